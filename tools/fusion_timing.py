"""Device fusion (apd_fuse_views) against the reference's sequential host loop on a synthetic ring: time and byte equality.
Usage: python tools/fusion_timing.py [W H views sources]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from apd_mvs_amd import pipeline, synth
import test_gpu_dropin_binary as T

W, H, V, S = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (1920, 1080, 12, 8)))
scene, results = T._fusion_inputs(synth, pipeline, pkg, W, H, V, S, 0.0005, seed=5)
out = "/tmp/fusion_timing"
os.makedirs(out, exist_ok=True)
t = {}
for mode in ("gpu", "cpu"):
    if mode == "cpu":
        os.environ["APD_FUSION"] = "cpu"
    else:
        os.environ.pop("APD_FUSION", None)
    pipeline.fuse(scene, results, os.path.join(out, mode + "_warm.ply")) if mode == "gpu" else None
    t0 = time.time()
    n = pipeline.fuse(scene, results, os.path.join(out, mode + ".ply"))
    t[mode] = time.time() - t0
    print("%s fusion: %d points from %d views of %dx%d with %d sources each in %.2f s" % (mode, n, V, W, H, S, t[mode]), flush=True)
same = open(os.path.join(out, "gpu.ply"), "rb").read() == open(os.path.join(out, "cpu.ply"), "rb").read()
print("identical files:", same, " speed-up %.1fx" % (t["cpu"] / t["gpu"]))
