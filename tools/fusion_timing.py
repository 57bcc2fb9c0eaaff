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
from oracle import binding as ob
pipeline.fuse(scene, results, os.path.join(out, "gpu_warm.ply"))
t0 = time.time()
n = pipeline.fuse(scene, results, os.path.join(out, "gpu.ply"))
t_gpu = time.time() - t0
print("device fusion: %d points from %d views of %dx%d with %d sources each in %.2f s" % (n, V, W, H, S, t_gpu), flush=True)
cams = (type(scene.cameras[0]) * V)(*scene.cameras)
t0 = time.time()
n = ob.fuse(cams, scene.images, [results[v].depth for v in range(V)], [results[v].normal for v in range(V)],
            [results[v].weak for v in range(V)], scene.pairs, os.path.join(out, "cpu.ply"))
t_cpu = time.time() - t0
print("sequential host loop (oracle): %d points in %.2f s" % (n, t_cpu), flush=True)
same = open(os.path.join(out, "gpu.ply"), "rb").read() == open(os.path.join(out, "cpu.ply"), "rb").read()
print("identical files:", same, " speed-up %.1fx" % (t_cpu / t_gpu))
