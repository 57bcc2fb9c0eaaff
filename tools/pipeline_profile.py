import os, sys, time
ROOT = "/root/repo"
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
from apd_mvs_amd import pipeline, synth
T = {}
def timed(name, f):
    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(*a, **k); torch.cuda.synchronize()
        T[name] = T.get(name, 0.0) + time.perf_counter() - t0
        return r
    return g
H = pkg.Handle
H.__init__ = timed("Handle()", H.__init__)
H.upload_views = timed("upload_views", H.upload_views)
H.upload_prior = timed("upload_prior", H.upload_prior)
H.run = timed("run", H.run)
H.download = timed("download", H.download)
H.close = timed("close", H.close)
pipeline.rescale_nearest = timed("rescale_nearest", pipeline.rescale_nearest)
H.download_device = timed("download_device", H.download_device)
H.reset = timed("reset", H.reset)
pipeline.level_inputs = timed("level_inputs", pipeline.level_inputs)
scene = pipeline.synthetic_ring(synth, 1920, 1080, 6, 5, pkg.make_camera, seed=0, textureless=0.2)
t0 = time.perf_counter()
res = pipeline.run_pipeline(scene, pipeline.HipBackend(pkg, device=0))
tot = time.perf_counter() - t0
print("total %.2f s for 6 views" % tot)
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    print("  %-18s %.2f s" % (k, v))
print("  other              %.2f s" % (tot - sum(v for k, v in T.items() if k not in ("rescale_nearest",)) ))
