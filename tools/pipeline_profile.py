"""Host/kernels split of the in-memory pipeline.  Usage: python tools/pipeline_profile.py [W H views src]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
KT = {}
_run = H.run
def run_profiled(self, *a, **k):
    self.profile_enable(True); self.profile_reset()
    r = _run(self, *a, **k)
    for kid, (ms, n) in self.profile().items():
        KT[kid] = KT.get(kid, 0.0) + ms
    return r
H.run = timed("run", run_profiled)
H.download = timed("download", H.download)
H.close = timed("close", H.close)
pipeline.rescale_nearest = timed("rescale_nearest", pipeline.rescale_nearest)
H.download_device = timed("download_device", H.download_device)
H.reset = timed("reset", H.reset)
pipeline.level_inputs = timed("level_inputs", pipeline.level_inputs)
Wd, Ht, NV, NS = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (1920, 1080, 6, 5)
t0 = time.perf_counter()
scene = pipeline.synthetic_ring(synth, Wd, Ht, NV, NS, pkg.make_camera, seed=0, textureless=0.2)
print("scene %.2f s" % (time.perf_counter() - t0))
t0 = time.perf_counter()
res = pipeline.run_pipeline(scene, pipeline.HipBackend(pkg, device=0))
tot = time.perf_counter() - t0
print("total %.2f s for %d views of %dx%d" % (tot, NV, Wd, Ht))
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    print("  %-18s %.2f s" % (k, v))
print("  other              %.2f s" % (tot - sum(v for k, v in T.items() if k not in ("rescale_nearest",)) ))
print("kernel time by kernel (all passes, all levels):")
for kid, ms in sorted(KT.items(), key=lambda kv: -kv[1]):
    print("  K%-2d %-26s %8.1f ms" % (kid, pkg.KERNEL_NAMES[kid], ms))
