"""Per-kernel timing of a three-pass pipeline (FIRST_INIT -> REFINE_INIT+APD -> REFINE_ITER+APD+geom) at a given size.
Usage: python tools/pass_timing.py [W H N textureless [float]]      float: non-integer grey values, i.e. the float-image path that
every pyramid level but the finest takes (APD.cpp:474)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
from apd_mvs_amd import synth
import common

W, H, N = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (4096, 3072, 8)))
tl = float(sys.argv[4]) if len(sys.argv) > 4 else 0.2
sc = synth.make_scene(W, H, N, seed=3, textureless=tl, device="cuda")
if len(sys.argv) > 5 and sys.argv[5] == "float":
    sc.images = [im * 0.97 + 0.3 for im in sc.images]
cams = [pkg.make_camera(sc.K[i], sc.R[i], sc.t[i], W, H, sc.depth_min, sc.depth_max) for i in range(N + 1)]
dmin, dmax = 0.6 * sc.depth_min, 1.2 * sc.depth_max
passes = [dict(state=0, use_APD=0, weak_peak_radius=6),
          dict(state=1, use_APD=1, weak_peak_radius=6, rotate_time=4, ransac_threshold=0.01 - 0.00125 * 3),
          dict(state=2, use_APD=1, weak_peak_radius=4, rotate_time=4, ransac_threshold=0.01 - 0.00125 * 3, geom_consistency=1)]
prior = None
deps = None
for pi, extra in enumerate(passes):
    p = pkg.default_params(num_images=N + 1, depth_min=dmin, depth_max=dmax, max_iterations=3, seed=5 + pi, **extra)
    h = pkg.Handle(W, H, p, device=0)
    if extra.get("geom_consistency"):
        deps = [torch.from_numpy(prior[0][..., 3].copy()).cuda()] * (N + 1)   # stand-in depth maps
    h.upload_views(cams, sc.images, deps if extra.get("geom_consistency") else None)
    if prior is not None:
        h.upload_prior(*prior)
    h.profile_enable(True); h.profile_reset()
    t0 = time.time(); h.run(); t1 = time.time()
    prof = h.profile()
    tot = sum(v[0] for v in prof.values())
    print("== pass %d state=%d weak=%d (%.1f%%) wall %.0f ms, kernels %.0f ms" % (pi, extra["state"], h.weak_count, 100.0 * h.weak_count / (W * H), (t1 - t0) * 1e3, tot))
    for k, (ms, n) in sorted(prof.items()):
        print("   K%-2d %-24s %9.2f ms  x%d" % (k, pkg.KERNEL_NAMES[k], ms, n))
    planes, weak, views = h.download()
    prior = common.postprocess(planes, weak, views, np.float32(dmin), np.float32(dmax))
    print("   states:", np.bincount(prior[2].ravel(), minlength=3))
    h.close()
