"""Lane-level against wave-level work of the weak sweep K9/K10 (library built with -DAPD_LAB_WIN_STATS).
Usage: python tools/weak_stats.py [W H N iters]"""
import os as _os
_os.environ.setdefault("APD_ALLOW_STALE_LIBRARY", "1")   # a lab build (-DAPD_LAB_WIN_STATS): not the digest of the tree's flags
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
from apd_mvs_amd import synth
import common
W, H, N, iters = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (2048, 1536, 10, 3)))
sc = synth.make_scene(W, H, N, seed=0, device="cuda", textureless=0.2)
cams = [pkg.make_camera(sc.K[i], sc.R[i], sc.t[i], W, H, sc.depth_min, sc.depth_max) for i in range(N + 1)]
dmin, dmax = 0.6 * sc.depth_min, 1.2 * sc.depth_max
p0 = pkg.default_params(num_images=N + 1, depth_min=dmin, depth_max=dmax, use_APD=0, state=pkg.FIRST_INIT, max_iterations=3, weak_peak_radius=6, seed=12345)
h0 = pkg.Handle(W, H, p0, device=0)
h0.upload_views(cams, sc.images)
h0.run()
planes, weak, views = h0.download()
h0.close()
prior = common.postprocess(planes, weak, views, np.float32(dmin), np.float32(dmax))
p = pkg.default_params(num_images=N + 1, depth_min=dmin, depth_max=dmax, use_APD=1, state=pkg.REFINE_INIT, max_iterations=iters,
                       weak_peak_radius=6, rotate_time=4, ransac_threshold=0.01 - 0.00125 * 3, seed=12346)
h = pkg.Handle(W, H, p, device=0)
h.upload_views(cams, sc.images)
h.upload_prior(*prior)
for k in (pkg.K1, pkg.K2, pkg.K3, pkg.K4, pkg.K5):
    h.run_kernel(k)
L = pkg.lib()
out = (C.c_ulonglong * 8)()
win = (C.c_ulonglong * 8)()
L.apd_debug_weak_stats(out, 1)
L.apd_debug_win_stats_weak(win, 1)
weak_px = h.weak_count
for it in range(iters):
    h.run_sweeps(it, 1)
    L.apd_debug_weak_stats(out, 1)
    s = [float(v) for v in out]
    print("iter %d (%d WEAK pixels, %.1f %%): per weak pixel: propagation %.1f NCCNew needed, %.1f executed by its wave (%.3f); "
          "hypotheses 9..14: %.1f needed, %.1f executed (%.3f); sub-patches %.1f needed, %.1f executed (%.3f)"
          % (it, weak_px, 100.0 * weak_px / (W * H), s[0] / weak_px, s[1] * 64 / weak_px, s[0] / max(s[1] * 64, 1), s[2] / weak_px,
             s[3] * 64 / weak_px, s[2] / max(s[3] * 64, 1), s[4] / weak_px, s[5] * 64 / weak_px, s[4] / max(s[5] * 64, 1)))
    if s[7] > 0:
        print("        sub-patch rows of three taps in which two neighbouring taps lie in one 16-byte row segment: %.1f %%" % (100.0 * s[6] / s[7]))
    L.apd_debug_win_stats_weak(win, 1)
    w = [float(v) for v in win]
    lanes = max(w[0] + w[1] + w[2], 1.0)
    print("        centre patches: %.1f %% of the lane-level NCCs through the window, %.1f %% global (wave-level: %.1f %% of the calls have lanes on "
          "both paths); windows staged %d" % (100.0 * w[0] / lanes, 100.0 * (w[1] + w[2]) / lanes, 100.0 * w[4] / max(w[3], 1.0), int(w[5])))
