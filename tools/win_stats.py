"""Window hit statistics of the K6/K7 LDS-window kernel, per iteration (library built with -DAPD_LAB_WIN_STATS).
Usage: python tools/win_stats.py [--hard] [W H N iters]      --hard: the synth.HARD scene (occlusions, gain, lost overlap)"""
import os as _os
_os.environ.setdefault("APD_ALLOW_STALE_LIBRARY", "1")   # a lab build (-DAPD_LAB_WIN_STATS): not the digest of the tree's flags
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
from apd_mvs_amd import synth
hard = "--hard" in sys.argv
if hard:
    sys.argv.remove("--hard")
W, H, N, iters = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (2048, 1536, 8, 4)))
sc = synth.make_scene(W, H, N, seed=0, device="cuda", **(synth.HARD if hard else {}))
cams = [pkg.make_camera(sc.K[i], sc.R[i], sc.t[i], W, H, sc.depth_min, sc.depth_max) for i in range(N + 1)]
p = pkg.default_params(num_images=N + 1, depth_min=0.6 * sc.depth_min, depth_max=1.2 * sc.depth_max, use_APD=0, state=pkg.FIRST_INIT,
                       max_iterations=iters, seed=12345)
h = pkg.Handle(W, H, p, device=0)
h.upload_views(cams, sc.images)
L = pkg.lib()
out = (C.c_ulonglong * 8)()
for k in (pkg.K1, pkg.K2, pkg.K5):
    h.run_kernel(k)
L.apd_debug_win_stats(out, 1)
for it in range(iters):
    h.run_sweeps(it, 1)
    L.apd_debug_win_stats(out, 1)
    s = list(out)
    lane = max(s[0] + s[1] + s[2], 1)
    print("iter %d: lane-NCCs %.3g  window %.1f%%  global fast %.1f%%  global slow %.2f%% | wave-NCCs %.3g, mixed %.1f%% | windows staged %d"
          % (it, lane, 100.0 * s[0] / lane, 100.0 * s[1] / lane, 100.0 * s[2] / lane, s[3], 100.0 * s[4] / max(s[3], 1), s[5]))
    # lane utilisation: NCCs the lanes need against 64 x the NCCs the waves execute; phase A (9 hypotheses x N views) runs with full
    # waves, so what is left is the refinement (5 hypotheses x selected views, early-outs per lane, one wave-level NCC if ANY lane needs it)
    pixels = W * H  # two launches per iteration, each over half the pixels
    per_px, per_wave = lane / pixels, s[3] * 64.0 / pixels
    print("        per pixel: %.1f NCCs needed, %.1f executed by its wave (utilisation %.3f); beyond the 9 x %d of phase A: %.1f needed, %.1f executed (%.3f)"
          % (per_px, per_wave, per_px / per_wave, N, per_px - 9 * N, per_wave - 9 * N, (per_px - 9 * N) / max(per_wave - 9 * N, 1e-9)))

# post-loop kernels of the same pass: K11..K13, then K14 and K15 with their own counters
for k in (pkg.K11, pkg.K12, pkg.K13):
    h.run_kernel(k)
L.apd_debug_win_stats_k1415(out, 1)
for name, k in (("K14", pkg.K14), ("K15", pkg.K15)):
    h.run_kernel(k)
    L.apd_debug_win_stats_k1415(out, 1)
    s = list(out)
    lane = max(s[0] + s[1] + s[2], 1)
    print("%s: lane-NCCs %.3g  window %.1f%%  global fast %.1f%%  global slow %.2f%% | wave-NCCs %.3g, mixed %.1f%% | windows staged %d"
          % (name, lane, 100.0 * s[0] / lane, 100.0 * s[1] / lane, 100.0 * s[2] / lane, s[3], 100.0 * s[4] / max(s[3], 1), s[5]))
