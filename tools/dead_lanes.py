#!/usr/bin/env python3
"""Dead lanes of K6/K7 (and K14/K15) in an APD pass: the strong sweep skips WEAK pixels but keeps their lanes.

Runs the untimed prelude of bench.py's APD workload on the GPU (a FIRST_INIT pass with weak_peak_radius 6 + ProcessProblem's
post-processing), takes the weak map the REFINE_INIT + APD pass starts from and counts, for every wave of K6/K7 -- the 64
same-colour pixels of a 32 x 4 footprint, four waves per 32 x 16 tile -- how many lanes have a pixel to update (not WEAK):

  * waves with no live lane leave at once (the kernel's ballot): free;
  * partly filled waves run every NCC for the lanes they have: the dead share of their lanes is what the launch wastes;
  * what re-packing the live pixels of a TILE into fewer waves (same LDS reference tile) could win: ceil(live / 64) waves per
    tile and colour instead of the number of non-empty waves.

usage: python tools/dead_lanes.py [W H N textureless]      (default 6200 4130 10 0.2 = configs[2])"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import __graft_entry__ as ge

pkg = ge.load_package()
from apd_mvs_amd import synth

W, H, N = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (6200, 4130, 10)))
tex = float(sys.argv[4]) if len(sys.argv) > 4 else 0.2
dev = torch.device("cuda", 0)
sc = synth.make_scene(W, H, N, seed=0, device=dev, textureless=tex)
cams = [pkg.make_camera(sc.K[i], sc.R[i], sc.t[i], W, H, sc.depth_min, sc.depth_max) for i in range(N + 1)]
dmin, dmax = 0.6 * sc.depth_min, 1.2 * sc.depth_max
p0 = pkg.default_params(num_images=N + 1, depth_min=dmin, depth_max=dmax, use_APD=0, state=pkg.FIRST_INIT, max_iterations=3,
                        weak_peak_radius=6, seed=12345)
h0 = pkg.Handle(W, H, p0, device=0)
h0.upload_views(cams, sc.images)
h0.run()
planes, weak, views = h0.download()
h0.close()
bad = (planes[..., 3] < np.float32(dmin)) | (planes[..., 3] > np.float32(dmax))
weak[bad] = pkg.UNKNOWN
live = weak != pkg.WEAK          # K6/K7 update STRONG and UNKNOWN pixels (APD.cu:1547-1585 skip WEAK only)
print("%d x %d, %d sources: WEAK %.2f %%, UNKNOWN %.2f %%" % (W, H, N, 100.0 * (weak == pkg.WEAK).mean(), 100.0 * (weak == pkg.UNKNOWN).mean()))

TW, TH, WH = 32, 16, 4           # tile, wave footprint rows (csrc/apd_sweep.h)
Hp, Wp = -(-H // TH) * TH, -(-W // TW) * TW
pad = np.zeros((Hp, Wp), bool)
pad[:H, :W] = live
inimg = np.zeros((Hp, Wp), bool)
inimg[:H, :W] = True
ys, xs = np.mgrid[0:Hp, 0:Wp]
tot_waves = tot_nonempty = tot_live = tot_packed = 0
hist = np.zeros(65, np.int64)
for colour in (0, 1):
    m = ((xs + ys) & 1) == colour
    lv = (pad & m).reshape(Hp // WH, WH, Wp // TW, TW).sum(axis=(1, 3))          # live lanes per wave
    px = (inimg & m).reshape(Hp // WH, WH, Wp // TW, TW).sum(axis=(1, 3))        # lanes with a pixel at all
    waves = px > 0
    hist += np.bincount(lv[waves].ravel(), minlength=65)
    tot_waves += int(waves.sum())
    tot_nonempty += int((lv > 0).sum())
    tot_live += int(lv.sum())
    per_tile = lv.reshape(Hp // TH, TH // WH, Wp // TW).sum(axis=1)              # live lanes per tile and colour
    tot_packed += int(np.ceil(per_tile / 64.0).sum())
print("K6/K7 waves (both colours): %d, non-empty %d (%.1f %%), live lanes %.1f %% of all lanes, %.1f %% of the lanes of non-empty waves"
      % (tot_waves, tot_nonempty, 100.0 * tot_nonempty / tot_waves, 100.0 * tot_live / (64.0 * tot_waves), 100.0 * tot_live / (64.0 * tot_nonempty)))
print("  wave-level work today = non-empty waves; if the live pixels of a tile were packed into ceil(live / 64) waves: %d waves (%.1f %% of today's)"
      % (tot_packed, 100.0 * tot_packed / tot_nonempty))
edges = [0, 1, 16, 32, 48, 60, 64, 65]
names = ["0", "1-15", "16-31", "32-47", "48-59", "60-63", "64"]
print("  live lanes per wave: " + ", ".join("%s: %.1f %%" % (names[i], 100.0 * hist[edges[i]:edges[i + 1]].sum() / hist.sum()) for i in range(7)))

# K14 / K15: one lane per pixel, 8 x 8 footprints; a lane is dead when the pixel has no estimate (depth 0 -> UNKNOWN at once)
# or lies in the 6-px margin (K14 only); WEAK pixels are scored like STRONG ones
has = planes[..., 3] != 0
ins = np.zeros((H, W), bool)
ins[6:H - 6, 6:W - 6] = True
for name, lv in (("K14", has & ins), ("K15", has)):
    Hq, Wq = -(-H // 8) * 8, -(-W // 8) * 8
    padq = np.zeros((Hq, Wq), bool)
    padq[:H, :W] = lv
    per = padq.reshape(Hq // 8, 8, Wq // 8, 8).sum(axis=(1, 3))
    ne = per > 0
    print("%s: live lanes %.1f %% of the lanes of its %d non-empty waves" % (name, 100.0 * per.sum() / (64.0 * ne.sum()), int(ne.sum())))
