"""How many of a strong pixel's nine propagation hypotheses (eight checkerboard arms + its own plane, APD.cu:1012-1209)
are the same plane bit for bit?  A pixel that adopts a neighbour's plane copies the float4 (APD.cu:1300), so converged
regions share planes exactly and K6/K7 scores the same (plane, view) several times.

Measured from the state the kernel itself sees: before the black launch of iteration i for black pixels, between the black and
the red launch for red pixels.  The arm search is restated with torch on the GPU (strict `<`, first minimum wins, exactly
arm_pos() of csrc/apd_kernels_k67w.hip); nothing in the library is instrumented.

Prints per (iteration, colour): the mean number of DISTINCT hypotheses per pixel, the histogram, and the mean over wave
footprints (32 x 4 px, the 64 same-colour pixels one wave64 owns) of the per-wave MAXIMUM -- the trip count a wave would
need if every lane looped over its own distinct hypotheses only.

Usage: python tools/dup_stats.py [W H N iters [apd]]"""
import os as _os
_os.environ.setdefault("APD_ALLOW_STALE_LIBRARY", "1")   # a lab build (-DAPD_LAB_WIN_STATS): not the digest of the tree's flags
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import __graft_entry__ as ge

pkg = ge.load_package()
from apd_mvs_amd import synth

args = sys.argv[1:]
W, H, N, iters = (int(v) for v in (args[:4] if len(args) >= 4 else (4096, 3072, 8, 5)))
dev = torch.device("cuda", 0)


def shifted(costs_pad, pad, dx, dy):
    """costs at (x + dx, y + dy) for every pixel; +inf outside the image."""
    Hh, Ww = costs_pad.shape[0] - 2 * pad, costs_pad.shape[1] - 2 * pad
    return costs_pad[pad + dy:pad + dy + Hh, pad + dx:pad + dx + Ww]


def arm_candidates(arm):
    d = arm >> 1
    dx = -1 if d == 2 else (1 if d == 3 else 0)
    dy = -1 if d == 0 else (1 if d == 1 else 0)
    if arm & 1:  # far
        return [((3 + 2 * i) * dx, (3 + 2 * i) * dy) for i in range(11)]
    ex, ey = (1 if dy != 0 else 0), (1 if dx != 0 else 0)
    out = [(dx, dy)]
    for i in range(3):
        for sgn in (-1, 1):
            out.append(((2 + i) * dx + sgn * (1 + i) * ex, (2 + i) * dy + sgn * (1 + i) * ey))
    return out


def distinct_counts(costs, planes_bits):
    """costs (H, W) float32, planes_bits (H, W, 4) int32 -> (H, W) int8: distinct planes among valid arms + own."""
    Hh, Ww = costs.shape
    pad = 24
    inf = torch.full((Hh + 2 * pad, Ww + 2 * pad), float("inf"), device=costs.device)
    inf[pad:pad + Hh, pad:pad + Ww] = torch.nan_to_num(costs, nan=float("inf"))
    inside = torch.zeros_like(inf, dtype=torch.bool)
    inside[pad:pad + Hh, pad:pad + Ww] = True
    ys, xs = torch.meshgrid(torch.arange(Hh, device=costs.device), torch.arange(Ww, device=costs.device), indexing="ij")
    hyps, valids = [], []
    for arm in range(8):
        cand = arm_candidates(arm)
        stack = torch.stack([shifted(inf, pad, dx, dy) for dx, dy in cand])
        valid = shifted(inside, pad, cand[0][0], cand[0][1])
        k = torch.argmin(stack, dim=0)  # strict '<' in the kernel == first minimum; ties in float costs are rare
        offs = torch.tensor(cand, device=costs.device)
        qx = (xs + offs[k, 0]).clamp(0, Ww - 1)
        qy = (ys + offs[k, 1]).clamp(0, Hh - 1)
        hyps.append(planes_bits[qy, qx])
        valids.append(valid)
        del stack
    hyps.append(planes_bits)
    valids.append(torch.ones_like(valids[0]))
    distinct = torch.zeros((Hh, Ww), dtype=torch.int8, device=costs.device)
    for i in range(9):
        dup = torch.zeros_like(valids[0])
        for j in range(i):
            dup |= valids[j] & (hyps[i] == hyps[j]).all(dim=-1)
        distinct += (valids[i] & ~dup).to(torch.int8)
    nvalid = sum(v.to(torch.int8) for v in valids)
    return distinct, nvalid


def report(tag, distinct, nvalid, colour, strong):
    Hh, Ww = distinct.shape
    ys, xs = torch.meshgrid(torch.arange(Hh, device=dev), torch.arange(Ww, device=dev), indexing="ij")
    mask = (((xs + ys) & 1) == colour) & strong
    d = distinct[mask].float()
    nv = nvalid[mask].float()
    hist = torch.bincount(distinct[mask].long(), minlength=10)[:10].float()
    hist = hist / hist.sum()
    # wave footprints: 32 x 4 px
    Hc, Wc = (Hh // 4) * 4, (Ww // 32) * 32
    dm = torch.where(mask, distinct, torch.zeros_like(distinct))[:Hc, :Wc].reshape(Hc // 4, 4, Wc // 32, 32)
    wave_max = dm.amax(dim=(1, 3)).float()
    print("%s colour %d: valid hypotheses %.3f, distinct %.3f per pixel (%.1f %% duplicates) | wave max distinct mean %.3f | hist[1..9] %s"
          % (tag, colour, nv.mean().item(), d.mean().item(), 100.0 * (1 - d.sum().item() / nv.sum().item()), wave_max.mean().item(),
             " ".join("%.3f" % v for v in hist[1:].tolist())), flush=True)


def state_tensors(h):
    planes, weak, _ = h.download()
    costs = h.state(pkg.STATE_COSTS)
    pb = torch.from_numpy(np.ascontiguousarray(planes).view(np.int32)).to(dev)
    return torch.from_numpy(costs).to(dev), pb, torch.from_numpy(weak).to(dev) != pkg.WEAK


sc = synth.make_scene(W, H, N, seed=0, device=dev)
cams = [pkg.make_camera(sc.K[i], sc.R[i], sc.t[i], W, H, sc.depth_min, sc.depth_max) for i in range(N + 1)]
p = pkg.default_params(num_images=N + 1, depth_min=0.6 * sc.depth_min, depth_max=1.2 * sc.depth_max, use_APD=0, state=pkg.FIRST_INIT,
                       max_iterations=iters, seed=12345)
h = pkg.Handle(W, H, p, device=0)
h.upload_views(cams, sc.images)
del sc.images[:]
for k in (pkg.K1, pkg.K2, pkg.K5):
    h.run_kernel(k)
print("dup_stats: %dx%d, %d sources, FIRST_INIT pass" % (W, H, N))
for it in range(iters):
    for colour, kid in ((0, pkg.K6), (1, pkg.K7)):
        costs, pb, strong = state_tensors(h)
        distinct, nvalid = distinct_counts(costs, pb)
        report("iter %d" % it, distinct, nvalid, colour, strong)
        del costs, pb, distinct, nvalid
        h.run_kernel(kid, it)
    h.run_kernel(pkg.K8, it)
h.close()
