"""How well do the sub-patch anchors of a K9/K10 wave cluster?  ("the share of pairs whose anchors fall inside a window is
the number to print first".)

HIP path at a BASELINE size (default configs[2]: 6200 x 4130, 10 sources, 20 % textureless): FIRST_INIT pass, REFINE_INIT + APD handle, K1..K5, then
the WEAK lists' order is emulated on the host (16 x 8 px tiles of one colour in supertile order, apd_kernels_weak.hip) and for every wave of 64
list entries its 64 x 8 (pixel, slot) anchors are grouped in several ways; for each grouping the share of pairs whose anchor lies inside a
64-column x R-row window centred on the group's bounding box is printed (margin: the sub-patch's +-5 px and one texel for the bilinear tap, in
REFERENCE-image pixels: the homographies of a converged region are close to a translation, so the source-image figure is the same to first order).

Groupings:  slot      -- slot k of every lane (today's loop order; slots are sorted by plane-fit weight per pixel, APD.cu:1974-1980)
            angle     -- the g-th anchor of every lane by angle around its pixel
            sorted-y  -- all (lane, slot) pairs of the wave sorted by anchor row-band then column, cut into groups of 64
Usage: python tools/nb_cluster.py [W H N]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge

pkg = ge.load_package()
from apd_mvs_amd import synth
import common


def list_order(weak, colour, super_shift=4):
    """Pixel coordinates of the WEAK pixels of one colour in the order of the K9/K10 lists: 16 x 8 px tiles, raster order inside supertiles of
    2^shift x 2^shift tiles, supertiles in raster order; inside a tile raster order."""
    H, W = weak.shape
    ys, xs = np.nonzero(weak == 0)
    keep = ((xs + ys) & 1) == colour
    xs, ys = xs[keep], ys[keep]
    tx, ty = xs // 16, ys // 8
    sx, sy = tx >> super_shift, ty >> super_shift
    supers_x = ((W + 15) // 16 + (1 << super_shift) - 1) >> super_shift
    key = ((((sy * supers_x + sx) << (2 * super_shift)) + ((ty & ((1 << super_shift) - 1)) << super_shift) + (tx & ((1 << super_shift) - 1))).astype(np.int64) << 8) + (
        (ys & 7) * 16 + (xs & 15))
    o = np.argsort(key, kind="stable")
    return xs[o], ys[o]


def window_share(ax, ay, valid, rows, margin=7):
    """ax, ay, valid: [waves, groups, 64].  Share of valid pairs inside a 64 x rows window centred on the group's bounding box."""
    big = 1 << 20
    x_lo = np.where(valid, ax, big).min(-1, keepdims=True)
    x_hi = np.where(valid, ax, -big).max(-1, keepdims=True)
    y_lo = np.where(valid, ay, big).min(-1, keepdims=True)
    y_hi = np.where(valid, ay, -big).max(-1, keepdims=True)
    cx, cy = (x_lo + x_hi) // 2, (y_lo + y_hi) // 2
    inside = valid & (np.abs(ax - cx) <= 32 - margin) & (np.abs(ay - cy) <= rows // 2 - margin)
    return inside.sum() / max(valid.sum(), 1), float(np.median((x_hi - x_lo)[valid.any(-1, keepdims=True)])), float(np.median((y_hi - y_lo)[valid.any(-1, keepdims=True)]))


def best_window_share(ax, ay, valid, rows, margin=7):
    """Upper bound for ONE window per group: the densest 64 x rows placement (by a coarse grid search over the group's pairs as centres)."""
    hw, hh = 32 - margin, rows // 2 - margin
    # candidate centres: every pair's anchor; count pairs within the box: O(64^2) per group -- sample groups to bound the cost
    G = ax.shape[0] * ax.shape[1]
    a = ax.reshape(G, 64)
    b = ay.reshape(G, 64)
    v = valid.reshape(G, 64)
    pick = np.random.RandomState(0).choice(G, min(G, 20000), replace=False)
    a, b, v = a[pick], b[pick], v[pick]
    dx = np.abs(a[:, :, None] - a[:, None, :]) <= hw
    dy = np.abs(b[:, :, None] - b[:, None, :]) <= hh
    cnt = (dx & dy & v[:, None, :] & v[:, :, None]).sum(-1).max(-1)
    return cnt.sum() / max(v.sum(), 1)


def main():
    W, H, N = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (6200, 4130, 10)
    hard = "--hard" in sys.argv
    import torch
    sc = synth.make_scene(W, H, N, seed=0, device=torch.device("cuda", 0), textureless=0.2, **(synth.HARD if hard else {}))
    imgs = sc.images_numpy()
    del sc.images[:]
    p0 = common.base_params(sc, N, max_iterations=3, seed=12345, weak_peak_radius=6)
    h0 = common.make_handle(pkg, sc, imgs, N, p0)
    h0.run()
    planes, weak, views = h0.download()
    h0.close()
    prior = common.postprocess(planes, weak, views, p0["depth_min"], p0["depth_max"])
    p = common.base_params(sc, N, max_iterations=3, seed=12346, state=1, use_APD=1, weak_peak_radius=6, rotate_time=4, ransac_threshold=0.01 - 0.00125 * 3)
    h = common.make_handle(pkg, sc, imgs, N, p, prior=prior)
    for k in (1, 2, 3, 4, 5):
        h.run_kernel(k)
    wi = h.state(pkg.STATE_WEAK_INFO)
    nmap = h.state(pkg.STATE_NEIGHBOURS_MAP)
    nb = h.state(pkg.STATE_NEIGHBOURS).astype(np.int32)   # [weak, 9, 2]
    h.close()
    print("%dx%d N=%d%s: %.1f %% WEAK after K4" % (W, H, N, " hard" if hard else "", 100.0 * (wi == 0).mean()))
    for colour in (0,):
        xs, ys = list_order(wi, colour)
        nw = len(xs) // 64
        xs, ys = xs[:nw * 64], ys[:nw * 64]
        q = nb[nmap[ys, xs]]                      # [n, 9, 2]
        ax, ay = q[:, 1:, 0].reshape(nw, 64, 8), q[:, 1:, 1].reshape(nw, 64, 8)
        valid = ax >= 0
        px, py = xs.reshape(nw, 64, 1), ys.reshape(nw, 64, 1)
        dist = np.hypot(ax - px, ay - py)[valid]
        print("colour %d: %d waves, %.2f valid anchors per pixel, anchor distance median %.0f px, p90 %.0f, max %.0f" % (
            colour, nw, valid.sum() / (nw * 64), np.median(dist), np.percentile(dist, 90), dist.max()))
        # grouping "slot": [waves, 8 slots, 64 lanes]
        groupings = {"slot": (ax.transpose(0, 2, 1), ay.transpose(0, 2, 1), valid.transpose(0, 2, 1))}
        # grouping "angle": per lane sort the 8 slots by angle (invalid last)
        ang = np.where(valid, np.arctan2(ay - py, ax - px), 10.0)
        o = np.argsort(ang, axis=-1, kind="stable")
        gx, gy, gv = np.take_along_axis(ax, o, -1), np.take_along_axis(ay, o, -1), np.take_along_axis(valid, o, -1)
        groupings["angle"] = (gx.transpose(0, 2, 1), gy.transpose(0, 2, 1), gv.transpose(0, 2, 1))
        # grouping "sorted": all 512 pairs of a wave sorted by (row band of 16 px, column), groups of 64
        fx, fy, fv = ax.reshape(nw, 512), ay.reshape(nw, 512), valid.reshape(nw, 512)
        for band in (12, 16, 24):
            key = np.where(fv, (fy // band).astype(np.int64) * 65536 + fx, np.int64(1) << 40)
            o = np.argsort(key, axis=-1, kind="stable")
            groupings["sorted-y%d" % band] = tuple(np.take_along_axis(a, o, -1).reshape(nw, 8, 64) for a in (fx, fy, fv))
        for name, (a, b, v) in groupings.items():
            line = "  %-10s" % name
            for rows in (29, 40, 56):
                share, mx, my = window_share(a, b, v, rows)
                line += " | R=%d: bbox-centred %.3f, best placement %.3f" % (rows, share, best_window_share(a, b, v, rows))
            print(line + " | group bbox median %.0f x %.0f px" % (mx, my))
        # ONE byte window per workgroup of `nwg` consecutive waves and view, centred on the bounding box of the workgroup's own pixels
        for nwg in (1, 2, 4):
            g = nw // nwg
            wx, wy = xs[:g * nwg * 64].reshape(g, nwg * 64), ys[:g * nwg * 64].reshape(g, nwg * 64)
            cxx, cyy = (wx.min(-1) + wx.max(-1)) // 2, (wy.min(-1) + wy.max(-1)) // 2
            span_x, span_y = wx.max(-1) - wx.min(-1), wy.max(-1) - wy.min(-1)
            a = ax[:g * nwg].reshape(g, nwg * 64 * 8)
            b = ay[:g * nwg].reshape(g, nwg * 64 * 8)
            v = valid[:g * nwg].reshape(g, nwg * 64 * 8)
            line = "  workgroup of %d wave(s): pixel bbox median %d x %d (p90 %d x %d) |" % (nwg, np.median(span_x), np.median(span_y), np.percentile(span_x, 90), np.percentile(span_y, 90))
            for (cw, ch) in ((128, 96), (128, 128), (160, 128), (192, 128), (192, 160), (256, 128)):
                inside = v & (np.abs(a - cxx[:, None]) <= cw // 2 - 7) & (np.abs(b - cyy[:, None]) <= ch // 2 - 7)
                allin = (inside | ~v).reshape(g * nwg, 512).all(-1).mean()
                line += " %dx%d (%.0f KB): %.3f of the anchors, %.3f of the waves complete |" % (cw, ch, cw * ch / 1024.0, inside.sum() / v.sum(), allin)
            print(line)
        # The lane = (pixel, hypothesis) mapping of round 6 (k910_update_weak, APD_K910_REMAP): a wave-level sub-patch tap is issued for the eight
        # pixels of a group x their eight hypotheses at ONE slot index.  Lanes with the same (anchor of the slot, anchor of the hypothesis) pair
        # read the same addresses (one L1 tag access for all of them).  How many distinct pairs per 64 lanes -- in K3's slot order (sorted by
        # plane-fit weight per pixel) and with every pixel walking its anchors in the order of their angle around it?
        code = np.where(valid, ay.astype(np.int64) * 65536 + ax, -1 - np.arange(8)[None, None, :])     # [waves, 64, 8] anchor id (invalid: unique)
        ang = np.where(valid, np.arctan2(ay - py, ax - px), 10.0)
        order = np.argsort(ang, axis=-1, kind="stable")
        for name, c in (("K3 slot order", code), ("angle order", np.take_along_axis(code, order, -1))):
            g = c.reshape(nw, 8, 8, 8)                      # [wave, group, pixel in group, slot]
            pick = np.random.RandomState(2).choice(nw, min(nw, 3000), replace=False)
            g = g[pick]
            pairs = g[:, :, :, :, None] * (1 << 40) + g[:, :, :, None, :]     # [wave, group, pixel, slot k, hyp h]
            # one wave-level tap = fixed (group, k): 8 pixels x 8 hyps = 64 lanes
            lanes = pairs.transpose(0, 1, 3, 2, 4).reshape(len(pick) * 8 * 8, 64)
            distinct = np.array([len(np.unique(r)) for r in lanes[np.random.RandomState(3).choice(len(lanes), 20000, replace=False)]])
            places = g.transpose(0, 1, 3, 2).reshape(len(pick) * 8 * 8, 8)
            dplaces = np.array([len(np.unique(r)) for r in places[np.random.RandomState(4).choice(len(places), 20000, replace=False)]])
            print("  lane = (pixel, hypothesis), %-13s: distinct (anchor, plane) pairs per wave-level tap %.1f of 64; distinct anchors per tap %.2f of 8" % (
                name, distinct.mean(), dplaces.mean()))
        if "--global-pairs" in sys.argv:
            # how many DISTINCT (anchor of the slot, anchor of the hypothesis) pairs does a whole launch hold?  c(anchor, plane of anchor', view) does not
            # depend on the WEAK pixel that asks for it (the two-kernel form of the propagation phase)
            ids = np.where(valid, ay.astype(np.int64) * 65536 + ax, -1).reshape(-1, 8)          # [pixels, 8]
            strong = valid.reshape(-1, 8)
            pair = ids[:, :, None] * (1 << 32) + ids[:, None, :]                                 # [pixels, slot k, hyp h]
            ok = strong[:, :, None] & strong[:, None, :]
            flat = pair[ok]
            uniq = np.unique(flat)
            print("  whole launch (colour %d): %d (slot, hypothesis) pairs over %d WEAK pixels, %d distinct (anchor, anchor') pairs = %.3f; distinct anchors %d (%.1f pairs each)" % (
                colour, flat.size, ids.shape[0], uniq.size, uniq.size / flat.size, np.unique(ids[strong]).size, uniq.size / max(np.unique(ids[strong]).size, 1)))
            per_anchor = np.unique(uniq >> 32, return_counts=True)[1]
            print("    distinct partners per anchor: median %d, p90 %d, p99 %d, p99.9 %d, max %d; share of pairs whose anchor has > 128 partners %.4f, > 256: %.4f" % (
                np.median(per_anchor), np.percentile(per_anchor, 90), np.percentile(per_anchor, 99), np.percentile(per_anchor, 99.9), per_anchor.max(),
                per_anchor[per_anchor > 128].sum() / per_anchor.sum(), per_anchor[per_anchor > 256].sum() / per_anchor.sum()))
            for tile_waves in (1, 4, 16, 64, 256):
                n = (ids.shape[0] // (64 * tile_waves)) * 64 * tile_waves
                blk = np.where(ok[:n], pair[:n], -1).reshape(-1, 64 * tile_waves * 64)
                pick = np.random.RandomState(5).choice(blk.shape[0], min(blk.shape[0], 300), replace=False)
                ratio = np.mean([(np.unique(b[b >= 0]).size) / max((b >= 0).sum(), 1) for b in blk[pick]])
                print("    inside %4d consecutive waves of the list: distinct / all = %.3f" % (tile_waves, ratio))
        # distinct anchors per wave: how much do lanes share?
        code = np.where(valid, ay * 65536 + ax, -1).reshape(nw, 512)
        distinct = np.array([len(np.unique(c[c >= 0])) for c in code[np.random.RandomState(1).choice(nw, min(nw, 4000), replace=False)]])
        print("  distinct anchors per wave (of <= 512 pairs): median %d, p10 %d, p90 %d" % (np.median(distinct), np.percentile(distinct, 10), np.percentile(distinct, 90)))


if __name__ == "__main__":
    t0 = time.time()
    main()
    print("%.0f s" % (time.time() - t0))
