"""Known answers for the oracle's three cost functions from an INDEPENDENT float64 restatement (numpy, matrix form) of what the
reference computes: ComputeBilateralNCCOld (APD.cu:530-614), ComputeBilateralNCCNew (:400-528) and
ComputeGeomConsistencyCost (:760-789).  The oracle evaluates them in binary32 in the reference's operation order; this file
evaluates the same definitions with 3x3 matrix algebra in double and compares on hundreds of random (pixel, view, plane)
triples.  Agreement to ~1e-3 says the oracle computes the right FUNCTION; the bit-level contract is then a matter of
rounding order, which the HIP-vs-oracle tests pin."""
import numpy as np
import pytest

import common


def _cams(sc, i):
    K = sc.K[i].astype(np.float64).reshape(3, 3)
    R = sc.R[i].astype(np.float64).reshape(3, 3)
    t = sc.t[i].astype(np.float64)
    return K, R, t, -R.T @ t


def _homography(sc, src, plane):
    """x_src ~ K_s (R_rel - t_rel n^T / d) K_r^-1 x_ref for the plane n.X + d = 0 in the reference camera frame."""
    Kr, Rr, tr, Cr = _cams(sc, 0)
    Ks, Rs, ts, Cs = _cams(sc, src)
    n, d = np.asarray(plane[:3], np.float64), float(plane[3])
    R_rel = Rs @ Rr.T
    t_rel = Rs @ (Cr - Cs)
    return Ks @ (R_rel - np.outer(t_rel, n) / d) @ np.linalg.inv(Kr)


def _warp(H, x, y):
    v = H @ np.array([x, y, 1.0])
    return v[0] / v[2], v[1] / v[2]


def _texel(img, x, y):
    h, w = img.shape
    return float(img[min(max(int(y), 0), h - 1), min(max(int(x), 0), w - 1)])


def _bilinear(img, sx, sy):
    """tex2D(linear filter, clamp) at (sx + 0.5, sy + 0.5): between texels floor(s) and floor(s) + 1."""
    fx, fy = np.floor(sx), np.floor(sy)
    a, b = sx - fx, sy - fy
    x0, y0 = int(fx), int(fy)
    top = _texel(img, x0, y0) * (1 - a) + _texel(img, x0 + 1, y0) * a
    bot = _texel(img, x0, y0 + 1) * (1 - a) + _texel(img, x0 + 1, y0 + 1) * a
    return top * (1 - b) + bot * b


def _patch_cost(ref, src, H, cx, cy, radius, step):
    r, s = [], []
    for i in range(-radius, radius + 1, step):
        for j in range(-radius, radius + 1, step):
            r.append(_texel(ref, cx + i, cy + j))
            s.append(_bilinear(src, *_warp(H, cx + i, cy + j)))
    r, s = np.array(r), np.array(s)
    var_r, var_s = (r * r).mean() - r.mean() ** 2, (s * s).mean() - s.mean() ** 2
    if var_r < 1e-5 or var_s < 1e-5:
        return 2.0, min(var_r, var_s)
    cov = (r * s).mean() - r.mean() * s.mean()
    return float(np.clip(1.0 - cov / np.sqrt(var_r * var_s), 0.0, 2.0)), min(var_r, var_s)


def _ncc_old64(sc, imgs, px, py, src, plane):
    H = _homography(sc, src, plane)
    cx, cy = _warp(H, px, py)
    if not (0 <= cx < sc.width and 0 <= cy < sc.height):
        return 2.0, 1.0
    return _patch_cost(imgs[0], imgs[src], H, px, py, 5, 2)


def _random_planes(rng, sc, gt, px, py, n):
    """planes (normal, distance) through the pixel's ray at depths around the true one, normals within ~35 degrees of -z"""
    K = sc.K[0].astype(np.float64)
    out = []
    for _ in range(n):
        depth = gt[py, px] * rng.uniform(0.9, 1.1)
        nrm = np.array([rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), -1.0])
        nrm /= np.linalg.norm(nrm)
        X = depth * np.array([(px - K[2]) / K[0], (py - K[5]) / K[4], 1.0])
        out.append(np.array([nrm[0], nrm[1], nrm[2], -nrm @ X]))
    return out


def test_ncc_old_against_float64(ob, synth):
    W, H, N = 96, 72, 3
    sc, imgs = common.scene_inputs(synth, W, H, N, seed=5)
    o = common.make_oracle(ob, sc, imgs, N, common.base_params(sc, N))
    gt = sc.gt_depth.numpy()
    rng = np.random.RandomState(0)
    diffs, twos = [], 0
    for _ in range(150):
        px, py = int(rng.randint(0, W)), int(rng.randint(0, H))
        for plane in _random_planes(rng, sc, gt, px, py, 2):
            for src in range(1, N + 1):
                want, margin = _ncc_old64(sc, imgs, px, py, src, plane)
                got = o.ncc_old(px, py, src, plane.astype(np.float32))
                if margin < 1e-3:      # at the variance threshold the binary32 rounding may land on the other side
                    continue
                twos += want == 2.0
                diffs.append(abs(got - want))
    diffs = np.array(diffs)
    assert len(diffs) > 800 and twos > 5
    assert diffs.max() < 5e-3 and np.median(diffs) < 1e-4, (diffs.max(), np.median(diffs))
    o.close()


def test_ncc_new_against_float64(ob, synth):
    """The deformable cost of WEAK pixels: 0.25 * centre (6x6, stride 2) + 0.75 * mean over the reliable neighbours of a 3x3
    (stride 5) sub-patch cost, all under the centre pixel's homography; a neighbour that projects outside the image costs 2 if
    that view is selected there, and is skipped otherwise (:436-449)."""
    W, H, N = 96, 72, 3
    sc, imgs = common.scene_inputs(synth, W, H, N, seed=3, textureless=0.3)
    p0 = common.base_params(sc, N, max_iterations=2, weak_peak_radius=6)
    o0 = common.make_oracle(ob, sc, imgs, N, p0)
    o0.run()
    prior = common.postprocess(o0.planes.copy(), o0.weak_info.copy(), o0.selected_views.copy(), p0["depth_min"], p0["depth_max"])
    o0.close()
    p = common.base_params(sc, N, state=1, use_APD=1, weak_peak_radius=6, rotate_time=2, ransac_threshold=0.00875)
    o = common.make_oracle(ob, sc, imgs, N, p, prior=prior)
    for k in (1, 2, 3, 4):
        o.run_kernel(k)
    weak = np.argwhere(o.weak_info == ob.WEAK)
    assert len(weak) > 50
    gt = sc.gt_depth.numpy()
    rng = np.random.RandomState(1)
    sel = o.selected_views
    diffs = []
    for (py, px) in weak[rng.permutation(len(weak))[:80]]:
        py, px = int(py), int(px)
        nb = o.neighbours[o.neighbours_map[py, px]]
        for plane in _random_planes(rng, sc, gt, px, py, 2):
            for src in range(1, N + 1):
                Hm = _homography(sc, src, plane)
                cx, cy = _warp(Hm, px, py)
                risky = False
                if not (0 <= cx < W and 0 <= cy < H):
                    want = 2.0
                else:
                    centre, m = _patch_cost(imgs[0], imgs[src], Hm, px, py, 5, 2)
                    risky |= m < 1e-3
                    acc, cnt = 0.0, 0
                    for k in range(1, 9):
                        qx, qy = int(nb[k][0]), int(nb[k][1])
                        if qx == -1 or qy == -1:
                            continue
                        nx, ny = _warp(Hm, qx, qy)
                        if nx < 0 or ny < 0 or nx >= W or ny >= H:
                            if (int(sel[qy, qx]) >> (src - 1)) & 1:
                                acc += 2.0
                                cnt += 1
                            continue
                        c, m = _patch_cost(imgs[0], imgs[src], Hm, qx, qy, 5, 5)
                        risky |= m < 1e-3
                        acc += c
                        cnt += 1
                    want = centre if cnt == 0 else 0.25 * centre + 0.75 * min(acc / cnt, 2.0)
                if risky:
                    continue
                diffs.append(abs(o.ncc_new(px, py, src, plane.astype(np.float32)) - want))
    diffs = np.array(diffs)
    assert len(diffs) > 200
    # textureless patches: variances of a few grey levels^2 amplify the binary32 rounding of the moments
    assert diffs.max() < 5e-3 and np.median(diffs) < 6e-4, (diffs.max(), np.median(diffs))
    o.close()


def test_geometric_cost_against_float64(ob, synth):
    """Forward-backward reprojection error against the source view's depth map, truncated pixel look-up, 3 if that depth is 0,
    capped at 3 (:760-789)."""
    W, H, N = 96, 72, 3
    sc, imgs = common.scene_inputs(synth, W, H, N, seed=4)
    deps = common.fake_depth_maps(W, H, N + 1)
    o = common.make_oracle(ob, sc, imgs, N, common.base_params(sc, N, state=2, geom_consistency=1), depths=deps)
    gt = sc.gt_depth.numpy()
    rng = np.random.RandomState(2)
    Kr, Rr, tr, Cr = _cams(sc, 0)
    diffs, threes = [], 0
    for _ in range(300):
        px, py = int(rng.randint(0, W)), int(rng.randint(0, H))
        plane = _random_planes(rng, sc, gt, px, py, 1)[0]
        src = int(rng.randint(1, N + 1))
        Ks, Rs, ts, Cs = _cams(sc, src)
        n, w = plane[:3], plane[3]
        depth = -w * Kr[0, 0] / ((px - Kr[0, 2]) * n[0] + (Kr[0, 0] / Kr[1, 1]) * (py - Kr[1, 2]) * n[1] + Kr[0, 0] * n[2])
        X = Rr.T @ (depth * np.array([(px - Kr[0, 2]) / Kr[0, 0], (py - Kr[1, 2]) / Kr[1, 1], 1.0])) + Cr
        v = Ks @ (Rs @ X + ts)
        sx, sy = v[0] / v[2], v[1] / v[2]
        sd = _texel(deps[src], int(sx), int(sy))   # (int) truncates toward zero; the texture clamps
        if sd == 0.0:
            want = 3.0
            threes += 1
        else:
            Y = Rs.T @ (sd * np.array([(sx - Ks[0, 2]) / Ks[0, 0], (sy - Ks[1, 2]) / Ks[1, 1], 1.0])) + Cs
            b = Kr @ (Rr @ Y + tr)
            want = min(3.0, float(np.hypot(px - b[0] / b[2], py - b[1] / b[2])))
        if abs(sx - round(sx)) < 1e-3 or abs(sy - round(sy)) < 1e-3:
            continue  # the truncation may pick the neighbouring texel in binary32
        diffs.append(abs(o.geom_cost(px, py, src, plane.astype(np.float32)) - want))
    diffs = np.array(diffs)
    assert len(diffs) > 250 and threes >= 1
    assert diffs.max() < 2e-2 and np.median(diffs) < 2e-4, (diffs.max(), np.median(diffs))
    o.close()
