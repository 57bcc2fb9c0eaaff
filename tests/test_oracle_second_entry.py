"""Double entry for the control-flow-heavy stages of the path.  The reference ships no tests or vectors and cannot be built
here, so the oracle (oracle/apd_oracle.c) is pinned by inspection only.  This file is a SECOND restatement of four stages,
written from the reference's text (APD.cu line numbers below) and not from the oracle's: plain Python over numpy binary32
scalars, one statement per statement.  It shares with the oracle only leaf functions that have their own independent checks --
the NCC / geometric cost of one (pixel, view, plane) (tests/test_oracle_float64.py), the XORWOW stream (tests/test_rng.py,
pinned to rocRAND) and the polynomial exp of the arithmetic contract -- and must reproduce the oracle's state BIT FOR BIT:

  * adaptive checkerboard arm search + multi-hypothesis joint view selection of CheckerboardPropagationStrong
    (APD.cu:1012-1259): the view weights of every pixel of a colour;
  * GenNeighbours, K3 (APD.cu:1750-1969): neighbour table, reliability flags and the random state it leaves behind;
  * the peak classifier of DepthToWeak, K14 (APD.cu:1990-2143): the weak map.

Two independent transcriptions that agree on every bit do not prove either right, but a slip in one of them (a swapped arm,
a `<` for a `<=`, a draw out of order) shows up here."""
import ctypes as C
import math

import numpy as np

import common

f32 = np.float32
FLT_EPSILON = f32(1.1920929e-07)
FLT_MAX = f32(3.4028234663852886e38)
WEAK, STRONG, UNKNOWN = 0, 1, 2


# ---- leaves shared with the oracle (each has its own independent test) ----------------------------------------------------

class Rng:
    """curand / curand_uniform on one pixel's XORWOW state (6 words: x0..x4, d)."""

    def __init__(self, ob, words):
        self.L = ob.lib()
        self.state = (C.c_uint32 * 6)(*[int(w) for w in words])

    def next(self):
        return int(self.L.orc_xorwow_next(self.state))

    def uniform(self):
        return f32(self.L.orc_xorwow_uniform(self.state))

    def words(self):
        return np.array(list(self.state), np.uint32)


def expf(ob, x):
    return f32(ob.lib().orc_expf(C.c_float(float(x))))


# ---- small device helpers, APD.cu:29-142 ----------------------------------------------------------------------------------

def is_set(v, n):  # :52-55
    return (int(v) >> n) & 1


def normalize2(x, y):  # :136-142, rsqrtf := 1 / sqrtf (contract C4)
    inv = f32(1.0) / np.sqrt(f32(x * x + y * y))
    return f32(x * inv), f32(y * inv)


def normalize3(x, y, z):  # :128-134
    inv = f32(1.0) / np.sqrt(f32(f32(x * x + y * y) + z * z))
    return f32(x * inv), f32(y * inv), f32(z * inv)


def point_in_triangle(A, B, Cc, P):  # :91-112, short2 / int2 arguments
    ABx, ABy = f32(B[0] - A[0]), f32(B[1] - A[1])
    BCx, BCy = f32(Cc[0] - B[0]), f32(Cc[1] - B[1])
    CAx, CAy = f32(A[0] - Cc[0]), f32(A[1] - Cc[1])
    ab = np.sqrt(f32(ABx * ABx + ABy * ABy))
    bc = np.sqrt(f32(BCx * BCx + BCy * BCy))
    ca = np.sqrt(f32(CAx * CAx + CAy * CAy))
    if ab <= 2 or bc <= 2 or ca <= 2:
        return False
    if not (f32(ab + bc) > ca and f32(bc + ca) > ab and f32(ab + ca) > bc):
        return False
    PAx, PAy = f32(A[0] - P[0]), f32(A[1] - P[1])
    PBx, PBy = f32(B[0] - P[0]), f32(B[1] - P[1])
    PCx, PCy = f32(Cc[0] - P[0]), f32(Cc[1] - P[1])
    t1 = f32(PAx * PBy - PAy * PBx)
    t2 = f32(PBx * PCy - PBy * PCx)
    t3 = f32(PCx * PAy - PCy * PAx)
    return bool(f32(t1 * t2) >= 0 and f32(t1 * t3) >= 0)


def get_3d_point(K, px, py, depth):  # :159-172
    depth = f32(depth)
    return (f32(f32(depth * f32(f32(px) - K[2])) / K[0]), f32(f32(depth * f32(f32(py) - K[5])) / K[4]), depth)


def distance_to_origin(K, px, py, depth, n):  # :186-192
    X = get_3d_point(K, px, py, depth)
    return f32(-f32(f32(f32(n[0] * X[0]) + f32(n[1] * X[1])) + f32(n[2] * X[2])))


def normal_to_ref_cam(R, p):  # TransformNormal2RefCam, :383-392
    return (f32(f32(f32(R[0] * p[0]) + f32(R[1] * p[1])) + f32(R[2] * p[2])),
            f32(f32(f32(R[3] * p[0]) + f32(R[4] * p[1])) + f32(R[5] * p[2])),
            f32(f32(f32(R[6] * p[0]) + f32(R[7] * p[1])) + f32(R[8] * p[2])), f32(p[3]))


# ---- stage 1: arm search + joint view selection, APD.cu:1012-1259 ---------------------------------------------------------

def arm_search(costs, W, H, px, py):
    """positions[8] and flag[8] in the reference's order: 0 up_near, 1 up_far, 2 down_near, 3 down_far, 4 left_near,
    5 left_far, 6 right_near, 7 right_far (:1020); every probe is `costs[q] < costMin` on the flat index."""
    c = costs.reshape(-1)
    center = py * W + px
    pos = [center - W, center - 3 * W, center + W, center + 3 * W, center - 1, center - 3, center + 1, center + 3]
    flag = [False] * 8

    def scan(first, probes):
        best, cmin = first, c[first]
        for ok, q in probes:
            if ok and c[q] < cmin:
                cmin, best = c[q], q
        return best

    if py > 2:  # up_far :1021-1038
        flag[1] = True
        pos[1] = scan(pos[1], [(py > 2 + 2 * i, center - 3 * W - 2 * i * W) for i in range(1, 11)])
    if py < H - 3:  # down_far :1040-1057
        flag[3] = True
        pos[3] = scan(pos[3], [(py < H - 3 - 2 * i, center + 3 * W + 2 * i * W) for i in range(1, 11)])
    if px > 2:  # left_far :1059-1076
        flag[5] = True
        pos[5] = scan(pos[5], [(px > 2 + 2 * i, center - 3 - 2 * i) for i in range(1, 11)])
    if px < W - 3:  # right_far :1078-1095
        flag[7] = True
        pos[7] = scan(pos[7], [(px < W - 3 - 2 * i, center + 3 + 2 * i) for i in range(1, 11)])
    if py > 0:  # up_near :1097-1121
        flag[0] = True
        probes = []
        for i in range(3):
            probes.append((py > 1 + i and px > i, center - W - (1 + i) * W - (1 + i)))
            probes.append((py > 1 + i and px < W - 1 - i, center - W - (1 + i) * W + (1 + i)))
        pos[0] = scan(pos[0], probes)
    if py < H - 1:  # down_near :1123-1147
        flag[2] = True
        probes = []
        for i in range(3):
            probes.append((py < H - 2 - i and px > i, center + W + (1 + i) * W - (1 + i)))
            probes.append((py < H - 2 - i and px < W - 1 - i, center + W + (1 + i) * W + (1 + i)))
        pos[2] = scan(pos[2], probes)
    if px > 0:  # left_near :1149-1173
        flag[4] = True
        probes = []
        for i in range(3):
            probes.append((px > 1 + i and py > i, center - 1 - (1 + i) - (1 + i) * W))
            probes.append((px > 1 + i and py < H - 1 - i, center - 1 - (1 + i) + (1 + i) * W))
        pos[4] = scan(pos[4], probes)
    if px < W - 1:  # right_near :1175-1199
        flag[6] = True
        probes = []
        for i in range(3):
            probes.append((px < W - 2 - i and py > i, center + 1 + (1 + i) - (1 + i) * W))
            probes.append((px < W - 2 - i and py < H - 1 - i, center + 1 + (1 + i) + (1 + i) * W))
        pos[6] = scan(pos[6], probes)
    return pos, flag


def view_weights_of_pixel(ob, o, snap, W, H, nsrc, px, py, it):
    costs, planes, sel, rng_words = snap
    pos, flag = arm_search(costs, W, H, px, py)
    cost_array = np.zeros((8, 32), np.float32)
    cost_array[0, 0] = f32(2.0)  # `= { 2.0f }` sets one element (:1004)
    flat_planes = planes.reshape(-1, 4)
    for a in range(8):
        if flag[a]:
            for v in range(nsrc):  # ComputeMultiViewCostVectorOld :696-705
                cost_array[a, v] = f32(o.ncc_old(px, py, v + 1, flat_planes[pos[a]]))
    center = py * W + px
    priors = np.zeros(32, np.float32)
    nb = [center - W, center + W, center - 1, center + 1]
    flat_sel = sel.reshape(-1)
    for i in range(4):  # :1210-1222
        if flag[2 * i]:
            for j in range(nsrc):
                priors[j] = f32(priors[j] + (f32(0.9) if is_set(flat_sel[nb[i]], j) == 1 else f32(0.1)))
    probs = np.zeros(32, np.float32)
    thr = f32(0.8 * float(expf(ob, f32(it * it) / f32(-90.0))))  # double product, then float (:1225)
    for i in range(nsrc):
        count, count_false, tmpw = f32(0), 0, f32(0)
        for j in range(8):
            cij = cost_array[j, i]
            if cij < thr:
                tmpw = f32(tmpw + expf(ob, f32(f32(cij * cij) / f32(-0.18))))
                count = f32(count + f32(1))
            if cij > f32(1.2):
                count_false += 1
        if count > 2 and count_false < 3:
            probs[i] = f32(tmpw / count)
        elif count_false < 3:
            probs[i] = expf(ob, f32(f32(thr * thr) / f32(-0.32)))
        probs[i] = f32(probs[i] * priors[i])
    with np.errstate(divide="ignore", invalid="ignore"):  # TransformPDFToCDF :143-157; a zero sum gives inf / NaN like the device
        s = f32(0)
        for i in range(nsrc):
            s = f32(s + probs[i])
        inv = f32(1.0) / s
        cum = f32(0)
        for i in range(nsrc):
            cum = f32(cum + f32(probs[i] * inv))
            probs[i] = cum
    rng = Rng(ob, rng_words[py, px])
    weights = np.zeros(32, np.uint8)
    for _ in range(15):  # :1249-1259
        rand_prob = f32(rng.uniform() - FLT_EPSILON)
        for v in range(nsrc):
            if probs[v] > rand_prob:
                weights[v] += 1
                break
    return weights


def test_arm_search_and_view_selection(synth, ob):
    W, H, N = 40, 30, 3
    sc, imgs = common.scene_inputs(synth, W, H, N, seed=21)
    o = common.make_oracle(ob, sc, imgs, N, common.base_params(sc, N, seed=77, max_iterations=2))
    for kid in (1, 2, 5):
        o.run_kernel(kid)
    checked = 0
    for it in (0, 1):
        for colour, kid in ((0, 6), (1, 7)):
            snap = (o.costs.copy(), o.planes.copy(), o.selected_views.copy(), o.rng.copy())
            o.run_kernel(kid, it)
            got = o.view_weight
            for py in range(H):
                for px in range(W):
                    if (px + py) % 2 != colour:
                        continue
                    want = view_weights_of_pixel(ob, o, snap, W, H, N, px, py, it)
                    assert np.array_equal(got[py, px], want), (it, colour, px, py, got[py, px][:N], want[:N])
                    checked += 1
    assert checked == 2 * W * H
    o.close()


# ---- stage 2: GenNeighbours, APD.cu:1750-1969 -----------------------------------------------------------------------------

def to_short(v):
    """float -> short2 member: truncation toward zero (values stay far inside the short range here)."""
    return int(np.trunc(v))


def gen_neighbours_pixel(ob, W, H, K, params, weak, nearest, planes, rng_words, px, py):
    """Returns (neighbours[9] as (x, y), reliable, rng words after) for one WEAK pixel."""
    rng = Rng(ob, rng_words[py, px])
    min_margin = 6
    depth_diff = f32(f32(params["depth_max"]) - f32(params["depth_min"]))
    rotate_time = params["rotate_time"]
    angle = f32(f32(45.0) / f32(rotate_time))
    cos_angle = f32(math.cos(float(angle) * math.pi / float(f32(180.0))))
    sin_angle = f32(math.sin(float(angle) * math.pi / float(f32(180.0))))
    threshold = f32(math.cos(float(f32(angle / f32(2.0))) * math.pi / float(f32(180.0))))
    shift_range = max(int(math.tan(float(f32(angle / f32(2.0))) * math.pi / float(f32(180.0))) * 20), 1)
    ransac_threshold = f32(params["ransac_threshold"])
    strong = [(-1, -1)] * 32
    valid = [False] * 32
    dir_index0 = -1
    found = 0

    def shift():  # (curand() % 2 == 0 ? 1 : -1) * curand() % shift_range, unsigned; sign draw first (contract C8)
        sign = 1 if rng.next() % 2 == 0 else 0xFFFFFFFF
        mag = rng.next()
        return ((sign * mag) & 0xFFFFFFFF) % shift_range

    for ox in (-1, 0, 1):
        for oy in (-1, 0, 1):
            if ox == 0 and oy == 0:
                continue
            dx, dy = normalize2(f32(ox), f32(oy))
            dir_index0 += 1
            for rot in range(rotate_time):
                slot = dir_index0 * 4 + rot
                radius = 2
                while radius <= 4096:
                    tx, ty = f32(f32(px) + f32(dx * f32(radius))), f32(f32(py) + f32(dy * f32(radius)))
                    if tx < 0 or ty < 0 or tx >= W or ty >= H:
                        break
                    for _ in range(4):
                        sx = shift()
                        sy = shift()
                        ddx, ddy = normalize2(f32(f32(dx * f32(20)) + f32(sx)), f32(f32(dy * f32(20)) + f32(sy)))
                        qx, qy = to_short(f32(f32(px) + f32(ddx * f32(radius)))), to_short(f32(f32(py) + f32(ddy * f32(radius))))
                        if qx < min_margin or qy < min_margin or qx >= W - min_margin or qy >= H - min_margin:
                            continue
                        if weak[qy, qx] != STRONG:
                            qx, qy = int(nearest[qy, qx, 0]), int(nearest[qy, qx, 1])
                            if qx == -1 or qy == -1:
                                continue
                        tdx, tdy = normalize2(f32(qx - px), f32(qy - py))
                        ca = f32(f32(tdx * dx) + f32(tdy * dy))
                        if ca > threshold:
                            strong[slot] = (qx, qy)
                            valid[slot] = True
                            found += 1
                            break
                    if valid[slot]:
                        break
                    radius = min(radius * 2, radius + 25)
                rx = f32(f32(dx * cos_angle) - f32(dy * sin_angle))
                ry = f32(f32(dx * sin_angle) + f32(dy * cos_angle))
                dx, dy = normalize2(rx, ry)
    out = [(-1, -1)] * 9
    out[0] = (px, py)
    if found <= 3:
        return out, 0, rng.words()
    center3 = get_3d_point(K, px, py, planes[py, px, 3])  # .w still holds the DEPTH before K5 (:1866)
    pts, pts3 = [], []
    for i in range(32):
        if valid[i]:
            q = strong[i]
            pts.append(q)
            pts3.append(get_3d_point(K, q[0], q[1], planes[q[1], q[0], 3]))
    n = len(pts)
    pts = pts + [(-1, -1)] * (32 - n)
    best, use, has_plane = None, (-1, -1, -1), False
    min_cost, max_count = FLT_MAX, 3

    def plane_dist(pl, P):
        return np.abs(f32(f32(f32(f32(pl[0] * P[0]) + f32(pl[1] * P[1])) + f32(pl[2] * P[2])) + pl[3]))

    for _ in range(50):
        a = rng.next() % n
        b = rng.next() % n
        c = rng.next() % n
        if a == b or b == c or a == c:
            continue
        if not point_in_triangle(pts[a], pts[b], pts[c], (px, py)):
            continue
        A, B, Cc = pts3[a], pts3[b], pts3[c]
        ACx, ACy, ACz = f32(A[0] - Cc[0]), f32(A[1] - Cc[1]), f32(A[2] - Cc[2])
        BCx, BCy, BCz = f32(B[0] - Cc[0]), f32(B[1] - Cc[1]), f32(B[2] - Cc[2])
        nx = f32(f32(ACy * BCz) - f32(BCy * ACz))
        ny = f32(-f32(f32(ACx * BCz) - f32(BCx * ACz)))
        nz = f32(f32(ACx * BCy) - f32(BCx * ACy))
        if (nx == 0 and ny == 0 and nz == 0) or np.isnan(nx) or np.isnan(ny) or np.isnan(nz):
            continue
        nx, ny, nz = normalize3(nx, ny, nz)
        nw = f32(-f32(f32(f32(nx * A[0]) + f32(ny * A[1])) + f32(nz * A[2])))
        pl = (nx, ny, nz, nw)
        count = 0
        for k in range(n):
            if f32(plane_dist(pl, pts3[k]) / depth_diff) < ransac_threshold:
                count += 1
        if count < 6:
            continue
        if count > max_count:
            max_count = count
            min_cost = plane_dist(pl, center3)
            best, use, has_plane = pl, (a, b, c), True
        elif count == max_count:
            cd = plane_dist(pl, center3)
            if cd < min_cost:
                min_cost = cd
                best, use = pl, (a, b, c)
    if not has_plane:
        return out, 0, rng.words()
    weight = [f32(0)] * n
    for i in range(n):
        d = plane_dist(best, pts3[i])
        if f32(d / depth_diff) >= ransac_threshold:
            pts[i] = (-1, -1)
            weight[i] = FLT_MAX
            continue
        if i in use:
            d = f32(d - f32(1))
        weight[i] = d
    for i in range(1, n):  # sort_small_weighted :14-27
        tmp, tw = pts[i], weight[i]
        j = i
        while j >= 1 and tw < weight[j - 1]:
            pts[j], weight[j] = pts[j - 1], weight[j - 1]
            j -= 1
        pts[j], weight[j] = tmp, tw
    for i in range(1, 9):
        out[i] = pts[i - 1]
    return out, 1, rng.words()


def _apd_oracle(synth, ob, W, H, N, seed):
    """A REFINE_INIT + APD oracle whose prior comes from a complete FIRST_INIT pass of the same scene."""
    sc, imgs = common.scene_inputs(synth, W, H, N, seed=seed, textureless=0.3)
    p0 = common.base_params(sc, N, seed=5, max_iterations=2, weak_peak_radius=6)
    o0 = common.make_oracle(ob, sc, imgs, N, p0)
    o0.run()
    prior = common.postprocess(o0.planes.copy(), o0.weak_info.copy(), o0.selected_views.copy(), f32(p0["depth_min"]), f32(p0["depth_max"]))
    o0.close()
    p1 = common.base_params(sc, N, seed=6, max_iterations=1, state=1, use_APD=1, weak_peak_radius=6, rotate_time=2, ransac_threshold=0.00875)
    return sc, p1, common.make_oracle(ob, sc, imgs, N, p1, prior=prior)


def test_gen_neighbours(synth, ob):
    W, H, N = 128, 96, 3   # a textureless region wide enough for the radius sequence to leave its doubling phase (57, 82, ...)
    sc, p1, o = _apd_oracle(synth, ob, W, H, N, seed=12)
    for kid in (1, 2):
        o.run_kernel(kid)
    weak = o.weak_info.copy()
    assert (weak == WEAK).sum() > 40, "the scene must have a textureless region"
    snap = dict(weak=weak, nearest=o.nearest_strong.copy(), planes=o.planes.copy(), rng=o.rng.copy())
    nmap = o.neighbours_map.copy()
    o.run_kernel(3)
    K = [f32(v) for v in sc.K[0].reshape(-1)]
    nb, reliable, rng_after = o.neighbours, o.weak_reliable, o.rng
    n_reliable = 0
    for py in range(H):
        for px in range(W):
            if weak[py, px] != WEAK:
                assert np.array_equal(rng_after[py, px], snap["rng"][py, px])
                continue
            want_nb, want_rel, want_rng = gen_neighbours_pixel(ob, W, H, K, p1, snap["weak"], snap["nearest"], snap["planes"], snap["rng"], px, py)
            row = nb[nmap[py, px]]
            assert [tuple(int(v) for v in q) for q in row] == want_nb, (px, py, row.tolist(), want_nb)
            assert int(reliable[py, px]) == want_rel, (px, py)
            assert np.array_equal(rng_after[py, px], want_rng), (px, py)
            n_reliable += want_rel
    assert n_reliable > 10, "some pixels must get a full neighbour set, or the RANSAC half is not exercised"
    o.close()


def test_gen_neighbours_long_rays(synth, ob):
    """The same on a hand-made WEAK block 150 px wide: the rays leave the doubling phase of the radius sequence
    (2, 4, 8, 16, 32, 57, 82, ...; :1807) and run into the substitution by the nearest STRONG pixel; every 7th WEAK pixel."""
    W, H, N = 200, 140, 2
    sc, imgs = common.scene_inputs(synth, W, H, N, seed=4)
    p0 = common.base_params(sc, N, seed=5, max_iterations=1, weak_peak_radius=6)
    o0 = common.make_oracle(ob, sc, imgs, N, p0)
    o0.run()
    planes, views, weak0 = common.postprocess(o0.planes.copy(), o0.weak_info.copy(), o0.selected_views.copy(), f32(p0["depth_min"]), f32(p0["depth_max"]))
    o0.close()
    weak = np.where(weak0 == UNKNOWN, UNKNOWN, STRONG).astype(np.uint8)
    weak[20:120, 25:175] = WEAK
    weak[60:64, 90:96] = STRONG   # an island inside the block
    p1 = common.base_params(sc, N, seed=6, max_iterations=1, state=1, use_APD=1, weak_peak_radius=6, rotate_time=4, ransac_threshold=0.0075)
    o = common.make_oracle(ob, sc, imgs, N, p1, prior=(planes, views, weak))
    for kid in (1, 2):
        o.run_kernel(kid)
    snap = dict(weak=o.weak_info.copy(), nearest=o.nearest_strong.copy(), planes=o.planes.copy(), rng=o.rng.copy())
    nmap = o.neighbours_map.copy()
    o.run_kernel(3)
    K = [f32(v) for v in sc.K[0].reshape(-1)]
    nb, reliable, rng_after = o.neighbours, o.weak_reliable, o.rng
    checked = far = 0
    for py in range(H):
        for px in range(W):
            if snap["weak"][py, px] != WEAK or (px * 31 + py * 17) % 7 != 0:
                continue
            want_nb, want_rel, want_rng = gen_neighbours_pixel(ob, W, H, K, p1, snap["weak"], snap["nearest"], snap["planes"], snap["rng"], px, py)
            row = nb[nmap[py, px]]
            assert [tuple(int(v) for v in q) for q in row] == want_nb, (px, py, row.tolist(), want_nb)
            assert int(reliable[py, px]) == want_rel and np.array_equal(rng_after[py, px], want_rng), (px, py)
            checked += 1
            far += any(q[0] >= 0 and max(abs(q[0] - px), abs(q[1] - py)) > 45 for q in want_nb[1:])
    assert checked > 1500 and far > 100, (checked, far)
    o.close()


# ---- stage 3: DepthToWeak, APD.cu:1990-2143 -------------------------------------------------------------------------------

def depth_to_weak_pixel(o, cams_c, K, R, params, planes, sel, vweight, nsrc, W, H, px, py, geom):
    min_margin = 6
    if px < min_margin or py < min_margin or px >= W - min_margin or py >= H - min_margin:
        return UNKNOWN
    origin = normal_to_ref_cam(R, planes[py, px])
    origin_depth = origin[3]
    if origin_depth == 0:
        return UNKNOWN
    s = int(sel[py, px])
    vw = vweight[py, px]
    base_line, valid, weight_normal = f32(0), 0, f32(0)
    for v in range(nsrc):
        if is_set(s, v):
            cd = [f32(cams_c[0][k] - cams_c[v + 1][k]) for k in range(3)]
            tv = float(f32(f32(f32(cd[0] * cd[0]) + f32(cd[1] * cd[1])) + f32(cd[2] * cd[2])))  # float expression stored in a double (:2043)
            base_line = f32(base_line + f32(np.sqrt(f32(tv))))   # sqrtf(double) converts back to float first
            weight_normal = f32(weight_normal + f32(vw[v]))
            valid += 1
    if valid == 0:
        return UNKNOWN
    base_line = f32(base_line / f32(valid))
    disp = f32(f32(K[0] * base_line) / origin_depth)
    radius, n = 30, 61
    p_costs = [f32(0)] * n
    for p_disp in range(-radius, radius + 1):
        p_depth = f32(f32(K[0] * base_line) / f32(disp + f32(p_disp)))
        if p_depth < f32(params["depth_min"]) or p_depth > f32(params["depth_max"]):
            p_costs[p_disp + radius] = f32(2.0)
            continue
        pl = np.array([origin[0], origin[1], origin[2], distance_to_origin(K, px, py, p_depth, origin)], np.float32)
        p_cost = f32(0)
        for v in range(nsrc):
            if is_set(s, v):
                t = f32(0)
                t = f32(t + f32(o.ncc_old(px, py, v + 1, pl)))
                if geom:
                    t = f32(t + f32(f32(params["geom_factor"]) * f32(o.geom_cost(px, py, v + 1, pl))))
                p_cost = f32(p_cost + f32(t * f32(vw[v])))
        with np.errstate(divide="ignore", invalid="ignore"):
            p_cost = f32(p_cost / weight_normal)
        p_costs[p_disp + radius] = f32(2.0) if f32(2.0) < p_cost else p_cost   # MIN(2.0f, p_cost) = (a < b) ? a : b
    is_peak = [False] * n
    peak_count, min_peak, min_cost = 0, 0, f32(2.0)
    for i in range(2, n - 2):  # :2101-2110
        if p_costs[i - 1] > p_costs[i] and p_costs[i + 1] > p_costs[i]:
            is_peak[i] = True
            peak_count += 1
            if p_costs[i] < min_cost:
                min_peak, min_cost = i, p_costs[i]
    if abs(min_peak - radius) > params["weak_peak_radius"] or p_costs[min_peak] > f32(0.5):
        return WEAK
    if peak_count == 1:
        return STRONG if p_costs[min_peak] <= f32(0.15) else WEAK
    var = f32(0)
    for i in range(2, n - 2):
        if is_peak[i] and i != min_peak:
            d = f32(p_costs[i] - min_cost)
            var = f32(var + f32(d * d))
    var = f32(np.sqrt(var))
    var = f32(var / f32(peak_count - 1))
    return STRONG if var > f32(0.2) else WEAK


def test_depth_to_weak_classifier(synth, ob):
    W, H, N = 56, 40, 3
    sc, imgs = common.scene_inputs(synth, W, H, N, seed=14, textureless=0.25)
    for geom in (0, 1):
        kw = dict(seed=9, max_iterations=2, weak_peak_radius=4 if geom else 6)
        depths = None
        prior = None
        if geom:
            kw.update(state=2, geom_consistency=1)
            depths = common.fake_depth_maps(W, H, N + 1)
            o0 = common.make_oracle(ob, sc, imgs, N, common.base_params(sc, N, seed=3, max_iterations=2, weak_peak_radius=6))
            o0.run()
            p0 = common.base_params(sc, N)
            prior = common.postprocess(o0.planes.copy(), o0.weak_info.copy(), o0.selected_views.copy(), f32(p0["depth_min"]), f32(p0["depth_max"]))
            o0.close()
        params = common.base_params(sc, N, **kw)
        full = ob.default_params(**params)
        params["geom_factor"] = full.geom_factor
        o = common.make_oracle(ob, sc, imgs, N, params, depths=depths, prior=prior)
        for kid in (1, 2, 5):
            o.run_kernel(kid)
        o.run_sweeps(0, 2)
        for kid in (11, 12, 13):
            o.run_kernel(kid)
        planes, sel, vw = o.planes.copy(), o.selected_views.copy(), o.view_weight.copy()
        o.run_kernel(14)
        got = o.weak_info
        K = [f32(v) for v in sc.K[0].reshape(-1)]
        R = [f32(v) for v in sc.R[0].reshape(-1)]
        # Camera::c = -R^T t accumulated in double, stored as float (APD.cpp:75-77); ob.make_camera does the same
        cams_c = [[f32(v) for v in ob.make_camera(sc.K[i], sc.R[i], sc.t[i], W, H, sc.depth_min, sc.depth_max).c] for i in range(N + 1)]
        counts = [0, 0, 0]
        for py in range(H):
            for px in range(W):
                want = depth_to_weak_pixel(o, cams_c, K, R, params, planes, sel, vw, N, W, H, px, py, geom)
                assert int(got[py, px]) == want, (geom, px, py, int(got[py, px]), want)
                counts[want] += 1
        assert min(counts) > 0, counts   # all three classes occur
        o.close()
