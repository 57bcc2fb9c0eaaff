"""Edge cases of the HIP path against the oracle (bit-exact, through the C ABI): degenerate plane hypotheses
that leave the fast-reciprocal range, the float-image (non texel-quad) weak path, crafted WEAK maps for the
separable nearest-strong search, and the exhaustive ISA checks the arithmetic contract leans on."""
import os
import subprocess

import numpy as np
import pytest

import common

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _canon(a):
    """Raw words with every NaN mapped to one pattern (x86 and gfx950 differ in the sign/payload of generated NaNs)."""
    a = np.ascontiguousarray(a)
    if a.dtype == np.float32:
        w = a.view(np.uint32).copy()
        w[np.isnan(a)] = 0x7FC00000
        return w
    return a.view(np.uint8)


def _assert_states(pkg, h, o, where):
    for name, hs, oa in common.ORACLE_STATES:
        if name == "neighbours" and h.weak_count == 0:
            continue
        a, b = h.state(getattr(pkg, hs)), getattr(o, oa)
        assert np.array_equal(_canon(a), _canon(b)), "%s: `%s` differs" % (where, name)


def test_degenerate_hypotheses_take_the_ieee_path(gpu_pkg, ob, synth):
    """Planes through / next to the camera centre, huge distances, normals at 90 degrees to the viewing ray and a
    zero normal: denominators of the warp cross zero or leave [2^-100, 2^100], so the sample loop must fall back
    to the IEEE division and still agree with the oracle bit for bit (Inf/NaN coordinates included)."""
    W, H, N = 64, 48, 3
    sc, imgs = common.scene_inputs(synth, W, H, N)
    p = common.base_params(sc, N, max_iterations=1)
    h = common.make_handle(gpu_pkg, sc, imgs, N, p)
    o = common.make_oracle(ob, sc, imgs, N, p)
    for kid in (1, 2, 5):
        h.run_kernel(kid)
        o.run_kernel(kid)
    planes = h.state(gpu_pkg.STATE_PLANES).copy()
    costs = h.state(gpu_pkg.STATE_COSTS).copy()
    bad = [(0.0, 0.0, -1.0, 1e-30), (0.0, 0.0, -1.0, -1e-30), (0.0, 0.0, -1.0, 0.0), (0.6, 0.0, -0.8, 1e-38),
           (0.0, 0.0, -1.0, 3e38), (1.0, 0.0, 0.0, 2.0), (0.0, 1.0, 0.0, -2.0), (0.0, 0.0, 0.0, 2.0),
           (0.7071068, 0.0, -0.7071068, 1e-3), (-0.9999, 0.0, -0.0141, 5e-5), (1e20, -1e20, 1e20, 1.0), (0.0, 0.0, -1.0, 1e-12)]
    rng = np.random.RandomState(5)
    ys, xs = rng.randint(0, H, 400), rng.randint(0, W, 400)
    for k, (y, x) in enumerate(zip(ys, xs)):
        planes[y, x] = bad[k % len(bad)]
        costs[y, x] = 0.0  # cheapest neighbour of every arm that reaches it: the plane gets evaluated elsewhere too
    h.set_state(gpu_pkg.STATE_PLANES, planes)
    h.set_state(gpu_pkg.STATE_COSTS, costs)
    o.planes[...] = planes
    o.costs[...] = costs
    for kid in (6, 7, 8, 11, 12, 13, 14, 15):
        h.run_kernel(kid)
        o.run_kernel(kid)
        _assert_states(gpu_pkg, h, o, "degenerate planes after K%d" % kid)
    h.close()
    o.close()


def test_weak_path_on_float_images(gpu_pkg, ob, synth):
    """REFINE_INIT + APD on non-integer images: K9/K10 must use the float sampler and the generic sub-patch."""
    W, H, N = 80, 60, 3
    sc, imgs = common.scene_inputs(synth, W, H, N, seed=3, textureless=0.25)
    imgs = [im * np.float32(0.5) + np.float32(0.25) for im in imgs]
    prior = None
    weak_seen = 0
    for pi, extra in enumerate([dict(state=0, use_APD=0, weak_peak_radius=6),
                                dict(state=1, use_APD=1, weak_peak_radius=6, rotate_time=2, ransac_threshold=0.00875)]):
        p = common.base_params(sc, N, seed=13, **extra)
        h = common.make_handle(gpu_pkg, sc, imgs, N, p, prior=prior)
        o = common.make_oracle(ob, sc, imgs, N, p, prior=prior)
        weak_seen = max(weak_seen, h.weak_count)
        sched = [1, 2] + ([3, 4] if h.weak_count else []) + [5]
        for i in range(p["max_iterations"]):
            sched += [(6, i), (7, i), (8, i)] + ([(9, i), (10, i)] if h.weak_count else [])
        sched += [11, 12, 13, 14, 15]
        for s in sched:
            kid, it = (s, 0) if isinstance(s, int) else s
            h.run_kernel(kid, it)
            o.run_kernel(kid, it)
            common.assert_state_equal(gpu_pkg, h, o, "float weak pass %d after K%d" % (pi, kid))
        planes, weak, views = h.download()
        prior = common.postprocess(planes, weak, views, p["depth_min"], p["depth_max"])
        h.close()
        o.close()
    assert weak_seen > 50


@pytest.mark.parametrize("W,H", [(430, 330), (97, 150), (301, 64)])
def test_nearest_strong_search_on_crafted_maps(gpu_pkg, ob, synth, W, H):
    """K2: big WEAK blocks (interior farther than 100 px from any STRONG pixel -> (-1,-1)), isolated STRONG pixels
    (equidistant candidates: the reference's column-major scan order breaks the tie), UNKNOWN pixels (never a
    target), image borders and windows clipped by the image."""
    N = 1
    sc, imgs = common.scene_inputs(synth, W, H, N)
    p = common.base_params(sc, N, max_iterations=1)
    h = common.make_handle(gpu_pkg, sc, imgs, N, p)
    o = common.make_oracle(ob, sc, imgs, N, p)
    rng = np.random.RandomState(W * 7 + H)
    weak = np.ones((H, W), np.uint8)                      # STRONG
    weak[5:H - 3, 4:W - 2] = 0                            # one huge WEAK block, thin STRONG frame
    xmax = min(W - 3, 90)
    for _ in range(12):                                   # isolated STRONG pixels in its left part
        weak[rng.randint(6, H - 4), rng.randint(5, xmax)] = 1
    cx, cy = min(W // 2, 60), H // 2                      # symmetric cross: four candidates at the same distance
    for dx, dy in ((-7, 0), (7, 0), (0, -7), (0, 7), (-5, -5), (5, 5), (-5, 5), (5, -5)):
        weak[cy + dy, cx + dx] = 1
    sy, sx = rng.randint(0, H, 200), rng.randint(0, xmax, 200)
    weak[sy, sx] = 2                                      # UNKNOWN specks
    h.set_state(gpu_pkg.STATE_WEAK_INFO, weak)
    o.weak_info[...] = weak
    h.run_kernel(2)
    o.run_kernel(2)
    got, want = h.state(gpu_pkg.STATE_NEAREST_STRONG), o.nearest_strong
    assert np.array_equal(got, want)
    if W > 410 and H > 210:
        assert (want[weak == 0][:, 0] == -1).any(), "the case must contain WEAK pixels without a STRONG pixel in reach"
    assert (want[weak == 0][:, 0] >= 0).any()
    assert (want[weak != 0] == -1).all()
    h.close()
    o.close()


def test_isa_contract_exhaustive():
    """The hardware facts the bit-exact contract relies on, checked over all 2^32 float inputs on this GPU (and, for the two-operand
    division, over 2^32 pseudo-random pairs):
    v_rcp_f32 + one FMA Newton step == IEEE 1/z for biased exponents 27..227; v_fract_f32 == min(x - floor(x),
    1 - 2^-24) for finite x and NaN otherwise; v_cvt_flr_i32_f32 == saturating (int)floor(x) for non-NaN x."""
    exe = os.path.join(ROOT, "tools", "_build", "valu_rates")
    assert os.path.exists(exe), "tools/_build/valu_rates not built: run __graft_entry__.build()"
    out = subprocess.run([exe, "--check"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600).stdout
    facts = dict(line.split("=", 1) for line in out.splitlines() if line.startswith("CHECK_"))
    assert facts.get("CHECK_recip_midrange_mismatches") == "0", out
    assert facts.get("CHECK_fract_finite_mismatches") == "0", out
    assert facts.get("CHECK_fract_nonfinite_not_nan") == "0", out
    assert facts.get("CHECK_cvt_flr_non_nan_mismatches") == "0", out
    # the NCC epilogue: square root and division without the compiler's range scaling == sqrtf and `/` on the ranges the kernels use
    assert facts.get("CHECK_sqrt_midrange_mismatches") == "0" and facts.get("CHECK_sqrt_nan_not_nan") == "0", out
    assert facts.get("CHECK_div_midrange_mismatches") == "0" and int(facts.get("CHECK_div_pairs", "0")) == 1 << 32, out


@pytest.mark.parametrize("threshold,rotate", [(float("inf"), 4), (0.0, 4), (0.005, 3), (1e-30, 2)])
def test_gen_neighbours_with_and_without_a_distance_cut(gpu_pkg, ob, synth, threshold, rotate):
    """K3's inlier test is a comparison with a host-computed cut where one exists (csrc/apd_capi.hip: ransac_distance_cut) and
    the reference's division where none does (an infinite threshold); with rotate_time 4 the jitter range is 1 and a
    (slot, radius) is probed once instead of four identical times, with 3 it is 2 and the four attempts differ.  Every
    combination must leave the oracle's bits: neighbours, reliability, RNG words."""
    W, H, N = 96, 72, 4
    sc, imgs = common.scene_inputs(synth, W, H, N, seed=3, textureless=0.25)
    prior = None
    weak_seen = 0
    for pi, extra in enumerate([dict(state=0, use_APD=0, weak_peak_radius=6),
                                dict(state=1, use_APD=1, weak_peak_radius=6, rotate_time=rotate, ransac_threshold=threshold)]):
        p = common.base_params(sc, N, seed=11, **extra)
        h = common.make_handle(gpu_pkg, sc, imgs, N, p, prior=prior)
        o = common.make_oracle(ob, sc, imgs, N, p, prior=prior)
        weak_seen = max(weak_seen, h.weak_count)
        sched = [1, 2] + ([3, 4] if h.weak_count else []) + [5]
        for i in range(p["max_iterations"]):
            sched += [(6, i), (7, i), (8, i)] + ([(9, i), (10, i)] if h.weak_count else [])
        sched += [11, 12, 13, 14, 15]
        for s in sched:
            kid, it = (s, 0) if isinstance(s, int) else s
            h.run_kernel(kid, it)
            o.run_kernel(kid, it)
            common.assert_state_equal(gpu_pkg, h, o, "threshold %r rotate %d pass %d after K%d" % (threshold, rotate, pi, kid))
        planes, weak, views = h.download()
        prior = common.postprocess(planes, weak, views, p["depth_min"], p["depth_max"])
        h.close()
        o.close()
    assert weak_seen > 50
