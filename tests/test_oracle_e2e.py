"""Behaviour of the whole oracle schedule (APD.cu:2386-2495): convergence the survey measured on the
reference's own device code, determinism, reference quirks."""
import hashlib

import numpy as np
import pytest

import common


def _digest(o):
    h = hashlib.sha256()
    for a in (o.planes, o.costs, o.selected_views, o.view_weight, o.weak_info, o.rng, o.fit_planes):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def _first_pass(ob, synth, W=96, H=72, N=4, iters=3, threads=None, textureless=0.0, seed=7):
    sc, imgs = common.scene_inputs(synth, W, H, N, textureless=textureless)
    if threads:
        ob.lib().orc_set_threads(threads)
    o = common.make_oracle(ob, sc, imgs, N, common.base_params(sc, N, max_iterations=iters, seed=seed, weak_peak_radius=6))
    return sc, imgs, o


def test_first_pass_converges(ob, synth):
    """Slanted-plane scene, 3 iterations: >= 99 % of interior pixels within 1 % depth (the reference's
    device code compiled for the CPU reached 99.8 % in the survey), 100 % after K12-K15."""
    sc, imgs, o = _first_pass(ob, synth)
    for k in (1, 2, 5):
        o.run_kernel(k)
    o.run_sweeps(0, 3)
    gt = sc.gt_depth.numpy()
    d = common.depth_of_planes(o.planes, sc.K[0].astype(np.float64))
    assert ((np.abs(d - gt) / gt)[8:-8, 8:-8] < 0.01).mean() >= 0.99
    assert not np.isnan(o.costs).any()
    for k in (11, 12, 13, 14, 15):
        o.run_kernel(k)
    d = o.planes[..., 3]
    assert ((np.abs(d - gt) / gt)[8:-8, 8:-8] < 0.01).mean() >= 0.995
    # K14: 6 px border -> UNKNOWN (APD.cu:2001-2004); textured interior -> STRONG
    wi = o.weak_info
    assert (wi[:6] == ob.UNKNOWN).all() and (wi[-6:] == ob.UNKNOWN).all()
    assert (wi[:, :6] == ob.UNKNOWN).all() and (wi[:, -6:] == ob.UNKNOWN).all()
    assert (wi[6:-6, 6:-6] == ob.STRONG).mean() > 0.95
    # normals are unit length, world frame
    nrm = np.linalg.norm(o.planes[..., :3], axis=-1)
    assert np.allclose(nrm, 1.0, atol=1e-4)


def test_deterministic_across_thread_counts(ob, synth):
    digests = []
    for threads in (1, 3, 8):
        sc, imgs, o = _first_pass(ob, synth, W=64, H=48, N=3, iters=2, threads=threads)
        o.run()
        digests.append(_digest(o))
        o.close()
    ob.lib().orc_set_threads(0)
    assert digests[0] == digests[1] == digests[2]


def test_seed_changes_result_and_same_seed_repeats(ob, synth):
    outs = []
    for seed in (7, 7, 8):
        sc, imgs, o = _first_pass(ob, synth, W=64, H=48, N=3, iters=1, seed=seed)
        o.run()
        outs.append(_digest(o))
    assert outs[0] == outs[1] and outs[0] != outs[2]


def test_border_pixels_never_adopt_a_neighbour(ob, synth):
    """Appendix A #2: an out-of-image arm has cost 0 for every view and wins FindMinCostIndex, so a
    pixel within 3 px of the border keeps its own plane in the propagation step (refinement may still
    change it): after K6 only, its plane is either unchanged or one of its own refinement planes --
    in particular it is never equal to the plane of any neighbour it could have propagated from."""
    sc, imgs, o = _first_pass(ob, synth, W=64, H=48, N=3, iters=1)
    for k in (1, 2, 5):
        o.run_kernel(k)
    before = o.planes.copy()
    views_before = o.selected_views.copy()
    o.run_kernel(6, 0)
    after = o.planes
    H, W = before.shape[:2]
    ys, xs = np.mgrid[0:H, 0:W]
    black = (xs + ys) % 2 == 0
    corner = black & ((xs < 1) | (ys < 1) | (xs > W - 2) | (ys > H - 2))
    # selected_views is only overwritten when a neighbour's plane is adopted (APD.cu:1306)
    assert np.array_equal(o.selected_views[corner], views_before[corner])
    # red pixels untouched by the black launch
    assert np.array_equal(after[~black], before[~black])
    # interior black pixels do change
    assert (after[black] != before[black]).any()


def test_odd_height_quirk_last_row_unvisited(ob, synth):
    """Appendix A #13: HALF launch uses height/2; for odd H with (H/2) % 16 == 0 the last row is never
    visited by a red/black kernel."""
    W, H, N = 40, 33, 2
    sc, imgs = common.scene_inputs(synth, W, H, N)
    o = common.make_oracle(ob, sc, imgs, N, common.base_params(sc, N, max_iterations=1))
    for k in (1, 2, 5):
        o.run_kernel(k)
    before = o.planes.copy()
    o.run_sweeps(0, 1)
    assert np.array_equal(o.planes[H - 1], before[H - 1])
    assert (o.planes[H - 2] != before[H - 2]).any()
    # with H = 35 (H/2 = 17 -> padded to 32 rows of pairs) the last row IS visited
    sc, imgs = common.scene_inputs(synth, W, 35, N)
    o = common.make_oracle(ob, sc, imgs, N, common.base_params(sc, N, max_iterations=1))
    for k in (1, 2, 5):
        o.run_kernel(k)
    before = o.planes.copy()
    o.run_sweeps(0, 1)
    assert (o.planes[34] != before[34]).any()


def test_apd_pass_exercises_weak_path(ob, synth):
    """pass 1 (FIRST_INIT) -> pass 2 (REFINE_INIT + APD): WEAK pixels exist, get reliable neighbours,
    and the pass does not destroy the depth map."""
    W, H, N = 96, 72, 4
    sc, imgs = common.scene_inputs(synth, W, H, N, seed=3, textureless=0.25)
    p1 = common.base_params(sc, N, seed=11, weak_peak_radius=6)
    o = common.make_oracle(ob, sc, imgs, N, p1)
    o.run()
    prior = common.postprocess(o.planes, o.weak_info, o.selected_views, p1["depth_min"], p1["depth_max"])
    n_weak = int((prior[2] == ob.WEAK).sum())
    assert n_weak > 50
    p2 = common.base_params(sc, N, seed=11, weak_peak_radius=6, state=ob.REFINE_INIT, use_APD=1, rotate_time=2,
                            ransac_threshold=0.01 - 0.00125)
    o2 = common.make_oracle(ob, sc, imgs, N, p2, prior=prior)
    assert o2.weak_count == n_weak
    for k in (1, 2, 3, 4):
        o2.run_kernel(k)
    nb = o2.neighbours
    rel = o2.weak_reliable[prior[2] == ob.WEAK]
    assert rel.sum() > 0
    # slot 0 of every weak pixel is the pixel itself (APD.cu:1781)
    ys, xs = np.nonzero(prior[2] == ob.WEAK)
    assert np.array_equal(nb[:, 0, 0], xs.astype(np.int16)) and np.array_equal(nb[:, 0, 1], ys.astype(np.int16))
    # reliable pixels have at least 6 neighbours (RANSAC needs >= 6 inliers, :1918), all STRONG
    wi = o2.weak_info
    for i in np.nonzero(rel)[0][:200]:
        pts = nb[i, 1:]
        valid = pts[pts[:, 0] >= 0]
        assert len(valid) >= 6
        assert (prior[2][valid[:, 1], valid[:, 0]] == ob.STRONG).all()
    # unreliable WEAK -> UNKNOWN (K4)
    assert ((wi == ob.WEAK).sum()) == int(rel.sum())
    o2.run_kernel(5)
    o2.run_sweeps(0, 2)
    for k in (11, 12, 13, 14, 15):
        o2.run_kernel(k)
    gt = sc.gt_depth.numpy()
    d = o2.planes[..., 3]
    assert ((np.abs(d - gt) / gt)[8:-8, 8:-8] < 0.02).mean() > 0.9
    assert not np.isnan(o2.planes).any()


def test_geometric_pass_runs_and_uses_depth_maps(ob, synth):
    W, H, N = 64, 48, 3
    sc, imgs = common.scene_inputs(synth, W, H, N, seed=3, textureless=0.25)
    p1 = common.base_params(sc, N, seed=5, weak_peak_radius=6)
    o = common.make_oracle(ob, sc, imgs, N, p1)
    o.run()
    prior = common.postprocess(o.planes, o.weak_info, o.selected_views, p1["depth_min"], p1["depth_max"])
    p3 = common.base_params(sc, N, seed=5, state=ob.REFINE_ITER, use_APD=1, geom_consistency=1, weak_peak_radius=4)
    deps = common.fake_depth_maps(W, H, N + 1)
    o3 = common.make_oracle(ob, sc, imgs, N, p3, depths=deps, prior=prior)
    # geometric cost: zero depth in the source map -> max cost 3.0 (APD.cu:774-776)
    zero = [np.zeros((H, W), np.float32) for _ in range(N + 1)]
    oz = common.make_oracle(ob, sc, imgs, N, p3, depths=zero, prior=prior)
    assert oz.geom_cost(30, 20, 1, [0.0, 0.0, -1.0, 2.0]) == 3.0
    c = o3.geom_cost(30, 20, 1, [0.0, 0.0, -1.0, 2.2])
    assert 0.0 <= c <= 3.0
    o3.run()
    assert not np.isnan(o3.planes).any()
