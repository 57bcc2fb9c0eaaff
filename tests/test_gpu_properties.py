"""Size-independent properties at BASELINE.json's full sizes (the oracle cannot finish these in
seconds): determinism, value ranges, quality against the analytic scene, and consistency of the
full-size run with an oracle run on a window whose dependency cone it fully contains."""
import hashlib

import numpy as np
import pytest

import common

pytestmark = pytest.mark.gpu


def _digest(arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def _run_full(gpu_pkg, synth, W, H, N, iters, seed=12345):
    import torch
    sc = synth.make_scene(W, H, N, seed=0, device="cuda")
    cams = [gpu_pkg.make_camera(sc.K[i], sc.R[i], sc.t[i], W, H, sc.depth_min, sc.depth_max) for i in range(N + 1)]
    p = gpu_pkg.default_params(num_images=N + 1, depth_min=0.6 * sc.depth_min, depth_max=1.2 * sc.depth_max, use_APD=0,
                               state=gpu_pkg.FIRST_INIT, max_iterations=iters, seed=seed)
    h = gpu_pkg.Handle(W, H, p, device=0)
    h.upload_views(cams, sc.images)
    for k in (1, 2, 5):
        h.run_kernel(k)
    h.run_sweeps(0, iters)
    planes = h.state(gpu_pkg.STATE_PLANES)
    costs = h.state(gpu_pkg.STATE_COSTS)
    views = h.state(gpu_pkg.STATE_SELECTED_VIEWS)
    gt = sc.gt_depth.cpu().numpy()
    K = sc.K[0].astype(np.float64)
    h.close()
    del sc
    torch.cuda.empty_cache()
    return planes, costs, views, gt, K, p


def test_full_size_config1_properties(gpu_pkg, synth):
    """configs[1] shape: 6200x4130, 8 source views."""
    W, H, N, iters = 6200, 4130, 8, 2
    planes, costs, views, gt, K, p = _run_full(gpu_pkg, synth, W, H, N, iters)
    d1 = _digest([planes, costs, views])
    # ranges: costs in [0, 2] or NaN (live NaN path, Appendix A #9); bit j of the view mask < N
    finite = np.isfinite(costs)
    assert finite.mean() > 0.999
    assert (costs[finite] >= 0).all() and (costs[finite] <= 2.0).all()
    assert (views >> N == 0).all()
    nrm = np.linalg.norm(planes[..., :3].astype(np.float64), axis=-1)
    assert np.abs(nrm - 1).max() < 1e-3
    # quality against the analytic scene after 2 sweeps
    d = common.depth_of_planes(planes[::7, ::7], K) if False else None
    ys, xs = np.mgrid[0:H:5, 0:W:5]
    pl = planes[::5, ::5].astype(np.float64)
    dd = -pl[..., 3] * K[0] / ((xs - K[2]) * pl[..., 0] + (K[0] / K[4]) * (ys - K[5]) * pl[..., 1] + K[0] * pl[..., 2])
    g = gt[::5, ::5]
    assert ((np.abs(dd - g) / g)[4:-4, 4:-4] < 0.01).mean() > 0.97
    # determinism: a second handle with the same seed gives the same bits
    planes2, costs2, views2, _, _, _ = _run_full(gpu_pkg, synth, W, H, N, iters)
    assert _digest([planes2, costs2, views2]) == d1


def test_full_size_stress_shape_determinism(gpu_pkg, synth):
    """configs[4] shape: 4096x3072, 16 source views (NMAX = 16 kernel), one sweep."""
    W, H, N = 4096, 3072, 16
    a = _run_full(gpu_pkg, synth, W, H, N, 1, seed=99)
    b = _run_full(gpu_pkg, synth, W, H, N, 1, seed=99)
    assert _digest(a[:3]) == _digest(b[:3])
    c = _run_full(gpu_pkg, synth, W, H, N, 1, seed=100)
    assert _digest(a[:3]) != _digest(c[:3])


def test_red_black_independence(gpu_pkg, ob, synth):
    """Within one colour the processing order must not matter: the XCD-banded tile order of the HIP
    launch and the row-major order of the oracle give identical bits at a size with many tiles."""
    W, H, N = 416, 304, 4
    sc, imgs = common.scene_inputs(synth, W, H, N)
    p = common.base_params(sc, N, max_iterations=1)
    h = common.make_handle(gpu_pkg, sc, imgs, N, p)
    o = common.make_oracle(ob, sc, imgs, N, p)
    for kid in (1, 2, 5, 6, 7):
        h.run_kernel(kid, 0)
        o.run_kernel(kid, 0)
    common.assert_state_equal(gpu_pkg, h, o, "416x304 one sweep", skip=("fit",))
    h.close()
    o.close()


def test_window_and_global_kernels_agree(gpu_pkg, synth, monkeypatch):
    """K6/K7 with per-wave LDS source windows (apd_kernels_k67w.hip, default) against the kernel that gathers every
    sample from HBM (APD_OPT_K67_WINDOWS = 0): all state bit-identical after every iteration at a size where, from the second
    iteration on, most NCCs read the windows (the oracle is too slow for this size)."""
    W, H, N, iters = 1536, 1152, 5, 4
    runs = {}
    for mode in ("1", "0"):
        sc = synth.make_scene(W, H, N, seed=2, device="cuda", textureless=0.1)
        cams = [gpu_pkg.make_camera(sc.K[i], sc.R[i], sc.t[i], W, H, sc.depth_min, sc.depth_max) for i in range(N + 1)]
        p = gpu_pkg.default_params(num_images=N + 1, depth_min=0.6 * sc.depth_min, depth_max=1.2 * sc.depth_max, use_APD=0,
                                   state=gpu_pkg.FIRST_INIT, max_iterations=iters, seed=31)
        h = gpu_pkg.Handle(W, H, p, device=0)
        h.set_option("k67_windows", int(mode))
        assert h.get_option("k67_windows") == int(mode)
        h.upload_views(cams, sc.images)
        for k in (1, 2, 5):
            h.run_kernel(k)
        digests = []
        for it in range(iters):
            h.run_sweeps(it, 1)
            digests.append(_digest([h.state(gpu_pkg.STATE_PLANES), h.state(gpu_pkg.STATE_COSTS), h.state(gpu_pkg.STATE_SELECTED_VIEWS),
                                    h.state(gpu_pkg.STATE_VIEW_WEIGHT), h.state(gpu_pkg.STATE_RNG)]))
        runs[mode] = digests
        h.close()
    assert runs["1"] == runs["0"]


def test_window_and_global_post_loop_kernels_agree(gpu_pkg, synth, monkeypatch):
    """K14 DepthToWeak / K15 LocalRefine with LDS source windows (apd_kernels_k1415w.hip, default) against the
    window-less kernels (APD_OPT_K1415_WINDOWS = 0): identical weak map and depths after a complete pass, photometric and with
    the geometric term, at a size the oracle cannot reach."""
    import torch
    W, H, N = 1280, 960, 5
    out = {}
    for mode in ("1", "0"):
        sc = synth.make_scene(W, H, N, seed=4, device="cuda", textureless=0.15)
        cams = [gpu_pkg.make_camera(sc.K[i], sc.R[i], sc.t[i], W, H, sc.depth_min, sc.depth_max) for i in range(N + 1)]
        dmin, dmax = 0.6 * sc.depth_min, 1.2 * sc.depth_max
        digests = []
        prior = None
        for state, geom in ((gpu_pkg.FIRST_INIT, 0), (gpu_pkg.REFINE_ITER, 1)):
            p = gpu_pkg.default_params(num_images=N + 1, depth_min=dmin, depth_max=dmax, use_APD=0, state=state, max_iterations=2,
                                       weak_peak_radius=6, geom_consistency=geom, seed=77)
            h = gpu_pkg.Handle(W, H, p, device=0)
            h.set_option("k1415_windows", int(mode))
            deps = [sc.gt_depth] * (N + 1) if geom else None   # stand-in depth maps of the sources
            h.upload_views(cams, sc.images, deps)
            if prior is not None:
                h.upload_prior(*prior)
            h.run()
            planes, weak, views = h.download()
            digests.append(_digest([planes, weak, views]))
            prior = common.postprocess(planes, weak, views, np.float32(dmin), np.float32(dmax))
            h.close()
        out[mode] = digests
        del sc
        torch.cuda.empty_cache()
    assert out["1"] == out["0"]


def test_early_outs_change_no_bit(gpu_pkg, synth, monkeypatch):
    """The exact early-outs (refinement hypotheses of K6/K7 and K9/K10 that can no longer beat the running cost, K15's depth
    samples that cannot be adopted, K14's centre-first classification) against APD_OPT_EARLY_OUT = 0, which
    evaluates every NCC the reference evaluates: every state array identical after each of the three pass kinds, weak
    pixels and the geometric term included, at a size the oracle cannot reach."""
    import torch
    W, H, N = 1024, 768, 6
    out = {}
    for mode in ("1", "0"):
        sc = synth.make_scene(W, H, N, seed=9, device="cuda", textureless=0.25)
        cams = [gpu_pkg.make_camera(sc.K[i], sc.R[i], sc.t[i], W, H, sc.depth_min, sc.depth_max) for i in range(N + 1)]
        dmin, dmax = 0.6 * sc.depth_min, 1.2 * sc.depth_max
        digests, weak_counts = [], []
        prior = None
        passes = [dict(state=gpu_pkg.FIRST_INIT, use_APD=0, weak_peak_radius=6),
                  dict(state=gpu_pkg.REFINE_INIT, use_APD=1, weak_peak_radius=6, rotate_time=2, ransac_threshold=0.01 - 0.00125),
                  dict(state=gpu_pkg.REFINE_ITER, use_APD=1, weak_peak_radius=2, rotate_time=2, ransac_threshold=0.01 - 0.00125,
                       geom_consistency=1)]
        for pi, extra in enumerate(passes):
            p = gpu_pkg.default_params(num_images=N + 1, depth_min=dmin, depth_max=dmax, max_iterations=3, seed=41 + pi, **extra)
            h = gpu_pkg.Handle(W, H, p, device=0)
            h.set_option("early_out", int(mode))
            deps = [sc.gt_depth] * (N + 1) if extra.get("geom_consistency") else None
            h.upload_views(cams, sc.images, deps)
            if prior is not None:
                h.upload_prior(*prior)
            h.run()
            weak_counts.append(h.weak_count)
            planes, weak, views = h.download()
            digests.append(_digest([planes, weak, views, h.state(gpu_pkg.STATE_COSTS), h.state(gpu_pkg.STATE_RNG),
                                    h.state(gpu_pkg.STATE_VIEW_WEIGHT)]))
            prior = common.postprocess(planes, weak, views, np.float32(dmin), np.float32(dmax))
            h.close()
        out[mode] = digests
        assert weak_counts[1] > 1000   # the weak sweep really ran
        del sc
        torch.cuda.empty_cache()
    assert out["1"] == out["0"]


def test_recycled_handle_equals_a_new_one(gpu_pkg, synth):
    """apd_reset: a handle that already ran another (view, pass) -- other images, other parameters, geometric term, WEAK
    pixels -- gives the same bits as a freshly created one."""
    W, H, N = 160, 120, 4
    sc_a, imgs_a = common.scene_inputs(synth, W, H, N, seed=5, textureless=0.25)
    sc_b, imgs_b = common.scene_inputs(synth, W, H, N - 1, seed=9)
    deps = common.fake_depth_maps(W, H, N + 1)
    pa = common.base_params(sc_a, N, seed=21, state=0, use_APD=0, weak_peak_radius=6, max_iterations=2)
    h = common.make_handle(gpu_pkg, sc_a, imgs_a, N, pa)
    h.run()
    planes, weak, views = h.download()
    prior = common.postprocess(planes, weak, views, pa["depth_min"], pa["depth_max"])
    pa2 = common.base_params(sc_a, N, seed=22, state=2, use_APD=1, weak_peak_radius=4, rotate_time=2, ransac_threshold=0.00875,
                             geom_consistency=1, max_iterations=2)
    h.reset(gpu_pkg.default_params(**pa2))
    cams_a = [gpu_pkg.make_camera(sc_a.K[i], sc_a.R[i], sc_a.t[i], W, H, sc_a.depth_min, sc_a.depth_max) for i in range(N + 1)]
    h.upload_views(cams_a, imgs_a, deps)
    h.upload_prior(*prior)
    h.run()
    assert h.weak_count > 0
    # now a completely different job on the recycled handle and on a new one
    pb = common.base_params(sc_b, N - 1, seed=33, state=0, use_APD=0, weak_peak_radius=6, max_iterations=2)
    h.reset(gpu_pkg.default_params(**pb))
    cams_b = [gpu_pkg.make_camera(sc_b.K[i], sc_b.R[i], sc_b.t[i], W, H, sc_b.depth_min, sc_b.depth_max) for i in range(N)]
    h.upload_views(cams_b, imgs_b)
    h.run()
    fresh = common.make_handle(gpu_pkg, sc_b, imgs_b, N - 1, pb)
    fresh.run()
    for which in (gpu_pkg.STATE_PLANES, gpu_pkg.STATE_COSTS, gpu_pkg.STATE_SELECTED_VIEWS, gpu_pkg.STATE_VIEW_WEIGHT,
                  gpu_pkg.STATE_WEAK_INFO, gpu_pkg.STATE_RNG, gpu_pkg.STATE_FIT_PLANES):
        assert np.array_equal(common.bits(h.state(which)), common.bits(fresh.state(which))), which
    h.close()
    fresh.close()


@pytest.mark.parametrize("use_apd", [0, 1])
def test_split_pass_around_the_depth_maps_changes_no_bit(gpu_pkg, ob, synth, use_apd):
    """apd_upload_views_split + apd_run_before_depths + apd_upload_depths + apd_run_after_depths (a scheduler that starts a
    view before its sources of the same pass have published their depth maps) == apd_run == the oracle, for a geometric pass
    with and without WEAK pixels and for a photometric pass; the kernels that read depth maps are refused while the maps are
    outstanding, and rubbish in the depth buffers before the late upload changes nothing."""
    W, H, N = 96, 72, 4
    sc, imgs = common.scene_inputs(synth, W, H, N, seed=3, textureless=0.25 if use_apd else 0.0)
    cams = [gpu_pkg.make_camera(sc.K[i], sc.R[i], sc.t[i], W, H, sc.depth_min, sc.depth_max) for i in range(N + 1)]
    p0 = common.base_params(sc, N, seed=11, state=0, use_APD=0, weak_peak_radius=6)
    h = gpu_pkg.Handle(W, H, gpu_pkg.default_params(**p0))
    h.upload_views_split(cams, imgs)          # photometric: nothing to wait for, the first half is the whole pass
    h.run_before_depths()
    h.run_after_depths()
    o = common.make_oracle(ob, sc, imgs, N, p0)
    o.run()
    common.assert_state_equal(gpu_pkg, h, o, "split photometric pass")
    planes, weak, views = h.download()
    prior = common.postprocess(planes, weak, views, p0["depth_min"], p0["depth_max"])
    o.close()
    depths = common.fake_depth_maps(W, H, N + 1)
    p1 = common.base_params(sc, N, seed=12, state=2, use_APD=use_apd, weak_peak_radius=4, geom_consistency=1, max_iterations=2)
    h.reset(gpu_pkg.default_params(**p1))
    h.upload_views_split(cams, imgs)
    h.upload_prior(prior[0], prior[1], prior[2] if use_apd else None)
    assert (h.weak_count > 50) == bool(use_apd)
    for kid in (9, 10, 14, 15):
        with pytest.raises(gpu_pkg.ApdError, match="apd_upload_depths"):
            h.run_kernel(kid)
    with pytest.raises(gpu_pkg.ApdError, match="apd_run_before_depths has not run"):
        h.run_after_depths()             # the second half alone: K9 / K10 / K14 / K15 on planes nobody initialised
    with pytest.raises(gpu_pkg.ApdError, match="apd_upload_depths"):
        h.run()                          # a whole pass is refused up front while the maps are outstanding, not at K9 with the handle half-run
    h.profile_enable(True)
    for kid in (9, 14):                  # refused launches take nothing from the event pool and record nothing
        with pytest.raises(gpu_pkg.ApdError, match="apd_upload_depths"):
            h.run_kernel(kid)
    assert h.profile() == {}
    h.profile_enable(False)
    h.run_before_depths()
    with pytest.raises(gpu_pkg.ApdError, match="already ran"):
        h.run_before_depths()
    with pytest.raises(gpu_pkg.ApdError, match="apd_run_after_depths"):
        h.run()
    with pytest.raises(gpu_pkg.ApdError, match="apd_upload_depths"):
        h.run_after_depths()
    h.synchronize()
    h.upload_depths(depths)
    h.run_after_depths()
    with pytest.raises(gpu_pkg.ApdError):
        h.run_after_depths()             # the pass is complete
    o = common.make_oracle(ob, sc, imgs, N, p1, depths=depths, prior=(prior[0], prior[1], prior[2] if use_apd else None))
    o.run()
    common.assert_state_equal(gpu_pkg, h, o, "split geometric pass, use_APD=%d" % use_apd)
    h.close()
    o.close()


def test_shared_images_give_the_bits_of_copied_ones(gpu_pkg, ob, synth):
    """apd_image_create + apd_upload_views_shared (level images created once on the device and handed to many handles by reference)
    == apd_upload_views (every handle copies, tests and packs its images) == the oracle: a FIRST_INIT pass (tiled copies on demand), an
    APD pass and a geometric pass, 8-bit and float images, two handles sharing the images at the same time."""
    W, H, N = 96, 72, 4
    for float_images in (False, True):
        sc, imgs = common.scene_inputs(synth, W, H, N, seed=5, textureless=0.25)
        if float_images:
            imgs = [(im * np.float32(0.731) + np.float32(2.5)).astype(np.float32) for im in imgs]
        cams = [gpu_pkg.make_camera(sc.K[i], sc.R[i], sc.t[i], W, H, sc.depth_min, sc.depth_max) for i in range(N + 1)]
        shared = [gpu_pkg.SharedImage(W, H, im) for im in imgs]
        depths = common.fake_depth_maps(W, H, N + 1)
        passes = [dict(state=0, use_APD=0, weak_peak_radius=6), dict(state=1, use_APD=1, weak_peak_radius=6, rotate_time=2),
                  dict(state=2, use_APD=1, weak_peak_radius=4, geom_consistency=1)]
        prior = None
        ha = gpu_pkg.Handle(W, H, gpu_pkg.default_params(**common.base_params(sc, N)))
        hb = gpu_pkg.Handle(W, H, gpu_pkg.default_params(**common.base_params(sc, N)))
        for pi, extra in enumerate(passes):
            p = common.base_params(sc, N, seed=21 + pi, **extra)
            o = common.make_oracle(ob, sc, imgs, N, p, depths=depths if extra.get("geom_consistency") else None, prior=prior)
            o.run()
            for h in (ha, hb):      # both handles read the same shared images, one after the other launching, both in flight
                h.reset(gpu_pkg.default_params(**p))
                h.upload_views_shared(cams, shared)
                if prior is not None:
                    h.upload_prior(*prior)
                h.run_before_depths()
            for h in (ha, hb):
                if extra.get("geom_consistency"):
                    h.upload_depths(depths)
                h.run_after_depths()
                common.assert_state_equal(gpu_pkg, h, o, "shared images, %s, pass %d" % ("float" if float_images else "8-bit", pi))
            planes, weak, views = ha.download()
            prior = common.postprocess(planes, weak, views, p["depth_min"], p["depth_max"])
            o.close()
        ha.close()
        hb.close()
        for im in shared:
            im.close()
