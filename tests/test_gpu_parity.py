"""HIP path vs CPU oracle through the C ABI, kernel by kernel, BIT-EXACT (float32 planes/costs are
compared as raw words; masks, weights, RNG state and neighbour tables are integer data).

The tolerance north_star states (1e-3 relative depth, 1 degree normals) is implied by bit equality;
it is asserted explicitly at the end of each case as well."""
import numpy as np
import pytest

import common
import golden_io

pytestmark = pytest.mark.gpu


def _schedule(iters, weak):
    s = [1, 2] + ([3, 4] if weak else []) + [5]
    for i in range(iters):
        s += [(6, i), (7, i), (8, i)] + ([(9, i), (10, i)] if weak else [])
    s += [11, 12, 13, 14, 15]
    return [(k, 0) if isinstance(k, int) else k for k in s]


def _lockstep(pkg, h, o, iters, label):
    for kid, it in _schedule(iters, h.weak_count > 0):
        h.run_kernel(kid, it)
        o.run_kernel(kid, it)
        common.assert_state_equal(pkg, h, o, "%s after K%d(iter %d)" % (label, kid, it))


def _tolerance_check(h_planes, o_planes):
    d_h, d_o = h_planes[..., 3].astype(np.float64), o_planes[..., 3].astype(np.float64)
    ok = np.isfinite(d_o) & (d_o != 0)
    assert (np.abs(d_h[ok] - d_o[ok]) <= 1e-3 * np.abs(d_o[ok])).all()
    cosang = np.clip((h_planes[..., :3] * o_planes[..., :3]).sum(-1), -1, 1)
    assert (np.degrees(np.arccos(cosang[ok])) <= 1.0).all()


@pytest.mark.parametrize("W,H,N,iters", [(64, 48, 3, 2), (97, 71, 5, 2), (40, 33, 2, 1), (33, 35, 1, 1), (80, 64, 8, 1)])
def test_first_pass_lockstep(gpu_pkg, ob, synth, W, H, N, iters):
    """FIRST_INIT pass; odd sizes, the odd-height HALF-launch quirk (H=33), N = 1..8."""
    sc, imgs = common.scene_inputs(synth, W, H, N)
    p = common.base_params(sc, N, max_iterations=iters, weak_peak_radius=6)
    h = common.make_handle(gpu_pkg, sc, imgs, N, p)
    o = common.make_oracle(ob, sc, imgs, N, p)
    _lockstep(gpu_pkg, h, o, iters, "first pass %dx%d N=%d" % (W, H, N))
    _tolerance_check(h.state(gpu_pkg.STATE_PLANES), o.planes)
    h.close()
    o.close()


def test_float_image_path(gpu_pkg, ob, synth):
    """Non-integer images (what a downscaled pyramid level holds, APD.cpp:474) cannot use the packed
    8-bit texel quads: the float sampler path must give the same bits as the oracle too."""
    W, H, N = 72, 56, 3
    sc, imgs = common.scene_inputs(synth, W, H, N)
    imgs = [im * np.float32(0.5) + np.float32(0.37) for im in imgs]
    p = common.base_params(sc, N, max_iterations=2, weak_peak_radius=6)
    h = common.make_handle(gpu_pkg, sc, imgs, N, p)
    o = common.make_oracle(ob, sc, imgs, N, p)
    _lockstep(gpu_pkg, h, o, 2, "float images")
    h.close()
    o.close()


@pytest.mark.parametrize("N", [9, 16, 17, 31])
def test_many_source_views(gpu_pkg, ob, synth, N):
    """NMAX = 16 and 32 instantiations of the sweep kernels (MAX_IMAGES = 32 -> at most 31 sources)."""
    W, H = 48, 40
    sc, imgs = common.scene_inputs(synth, W, H, N)
    p = common.base_params(sc, N, max_iterations=1)
    h = common.make_handle(gpu_pkg, sc, imgs, N, p)
    o = common.make_oracle(ob, sc, imgs, N, p)
    for kid, it in [(1, 0), (2, 0), (5, 0), (6, 0), (7, 0), (8, 0), (11, 0), (12, 0), (13, 0)]:
        h.run_kernel(kid, it)
        o.run_kernel(kid, it)
    common.assert_state_equal(gpu_pkg, h, o, "N=%d" % N)
    h.close()
    o.close()


def test_three_pass_pipeline_with_apd_and_geometric_term(gpu_pkg, ob, synth):
    """pass 1 FIRST_INIT -> pass 2 REFINE_INIT + adaptive patch deformation -> pass 3 REFINE_ITER +
    geometric consistency: the per-pass parameters of main.cpp:168-215."""
    W, H, N = 96, 72, 4
    sc, imgs = common.scene_inputs(synth, W, H, N, seed=3, textureless=0.25)
    deps = common.fake_depth_maps(W, H, N + 1)
    passes = [dict(state=0, use_APD=0, weak_peak_radius=6),
              dict(state=1, use_APD=1, weak_peak_radius=6, rotate_time=2, ransac_threshold=0.01 - 0.00125),
              dict(state=2, use_APD=1, weak_peak_radius=4, rotate_time=4, ransac_threshold=0.01 - 0.0025, geom_consistency=1)]
    prior = None
    weak_seen = 0
    for pi, extra in enumerate(passes):
        p = common.base_params(sc, N, seed=11, **extra)
        geom = bool(p.get("geom_consistency"))
        h = common.make_handle(gpu_pkg, sc, imgs, N, p, depths=deps if geom else None, prior=prior)
        o = common.make_oracle(ob, sc, imgs, N, p, depths=deps if geom else None, prior=prior)
        assert h.weak_count == o.weak_count
        weak_seen = max(weak_seen, h.weak_count)
        _lockstep(gpu_pkg, h, o, p["max_iterations"], "pass %d" % pi)
        planes, weak, views = h.download()
        prior = common.postprocess(planes, weak, views, p["depth_min"], p["depth_max"])
        h.close()
        o.close()
    assert weak_seen > 50, "the scene must drive pixels through the weak path"


def test_run_equals_stepwise(gpu_pkg, ob, synth):
    """apd_run (whole schedule, asynchronous) == oracle's orc_run."""
    W, H, N = 64, 48, 3
    sc, imgs = common.scene_inputs(synth, W, H, N)
    p = common.base_params(sc, N, max_iterations=3, weak_peak_radius=6)
    h = common.make_handle(gpu_pkg, sc, imgs, N, p)
    o = common.make_oracle(ob, sc, imgs, N, p)
    h.run()
    o.run()
    common.assert_state_equal(gpu_pkg, h, o, "apd_run")
    gt = sc.gt_depth.numpy()
    d = h.state(gpu_pkg.STATE_PLANES)[..., 3]
    assert ((np.abs(d - gt) / gt)[8:-8, 8:-8] < 0.01).mean() > 0.99
    h.close()
    o.close()


@pytest.mark.parametrize("case", golden_io.CASES)
def test_golden_fixtures(gpu_pkg, case):
    """HIP path against the committed golden vectors (no oracle involved at run time)."""
    fx = golden_io.Fixture(case)
    h = gpu_pkg.Handle(fx.W, fx.H, gpu_pkg.default_params(**fx.params), device=0)
    h.upload_views(fx.cameras(gpu_pkg), fx.imgs, fx.depths)
    if fx.prior is not None:
        h.upload_prior(*fx.prior)
    h.run()
    fx.check(h.state(gpu_pkg.STATE_PLANES), h.state(gpu_pkg.STATE_COSTS), h.state(gpu_pkg.STATE_SELECTED_VIEWS),
             h.state(gpu_pkg.STATE_WEAK_INFO), h.state(gpu_pkg.STATE_VIEW_WEIGHT), h.state(gpu_pkg.STATE_RNG),
             h.state(gpu_pkg.STATE_NEIGHBOURS) if h.weak_count else None)
    h.close()


def test_export_matches_process_problem_postprocessing(gpu_pkg, synth):
    import torch
    W, H, N = 64, 48, 3
    sc, imgs = common.scene_inputs(synth, W, H, N)
    p = common.base_params(sc, N, max_iterations=1)
    h = common.make_handle(gpu_pkg, sc, imgs, N, p)
    h.run()
    planes = h.state(gpu_pkg.STATE_PLANES)
    planes[:5, :, 3] = 100.0   # out of [depth_min, depth_max] -> exported as 0 (main.cpp:109-112)
    planes[5:8, :, 3] = 0.01
    h.set_state(gpu_pkg.STATE_PLANES, planes)
    planes, weak, views = h.download()
    depth = torch.empty((H, W), device="cuda", dtype=torch.float32)
    normal = torch.empty((H, W, 3), device="cuda", dtype=torch.float32)
    h.export_depth_normal(depth, normal)
    ref = common.postprocess(planes, weak, views, p["depth_min"], p["depth_max"])[0]
    assert np.array_equal(depth.cpu().numpy().view(np.uint32), ref[..., 3].view(np.uint32))
    assert np.array_equal(normal.cpu().numpy().view(np.uint32), np.ascontiguousarray(ref[..., :3]).view(np.uint32))
    assert (ref[..., 3] == 0).any()
    h.close()


@pytest.mark.parametrize("case", [3, 6, 16, 18, 29, 39])
def test_randomised_three_pass_cases(gpu_pkg, ob, synth, case):
    """Cases of tools/parity_fuzz.py (random size, N = 1..9, textureless share, iterations, 8-bit or float images) through
    FIRST_INIT, REFINE_INIT + APD and REFINE_ITER + APD + geometric term: every state array bit-identical after each pass.
    The full sweep (`python tools/parity_fuzz.py 40`) is part of the round's profile set."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("parity_fuzz", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                               "tools", "parity_fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.run_case(case)


@pytest.mark.parametrize("case", [1, 4, 7, 12, 22, 33])
def test_randomised_three_pass_cases_on_hard_scenes(gpu_pkg, ob, synth, case):
    """The same sweep on synth.HARD-like scenes (APD_FUZZ_HARD=1): slabs in front of the planes (depth steps, occlusions), per-view
    gain / offset, sources aiming off the target so that parts of the frame project outside them (APD.cu:546-548 returns 2.0
    there) -- the windows, early-outs and view selection away from their best case.  Full sweep: profiles/r05/parity_fuzz_hard_*.txt."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("parity_fuzz", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                               "tools", "parity_fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert "hard(" in mod.run_case(case, hard=True)


def _fuzz():
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("parity_fuzz", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                               "tools", "parity_fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("case", list(range(80000, 80036)))
def test_fuzz_sweep_easy(gpu_pkg, ob, synth, case):
    """36 further cases of tools/parity_fuzz.py in the driver's suite (the 700-case sweeps were builder-side logs):
    random size 36..260 x 30..180, N = 1..18, textureless share, iterations, 8-bit / float images, one handle per pass or one
    recycled, whole pass or split around the depth maps, shared level images -- every state array bit-identical after each of the
    three pass kinds."""
    _fuzz().run_case(case, hard=False)


@pytest.mark.parametrize("case", list(range(81000, 81030)))
def test_fuzz_sweep_hard(gpu_pkg, ob, synth, case):
    """30 further hard cases (slabs, gain / offset, sources aiming off the target)."""
    assert "hard(" in _fuzz().run_case(case, hard=True)


@pytest.mark.parametrize("W,H,N,float_images", [(80, 60, 20, False), (72, 56, 12, True), (64, 48, 31, False)])
def test_many_source_views_three_pass(gpu_pkg, ob, synth, W, H, N, float_images):
    """The 16- and 32-view instantiations of the sweep kernels (the reference allows MAX_IMAGES = 32 including the
    reference view, main.h:8) through the three pass kinds, bit-identical after every pass."""
    sc, imgs = common.scene_inputs(synth, W, H, N, seed=2, textureless=0.2)
    if float_images:
        imgs = [(im * np.float32(0.9) + np.float32(1.7)).astype(np.float32) for im in imgs]
    deps = common.fake_depth_maps(W, H, N + 1)
    passes = [dict(state=0, use_APD=0, weak_peak_radius=6),
              dict(state=1, use_APD=1, weak_peak_radius=6, rotate_time=2, ransac_threshold=0.00875),
              dict(state=2, use_APD=1, weak_peak_radius=4, rotate_time=4, ransac_threshold=0.0075, geom_consistency=1)]
    prior = None
    for pi, extra in enumerate(passes):
        p = common.base_params(sc, N, seed=50, max_iterations=2, **extra)
        geom = bool(p.get("geom_consistency"))
        h = common.make_handle(gpu_pkg, sc, imgs, N, p, depths=deps if geom else None, prior=prior)
        o = common.make_oracle(ob, sc, imgs, N, p, depths=deps if geom else None, prior=prior)
        h.run()
        o.run()
        common.assert_state_equal(gpu_pkg, h, o, "N=%d pass %d" % (N, pi))
        planes, weak, views = h.download()
        prior = common.postprocess(planes, weak, views, p["depth_min"], p["depth_max"])
        h.close()
        o.close()
