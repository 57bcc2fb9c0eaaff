"""HIP path vs CPU oracle at the FULL sizes of BASELINE.json's configs, bit for bit, kernel by kernel.

The oracle cannot sweep 25.6 Mpix in a test's time budget (0.3 Mpix*iter/s on 128 threads), so it runs in region-of-interest
mode (orc_set_roi, oracle/apd_oracle.h): after every kernel of the HIP path -- launched over the whole image, with the
XCD-banded grids, LDS windows and 16/32-bit index arithmetic of the real size -- the same kernel runs on the CPU on a few
windows from the HIP path's own pre-kernel state, and every state array is compared inside the windows as raw bits.
Windows: the image corner (clamped patches, the 6-pixel UNKNOWN border), the centre, a window across the seam between the
tile bands of two XCDs, the bottom-right corner (last partial tile, HALF-launch rows), and for the APD configs windows
inside and on the rim of a textureless region, where K3's neighbour search walks hundreds of pixels (APD.cu:1750-1969)."""
import numpy as np
import pytest

import common

pytestmark = pytest.mark.gpu


def _schedule(iters, weak, tail=True):
    s = [(1, 0), (2, 0)] + ([(3, 0), (4, 0)] if weak else []) + [(5, 0)]
    for i in range(iters):
        s += [(6, i), (7, i), (8, i)] + ([(9, i), (10, i)] if weak else [])
    return s + ([(11, 0), (12, 0), (13, 0), (14, 0), (15, 0)] if tail else [])


def _fixed_windows(W, H, w=256, hh=160):
    """corner, centre, XCD band seam (tile rows are split into 8 contiguous bands, apd_sweep.h), bottom-right corner, top-right
    and bottom-left corners (narrower), and two windows at pseudo-random interior positions."""
    tiles_y = (H + 15) // 16
    seam_y = ((tiles_y + 7) // 8) * 16  # first tile row of the second band
    rng = np.random.RandomState(W * 7 + H)
    wins = [(0, 0, w, hh), (W // 2 - w // 2, H // 2 - hh // 2, W // 2 + w // 2, H // 2 + hh // 2),
            (W // 3, max(seam_y - hh // 2, 0), W // 3 + w, seam_y + hh // 2), (W - w, H - hh, W, H),
            (W - w // 2, 0, W, hh // 2), (0, H - hh // 2, w // 2, H)]
    for _ in range(2):
        x0, y0 = int(rng.randint(0, W - w)), int(rng.randint(0, H - hh))
        wins.append((x0, y0, x0 + w, y0 + hh))
    return [(max(x0, 0), max(y0, 0), min(x1, W), min(y1, H)) for (x0, y0, x1, y1) in wins]


def _weak_windows(weak, w=128, hh=80):
    """Two windows chosen from the WEAK map: around the WEAK pixel deepest inside a WEAK region (largest distance to a
    non-WEAK pixel along its row) and around the first WEAK pixel in raster order (a rim, mixed STRONG / WEAK)."""
    H, W = weak.shape
    is_weak = weak == 0
    ys, xs = np.nonzero(is_weak)
    assert len(ys) > 1000, "the scene must have WEAK pixels"
    # run length of WEAK pixels to the left, per row (vectorised), as a cheap "depth inside the region"
    run = np.zeros((H, W), np.int32)
    acc = np.zeros(H, np.int32)
    for x in range(W):
        acc = np.where(is_weak[:, x], acc + 1, 0)
        run[:, x] = acc
    cy, cx = np.unravel_index(int(np.argmax(run)), run.shape)
    cx = max(cx - int(run[cy, cx]) // 2, 0)
    wins = []
    far = (xs > 400) | (ys > 300)  # not inside the fixed corner window
    first = int(np.argmax(far)) if far.any() else 0
    for (px, py) in ((cx, cy), (int(xs[first]), int(ys[first]))):
        x0, y0 = min(max(px - w // 2, 0), W - w), min(max(py - hh // 2, 0), H - hh)
        wins.append((x0, y0, x0 + w, y0 + hh))
    return wins


def _scene(synth, W, H, N, textureless=0.0, **kw):
    import torch
    sc = synth.make_scene(W, H, N, seed=0, device=torch.device("cuda", 0), textureless=textureless, **kw)
    imgs = sc.images_numpy()
    del sc.images[:]
    torch.cuda.empty_cache()
    return sc, imgs


def test_configs1_office_fullres_8src_first_pass(gpu_pkg, ob, synth, record_property):
    """configs[1]: 6200x4130, 8 source views, FIRST_INIT: the whole pass (K1, K2, K5, three iterations of K6, K7, K8,
    K11..K15), i.e. the metric's sweep from random planes into the converged regime where the LDS windows serve most NCCs."""
    W, H, N = 6200, 4130, 8
    sc, imgs = _scene(synth, W, H, N)
    p = common.base_params(sc, N, max_iterations=3, seed=12345, weak_peak_radius=6)
    h = common.make_handle(gpu_pkg, sc, imgs, N, p)
    o = common.make_oracle(ob, sc, imgs, N, p)
    log = []
    n = common.fullsize_lockstep(gpu_pkg, h, o, _schedule(3, False), _fixed_windows(W, H), "configs[1]", log)
    record_property("kernels_compared", n)
    print("\n".join(log))
    assert n == 17
    h.close()
    o.close()


def test_configs2_pipes_fullres_10src_apd_pass(gpu_pkg, ob, synth, record_property):
    """configs[2]: 6200x4130, 10 source views, adaptive patch deformation on: the REFINE_INIT + APD pass (K1..K5, two
    iterations of K6..K10, K11..K15) on the WEAK map a complete FIRST_INIT pass leaves on a scene with 20 % textureless area."""
    W, H, N = 6200, 4130, 10
    sc, imgs = _scene(synth, W, H, N, textureless=0.2)
    p0 = common.base_params(sc, N, max_iterations=3, seed=12345, weak_peak_radius=6)
    h0 = common.make_handle(gpu_pkg, sc, imgs, N, p0)
    h0.run()
    planes, weak, views = h0.download()
    h0.close()
    prior = common.postprocess(planes, weak, views, p0["depth_min"], p0["depth_max"])
    weak_fraction = float((prior[2] == 0).mean())
    assert 0.05 < weak_fraction < 0.5, weak_fraction
    p = common.base_params(sc, N, max_iterations=2, seed=12346, state=1, use_APD=1, weak_peak_radius=6, rotate_time=4,
                           ransac_threshold=0.01 - 0.00125 * 3)
    h = common.make_handle(gpu_pkg, sc, imgs, N, p, prior=prior)
    o = common.make_oracle(ob, sc, imgs, N, p, prior=prior)
    assert h.weak_count == o.weak_count > 0
    assert np.array_equal(h.state(gpu_pkg.STATE_NEIGHBOURS_MAP), o.neighbours_map)
    windows = _fixed_windows(W, H, 160, 96) + _weak_windows(prior[2], 192, 128)
    log = []
    n = common.fullsize_lockstep(gpu_pkg, h, o, _schedule(2, True), windows, "configs[2]", log)
    print("\n".join(log))
    # the neighbour search must really have walked far: some accepted neighbour lies > 100 px from its pixel
    nb = h.state(gpu_pkg.STATE_NEIGHBOURS).astype(np.int32)
    valid = nb[:, 1:, 0] >= 0
    far = np.abs(nb[:, 1:, :] - nb[:, :1, :]).max(-1)[valid]
    record_property("weak_fraction", weak_fraction)
    record_property("max_neighbour_distance_px", int(far.max()))
    assert far.max() > 100
    assert n == 20
    h.close()
    o.close()


def _edge_windows(gt, count=4, w=160, hh=96):
    """Windows centred on depth steps of the reference view (slab edges: a source sees the background where the reference sees the
    slab): the `count` strongest steps at least 600 px apart, away from the frame border."""
    H, W = gt.shape
    step = np.zeros((H, W), np.float32)
    step[:, 1:] = np.abs(gt[:, 1:] - gt[:, :-1])
    step[1:, :] = np.maximum(step[1:, :], np.abs(gt[1:, :] - gt[:-1, :]))
    step[:200, :] = step[-200:, :] = 0
    step[:, :300] = step[:, -300:] = 0
    wins = []
    for _ in range(count):
        cy, cx = np.unravel_index(int(np.argmax(step)), step.shape)
        if step[cy, cx] < 0.05:
            break
        wins.append((cx - w // 2, cy - hh // 2, cx + w // 2, cy + hh // 2))
        step[max(cy - 600, 0):cy + 600, max(cx - 600, 0):cx + 600] = 0
    return wins


def test_configs2_pipes_fullres_10src_apd_pass_on_the_hard_scene(gpu_pkg, ob, synth, record_property):
    """configs[2] at full size on synth.HARD (the bench's *_hard lines): slabs in front of the planes, per-view gain / offset, sources
    aiming off the target.  REFINE_INIT + APD pass (K1..K5, two iterations of K6..K10, K11..K15) on the WEAK map a FIRST_INIT pass
    leaves; oracle windows across four depth steps (LDS windows staged over an occlusion edge, lanes of one wave landing on both
    sides of it in the sources), on the frame corners and inside a textureless region."""
    W, H, N = 6200, 4130, 10
    sc, imgs = _scene(synth, W, H, N, textureless=0.2, **synth.HARD)
    gt = sc.gt_depth.cpu().numpy()
    p0 = common.base_params(sc, N, max_iterations=3, seed=12345, weak_peak_radius=6)
    h0 = common.make_handle(gpu_pkg, sc, imgs, N, p0)
    h0.run()
    planes, weak, views = h0.download()
    h0.close()
    prior = common.postprocess(planes, weak, views, p0["depth_min"], p0["depth_max"])
    weak_fraction = float((prior[2] == 0).mean())
    good = float(((np.abs(planes[..., 3] - gt) / gt)[8:-8, 8:-8] < 0.01).mean())
    record_property("weak_fraction", weak_fraction)
    record_property("within_1pct_after_first_pass", good)
    assert 0.05 < weak_fraction < 0.7, weak_fraction
    assert 0.5 < good < 0.999, good   # a hard scene: the first pass must NOT converge everywhere
    p = common.base_params(sc, N, max_iterations=2, seed=12346, state=1, use_APD=1, weak_peak_radius=6, rotate_time=4,
                           ransac_threshold=0.01 - 0.00125 * 3)
    h = common.make_handle(gpu_pkg, sc, imgs, N, p, prior=prior)
    o = common.make_oracle(ob, sc, imgs, N, p, prior=prior)
    assert h.weak_count == o.weak_count > 0
    edges = _edge_windows(gt)
    assert len(edges) >= 3, edges
    windows = edges + _fixed_windows(W, H, 160, 96)[:4] + _weak_windows(prior[2], 160, 96)
    log = []
    n = common.fullsize_lockstep(gpu_pkg, h, o, _schedule(2, True), windows, "configs[2] hard", log)
    print("\n".join(log))
    assert n == 20
    h.close()
    o.close()


def test_configs4_synthetic_4096x3072_16src(gpu_pkg, ob, synth):
    """configs[4]: 4096x3072, 16 source views (the NMAX = 16 kernels at full size), FIRST_INIT, two iterations + the
    post-loop kernels."""
    W, H, N = 4096, 3072, 16
    sc, imgs = _scene(synth, W, H, N)
    p = common.base_params(sc, N, max_iterations=2, seed=777, weak_peak_radius=6)
    h = common.make_handle(gpu_pkg, sc, imgs, N, p)
    o = common.make_oracle(ob, sc, imgs, N, p)
    log = []
    n = common.fullsize_lockstep(gpu_pkg, h, o, _schedule(2, False), _fixed_windows(W, H, 192, 128), "configs[4]", log)
    print("\n".join(log))
    assert n == 14
    h.close()
    o.close()


def test_configs2_geometric_apd_pass_fullres(gpu_pkg, ob, synth):
    """The third pass kind at the full size of configs[2]: REFINE_ITER with the geometric-consistency term (depth maps of all
    eleven views resident, APD.cu:760-772) and adaptive patch deformation, 10 source views, on the state a FIRST_INIT pass
    leaves: K1..K5, one iteration of K6..K10, K11..K15."""
    W, H, N = 6200, 4130, 10
    sc, imgs = _scene(synth, W, H, N, textureless=0.2)
    p0 = common.base_params(sc, N, max_iterations=2, seed=4321, weak_peak_radius=6)
    h0 = common.make_handle(gpu_pkg, sc, imgs, N, p0)
    h0.run()
    planes, weak, views = h0.download()
    h0.close()
    prior = common.postprocess(planes, weak, views, p0["depth_min"], p0["depth_max"])
    # depth maps "of the previous pass" for every view: the reference view's own estimate and smooth analytic maps with holes
    deps = [np.ascontiguousarray(prior[0][..., 3])] + common.fake_depth_maps(W, H, N)
    p = common.base_params(sc, N, max_iterations=1, seed=4322, state=2, use_APD=1, geom_consistency=1, weak_peak_radius=4,
                           rotate_time=4, ransac_threshold=0.01 - 0.00125 * 3)
    h = common.make_handle(gpu_pkg, sc, imgs, N, p, depths=deps, prior=prior)
    o = common.make_oracle(ob, sc, imgs, N, p, depths=deps, prior=prior)
    assert h.weak_count == o.weak_count > 0
    windows = _fixed_windows(W, H, 128, 80)[:5] + _weak_windows(prior[2], 128, 96)
    n = common.fullsize_lockstep(gpu_pkg, h, o, _schedule(1, True), windows, "configs[2] geometric")
    assert n == 15
    h.close()
    o.close()


def test_float_images_at_a_pyramid_level_of_configs1(gpu_pkg, ob, synth):
    """Level 1 of the configs[1] pyramid: 3100 x 2065, non-integer grey values (what cv::resize leaves, APD.cpp:474), i.e. the
    float texel-quad path with binary32 LDS windows, through a REFINE_INIT + APD pass on a FIRST_INIT prior."""
    W, H, N = 3100, 2065, 8
    sc, imgs = _scene(synth, W, H, N, textureless=0.2)
    imgs = [(im * np.float32(0.75) + np.float32(13.37)).astype(np.float32) for im in imgs]
    p0 = common.base_params(sc, N, max_iterations=2, seed=99, weak_peak_radius=6)
    h0 = common.make_handle(gpu_pkg, sc, imgs, N, p0)
    o0 = common.make_oracle(ob, sc, imgs, N, p0)
    wins = _fixed_windows(W, H, 160, 96)
    n0 = common.fullsize_lockstep(gpu_pkg, h0, o0, _schedule(2, False), wins, "float level FIRST_INIT")
    assert n0 == 14
    planes, weak, views = h0.download()
    h0.close()
    o0.close()
    prior = common.postprocess(planes, weak, views, p0["depth_min"], p0["depth_max"])
    p = common.base_params(sc, N, max_iterations=1, seed=100, state=1, use_APD=1, weak_peak_radius=6, rotate_time=2, ransac_threshold=0.00875)
    h = common.make_handle(gpu_pkg, sc, imgs, N, p, prior=prior)
    o = common.make_oracle(ob, sc, imgs, N, p, prior=prior)
    assert h.weak_count == o.weak_count > 0
    n = common.fullsize_lockstep(gpu_pkg, h, o, _schedule(1, True), wins[:5] + _weak_windows(prior[2], 128, 96), "float level APD")
    assert n == 15
    h.close()
    o.close()


def test_configs0_office_halfres_2src_3iter(gpu_pkg, ob, synth, record_property):
    """configs[0]: ETH3D office at half resolution (scale_factor 2), 2 source views, 3 iterations -- the shape BASELINE.json runs on
    the reference's CPU path.  The full-size 8-bit frames are resampled as the reference does for scale_size = 2 (cv::resize
    INTER_LINEAR on the float image, intrinsics scaled by the rounded size ratio, APD.cpp:464-488), so the level holds
    non-integer grey values: 3100 x 2065, the float texel-quad path, the N = 2 instantiation of every sweep kernel at a size
    where the XCD-banded grid and the LDS windows are the real ones.  Whole FIRST_INIT pass, main.cpp:171-190."""
    from apd_mvs_amd import pipeline
    W0, H0, N = 6200, 4130, 2
    sc, imgs0 = _scene(synth, W0, H0, N)
    W, H = W0 // 2, H0 // 2
    imgs = [pipeline.resize_linear(im, W, H) for im in imgs0]
    assert any(np.any(im != np.round(im)) for im in imgs), "a resampled level holds non-integer grey values"
    del imgs0
    sx, sy = np.float32(W) / np.float32(W0), np.float32(H) / np.float32(H0)
    for k in range(N + 1):   # APD.cpp:480-487
        K = sc.K[k].copy()
        K[0], K[2], K[4], K[5] = K[0] * sx, K[2] * sx, K[4] * sy, K[5] * sy
        sc.K[k] = K
    sc.width, sc.height = W, H
    p = common.base_params(sc, N, max_iterations=3, seed=12345, weak_peak_radius=6)
    h = common.make_handle(gpu_pkg, sc, imgs, N, p)
    o = common.make_oracle(ob, sc, imgs, N, p)
    log = []
    n = common.fullsize_lockstep(gpu_pkg, h, o, _schedule(3, False), _fixed_windows(W, H), "configs[0]", log)
    print("\n".join(log))
    record_property("kernels_compared", n)
    assert n == 17
    # the pass must have done its job on this shape: depth within 1 % of the analytic depth (rendered at full size: every other pixel)
    d = h.state(gpu_pkg.STATE_PLANES)[..., 3]
    gt = sc.gt_depth.cpu().numpy()[::2, ::2][:H, :W]
    good = float(((np.abs(d - gt) / gt)[8:-8, 8:-8] < 0.02).mean())
    record_property("within_2pct_depth", good)
    assert good > 0.9, good
    h.close()
    o.close()


def test_configs1_full_frame_oracle_iteration2(gpu_pkg, ob, synth, record_property):
    """configs[1], ONE iteration over the WHOLE frame against the oracle (no region of interest): after K1, K2, K5 and
    iterations 0 and 1 on the HIP path, its state is loaded into the oracle and K6, K7, K8 of iteration 2 -- the converged regime,
    LDS windows on, compacted refinement, early-outs -- run on both over all 25.6 Mpix; every state array of every pixel is compared
    as raw bits.  The window tests above cover 1.3 % of the frame after every kernel; this one covers every wave's footprint once
    (APD.cu:1547-1585: the launch visits every pixel of a colour)."""
    import time
    W, H, N = 6200, 4130, 8
    sc, imgs = _scene(synth, W, H, N)
    p = common.base_params(sc, N, max_iterations=3, seed=12345, weak_peak_radius=6)
    h = common.make_handle(gpu_pkg, sc, imgs, N, p)
    o = common.make_oracle(ob, sc, imgs, N, p)
    for kid, it in _schedule(2, False, tail=False):
        h.run_kernel(kid, it)
    t0 = time.perf_counter()
    log = []
    n = common.fullsize_lockstep(gpu_pkg, h, o, [(6, 2), (7, 2), (8, 2)], [(0, 0, W, H)], "configs[1] full frame", log)
    print("\n".join(log))
    record_property("oracle_full_frame_s", round(time.perf_counter() - t0, 1))
    record_property("pixels_compared", W * H)
    assert n == 3
    h.close()
    o.close()


def test_configs3_frame_every_kernel_over_the_whole_frame(gpu_pkg, ob, synth, record_property):
    """configs[3]'s frame (1920 x 1080, 10 source views) with NO region of interest: every kernel of a REFINE_INIT + APD pass (K1..K5, two
    iterations of K6..K10, K11..K15: 20 kernels) and of the REFINE_ITER + APD + geometric pass that follows it (K1..K5, one iteration,
    K11..K15: 15 kernels) runs on the HIP path and on the oracle over all 2.07 Mpix from the same pre-kernel state, and every state array of
    every pixel is compared as raw bits after every kernel.  The 6200 x 4130 tests compare windows (1.3 % of the frame) after every kernel
    and one strong iteration over the whole frame; this one puts K3, K8, K9/K10, K14 and K15 -- list compaction in supertile order, XCD
    chunking of the WEAK lists, the chunk-major K14 with its pair walk (ten sources) -- under a whole-frame comparison too."""
    import time
    W, H, N = 1920, 1080, 10
    sc, imgs = _scene(synth, W, H, N, textureless=0.2)
    p0 = common.base_params(sc, N, max_iterations=3, seed=2024, weak_peak_radius=6)
    h0 = common.make_handle(gpu_pkg, sc, imgs, N, p0)
    h0.run()
    planes, weak, views = h0.download()
    h0.close()
    prior = common.postprocess(planes, weak, views, p0["depth_min"], p0["depth_max"])
    weak_fraction = float((prior[2] == 0).mean())
    assert 0.05 < weak_fraction < 0.5, weak_fraction
    t0 = time.perf_counter()
    p = common.base_params(sc, N, max_iterations=2, seed=2025, state=1, use_APD=1, weak_peak_radius=6, rotate_time=4, ransac_threshold=0.01 - 0.00125 * 3)
    h = common.make_handle(gpu_pkg, sc, imgs, N, p, prior=prior)
    o = common.make_oracle(ob, sc, imgs, N, p, prior=prior)
    assert h.weak_count == o.weak_count > 10000
    log = []
    n = common.fullsize_lockstep(gpu_pkg, h, o, _schedule(2, True), [(0, 0, W, H)], "configs[3] frame, APD pass, whole frame", log)
    assert n == 20
    planes, weak, views = h.download()
    h.close()
    o.close()
    prior = common.postprocess(planes, weak, views, p["depth_min"], p["depth_max"])
    deps = [np.ascontiguousarray(prior[0][..., 3])] + common.fake_depth_maps(W, H, N)
    pg = common.base_params(sc, N, max_iterations=1, seed=2026, state=2, use_APD=1, geom_consistency=1, weak_peak_radius=4, rotate_time=4,
                            ransac_threshold=0.01 - 0.00125 * 3)
    h = common.make_handle(gpu_pkg, sc, imgs, N, pg, depths=deps, prior=prior)
    o = common.make_oracle(ob, sc, imgs, N, pg, depths=deps, prior=prior)
    assert h.weak_count == o.weak_count > 0
    n = common.fullsize_lockstep(gpu_pkg, h, o, _schedule(1, True), [(0, 0, W, H)], "configs[3] frame, geometric pass, whole frame", log)
    assert n == 15
    print("\n".join(log))
    record_property("whole_frame_kernels_compared", 35)
    record_property("weak_fraction", weak_fraction)
    record_property("seconds", round(time.perf_counter() - t0, 1))
    h.close()
    o.close()
