"""Known-answer tests of the oracle's building blocks (the reference has no tests of its own:
these anchors are analytic facts about the algorithm of APD.cu, not reference outputs)."""
import ctypes as C

import numpy as np
import pytest

import common


def _f4(v):
    return (C.c_float * 4)(*[float(x) for x in v])


def test_sin_cos_exp_accuracy(ob):
    L = ob.lib()
    xs = np.linspace(-0.0314, 0.0314, 401).astype(np.float32)
    for x in xs:
        assert abs(L.orc_sinf(float(x)) - np.sin(np.float64(x))) <= 2 * np.spacing(np.float32(abs(np.sin(np.float64(x))) + 1e-30))
        assert abs(L.orc_cosf(float(x)) - np.cos(np.float64(x))) <= 2 * np.spacing(np.float32(1.0))
    for x in np.linspace(-0.78, 0.78, 101).astype(np.float32):
        assert abs(L.orc_sinf(float(x)) - np.sin(np.float64(x))) < 3e-7
        assert abs(L.orc_cosf(float(x)) - np.cos(np.float64(x))) < 3e-7
    for x in np.concatenate([np.linspace(-30, 0, 601), [-86.9, -50.0, -1e-8, 0.0]]).astype(np.float32):
        ref = np.exp(np.float64(x))
        assert abs(L.orc_expf(float(x)) - ref) <= 4e-7 * ref + 1e-45
    assert L.orc_expf(0.0) == 1.0
    assert L.orc_expf(-100.0) == 0.0
    assert np.isnan(L.orc_expf(float("nan")))


def test_view_selection_threshold_values(ob):
    """cost_threshold = 0.8*exp(-iter^2/90) (APD.cu:1225) for the iterations the schedule uses."""
    L = ob.lib()
    for it in range(0, 9):
        thr = np.float32(0.8 * np.float64(np.float32(L.orc_expf(float(np.float32(it * it) / np.float32(-90.0))))))
        assert abs(float(thr) - 0.8 * np.exp(-it * it / 90.0)) < 2e-7


def test_sampler_texel_centres_half_texels_and_clamp(ob):
    L = ob.lib()
    W, H = 7, 5
    img = (np.arange(W * H, dtype=np.float32) * 3 + 1).reshape(H, W)
    p = img.ctypes.data_as(C.POINTER(C.c_float))
    for y in range(H):
        for x in range(W):
            assert L.orc_sample_bilinear(p, W, H, float(x), float(y)) == img[y, x]
    assert L.orc_sample_bilinear(p, W, H, 2.5, 1.0) == (img[1, 2] + img[1, 3]) / 2
    assert L.orc_sample_bilinear(p, W, H, 2.0, 1.5) == (img[1, 2] + img[2, 2]) / 2
    assert L.orc_sample_bilinear(p, W, H, 2.5, 1.5) == (img[1, 2] + img[1, 3] + img[2, 2] + img[2, 3]) / 4
    # clamp-to-edge (cudaAddressModeWrap degrades to clamp for unnormalised coordinates, APD.cpp:598-602)
    assert L.orc_sample_bilinear(p, W, H, -3.7, -9.0) == img[0, 0]
    assert L.orc_sample_bilinear(p, W, H, 100.0, 2.0) == img[2, W - 1]
    assert L.orc_sample_bilinear(p, W, H, 3.0, 1e9) == img[H - 1, 3]
    assert L.orc_sample_bilinear(p, W, H, -0.5, 0.0) == img[0, 0]
    assert L.orc_sample_bilinear(p, W, H, W - 0.5, 0.0) == img[0, W - 1]
    assert np.isnan(L.orc_sample_bilinear(p, W, H, float("nan"), 1.0))


def test_study_knob_for_fixed_point_weights_is_off_by_default_and_quantises_to_1_256(ob):
    """orc_set_study_weights_q8 (tools/sampler_sensitivity.py): the CUDA texture unit's 8 fractional weight bits.  Not part of
    the contract: off unless a study switches it on, and it must be switched off again."""
    L = ob.lib()
    assert L.orc_get_study_weights_q8() == 0
    W, H = 4, 3
    img = np.array([[0, 256, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0]], np.float32)
    p = img.ctypes.data_as(C.POINTER(C.c_float))
    x = 0.3  # weight 0.3 -> contract: 0.3 * 256; 8-bit weight: round(76.8) / 256 * 256 = 77
    assert abs(L.orc_sample_bilinear(p, W, H, x, 0.0) - np.float32(x) * 256) < 1e-4
    L.orc_set_study_weights_q8(1)
    try:
        assert L.orc_sample_bilinear(p, W, H, x, 0.0) == 77.0
        assert L.orc_sample_bilinear(p, W, H, 0.5, 0.0) == 128.0 and L.orc_sample_bilinear(p, W, H, 1.0, 0.0) == 256.0
    finally:
        L.orc_set_study_weights_q8(0)
    assert L.orc_get_study_weights_q8() == 0


def _cam(ob, f, cx, cy, R, t, W, H):
    return ob.make_camera([f, 0, cx, 0, f, cy, 0, 0, 1], R, t, W, H, 1.0, 4.0)


def test_homography_fronto_parallel_translation(ob):
    """Identity rotations, baseline b along x, plane z = d: H maps (x, y) -> (x - f*b/d, y)."""
    L = ob.lib()
    f, cx, cy, W, H = 100.0, 32.0, 24.0, 64, 48
    ref = _cam(ob, f, cx, cy, np.eye(3), [0, 0, 0], W, H)
    b = 0.1
    src = _cam(ob, f, cx, cy, np.eye(3), [-b, 0, 0], W, H)  # t = -R C, C = (b, 0, 0)
    d = 2.0
    plane = _f4([0, 0, -1, d])  # n.X + w = 0 with n = (0,0,-1), w = d  ->  z = d
    Hm = (C.c_float * 9)()
    L.orc_homography(C.byref(ref), C.byref(src), plane, Hm)
    Hm = np.array(list(Hm), np.float64).reshape(3, 3)
    for (x, y) in [(0, 0), (10, 7), (63, 47)]:
        q = Hm @ np.array([x, y, 1.0])
        assert abs(q[0] / q[2] - (x - f * b / d)) < 1e-3
        assert abs(q[1] / q[2] - y) < 1e-3


def test_homography_matches_float64_formula(ob, synth):
    """H = K_s (R_rel - t_rel n^T / d) K_r^-1 (APD.cu:303-363) on random rotated cameras."""
    L = ob.lib()
    sc = synth.make_scene(64, 48, 4, seed=5)
    cams = [ob.make_camera(sc.K[i], sc.R[i], sc.t[i], 64, 48, 1, 4) for i in range(5)]
    rng = np.random.RandomState(0)
    for src in range(1, 5):
        for _ in range(20):
            n = rng.normal(size=3)
            n /= np.linalg.norm(n)
            w = rng.uniform(1.0, 4.0)
            Hm = (C.c_float * 9)()
            L.orc_homography(C.byref(cams[0]), C.byref(cams[src]), _f4([n[0], n[1], n[2], w]), Hm)
            Hm = np.array(list(Hm), np.float64).reshape(3, 3)
            Kr, Ks = sc.K[0].reshape(3, 3).astype(np.float64), sc.K[src].reshape(3, 3).astype(np.float64)
            Rr, Rs = sc.R[0].reshape(3, 3).astype(np.float64), sc.R[src].reshape(3, 3).astype(np.float64)
            Cr, Cs = -Rr.T @ sc.t[0].astype(np.float64), -Rs.T @ sc.t[src].astype(np.float64)
            Rrel = Rs @ Rr.T
            trel = Rs @ (Cr - Cs)
            ref = Ks @ (Rrel - np.outer(trel, n) / w) @ np.linalg.inv(Kr)
            assert np.allclose(Hm, ref, rtol=2e-5, atol=2e-4)


def test_depth_distance_round_trip(ob, synth):
    L = ob.lib()
    sc = synth.make_scene(64, 48, 1, seed=2)
    cam = ob.make_camera(sc.K[0], sc.R[0], sc.t[0], 64, 48, 1, 4)
    rng = np.random.RandomState(1)
    for _ in range(200):
        x, y = int(rng.randint(0, 64)), int(rng.randint(0, 48))
        n = rng.normal(size=3)
        n[2] = -abs(n[2]) - 0.5
        n /= np.linalg.norm(n)
        depth = float(rng.uniform(0.7, 4.5))
        w = L.orc_distance_to_origin(C.byref(cam), x, y, depth, _f4([n[0], n[1], n[2], 0]))
        back = L.orc_depth_from_plane(C.byref(cam), _f4([n[0], n[1], n[2], w]), x, y)
        assert abs(back - depth) < 2e-5 * depth


def _oracle_for(ob, synth, W=64, H=48, N=3, **kw):
    sc, imgs = common.scene_inputs(synth, W, H, N)
    o = common.make_oracle(ob, sc, imgs, N, common.base_params(sc, N, **kw))
    return sc, imgs, o


def test_ncc_true_plane_is_cheap_wrong_plane_is_not(ob, synth):
    sc, imgs, o = _oracle_for(ob, synth)
    gt = sc.gt_depth.numpy()
    L = ob.lib()
    cam = ob.make_camera(sc.K[0], sc.R[0], sc.t[0], 64, 48, 1, 4)
    # true plane of the scene (synth.py): n.P + d = 0 with n = (-0.15,-0.10,1), d = -2.0 where it is the nearer one
    n = np.array([-0.15, -0.10, 1.0])
    nn = -n / np.linalg.norm(n)  # facing the camera
    good, bad = [], []
    for (x, y) in [(20, 20), (30, 25), (40, 30), (25, 35)]:
        w = L.orc_distance_to_origin(C.byref(cam), x, y, float(gt[y, x]), _f4([nn[0], nn[1], nn[2], 0]))
        for src in (1, 2, 3):
            good.append(o.ncc_old(x, y, src, [nn[0], nn[1], nn[2], w]))
            w_bad = L.orc_distance_to_origin(C.byref(cam), x, y, float(gt[y, x]) * 1.3, _f4([nn[0], nn[1], nn[2], 0]))
            bad.append(o.ncc_old(x, y, src, [nn[0], nn[1], nn[2], w_bad]))
    assert max(good) < 0.15, good
    # short-baseline views barely see a 30 % depth error; on average the wrong plane is far worse
    assert np.mean(bad) > 0.2 and np.mean(bad) > 4 * np.mean(good), (good, bad)
    assert all(0.0 <= c <= 2.0 for c in good + bad)


def test_ncc_out_of_image_centre_costs_two(ob, synth):
    sc, imgs, o = _oracle_for(ob, synth)
    # a plane almost at the camera: the projection of the centre leaves the source image (APD.cu:546-548)
    assert o.ncc_old(5, 5, 1, [0.0, 0.0, -1.0, 0.02]) == 2.0


def test_ncc_zero_variance_patch_costs_two(ob, synth):
    sc, imgs = common.scene_inputs(synth, 64, 48, 2)
    imgs = [np.full_like(im, 77.0) for im in imgs]
    o = common.make_oracle(ob, sc, imgs, 2, common.base_params(sc, 2))
    assert o.ncc_old(30, 20, 1, [0.0, 0.0, -1.0, 2.0]) == 2.0  # var < 1e-5 (APD.cu:602-605)


def test_identical_images_identity_pose_cost_zero(ob, synth):
    """Source == reference with the same camera: every plane warps onto itself, NCC cost == 0."""
    sc, imgs = common.scene_inputs(synth, 64, 48, 1)
    W, H = 64, 48
    cams = [ob.make_camera(sc.K[0], sc.R[0], sc.t[0], W, H, 1, 4)] * 2
    p = ob.default_params(**common.base_params(sc, 1))
    o = ob.Oracle(W, H, p, cams, [imgs[0], imgs[0]])
    for (x, y) in [(10, 10), (32, 24), (50, 40)]:
        assert o.ncc_old(x, y, 1, [0.1, -0.2, -0.97, 2.0]) < 1e-6


def test_median_filter_is_the_median_of_the_21_tap_stencil(ob, synth):
    """K12/K13 (CheckerboardFilterStrong, APD.cu:1604-1714): away from the border and with every tap STRONG, the depth of a pixel
    becomes the median of itself and twenty taps -- (0, +-1), (0, +-3), (0, +-5), (+-1, 0), (+-3, 0), (+-5, 0), (+-2, +-1), (+-1, +-2) --
    all of the other colour, so black pixels read the pre-filter state and red pixels the black result; pixels whose cost is
    below 0.001 are left alone (:1638)."""
    W, H, N = 40, 32, 2
    sc, imgs = common.scene_inputs(synth, W, H, N)
    o = common.make_oracle(ob, sc, imgs, N, common.base_params(sc, N))
    rng = np.random.RandomState(4)
    o.planes[..., 3] = rng.uniform(1.0, 3.0, (H, W)).astype(np.float32)
    o.costs[...] = 1.0
    o.costs[10, 10] = 0.0005
    o.costs[11, 10] = 0.0005
    o.weak_info[...] = ob.STRONG
    taps = [(0, -1), (0, -3), (0, -5), (0, 1), (0, 3), (0, 5), (-1, 0), (-3, 0), (-5, 0), (1, 0), (3, 0), (5, 0),
            (2, -1), (2, 1), (-2, -1), (-2, 1), (-1, -2), (1, -2), (-1, 2), (1, 2)]
    for kid, colour in ((12, 0), (13, 1)):
        before = o.planes[..., 3].copy()
        o.run_kernel(kid)
        after = o.planes[..., 3]
        checked = 0
        for y in range(6, H - 6):
            for x in range(6, W - 6):
                # colour of a pixel as the checkerboard launches see it: black = (x + y) even
                if (x + y) % 2 != colour:
                    assert after[y, x] == before[y, x]
                    continue
                if o.costs[y, x] < 0.001:
                    assert after[y, x] == before[y, x]
                    continue
                vals = np.array([before[y, x]] + [before[y + dy, x + dx] for dx, dy in taps], np.float32)
                assert after[y, x] == np.sort(vals)[10], (kid, x, y)
                checked += 1
        assert checked > 100
    o.close()
