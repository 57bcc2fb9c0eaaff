"""Error behaviour of the C ABI: the reference exit()s (APD.cpp:321, 430); the library returns codes."""
import numpy as np
import pytest

import common

pytestmark = pytest.mark.gpu


def test_too_many_images(gpu_pkg, synth):
    W, H = 32, 32
    sc, imgs = common.scene_inputs(synth, W, H, 1)
    cams = [gpu_pkg.make_camera(sc.K[0], sc.R[0], sc.t[0], W, H, 1, 4)] * 33
    h = gpu_pkg.Handle(W, H, gpu_pkg.default_params())
    with pytest.raises(gpu_pkg.ApdError, match="so much images"):
        h.upload_views(cams, [imgs[0]] * 33)
    h.close()


def test_run_before_upload(gpu_pkg):
    h = gpu_pkg.Handle(32, 32, gpu_pkg.default_params())
    with pytest.raises(gpu_pkg.ApdError, match="apd_upload_views"):
        h.run()
    h.close()


def test_refine_state_needs_prior(gpu_pkg, synth):
    W, H, N = 32, 32, 2
    sc, imgs = common.scene_inputs(synth, W, H, N)
    h = common.make_handle(gpu_pkg, sc, imgs, N, common.base_params(sc, N, state=gpu_pkg.REFINE_INIT, use_APD=1))
    with pytest.raises(gpu_pkg.ApdError, match="apd_upload_prior"):
        h.run()
    h.close()


def test_geometric_pass_needs_depth_maps(gpu_pkg, synth):
    W, H, N = 32, 32, 2
    sc, imgs = common.scene_inputs(synth, W, H, N)
    with pytest.raises(gpu_pkg.ApdError, match="depth maps"):
        common.make_handle(gpu_pkg, sc, imgs, N, common.base_params(sc, N, geom_consistency=1))


def test_unsupported_patch_geometry(gpu_pkg):
    with pytest.raises(gpu_pkg.ApdError, match="patch geometry"):
        gpu_pkg.Handle(32, 32, gpu_pkg.default_params(strong_radius=7))


def test_camera_size_mismatch(gpu_pkg, synth):
    W, H = 32, 32
    sc, imgs = common.scene_inputs(synth, W, H, 1)
    cams = [gpu_pkg.make_camera(sc.K[i], sc.R[i], sc.t[i], W + 1, H, 1, 4) for i in range(2)]
    h = gpu_pkg.Handle(W, H, gpu_pkg.default_params())
    with pytest.raises(gpu_pkg.ApdError, match="camera"):
        h.upload_views(cams, imgs)
    h.close()


def test_profile_counters(gpu_pkg, synth):
    W, H, N = 64, 48, 2
    sc, imgs = common.scene_inputs(synth, W, H, N)
    h = common.make_handle(gpu_pkg, sc, imgs, N, common.base_params(sc, N))
    for k in (1, 2, 5):
        h.run_kernel(k)
    h.profile_enable(True)
    h.profile_reset()
    h.run_sweeps(0, 2)
    prof = h.profile()
    assert prof[gpu_pkg.K6][1] == 2 and prof[gpu_pkg.K7][1] == 2 and prof[gpu_pkg.K8][1] == 2
    assert prof[gpu_pkg.K6][0] > 0
    assert gpu_pkg.K9 not in prof  # no WEAK pixel -> weak kernels are not launched
    h.close()


def test_weak_lists_cannot_be_overrun_by_a_second_pass(gpu_pkg, ob, synth):
    """the WEAK lists, the neighbour table and its index map are sized by the weak map apd_upload_prior loads.
    K14 rewrites that map (usually with more WEAK pixels) and apd_upload_state can too; the kernels that walk the lists must
    then be refused (APD_ERR_STATE), not run past the end of them -- and a new apd_upload_prior / apd_reset re-arms the
    handle with the oracle's bits."""
    W, H, N = 96, 72, 4
    sc, imgs = common.scene_inputs(synth, W, H, N, seed=3, textureless=0.25)
    p0 = common.base_params(sc, N, seed=11, state=0, use_APD=0, weak_peak_radius=6)
    h = common.make_handle(gpu_pkg, sc, imgs, N, p0)
    h.run()
    planes, weak, views = h.download()
    with pytest.raises(gpu_pkg.ApdError, match="apd_reset or apd_upload_prior"):
        h.run()                      # a second pass on the same object
    for kid in (3, 8, 9, 10):
        with pytest.raises(gpu_pkg.ApdError, match="rewritten"):
            h.run_kernel(kid)        # K14 followed by a list walker
    prior = common.postprocess(planes, weak, views, p0["depth_min"], p0["depth_max"])
    assert (prior[2] == 0).sum() > 50
    p1 = common.base_params(sc, N, seed=11, state=1, use_APD=1, weak_peak_radius=6)
    # the same handle re-armed == a fresh oracle, through a whole APD pass whose K14 grows the WEAK set again
    h.reset(gpu_pkg.default_params(**p1))
    cams = [gpu_pkg.make_camera(sc.K[i], sc.R[i], sc.t[i], W, H, sc.depth_min, sc.depth_max) for i in range(N + 1)]
    h.upload_views(cams, imgs)
    h.upload_prior(*prior)
    o = common.make_oracle(ob, sc, imgs, N, p1, prior=prior)
    h.run()
    o.run()
    common.assert_state_equal(gpu_pkg, h, o, "APD pass on a re-armed handle")
    # a grown map through apd_upload_state: refused as well, until the prior is uploaded again
    grown = np.zeros((H, W), np.uint8)   # every pixel WEAK: twice what any list has room for
    h.set_state(gpu_pkg.STATE_WEAK_INFO, grown)
    with pytest.raises(gpu_pkg.ApdError, match="rewritten"):
        h.run_kernel(9)
    h.upload_prior(prior[0], prior[1], grown)
    assert h.weak_count == W * H
    for kid in (1, 2, 3, 4, 5):
        h.run_kernel(kid)
    h.run_sweeps(0, 1)                   # every list at its maximum length
    h.synchronize()
    h.close()
    o.close()
