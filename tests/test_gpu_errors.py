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
