"""The C-ABI shared library loads without a GPU and exports every symbol include/apd_mi355x.h declares
(no compute calls here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "apd_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(apd_[a-z_0-9]+)\s*\(", text)
    return sorted(set(names))


def test_header_declares_the_expected_entry_points():
    names = _declared_functions()
    for must in ("apd_create", "apd_destroy", "apd_upload_views", "apd_upload_prior", "apd_run", "apd_run_kernel",
                 "apd_run_sweeps", "apd_download", "apd_export_depth_normal_device", "apd_profile_get", "apd_reset",
                 "apd_fuse_views", "apd_fusion_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol(pkg):
    path = pkg.library_path()
    assert os.path.exists(path), "run __graft_entry__.build() first"
    L = C.CDLL(path)
    missing = [n for n in _declared_functions() if not hasattr(L, n)]
    assert not missing, missing


def test_struct_layouts_match_the_reference(pkg, ob):
    # Camera is 112 bytes in the reference: 9+9+3+3 floats, 2 ints, 2 floats
    assert C.sizeof(pkg.Camera) == 112
    assert C.sizeof(ob.Camera) == 112
    p = pkg.default_params()
    # defaults of PatchMatchParams, main.h:75-94
    assert (p.max_iterations, p.top_k, p.strong_radius, p.strong_increment, p.weak_radius, p.weak_increment) == (3, 4, 5, 2, 5, 5)
    assert (p.weak_peak_radius, p.rotate_time) == (2, 4)
    assert abs(p.ransac_threshold - 0.005) < 1e-9 and abs(p.geom_factor - 0.2) < 1e-7
    assert abs(p.sigma_spatial - 5.0) < 1e-9 and abs(p.sigma_color - 3.0) < 1e-9


def test_no_gpu_means_loud_failure_not_fallback(pkg):
    """Without a device the product must raise, never compute on the CPU."""
    if pkg.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(pkg.ApdError):
        pkg.Handle(32, 32, pkg.default_params(), device=0)


def test_no_gpu_means_the_fusion_fails_loudly_too(pkg, tmp_path):
    """apd_fuse_views has no host fallback either: without a device it returns an error and writes nothing."""
    import numpy as np
    if pkg.device_count() > 0:
        pytest.skip("a GPU is visible")
    L = pkg.lib()
    L.apd_fusion_last_error.restype = C.c_char_p
    cams = (pkg.Camera * 2)()
    img = np.zeros((4, 4), np.float32)
    nrm = np.zeros((4, 4, 3), np.float32)
    weak = np.zeros((4, 4), np.uint8)
    fptr = (C.c_void_p * 2)(img.ctypes.data, img.ctypes.data)
    nptr = (C.c_void_p * 2)(nrm.ctypes.data, nrm.ctypes.data)
    wptr = (C.c_void_p * 2)(weak.ctypes.data, weak.ctypes.data)
    rows, cols = (C.c_int * 2)(4, 4), (C.c_int * 2)(4, 4)
    offs, idx = (C.c_int * 3)(0, 1, 2), (C.c_int * 2)(1, 0)
    n = C.c_longlong(0)
    out = tmp_path / "x.ply"
    st = L.apd_fuse_views(0, 2, cams, fptr, 1, fptr, nptr, wptr, None, rows, cols, offs, idx, 0, str(out).encode(), C.byref(n))
    assert st != 0 and not out.exists() and L.apd_fusion_last_error()
