"""Host-side constants the kernels rely on, checked without a GPU.

K3 (GenNeighbours, APD.cu:1911 / :1946) tests `dist / (depth_max - depth_min) < ransac_threshold` for dist >= 0.  The device
compares `dist < cut` instead; apd_ransac_distance_cut (csrc/apd_capi.hip) finds the cut with IEEE binary32 divisions.
The two predicates have to agree for every binary32 dist, or K3 would pick other neighbours than the reference."""
import ctypes as C

import numpy as np


def _cut(pkg, dmin, dmax, thr):
    L = C.CDLL(pkg.library_path())
    L.apd_ransac_distance_cut.argtypes = [C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float)]
    L.apd_ransac_distance_cut.restype = C.c_int
    out = C.c_float(0)
    ok = L.apd_ransac_distance_cut(dmin, dmax, thr, C.byref(out))
    return ok, np.float32(out.value)


def _neighbours(x, n):
    """The 2n+1 binary32 values around x >= 0 (clipped at 0)."""
    bits = np.float32(x).view(np.uint32).astype(np.int64)
    b = np.clip(bits + np.arange(-n, n + 1), 0, 0x7F7FFFFF).astype(np.uint32)
    return b.view(np.float32)


def test_cut_equals_the_division_for_every_float_near_it(pkg):
    rng = np.random.default_rng(5)
    cases = [(0.5, 3.0, 0.005), (1.0, 100.0, 0.005), (0.3, 1.7, 0.01), (2.4123, 17.913, 0.005), (0.0, 1.0, 0.005)]
    for _ in range(400):
        dmin = float(np.float32(rng.uniform(0.01, 50.0)))
        dmax = float(np.float32(dmin + rng.uniform(1e-3, 500.0)))
        cases.append((dmin, dmax, float(np.float32(10.0 ** rng.uniform(-5, 0)))))
    for dmin, dmax, thr in cases:
        ok, cut = _cut(pkg, dmin, dmax, thr)
        assert ok == 1
        dd = np.float32(dmax) - np.float32(dmin)
        t = np.float32(thr)
        xs = np.concatenate([_neighbours(cut, 2000), rng.uniform(0, 4 * float(cut) + 1e-6, 4000).astype(np.float32),
                             np.array([0.0, np.inf, np.nan, 1e-45, 3e38], np.float32)])
        with np.errstate(invalid="ignore", over="ignore"):
            by_division = (xs / dd) < t          # numpy divides binary32 by binary32 in binary32: the reference's test
            by_cut = xs < cut
            assert np.array_equal(by_division, by_cut), (dmin, dmax, thr, float(cut))
            # the second use (:1946) is the >= form
            assert np.array_equal((xs / dd) >= t, xs >= cut)


def test_parameters_without_a_cut_are_reported(pkg):
    assert _cut(pkg, 2.0, 2.0, 0.005)[0] == 0      # depth range 0: the division gives inf / NaN, no cut
    assert _cut(pkg, 3.0, 2.0, 0.005)[0] == 0      # negative range: the quotient is not monotone the same way
    assert _cut(pkg, 1.0, float("inf"), 0.005)[0] == 0
    ok, cut = _cut(pkg, 1.0, 2.0, 0.0)             # nothing is below a threshold of 0
    assert ok == 1 and cut == 0.0
    ok, cut = _cut(pkg, 1.0, 2.0, -1.0)
    assert ok == 1 and cut == 0.0
