"""The oracle's region-of-interest mode (orc_set_roi), the checker of the full-resolution GPU parity tests
(tests/test_gpu_fullsize_parity.py): from the same pre-kernel state, a kernel run on a window must give, inside the window, the
bits the whole-image run gives, and must not touch anything outside -- for every kernel of the three pass kinds
(APD.cu:2386-2495; the colouring of APD.cu:1510-1585 makes every launch a function of state it does not write)."""
import numpy as np

import common

ARRAYS = ("planes", "fit_planes", "costs", "rng", "selected_views", "view_weight", "weak_info", "weak_reliable",
          "nearest_strong", "neighbours")


def _snapshot(o):
    return {k: np.array(getattr(o, k), copy=True) for k in ARRAYS}


def _restore(o, snap):
    for k in ARRAYS:
        getattr(o, k)[...] = snap[k]


def _schedule(iters, weak):
    s = [(1, 0), (2, 0)] + ([(3, 0), (4, 0)] if weak else []) + [(5, 0)]
    for i in range(iters):
        s += [(6, i), (7, i), (8, i)] + ([(9, i), (10, i)] if weak else [])
    return s + [(11, 0), (12, 0), (13, 0), (14, 0), (15, 0)]


def test_roi_equals_full_run_on_every_kernel(ob, synth):
    W, H, N = 83, 61, 3
    sc, imgs = common.scene_inputs(synth, W, H, N, seed=3, textureless=0.25)
    deps = common.fake_depth_maps(W, H, N + 1)
    passes = [dict(state=0, use_APD=0, weak_peak_radius=6),
              dict(state=1, use_APD=1, weak_peak_radius=6, rotate_time=2, ransac_threshold=0.00875),
              dict(state=2, use_APD=1, weak_peak_radius=4, rotate_time=4, ransac_threshold=0.0075, geom_consistency=1)]
    windows = [(0, 0, 30, 20), (17, 9, 64, 40), (50, 33, W, H), (0, 40, 83, 41)]
    prior = None
    weak_kernels_seen = 0
    for pi, extra in enumerate(passes):
        p = common.base_params(sc, N, seed=21, max_iterations=2, **extra)
        geom = bool(p.get("geom_consistency"))
        o = common.make_oracle(ob, sc, imgs, N, p, depths=deps if geom else None, prior=prior)
        weak = o.weak_count > 0
        for kid, it in _schedule(2, weak):
            before = _snapshot(o)
            o.run_kernel(kid, it)
            full = _snapshot(o)
            for (x0, y0, x1, y1) in windows:
                _restore(o, before)
                o.set_roi(x0, y0, x1, y1)
                o.run_kernel(kid, it)
                o.set_roi()
                inside = np.zeros((H, W), bool)
                inside[y0:y1, x0:x1] = True
                for k in ARRAYS:
                    got = getattr(o, k)
                    if k == "neighbours":  # indexed by WEAK pixel: rows of the window's WEAK pixels vs everything else
                        if not weak:
                            continue
                        rows = o.neighbours_map[inside & (before["weak_info"] == ob.WEAK)]
                        sel = np.zeros(got.shape[0], bool)
                        sel[rows] = True
                        assert np.array_equal(got[sel], full[k][sel]), (pi, kid, it, k)
                        assert np.array_equal(got[~sel], before[k][~sel]), (pi, kid, it, k, "outside")
                        continue
                    assert np.array_equal(common.bits(got[inside]), common.bits(full[k][inside])), (pi, kid, it, k, (x0, y0))
                    assert np.array_equal(common.bits(got[~inside]), common.bits(before[k][~inside])), (pi, kid, it, k, "outside")
            _restore(o, full)
            if kid in (3, 9, 10) and weak:
                weak_kernels_seen += 1
        planes, weak_map, views = o.planes.copy(), o.weak_info.copy(), o.selected_views.copy()
        prior = common.postprocess(planes, weak_map, views, p["depth_min"], p["depth_max"])
        o.close()
    assert weak_kernels_seen >= 6, "the scene must drive the weak kernels"
