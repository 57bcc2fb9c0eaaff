"""The fusion checker itself (oracle/fusion_oracle.cpp: RunFusion, APD.cpp:826-977, as the sequential loop it is) on a
synthetic ring with exact depth maps.  No GPU needed."""
import numpy as np

import test_gpu_dropin_binary as T


def test_sequential_fusion_on_exact_maps(pkg, ob, synth, tmp_path):
    from apd_mvs_amd import pipeline
    W, H, V, S = 96, 72, 4, 3
    scene, results = T._fusion_inputs(synth, pipeline, pkg, W, H, V, S, 0.0, seed=3)
    cams = (type(scene.cameras[0]) * V)(*scene.cameras)
    args = (cams, scene.images, [results[v].depth for v in range(V)], [results[v].normal for v in range(V)],
            [results[v].weak for v in range(V)], scene.pairs)
    n = ob.fuse(*args, tmp_path / "a.ply")
    xyz, bgr = T._read_ply(tmp_path / "a.ply")
    assert n == len(xyz) and 0.05 * W * H * V < n < W * H * V
    assert (bgr[:, 0] == bgr[:, 1]).all() and (bgr[:, 1] == bgr[:, 2]).all()
    # every fused point lies on one of the generator's two planes n.P + d = 0
    P = xyz.astype(np.float64)
    r1 = np.abs(P @ np.array([-0.15, -0.10, 1.0]) - 2.0) / np.linalg.norm([-0.15, -0.10, 1.0])
    r2 = np.abs(P @ np.array([0.25, 0.05, 1.0]) - 2.6) / np.linalg.norm([0.25, 0.05, 1.0])
    assert np.minimum(r1, r2).max() < 2e-3
    # a consumed source pixel never supports a second point: fusing twice is deterministic and order matters
    n2 = ob.fuse(*args, tmp_path / "b.ply")
    assert n2 == n and (tmp_path / "a.ply").read_bytes() == (tmp_path / "b.ply").read_bytes()
    rev = list(reversed(range(V)))
    cams_r = (type(scene.cameras[0]) * V)(*[scene.cameras[v] for v in rev])
    pairs_r = [[rev.index(s) for s in scene.pairs[v]] for v in rev]
    n3 = ob.fuse(cams_r, [scene.images[v] for v in rev], [results[v].depth for v in rev], [results[v].normal for v in rev],
                 [results[v].weak for v in rev], pairs_r, tmp_path / "c.ply")
    assert (tmp_path / "c.ply").read_bytes() != (tmp_path / "a.ply").read_bytes() and abs(n3 - n) < 0.5 * n


def test_oracle_kernels_against_libm_and_against_the_products_kernels(pkg, ob):
    """The fusion oracle restates acosf (fdlibm) and exp (contract C5) on its own (oracle/fusion_oracle.cpp includes nothing
    from the product).  Known answers: within a few ulp of libm; NaN outside [-1, 1].  Cross-check: the product's separately
    written kernels (csrc/apd_fusion_math.h, reached through libapd_host.so) give the same bits on a dense sample -- the
    byte-exact APD.ply comparison of the GPU tests rests on that agreement, and a slip in either copy shows up here."""
    import ctypes as C
    import os
    ob.build()
    L = C.CDLL(os.path.join(os.path.dirname(ob.__file__), "_build", "libapd_fusion_oracle.so"))
    for f in (L.orc_fusion_acos, L.orc_fusion_exp):
        f.restype = C.c_float
        f.argtypes = [C.c_float]
    pkg.lib()
    host = C.CDLL(os.path.join(os.path.dirname(pkg.library_path()), "libapd_host.so"))
    fp = C.POINTER(C.c_float)
    host.apdhost_fusion_math.argtypes = [fp, C.c_int, C.c_int, fp]
    rng = np.random.RandomState(11)
    xs = np.concatenate([np.linspace(-1, 1, 40001), rng.uniform(-1, 1, 20000), 1 - np.logspace(-8, -1, 2000), np.logspace(-8, -1, 2000) - 1,
                         [1.0, -1.0, 0.0, -0.0, 0.5, -0.5, 0.49999997, 0.50000006, 1e-9, -1e-9, 1.0000001, -1.0000001, 3.0, np.nan]]).astype(np.float32)
    es = np.concatenate([np.linspace(-30, 0, 40001), -rng.uniform(0, 25, 20000), [-87.5, -100.0, 0.0, 1.0, 89.0, np.nan]]).astype(np.float32)
    for which, fn, inp, ref_fn in ((0, L.orc_fusion_acos, xs, np.arccos), (1, L.orc_fusion_exp, es, np.exp)):
        mine = np.array([fn(float(v)) for v in inp], np.float32)
        theirs = np.zeros_like(inp)
        host.apdhost_fusion_math(inp.ctypes.data_as(fp), len(inp), which, theirs.ctypes.data_as(fp))
        assert np.array_equal(mine.view(np.uint32) | (np.isnan(mine) * 0x7fffffff).astype(np.uint32),
                              theirs.view(np.uint32) | (np.isnan(theirs) * 0x7fffffff).astype(np.uint32)), which
        with np.errstate(invalid="ignore", over="ignore"):
            ref = ref_fn(inp.astype(np.float64))
        ok = np.isfinite(ref) & (ref != 0) & (np.abs(ref) < 3e38) & (np.abs(ref) > 1e-37)
        assert (np.abs(mine[ok] - ref[ok]) <= 3 * np.spacing(np.abs(ref[ok]).astype(np.float32))).all(), which
        assert np.array_equal(np.isnan(mine), np.isnan(ref)), which
