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
