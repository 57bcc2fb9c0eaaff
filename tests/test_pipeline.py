"""In-memory multi-pass scheduler (apd-mvs_amd/pipeline.py) without a GPU: schedule and resampling against the C++ host,
and a 2-rank gloo run with the ORACLE plugged in as the compute backend (standing in for the per-GPU handle)."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from common import OracleBackend  # the CPU oracle behind the pipeline's backend interface

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_LIB = os.path.join(ROOT, "apd-mvs_amd", "_build", "libapd_host.so")


def test_schedule_matches_reference_driver(pkg):
    from apd_mvs_amd import pipeline
    assert pipeline.compute_round_num(6200, 4130) == 4 and pipeline.compute_round_num(1920, 1080) == 2
    assert pipeline.compute_round_num(1000, 900) == 1 and pipeline.compute_round_num(1001, 10) == 2
    s = pipeline.pass_schedule(3, iters=3)
    assert len(s) == 12 and [x.scale_size for x in s] == [4] * 4 + [2] * 4 + [1] * 4
    assert [x.iteration_index for x in s] == list(range(12))
    first, g0, g1, g2 = s[0].params, s[1].params, s[2].params, s[3].params
    assert first["state"] == 0 and first["use_APD"] == 0 and first["geom_consistency"] == 0 and first["weak_peak_radius"] == 6
    assert [g["weak_peak_radius"] for g in (g0, g1, g2)] == [4, 2, 2] and all(g["state"] == 2 and g["geom_consistency"] == 1 for g in (g0, g1, g2))
    r1, r2 = s[4].params, s[8].params
    assert r1["state"] == 1 and r1["use_APD"] == 1 and r1["rotate_time"] == 2 and abs(r1["ransac_threshold"] - 0.00875) < 1e-7
    assert r2["rotate_time"] == 4 and abs(r2["ransac_threshold"] - 0.0075) < 1e-7
    assert s[9].params["use_APD"] == 1 and s[9].params["state"] == 2


@pytest.mark.parametrize("rows,cols,nr,nc", [(40, 64, 20, 32), (41, 63, 21, 32), (30, 50, 8, 13), (17, 9, 17, 5)])
def test_resampling_matches_cpp_host(pkg, rows, cols, nr, nc):
    from apd_mvs_amd import pipeline
    pkg.lib()
    L = C.CDLL(HOST_LIB)
    fp = C.POINTER(C.c_float)
    L.apdhost_resize_linear.argtypes = [fp, C.c_int, C.c_int, fp, C.c_int, C.c_int]
    L.apdhost_rescale_nearest_f32.argtypes = [fp, C.c_int, C.c_int, fp, C.c_int, C.c_int]
    rng = np.random.RandomState(rows * 100 + cols)
    src = (rng.rand(rows, cols) * 255).astype(np.float32)
    want = np.zeros((nr, nc), np.float32)
    L.apdhost_resize_linear(src.ctypes.data_as(fp), rows, cols, want.ctypes.data_as(fp), nr, nc)
    got = pipeline.resize_linear(src, nc, nr)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # nearest upsampling of prior state, swapped scale factors included (up and down)
    for tr, tc in ((rows * 2, cols * 2), (rows * 2 + 1, cols * 2 - 1), (nr, nc)):
        want = np.zeros((tr, tc), np.float32)
        L.apdhost_rescale_nearest_f32(src.ctypes.data_as(fp), rows, cols, want.ctypes.data_as(fp), tr, tc)
        got = pipeline.rescale_nearest(src, tc, tr)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (tr, tc)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene(pipeline, synth, ob):
    return pipeline.synthetic_ring(synth, 40, 32, 3, 2, ob.make_camera, seed=2)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import __graft_entry__ as ge
    ge.load_package()
    from apd_mvs_amd import pipeline, synth
    backend = OracleBackend()
    scene = _scene(pipeline, synth, backend.ob)
    out = pipeline.run_pipeline(scene, backend, iters=1, seed=5, max_passes=2)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **{"%s%d" % (k, v): getattr(st, k) for v, st in out.items()
                                                            for k in ("depth", "normal", "weak", "views")})
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_jacobi_vs_one_rank_gauss_seidel(pkg, synth, ob, tmp_path):
    """World 2: ranks end with identical gathered maps.  View 0 (first in the reference's order) reads only previous-pass
    depth maps in either order, so it is bit-identical to the single-rank run; the other views differ only through the
    Gauss-Seidel -> Jacobi change of the geometric term and stay close."""
    from apd_mvs_amd import pipeline
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert sorted(a.files) == sorted(b.files) and len(a.files) == 12
    for k in a.files:
        assert np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8)), k
    single = pipeline.run_pipeline(_scene(pipeline, synth, ob), OracleBackend(), iters=1, seed=5, max_passes=2)
    assert np.array_equal(a["depth0"].view(np.uint32), single[0].depth.view(np.uint32))
    assert np.array_equal(a["views0"], single[0].views) and np.array_equal(a["weak0"], single[0].weak)
    for v in (1, 2):
        d1, d2 = a["depth%d" % v], single[v].depth
        ok = (d1 > 0) & (d2 > 0)
        assert ok.mean() > 0.5
        assert (np.abs(d1[ok] - d2[ok]) <= 0.05 * d2[ok]).mean() > 0.8


def test_source_only_views_are_loaded_not_processed(pkg, synth, ob, tmp_path):
    """A subset run: pair.txt has entries for views 0..2, whose sources include image 3 (no entry of its own).  The reference
    loads such an image like any other (APD.cpp:419-452).  The scheduler processes three views, gives the geometric term an
    empty (all-zero) depth map for the fourth, and view 0 -- whose sources are full views -- ends exactly as in the run that
    reconstructs all four."""
    from apd_mvs_amd import pipeline
    full = pipeline.synthetic_ring(synth, 40, 32, 4, 2, ob.make_camera, seed=3)
    assert 3 in full.pairs[2] and 3 not in full.pairs[0]
    subset = pipeline.MvsScene(full.cameras, full.images, full.pairs[:3])
    assert subset.num_views == 3 and len(subset.images) == 4
    out_full = pipeline.run_pipeline(full, OracleBackend(), iters=1, seed=5, max_passes=2)
    out = pipeline.run_pipeline(subset, OracleBackend(), iters=1, seed=5, max_passes=2)
    assert sorted(out) == [0, 1, 2]
    for k in ("depth", "normal", "weak", "views"):
        assert np.array_equal(getattr(out[0], k), getattr(out_full[0], k)), k
    assert (out[2].depth > 0).mean() > 0.5
    # the loader builds such a scene from a folder: reference views first, source-only images after them
    (tmp_path / "images").mkdir()
    (tmp_path / "cams").mkdir()
    for i in range(4):
        img = np.clip(full.images[i], 0, 255).astype(np.uint8)
        (tmp_path / "images" / ("%08d.pgm" % (10 + i))).write_bytes(b"P5\n%d %d\n255\n" % (40, 32) + img.tobytes())
        cam = full.cameras[i]
        R, t, K = list(cam.R), list(cam.t), list(cam.K)
        txt = "extrinsic\n" + "".join("%.9g %.9g %.9g %.9g\n" % (R[3 * r], R[3 * r + 1], R[3 * r + 2], t[r]) for r in range(3))
        txt += "0 0 0 1\n\nintrinsic\n" + "".join("%.9g %.9g %.9g\n" % tuple(K[3 * r:3 * r + 3]) for r in range(3))
        txt += "\n%.9g 0.01 192 %.9g\n" % (cam.depth_min, cam.depth_max)
        (tmp_path / "cams" / ("%08d_cam.txt" % (10 + i))).write_text(txt)
    (tmp_path / "pair.txt").write_text("2\n10\n2 11 5.0 13 4.0\n11\n3 10 5.0 13 4.0 12 0.0\n")
    scene = pipeline.load_dense_folder(str(tmp_path), type(full.cameras[0]))
    assert scene.num_views == 2 and scene.ids == [10, 11, 13] and scene.pairs == [[1, 2], [0, 2]]   # score 0 dropped (main.cpp:41)
    assert len(scene.images) == 3 and np.array_equal(scene.images[2], np.clip(full.images[3], 0, 255).astype(np.uint8).astype(np.float32))
    (tmp_path / "pair.txt").write_text("1\n10\n1 10 5.0\n")
    with pytest.raises(ValueError):
        pipeline.load_dense_folder(str(tmp_path), type(full.cameras[0]))


def test_rescale_nearest_tensor_path_matches_numpy(pkg):
    """The device-resident pipeline resamples prior state with torch (RescaleMatToTargetSize, APD.cpp:752-774, swapped
    factors included); it must pick exactly the pixels the numpy / C++ host version picks, for every dtype it carries."""
    import torch
    from apd_mvs_amd import pipeline
    rng = np.random.RandomState(3)
    for (h, w), (th, tw) in (((30, 41), (60, 82)), ((31, 40), (61, 80)), ((33, 47), (67, 93)), ((20, 20), (20, 20)), ((50, 37), (25, 19))):
        for arr in (rng.rand(h, w).astype(np.float32), rng.rand(h, w, 4).astype(np.float32),
                    rng.randint(0, 3, (h, w)).astype(np.uint8), rng.randint(-2**31, 2**31 - 1, (h, w)).astype(np.int32)):
            ref = pipeline.rescale_nearest(arr, tw, th)
            out = pipeline.rescale_nearest(torch.from_numpy(arr), tw, th).numpy()
            assert out.shape == ref.shape and np.array_equal(out, ref), ((h, w), (th, tw), arr.dtype)
