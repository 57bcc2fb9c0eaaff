"""The in-memory multi-scale scheduler (apd-mvs_amd/pipeline.py) on the GPU against the CPU oracle.

* two pyramid levels, all four pass kinds of main.cpp:164-217 (FIRST_INIT, REFINE_ITER + geometric term, REFINE_INIT + APD,
  REFINE_ITER + APD + geometric term), image / intrinsics downscaling (APD.cpp:464-488) and the nearest-neighbour upsampling
  of the prior state (APD.cpp:752-774): run_pipeline(HipBackend) == run_pipeline(OracleBackend), every map of every view,
  bit for bit;
* BASELINE.json configs[3] (Tanks&Temples-shaped: 1920x1080, 10 source views per reference view, views sharded by reference
  image, RCCL all-gather before fusion) on the one GPU of the box: the whole schedule under an initialised `nccl` process group
  (the collective path is taken with one rank too), selected (view, pass) pairs re-run in lock step against the oracle's
  region-of-interest mode at full size, the gathered maps fused on the GPU and compared with the sequential fusion loop."""
import os
import socket

import numpy as np
import pytest

import common

pytestmark = pytest.mark.gpu


def test_two_level_pipeline_hip_backend_equals_oracle_backend(gpu_pkg, ob, synth):
    from apd_mvs_amd import pipeline
    W, H, V, S = 1040, 72, 3, 2   # max(W, H) > 1000 -> two levels: 520x36 and 1040x72 (main.cpp:72-88)
    hip_scene = pipeline.synthetic_ring(synth, W, H, V, S, gpu_pkg.make_camera, seed=6, textureless=0.25)
    orc_scene = pipeline.synthetic_ring(synth, W, H, V, S, ob.make_camera, seed=6, textureless=0.25)
    seen = []
    got = pipeline.run_pipeline(hip_scene, pipeline.HipBackend(gpu_pkg, device=0), iters=2, seed=91, log=seen.append)
    want = pipeline.run_pipeline(orc_scene, common.OracleBackend(threads=0), iters=2, seed=91)
    assert len(seen) == 8 * V and "scale 2" in seen[0] and "scale 1" in seen[-1]
    weak_pixels = 0
    for v in range(V):
        for k in ("depth", "normal", "weak", "views"):
            a, b = getattr(got[v], k), getattr(want[v], k)
            assert a.shape == b.shape and np.array_equal(common.bits(a), common.bits(b)), (v, k)
        weak_pixels += int((want[v].weak == 0).sum())
        assert got[v].depth.shape == (H, W)
    assert weak_pixels > 200, "the finer level must run the APD (weak) path"
    assert (got[0].depth > 0).mean() > 0.5


class CheckedHipBackend:
    """HipBackend that, for chosen (pass, view) pairs, first replays the pass kernel by kernel on a fresh handle in lock step
    with the oracle's region-of-interest mode (tests/common.py::fullsize_lockstep), then returns the normal result."""

    accepts_tensors = True

    def __init__(self, pkg, ob, inner, check, windows_of):
        self.pkg, self.ob, self.inner, self.check, self.windows_of = pkg, ob, inner, set(check), windows_of
        self.device = inner.device
        self.calls = {}
        self.checked = []

    @property
    def camera_type(self):
        return self.inner.camera_type

    def run_pass(self, width, height, params, cameras, images, depths, prior):
        key = (params["state"], params["geom_consistency"], params["use_APD"], width)
        n = self.calls.get(key, 0)
        self.calls[key] = n + 1
        if n == 0 and (params["state"], params["geom_consistency"], params["use_APD"]) in self.check:
            self._lockstep(width, height, params, cameras, images, depths, prior)
        return self.inner.run_pass(width, height, params, cameras, images, depths, prior)

    def _lockstep(self, W, H, params, cameras, images, depths, prior):
        pkg, ob = self.pkg, self.ob
        npy = lambda t: None if t is None else t.detach().cpu().contiguous().numpy()
        imgs = [npy(t) for t in images]
        deps = None if depths is None else [npy(t) for t in depths]
        pr = None
        if prior is not None:
            pr = (npy(prior[0]), npy(prior[1]).view(np.uint32), npy(prior[2]))
        h = pkg.Handle(W, H, pkg.default_params(**params), device=self.device)
        h.upload_views(cameras, imgs, deps)
        if pr is not None:
            h.upload_prior(*pr)
        ocams = [ob.Camera.from_buffer_copy(c) for c in cameras]
        o = ob.Oracle(W, H, ob.default_params(**params), ocams, imgs, depths=deps, prior_planes=None if pr is None else pr[0],
                      prior_views=None if pr is None else pr[1], prior_weak=None if pr is None else pr[2])
        assert h.weak_count == o.weak_count
        weak = h.weak_count > 0
        sched = [(1, 0), (2, 0)] + ([(3, 0), (4, 0)] if weak else []) + [(5, 0)]
        for i in range(params["max_iterations"]):
            sched += [(6, i), (7, i), (8, i)] + ([(9, i), (10, i)] if weak else [])
        sched += [(11, 0), (12, 0), (13, 0), (14, 0), (15, 0)]
        windows = self.windows_of(W, H, None if pr is None else pr[2])
        n = common.fullsize_lockstep(pkg, h, o, sched, windows, "configs[3] state %d geom %d APD %d %dx%d"
                                     % (params["state"], params["geom_consistency"], params["use_APD"], W, H))
        self.checked.append((params["state"], params["geom_consistency"], params["use_APD"], W, H, n, h.weak_count))
        h.close()
        o.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_configs3_1080p_10src_sharded_pipeline_allgather_fusion(gpu_pkg, ob, synth, tmp_path):
    import torch
    import torch.distributed as dist
    from apd_mvs_amd import pipeline
    W, H, V, S = 1920, 1080, 12, 10
    scene = pipeline.synthetic_ring(synth, W, H, V, S, gpu_pkg.make_camera, seed=3, textureless=0.2)

    def windows_of(w, h, weak):
        wins = [(0, 0, 160, 96), (w // 2 - 80, h // 2 - 48, w // 2 + 80, h // 2 + 48), (w - 160, h - 96, w, h)]
        if weak is not None and (weak == 0).sum() > 1000:
            ys, xs = np.nonzero(weak == 0)
            k = len(ys) // 2
            x0, y0 = min(max(int(xs[k]) - 64, 0), w - 128), min(max(int(ys[k]) - 40, 0), h - 80)
            wins.append((x0, y0, x0 + 128, y0 + 80))
        return wins

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        assert dist.get_backend() == "nccl"
        # checked pass kinds: (state, geom, APD) -- the first view of the first pass of each kind, at whichever level it occurs
        check = [(0, 0, 0), (2, 1, 0), (1, 0, 1), (2, 1, 1)]
        backend = CheckedHipBackend(gpu_pkg, ob, pipeline.HipBackend(gpu_pkg, device=0), check, windows_of)
        out = pipeline.run_pipeline(scene, backend, iters=3, seed=12345)
    finally:
        dist.destroy_process_group()
    kinds = sorted((c[0], c[1], c[2]) for c in backend.checked)
    assert kinds == sorted(check), backend.checked
    assert any(c[3] == W and c[6] > 0 for c in backend.checked), "an APD pass at 1920x1080 with WEAK pixels must have been checked"
    assert len(out) == V and all(out[v].depth.shape == (H, W) for v in range(V))
    # quality: the gathered depth maps lie on the generator's surfaces
    sc = synth.make_scene(W, H, V - 1, seed=3, textureless=0.2)
    gt = sc.gt_depth.numpy()
    d = out[0].depth
    ok = d > 0
    assert ok.mean() > 0.8 and (np.abs(d[ok] - gt[ok]) / gt[ok] < 0.01).mean() > 0.9
    # fusion of the gathered maps on the GPU == the reference's sequential loop on the same maps
    cams = (type(scene.cameras[0]) * V)(*scene.cameras)
    n_cpu = ob.fuse(cams, scene.images, [out[v].depth for v in range(V)], [out[v].normal for v in range(V)],
                    [out[v].weak for v in range(V)], scene.pairs, tmp_path / "cpu.ply")
    n_gpu = pipeline.fuse(scene, out, tmp_path / "gpu.ply")
    assert n_cpu == n_gpu and n_gpu > 0.5 * W * H
    assert (tmp_path / "cpu.ply").read_bytes() == (tmp_path / "gpu.ply").read_bytes()
