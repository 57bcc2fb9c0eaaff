"""world_size-2 (and 3) gloo tests of the multi-GPU plumbing on CPU: view sharding, the padded
all-gather of per-view maps, and a 2-rank run in which every rank computes its own views' depth maps
with the ORACLE (standing in for the per-GPU handle) and all ranks end with identical gathered maps."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, num_views, with_oracle, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import __graft_entry__ as ge
    ge.load_package()
    from apd_mvs_amd import sharding
    mine = sharding.shard_views(num_views, world, rank)
    H, W = 24, 32
    local = {}
    if with_oracle:
        import common
        from apd_mvs_amd import synth
        from oracle import binding as ob
        ob.lib().orc_set_threads(2)
        for v in mine:
            sc = synth.make_scene(W, H, 2, seed=0, ref_view=v)
            imgs = sc.images_numpy()
            o = common.make_oracle(ob, sc, imgs, 2, common.base_params(sc, 2, max_iterations=1, seed=100 + v))
            o.run()
            local[v] = torch.from_numpy(o.planes.copy())
            o.close()
    else:
        for v in mine:
            local[v] = torch.full((H, W, 4), float(v + 1)) + torch.arange(W).view(1, W, 1) * 0.001
    gathered = sharding.allgather_maps(local, num_views)
    assert gathered.shape == (num_views, H, W, 4)
    for v in mine:
        assert torch.equal(gathered[v], local[v])
    t = sharding.timed_region_max(0.1 * (rank + 1), torch.device("cpu"))
    assert abs(t - 0.1 * world) < 1e-12
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), gathered.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_views_round_robin(pkg):
    from apd_mvs_amd import sharding
    assert sharding.shard_views(10, 4, 0) == [0, 4, 8]
    assert sharding.shard_views(10, 4, 3) == [3, 7]
    assert sharding.shard_views(2, 4, 3) == []
    allv = sorted(v for r in range(8) for v in sharding.shard_views(152, 8, r))
    assert allv == list(range(152))
    assert sharding.max_views_per_rank(152, 8) == 19
    assert all(sharding.owner_of(v, 8) == v % 8 for v in range(152))


@pytest.mark.parametrize("world,num_views", [(2, 5), (3, 2), (2, 2)])
def test_allgather_maps_gloo(tmp_path, world, num_views):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, num_views, False, str(tmp_path)), nprocs=world, join=True)
    outs = [np.load(tmp_path / ("rank%d.npy" % r)) for r in range(world)]
    for r in range(1, world):
        assert np.array_equal(outs[0], outs[r])
    for v in range(num_views):
        assert abs(outs[0][v, 0, 0, 0] - (v + 1)) < 1e-6


def test_two_ranks_compute_and_exchange_depth_maps(tmp_path):
    world, num_views = 2, 3
    port = _free_port()
    mp.spawn(_worker, args=(world, port, num_views, True, str(tmp_path)), nprocs=world, join=True)
    a, b = np.load(tmp_path / "rank0.npy"), np.load(tmp_path / "rank1.npy")
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    # three different reference views -> three different maps
    assert not np.array_equal(a[0], a[1]) and not np.array_equal(a[1], a[2])
    assert np.isfinite(a[..., 3]).all()
