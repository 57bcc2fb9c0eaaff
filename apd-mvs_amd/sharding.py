"""Multi-GPU plumbing: reference views are independent units, so ranks shard the views
round-robin and exchange only the per-view depth / normal / weak maps after a pass (the reference does
that exchange through depths.dmb files, APD.cpp:492-509).  Works with any torch.distributed backend:
"nccl" (= RCCL over xGMI) on MI355X, "gloo" in the CPU tests."""
import torch
import torch.distributed as dist


def shard_views(num_views, world_size, rank):
    """Round-robin by problem index: contiguous blocks would put mutual source views on one rank."""
    return list(range(rank, num_views, world_size))


def owner_of(view, world_size):
    return view % world_size


def max_views_per_rank(num_views, world_size):
    return (num_views + world_size - 1) // world_size


def allgather_maps(local_maps, num_views, group=None):
    """All-gather per-view maps.

    local_maps: dict view_index -> tensor [H, W, C] (same shape/dtype on every rank) for the views this
    rank owns.  Returns a tensor [num_views, H, W, C] identical on every rank.  Ranks owning fewer views
    pad with zeros so that one equal-count all_gather_into_tensor moves everything (one large collective
    instead of one per view: xGMI is per-link bound).
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    mine = shard_views(num_views, world, rank)
    assert sorted(local_maps.keys()) == mine, (sorted(local_maps.keys()), mine)
    slots = max_views_per_rank(num_views, world)
    sample = next(iter(local_maps.values())) if local_maps else None
    shape = torch.Size(_broadcast_shape(sample, group))
    dtype = sample.dtype if sample is not None else torch.float32
    device = sample.device if sample is not None else torch.device("cpu")
    send = torch.zeros((slots,) + tuple(shape), dtype=dtype, device=device)
    for k, v in enumerate(mine):
        send[k].copy_(local_maps[v])
    recv = torch.empty((world * slots,) + tuple(shape), dtype=dtype, device=device)
    dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.view(world, slots, *shape)
    out = torch.empty((num_views,) + tuple(shape), dtype=dtype, device=device)
    for v in range(num_views):
        out[v].copy_(recv[owner_of(v, world), v // world])
    return out


def _broadcast_shape(sample, group):
    """Every rank must agree on the map shape even if one of them owns no view."""
    world = dist.get_world_size(group)
    mine = list(sample.shape) if sample is not None else None
    shapes = [None] * world
    dist.all_gather_object(shapes, mine, group=group)
    known = [s for s in shapes if s is not None]
    assert known and all(s == known[0] for s in known), shapes
    return known[0]


def timed_region_max(elapsed_seconds, device, group=None):
    """MAX over ranks of a wall time (bench.py's contract)."""
    t = torch.tensor([elapsed_seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
