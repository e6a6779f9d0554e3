"""torch.distributed helper of tests/test_distributed_cpu.py (the product package imports no torch: the GPU path gathers with
RCCL from inside libmpcbatch, mpcb_allgather).  TEST INFRASTRUCTURE ONLY."""
from mpc_motion_planning_amd.sharding import shard_bounds


def gather_rows(local, dist, B_total):
    """All-gather row blocks of unequal length into a [B_total, ...] tensor (torch tensors in, torch tensor out).
    Ragged shards are padded to the longest one so that a single all_gather_into_tensor does the exchange."""
    import torch
    world = dist.get_world_size()
    sizes = [shard_bounds(B_total, world, r) for r in range(world)]
    longest = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((longest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world * longest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad)
    parts = [out[r * longest: r * longest + (hi - lo)] for r, (lo, hi) in enumerate(sizes)]
    return torch.cat(parts, dim=0)
