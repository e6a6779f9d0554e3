"""Sharding of a scenario batch over the GPUs of one node (one process per GPU).

Instances are independent NLPs (there is no coupling between problems in the reference either: each
`solver(...)` call of main_cbf_kin_c_sim.py:100 stands alone), so the batch is cut into contiguous slices, every rank
solves its own slice with no data-path collective, and ONE all-gather at the end gives every rank all converged
trajectories.  `torch.distributed` is plumbing only (backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU
tests); nothing here touches the solver.
"""


def shard_bounds(B, world, rank):
    """Contiguous slice [lo, hi) of rank `rank`: sizes differ by at most one, earlier ranks take the remainder."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    q, r = divmod(B, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard(arr, world, rank):
    lo, hi = shard_bounds(len(arr), world, rank)
    return arr[lo:hi]


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
