"""The N > 1 path on CPU: two `gloo` ranks shard one scenario batch (contiguous slices, no data-path collective),
solve their slices and exchange the converged trajectories with ONE all-gather — the same host logic bench.py runs
over RCCL.  On this GPU-less box the per-rank solve is the CPU oracle standing in for the device solve (test only)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, B, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import oracle
    from mpc_motion_planning_amd import scenes
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from mpc_motion_planning_amd.sharding import shard, shard_bounds
    from dist_helpers import gather_rows
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = oracle.default_config(N=30, n_obs=1); cfg.init_rollout = 1; cfg.mu_init = 10.0
    x0, xs, obs = scenes.sample_c2(B, seed=33)
    lo, hi = shard_bounds(B, world, rank)
    r = oracle.solve(cfg, shard(x0, world, rank), shard(xs, world, rank), shard(obs, world, rank), threads=2)
    z_all = gather_rows(torch.from_numpy(r["z"]), dist, B)
    st_all = gather_rows(torch.from_numpy(r["status"].astype(np.int64))[:, None], dist, B)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), z=z_all.numpy(), status=st_all.numpy()[:, 0], lo=lo, hi=hi)
    dist.barrier(); dist.destroy_process_group()


def test_two_rank_sharded_solve_and_allgather(tmp_path):
    import torch.multiprocessing as mp
    from oracle import oracle
    from mpc_motion_planning_amd import scenes
    B, world, port = 37, 2, 29517 + (os.getpid() % 500)          # ragged: 19 + 18
    mp.spawn(_worker, args=(world, port, B, str(tmp_path)), nprocs=world, join=True)
    cfg = oracle.default_config(N=30, n_obs=1); cfg.init_rollout = 1; cfg.mu_init = 10.0
    x0, xs, obs = scenes.sample_c2(B, seed=33)
    ref = oracle.solve(cfg, x0, xs, obs)
    outs = [np.load(os.path.join(tmp_path, "rank%d.npz" % r)) for r in range(world)]
    assert (int(outs[0]["lo"]), int(outs[0]["hi"]), int(outs[1]["lo"]), int(outs[1]["hi"])) == (0, 19, 19, 37)
    for o in outs:                                               # every rank holds every trajectory, bit-identical
        assert np.array_equal(o["z"], ref["z"]) and np.array_equal(o["status"], ref["status"])


def _id_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    from mpc_motion_planning_amd.sharding import exchange_unique_id
    uid = exchange_unique_id(rank, world, lambda: bytes(range(128)), addr="127.0.0.1", port=port)
    open(os.path.join(out_dir, "id%d.bin" % rank), "wb").write(uid)


def test_group_id_travels_from_rank0_over_tcp(tmp_path):
    """The host channel of the in-library RCCL group (bench.py, N > 1 without torch.distributed): rank 0 hands the 128-byte
    id to the other ranks over a plain TCP socket."""
    import multiprocessing as mp
    world, port = 3, 31017 + (os.getpid() % 500)
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=_id_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p_ in ps:
        p_.start()
    for p_ in ps:
        p_.join(60)
        assert p_.exitcode == 0
    for r in range(world):
        assert open(os.path.join(tmp_path, "id%d.bin" % r), "rb").read() == bytes(range(128))
