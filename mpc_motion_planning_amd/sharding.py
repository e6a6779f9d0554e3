"""Sharding of a scenario batch over the GPUs of one node (one process per GPU).

Instances are independent NLPs (there is no coupling between problems in the reference either: each
`solver(...)` call of main_cbf_kin_c_sim.py:100 stands alone), so the batch is cut into contiguous slices, every rank
solves its own slice with no data-path collective, and ONE all-gather at the end gives every rank all converged
trajectories (mpcb_allgather: RCCL inside the library).  No torch here: the two-rank gloo test of this bookkeeping keeps its
torch helper in tests/dist_helpers.py.
"""


def shard_bounds(B, world, rank):
    """Contiguous slice [lo, hi) of rank `rank`: sizes differ by at most one, earlier ranks take the remainder.
    (Same rule as mpcb_shard_bounds of the C ABI, which the library uses for its own device groups; tests/test_abi.py
    checks the two against each other.)"""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    q, r = divmod(B, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def exchange_unique_id(rank, world, make_id, addr=None, port=None, timeout=120.0):
    """Host channel for the 128-byte RCCL group id when processes are started one per GPU (RANK / WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT in the environment, the launch bench.py is given): rank 0 listens on MASTER_PORT + 17 and hands the id to the
    other ranks over TCP.  Plain sockets, no torch.distributed.  `make_id` is called on rank 0 only."""
    import os
    import socket
    import time
    if world == 1:
        return make_id()
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(port if port is not None else int(os.environ.get("MASTER_PORT", "29500")) + 17)
    if rank == 0:
        uid = make_id()
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind((addr, port)); srv.listen(world); srv.settimeout(timeout)
        for _ in range(world - 1):
            conn, _a = srv.accept()
            conn.sendall(uid); conn.close()
        srv.close()
        return uid
    deadline = time.time() + timeout
    while True:
        try:
            c = socket.create_connection((addr, port), timeout=5.0)
            break
        except OSError:
            if time.time() > deadline:
                raise
            time.sleep(0.05)
    buf = b""
    while len(buf) < 128:
        chunk = c.recv(128 - len(buf))
        if not chunk:
            raise ConnectionError("group id exchange: connection closed after %d bytes" % len(buf))
        buf += chunk
    c.close()
    return buf


def shard(arr, world, rank):
    lo, hi = shard_bounds(len(arr), world, rank)
    return arr[lo:hi]
