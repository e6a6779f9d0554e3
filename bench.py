#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path: MPC solves/sec, kinematic bicycle N=30 + obstacle rows (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the solver over one batch per GPU: BASELINE config C2 — kinematic bicycle, N=30, T=0.1 s,
1 static obstacle [50,3.5,0,8,4.8,1.8], set-point [400,3.5,0,30], batch 4096 random feasible-at-node-0 initial states
per GPU (seeded), cold start z0 = 0 (as the reference's first step, main_cbf_kin_c_sim.py:47-50).
Inputs are resident in HBM before the timed region.  Instances are independent, so ranks shard the batch with no
data-path collective; for N > 1 every step ends with ONE RCCL all-gather of the converged trajectories
(torch.distributed, backend nccl = RCCL) so that every rank holds all of them — it is inside the timed region.
"scaling" is weak: per-GPU batch fixed.

value = instances that reached the KKT tolerance (status 0) on all ranks * K / max-over-ranks wall time.  Instances
that end with another status (the random scenes include unavoidable collisions, i.e. infeasible NLPs) are counted
in config.failed_per_step and are NOT part of value, though their time is.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# HIP maps streams onto a small pool of hardware queues (default 4 per process) and two streams on one queue run one
# after the other; torch + RCCL take several.  More queues keep the solver handles' streams (--inflight) concurrent.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md, chip-level parameters)
FP64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X FP64 vector peak (spec)


def measured_traffic(workload):
    """HBM bytes per launch of this workload from the rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate passes, KiB units,
    2x correction on FETCH_SIZE as MI355X_MICROARCH.md prescribes), recorded by tools/profile_round.sh in profiles/traffic.json.
    bench.py cannot collect PMC counters on itself, so this is the committed measurement of the same command; None if absent."""
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "traffic.json")
    try:
        t = json.load(open(p))
        if t.get("workload") == workload:
            return float(t["bytes_per_launch"]), "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE)"
    except (OSError, ValueError, KeyError):
        pass
    return None, None


def algorithmic_bytes_per_solve(nx, nz, n_obs_values):
    # 8*(2*nx + nz_in + obs_in + nz_out) + 16   (status i32 + iters i32 + obj f64)
    return 8 * (2 * nx + nz + n_obs_values + nz) + 16


def cpu_baseline(cfg, x0, xs, obs, min_seconds=1.5):
    """The CPU oracle (oracle/mpc_oracle.cpp, same NLP, same algorithm, OpenMP over the batch) timed on this box's
    host cores on a bounded sample of the same workload.  Reported next to the GPU number; CasADi+IPOPT cannot be
    timed (not installed here nor on the GPU box, no network: SURVEY.md §0 F2)."""
    from oracle import oracle
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("MPCB_CPU_THREADS", "16"))))   # the 1-GPU box's CPU share is 16 cores
    n = min(len(x0), 2048)
    t0 = time.perf_counter(); reps = 0; solved = 0
    while True:
        r = oracle.solve(cfg, x0[:n], xs[:n], obs[:n], threads=cores, want_multipliers=False)
        solved += int((r["status"] == 0).sum()); reps += 1
        dt = time.perf_counter() - t0
        if dt >= min_seconds:          # ~16 threads x 1.5 s = about 25 core-seconds of CPU work
            break
    m = min(len(x0), 24)                                  # single-thread latency per solve
    t1 = time.perf_counter()
    oracle.solve(cfg, x0[:m], xs[:m], obs[:m], threads=1, want_multipliers=False)
    lat_ms = 1e3 * (time.perf_counter() - t1) / m
    return {"value": solved / dt, "unit": "solves/s", "cores": cores, "kind": "port", "single_thread_ms_per_instance": lat_ms,
            "sample": "%d x first %d instances of the step's batch, OpenMP over instances, %.1f s wall (%.0f core-s); latency: %d instances on one thread" % (reps, n, dt, dt * cores, m),
            "note": "own FP64 C++ restatement of the NLP + IPOPT-style solver; CasADi+IPOPT baseline unavailable (casadi not installed)"}


def closed_loop_bench(args, bs, cfg, x0, xs, obs, workload, rank, world):
    """C5: `steps` = number of complete 80-step closed loops; value counts solved MPC steps per second.  The scenes are split
    over --inflight solver handles, each running its own closed loop from its own host thread (the steps of one loop depend on
    each other, independent loops overlap on the GPU like the launches of the other configurations)."""
    import threading
    from mpc_motion_planning_amd import _abi
    from mpc_motion_planning_amd.solver import BatchSolver
    sim_steps = 80                                            # sim_time 8 s / T_S 0.1 (main_cbf_kin_c_sim.py:68,87)
    F = max(1, min(2, args.inflight))                         # measured: 2 loops 879 k, 1 loop 739 k, 3 loops 640 k solved steps/s (host side of 3 loops contends)
    H = [bs] + [BatchSolver(cfg, device=bs.device) for _ in range(F - 1)]
    parts = np.array_split(np.arange(len(x0)), F)
    for h_ in H:
        h_.closed_loop(x0[:256], xs[:256], obs[:256], steps=4, obs_motion=_abi.OBSMOVE_PREDICTED)
        h_.timing(reset=True)
    res = [None] * F

    def run(q):
        res[q] = H[q].closed_loop(x0[parts[q]], xs[parts[q]], obs[parts[q]], steps=sim_steps, obs_motion=_abi.OBSMOVE_PREDICTED)

    t0 = time.perf_counter()
    reps = max(1, args.steps // 10)
    for _ in range(reps):
        th = [threading.Thread(target=run, args=(q,)) for q in range(F)]
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join()
    dt = time.perf_counter() - t0
    tms = [h_.timing() for h_ in H]
    status = np.concatenate([r_["status"] for r_ in res]); iters = np.concatenate([r_["iters"] for r_ in res])
    solved = int((status == 0).sum())
    out = {"metric": "mpc_solves_per_sec", "value": solved * reps / dt, "unit": "solves/s", "n_gpus": world, "steps": reps, "warmup": args.warmup,
           "ms_per_step": 1e3 * dt / reps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": workload, "scenes_per_gpu": len(x0), "sim_steps": sim_steps, "solved_steps": solved,
                      "failed_steps": int(status.size - solved), "scenes_all_steps_solved": int((status == 0).all(axis=1).sum()),
                      "iters_mean": float(iters.mean()), "host_pointer_entry": True, "loops_in_flight": F,
                      "kernel_ms_avg": sum(t_["total_ms"] for t_ in tms) / max(1, sum(t_["launches"] for t_ in tms)),
                      "launches": sum(t_["launches"] for t_ in tms)}}
    if rank == 0:
        print(json.dumps(out))
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="instances per GPU per step (default: the config's BASELINE batch)")
    ap.add_argument("--config", default="C2", choices=["C2", "C3", "C4", "C5"],
                    help="BASELINE.json config: C2 (default, the metric's config) kin+1 static obstacle B=4096; C3 kin+3 predicted "
                         "obstacles B=32768; C4 dyn N=40 3 obstacles B=8192/GPU; C5 closed loop 80 steps (solves = scenes x steps)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inflight", type=int, default=3,
                    help="solver handles (= HIP streams) used round-robin: step k+1 is launched while the tail of step k drains (a launch "
                         "ends with its slowest instance and leaves most SIMDs idle before that); 1 = strictly one launch at a time")
    ap.add_argument("--batches", type=int, default=5,
                    help="distinct synthetic batches resident in HBM, used round-robin by the steps (the launch time of a 4096-instance "
                         "batch moves +-15 %% with where its slowest instances fall in the dispatch order; one batch would report one draw)")
    ap.add_argument("--warm", action="store_true",
                    help="C2/C3/C4 only: time the NEXT receding-horizon step, started from the shifted solution of a cold solve "
                         "(main_cbf_kin_c_sim.py:16-26,92) instead of the cold start z0=0")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    if world == 1 and args.gpus > 1:
        raise SystemExit("--gpus %d needs the torch.distributed launcher (one process per GPU)" % args.gpus)

    from mpc_motion_planning_amd import scenes, _abi
    from mpc_motion_planning_amd.solver import BatchSolver, default_config, dims

    dist = None; torch = None
    use_dist = world > 1 or os.environ.get("MPCB_BENCH_FORCE_DIST") == "1"    # the env switch rehearses the N > 1 plumbing with one rank
    if use_dist:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    conf = args.config
    B = args.batch or {"C2": 4096, "C3": 32768, "C4": 8192, "C5": 4096}[conf]
    obs_kind = _abi.OBSIN_STATIC
    NB = 1 if (args.warm or conf == "C5") else max(1, args.batches)
    sets = []                                               # NB x (x0, xs, obs)
    if conf == "C2":
        cfg = default_config(model=_abi.MODEL_KIN, N=30, T=0.1, n_obs=1)
        for q in range(NB):
            sets.append(scenes.sample_c2(B, seed=q + 16 * rank))          # seeds 0..4
        workload = "C2: kinematic bicycle + 1 static CBF/keep-out obstacle row set (MPC_CBF_optimize_kin), N=30, T=0.1, batch %d random x0 per GPU, cold start z0=0" % B
    elif conf == "C3":
        cfg = default_config(model=_abi.MODEL_KIN, N=30, T=0.1, n_obs=3)
        for q in range(NB):
            a0, a1, _, a3 = scenes.sample_c3(B, N=30, dt=0.1, seed=100 + q + 16 * rank)
            sets.append((a0, a1, a3))
        obs_kind = _abi.OBSIN_PREDICTED
        workload = "C3: kinematic bicycle + 3 predicted moving obstacles (MPC_CBF_optimize_kin_pre + Obs_prediction), N=30, batch %d per GPU, cold start" % B
    elif conf == "C4":
        cfg = default_config(model=_abi.MODEL_DYN, N=40, T=0.1, n_obs=3)
        for q in range(NB):
            sets.append(scenes.sample_c4(B, seed=200 + q + 16 * rank, n_obs=3))
        workload = "C4: dynamic bicycle (MPC_CBF_optimize_dyn, aligned rows), N=40, 3 static obstacles, batch %d per GPU, cold start" % B
    else:
        cfg = default_config(model=_abi.MODEL_KIN, N=30, T=0.1, n_obs=3)
        a0, a1, a2, _ = scenes.sample_c3(B, N=30, dt=0.1, seed=4000 + rank)
        sets.append((a0, a1, a2))
        workload = "C5: closed loop, %d scenes per GPU x 80 receding-horizon steps, kinematic bicycle + 3 moving obstacles re-predicted every step" % B
    x0, xs, obs = sets[0]
    nx, nz, ng = dims(cfg)
    bs = BatchSolver(cfg, device=local_rank)
    if conf == "C5":
        return closed_loop_bench(args, bs, cfg, x0, xs, obs, workload, rank, world)
    D = []                                                  # per batch: inputs and the status / iteration outputs, all resident in HBM
    for (a0, a1, a2) in sets:
        D.append(dict(x0=bs.device_array((B, nx)).upload(a0), xs=bs.device_array((B, nx)).upload(a1), obs=bs.device_array(a2.shape).upload(a2),
                      st=bs.device_array((B,), np.int32), it=bs.device_array((B,), np.int32)))
    d_x0 = D[0]["x0"]
    cyc = [0]                                               # step counter: step k solves batch k mod NB on handle k mod F
    F = max(1, args.inflight)
    H = [bs] + [BatchSolver(cfg, device=local_rank) for _ in range(F - 1)]     # one handle = one stream (include/mpcbatch.h)
    d_obj = [bs.device_array((B,)) for _ in range(F)]; d_kkt = [bs.device_array((B, 4)) for _ in range(F)]
    if use_dist:    # z lives in torch tensors so that RCCL can gather it; the solver only sees their raw pointers
        z_bufs = [torch.empty((B, nz), dtype=torch.float64, device="cuda") for _ in range(max(2, F))]
        z_all = torch.empty((world * B, nz), dtype=torch.float64, device="cuda")
        z_ptrs = [t.data_ptr() for t in z_bufs]
        pending = [None] * len(z_bufs); waiting = []        # gathers in flight per z buffer; solves whose gather is not issued yet
    else:
        z_ptrs = [bs.device_array((B, nz)) for _ in range(F)]

    d_z0 = None
    if args.warm:
        # one cold solve, then the reference's shift: x0 <- X_1 (the plant is the model's own Euler step), U <- [U_1.., U_N-1, U_N-1],
        # X <- [X_1.., X_N, X_N]; instances the cold solve did not finish keep their cold start
        N = cfg.N
        cold = bs.solve_batch(x0, xs, obs)            # (C3: the predicted obstacle trajectories are kept as they are)
        Z = cold["z"]; ok = cold["status"] == 0
        U = Z[:, :2 * N].reshape(B, N, 2); X = Z[:, 2 * N:].reshape(B, N + 1, nx)
        z0 = np.concatenate([np.concatenate([U[:, 1:], U[:, -1:]], 1).reshape(B, -1), np.concatenate([X[:, 1:], X[:, -1:]], 1).reshape(B, -1)], 1)
        z0[~ok] = 0.0
        x0 = np.where(ok[:, None], X[:, 1], x0)
        d_x0.upload(x0)
        d_z0 = bs.device_array((B, nz)).upload(z0)
        workload = workload.replace("cold start z0=0", "cold start").replace("cold start", "WARM start: next receding-horizon step from the shifted previous solution")

    def issue_gathers(keep):
        # N > 1: a solve whose successor has been launched is waited for (its handle's stream only) and its z is gathered on
        # RCCL's stream, while the younger solves keep the GPU busy
        while len(waiting) > keep:
            hq, zq = waiting.pop(0)
            H[hq].sync()
            pending[zq] = dist.all_gather_into_tensor(z_all, z_bufs[zq], async_op=True)

    def step(gather=True):
        k = cyc[0]; cyc[0] += 1
        d = D[k % NB]; hq = k % F
        if not (use_dist and gather):
            H[hq].solve_device(B, d["x0"], d["xs"], d["obs"], obs_kind, d_z0, z_ptrs[hq], d_obj[hq], d["st"], d["it"], d_kkt[hq])
            return
        zq = k % len(z_bufs)
        if pending[zq] is not None:                     # the gather that last read this buffer must be done
            pending[zq].wait(); torch.cuda.current_stream().synchronize(); pending[zq] = None
        H[hq].solve_device(B, d["x0"], d["xs"], d["obs"], obs_kind, d_z0, z_ptrs[zq], d_obj[hq], d["st"], d["it"], d_kkt[hq])
        waiting.append((hq, zq))
        issue_gathers(F - 1)

    def fence():
        if use_dist:
            issue_gathers(0)
        for h_ in H:
            h_.sync()
        if use_dist:
            for w in pending:
                if w is not None:
                    w.wait()
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    for h_ in H:
        h_.timing(reset=True)
    cyc[0] = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if use_dist and world == 1:       # one-rank rehearsal of the N > 1 plumbing: the gathered block must equal what the last solve wrote
        assert torch.equal(z_all[:B], z_bufs[(args.steps - 1) % len(z_bufs)]), "all_gather result differs from the solver output"
    tms = [h_.timing() for h_ in H]
    tm = {"total_ms": sum(t_["total_ms"] for t_ in tms), "launches": sum(t_["launches"] for t_ in tms)}

    # every batch's status / iteration arrays hold its latest (identical, deterministic) result; weight by how often it ran
    uses = [args.steps // NB + (1 if q < args.steps % NB else 0) for q in range(NB)]
    st_b = [d["st"].download() for d in D]; it_b = [d["it"].download() for d in D]
    solved = sum(u * int((s_ == 0).sum()) for u, s_ in zip(uses, st_b))            # solved instances over all timed steps of this rank
    status = np.concatenate([s_ for u, s_ in zip(uses, st_b) if u]); iters = np.concatenate([i_ for u, i_ in zip(uses, it_b) if u])
    iters_per_launch = sum(u * float(i_.sum()) for u, i_ in zip(uses, it_b)) / max(1, args.steps)
    dt_nogather = None
    if use_dist:    # the same K steps once more without the gather, reported next to the headline value
        cyc[0] = 0
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step(gather=False)
        fence()
        dt_nogather = time.perf_counter() - t1
    if use_dist:
        t = torch.tensor([dt, float(solved), float(tm["total_ms"]), dt_nogather], dtype=torch.float64, device="cuda")
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt_max = float(tmax[0]); solved_all = int(round(float(tsum[1]))); dt_nogather = float(tmax[3])
    else:
        dt_max = dt; solved_all = solved

    if rank == 0:
        kernel_ms = tm["total_ms"] / max(1, tm["launches"])
        abytes = algorithmic_bytes_per_solve(nx, nz, int(obs[0].size)) * B
        achieved = abytes / (kernel_ms * 1e-3) / 1e9
        it_ok = iters[status == 0]
        flops = (147e3 if cfg.model == _abi.MODEL_DYN else 56e3) * iters_per_launch   # ~56 kflop (kin N=30) / ~147 kflop (dyn N=40) per iteration
        traffic, traffic_src = measured_traffic(workload)
        out = {
            "metric": "mpc_solves_per_sec", "value": solved_all / dt_max, "unit": "solves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt_max / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload, "batch_per_gpu": B, "distinct_batches": NB,
                       "solved_per_step": solved_all / args.steps, "failed_per_step": world * B - solved_all / args.steps,
                       "iters_mean_solved": float(it_ok.mean()) if len(it_ok) else None, "iters_max": int(iters.max()),
                       "launches_in_flight": F, "tol": cfg.tol, "collective": "rccl all_gather of z per step, overlapped with the next step's solve" if use_dist else "none",
                       "value_without_gather": (solved_all / dt_nogather) if dt_nogather else None},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "kernel": "mpcb_kernel_%s<%d>" % ("dyn" if cfg.model == _abi.MODEL_DYN else "kin", 1 if cfg.n_obs <= 1 else 3), "kernel_ms_avg": kernel_ms,
                         "algorithmic_bytes_per_launch": abytes,
                         "achieved_over_wall_clock": abytes * args.steps / dt_max / 1e9,   # launches overlap (launches_in_flight): each one lasts longer than a step

                         "note": "compulsory I/O is 3072 B/solve; the solve is LDS-resident, bound by FP64 VALU issue and the serial "
                                 "Riccati chain, not by HBM (SURVEY.md F10, DESIGN.md §5)"},
            "roofline_fp64": {"bound": "fp64_valu", "achieved": flops * args.steps / dt_max / 1e12, "peak": FP64_VECTOR_PEAK_TFLOPS,
                              "unit": "TFLOP/s", "frac": flops * args.steps / dt_max / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                              "model": "56 kflop x interior-point iterations summed over the batch, per step, over the wall clock of the "
                                       "timed region (launches overlap, so a launch lasts longer than a step)"},
        }
        if not args.no_cpu_baseline and world == 1:      # rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(cfg, x0, xs, obs)
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
