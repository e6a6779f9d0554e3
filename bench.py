#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path: MPC solves/sec, kinematic bicycle N=30 + obstacle rows (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the solver over one batch per GPU: BASELINE config C2 — kinematic bicycle, N=30, T=0.1 s,
1 static obstacle [50,3.5,0,8,4.8,1.8], set-point [400,3.5,0,30], batch 4096 random feasible-at-node-0 initial states
per GPU (seeded), cold start z0 = 0 (as the reference's first step, main_cbf_kin_c_sim.py:47-50).
Inputs are resident in HBM before the timed region.  Instances are independent, so ranks shard the work with no
data-path collective; for N > 1 every step ends with ONE RCCL all-gather of the converged trajectories so that every
rank holds all of them — it is inside the timed region.  RCCL is called by libmpcbatch.so itself (mpcb_comm_init_rank /
mpcb_allgather / mpcb_allreduce); this file imports no torch: the launcher only provides RANK / WORLD_SIZE / MASTER_*.
"scaling" is weak: per-GPU batch fixed.

value = instances that reached the KKT tolerance (status 0) on all ranks * K / max-over-ranks wall time.  Instances
that end with another status (the random scenes include unavoidable collisions, i.e. locally infeasible NLPs) are
counted in config.failed_per_step and are NOT part of value, though their time is.

PARITY STATUS: unpinned versus the reference's own solver (CasADi + IPOPT is not installed here or on the GPU box, the
reference holds no golden outputs): solutions are checked against this repository's CPU oracle and an independent KKT
certificate, see DESIGN.md §4.
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
# after the other; RCCL takes some too.  More queues keep the solver handles' streams (--inflight) concurrent.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md, chip-level parameters)
FP64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X FP64 vector peak (spec)
PARITY_NOTE = ("parity UNPINNED vs the reference's CasADi+IPOPT (not installed here nor on the GPU box, no golden outputs in the "
               "reference): results are checked against this repo's CPU oracle (same NLP, IPOPT's published algorithm) and an "
               "independent KKT certificate (oracle/kkt_check.py)")


def measured_traffic(workload):
    """HBM bytes per launch of this workload from the rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate passes, KiB units,
    2x correction on FETCH_SIZE as MI355X_MICROARCH.md prescribes), recorded by tools/profile_round.sh in profiles/traffic.json.
    A process cannot collect PMC counters on itself, so this is the committed OFFLINE measurement of the same command; None if absent."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        t = json.load(open(p))
        for e in (t if isinstance(t, list) else [t]):          # one entry per profiled workload
            if e.get("workload_key") == workload.split(":")[0]:
                return float(e["bytes_per_launch"]), "offline: profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE of this command, committed)"
    except (OSError, ValueError, KeyError):
        pass
    return None, None


def algorithmic_bytes_per_solve(nx, nz, n_obs_values):
    # 8*(2*nx + nz_in + obs_in + nz_out) + 16   (status i32 + iters i32 + obj f64)
    return 8 * (2 * nx + nz + n_obs_values + nz) + 16


def host_cores():
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    return max(1, min(cores, int(os.environ.get("MPCB_CPU_THREADS", "16"))))   # the 1-GPU box's CPU share is 16 cores


def cpu_baseline(cfg, x0, xs, obs, min_seconds=1.5):
    """The CPU oracle (oracle/mpc_oracle.cpp, same NLP, same algorithm, OpenMP over the batch) timed on this box's
    host cores on a bounded sample of the same workload.  Reported next to the GPU number; CasADi+IPOPT cannot be
    timed (not installed here nor on the GPU box, no network: SURVEY.md §0 F2)."""
    from oracle import oracle
    cores = host_cores()
    n = min(len(x0), 2048)
    oracle.solve(cfg, x0[:4 * cores], xs[:4 * cores], obs[:4 * cores], threads=cores, want_multipliers=False)   # untimed: starts the OpenMP team, first-touch of the per-thread arenas
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


def cpu_baseline_closed_loop(cfg, x0, xs, obs, sim_steps, min_seconds=8.0):
    """C5 on the host: the oracle driving the same receding-horizon loop (solve, plant step, shift, obstacle advance and
    re-prediction: main_cbf_kin_c_sim_pre.py:86-126) on a bounded sample of the scenes, OpenMP over scenes inside every step."""
    from oracle import oracle
    from mpc_motion_planning_amd import scenes
    cores = host_cores()
    n = min(len(x0), 768); N = cfg.N
    xc = x0[:n].copy(); oc = obs[:n].copy(); z0 = np.zeros((n, 2 * N + 4 * (N + 1)))
    oracle.solve(cfg, xc[:4 * cores], xs[:4 * cores], scenes.predict_obstacles(oc[:4 * cores], cfg.T, N), threads=cores, want_multipliers=False)   # untimed warm-up of the OpenMP team
    t0 = time.perf_counter(); solved = 0; done = 0
    for _ in range(sim_steps):
        r = oracle.solve(cfg, xc, xs[:n], scenes.predict_obstacles(oc, cfg.T, N), z0=z0, threads=cores, want_multipliers=False)
        solved += int((r["status"] == 0).sum()); done += 1
        U = r["z"][:, :2 * N].reshape(n, N, 2); X = r["z"][:, 2 * N:].reshape(n, N + 1, 4)
        f = np.stack([xc[:, 3] * np.cos(xc[:, 2]), xc[:, 3] * np.sin(xc[:, 2]), xc[:, 3] * np.tan(U[:, 0, 0]) / cfg.veh_l, U[:, 0, 1]], axis=1)
        xc = xc + cfg.T * f
        z0 = np.concatenate([np.concatenate([U[:, 1:], U[:, -1:]], 1).reshape(n, -1), np.concatenate([X[:, 1:], X[:, -1:]], 1).reshape(n, -1)], 1)
        oc[:, :, 0] += oc[:, :, 3] * np.cos(oc[:, :, 2]) * cfg.T; oc[:, :, 1] += oc[:, :, 3] * np.sin(oc[:, :, 2]) * cfg.T
        if time.perf_counter() - t0 >= min_seconds:
            break
    dt = time.perf_counter() - t0
    return {"value": solved / dt, "unit": "solves/s", "cores": cores, "kind": "port",
            "sample": "%d scenes x %d of %d closed-loop steps, %.1f s wall (%.0f core-s)" % (n, done, sim_steps, dt, dt * cores),
            "note": "own FP64 C++ restatement (oracle) driving the same loop; CasADi+IPOPT baseline unavailable (casadi not installed)"}


class Group:
    """The ranks of this run.  world = 1: nothing to do.  world > 1 (or MPCB_BENCH_FORCE_DIST=1, the one-rank rehearsal): a
    communication handle joins the RCCL group inside libmpcbatch (mpcb_comm_init_rank); the 128-byte group id travels from rank
    0 to the others over a plain TCP socket on MASTER_PORT + 17."""

    def __init__(self, cfg, rank, world, device, force):
        from mpc_motion_planning_amd import sharding, solver
        self.rank, self.world, self.active = rank, world, (world > 1 or force)
        self.h = None
        if self.active:
            self.h = solver.BatchSolver(cfg, device=device)
            self.h.comm_init(sharding.exchange_unique_id(rank, world, solver.comm_unique_id), rank, world)

    def barrier(self):
        if self.active:
            self.h.allreduce([0.0], "sum")

    def reduce(self, values, op):
        return self.h.allreduce(values, op) if self.active else np.asarray(values, dtype=np.float64)


def closed_loop_bench(args, bs, cfg, x0, xs, obs, workload, grp, local_rank):
    """C5: `steps` // 10 (at least one) complete 80-step closed loops over all scenes; value counts solved MPC steps per second.
    The scenes are split over a few solver handles, each running its own closed loop from its own host thread (the steps of one
    loop depend on each other, independent loops overlap on the GPU like the launches of the other configurations)."""
    import threading
    from mpc_motion_planning_amd import _abi
    from mpc_motion_planning_amd.solver import BatchSolver, dims
    sim_steps = 80                                            # sim_time 8 s / T_S 0.1 (main_cbf_kin_c_sim.py:68,87)
    F = max(1, min(int(os.environ.get("MPCB_C5_LOOPS", "6")), args.inflight + 1))
    H = [bs] + [BatchSolver(cfg, device=local_rank) for _ in range(F - 1)]
    parts = np.array_split(np.arange(len(x0)), F)
    for h_ in H:
        h_.closed_loop_sampled(_abi.SCENES_C3, 256, 1, steps=4, obs_motion=_abi.OBSMOVE_PREDICTED)
        h_.timing(reset=True)
    res = [None] * F

    seed = 4000                                               # one Monte-Carlo population; rank r draws scenes [r*B, (r+1)*B) of it
    first = grp.rank * len(x0)

    def run(q, hold):
        # scenes are drawn on the device (mpcb_closed_loop_sampled: counter-based generator, global scene index), nothing but the
        # histories crosses PCIe
        res[q] = H[q].closed_loop_sampled(_abi.SCENES_C3, len(parts[q]), seed, first_index=first + int(parts[q][0]), steps=sim_steps,
                                          obs_motion=_abi.OBSMOVE_PREDICTED, hold_on_failure=hold)

    def loops(hold):
        th = [threading.Thread(target=run, args=(q, hold)) for q in range(F)]
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join()
        return {k: np.concatenate([r_[k] for r_ in res]) for k in ("status", "iters", "x_hist", "u_hist", "x0", "obs0")}

    grp.barrier()
    t0 = time.perf_counter()
    reps = max(1, args.steps // 10)
    for _ in range(reps):
        out = loops(False)
    for h_ in H:
        h_.sync()
    grp.barrier()
    dt = time.perf_counter() - t0
    tms = [h_.timing() for h_ in H]
    held = loops(True)                                        # the same scenes once more with the hold-and-shift fallback (untimed)
    status, iters = out["status"], out["iters"]
    x0, obs = out["x0"], out["obs0"]                          # the scenes the device drew
    xs = np.tile(x0[:1] * 0 + np.array([400.0, 3.5, 0.0, 30.0]), (len(x0), 1))
    B = len(x0)
    solved = int((status == 0).sum())

    def min_margin(run_):                                     # smallest obstacle-row value h along the executed trajectories
        X = run_["x_hist"]
        t = np.arange(sim_steps + 1)[None, :, None]
        ox = obs[:, None, :, 0] + obs[:, None, :, 3] * np.cos(obs[:, None, :, 2]) * cfg.T * t
        oy = obs[:, None, :, 1] + obs[:, None, :, 3] * np.sin(obs[:, None, :, 2]) * cfg.T * t
        sx = cfg.ego_hl + obs[:, None, :, 4] / 2 + cfg.safe_disl; sy = cfg.ego_hw + obs[:, None, :, 5] / 2 + cfg.safe_disw
        h = (X[:, :, None, 0] - ox) ** 2 / sx ** 2 + (X[:, :, None, 1] - oy) ** 2 / sy ** 2 - 1.0
        ok = (run_["status"] == 0).all(axis=1)
        return (float(h[ok].min()) if ok.any() else None), float(np.nanmin(h)), int(ok.sum()), int((np.nanmin(h, axis=(1, 2)) < -1e-6).sum())

    m_ok, m_all, n_ok, n_coll = min_margin(out)
    mh_ok, mh_all, nh_ok, nh_coll = min_margin(held)
    tot = grp.reduce([dt, 0.0], "max"); dt_max = float(tot[0])
    sums = grp.reduce([float(solved), float(B), float(n_ok)], "sum")
    nx, nz, ng = dims(cfg)
    launches = sum(t_["launches"] for t_ in tms)
    kernel_ms = sum(t_["total_ms"] for t_ in tms) / max(1, launches)
    per_launch = B // F
    abytes = algorithmic_bytes_per_solve(nx, nz, int(obs[0].size)) * per_launch      # obstacles predicted on device from [n_obs, 6]
    res_line = {
        "metric": "mpc_solves_per_sec", "value": sums[0] * reps / dt_max, "unit": "solves/s", "n_gpus": grp.world, "steps": reps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt_max / reps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": workload, "scenes_per_gpu": B, "sim_steps": sim_steps, "scenes_per_sec": sums[1] * reps / dt_max,
                   "solved_steps": int(sums[0]), "failed_steps": int(sums[1]) * sim_steps - int(sums[0]),
                   "failure_rate_steps": 1.0 - sums[0] / (sums[1] * sim_steps),
                   "status_histogram": {str(k): int(v) for k, v in enumerate(np.bincount(status.ravel(), minlength=7))},
                   "scenes_all_steps_solved": int(sums[2]), "min_h_scenes_all_solved": m_ok, "min_h_all_scenes": m_all, "scenes_with_collision": n_coll,
                   "with_hold_on_failure": {"scenes_all_steps_solved": nh_ok, "min_h_scenes_all_solved": mh_ok, "min_h_all_scenes": mh_all,
                                            "scenes_with_collision": nh_coll, "failed_steps": int((held["status"] != 0).sum())},
                   "iters_mean": float(iters.mean()), "iters_max": int(iters.max()), "scenes_drawn_on_device": True, "loops_in_flight": F,
                   "launches": launches, "restoration": bool(cfg.restoration), "collective": "none (scenes are independent for their whole horizon)",
                   "parity": PARITY_NOTE},
        "roofline": {"bound": "hbm", "achieved": abytes / (kernel_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": abytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                     "kernel": "mpcb_kernel_kin<3> + mpcb_kernel_kin_resto<3> (one solve = both passes)", "kernel_ms_avg": kernel_ms,
                     "algorithmic_bytes_per_launch": abytes,
                     "note": "compulsory I/O only; the solve is LDS-resident, bound by FP64 VALU issue and the serial Riccati chain (DESIGN.md §5)"},
    }
    if grp.rank == 0:
        if not args.no_cpu_baseline and grp.world == 1:
            res_line["cpu_baseline"] = cpu_baseline_closed_loop(cfg, x0, xs, obs, sim_steps)
        print(json.dumps(res_line))
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="instances per GPU per step (default: the config's BASELINE batch)")
    ap.add_argument("--config", default="C2", choices=["C2", "C3", "C4", "C5"],
                    help="BASELINE.json config: C2 (default, the metric's config) kin+1 static obstacle B=4096; C3 kin+3 predicted "
                         "obstacles B=32768; C4 dyn N=40 3 obstacles B=8192/GPU; C5 closed loop 80 steps (solves = scenes x steps)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-restoration", action="store_true", help="cfg.restoration = 0: a failed line search ends the solve (round-1 behaviour)")
    ap.add_argument("--inflight", type=int, default=6,
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
        raise SystemExit("--gpus %d needs one process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the "
                         "environment, e.g. python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d)" % (args.gpus, args.gpus, args.gpus))

    from mpc_motion_planning_amd import scenes, _abi
    from mpc_motion_planning_amd.solver import BatchSolver, default_config, dims

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
    if args.no_restoration:
        cfg.restoration = 0
    x0, xs, obs = sets[0]
    nx, nz, ng = dims(cfg)
    bs = BatchSolver(cfg, device=local_rank)
    grp = Group(cfg, rank, world, local_rank, os.environ.get("MPCB_BENCH_FORCE_DIST") == "1")
    if conf == "C5":
        return closed_loop_bench(args, bs, cfg, x0, xs, obs, workload, grp, local_rank)
    D = []                                                  # per batch: inputs and the status / iteration outputs, all resident in HBM
    for (a0, a1, a2) in sets:
        D.append(dict(x0=bs.device_array((B, nx)).upload(a0), xs=bs.device_array((B, nx)).upload(a1), obs=bs.device_array(a2.shape).upload(a2),
                      st=bs.device_array((B,), np.int32), it=bs.device_array((B,), np.int32)))
    d_x0 = D[0]["x0"]
    cyc = [0]                                               # step counter: step k solves batch k mod NB on handle k mod F
    F = max(1, args.inflight)
    H = [bs] + [BatchSolver(cfg, device=local_rank) for _ in range(F - 1)]     # one handle = one stream (include/mpcbatch.h)
    d_obj = [bs.device_array((B,)) for _ in range(F)]; d_kkt = [bs.device_array((B, 4)) for _ in range(F)]
    z_bufs = [bs.device_array((B, nz)) for _ in range(F)]
    z_all = grp.h.device_array((world * B, nz)) if grp.active else None

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

    def step(gather=True):
        k = cyc[0]; cyc[0] += 1
        d = D[k % NB]; hq = k % F
        if grp.active and gather:
            H[hq].wait_for(grp.h)                       # the gather that last read this handle's z buffer must be done
        H[hq].solve_device(B, d["x0"], d["xs"], d["obs"], obs_kind, d_z0, z_bufs[hq], d_obj[hq], d["st"], d["it"], d_kkt[hq])
        if grp.active and gather:
            # the all-gather of this step's trajectories runs on the communication handle's stream behind this solve, while the
            # younger solves of the other handles keep the GPU busy
            grp.h.wait_for(H[hq])
            grp.h.allgather(z_bufs[hq], z_all, B * nz)

    def fence():
        for h_ in H:
            h_.sync()
        if grp.active:
            grp.h.sync()
        grp.barrier()

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
    if grp.active:      # the gathered block of this rank must equal what its last solve wrote
        got = z_all.download()[rank * B:(rank + 1) * B]
        assert np.array_equal(got, z_bufs[(args.steps - 1) % F].download()), "all-gather result differs from the solver output"
    tms = [h_.timing() for h_ in H]
    tm = {"total_ms": sum(t_["total_ms"] for t_ in tms), "launches": sum(t_["launches"] for t_ in tms)}

    # every batch's status / iteration arrays hold its latest (identical, deterministic) result; weight by how often it ran
    uses = [args.steps // NB + (1 if q < args.steps % NB else 0) for q in range(NB)]
    st_b = [d["st"].download() for d in D]; it_b = [d["it"].download() for d in D]
    solved = sum(u * int((s_ == 0).sum()) for u, s_ in zip(uses, st_b))            # solved instances over all timed steps of this rank
    status = np.concatenate([s_ for u, s_ in zip(uses, st_b) if u]); iters = np.concatenate([i_ for u, i_ in zip(uses, it_b) if u])
    iters_per_launch = sum(u * float(i_.sum()) for u, i_ in zip(uses, it_b)) / max(1, args.steps)
    dt_nogather = None
    if grp.active:      # the same K steps once more without the gather, reported next to the headline value
        cyc[0] = 0
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step(gather=False)
        fence()
        dt_nogather = time.perf_counter() - t1
    mx = grp.reduce([dt, dt_nogather or 0.0], "max"); dt_max, dt_nogather_max = float(mx[0]), float(mx[1])
    solved_all = int(round(float(grp.reduce([float(solved)], "sum")[0])))

    if rank == 0:
        kernel_ms = tm["total_ms"] / max(1, tm["launches"])
        abytes = algorithmic_bytes_per_solve(nx, nz, int(obs[0].size)) * B
        achieved = abytes / (kernel_ms * 1e-3) / 1e9
        it_ok = iters[status == 0]
        flops = (147e3 if cfg.model == _abi.MODEL_DYN else 56e3) * iters_per_launch   # ~56 kflop (kin N=30) / ~147 kflop (dyn N=40) per iteration
        traffic, traffic_src = measured_traffic(workload)
        kk = ("dyn" if cfg.model == _abi.MODEL_DYN else "kin", 1 if cfg.n_obs <= 1 else 3)
        kname = ("mpcb_kernel_%s<%d>" % kk) + ((" + mpcb_kernel_%s_resto<%d> (one solve = first pass + restoration pass, timed together)" % kk) if cfg.restoration else "")
        out = {
            "metric": "mpc_solves_per_sec", "value": solved_all / dt_max, "unit": "solves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt_max / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload, "batch_per_gpu": B, "distinct_batches": NB,
                       "solved_per_step": solved_all / args.steps, "failed_per_step": world * B - solved_all / args.steps,
                       "status_histogram_rank0": {str(k): int(v) for k, v in enumerate(np.bincount(status, minlength=7))},
                       "iters_mean_solved": float(it_ok.mean()) if len(it_ok) else None, "iters_max": int(iters.max()),
                       "iters_share_of_unsolved": float(iters[status != 0].sum() / max(1, iters.sum())),
                       "restoration": bool(cfg.restoration),
                       "launches_in_flight": F, "tol": cfg.tol,
                       "collective": "RCCL all-gather of z per step inside libmpcbatch (mpcb_allgather), overlapped with the next steps' solves" if grp.active else "none",
                       "value_without_gather": (solved_all / dt_nogather_max) if dt_nogather else None,
                       "multi_gpu_note": "no N > 1 number has been measured on hardware by the builder (one-GPU boxes only); the driver's SCALE run is the measurement",
                       "parity": PARITY_NOTE},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "kernel": kname, "kernel_ms_avg": kernel_ms,
                         "algorithmic_bytes_per_launch": abytes,
                         "achieved_over_wall_clock": abytes * args.steps / dt_max / 1e9,   # launches overlap (launches_in_flight): each one lasts longer than a step
                         "note": "compulsory I/O is 3072 B/solve; the solve is LDS-resident, bound by FP64 VALU issue and the serial "
                                 "Riccati chain, not by HBM (SURVEY.md F10, DESIGN.md §5)"},
            "roofline_fp64": {"bound": "fp64_valu", "achieved": flops * args.steps / dt_max / 1e12, "peak": FP64_VECTOR_PEAK_TFLOPS,
                              "unit": "TFLOP/s", "frac": flops * args.steps / dt_max / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                              "model": "56 kflop (kin) / 147 kflop (dyn) x interior-point iterations summed over the batch, per step, over the "
                                       "wall clock of the timed region (launches overlap, so a launch lasts longer than a step)"},
        }
        if not args.no_cpu_baseline and world == 1:      # rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(cfg, x0, xs, obs)
        print(json.dumps(out))
    grp.barrier()


if __name__ == "__main__":
    main()
