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
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md, chip-level parameters)
FP64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X FP64 vector peak (spec)
PARITY_NOTE = ("parity UNPINNED vs the reference's CasADi+IPOPT (not installed here nor on the GPU box, no golden outputs in the "
               "reference): results are checked against this repo's CPU oracle (same NLP, IPOPT's published algorithm) and an "
               "independent KKT certificate (oracle/kkt_check.py)")


def measured_counters(workload, settings=None):
    """Counters of this workload from the rocprofv3 PMC passes of tools/profile_round.sh (separate passes, never mixed with tracing),
    kept in profiles/traffic.json: HBM bytes per solve launch (FETCH_SIZE / WRITE_SIZE, KiB units, 2x correction on FETCH_SIZE as
    MI355X_MICROARCH.md prescribes) and FP64 operations per solve launch (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64: wave instructions
    x 64 lanes, FMA counted twice).  A process cannot collect PMC counters on itself, so this is the committed OFFLINE measurement
    of the same command; the entry of the latest round wins.  {} if absent."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        t = json.load(open(p))
        hits = [e for e in (t if isinstance(t, list) else [t]) if e.get("workload_key") == workload.split(":")[0]]
        if settings is not None:      # counters of a run with other solver settings say nothing about this one
            hits = [e for e in hits if e.get("workload") == workload and all(e.get("settings", {}).get(k) == v for k, v in settings.items())]
        if hits:
            e = hits[-1]
            return {"bytes_per_launch": e.get("bytes_per_launch"), "fp64_flop_per_launch": e.get("fp64_flop_per_launch"),
                    "valu_busy": e.get("valu_busy"), "wait_any": e.get("wait_any"), "round": e.get("round"),
                    "source": "offline: profiles/traffic.json (rocprofv3 --pmc passes of this command at --inflight 1, committed)"}
    except (OSError, ValueError, KeyError):
        pass
    return {}


def parity_evidence():
    """The numbers of the independent-evidence tests of the GPU suite (tests/test_parity_evidence.py writes them on the GPU box,
    the committed copy is profiles/parity_evidence.json): same-basin fractions against SciPy SLSQP, the IPOPT-default-start batches,
    the audit of the MPCB_ST_INFEASIBLE verdicts.  None if the file is absent."""
    for p in (os.path.join(ROOT, "gpurun_out", "parity_evidence.json"), os.path.join(ROOT, "profiles", "parity_evidence.json")):
        try:
            return json.load(open(p))
        except (OSError, ValueError):
            continue
    return None


def kernel_launches_per_solve(cfg, with_z0):
    """Kernel launches behind one mpcb_solve_device call, as mpcb_api.hip's launch_solve orders them (used by tools/summarize_profile.py to
    add up the counters of one solve): first attempt, [its restoration pass], [second start — none where the kernel runs it inside the first
    launch: the kin and dyn instantiations for up to three obstacles (no general-gamma rows, no RK4) with a second start of kind 1], [restoration pass]."""
    from mpc_motion_planning_amd import _abi
    second = bool(cfg.second_start) and bool(cfg.init_rollout)
    kind = (2 if with_z0 else 1) if cfg.second_start == 3 else int(cfg.second_start)
    gen = cfg.model == _abi.MODEL_KIN and cfg.obs_mode == _abi.OBS_DCBF and cfg.gamma < 1.0 - 1e-12 and cfg.n_obs > 0
    fused = second and kind == 1 and cfg.n_obs <= 3 and (cfg.model == _abi.MODEL_DYN or (not gen and cfg.integrator != _abi.INT_RK4))
    n = 1
    if cfg.restoration and not (second and kind == 1):
        n += 1
    if second:
        n += (0 if fused else 1) + (1 if cfg.restoration else 0)
    return n


def algorithmic_bytes_per_solve(nx, nz, n_obs_values, with_z0=True):
    # 8*(2*nx + nz_in + obs_in + nz_out) + 16   (status i32 + iters i32 + obj f64).  Without a start vector
    # (z0 = NULL: the cold start of main_cbf_kin_c_sim.py:47-50) the nz_in term is not compulsory: nothing is read.
    return 8 * (2 * nx + (nz if with_z0 else 0) + n_obs_values + nz) + 16


def host_cores():
    """The cores this process may actually use: its affinity mask, cut to the CPU quota of its control group when there is one (on the
    one-GPU boxes the mask shows every core of the host, 256, while the container's share is 16: 256 OpenMP threads on 16 cores'
    worth of quota ran the oracle three times SLOWER than 16 threads).  MPCB_CPU_THREADS overrides."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]           # cgroup v2
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota:
        cores = min(cores, max(1, int(round(quota))))
    elif cores > 64 and not os.environ.get("MPCB_CPU_THREADS"):
        cores = 16          # no quota visible but a whole host's mask: the documented CPU share of a one-GPU box
    if os.environ.get("MPCB_CPU_THREADS"):
        cores = int(os.environ["MPCB_CPU_THREADS"])
    return max(1, cores)


def cpu_baseline(cfg, x0, xs, obs, min_seconds=1.5):
    """The CPU oracle (oracle/mpc_oracle.cpp, same NLP, same algorithm, OpenMP over the batch) timed on this box's
    host cores on a bounded sample of the same workload.  Reported next to the GPU number; CasADi+IPOPT cannot be
    timed (not installed here nor on the GPU box, no network: SURVEY.md §0 F2)."""
    from oracle import oracle
    native = oracle.use_native_build()
    cores = host_cores()
    n = min(len(x0), 2048)
    w = min(len(x0), 4 * cores)
    oracle.solve(cfg, x0[:w], xs[:w], obs[:w], threads=cores, want_multipliers=False)   # untimed: starts the OpenMP team, first-touch of the per-thread arenas
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
            "build": ("g++ %s -fopenmp, compiled on this host" % native) if native else "g++ -O2 -fopenmp (portable build; no compiler on this host)",
            "threads": "affinity mask cut to the cgroup CPU quota (%d)" % cores,
            "sample": "%d x first %d instances of the step's batch, OpenMP over instances, %.1f s wall (%.0f core-s); latency: %d instances on one thread" % (reps, n, dt, dt * cores, m),
            "note": "own FP64 C++ restatement of the NLP + IPOPT-style solver; CasADi+IPOPT baseline unavailable (casadi not installed)"}


def cpu_baseline_closed_loop(cfg, x0, xs, obs, sim_steps, min_seconds=8.0):
    """C5 on the host: the oracle driving the same receding-horizon loop (solve, plant step, shift, obstacle advance and
    re-prediction: main_cbf_kin_c_sim_pre.py:86-126) on a bounded sample of the scenes, OpenMP over scenes inside every step."""
    from oracle import oracle
    from mpc_motion_planning_amd import scenes
    native = oracle.use_native_build()
    cores = host_cores()
    n = min(len(x0), 768); N = cfg.N
    xc = x0[:n].copy(); oc = obs[:n].copy(); z0 = np.zeros((n, 2 * N + 4 * (N + 1)))
    w = min(n, 4 * cores)
    oracle.solve(cfg, xc[:w], xs[:w], scenes.predict_obstacles(oc[:w], cfg.T, N), threads=cores, want_multipliers=False)   # untimed warm-up of the OpenMP team
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
            "build": ("g++ %s -fopenmp, compiled on this host" % native) if native else "g++ -O2 -fopenmp (portable build)",
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
    ap.add_argument("--inflight", type=int, default=16,
                    help="launch lanes of the solver handle (mpcb_set_inflight): a batch is cut into that many chunks, chunk c of step k+1 "
                         "starts when chunk c of step k has finished, while the slowest instances of the other chunks still run (a launch "
                         "ends with its slowest instance and leaves most SIMDs idle before that); 1 = strictly one launch at a time")
    ap.add_argument("--handles", type=int, default=1, help="independent solver handles used round-robin (rounds 1-2 overlapped launches this way)")
    ap.add_argument("--batches", type=int, default=5,
                    help="distinct synthetic batches resident in HBM, used round-robin by the steps (the launch time of a 4096-instance "
                         "batch moves +-15 %% with where its slowest instances fall in the dispatch order; one batch would report one draw)")
    ap.add_argument("--start-steer", type=float, default=None,
                    help="cfg.start_steer [rad] (default: what mpcb_default_config ships, 0.03): the slight constant turn of a cold start whose straight "
                         "roll-out runs into an obstacle row; 0 = the straight roll-out of rounds 1-2")
    ap.add_argument("--second-start", type=int, default=None, choices=[0, 1, 2, 3],
                    help="cfg.second_start (default: what mpcb_default_config ships, 3 = by the kind of start: 1 for a cold start, 2 with a start "
                         "vector): 0 = one attempt per instance (round-2 behaviour), 1 = second attempt instead of the first attempt's restoration "
                         "phase, 2 = after it")
    ap.add_argument("--integrator", default="euler", choices=["euler", "rk4"],
                    help="shooting rows: explicit Euler (the reference's NLP, kin.py:207: the headline) or the RK4 instantiations (kinematic configurations)")
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
    if args.second_start is not None:
        cfg.second_start = args.second_start
    if args.start_steer is not None:
        cfg.start_steer = args.start_steer
    if args.integrator == "rk4":
        cfg.integrator = _abi.INT_RK4
        workload += " [RK4 shooting rows]"
    x0, xs, obs = sets[0]
    nx, nz, ng = dims(cfg)
    bs = BatchSolver(cfg, device=local_rank)
    grp = Group(cfg, rank, world, local_rank, os.environ.get("MPCB_BENCH_FORCE_DIST") == "1")
    if conf == "C5":
        return closed_loop_bench(args, bs, cfg, x0, xs, obs, workload, grp, local_rank)
    # One solver handle with `--inflight` launch lanes (mpcb_set_inflight): consecutive asynchronous solves of ONE handle go to its
    # lanes in turn and overlap inside the library.  --handles H > 1 adds independent handles (how rounds 1-2 overlapped launches).
    F = max(1, args.inflight)
    bs.set_inflight(F)
    H = [bs] + [BatchSolver(cfg, device=local_rank, inflight=F) for _ in range(max(1, args.handles) - 1)]
    HN = len(H)
    R = F                                                   # ring of output-buffer sets per handle: solves closer than F apart may run concurrently
    if HN * R > 32:
        raise SystemExit("--handles x --inflight must be <= 32 (event slots of the gather ring)")
    D = []                                                  # per batch: the inputs, resident in HBM
    for (a0, a1, a2) in sets:
        D.append(dict(x0=bs.device_array((B, nx)).upload(a0), xs=bs.device_array((B, nx)).upload(a1), obs=bs.device_array(a2.shape).upload(a2)))
    d_x0 = D[0]["x0"]
    cyc = [0]                                               # step counter: step k solves batch k mod NB on handle k mod HN, output set (k div HN) mod R
    OUT = [[dict(z=bs.device_array((B, nz)), obj=bs.device_array((B,)), kkt=bs.device_array((B, 4)), st=bs.device_array((B,), np.int32),
                 it=bs.device_array((B,), np.int32)) for _ in range(R)] for _ in range(HN)]
    z_all = grp.h.device_array((world * B, nz)) if grp.active else None
    last = {}                                               # batch -> the output set its latest solve wrote

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
        d = D[k % NB]; hq = k % HN; sl = (k // HN) % R; mark = hq * R + sl
        o = OUT[hq][sl]; last[k % NB] = o; last["z"] = o["z"]
        if grp.active and gather:
            H[hq].wait_mark(grp.h, mark)                # only the gather that last read THIS z buffer must be done (R steps ago), not the latest one
        H[hq].solve_device(B, d["x0"], d["xs"], d["obs"], obs_kind, d_z0, o["z"], o["obj"], o["st"], o["it"], o["kkt"])
        if grp.active and gather:
            # the all-gather of this step's trajectories runs on the communication handle's stream behind this solve, while the
            # younger solves keep the GPU busy
            H[hq].record(mark)
            grp.h.wait_mark(H[hq], mark)
            grp.h.allgather(o["z"], z_all, B * nz)
            grp.h.record(mark)

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
        assert np.array_equal(got, last["z"].download()), "all-gather result differs from the solver output"
    tms = [h_.timing() for h_ in H]
    tm = {"total_ms": sum(t_["total_ms"] for t_ in tms), "launches": sum(t_["launches"] for t_ in tms)}

    # every batch's status / iteration arrays hold its latest (identical, deterministic) result; weight by how often it ran
    uses = [args.steps // NB + (1 if q < args.steps % NB else 0) for q in range(NB)]
    # (deterministic solves: a batch's result is the same every time it runs; read it from the set its latest solve wrote, unless a
    # later step has reused that set — then from a fresh synchronous solve)
    st_b = []; it_b = []
    live = {id(OUT[k % HN][(k // HN) % R]): k % NB for k in range(args.steps)}           # set -> batch of the last step that wrote it
    for q in range(NB):
        o = last.get(q)
        if o is None or live.get(id(o)) != q:
            o = OUT[0][0]
            bs.solve_device(B, D[q]["x0"], D[q]["xs"], D[q]["obs"], obs_kind, d_z0, o["z"], o["obj"], o["st"], o["it"], o["kkt"], sync=True)
        st_b.append(o["st"].download()); it_b.append(o["it"].download())
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

    # end to end through the host-pointer entry (mpcb_solve: pageable H2D of the inputs + solve + D2H of z, obj, status, iters, kkt),
    # the PCIe-inclusive rate; never `value`
    e2e = None
    if rank == 0 and not args.warm:
        bs.solve_batch(x0, xs, obs)
        t1 = time.perf_counter(); reps_e = 3; ok_e = 0
        for _ in range(reps_e):
            ok_e += int((bs.solve_batch(x0, xs, obs)["status"] == 0).sum())
        dte = (time.perf_counter() - t1) / reps_e
        e2e = {"value": ok_e / reps_e / dte, "unit": "solves/s", "ms_per_call": 1e3 * dte,
               "what": "mpcb_solve with host pointers, one call at a time: H2D of x0/xs/obs + all launches of the solve + D2H of z, obj, status, iters, kkt"}

    if rank == 0:
        launches = max(1, tm["launches"])
        kernel_ms = tm["total_ms"] / launches
        inst_per_launch = B * args.steps / launches                                # a launch = one solve call on one lane (all its passes)
        with_z0 = d_z0 is not None
        bytes_solve = algorithmic_bytes_per_solve(nx, nz, int(obs[0].size), with_z0)
        abytes = bytes_solve * inst_per_launch
        achieved = abytes / (kernel_ms * 1e-3) / 1e9
        it_ok = iters[status == 0]
        flop_iter = 147e3 if cfg.model == _abi.MODEL_DYN else 56e3                  # ~56 kflop (kin N=30) / ~147 kflop (dyn N=40) per iteration
        flops_model = flop_iter * iters_per_launch / B * inst_per_launch            # model flops of one launch
        ctr = measured_counters(workload, {"second_start": int(cfg.second_start), "restoration": bool(cfg.restoration), "start_steer": float(cfg.start_steer),
                                           "integrator": "rk4" if cfg.integrator == _abi.INT_RK4 else "euler"})
        flops_ctr = (ctr["fp64_flop_per_launch"] / B * inst_per_launch) if ctr.get("fp64_flop_per_launch") else None
        traffic = (ctr["bytes_per_launch"] / B * inst_per_launch) if ctr.get("bytes_per_launch") else None
        flops_used = flops_ctr if flops_ctr else flops_model
        kk = ("dyn" if cfg.model == _abi.MODEL_DYN else "kin", 1 if cfg.n_obs <= 1 else 3)
        kname = ("mpcb_kernel_%s<%d>" % kk) + ((" + mpcb_kernel_%s_resto<%d>" % kk) if cfg.restoration else "") + \
            " (one solve = its launches on one lane, timed together: first attempt, second start, restoration pass)"
        out = {
            "metric": "mpc_solves_per_sec", "value": solved_all / dt_max, "unit": "solves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt_max / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload, "batch_per_gpu": B, "distinct_batches": NB,
                       "solved_per_step": solved_all / args.steps, "failed_per_step": world * B - solved_all / args.steps,
                       "status_histogram_rank0": {str(k): int(v) for k, v in enumerate(np.bincount(status, minlength=7))},
                       "iters_mean_solved": float(it_ok.mean()) if len(it_ok) else None, "iters_max": int(iters.max()),
                       "iters_share_of_unsolved": float(iters[status != 0].sum() / max(1, iters.sum())),
                       "restoration": bool(cfg.restoration), "second_start": int(cfg.second_start), "start_steer": float(cfg.start_steer),
                       "integrator": "rk4" if cfg.integrator == _abi.INT_RK4 else "euler",
                       "kernel_launches_per_solve": kernel_launches_per_solve(cfg, d_z0 is not None),
                       "solver_handles": HN, "launch_lanes_per_handle": F, "launches_per_step": launches / args.steps, "tol": cfg.tol,
                       "collective": "RCCL all-gather of z per step inside libmpcbatch (mpcb_allgather), overlapped with the next steps' solves" if grp.active else "none",
                       "value_without_gather": (solved_all / dt_nogather_max) if dt_nogather else None,
                       "multi_gpu_note": "no N > 1 number has been measured on hardware by the builder (one-GPU boxes only); the driver's SCALE run is the measurement",
                       "parity": PARITY_NOTE, "parity_evidence": parity_evidence()},
            # what binds: FP64 VALU issue (one wave per SIMD).  flops per launch from the committed PMC pass when there is one
            # (SQ_INSTS_VALU_{ADD,MUL,TRANS}_F64 x 64 + FMA_F64 x 128), the 56/147 kflop-per-iteration model next to it
            "roofline": {"bound": "fp64_valu_issue", "achieved": flops_used / (kernel_ms * 1e-3) / 1e12, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": flops_used / (kernel_ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                         "flop_per_launch": flops_used, "flop_source": ("counters, " + ctr["source"]) if flops_ctr else "model (no committed PMC pass for this workload)",
                         "flop_per_launch_model": flops_model, "traffic": traffic, "traffic_source": ctr.get("source"),
                         "kernel": kname, "kernel_ms_avg": kernel_ms, "instances_per_launch": inst_per_launch,
                         "achieved_over_wall_clock": flops_used * launches / dt_max / 1e12,   # launches overlap: each one lasts longer than its share of a step
                         "frac_over_wall_clock": flops_used * launches / dt_max / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                         "valu_busy": ctr.get("valu_busy"), "wait_any": ctr.get("wait_any"),
                         "note": "peak = FP64 vector rate of the chip (256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz); the solve is LDS-resident and "
                                 "bound by instruction issue at one wave per SIMD, not by HBM (SURVEY.md F10, DESIGN.md §5)"},
            "roofline_hbm": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                             "traffic": traffic, "algorithmic_bytes_per_launch": abytes, "algorithmic_bytes_per_solve": bytes_solve,
                             "achieved_over_wall_clock": abytes * launches / dt_max / 1e9,
                             "note": "compulsory I/O of the inputs actually passed (%s z0): x0, xs, obs in; z, obj, status, iters out" % ("with" if with_z0 else "no")},
            "end_to_end": e2e,
        }
        if not args.no_cpu_baseline and world == 1:      # rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(cfg, x0, xs, obs)
        print(json.dumps(out))
    grp.barrier()


if __name__ == "__main__":
    main()
