"""Independent evidence for the solve itself, in the GPU suite (the round-2 verdict's first item).  Parity with the reference's own
solver stays UNPINNED — CasADi + IPOPT is installed neither here nor on the GPU box and the reference holds no result files — so
these tests measure what can be measured with what IS importable (SciPy), on the DEVICE results:

  (a) IPOPT's own start (the X guess taken as given, mu_init = 0.1, z0 = 0: what `solver(x0=init_control, ...)` of
      main_cbf_kin_c_sim.py:92,100 hands to IPOPT) on 256-instance C2 / C3 batches: HIP path against the oracle, and the fraction
      of instances that end at the SAME trajectory as the batch default (roll-out start, mu_init = 10);
  (b) an audit of the MPCB_ST_INFEASIBLE verdicts: SciPy SLSQP on the pure feasibility problem of the reference-form NLP
      (oracle/kkt_check.py), from the device's last iterate and from a start biased to the free side of the road;
  (c) the same-basin fraction against SciPy SLSQP on 48 / 48 / 24 solved instances of C2 / C3 / C4.

The numbers go to gpurun_out/parity_evidence.json (committed copy: profiles/parity_evidence.json), which bench.py attaches to its
line as config.parity_evidence."""
import json
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

from mpc_motion_planning_amd import scenes, _abi
from mpc_motion_planning_amd.solver import default_config

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKERS = max(2, min(12, (os.cpu_count() or 4) - 2))


@pytest.fixture(scope="module")
def evidence():
    ev = {}
    yield ev
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", "parity_evidence.json")
    old = {}
    try:
        old = json.load(open(path))
    except (OSError, ValueError):
        pass
    old.update(ev)
    json.dump(old, open(path, "w"), indent=1, sort_keys=True)


def _pool():
    # spawn: the workers must not inherit a process image in which the HIP runtime is initialised; they never touch the GPU
    return mp.get_context("spawn").Pool(WORKERS)


def _sample(conf, B, seed):
    if conf in ("C2", "C2rk4"):
        cfg = default_config(N=30, n_obs=1); x0, xs, obs = scenes.sample_c2(B, seed=seed)
        if conf == "C2rk4":
            cfg.integrator = _abi.INT_RK4                      # Runge-Kutta shooting rows (BASELINE.json's north_star; opt-in)
    elif conf == "C3":
        cfg = default_config(N=30, n_obs=3); x0, xs, _, obs = scenes.sample_c3(B, N=30, dt=0.1, seed=seed)
    else:
        cfg = default_config(model=_abi.MODEL_DYN, N=40, n_obs=3); x0, xs, obs = scenes.sample_c4(B, seed=seed, n_obs=3)
    return cfg, x0, xs, obs


# ---------------------------------------------------------------------------------------------------------------- (a)
@pytest.mark.parametrize("conf", ["C2", "C3"])
def test_ipopt_default_start_on_batches(gpu_solver_factory, oracle_mod, evidence, conf):
    B = 256
    cfg, x0, xs, obs = _sample(conf, B, seed=301 if conf == "C2" else 302)
    ci = cfg.copy(); ci.init_rollout = 0; ci.mu_init = 0.1                        # IPOPT's documented defaults, X guess as given
    g_i = gpu_solver_factory(ci).solve_batch(x0, xs, obs)                        # z0 = None = zeros (main_cbf_kin_c_sim.py:47-50)
    o_i = oracle_mod.solve(ci, x0, xs, obs)
    g_d = gpu_solver_factory(cfg).solve_batch(x0, xs, obs)
    # HIP path vs oracle under the IPOPT start: long paths (50 iterations on average) through a non-convex NLP, so two roundings of
    # one algorithm part ways on a few instances; what must hold: nearly all statuses equal, equal trajectories where both solve
    same = (g_i["status"] == o_i["status"]).mean()
    both = (g_i["status"] == 0) & (o_i["status"] == 0)
    d_io = np.abs(g_i["z"] - o_i["z"]).max(axis=1)
    assert same >= 0.93 and both.sum() >= 0.5 * B
    assert (d_io[both] <= 1e-5).mean() >= 0.97, "HIP vs oracle under the IPOPT start: %d of %d beyond 1e-5" % ((d_io[both] > 1e-5).sum(), both.sum())
    # same basin as the batch default?
    bd = (g_i["status"] == 0) & (g_d["status"] == 0)
    d_id = np.abs(g_i["z"] - g_d["z"]).max(axis=1)
    frac = float((d_id[bd] <= 1e-4).mean())
    lower = float((g_i["obj"][bd] < g_d["obj"][bd] * (1 - 1e-6))[d_id[bd] > 1e-4].mean()) if (d_id[bd] > 1e-4).any() else None
    evidence["ipopt_start_" + conf] = {
        "instances": B, "solved_ipopt_start": int((g_i["status"] == 0).sum()), "solved_default_start": int((g_d["status"] == 0).sum()),
        "solved_both": int(bd.sum()), "same_trajectory_1e-4": frac, "status_equal_hip_vs_oracle": float(same),
        "of_the_differing_ipopt_start_has_lower_objective": lower,
        "iters_mean_ipopt_start": float(g_i["iters"].mean()), "iters_mean_default_start": float(g_d["iters"].mean()),
        "status_histogram_ipopt_start": np.bincount(g_i["status"], minlength=9).tolist()}
    print(conf, evidence["ipopt_start_" + conf])
    assert 0.5 <= frac <= 1.0


# ---------------------------------------------------------------------------------------------------------------- (b)
def _audit_one(args):
    sys.path.insert(0, ROOT)
    from oracle import kkt_check, scipy_crosscheck as sc
    x0, xs, ob, z_dev = args[:4]
    lanes = args[4] if len(args) > 4 else (0.0,)            # lateral positions the constructed starts are drawn to
    nlp = kkt_check.KinNlp(30, 0.1, x0, xs, ob)
    N = nlp.N
    best = (np.inf, np.inf)
    # start 1: the device's last iterate; start 2: controls 0, X rolled out at constant speed with y drawn to the side of the road the
    # obstacle leaves free (y = 0: the shipped obstacle occupies y in [1.2, 5.8] of the road [-1, 5]); start 3: the same with the
    # speed braked to 0 as fast as the acceleration bound allows
    starts = [z_dev]
    for ylane in lanes:
        for brake in (False, True):
            X = np.zeros((N + 1, 4)); X[0] = x0; U = np.zeros((N, 2))
            for k in range(N):
                a = -3.0 if (brake and X[k, 3] > 0.35) else 0.0
                U[k, 1] = a
                X[k + 1] = X[k] + 0.1 * np.array([X[k, 3] * np.cos(X[k, 2]), X[k, 3] * np.sin(X[k, 2]), 0.0, a])
            X[1:, 1] = x0[1] + (ylane - x0[1]) * np.minimum(1.0, np.arange(1, N + 1) / 10.0)
            starts.append(np.concatenate([U.reshape(-1), X.reshape(-1)]))
    for z0 in starts:
        S, viol, _ = sc.min_violation_slsqp(nlp, z0, maxiter=250)
        if viol < best[1]:
            best = (S, viol)
    return best


def test_infeasible_verdicts_audited_by_an_independent_solver(gpu_solver_factory, evidence):
    """Round 2 shipped ONE attempt per instance and labelled 17 % of the C2 workload MPCB_ST_INFEASIBLE.  The audit: SLSQP finds a
    feasible point for nearly every one of them — those verdicts are local statements about the path the solver took (as IPOPT's
    "Converged to a point of local infeasibility" is), not about the instance.  That is what cfg.second_start answers: the same
    instances solved once more from the reference's own start z = 0.  Both halves are measured here: (1) the one-attempt verdicts
    and how many SLSQP overturns, (2) what the shipped configuration leaves unsolved, audited the same way."""
    cfg, x0, xs, obs = _sample("C2", 1024, seed=303)
    one = cfg.copy(); one.second_start = 0; one.start_steer = 0.0           # the round-2 solver: straight roll-out start, one attempt
    r1 = gpu_solver_factory(one).solve_batch(x0, xs, obs)
    idx = np.nonzero(r1["status"] == _abi.ST_INFEASIBLE)[0][:128]
    assert len(idx) == 128, "only %d MPCB_ST_INFEASIBLE verdicts in 1024 C2 scenes" % len(idx)
    ones = cfg.copy(); ones.second_start = 0                                 # one attempt from the shipped start (cfg.start_steer)
    r1s = gpu_solver_factory(ones).solve_batch(x0, xs, obs)
    idx_s = np.nonzero(r1s["status"] == _abi.ST_INFEASIBLE)[0][:128]
    r2 = gpu_solver_factory(cfg).solve_batch(x0, xs, obs)                    # the shipped configuration (second start on)
    left = np.nonzero(r2["status"] != 0)[0]
    with _pool() as p:
        res = p.map(_audit_one, [(x0[i], xs[i], obs[i], r1["z"][i]) for i in idx], chunksize=2)
        res_s = p.map(_audit_one, [(x0[i], xs[i], obs[i], r1s["z"][i]) for i in idx_s], chunksize=1) if len(idx_s) else []
        res2 = p.map(_audit_one, [(x0[i], xs[i], obs[i], r2["z"][i]) for i in left], chunksize=1) if len(left) else []
    viol = np.array([v for _, v in res]); viol_s = np.array([v for _, v in res_s]) if len(res_s) else np.zeros(0)
    viol2 = np.array([v for _, v in res2]) if len(res2) else np.zeros(0)
    wrong = int((viol <= 1e-8).sum())
    evidence["infeasible_audit_C2"] = {
        "one_attempt_straight_start": {"what": "the round-2 solver: start_steer = 0, second_start = 0", "solved": int((r1["status"] == 0).sum()),
                                       "verdicts_audited": len(idx), "feasible_point_found_by_slsqp": wrong, "overturned_rate": wrong / len(idx),
                                       "solved_by_the_shipped_configuration": int((r2["status"][idx] == 0).sum())},
        "one_attempt": {"what": "the shipped start (start_steer), second_start = 0", "solved": int((r1s["status"] == 0).sum()),
                        "infeasible_verdicts": int((r1s["status"] == _abi.ST_INFEASIBLE).sum()), "verdicts_audited": len(idx_s),
                        "feasible_point_found_by_slsqp": int((viol_s <= 1e-8).sum()),
                        "solved_by_the_second_start": int((r2["status"][idx_s] == 0).sum())},
        "shipped_configuration": {"instances": len(x0), "solved": int((r2["status"] == 0).sum()), "unsolved": int(len(left)),
                                  "unsolved_with_a_feasible_point_by_slsqp": int((viol2 <= 1e-8).sum()),
                                  "status_histogram": np.bincount(r2["status"], minlength=9).tolist()},
        "method": "SciPy SLSQP, min sum(s) with elastic obstacle rows on the reference-form NLP (oracle/kkt_check.py), 3 starts"}
    print(evidence["infeasible_audit_C2"])
    # the shipped configuration must overturn what the audit overturns (it solves the NLP, SLSQP only finds a feasible point), and may
    # leave at most 2 % of the scenes unsolved; the turned start alone must already solve most of what one attempt lost
    assert (r2["status"][idx] == 0).sum() >= 0.9 * wrong and len(left) <= 0.02 * len(x0)
    assert (r1s["status"] == 0).sum() >= (r1["status"] == 0).sum() + 0.5 * (r1["status"] != 0).sum()
    # what one attempt solves WITHOUT entering its restoration phase is untouched, bit for bit (with second_start = 1 the first attempt's
    # restoration phase is skipped in favour of the second start; second_start = 2 keeps every one-attempt result)
    two = cfg.copy(); two.second_start = 2
    r3 = gpu_solver_factory(two).solve_batch(x0, xs, obs)
    assert np.array_equal(r3["z"][r1s["status"] == 0], r1s["z"][r1s["status"] == 0]) and (r3["status"] == 0).sum() >= (r2["status"] == 0).sum() - 2


def test_unsolved_c3_instances_audited_by_an_independent_solver(gpu_solver_factory, evidence):
    """The same audit on C3 (three predicted moving obstacles on two lanes, 2.7 % of the bench workload ends unsolved): what the shipped
    configuration leaves unsolved in 1024 scenes, SLSQP on the pure feasibility problem from the device's iterate and from constructed
    paths on either lane, braked and unbraked.  Reported, not asserted beyond sanity: scenes with three obstacles across two lanes can be
    infeasible for real."""
    cfg, x0, xs, obs = _sample("C3", 1024, seed=304)
    r = gpu_solver_factory(cfg).solve_batch(x0, xs, obs)
    left = np.nonzero((r["status"] != 0) & (r["status"] != _abi.ST_INFEASIBLE_X0))[0][:48]
    with _pool() as p:
        res = p.map(_audit_one, [(x0[i], xs[i], obs[i], r["z"][i], (0.0, 3.5)) for i in left], chunksize=1) if len(left) else []
    viol = np.array([v for _, v in res]) if len(res) else np.zeros(0)
    evidence["unsolved_audit_C3"] = {"instances": len(x0), "solved": int((r["status"] == 0).sum()), "unsolved": int((r["status"] != 0).sum()),
                                     "audited": int(len(left)), "feasible_point_found_by_slsqp": int((viol <= 1e-8).sum()),
                                     "status_histogram": np.bincount(r["status"], minlength=9).tolist(),
                                     "method": "as infeasible_audit_C2, 5 starts (device iterate, both lanes x braked / unbraked)"}
    print(evidence["unsolved_audit_C3"])
    assert (r["status"] == 0).mean() >= 0.95 and len(left) >= 1


def _audit_one_dyn(args):
    sys.path.insert(0, ROOT)
    from oracle import kkt_check, scipy_crosscheck as sc
    x0, xs, ob, z_dev = args
    nlp = kkt_check.DynNlp(40, 0.1, x0, xs, ob)
    N = nlp.N
    starts = [z_dev]
    for ylane in (-0.5, 1.75, 4.5):                           # between / beside the obstacle rows of the C4 scenes, inside the y box [-1, 5]
        for brake in (False, True):
            X = np.zeros((N + 1, 6)); X[0] = x0; U = np.zeros((N, 2))
            for k in range(N):
                a = -3.0 if (brake and X[k, 3] > 1.0) else 0.0
                U[k, 1] = a
                X[k + 1] = X[k] + 0.1 * np.array([X[k, 3] * np.cos(X[k, 2]), X[k, 3] * np.sin(X[k, 2]), 0.0, a, 0.0, 0.0])
            X[1:, 1] = x0[1] + (ylane - x0[1]) * np.minimum(1.0, np.arange(1, N + 1) / 15.0)
            starts.append(np.concatenate([U.reshape(-1), X.reshape(-1)]))
    best = (np.inf, np.inf)
    for z0 in starts:
        S, viol, _ = sc.min_violation_slsqp(nlp, z0, maxiter=200)
        if viol < best[1]:
            best = (S, viol)
        if best[1] <= 1e-8:
            break
    return best


def test_unsolved_c4_instances_audited_by_an_independent_solver(gpu_solver_factory, evidence):
    """The audit on C4 (dynamic bicycle, N = 40, three static obstacles; 0.8 % of the bench workload ends unsolved), reference row form
    sqrt(h) >= 1 (oracle/kkt_check.DynNlp): what the shipped configuration leaves unsolved in 1024 scenes."""
    cfg, x0, xs, obs = _sample("C4", 1024, seed=305)
    r = gpu_solver_factory(cfg).solve_batch(x0, xs, obs)
    left = np.nonzero((r["status"] != 0) & (r["status"] != _abi.ST_INFEASIBLE_X0))[0][:24]
    with _pool() as p:
        res = p.map(_audit_one_dyn, [(x0[i], xs[i], obs[i], r["z"][i]) for i in left], chunksize=1) if len(left) else []
    viol = np.array([v for _, v in res]) if len(res) else np.zeros(0)
    evidence["unsolved_audit_C4"] = {"instances": len(x0), "solved": int((r["status"] == 0).sum()), "unsolved": int((r["status"] != 0).sum()),
                                     "audited": int(len(left)), "feasible_point_found_by_slsqp": int((viol <= 1e-8).sum()),
                                     "status_histogram": np.bincount(r["status"], minlength=9).tolist(),
                                     "method": "as infeasible_audit_C2 on kkt_check.DynNlp, up to 7 starts (device iterate, three lateral positions x braked / unbraked)"}
    print(evidence["unsolved_audit_C4"])
    assert (r["status"] == 0).mean() >= 0.97


# ---------------------------------------------------------------------------------------------------------------- (c)
def _slsqp_one(args):
    sys.path.insert(0, ROOT)
    from oracle import kkt_check, scipy_crosscheck as sc
    conf, x0, xs, ob, z_dev, f_dev = args
    if conf == "C4":
        nlp = kkt_check.DynNlp(40, 0.1, x0, xs, ob); rhs0 = lambda x: nlp.rhs(x[None, :], np.zeros((1, 2)))[0]      # noqa: E731
    else:
        nlp = kkt_check.KinNlp(30, 0.1, x0, xs, ob, integrator="rk4" if conf == "C2rk4" else "euler")
        rhs0 = lambda x: np.array([x[3] * np.cos(x[2]), x[3] * np.sin(x[2]), 0.0, 0.0])   # noqa: E731
    z, f, s = sc.solve_slsqp(nlp, sc.cold_start(nlp, 0.1, rhs0), maxiter=600)
    g = nlp.g(z)
    viol = float(np.maximum(0, np.maximum(nlp.lbg - g, g - nlp.ubg)).max())
    return float(np.abs(z - z_dev).max()), float(f / f_dev - 1.0), int(s.status), viol


@pytest.mark.parametrize("conf,count", [("C2", 48), ("C3", 48), ("C4", 24), ("C2rk4", 24)])
def test_same_basin_fraction_against_slsqp(gpu_solver_factory, evidence, conf, count):
    cfg, x0, xs, obs = _sample(conf, 4 * count, seed=123)
    r = gpu_solver_factory(cfg).solve_batch(x0, xs, obs)
    idx = np.nonzero(r["status"] == 0)[0][:count]
    assert len(idx) == count
    with _pool() as p:
        res = p.map(_slsqp_one, [(conf, x0[i], xs[i], obs[i], r["z"][i], r["obj"][i]) for i in idx], chunksize=1)
    dz = np.array([a[0] for a in res]); df = np.array([a[1] for a in res]); ok = np.array([a[3] <= 1e-6 for a in res])
    # same basin: SLSQP ends feasible at the same objective (1e-6 relative) — along flat directions of the dyn problem its own
    # accuracy is ~1e-3 in z at equal f; "within 1e-4" is north_star's trajectory bound
    same = ok & (np.abs(df) <= 1e-6)
    evidence["slsqp_" + conf] = {"instances": count, "slsqp_feasible": int(ok.sum()), "within_1e-4": int((ok & (dz <= 1e-4)).sum()),
                                 "same_basin": int(same.sum()), "same_basin_fraction": float(same.sum() / max(1, ok.sum())),
                                 "other_minimum_slsqp_lower": int((ok & (df < -1e-6)).sum()), "other_minimum_device_lower": int((ok & (df > 1e-6)).sum()),
                                 "median_linf": float(np.median(dz[same])) if same.any() else None}
    print(conf, evidence["slsqp_" + conf])
    # (SLSQP itself ends infeasible from the roll-out cold start on a good part of the instances only the second start solves)
    assert same.sum() >= 0.75 * ok.sum() and ok.sum() >= 0.6 * count
