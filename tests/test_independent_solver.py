"""An off-the-shelf NLP solver (SciPy SLSQP: sequential quadratic programming, no code or algorithm shared with this repository)
on the reference's NLP as oracle/kkt_check.py restates it, from the solvers' own cold start, against the CPU oracle.  The GPU
suite repeats it against the HIP path.  This is NOT the reference's CasADi+IPOPT (parity stays unpinned), but it is an
independent SOLVER, where oracle/kkt_check.certificate is an independent optimality CHECK."""
import numpy as np
import pytest

from mpc_motion_planning_amd import _abi, scenes
from mpc_motion_planning_amd.solver import default_config
from oracle import kkt_check, oracle, scipy_crosscheck as sc


def kin_rhs0(x):
    return np.array([x[3] * np.cos(x[2]), x[3] * np.sin(x[2]), 0.0, 0.0])


def cases():
    """(name, cfg, x0, xs, obs-as-the-solver-takes-it, nlp, rhs for the cold start, tolerance on z).  The tolerance is north_star's
    1e-4; measured: 2e-6 .. 4e-5 (SLSQP stops where its line search finds no further descent at f ~ 1e8)."""
    out = []
    cfg = default_config(N=30, n_obs=1)                                             # C2, the shipped scene
    out.append(("C2 shipped", cfg, scenes.SHIPPED_X0, scenes.SHIPPED_XS, scenes.SHIPPED_OBS.reshape(1, 6),
                kkt_check.KinNlp(30, 0.1, scenes.SHIPPED_X0, scenes.SHIPPED_XS, scenes.SHIPPED_OBS.reshape(1, 6)), kin_rhs0, 1e-4))
    x0, xs, _ = scenes.sample_c2(2, seed=7)                                         # C1-like: N = 20, no obstacle
    out.append(("C1", default_config(N=20, n_obs=0), x0[0], xs[0], None, kkt_check.KinNlp(20, 0.1, x0[0], xs[0], None), kin_rhs0, 1e-4))
    x0, xs, o0, traj = scenes.sample_c3(2, N=30, dt=0.1, seed=4)                    # C3: three predicted obstacles
    out.append(("C3", default_config(N=30, n_obs=3), x0[0], xs[0], traj[0], kkt_check.KinNlp(30, 0.1, x0[0], xs[0], traj[0]), kin_rhs0, 1e-4))
    x0, xs, ob = scenes.sample_c4(2, seed=61, n_obs=1)                              # dynamic bicycle, reference row form sqrt(h) >= 1
    nlp = kkt_check.DynNlp(20, 0.1, x0[0], xs[0], ob[0])
    out.append(("dyn", default_config(model=_abi.MODEL_DYN, N=20, n_obs=1), x0[0], xs[0], ob[0], nlp,
                lambda x, nlp=nlp: nlp.rhs(x[None, :], np.zeros((1, 2)))[0], 1e-4))
    return out


@pytest.mark.parametrize("case", cases(), ids=lambda c: c[0])
def test_oracle_against_scipy_slsqp(case):
    name, cfg, x0, xs, obs, nlp, rhs0, tol = case
    r = oracle.solve(cfg, x0[None], xs[None], None if obs is None else obs[None])
    assert r["status"][0] == 0
    z, f, s = sc.solve_slsqp(nlp, sc.cold_start(nlp, cfg.T, rhs0))
    assert np.abs(nlp.g(z)[nlp.lbg == nlp.ubg] - nlp.lbg[nlp.lbg == nlp.ubg]).max() <= 1e-9      # SLSQP's point is feasible
    assert np.abs(z - r["z"][0]).max() <= tol, "%s: L-inf %.2e" % (name, np.abs(z - r["z"][0]).max())
    # the interior-point solution sits inside bounds relaxed by 1e-8 (IPOPT's bound_relax_factor): its objective is lower by ~1e-8 relative
    assert abs(f / r["obj"][0] - 1) <= 1e-7
