"""The kernel source (mpc_motion_planning_amd/csrc/mpcb_kernel.h) stepped on the CPU by tests/emu (64 host threads,
one per lane) against the oracle.  No GPU.  This checks the wave-parallel formulation — lane/entry mappings, LDS
tables, reductions — before any GPU time is spent; the GPU parity tests proper are in test_gpu_parity.py."""
import os

import numpy as np
import pytest

from oracle import oracle
from tests.emu import emu
from mpc_motion_planning_amd import scenes, _abi

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "solutions.npz"))


def product_cfg(N=30, n_obs=1):
    c = oracle.default_config(N=N, n_obs=n_obs)
    c.init_rollout = 1; c.mu_init = 10.0; c.second_start = 3; c.start_steer = 0.03        # the settings mpcb_default_config ships
    return c


def test_emulated_kernel_equals_oracle_on_shipped_scene():
    cfg = product_cfg()
    e = emu.solve(cfg, G["S_x0"], G["S_xs"], G["S_obs"], trace_instance=0)
    assert e["status"][0] == 0 and e["iters"][0] == G["S_iters"][0]
    assert np.abs(e["z"] - G["S_z"]).max() <= 1e-10
    assert np.abs(e["lam_g"] - G["S_lam_g"]).max() <= 1e-6 * np.abs(G["S_lam_g"]).max()
    assert np.abs(e["lam_x"] - G["S_lam_x"]).max() <= 1e-6 * max(1.0, np.abs(G["S_lam_x"]).max())
    tr = e["trace"][: e["iters"][0] + 1]
    assert tr[0, 0] == 10.0 and tr[-1, 1] <= 1e-8 and np.all(np.diff(tr[:, 0]) <= 0)      # mu monotone, converged


def test_emulated_kernel_variants():
    # start taken as given (IPOPT-like), no obstacle, three predicted obstacles, DCBF rows
    c = oracle.default_config(N=30, n_obs=1)
    e = emu.solve(c, G["S_x0"], G["S_xs"], G["S_obs"]); r = oracle.solve(c, G["S_x0"], G["S_xs"], G["S_obs"])
    assert e["status"][0] == r["status"][0] == 0 and np.abs(e["z"] - r["z"]).max() <= 1e-8
    c = product_cfg(20, 0)
    e = emu.solve(c, G["C1_x0"], G["C1_xs"])
    assert e["status"][0] == 0 and np.abs(e["z"] - G["C1_z"]).max() <= 1e-9
    c = product_cfg(30, 3)
    e = emu.solve(c, G["C3_x0"][:2], G["C3_xs"][:2], G["C3_traj"][:2])
    assert np.array_equal(e["status"], G["C3_status"][:2]) and np.abs(e["z"] - G["C3_z"][:2]).max() <= 1e-8
    c = product_cfg(); c.obs_mode = _abi.OBS_DCBF
    e = emu.solve(c, G["S_x0"], G["S_xs"], G["S_obs"]); r = oracle.solve(c, G["S_x0"], G["S_xs"], G["S_obs"])
    assert e["status"][0] == 0 and np.abs(e["z"] - r["z"]).max() <= 1e-8 and np.abs(e["lam_g"] - r["lam_g"]).max() <= 1e-5 * np.abs(r["lam_g"]).max()


def test_emulated_kernel_general_gamma_cbf_rows():
    """GEN kernels (0 < gamma < 1): closed-form gradient and Hessian of c_i(X_i) = h(F(X_i)) - (1-gamma) h(X_i) against the
    oracle's AD, multipliers converted to the reference's row form."""
    c = product_cfg(30, 1); c.obs_mode = _abi.OBS_DCBF; c.gamma = 0.5
    e = emu.solve(c, G["S_x0"], G["S_xs"], G["S_obs"]); r = oracle.solve(c, G["S_x0"], G["S_xs"], G["S_obs"])
    assert e["status"][0] == r["status"][0] == 0 and e["iters"][0] == r["iters"][0]
    assert np.abs(e["z"] - r["z"]).max() <= 1e-9
    assert np.abs(e["lam_g"] - r["lam_g"]).max() <= 1e-8 * np.abs(r["lam_g"]).max()


def test_emulated_kernel_failure_paths():
    c = product_cfg()
    e = emu.solve(c, [[48.0, 3.5, 0, 10]], G["S_xs"], G["S_obs"])
    assert e["status"][0] == _abi.ST_INFEASIBLE_X0 and e["iters"][0] == 0
    e = emu.solve(c, [[40.0, 3.5, 0, 25]], G["S_xs"], G["S_obs"]); r = oracle.solve(c, [[40.0, 3.5, 0, 25]], G["S_xs"], G["S_obs"])
    assert e["status"][0] == r["status"][0] != 0 and np.all(np.isfinite(e["z"]))


def test_emulated_dyn_kernel_and_closed_form_derivatives():
    """6-state dynamic bicycle: the kernel's hand-written tyre-model Jacobian / Hessian against the oracle's
    forward-mode AD, and one small solve (N = 10, two entries per lane in the 10x10 Riccati block)."""
    cfg = oracle.default_config(model=_abi.MODEL_DYN, N=10, n_obs=1); cfg.init_rollout = 1; cfg.mu_init = 10.0
    rng = np.random.default_rng(0)
    for _ in range(25):
        X = np.array([rng.uniform(0, 100), rng.uniform(-1, 5), rng.uniform(-0.5, 0.5), rng.uniform(3, 30), rng.uniform(-2, 2), rng.uniform(-0.5, 0.5)])
        U = np.array([rng.uniform(-0.5, 0.5), rng.uniform(-3, 3)]); lam = rng.normal(size=6) * 10
        F, A, Bm, H = oracle.model_eval(cfg, X, U, lam, ad=True)
        Fk, j, h = emu.dyn_model(cfg, X, U, lam)
        jr = np.array([A[0, 2], A[0, 3], A[0, 4], A[1, 2], A[1, 3], A[1, 4], A[3, 4], A[3, 5], A[4, 3], A[4, 4], A[4, 5], A[5, 3], A[5, 4], A[5, 5], Bm[4, 0], Bm[5, 0]])
        hr = np.array([H[2, 2], H[2, 3], H[2, 4], H[3, 3], H[3, 4], H[3, 5], H[4, 4], H[4, 5], H[5, 5], H[3, 6], H[4, 6], H[5, 6], H[6, 6]])
        assert np.allclose(F, Fk, rtol=1e-13, atol=1e-12) and np.allclose(j, jr, rtol=1e-12, atol=1e-12) and np.allclose(h, hr, rtol=1e-11, atol=1e-10)
        # entries the kernel treats as structural constants / zeros
        mask = np.ones((6, 6), bool)
        for (a_, b_) in [(0, 0), (1, 1), (2, 2), (3, 3), (0, 2), (0, 3), (0, 4), (1, 2), (1, 3), (1, 4), (2, 5), (3, 4), (3, 5), (4, 3), (4, 4), (4, 5), (5, 3), (5, 4), (5, 5)]:
            mask[a_, b_] = False
        assert np.abs(A[mask]).max() == 0.0 and A[2, 5] == pytest.approx(0.1) and Bm[3, 1] == pytest.approx(0.1) and np.abs(Bm[:3]).max() == 0.0
    x0 = scenes.DYN_X0[None]; xs = scenes.DYN_XS[None]; obs = np.array([[[30.0, -3.0, 0, 0, 0, 0]]])
    e = emu.solve(cfg, x0, xs, obs); r = oracle.solve(cfg, x0, xs, obs)
    assert e["status"][0] == r["status"][0] == 0 and e["iters"][0] == r["iters"][0]
    assert np.abs(e["z"] - r["z"]).max() <= 1e-9 and np.abs(e["lam_g"] - r["lam_g"]).max() <= 1e-6 * np.abs(r["lam_g"]).max()


def test_emulated_kernel_end_game_needs_symmetric_p():
    """Five C3 instances (tests/golden/endgame_c3.npz, captured from a GPU run) that the device kernel failed with
    MPCB_ST_LINESEARCH at mu = mu_floor while the oracle solved them: P(i,j) and P(j,i) were stored from two lanes with different
    cancellation errors, the asymmetry compounded through the Riccati sweep and the inertia correction escalated.  With P
    mirrored across the diagonal at the store the kernel converges like the oracle."""
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "endgame_c3.npz"))
    cfg = product_cfg(30, 3); cfg.second_start = 0; cfg.start_steer = 0.0       # the straight roll-out start these instances were captured with
    e = emu.solve(cfg, d["x0"], d["xs"], d["obs"]); r = oracle.solve(cfg, d["x0"], d["xs"], d["obs"])
    assert np.all(r["status"] == 0) and np.all(e["status"] == 0)
    assert np.abs(e["iters"] - r["iters"]).max() <= 4 and np.abs(e["z"] - r["z"]).max() <= 1e-5


def test_emulated_kernels_with_a_time_grid():
    """Per-stage step lengths (mpcb_set_time_grid; the two-rate grid of kin.py:19-25 made effective) through the kernel source
    and the oracle: same status, same iteration count, same point; the shooting rows hold with T_i and the rate rows with
    rate * T_{i-1}."""
    c = product_cfg()
    tg = np.concatenate([np.full(24, 0.1), np.full(6, 0.5)])
    x0, xs, obs = scenes.sample_c2(3, seed=3)
    r = oracle.solve(c, x0, xs, obs, tgrid=tg); e = emu.solve(c, x0, xs, obs, tgrid=tg)
    # (iteration counts: equal up to one step of the end game on an instance, where the two sides round differently)
    assert np.array_equal(r["status"], e["status"]) and np.abs(r["iters"] - e["iters"]).max() <= 1 and (r["iters"] != e["iters"]).sum() <= 1 and r["status"][0] == 0
    assert np.abs(r["z"] - e["z"]).max() <= 1e-9
    z = e["z"][0]; X = z[60:].reshape(31, 4); U = z[:60].reshape(30, 2)
    f = np.stack([X[:-1, 3] * np.cos(X[:-1, 2]), X[:-1, 3] * np.sin(X[:-1, 2]), X[:-1, 3] * np.tan(U[:, 0]) / 2.6, U[:, 1]], 1)
    assert np.abs(X[1:] - (X[:-1] + tg[:, None] * f)).max() <= 1e-8
    assert (np.abs(np.diff(U[:, 0])) - 1e-8).max() <= (5 * np.pi / 180 * tg[:-1]).max() and ((np.abs(np.diff(U[:, 0])) - 2e-8) / tg[:-1]).max() <= 5 * np.pi / 180
    u = oracle.solve(c, x0, xs, obs); g = oracle.solve(c, x0, xs, obs, tgrid=np.full(30, 0.1))
    assert np.array_equal(u["z"], g["z"])                              # a grid of T_S everywhere is the fixed grid, bit for bit


def test_lds_budget_of_the_benchmark_instances():
    """The kernels run one wave per SIMD (register-bound: four workgroups per CU), so an instance must stay within 160 KB / 4 of LDS
    where it can, and within 160 KB / 3 where it cannot: a table that grows past these lines costs a quarter or a third of the
    throughput without failing any parity test (it happened in round 2 to N = 50 with a first version of the roll-out records)."""
    CU = 160 * 1024
    kin30 = oracle.default_config(N=30, n_obs=1)
    assert emu.lds_bytes(kin30) <= CU // 4 and emu.lds_bytes(kin30, True) <= CU // 4             # C2: both passes four per CU
    kin30_3 = oracle.default_config(N=30, n_obs=3)
    assert emu.lds_bytes(kin30_3) <= CU // 4 and emu.lds_bytes(kin30_3, True) <= CU // 4         # C3 / C5
    kin50 = oracle.default_config(N=50, n_obs=1)
    assert emu.lds_bytes(kin50) <= CU // 3                                                        # the shipped YAML (N_p = 50): three per CU
    dyn40 = oracle.default_config(model=_abi.MODEL_DYN, N=40, n_obs=3)
    assert emu.lds_bytes(dyn40) <= CU // 3                                                        # C4: three per CU (two in round 1)
    kin63_8 = oracle.default_config(N=63, n_obs=8)
    assert emu.lds_bytes(kin63_8, True) <= CU                                                     # the largest instance still fits one CU
    dyn63_8 = oracle.default_config(model=_abi.MODEL_DYN, N=63, n_obs=8)
    assert emu.lds_bytes(dyn63_8, True) <= CU


def test_rk4_shooting_rows_kernel_source_against_oracle_and_certificate():
    """cfg.integrator = MPCB_INT_RK4 (BASELINE's north_star; the reference's NLP is explicit Euler): the kernel's closed-form
    Runge-Kutta step, its Jacobian and its Hessian contraction (kin_rk4_step / kin_rk4_derivs in mpcb_kernel.h) stepped on the CPU
    against the oracle, which differentiates four rhs evaluations by second-order AD: same iteration counts, same trajectories;
    and the independent certificate of oracle/kkt_check.py with its own RK4 rows (complex-step derivatives)."""
    from oracle import kkt_check
    cfg = product_cfg(30, 1); cfg.integrator = _abi.INT_RK4
    x0, xs, obs = scenes.sample_c2(5, seed=3); x0[0] = scenes.SHIPPED_X0
    r = oracle.solve(cfg, x0, xs, obs); e = emu.solve(cfg, x0, xs, obs)
    assert np.array_equal(r["status"], e["status"]) and np.array_equal(r["iters"], e["iters"]) and (r["status"] == 0).all()
    assert np.abs(r["z"] - e["z"]).max() <= 1e-10 and np.abs(r["lam_g"] - e["lam_g"]).max() <= 1e-9 * np.abs(r["lam_g"]).max()
    eu = oracle.solve(product_cfg(30, 1), x0, xs, obs)
    assert 1e-3 < np.abs(eu["z"] - r["z"]).max() < 2.0                      # a different discretisation, the same manoeuvre
    for b in range(3):
        nlp = kkt_check.KinNlp(30, 0.1, x0[b], xs[b], obs[b], integrator="rk4")
        c = kkt_check.certificate(nlp, e["z"][b], e["lam_g"][b], e["lam_x"][b])
        assert c["stationarity"] <= 1e-6 * c["lam_scale"] and c["feas_g"] <= 2e-8 and c["compl"] <= 1e-3 and c["sign"] == 0.0, c
        assert c["f"] == pytest.approx(e["obj"][b], rel=1e-11)
    c3 = product_cfg(30, 3); c3.integrator = _abi.INT_RK4                  # three predicted obstacles through the <3> instantiation
    x0, xs, _, traj = scenes.sample_c3(3, N=30, dt=0.1, seed=5)
    r = oracle.solve(c3, x0, xs, traj); e = emu.solve(c3, x0, xs, traj)
    assert np.array_equal(r["status"], e["status"]) and np.array_equal(r["iters"], e["iters"])
    ok = r["status"] == 0
    assert ok.any() and np.abs(r["z"][ok] - e["z"][ok]).max() <= 1e-10
