"""Diagnostic: status / iteration-count agreement GPU vs oracle for every kernel instantiation on the batches of
test_every_kernel_instantiation_full_outputs (B = 48) and on 256-instance batches (restoration pass included)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from mpc_motion_planning_amd import scenes, _abi
from mpc_motion_planning_amd.solver import BatchSolver, default_config
from oracle import oracle
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for model, n_obs, gamma in [(0, 0, 1), (0, 1, 1), (0, 2, 1), (0, 3, 1), (0, 5, 1), (0, 8, 1), (0, 1, 0.8), (0, 3, 0.5), (0, 5, 0.3), (1, 0, 1), (1, 1, 1), (1, 2, 1), (1, 5, 1), (1, 8, 1)]:
    rng = np.random.default_rng(100 + n_obs)
    if model == 0:
        cfg = default_config(N=30, n_obs=n_obs)
        x0, xs, ob1 = scenes.sample_c2(B, seed=40 + n_obs)
        obs = np.tile(np.array([[900.0, 3.5, 0, 0, 4.8, 1.8]]), (B, max(n_obs, 1), 1))
        if n_obs:
            obs[:, 0] = ob1[:, 0]
            obs[:, 1:, 0] = rng.uniform(150, 400, (B, n_obs - 1)); obs[:, 1:, 1] = rng.uniform(-0.5, 4.0, (B, n_obs - 1))
        obs = obs[:, :n_obs]
        if gamma < 1: cfg.obs_mode = _abi.OBS_DCBF; cfg.gamma = gamma
    else:
        cfg = default_config(model=_abi.MODEL_DYN, N=20, n_obs=n_obs)
        x0, xs, obs = scenes.sample_c4(B, seed=60 + n_obs, n_obs=max(n_obs, 1)); obs = obs[:, :n_obs]
    bs = BatchSolver(cfg)
    g = bs.solve_batch(x0, xs, obs if n_obs else None); r = oracle.solve(cfg, x0, xs, obs if n_obs else None)
    bs.close()
    both = (g["status"] == 0) & (r["status"] == 0)
    resto = (r["status"] >= 5).sum()
    print("model %d n_obs %d gamma %.1f: status equal %.4f  iters equal %.4f  iters equal on both-solved %.4f  solved %d/%d  oracle ended in restoration statuses %d  dz %.1e" % (
        model, n_obs, gamma, (g["status"] == r["status"]).mean(), (g["iters"] == r["iters"]).mean(), (g["iters"][both] == r["iters"][both]).mean(), both.sum(), B, resto,
        np.abs(g["z"][both] - r["z"][both]).max()))
