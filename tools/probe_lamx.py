"""Diagnostic: dyn<3> multipliers vs oracle on the instantiation test's batch (which instance / entry differs, iteration counts)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_motion_planning_amd import _abi, scenes
from mpc_motion_planning_amd.solver import BatchSolver, default_config
from oracle import oracle
n_obs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg = default_config(model=_abi.MODEL_DYN, N=20, n_obs=n_obs)
x0, xs, obs = scenes.sample_c4(48, seed=60 + n_obs, n_obs=max(n_obs, 1)); obs = obs[:, :n_obs]
g = BatchSolver(cfg).solve_batch(x0, xs, obs, multipliers=True)
r = oracle.solve(cfg, x0, xs, obs)
both = (g["status"] == 0) & (r["status"] == 0)
print("status equal", (g["status"] == r["status"]).mean(), "iters equal", (g["iters"] == r["iters"]).mean())
sc = np.maximum(1.0, np.abs(r["lam_x"]).max(axis=1, keepdims=True))
d = np.abs(g["lam_x"] - r["lam_x"]) / sc
d[~both] = 0
i, j = np.unravel_index(d.argmax(), d.shape)
print("worst", d.max(), "instance", i, "entry", j, "gpu", g["lam_x"][i, j], "oracle", r["lam_x"][i, j], "scale", sc[i, 0], "iters", g["iters"][i], r["iters"][i],
      "dz", np.abs(g["z"][i] - r["z"][i]).max(), "kkt", g.get("kkt", [None])[i] if "kkt" in g else None)
print("per-instance worst rel diff (sorted):", np.sort(d.max(axis=1))[-6:])
