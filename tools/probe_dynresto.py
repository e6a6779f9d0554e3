import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_motion_planning_amd import scenes, _abi
from mpc_motion_planning_amd.solver import BatchSolver, default_config
from oracle import oracle
for (N, n_obs, B) in ((40, 1, 512), (40, 3, 512), (63, 3, 256), (40, 5, 256), (63, 4, 128), (20, 8, 128)):
    cfg = default_config(model=_abi.MODEL_DYN, N=N, n_obs=n_obs)
    x0, xs, obs = scenes.sample_c4(B, seed=77 + n_obs, n_obs=n_obs)
    g = BatchSolver(cfg).solve_batch(x0, xs, obs); r = oracle.solve(cfg, x0, xs, obs, want_multipliers=False)
    went = (g["status"] != 0) | (r["status"] != 0) | (g["iters"] != r["iters"])
    print("N", N, "n_obs", n_obs, "gpu", np.bincount(g["status"], minlength=7), "oracle", np.bincount(r["status"], minlength=7),
          "status eq %.3f iters eq %.3f" % ((g["status"] == r["status"]).mean(), (g["iters"] == r["iters"]).mean()))
    d = np.nonzero(g["iters"] != r["iters"])[0][:12]
    print("   differing:", d, "gpu it", g["iters"][d], "ora it", r["iters"][d], "gpu st", g["status"][d], "ora st", r["status"][d])
