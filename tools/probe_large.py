"""Diagnostic: large-batch agreement GPU vs oracle (C2 65 536 instances, C3 32 768, C4 8 192)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from mpc_motion_planning_amd import scenes, _abi
from mpc_motion_planning_amd.solver import BatchSolver, default_config
from oracle import oracle
for name, cfg, data in (("C2 65536", default_config(N=30, n_obs=1), scenes.sample_c2(65536, seed=77)),
                        ("C3 32768", default_config(N=30, n_obs=3), (lambda t: (t[0], t[1], t[3]))(scenes.sample_c3(32768, N=30, dt=0.1, seed=78))),
                        ("C4 8192", default_config(model=_abi.MODEL_DYN, N=40, n_obs=3), scenes.sample_c4(8192, seed=79, n_obs=3))):
    x0, xs, obs = data
    bs = BatchSolver(cfg); g = bs.solve_batch(x0, xs, obs); bs.close()
    r = oracle.solve(cfg, x0, xs, obs)
    both = (g["status"] == 0) & (r["status"] == 0)
    dz = np.abs(g["z"][both] - r["z"][both]).max(axis=1)
    print("%s: status equal %.5f (%d differ)  solved gpu %d oracle %d  iters equal on both-solved %.4f  L-inf: max %.2e, above 1e-5: %d, above 1e-4: %d" % (
        name, (g["status"] == r["status"]).mean(), int((g["status"] != r["status"]).sum()), int((g["status"] == 0).sum()), int((r["status"] == 0).sum()),
        (g["iters"][both] == r["iters"][both]).mean(), dz.max(), int((dz > 1e-5).sum()), int((dz > 1e-4).sum())))
