"""Diagnostic: the 512-instance C2 batches of test_random_c2_batches_against_oracle on the GPU, against the oracle and (for the
instances that differ) the emulated kernel."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from mpc_motion_planning_amd import scenes
from mpc_motion_planning_amd.solver import BatchSolver, default_config
from oracle import oracle
cfg = default_config(N=30, n_obs=1)
bs = BatchSolver(cfg)
for seed in (0, 1, 2):
    x0, xs, obs = scenes.sample_c2(512, seed=seed)
    g = bs.solve_batch(x0, xs, obs); r = oracle.solve(cfg, x0, xs, obs)
    d = np.nonzero(g["status"] != r["status"])[0]
    print("seed", seed, "status equal %.4f" % (g["status"] == r["status"]).mean(), "iters equal %.4f" % (g["iters"] == r["iters"]).mean(), "differ:", [(int(i), int(g["status"][i]), int(g["iters"][i]), int(r["status"][i]), int(r["iters"][i])) for i in d])
    if len(d):
        try:
            from emu import emu
            e = emu.solve(cfg, x0[d], xs[d], obs[d])
            print("   emulated kernel on those:", list(zip(e["status"].tolist(), e["iters"].tolist())))
        except Exception as ex:
            print("   emulator unavailable", ex)
