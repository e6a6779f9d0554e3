"""Diagnostic: C5 closed loop on the device, step count and batch from the command line."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_motion_planning_amd import scenes, _abi
from mpc_motion_planning_amd.solver import BatchSolver, default_config
B = int(sys.argv[1]); steps = int(sys.argv[2]); resto = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cfg = default_config(N=30, n_obs=3); cfg.restoration = resto
x0, xs, obs, _ = scenes.sample_c3(B, N=30, dt=0.1, seed=4000)
bs = BatchSolver(cfg)
for s in ([1, 2, 5, 10, 20, 40, 80] if steps == 0 else [steps]):
    t = time.time()
    dev = bs.closed_loop(x0, xs, obs, steps=s, obs_motion=_abi.OBSMOVE_PREDICTED)
    print("steps", s, "%.2fs" % (time.time() - t), np.bincount(dev["status"].ravel(), minlength=8), "iters max", dev["iters"].max(), flush=True)
