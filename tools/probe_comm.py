"""Diagnostic: the in-library multi-GPU paths on whatever devices are visible (a one-device group exercises the same calls)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_motion_planning_amd import scenes, _abi, solver
from mpc_motion_planning_amd.solver import BatchSolver, default_config
cfg = default_config(N=30, n_obs=1)
x0, xs, obs = scenes.sample_c2(37, seed=33)
ref = BatchSolver(cfg).solve_batch(x0, xs, obs)
nd = solver.device_count()
g = BatchSolver(cfg); g.set_devices(list(range(nd)))
print("group of", nd, "devices:", g.comm_info())
r = g.solve_batch(x0, xs, obs, multipliers=True)
print("group solve == single solve:", np.array_equal(r["z"], ref["z"]), np.array_equal(r["status"], ref["status"]), np.array_equal(r["iters"], ref["iters"]))
for i in range(nd):
    print("gathered copy on device", i, "equal:", np.array_equal(g.gathered_z(i, 37), ref["z"]))
a = BatchSolver(cfg); a.comm_init(solver.comm_unique_id(), 0, 1)
print("rank mode:", a.comm_info(), a.allreduce([1.5, 2.0], "max"), a.allreduce([1.5, 2.0], "sum"))
s = a.device_array((5,)).upload(np.arange(5.0)); d = a.device_array((5,))
a.allgather(s, d, 5); a.sync(); print("allgather", d.download())
