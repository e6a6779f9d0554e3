"""Probe: kernel time vs horizon N / batch for the kin solve (C2-style scenes).  Used for occupancy experiments:
    python tools/probe_batch.py N B [reps]
Prints ms per launch (HIP events inside the library) and solved count."""
import sys
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from mpc_motion_planning_amd import scenes
from mpc_motion_planning_amd.solver import BatchSolver, default_config

N = int(sys.argv[1]); B = int(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
cfg = default_config(N=N, T=0.1, n_obs=1)
bs = BatchSolver(cfg)
x0, xs, obs = scenes.sample_c2(B, seed=0)
out = bs.solve_batch(x0, xs, obs)
import time
bs.timing(reset=True)
t0 = time.perf_counter()
for _ in range(reps):
    out = bs.solve_batch(x0, xs, obs)
wall_ms = 1e3 * (time.perf_counter() - t0) / reps
t = bs.timing(); n, tot = t['launches'], t['total_ms']
solved = int((out["status"] == 0).sum())
print(f"N={N} B={B} host-pointer call {wall_ms:.3f} ms (PCIe both ways + sync) ms/launch={tot / n:.3f} solved={solved} iters_mean={out['iters'].mean():.2f} solves/s={solved / (tot / n) * 1e3:.0f}")
