"""Diagnostic: per-phase cycle counts of one interior-point iteration (s_memtime stamps inside the kernel).
Builds a second library with -DMPCB_STAMPS (never shipped) and runs the shipped C2 scene through mpcb_solve_trace:
    python tools/phase_stamps.py [kin|dyn]
Columns: condense+KKT, Riccati sweep, forward+costates+ratios, line search (s_memtime ticks)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "mpc_motion_planning_amd", "lib", "libmpcbatch_stamps.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-fno-gpu-rdc", "-mllvm", "-amdgpu-opt-vgpr-liverange=false",
                       "-DMPCB_STAMPS"] + [a for a in sys.argv[1:] if a.startswith("-D")] + ["-o", out, os.path.join(ROOT, "mpc_motion_planning_amd", "csrc", "mpcb_api.hip")],
                      stderr=subprocess.DEVNULL)
import mpc_motion_planning_amd._lib as _lib
_lib.LIB_PATH = out
from mpc_motion_planning_amd import scenes
from mpc_motion_planning_amd.solver import BatchSolver, default_config

from mpc_motion_planning_amd import _abi
if "dyn" in sys.argv[1:]:
    cfg = default_config(model=_abi.MODEL_DYN, N=40, T=0.1, n_obs=3)
    x0, xs, obs = scenes.sample_c4(1, seed=3000, n_obs=3)
    x0, xs = x0[0], xs[0]
else:
    cfg = default_config(N=30, T=0.1, n_obs=1)
    x0, xs, obs = scenes.SHIPPED_X0, scenes.SHIPPED_XS, scenes.SHIPPED_OBS.reshape(1, 1, 6)
bs = BatchSolver(cfg)
r = bs.solve_trace(x0, xs, obs)
tr = r["trace"][: int(r["iters"])]
c = tr[:, 4:8]
print("iters", int(r["iters"]), "status", int(r["status"]))
print("mean cycles/iter: condense+KKT %.0f  Riccati %.0f  forward+costates+ratios %.0f  line search %.0f  total %.0f"
      % (*c.mean(0), c.sum(1).mean()))
os.remove(out)
