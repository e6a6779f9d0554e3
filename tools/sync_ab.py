"""Diagnostic for the round-2 GPU-only wrong-result incidents: one build of the library (MPCB_LIB) against the
oracle on the two recorded reproducers — fuzz case (11, 4) = kin<8, GEN> with restoration, and random C2 batches through
kin<1> / kin_resto<1>.  Prints one JSON line.   MPCB_LIB=<variant.so> python tools/sync_ab.py <label>"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import tools.fuzz_gpu_vs_oracle as fz
from mpc_motion_planning_amd import scenes
from mpc_motion_planning_amd.solver import BatchSolver, default_config
from oracle import oracle

label = sys.argv[1] if len(sys.argv) > 1 else "default"
out = {"label": label, "lib": os.environ.get("MPCB_LIB", "default")}
# reproducer 1: fuzz case 11 / 4
cap = {}
class FakeBS:
    def __init__(self, cfg): self.cfg = cfg
    def solve_batch(self, x0, xs, ob, multipliers=True):
        cap["last"] = (self.cfg, x0, xs, ob); return oracle.solve(self.cfg, x0, xs, ob)
    def close(self): pass
real = fz.BatchSolver; fz.BatchSolver = FakeBS
rng = np.random.default_rng(11)
for c in range(5):
    ok, line = fz.one_case(rng, c)
fz.BatchSolver = real
cfg, x0, xs, ob = cap["last"]
out["fuzz_case"] = line.split("->")[0].strip()
bs = BatchSolver(cfg)
g = bs.solve_batch(x0, xs, ob); r = oracle.solve(cfg, x0, xs, ob)
bs.close()
out["fuzz_status_equal"] = float((g["status"] == r["status"]).mean())
out["fuzz_iters_equal"] = float((g["iters"] == r["iters"]).mean())
# reproducer 2: C2 batches (kin<1> first pass, kin_resto<1> restoration pass)
eq_s, eq_i, n = 0, 0, 0
for seed in (0, 1, 2):
    cfg2 = default_config(N=30, n_obs=1)
    x0, xs, obs = scenes.sample_c2(512, seed=seed)
    bs = BatchSolver(cfg2); g = bs.solve_batch(x0, xs, obs); bs.close()
    r = oracle.solve(cfg2, x0, xs, obs)
    eq_s += int((g["status"] == r["status"]).sum()); eq_i += int((g["iters"] == r["iters"]).sum()); n += 512
out["c2_status_equal"] = eq_s / n; out["c2_iters_equal"] = eq_i / n
print(json.dumps(out))
