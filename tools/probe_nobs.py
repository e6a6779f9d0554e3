"""Probe: kernel time vs number of obstacles (which NOBS template instance runs) for the kin solve, C3-style scenes.
    python tools/probe_nobs.py [B]"""
import sys
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from mpc_motion_planning_amd import scenes, _abi
from mpc_motion_planning_amd.solver import BatchSolver, default_config

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
for model in (0, 1):
    for n_obs in (0, 1, 2, 3, 4, 5, 8):
        if model == 0:
            cfg = default_config(N=30, n_obs=n_obs)
            if n_obs:
                x0, xs, obs, _ = scenes.sample_c3(B, N=30, dt=0.1, seed=1, n_obs=n_obs)
            else:
                x0, xs, obs = scenes.sample_c2(B, seed=1); obs = None
        else:
            cfg = default_config(model=_abi.MODEL_DYN, N=40, n_obs=n_obs)
            x0, xs, obs = scenes.sample_c4(B, seed=1, n_obs=max(n_obs, 1)); obs = obs[:, :n_obs] if n_obs else None
        bs = BatchSolver(cfg)
        bs.solve_batch(x0, xs, obs); bs.timing(reset=True)
        for _ in range(3):
            out = bs.solve_batch(x0, xs, obs)
        t = bs.timing(); ms = t["total_ms"] / t["launches"]
        it = out["iters"].sum()
        print("%s n_obs=%d: %.2f ms/launch, %d solved of %d, %.1f us per instance-iteration x 1024 SIMDs" % (
            "dyn" if model else "kin", n_obs, ms, int((out["status"] == 0).sum()), B, 1e3 * ms / it * 1024))
        bs.close()
