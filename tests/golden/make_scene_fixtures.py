"""Generates tests/golden/scene_helpers.json by IMPORTING the reference's numpy-only scene helpers
(CasaDi_MPC_Optimize_Multishoot/Obs_prediction.py, RefPathGenerator.py, helpers.py — none of them imports casadi)
in the build container.  /root/reference does not travel to the GPU box; only this data file does.

    python tests/golden/make_scene_fixtures.py
"""
import json
import os
import sys
import warnings

import numpy as np

REF = "/root/reference/CasaDi_MPC_Optimize_Multishoot"
sys.path.insert(0, REF)
warnings.simplefilter("ignore")
import Obs_prediction as ref_obs          # noqa: E402
import RefPathGenerator as ref_path       # noqa: E402
import helpers as ref_helpers             # noqa: E402

out = {}
cases = []
for obs, dt, n in [([[50, 3.5, 0, 10, 4.8, 1.8]], 0.1, 30), ([[50, 3.5, 0.05, 8, 4.8, 1.8], [70, 0.0, -0.02, 12.5, 4.2, 1.7]], 0.1, 20),
                   ([[10.0, 1.0, 0.3, 3.0, 5.0, 2.0]], 0.05, 50)]:
    tr = ref_obs.obs_prediction([np.array([o], dtype=float) for o in obs], dt, n)
    cases.append({"obs": obs, "dt": dt, "N_p": n, "traj": [t.tolist() for t in tr]})
out["obs_prediction"] = cases

paths = []
for x0, xs, H, dt, last in [([0, 3, 0, 15], [400, 3.5, 0, 30], 3, 0.1, 0), ([37.2, 2.1, 0.02, 22.0], [400, 3.5, 0, 30], 3, 0.1, 30),
                            ([120.0, 3.4, 0.0, 29.0], [400, 3.5, 0, 30], 5, 0.1, 118), ([390.0, 3.5, 0.0, 30.0], [400, 3.5, 0, 30], 3, 0.1, 385)]:
    g = ref_path.RefPathGenerator()
    glob = g.define_ref_path(np.array([0, 3, 0, 15]).reshape(-1, 1), np.array(xs).reshape(-1, 1), dt)
    loc, idx = g.find_ref_traj(np.array(x0, dtype=float).reshape(-1, 1), np.array(xs, dtype=float).reshape(-1, 1), H, dt, last)
    paths.append({"x0": x0, "xs": xs, "H": H, "dt": dt, "last_idx": last, "global_shape": list(glob.shape),
                  "global_first": glob[0].tolist(), "global_last": glob[-1].tolist(), "local": np.asarray(loc).tolist(), "min_idx": int(idx)})
out["ref_path"] = paths

cfg = ref_helpers.load_config(os.path.join(REF, "mpc_parameters.yaml"))
out["yaml"] = cfg
for H in (2, 3, 4, 5):
    out.setdefault("N_p", {})[str(H)] = len(np.arange(0, H + 0.1, 0.1, dtype=float)) - 1

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "scene_helpers.json"), "w") as fh:
    json.dump(out, fh, indent=1)
print("wrote scene_helpers.json")
