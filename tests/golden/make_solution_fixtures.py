"""Generates tests/golden/solutions.npz with the CPU oracle (oracle/mpc_oracle.cpp).

PARITY UNPINNED: these vectors come from this repo's own restatement of the NLP + IPOPT's published algorithm,
not from CasADi+IPOPT (absent).  Each vector carries an independent KKT certificate
(oracle/kkt_check.py) computed at generation time and re-checked by the tests.

    python tests/golden/make_solution_fixtures.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle, kkt_check                     # noqa: E402
from mpc_motion_planning_amd import scenes, _abi         # noqa: E402

out = {}


def product_cfg(N, n_obs):
    c = oracle.default_config(N=N, n_obs=n_obs)
    c.init_rollout = 1; c.mu_init = 10.0; c.second_start = 3; c.start_steer = 0.03   # the settings mpcb_default_config ships
    return c


def ipopt_like_cfg(N, n_obs):
    return oracle.default_config(N=N, n_obs=n_obs)        # mu_init 0.1, start taken as given (IPOPT defaults)


# S: shipped closed-loop scene, first step (main_cbf_kin_c_sim.py:45-55), cold start
x0, xs, obs = scenes.SHIPPED_X0[None], scenes.SHIPPED_XS[None], scenes.SHIPPED_OBS[None]
rp = oracle.solve(product_cfg(30, 1), x0, xs, obs)
ri = oracle.solve(ipopt_like_cfg(30, 1), x0, xs, obs)
assert rp["status"][0] == 0 and ri["status"][0] == 0
out.update(S_x0=x0, S_xs=xs, S_obs=obs, S_z=rp["z"], S_obj=rp["obj"], S_lam_g=rp["lam_g"], S_lam_x=rp["lam_x"],
           S_z_ipoptlike=ri["z"], S_iters=np.array([rp["iters"][0], ri["iters"][0]]))
cert = kkt_check.certificate(kkt_check.KinNlp(30, 0.1, x0[0], xs[0], obs[0]), rp["z"][0], rp["lam_g"][0], rp["lam_x"][0])
print("S:", rp["obj"], rp["iters"], ri["iters"], "dz(ipopt-like vs product)", np.abs(rp["z"] - ri["z"]).max(), cert)

# C1: plumbing case, N=20, no obstacle (main_kin_s_sim.py:42,46)
x0 = np.array([[0.0, 0.0, 0.0, 20.0]]); xs = np.array([[500.0, 3.5, 0.0, 30.0]])
r = oracle.solve(product_cfg(20, 0), x0, xs)
assert r["status"][0] == 0
out.update(C1_x0=x0, C1_xs=xs, C1_z=r["z"], C1_obj=r["obj"])
print("C1:", r["obj"], r["iters"])

# C2: 16 seeded random scenes, 1 static obstacle
x0, xs, obs = scenes.sample_c2(16, seed=7)
r = oracle.solve(product_cfg(30, 1), x0, xs, obs)
out.update(C2_x0=x0, C2_xs=xs, C2_obs=obs, C2_z=r["z"], C2_obj=r["obj"], C2_status=r["status"], C2_iters=r["iters"])
print("C2:", r["status"], r["iters"])

# C3: 8 seeded scenes with 3 predicted moving obstacles
x0, xs, obs0, traj = scenes.sample_c3(8, N=30, dt=0.1, seed=11)
r = oracle.solve(product_cfg(30, 3), x0, xs, traj)
out.update(C3_x0=x0, C3_xs=xs, C3_obs0=obs0, C3_traj=traj, C3_z=r["z"], C3_obj=r["obj"], C3_status=r["status"], C3_iters=r["iters"])
print("C3:", r["status"], r["iters"])

# W: warm start = shifted solution of S after one plant step
z = rp["z"][0]; N = 30
U = z[:2 * N].reshape(N, 2); X = z[2 * N:].reshape(N + 1, 4)
f = np.array([X[0, 3] * np.cos(X[0, 2]), X[0, 3] * np.sin(X[0, 2]), X[0, 3] * np.tan(U[0, 0]) / 2.6, U[0, 1]])
x1 = (X[0] + 0.1 * f)[None]
z0 = np.concatenate([np.vstack([U[1:], U[-1:]]).reshape(-1), np.vstack([X[1:], X[-1:]]).reshape(-1)])[None]
rw = oracle.solve(product_cfg(30, 1), x1, scenes.SHIPPED_XS[None], scenes.SHIPPED_OBS[None], z0=z0)
assert rw["status"][0] == 0
out.update(W_x0=x1, W_z0=z0, W_z=rw["z"], W_obj=rw["obj"], W_iters=rw["iters"])
print("W:", rw["obj"], rw["iters"])

# D: dynamic bicycle, shipped closed-loop scene, first step (main_cbf_dyn_c_sim.py:44-51) at N = 40 (BASELINE C4 horizon)
cd = oracle.default_config(model=_abi.MODEL_DYN, N=40, n_obs=1); cd.init_rollout = 1; cd.mu_init = 10.0; cd.second_start = 3; cd.start_steer = 0.03
rd = oracle.solve(cd, scenes.DYN_X0[None], scenes.DYN_XS[None], scenes.DYN_OBS[None])
assert rd["status"][0] == 0
out.update(D_z=rd["z"], D_obj=rd["obj"], D_lam_g=rd["lam_g"], D_lam_x=rd["lam_x"], D_iters=rd["iters"])
print("D:", rd["obj"], rd["iters"])
# C4: 8 seeded dyn scenes with 3 static obstacles
x0, xs, obs = scenes.sample_c4(8, seed=13, n_obs=3)
c4 = oracle.default_config(model=_abi.MODEL_DYN, N=40, n_obs=3); c4.init_rollout = 1; c4.mu_init = 10.0; c4.second_start = 3; c4.start_steer = 0.03
r4 = oracle.solve(c4, x0, xs, obs)
out.update(C4_x0=x0, C4_xs=xs, C4_obs=obs, C4_z=r4["z"], C4_obj=r4["obj"], C4_status=r4["status"], C4_iters=r4["iters"])
print("C4:", r4["status"], r4["iters"])

np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "solutions.npz"), **out)
print("wrote solutions.npz")
