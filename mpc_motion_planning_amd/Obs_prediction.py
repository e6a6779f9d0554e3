"""Constant-velocity, constant-heading obstacle roll-out.

Same call and return shape as CasaDi_MPC_Optimize_Multishoot/Obs_prediction.py:3-40:
    obs_prediction(obs_list, dt, N_p) -> list of (N_p+1, 6) arrays, rows [x, y, theta, v, l, w]
with row 0 the given state and x, y advanced by v*cos(theta)*dt, v*sin(theta)*dt per step (:27-30), accumulated
step by step so the sums round exactly as the reference's do.  A batched device version of the same rule lives in
libmpcbatch (mpcb_closed_loop, predict = 1); `scenes.predict_obstacles` is the numpy batch form.
"""
import numpy as np


def obs_prediction(obs_list, dt, N_p):
    out = []
    for obs in obs_list:
        s = np.asarray(obs, dtype=np.float64).reshape(-1)[:6]
        x, y, theta, v, l, w = (float(t) for t in s)
        traj = np.empty((N_p + 1, 6))
        for i in range(N_p + 1):
            traj[i] = (x, y, theta, v, l, w)
            x = x + v * np.cos(theta) * dt
            y = y + v * np.sin(theta) * dt
        out.append(traj)
    return out
