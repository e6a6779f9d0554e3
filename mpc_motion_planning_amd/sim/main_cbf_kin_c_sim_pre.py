"""Closed-loop simulation with a moving obstacle re-predicted every step — counterpart of
CasaDi_MPC_Optimize_Multishoot/main_cbf_kin_c_sim_pre.py (:40-126; obstacle [50,3.5,0,10,4.8,1.8], advanced by one
predicted step per MPC step, :106).

    python -m mpc_motion_planning_amd.sim.main_cbf_kin_c_sim_pre [--out run.npz]
"""
import argparse

import numpy as np

from mpc_motion_planning_amd import MPC_CBF_optimize_kin_pre, shift_movement
from mpc_motion_planning_amd.Obs_prediction import obs_prediction


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--sim-time", type=float, default=8.0)
    args = ap.parse_args(argv)
    mpc = MPC_CBF_optimize_kin_pre.MPC_optimize()
    N_p, T_S = mpc.N_p, mpc.T_S
    x0 = np.array([0, 3, 0, 15], dtype=float).reshape(-1, 1)
    xs = np.array([400, 3.5, 0, 30], dtype=float).reshape(-1, 1)
    obs = [np.array([[50, 3.5, 0, 10, 4.8, 1.8]], dtype=float)]
    lbg, ubg, lbx, ubx = mpc.initialize_constraints(obs)
    u0 = np.zeros((N_p, 2)); next_states = np.zeros((N_p + 1, 4))
    t_now, xh, uh, oh = 0.0, [x0[:, 0].copy()], [], [obs[0][0].copy()]
    for _ in range(int(round(args.sim_time / T_S))):
        traj = obs_prediction(obs, T_S, N_p)
        solver = mpc.optimize_problem(ego_state=x0, ref_state=None, obs_trajectories=traj)
        res = solver(x0=np.concatenate((u0.reshape(-1, 1), next_states.reshape(-1, 1))), p=np.concatenate((x0, xs)),
                     lbg=lbg, lbx=lbx, ubg=ubg, ubx=ubx)
        z = res["x"].full()
        u0 = z[:2 * N_p].reshape(N_p, 2); x_m = z[2 * N_p:].reshape(N_p + 1, 4)
        obs = [traj[0][1].reshape(1, -1)]
        uh.append(u0[0].copy())
        t_now, x0, u0, next_states = shift_movement(T_S, t_now, x0, u0, x_m, mpc.f)
        x0 = np.asarray(x0).reshape(-1, 1)
        xh.append(x0[:, 0].copy()); oh.append(obs[0][0].copy())
    xh, uh, oh = np.array(xh), np.array(uh), np.array(oh)
    h = ((xh[:, 0] - oh[:, 0]) / 5.8) ** 2 + ((xh[:, 1] - oh[:, 1]) / 2.3) ** 2 - 1
    print("final state %s, obstacle at x = %.1f, min margin h = %.3f" % (np.round(xh[-1], 3), oh[-1, 0], h.min()))
    if args.out:
        np.savez(args.out, x_hist=xh, u_hist=uh, obs_hist=oh)
    return xh, uh, oh


if __name__ == "__main__":
    main()
