"""Closed-loop simulation without obstacles — counterpart of CasaDi_MPC_Optimize_Multishoot/main_kin_c_sim.py (scene
constants :40,44,53; loop :64-84 with its `|x0 - xs| > 1e-2` stop test).  The reference imports `MPC_optimize_kin`, whose
source is missing from the repository; the CBF-kin formulation with zero obstacles is the same NLP.

    python -m mpc_motion_planning_amd.sim.main_kin_c_sim [--sim-time 10] [--out run.npz]
"""
import argparse
import time

import numpy as np

from mpc_motion_planning_amd import MPC_CBF_optimize_kin, shift_movement


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--sim-time", type=float, default=10.0)
    args = ap.parse_args(argv)

    mpc = MPC_CBF_optimize_kin.MPC_optimize()
    N_p, n_states, n_controls, T_S = mpc.N_p, mpc.num_states, mpc.num_controls, mpc.T_S
    x0 = np.array([0, 0, 0, 20], dtype=float).reshape(-1, 1)
    xs = np.array([500, 3.5, 0, 30], dtype=float).reshape(-1, 1)
    lbg, ubg, lbx, ubx = mpc.initialize_constraints(None)
    u0 = np.zeros((N_p, n_controls)); next_states = np.zeros((N_p + 1, n_states))
    t_now, it, xh, uh, xc, ms = 0.0, 0, [x0[:, 0].copy()], [], [], []
    while np.linalg.norm(x0 - xs) > 1e-2 and it - args.sim_time / T_S < 0.0:
        tic = time.time()
        c_p = np.concatenate((x0, xs))
        init = np.concatenate((u0.reshape(-1, 1), next_states.reshape(-1, 1)))
        solver = mpc.optimize_problem(ego_state=x0, ref_state=xs, obstacle=None)
        res = solver(x0=init, p=c_p, lbg=lbg, lbx=lbx, ubg=ubg, ubx=ubx)
        z = res["x"].full()
        u0 = z[:N_p * n_controls].reshape(N_p, n_controls)
        x_m = z[N_p * n_controls:].reshape(N_p + 1, n_states)
        xc.append(x_m.T.copy()); uh.append(u0[0].copy())
        t_now, x0, u0, next_states = shift_movement(T_S, t_now, x0, u0, x_m, mpc.f)
        x0 = np.asarray(x0).reshape(-1, 1)
        xh.append(x0[:, 0].copy()); ms.append(1e3 * (time.time() - tic)); it += 1
    xh, uh = np.array(xh), np.array(uh)
    print("host loop: %d steps, %.2f ms/step (incl. PCIe + launch), last status %s" % (it, np.mean(ms), solver.stats()["return_status"]))
    print("final state %s" % np.round(xh[-1], 3))
    if args.out:
        np.savez(args.out, x_hist=xh, u_hist=uh, x_pred=np.array(xc))
    return xh, uh


if __name__ == "__main__":
    main()
