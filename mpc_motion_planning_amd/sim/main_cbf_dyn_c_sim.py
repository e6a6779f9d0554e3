"""Closed-loop simulation with the dynamic bicycle model — counterpart of
CasaDi_MPC_Optimize_Multishoot/main_cbf_dyn_c_sim.py (scene constants :44,48,51,59; loop :75-108, including the
zeroed control at step 10, :97-99).

    python -m mpc_motion_planning_amd.sim.main_cbf_dyn_c_sim [--sim-time 10] [--out run.npz]

The reference's flow step by step through the drop-in surface (optimize_problem -> solver -> shift_movement); the bound
lists come back g-aligned.  Figures are out of scope; the histories go to an .npz file.
"""
import argparse
import time

import numpy as np

from mpc_motion_planning_amd import MPC_CBF_optimize_dyn, shift_movement


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--sim-time", type=float, default=10.0)
    args = ap.parse_args(argv)

    mpc = MPC_CBF_optimize_dyn.MPC_optimize()
    N_p, n_states, n_controls, T_S = mpc.N_p, mpc.num_states, mpc.num_controls, mpc.T_S
    x0 = np.array([0, 0, 0, 10, 0, 0], dtype=float).reshape(-1, 1)        # x y phi vx vy r
    xs = np.array([600, 3.5, 0, 15, 0, 0], dtype=float).reshape(-1, 1)
    obs = np.array([100, -3.5], dtype=float)
    steps = int(round(args.sim_time / T_S))

    lbg, ubg, lbx, ubx = mpc.initialize_constraints()
    u0 = np.zeros((N_p, n_controls)); next_states = np.zeros((N_p + 1, n_states))
    t_now, xh, uh, ms, fails = 0.0, [x0[:, 0].copy()], [], [], 0
    for it in range(steps):
        tic = time.time()
        c_p = np.concatenate((x0, xs))
        init = np.concatenate((u0.reshape(-1, 1), next_states.reshape(-1, 1)))
        solver = mpc.optimize_problem(ego_state=x0, ref_state=xs, obstacle=obs)
        res = solver(x0=init, p=c_p, lbg=lbg, lbx=lbx, ubg=ubg, ubx=ubx)
        z = res["x"].full()
        u0 = z[:N_p * n_controls].reshape(N_p, n_controls)
        x_m = z[N_p * n_controls:].reshape(N_p + 1, n_states)
        fails += 0 if solver.stats()["success"] else 1
        if it == 10:                                                      # the reference's disturbance: no input at step 10
            u0[0, :] = 0.0
        uh.append(u0[0].copy())
        t_now, x0, u0, next_states = shift_movement(T_S, t_now, x0, u0, x_m, mpc.f)
        x0 = np.asarray(x0).reshape(-1, 1)
        xh.append(x0[:, 0].copy()); ms.append(1e3 * (time.time() - tic))
    xh, uh = np.array(xh), np.array(uh)
    h = ((xh[:, 0] - obs[0]) / 4.0) ** 2 + ((xh[:, 1] - obs[1]) / 1.0) ** 2 - 1      # dyn.py:238-243, fixed axes 4 x 1
    print("host loop: %d steps, %.2f ms/step (incl. PCIe + launch), %d unsolved steps" % (steps, np.mean(ms), fails))
    print("final state %s, min obstacle margin h = %.3f" % (np.round(xh[-1], 3), h.min()))
    if args.out:
        np.savez(args.out, x_hist=xh, u_hist=uh)
    return xh, uh


if __name__ == "__main__":
    main()
