"""Closed-loop simulation, one static obstacle — counterpart of
CasaDi_MPC_Optimize_Multishoot/main_cbf_kin_c_sim.py (same scene constants :45,49,55,68, same loop :87-123).

    python -m mpc_motion_planning_amd.sim.main_cbf_kin_c_sim [--device-loop] [--out run.npz]

Default: the reference's flow step by step through the drop-in surface (optimize_problem -> solver -> shift_movement).
--device-loop: the same 80 steps inside one mpcb_closed_loop call (no host round trips).  Figures are out of scope;
the histories go to an .npz file.
"""
import argparse
import time

import numpy as np

from mpc_motion_planning_amd import MPC_CBF_optimize_kin, RefPathGenerator, shift_movement
from mpc_motion_planning_amd.helpers import load_config, find_params_file


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--device-loop", action="store_true")
    ap.add_argument("--out", default=None)
    ap.add_argument("--sim-time", type=float, default=8.0)
    args = ap.parse_args(argv)

    cfg = load_config(find_params_file())
    T_horizon, T_S = cfg["mpc_params"]["horizon"], cfg["mpc_params"]["T_S"]
    mpc = MPC_CBF_optimize_kin.MPC_optimize()
    N_p, n_states, n_controls = mpc.N_p, mpc.num_states, mpc.num_controls
    x0 = np.array([0, 3, 0, 15], dtype=float).reshape(-1, 1)
    xs = np.array([400, 3.5, 0, 30], dtype=float).reshape(-1, 1)
    obs = np.array([[50, 3.5, 0, 8, 4.8, 1.8]], dtype=float)
    steps = int(round(args.sim_time / T_S))

    if args.device_loop:
        bs = mpc._batch_solver(mpc._make_cfg(1))
        t0 = time.time()
        r = bs.closed_loop(x0.T, xs.T, obs[None], steps=steps)            # obstacles stay put, as in the reference's loop
        print("device loop: %d steps in %.1f ms, statuses %s" % (steps, 1e3 * (time.time() - t0), np.bincount(r["status"][0], minlength=5)))
        xh, uh = r["x_hist"][0], r["u_hist"][0]
    else:
        ref = RefPathGenerator.RefPathGenerator()
        ref.define_ref_path(x0, xs, T_S)
        lbg, ubg, lbx, ubx = mpc.initialize_constraints(obs)
        u0 = np.zeros((N_p, n_controls)); next_states = np.zeros((N_p + 1, n_states))
        t_now, last_idx, xh, uh, ms = 0.0, 0, [x0[:, 0].copy()], [], []
        for _ in range(steps):
            tic = time.time()
            c_p = np.concatenate((x0, xs))
            init = np.concatenate((u0.reshape(-1, 1), next_states.reshape(-1, 1)))
            ref_traj, last_idx = ref.find_ref_traj(x0, xs, T_horizon, T_S, last_idx)
            solver = mpc.optimize_problem(ego_state=x0, ref_state=ref_traj, obstacle=obs)
            res = solver(x0=init, p=c_p, lbg=lbg, lbx=lbx, ubg=ubg, ubx=ubx)
            z = res["x"].full()
            u0 = z[:N_p * n_controls].reshape(N_p, n_controls)
            x_m = z[N_p * n_controls:].reshape(N_p + 1, n_states)
            uh.append(u0[0].copy())
            t_now, x0, u0, next_states = shift_movement(T_S, t_now, x0, u0, x_m, mpc.f)
            x0 = np.asarray(x0).reshape(-1, 1)
            xh.append(x0[:, 0].copy()); ms.append(1e3 * (time.time() - tic))
        xh, uh = np.array(xh), np.array(uh)
        print("host loop: %d steps, %.2f ms/step (incl. PCIe + launch), last status %s" % (steps, np.mean(ms), solver.stats()["return_status"]))
    h = ((xh[:, 0] - 50) / 5.8) ** 2 + ((xh[:, 1] - 3.5) / 2.3) ** 2 - 1
    print("final state %s, min obstacle margin h = %.3f" % (np.round(xh[-1], 3), h.min()))
    if args.out:
        np.savez(args.out, x_hist=xh, u_hist=uh)
    return xh, uh


if __name__ == "__main__":
    main()
