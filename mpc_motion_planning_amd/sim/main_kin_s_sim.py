"""One open-loop solve without obstacles — counterpart of CasaDi_MPC_Optimize_Multishoot/main_kin_s_sim.py (:36-99),
BASELINE config 1 (plumbing).  The reference imports `MPC_optimize_kin`, whose source is missing from the repository
(only stale bytecode is left); the CBF-kin formulation with zero obstacles is the same NLP.

    python -m mpc_motion_planning_amd.sim.main_kin_s_sim
"""
import numpy as np

from mpc_motion_planning_amd import MPC_CBF_optimize_kin, shift_movement


def main():
    mpc = MPC_CBF_optimize_kin.MPC_optimize()
    N_p = mpc.N_p
    x0 = np.array([0, 0, 0, 20], dtype=float).reshape(-1, 1)
    xs = np.array([500, 3.5, 0, 30], dtype=float).reshape(-1, 1)
    lbg, ubg, lbx, ubx = mpc.initialize_constraints(None)
    solver = mpc.optimize_problem(ego_state=x0, ref_state=xs, obstacle=None)
    res = solver(x0=np.zeros((2 * N_p + 4 * (N_p + 1), 1)), p=np.concatenate((x0, xs)), lbg=lbg, lbx=lbx, ubg=ubg, ubx=ubx)
    z = res["x"].full()
    u0 = z[:2 * N_p].reshape(N_p, 2); x_m = z[2 * N_p:].reshape(N_p + 1, 4)
    t, x1, _, _ = shift_movement(mpc.T_S, 0.0, x0, u0, x_m, mpc.f)
    print("status %s after %d iterations, J = %.6e" % (solver.stats()["return_status"], solver.stats()["iter_count"], float(res["f"])))
    print("first control", u0[0], "next state", np.asarray(x1).reshape(-1))
    return z


if __name__ == "__main__":
    main()
