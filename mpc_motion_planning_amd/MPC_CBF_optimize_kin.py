"""Kinematic-bicycle MPC with static obstacles — drop-in for
CasaDi_MPC_Optimize_Multishoot/MPC_CBF_optimize_kin.py (class MPC_optimize), solved on MI355X.

    mpc = MPC_optimize()                                   # reads mpc_parameters.yaml          (ref :10-82)
    lbg, ubg, lbx, ubx = mpc.initialize_constraints(obs)   # obs: (n_obs, 6) rows [x,y,th,v,l,w] (ref :84-134)
    solver = mpc.optimize_problem(ego_state=x0, ref_state=ref, obstacle=obs)                     (ref :136-255)
    res = solver(x0=z_init, p=[x0; xs], lbg=lbg, lbx=lbx, ubg=ubg, ubx=ubx);  z = res['x'].full()
"""
import numpy as np

from . import _abi
from ._mpc_base import MpcBase, NlpSolver


class MPC_optimize(MpcBase):
    MODEL = _abi.MODEL_KIN

    def initialize_constraints(self, obstacle):
        n_obs = 0 if obstacle is None else np.asarray(obstacle).reshape(-1, 6).shape[0]
        lbx, ubx = self._box_lists()
        N = self.N_p
        lbg = [0.0] * (4 * (N + 1))                       # X_0 - P, dynamics rows
        ubg = [0.0] * (4 * (N + 1))
        lbg += [self.df_dot_min * self.T_S] * (N - 1)     # steering-rate rows, i = 1..N-1
        ubg += [self.df_dot_max * self.T_S] * (N - 1)
        lbg += [0.0] * (N * n_obs)                        # obstacle rows h >= 0, nodes 0..N-1
        ubg += [np.inf] * (N * n_obs)
        return lbg, ubg, lbx, ubx

    def optimize_problem(self, ego_state, ref_state, obstacle):
        # ego_state and ref_state do not enter the reference's NLP (x0 comes in through p; aa = 0, ref :194-197)
        obs = None if obstacle is None else np.asarray(obstacle, dtype=np.float64).reshape(-1, 6)
        n_obs = 0 if obs is None else obs.shape[0]
        cfg = self._make_cfg(n_obs)
        return NlpSolver(self, cfg, None if n_obs == 0 else obs.reshape(1, n_obs, 6), _abi.OBSIN_STATIC)
