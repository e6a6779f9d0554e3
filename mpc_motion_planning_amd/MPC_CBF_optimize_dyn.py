"""Dynamic-bicycle MPC (6 states, nonlinear tyre) — surface of
CasaDi_MPC_Optimize_Multishoot/MPC_CBF_optimize_dyn.py.

Solved on the device by mpcb_kernel_dyn (closed-form tyre-model derivatives, 10x10 stage block).  The obstacle row is
posed as h >= 1 instead of the reference's sqrt(h) >= 1 (same feasible set and KKT points, no NaN inside the ellipse).
Two defects of the reference file are not reproduced: it reads vehicle_params['Veh_w'] while
the YAML key is 'Veh_W' (ref :45), and its lbg/ubg lists are interleaved one stage off its g rows (ref :112-129
vs :215-231); `initialize_constraints` below returns bounds aligned with g.
"""
import numpy as np

from . import _abi
from ._mpc_base import MpcBase, NlpSolver


class MPC_optimize(MpcBase):
    MODEL = _abi.MODEL_DYN

    def initialize_constraints(self):
        lbx, ubx = self._box_lists()
        N = self.N_p
        lbg, ubg = [0.0] * 6, [0.0] * 6
        for i in range(N):
            lbg += [0.0] * 6
            ubg += [0.0] * 6
            if i > 0:
                lbg += [self.df_dot_min * self.T_S, self.jerk_min * self.T_S]
                ubg += [self.df_dot_max * self.T_S, self.jerk_max * self.T_S]
        lbg += [1.0] * (N + 1)
        ubg += [np.inf] * (N + 1)
        return lbg, ubg, lbx, ubx

    def optimize_problem(self, ego_state, ref_state, obstacle):
        o = np.asarray(obstacle, dtype=np.float64).reshape(-1)
        row = np.array([[o[0], o[1], 0.0, 0.0, 0.0, 0.0]])
        cfg = self._make_cfg(1)
        return NlpSolver(self, cfg, row.reshape(1, 1, 6), _abi.OBSIN_STATIC)
