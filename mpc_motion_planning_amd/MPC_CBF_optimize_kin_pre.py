"""Kinematic-bicycle MPC with predicted (moving) obstacles — drop-in for
CasaDi_MPC_Optimize_Multishoot/MPC_CBF_optimize_kin_pre.py.  Differs from the static variant only in where the
obstacle rows take the obstacle from: node i uses obs_trajectories[j][i] (ref :236-253)."""
import numpy as np

from . import _abi
from ._mpc_base import MpcBase, NlpSolver


class MPC_optimize(MpcBase):
    MODEL = _abi.MODEL_KIN

    def initialize_constraints(self, obs_trajectories):
        n_obs = 0 if obs_trajectories is None else len(obs_trajectories)
        lbx, ubx = self._box_lists()
        N = self.N_p
        lbg = [0.0] * (4 * (N + 1)) + [self.df_dot_min * self.T_S] * (N - 1) + [0.0] * (N * n_obs)
        ubg = [0.0] * (4 * (N + 1)) + [self.df_dot_max * self.T_S] * (N - 1) + [np.inf] * (N * n_obs)
        return lbg, ubg, lbx, ubx

    def optimize_problem(self, ego_state, ref_state, obs_trajectories):
        n_obs = 0 if obs_trajectories is None else len(obs_trajectories)
        cfg = self._make_cfg(n_obs)
        traj = None
        if n_obs:
            traj = np.stack([np.asarray(t, dtype=np.float64).reshape(-1, 6)[: self.N_p + 1] for t in obs_trajectories])
            if traj.shape[1] != self.N_p + 1:
                raise ValueError("each obstacle trajectory needs N_p + 1 = %d rows" % (self.N_p + 1))
            traj = traj.reshape(1, n_obs, self.N_p + 1, 6)
        return NlpSolver(self, cfg, traj, _abi.OBSIN_PREDICTED)
