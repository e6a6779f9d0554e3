"""Straight global reference path and local preview window.

Same class, method names, arguments and return values as CasaDi_MPC_Optimize_Multishoot/RefPathGenerator.py:
  define_ref_path(x0, xs, dt)  -> (M,4) array [x, y=xs[1], phi=xs[2], vx=xs[3]] at 1 m spacing from x0[0] to xs[0] (:9-24)
  find_ref_traj(x0, xs, T_horizon, dt, last_idx) -> ((N_p+1,4) window, index of the nearest path point) (:27-59)
The window is numerically inert in the reference's NLP (blend weight aa = 0, MPC_CBF_optimize_kin.py:194-197);
it is kept because the drivers call it every step and plot it.
"""
import numpy as np


def _scalar(v):
    return float(np.asarray(v, dtype=np.float64).reshape(-1)[0])


class RefPathGenerator:
    def __init__(self):
        self.ref_global = None
        self.step_x = None
        self.ref_len = None

    def define_ref_path(self, x0, xs, dt):
        xa, xb = _scalar(x0[0]), _scalar(xs[0])
        self.step_x = 1
        if xb > xa:
            gx = np.arange(xa, xb + self.step_x, self.step_x)
        else:
            gx = np.arange(xa, xb - self.step_x, -self.step_x)
        cols = [gx] + [np.full_like(gx, _scalar(xs[i]), dtype=np.float64) for i in (1, 2, 3)]
        self.ref_global = np.stack(cols, axis=1).astype(np.float64)
        self.ref_len = len(self.ref_global)
        return self.ref_global

    def find_ref_traj(self, x0, xs, T_horizon, dt, last_idx):
        N_p = int(T_horizon / dt)
        preview_v = 0.5 * _scalar(x0[3]) + 0.5 * _scalar(xs[3])
        preview_idx = int(preview_v * T_horizon / self.step_x)
        lo = max(0, last_idx - 5)
        hi = min(self.ref_len, last_idx + preview_idx)
        px, py = _scalar(x0[0]), _scalar(x0[1])
        # first local minimum of the distance along the search window (the reference stops at the first increase)
        min_idx = lo
        best = np.inf
        for i in range(lo, hi):
            d = np.hypot(self.ref_global[i, 0] - px, self.ref_global[i, 1] - py)
            if d < best:
                best = d
                min_idx = i
            else:
                break
        idx = np.linspace(min_idx, min_idx + preview_idx, N_p + 1)
        idx = np.clip(idx, 0, self.ref_len - 1).astype(int)
        return self.ref_global[idx, :], min_idx
