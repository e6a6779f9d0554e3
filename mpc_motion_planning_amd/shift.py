"""Receding-horizon shift — the `shift_movement` every reference driver defines
(CasaDi_MPC_Optimize_Multishoot/main_cbf_kin_c_sim.py:16-26).

    t, st, u_end, x_f = shift(T, t0, x0, u, x_f, f)
      st    = x0 + T * f(x0, u[0])            plant step with the first control (explicit Euler)
      u_end = [u[1:]; u[-1]]                  (N,2)  warm start of the controls
      x_f   = [x_f[1:]; x_f[-1]]              (N+1,nx) warm start of the states
"""
import numpy as np


def shift(T, t0, x0, u, x_f, f):
    fv = f(x0, u[0, :])
    fv = fv.full() if hasattr(fv, "full") else np.asarray(fv, dtype=np.float64).reshape(-1, 1)
    st = np.asarray(x0, dtype=np.float64).reshape(-1, 1) + T * fv
    t = t0 + T
    u_end = np.concatenate((u[1:], u[-1:]))
    x_f = np.concatenate((x_f[1:], x_f[-1:]), axis=0)
    return t, st, u_end, x_f


shift_movement = shift
