"""ctypes front-end of tests/emu (CPU stepping of the kernel source).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

from mpc_motion_planning_amd._abi import dptr, iptr, OBSIN_STATIC, OBSIN_PREDICTED

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
        _LIB = C.CDLL(os.path.join(_HERE, "_build", "libmpcbemu.so"))
        _LIB.mpcb_emu_solve.restype = C.c_int
    return _LIB


def solve(cfg, x0, xs, obs=None, z0=None, trace_instance=-1, tgrid=None):
    x0 = np.ascontiguousarray(np.atleast_2d(x0), dtype=np.float64)
    xs = np.ascontiguousarray(np.atleast_2d(xs), dtype=np.float64)
    B = x0.shape[0]; N = cfg.N; nx = cfg.nx()
    nz = 2 * N + nx * (N + 1)
    nrate = sum(1 for i in range(2) if np.isfinite(cfg.du_lo[i]) or np.isfinite(cfg.du_hi[i]))
    ng = nx * (N + 1) + nrate * (N - 1) + cfg.n_obs * (N + 1 if cfg.obs_terminal else N)
    kind = OBSIN_STATIC
    if cfg.n_obs > 0:
        obs = np.ascontiguousarray(obs, dtype=np.float64)
        if obs.size == B * cfg.n_obs * (N + 1) * 6:
            kind = OBSIN_PREDICTED
    else:
        obs = None
    if z0 is not None:
        z0 = np.ascontiguousarray(z0, dtype=np.float64).reshape(B, nz)
    z = np.zeros((B, nz)); obj = np.zeros(B); st = np.zeros(B, np.int32); it = np.zeros(B, np.int32)
    kkt = np.zeros((B, 4)); lam_g = np.zeros((B, ng)); lam_x = np.zeros((B, nz))
    trace = np.zeros((cfg.max_iter + 1, 8)) if trace_instance >= 0 else None
    rc = lib().mpcb_emu_solve(C.byref(cfg), C.c_int32(B), dptr(x0), dptr(xs), dptr(obs), C.c_int32(kind), dptr(z0),
                              dptr(z), dptr(obj), iptr(st), iptr(it), dptr(kkt), dptr(lam_g), dptr(lam_x),
                              dptr(trace), C.c_int32(trace_instance), dptr(None if tgrid is None else np.ascontiguousarray(tgrid, dtype=np.float64)))
    if rc != 0:
        raise RuntimeError("mpcb_emu_solve failed with code %d" % rc)
    return dict(z=z, obj=obj, status=st, iters=it, kkt=kkt, lam_g=lam_g, lam_x=lam_x, trace=trace)


def lds_bytes(cfg, restoration_pass=False):
    """LDS bytes of one instance as the kernels lay it out (layout_kin / layout_dyn of the kernel headers)."""
    f = lib().mpcb_emu_lds_bytes; f.restype = C.c_int64
    return int(f(C.byref(cfg), C.c_int32(1 if restoration_pass else 0)))


def dyn_model(cfg, X, U, lam):
    X = np.ascontiguousarray(X, np.float64); U = np.ascontiguousarray(U, np.float64); lam = np.ascontiguousarray(lam, np.float64)
    F = np.zeros(6); jac = np.zeros(16); hess = np.zeros(13)
    lib().mpcb_emu_dyn_model(C.byref(cfg), dptr(X), dptr(U), dptr(lam), dptr(F), dptr(jac), dptr(hess))
    return F, jac, hess
