"""ctypes front-end of the CPU oracle (oracle/mpc_oracle.cpp).  TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the product package.
Parity status: unpinned versus CasADi+IPOPT (absent, see the header of mpc_oracle.cpp and DESIGN.md).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from mpc_motion_planning_amd._abi import MpcbConfig, dptr, iptr, OBSIN_STATIC, OBSIN_PREDICTED, MODEL_KIN

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "_build", "libmpcoracle.so")
    src = os.path.join(_HERE, "mpc_oracle.cpp")
    hdr = os.path.join(_HERE, "..", "include", "mpcbatch.h")
    stale = (not os.path.exists(so)) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(so) for p in (src, hdr))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.mpco_solve.restype = C.c_int
        _LIB.mpco_default_config.restype = C.c_int
        _LIB.mpco_dims.restype = C.c_int
    return _LIB


def use_native_build():
    """bench.py's cpu_baseline leg only: compile the oracle once more with -O3 -march=native for THE MACHINE THIS RUNS ON (a
    temporary directory: such a binary must not travel between hosts) and make it the library solve() calls from now on.
    The test-suite keeps the portable -O2 build, whose results the golden vectors were generated with (FMA contraction under
    -march=native moves last digits).  Returns the flags used, or None when no compiler is at hand (the -O2 build stays)."""
    global _LIB
    import tempfile
    flags = ["-O3", "-march=native", "-std=c++17", "-fPIC", "-fopenmp", "-shared", "-Wl,-Bsymbolic"]
    so = os.path.join(tempfile.mkdtemp(prefix="mpcoracle_native_"), "libmpcoracle_native.so")
    try:
        subprocess.check_call(["g++"] + flags + ["-o", so, os.path.join(_HERE, "mpc_oracle.cpp")], stderr=subprocess.DEVNULL)
    except (OSError, subprocess.CalledProcessError):
        return None
    _LIB = C.CDLL(so)
    _LIB.mpco_solve.restype = C.c_int
    _LIB.mpco_default_config.restype = C.c_int
    _LIB.mpco_dims.restype = C.c_int
    return " ".join(flags[:2])


def default_config(model=MODEL_KIN, N=30, T=0.1, n_obs=0):
    cfg = MpcbConfig()
    rc = lib().mpco_default_config(C.byref(cfg), C.c_int32(model), C.c_int32(N), C.c_double(T))
    assert rc == 0
    cfg.n_obs = n_obs
    return cfg


def dims(cfg):
    nx, nz, ng = C.c_int32(), C.c_int32(), C.c_int32()
    lib().mpco_dims(C.byref(cfg), C.byref(nx), C.byref(nz), C.byref(ng))
    return nx.value, nz.value, ng.value


def solve(cfg, x0, xs, obs=None, z0=None, threads=0, want_multipliers=True, tgrid=None):
    """Returns dict(z, obj, status, iters, kkt, lam_g, lam_x); arrays are [B, ...]."""
    x0 = np.ascontiguousarray(np.atleast_2d(x0), dtype=np.float64)
    xs = np.ascontiguousarray(np.atleast_2d(xs), dtype=np.float64)
    B = x0.shape[0]
    nx, nz, ng = dims(cfg)
    assert x0.shape == (B, nx) and xs.shape == (B, nx)
    kind = OBSIN_STATIC
    if cfg.n_obs > 0:
        obs = np.ascontiguousarray(obs, dtype=np.float64)
        if obs.ndim == 4 or (obs.ndim == 3 and obs.shape[0] == cfg.n_obs and obs.shape[1] == cfg.N + 1 and B == 1 and obs.shape[-1] == 6 and obs.shape != (B, cfg.n_obs, 6)):
            kind = OBSIN_PREDICTED
            obs = obs.reshape(B, cfg.n_obs, cfg.N + 1, 6)
        else:
            obs = obs.reshape(B, cfg.n_obs, 6)
    else:
        obs = None
    if z0 is not None:
        z0 = np.ascontiguousarray(z0, dtype=np.float64).reshape(B, nz)
    tg = None
    if tgrid is not None:
        tg = np.ascontiguousarray(tgrid, dtype=np.float64).reshape(-1)
        assert tg.size == cfg.N, "the time grid has one step length per stage"
    z = np.zeros((B, nz)); obj = np.zeros(B); st = np.zeros(B, np.int32); it = np.zeros(B, np.int32)
    kkt = np.zeros((B, 4))
    lam_g = np.zeros((B, ng)) if want_multipliers else None
    lam_x = np.zeros((B, nz)) if want_multipliers else None
    rc = lib().mpco_solve(C.byref(cfg), C.c_int32(B), dptr(x0), dptr(xs), dptr(obs), C.c_int32(kind), dptr(z0),
                          dptr(z), dptr(obj), iptr(st), iptr(it), dptr(kkt), dptr(lam_g), dptr(lam_x),
                          C.c_int32(threads), dptr(tg))
    if rc != 0:
        raise RuntimeError("mpco_solve failed with code %d" % rc)
    return dict(z=z, obj=obj, status=st, iters=it, kkt=kkt, lam_g=lam_g, lam_x=lam_x)


def model_eval(cfg, X, U, lam, ad=False):
    nx = cfg.nx()
    X = np.ascontiguousarray(X, np.float64); U = np.ascontiguousarray(U, np.float64)
    lam = np.ascontiguousarray(lam, np.float64)
    F = np.zeros(nx); A = np.zeros((nx, nx)); Bm = np.zeros((nx, 2)); H = np.zeros((nx + 2, nx + 2))
    rc = lib().mpco_model_eval(C.byref(cfg), dptr(X), dptr(U), dptr(lam), dptr(F), dptr(A), dptr(Bm), dptr(H),
                               C.c_int32(1 if ad else 0))
    assert rc == 0
    return F, A, Bm, H
