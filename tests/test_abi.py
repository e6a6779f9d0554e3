"""C-ABI checks that need no GPU: the library loads, exports every symbol include/mpcbatch.h declares, the ctypes
mirror of mpcb_config has the C size, and the host-only entry points behave."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from mpc_motion_planning_amd import _abi, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "mpcbatch.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mpcb_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    names = header_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(L, n), "libmpcbatch.so lacks %s" % n
    assert sorted(_lib.SIGNATURES) == names, "ctypes SIGNATURES out of sync with the header"


def test_config_struct_matches_c_layout():
    cfg = _abi.MpcbConfig()
    assert _lib.lib().mpcb_default_config(C.byref(cfg), _abi.MODEL_KIN, 30, 0.1) == 0
    assert cfg.struct_size == C.sizeof(_abi.MpcbConfig)
    assert (cfg.N, cfg.T, cfg.max_iter) == (30, 0.1, 100)
    assert list(cfg.Q)[:4] == [1e1, 1e5, 3e5, 1e4] and list(cfg.R) == [1e4, 1e4] and list(cfg.DR) == [1e5, 1e2]
    assert cfg.u_hi[0] == pytest.approx(35 * np.pi / 180) and cfg.du_hi[0] == pytest.approx(5 * np.pi / 180 * 0.1)
    assert cfg.tol == 1e-8 and cfg.bound_relax == 1e-8 and cfg.max_gradient == 100.0
    cfg.struct_size = 12
    h = C.c_void_p()
    assert _lib.lib().mpcb_create(C.byref(cfg), 0, C.byref(h)) == _abi.E_INVALID


def test_dims_follow_the_reference_counts():
    from mpc_motion_planning_amd.solver import default_config, dims
    # kin N=30: len(lbx)=184, len(lbg)=124+29+30*n_obs
    assert dims(default_config(N=30, n_obs=1)) == (4, 184, 183)
    assert dims(default_config(N=30, n_obs=3)) == (4, 184, 243)
    assert dims(default_config(N=20, n_obs=0)) == (4, 124, 103)
    assert dims(default_config(N=50, n_obs=1)) == (4, 304, 303)
    assert dims(default_config(model=_abi.MODEL_DYN, N=40, n_obs=1))[:2] == (6, 326)


def test_model_rhs_host():
    from mpc_motion_planning_amd.solver import default_config, model_rhs
    f = model_rhs(default_config(), [0, 3, 0.1, 15], [0.05, 1.5])
    assert np.allclose(f, [15 * np.cos(0.1), 15 * np.sin(0.1), 15 * np.tan(0.05) / 2.6, 1.5], rtol=0, atol=1e-15)


def test_version_string():
    assert b"gfx950" in _lib.lib().mpcb_version()


def test_shard_bounds_of_the_library_match_the_host_rule():
    """mpcb_shard_bounds (what the library cuts a device group's batch with) against sharding.shard_bounds (what bench.py and
    the gloo test use): contiguous, covering, sizes differing by at most one."""
    from mpc_motion_planning_amd.sharding import shard_bounds
    from mpc_motion_planning_amd import solver
    for B in (0, 1, 7, 37, 4096, 65537):
        for W in (1, 2, 3, 8):
            cuts = [solver.shard_bounds(B, W, r) for r in range(W)]
            assert cuts == [shard_bounds(B, W, r) for r in range(W)]
            assert cuts[0][0] == 0 and cuts[-1][1] == B and all(cuts[i][1] == cuts[i + 1][0] for i in range(W - 1))
    lo, hi = C.c_int64(), C.c_int64()
    assert _lib.lib().mpcb_shard_bounds(10, 2, 2, C.byref(lo), C.byref(hi)) == _abi.E_INVALID
    w, r = C.c_int32(), C.c_int32()
    assert _lib.lib().mpcb_comm_info(None, C.byref(w), C.byref(r)) == _abi.E_INVALID


def test_missing_rccl_is_an_error_code_not_a_crash():
    """librccl is loaded on first use; when it cannot be loaded the group entry points return MPCB_E_DEVICE with a message
    (abi 2 assigned a NULL dlerror() to a std::string there).  Own process: the load is attempted once per process."""
    import subprocess
    import sys
    code = ("import ctypes as C, sys; sys.path.insert(0, %r)\n"
            "from mpc_motion_planning_amd import _lib, _abi\n"
            "L = _lib.lib(); buf = C.create_string_buffer(128)\n"
            "rc = L.mpcb_comm_unique_id(C.cast(buf, C.c_void_p)); msg = L.mpcb_last_error(None)\n"
            "rc2 = L.mpcb_comm_unique_id(C.cast(buf, C.c_void_p))\n"
            "print(rc, rc2, msg.decode())\n" % ROOT)
    env = dict(os.environ, MPCB_RCCL_LIB="/nonexistent/librccl.so")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    rc, rc2, msg = out.stdout.strip().split(" ", 2)
    assert int(rc) == _abi.E_DEVICE and int(rc2) == _abi.E_DEVICE and "librccl" in msg and "nonexistent" in msg


def test_inflight_argument_checks_need_no_device():
    assert _lib.lib().mpcb_set_inflight(None, 2) == _abi.E_INVALID


def test_integration_stub_declares_the_same_struct():
    """The ctypes structure INTEGRATION.md §B shows a maintainer is the one the library expects (same fields, same order, same size)."""
    text = open(os.path.join(os.path.dirname(__file__), "..", "INTEGRATION.md")).read()
    m = re.search(r"class _Cfg\(C\.Structure\):.*?\n(?=\n)", text, re.S)
    assert m, "stub not found"
    ns = {"C": C}
    exec(m.group(0), ns)
    stub = ns["_Cfg"]
    assert [f[0] for f in stub._fields_] == [f[0] for f in _abi.MpcbConfig._fields_]
    assert C.sizeof(stub) == C.sizeof(_abi.MpcbConfig)
    cfg = stub()
    raw = C.CDLL(_lib.lib()._name)                       # an untyped binding, as the stub's own C.CDLL(...)
    assert raw.mpcb_default_config(C.byref(cfg), 0, C.c_int32(30), C.c_double(0.1)) == 0 and cfg.struct_size == C.sizeof(stub)
    assert cfg.second_start == 3 and cfg.acceptable_iter == 15 and cfg.acceptable_tol == 1e-8 and cfg.start_steer == 0.03
