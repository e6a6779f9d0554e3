"""Loader of libmpcbatch.so (the HIP library).  There is no fallback of any kind: if the shared library is not
built, or no HIP device is present, the calls raise."""
import ctypes as C
import os

from ._abi import MpcbConfig

_HERE = os.path.dirname(os.path.abspath(__file__))
# MPCB_LIB names another build of the same library (diagnostic builds of tools/sync_ab.sh); there is still no fallback
LIB_PATH = os.environ.get("MPCB_LIB") or os.path.join(_HERE, "lib", "libmpcbatch.so")

_lib = None

_PD = C.POINTER(C.c_double)
_PI = C.POINTER(C.c_int32)
_H = C.c_void_p

# name -> (restype, argtypes); must list every function of include/mpcbatch.h (tests/test_abi.py checks)
SIGNATURES = {
    "mpcb_version": (C.c_char_p, []),
    "mpcb_default_config": (C.c_int, [C.POINTER(MpcbConfig), C.c_int32, C.c_int32, C.c_double]),
    "mpcb_dims": (C.c_int, [C.POINTER(MpcbConfig), _PI, _PI, _PI]),
    "mpcb_device_count": (C.c_int, []),
    "mpcb_create": (C.c_int, [C.POINTER(MpcbConfig), C.c_int32, C.POINTER(_H)]),
    "mpcb_destroy": (C.c_int, [_H]),
    "mpcb_last_error": (C.c_char_p, [_H]),
    "mpcb_set_bounds": (C.c_int, [_H, _PD, _PD, C.c_int32, _PD, _PD, C.c_int32]),
    "mpcb_set_time_grid": (C.c_int, [_H, _PD, C.c_int32]),
    "mpcb_solve": (C.c_int, [_H, C.c_int32, _PD, _PD, _PD, C.c_int32, _PD, _PD, _PD, _PI, _PI, _PD, _PD, _PD]),
    "mpcb_solve_device": (C.c_int, [_H, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    "mpcb_set_inflight": (C.c_int, [_H, C.c_int32]),
    "mpcb_solve_trace": (C.c_int, [_H, _PD, _PD, _PD, C.c_int32, _PD, _PD, _PI, _PI, _PD]),
    "mpcb_closed_loop": (C.c_int, [_H, C.c_int32, C.c_int32, _PD, _PD, _PD, C.c_int32, C.c_int32, _PD, _PD, _PI, _PI]),
    "mpcb_sample_scenes": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mpcb_closed_loop_sampled": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, C.c_int32, C.c_int32, C.c_int32, _PD, _PD, _PD, _PD, _PI, _PI]),
    "mpcb_predict_obstacles": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_double, _PD, _PD]),
    "mpcb_ref_path_window": (C.c_int, [_H, C.c_int32, C.c_double, _PD, _PD, C.c_double, C.c_double, _PI, _PD]),
    "mpcb_shard_bounds": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "mpcb_comm_unique_id": (C.c_int, [C.c_void_p]),
    "mpcb_comm_init_rank": (C.c_int, [_H, C.c_void_p, C.c_int32, C.c_int32]),
    "mpcb_set_devices": (C.c_int, [_H, _PI, C.c_int32]),
    "mpcb_comm_info": (C.c_int, [_H, _PI, _PI]),
    "mpcb_allgather": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_uint64]),
    "mpcb_allreduce": (C.c_int, [_H, _PD, C.c_int32, C.c_int32]),
    "mpcb_gathered_z": (C.c_int, [_H, C.c_int32, C.POINTER(C.c_void_p)]),
    "mpcb_dev_alloc": (C.c_int, [_H, C.c_uint64, C.POINTER(C.c_void_p)]),
    "mpcb_dev_free": (C.c_int, [_H, C.c_void_p]),
    "mpcb_dev_upload": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_uint64]),
    "mpcb_dev_download": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_uint64]),
    "mpcb_sync": (C.c_int, [_H]),
    "mpcb_stream_wait": (C.c_int, [_H, _H]),
    "mpcb_event_record": (C.c_int, [_H, C.c_int32]),
    "mpcb_event_wait": (C.c_int, [_H, _H, C.c_int32]),
    "mpcb_timing": (C.c_int, [_H, C.c_int32, _PI, _PD, _PD]),
    "mpcb_model_rhs": (C.c_int, [C.POINTER(MpcbConfig), _PD, _PD, _PD]),
}


class MpcbError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libmpcbatch error %d: %s" % (code, msg))
        self.code = code


def lib():
    """The loaded library.  Raises if it has not been built (run `python -c 'import __graft_entry__ as g; g.build()'`
    or `make -C mpc_motion_planning_amd/csrc`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libmpcbatch.so is not built (%s missing); there is no CPU fallback. "
                              "Build it with: make -C mpc_motion_planning_amd/csrc" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, handle=None):
    if rc != 0:
        msg = lib().mpcb_last_error(handle)
        raise MpcbError(rc, msg.decode("utf-8", "replace") if msg else "")
