"""ctypes mirror of include/mpcbatch.h (struct mpcb_config and constants).

Pure declarations, no library is loaded here.  Field order and types must match the header exactly;
tests/test_abi.py checks sizeof and the exported symbols against the header.
"""
import ctypes as C

ABI_VERSION = 3

OK, E_INVALID, E_BOUNDS, E_DEVICE, E_UNSUPPORTED = 0, -1, -2, -3, -4
ST_SOLVED, ST_MAXITER, ST_LINESEARCH, ST_INFEASIBLE_X0, ST_NUMERIC, ST_INFEASIBLE, ST_RESTO_FAILED = 0, 1, 2, 3, 4, 5, 6
ST_ACCEPTABLE = 8
STATUS_NAMES = {0: "solved", 1: "max_iter", 2: "line_search", 3: "infeasible_x0", 4: "numeric", 5: "locally_infeasible",
                6: "restoration_failed", 8: "acceptable"}
MODEL_KIN, MODEL_DYN = 0, 1
OBS_KEEPOUT, OBS_DCBF = 0, 1
OBSIN_STATIC, OBSIN_PREDICTED = 0, 1
MU_MONOTONE = 0
INT_EULER, INT_RK4 = 0, 1
CL_HOLD_ON_FAILURE, CL_ADVANCE_FIRST_ONLY = 1, 2
OBSMOVE_STATIC, OBSMOVE_PREDICTED, OBSMOVE_CURRENT = 0, 1, 2
NX_MAX, NU, NOBS_MAX, N_MAX = 6, 2, 8, 63
UNIQUE_ID_BYTES = 128
SCENES_C2, SCENES_C3, SCENES_C4 = 2, 3, 4

_d = C.c_double
_i = C.c_int32


class MpcbConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("model", _i), ("N", _i), ("n_obs", _i), ("obs_mode", _i), ("obs_terminal", _i),
        ("du0_cost", _i), ("rate_interleaved", _i), ("max_iter", _i), ("mu_strategy", _i), ("init_rollout", _i),
        ("integrator", _i), ("restoration", _i),
        ("T", _d), ("gamma", _d),
        ("Q", _d * NX_MAX), ("R", _d * NU), ("DR", _d * NU), ("u_last", _d * NU),
        ("u_lo", _d * NU), ("u_hi", _d * NU),
        ("x_lo", _d * NX_MAX), ("x_hi", _d * NX_MAX),
        ("du_lo", _d * NU), ("du_hi", _d * NU),
        ("obs_hmin", _d), ("ego_hl", _d), ("ego_hw", _d), ("safe_disl", _d), ("safe_disw", _d),
        ("obs_sx_fixed", _d), ("obs_sy_fixed", _d),
        ("veh_l", _d), ("veh_m", _d), ("veh_lf", _d), ("veh_lr", _d), ("veh_Iz", _d),
        ("Fymax_f", _d), ("Fymax_r", _d), ("aopt_f", _d), ("aopt_r", _d),
        ("tol", _d), ("mu_init", _d), ("bound_push", _d), ("bound_frac", _d), ("bound_relax", _d),
        ("max_gradient", _d),
        ("dual_inf_tol", _d), ("constr_viol_tol", _d), ("compl_inf_tol", _d),
        ("acceptable_tol", _d), ("acceptable_obj_change_tol", _d), ("acceptable_constr_viol_tol", _d),
        ("acceptable_dual_inf_tol", _d), ("acceptable_compl_inf_tol", _d),
        ("acceptable_iter", _i), ("second_start", _i), ("start_steer", _d),
    ]

    def copy(self):
        other = MpcbConfig()
        C.memmove(C.byref(other), C.byref(self), C.sizeof(MpcbConfig))
        return other

    def nx(self):
        return 6 if self.model == MODEL_DYN else 4

    def nz(self):
        return NU * self.N + self.nx() * (self.N + 1)


def dptr(a):
    """double* of a C-contiguous float64 numpy array (or NULL for None)."""
    if a is None:
        return None
    return a.ctypes.data_as(C.POINTER(C.c_double))


def iptr(a):
    if a is None:
        return None
    return a.ctypes.data_as(C.POINTER(C.c_int32))
