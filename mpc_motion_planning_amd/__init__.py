"""mpc_motion_planning_amd — batched multiple-shooting MPC on AMD MI355X (gfx950).

Drop-in for the CasADi+IPOPT solve of ZhuorenLi/MPC_motion_planning (CasaDi_MPC_Optimize_Multishoot/):
  MPC_CBF_optimize_kin / _kin_pre / _dyn .MPC_optimize   the reference's problem classes, same surface
  shift(T, t0, x0, u, x_f, f)                             the drivers' shift_movement
  BatchSolver                                             batched extension: thousands of instances per call
The numerical work happens in lib/libmpcbatch.so (hand-written HIP, C ABI in include/mpcbatch.h); Python is
numpy + ctypes only.
"""
from ._abi import (MpcbConfig, MODEL_KIN, MODEL_DYN, OBS_KEEPOUT, OBS_DCBF, OBSIN_STATIC, OBSIN_PREDICTED,
                   ST_SOLVED, ST_MAXITER, ST_LINESEARCH, ST_INFEASIBLE_X0, ST_NUMERIC, STATUS_NAMES)
from .shift import shift, shift_movement

__all__ = ["MpcbConfig", "BatchSolver", "default_config", "shift", "shift_movement", "MODEL_KIN", "MODEL_DYN",
           "OBS_KEEPOUT", "OBS_DCBF", "OBSIN_STATIC", "OBSIN_PREDICTED", "ST_SOLVED", "ST_MAXITER", "ST_LINESEARCH",
           "ST_INFEASIBLE_X0", "ST_NUMERIC", "STATUS_NAMES"]


def __getattr__(name):   # the solver front-end needs the built library; import it lazily
    if name in ("BatchSolver", "default_config", "DeviceArray", "dims", "device_count", "model_rhs"):
        from . import solver
        return getattr(solver, name)
    raise AttributeError(name)
