"""Configuration loading for the drop-in surface.

Counterpart of CasaDi_MPC_Optimize_Multishoot/helpers.py:4-24 (`load_config`): same name, same argument, same
return value (the parsed YAML mapping), errors propagate.  The reference reads "mpc_parameters.yaml" from the
current working directory by bare file name (MPC_CBF_optimize_kin.py:7,11); `find_params_file` keeps that rule and
adds the packaged copy as the place to look when the CWD has none.
"""
import os

import yaml

PARAMS_FILE = "mpc_parameters.yaml"
_PKG_DIR = os.path.dirname(os.path.abspath(__file__))


def load_config(config_file):
    with open(config_file, "r", encoding="utf-8") as fh:
        return yaml.safe_load(fh)


def find_params_file(name=PARAMS_FILE):
    if os.path.exists(name):
        return name
    return os.path.join(_PKG_DIR, "sim", name)


def horizon_steps(T_horizon, T_S):
    """N_p exactly as the reference computes it: len(arange(0, H + T_S, T_S)) - 1  (kin.py:32-33)."""
    import numpy as np
    return len(np.arange(0, T_horizon + T_S, T_S, dtype=float)) - 1
