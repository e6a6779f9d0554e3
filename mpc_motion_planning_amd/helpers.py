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


def get_config_path(config_file):
    """Full path of a configuration file that sits next to the package's own copy (helpers.py:26-38 of the reference resolves
    against the directory of helpers.py; here that directory is the package's sim/ folder, where mpc_parameters.yaml ships)."""
    return os.path.join(_PKG_DIR, "sim", config_file)


def validate_config(config):
    """True when every section the classes read is present.  The reference's version (helpers.py:40-55) asks for a section
    `velocity_constraints` that its own YAML does not have and is never called; this one checks the sections that are used."""
    needed = ("mpc_params", "vehicle_params", "tire_params", "kinematics_constraints", "dynamics_constraints")
    missing = [k for k in needed if k not in config]
    for k in missing:
        print("configuration lacks the section '%s'" % k)
    return not missing


def print_config(config):
    """Dump of the parsed YAML, section by section (helpers.py:57-68)."""
    for section, values in config.items():
        print("\n%s:" % section)
        if isinstance(values, dict):
            for key, value in values.items():
                print("  %s: %s" % (key, value))
        else:
            print("  %s" % (values,))


def find_params_file(name=PARAMS_FILE):
    if os.path.exists(name):
        return name
    return os.path.join(_PKG_DIR, "sim", name)


def horizon_steps(T_horizon, T_S):
    """N_p exactly as the reference computes it: len(arange(0, H + T_S, T_S)) - 1  (kin.py:32-33)."""
    import numpy as np
    return len(np.arange(0, T_horizon + T_S, T_S, dtype=float)) - 1
