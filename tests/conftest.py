import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def gpu_solver_factory():
    """BatchSolver factory; fails (does not skip) when the HIP library or a device is missing: GPU tests must
    never pass on anything but the native path."""
    from mpc_motion_planning_amd import solver
    assert solver.device_count() >= 1, "no HIP device visible: -m gpu tests need the MI355X box"
    return solver.BatchSolver


GOLDEN = os.path.join(ROOT, "tests", "golden")
