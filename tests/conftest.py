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


@pytest.fixture
def yaml_horizon3(tmp_path, monkeypatch):
    """BASELINE's N = 30: the package's mpc_parameters.yaml (the reference's values, horizon 5 -> N_p = 50) with horizon: 3,
    placed in the current working directory, which is where the reference's classes look for it (kin.py:7,11)."""
    src = os.path.join(ROOT, "mpc_motion_planning_amd", "sim", "mpc_parameters.yaml")
    text = open(src).read().replace("horizon: 5", "horizon: 3")
    assert "horizon: 3" in text
    (tmp_path / "mpc_parameters.yaml").write_text(text)
    monkeypatch.chdir(tmp_path)
    return tmp_path
