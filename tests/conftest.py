import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_swarm
    oracle_swarm.lib()
    return oracle_swarm


@pytest.fixture(scope="session")
def mrs():
    """The product package with libmrs_swarm.so built in-tree (hipcc cross-compiles on CPU boxes)."""
    import mrs_multirotor_simulator_amd as M
    from mrs_multirotor_simulator_amd import build
    build.build_library()
    M.load_library()
    return M
