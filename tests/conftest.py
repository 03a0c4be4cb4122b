import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def po():
    """The CPU oracle (test infrastructure)."""
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def small_workload():
    from srrg2_laser_slam_2d_amd import synth
    return synth.make_workload(6, 20000, seed=3)


@pytest.fixture(scope="session")
def ctx():
    """One lsm2d context on cuda:0 -- fails loudly if the HIP library or the GPU is missing."""
    from srrg2_laser_slam_2d_amd import api
    c = api.Context(0)
    yield c
    c.close()


def golden_path(name):
    return os.path.join(ROOT, "tests", "golden", name)
