import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG = "a-2d-lidar-based-slam-system-for-wheeled-mobile-robots_amd"
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pkg(sub=None):
    return importlib.import_module(PKG + ("." + sub if sub else ""))


@pytest.fixture(scope="session")
def syn():
    return pkg("synthetic")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def g1():
    return load_golden("g1_bresenham.npz")


@pytest.fixture(scope="session")
def g2():
    return load_golden("g2_mapping.npz")


@pytest.fixture(scope="session")
def g3():
    return load_golden("g3_icp.npz")


@pytest.fixture(scope="session")
def g4():
    return load_golden("g4_pipeline.npz")
