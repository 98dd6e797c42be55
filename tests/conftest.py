import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def lib():
    from parakeet_slam_amd import _lib

    if os.environ.get("PK_TEST_LIB"):  # diagnostics only: the suite against another build of the library (an A/B, a regression check)
        _lib.LIB_PATH = os.path.join(ROOT, "parakeet_slam_amd", os.environ["PK_TEST_LIB"])
    return _lib


def relerr(a, b, floor=1e-300):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b) / (np.abs(b) + floor)))
