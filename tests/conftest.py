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


def golden(name):
    path = os.path.join(GOLDEN, name + ".npz")
    if not os.path.exists(path):
        pytest.skip("golden fixture %s not generated" % name)
    return np.load(path, allow_pickle=False)


def gaussian_design(m, n, seed):
    """Same call sequence as the reference factory (accbpg/applications.py:47-49)."""
    if seed > 0:
        np.random.seed(seed)
    return np.random.randn(m, n)
