import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# include/bark_hip_testing.h: the launch-failure hook exists only in processes started with this variable (read when the
# library is loaded, which no test module has done yet; child processes of the tests inherit it)
os.environ.setdefault("BARK_TEST_HOOKS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: g[k] for k in g.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden
