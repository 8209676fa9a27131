import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_err(a, b):
    """max-norm relative error used throughout: max|a-b| / max|b|."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.max(np.abs(b))
    err = float(np.max(np.abs(a - b)) / (den if den > 0 else 1.0))
    log = os.environ.get("SCARLET_LOG_REL_ERR")
    if log:            # evidence for profiles/: every max-norm error a test computed, with the test's name
        with open(log, "a") as f:
            f.write("%.3e  %s\n" % (err, os.environ.get("PYTEST_CURRENT_TEST", "?")))
    return err
