"""pytest configuration: markers, repo root on sys.path, golden-fixture loader."""
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as d:
        return {k: d[k] for k in d.files}


@pytest.fixture
def golden():
    return load_golden


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def bar(name, measured, limit):
    """assert measured < limit — and, with TRK_BARS_LOG=<file>, append `name measured limit` there: the record the bars of the
    loosened tests are set from (bar = 2 x the measured value on the MI355X)."""
    measured = float(measured)
    log = os.environ.get("TRK_BARS_LOG")
    if log:
        with open(log, "a") as fh:
            fh.write(f"{name} {measured:.3e} {limit:.3e}\n")
    assert measured < limit, (name, measured, limit)


def maxrel(a, b):
    """max_i |a_i / b_i - 1| of two scalar histories."""
    a, b = np.asarray(a, dtype=np.float64).reshape(-1), np.asarray(b, dtype=np.float64).reshape(-1)
    return float(np.max(np.abs(a / b - 1.0))) if a.shape == b.shape else float("inf")
