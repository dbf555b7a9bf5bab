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


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))
        return cache[name]

    return load


def assert_close(a, b, rtol=1e-4, atol=0.0, what="", floor=0.1):
    """north_star tolerance: 1e-4 relative (fp32).  `atol` is stated per call site where the
    quantity has a natural absolute floor (e.g. likelihoods are floored at 1e-9).  fp32 summation
    error is relative to the magnitude of the summed terms, not of a cancelling result, so elements
    smaller than `floor` x max|b| are held to rtol x floor x max|b|."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    scale = max(float(np.abs(b).max()), 1e-30)
    err = np.abs(a - b)
    tol = atol + rtol * np.maximum(np.abs(b), floor * scale)
    bad = err > tol
    assert not bad.any(), (f"{what}: {bad.sum()}/{bad.size} out of tolerance; max err {err.max():.3e} "
                           f"(scale {scale:.3e}) at {np.unravel_index(err.argmax(), err.shape)}")
