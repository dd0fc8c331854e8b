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


def pytest_collection_modifyitems(config, items):
    """`gpu` tests are skipped (not failed) where no GPU is visible, so a plain
    `pytest tests/` works in the CPU container; with a GPU they must run through
    the HIP library (the product path raises if it is missing)."""
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def assert_params_close(actual, desired, lr, nsteps, tight=1e-6, frac_loose=0.002):
    """Post-Adam parameters.  Adam's update lr*m/(sqrt(v)+eps) is ill-conditioned
    for elements whose gradient is at rounding-noise level (|g| ~ 1e-8: a 1e-10
    reassociation difference moves the update by ~1 % of lr), so the check is
    two-tier: all but max(3, 0.2 %) of the elements agree to `tight`, and every
    element agrees to 10 % of the total distance Adam can move it (lr * nsteps):
    at |g| ~ eps = 1e-8 the first update is lr*g/(|g|+eps), so a 2e-9 difference
    in g already moves it by 5 % of lr."""
    a = np.asarray(actual, dtype=np.float64).ravel()
    d = np.asarray(desired, dtype=np.float64).ravel()
    err = np.abs(a - d)
    assert err.max() <= 0.10 * lr * nsteps + tight, f"max err {err.max():.3e}"
    n_loose = int((err > tight).sum())
    assert n_loose <= max(3, frac_loose * err.size), f"{n_loose} of {err.size} elements beyond {tight}"
