"""pytest configuration: the ``gpu`` marker, import paths, and shared golden-vector loaders."""
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, GOLDEN)
sys.path.insert(0, os.path.join(REPO, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _load(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden_zarr():
    """The reference's own 18 known-answer arrays (float32)."""
    return _load("reference_zarr.npz")


@pytest.fixture(scope="session")
def golden_generated():
    """float64 outputs captured by importing the reference (tests/golden/make_golden.py)."""
    return _load("reference_generated.npz")


@pytest.fixture(scope="session")
def golden_spec():
    return _load("reference_spec.npz")
