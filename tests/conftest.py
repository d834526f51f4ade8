"""Shared pytest configuration: the `gpu` marker and helpers to reach the oracle and the fixtures."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
DEFAULT_SEED = 71892305  # the reference's DEFAULT_SEED (tests/conftest.py:22 there)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture
def rng():
    return np.random.default_rng(DEFAULT_SEED)


@pytest.fixture(autouse=True)
def _clean_pivot_flags():
    """Tests that feed non-positive-definite matrices on purpose must not leak their (deferred) error into the next test."""
    yield
    try:
        from markovflow_amd import _lib
        if _lib._flags:
            _lib._take_failures(synchronise=True)
    except Exception:
        pass
