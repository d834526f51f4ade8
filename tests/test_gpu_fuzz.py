"""A short run of the randomised GPU-vs-oracle campaign (scripts/fuzz_parity.py): random shapes across the serial /
parallel-in-time thresholds, ragged chunk tails, d = 1..9, m = 1..3, random explicit chunk counts; log-likelihood, posterior
marginals, prior covariance scan, Cholesky, both solves and the KL against the numpy oracle / dense linear algebra."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def test_random_shapes_against_the_oracle():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "fuzz_parity.py")
    spec = importlib.util.spec_from_file_location("fuzz_parity", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    worst = mod.run(40, seed=71892305)
    assert all(v < 1e-8 for v in worst.values()), worst
