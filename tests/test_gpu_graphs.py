"""HIP-graph capture of whole evaluations (markovflow_amd.graphs): replays give the eager results and follow in-place updates
of the inputs."""
import numpy as np
import pytest
import torch

import markovflow_amd as mfa
from test_gpu_kalman import build_kf, random_ssm

pytestmark = pytest.mark.gpu


def test_captured_log_likelihood_and_solve_follow_input_updates(rng):
    kw = random_ssm(rng, (3,), 200, 4, 1, well=True)
    kf = build_kf(kw, np.array([[0.7]]))
    eager = float(kf.log_likelihood())
    call = mfa.graphs.capture(kf.log_likelihood)
    assert float(call()) == pytest.approx(eager, rel=1e-13)
    # new observations, written in place: the replay sees them
    y_new = torch.randn_like(kf.observations)
    kf.observations.copy_(y_new)
    replayed = float(call())
    assert replayed == pytest.approx(float(kf.log_likelihood()), rel=1e-13)
    assert abs(replayed - eager) > 1e-6

    chol = kf._k_inv_post.cholesky
    rhs = torch.randn(3, 200, 4, dtype=torch.float64, device="cuda:0")
    solve = mfa.graphs.capture(lambda: chol.solve(rhs))
    np.testing.assert_allclose(solve().cpu().numpy(), chol.solve(rhs).cpu().numpy(), rtol=1e-13)
    rhs.mul_(2.0)
    np.testing.assert_allclose(solve().cpu().numpy(), chol.solve(rhs).cpu().numpy(), rtol=1e-13)
