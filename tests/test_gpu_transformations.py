"""
GPU tests of markovflow_amd.ssm_gaussian_transformations - re-expressing
/root/reference/tests/unit/test_ssm_gaussian_transformations.py: every transform against the dense joint Gaussian
(theta = K^-1 mu, Theta = -1/2 K^-1, eta = mu, H = Sigma + mu mu^T) and the round trips back to the SSM parameters.
"""
import numpy as np
import pytest
import torch

import markovflow_amd as mfa
from markovflow_amd import ssm_gaussian_transformations as G
from oracle import numpy_oracle as O
from test_gpu_kalman import random_ssm

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def tt(x):
    return torch.tensor(np.ascontiguousarray(x), dtype=torch.float64, device=DEV)


def nn(x):
    return x.detach().cpu().numpy()


def dense_joint(kw, s):
    """Mean and covariance of the whole trajectory of series s."""
    a, b, cq = kw["a_s"][s], kw["b_s"][s], kw["chol_q"][s]
    n, d = a.shape[0] + 1, a.shape[-1]
    means, covs = [kw["mu0"][s]], [kw["chol_p0"][s] @ kw["chol_p0"][s].T]
    for k in range(n - 1):
        means.append(a[k] @ means[k] + b[k])
        covs.append(a[k] @ covs[k] @ a[k].T + cq[k] @ cq[k].T)
    big = np.zeros((n * d, n * d))
    for i in range(n):
        c = covs[i]
        big[i * d:(i + 1) * d, i * d:(i + 1) * d] = c
        for j in range(i + 1, n):
            c = a[j - 1] @ c
            big[j * d:(j + 1) * d, i * d:(i + 1) * d] = c
            big[i * d:(i + 1) * d, j * d:(j + 1) * d] = c.T
    return np.concatenate(means), big


@pytest.mark.parametrize("d,t,bsz", [(1, 5, 2), (3, 6, 3), (6, 80, 2)])
def test_naturals_and_expectations_vs_dense_and_round_trips(rng, d, t, bsz):
    kw = random_ssm(rng, (bsz,), t, d, 1, well=True)
    ssm = mfa.StateSpaceModel(tt(kw["mu0"]), tt(kw["chol_p0"]), tt(kw["a_s"]), tt(kw["b_s"]), tt(kw["chol_q"]))
    th_lin, th_diag, th_sub = G.ssm_to_naturals(ssm)
    et_lin, et_diag, et_sub = G.ssm_to_expectations(ssm)
    if t <= 10:
        for s in range(bsz):
            mu, cov = dense_joint(kw, s)
            prec = np.linalg.inv(cov)
            blk = lambda m, i, j: m[i * d:(i + 1) * d, j * d:(j + 1) * d]  # noqa: E731
            np.testing.assert_allclose(nn(th_lin)[s].reshape(-1), prec @ mu, rtol=1e-8, atol=1e-9)
            second = cov + np.outer(mu, mu)
            for i in range(t):
                np.testing.assert_allclose(nn(th_diag)[s, i], -0.5 * blk(prec, i, i), rtol=1e-8, atol=1e-9)
                np.testing.assert_allclose(nn(et_diag)[s, i], blk(second, i, i), rtol=1e-8, atol=1e-9)
                if i + 1 < t:
                    np.testing.assert_allclose(nn(th_sub)[s, i], -blk(prec, i + 1, i), rtol=1e-8, atol=1e-9)
                    np.testing.assert_allclose(nn(et_sub)[s, i], blk(second, i + 1, i), rtol=1e-8, atol=1e-9)
            np.testing.assert_allclose(nn(et_lin)[s].reshape(-1), mu, rtol=1e-9, atol=1e-10)
    want = (kw["a_s"], kw["b_s"], kw["chol_p0"], kw["chol_q"], kw["mu0"])
    for got in (G.naturals_to_ssm_params(th_lin, th_diag, th_sub), G.expectations_to_ssm_params(et_lin, et_diag, et_sub)):
        for g, w in zip(got, want):
            np.testing.assert_allclose(nn(g), w, rtol=1e-6, atol=1e-8)
    ns = G.ssm_to_naturals_no_smoothing(ssm)
    for g, w in zip(G.naturals_to_ssm_params_no_smoothing(*ns), want):
        np.testing.assert_allclose(nn(g), w, rtol=1e-8, atol=1e-10)
    # the natural parameters rebuild the same distribution: precision blocks agree with StateSpaceModel.precision
    np.testing.assert_allclose(nn(-2 * th_diag), nn(ssm.precision.block_diagonal), rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(nn(-th_sub), nn(ssm.precision.block_sub_diagonal), rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("d,t", [(12, 40), (30, 1001)])
def test_round_trips_at_large_state_dim(rng, d, t):
    """The largest problem of the reference's suite (tests/unit/test_ssm_gaussian_transformations.py:40-46: d = 30, T = 1001) runs
    on the LDS-tile / MFMA operator kernels (d > 9, fp64)."""
    kw = dict(mu0=rng.normal(size=(d,)), chol_p0=np.tril(0.2 * rng.normal(size=(d, d))) / np.sqrt(d) + np.eye(d),
              a_s=0.6 * np.eye(d) + 0.3 * rng.normal(size=(t - 1, d, d)) / np.sqrt(d), b_s=0.3 * rng.normal(size=(t - 1, d)),
              chol_q=np.tril(0.2 * rng.normal(size=(t - 1, d, d))) / np.sqrt(d) + 0.7 * np.eye(d))
    for key in ("chol_p0", "chol_q"):
        idx = np.arange(d)
        kw[key][..., idx, idx] = np.abs(kw[key][..., idx, idx])
    ssm = mfa.StateSpaceModel(tt(kw["mu0"]), tt(kw["chol_p0"]), tt(kw["a_s"]), tt(kw["b_s"]), tt(kw["chol_q"]))
    want = (kw["a_s"], kw["b_s"], kw["chol_p0"], kw["chol_q"], kw["mu0"])
    for got in (G.naturals_to_ssm_params(*G.ssm_to_naturals(ssm)), G.expectations_to_ssm_params(*G.ssm_to_expectations(ssm))):
        for g, w in zip(got, want):
            np.testing.assert_allclose(nn(g), w, rtol=1e-6, atol=1e-8)
