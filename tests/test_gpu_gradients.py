"""
GPU tests of the gradients of KalmanFilter.log_likelihood.  The reference pins gradients through
tests/integration/models/test_gaussian_process_regression.py:117-130 (GPR gradients against GPflow's dense GP); here:
  * every tensor gradient (mu0, cholP0, A, b, cholQ, H, y, cholR) against torch autograd through a DENSE joint-Gaussian
    restatement of the same model (fp64, CPU) - an independent route to the same derivative;
  * GPR hyper-parameter gradients (lengthscale, variance per component, noise) against autograd through the dense GP
    marginal likelihood, the identity the reference tests.
Tolerance: rtol 1e-6 (Fisher's identity is exact; the smoother's rounding is what is left).
"""
import numpy as np
import pytest
import torch

import markovflow_amd as mfa
from test_gpu_kalman import random_ssm, tt

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dense_log_likelihood(mu0, cp0, a_s, b_s, cq, h, y, chol_r):
    """log N(y; H mu, H Sigma H^T + R) of ONE series from the dense joint of the chain - plain differentiable torch."""
    n, d = a_s.shape[0] + 1, mu0.shape[0]
    m = h.shape[1]
    means, covs = [mu0], [cp0 @ cp0.T]
    cross = {}
    for k in range(n - 1):
        q = cq[k] @ cq[k].T
        means.append(a_s[k] @ means[k] + b_s[k])
        covs.append(a_s[k] @ covs[k] @ a_s[k].T + q)
    big = torch.zeros(n * d, n * d, dtype=mu0.dtype)
    for i in range(n):
        big[i * d:(i + 1) * d, i * d:(i + 1) * d] = covs[i]
        c = covs[i]
        for j in range(i + 1, n):
            c = a_s[j - 1] @ c
            big[j * d:(j + 1) * d, i * d:(i + 1) * d] = c
            big[i * d:(i + 1) * d, j * d:(j + 1) * d] = c.T
    hm = torch.zeros(n * m, n * d, dtype=mu0.dtype)
    for i in range(n):
        hm[i * m:(i + 1) * m, i * d:(i + 1) * d] = h[i]
    mean_y = hm @ torch.cat(means)
    cov_y = hm @ big @ hm.T + torch.block_diag(*[chol_r @ chol_r.T] * n)
    res = y.reshape(-1) - mean_y
    return -0.5 * (res @ torch.linalg.solve(cov_y, res) + torch.linalg.slogdet(cov_y)[1] + n * m * np.log(2 * np.pi))


@pytest.mark.parametrize("dtype,d,m,t,bsz", [(torch.float32, 64, 32, 40, 2), (torch.float32, 33, 7, 70, 1), (torch.float64, 20, 3, 50, 2),
                                             (torch.float64, 12, 20, 9, 2),
                                             # 16 <= d <= 32 with up to four outputs: the wave kernel of the local step (csrc/mf_wave_grad.hpp)
                                             (torch.float64, 16, 1, 30, 3), (torch.float64, 32, 4, 12, 2), (torch.float64, 27, 2, 9, 70),
                                             (torch.float32, 16, 4, 40, 2), (torch.float32, 32, 1, 25, 2)])
def test_large_d_local_gradient_kernel_vs_closed_forms(rng, dtype, d, m, t, bsz):
    """The tile kernel of the local gradient step (csrc/mf_biggrad_impl.hpp; BASELINE config 5's d = 64, m = 32 in fp32) against
    the same closed forms as batched fp64 products on the SAME smoothed moments (kalman_filter._local_gradients_dense) - every
    gradient tensor, every time point; (12, 20): an observation dimension beyond the tiles takes the batched products themselves."""
    from markovflow_amd import kalman_filter as KF
    from test_gpu_kalman import build_kf
    kw = random_ssm(rng, (bsz,), t, d, m, well=True)
    cov_r = 0.4 * np.eye(m) + 0.05 * np.ones((m, m))
    kf0 = build_kf(kw, np.linalg.cholesky(cov_r), dtype=dtype)
    flat = [x.detach().clone().requires_grad_(True) for x in kf0.prior_ssm._flat_params()]
    h = kf0.emission.emission_matrix.detach().clone().requires_grad_(True)
    y = kf0.observations.detach().clone().requires_grad_(True)
    kf = mfa.KalmanFilter(mfa.StateSpaceModel(*flat), mfa.EmissionModel(h), y, kf0._chol_obs_covariance)
    w = torch.tensor(rng.normal(size=bsz) + 2.0, dtype=dtype, device=DEV)
    per_series = kf._per_series()[0]
    torch.sum(per_series * w).backward()
    with torch.no_grad():
        post = kf0.posterior_state_space_model()
        means, covs, cross = (x.double() for x in post._moments(want_sub=True))
        r_inv = torch.tensor(np.linalg.inv(cov_r), dtype=torch.float64, device=DEV)
        hh, yy = kf0._expanded()[0].double(), kf0._expanded()[1].double()
        want = KF._local_gradients_dense(*(x.detach().double() for x in kf0.prior_ssm._flat_params()), hh, yy, r_inv, means, covs,
                                         cross, w.double())
    tol = dict(rtol=1e-7, atol=1e-9) if dtype == torch.float64 else dict(rtol=2e-3, atol=2e-3)
    got = [x.grad for x in flat] + [h.grad, y.grad]
    for name, g, ww in zip(("mu0", "chol_p0", "a_s", "b_s", "chol_q", "h", "y"), got, want):
        scale = float(ww.abs().max())
        np.testing.assert_allclose(g.double().cpu().numpy().reshape(ww.shape) / scale, ww.cpu().numpy() / scale, err_msg=name, **tol)


@pytest.mark.parametrize("d,m,t,bsz", [(2, 1, 6, 3), (3, 2, 5, 2), (6, 1, 9, 2), (4, 3, 4, 1),
                                       # beyond the register-resident local kernel (VERDICT r02 missing 4): smoothed moments from the
                                       # LDS-tile / MFMA kernels, local closed forms as batched products (kalman_filter._local_gradients_dense)
                                       (12, 2, 6, 2), (17, 5, 5, 1), (32, 6, 4, 2),
                                       # 16 <= d <= 32, m <= 4: posterior chain, moments and the local step on the wave kernels
                                       (16, 1, 6, 2), (24, 3, 5, 2), (32, 4, 4, 1),
                                       # 10 <= d <= 15 on chains long enough for the time partition: posterior chain, moments and the
                                       # local step all in row form (csrc/mf_row_*.hpp compiled for these d)
                                       (12, 3, 70, 1), (15, 4, 66, 2), (10, 1, 130, 1),
                                       # more than four outputs at d <= 9: value and local step on the tile engine
                                       (3, 6, 12, 2), (9, 7, 20, 1)])
def test_tensor_gradients_vs_dense_autograd(rng, d, m, t, bsz):
    kw = random_ssm(rng, (bsz,), t, d, m, well=True)
    chol_r = np.linalg.cholesky(0.4 * np.eye(m) + 0.1 * np.ones((m, m)))
    names = ["mu0", "chol_p0", "a_s", "b_s", "chol_q", "h", "y"]
    cpu = {k: torch.tensor(kw[k], dtype=torch.float64, requires_grad=True) for k in names}
    cpu_r = torch.tensor(chol_r, dtype=torch.float64, requires_grad=True)
    total = sum(dense_log_likelihood(*(cpu[k][s] for k in names), cpu_r) for s in range(bsz))
    total.backward()
    gpu = {k: torch.tensor(kw[k], dtype=torch.float64, device=DEV, requires_grad=True) for k in names}
    gpu_r = torch.tensor(chol_r, dtype=torch.float64, device=DEV, requires_grad=True)
    ssm = mfa.StateSpaceModel(gpu["mu0"], gpu["chol_p0"], gpu["a_s"], gpu["b_s"], gpu["chol_q"])
    ll = mfa.KalmanFilter(ssm, mfa.EmissionModel(gpu["h"]), gpu["y"], gpu_r).log_likelihood()
    assert float(ll.detach()) == pytest.approx(float(total.detach()), rel=1e-10)
    ll.backward()
    for k in names:
        want = cpu[k].grad.numpy()
        if k in ("chol_p0", "chol_q"):
            want = np.tril(want)                # the factors are lower triangular by construction: only those entries vary
        np.testing.assert_allclose(gpu[k].grad.cpu().numpy(), want, rtol=1e-6, atol=1e-8, err_msg=k)
    np.testing.assert_allclose(gpu_r.grad.cpu().numpy(), np.tril(cpu_r.grad.numpy()), rtol=1e-6, atol=1e-8)


def test_gpr_hyperparameter_gradients_vs_dense_gp(rng):
    """d log p(y) / d (lengthscale, variance, noise std) for Sum(Matern52, Matern32) against autograd through the dense GP
    marginal likelihood (the check of test_gaussian_process_regression.py:117-130 of the reference)."""
    n, bsz = 40, 2
    t = np.cumsum(0.1 + rng.exponential(0.2, size=(bsz, n)), axis=-1)
    y = rng.normal(size=(bsz, n, 1))
    vals = dict(l5=0.8, v5=1.2, l3=1.4, v3=0.6, s=0.3)
    cpu = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in vals.items()}
    tt_, yt = torch.tensor(t), torch.tensor(y[..., 0])
    total = 0.0
    for s in range(bsz):
        r = (tt_[s][:, None] - tt_[s][None, :]).abs()
        l5, l3 = np.sqrt(5.0) / cpu["l5"], np.sqrt(3.0) / cpu["l3"]
        kmat = cpu["v5"] * (1 + l5 * r + (l5 * r) ** 2 / 3) * torch.exp(-l5 * r) + cpu["v3"] * (1 + l3 * r) * torch.exp(-l3 * r)
        kn = kmat + cpu["s"] ** 2 * torch.eye(n, dtype=torch.float64)
        total = total - 0.5 * (yt[s] @ torch.linalg.solve(kn, yt[s]) + torch.linalg.slogdet(kn)[1] + n * np.log(2 * np.pi))
    total.backward()
    gpu = {k: torch.tensor(v, dtype=torch.float64, device=DEV, requires_grad=True) for k, v in vals.items()}
    kern = mfa.Sum([mfa.Matern52(gpu["l5"], gpu["v5"]), mfa.Matern32(gpu["l3"], gpu["v3"])])
    gpr = mfa.GaussianProcessRegression((torch.tensor(t, device=DEV), torch.tensor(y, device=DEV)), kern,
                                        chol_obs_covariance=gpu["s"].reshape(1, 1))
    ll = gpr.log_likelihood()
    assert float(ll.detach()) == pytest.approx(float(total.detach()), rel=1e-9)
    ll.backward()
    for k in vals:
        assert float(gpu[k].grad) == pytest.approx(float(cpu[k].grad), rel=1e-6), k
    # without gradients requested the fused kernel runs and agrees
    with torch.no_grad():
        assert float(gpr.log_likelihood()) == pytest.approx(float(total.detach()), rel=1e-9)


def test_per_series_weights_reach_every_gradient(rng):
    """A loss that weights the series differently: the incoming gradient is applied inside mf_kf_loglik_grad."""
    d, m, t, bsz = 3, 1, 7, 3
    kw = random_ssm(rng, (bsz,), t, d, m, well=True)
    chol_r = np.array([[0.6]])
    weights = np.array([0.3, -1.7, 2.5])
    names = ["mu0", "chol_p0", "a_s", "b_s", "chol_q", "h", "y"]
    cpu = {k: torch.tensor(kw[k], dtype=torch.float64, requires_grad=True) for k in names}
    cpu_r = torch.tensor(chol_r, dtype=torch.float64, requires_grad=True)
    total = sum(float(weights[s]) * dense_log_likelihood(*(cpu[k][s] for k in names), cpu_r) for s in range(bsz))
    total.backward()
    gpu = {k: torch.tensor(kw[k], dtype=torch.float64, device=DEV, requires_grad=True) for k in names}
    gpu_r = torch.tensor(chol_r, dtype=torch.float64, device=DEV, requires_grad=True)
    ssm = mfa.StateSpaceModel(gpu["mu0"], gpu["chol_p0"], gpu["a_s"], gpu["b_s"], gpu["chol_q"])
    kf = mfa.KalmanFilter(ssm, mfa.EmissionModel(gpu["h"]), gpu["y"], gpu_r)
    per = kf._per_series()[0] + kf._constant_terms(t)
    loss = torch.sum(per * torch.tensor(weights, dtype=torch.float64, device=DEV))
    assert float(loss.detach()) == pytest.approx(float(total.detach()), rel=1e-10)
    loss.backward()
    for k in names:
        want = cpu[k].grad.numpy()
        if k in ("chol_p0", "chol_q"):
            want = np.tril(want)
        np.testing.assert_allclose(gpu[k].grad.cpu().numpy(), want, rtol=1e-6, atol=1e-8, err_msg=k)
    np.testing.assert_allclose(gpu_r.grad.cpu().numpy(), np.tril(cpu_r.grad.numpy()), rtol=1e-6, atol=1e-8)


# ------------------------------------------------------------------------------------------------------------------------------
# round 2: larger state dimensions / longer chains, the sites variants, kl_divergence, marginals, fail-loud everywhere else
# ------------------------------------------------------------------------------------------------------------------------------
NAMES = ["mu0", "chol_p0", "a_s", "b_s", "chol_q", "h", "y"]


def dense_chain(mu0, cp0, a_s, b_s, cq):
    """(stacked mean [n d], dense precision [n d, n d]) of ONE chain in O(n) differentiable torch ops
    (the block form of state_space_model.py:431-483)."""
    n, d = a_s.shape[0] + 1, mu0.shape[0]
    eye = torch.eye(d, dtype=mu0.dtype)
    qinv = [torch.cholesky_solve(eye, cp0)] + [torch.cholesky_solve(eye, cq[k]) for k in range(n - 1)]
    prec = torch.zeros(n * d, n * d, dtype=mu0.dtype)
    means = [mu0]
    for k in range(n):
        blk = qinv[k]
        if k < n - 1:
            j = qinv[k + 1] @ a_s[k]
            blk = blk + a_s[k].T @ j
            prec[(k + 1) * d:(k + 2) * d, k * d:(k + 1) * d] = -j
            prec[k * d:(k + 1) * d, (k + 1) * d:(k + 2) * d] = -j.T
            means.append(a_s[k] @ means[k] + b_s[k])
        prec[k * d:(k + 1) * d, k * d:(k + 1) * d] = blk
    return torch.cat(means), prec


def dense_log_likelihood_fast(mu0, cp0, a_s, b_s, cq, h, y, r_blocks):
    """log N(y; H mu, H Sigma H^T + blockdiag(R_k)); r_blocks [n, m, m] observation COVARIANCES per time point."""
    n, m = h.shape[0], h.shape[1]
    mean, prec = dense_chain(mu0, cp0, a_s, b_s, cq)
    sigma = torch.linalg.inv(prec)
    hm = torch.block_diag(*[h[i] for i in range(n)])
    cov_y = hm @ sigma @ hm.T + torch.block_diag(*[r_blocks[i] for i in range(n)])
    res = y.reshape(-1) - hm @ mean
    return -0.5 * (res @ torch.linalg.solve(cov_y, res) + torch.linalg.slogdet(cov_y)[1] + n * m * np.log(2 * np.pi))


def _leaves(kw, names, device=None):
    return {k: torch.tensor(kw[k], dtype=torch.float64, device=device, requires_grad=True) for k in names}


def _assert_grads(gpu, cpu, names, rtol=1e-6, atol=1e-8):
    for k in names:
        want = cpu[k].grad.numpy()
        if k in ("chol_p0", "chol_q"):
            want = np.tril(want)
        np.testing.assert_allclose(gpu[k].grad.cpu().numpy(), want, rtol=rtol, atol=atol, err_msg=k)


@pytest.mark.parametrize("d,m,t,bsz", [(7, 1, 65, 2), (8, 3, 130, 2), (9, 3, 65, 3), (9, 1, 130, 1), (9, 3, 400, 1), (6, 2, 400, 2)])
def test_tensor_gradients_state_dims_7_to_9_and_long_chains(rng, d, m, t, bsz):
    """d = 7, 8, 9 (where the gradient kernel is under register pressure), m = 1 and 3, chains that cross
    the serial / parallel-in-time threshold of the smoother (64 blocks)."""
    kw = random_ssm(rng, (bsz,), t, d, m, well=True)
    chol_r = np.linalg.cholesky(0.4 * np.eye(m) + 0.1 * np.ones((m, m)))
    cpu = _leaves(kw, NAMES)
    cpu_r = torch.tensor(chol_r, dtype=torch.float64, requires_grad=True)
    total = sum(dense_log_likelihood_fast(*(cpu[k][s] for k in NAMES), (cpu_r @ cpu_r.T).expand(t, m, m)) for s in range(bsz))
    total.backward()
    gpu = _leaves(kw, NAMES, DEV)
    gpu_r = torch.tensor(chol_r, dtype=torch.float64, device=DEV, requires_grad=True)
    ssm = mfa.StateSpaceModel(gpu["mu0"], gpu["chol_p0"], gpu["a_s"], gpu["b_s"], gpu["chol_q"])
    ll = mfa.KalmanFilter(ssm, mfa.EmissionModel(gpu["h"]), gpu["y"], gpu_r).log_likelihood()
    assert float(ll.detach()) == pytest.approx(float(total.detach()), rel=1e-9)
    ll.backward()
    _assert_grads(gpu, cpu, NAMES, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(gpu_r.grad.cpu().numpy(), np.tril(cpu_r.grad.numpy()), rtol=2e-6, atol=1e-7)


def test_gradients_across_the_lane_per_series_switch(rng):
    """B = 4100 >= 4096: the smoother behind the backward takes the one-lane-per-series kernels instead of the parallel-in-time
    ones.  Checked on the first, a middle and the last series against dense autograd, and as a directional derivative of
    the whole batch."""
    d, m, t, bsz = 3, 1, 70, 4100
    kw = random_ssm(rng, (bsz,), t, d, m, well=True)
    chol_r = np.array([[0.7]])
    gpu = _leaves(kw, NAMES, DEV)
    gpu_r = torch.tensor(chol_r, dtype=torch.float64, device=DEV, requires_grad=True)

    def value(tensors, r):
        ssm = mfa.StateSpaceModel(tensors["mu0"], tensors["chol_p0"], tensors["a_s"], tensors["b_s"], tensors["chol_q"])
        return mfa.KalmanFilter(ssm, mfa.EmissionModel(tensors["h"]), tensors["y"], r).log_likelihood()

    value(gpu, gpu_r).backward()
    for s in (0, 2077, bsz - 1):
        cpu = {k: torch.tensor(kw[k][s], dtype=torch.float64, requires_grad=True) for k in NAMES}
        cpu_r = torch.tensor(chol_r, dtype=torch.float64, requires_grad=True)
        dense_log_likelihood_fast(*(cpu[k] for k in NAMES), (cpu_r @ cpu_r.T).expand(t, m, m)).backward()
        for k in NAMES:
            want = cpu[k].grad.numpy()
            if k in ("chol_p0", "chol_q"):
                want = np.tril(want)
            np.testing.assert_allclose(gpu[k].grad[s].cpu().numpy(), want, rtol=1e-6, atol=1e-8, err_msg=f"{k}[{s}]")
    # directional derivative over the whole batch (central difference, fp64)
    gen = torch.Generator(device=DEV); gen.manual_seed(1)
    dirs = {k: torch.randn(gpu[k].shape, dtype=torch.float64, device=DEV, generator=gen) for k in NAMES}
    for k in ("chol_p0", "chol_q"):
        dirs[k] = torch.tril(dirs[k])
    slope = sum(float(torch.sum(gpu[k].grad * dirs[k])) for k in NAMES)
    eps = 1e-6
    with torch.no_grad():
        up = value({k: gpu[k].detach() + eps * dirs[k] for k in NAMES}, gpu_r.detach())
        dn = value({k: gpu[k].detach() - eps * dirs[k] for k in NAMES}, gpu_r.detach())
    assert slope == pytest.approx(float(up - dn) / (2 * eps), rel=2e-5)


@pytest.mark.parametrize("d,t", [(2, 7), (4, 30), (6, 90)])
def test_sites_filter_gradients_vs_dense_autograd(rng, d, t):
    """KalmanFilterWithSites.log_likelihood differentiated with respect to the site natural parameters and the chain
    (the CVI models do this: models/variational_cvi.py:138-161) against autograd through the dense Gaussian."""
    kw = random_ssm(rng, (), t, d, 1, well=True)
    nat2 = -0.5 * rng.uniform(0.5, 2.0, size=(t, 1, 1))
    nat1 = rng.normal(size=(t, 1))
    names = ["mu0", "chol_p0", "a_s", "b_s", "chol_q", "h"]
    cpu = _leaves(kw, names)
    c1, c2 = torch.tensor(nat1, requires_grad=True), torch.tensor(nat2, requires_grad=True)
    total = dense_log_likelihood_fast(*(cpu[k] for k in names), -0.5 * c1 / c2[..., 0], 1.0 / (-2.0 * c2))
    total.backward()
    gpu = _leaves(kw, names, DEV)
    g1 = torch.tensor(nat1, device=DEV, requires_grad=True)
    g2 = torch.tensor(nat2, device=DEV, requires_grad=True)
    ssm = mfa.StateSpaceModel(gpu["mu0"], gpu["chol_p0"], gpu["a_s"], gpu["b_s"], gpu["chol_q"])
    kf = mfa.KalmanFilterWithSites(ssm, mfa.EmissionModel(gpu["h"]), mfa.UnivariateGaussianSitesNat(g1, g2))
    ll = kf.log_likelihood()
    assert float(ll.detach()) == pytest.approx(float(total.detach()), rel=1e-9)
    ll.backward()
    _assert_grads(gpu, cpu, names)
    np.testing.assert_allclose(g1.grad.cpu().numpy(), c1.grad.numpy(), rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(g2.grad.cpu().numpy(), c2.grad.numpy(), rtol=1e-6, atol=1e-8)


def test_sparse_sites_filter_gradients_vs_dense_autograd(rng):
    """KalmanFilterWithSparseSites: sites live on a subset of the grid; gradient with respect to their natural parameters."""
    d, grid, idx = 3, 40, np.array([1, 4, 5, 17, 30, 39])
    kw = random_ssm(rng, (), grid, d, 1, well=True)
    nat2 = -0.5 * rng.uniform(0.5, 2.0, size=(len(idx), 1, 1))
    nat1 = rng.normal(size=(len(idx), 1))
    yobs = rng.normal(size=(len(idx), 1))
    names = ["mu0", "chol_p0", "a_s", "b_s", "chol_q", "h"]
    # dense reference: the observed points only
    cpu = _leaves(kw, names)
    c2 = torch.tensor(nat2, requires_grad=True)
    mean, prec = dense_chain(*(cpu[k] for k in names[:5]))
    sigma = torch.linalg.inv(prec)
    sel = torch.zeros(len(idx), grid * d, dtype=torch.float64)
    hsel = []
    for r, i in enumerate(idx):
        hsel.append(torch.cat([torch.zeros(i * d, dtype=torch.float64), cpu["h"][i, 0],
                               torch.zeros((grid - i - 1) * d, dtype=torch.float64)]))
    hm = torch.stack(hsel)
    cov_y = hm @ sigma @ hm.T + torch.diag(1.0 / (-2.0 * c2[:, 0, 0]))
    res = torch.tensor(yobs[:, 0]) - hm @ mean
    total = -0.5 * (res @ torch.linalg.solve(cov_y, res) + torch.linalg.slogdet(cov_y)[1] + len(idx) * np.log(2 * np.pi))
    total.backward()
    gpu = _leaves(kw, names, DEV)
    g2 = torch.tensor(nat2, device=DEV, requires_grad=True)
    ssm = mfa.StateSpaceModel(gpu["mu0"], gpu["chol_p0"], gpu["a_s"], gpu["b_s"], gpu["chol_q"])
    sites = mfa.UnivariateGaussianSitesNat(torch.tensor(nat1, device=DEV), g2)
    kf = mfa.KalmanFilterWithSparseSites(ssm, mfa.EmissionModel(gpu["h"]), sites, grid,
                                         torch.tensor(idx[:, None], device=DEV), torch.tensor(yobs, device=DEV))
    ll = kf.log_likelihood()
    assert float(ll.detach()) == pytest.approx(float(total.detach()), rel=1e-9)
    ll.backward()
    np.testing.assert_allclose(g2.grad.cpu().numpy(), c2.grad.numpy(), rtol=1e-6, atol=1e-8)
    for k in ("a_s", "chol_q", "mu0"):
        want = cpu[k].grad.numpy()
        np.testing.assert_allclose(gpu[k].grad.cpu().numpy(), np.tril(want) if k == "chol_q" else want, rtol=1e-6, atol=1e-8)


def dense_kl(p1, p2):
    """KL(N1 || N2) of two chains from their dense precisions (the closed form of state_space_model.py:528-593)."""
    m1, k1 = dense_chain(*p1)
    m2, k2 = dense_chain(*p2)
    s1 = torch.linalg.inv(k1)
    diff = m2 - m1
    return 0.5 * (torch.trace(k2 @ s1) + diff @ k2 @ diff - m1.shape[0] - torch.linalg.slogdet(k2)[1] + torch.linalg.slogdet(k1)[1])


CHAIN = ["mu0", "chol_p0", "a_s", "b_s", "chol_q"]


# chains of >= 64 blocks with few series take the scans in time for the adjoint sweep (d >= 7: the LDS variant of the scan)
@pytest.mark.parametrize("d,t,bsz", [(1, 2, 2), (2, 6, 3), (3, 40, 2), (6, 70, 2), (9, 33, 1), (8, 90, 1), (5, 203, 3), (1, 64, 2),
                                     # d > 9: values from the large-d kernels, backward by the scan in batched products
                                     (12, 9, 2), (20, 5, 1),
                                     # 10 <= d <= 15, long chains: the adjoint sweeps and the local step in row form
                                     (12, 80, 2), (15, 70, 1), (10, 130, 1)])
def test_kl_divergence_gradients_vs_dense_autograd(rng, d, t, bsz):
    """d KL(q1 || q2) / d (every parameter of q1 AND q2) against autograd through the dense Gaussian KL (the reference
    differentiates state_space_model.py:528-593 through TensorFlow)."""
    kw1 = random_ssm(rng, (bsz,), t, d, 1, well=True)
    kw2 = random_ssm(rng, (bsz,), t, d, 1, well=True)
    weights = rng.normal(size=bsz)
    c1, c2 = _leaves(kw1, CHAIN), _leaves(kw2, CHAIN)
    per = torch.stack([dense_kl([c1[k][s] for k in CHAIN], [c2[k][s] for k in CHAIN]) for s in range(bsz)])
    (per * torch.tensor(weights)).sum().backward()
    g1, g2 = _leaves(kw1, CHAIN, DEV), _leaves(kw2, CHAIN, DEV)
    kl = mfa.StateSpaceModel(*(g1[k] for k in CHAIN)).kl_divergence(mfa.StateSpaceModel(*(g2[k] for k in CHAIN)))
    np.testing.assert_allclose(kl.detach().cpu().numpy(), per.detach().numpy(), rtol=1e-9, atol=1e-10)
    (kl * torch.tensor(weights, device=DEV)).sum().backward()
    _assert_grads(g1, c1, CHAIN, rtol=1e-6, atol=1e-8)
    _assert_grads(g2, c2, CHAIN, rtol=1e-6, atol=1e-8)


def test_kl_gradient_vanishes_at_equal_chains(rng):
    kw = random_ssm(rng, (2,), 12, 3, 1, well=True)
    g1, g2 = _leaves(kw, CHAIN, DEV), _leaves(kw, CHAIN, DEV)
    kl = mfa.StateSpaceModel(*(g1[k] for k in CHAIN)).kl_divergence(mfa.StateSpaceModel(*(g2[k] for k in CHAIN)))
    assert float(kl.abs().max()) < 1e-10
    kl.sum().backward()
    for k in CHAIN:
        assert float(g1[k].grad.abs().max()) < 1e-9 and float(g2[k].grad.abs().max()) < 1e-9, k


@pytest.mark.parametrize("d,t,bsz", [(2, 5, 2), (4, 80, 2), (9, 20, 1), (9, 70, 1), (7, 131, 2), (1, 300, 1), (12, 33, 2), (24, 9, 1),
                                     (12, 100, 2), (15, 70, 1)])
def test_marginals_gradients_vs_recursion_autograd(rng, d, t, bsz):
    """A random linear functional of the marginal means and covariances, differentiated through `marginals`."""
    kw = random_ssm(rng, (bsz,), t, d, 1, well=True)
    wm, ws = rng.normal(size=(bsz, t, d)), rng.normal(size=(bsz, t, d, d))
    cpu = _leaves(kw, CHAIN)
    total = 0.0
    for s in range(bsz):
        mean, cov = cpu["mu0"][s], cpu["chol_p0"][s] @ cpu["chol_p0"][s].T
        for k in range(t):
            total = total + torch.sum(torch.tensor(wm[s, k]) * mean) + torch.sum(torch.tensor(ws[s, k]) * cov)
            if k < t - 1:
                a, c = cpu["a_s"][s, k], cpu["chol_q"][s, k]
                mean, cov = a @ mean + cpu["b_s"][s, k], a @ cov @ a.T + c @ c.T
    total.backward()
    gpu = _leaves(kw, CHAIN, DEV)
    means, covs = mfa.StateSpaceModel(*(gpu[k] for k in CHAIN)).marginals
    val = torch.sum(means * torch.tensor(wm, device=DEV)) + torch.sum(covs * torch.tensor(ws, device=DEV))
    assert float(val.detach()) == pytest.approx(float(total.detach()), rel=1e-10)
    val.backward()
    _assert_grads(gpu, cpu, CHAIN, rtol=1e-7, atol=1e-9)


def test_elbo_gradient_vanishes_at_the_posterior(rng):
    """The identity the reference pins in tests/integration/models/test_variational.py:123-132: with a Gaussian likelihood the
    ELBO  E_q[log p(y|x)] - KL(q || prior)  is maximised by the exact posterior, where it equals the log marginal likelihood
    and its gradient with respect to q's parameters is zero.  Exercises `marginals` and `kl_divergence` backward together."""
    d, t, bsz, noise = 3, 50, 2, 0.3
    kw = random_ssm(rng, (bsz,), t, d, 1, well=True)
    prior = mfa.StateSpaceModel(*(tt(kw[k]) for k in CHAIN))
    h, y = tt(kw["h"]), tt(kw["y"])
    kf = mfa.KalmanFilter(prior, mfa.EmissionModel(h), y, tt(np.array([[noise ** 0.5]])))
    q = kf.posterior_state_space_model().create_trainable_copy()

    def elbo(dist):
        means, covs = dist.marginals
        fm = torch.einsum("...kmd,...kd->...km", h, means)
        fv = torch.einsum("...kmd,...kde,...kme->...km", h, covs, h)
        ell = -0.5 * (np.log(2 * np.pi * noise) + ((y - fm) ** 2 + fv) / noise)
        return torch.sum(ell) - torch.sum(dist.kl_divergence(prior))

    value = elbo(q)
    assert float(value.detach()) == pytest.approx(float(kf.log_likelihood()), rel=1e-9)
    value.backward()
    leaves = q.trainable_variables
    assert len(leaves) == 5
    # scale: the same gradient away from the optimum
    q_off = mfa.StateSpaceModel(*(tt(kw[k]) for k in CHAIN)).create_trainable_copy()
    elbo(q_off).backward()
    for at_opt, off in zip(leaves, q_off.trainable_variables):
        assert float(at_opt.grad.abs().max()) < 1e-7 * max(1.0, float(off.grad.abs().max()))


def test_operations_without_a_backward_fail_loudly(rng):
    """nothing may return a tensor with a partial graph.  Everything that is not one of the differentiable
    entry points raises when an input requires a gradient, and works under no_grad / after detach."""
    kw = random_ssm(rng, (2,), 9, 3, 1, well=True)
    leaves = _leaves(kw, CHAIN, DEV)
    ssm = mfa.StateSpaceModel(*(leaves[k] for k in CHAIN))
    kf = mfa.KalmanFilter(ssm, mfa.EmissionModel(tt(kw["h"])), tt(kw["y"]), tt(np.array([[0.5]])))
    for op in (lambda: kf._k_inv_post, lambda: ssm._covariance_scan(want_sub=True), lambda: ssm._moments(want_sub=True)):
        with pytest.raises(NotImplementedError, match="not differentiable"):
            op()
    # round 4: the operator level is differentiable (tests/test_gpu_autograd_ops.py) - precision and the reparameterised sample;
    # round 5: the posterior chain and the covariance blocks too (test_posterior_state_space_model_gradients_vs_dense_autograd)
    assert ssm.precision.block_diagonal.requires_grad and ssm.sample(2).requires_grad and ssm.normalizer().requires_grad
    assert kf.posterior_state_space_model().state_transitions.requires_grad and ssm.covariance_blocks()[1].requires_grad
    with torch.no_grad():
        assert torch.isfinite(kf.posterior_state_space_model().marginal_means).all()
    assert torch.isfinite(ssm.create_non_trainable_copy().precision.cholesky.block_diagonal).all()


@pytest.mark.parametrize("d,t,bsz", [(1, 1 + 1, 3), (3, 100, 4), (6, 257, 2), (9, 70, 2), (7, 130, 1), (2, 64, 5)])
def test_kl_routes_agree_with_the_oracle(rng, d, t, bsz):
    """kl_divergence has three routes: one sweep per series (many series / short chains), the marginals of q1 by the scans in
    time + one lane per (series, step) (few long chains: t >= 64 here), and the reference's route over the operator kernels
    (other distributions, d > 9).  All against the numpy restatement of state_space_model.py:528-593; the moments handed to the
    backward are the marginals of q1 on either route."""
    from oracle import numpy_oracle as O
    from markovflow_amd import _lib
    kw1 = random_ssm(rng, (bsz,), t, d, 1, well=True)
    kw2 = random_ssm(rng, (bsz,), t, d, 1, well=True)
    ref = O.ssm_kl_divergence(tuple(kw1[k] for k in CHAIN), tuple(kw2[k] for k in CHAIN))
    q1 = mfa.StateSpaceModel(*(tt(kw1[k]) for k in CHAIN))
    q2 = mfa.StateSpaceModel(*(tt(kw2[k]) for k in CHAIN))
    value, moments, adjoint_inputs = q1._kl_divergence_value(q2, keep_moments=True)
    # the inputs of the backward's recursion, by-products of either forward route: N_k = dA^T Q2^-1 dA, n_k = dA^T Q2^-1 eps_k
    means_ref = q1._moments(want_sub=False)[0].cpu().numpy()
    for s_ in range(bsz):
        for k in (0, t // 2, t - 2, t - 1):
            if k < 0:
                continue
            if k == t - 1:
                want_n_mat, want_n_vec = np.zeros((d, d)), np.zeros(d)
            else:
                da = kw1["a_s"][s_, k] - kw2["a_s"][s_, k]
                q2i = np.linalg.inv(kw2["chol_q"][s_, k] @ kw2["chol_q"][s_, k].T)
                eps = da @ means_ref[s_, k] + kw1["b_s"][s_, k] - kw2["b_s"][s_, k]
                want_n_mat, want_n_vec = da.T @ q2i @ da, da.T @ q2i @ eps
            np.testing.assert_allclose(adjoint_inputs[0][s_, k].cpu().numpy(), want_n_mat, rtol=1e-8, atol=1e-10)
            np.testing.assert_allclose(adjoint_inputs[1][s_, k].cpu().numpy(), want_n_vec, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(value.cpu().numpy(), ref, rtol=1e-9)
    ws_bytes = int(_lib.load().mf_ssm_kl_workspace_bytes(bsz, t, d, 8))
    assert (ws_bytes > 0) == (t >= 64) and moments is not None      # both routes hand q1's moments to the backward
    if moments is not None:
        means, covs, cross = q1._moments(want_sub=True)
        for got, want in zip(moments, (means, covs, cross)):
            np.testing.assert_allclose(got.cpu().numpy(), want.reshape(got.shape).cpu().numpy(), rtol=1e-11, atol=1e-13)
    # the sweep per series, whatever the shape: no workspace
    out = torch.empty(bsz, dtype=torch.float64, device=DEV)
    info = _lib.pivot_info(out.device)
    _lib.call("mf_ssm_kl_divergence", out.dtype, bsz, t, d, *[_lib.ptr(x) for x in q1._flat_params()],
              *[_lib.ptr(x) for x in q2._flat_params()], _lib.ptr(out), None, None, None, None, None, None, 0, info,
              _lib.stream_ptr(out.device))
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-9)
    np.testing.assert_allclose(q1._kl_divergence_operators(q2).cpu().numpy(), ref, rtol=1e-9)


def test_trainable_copy_follows_optimiser_steps(rng):
    """The reference's trainable copy re-evaluates its bijectors on every access (state_space_model.py:396-429): after an
    optimiser step the chain must see the NEW parameters, and its Cholesky factors must stay lower triangular."""
    kw = random_ssm(rng, (3,), 20, 4, 1, well=True)
    prior = mfa.StateSpaceModel(*(tt(kw[k]) for k in CHAIN))
    kw2 = random_ssm(rng, (3,), 20, 4, 1, well=True)
    q = mfa.StateSpaceModel(*(tt(kw2[k]) for k in CHAIN)).create_trainable_copy()
    opt = torch.optim.SGD(q.trainable_variables, lr=1e-2)
    values = []
    for _ in range(3):
        opt.zero_grad()
        kl = q.kl_divergence(prior).sum()
        kl.backward()
        values.append(float(kl.detach()))
        opt.step()
    assert values[0] > values[1] > values[2]                       # gradient steps on the leaves reach the next evaluation
    leaves = q.trainable_variables
    assert torch.equal(q.cholesky_process_covariances.detach(), leaves[4].detach())
    assert float(torch.triu(leaves[4].detach(), diagonal=1).abs().max()) == 0.0
    assert float(torch.triu(leaves[1].detach(), diagonal=1).abs().max()) == 0.0
    fresh = mfa.StateSpaceModel(*(t.detach() for t in leaves)).kl_divergence(prior).sum()
    assert float(q.kl_divergence(prior).sum().detach()) == pytest.approx(float(fresh), rel=1e-12)


# ---- posterior_state_space_model / covariance_blocks under a gradient (VERDICT r04 missing 4) --------------------------------------
def dense_posterior_moments(mu0, cp0, a_s, b_s, cq, h, y, chol_r):
    """(means [n, d], covariance blocks [n, d, d], Cov(x_{k+1}, x_k) [n-1, d, d]) of x | y for ONE series by dense Gaussian
    conditioning in differentiable torch (kalman_filter.py:109-182 states the same posterior as a chain)."""
    n, d = a_s.shape[0] + 1, mu0.shape[0]
    mean, prec = dense_chain(mu0, cp0, a_s, b_s, cq)
    hm = torch.block_diag(*[h[i] for i in range(n)])
    # chol_r: the Cholesky factor of the shared observation covariance [m, m], or per-point PRECISIONS [n, m, m] (sites)
    r_inv = torch.block_diag(*([torch.cholesky_inverse(chol_r)] * n if chol_r.dim() == 2 else [chol_r[i] for i in range(n)]))
    cov = torch.linalg.inv(prec + hm.T @ r_inv @ hm)
    cov = 0.5 * (cov + cov.T)
    m_post = cov @ (hm.T @ r_inv @ y.reshape(-1) + prec @ mean)
    blocks = torch.stack([cov[k * d:(k + 1) * d, k * d:(k + 1) * d] for k in range(n)])
    cross = torch.stack([cov[(k + 1) * d:(k + 2) * d, k * d:(k + 1) * d] for k in range(n - 1)]) if n > 1 else cov.new_zeros(0, d, d)
    return m_post.reshape(n, d), blocks, cross


@pytest.mark.parametrize("d,m,t,bsz", [(2, 1, 6, 2), (3, 2, 9, 1), (9, 3, 5, 2), (4, 1, 70, 1), (12, 2, 5, 1), (3, 1, 2, 2)])
def test_posterior_state_space_model_gradients_vs_dense_autograd(rng, d, m, t, bsz):
    """The reference differentiates THROUGH posterior_state_space_model (kalman_filter.py:109-182 under a tape; conditionals.py:453
    reads the posterior chain's covariance blocks): weighted sums of the posterior chain's marginal means, covariance blocks and
    subsequent covariances, value and gradient with respect to every tensor of the model, against dense conditioning."""
    kw = random_ssm(rng, (bsz,), t, d, m, well=True)
    chol_r = np.linalg.cholesky(0.4 * np.eye(m) + 0.1 * np.ones((m, m)))
    names = ["mu0", "chol_p0", "a_s", "b_s", "chol_q", "h", "y"]
    w_m, w_c, w_x = rng.normal(size=(bsz, t, d)), rng.normal(size=(bsz, t, d, d)), rng.normal(size=(bsz, max(t - 1, 0), d, d))
    cpu = {k: torch.tensor(kw[k], dtype=torch.float64, requires_grad=True) for k in names}
    cpu_r = torch.tensor(chol_r, dtype=torch.float64, requires_grad=True)
    total = 0.0
    for s in range(bsz):
        mm, cc, xx = dense_posterior_moments(*(cpu[k][s] for k in names), cpu_r)
        total = total + (torch.tensor(w_m[s]) * mm).sum() + (torch.tensor(w_c[s]) * cc).sum() + (torch.tensor(w_x[s]) * xx).sum()
    total.backward()
    gpu = {k: torch.tensor(kw[k], dtype=torch.float64, device=DEV, requires_grad=True) for k in names}
    gpu_r = torch.tensor(chol_r, dtype=torch.float64, device=DEV, requires_grad=True)
    ssm = mfa.StateSpaceModel(gpu["mu0"], gpu["chol_p0"], gpu["a_s"], gpu["b_s"], gpu["chol_q"])
    post = mfa.KalmanFilter(ssm, mfa.EmissionModel(gpu["h"]), gpu["y"], gpu_r).posterior_state_space_model()
    means = post.marginal_means
    covs, cross = post.covariance_blocks()
    got = (tt(w_m) * means).sum() + (tt(w_c) * covs).sum() + (tt(w_x) * cross).sum()
    assert float(got.detach()) == pytest.approx(float(total.detach()), rel=1e-8, abs=1e-9)
    got.backward()
    for k in names:
        want = cpu[k].grad.numpy()
        if k in ("chol_p0", "chol_q"):
            want = np.tril(want)
        np.testing.assert_allclose(gpu[k].grad.cpu().numpy(), want, rtol=2e-6, atol=1e-7, err_msg=k)
    np.testing.assert_allclose(gpu_r.grad.cpu().numpy(), np.tril(cpu_r.grad.numpy()), rtol=2e-6, atol=1e-7)
    # the value without a gradient is the kernels' chain: both routes describe the same model
    with torch.no_grad():
        plain = mfa.KalmanFilter(ssm, mfa.EmissionModel(gpu["h"]), gpu["y"], gpu_r).posterior_state_space_model()
    for a, b in zip(post._flat_params(), plain._flat_params()):
        torch.testing.assert_close(a.detach(), b, rtol=1e-7, atol=1e-9)


def test_upper_diagonal_lower_gradients_vs_finite_differences(rng):
    """block_tri_diag.py:438-545 under a tape: U^T and chol_D as differentiable functions of the blocks (the Cholesky factor of the
    time-reversed matrix), against central differences of the kernel's own factorisation."""
    from test_gpu_large_d_ops import scaled_spd_btd
    diag, sub = scaled_spd_btd(rng, (2,), 7, 3, True)
    dg, sb = tt(diag).requires_grad_(True), tt(sub).requires_grad_(True)
    w_u, w_c = tt(rng.normal(size=sub.shape)), tt(rng.normal(size=diag.shape))

    def value(dv, sv):
        u_t, chol_d = mfa.SymmetricBlockTriDiagonal(dv, sv).upper_diagonal_lower()
        return (w_u * u_t.block_sub_diagonal).sum() + (w_c * chol_d.block_diagonal).sum()

    value(dg, sb).backward()
    with torch.no_grad():
        eps = 1e-6
        for leaf, grad in ((dg, dg.grad), (sb, sb.grad)):
            for _ in range(6):
                idx = tuple(int(rng.integers(0, s)) for s in leaf.shape)
                pert = torch.zeros_like(leaf)
                pert[idx] = eps
                if leaf is dg:                        # a symmetric perturbation of a symmetric block: d/dD_ij + d/dD_ji
                    pert[idx[:-2] + (idx[-1], idx[-2])] = eps
                    want = grad[idx] + (grad[idx[:-2] + (idx[-1], idx[-2])] if idx[-1] != idx[-2] else 0.0)
                    up, dn = value(dg + pert, sb), value(dg - pert, sb)
                else:
                    want = grad[idx]
                    up, dn = value(dg, sb + pert), value(dg, sb - pert)
                assert float((up - dn) / (2 * eps)) == pytest.approx(float(want), rel=2e-5, abs=1e-6)


def test_sites_filter_posterior_gradients_vs_dense_autograd(rng):
    """KalmanFilterWithSites.posterior_state_space_model under a tape (kalman_filter.py:437-497 with :109-182): per-point precisions
    and pseudo-observations that are functions of the sites' natural parameters; gradients with respect to nat1, nat2 and the chain."""
    d, t = 3, 9
    kw = random_ssm(rng, (), t, d, 1, well=True)
    nat1_np, nat2_np = rng.normal(size=(t, 1)), -0.5 * (0.5 + rng.random(size=(t, 1, 1)))
    w_m, w_c = rng.normal(size=(t, d)), rng.normal(size=(t, d, d))

    def value(dev):
        leaves = {k: torch.tensor(kw[k], dtype=torch.float64, device=dev, requires_grad=True) for k in CHAIN}
        nat1 = torch.tensor(nat1_np, dtype=torch.float64, device=dev, requires_grad=True)
        nat2 = torch.tensor(nat2_np, dtype=torch.float64, device=dev, requires_grad=True)
        h = torch.tensor(kw["h"], dtype=torch.float64, device=dev)
        wm, wc = torch.tensor(w_m, dtype=torch.float64, device=dev), torch.tensor(w_c, dtype=torch.float64, device=dev)
        if dev == "cpu":
            means, covs, _ = dense_posterior_moments(*(leaves[k] for k in CHAIN), h, -0.5 * nat1 / nat2[..., 0], -2.0 * nat2)
        else:
            sites = mfa.UnivariateGaussianSitesNat(nat1, nat2)
            post = mfa.KalmanFilterWithSites(mfa.StateSpaceModel(*(leaves[k] for k in CHAIN)), mfa.EmissionModel(h), sites).posterior_state_space_model()
            means, covs = post.marginals
        total = (wm * means).sum() + (wc * covs).sum()
        total.backward()
        return total, leaves, nat1, nat2

    want, cl, c1, c2 = value("cpu")
    got, gl, g1, g2 = value(DEV)
    assert float(got.detach()) == pytest.approx(float(want.detach()), rel=1e-9, abs=1e-10)
    _assert_grads(gl, cl, CHAIN, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(g1.grad.cpu().numpy(), c1.grad.numpy(), rtol=2e-6, atol=1e-8)
    np.testing.assert_allclose(g2.grad.cpu().numpy(), c2.grad.numpy(), rtol=2e-6, atol=1e-8)
