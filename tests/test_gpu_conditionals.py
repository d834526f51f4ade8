"""
GPU parity tests of markovflow_amd/conditionals.py - the function surface of /root/reference/markovflow/conditionals.py
(conditional_predict :29-83, conditional_statistics :87-120, base_conditional_predict :380-421, pairwise_marginals :424-485):

* ``conditional_statistics`` (``mf_sde_conditional_statistics_*``: one lane per new point) against direct Gaussian conditioning on
  the kernel's dense covariance of the triple (x_-, x_t, x_+);
* ``pairwise_marginals -> conditional_predict`` against the dense GP predictive distribution and against the fused route
  ``ConditionalProcess.predict_state`` (``mf_sde_conditional_predict_*``) - the two compositions the reference has
  (posterior.py:207-229 calls exactly these functions);
* shapes and the conditional (no covariances) form.
fp64; tolerances as tests/test_gpu_kernels.py::test_posterior_predict_f_vs_dense_gp.
"""
import numpy as np
import pytest
import torch

import markovflow_amd as mfa
from markovflow_amd import conditionals as C
from oracle import numpy_kernels as K
from test_gpu_kalman import DEV, nn, tt

pytestmark = pytest.mark.gpu
CLS = {1: mfa.Matern12, 3: mfa.Matern32, 5: mfa.Matern52}


def _kernel(sig, ls, var):
    parts = [CLS[o](l, v, device=DEV) for o, l, v in zip(sig, ls, var)]
    return parts[0] if len(parts) == 1 else mfa.Sum(parts, jitter=1e-10)


@pytest.mark.parametrize("sig", [(3,), (5,), (5, 5), (1, 3, 5)])
def test_conditional_statistics_against_direct_gaussian_conditioning(rng, sig):
    """p(x_t | x_-, x_+) = N(P [x_-, x_+], T): from the stationary joint covariance of the three states, built from the kernel's own
    transitions, K = [[P, P A1^T, P A1^T A2^T], [., P, P A2^T], [., ., P]] (stationary: Cov(x_s, x_u) = A(u - s) P for u > s)."""
    bsz, n = 2, 20
    ls, var = [0.6 + 0.5 * j for j in range(len(sig))], [1.0 + 0.3 * j for j in range(len(sig))]
    kern = _kernel(sig, ls, var)
    d = kern.state_dim
    t = np.cumsum(0.2 + rng.exponential(0.3, size=(bsz, n)), axis=-1)
    gaps = np.diff(t, axis=-1)
    t_new = t[:, :-1] + gaps * (0.5 + 0.3 * (rng.random((bsz, n - 1)) - 0.5))        # one new point inside every gap
    n_new = n - 1
    proj, cov = C.conditional_statistics(tt(t_new), tt(t), kern)
    assert tuple(proj.shape) == (bsz, n_new, d, 2 * d) and tuple(cov.shape) == (bsz, n_new, d, d)
    minus, plus = t[:, :-1], t[:, 1:]
    a1, _ = kern.transition_statistics(tt(minus), tt(t_new - minus))
    a2, _ = kern.transition_statistics(tt(t_new), tt(plus - t_new))
    ps = nn(kern.steady_state_covariance).reshape(d, d)
    a1, a2 = nn(a1), nn(a2)
    for s in range(bsz):
        for i in range(n_new):
            k_mt = ps @ a1[s, i].T                      # Cov(x_-, x_t)
            k_tp = ps @ a2[s, i].T                      # Cov(x_t, x_+)
            k_mp = ps @ a1[s, i].T @ a2[s, i].T         # Cov(x_-, x_+)
            k_ends = np.block([[ps, k_mp], [k_mp.T, ps]])
            k_t_ends = np.hstack([k_mt.T, k_tp])
            want_p = k_t_ends @ np.linalg.inv(k_ends)
            want_t = ps - want_p @ k_t_ends.T
            # (the dense reference inverts the 2d x 2d covariance of the two ends, ill-conditioned for Matern-5/2 states)
            np.testing.assert_allclose(nn(proj)[s, i], want_p, rtol=2e-3, atol=2e-5)
            np.testing.assert_allclose(nn(cov)[s, i], want_t, rtol=2e-3, atol=1e-6)


@pytest.mark.parametrize("sig", [(3,), (5, 5), (1, 5)])
def test_pairwise_marginals_then_conditional_predict_against_the_dense_gp_and_the_fused_route(rng, sig):
    bsz, n, noise = 2, 50, 0.05
    ls, var = [0.6 + 0.5 * j for j in range(len(sig))], [1.0 + 0.3 * j for j in range(len(sig))]
    kern = _kernel(sig, ls, var)
    d = kern.state_dim
    t = np.cumsum(0.05 + rng.exponential(0.15, size=(bsz, n)), axis=-1)
    y = rng.normal(size=(bsz, n, 1))
    t_new = np.sort(np.concatenate([t[:, :1] - rng.random((bsz, 4)) * 2.0, t[:, -1:] + rng.random((bsz, 4)) * 2.0, t[:, 3:6],
                                    t[:, :1] + rng.random((bsz, 25)) * (t[:, -1:] - t[:, :1])], axis=-1), axis=-1)
    gpr = mfa.GaussianProcessRegression((tt(t), tt(y)), kern, chol_obs_covariance=tt(np.sqrt(noise) * np.eye(1)))
    post = gpr.posterior
    dist = post.gauss_markov_model
    m0 = kern.initial_mean((bsz,))
    p0 = kern.initial_covariance(tt(t[..., :1]))
    pair_mean, pair_cov = C.pairwise_marginals(dist, m0, p0)
    assert tuple(pair_mean.shape) == (bsz, n + 1, 2 * d) and tuple(pair_cov.shape) == (bsz, n + 1, 2 * d, 2 * d)
    np.testing.assert_allclose(nn(pair_cov), np.swapaxes(nn(pair_cov), -1, -2), atol=1e-12)
    mean, cov = C.conditional_predict(tt(t_new), tt(t), kern, pair_mean, pair_cov)
    fused_mean, fused_cov = post.predict_state(tt(t_new))
    np.testing.assert_allclose(nn(mean), nn(fused_mean), rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(nn(cov), nn(fused_cov), rtol=1e-8, atol=1e-10)
    f_mean, f_var = kern.generate_emission_model(tt(t_new)).project_state_marginals_to_f(mean, cov)
    for s in range(bsz):
        want_mean, want_var = K.dense_gp_predict(sig, ls, var, t[s], y[s, :, 0], noise, t_new[s])
        np.testing.assert_allclose(nn(f_mean)[s, :, 0], want_mean, rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(nn(f_var)[s, :, 0], want_var, rtol=1e-5, atol=1e-7)
    # without the pairwise covariances: the conditional density given [x_-, x_+] = the pairwise means
    cmean, ccov = C.conditional_predict(tt(t_new), tt(t), kern, pair_mean)
    proj, t_cov = C.conditional_statistics(tt(t_new), tt(t), kern)
    np.testing.assert_allclose(nn(cmean), nn(mean), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(nn(ccov), nn(t_cov), rtol=0, atol=0)
    # base_conditional_predict on hand-made inputs (conditionals.py:380-421)
    states = torch.randn(bsz, t_new.shape[-1], 2 * d, dtype=torch.float64, device=DEV)
    bm, bc = C.base_conditional_predict(proj, t_cov, states)
    np.testing.assert_allclose(nn(bm), nn((proj @ states[..., None])[..., 0]), rtol=1e-12)
    np.testing.assert_allclose(nn(bc), nn(t_cov), rtol=0, atol=0)


def test_pairwise_marginals_carries_gradients_to_the_chain(rng):
    """under a tape ``pairwise_marginals`` is built from the differentiable moments (the reference forms it from
    ``dist.marginals`` / ``dist.covariance_blocks()`` under its tape, conditionals.py:449-450; its sparse / PEP models differentiate
    through it).  Gradient of a scalar of both outputs w.r.t. every parameter of the chain against the dense torch composition
    ``P_{k+1} = A_k P_k A_k^T + Q_k``, ``m_{k+1} = A_k m_k + b_k``, ``Cov(x_{k+1}, x_k) = A_k P_k``; fp64, rtol 1e-8."""
    bsz, n, d = 2, 7, 3
    g = torch.Generator(device=DEV); g.manual_seed(11)
    rnd = lambda *s: torch.randn(*s, dtype=torch.float64, device=DEV, generator=g)                    # noqa: E731
    eye = torch.eye(d, dtype=torch.float64, device=DEV)
    leaves = [rnd(bsz, d), torch.tril(0.3 * rnd(bsz, d, d)) + eye, 0.5 * rnd(bsz, n - 1, d, d), 0.3 * rnd(bsz, n - 1, d),
              torch.tril(0.3 * rnd(bsz, n - 1, d, d)) + eye]
    m0, p0 = rnd(bsz, d), eye.expand(bsz, d, d) * 2.0
    wm, wc = rnd(bsz, n + 1, 2 * d), rnd(bsz, n + 1, 2 * d, 2 * d)

    def scalar(pm, pc):
        return torch.sum(pm * wm) + torch.sum(pc * wc)

    def ours(mu0, cp0, a, b, cq):
        return scalar(*C.pairwise_marginals(mfa.StateSpaceModel(mu0, cp0, a, b, cq), m0, p0))

    def dense(mu0, cp0, a, b, cq):
        means, covs, subs = [mu0], [cp0 @ cp0.transpose(-1, -2)], []
        for k in range(n - 1):
            subs.append(a[:, k] @ covs[-1])
            covs.append(a[:, k] @ covs[-1] @ a[:, k].transpose(-1, -2) + cq[:, k] @ cq[:, k].transpose(-1, -2))
            means.append((a[:, k] @ means[-1][..., None])[..., 0] + b[:, k])
        ext_m = torch.stack([m0] + means + [m0], dim=1)
        ext_c = torch.stack([p0] + covs + [p0], dim=1)
        zero = torch.zeros_like(p0)
        ext_s = torch.stack([zero] + subs + [zero], dim=1)
        pm = torch.cat([ext_m[:, :-1], ext_m[:, 1:]], dim=-1)
        pc = torch.cat([torch.cat([ext_c[:, :-1], ext_s.transpose(-1, -2)], dim=-1), torch.cat([ext_s, ext_c[:, 1:]], dim=-1)], dim=-2)
        return scalar(pm, pc)

    a_leaves = [t.clone().requires_grad_(True) for t in leaves]
    b_leaves = [t.clone().requires_grad_(True) for t in leaves]
    va, vb = ours(*a_leaves), dense(*b_leaves)
    assert float(va.detach()) == pytest.approx(float(vb.detach()), rel=1e-10)
    ga, gb = torch.autograd.grad(va, a_leaves), torch.autograd.grad(vb, b_leaves)
    for x, y, name in zip(ga, gb, ("mu0", "cholP0", "A", "b", "cholQ")):
        if name.startswith("chol"):
            x, y = torch.tril(x), torch.tril(y)
        np.testing.assert_allclose(nn(x), nn(y), rtol=1e-8, atol=1e-10, err_msg=name)
