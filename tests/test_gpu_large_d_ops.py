"""
GPU parity tests of the operator entry points at LARGE state dimension (10 <= d <= 64 fp32, <= 32 fp64), which run on the
LDS-tile / MFMA engine (csrc/mf_bigops_impl.hpp): the same identities as tests/test_gpu_block_tri_diag.py and
tests/test_gpu_kalman.py (which re-express /root/reference/tests/unit/test_block_tri_diag.py:79-225,
tests/unit/test_state_space_model.py and tests/integration/test_kalman_filter.py:105-139), against the numpy oracle.
Tolerances: fp64 rtol 1e-8; fp32 rtol 3e-3, atol 3e-4 on inputs whose conditioning does not grow with d.
"""
import numpy as np
import pytest
import torch

import markovflow_amd as mfa
from oracle import numpy_oracle as O
from test_gpu_block_tri_diag import nn, tt
from test_gpu_kalman import build_kf, random_ssm

pytestmark = pytest.mark.gpu


def scaled_spd_btd(rng, batch, n, d, has_sub):
    """SPD block tridiagonal from a random lower factor whose off-diagonal entries shrink like 1/sqrt(d), so that the
    conditioning does not grow with the state dimension (an unscaled 64 x 64 coupling block has norm ~5 against unit
    pivots, which fp32 does not survive - a property of the input, not of the kernels)."""
    sc = 0.3 / np.sqrt(d)
    ldiag = np.tril(sc * rng.normal(size=batch + (n, d, d)))
    idx = np.arange(d)
    ldiag[..., idx, idx] = 1.0 + np.abs(rng.normal(size=batch + (n, d)))
    diag = ldiag @ np.swapaxes(ldiag, -1, -2)
    sub = None
    if has_sub:
        lsub = sc * rng.normal(size=batch + (n - 1, d, d))
        diag[..., 1:, :, :] += lsub @ np.swapaxes(lsub, -1, -2)
        sub = lsub @ np.swapaxes(ldiag[..., :-1, :, :], -1, -2)
    return diag, sub
TOL = {torch.float64: dict(rtol=1e-8, atol=1e-10), torch.float32: dict(rtol=3e-3, atol=3e-4)}
CASES = [(torch.float64, 10), (torch.float64, 16), (torch.float64, 19), (torch.float64, 32),
         (torch.float32, 12), (torch.float32, 33), (torch.float32, 48), (torch.float32, 64)]


@pytest.mark.parametrize("dtype,d", CASES)
@pytest.mark.parametrize("batch,n,has_sub", [((2,), 5, True), ((), 1, False), ((1,), 3, False), ((2, 1), 9, True)])
def test_large_d_cholesky_solve_logdet_mult_inverse(rng, dtype, d, batch, n, has_sub):
    diag, sub = scaled_spd_btd(rng, batch, n, d, has_sub)
    if dtype == torch.float32:
        diag = diag.astype(np.float32).astype(np.float64)
        sub = None if sub is None else sub.astype(np.float32).astype(np.float64)
    rhs = rng.normal(size=batch + (n, d))
    tol = TOL[dtype]
    sym = mfa.SymmetricBlockTriDiagonal(tt(diag, dtype), tt(sub, dtype))
    chol = sym.cholesky
    ld, ls = O.btd_cholesky(diag, sub)
    np.testing.assert_allclose(nn(chol.block_diagonal), np.tril(ld), **tol)
    if has_sub:
        np.testing.assert_allclose(nn(chol.block_sub_diagonal), ls, **tol)
    np.testing.assert_allclose(nn(chol.abs_log_det()), O.btd_abs_log_det(ld), **tol)
    exact = mfa.LowerTriangularBlockTriDiagonal(tt(np.tril(ld), dtype), tt(ls, dtype))
    r = tt(rhs, dtype)
    np.testing.assert_allclose(nn(exact.solve(r)), O.btd_solve(ld, ls, rhs), **tol)
    np.testing.assert_allclose(nn(exact.solve(r, transpose_left=True)), O.btd_solve(ld, ls, rhs, transpose_left=True), **tol)
    np.testing.assert_allclose(nn(sym.dense_mult(r)), O.btd_dense_mult(diag, sub, rhs, symmetric=True), **tol)
    np.testing.assert_allclose(nn(exact.dense_mult(r)), O.btd_dense_mult(ld, ls, rhs, symmetric=False), **tol)
    np.testing.assert_allclose(nn(exact.dense_mult(r, transpose_left=True)),
                               O.btd_dense_mult(ld, ls, rhs, symmetric=False, transpose_left=True), **tol)
    inv_d, inv_s = O.btd_block_diagonal_of_inverse(ld, ls, return_sub=True)
    got_d, got_s = exact._diag_and_sub_of_inverse(want_sub=True)
    np.testing.assert_allclose(nn(got_d), inv_d, **tol)
    if has_sub:
        np.testing.assert_allclose(nn(got_s), inv_s, **tol)
        u_t, chol_d = sym.upper_diagonal_lower()
        want_u, want_c = O.btd_upper_diagonal_lower(diag, sub)
        np.testing.assert_allclose(nn(u_t.block_sub_diagonal), want_u, **tol)
        np.testing.assert_allclose(nn(chol_d.block_diagonal), np.tril(want_c), **tol)


@pytest.mark.parametrize("dtype,d,m", [(torch.float64, 14, 1), (torch.float64, 32, 3), (torch.float32, 20, 2),
                                       (torch.float32, 64, 32)])
def test_large_d_posterior_marginals_and_kl(rng, dtype, d, m):
    """posterior_state_space_model, marginals and KL for a state dimension the register-resident kernels do not cover."""
    t = 7
    kw = random_ssm(rng, (2,), t, d, m, well=True)
    if dtype == torch.float32:
        kw = {k: v.astype(np.float32).astype(np.float64) for k, v in kw.items()}
    cov = 0.4 * np.eye(m)
    kf = build_kf(kw, np.linalg.cholesky(cov), dtype=dtype)
    tol = dict(rtol=1e-7, atol=1e-9) if dtype == torch.float64 else dict(rtol=5e-3, atol=5e-4)
    post = kf.posterior_state_space_model()
    want = O.kf_posterior_ssm(**kw, r_inv=np.linalg.inv(cov))
    got = (post.initial_mean, post.cholesky_initial_covariance, post.state_transitions, post.state_offsets,
           post.cholesky_process_covariances)
    for g, w in zip(got, want):
        np.testing.assert_allclose(nn(g), w, **tol)
    np.testing.assert_allclose(nn(post.marginal_means), O.ssm_marginal_means(want[0], want[2], want[3]), **tol)
    np.testing.assert_allclose(nn(post.marginal_covariances), O.ssm_marginal_covariances(want[1], want[2], want[4]), **tol)
    prior = (kw["mu0"], kw["chol_p0"], kw["a_s"], kw["b_s"], kw["chol_q"])
    kl = O.ssm_kl_divergence(want, prior)
    np.testing.assert_allclose(nn(post.kl_divergence(kf.prior_ssm)), kl, rtol=1e-6 if dtype == torch.float64 else 2e-2)
    # the log-likelihood of the same model agrees with the operator route: 0.5 |L^-1 eta|^2 - log|L| through cholesky/solve
    ref = O.kf_log_likelihood(**kw, r_inv=np.linalg.inv(cov))
    np.testing.assert_allclose(float(kf.log_likelihood().cpu()), ref, rtol=1e-9 if dtype == torch.float64 else 3e-4)


@pytest.mark.parametrize("dtype,d", [(torch.float64, 10), (torch.float64, 19), (torch.float64, 32), (torch.float32, 12),
                                     (torch.float32, 33), (torch.float32, 64)])
@pytest.mark.parametrize("bsz,t", [(2, 3), (3, 40), (1, 700), (5, 129)])
def test_large_d_marginal_covariances_partitioned_in_time(rng, dtype, d, bsz, t):
    """state_space_model.py:254-275,326-341 for d > 9: the forward recursion on the LDS-tile / MFMA engine.  Chains long enough
    for more than one chunk per series (t = 129, 700) take the three-pass partition, the others one workgroup per series; both
    against the explicit recursion, and the reference's route (precision -> cholesky -> block_diagonal_of_inverse) as a check
    of the identity itself in fp64."""
    kw = random_ssm(rng, (bsz,), t, d, 1, well=True)
    if dtype == torch.float32:
        kw = {k: v.astype(np.float32).astype(np.float64) for k, v in kw.items()}
    ssm = mfa.StateSpaceModel(*(tt(kw[k], dtype) for k in ("mu0", "chol_p0", "a_s", "b_s", "chol_q")))
    covs, sub = ssm.covariance_blocks()
    ec, es, em = [], [], []
    for s_ in range(bsz):
        mean, cov = kw["mu0"][s_], kw["chol_p0"][s_] @ kw["chol_p0"][s_].T
        cs, ss, ms = [cov], [], [mean]
        for k in range(t - 1):
            a, c = kw["a_s"][s_, k], kw["chol_q"][s_, k]
            ss.append(a @ cov)
            mean, cov = a @ mean + kw["b_s"][s_, k], a @ cov @ a.T + c @ c.T
            cs.append(cov); ms.append(mean)
        ec.append(np.stack(cs)); es.append(np.stack(ss)); em.append(np.stack(ms))
    tol = TOL[dtype]
    np.testing.assert_allclose(nn(covs), np.stack(ec), **tol)
    np.testing.assert_allclose(nn(sub), np.stack(es), **tol)
    np.testing.assert_allclose(nn(ssm.marginal_covariances), np.stack(ec), **tol)
    # `marginals`: the means ride along the same three passes
    means, covs2, sub2 = ssm._moments(want_sub=True)
    np.testing.assert_allclose(nn(means), np.stack(em), **tol)
    np.testing.assert_allclose(nn(covs2), np.stack(ec), **tol)
    np.testing.assert_allclose(nn(sub2), np.stack(es), **tol)
    np.testing.assert_allclose(nn(ssm.marginal_means), np.stack(em), **tol)
    m2, c2 = ssm.marginals
    np.testing.assert_allclose(nn(m2), np.stack(em), **tol)
    if dtype == torch.float64 and t <= 40:
        ref = ssm.precision.cholesky.block_diagonal_of_inverse()
        np.testing.assert_allclose(nn(ref), np.stack(ec), rtol=1e-7, atol=1e-9)


# ---- time-partitioned factorisations (csrc/mf_bigpar_impl.hpp): chains long enough for >= 4 chunks per series ----------------------
PAR_CASES = [(torch.float64, 10), (torch.float64, 19), (torch.float64, 32), (torch.float32, 12), (torch.float32, 33),
             (torch.float32, 48), (torch.float32, 64)]


@pytest.mark.parametrize("dtype,d", PAR_CASES)
@pytest.mark.parametrize("bsz,n", [(1, 67), (3, 300), (2, 32)])
def test_large_d_cholesky_and_udl_partitioned_in_time(rng, dtype, d, bsz, n):
    """block_tri_diag.py:423-436 / :438-545 on chains cut into chunks (up-sweep with a spike, chunk-end pivots, emit): the NATURAL
    ORDER factors, block by block, against the oracle's serial recursions; ragged last chunks (67 = 8 x 9 - 5, 300 = 37 x 9 - 33)."""
    diag, sub = scaled_spd_btd(rng, (bsz,), n, d, True)
    if dtype == torch.float32:
        diag, sub = diag.astype(np.float32).astype(np.float64), sub.astype(np.float32).astype(np.float64)
    tol = TOL[dtype]
    sym = mfa.SymmetricBlockTriDiagonal(tt(diag, dtype), tt(sub, dtype))
    chol = sym.cholesky
    ld, ls = O.btd_cholesky(diag, sub)
    np.testing.assert_allclose(nn(chol.block_diagonal), np.tril(ld), **tol)
    np.testing.assert_allclose(nn(chol.block_sub_diagonal), ls, **tol)
    u_t, chol_d = sym.upper_diagonal_lower()
    want_u, want_c = O.btd_upper_diagonal_lower(diag, sub)
    np.testing.assert_allclose(nn(u_t.block_sub_diagonal), want_u, **tol)
    np.testing.assert_allclose(nn(chol_d.block_diagonal), np.tril(want_c), **tol)
    # solve (both orientations; a second right-hand side per factor: the broadcast of block_tri_diag.py:339-351) and the block
    # diagonal / sub-diagonal of the inverse, on the exact factor
    exact = mfa.LowerTriangularBlockTriDiagonal(tt(np.tril(ld), dtype), tt(ls, dtype))
    rhs = rng.normal(size=(2, bsz, n, d))
    r = tt(rhs, dtype)
    np.testing.assert_allclose(nn(exact.solve(r)), O.btd_solve(ld, ls, rhs), **tol)
    np.testing.assert_allclose(nn(exact.solve(r, transpose_left=True)), O.btd_solve(ld, ls, rhs, transpose_left=True), **tol)
    inv_d, inv_s = O.btd_block_diagonal_of_inverse(ld, ls, return_sub=True)
    got_d, got_s = exact._diag_and_sub_of_inverse(want_sub=True)
    np.testing.assert_allclose(nn(got_d), inv_d, **tol)
    np.testing.assert_allclose(nn(got_s), inv_s, **tol)


@pytest.mark.parametrize("dtype,d,m,t", [(torch.float64, 14, 1, 90), (torch.float64, 32, 3, 41), (torch.float32, 20, 2, 260),
                                         (torch.float32, 64, 32, 70)])
def test_large_d_posterior_chain_partitioned_in_time(rng, dtype, d, m, t):
    """kalman_filter.py:109-182 for d > 9 on a chain long enough for the partition: all five tensors of the posterior chain (the
    means come from the affine recursion restarted at the chunk boundaries) against the oracle."""
    kw = random_ssm(rng, (2,), t, d, m, well=True)
    if dtype == torch.float32:
        kw = {k: v.astype(np.float32).astype(np.float64) for k, v in kw.items()}
    cov = 0.4 * np.eye(m)
    kf = build_kf(kw, np.linalg.cholesky(cov), dtype=dtype)
    tol = dict(rtol=1e-7, atol=1e-9) if dtype == torch.float64 else dict(rtol=5e-3, atol=5e-4)
    post = kf.posterior_state_space_model()
    want = O.kf_posterior_ssm(**kw, r_inv=np.linalg.inv(cov))
    got = (post.initial_mean, post.cholesky_initial_covariance, post.state_transitions, post.state_offsets,
           post.cholesky_process_covariances)
    for g, w in zip(got, want):
        np.testing.assert_allclose(nn(g), w, **tol)
    # marginal means alone: the affine recursion cut into chunks (state_space_model.py:232-251)
    np.testing.assert_allclose(nn(post.marginal_means), O.ssm_marginal_means(want[0], want[2], want[3]), **tol)
    np.testing.assert_allclose(nn(kf.prior_ssm.marginal_means), O.ssm_marginal_means(kw["mu0"], kw["a_s"], kw["b_s"]), **tol)
    kl = O.ssm_kl_divergence(want, (kw["mu0"], kw["chol_p0"], kw["a_s"], kw["b_s"], kw["chol_q"]))
    np.testing.assert_allclose(nn(post.kl_divergence(kf.prior_ssm)), kl, rtol=1e-6 if dtype == torch.float64 else 2e-2)


# ---- precision assembly at 32 < d <= 64 on the panel kernels (csrc/mf_panel.hpp, PREC): chunks of 16 blocks -----------------------
@pytest.mark.parametrize("d,m,t,bsz", [(33, 1, 3, 2), (40, 3, 2, 2), (48, 17, 16, 1), (49, 2, 17, 2), (64, 32, 18, 2), (64, 5, 50, 1),
                                       (56, 20, 33, 3)])
def test_panel_precision_prior_and_posterior(rng, d, m, t, bsz):
    """`ssm.precision` (state_space_model.py:431-483) and `kf._k_inv_post` (kalman_filter.py:86-101) for 32 < d <= 64 (fp32) against
    the numpy oracle, block by block: chains shorter / one longer / several times longer than a chunk, m up to 32; the
    information vector through the posterior chain's means is covered by test_large_d_posterior_marginals_and_kl."""
    kw = random_ssm(rng, (bsz,), t, d, m, well=True)
    kw = {k: v.astype(np.float32).astype(np.float64) for k, v in kw.items()}
    cov = 0.5 * np.eye(m) + 0.1 * np.ones((m, m)) / m
    chol_r = np.linalg.cholesky(cov).astype(np.float32).astype(np.float64)
    kf = build_kf(kw, chol_r, dtype=torch.float32)
    tol = dict(rtol=2e-3, atol=2e-3)
    prec = kf.prior_ssm.precision
    want_d, want_s = O.ssm_precision(kw["chol_p0"], kw["a_s"], kw["chol_q"])
    np.testing.assert_allclose(nn(prec.block_diagonal), want_d, **tol)
    if t > 1:
        np.testing.assert_allclose(nn(prec.block_sub_diagonal), want_s, **tol)
    post = kf._k_inv_post
    want_d, want_s = O.kf_posterior_precision(kw["chol_p0"], kw["a_s"], kw["chol_q"], kw["h"], np.linalg.inv(chol_r @ chol_r.T))
    np.testing.assert_allclose(nn(post.block_diagonal), want_d, **tol)
    if t > 1:
        np.testing.assert_allclose(nn(post.block_sub_diagonal), want_s, **tol)
