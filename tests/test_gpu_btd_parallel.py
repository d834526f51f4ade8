"""
GPU parity tests of the parallel-in-time Cholesky / solve (csrc/mf_btd_par.hpp), the path BASELINE config 3 takes
(few series, long chains).  The contract is the reference's: `cholesky` returns the NATURAL-ORDER factor
(/root/reference/tests/unit/test_block_tri_diag.py:94-107) and `solve` inverts it (:110-137).

Inputs are built from a random lower block-bidiagonal factor, so the exact answer is known for any length; at
oracle-sized lengths the numpy oracle is compared too.  Tolerances: fp64 rtol 1e-8 on the factor and the
solutions (the partitioned elimination reorders the arithmetic, the result is the same matrix); fp32 2e-3.
"""
import numpy as np
import pytest
import torch

import markovflow_amd as mfa
from markovflow_amd import _lib
from oracle import numpy_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def factor_and_matrix(rng, batch, n, d, dtype=np.float64):
    """Random well-conditioned lower factor (ldiag, lsub) and the SPD matrix (diag, sub) = L L^T, block-wise."""
    ldiag = np.tril(0.3 * rng.normal(size=batch + (n, d, d)))
    idx = np.arange(d)
    ldiag[..., idx, idx] = 1.0 + np.abs(rng.normal(size=batch + (n, d)))
    lsub = 0.3 * rng.normal(size=batch + (n - 1, d, d))
    diag = ldiag @ np.swapaxes(ldiag, -1, -2)
    diag[..., 1:, :, :] += lsub @ np.swapaxes(lsub, -1, -2)
    sub = lsub @ np.swapaxes(ldiag[..., :-1, :, :], -1, -2)
    return ldiag.astype(dtype), lsub.astype(dtype), diag.astype(dtype), sub.astype(dtype)


def tt(x, dtype=torch.float64):
    return torch.tensor(np.ascontiguousarray(x), dtype=dtype, device=DEV)


def nn(x):
    return x.detach().cpu().numpy().astype(np.float64)


def uses_parallel_path(bsz, n, d, esz=8):
    return _lib.load().mf_btd_cholesky_workspace_bytes(bsz, n, d, esz) > 0


# lengths chosen to hit: one reduced level (64), ragged last chunks (777, 4099), several levels (4099, 20000)
@pytest.mark.parametrize("d,n,batch", [(1, 64, ()), (2, 777, (3,)), (3, 100, (2,)), (4, 4099, ()), (6, 1000, (1,)),
                                       (6, 20000, ()), (9, 513, (2, 1)), (5, 65, (1,)),
                                       # 10 <= d <= 15: compiled with the row kernels only (one 16-lane row per chunk)
                                       (10, 300, (2,)), (12, 777, ()), (13, 1000, (1,)), (15, 200, (2,))])
def test_parallel_cholesky_and_solve_fp64(rng, d, n, batch):
    bsz = int(np.prod(batch)) if batch else 1
    assert uses_parallel_path(bsz, n, d), "this shape is meant to take the parallel-in-time path"
    assert _lib.load().mf_row_operators_cover(bsz, n, d, 8) == 1
    ldiag, lsub, diag, sub = factor_and_matrix(rng, batch, n, d)
    chol = mfa.SymmetricBlockTriDiagonal(tt(diag), tt(sub)).cholesky
    np.testing.assert_allclose(nn(chol.block_diagonal), ldiag, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(nn(chol.block_sub_diagonal), lsub, rtol=1e-8, atol=1e-10)
    rhs = rng.normal(size=batch + (n, d))
    exact = mfa.LowerTriangularBlockTriDiagonal(tt(ldiag), tt(lsub))
    for transpose in (False, True):
        got = nn(exact.solve(tt(rhs), transpose_left=transpose))
        # L (L^-1 r) = r through the independent (time-parallel) mat-vec kernel
        back = nn(exact.dense_mult(tt(got), transpose_left=transpose))
        np.testing.assert_allclose(back, rhs, rtol=1e-8, atol=1e-9)
        if n <= 1000:
            np.testing.assert_allclose(got, O.btd_solve(ldiag, lsub, rhs, transpose_left=transpose), rtol=1e-8, atol=1e-10)
    if n <= 1000:
        ld, ls = O.btd_cholesky(diag, sub)
        np.testing.assert_allclose(nn(chol.block_diagonal), np.tril(ld), rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(nn(chol.block_sub_diagonal), ls, rtol=1e-8, atol=1e-10)


def test_parallel_solve_with_extra_leading_rhs_dims(rng):
    """One factor, several right-hand sides (sample_shape + batch_shape, state_space_model.py:307-322)."""
    d, n = 3, 300
    ldiag, lsub, _, _ = factor_and_matrix(rng, (2,), n, d)
    low = mfa.LowerTriangularBlockTriDiagonal(tt(ldiag), tt(lsub))
    rhs = rng.normal(size=(4, 2, n, d))
    got = nn(low.solve(tt(rhs)))
    want = np.stack([O.btd_solve(ldiag, lsub, rhs[i]) for i in range(4)])
    np.testing.assert_allclose(got, want, rtol=1e-8, atol=1e-10)


def test_parallel_cholesky_flags_non_positive_definite(rng):
    d, n = 3, 200
    _, _, diag, sub = factor_and_matrix(rng, (), n, d)
    diag[137] = -diag[137]
    dg, sb = tt(diag[None]), tt(sub[None])
    ld, ls = torch.empty_like(dg), torch.empty_like(sb)
    lib = _lib.load()
    ws_bytes = int(lib.mf_btd_cholesky_workspace_bytes(1, n, d, 8))
    assert ws_bytes > 0
    ws = _lib.workspace(ws_bytes, dg.device)
    info = _lib.new_info(dg.device)
    _lib.call("mf_btd_cholesky", torch.float64, 1, n, d, _lib.ptr(dg), _lib.ptr(sb), _lib.ptr(ld), _lib.ptr(ls),
              _lib.ptr(ws), ws_bytes, _lib.ptr(info), _lib.stream_ptr(dg.device))
    # (the up / down sweeps raise without a location, the emit pass names the block: the smallest flat index survives)
    assert int(info.item()) > 0 and int(lib.mf_info_flat_index(int(info.item()))) == 137


def test_too_small_workspace_falls_back_to_the_serial_kernel(rng):
    """ws == NULL is legal: the natural-order serial kernel runs instead and gives the same factor."""
    d, n = 4, 256
    ldiag, lsub, diag, sub = factor_and_matrix(rng, (), n, d)
    dg, sb = tt(diag[None]), tt(sub[None])
    ld, ls = torch.empty_like(dg), torch.empty_like(sb)
    info = _lib.new_info(dg.device)
    _lib.call("mf_btd_cholesky", torch.float64, 1, n, d, _lib.ptr(dg), _lib.ptr(sb), _lib.ptr(ld), _lib.ptr(ls),
              None, 0, _lib.ptr(info), _lib.stream_ptr(dg.device))
    np.testing.assert_allclose(nn(ld[0]), ldiag, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(nn(ls[0]), lsub, rtol=1e-9, atol=1e-11)


def test_config3_full_size_fp32_recombination_and_round_trip(rng):
    """BASELINE config 3: T=100000, d=6, fp32, one chain.  Size-independent properties:
    L L^T reproduces the matrix block by block, L (L^-1 r) = r and L^T (L^-T r) = r."""
    d, n = 6, 100000
    ldiag, lsub, diag, sub = factor_and_matrix(rng, (), n, d, dtype=np.float32)
    f32 = torch.float32
    sym = mfa.SymmetricBlockTriDiagonal(tt(diag, f32), tt(sub, f32))
    chol = sym.cholesky
    gl, gw = chol.block_diagonal.double(), chol.block_sub_diagonal.double()
    rec_diag = gl @ gl.transpose(-1, -2)
    rec_diag[1:] += gw @ gw.transpose(-1, -2)
    rec_sub = gw @ gl[:-1].transpose(-1, -2)
    np.testing.assert_allclose(nn(rec_diag), diag.astype(np.float64), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(nn(rec_sub), sub.astype(np.float64), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(nn(chol.block_diagonal), ldiag.astype(np.float64), rtol=2e-3, atol=2e-4)
    rhs = tt(rng.normal(size=(n, d)), f32)
    for transpose in (False, True):
        z = chol.solve(rhs, transpose_left=transpose)
        back = chol.dense_mult(z, transpose_left=transpose)
        np.testing.assert_allclose(nn(back), nn(rhs), rtol=2e-3, atol=2e-4)
    # log-determinant through the factor equals the one through the partitioned scalar reduction
    assert torch.isfinite(chol.abs_log_det()).all()


# ---- parallel-in-time Takahashi (block_diagonal_of_inverse) and marginal means --------------------------------------------------
@pytest.mark.parametrize("d,n,batch", [(1, 64, ()), (3, 100, (2,)), (2, 777, (3,)), (6, 1000, (1,)), (9, 513, (2, 1)),
                                       (11, 200, (2,)), (15, 130, ()), (7, 300, (2,)), (8, 1000, ()), (12, 2000, (1,))])
def test_parallel_block_diagonal_of_inverse_vs_oracle(rng, d, n, batch):
    bsz = int(np.prod(batch)) if batch else 1
    assert _lib.load().mf_btd_diag_of_inverse_workspace_bytes(bsz, n, d, 8) > 0
    ldiag, lsub, _, _ = factor_and_matrix(rng, batch, n, d)
    low = mfa.LowerTriangularBlockTriDiagonal(tt(ldiag), tt(lsub))
    inv_d, inv_s = O.btd_block_diagonal_of_inverse(ldiag, lsub, return_sub=True)
    got_d, got_s = low._diag_and_sub_of_inverse(want_sub=True)
    np.testing.assert_allclose(nn(got_d), inv_d, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(nn(got_s), inv_s, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(nn(low.block_diagonal_of_inverse()), inv_d, rtol=1e-8, atol=1e-10)


def test_parallel_block_diagonal_of_inverse_long_chain_identity(rng):
    """Size-independent property at T = 20000: block row k of  M Sigma = I  restricted to the tridiagonal pattern,
    S_{k-1} Sigma_{k-1,k} + D_k Sigma_{kk} + S_k^T Sigma_{k+1,k} = I."""
    d, n = 6, 20000
    ldiag, lsub, diag, sub = factor_and_matrix(rng, (), n, d)
    low = mfa.LowerTriangularBlockTriDiagonal(tt(ldiag), tt(lsub))
    sig_d, sig_s = low._diag_and_sub_of_inverse(want_sub=True)
    dg, sb = tt(diag), tt(sub)
    row = dg @ sig_d
    row[1:] += sb @ sig_s.transpose(-1, -2)
    row[:-1] += sb.transpose(-1, -2) @ sig_s
    eye = torch.eye(d, dtype=torch.float64, device=DEV).expand(n, d, d)
    np.testing.assert_allclose(nn(row), nn(eye), rtol=0, atol=1e-8)


@pytest.mark.parametrize("d,n,batch", [(2, 64, ()), (6, 1000, (2,)), (4, 4099, ()), (9, 300, (3,)), (12, 300, (2,)), (15, 100, ())])
def test_parallel_marginal_means_and_sample_propagation(rng, d, n, batch):
    a = 0.9 * np.eye(d) + 0.1 * rng.normal(size=batch + (n - 1, d, d)) / np.sqrt(d)
    mu0, b = rng.normal(size=batch + (d,)), rng.normal(size=batch + (n - 1, d))
    chol = np.tile(np.eye(d), batch + (n - 1, 1, 1))
    ssm = mfa.StateSpaceModel(tt(mu0), tt(np.tile(np.eye(d), batch + (1, 1))), tt(a), tt(b), tt(chol))
    np.testing.assert_allclose(nn(ssm.marginal_means), O.ssm_marginal_means(mu0, a, b), rtol=1e-9, atol=1e-10)
    # extra leading dims (sample_shape + batch_shape, state_space_model.py:307-322): one set of transitions, 3 offsets
    offs = rng.normal(size=(3,) + batch + (n, d))
    got = nn(ssm._propagate(tt(offs)))
    want = np.stack([O.ssm_marginal_means(offs[i][..., 0, :], a, offs[i][..., 1:, :]) for i in range(3)])
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-10)


# ---- parallel-in-time U D U^T and posterior chain ----------------------------------------------------------------------------------
@pytest.mark.parametrize("d,n,batch", [(1, 64, ()), (3, 100, (2,)), (2, 777, (3,)), (6, 1000, (1,)), (9, 300, (2, 1)),
                                       (10, 100, (2,)), (14, 300, ()), (15, 777, (1,))])
def test_parallel_upper_diagonal_lower_vs_oracle(rng, d, n, batch):
    bsz = int(np.prod(batch)) if batch else 1
    assert _lib.load().mf_btd_udl_workspace_bytes(bsz, n, d, 8) > 0
    _, _, diag, sub = factor_and_matrix(rng, batch, n, d)
    u_t, chol_d = mfa.SymmetricBlockTriDiagonal(tt(diag), tt(sub)).upper_diagonal_lower()
    want_u, want_c = O.btd_upper_diagonal_lower(diag, sub)
    np.testing.assert_allclose(nn(u_t.block_sub_diagonal), want_u, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(nn(chol_d.block_diagonal), np.tril(want_c), rtol=1e-8, atol=1e-10)


def test_parallel_udl_long_chain_recombines(rng):
    """U D U^T = M block by block at T = 20000 (test_block_tri_diag.py:205-225 of the reference, size-independent form):
    D_k = Delta_k + U_k Delta_{k+1} U_k^T,  S_k = Delta_{k+1} U_k^T."""
    d, n = 6, 20000
    _, _, diag, sub = factor_and_matrix(rng, (), n, d)
    u_t, chol_d = mfa.SymmetricBlockTriDiagonal(tt(diag), tt(sub)).upper_diagonal_lower()
    ut, cd = u_t.block_sub_diagonal, chol_d.block_diagonal
    delta = cd @ cd.transpose(-1, -2)
    rec_sub = delta[1:] @ ut
    rec_diag = delta.clone()
    rec_diag[:-1] += ut.transpose(-1, -2) @ delta[1:] @ ut
    np.testing.assert_allclose(nn(rec_sub), sub, rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(nn(rec_diag), diag, rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("d,m,n", [(2, 1, 300), (6, 1, 500), (4, 2, 129), (10, 1, 200), (12, 3, 300), (15, 4, 129),
                                   # 10 <= d <= 15, chains too short to cut: the same row kernels with one chunk per series
                                   (12, 3, 9), (15, 2, 2), (10, 4, 40)])
def test_parallel_posterior_state_space_model_vs_oracle(rng, d, m, n):
    """KalmanFilter.posterior_state_space_model (kalman_filter.py:109-182) on the parallel-in-time path (few series)."""
    from test_gpu_kalman import build_kf, random_ssm
    kw = random_ssm(rng, (2,), n, d, m, well=d >= 10)
    cov = 0.4 * np.eye(m)
    kf = build_kf(kw, np.linalg.cholesky(cov))
    post = kf.posterior_state_space_model()
    want = O.kf_posterior_ssm(**kw, r_inv=np.linalg.inv(cov))
    got = (post.initial_mean, post.cholesky_initial_covariance, post.state_transitions, post.state_offsets,
           post.cholesky_process_covariances)
    for g, w in zip(got, want):
        np.testing.assert_allclose(nn(g), w, rtol=1e-7, atol=1e-9)
    # smoothed marginals against the oracle's (means: affine scan; covariances: Cholesky + Takahashi, all parallel in time)
    means = O.ssm_marginal_means(want[0], want[2], want[3])
    np.testing.assert_allclose(nn(post.marginal_means), means, rtol=1e-7, atol=1e-9)
    covs = O.ssm_marginal_covariances(want[1], want[2], want[4])
    np.testing.assert_allclose(nn(post.marginal_covariances), covs, rtol=1e-6, atol=1e-9)
    # KL(posterior || prior) from its local form (state_space_model.py:528-593), every series
    kl = O.ssm_kl_divergence(want, (kw["mu0"], kw["chol_p0"], kw["a_s"], kw["b_s"], kw["chol_q"]))
    np.testing.assert_allclose(nn(post.kl_divergence(kf.prior_ssm)), kl, rtol=1e-7)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("d,n,batch", [(1, 2, ()), (3, 9, (2,)), (6, 64, (3,)), (2, 777, (2, 1)), (6, 1000, (1,)), (9, 300, (2,)), (7, 130, ()),
                                       (12, 300, (2,)), (15, 130, ())])
def test_covariance_scan_equals_the_reference_route(rng, dtype, d, n, batch):
    """marginal / subsequent covariances by the forward recursion (mf_ssm_marginal_covariances) against the reference's route:
    assembled precision -> Cholesky -> block diagonal of the inverse, and A_k Sigma_k (state_space_model.py:254-275,326-341)."""
    from test_gpu_kalman import random_ssm
    kw = random_ssm(rng, batch, n, d, 1, well=True)
    t = lambda x: torch.tensor(np.ascontiguousarray(x), dtype=dtype, device="cuda:0")   # noqa: E731
    ssm = mfa.StateSpaceModel(t(kw["mu0"]), t(kw["chol_p0"]), t(kw["a_s"]), t(kw["b_s"]), t(kw["chol_q"]))
    covs, sub = ssm.covariance_blocks()
    ref_covs = ssm.precision.cholesky.block_diagonal_of_inverse()
    ref_sub = ssm.subsequent_covariances(ref_covs)
    tol = dict(rtol=1e-9, atol=1e-11) if dtype == torch.float64 else dict(rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(covs.cpu().numpy(), ref_covs.cpu().numpy(), **tol)
    np.testing.assert_allclose(sub.cpu().numpy(), ref_sub.cpu().numpy(), **tol)
    np.testing.assert_allclose(ssm.marginal_covariances.cpu().numpy(), covs.cpu().numpy(), **tol)
    # explicit recursion in numpy (tests/unit/test_state_space_model.py:63-89 of the reference)
    if dtype == torch.float64 and not batch:
        cov = kw["chol_p0"] @ kw["chol_p0"].T
        for k in range(n - 1):
            np.testing.assert_allclose(covs[k].cpu().numpy(), cov, rtol=1e-9, atol=1e-11)
            cov = kw["a_s"][k] @ cov @ kw["a_s"][k].T + kw["chol_q"][k] @ kw["chol_q"][k].T
        np.testing.assert_allclose(covs[n - 1].cpu().numpy(), cov, rtol=1e-9, atol=1e-11)


def test_row_only_dimensions_with_many_short_series(rng):
    """d = 13 with 4100 series of 12 points: beyond the partition's series limit the row-only build runs one chunk per series
    (mf_row_operators_cover = 1 for every shape) - posterior chain, marginals and KL of every series against the oracle."""
    from test_gpu_kalman import build_kf, random_ssm
    d, m, n, bsz = 13, 2, 12, 4100
    assert _lib.load().mf_row_operators_cover(bsz, n, d, 8) == 1
    kw = random_ssm(rng, (bsz,), n, d, m, well=True)
    cov = 0.4 * np.eye(m)
    kf = build_kf(kw, np.linalg.cholesky(cov))
    post = kf.posterior_state_space_model()
    want = O.kf_posterior_ssm(**kw, r_inv=np.linalg.inv(cov))
    got = (post.initial_mean, post.cholesky_initial_covariance, post.state_transitions, post.state_offsets,
           post.cholesky_process_covariances)
    for g, w in zip(got, want):
        np.testing.assert_allclose(nn(g), w, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(nn(post.marginal_covariances), O.ssm_marginal_covariances(want[1], want[2], want[4]), rtol=1e-6, atol=1e-9)
    kl = O.ssm_kl_divergence(want, (kw["mu0"], kw["chol_p0"], kw["a_s"], kw["b_s"], kw["chol_q"]))
    np.testing.assert_allclose(nn(post.kl_divergence(kf.prior_ssm)), kl, rtol=1e-7)
    ref = O.kf_log_likelihood(**kw, r_inv=np.linalg.inv(cov), per_series=True)
    np.testing.assert_allclose(nn(kf._log_likelihood_per_series() + kf._constant_terms(n)), ref, rtol=1e-9)
