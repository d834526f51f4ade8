"""
CPU oracle (numpy, fp64) for the Kalman / block-tridiagonal hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.  The product
path (``markovflow_amd``) never routes through this module.

It restates, block by block, what the reference executes for this path.  The arithmetic of the
banded ops lives in the un-vendored third-party library ``banded-matrices==0.0.6``
(``/root/reference/pyproject.toml:17``); its published semantics (banded Cholesky, triangular solve,
band x vector product, sparse-inverse subset) are restated here in block form and anchored on the
reference's own call sites and on the dense identities of
``/root/reference/tests/unit/test_block_tri_diag.py:79-225``.

Parity pin: ``tests/test_oracle_golden.py`` checks every function below against
(1) golden vectors produced in the build container by the reference's own numpy tools
    (``tests/tools/numpy_kalman_filter.py``, ``tests/tools/kernels/kernels.py``; generator:
    ``tests/golden/make_golden.py``) and (2) dense ``numpy.linalg`` identities.

All functions take arrays with arbitrary leading batch dims and are written for clarity, with plain
Python loops over the time axis.
"""
from typing import Optional, Tuple

import numpy as np

_T = lambda x: np.swapaxes(x, -1, -2)  # noqa: E731


def _chol_solve(chol: np.ndarray, rhs: np.ndarray) -> np.ndarray:
    """(chol cholᵀ)⁻¹ rhs, batched; the meaning of ``tf.linalg.cholesky_solve``."""
    full = chol @ _T(chol)
    return np.linalg.solve(full, rhs)


# ----------------------------------------------------------------------------------------------
# Block-tridiagonal operator                       (reference: markovflow/block_tri_diag.py)
# ----------------------------------------------------------------------------------------------


def btd_to_dense(diag: np.ndarray, sub: Optional[np.ndarray], symmetric: bool) -> np.ndarray:
    """``BlockTriDiagonal.to_dense`` (block_tri_diag.py:150-173).

    Lower-triangular objects only keep the lower triangle of each diagonal block (the band layout
    of block_tri_diag.py:206-237 drops the rest); symmetric objects are mirrored.
    """
    *batch, n, d, _ = diag.shape
    dense = np.zeros(tuple(batch) + (n * d, n * d), dtype=diag.dtype)
    for i in range(n):
        dense[..., i * d:(i + 1) * d, i * d:(i + 1) * d] = np.tril(diag[..., i, :, :])
        if sub is not None and i < n - 1:
            dense[..., (i + 1) * d:(i + 2) * d, i * d:(i + 1) * d] = sub[..., i, :, :]
    if symmetric:
        dg = np.einsum("...ii->...i", dense)
        dense = dense + _T(dense)
        idx = np.arange(n * d)
        dense[..., idx, idx] = dg
    return dense


def btd_cholesky(diag: np.ndarray, sub: Optional[np.ndarray]) -> Tuple[np.ndarray, Optional[np.ndarray]]:
    """``SymmetricBlockTriDiagonal.cholesky`` (block_tri_diag.py:423-436), natural order.

    L_0 = chol(D_0);  W_{k-1} = S_{k-1} L_{k-1}⁻ᵀ;  L_k = chol(D_k - W_{k-1} W_{k-1}ᵀ).
    Pinned by tests/unit/test_block_tri_diag.py:94-107 (== np.linalg.cholesky of the dense matrix).
    """
    n = diag.shape[-3]
    l_diag = np.zeros_like(diag)
    l_sub = None if sub is None else np.zeros_like(sub)
    l_diag[..., 0, :, :] = np.linalg.cholesky(diag[..., 0, :, :])
    for k in range(1, n):
        if sub is None:
            l_diag[..., k, :, :] = np.linalg.cholesky(diag[..., k, :, :])
            continue
        # W = S L⁻ᵀ  <=>  L Wᵀ = Sᵀ
        w = _T(np.linalg.solve(l_diag[..., k - 1, :, :], _T(sub[..., k - 1, :, :])))
        l_sub[..., k - 1, :, :] = w
        l_diag[..., k, :, :] = np.linalg.cholesky(diag[..., k, :, :] - w @ _T(w))
    return l_diag, l_sub


def btd_solve(l_diag: np.ndarray, l_sub: Optional[np.ndarray], rhs: np.ndarray,
              transpose_left: bool = False) -> np.ndarray:
    """``LowerTriangularBlockTriDiagonal.solve`` (block_tri_diag.py:339-351): L⁻¹x or L⁻ᵀx.

    Only the lower triangle of each diagonal block is used (band layout).  ``rhs`` may carry
    extra leading dims / broadcast against the factor's batch (block_tri_diag.py:261-287).
    Pinned by tests/unit/test_block_tri_diag.py:110-136.
    """
    n = l_diag.shape[-3]
    ld = np.tril(l_diag)
    out_shape = np.broadcast_shapes(rhs.shape[:-2], l_diag.shape[:-3]) + rhs.shape[-2:]
    out = np.zeros(out_shape, dtype=np.result_type(rhs.dtype, l_diag.dtype))
    rhs = np.broadcast_to(rhs, out_shape)

    def mv(m, v):
        return (m @ v[..., None])[..., 0]

    def sv(m, v):
        m, v = np.broadcast_arrays(m, v[..., None])
        return np.linalg.solve(m, v)[..., 0]

    if not transpose_left:
        for k in range(n):
            r = rhs[..., k, :]
            if l_sub is not None and k > 0:
                r = r - mv(l_sub[..., k - 1, :, :], out[..., k - 1, :])
            out[..., k, :] = sv(ld[..., k, :, :], r)
    else:
        for k in reversed(range(n)):
            r = rhs[..., k, :]
            if l_sub is not None and k < n - 1:
                r = r - mv(_T(l_sub[..., k, :, :]), out[..., k + 1, :])
            out[..., k, :] = sv(_T(ld[..., k, :, :]), r)
    return out


def btd_dense_mult(diag: np.ndarray, sub: Optional[np.ndarray], right: np.ndarray,
                   symmetric: bool, transpose_left: bool = False) -> np.ndarray:
    """``BlockTriDiagonal.dense_mult`` (block_tri_diag.py:175-199): Mx, Mᵀx or symmetrised Mx.

    Pinned by tests/unit/test_block_tri_diag.py:139-180.
    """
    n = diag.shape[-3]

    def mv(m, v):
        return (m @ v[..., None])[..., 0]

    if symmetric:
        lo = np.tril(diag)
        dblk = lo + _T(lo) - lo * np.eye(diag.shape[-1])
    else:
        dblk = np.tril(diag)
        if transpose_left:
            dblk = _T(dblk)
    out = mv(dblk, right)
    if sub is not None:
        below = mv(sub, right[..., :-1, :])        # contributes to rows 1..n-1  (M x)
        above = mv(_T(sub), right[..., 1:, :])     # contributes to rows 0..n-2  (Mᵀ x)
        if symmetric:
            out[..., 1:, :] += below
            out[..., :-1, :] += above
        elif transpose_left:
            out[..., :-1, :] += above
        else:
            out[..., 1:, :] += below
    assert out.shape[-2] == n
    return out


def btd_abs_log_det(l_diag: np.ndarray) -> np.ndarray:
    """``LowerTriangularBlockTriDiagonal.abs_log_det`` (block_tri_diag.py:353-366)."""
    dg = np.einsum("...ii->...i", l_diag)
    return 0.5 * np.sum(np.log(np.square(dg)), axis=(-1, -2))


def btd_block_diagonal_of_inverse(l_diag: np.ndarray, l_sub: Optional[np.ndarray],
                                  return_sub: bool = False):
    """``block_diagonal_of_inverse`` (block_tri_diag.py:318-337): diagonal blocks of (LLᵀ)⁻¹.

    Block Takahashi recursion, backward:  Σ_{n-1} = L⁻ᵀL⁻¹;  G_k = W_k L_k⁻¹;
    Σ_kk = L_k⁻ᵀL_k⁻¹ + G_kᵀ Σ_{k+1,k+1} G_k;  Σ_{k+1,k} = -Σ_{k+1,k+1} G_k.
    (The sub-diagonal blocks are what ssm_gaussian_transformations.py:453-458 reads.)
    Pinned by tests/unit/test_block_tri_diag.py:183-202.
    """
    n = l_diag.shape[-3]
    ld = np.tril(l_diag)
    eye = np.eye(l_diag.shape[-1])
    out = np.zeros_like(l_diag)
    out_sub = None if l_sub is None else np.zeros_like(l_sub)
    for k in reversed(range(n)):
        linv = np.linalg.solve(ld[..., k, :, :], np.broadcast_to(eye, ld[..., k, :, :].shape))
        s = _T(linv) @ linv
        if l_sub is not None and k < n - 1:
            g = l_sub[..., k, :, :] @ linv
            s = s + _T(g) @ out[..., k + 1, :, :] @ g
            out_sub[..., k, :, :] = -out[..., k + 1, :, :] @ g
        out[..., k, :, :] = s
    if return_sub:
        return out, out_sub
    return out


def btd_upper_diagonal_lower(diag: np.ndarray, sub: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """``SymmetricBlockTriDiagonal.upper_diagonal_lower`` (block_tri_diag.py:438-545).

    Backward: Δ_{n-1} = D_{n-1}; Δ_k = D_k - S_kᵀ Δ_{k+1}⁻¹ S_k; U_kᵀ = Δ_{k+1}⁻¹ S_k.
    Returns (U_sᵀ as the sub-diagonal of a unit lower block-bidiagonal, chol(Δ_k) block diagonal).
    Pinned by tests/unit/test_block_tri_diag.py:205-225.
    """
    n = diag.shape[-3]
    chol_d = np.zeros_like(diag)
    u_t = np.zeros_like(sub)
    chol_d[..., n - 1, :, :] = np.linalg.cholesky(diag[..., n - 1, :, :])
    for k in reversed(range(n - 1)):
        d_inv_s = _chol_solve(chol_d[..., k + 1, :, :], sub[..., k, :, :])
        u_t[..., k, :, :] = d_inv_s
        chol_d[..., k, :, :] = np.linalg.cholesky(diag[..., k, :, :] - _T(sub[..., k, :, :]) @ d_inv_s)
    return u_t, chol_d


# ----------------------------------------------------------------------------------------------
# State space model                                (reference: markovflow/state_space_model.py)
# ----------------------------------------------------------------------------------------------


def ssm_precision(chol_p0: np.ndarray, a_s: np.ndarray, chol_q: np.ndarray):
    """``StateSpaceModel._build_precision`` (state_space_model.py:431-483).

    sub_k = -Q_{k+1}⁻¹A_{k+1};  diag_k = [P₀⁻¹, Q⁻¹...]_k + A_{k+1}ᵀQ_{k+1}⁻¹A_{k+1} (absent for last).
    """
    inv_q_a = _chol_solve(chol_q, a_s)
    aqa = _T(a_s) @ inv_q_a
    cat = np.concatenate([chol_p0[..., None, :, :], chol_q], axis=-3)
    eye = np.broadcast_to(np.eye(cat.shape[-1]), cat.shape)
    diag = _chol_solve(cat, eye)
    diag[..., :-1, :, :] += aqa
    return diag, -inv_q_a


def ssm_marginal_means(mu0: np.ndarray, a_s: np.ndarray, b_s: np.ndarray) -> np.ndarray:
    """``StateSpaceModel.marginal_means`` (state_space_model.py:232-251): μ_{k+1} = A_k μ_k + b_k."""
    n = a_s.shape[-3]
    out = np.zeros(mu0.shape[:-1] + (n + 1, mu0.shape[-1]), dtype=mu0.dtype)
    out[..., 0, :] = mu0
    for k in range(n):
        out[..., k + 1, :] = (a_s[..., k, :, :] @ out[..., k, :, None])[..., 0] + b_s[..., k, :]
    return out


def ssm_log_det_precision(chol_p0: np.ndarray, chol_q: np.ndarray) -> np.ndarray:
    """``StateSpaceModel.log_det_precision`` (state_space_model.py:343-373)."""
    d0 = np.einsum("...ii->...i", chol_p0)
    dq = np.einsum("...ii->...i", chol_q)
    return -(np.sum(np.log(np.square(d0)), axis=-1) + np.sum(np.log(np.square(dq)), axis=(-1, -2)))


def ssm_marginal_covariances(chol_p0, a_s, chol_q) -> np.ndarray:
    """``StateSpaceModel.marginal_covariances`` (state_space_model.py:254-262)."""
    diag, sub = ssm_precision(chol_p0, a_s, chol_q)
    return btd_block_diagonal_of_inverse(*btd_cholesky(diag, sub))


def ssm_subsequent_covariances(a_s, marginal_covs) -> np.ndarray:
    """``StateSpaceModel.subsequent_covariances`` (state_space_model.py:326-341)."""
    return a_s @ marginal_covs[..., :-1, :, :]


def ssm_kl_divergence(ssm1, ssm2) -> np.ndarray:
    """``StateSpaceModel.kl_divergence`` (state_space_model.py:528-593); ssm = (mu0, cholP0, A, b, cholQ)."""
    mu0_1, cp0_1, a_1, b_1, cq_1 = ssm1
    mu0_2, cp0_2, a_2, b_2, cq_2 = ssm2
    covs_1 = ssm_marginal_covariances(cp0_1, a_1, cq_1)
    diag_2, sub_2 = ssm_precision(cp0_2, a_2, cq_2)
    sub_covs_1 = ssm_subsequent_covariances(a_1, covs_1)
    trace = np.sum(diag_2 * covs_1, axis=(-3, -2, -1)) + 2.0 * np.sum(sub_2 * sub_covs_1, axis=(-3, -2, -1))
    mean_diff = ssm_marginal_means(mu0_2, a_2, b_2) - ssm_marginal_means(mu0_1, a_1, b_1)
    l2d, l2s = btd_cholesky(diag_2, sub_2)
    lmd = btd_dense_mult(l2d, l2s, mean_diff, symmetric=False, transpose_left=True)
    mahalanobis = np.sum(lmd * lmd, axis=(-2, -1))
    dim = (a_1.shape[-3] + 1) * a_1.shape[-1]
    return 0.5 * (trace + mahalanobis - dim
                  - ssm_log_det_precision(cp0_2, cq_2) + ssm_log_det_precision(cp0_1, cq_1))


def ssm_log_pdf(ssm, states: np.ndarray) -> np.ndarray:
    """``StateSpaceModel.log_pdf`` (state_space_model.py:485-526)."""
    mu0, cp0, a_s, b_s, cq = ssm
    d = mu0.shape[-1]

    def mvn_tril(loc, tril, x):
        diff = x - loc
        tril_b, diff_b = np.broadcast_arrays(tril, diff[..., None])
        z = np.linalg.solve(tril_b, diff_b)[..., 0]
        return (-0.5 * np.sum(z * z, axis=-1) - np.sum(np.log(np.abs(np.einsum("...ii->...i", tril))), axis=-1)
                - 0.5 * d * np.log(2 * np.pi))

    init = mvn_tril(mu0, cp0, states[..., 0, :])
    cond = (a_s @ states[..., :-1, :, None])[..., 0] + b_s
    rest = mvn_tril(cond, cq, states[..., 1:, :])
    return init + np.sum(rest, axis=-1)


def cholesky_or_zero(cov: np.ndarray) -> np.ndarray:
    """``state_space_model_from_covariances.cholesky_or_zero`` (state_space_model.py:634-656)."""
    mask = np.all(cov == 0, axis=(-2, -1))[..., None, None]
    fix = np.where(mask, np.eye(cov.shape[-1]), 0.0)
    return np.where(mask, 0.0, np.linalg.cholesky(cov + fix))


# ----------------------------------------------------------------------------------------------
# Kalman filter                                     (reference: markovflow/kalman_filter.py)
# ----------------------------------------------------------------------------------------------


def _r_inv_from_chol(chol_r: np.ndarray) -> np.ndarray:
    """``KalmanFilter._r_inv`` (kalman_filter.py:341-348)."""
    return _chol_solve(chol_r, np.eye(chol_r.shape[-1]))


def kf_posterior_precision(chol_p0, a_s, chol_q, h, r_inv):
    """``BaseKalmanFilter._k_inv_post`` (kalman_filter.py:86-101): K⁻¹ + GᵀΣ⁻¹G."""
    diag, sub = ssm_precision(chol_p0, a_s, chol_q)
    hrh = np.einsum("...ji,...jk,...kl->...il", h, r_inv, h)
    return diag + hrh, sub


def kf_back_project(h, r_inv, obs):
    """``BaseKalmanFilter._back_project_y_to_state`` (kalman_filter.py:257-271): (GᵀΣ⁻¹) y."""
    back = np.einsum("...ij,...ki->...kj", h, r_inv)
    return np.einsum("...ij,...i->...j", back, obs)


def kf_log_likelihood(mu0, chol_p0, a_s, b_s, chol_q, h, y, r_inv,
                      log_det_obs_precision: Optional[np.ndarray] = None,
                      per_series: bool = False):
    """``BaseKalmanFilter.log_likelihood`` (kalman_filter.py:184-255).

    ``r_inv`` is [m, m] (shared, ``KalmanFilter``) or [..., T, m, m] (``KalmanFilterWithSites``).
    ``log_det_obs_precision`` defaults to T·logdet(R⁻¹) (kalman_filter.py:103-107); the sites
    variant passes Σ_k logdet R_k⁻¹ (kalman_filter.py:489-492).
    Returns the scalar summed over batch (kalman_filter.py:255) unless ``per_series``.
    """
    num_data = a_s.shape[-3] + 1
    m = h.shape[-2]
    pd, ps = kf_posterior_precision(chol_p0, a_s, chol_q, h, r_inv)
    ld, ls = btd_cholesky(pd, ps)
    marginal = (h @ ssm_marginal_means(mu0, a_s, b_s)[..., None])[..., 0]
    disp = y - marginal
    cst = -0.5 * np.log(2 * np.pi) * (m * num_data)
    term1 = -0.5 * np.sum(np.einsum("...op,...p,...o->...o", r_inv, disp, disp), axis=(-1, -2))
    obs_proj = kf_back_project(h, r_inv, disp)
    term2 = 0.5 * np.sum(np.square(btd_solve(ld, ls, obs_proj)), axis=(-1, -2))
    if log_det_obs_precision is None:
        log_det_obs_precision = num_data * np.linalg.slogdet(r_inv)[1]
    term3 = 0.5 * ssm_log_det_precision(chol_p0, chol_q) - btd_abs_log_det(ld) + 0.5 * log_det_obs_precision
    out = cst + term1 + term2 + term3
    return out if per_series else np.sum(out)


def kf_posterior_ssm(mu0, chol_p0, a_s, b_s, chol_q, h, y, r_inv):
    """``BaseKalmanFilter.posterior_state_space_model`` (kalman_filter.py:109-182).

    Returns (mu0', cholP0', A', b', cholQ').
    """
    pd, ps = kf_posterior_precision(chol_p0, a_s, chol_q, h, r_inv)
    u_t, chol_d = btd_upper_diagonal_lower(pd, ps)
    obs_proj = kf_back_project(h, r_inv, y)
    prior_d, prior_s = ssm_precision(chol_p0, a_s, chol_q)
    k_inv_mu = btd_dense_mult(prior_d, prior_s, ssm_marginal_means(mu0, a_s, b_s), symmetric=True)
    eye_blocks = np.broadcast_to(np.eye(pd.shape[-1]), pd.shape).copy()
    x = btd_solve(eye_blocks, u_t, obs_proj + k_inv_mu, transpose_left=True)
    m_post = btd_solve(chol_d, None, btd_solve(chol_d, None, x), transpose_left=True)
    eye = np.broadcast_to(np.eye(pd.shape[-1]), pd.shape)
    qs = np.linalg.cholesky(_chol_solve(chol_d, eye))
    return (m_post[..., 0, :], qs[..., 0, :, :], -u_t, m_post[..., 1:, :], qs[..., 1:, :, :])


def kf_sparse_sites_log_likelihood(mu0, chol_p0, a_s, b_s, chol_q, h, obs_index, sparse_obs,
                                   site_precisions):
    """``KalmanFilterWithSparseSites.log_likelihood`` (kalman_filter.py:579-626); no batch dims.

    ``obs_index`` [N] int, ``sparse_obs`` [N, 1], ``site_precisions`` [N, 1, 1].
    """
    grid = a_s.shape[-3] + 1
    r_inv = np.zeros((grid, 1, 1))
    r_inv[obs_index] = site_precisions                       # sparse_to_dense (kalman_filter.py:561-565)
    obs = np.zeros((grid, 1))
    obs[obs_index] = sparse_obs
    num_data = obs_index.shape[0]
    pd, ps = kf_posterior_precision(chol_p0, a_s, chol_q, h, r_inv)
    ld, ls = btd_cholesky(pd, ps)
    marginal = (h @ ssm_marginal_means(mu0, a_s, b_s)[..., None])[..., 0]
    disp = obs - marginal
    disp_data = sparse_obs - marginal[obs_index]
    cst = -0.5 * np.log(2 * np.pi) * (h.shape[-2] * num_data)
    term1 = -0.5 * np.sum(np.einsum("...op,...p,...o->...o", site_precisions, disp_data, disp_data))
    obs_proj = kf_back_project(h, r_inv, disp)
    term2 = 0.5 * np.sum(np.square(btd_solve(ld, ls, obs_proj)))
    term3 = (0.5 * ssm_log_det_precision(chol_p0, chol_q) - btd_abs_log_det(ld)
             + 0.5 * np.sum(np.linalg.slogdet(site_precisions)[1]))
    return cst + term1 + term2 + term3


# ----------------------------------------------------------------------------------------------
# Dense helpers used only to pin the block recurrences above
# ----------------------------------------------------------------------------------------------


def dense_gaussian_kl(mean1, cov1, mean2, cov2) -> float:
    """Closed-form KL(N1 || N2); stands in for tfp in tests/unit/test_state_space_model.py:144-176."""
    n = mean1.shape[-1]
    sol = np.linalg.solve(cov2, cov1)
    diff = mean2 - mean1
    return 0.5 * (np.trace(sol) + diff @ np.linalg.solve(cov2, diff) - n
                  + np.linalg.slogdet(cov2)[1] - np.linalg.slogdet(cov1)[1])
