"""
GPU parity tests of the large-state-dimension log-likelihood (csrc/mf_big.hpp: one workgroup per (series, chunk),
LDS-resident d x d tiles, f32 MFMA) - the path BASELINE config 5 (state_dim = 64, fp32) takes.  Same contract as the
small-d tests: KalmanFilter.log_likelihood of /root/reference/markovflow/kalman_filter.py:184-255, compared with the
fp64 numpy oracle on identical (fp32-rounded) inputs.  Tolerance: fp32 arithmetic, rtol 3e-4 on the scalar.
"""
import numpy as np
import pytest
import torch

import markovflow_amd as mfa
from markovflow_amd import _lib
from oracle import numpy_oracle as O
from test_gpu_kalman import DEV, build_kf, loglik_with_chunks, nn, random_ssm, tt

pytestmark = pytest.mark.gpu
F32 = torch.float32
RTOL = 3e-4


def rounded(kw):
    """The values the fp32 kernel actually sees."""
    return {k: v.astype(np.float32).astype(np.float64) for k, v in kw.items()}


@pytest.mark.parametrize("d,m,t,batch", [(10, 1, 7, (2,)), (16, 2, 12, (3,)), (17, 1, 9, ()), (32, 4, 20, (2,)),
                                         (40, 3, 6, (1,)), (48, 1, 33, (2,)), (64, 1, 2, (2,)), (64, 2, 3, (1,)),
                                         (64, 32, 24, (2,)), (64, 5, 40, (3,))])
def test_large_d_log_likelihood_vs_oracle(rng, d, m, t, batch):
    kw = rounded(random_ssm(rng, batch, t, d, m, well=True))
    cov = 0.5 * np.eye(m) + 0.1 * np.ones((m, m))
    chol_r = np.linalg.cholesky(cov).astype(np.float32).astype(np.float64)
    kf = build_kf(kw, chol_r, dtype=F32)
    ref = O.kf_log_likelihood(**kw, r_inv=np.linalg.inv(chol_r @ chol_r.T))
    np.testing.assert_allclose(float(kf.log_likelihood().cpu()), ref, rtol=RTOL)


@pytest.mark.parametrize("chunks", [1, 2, 3, 5, 16, 24, 70])
def test_large_d_time_partition_invariance(rng, chunks):
    """Any partition of the time axis (including >1 reduction level: 70 chunks -> 9 -> 2) gives the same scalar."""
    d, m, t = 24, 2, 281
    kw = rounded(random_ssm(rng, (2,), t, d, m, well=True))
    r_inv = np.array([[2.0, 0.3], [0.3, 1.5]])
    ref = O.kf_log_likelihood(**kw, r_inv=r_inv, per_series=True)
    cst = -0.5 * np.log(2 * np.pi) * m * t + 0.5 * t * np.linalg.slogdet(r_inv)[1]
    got = loglik_with_chunks(kw, r_inv, chunks, dtype=F32)
    np.testing.assert_allclose(got + cst, ref, rtol=RTOL)


def test_large_d_per_step_precisions(rng):
    """KalmanFilterWithSites (per-step R^-1, m = 1) through the large-d kernel (kalman_filter.py:437-497)."""
    d, t = 20, 30
    kw = rounded(random_ssm(rng, (), t, d, 1, well=True))
    prec = (0.5 + rng.random(size=(t, 1, 1))).astype(np.float32).astype(np.float64)
    means = kw["y"]
    ssm = mfa.StateSpaceModel(*(tt(kw[k], F32) for k in ("mu0", "chol_p0", "a_s", "b_s", "chol_q")))
    sites = mfa.UnivariateGaussianSitesNat(nat1=tt(means * prec[..., 0], F32), nat2=tt(-0.5 * prec, F32))
    kf = mfa.KalmanFilterWithSites(ssm, mfa.EmissionModel(tt(kw["h"], F32)), sites)
    ref = O.kf_log_likelihood(**kw, r_inv=prec, log_det_obs_precision=np.sum(np.log(prec)))
    np.testing.assert_allclose(float(kf.log_likelihood().cpu()), ref, rtol=RTOL)


def test_config5_shape_state_dim_64(rng):
    """BASELINE config 5 shape (d = 64, fp32, m = 32 spatial outputs), time axis cut to what the oracle does in seconds,
    plus the size-independent property at the full T = 2048: the scalar does not depend on the time partition."""
    d, m = 64, 32
    kw = rounded(random_ssm(rng, (1,), 48, d, m, well=True))
    chol_r = np.sqrt(0.1) * np.eye(m)
    chol_r = chol_r.astype(np.float32).astype(np.float64)
    kf = build_kf(kw, chol_r, dtype=F32)
    ref = O.kf_log_likelihood(**kw, r_inv=np.linalg.inv(chol_r @ chol_r.T))
    np.testing.assert_allclose(float(kf.log_likelihood().cpu()), ref, rtol=RTOL)
    big = rounded(random_ssm(rng, (1,), 2048, d, 4, well=True))
    r_inv = 2.0 * np.eye(4)
    a = loglik_with_chunks(big, r_inv, 0, dtype=F32)
    b = loglik_with_chunks(big, r_inv, 37, dtype=F32)
    np.testing.assert_allclose(a, b, rtol=RTOL)


@pytest.mark.parametrize("d,m,t,batch", [(10, 1, 7, (2,)), (14, 2, 30, (3,)), (16, 3, 12, ()), (24, 1, 50, (2,)),
                                         (32, 32, 9, (1,)), (32, 2, 64, (2,))])
def test_large_d_fp64_log_likelihood_vs_oracle(rng, d, m, t, batch):
    """fp64 on f64 MFMA (d <= 32): same tolerance as the register-resident fp64 kernels, rtol 1e-9."""
    kw = random_ssm(rng, batch, t, d, m, well=True)
    cov = 0.5 * np.eye(m) + 0.1 * np.ones((m, m))
    kf = build_kf(kw, np.linalg.cholesky(cov))
    ref = O.kf_log_likelihood(**kw, r_inv=np.linalg.inv(cov))
    np.testing.assert_allclose(float(kf.log_likelihood().cpu()), ref, rtol=1e-9)


@pytest.mark.parametrize("chunks", [1, 3, 16, 70])
def test_large_d_fp64_time_partition_invariance(rng, chunks):
    d, m, t = 14, 1, 281       # e.g. a periodic kernel's state dimension
    kw = random_ssm(rng, (2,), t, d, m, well=True)
    r_inv = np.array([[2.0]])
    ref = O.kf_log_likelihood(**kw, r_inv=r_inv, per_series=True)
    cst = -0.5 * np.log(2 * np.pi) * m * t + 0.5 * t * np.linalg.slogdet(r_inv)[1]
    np.testing.assert_allclose(loglik_with_chunks(kw, r_inv, chunks) + cst, ref, rtol=1e-9)


@pytest.mark.parametrize("d,m,t,batch", [(33, 1, 9, (2,)), (40, 3, 21, (1,)), (48, 17, 12, (2,)), (49, 2, 30, ()),
                                         (64, 1, 40, (3,)), (64, 32, 24, (2,)), (57, 16, 11, (1,))])
def test_panel_fp64_log_likelihood_vs_oracle(rng, d, m, t, batch):
    """fp64 for 32 < d <= 64 (the reference's default float at the spatio-temporal model's state dimension,
    models/spatio_temporal_variational.py:45-85): the panel kernels (csrc/mf_panel.hpp) on v_mfma_f64_16x16x4_f64.  rtol 1e-9 as
    for every other fp64 log-likelihood."""
    kw = random_ssm(rng, batch, t, d, m, well=True)
    cov = 0.5 * np.eye(m) + 0.1 * np.ones((m, m))
    kf = build_kf(kw, np.linalg.cholesky(cov))
    ref = O.kf_log_likelihood(**kw, r_inv=np.linalg.inv(cov))
    np.testing.assert_allclose(float(kf.log_likelihood().cpu()), ref, rtol=1e-9)


@pytest.mark.parametrize("dtype", [F32, torch.float64])
@pytest.mark.parametrize("d,chunks", [(64, 1), (64, 2), (64, 5), (64, 16), (64, 70), (40, 3), (40, 24), (48, 70)])
def test_panel_time_partition_invariance(rng, dtype, d, chunks):
    """Panel kernels: any partition of the time axis (chunks without / with a spike, one and two reduction levels: 70 chunks -> 9
    -> 2) gives the oracle's per-series value; d = 40 exercises the identity padding of a 48 x 48 problem."""
    m, t = 2, 281
    kw = random_ssm(rng, (2,), t, d, m, well=True)
    if dtype == F32:
        kw = rounded(kw)
    r_inv = np.array([[2.0, 0.3], [0.3, 1.5]])
    ref = O.kf_log_likelihood(**kw, r_inv=r_inv, per_series=True)
    cst = -0.5 * np.log(2 * np.pi) * m * t + 0.5 * t * np.linalg.slogdet(r_inv)[1]
    got = loglik_with_chunks(kw, r_inv, chunks, dtype=dtype)
    np.testing.assert_allclose(got + cst, ref, rtol=RTOL if dtype == F32 else 1e-9)


@pytest.mark.parametrize("dtype", [F32, torch.float64])
def test_panel_per_step_precisions(rng, dtype):
    """KalmanFilterWithSites (per-step R^-1, m = 1; kalman_filter.py:437-497) at d = 50 through the panel kernels."""
    d, t = 50, 30
    kw = random_ssm(rng, (), t, d, 1, well=True)
    prec = 0.5 + rng.random(size=(t, 1, 1))
    if dtype == F32:
        kw, prec = rounded(kw), prec.astype(np.float32).astype(np.float64)
    means = kw["y"]
    ssm = mfa.StateSpaceModel(*(tt(kw[k], dtype) for k in ("mu0", "chol_p0", "a_s", "b_s", "chol_q")))
    sites = mfa.UnivariateGaussianSitesNat(nat1=tt(means * prec[..., 0], dtype), nat2=tt(-0.5 * prec, dtype))
    kf = mfa.KalmanFilterWithSites(ssm, mfa.EmissionModel(tt(kw["h"], dtype)), sites)
    ref = O.kf_log_likelihood(**kw, r_inv=prec, log_det_obs_precision=np.sum(np.log(prec)))
    np.testing.assert_allclose(float(kf.log_likelihood().cpu()), ref, rtol=RTOL if dtype == F32 else 1e-9)


def test_large_d_unsupported_cases_fail_loudly(rng):
    kw = random_ssm(rng, (1,), 4, 65, 1, well=True)
    with pytest.raises(NotImplementedError):          # fp64 beyond the panel kernels
        build_kf(kw, np.eye(1)).log_likelihood()
    kw = random_ssm(rng, (1,), 4, 65, 1, well=True)
    with pytest.raises(NotImplementedError):          # beyond the LDS-tiled sizes
        build_kf(kw, np.eye(1), dtype=F32).log_likelihood()
    kw = random_ssm(rng, (1,), 4, 40, 1, well=True)
    with pytest.raises(NotImplementedError):          # the operators too are limited to d <= 32 in fp64
        build_kf(kw, np.eye(1)).posterior_state_space_model()


@pytest.mark.parametrize("dtype", [F32, torch.float64])
def test_large_d_edge_shapes(rng, dtype):
    """A chain of ONE block (T = 1, C ABI level: the Python classes need a transition) and more outputs than padded state
    columns (m = 32 with d = 10)."""
    tol = RTOL if dtype == F32 else 1e-9
    kw = random_ssm(rng, (3,), 2, 16, 2, well=True)
    one = {k: (v[:, :0] if k in ("a_s", "b_s", "chol_q") else (v[:, :1] if k in ("h", "y") else v)) for k, v in kw.items()}
    if dtype == F32:
        one = rounded(one)
    r_inv = np.array([[2.0, 0.5], [0.5, 1.0]])
    # oracle for T = 1: Gaussian evidence of y0 = H x0 + e
    ref = []
    for s in range(3):
        p0 = one["chol_p0"][s] @ one["chol_p0"][s].T
        h, y = one["h"][s, 0], one["y"][s, 0]
        cov = h @ p0 @ h.T + np.linalg.inv(r_inv)
        res = y - h @ one["mu0"][s]
        ref.append(-0.5 * (res @ np.linalg.solve(cov, res) + np.linalg.slogdet(cov)[1] + 2 * np.log(2 * np.pi)))
    cst = -0.5 * np.log(2 * np.pi) * 2 + 0.5 * np.linalg.slogdet(r_inv)[1]
    np.testing.assert_allclose(loglik_with_chunks(one, r_inv, 0, dtype=dtype) + cst, ref, rtol=tol)
    kw = random_ssm(rng, (2,), 6, 10, 32, well=True)
    if dtype == F32:
        kw = rounded(kw)
    cov = 0.5 * np.eye(32)
    ref = O.kf_log_likelihood(**kw, r_inv=np.linalg.inv(cov))
    np.testing.assert_allclose(float(build_kf(kw, np.linalg.cholesky(cov), dtype=dtype).log_likelihood().cpu()), ref, rtol=tol)


# ---- 10 <= d <= 15: the row kernels (csrc/mf_rowwide_inst.hip), as for d = 7 ... 9 ---------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float64, F32])
@pytest.mark.parametrize("d,m,t,batch", [(10, 1, 7, (2,)), (11, 3, 130, (3,)), (12, 4, 64, (2,)), (13, 2, 33, ()), (14, 3, 200, (2,)),
                                         (15, 4, 90, (3,)), (15, 1, 2, (1,)), (12, 3, 1000, (5,)),
                                         # five to eight outputs: still the row kernels in the row-only builds
                                         (10, 5, 60, (2,)), (14, 7, 130, (2,)), (15, 8, 33, (1,)), (12, 6, 200, (3,))])
def test_row_kernels_10_to_15_vs_oracle(rng, dtype, d, m, t, batch):
    """kalman_filter.py:184-255 for 10 <= d <= 15 with up to four outputs: one 16-lane row per (series, chunk); every series
    against the oracle (fp64: 1e-9; fp32 on fp32-rounded inputs: 3e-4)."""
    kw = random_ssm(rng, batch, t, d, m, well=True)
    if dtype == F32:
        kw = rounded(kw)
    cov = 0.5 * np.eye(m) + 0.1 * np.ones((m, m))
    chol_r = np.linalg.cholesky(cov).astype(np.float32).astype(np.float64)
    r_inv = np.linalg.inv(chol_r @ chol_r.T)
    kf = build_kf(kw, chol_r, dtype=dtype)
    ref = O.kf_log_likelihood(**kw, r_inv=r_inv, per_series=True)
    got = nn(kf._log_likelihood_per_series() + kf._constant_terms(t)).reshape(np.shape(ref))
    np.testing.assert_allclose(got, ref, rtol=1e-9 if dtype == torch.float64 else RTOL)


@pytest.mark.parametrize("chunks", [1, 2, 7, 24, 70, 140])
@pytest.mark.parametrize("d", [10, 13, 15])
def test_row_kernels_10_to_15_time_partition_invariance(rng, d, chunks):
    """Any partition of the time axis (up to three reduction levels: 140 chunks -> 24 -> 4) gives the same value per series."""
    m, t = 3, 281
    kw = random_ssm(rng, (2,), t, d, m, well=True)
    r_inv = np.array([[2.0, 0.3, 0.0], [0.3, 1.5, 0.1], [0.0, 0.1, 1.0]])
    ref = O.kf_log_likelihood(**kw, r_inv=r_inv, per_series=True)
    cst = -0.5 * np.log(2 * np.pi) * m * t + 0.5 * t * np.linalg.slogdet(r_inv)[1]
    got = loglik_with_chunks(kw, r_inv, chunks, dtype=torch.float64)
    np.testing.assert_allclose(got + cst, ref, rtol=1e-9)


def test_row_kernels_10_to_15_per_step_precisions_and_fallback(rng):
    """Per-step observation precisions (sites, kalman_filter.py:437-497) at d = 12 through the row kernel; five outputs at d = 12
    are within the row-only builds' eight, nine outputs are beyond them and take the LDS-tile path - same oracle, same tolerance."""
    d, t = 12, 60
    kw = random_ssm(rng, (), t, d, 1, well=True)
    prec = 0.5 + rng.random(size=(t, 1, 1))
    ssm = mfa.StateSpaceModel(*(tt(kw[k], torch.float64) for k in ("mu0", "chol_p0", "a_s", "b_s", "chol_q")))
    sites = mfa.UnivariateGaussianSitesNat(nat1=tt(kw["y"] * prec[..., 0], torch.float64), nat2=tt(-0.5 * prec, torch.float64))
    kf = mfa.KalmanFilterWithSites(ssm, mfa.EmissionModel(tt(kw["h"], torch.float64)), sites)
    ref = O.kf_log_likelihood(**kw, r_inv=prec, log_det_obs_precision=np.sum(np.log(prec)))
    np.testing.assert_allclose(float(kf.log_likelihood().cpu()), ref, rtol=1e-9)
    kw5 = random_ssm(rng, (2,), t, d, 9, well=True)
    chol_r = np.linalg.cholesky(0.5 * np.eye(9) + 0.1)
    ref5 = O.kf_log_likelihood(**kw5, r_inv=np.linalg.inv(chol_r @ chol_r.T))
    np.testing.assert_allclose(float(build_kf(kw5, chol_r, dtype=torch.float64).log_likelihood().cpu()), ref5, rtol=1e-9)


# ---- more than four outputs at d <= 9: the tile engine (its tiles pad d to 16), the reference has no such cap ----------------------------
@pytest.mark.parametrize("dtype", [torch.float64, F32])
@pytest.mark.parametrize("d,m,t,batch", [(3, 6, 40, (2,)), (9, 7, 130, (3,)), (1, 5, 9, ()), (6, 32, 20, (2,)), (5, 5, 300, (1,))])
def test_more_than_four_outputs_at_small_state_dimension(rng, dtype, d, m, t, batch):
    """kalman_filter.py:184-255 / :109-182 with output_dim > 4 and state_dim <= 9 (five independent Matern-1/2 outputs are d = m = 5):
    log-likelihood per series and the posterior chain against the oracle."""
    kw = random_ssm(rng, batch, t, d, m, well=True)
    if dtype == F32:
        kw = rounded(kw)
    cov = 0.5 * np.eye(m) + 0.1 * np.ones((m, m))
    chol_r = np.linalg.cholesky(cov).astype(np.float32).astype(np.float64)
    r_inv = np.linalg.inv(chol_r @ chol_r.T)
    kf = build_kf(kw, chol_r, dtype=dtype)
    ref = O.kf_log_likelihood(**kw, r_inv=r_inv, per_series=True)
    got = nn(kf._log_likelihood_per_series() + kf._constant_terms(t)).reshape(np.shape(ref))
    np.testing.assert_allclose(got, ref, rtol=1e-9 if dtype == torch.float64 else RTOL)
    post = kf.posterior_state_space_model()
    want = O.kf_posterior_ssm(**kw, r_inv=r_inv)
    got = (post.initial_mean, post.cholesky_initial_covariance, post.state_transitions, post.state_offsets,
           post.cholesky_process_covariances)
    tol = dict(rtol=1e-7, atol=1e-9) if dtype == torch.float64 else dict(rtol=5e-3, atol=5e-4)
    for g, w in zip(got, want):
        np.testing.assert_allclose(nn(g), w, **tol)
