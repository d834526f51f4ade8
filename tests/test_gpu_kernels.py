"""
GPU parity tests of the SDE-kernel -> state-space-model step (markovflow_amd/kernels.py, csrc/mf_sde.hip) and of the
GaussianProcessRegression harness: the device-generated tensors against the numpy oracle (oracle/numpy_kernels.py, pinned
on the reference's expm test kernels) and against the golden vectors themselves; the GPR log marginal likelihood
against a dense GP (the identity of /root/reference/tests/integration/models/test_gaussian_process_regression.py:99-105).
Tolerances: fp64 rtol 1e-10 on A, 1e-8 on chol Q (relative to the largest entry), 1e-9 on the log-likelihood.
"""
import numpy as np
import pytest
import torch

import markovflow_amd as mfa
from oracle import numpy_kernels as K
from oracle import numpy_oracle as O
from conftest import golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def tt(x, dtype=torch.float64):
    return torch.tensor(np.ascontiguousarray(x), dtype=dtype, device=DEV)


def nn(x):
    return x.detach().cpu().numpy().astype(np.float64)


KERNELS = {"m12": (mfa.Matern12, 1), "m32": (mfa.Matern32, 3), "m52": (mfa.Matern52, 5)}


@pytest.mark.parametrize("name", ["m12", "m32", "m52"])
def test_matern_state_space_model_vs_reference_fixture(name):
    cls, order = KERNELS[name]
    g = golden("kernels_matern_T24.npz")
    kern = cls(lengthscale=float(g[f"{name}_length_scale"]), variance=float(g[f"{name}_variance"]), device=DEV)
    t = tt(g["t"])
    a_s, q_s = kern.transition_statistics_from_time_points(t)
    np.testing.assert_allclose(nn(a_s), g[f"{name}_A"], rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(nn(q_s), g[f"{name}_Q"], rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(nn(kern.steady_state_covariance), g[f"{name}_Pinf"], rtol=1e-13)
    ssm = kern.state_space_model(t)
    chol = nn(ssm.cholesky_process_covariances)
    np.testing.assert_allclose(chol @ np.swapaxes(chol, -1, -2), g[f"{name}_Q"], rtol=1e-8, atol=1e-12)
    assert np.all(np.triu(chol, 1) == 0)
    p0 = nn(ssm.initial_covariance)
    np.testing.assert_allclose(p0, g[f"{name}_P0"], rtol=1e-12)
    assert tuple(kern.generate_emission_model(t).emission_matrix.shape) == (2, 24, 1, kern.state_dim)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_sum_and_independent_multi_output_vs_oracle(rng, dtype):
    orders, ls, var = [5, 1, 3, 5], [0.7, 1.3, 0.9, 1.8], [1.3, 0.4, 0.8, 1.1]
    t = np.cumsum(0.05 + rng.exponential(0.2, size=(3, 40)), axis=-1)
    parts = [KERNELS[{1: "m12", 3: "m32", 5: "m52"}[o]][0](l, v, device=DEV, dtype=dtype) for o, l, v in zip(orders, ls, var)]
    tol = dict(rtol=1e-10, atol=1e-12) if dtype == torch.float64 else dict(rtol=2e-5, atol=2e-6)
    a_ref, q_ref, p_ref = K.concat_transitions(orders, ls, var, np.diff(t, axis=-1), jitter=1e-6)
    for kern, indep in ((mfa.Sum(parts, jitter=1e-6), False), (mfa.IndependentMultiOutput(parts, jitter=1e-6), True)):
        assert kern.state_dim == 9 and kern.output_dim == (4 if indep else 1)
        ssm = kern.state_space_model(tt(t, dtype))
        np.testing.assert_allclose(nn(ssm.state_transitions), a_ref, **tol)
        chol = nn(ssm.cholesky_process_covariances)
        np.testing.assert_allclose(chol @ np.swapaxes(chol, -1, -2), q_ref, rtol=tol["rtol"] * 100, atol=tol["atol"] * 100)
        np.testing.assert_allclose(nn(ssm.initial_covariance), np.broadcast_to(p_ref + 1e-6 * np.eye(9), (3, 9, 9)), **tol)
        np.testing.assert_allclose(nn(kern.generate_emission_model(tt(t, dtype)).emission_matrix),
                                   K.emission(orders, t.shape, indep), rtol=0, atol=0)
        np.testing.assert_allclose(nn(kern.steady_state_covariance), p_ref, **tol)
    # per-series hyper-parameters (one kernel per series)
    ls_b, var_b = 0.5 + rng.random(3), 0.5 + rng.random(3)
    kern = mfa.Matern52(tt(ls_b, dtype), tt(var_b, dtype))
    a_s = nn(kern.state_transitions(None, tt(np.diff(t, axis=-1), dtype)))
    for s in range(3):
        want, _, _ = K.matern_transitions(5, ls_b[s], var_b[s], np.diff(t[s]))
        np.testing.assert_allclose(a_s[s], want, **tol)


def test_zero_time_gap_gives_zero_cholesky():
    """dt = 0: A = I, Q = 0 and the all-zero covariance passes through as a zero factor (state_space_model.py:634-656)."""
    kern = mfa.Matern32(1.0, 2.0, device=DEV)
    t = tt(np.array([[0.0, 0.5, 0.5, 1.0]]))
    ssm = kern.state_space_model(t)
    np.testing.assert_allclose(nn(ssm.state_transitions)[0, 1], np.eye(2), atol=1e-15)
    assert np.all(nn(ssm.cholesky_process_covariances)[0, 1] == 0)
    assert np.all(np.isfinite(nn(ssm.cholesky_process_covariances)))


@pytest.mark.parametrize("n", [15, 500])
def test_gpr_log_likelihood_matern32_vs_dense_gp(n):
    """BASELINE config 1: GPR (Matern-3/2, state_dim 2) on 500 1-D points against the dense log marginal likelihood."""
    g = golden(f"gpr_matern32_N{n}.npz")
    kern = mfa.Matern32(lengthscale=float(g["length_scale"]), variance=float(g["variance"]), device=DEV)
    gpr = mfa.GaussianProcessRegression((tt(g["t"]), tt(g["y"])), kern,
                                        chol_obs_covariance=tt(np.sqrt(g["noise"]) * np.eye(1)))
    # N = 500: the tool's exponential gaps go down to 7e-6, where Q_k = Pinf - A Pinf A^T has eigenvalues of 2e-16: any
    # state-space route agrees with the dense GP to ~1e-6 here (same bar as tests/test_gpu_kalman.py::test_golden_gpr_matern32
    # and tests/test_oracle_golden.py::test_gpr_log_marginal_likelihood, which feed the fixture's expm-based A, Q)
    rel = 1e-9 if n == 15 else 1e-6
    assert float(gpr.log_likelihood().cpu()) == pytest.approx(float(g["log_marginal_likelihood"]), rel=rel)
    assert float(gpr.loss().cpu()) == pytest.approx(-float(g["log_marginal_likelihood"]), rel=rel)


def test_gpr_sum_kernel_batch_vs_dense_gp(rng):
    """Sum(Matern52, Matern12, Matern32) on a batch of irregular series: state-space log-likelihood = dense GP."""
    orders, ls, var, noise = [5, 1, 3], [0.8, 1.5, 0.6], [1.0, 0.5, 0.7], 0.05
    bsz, n = 3, 120
    t = np.cumsum(0.02 + rng.exponential(0.1, size=(bsz, n)), axis=-1)
    y = rng.normal(size=(bsz, n, 1))
    kern = mfa.Matern52(ls[0], var[0], device=DEV) + mfa.Matern12(ls[1], var[1], device=DEV) + mfa.Matern32(ls[2], var[2], device=DEV)
    assert isinstance(kern, mfa.Sum) and kern.state_dim == 6
    gpr = mfa.GaussianProcessRegression((tt(t), tt(y)), kern, chol_obs_covariance=tt(np.sqrt(noise) * np.eye(1)))
    want = 0.0
    for s in range(bsz):
        kn = K.dense_kernel_matrix(orders, ls, var, t[s]) + noise * np.eye(n)
        want += -0.5 * y[s, :, 0] @ np.linalg.solve(kn, y[s, :, 0]) - 0.5 * np.linalg.slogdet(kn)[1] - 0.5 * n * np.log(2 * np.pi)
    assert float(gpr.log_likelihood().cpu()) == pytest.approx(want, rel=1e-8)
    # posterior mean of f at the training inputs = dense GP posterior mean
    post = gpr.posterior_state_space_model()
    f_mean = nn(post.marginal_means)[..., [0, 3, 4]].sum(-1)
    for s in range(bsz):
        kd = K.dense_kernel_matrix(orders, ls, var, t[s])
        np.testing.assert_allclose(f_mean[s], kd @ np.linalg.solve(kd + noise * np.eye(n), y[s, :, 0]), rtol=1e-6, atol=1e-8)


# ---- fused route: kernel -> SSM generation inside the Kalman sweep -----------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("sig", [(1,), (3,), (5,), (3, 3), (5, 3), (3, 5), (5, 5)])
def test_fused_gpr_log_likelihood_equals_materialised_route(rng, dtype, sig):
    """mf_gpr_matern_loglik (generation fused) against the materialised route (mf_sde_matern_transitions + mf_kf_loglik) and,
    through it, the oracle; several time partitions; per-series hyper-parameters."""
    cls = {1: mfa.Matern12, 3: mfa.Matern32, 5: mfa.Matern52}
    bsz, n = 4, 150
    # fp32: Q = Pinf - A Pinf A^T loses its digits at small gaps ("fp32 conditioning"): wider gaps + jitter
    f64 = dtype == torch.float64
    t = np.cumsum((0.05 if f64 else 0.4) + rng.exponential(0.1 if f64 else 0.3, size=(bsz, n)), axis=-1)
    y = rng.normal(size=(bsz, n, 1))
    ls = [tt(0.5 + rng.random(bsz), dtype) for _ in sig]
    var = [tt(0.5 + rng.random(bsz), dtype) for _ in sig]
    jit = 1e-9 if f64 else 1e-4
    parts = [cls[o](l, v, jitter=jit) for o, l, v in zip(sig, ls, var)]
    kern = parts[0] if len(parts) == 1 else mfa.Sum(parts, jitter=jit)
    chol_r = tt(np.sqrt(0.1) * np.eye(1), dtype)
    gpr = mfa.GaussianProcessRegression((tt(t, dtype), tt(y, dtype)), kern, chol_obs_covariance=chol_r)
    fused = gpr._fused_log_likelihood_per_series()
    assert fused is not None, "this signature is meant to be covered by the fused kernel"
    ref = gpr._kalman._log_likelihood_per_series() + gpr._kalman._constant_terms(n)
    tol = 1e-9 if dtype == torch.float64 else 2e-3
    np.testing.assert_allclose(nn(fused), nn(ref), rtol=tol)
    for chunks in (1, 3, 16):
        gpr._chunks = chunks
        np.testing.assert_allclose(nn(gpr._fused_log_likelihood_per_series()), nn(ref), rtol=tol)
    gpr._chunks = 0
    assert torch.isfinite(ref).all()
    assert float(gpr.log_likelihood().cpu()) == pytest.approx(float(ref.sum().cpu()), rel=tol)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("sig,multi", [((5, 5, 5), False), ((5, 5, 5), True), ((5, 3, 1, 3), False), ((5, 5, 5, 5), True),
                                       ((3, 3, 3, 3, 3, 3, 3), False), ((5, 5, 5, 5, 5), False), ((5, 3, 3), True), ((1,) * 8, False),
                                       # five to eight independent outputs (row-only builds, d >= 10)
                                       ((3,) * 5, True), ((3,) * 7, True), ((5,) * 5, True), ((1,) * 8 + (3,), False), ((3, 1) * 4, True)])
def test_fused_gpr_row_kernel_equals_materialised_route(rng, dtype, sig, multi):
    """The row form of the fused route (csrc/mf_row_gpr.hpp, 7 <= d <= 15): any concatenation of Matern components as a Sum kernel
    (one output) or as IndependentMultiOutput (one output per component; BASELINE config 4 = 3 x Matern-5/2, 3 outputs), against
    the materialised route (mf_sde_matern_transitions + mf_kf_loglik) and, through it, the oracle; several time partitions;
    per-series hyper-parameters; a full observation noise covariance for the multi-output form."""
    cls = {1: mfa.Matern12, 3: mfa.Matern32, 5: mfa.Matern52}
    bsz, n = 3, 170
    f64 = dtype == torch.float64
    t = np.cumsum((0.05 if f64 else 0.4) + rng.exponential(0.1 if f64 else 0.3, size=(bsz, n)), axis=-1)
    m = len(sig) if multi else 1
    y = rng.normal(size=(bsz, n, m))
    jit = 1e-9 if f64 else 1e-4
    parts = [cls[o](tt(0.5 + rng.random(bsz), dtype), tt(0.5 + rng.random(bsz), dtype), jitter=jit) for o in sig]
    kern = mfa.IndependentMultiOutput(parts, jitter=jit) if multi else mfa.Sum(parts, jitter=jit)
    assert 7 <= kern.state_dim <= 15
    cov = 0.1 * np.eye(m) + 0.02 * np.ones((m, m))
    gpr = mfa.GaussianProcessRegression((tt(t, dtype), tt(y, dtype)), kern, chol_obs_covariance=tt(np.linalg.cholesky(cov), dtype))
    fused = gpr._fused_log_likelihood_per_series()
    if not f64 and kern.state_dim < 9:
        # fp32 keeps the register kernels up to d = 8 (csrc/mf_inst.hip: row_path): no row form, the model is materialised
        assert fused is None and torch.isfinite(gpr.log_likelihood())
        return
    assert fused is not None, "this signature is meant to be covered by the row form of the fused kernel"
    ref = gpr._kalman._log_likelihood_per_series() + gpr._kalman._constant_terms(n)
    tol = 1e-9 if f64 else 2e-3
    np.testing.assert_allclose(nn(fused), nn(ref), rtol=tol)
    for chunks in (1, 3, 16):
        gpr._chunks = chunks
        np.testing.assert_allclose(nn(gpr._fused_log_likelihood_per_series()), nn(ref), rtol=tol)
    gpr._chunks = 0
    assert float(gpr.log_likelihood().cpu()) == pytest.approx(float(ref.sum().cpu()), rel=tol)


def test_fused_gpr_falls_back_when_not_covered(rng):
    """Three components below d = 7 (the register kernels generate one or two) are not fused: log_likelihood takes the
    materialised route (still all HIP)."""
    t = np.cumsum(0.1 + rng.random(size=(2, 30)), axis=-1)
    kern = mfa.Sum([mfa.Matern12(1.0, 1.0, device=DEV), mfa.Matern32(1.0, 1.0, device=DEV), mfa.Matern52(1.0, 1.0, device=DEV)])
    assert kern.state_dim == 6
    gpr = mfa.GaussianProcessRegression((tt(t), tt(rng.normal(size=(2, 30, 1)))), kern, chol_obs_covariance=tt(0.3 * np.eye(1)))
    assert gpr._fused_log_likelihood_per_series() is None
    assert torch.isfinite(gpr.log_likelihood())


# ---- prediction at new time points (posterior.predict_f) ------------------------------------------------------------------------------
@pytest.mark.parametrize("sig", [(3,), (5, 1), (5, 5), (1, 3, 5), (5, 5, 5, 5), (3, 3, 3, 3, 3)])
def test_posterior_predict_f_vs_dense_gp(rng, sig):
    """GaussianProcessRegression.posterior.predict_f / predict_y at new points - before, between, ON and after the training
    points - against the dense GP predictive distribution (markovflow/posterior.py:231-258, conditionals.py:29-83)."""
    cls = {1: mfa.Matern12, 3: mfa.Matern32, 5: mfa.Matern52}
    bsz, n, n_new, noise = 2, 60, 45, 0.05
    ls, var = [0.6 + 0.5 * j for j in range(len(sig))], [1.0 + 0.3 * j for j in range(len(sig))]
    t = np.cumsum(0.05 + rng.exponential(0.15, size=(bsz, n)), axis=-1)
    y = rng.normal(size=(bsz, n, 1))
    t_new = np.sort(np.concatenate([t[:, :1] - rng.random((bsz, 5)) * 2.0, t[:, -1:] + rng.random((bsz, 5)) * 2.0,
                                    t[:, 3:8], t[:, :1] + rng.random((bsz, 30)) * (t[:, -1:] - t[:, :1])], axis=-1), axis=-1)
    parts = [cls[o](l, v, device=DEV) for o, l, v in zip(sig, ls, var)]
    kern = parts[0] if len(parts) == 1 else mfa.Sum(parts, jitter=1e-10)
    chol_r = tt(np.sqrt(noise) * np.eye(1))
    gpr = mfa.GaussianProcessRegression((tt(t), tt(y)), kern, chol_obs_covariance=chol_r)
    post = gpr.posterior
    f_mean, f_var = post.predict_f(tt(t_new))
    y_mean, y_var = post.predict_y(tt(t_new))
    assert tuple(f_mean.shape) == (bsz, n_new, 1) and tuple(f_var.shape) == (bsz, n_new, 1)
    for s in range(bsz):
        mean, v = K.dense_gp_predict(sig, ls, var, t[s], y[s, :, 0], noise, t_new[s])
        np.testing.assert_allclose(nn(f_mean)[s, :, 0], mean, rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(nn(f_var)[s, :, 0], v, rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(nn(y_var)[s, :, 0], v + noise, rtol=1e-5, atol=1e-7)
    # state covariances are symmetric positive definite, full_output_cov has the output-by-output shape
    _, cov = post.predict_state(tt(t_new))
    np.testing.assert_allclose(nn(cov), np.swapaxes(nn(cov), -1, -2), atol=1e-10)
    assert np.all(np.linalg.eigvalsh(nn(cov)) > -1e-9)
    _, full = post.predict_f(tt(t_new), full_output_cov=True)
    assert tuple(full.shape) == (bsz, n_new, 1, 1)


def test_fused_gpr_follows_hyper_parameter_updates(rng):
    """The fused route keeps its hyper-parameter tensors between calls: an in-place update (an optimiser step) or a replaced
    tensor must be seen by the next call."""
    bsz, n = 2, 80
    t = np.cumsum(0.05 + rng.exponential(0.1, size=(bsz, n)), axis=-1)
    y = rng.normal(size=(bsz, n, 3))
    parts = [mfa.Matern52(tt(0.5 + rng.random(bsz)), tt(0.5 + rng.random(bsz)), jitter=1e-9) for _ in range(3)]
    kern = mfa.IndependentMultiOutput(parts, jitter=1e-9)
    chol = tt(np.sqrt(0.1) * np.eye(3))
    gpr = mfa.GaussianProcessRegression((tt(t), tt(y)), kern, chol_obs_covariance=chol)

    def both():
        gpr.fused = True
        a = float(gpr.log_likelihood().cpu())
        gpr.fused = False
        return a, float(gpr.log_likelihood().cpu())
    a0, b0 = both()
    assert a0 == pytest.approx(b0, rel=1e-10)
    parts[1]._lengthscale_t.mul_(1.7)                    # in place
    a1, b1 = both()
    assert a1 == pytest.approx(b1, rel=1e-10) and abs(a1 - a0) > 1e-6 * abs(a0)
    parts[2]._variance_t = parts[2]._variance_t * 0.5    # replaced
    a2, b2 = both()
    assert a2 == pytest.approx(b2, rel=1e-10) and abs(a2 - a1) > 1e-6 * abs(a1)
    chol.mul_(1.3)                                       # the noise factor, in place
    a3, b3 = both()
    assert a3 == pytest.approx(b3, rel=1e-10) and abs(a3 - a2) > 1e-6 * abs(a2)


@pytest.mark.parametrize("per_series", [True, False])
@pytest.mark.parametrize("sig", [(5, 3, 1), (5, 5, 5), (3,), (1, 1)])
def test_generator_backward_equals_autograd_through_the_closed_forms(rng, sig, per_series):
    """mf_sde_matern_transitions_grad_* (the closed forms of matern.py / sde_kernel.py:421-446 + Cholesky in forward mode, one kernel)
    against torch autograd through the same closed forms (kernels._torch_transitions): gradients of a random linear functional of
    (A, chol Q) with respect to every lengthscale and variance, per-series and shared hyper-parameters."""
    cls = {1: mfa.Matern12, 3: mfa.Matern32, 5: mfa.Matern52}
    bsz, n = 3, 40
    dts = tt(0.05 + rng.exponential(0.2, size=(bsz, n)))

    def leaves():
        shape = (bsz,) if per_series else ()
        return ([tt(0.5 + rng.random(shape)).requires_grad_(True) for _ in sig], [tt(0.5 + rng.random(shape)).requires_grad_(True) for _ in sig])
    ls, var = leaves()
    kern_parts = [cls[o](l, v, jitter=1e-8) for o, l, v in zip(sig, ls, var)]
    kern = kern_parts[0] if len(sig) == 1 else mfa.Sum(kern_parts, jitter=1e-8)
    d = kern.state_dim
    w_a, w_c = tt(rng.normal(size=(bsz, n, d, d))), torch.tril(tt(rng.normal(size=(bsz, n, d, d))))
    a_s, chol, _ = kern._device_transitions(dts, True, False)               # HIP forward + backward
    (torch.sum(w_a * a_s) + torch.sum(w_c * chol)).backward()
    got = [x.grad.clone() for x in ls + var]
    for x in ls + var:
        x.grad = None
    a_t, chol_t, _ = kern._torch_transitions(dts, True, False)              # differentiable torch ops
    np.testing.assert_allclose(nn(a_s), nn(a_t), rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(nn(chol), nn(chol_t), rtol=1e-9, atol=1e-12)
    (torch.sum(w_a * a_t) + torch.sum(w_c * chol_t)).backward()
    for g, x in zip(got, ls + var):
        np.testing.assert_allclose(nn(g), nn(x.grad), rtol=1e-8, atol=1e-10)


def test_kernel_under_inference_mode_on_the_device():
    """kernels are built and evaluated under torch.inference_mode() (a common wrapper for prediction): the version-keyed
    caches step aside for inference tensors; the transitions equal the ones formed outside it."""
    t = torch.cumsum(0.1 + torch.rand(2, 50, dtype=torch.float64, device="cuda:0"), dim=-1)
    ls, var = torch.tensor([0.7, 1.3], dtype=torch.float64, device="cuda:0"), torch.tensor([1.1, 0.6], dtype=torch.float64, device="cuda:0")
    ref = mfa.Matern52(ls, var).state_space_model(t)
    with torch.inference_mode():
        got = mfa.Matern52(ls.clone(), var.clone()).state_space_model(t)
        assert torch.equal(got.state_transitions, ref.state_transitions)
        assert torch.equal(got.cholesky_process_covariances, ref.cholesky_process_covariances)
