"""Differential test: the C restatement (timed CPU baseline) against the numpy oracle and the golden vectors."""
import numpy as np
import pytest

from oracle import c_oracle as C
from oracle import numpy_oracle as O
from conftest import golden


def random_ssm(rng, bsz, t, d, m):
    a = 0.5 * rng.normal(size=(bsz, t - 1, d, d)) / np.sqrt(d)
    cq = np.tril(0.3 * rng.normal(size=(bsz, t - 1, d, d))) + np.eye(d)
    cp0 = np.tril(0.3 * rng.normal(size=(bsz, d, d))) + np.eye(d)
    return dict(mu0=rng.normal(size=(bsz, d)), chol_p0=cp0, a_s=a, b_s=0.3 * rng.normal(size=(bsz, t - 1, d)),
                chol_q=cq, h=rng.normal(size=(bsz, t, m, d)), y=rng.normal(size=(bsz, t, m)))


@pytest.mark.parametrize("d,m,t", [(1, 1, 1), (2, 1, 5), (3, 2, 8), (6, 1, 40), (9, 3, 17)])
def test_c_loglik_matches_numpy_oracle(rng, d, m, t):
    kw = random_ssm(rng, 5, t, d, m)
    r = rng.normal(size=(m, m)); r_inv = np.linalg.inv(r @ r.T + np.eye(m))
    np.testing.assert_allclose(C.kf_loglik(**kw, r_inv=r_inv),
                               O.kf_log_likelihood(**kw, r_inv=r_inv, per_series=True), rtol=1e-10)


def test_c_loglik_per_step_precisions(rng):
    kw = random_ssm(rng, 3, 11, 4, 1)
    r_inv = rng.uniform(0.5, 2.0, size=(3, 11, 1, 1))
    ref = O.kf_log_likelihood(**kw, r_inv=r_inv, per_series=True, log_det_obs_precision=0.0)
    np.testing.assert_allclose(C.kf_loglik(**kw, r_inv=r_inv, per_step=True), ref, rtol=1e-10)


def test_c_loglik_golden_reference_fixture():
    g = golden("kf_T8_d3_m2_b3.npz")
    y = g["y"]; bsz, t = y.shape[0], y.shape[1]
    out = C.kf_loglik(np.tile(g["mu0"], (bsz, 1)), np.tile(g["cholP0"], (bsz, 1, 1)),
                      np.tile(g["A"], (bsz, t - 1, 1, 1)), np.tile(g["b"], (bsz, t - 1, 1)),
                      np.tile(g["cholQ"], (bsz, t - 1, 1, 1)), np.tile(g["H"], (bsz, t, 1, 1)), y,
                      np.linalg.inv(g["R"]))
    np.testing.assert_allclose(out, g["log_liks"].sum(-1), rtol=1e-9)


@pytest.mark.parametrize("name", ["btd_d3_T4_sub1", "btd_d6_T64_sub1", "btd_d3_T4_sub0"])
def test_c_cholesky_solve(name):
    g = golden(name + ".npz")
    sub = g["sub"] if bool(g["has_sub"]) else None
    ld, ls = C.btd_cholesky(g["diag"], sub)
    np.testing.assert_allclose(ld, g["chol_diag"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(C.btd_solve(ld, ls, g["rhs"]), g["solve_l"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(C.btd_solve(ld, ls, g["rhs"], transpose=True), g["solve_lt"], rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("d", [4, 6])
def test_c_loglik_fixed_size_instance_equals_generic(rng, d):
    """d = 6 / d = 4 with one output run a compile-time-sized instance (the timed CPU baseline): same numbers as the generic one."""
    kw = random_ssm(rng, 4, 50, d, 1)
    r_inv = np.array([[2.5]])
    np.testing.assert_allclose(C.kf_loglik(**kw, r_inv=r_inv), C.kf_loglik(**kw, r_inv=r_inv, generic=True), rtol=1e-12)
    np.testing.assert_allclose(C.kf_loglik(**kw, r_inv=r_inv), O.kf_log_likelihood(**kw, r_inv=r_inv, per_series=True), rtol=1e-10)
