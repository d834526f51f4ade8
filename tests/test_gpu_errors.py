"""
The non-positive-pivot channel is deterministic: TensorFlow's Cholesky raises inside the op
(/root/reference/markovflow/block_tri_diag.py:423-436); here a kernel raises ONE int in device memory and the host learns of
it through a 4-byte copy queued on the same stream right behind the kernel (markovflow_amd/_lib.py) - after any
synchronisation with that stream the answer is final: no failure is missed, none is attributed to a later call.  No sleeps,
no retries.
"""
import numpy as np
import pytest
import torch

import markovflow_amd as mfa
from markovflow_amd import _lib, synthetic
from test_gpu_kalman import DEV, build_kf, random_ssm, tt

pytestmark = pytest.mark.gpu


def _spd_and_not(rng, n=12, d=3):
    good = np.tile(4.0 * np.eye(d), (2, n, 1, 1)) + 0.1 * rng.normal(size=(2, n, d, d))
    good = good + np.swapaxes(good, -1, -2)
    bad = good.copy()
    bad[1, n // 2] = -np.eye(d)
    sub = 0.1 * rng.normal(size=(2, n - 1, d, d))
    return good, bad, sub


def test_failure_injection_500_times_no_stale_and_no_missed_flag(rng):
    """Alternate factorisations of a positive definite and of an indefinite matrix, 500 times: every bad one is reported by the
    next check_errors(), no good one ever is - also when the good one is issued right after a bad one was reported."""
    good, bad, sub = _spd_and_not(rng)
    g, b, s = tt(good), tt(bad), tt(sub)
    _lib.check_errors()
    missed = stale = 0
    for i in range(500):
        mfa.SymmetricBlockTriDiagonal(b, s).cholesky
        try:
            _lib.check_errors()
            missed += 1
        except mfa.MarkovflowAmdError as exc:
            assert "cholesky" in str(exc)
        mfa.SymmetricBlockTriDiagonal(g, s).cholesky
        try:
            _lib.check_errors()
        except mfa.MarkovflowAmdError:
            stale += 1
    assert (missed, stale) == (0, 0)


def test_host_read_of_a_result_raises_every_time_and_only_then(rng):
    """`float(kf.log_likelihood())`: 200 alternations of a model with a singular process covariance and a sound one, no
    synchronisation but the host read itself."""
    inp = synthetic.make_ssm(3, 40, (3, 3), dtype=torch.float64, device=DEV)
    bad = dict(inp)
    bad["cholQ"] = inp["cholQ"].clone()
    bad["cholQ"][1, 17] = 0.0
    kf_good, kf_bad = synthetic.kalman_filter_from(inp), synthetic.kalman_filter_from(bad)
    _lib.check_errors()
    for i in range(200):
        with pytest.raises(mfa.MarkovflowAmdError, match="log_likelihood"):
            float(kf_bad.log_likelihood())
        assert np.isfinite(float(kf_good.log_likelihood()))
    _lib.check_errors()


def test_flags_are_per_stream(rng):
    """A failing factorisation on one stream while another stream computes sound results: the failure is reported exactly
    once, naming the operation - by the first library call that finds its flag copy landed, by the host read of ANY result
    (which waits for every stream that holds a flag) or by check_errors() at the latest - and afterwards everything is clean."""
    good, bad, sub = _spd_and_not(rng)
    g, b, s = tt(good), tt(bad), tt(sub)
    inp = synthetic.make_ssm(3, 40, (3, 3), dtype=torch.float64, device=DEV)
    kf = synthetic.kalman_filter_from(inp)
    torch.cuda.synchronize()
    _lib.check_errors()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for rep in range(50):
        reports = 0
        with torch.cuda.stream(s1):
            mfa.SymmetricBlockTriDiagonal(b, s).cholesky
        try:
            with torch.cuda.stream(s2):
                value = float(kf.log_likelihood())        # a sound result on s2; its host read waits for s1's flag too
            pytest.fail("the failure on the other stream was lost")
        except mfa.MarkovflowAmdError as exc:
            assert "cholesky" in str(exc)
            reports += 1
        assert len({k[1] for k in _lib._flags}) >= 2      # one flag per stream
        with torch.cuda.stream(s2):                       # reported once: from here on clean, on both streams
            assert np.isfinite(float(kf.log_likelihood()))
            mfa.SymmetricBlockTriDiagonal(g, s).cholesky
        with torch.cuda.stream(s1):
            mfa.SymmetricBlockTriDiagonal(g, s).cholesky
        _lib.check_errors()
        assert reports == 1


def test_torch_factorisations_on_the_path_report_through_the_same_channel(rng):
    """the d > 9 prediction route and the kernels' initial covariance factor with torch; a matrix that is not positive
    definite must raise there as it does in the HIP kernels (d <= 9), not return silent NaNs."""
    _lib.check_errors()
    good = tt(np.tile(np.eye(3), (4, 1, 1)))
    bad = good.clone()
    bad[2] = -bad[2]
    assert bool(torch.isfinite(_lib.checked_cholesky(good, "stand-in")).all())
    _lib.check_errors()
    _lib.checked_cholesky(bad, "ConditionalProcess.predict_state")
    with pytest.raises(mfa.MarkovflowAmdError, match="predict_state"):
        _lib.check_errors()
    _lib.check_errors()


def test_fused_gpr_cache_and_writes_that_bypass_the_version_counter():
    """The fused GPR route keeps its derived hyper-parameter tensors until a source is replaced or written in place through torch;
    a `.data` write is invisible to that check and needs invalidate_hyperparameter_cache() (documented)."""
    t = torch.cumsum(0.1 + 0.1 * torch.rand(3, 50, dtype=torch.float64, device=DEV), dim=-1)
    y = torch.randn(3, 50, 1, dtype=torch.float64, device=DEV)
    ls = torch.full((3,), 0.8, dtype=torch.float64, device=DEV)
    noise = 0.3 * torch.eye(1, dtype=torch.float64, device=DEV)
    gpr = mfa.GaussianProcessRegression((t, y), mfa.Matern52(ls, 1.1, jitter=1e-9), chol_obs_covariance=noise)
    v0 = float(gpr.log_likelihood())
    ls.mul_(1.5)                                               # through torch: seen (version counter)
    v1 = float(gpr.log_likelihood())
    fresh = mfa.GaussianProcessRegression((t, y), mfa.Matern52(ls.clone(), 1.1, jitter=1e-9), chol_obs_covariance=noise)
    assert v1 != v0 and v1 == pytest.approx(float(fresh.log_likelihood()), rel=1e-12)
    ls.data.mul_(0.5)                                          # bypasses the counter
    gpr.invalidate_hyperparameter_cache()
    fresh = mfa.GaussianProcessRegression((t, y), mfa.Matern52(ls.clone(), 1.1, jitter=1e-9), chol_obs_covariance=noise)
    assert float(gpr.log_likelihood()) == pytest.approx(float(fresh.log_likelihood()), rel=1e-12)


def test_non_positive_pivot_is_reported_like_the_reference(rng):
    """SURVEY 8(b) / VERDICT r05 item 9: the `info` word names the FIRST failing block, LAPACK style (`info = 1 + flat index`,
    flat index = series * blocks + block).  (a) C ABI: `mf_btd_cholesky_f64` on three series of which series 2 loses positive
    definiteness at block 4 and series 1 at block 5 -> the SMALLEST flat index wins: 1 * 7 + 5;
    (b) the Python classes raise MarkovflowAmdError carrying series / block; (c) the fused log-likelihood names the block whose
    elimination step failed."""
    from markovflow_amd import _lib
    bsz, n, d = 3, 7, 4
    a = rng.normal(size=(bsz, n, d, d))
    diag = a @ a.transpose(0, 1, 3, 2) + 4 * np.eye(d)
    sub = 0.1 * rng.normal(size=(bsz, n - 1, d, d))
    diag[2, 4] -= 50 * np.eye(d)
    diag[1, 5] -= 50 * np.eye(d)
    dg, sb = torch.tensor(diag, device=DEV), torch.tensor(sub, device=DEV)
    lib = _lib.load()
    info = _lib.new_info(torch.device(DEV))
    ld, ls = torch.empty_like(dg), torch.empty_like(sb)
    rc = lib.mf_btd_cholesky_f64(bsz, n, d, _lib.ptr(dg), _lib.ptr(sb), _lib.ptr(ld), _lib.ptr(ls), None, 0, _lib.ptr(info),
                                 _lib.stream_ptr(torch.device(DEV)))
    assert rc == 0
    word = int(info.item())
    assert word >= 2 and int(lib.mf_info_flat_index(word)) == 1 * n + 5
    assert int(lib.mf_info_flat_index(0)) == -1 and int(lib.mf_info_flat_index(1)) == -1
    _lib.set_synchronous_checks(True)
    try:
        with pytest.raises(mfa.MarkovflowAmdError) as exc:
            mfa.SymmetricBlockTriDiagonal(dg, sb).cholesky
        assert exc.value.flat_index == 1 * n + 5 and exc.value.series == 1 and exc.value.block == 5
        assert "series 1, block 5" in str(exc.value)
        # the fused log-likelihood: a process covariance factor with a zero on its diagonal at transition 3 of series 0
        kw = random_ssm(rng, (2,), 9, 3, 1, well=True)
        kw["chol_q"][0, 3, 1, 1] = 0.0
        with pytest.raises(mfa.MarkovflowAmdError) as exc:
            build_kf(kw, np.eye(1)).log_likelihood()
        assert exc.value.series == 0 and exc.value.block in (3, 4)
    finally:
        _lib.set_synchronous_checks(False)
