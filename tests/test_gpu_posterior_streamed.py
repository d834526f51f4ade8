"""
GPU parity tests of the streamed, time-partitioned posterior chain (csrc/mf_post_lds.hpp: `mf_kf_posterior_chain_*` with a
workspace) - BaseKalmanFilter.posterior_state_space_model (/root/reference/markovflow/kalman_filter.py:109-182,
block_tri_diag.py:438-545) for few, long series.  Through the C ABI, all five tensors of the posterior chain, against
the numpy restatement (oracle.numpy_oracle.kf_posterior_ssm); fp64 rtol 1e-8 as for the other two routes
(tests/test_gpu_kalman.py), fp32 2e-3 on well-conditioned chains.  The same arithmetic runs on the CPU, lane by lane, in
tests/test_post_host_sim.py.
"""
import numpy as np
import pytest
import torch

import markovflow_amd as mfa
from markovflow_amd import _lib
from oracle import numpy_oracle as O
from test_gpu_kalman import DEV, build_kf, nn, random_ssm, tt

pytestmark = pytest.mark.gpu


def posterior_chain_abi(kw, r_inv, chunks, dtype=torch.float64, per_step=False, streamed=True):
    """Call mf_kf_posterior_chain directly; returns the five tensors as numpy arrays."""
    mu0, cp0, a, b, cq, h, y = (tt(kw[k], dtype) for k in ("mu0", "chol_p0", "a_s", "b_s", "chol_q", "h", "y"))
    bsz, t, m, d = h.shape
    lib = _lib.load()
    esz = 8 if dtype == torch.float64 else 4
    ws, wsb = None, 0
    if streamed:
        wsb = int(lib.mf_kf_posterior_chain_workspace_bytes(bsz, t, d, m, int(per_step), esz, chunks))
        assert wsb > 0, "the streamed kernels should cover this call"
        ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    ap, bp, cqp, mp, cpp = (torch.full_like(x, float("nan")) for x in (a, b, cq, mu0, cp0))
    info = _lib.new_info(torch.device(DEV))
    ri = tt(r_inv, dtype)
    _lib.call("mf_kf_posterior_chain", dtype, bsz, t, d, m, _lib.ptr(mu0), _lib.ptr(cp0), _lib.ptr(a), _lib.ptr(b), _lib.ptr(cq),
              _lib.ptr(h), _lib.ptr(y), _lib.ptr(ri), int(per_step), _lib.ptr(ap), _lib.ptr(mp), _lib.ptr(bp), _lib.ptr(cpp),
              _lib.ptr(cqp), _lib.ptr(ws), wsb, _lib.ptr(info), chunks, None, None, _lib.stream_ptr(torch.device(DEV)))
    torch.cuda.synchronize()
    assert int(info.item()) == 0
    return tuple(nn(x) for x in (mp, cpp, ap, bp, cqp))


def assert_chain(got, want, rtol, atol):
    for name, g, w in zip(("mu0", "cholP0", "A", "b", "cholQ"), got, want):
        assert np.all(np.isfinite(g)), name
        np.testing.assert_allclose(g, w, rtol=rtol, atol=atol, err_msg=name)


# chunks: 0 = automatic (chains of >= 8 transitions are cut into chunks of >= 4); explicit counts exercise ragged last chunks,
# one chunk per series (emit pass alone), more than 64 chunks per series (the scan's folded runs) and waves that hold several series
@pytest.mark.parametrize("d,m,t,bsz,chunks,per_step", [
    (6, 1, 100, 3, 0, False), (6, 1, 101, 3, 7, False), (6, 1, 64, 2, 1, False), (6, 1, 300, 2, 100, False),
    (6, 1, 1000, 1, 0, False), (6, 2, 90, 3, 0, False), (6, 3, 70, 2, 5, False), (6, 1, 57, 3, 4, True),
    (4, 1, 200, 5, 0, False), (4, 2, 37, 2, 3, False), (4, 1, 37, 2, 6, True), (2, 1, 50, 70, 0, False),
    (2, 2, 33, 3, 4, False), (1, 1, 40, 3, 5, False), (3, 1, 300, 2, 0, False), (5, 1, 129, 4, 11, False),
    (5, 3, 45, 2, 2, False), (3, 2, 20, 130, 2, False), (6, 1, 2, 3, 0, False), (6, 1, 3, 70, 2, False),
])
def test_streamed_posterior_chain_against_the_oracle(rng, d, m, t, bsz, chunks, per_step):
    kw = random_ssm(rng, (bsz,), t, d, m, well=True)
    if per_step:
        r_inv = rng.uniform(0.5, 2.0, size=(bsz, t, m, m))
    else:
        r = rng.normal(size=(m, m))
        r_inv = np.linalg.inv(r @ r.T + np.eye(m))
    want = O.kf_posterior_ssm(**kw, r_inv=r_inv)
    got = posterior_chain_abi(kw, r_inv, chunks, per_step=per_step)
    assert_chain(got, want, rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("d,m,t,bsz,chunks", [(6, 1, 200, 3, 0), (4, 2, 100, 4, 5), (2, 1, 64, 5, 3)])
def test_streamed_posterior_chain_fp32(rng, d, m, t, bsz, chunks):
    kw = random_ssm(rng, (bsz,), t, d, m, well=True)
    kw = {k: v.astype(np.float32).astype(np.float64) for k, v in kw.items()}
    r_inv = np.eye(m) * 2.0
    want = O.kf_posterior_ssm(**kw, r_inv=r_inv)
    got = posterior_chain_abi(kw, r_inv, chunks, dtype=torch.float32)
    assert_chain(got, want, rtol=2e-3, atol=2e-4)


def test_streamed_route_is_the_one_the_class_takes_for_few_long_series(rng, monkeypatch):
    """KalmanFilter.posterior_state_space_model: B = 4, T = 500 -> the streamed kernels; identical (to rounding) to the
    lane-per-series sweep and to the operator route, and the posterior's marginals agree with the oracle's."""
    kw = random_ssm(rng, (4,), 500, 6, 1, well=True)
    kf = build_kf(kw, np.array([[0.7]]))
    seen = []
    real_call = _lib.call

    def spy(name, *args):
        if name == "mf_kf_posterior_chain":
            seen.append(args[-7] is not None and args[-6] > 0)      # ws, ws_bytes
        return real_call(name, *args)

    monkeypatch.setattr(_lib, "call", spy)
    post = kf.posterior_state_space_model()
    assert seen == [True]
    want = O.kf_posterior_ssm(**kw, r_inv=np.array([[1.0 / 0.49]]))
    got = (post.initial_mean, post.cholesky_initial_covariance, post.state_transitions, post.state_offsets,
           post.cholesky_process_covariances)
    assert_chain(tuple(nn(g) for g in got), want, rtol=1e-8, atol=1e-10)
    monkeypatch.setattr(mfa.BaseKalmanFilter, "_POST_STREAMED", False)
    ops = kf.posterior_state_space_model()
    np.testing.assert_allclose(nn(post.marginal_means), nn(ops.marginal_means), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(nn(post.marginal_covariances), nn(ops.marginal_covariances), rtol=1e-9, atol=1e-11)


def test_streamed_posterior_chain_reports_a_matrix_that_is_not_positive_definite(rng):
    kw = random_ssm(rng, (3,), 120, 6, 1, well=True)
    kw["chol_q"][1, 60] = 0.0                 # a singular process covariance: Q^-1 does not exist
    kf = build_kf(kw, np.array([[0.7]]))
    mfa.set_synchronous_checks(True)
    try:
        with pytest.raises(mfa.MarkovflowAmdError):
            kf.posterior_state_space_model()
    finally:
        mfa.set_synchronous_checks(False)


def test_full_size_streamed_posterior_chain():
    """The north-star shape (B=1024, T=10000, d=6, m=1, fp64; SURVEY section 8d's Matern-5/2 + Matern-5/2 chains): every series against
    the lane-per-series sweep (another decomposition of the same recursion: serial over the whole chain) and 24 full-length
    series against the numpy oracle, all five tensors."""
    from markovflow_amd import synthetic
    inp = synthetic.make_ssm(1024, 10000, (5, 5), dtype=torch.float64, device=DEV)
    kf = synthetic.kalman_filter_from(inp)
    post = kf.posterior_state_space_model()
    got = (post.initial_mean, post.cholesky_initial_covariance, post.state_transitions, post.state_offsets,
           post.cholesky_process_covariances)
    for g in got:
        assert bool(torch.isfinite(g).all())
    saved = mfa.BaseKalmanFilter._POST_FUSED_MIN_SERIES
    try:
        mfa.BaseKalmanFilter._POST_FUSED_MIN_SERIES = 1          # one lane per series, no time partition
        serial = kf.posterior_state_space_model()
    finally:
        mfa.BaseKalmanFilter._POST_FUSED_MIN_SERIES = saved
    want = (serial.initial_mean, serial.cholesky_initial_covariance, serial.state_transitions, serial.state_offsets,
            serial.cholesky_process_covariances)
    # (two decompositions of a 10^4-step recursion on chains whose process covariances reach the 1e-9 jitter: 1.2e-8 of the
    # tensor's scale between them was measured; each is within 1e-8 of the oracle below)
    for name, g, w in zip(("mu0", "cholP0", "A", "b", "cholQ"), got, want):
        scale = float(w.abs().max())
        assert float((g - w).abs().max()) <= 1e-7 * scale, name
    pick = np.r_[0:8, 500:508, 1016:1024]
    kw = {k2: nn(inp[k1][pick]) for k1, k2 in (("mu0", "mu0"), ("cholP0", "chol_p0"), ("A", "a_s"), ("b", "b_s"),
                                               ("cholQ", "chol_q"), ("H", "h"), ("y", "y"))}
    ref = O.kf_posterior_ssm(**kw, r_inv=np.array([[1.0 / 0.1]]))
    for name, g, w in zip(("mu0", "cholP0", "A", "b", "cholQ"), got, ref):
        gg = nn(g[pick])
        assert np.abs(gg - w).max() <= 1e-8 * np.abs(w).max(), name


# ---- the smoother after the filter: posterior_state_space_model() from the summaries log_likelihood() left behind -----------------
@pytest.mark.parametrize("d,m,t,bsz,chunks", [(6, 1, 500, 4, 0), (6, 1, 301, 3, 13), (4, 2, 200, 5, 0), (2, 1, 150, 70, 0),
                                              (6, 3, 90, 2, 6), (5, 1, 1000, 1, 0), (3, 1, 70, 3, 2)])
def test_posterior_from_the_filters_summaries(rng, monkeypatch, d, m, t, bsz, chunks):
    kw = random_ssm(rng, (bsz,), t, d, m, well=True)
    r = rng.normal(size=(m, m))
    chol_r = np.linalg.cholesky(r @ r.T + np.eye(m))
    kf = build_kf(kw, chol_r)
    kf._chunks = chunks
    seen = []
    real = _lib.call_rc

    def spy(name, *args):
        seen.append(name)
        return real(name, *args)

    monkeypatch.setattr(_lib, "call_rc", spy)
    kf.log_likelihood()
    post = kf.posterior_state_space_model()
    assert "mf_kf_posterior_chain_from_filter" in seen
    want = O.kf_posterior_ssm(**kw, r_inv=np.linalg.inv(chol_r @ chol_r.T))
    got = (post.initial_mean, post.cholesky_initial_covariance, post.state_transitions, post.state_offsets,
           post.cholesky_process_covariances)
    assert_chain(tuple(nn(g) for g in got), want, rtol=1e-8, atol=1e-10)


def test_filter_cache_is_dropped_when_an_input_changes(rng, monkeypatch):
    """An in-place write to any input (torch bumps its version counter) invalidates the summaries: the next posterior runs its own
    passes and is the posterior of the NEW inputs."""
    kw = random_ssm(rng, (3,), 400, 6, 1, well=True)
    kf = build_kf(kw, np.array([[0.7]]))
    kf.log_likelihood()
    kf.observations.mul_(1.5)
    seen = []
    real = _lib.call_rc

    def spy(name, *args):
        seen.append(name)
        return real(name, *args)

    monkeypatch.setattr(_lib, "call_rc", spy)
    post = kf.posterior_state_space_model()
    assert "mf_kf_posterior_chain_from_filter" not in seen
    kw2 = dict(kw)
    kw2["y"] = kw["y"] * 1.5
    want = O.kf_posterior_ssm(**kw2, r_inv=np.array([[1.0 / 0.49]]))
    got = (post.initial_mean, post.cholesky_initial_covariance, post.state_transitions, post.state_offsets,
           post.cholesky_process_covariances)
    assert_chain(tuple(nn(g) for g in got), want, rtol=1e-8, atol=1e-10)


def _posterior_tuple(post):
    return tuple(nn(g) for g in (post.initial_mean, post.cholesky_initial_covariance, post.state_transitions, post.state_offsets,
                                 post.cholesky_process_covariances))


def test_filter_cache_with_an_emission_matrix_broadcast_over_the_batch(rng):
    """the emission matrix ``[T, m, d]`` is SHARED by the batch, so the kernels see an expanded temporary
    of it.  An in-place write to the shared tensor between log_likelihood() and posterior_state_space_model() must give the
    posterior of the NEW emission matrix (kalman_filter.py:109-182: nothing is cached in the reference)."""
    bsz, t, d, m = 3, 400, 6, 1
    kw = random_ssm(rng, (bsz,), t, d, m, well=True)
    h_shared = kw["h"][0].copy()
    ssm = mfa.StateSpaceModel(tt(kw["mu0"]), tt(kw["chol_p0"]), tt(kw["a_s"]), tt(kw["b_s"]), tt(kw["chol_q"]))
    h_dev = tt(h_shared)
    kf = mfa.KalmanFilter(ssm, mfa.EmissionModel(h_dev), tt(kw["y"]), tt(np.array([[0.7]])))
    for scale in (2.0, 0.25, 3.0):
        kf.log_likelihood()
        h_dev.mul_(scale)
        h_shared = h_shared * scale
        post = kf.posterior_state_space_model()
        kw2 = dict(kw)
        kw2["h"] = np.broadcast_to(h_shared, (bsz, t, m, d)).copy()
        want = O.kf_posterior_ssm(**kw2, r_inv=np.array([[1.0 / 0.49]]))
        assert_chain(_posterior_tuple(post), want, rtol=1e-8, atol=1e-10)


def test_filter_cache_with_non_contiguous_inputs(rng):
    """Non-contiguous observations (a strided view of a larger tensor: flattened into a temporary on every call) and a chain
    built from a permuted ``state_transitions`` (the constructor takes a contiguous snapshot, state_space_model.py:74-116 holds
    values): writes through the view's base and through the tensor the chain holds must both be seen."""
    bsz, t, d, m = 3, 300, 4, 1
    kw = random_ssm(rng, (bsz,), t, d, m, well=True)
    a_view = tt(np.ascontiguousarray(np.swapaxes(kw["a_s"], -1, -2))).transpose(-1, -2)
    assert not a_view.is_contiguous()
    ssm = mfa.StateSpaceModel(tt(kw["mu0"]), tt(kw["chol_p0"]), a_view, tt(kw["b_s"]), tt(kw["chol_q"]))
    y_store = tt(np.repeat(kw["y"], 2, axis=-1))
    y_view = y_store[..., ::2]
    assert not y_view.is_contiguous()
    kf = mfa.KalmanFilter(ssm, mfa.EmissionModel(tt(kw["h"])), y_view, tt(np.array([[0.7]])))
    a_now, y_now = kw["a_s"].copy(), kw["y"].copy()
    for step in range(4):
        kf.log_likelihood()
        if step % 2 == 0:
            y_store.add_(1.0)
            y_now = y_now + 1.0
        else:
            ssm.state_transitions.mul_(0.5)
            a_now = a_now * 0.5
        post = kf.posterior_state_space_model()
        kw2 = dict(kw)
        kw2["a_s"], kw2["y"] = a_now, y_now
        want = O.kf_posterior_ssm(**kw2, r_inv=np.array([[1.0 / 0.49]]))
        assert_chain(_posterior_tuple(post), want, rtol=1e-8, atol=1e-10)


def test_filter_cache_still_hits_on_unchanged_broadcast_inputs(rng, monkeypatch):
    """... and the cache is still used when nothing changed, broadcast inputs included."""
    bsz, t, d, m = 3, 400, 6, 1
    kw = random_ssm(rng, (bsz,), t, d, m, well=True)
    ssm = mfa.StateSpaceModel(tt(kw["mu0"]), tt(kw["chol_p0"]), tt(kw["a_s"]), tt(kw["b_s"]), tt(kw["chol_q"]))
    kf = mfa.KalmanFilter(ssm, mfa.EmissionModel(tt(kw["h"][0])), tt(kw["y"]), tt(np.array([[0.7]])))
    seen = []
    real = _lib.call_rc

    def spy(name, *args):
        seen.append(name)
        return real(name, *args)

    monkeypatch.setattr(_lib, "call_rc", spy)
    kf.log_likelihood()
    post = kf.posterior_state_space_model()
    assert "mf_kf_posterior_chain_from_filter" in seen
    kw2 = dict(kw)
    kw2["h"] = np.broadcast_to(kw["h"][0], (bsz, t, m, d)).copy()
    want = O.kf_posterior_ssm(**kw2, r_inv=np.array([[1.0 / 0.49]]))
    assert_chain(_posterior_tuple(post), want, rtol=1e-8, atol=1e-10)
