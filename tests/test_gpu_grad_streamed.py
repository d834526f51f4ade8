"""
GPU parity tests of the streamed backward of KalmanFilter.log_likelihood (csrc/mf_grad_lds.hpp: `mf_kf_loglik_grad_streamed_*`)
- the gradients TensorFlow's reverse mode produces through /root/reference/markovflow/kalman_filter.py:184-255 (pinned there by
tests/integration/models/test_gaussian_process_regression.py:117-130, test_variational.py:123-132).  Through the C ABI, every
gradient tensor, against torch autograd of the DENSE log-likelihood on the CPU (rtol 1e-6) and, at sizes beyond it, against
the route the kernels replace (posterior chain -> marginal scans -> one lane per time point).  The same arithmetic runs on the
CPU, lane by lane, in tests/test_post_host_sim.py.
"""
import numpy as np
import pytest
import torch

import markovflow_amd as mfa
from markovflow_amd import _lib, kalman_filter as kfm
from test_gpu_kalman import DEV, build_kf, nn, random_ssm, tt
from test_post_host_sim import _dense_log_likelihood

pytestmark = pytest.mark.gpu
NAMES = ("mu0", "cholP0", "A", "b", "cholQ", "H", "y", "Omega")


def grad_streamed_abi(kw, r_inv, w, chunks, dtype=torch.float64, per_step=False, fwd_chunks=None, strict=True, skip=()):
    """Call mf_kf_loglik_grad_streamed directly; the eight gradient tensors as numpy arrays.  fwd_chunks (0 = automatic): evaluate
    mf_kf_loglik first, on that many chunks per series, and hand its workspace (the chunk summaries) to the backward."""
    import ctypes
    ins = [tt(kw[k], dtype) for k in ("mu0", "chol_p0", "a_s", "b_s", "chol_q", "h", "y")] + [tt(r_inv, dtype)]
    bsz, t, m, d = ins[5].shape
    lib = _lib.load()
    fwd = (None, 0, 0)
    if fwd_chunks is not None:
        esz = ins[0].element_size()
        fws = torch.empty(int(lib.mf_kf_loglik_workspace_bytes(bsz, t, d, esz, fwd_chunks)), dtype=torch.uint8, device=DEV)
        val = torch.empty(bsz, dtype=dtype, device=DEV)
        finfo = _lib.new_info(torch.device(DEV))
        _lib.call("mf_kf_loglik", dtype, bsz, t, d, m, *[_lib.ptr(x) for x in ins], int(per_step), 0.0, _lib.ptr(val), _lib.ptr(fws),
                  fws.numel(), _lib.ptr(finfo), fwd_chunks, None, None, _lib.stream_ptr(torch.device(DEV)))
        path, p_f, l_f = ctypes.c_int(0), ctypes.c_int64(0), ctypes.c_int64(0)
        assert lib.mf_kf_loglik_plan(bsz, t, d, m, int(per_step), esz, fwd_chunks, 1, ctypes.byref(path), ctypes.byref(p_f),
                                     ctypes.byref(l_f)) == 0
        if strict:
            assert path.value == 2 and p_f.value >= 2, "the forward should have taken the streaming kernel on several chunks"
        if path.value == 2 and p_f.value >= 2:
            fwd = (fws, int(p_f.value), int(l_f.value))
    wsb = int(lib.mf_kf_loglik_grad_streamed_workspace_bytes(bsz, t, d, m, int(per_step), ins[0].element_size(), chunks))
    assert wsb > 0, "the streamed kernels should cover this call"
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    outs = [torch.full_like(x, float("nan")) for x in ins[:7]] + [torch.full((bsz, t, m, m), float("nan"), dtype=dtype, device=DEV)]
    info = _lib.new_info(torch.device(DEV))
    _lib.call("mf_kf_loglik_grad_streamed", dtype, bsz, t, d, m, *[_lib.ptr(x) for x in ins], int(per_step), _lib.ptr(tt(w, dtype)),
              *[None if i in skip else _lib.ptr(x) for i, x in enumerate(outs)], _lib.ptr(ws), wsb, _lib.ptr(info), chunks, _lib.ptr(fwd[0]), fwd[1], fwd[2], None, None,
              _lib.stream_ptr(torch.device(DEV)))
    torch.cuda.synchronize()
    assert int(info.item()) == 0
    return [nn(x) for x in outs]


def dense_autograd(kw, r_inv, w, per_step):
    leaves = [torch.tensor(kw[k], requires_grad=True) for k in ("mu0", "chol_p0", "a_s", "b_s", "chol_q", "h", "y")]
    ri = torch.tensor(r_inv, requires_grad=True)
    total = 0.0
    for s in range(leaves[0].shape[0]):
        total = total + w[s] * _dense_log_likelihood(*[v[s] for v in leaves], ri[s] if per_step else ri)
    total.backward()
    g = [v.grad.numpy() for v in leaves]
    g[1], g[4] = np.tril(g[1]), np.tril(g[4])
    gr = ri.grad.numpy()
    return g, 0.5 * (gr + np.swapaxes(gr, -1, -2))


def precision_gradient(om, r_inv, w, t, per_step):
    """d/dR^-1 from the kernel's Omega: -1/2 Omega from the quadratic forms + 1/2 R from the log-determinant (the caller's)."""
    if per_step:
        return -0.5 * om + 0.5 * w[:, None, None, None] * np.linalg.inv(r_inv)
    return -0.5 * om.sum(axis=(0, 1)) + 0.5 * w.sum() * t * np.linalg.inv(r_inv)


@pytest.mark.parametrize("d,m,t,bsz,chunks,per_step", [
    (6, 1, 100, 3, 0, False), (6, 1, 101, 2, 7, False), (6, 1, 64, 2, 2, False), (6, 1, 150, 1, 70, False),
    (6, 2, 90, 2, 0, False), (6, 3, 70, 2, 5, False), (6, 1, 57, 3, 4, True), (4, 1, 120, 3, 0, False),
    (4, 2, 37, 2, 3, False), (4, 1, 37, 2, 6, True), (2, 1, 50, 70, 0, False), (2, 2, 33, 3, 4, False),
    (1, 1, 40, 3, 5, False), (3, 1, 130, 2, 0, False), (5, 1, 129, 2, 11, False), (5, 3, 45, 2, 2, False),
    (3, 2, 20, 130, 2, False), (6, 1, 3, 70, 2, False),
])
def test_streamed_backward_against_dense_autograd(rng, d, m, t, bsz, chunks, per_step):
    kw = random_ssm(rng, (bsz,), t, d, m, well=True)
    if per_step:
        r_inv = rng.uniform(0.5, 2.0, size=(bsz, t, m, m))
    else:
        r = rng.normal(size=(m, m))
        r_inv = np.linalg.inv(r @ r.T + np.eye(m))
    w = rng.uniform(0.5, 1.5, size=bsz)
    got = grad_streamed_abi(kw, r_inv, w, chunks, per_step=per_step)
    want, want_r = dense_autograd(kw, r_inv, w, per_step)
    for name, g, ref in zip(NAMES[:7], got[:7], want):
        assert np.all(np.isfinite(g)), name
        np.testing.assert_allclose(g, ref, rtol=1e-6, atol=1e-8 * (1 + np.abs(ref).max()), err_msg=name)
    got_r = precision_gradient(got[7], r_inv, w, t, per_step)
    np.testing.assert_allclose(got_r, want_r, rtol=1e-6, atol=1e-8 * (1 + np.abs(want_r).max()), err_msg="R^-1")


# the same gradients from the FORWARD evaluation's chunk summaries (three passes instead of five): the backward's chunks are groups
# of the forward's - one group per chunk, several, ragged last groups, more than 64 forward chunks per series
@pytest.mark.parametrize("d,m,t,bsz,chunks,fwd_chunks,per_step", [
    (6, 1, 100, 3, 0, 0, False), (6, 1, 101, 2, 5, 12, False), (6, 1, 64, 2, 2, 2, False), (6, 1, 150, 1, 7, 70, False),
    (6, 2, 90, 2, 4, 9, False), (6, 3, 70, 2, 5, 5, False), (6, 1, 57, 3, 4, 11, True), (4, 1, 120, 3, 0, 0, False),
    (4, 2, 37, 2, 3, 7, False), (2, 1, 50, 70, 0, 0, False), (1, 1, 40, 3, 5, 10, False), (3, 1, 130, 2, 3, 100, False),
    (5, 1, 129, 2, 11, 33, False), (5, 3, 45, 2, 2, 6, False), (6, 1, 3, 70, 2, 2, False),
])
def test_streamed_backward_from_the_forward_summaries(rng, d, m, t, bsz, chunks, fwd_chunks, per_step):
    kw = random_ssm(rng, (bsz,), t, d, m, well=True)
    if per_step:
        r_inv = rng.uniform(0.5, 2.0, size=(bsz, t, m, m))
    else:
        r = rng.normal(size=(m, m))
        r_inv = np.linalg.inv(r @ r.T + np.eye(m))
    w = rng.uniform(0.5, 1.5, size=bsz)
    got = grad_streamed_abi(kw, r_inv, w, chunks, per_step=per_step, fwd_chunks=fwd_chunks)
    want, want_r = dense_autograd(kw, r_inv, w, per_step)
    for name, g, ref in zip(NAMES[:7], got[:7], want):
        assert np.all(np.isfinite(g)), name
        np.testing.assert_allclose(g, ref, rtol=1e-6, atol=1e-8 * (1 + np.abs(ref).max()), err_msg=name)
    got_r = precision_gradient(got[7], r_inv, w, t, per_step)
    np.testing.assert_allclose(got_r, want_r, rtol=1e-6, atol=1e-8 * (1 + np.abs(want_r).max()), err_msg="R^-1")


def test_gradients_nobody_asked_for_are_not_stored(rng):
    """g_b, g_H, g_y, g_omega = NULL: their buffers stay untouched and the others do not change."""
    kw = random_ssm(rng, (3,), 90, 6, 2, well=True)
    r_inv = np.linalg.inv(np.array([[1.0, 0.2], [0.2, 0.7]]))
    w = rng.uniform(0.5, 1.5, size=3)
    full = grad_streamed_abi(kw, r_inv, w, 5, fwd_chunks=0)
    part = grad_streamed_abi(kw, r_inv, w, 5, fwd_chunks=0, skip=(3, 5, 6, 7))
    for i, name in enumerate(NAMES):
        if i in (3, 5, 6, 7):
            assert np.all(np.isnan(part[i])), name
        else:
            np.testing.assert_array_equal(part[i], full[i], err_msg=name)


def test_streamed_backward_is_the_route_of_few_long_series_and_agrees_with_the_route_it_replaces(rng, monkeypatch):
    kw = random_ssm(rng, (5,), 700, 6, 1, well=True)
    leaves = {k: tt(v).requires_grad_(True) for k, v in kw.items() if k != "y"}
    y = tt(kw["y"])

    def run():
        for v in leaves.values():
            v.grad = None
        ssm = mfa.StateSpaceModel(leaves["mu0"], leaves["chol_p0"], leaves["a_s"], leaves["b_s"], leaves["chol_q"])
        kf = mfa.KalmanFilter(ssm, mfa.EmissionModel(leaves["h"]), y, tt(np.array([[0.7]])))
        kf.log_likelihood().sum().backward()
        return {k: nn(v.grad) for k, v in leaves.items()}

    seen, from_forward = [], []
    real = _lib.call_rc

    def spy(name, *args):
        seen.append(name)
        if name == "mf_kf_loglik_grad_streamed":
            from_forward.append(args[-6] is not None and args[-5] >= 2)      # fwd_ws, chunks per series of the forward evaluation
        return real(name, *args)

    monkeypatch.setattr(_lib, "call_rc", spy)
    streamed = run()
    assert "mf_kf_loglik_grad_streamed" in seen and from_forward == [True]   # three passes, from the forward's summaries
    monkeypatch.setattr(kfm, "_GRAD_FROM_FORWARD", False)
    own = run()                                                            # five passes
    assert from_forward[-1] is False
    for k in streamed:
        np.testing.assert_allclose(streamed[k], own[k], rtol=1e-9, atol=1e-11 * (1 + np.abs(own[k]).max()), err_msg=k)
    monkeypatch.setattr(kfm, "_GRAD_STREAMED", False)
    seen.clear()
    other = run()
    assert "mf_kf_loglik_grad_streamed" not in seen
    for k in streamed:
        scale = np.abs(other[k]).max()
        np.testing.assert_allclose(streamed[k], other[k], rtol=1e-7, atol=1e-9 * (1 + scale), err_msg=k)


@pytest.mark.parametrize("d,m", [(1, 1), (2, 2), (3, 1), (4, 3), (5, 1)])
def test_streamed_backward_fp32_other_state_dimensions(rng, d, m):
    kw = random_ssm(rng, (3,), 90, d, m, well=True)
    kw = {k: v.astype(np.float32).astype(np.float64) for k, v in kw.items()}
    r_inv = np.eye(m) * 2.0
    w = np.ones(3)
    want, _ = dense_autograd(kw, r_inv, w, False)
    for fwd_chunks in (None, 5):
        got = grad_streamed_abi(kw, r_inv, w, 4, dtype=torch.float32, fwd_chunks=fwd_chunks)
        for name, g, ref in zip(NAMES[:7], got[:7], want):
            np.testing.assert_allclose(g, ref, rtol=5e-3, atol=5e-3 * (1 + np.abs(ref).max()), err_msg=name)


@pytest.mark.parametrize("fwd_chunks", [None, 0, 9])
def test_streamed_backward_fp32(rng, fwd_chunks):
    kw = random_ssm(rng, (3,), 120, 6, 1, well=True)
    kw = {k: v.astype(np.float32).astype(np.float64) for k, v in kw.items()}
    r_inv = np.eye(1) * 2.0
    w = np.ones(3)
    got = grad_streamed_abi(kw, r_inv, w, 0, dtype=torch.float32, fwd_chunks=fwd_chunks)
    want, _ = dense_autograd(kw, r_inv, w, False)
    for name, g, ref in zip(NAMES[:7], got[:7], want):
        np.testing.assert_allclose(g, ref, rtol=5e-3, atol=5e-3 * (1 + np.abs(ref).max()), err_msg=name)
