"""
The arithmetic of the streamed posterior-chain kernels WITHOUT a GPU: tests/host_sim/post_sim.cpp compiles the kernels' own
step functions (markovflow_amd/csrc/mf_post_math.hpp, `__host__ __device__`) for the CPU and runs the three passes - reversed
partitioned elimination per chunk, Kogge-Stone scan over the chunk summaries, emit - lane by lane, with the kernels' chunk
convention.  Checked against the numpy restatement of /root/reference/markovflow/kalman_filter.py:109-182
(oracle.numpy_oracle.kf_posterior_ssm), all five tensors, rtol 1e-9.  What this does NOT cover is the kernels' memory side
(LDS-DMA streams, stores): that is tests/test_gpu_posterior_streamed.py.
"""
import ctypes
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import numpy_oracle as O

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_sim")
LIB = os.path.join(HERE, "libmf_post_sim.so")
SRC = os.path.join(HERE, "post_sim.cpp")
CSRC = os.path.join(os.path.dirname(HERE), "..", "markovflow_amd", "csrc")
HDRS = [os.path.join(CSRC, f) for f in ("mf_post_math.hpp", "mf_grad_math.hpp", "mf_kernels.hpp", "mf_small.hpp")]


def _lib():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    stale = not os.path.exists(LIB) or any(os.path.getmtime(f) > os.path.getmtime(LIB) for f in [SRC] + HDRS)
    if stale:
        subprocess.check_call([hipcc, "-O1", "-std=c++17", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB, SRC])
    lib = ctypes.CDLL(LIB)
    lib.mf_post_host_sim_f64.restype = ctypes.c_int
    lib.mf_post_host_sim_f64.argtypes = ([ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 8 +
                                         [ctypes.c_int, ctypes.c_int64] + [ctypes.c_void_p] * 5)
    return lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.mark.parametrize("bsz,t,d,m,length,per_step", [
    (2, 30, 3, 1, 7, False), (2, 30, 6, 1, 5, False), (3, 101, 6, 2, 8, True), (1, 200, 4, 3, 3, False), (2, 2, 6, 1, 1, False),
    (2, 50, 1, 1, 49, False), (2, 50, 2, 1, 60, False), (1, 600, 6, 1, 4, False), (2, 65, 5, 1, 1, False),
])
def test_three_passes_on_the_host_agree_with_the_oracle(bsz, t, d, m, length, per_step):
    lib = _lib()
    rng = np.random.default_rng(7)
    a = 0.9 * np.eye(d) + 0.05 * rng.normal(size=(bsz, t - 1, d, d))
    cq = np.tril(0.1 * rng.normal(size=(bsz, t - 1, d, d))) + 0.5 * np.eye(d)
    cp0 = np.tril(0.1 * rng.normal(size=(bsz, d, d))) + np.eye(d)
    mu0 = rng.normal(size=(bsz, d))
    b = 0.1 * rng.normal(size=(bsz, t - 1, d))
    h = rng.normal(size=(bsz, t, m, d))
    y = rng.normal(size=(bsz, t, m))
    if per_step:
        r = rng.normal(size=(bsz, t, m, m))
        r_inv = r @ np.swapaxes(r, -1, -2) + np.eye(m)
    else:
        r = rng.normal(size=(m, m))
        r_inv = r @ r.T + np.eye(m)
    r_inv = np.ascontiguousarray(r_inv)
    want = O.kf_posterior_ssm(mu0, cp0, a, b, cq, h, y, r_inv)
    ap, bp, cqp, mp, cpp = (np.full_like(x, np.nan) for x in (a, b, cq, mu0, cp0))
    rc = lib.mf_post_host_sim_f64(bsz, t, d, m, _p(mu0), _p(cp0), _p(a), _p(b), _p(cq), _p(h), _p(y), _p(r_inv), int(per_step),
                                  length, _p(ap), _p(mp), _p(bp), _p(cpp), _p(cqp))
    assert rc == 0
    for name, g, w in zip(("mu0", "cholP0", "A", "b", "cholQ"), (mp, cpp, ap, bp, cqp), want):
        np.testing.assert_allclose(g, w, rtol=1e-9, atol=1e-12, err_msg=name)


def test_host_simulation_flags_a_singular_process_covariance():
    lib = _lib()
    rng = np.random.default_rng(3)
    bsz, t, d, m = 1, 40, 3, 1
    a = 0.5 * np.eye(d) + np.zeros((bsz, t - 1, d, d))
    cq = np.tile(np.eye(d), (bsz, t - 1, 1, 1))
    cq[0, 17] = 0.0
    args = [np.zeros((bsz, d)), np.tile(np.eye(d), (bsz, 1, 1)), a, np.zeros((bsz, t - 1, d)), cq, rng.normal(size=(bsz, t, m, d)),
            rng.normal(size=(bsz, t, m)), np.eye(m)]
    outs = [np.zeros_like(a), np.zeros((bsz, d)), np.zeros((bsz, t - 1, d)), np.zeros((bsz, d, d)), np.zeros_like(cq)]
    rc = lib.mf_post_host_sim_f64(bsz, t, d, m, *[_p(x) for x in args], 0, 8, *[_p(x) for x in outs])
    assert rc == 1


# ---- the streamed backward of log_likelihood (csrc/mf_grad_math.hpp) ------------------------------------------------------------
def _grad_lib():
    lib = _lib()
    lib.mf_grad_host_sim_f64.restype = ctypes.c_int
    lib.mf_grad_host_sim_f64.argtypes = ([ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 8 +
                                         [ctypes.c_int, ctypes.c_int64] + [ctypes.c_void_p] * 9)
    return lib


def _dense_log_likelihood(mu0, cp0, a_s, b_s, cq, h, y, r_inv):
    """log N(y; H mu, H Sigma H^T + blockdiag(R_k)) of ONE series through the dense precision of the chain (the block form of
    /root/reference/markovflow/state_space_model.py:431-483) - plain differentiable torch; r_inv [m, m] or [n, m, m]."""
    import torch
    n, d, m = a_s.shape[0] + 1, mu0.shape[0], h.shape[1]
    eye = torch.eye(d, dtype=mu0.dtype)
    qinv = [torch.cholesky_solve(eye, cp0)] + [torch.cholesky_solve(eye, cq[k]) for k in range(n - 1)]
    prec = torch.zeros(n * d, n * d, dtype=mu0.dtype)
    means = [mu0]
    for k in range(n):
        blk = qinv[k]
        if k < n - 1:
            j = qinv[k + 1] @ a_s[k]
            blk = blk + a_s[k].T @ j
            prec[(k + 1) * d:(k + 2) * d, k * d:(k + 1) * d] = -j
            prec[k * d:(k + 1) * d, (k + 1) * d:(k + 2) * d] = -j.T
            means.append(a_s[k] @ means[k] + b_s[k])
        prec[k * d:(k + 1) * d, k * d:(k + 1) * d] = blk
    hm = torch.block_diag(*[h[i] for i in range(n)])
    r_blocks = [torch.linalg.inv(r_inv if r_inv.dim() == 2 else r_inv[i]) for i in range(n)]
    cov_y = hm @ torch.linalg.inv(prec) @ hm.T + torch.block_diag(*r_blocks)
    res = y.reshape(-1) - hm @ torch.cat(means)
    return -0.5 * (res @ torch.linalg.solve(cov_y, res) + torch.linalg.slogdet(cov_y)[1] + n * m * np.log(2 * np.pi))


@pytest.mark.parametrize("bsz,t,d,m,length,per_step", [
    (2, 12, 3, 1, 4, False), (2, 17, 6, 1, 5, False), (2, 21, 6, 2, 8, True), (1, 30, 4, 3, 3, False), (2, 2, 6, 1, 1, False),
    (2, 9, 1, 1, 50, False), (1, 40, 6, 1, 1, False), (2, 26, 5, 2, 7, False), (2, 14, 2, 1, 2, True),
])
def test_streamed_backward_on_the_host_agrees_with_dense_autograd(bsz, t, d, m, length, per_step):
    """Three passes of the posterior chain, start moments per chunk from the two sides of every chunk boundary, then the forward
    pass with the pairwise marginals in registers: every gradient of sum_s w_s log p(y_s) against torch autograd of the dense
    log-likelihood (the precision's gradient through the quadratic form only: its log-determinant is the caller's)."""
    import torch
    lib = _grad_lib()
    rng = np.random.default_rng(11)
    a = 0.8 * np.eye(d) + 0.1 * rng.normal(size=(bsz, t - 1, d, d))
    cq = np.tril(0.1 * rng.normal(size=(bsz, t - 1, d, d))) + 0.5 * np.eye(d)
    cp0 = np.tril(0.1 * rng.normal(size=(bsz, d, d))) + np.eye(d)
    mu0 = rng.normal(size=(bsz, d))
    b = 0.1 * rng.normal(size=(bsz, t - 1, d))
    h = rng.normal(size=(bsz, t, m, d))
    y = rng.normal(size=(bsz, t, m))
    if per_step:
        r = rng.normal(size=(bsz, t, m, m))
        r_inv = r @ np.swapaxes(r, -1, -2) + np.eye(m)
    else:
        r = rng.normal(size=(m, m))
        r_inv = r @ r.T + np.eye(m)
    r_inv = np.ascontiguousarray(r_inv)
    w = rng.uniform(0.5, 1.5, size=bsz)
    outs = dict(gmu0=np.full_like(mu0, np.nan), gC0=np.full_like(cp0, np.nan), gA=np.full_like(a, np.nan), gb=np.full_like(b, np.nan),
                gC=np.full_like(cq, np.nan), gH=np.full_like(h, np.nan), gy=np.full_like(y, np.nan),
                gOm=np.full((bsz, t, m, m), np.nan))
    rc = lib.mf_grad_host_sim_f64(bsz, t, d, m, _p(mu0), _p(cp0), _p(a), _p(b), _p(cq), _p(h), _p(y), _p(r_inv), int(per_step),
                                  length, _p(w), *[_p(v) for v in outs.values()])
    assert rc == 0
    leaves = [torch.tensor(x, requires_grad=True) for x in (mu0, cp0, a, b, cq, h, y, r_inv)]
    total = 0.0
    for s in range(bsz):
        ri = leaves[7][s] if per_step else leaves[7]
        total = total + w[s] * _dense_log_likelihood(*[v[s] for v in leaves[:7]], ri)
    total.backward()
    want = [v.grad.numpy() for v in leaves]
    for name, g, ref in zip(("mu0", "cholP0", "A", "b", "cholQ", "H", "y"), list(outs.values())[:7], want[:7]):
        if name in ("cholP0", "cholQ"):
            ref = np.tril(ref)
        np.testing.assert_allclose(g, ref, rtol=1e-7, atol=1e-9, err_msg=name)
    # d/dR^-1: -1/2 Omega from the quadratic forms, + 1/2 R from the log-determinant (which the kernel leaves to the caller)
    om = outs["gOm"]
    if per_step:
        got = -0.5 * om + 0.5 * w[:, None, None, None] * np.linalg.inv(r_inv)
    else:
        got = -0.5 * om.sum(axis=(0, 1)) + 0.5 * w.sum() * t * np.linalg.inv(r_inv)
    np.testing.assert_allclose(got, 0.5 * (want[7] + np.swapaxes(want[7], -1, -2)), rtol=1e-7, atol=1e-9, err_msg="R^-1")
