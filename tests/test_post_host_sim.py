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
HDR = os.path.join(os.path.dirname(HERE), "..", "markovflow_amd", "csrc", "mf_post_math.hpp")


def _lib():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    stale = not os.path.exists(LIB) or any(os.path.getmtime(f) > os.path.getmtime(LIB) for f in (SRC, HDR))
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
