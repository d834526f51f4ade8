"""
ctypes binding of the C restatement ``oracle/c/mf_oracle.c`` (TEST INFRASTRUCTURE - see its header).

Used by tests (differential check against the numpy oracle) and by ``bench.py``'s ``cpu_baseline``
leg, where it is the timed "CPU port" of the reference algorithm.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "c", "libmf_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "c", "mf_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _c(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def num_threads() -> int:
    return int(lib().mf_oracle_num_threads())


def set_num_threads(n: int) -> None:
    lib().mf_oracle_set_num_threads(ctypes.c_int(int(n)))


def use_library(path: str) -> None:
    """Switch to another build of the same source (bench.py compiles one with -march=native on the box it runs on)."""
    global _lib
    _lib = ctypes.CDLL(path)


def kf_loglik(mu0, chol_p0, a_s, b_s, chol_q, h, y, r_inv, per_step=False, generic=False):
    """Per-series log-likelihood [B] for [B,...] inputs (see mf_oracle_kf_loglik_f64).  ``generic``: the entry point whose loop
    bounds are run-time values only (d = 6 / d = 4 with one output otherwise take a compile-time-sized instance)."""
    mu0, chol_p0, a_s, b_s, chol_q, h, y, r_inv = map(_c, (mu0, chol_p0, a_s, b_s, chol_q, h, y, r_inv))
    bsz, t, m, d = h.shape
    out = np.zeros(bsz)
    fn = lib().mf_oracle_kf_loglik_generic_f64 if generic else lib().mf_oracle_kf_loglik_f64
    rc = fn(
        ctypes.c_int64(bsz), ctypes.c_int64(t), ctypes.c_int(d), ctypes.c_int(m), _p(mu0), _p(chol_p0),
        _p(a_s), _p(b_s), _p(chol_q), _p(h), _p(y), _p(r_inv), ctypes.c_int(int(per_step)), _p(out))
    if rc != 0:
        raise RuntimeError(f"mf_oracle_kf_loglik_f64 returned {rc}")
    return out


def btd_cholesky(diag, sub):
    diag, sub = _c(diag), _c(sub)
    bsz, t, d, _ = diag.shape
    ld = np.zeros_like(diag)
    ls = None if sub is None else np.zeros_like(sub)
    rc = lib().mf_oracle_btd_cholesky_f64(ctypes.c_int64(bsz), ctypes.c_int64(t), ctypes.c_int(d),
                                          _p(diag), _p(sub), _p(ld), _p(ls))
    if rc != 0:
        raise RuntimeError(f"mf_oracle_btd_cholesky_f64 returned {rc}")
    return ld, ls


def btd_solve(ld, ls, rhs, transpose=False):
    ld, ls, rhs = _c(ld), _c(ls), _c(rhs)
    bsz, t, d, _ = ld.shape
    out = np.zeros_like(rhs)
    rc = lib().mf_oracle_btd_solve_f64(ctypes.c_int64(bsz), ctypes.c_int64(t), ctypes.c_int(d),
                                       _p(ld), _p(ls), _p(rhs), _p(out), ctypes.c_int(int(transpose)))
    if rc != 0:
        raise RuntimeError(f"mf_oracle_btd_solve_f64 returned {rc}")
    return out
