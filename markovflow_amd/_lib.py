"""
ctypes binding of ``libmarkovflow_amd.so`` (the C ABI declared in ``include/markovflow_amd.h``).

There is deliberately NO fallback: if the shared library is missing, or a tensor is not on a HIP
device, the call fails loudly.  PyTorch is used only as the owner of device memory and streams.
"""
import ctypes
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MF_LIB_PATH", os.path.join(_HERE, "libmarkovflow_amd.so"))   # override: A/B builds

_i64, _int, _vp, _sz = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t

# name -> (restype, argtypes with the scalar type written as "T" / "Tp")
_SIGS = {
    "mf_kf_loglik": (_int, [_i64, _i64, _int, _int, "Tp", "Tp", "Tp", "Tp", "Tp", "Tp", "Tp", "Tp", _int, "T",
                            "Tp", _vp, _sz, _vp, _i64, _vp, _vp, _vp]),
    "mf_btd_cholesky": (_int, [_i64, _i64, _int, "Tp", "Tp", "Tp", "Tp", _vp, _sz, _vp, _vp]),
    "mf_btd_solve": (_int, [_i64, _i64, _i64, _int, "Tp", "Tp", "Tp", "Tp", _int, _vp, _sz, _vp]),
    "mf_btd_matvec": (_int, [_i64, _i64, _i64, _int, "Tp", "Tp", "Tp", "Tp", _int, _vp]),
    "mf_btd_logdet": (_int, [_i64, _i64, _int, "Tp", "Tp", _vp]),
    "mf_btd_logdet_quad": (_int, [_i64, _i64, _int, "Tp", "Tp", "Tp", "Tp", _vp, _sz, _vp, _vp]),
    "mf_btd_diag_of_inverse": (_int, [_i64, _i64, _int, "Tp", "Tp", "Tp", "Tp", _vp, _sz, _vp]),
    "mf_ssm_marginal_covariances": (_int, [_i64, _i64, _int, "Tp", "Tp", "Tp", "Tp", "Tp", _vp, _sz, _vp]),
    "mf_btd_udl": (_int, [_i64, _i64, _int, "Tp", "Tp", "Tp", "Tp", "Tp", "Tp", "Tp", _int, _vp, _sz, _vp, _vp]),
    "mf_ssm_precision": (_int, [_i64, _i64, _int, _int, "Tp", "Tp", "Tp", "Tp", "Tp", "Tp", "Tp", "Tp", _int,
                                "Tp", "Tp", "Tp", _vp]),
    "mf_ssm_marginal_means": (_int, [_i64, _i64, _i64, _int, "Tp", "Tp", "Tp", _vp, _sz, _vp]),
    "mf_block_matmul": (_int, [_i64, _i64, _int, "Tp", _i64, "Tp", _i64, "Tp", _vp]),
    "mf_gpr_matern_loglik": (_int, [_i64, _i64, _int, ctypes.POINTER(ctypes.c_int), "Tp", "Tp", _int, "Tp", "Tp", "Tp", "T", "T",
                                    "Tp", _vp, _sz, _vp, _i64, _vp, _vp, _vp]),
    "mf_kf_loglik_total": (_int, [_i64, "Tp", _int, "Tp", _i64, "Tp", "T", "Tp", _vp]),
    "mf_kf_loglik_grad": (_int, [_i64, _i64, _int, _int] + ["Tp"] * 20 + [_vp, _vp]),
    "mf_sde_conditional_predict": (_int, [_i64, _i64, _i64, _int, _vp, "Tp", "Tp", "Tp", "Tp", "Tp", "Tp", "Tp", "Tp", "Tp",
                                          "Tp", "Tp", _vp, _vp]),
    "mf_sde_matern_transitions": (_int, [_i64, _i64, _int, ctypes.POINTER(ctypes.c_int), "Tp", "Tp", _int, "Tp", "T",
                                         "Tp", "Tp", "Tp", _vp]),
}
_PLAIN = {
    "mf_version": (_int, []),
    "mf_max_state_dim": (_int, []),
    "mf_max_state_dim_f32_loglik": (_int, []),
    "mf_max_state_dim_f64_loglik": (_int, []),
    "mf_kf_loglik_workspace_bytes": (_sz, [_i64, _i64, _int, _int, _i64]),
    "mf_btd_logdet_quad_workspace_bytes": (_sz, [_i64, _i64, _int, _int]),
    "mf_btd_cholesky_workspace_bytes": (_sz, [_i64, _i64, _int, _int]),
    "mf_btd_solve_workspace_bytes": (_sz, [_i64, _i64, _i64, _int, _int]),
    "mf_btd_diag_of_inverse_workspace_bytes": (_sz, [_i64, _i64, _int, _int]),
    "mf_btd_udl_workspace_bytes": (_sz, [_i64, _i64, _int, _int]),
}

_lib = None


def exported_symbols():
    """Every symbol ``include/markovflow_amd.h`` declares."""
    names = list(_PLAIN)
    for base in _SIGS:
        names += [base + "_f64", base + "_f32"]
    return names


def load():
    """Load the shared library (once).  Raises ImportError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import "
            "__graft_entry__ as g; g.build()'` (or `make -C markovflow_amd/csrc`). There is no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _PLAIN.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    for base, (res, args) in _SIGS.items():
        for suf, scalar in (("_f64", ctypes.c_double), ("_f32", ctypes.c_float)):
            fn = getattr(lib, base + suf)
            fn.restype = res
            fn.argtypes = [(_vp if a == "Tp" else scalar if a == "T" else a) for a in args]
    _lib = lib
    return lib


def suffix(dtype: torch.dtype) -> str:
    if dtype == torch.float64:
        return "_f64"
    if dtype == torch.float32:
        return "_f32"
    raise TypeError(f"markovflow_amd supports float32 and float64 tensors, got {dtype}")


def ptr(t: Optional[torch.Tensor]):
    """Device pointer of a contiguous HIP tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(
            "markovflow_amd kernels run on an MI355X only: got a CPU tensor and there is no CPU fallback. "
            "Move the inputs to device 'cuda'."
        )
    if not t.is_contiguous():
        raise RuntimeError("internal error: non-contiguous tensor passed to the C ABI")
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr(device) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class MarkovflowAmdError(RuntimeError):
    pass


def check(rc: int, what: str):
    if rc == 0:
        return
    if rc == -100:
        raise NotImplementedError(
            f"{what}: state dimension not instantiated in this build (supported: 1..{load().mf_max_state_dim()}; "
            f"log_likelihood up to {load().mf_max_state_dim_f32_loglik()} in float32, "
            f"{load().mf_max_state_dim_f64_loglik()} in float64)"
        )
    if rc == -1000:
        raise MarkovflowAmdError(f"{what}: kernel launch failed")
    raise ValueError(f"{what}: invalid argument #{-rc} (see include/markovflow_amd.h)")


def call(base: str, dtype: torch.dtype, *args):
    fn = getattr(load(), base + suffix(dtype))
    check(fn(*args), base)


def workspace(nbytes: int, device) -> Optional[torch.Tensor]:
    """Caller-owned scratch for the C ABI (None when the entry point asked for none)."""
    return torch.empty(int(nbytes), dtype=torch.uint8, device=device) if nbytes else None


def chol_solve(chol: torch.Tensor, rhs: torch.Tensor) -> torch.Tensor:
    """``(chol cholᵀ)⁻¹ rhs`` per block as two batched triangular solves.  NOT ``torch.cholesky_solve``: on this ROCm build
    its batched (MAGMA) path does not order itself after kernels queued on the current stream by this library - measured:
    35 of 40 runs of ``naturals_to_ssm_params`` returned garbage with it, 0 of 40 with the two trsm calls."""
    y = torch.linalg.solve_triangular(chol, rhs, upper=False)
    return torch.linalg.solve_triangular(chol.transpose(-1, -2), y, upper=True)


def new_info(device) -> torch.Tensor:
    return torch.zeros(1, dtype=torch.int32, device=device)


# Debug switch: when set, every factorisation synchronises and raises on a non-positive pivot, the way
# TensorFlow's Cholesky op raises in the reference.  Off by default (no host sync on the hot path).
CHECK_PIVOTS = os.environ.get("MF_CHECK_PIVOTS", "0") == "1"


def pivot_info(device) -> Optional[torch.Tensor]:
    """The `info` flag the product path hands to the C ABI: a zeroed device int when pivots are checked, else NULL (the flag
    is optional in every entry point; without it a non-positive pivot shows up as NaN, and no fill kernel is launched)."""
    return new_info(device) if CHECK_PIVOTS else None


def raise_on_info(info: Optional[torch.Tensor], what: str):
    if CHECK_PIVOTS and info is not None and int(info.item()) != 0:
        raise MarkovflowAmdError(f"{what}: matrix is not positive definite")
