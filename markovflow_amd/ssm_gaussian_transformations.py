"""
State space model <-> natural / expectation parameters of the equivalent Gaussian.

Mirror of ``markovflow/ssm_gaussian_transformations.py`` (reference): the six functions with the same names, arguments
and return tuples.  The sequential pieces run on the HIP kernels of the block-tridiagonal operator (marginal means and
covariances, Cholesky, Takahashi diagonal + sub-diagonal blocks of the inverse, bidiagonal solves); the rest is per-block
d x d algebra in torch on the device, as it is batched TensorFlow in the reference.

One deliberate difference in ``naturals_to_ssm_params``: the reference obtains the conditional precisions ``Q_k^-1`` with a
banded triangular solve (``solve_triang_band`` of ``A^-T`` against the precision, ``ssm_gaussian_transformations.py:473-490``);
the block structure of an SSM precision gives them locally, ``Q_k^-1 = D_k + A_{k+1}^T S_k`` (``D`` / ``S`` the diagonal /
sub-diagonal blocks of the precision; ``S_k = -Q_{k+1}^-1 A_{k+1}``), which is what is used here.
"""
from typing import Tuple

import torch

from . import _autograd_ops as _ag
from . import _lib
from .block_tri_diag import LowerTriangularBlockTriDiagonal, SymmetricBlockTriDiagonal
from .state_space_model import StateSpaceModel

T3 = Tuple[torch.Tensor, torch.Tensor, torch.Tensor]
T5 = Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]


def _outer(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    return a[..., :, None] * b[..., None, :]


def _chol_solve(chol: torch.Tensor, rhs: torch.Tensor) -> torch.Tensor:
    """``(chol chol^T)^-1 rhs`` per block: the HIP route of ``_autograd_ops.chol_solve_blocks`` (d <= 9), torch beyond."""
    return _ag.chol_solve_blocks(chol, rhs)


def _mm(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Product of two stacks of ``d x d`` blocks (``mf_block_matmul_*`` on the device, d <= 9)."""
    return _ag.block_matmul(a.contiguous(), b.contiguous()) if a.shape == b.shape else a @ b


def _mv(a: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """``a @ v`` for blocks against vectors ``[..., d]``: element-wise product + reduction instead of a batched GEMV."""
    return torch.sum(a * v[..., None, :], dim=-1)


def _eye_like(m: torch.Tensor) -> torch.Tensor:
    return torch.eye(m.shape[-1], dtype=m.dtype, device=m.device).expand(m.shape)


def ssm_to_expectations(ssm: StateSpaceModel) -> T3:
    """``(eta_linear [..,N+1,D], eta_diag [..,N+1,D,D], eta_subdiag [..,N,D,D])``: ``E[x]``, the diagonal and the lower
    sub-diagonal blocks of ``E[x x^T]`` (ssm_gaussian_transformations.py:32-89)."""
    means, covs, sub_covs = ssm._moments(want_sub=True)
    eta_diag = covs + _outer(means, means)
    eta_subdiag = sub_covs + _outer(means[..., 1:, :], means[..., :-1, :])
    return means, eta_diag, eta_subdiag


def expectations_to_ssm_params(eta_linear: torch.Tensor, eta_diag: torch.Tensor, eta_subdiag: torch.Tensor) -> T5:
    """``(As, offsets, chol_initial_covariance, chol_process_covariances, initial_mean)`` from the expectation parameters:
    ``A_i = S_{i,i-1} S_{i-1}^-1``, ``Q_i = S_i - A_i S_{i-1} A_i^T``, ``b_i = eta_i - A_i eta_{i-1}`` (:93-178)."""
    marginal_covs = eta_diag - _outer(eta_linear, eta_linear)
    covs_sub_diag = eta_subdiag.transpose(-1, -2) - _outer(eta_linear[..., :-1, :], eta_linear[..., 1:, :])
    marginal_chols = SymmetricBlockTriDiagonal(marginal_covs).cholesky.block_diagonal       # per-block Cholesky (HIP)
    a_s = _chol_solve(marginal_chols[..., :-1, :, :], covs_sub_diag).transpose(-1, -2)
    offsets = eta_linear[..., 1:, :] - _mv(a_s, eta_linear[..., :-1, :])
    conditional_covs = marginal_covs[..., 1:, :, :] - _mm(_mm(a_s, marginal_covs[..., :-1, :, :]), a_s.transpose(-1, -2))
    chol_q = SymmetricBlockTriDiagonal(conditional_covs.contiguous()).cholesky.block_diagonal
    return a_s, offsets, marginal_chols[..., 0, :, :], chol_q, eta_linear[..., 0, :]


def ssm_to_naturals(ssm: StateSpaceModel) -> T3:
    """``(theta_linear, theta_diag, theta_subdiag)`` of ``exp(theta^T x + x^T Theta x)``: ``theta = K^-1 mu``,
    ``Theta = -1/2 K^-1`` with the SUB-diagonal blocks returned as ``Q_{i}^-1 A_i`` (:182-253)."""
    a_s = ssm.state_transitions
    offsets = ssm.concatenated_state_offsets[..., None]
    chols = ssm.concatenated_cholesky_process_covariance
    linv_a = torch.linalg.solve_triangular(chols[..., 1:, :, :], a_s, upper=False)
    theta_subdiag = torch.linalg.solve_triangular(chols[..., 1:, :, :].transpose(-1, -2), linv_a, upper=True)
    tmp = _chol_solve(chols, offsets)
    theta_linear = torch.cat([tmp[..., :-1, :, :] - _mv(a_s.transpose(-1, -2), tmp[..., 1:, :, 0])[..., None], tmp[..., -1:, :, :]],
                             dim=-3)[..., 0]
    ata = _mm(linv_a.transpose(-1, -2), linv_a)
    ata = torch.cat([ata, torch.zeros_like(ata[..., :1, :, :])], dim=-3)
    theta_diag = -0.5 * (_chol_solve(chols, _eye_like(chols)) + ata)
    return theta_linear, theta_diag, theta_subdiag


def ssm_to_naturals_no_smoothing(ssm: StateSpaceModel) -> T3:
    """Natural parameters of the CONDITIONALS ``p(x_i | x_{i-1})`` (no smoothing across time) (:257-329)."""
    a_s = ssm.state_transitions
    offsets = ssm.concatenated_state_offsets[..., None]
    chols = ssm.concatenated_cholesky_process_covariance
    theta_subdiag = _chol_solve(chols[..., 1:, :, :], a_s)
    theta_linear = _chol_solve(chols, offsets)[..., 0]
    theta_diag = -0.5 * _chol_solve(chols, _eye_like(chols))
    return theta_linear, theta_diag, theta_subdiag


def naturals_to_ssm_params(theta_linear: torch.Tensor, theta_diag: torch.Tensor, theta_subdiag: torch.Tensor) -> T5:
    """State space model parameters from the natural parameters (:333-511): the joint precision is
    ``SymmetricBlockTriDiagonal(-2 theta_diag, -theta_subdiag)``; its Cholesky factor and the diagonal + sub-diagonal blocks
    of its inverse (block Takahashi) are HIP kernels."""
    precision = SymmetricBlockTriDiagonal((-2.0 * theta_diag).contiguous(), (-theta_subdiag).contiguous())
    marginal_covs, sub_covs = precision.cholesky._diag_and_sub_of_inverse(want_sub=True)      # S_ii, S_{i+1,i}
    # A_{i+1} = S_{i+1,i} S_ii^-1
    chol_m = SymmetricBlockTriDiagonal(marginal_covs[..., :-1, :, :].contiguous()).cholesky.block_diagonal
    a_s = _chol_solve(chol_m, sub_covs.transpose(-1, -2)).transpose(-1, -2)
    # conditional precisions Q_k^-1 = D_k + A_{k+1}^T S_k  (last block: D_n)
    d_blocks, s_blocks = precision.block_diagonal, precision.block_sub_diagonal
    cond_prec = torch.cat([d_blocks[..., :-1, :, :] + _mm(a_s.transpose(-1, -2), s_blocks), d_blocks[..., -1:, :, :]], dim=-3)
    cond_prec = 0.5 * (cond_prec + cond_prec.transpose(-1, -2))
    chol_prec = SymmetricBlockTriDiagonal(cond_prec.contiguous()).cholesky.block_diagonal
    covariances = _chol_solve(chol_prec, _eye_like(chol_prec))
    chols = SymmetricBlockTriDiagonal(covariances.contiguous()).cholesky.block_diagonal
    eye = _eye_like(d_blocks).contiguous()
    a_inv_block = LowerTriangularBlockTriDiagonal(eye, (-a_s).contiguous())
    precision_times_offsets = a_inv_block.solve(theta_linear.contiguous(), transpose_left=True)
    offsets = _mv(covariances, precision_times_offsets)
    return a_s, offsets[..., 1:, :], chols[..., 0, :, :], chols[..., 1:, :, :], offsets[..., 0, :]


def naturals_to_ssm_params_no_smoothing(theta_linear: torch.Tensor, theta_diag: torch.Tensor,
                                        theta_subdiag: torch.Tensor) -> T5:
    """Inverse of :func:`ssm_to_naturals_no_smoothing` (:515-593)."""
    chol_prec = SymmetricBlockTriDiagonal((-2.0 * theta_diag).contiguous()).cholesky.block_diagonal
    a_s = _chol_solve(chol_prec[..., 1:, :, :], theta_subdiag)
    offsets = _chol_solve(chol_prec, theta_linear[..., None])[..., 0]
    conditional_covs = _chol_solve(chol_prec, _eye_like(chol_prec))
    chols = SymmetricBlockTriDiagonal(conditional_covs.contiguous()).cholesky.block_diagonal
    return a_s, offsets[..., 1:, :], chols[..., 0, :, :], chols[..., 1:, :, :], offsets[..., 0, :]
