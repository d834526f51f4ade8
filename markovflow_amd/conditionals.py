"""
Conditional distributions of states between conditioning points - mirror of ``markovflow/conditionals.py`` (reference): the four
functions SURVEY.md 8(f3) names, with the reference's signatures and return conventions,

    conditional_predict            (conditionals.py:29-83)      base_conditional_predict   (conditionals.py:380-421)
    conditional_statistics         (conditionals.py:87-120)     pairwise_marginals         (conditionals.py:424-485)

and the private ``_conditional_statistics`` / ``_conditional_statistics_from_transitions`` they are built from (:122-256).
Callers in the reference bind these functions (``posterior.py:217-221``, ``models/sparse_pep.py:245,289``), not a class.

Device work: the transitions to / from the new points come from the kernel's HIP generator (``mf_sde_matern_transitions_*``), the
statistics ``(P_t, T_t)`` of every new point from ONE kernel (``mf_sde_conditional_statistics_*``: a lane per point - Cholesky of
``Q_tp + A_tp Q_mt A_tp^T``, two triangular solves and three small products in registers, d <= 9; batched closed forms beyond),
the pairwise marginals from the chain's moment kernels (``mf_ssm_marginals_*`` through ``GaussMarkovDistribution._moments``);
``base_conditional_predict`` is two batched ``[d, 2d]`` products.  ``ConditionalProcess.predict_state`` (posterior.py) evaluates
the composition ``pairwise_marginals -> conditional_predict`` in a single fused kernel (``mf_sde_conditional_predict_*``) without
materialising the ``[N + 1, 2d, 2d]`` joint covariances; ``tests/test_gpu_conditionals.py`` checks that the two routes agree.
"""
from typing import Optional, Tuple

import torch

from . import _lib
from .gauss_markov import GaussMarkovDistribution
from .kernels import SDEKernel

APPROX_INF = 1e10   # markovflow/base.py:46


def _flat(t: torch.Tensor, tail: int) -> torch.Tensor:
    return t.reshape((-1,) + tuple(t.shape[-tail:])).contiguous()


def _conditional_statistics_from_transitions(state_transitions_to_t: torch.Tensor, process_covariances_to_t: torch.Tensor,
                                             state_transitions_from_t: torch.Tensor, process_covariances_from_t: torch.Tensor,
                                             return_precision: bool = False) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """``(D_t, E_t, T_t)`` (or ``T_t^-1``) of ``p(x_t | x_-, x_+) = N(D_t x_- + E_t x_+, T_t)`` from the transitions
    ``x_- -> x_t`` (``A_mt, Q_mt``) and ``x_t -> x_+`` (``A_tp, Q_tp``), each ``batch + [num_points, d, d]``
    (conditionals.py:122-203): ``E = Q_mt A_tp^T (Q_tp + A_tp Q_mt A_tp^T)^-1``, ``D = A_mt - E A_tp A_mt``,
    ``T = Q_mt - Q_mt A_tp^T (...)^-1 A_tp Q_mt``."""
    a_mt, q_mt, a_tp, q_tp = state_transitions_to_t, process_covariances_to_t, state_transitions_from_t, process_covariances_from_t
    lead, d = tuple(a_mt.shape[:-2]), a_mt.shape[-1]
    # (the kernel takes four arrays of ONE shape; broadcastable inputs - which the reference's tf.matmul accepts - take the
    # torch route below, which broadcasts)
    same_shape = a_mt.shape == q_mt.shape == a_tp.shape == q_tp.shape
    if a_mt.is_cuda and same_shape and d <= _lib.load().mf_max_state_dim() and not return_precision and a_mt.numel() > 0 \
            and not (torch.is_grad_enabled() and any(x.requires_grad for x in (a_mt, q_mt, a_tp, q_tp))):
        f = [_flat(x, 2) for x in (a_mt, q_mt, a_tp, q_tp)]
        n = f[0].shape[0]
        proj = torch.empty((n, d, 2 * d), dtype=a_mt.dtype, device=a_mt.device)
        cov = torch.empty((n, d, d), dtype=a_mt.dtype, device=a_mt.device)
        info = _lib.pivot_info(a_mt.device)
        _lib.call("mf_sde_conditional_statistics", a_mt.dtype, n, d, *[_lib.ptr(x) for x in f], _lib.ptr(proj), _lib.ptr(cov), info,
                  _lib.stream_ptr(a_mt.device))
        _lib.raise_on_info(info, "conditional_statistics", a_mt.device)
        proj = proj.reshape(lead + (d, 2 * d))
        return proj[..., :d], proj[..., d:], cov.reshape(lead + (d, d))
    tr = lambda t: t.transpose(-1, -2)                                   # noqa: E731
    tri = torch.linalg.solve_triangular
    g = a_tp @ q_mt
    chol = _lib.checked_cholesky(q_tp + g @ tr(a_tp), "conditional_statistics")
    v = tri(chol, g, upper=False)                                         # L^-1 A_tp Q_mt
    e_m = tr(tri(tr(chol), v, upper=True))
    d_m = a_mt - e_m @ a_tp @ a_mt
    if return_precision:
        eye = torch.eye(d, dtype=a_mt.dtype, device=a_mt.device).expand(q_mt.shape)
        q_mt_inv = _lib.chol_solve(_lib.checked_cholesky(q_mt, "conditional_statistics"), eye)
        w = tri(_lib.checked_cholesky(q_tp, "conditional_statistics"), a_tp, upper=False)
        return d_m, e_m, q_mt_inv + tr(w) @ w
    return d_m, e_m, q_mt - tr(v) @ v


def _conditional_statistics(new_time_points: torch.Tensor, training_time_points: torch.Tensor, kernel: SDEKernel
                            ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """``P_t [.., num_new, d, 2d]``, ``T_t [.., num_new, d, d]`` and the insertion indices ``[.., num_new]`` of the new points
    among the (sorted) training points (conditionals.py:205-256); beyond both ends the neighbour sits at -/+ ``APPROX_INF``."""
    batch = tuple(new_time_points.shape[:-1])
    dtype, dev = training_time_points.dtype, training_time_points.device
    new = new_time_points.to(dtype).contiguous()
    indices = torch.searchsorted(training_time_points.contiguous(), new)
    inf = torch.full(batch + (1,), APPROX_INF, dtype=dtype, device=dev)
    aug = torch.cat([-inf, training_time_points, inf], dim=-1)
    minus, plus = torch.gather(aug, -1, indices), torch.gather(aug, -1, indices + 1)
    a_mt, q_mt = kernel.transition_statistics(minus, new - minus)
    a_tp, q_tp = kernel.transition_statistics(new, plus - new)
    d_m, e_m, t_m = _conditional_statistics_from_transitions(a_mt, q_mt, a_tp, q_tp)
    return torch.cat([d_m, e_m], dim=-1), t_m, indices


def conditional_statistics(new_time_points: torch.Tensor, training_time_points: torch.Tensor, kernel: SDEKernel
                           ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Statistics ``P_t`` (``batch + [num_new, d, 2d]``) and ``T_t`` (``batch + [num_new, d, d]``) of
    ``p(x_t | x_-, x_+) = N(P_t [x_-, x_+], T_t)`` for every new time point (conditionals.py:87-120).  Both sets of time points
    must be sorted."""
    proj, cov, _ = _conditional_statistics(new_time_points, training_time_points, kernel)
    return proj, cov


def base_conditional_predict(conditional_projections: torch.Tensor, conditional_covariances: torch.Tensor,
                             adjacent_states: torch.Tensor, pairwise_state_covariances: Optional[torch.Tensor] = None
                             ) -> Tuple[torch.Tensor, torch.Tensor]:
    """``p(x_t) = N(P_t m_t, T_t + P_t S_t P_t^T)``, or the conditional ``N(P_t m_t, T_t)`` when ``S_t`` is not given
    (conditionals.py:380-421).  ``adjacent_states``: ``batch + [num_points, 2d]``; ``pairwise_state_covariances``:
    ``batch + [num_points, 2d, 2d]``."""
    means = (conditional_projections @ adjacent_states[..., None])[..., 0]
    covs = conditional_covariances
    if pairwise_state_covariances is not None:
        covs = covs + (conditional_projections @ pairwise_state_covariances) @ conditional_projections.transpose(-1, -2)
    return means, covs


def conditional_predict(new_time_points: torch.Tensor, training_time_points: torch.Tensor, kernel: SDEKernel,
                        training_pairwise_means: torch.Tensor, training_pairwise_covariances: Optional[torch.Tensor] = None
                        ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Means ``batch + [num_new, d]`` and covariances ``batch + [num_new, d, d]`` of the states at ``new_time_points`` given the
    pairwise marginals of the states at the neighbouring training points (conditionals.py:29-83).
    ``training_pairwise_means``: ``batch + [num_training + 1, 2d]`` and ``training_pairwise_covariances``:
    ``batch + [num_training + 1, 2d, 2d]`` as ``pairwise_marginals`` returns them (pair i = the points left and right of
    insertion index i); without the covariances the conditional density given ``[x_-, x_+] = m_t`` is returned."""
    proj, cov, indices = _conditional_statistics(new_time_points, training_time_points, kernel)
    two_d = training_pairwise_means.shape[-1]
    means = torch.gather(training_pairwise_means, -2, indices[..., None].expand(tuple(indices.shape) + (two_d,)))
    covs = None
    if training_pairwise_covariances is not None:
        covs = torch.gather(training_pairwise_covariances, -3,
                            indices[..., None, None].expand(tuple(indices.shape) + (two_d, two_d)))
    return base_conditional_predict(proj, cov, means, pairwise_state_covariances=covs)


def pairwise_marginals(dist: GaussMarkovDistribution, initial_mean: torch.Tensor, initial_covariance: torch.Tensor
                       ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Mean ``batch + [num_transitions + 2, 2d]`` and covariance ``batch + [num_transitions + 2, 2d, 2d]`` of every pair of
    subsequent states ``(x_k, x_{k+1})``, the chain extended by the prior ``N(initial_mean, initial_covariance)`` before its
    first and after its last state (uncorrelated with the chain: the prior sits infinitely far away) - conditionals.py:424-485."""
    needs_grad = getattr(dist, "_needs_grad", None)
    if needs_grad is not None and needs_grad():
        # under a tape: the differentiable moments, as the reference builds them (dist.marginals + dist.covariance_blocks(),
        # conditionals.py:449-450) - `_moments` writes raw kernel outputs into fresh buffers and would drop the gradient silently
        means = dist.marginal_means
        covs, sub = dist.covariance_blocks()
        if dist.num_transitions == 0:
            sub = None
    else:
        means, covs, sub = dist._moments(want_sub=dist.num_transitions > 0)
    batch, d = tuple(dist.batch_shape), dist.state_dim
    m0 = initial_mean.to(means.dtype).expand(batch + (d,))[..., None, :]
    p0 = initial_covariance.to(covs.dtype).expand(batch + (d, d))[..., None, :, :]
    ext_m = torch.cat([m0, means, m0], dim=-2)
    joint_mean = torch.cat([ext_m[..., :-1, :], ext_m[..., 1:, :]], dim=-1)
    ext_c = torch.cat([p0, covs, p0], dim=-3)
    zero = torch.zeros_like(p0)
    ext_s = torch.cat([zero, sub, zero], dim=-3) if sub is not None else torch.cat([zero, zero], dim=-3)    # Cov(x_{k+1}, x_k)
    top = torch.cat([ext_c[..., :-1, :, :], ext_s.transpose(-1, -2)], dim=-1)
    bottom = torch.cat([ext_s, ext_c[..., 1:, :, :]], dim=-1)
    return joint_mean, torch.cat([top, bottom], dim=-2)
