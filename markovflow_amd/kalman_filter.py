"""
Kalman filter (SpInGP precision formulation) on the MI355X.

Mirror of ``markovflow/kalman_filter.py`` (reference): ``BaseKalmanFilter``, ``KalmanFilter``,
``GaussianSites``, ``UnivariateGaussianSitesNat``, ``KalmanFilterWithSites`` and
``KalmanFilterWithSparseSites`` with the same constructors, attributes and methods.

``log_likelihood`` is ONE fused HIP pipeline (``mf_kf_loglik_*``): precision assembly, posterior
Cholesky, forward solve and the log-determinants never touch HBM as intermediates, and the scan
over time is partitioned so that every SIMD of the chip has work.
``posterior_state_space_model`` is two kernels: a parallel assembly of the posterior precision and
information vector, and one backward UDUᵀ sweep that emits the posterior chain.
"""
import abc
import ctypes
import math
from typing import Optional

import torch

from . import _autograd_ops as _ag
from . import _lib
from .block_tri_diag import LowerTriangularBlockTriDiagonal, SymmetricBlockTriDiagonal, _flat
from .emission_model import EmissionModel
from .state_space_model import StateSpaceModel


def _sum_over_points(g: torch.Tensor) -> torch.Tensor:
    """``[B, T, m, m] -> [m, m]``, the sum over series and time points.  In two steps whose reduced axis is the OUTER one of a wide
    row (B over T m^2 columns, then T over m^2): torch's reduction of a ``[B T, m^2]`` array down its long axis runs at 0.4 % of the
    memory rate (1.04 ms for 37 MB at config 4's shape, a sixth of the training step: profiles/r05_config4_step.txt)."""
    bsz, n, m, _ = g.shape
    return g.reshape(bsz, n * m * m).sum(dim=0).reshape(n, m * m).sum(dim=0).reshape(m, m)


class _LogLikelihoodPerSeries(torch.autograd.Function):
    """
    Per-series log-likelihood (without the chain-independent constants) as a differentiable torch function of the flat
    model tensors and the observation PRECISION (shared ``[m, m]`` or per step ``[B, T, m, m]``: the sites variants).
    Forward: the fused HIP pipeline.  Backward: Fisher's identity on the smoothed marginals (``mf_kf_loglik_grad_*``,
    csrc/mf_kernels.hpp) - the posterior chain, its marginal means / covariances and cross-covariances are all HIP kernels,
    then ONE local kernel per (series, time point) produces every gradient.  How the precision depends on the user's
    parameters (``chol_obs_covariance``, site natural parameters, the scatter onto a grid) is plain torch outside this
    function.  Replaces the TensorFlow reverse mode over banded_matrices' registered gradients (SURVEY.md §8f rank 2;
    callers: models/gaussian_process_regression.py:150-160, models/variational_cvi.py:138-161).
    """

    @staticmethod
    def forward(ctx, mu0, cp0, a_s, b_s, cq, h, y, r_inv, chunks):
        with torch.no_grad():
            kf = _RawFilter(StateSpaceModel(mu0, cp0, a_s, b_s, cq), EmissionModel(h), y, r_inv)
            kf._chunks = chunks
            kf._keep_summaries = True
            out = kf._log_likelihood_per_series()
        ctx.save_for_backward(mu0, cp0, a_s, b_s, cq, h, y, r_inv)
        # the chunk summaries the forward elimination left in its workspace: the streamed backward starts from them
        ctx.fwd_summaries = kf._summaries
        return out

    @staticmethod
    def backward(ctx, grad_out):
        mu0, cp0, a_s, b_s, cq, h, y, r_inv = ctx.saved_tensors
        with torch.no_grad():
            streamed = _streamed_backward(mu0, cp0, a_s, b_s, cq, h, y, r_inv, grad_out, fwd=ctx.fwd_summaries,
                                          want=tuple(ctx.needs_input_grad[:8]))
            ctx.fwd_summaries = None
            if streamed is not None:
                return streamed + (None,)
            kf = _RawFilter(StateSpaceModel(mu0, cp0, a_s, b_s, cq), EmissionModel(h), y, r_inv)
            post = kf.posterior_state_space_model()
            means, covs, cross = post._moments(want_sub=True)  # one forward sweep (or two scans) of the posterior chain
            bsz, n, m, d = h.shape
            per_step = r_inv.dim() > 2
            if m > 16 * ((d + 15) // 16):
                # an observation dimension beyond the local kernels (register / row kernels: m <= 4; LDS tiles, which the C ABI
                # picks for more outputs at any d: m <= d rounded up to 16): the same closed forms as batched products on the
                # smoothed moments, which themselves come from the HIP kernels above.  BASELINE config 5 (d = 64, m = 32) takes
                # the tile kernel below (csrc/mf_biggrad_impl.hpp).
                return _local_gradients_dense(mu0, cp0, a_s, b_s, cq, h, y, r_inv, means, covs, cross, grad_out.reshape(bsz)) + (None,)
            g_mu0, g_cp0 = torch.empty_like(mu0), torch.empty_like(cp0)
            g_a, g_b, g_cq = torch.empty_like(a_s), torch.empty_like(b_s), torch.empty_like(cq)
            g_h, g_y = torch.empty_like(h), torch.empty_like(y)
            g_om = torch.empty((bsz, n, m, m), dtype=h.dtype, device=h.device)
            info = _lib.pivot_info(h.device)
            c = lambda t: _lib.ptr(t.contiguous())  # noqa: E731
            w = grad_out.reshape(bsz).contiguous()              # applied inside the kernel: no scaling passes over the outputs
            _lib.call("mf_kf_loglik_grad", h.dtype, bsz, n, d, m, c(mu0), c(cp0), c(a_s), c(b_s), c(cq), c(h), c(y), c(r_inv),
                      int(per_step), c(means), c(covs), c(cross), _lib.ptr(g_mu0), _lib.ptr(g_cp0), _lib.ptr(g_a),
                      _lib.ptr(g_b), _lib.ptr(g_cq), _lib.ptr(g_h), _lib.ptr(g_y), _lib.ptr(g_om), _lib.ptr(w), info,
                      _lib.stream_ptr(h.device))
            _lib.raise_on_info(info, "log_likelihood (backward)", h.device)
            # d/dR_k^-1 of  -1/2 E[r_k^T R_k^-1 r_k]  =  -1/2 Omega_k (already weighted); a shared precision collects every point.
            # The log-determinant of the precision lives in the constants, which torch differentiates outside this function.
            g_r_inv = -0.5 * g_om if per_step else -0.5 * _sum_over_points(g_om)
        return g_mu0, g_cp0, g_a, g_b, g_cq, g_h, g_y, g_r_inv, None


# few, long series (the condition under which posterior_state_space_model streams, below): the backward as five streamed passes
_GRAD_STREAMED = True
# (no upper limit on the batch: measured at B = 16384, T = 500 and B = 4096, T = 2000 the streamed passes beat one lane per
# series too - 10.6 -> 9.4 ms, 21.9 -> 9.2 ms before the forward's summaries were used; the constant is there for A/B timing)
_GRAD_STREAMED_MAX_SERIES = 1 << 40
_grad_prof_events = (None, None)     # optional hipEvent_t pair recorded around the kernels of the streamed backward (bench.py)


_GRAD_FROM_FORWARD = True            # start the streamed backward from the forward evaluation's chunk summaries when it left any


def _streamed_backward(mu0, cp0, a_s, b_s, cq, h, y, r_inv, grad_out, chunks=0, fwd=None, want=(True,) * 8):
    """``mf_kf_loglik_grad_streamed_*`` (csrc/mf_grad_lds.hpp): the smoothed marginals stay in registers.  ``None`` when the call
    is not that route's (short chains, d > 6, m > 3, unaligned views): the caller keeps the three-kernel route.
    ``fwd = (workspace, chunks per series, chunk length)`` of the forward ``mf_kf_loglik`` call on the same tensors, if its
    level-0 kernel was the streaming one: its chunk summaries replace the backward's own first two passes.
    ``want``: which of the eight inputs need a gradient (``ctx.needs_input_grad``) - those of b, H, y and the precision are not
    stored when nobody asked (their slots come back as ``None``)."""
    bsz, n, m, d = h.shape
    if not _GRAD_STREAMED or bsz < 1 or bsz >= _GRAD_STREAMED_MAX_SERIES or n <= 64:
        return None
    per_step = r_inv.dim() > 2
    lib = _lib.load()
    ws_bytes = int(lib.mf_kf_loglik_grad_streamed_workspace_bytes(bsz, n, d, m, int(per_step), h.element_size(), chunks))
    if ws_bytes == 0:
        return None
    tensors = [t.contiguous() for t in (mu0, cp0, a_s, b_s, cq, h, y, r_inv)]
    g_mu0, g_cp0 = torch.empty_like(tensors[0]), torch.empty_like(tensors[1])
    g_a, g_cq = torch.empty_like(tensors[2]), torch.empty_like(tensors[4])
    g_b = torch.empty_like(tensors[3]) if want[3] else None
    g_h = torch.empty_like(tensors[5]) if want[5] else None
    g_y = torch.empty_like(tensors[6]) if want[6] else None
    g_om = torch.empty((bsz, n, m, m), dtype=h.dtype, device=h.device) if want[7] else None
    if any(t.data_ptr() % 16 for t in (tensors[2], tensors[4], g_a, g_cq)):
        return None
    ws = _lib.workspace(ws_bytes, h.device)
    info = _lib.pivot_info(h.device)
    w = grad_out.reshape(bsz).contiguous()
    ev0, ev1 = _grad_prof_events
    fwd_ws, fwd_p, fwd_l = fwd if (fwd is not None and _GRAD_FROM_FORWARD) else (None, 0, 0)
    rc = _lib.call_rc("mf_kf_loglik_grad_streamed", h.dtype, bsz, n, d, m, *[_lib.ptr(t) for t in tensors], int(per_step),
                      _lib.ptr(w), _lib.ptr(g_mu0), _lib.ptr(g_cp0), _lib.ptr(g_a), _lib.ptr(g_b), _lib.ptr(g_cq), _lib.ptr(g_h),
                      _lib.ptr(g_y), _lib.ptr(g_om), _lib.ptr(ws), ws_bytes, info, chunks, _lib.ptr(fwd_ws), fwd_p, fwd_l,
                      ev0, ev1, _lib.stream_ptr(h.device))
    if rc == -101:
        return None
    _lib.check(rc, "mf_kf_loglik_grad_streamed")
    _lib.raise_on_info(info, "log_likelihood (backward)", h.device)
    g_r_inv = None if g_om is None else (-0.5 * g_om if per_step else -0.5 * _sum_over_points(g_om))
    return g_mu0, g_cp0, g_a, g_b, g_cq, g_h, g_y, g_r_inv


def _local_gradients_dense(mu0, cp0, a_s, b_s, cq, h, y, r_inv, means, covs, cross, w):
    """Fisher's identity, ``grad log p(y) = E_{x|y}[grad log p(x, y)]``, in closed form from the smoothed moments
    ``means [B,T,d]``, ``covs [B,T,d,d]``, ``cross = Cov(x_{k+1}, x_k) [B,T-1,d,d]`` - what ``mf_kf_loglik_grad_*`` evaluates
    with one lane per (series, time point) for d <= 9, here as batched matrix products for 10 <= d <= 64:

        e_k = x_{k+1} - A_k x_k - b_k,   E[e] = m_{k+1} - A m_k - b,   E[e x_k^T] = X_k - A S_k + E[e] m_k^T,
        Psi_k = E[e e^T] = S_{k+1} - A X_k^T - X_k A^T + A S_k A^T + E[e] E[e]^T,
        d/dA = Q^-1 E[e x^T],  d/db = Q^-1 E[e],  d/dC = tril(C^-T (C^-1 Psi C^-T - I)),           (Q = C C^T)
        r_k = y_k - H_k x_k:  d/dH = R^-1 (E[r] m^T - H S),  d/dy = -R^-1 E[r],  d/dR^-1 = -1/2 (E[r] E[r]^T + H S H^T).

    Returns the gradients of ``(mu0, cholP0, A, b, cholQ, H, y, R^-1)``, each weighted by the incoming ``w [B]``
    (reference: TensorFlow reverse mode through kalman_filter.py:184-255; pinned by
    tests/integration/models/test_variational.py:123-132 there)."""
    tri = torch.linalg.solve_triangular
    tr = lambda t: t.transpose(-1, -2)                                    # noqa: E731

    def chol_grad(chol, psi):
        """``tril(C^-T (C^-1 Psi C^-T - I))`` for symmetric ``psi``."""
        z = tri(chol, psi, upper=False)                                   # C^-1 Psi
        z = tri(chol, tr(z), upper=False)                                 # C^-1 Psi C^-T (symmetric)
        z = z - torch.eye(chol.shape[-1], dtype=chol.dtype, device=chol.device)
        return torch.tril(tri(tr(chol), z, upper=True))

    def q_inv(chol, rhs):
        return tri(tr(chol), tri(chol, rhs, upper=False), upper=True)

    wv, wm = w.reshape(-1, 1, 1), w.reshape(-1, 1, 1, 1)
    # the initial state
    d0 = means[:, 0] - mu0
    g_mu0 = q_inv(cp0, d0[..., None])[..., 0] * w.reshape(-1, 1)
    g_cp0 = chol_grad(cp0, covs[:, 0] + d0[..., :, None] * d0[..., None, :]) * wv
    # the transitions
    if a_s.shape[1] > 0:
        m_prev, m_next = means[:, :-1], means[:, 1:]
        s_prev, s_next = covs[:, :-1], covs[:, 1:]
        e_bar = m_next - (a_s @ m_prev[..., None])[..., 0] - b_s
        a_s_prev = a_s @ s_prev
        e_xt = cross - a_s_prev + e_bar[..., :, None] * m_prev[..., None, :]
        a_xt = a_s @ tr(cross)
        psi = s_next - a_xt - tr(a_xt) + a_s_prev @ tr(a_s) + e_bar[..., :, None] * e_bar[..., None, :]
        # one triangular solve pair for [E[e x^T] | E[e]]: a vector right-hand side would go to rocBLAS' trsv, which takes
        # 9 ms per call at config 5's shape against 0.7 ms for the d + 1 columns together
        both = q_inv(cq, torch.cat((e_xt, e_bar[..., None]), dim=-1))
        g_a = both[..., :-1] * wm
        g_b = both[..., -1] * wv
        g_cq = chol_grad(cq, psi) * wm
    else:
        g_a, g_b, g_cq = torch.zeros_like(a_s), torch.zeros_like(b_s), torch.zeros_like(cq)
    # the observations
    r_bar = y - (h @ means[..., None])[..., 0]
    h_s = h @ covs
    g_h = (r_inv @ (r_bar[..., :, None] * means[..., None, :] - h_s)) * wm
    g_y = -(r_inv @ r_bar[..., None])[..., 0] * wv
    g_om = (r_bar[..., :, None] * r_bar[..., None, :] + h_s @ tr(h)) * wm
    g_r_inv = -0.5 * g_om if r_inv.dim() > 2 else -0.5 * _sum_over_points(g_om)
    return g_mu0, g_cp0, g_a, g_b, g_cq, g_h, g_y, g_r_inv


class BaseKalmanFilter(abc.ABC):
    """Kalman filter over a ``StateSpaceModel`` and an ``EmissionModel`` (kalman_filter.py:32-271)."""

    def __init__(self, state_space_model: StateSpaceModel, emission_model: EmissionModel) -> None:
        self.prior_ssm = state_space_model
        self.emission = emission_model

    # tuning / measurement hooks (not part of the reference API): time partitions per series (0 = automatic) and
    # an optional pair of hipEvent_t handles recorded around the dominant kernel (used by bench.py)
    _chunks = 0
    _keep_summaries = False
    _summaries = None
    # the smoother after the filter: log_likelihood() leaves the chunk summaries of its level-0 kernel in its workspace; a
    # posterior_state_space_model() on the SAME inputs starts from them (mf_kf_posterior_chain_from_filter).  "Same" is decided on
    # the SOURCE tensors the filter holds (chain parameters, emission matrix, observations, observation covariance): identity,
    # storage, strides and autograd version of each - never on the flattened / broadcast copies handed to the kernels, which are
    # temporaries whose address the caching allocator recycles.  The cache entry keeps references to the
    # sources, so neither their ids nor their storage can be reused while it lives.  A write that bypasses torch's version counter
    # (`x.data.mul_()`, `set_()`) is invisible to that check: call invalidate_filter_cache() after one.
    _POST_FROM_FILTER = True
    _filter_cache = None

    def invalidate_filter_cache(self) -> None:
        self._filter_cache = None

    def _cache_sources(self):
        """The tensors this filter HOLDS that determine every kernel input, or None when some input is derived on the fly
        from tensors this class cannot name (sites): then nothing is cached."""
        return None

    def _chain_sources(self):
        s = self.prior_ssm
        return (s._mu_0, s._chol_P_0, s._A_s, s._b_s, s._chol_Q_s, self.emission.emission_matrix)

    @staticmethod
    def _source_key(sources):
        # identity + autograd version: the entry keeps the tensor OBJECTS alive, so an id cannot be handed to another tensor, and
        # every in-place write through torch bumps the version (shape / strides / storage of a live object only change through
        # `set_()` / `.data =`, which the docstring above sends to invalidate_filter_cache()).  Eight tensors cost ~2 us; the
        # (data_ptr, shape, stride, dtype, device) tuples this used to add cost 15 us on a 0.1-ms call
        return tuple((id(t), t._version) for t in sources)

    def _cache_key(self):
        sources = self._cache_sources()
        if sources is None:
            return None, None
        return self._source_key(sources), sources
    _prof_events = (None, None)
    _post_prof_events = (None, None)     # hipEvent_t pair around the kernels of posterior_state_space_model (bench.py)

    @property
    @abc.abstractmethod
    def _r_inv(self) -> torch.Tensor:
        """Precision of the observation model: ``[m, m]`` or ``[..., T, m, m]``."""

    @property
    @abc.abstractmethod
    def observations(self) -> torch.Tensor:
        """Observation vector ``batch_shape + [T, m]``."""

    @property
    def _r_inv_per_step(self) -> bool:
        return self._r_inv.dim() > 2

    @property
    def _k_inv_prior(self) -> SymmetricBlockTriDiagonal:
        return self.prior_ssm.precision

    def _expanded(self):
        """Emission matrix / observations / per-step precisions expanded to the chain's batch shape."""
        batch = tuple(self.prior_ssm.batch_shape)
        n, m, d = self.prior_ssm.num_transitions + 1, self.emission.output_dim, self.prior_ssm.state_dim
        h = self.emission.emission_matrix
        if tuple(h.shape[-3:]) != (n, m, d):
            raise ValueError(f"emission matrix has shape {tuple(h.shape)}, expected [..., {n}, {m}, {d}]")
        y = self.observations
        r_inv = self._r_inv                       # evaluated ONCE per call: the property is lazy and uncached
        _lib.same_dtype_device(self.prior_ssm.state_transitions, type(self).__name__, emission_matrix=h, observations=y,
                               observation_precision=r_inv)
        h = _flat(h.expand(batch + (n, m, d)), 3)
        y = _flat(y.expand(batch + (n, m)), 2)
        per_step = r_inv.dim() > 2
        if per_step:
            r_inv = _flat(r_inv.expand(batch + (n, m, m)), 3)
        else:
            r_inv = r_inv.contiguous()
        return h, y, r_inv, per_step

    @property
    def _k_inv_post(self) -> SymmetricBlockTriDiagonal:
        """Posterior precision ``K⁻¹ + GᵀΣ⁻¹G`` (kalman_filter.py:86-101)."""
        h, _, r_inv, per_step = self._expanded()
        diag, sub, _ = self.prior_ssm._precision_and_eta(h, None, r_inv, per_step, want_eta=False)
        return SymmetricBlockTriDiagonal(diag, sub)

    @property
    def _log_det_observation_precision(self) -> torch.Tensor:
        """``T · log|R⁻¹|`` (kalman_filter.py:103-107)."""
        num_data = self.prior_ssm.num_transitions + 1
        return num_data * torch.linalg.slogdet(self._r_inv)[1]

    # batches at least this large (or chains this short) take the fused one-lane-per-series sweep (mf_kf_posterior_chain
    # without a workspace); fewer, longer series the same sweep partitioned in time and streamed by LDS-DMA (with one)
    _POST_FUSED_MIN_SERIES = 2048
    _POST_FUSED_MAX_SERIAL_BLOCKS = 64
    _POST_STREAMED = True

    def _posterior_chain_fused(self, h, y, r_inv, per_step) -> Optional[StateSpaceModel]:
        mu0, cp0, a_s, b_s, cq = self.prior_ssm._flat_params()
        bsz, n, d, m = a_s.shape[0], self.prior_ssm.num_transitions + 1, self.prior_ssm.state_dim, h.shape[-2]
        lib = _lib.load()
        if d > lib.mf_max_state_dim() or m > 4 or bsz == 0:
            return None
        serial = bsz >= self._POST_FUSED_MIN_SERIES or n <= self._POST_FUSED_MAX_SERIAL_BLOCKS
        cache = self._filter_cache
        if (not serial and self._POST_STREAMED and self._POST_FROM_FILTER and cache is not None and m <= 3
                and (a_s.data_ptr() | cq.data_ptr()) % 16 == 0 and cache[0] == self._cache_key()[0]):
            # the smoother after the filter: start from the summaries log_likelihood() left behind
            wsb = int(lib.mf_kf_posterior_chain_from_filter_workspace_bytes(bsz, n, d, m, int(per_step), a_s.element_size(), cache[2]))
            if wsb:
                if a_s.is_cuda and cache[5] is not None and cache[5] != torch.cuda.current_stream(a_s.device):
                    torch.cuda.current_stream(a_s.device).wait_stream(cache[5])    # the summaries were written on another stream
                ws2 = _lib.workspace(wsb, a_s.device)
                outs = [torch.empty_like(x) for x in (a_s, mu0, b_s, cp0, cq)]
                info = _lib.pivot_info(a_s.device)
                rc = _lib.call_rc("mf_kf_posterior_chain_from_filter", a_s.dtype, bsz, n, d, m, _lib.ptr(mu0), _lib.ptr(cp0),
                                  _lib.ptr(a_s), _lib.ptr(b_s), _lib.ptr(cq), _lib.ptr(h), _lib.ptr(y), _lib.ptr(r_inv), int(per_step),
                                  *[_lib.ptr(x) for x in outs], _lib.ptr(ws2), wsb, info, _lib.ptr(cache[1]), cache[2], cache[3],
                                  self._post_prof_events[0], self._post_prof_events[1], _lib.stream_ptr(a_s.device))
                if rc != -101:
                    _lib.check(rc, "mf_kf_posterior_chain_from_filter")
                    _lib.raise_on_info(info, "posterior_state_space_model", a_s.device)
                    a_p, mu0_p, b_p, cp0_p, cq_p = outs
                    batch = tuple(self.prior_ssm.batch_shape)
                    return StateSpaceModel(initial_mean=mu0_p.reshape(batch + (d,)), chol_initial_covariance=cp0_p.reshape(batch + (d, d)),
                                           state_transitions=a_p.reshape(batch + (n - 1, d, d)),
                                           state_offsets=b_p.reshape(batch + (n - 1, d)),
                                           chol_process_covariances=cq_p.reshape(batch + (n - 1, d, d)))
        ws, ws_bytes = None, 0
        if not serial and self._POST_STREAMED and (a_s.data_ptr() | cq.data_ptr()) % 16 == 0:
            ws_bytes = int(lib.mf_kf_posterior_chain_workspace_bytes(bsz, n, d, m, int(per_step), a_s.element_size(),
                                                                     self._chunks))
            if ws_bytes:
                ws = torch.empty(ws_bytes, dtype=torch.uint8, device=a_s.device)
        if not serial and ws is None:
            return None
        a_p, b_p, cq_p = torch.empty_like(a_s), torch.empty_like(b_s), torch.empty_like(cq)
        mu0_p, cp0_p = torch.empty_like(mu0), torch.empty_like(cp0)
        info = _lib.pivot_info(a_s.device)
        _lib.call("mf_kf_posterior_chain", a_s.dtype, bsz, n, d, m, _lib.ptr(mu0), _lib.ptr(cp0), _lib.ptr(a_s), _lib.ptr(b_s),
                  _lib.ptr(cq), _lib.ptr(h), _lib.ptr(y), _lib.ptr(r_inv), int(per_step), _lib.ptr(a_p), _lib.ptr(mu0_p),
                  _lib.ptr(b_p), _lib.ptr(cp0_p), _lib.ptr(cq_p), _lib.ptr(ws), ws_bytes, info, self._chunks,
                  self._post_prof_events[0], self._post_prof_events[1], _lib.stream_ptr(a_s.device))
        _lib.raise_on_info(info, "posterior_state_space_model", a_s.device)
        batch = tuple(self.prior_ssm.batch_shape)
        return StateSpaceModel(initial_mean=mu0_p.reshape(batch + (d,)), chol_initial_covariance=cp0_p.reshape(batch + (d, d)),
                               state_transitions=a_p.reshape(batch + (n - 1, d, d)), state_offsets=b_p.reshape(batch + (n - 1, d)),
                               chol_process_covariances=cq_p.reshape(batch + (n - 1, d, d)))

    def posterior_state_space_model(self) -> StateSpaceModel:
        """Posterior as a state space model (kalman_filter.py:109-182)."""
        h, y, r_inv, per_step = self._expanded()
        if _ag.needs_grad(*self._chain_sources(), h, y, r_inv):
            return self._posterior_differentiable(h, y, r_inv, per_step)
        fused = self._posterior_chain_fused(h, y, r_inv, per_step)
        if fused is not None:
            return fused
        # posterior precision and  GᵀΣ⁻¹y + K⁻¹μ  (kalman_filter.py:149-156) in one parallel kernel
        diag, sub, eta = self.prior_ssm._precision_and_eta(h, y, r_inv, per_step, want_eta=True)
        # backward UDUᵀ sweep, m_post and chol(Δ⁻¹) fused (kalman_filter.py:159-174)
        if _lib.small_state_dim(self.prior_ssm.state_dim, int(math.prod(self.prior_ssm.batch_shape)),
                                self.prior_ssm.num_transitions + 1, diag.element_size()):
            # written by the kernels in the layout of the chain: transitions -Uᵀ, (mu0', b'), (cholP0', cholQ') - no slices,
            # copies or sign flips over the [B, T, d, d] tensors afterwards
            a_post, _, (mu0, offsets), (chol_p0, chol_q) = SymmetricBlockTriDiagonal(diag, sub)._udl(eta, chain=True)
            return StateSpaceModel(initial_mean=mu0, chol_initial_covariance=chol_p0, state_transitions=a_post,
                                   state_offsets=offsets, chol_process_covariances=chol_q)
        u_t, _, m_post, chol_dinv = SymmetricBlockTriDiagonal(diag, sub)._udl(eta)
        return StateSpaceModel(
            initial_mean=m_post[..., 0, :],
            chol_initial_covariance=chol_dinv[..., 0, :, :],
            state_transitions=-u_t,
            state_offsets=m_post[..., 1:, :],
            chol_process_covariances=chol_dinv[..., 1:, :, :],
        )

    def _posterior_differentiable(self, h, y, r_inv, per_step) -> StateSpaceModel:
        """The same chain (kalman_filter.py:109-182) as a composition of the DIFFERENTIABLE operators - the reference differentiates
        through this method under a tape: precision (state_space_model.py:431-483), ``U D U^T`` as the Cholesky factorisation of the
        time-reversed posterior precision (``SymmetricBlockTriDiagonal._udl_differentiable``; adjoint: ``mf_btd_cholesky_grad_*``),
        the two solves of kalman_filter.py:159-166 (``mf_btd_solve_*`` and its adjoint) and per-block Cholesky factors."""
        ssm = self.prior_ssm
        batch, n, d = tuple(ssm.batch_shape), ssm.num_transitions + 1, ssm.state_dim
        m = h.shape[-2]
        hb, yb = h.reshape(batch + (n, m, d)), y.reshape(batch + (n, m))
        rb = r_inv.reshape(batch + (n, m, m)) if per_step else r_inv
        prec = ssm.precision
        rh = rb @ hb                                                                       # R^-1 H
        diag = prec.block_diagonal + hb.transpose(-1, -2) @ rh                             # kalman_filter.py:86-101
        obs_proj = (rh.transpose(-1, -2) @ yb[..., None])[..., 0]                          # H^T R^-1 y
        # K^-1 mu = A^-T Q^-1 [mu0, b_0, ...]  (mu = A^-1 [mu0, b]; kalman_filter.py:153-156 without forming mu)
        qm = _ag.chol_solve_blocks(ssm.concatenated_cholesky_process_covariance, ssm.concatenated_state_offsets[..., None])[..., 0]
        eta = obs_proj + qm
        eye = torch.eye(d, dtype=diag.dtype, device=diag.device).expand(diag.shape).contiguous()
        back = (ssm.state_transitions.transpose(-1, -2) @ qm[..., 1:, :, None])[..., 0]
        eta = eta - torch.cat([back, torch.zeros_like(back[..., :1, :])], dim=-2)
        u_t, chol_d = SymmetricBlockTriDiagonal(diag, prec.block_sub_diagonal)._udl_differentiable()
        x = LowerTriangularBlockTriDiagonal(eye, u_t).solve(eta, transpose_left=True)      # kalman_filter.py:159-162
        m_post = _ag.chol_solve_blocks(chol_d, x[..., None])[..., 0]                       # kalman_filter.py:164-166
        d_inv = _ag.chol_solve_blocks(chol_d, eye)
        chol_dinv = _lib.checked_cholesky(0.5 * (d_inv + d_inv.transpose(-1, -2)), "posterior_state_space_model")
        return StateSpaceModel(initial_mean=m_post[..., 0, :], chol_initial_covariance=chol_dinv[..., 0, :, :],
                               state_transitions=-u_t, state_offsets=m_post[..., 1:, :],
                               chol_process_covariances=chol_dinv[..., 1:, :, :])

    def _constant_terms(self, num_points: int) -> torch.Tensor:
        """``cst + ½ log|Σ⁻¹|`` (kalman_filter.py:229-231,249-253), shape [] or batch_shape."""
        cst = -0.5 * math.log(2 * math.pi) * (self.emission.output_dim * num_points)
        return cst + 0.5 * self._log_det_observation_precision

    def _log_likelihood_per_series(self, expanded=None) -> torch.Tensor:
        """Per-series log-likelihood WITHOUT the chain-independent constant terms, shape [B]."""
        mu0, cp0, a_s, b_s, cq = self.prior_ssm._flat_params()
        h, y, r_inv, per_step = expanded if expanded is not None else self._expanded()
        bsz, n, d, m = a_s.shape[0], self.prior_ssm.num_transitions + 1, self.prior_ssm.state_dim, h.shape[-2]
        lib = _lib.load()
        esz = a_s.element_size()
        if bsz == 0:                                   # empty batch: nothing to launch
            return torch.empty(0, dtype=a_s.dtype, device=a_s.device)
        chunks = self._chunks
        aligned = int((a_s.data_ptr() | cq.data_ptr()) % 16 == 0)
        path, p_f, l_f = ctypes.c_int(-1), ctypes.c_int64(0), ctypes.c_int64(0)
        # which level-0 kernel runs, on which time partition: its chunk summaries are the first thing in the workspace, which stays
        # alive for whoever starts from them (the autograd function's backward; posterior_state_space_model, see _POST_FROM_FILTER)
        plan = lambda c: lib.mf_kf_loglik_plan(bsz, n, d, m, int(per_step), esz, c, aligned, ctypes.byref(path),       # noqa: E731
                                               ctypes.byref(p_f), ctypes.byref(l_f))
        planned = plan(chunks) == 0
        if self._keep_summaries and planned and chunks == 0 and path.value == 2 and p_f.value == 1 and n > 64:
            # (so many series that one chunk each fills the chip: two chunks cost the forward 6 % and save the backward its
            # first two passes - measured at B = 65536, T = 128: forward 1.52 -> 1.61 ms, backward 10.3 -> 8.6 ms)
            chunks = 2
            plan(chunks)
        ws_bytes = int(lib.mf_kf_loglik_workspace_bytes(bsz, n, d, esz, chunks))
        if ws_bytes == 0:
            _lib.check(-100, "mf_kf_loglik")
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=a_s.device)
        out = torch.empty(bsz, dtype=a_s.dtype, device=a_s.device)
        info = _lib.pivot_info(a_s.device)
        _lib.call("mf_kf_loglik", a_s.dtype, bsz, n, d, m, _lib.ptr(mu0), _lib.ptr(cp0), _lib.ptr(a_s),
                  _lib.ptr(b_s), _lib.ptr(cq), _lib.ptr(h), _lib.ptr(y), _lib.ptr(r_inv), int(per_step),
                  0.0, _lib.ptr(out), _lib.ptr(ws), ws_bytes, info, chunks, self._prof_events[0],
                  self._prof_events[1], _lib.stream_ptr(a_s.device))
        _lib.raise_on_info(info, "KalmanFilter.log_likelihood", a_s.device, blocks=n)
        usable = planned and path.value == 2 and p_f.value >= 2
        if self._keep_summaries:
            self._summaries = (ws, int(p_f.value), int(l_f.value)) if usable else None
        # (kept only where posterior_state_space_model would take the streamed kernels: the workspace stays alive with the filter)
        streamed_post = bsz < self._POST_FUSED_MIN_SERIES and n > self._POST_FUSED_MAX_SERIAL_BLOCKS and m <= 3
        key, sources = self._cache_key() if (usable and self._POST_FROM_FILTER and streamed_post) else (None, None)
        stream = torch.cuda.current_stream(a_s.device) if (key is not None and a_s.is_cuda) else None
        self._filter_cache = (key, ws, int(p_f.value), int(l_f.value), sources, stream) if key is not None else None
        return out

    def _per_series(self):
        """``(per-series values [B], differentiable?)``: through the autograd function when any tensor of the model (chain,
        emission, observations, observation precision / sites) requires a gradient, else the plain fused call.  The
        emission / observation tensors are expanded (and the lazy observation precision evaluated) once."""
        expanded = self._expanded()
        if torch.is_grad_enabled():
            tensors = self.prior_ssm._flat_params() + expanded[:3]
            if any(t.requires_grad for t in tensors):
                return _LogLikelihoodPerSeries.apply(*tensors, self._chunks), True
        return self._log_likelihood_per_series(expanded), False

    def log_likelihood(self) -> torch.Tensor:
        """Log marginal likelihood, summed over ``batch_shape`` (kalman_filter.py:184-255)."""
        num_data = self.prior_ssm.num_transitions + 1
        per_series, differentiable = self._per_series()
        if not differentiable:
            fused = self._fused_total(per_series, num_data)
            if fused is not None:
                return _lib.checked(fused)
        per_series = per_series.reshape(tuple(self.prior_ssm.batch_shape))
        return _lib.checked(torch.sum(per_series + self._constant_terms(num_data)))

    def _total_terms(self):
        """``(chol_obs | None, extra device scalar | None)`` for ``mf_kf_loglik_total``: how this filter's
        ``½ log|Σ⁻¹|`` term is handed to the kernel; ``None`` = no fused form, fall back to the torch expression."""
        return None

    def _fused_total(self, per_series: torch.Tensor, num_points: int) -> Optional[torch.Tensor]:
        """``Σ_batch (per_series + constant terms)`` in one kernel instead of ~10 elementwise / reduce launches (each
        ≈6 µs of dispatch on this GPU: 10 % of a B=1024, T=10000 evaluation)."""
        terms = self._total_terms()
        if terms is None:
            return None
        chol_obs, extra = terms
        out = torch.empty((), dtype=per_series.dtype, device=per_series.device)
        m = self.emission.output_dim
        cst = -0.5 * math.log(2 * math.pi) * (m * num_points)
        _lib.call("mf_kf_loglik_total", per_series.dtype, per_series.numel(), _lib.ptr(per_series), m,
                  _lib.ptr(None if chol_obs is None else chol_obs.contiguous()), num_points,
                  _lib.ptr(None if extra is None else extra.contiguous()), cst, _lib.ptr(out),
                  _lib.stream_ptr(per_series.device))
        return out

    def _back_project_y_to_state(self, observations: torch.Tensor) -> torch.Tensor:
        """``(GᵀΣ⁻¹) y`` (kalman_filter.py:257-271)."""
        back = torch.einsum("...ij,...ki->...kj", self.emission.emission_matrix, self._r_inv)
        return torch.einsum("...ij,...i->...j", back, observations)


class KalmanFilter(BaseKalmanFilter):
    """Kalman filter with one observation covariance shared by all time points (kalman_filter.py:275-353)."""

    def __init__(
        self,
        state_space_model: StateSpaceModel,
        emission_model: EmissionModel,
        observations: torch.Tensor,
        chol_obs_covariance: torch.Tensor,
    ) -> None:
        """
        :param observations: ``batch_shape + [num_transitions + 1, output_dim]``.
        :param chol_obs_covariance: ``[output_dim, output_dim]`` Cholesky of the observation covariance.
        """
        super().__init__(state_space_model, emission_model)
        assert isinstance(observations, torch.Tensor)   # kalman_filter.py:318 (tensor, not ndarray)
        m = emission_model.output_dim
        if tuple(chol_obs_covariance.shape) != (m, m):
            raise ValueError("The shape of the observation covariance matrix and the emission matrix are not compatible")
        shape = tuple(state_space_model.batch_shape) + (state_space_model.num_transitions + 1, m)
        if tuple(observations.shape) != shape:
            raise ValueError("The shape of the observations and the state-space-model parameters are not compatible")
        _lib.same_dtype_device(state_space_model.state_transitions, "KalmanFilter", observations=observations,
                               chol_obs_covariance=chol_obs_covariance, emission_matrix=emission_model.emission_matrix)
        self._chol_obs_covariance = chol_obs_covariance
        self._observations = observations

    def _cache_sources(self):
        return self._chain_sources() + (self._observations, self._chol_obs_covariance)

    @property
    def _r_inv(self) -> torch.Tensor:
        """``R⁻¹ = (chol cholᵀ)⁻¹`` (kalman_filter.py:341-348)."""
        chol = self._chol_obs_covariance
        m = self.emission.output_dim
        differentiable = chol.requires_grad and torch.is_grad_enabled()
        if chol.shape[-1] == 1 and (differentiable or not chol.is_cuda):
            return chol.pow(-2)                   # (two element-wise launches: square, reciprocal)
        if chol.is_cuda and m <= 32 and not differentiable:
            # one launch (mf_obs_precision_from_chol) instead of an identity + two triangular solves = eleven small kernels
            out = torch.empty_like(chol, memory_format=torch.contiguous_format)
            info = _lib.pivot_info(chol.device)
            _lib.call("mf_obs_precision_from_chol", chol.dtype, m, _lib.ptr(chol.contiguous()), _lib.ptr(out), info,
                      _lib.stream_ptr(chol.device))
            # (every caller evaluates a factorising kernel of its own right behind this one: that launch queues the flag copy)
            _lib.raise_on_info(info, "KalmanFilter (observation precision)", chol.device, more_follow=True)
            return out
        eye = torch.eye(m, dtype=chol.dtype, device=chol.device)
        return _lib.chol_solve(chol, eye.expand(chol.shape))               # differentiable route

    @property
    def _log_det_observation_precision(self) -> torch.Tensor:
        """``T · log|R⁻¹| = −2 T Σ log diag(chol R)`` (kalman_filter.py:103-107): read off the Cholesky factor that is
        already given instead of an LU-based slogdet of the inverse (a dozen launch-bound kernels per call)."""
        num_data = self.prior_ssm.num_transitions + 1
        diag = torch.diagonal(self._chol_obs_covariance, dim1=-2, dim2=-1)
        return (-2.0 * num_data) * torch.sum(torch.log(torch.abs(diag)))

    @property
    def observations(self) -> torch.Tensor:
        return self._observations

    def _total_terms(self):
        return self._chol_obs_covariance, None


class _RawFilter(BaseKalmanFilter):
    """Filter over flat tensors with the observation precision given as is (shared ``[m, m]`` or per step ``[B, T, m, m]``):
    what the autograd functions evaluate inside."""

    def __init__(self, state_space_model, emission_model, observations, r_inv):
        super().__init__(state_space_model, emission_model)
        self._obs, self._precision = observations, r_inv

    @property
    def _r_inv(self):
        return self._precision

    @property
    def observations(self):
        return self._obs


class GaussianSites(abc.ABC):
    """Parameters of independent Gaussian sites (kalman_filter.py:356-379)."""

    @property
    def means(self):
        raise NotImplementedError

    @property
    def precisions(self):
        raise NotImplementedError

    @property
    def log_det_precisions(self):
        raise NotImplementedError


class UnivariateGaussianSitesNat(GaussianSites):
    """Univariate Gaussian sites in natural parameters (kalman_filter.py:382-433)."""

    def __init__(self, nat1: torch.Tensor, nat2: torch.Tensor, log_norm: Optional[torch.Tensor] = None):
        """:param nat1: ``[N, 1]``; :param nat2: ``[N, 1, 1]``; :param log_norm: ``[N, 1]`` or None."""
        if nat1.dim() != 2 or nat1.shape[-1] != 1:
            raise ValueError(f"nat1 must have shape [N, 1], got {tuple(nat1.shape)}")
        if tuple(nat2.shape) != (nat1.shape[0], 1, 1):
            raise ValueError(f"nat2 must have shape [N, 1, 1], got {tuple(nat2.shape)}")
        if log_norm is not None and tuple(log_norm.shape) != (nat1.shape[0], 1):
            raise ValueError(f"log_norm must have shape [N, 1], got {tuple(log_norm.shape)}")
        self.num_data, self.output_dim = nat1.shape
        self.nat1 = nat1
        self.nat2 = nat2
        self.log_norm = log_norm

    @property
    def means(self):
        return -0.5 * self.nat1 / self.nat2[..., 0]

    @property
    def precisions(self):
        return -2 * self.nat2

    @property
    def log_det_precisions(self):
        return torch.log(-2 * self.nat2)


class KalmanFilterWithSites(BaseKalmanFilter):
    """Kalman filter with time-dependent Gaussian sites (kalman_filter.py:437-497)."""

    def __init__(self, state_space_model: StateSpaceModel, emission_model: EmissionModel, sites: GaussianSites) -> None:
        if sites.output_dim != emission_model.output_dim:
            raise ValueError("The shape of the site matrices and the emission matrix are not compatible")
        self.sites = sites
        super().__init__(state_space_model, emission_model)

    @property
    def _r_inv(self):
        return self.sites.precisions

    @property
    def _log_det_observation_precision(self):
        """``Σ_k log|R_k⁻¹|`` (kalman_filter.py:490-492).  Univariate sites: an element-wise log instead of a batched LU."""
        r_inv = self._r_inv
        if r_inv.shape[-1] == 1:
            return torch.sum(torch.log(torch.abs(r_inv[..., 0, 0])), dim=-1)
        return torch.sum(torch.linalg.slogdet(r_inv)[1], dim=-1)

    def _total_terms(self):
        extra = 0.5 * self._log_det_observation_precision
        return (None, extra.reshape(1)) if extra.numel() == 1 else None

    @property
    def observations(self):
        return self.sites.means


class KalmanFilterWithSparseSites(BaseKalmanFilter):
    """Kalman filter with Gaussian sites observed on a subset of a time grid (kalman_filter.py:501-626)."""

    def __init__(self, state_space_model: StateSpaceModel, emission_model: EmissionModel, sites: GaussianSites,
                 num_grid_points: int, observations_index: torch.Tensor, observations: torch.Tensor):
        """
        :param num_grid_points: number of grid points.
        :param observations_index: ``[N, 1]`` int64 positions of the observations in the grid.
        :param observations: ``[n_batch] + [N, output_dim]`` sparse observations.
        """
        self.sites = sites
        self.observations_index = observations_index
        self.sparse_observations = self._drop_batch_shape(observations)
        self.grid_shape = (num_grid_points, 1)
        super().__init__(state_space_model, emission_model)

    @property
    def _r_inv(self):
        return self.sparse_to_dense(self.sites.precisions, output_shape=self.grid_shape + (1,))

    def _drop_batch_shape(self, tensor: torch.Tensor):
        """Check the batch, if present, is 1 and drop it (kalman_filter.py:531-539)."""
        if tensor.dim() < 3:
            return tensor
        if tensor.shape[0] != 1:
            raise Exception("KalmanFilterWithSparseSites doesn't support batches")
        return tensor.squeeze(0)

    @property
    def _log_det_observation_precision(self):
        r_inv = self._r_inv_data
        if r_inv.shape[-1] == 1:                      # univariate sites: no batched LU
            return torch.sum(torch.log(torch.abs(r_inv[..., 0, 0])), dim=-1)
        return torch.sum(torch.linalg.slogdet(r_inv)[1], dim=-1)

    @property
    def observations(self):
        return self.sparse_to_dense(self.sparse_observations, self.grid_shape)

    @property
    def _r_inv_data(self):
        return self.sites.precisions

    def sparse_to_dense(self, tensor: torch.Tensor, output_shape) -> torch.Tensor:
        """Scatter onto the grid (the tf.scatter_nd of kalman_filter.py:561-565; duplicates add)."""
        out = torch.zeros(tuple(output_shape), dtype=tensor.dtype, device=tensor.device)
        idx = self.observations_index.reshape(-1).to(device=tensor.device, dtype=torch.long)
        return out.index_add_(0, idx, tensor.reshape((idx.shape[0],) + tuple(output_shape[1:])))

    def dense_to_sparse(self, tensor: torch.Tensor) -> torch.Tensor:
        """Gather from the grid (kalman_filter.py:567-577)."""
        expand_dims = tensor.dim() == 3
        idx = self.observations_index.reshape(-1).to(device=tensor.device, dtype=torch.long)
        out = tensor.reshape(-1, 1)[idx]
        if expand_dims:
            out = out[..., None]
        return out

    def log_likelihood(self) -> torch.Tensor:
        """Log marginal likelihood; only observed points count in the constants (kalman_filter.py:579-626)."""
        per_series, _ = self._per_series()
        num_data = self.observations_index.shape[0]
        return torch.sum(per_series + self._constant_terms(num_data))
