"""
Thin model harness over the Kalman path: ``GaussianProcessRegression`` (mirror of
``markovflow/models/gaussian_process_regression.py:29-160``, the direct caller of ``KalmanFilter`` in the reference).
Only what drives the hot path is mirrored: construction from ``(time_points, observations)`` and an SDE kernel,
``log_likelihood`` / ``loss``, the posterior state space model and ``posterior`` (prediction at new time points,
``markovflow_amd/posterior.py``); mean functions and training loops belong to the reference's outer layers.
"""
from typing import Optional, Tuple

import ctypes
import math

import torch

from . import _lib
from .kalman_filter import KalmanFilter
from .kernels import IndependentMultiOutput, SDEKernel
from .posterior import AnalyticPosteriorProcess
from .state_space_model import StateSpaceModel



def _gpr_partition(bsz: int, nt: int, chunks: int, lanes: int = 65536):
    """``(chunks to ask for, chunks per series P, transitions per chunk L)`` of the fused GPR forward - THE one place the Python
    side derives it (forward, backward and posterior chain all start from the same summaries).  Mirrors
    ``lds_partition()`` of csrc/mf_inst.hip for an EXPLICIT chunk count, which is what every caller passes down; the C side
    re-checks that ``(P, L)`` tile the ``nt`` transitions and refuses (-101) otherwise."""
    want = chunks if chunks > 0 else max(1, min(-(-lanes // max(bsz, 1)), max(nt // 4, 1)))
    want = max(1, min(want, nt))
    length = -(-nt // want)
    return want, -(-nt // length), length


class _GprFusedLogLik(torch.autograd.Function):
    """Per-series log-likelihood of GP regression (without the chain-independent constants) as a differentiable function of the
    stacked hyper-parameters ``lam`` (= sqrt(order) / lengthscale), ``var`` and the noise precision, with the kernel -> state space
    model step fused into BOTH directions: forward ``mf_gpr_matern_loglik_*`` on an explicit time partition (its chunk summaries
    stay in the workspace), backward ``mf_gpr_matern_loglik_grad_*`` (csrc/mf_gpr_grad.hpp: boundary scans on those summaries, an
    emit pass and a gradient pass that generate the transitions in registers) followed by the generator's backward
    (``mf_sde_matern_transitions_grad_packed_*``, and ``mf_sde_matern_prior_chol_grad_*`` for the stationary prior's factor, which
    the kernels generate themselves), which reduces the gradients of the transitions to the hyper-parameters.  Reference: TensorFlow reverse mode through
    models/gaussian_process_regression.py:150-160, kernels/matern.py, sde_kernel.py:421-446."""

    @staticmethod
    def forward(ctx, lam_t, var_t, rinv, t, y, orders, per_series, jitter, chunks):
        bsz, n = t.shape
        d = sum((o + 1) // 2 for o in orders)
        nt = n - 1
        lib = _lib.load()
        want, parts, length = _gpr_partition(bsz, nt, chunks)
        ws_bytes = int(lib.mf_kf_loglik_workspace_bytes(bsz, n, d, t.element_size(), want))
        ws = _lib.workspace(ws_bytes, t.device)
        out = torch.empty(bsz, dtype=t.dtype, device=t.device)
        info = _lib.pivot_info(t.device)
        c_orders = (ctypes.c_int * len(orders))(*orders)
        lam_c, var_c, rinv_c = lam_t.detach().contiguous(), var_t.detach().contiguous(), rinv.detach().contiguous()
        rc = _lib.call_rc("mf_gpr_matern_loglik", t.dtype, bsz, n, len(orders), c_orders, _lib.ptr(lam_c), _lib.ptr(var_c),
                          int(per_series), _lib.ptr(t), _lib.ptr(y), _lib.ptr(rinv_c), jitter, 0.0, _lib.ptr(out), _lib.ptr(ws),
                          ws_bytes, info, want, None, None, _lib.stream_ptr(t.device))
        _lib.check(rc, "mf_gpr_matern_loglik")
        _lib.raise_on_info(info, "GaussianProcessRegression.log_likelihood", t.device)
        ctx.save_for_backward(lam_c, var_c, rinv_c, t, y)
        ctx.meta = (orders, per_series, jitter, d)
        ctx.fwd = (ws, parts, length)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lam_c, var_c, rinv_c, t, y = ctx.saved_tensors
        orders, per_series, jitter, d = ctx.meta
        # (the summaries are only READ by the backward kernels and live as long as the graph node: a second backward through the
        # same node - retain_graph=True, per-parameter gradient loops - starts from them again)
        ws_f, parts, length = ctx.fwd
        bsz, n = t.shape
        lib = _lib.load()
        c_orders = (ctypes.c_int * len(orders))(*orders)
        with torch.no_grad():
            # one packed record per transition: the diagonal blocks of g_A | the lower triangles of those of g_cholQ (16-B padded parts)
            esz = t.element_size()
            sizes = [(o + 1) // 2 for o in orders]
            pad = lambda nel: -(-nel * esz // 16) * 16 // esz                                         # noqa: E731
            rec = pad(sum(k * k for k in sizes)) + pad(sum(k * (k + 1) // 2 for k in sizes))
            g_packed = torch.empty((bsz, n - 1, rec), dtype=t.dtype, device=t.device)
            g_cp0 = torch.empty((bsz, d, d), dtype=t.dtype, device=t.device)
            g_om = torch.empty((bsz, n), dtype=t.dtype, device=t.device)
            ws_bytes = int(lib.mf_gpr_matern_loglik_grad_workspace_bytes(bsz, n, d, t.element_size(), parts))
            ws = _lib.workspace(ws_bytes, t.device)
            info = _lib.pivot_info(t.device)
            w = grad_out.reshape(bsz).contiguous()
            rc = _lib.call_rc("mf_gpr_matern_loglik_grad", t.dtype, bsz, n, len(orders), c_orders, _lib.ptr(lam_c), _lib.ptr(var_c),
                              int(per_series), _lib.ptr(t), _lib.ptr(y), _lib.ptr(rinv_c), jitter, _lib.ptr(w), _lib.ptr(g_packed),
                              _lib.ptr(g_cp0), _lib.ptr(g_om), _lib.ptr(ws), ws_bytes, info, _lib.ptr(ws_f), parts, length,
                              _lib.stream_ptr(t.device))
            _lib.check(rc, "mf_gpr_matern_loglik_grad")
            _lib.raise_on_info(info, "GaussianProcessRegression.log_likelihood (backward)", t.device)
            dt = (t[:, 1:] - t[:, :-1]).contiguous()
            part = torch.empty((bsz, n - 1, len(orders), 2), dtype=t.dtype, device=t.device)
            _lib.call("mf_sde_matern_transitions_grad_packed", t.dtype, bsz, n - 1, len(orders), c_orders, _lib.ptr(lam_c),
                      _lib.ptr(var_c), int(per_series), _lib.ptr(dt), jitter, _lib.ptr(g_packed), _lib.ptr(part),
                      _lib.stream_ptr(t.device))
            prior = torch.empty((bsz, len(orders), 2), dtype=t.dtype, device=t.device)
            _lib.call("mf_sde_matern_prior_chol_grad", t.dtype, bsz, len(orders), c_orders, _lib.ptr(lam_c), _lib.ptr(var_c),
                      int(per_series), jitter, _lib.ptr(g_cp0), _lib.ptr(prior), _lib.stream_ptr(t.device))
            g = torch.sum(part, dim=1) + prior                           # [B, ncomp, 2]
            if not per_series:
                g = torch.sum(g, dim=0)
            g_rinv = (-0.5 * torch.sum(g_om)).reshape(1, 1)
        return g[..., 0].contiguous(), g[..., 1].contiguous(), g_rinv, None, None, None, None, None, None


class GaussianProcessRegression:
    """GP regression as a Kalman filter on the kernel's state space model (gaussian_process_regression.py:29-160)."""

    def __init__(self, input_data: Tuple[torch.Tensor, torch.Tensor], kernel: SDEKernel,
                 chol_obs_covariance: Optional[torch.Tensor] = None) -> None:
        """
        :param input_data: ``(time_points [batch + [num_data]], observations [batch + [num_data, observation_dim]])``.
        :param chol_obs_covariance: ``[observation_dim, observation_dim]`` Cholesky factor of the noise covariance
            (default: identity, as in the reference).
        """
        time_points, observations = input_data
        obs_dim = observations.shape[-1]
        if chol_obs_covariance is None:
            chol_obs_covariance = torch.eye(obs_dim, dtype=observations.dtype, device=observations.device)
        if tuple(chol_obs_covariance.shape) != (obs_dim, obs_dim):
            raise ValueError("chol_obs_covariance must have shape [observation_dim, observation_dim]")
        if tuple(time_points.shape) != tuple(observations.shape[:-1]):
            raise ValueError("time_points must have shape observations.shape[:-1]")
        # one dtype and one device for the whole model, as the reference's default_float() (a float32 buffer handed to a
        # float64 kernel would be read past its end)
        _lib.same_dtype_device(observations, "GaussianProcessRegression", time_points=time_points,
                               chol_obs_covariance=chol_obs_covariance)
        self._kernel = kernel
        self._time_points = time_points
        self._observations = observations
        self._chol_obs_covariance = chol_obs_covariance

    @property
    def time_points(self) -> torch.Tensor:
        return self._time_points

    @property
    def observations(self) -> torch.Tensor:
        return self._observations

    @property
    def kernel(self) -> SDEKernel:
        return self._kernel

    @property
    def _kalman(self) -> KalmanFilter:
        """gaussian_process_regression.py:112-124 (no mean function: residuals = observations)."""
        return KalmanFilter(
            state_space_model=self._kernel.state_space_model(self._time_points),
            emission_model=self._kernel.generate_emission_model(self._time_points),
            observations=self._observations,
            chol_obs_covariance=self._chol_obs_covariance,
        )

    # time partitions per series of the fused kernel (0 = automatic) and optional hipEvent_t pair around its level-0 kernel
    _chunks = 0
    _prof_events = (None, None)
    fused = True    # set False to force the materialised route (kernel tensors -> KalmanFilter)

    def invalidate_hyperparameter_cache(self) -> None:
        """Forget the hyper-parameter tensors the fused route keeps between calls.  They are re-derived automatically when a
        source tensor (a lengthscale, a variance, the noise factor) is REPLACED or written in place through torch (tensor identity
        + autograd version counter); a write that bypasses the version counter - ``lengthscale.data.clamp_()``, ``set_()`` - is
        invisible to that check: call this after such a write."""
        self._fused_cache = None
        for comp in self._kernel._components():
            comp._lambda_cached = None          # (the components keep sqrt(order) / lengthscale per lengthscale version)

    fused_backward = True    # set False to force the materialised route whenever a gradient is required

    def _fused_differentiable(self, comps, multi, rows, m, d) -> Optional[torch.Tensor]:
        """The fused route when a hyper-parameter or the noise requires a gradient: ``_GprFusedLogLik`` (forward AND backward with the
        kernel -> state space model step fused; d <= 6: one or two components, one output, chains of more than 64 points).  ``None``:
        the materialised, differentiable route runs (row signatures, observations that require a gradient, short chains)."""
        n = self._time_points.shape[-1]
        if (not self.fused_backward or rows or multi or m != 1 or d > 6 or n <= 64 or self._observations.requires_grad
                or self._time_points.requires_grad):
            return None
        dtype, dev = self._observations.dtype, self._observations.device
        batch = tuple(self._time_points.shape[:-1])
        t = self._time_points.reshape(-1, n).to(dtype).contiguous()
        y = self._observations.reshape(-1, n).contiguous()
        bsz = t.shape[0]
        nt = n - 1
        if bsz == 0 or _gpr_partition(bsz, nt, self._chunks)[1] < 2:
            return None                     # a single chunk per series leaves no summaries to start the backward from
        lam = [c._lambda.to(dtype=dtype, device=dev) for c in comps]
        var = [c._variance_t.to(dtype=dtype, device=dev) for c in comps]
        per_series = any(x.dim() > 0 for x in lam + var)
        if per_series:
            lam_t = torch.stack([x.expand(batch).reshape(-1) for x in lam], dim=-1)
            var_t = torch.stack([x.expand(batch).reshape(-1) for x in var], dim=-1)
        else:
            lam_t, var_t = torch.stack(lam), torch.stack(var)
        chol = self._chol_obs_covariance.to(dtype=dtype, device=dev)
        rinv = (1.0 / (chol * chol)).reshape(1, 1)
        log_det_rinv = torch.log(rinv[0, 0])
        orders = tuple(c.order for c in comps)
        out = _GprFusedLogLik.apply(lam_t, var_t, rinv, t, y, orders, per_series, self._kernel._jitter, self._chunks)
        const = -0.5 * math.log(2 * math.pi) * n * m + 0.5 * n * log_det_rinv
        return (out + const).reshape(batch)

    def _fused_log_likelihood_per_series(self) -> Optional[torch.Tensor]:
        """Per-series log-likelihood through ``mf_gpr_matern_loglik_*`` (kernel -> SSM generation fused into the Kalman
        sweep: 16 bytes per step instead of the materialised tensors); ``None`` when the kernel / shapes are not covered."""
        comps = self._kernel._components()
        multi = isinstance(self._kernel, IndependentMultiOutput)
        m = self._observations.shape[-1]
        d = self._kernel.state_dim
        # register kernels (d <= 6): one or two components, one output; row kernels (7 <= d <= 15): any concatenation, one
        # output (Sum) or one per component (IndependentMultiOutput, up to four; eight for d >= 10) - BASELINE config 4 is 3 x Matern-5/2, 3 outputs
        rows = 7 <= d <= 15 and len(comps) <= 15 and (m == len(comps) <= (8 if d >= 10 else 4) if multi else m == 1)
        if not self.fused or not self._observations.is_cuda or not (rows or (not multi and len(comps) <= 2 and m == 1)):
            return None
        if torch.is_grad_enabled() and (self._kernel._needs_grad() or self._chol_obs_covariance.requires_grad
                                        or self._observations.requires_grad):
            return self._fused_differentiable(comps, multi, rows, m, d)
        batch = tuple(self._time_points.shape[:-1])
        n, dtype, dev = self._time_points.shape[-1], self._observations.dtype, self._observations.device
        t = self._time_points.reshape(-1, n).to(dtype).contiguous()
        y = self._observations.reshape(-1, n, m).contiguous()
        bsz = t.shape[0]
        if bsz == 0 or n < 1:
            return None
        # the kernel's hyper-parameter tensors ([B, ncomp] or [ncomp]), the noise precision and its log-determinant: a dozen
        # small launches - at config 4's size more host time (0.25 ms) than the whole sweep takes on the device - so they are
        # kept until a source tensor is replaced or written in place (tensor identity + autograd version counter)
        sources = [c._lengthscale_t for c in comps] + [c._variance_t for c in comps] + [self._chol_obs_covariance]
        key = (dtype, dev, batch) + tuple(x._version for x in sources)
        cached = getattr(self, "_fused_cache", None)
        if (cached is not None and cached[0] == key and len(cached[1]) == len(sources)
                and all(a is b for a, b in zip(cached[1], sources))):          # (the cache holds the tensors: ids are not reused)
            per_series, lam_t, var_t, chol, rinv, log_det_rinv = cached[2:]
        else:
            lam = [c._lambda.to(dtype=dtype, device=dev) for c in comps]
            var = [c._variance_t.to(dtype=dtype, device=dev) for c in comps]
            per_series = any(x.dim() > 0 for x in lam + var)
            if per_series:
                lam_t = torch.stack([x.expand(batch).reshape(-1) for x in lam], dim=-1).contiguous()
                var_t = torch.stack([x.expand(batch).reshape(-1) for x in var], dim=-1).contiguous()
            else:
                lam_t, var_t = torch.stack(lam).contiguous(), torch.stack(var).contiguous()
            chol = self._chol_obs_covariance.to(dtype=dtype, device=dev)
            if m == 1:
                rinv = (1.0 / (chol * chol)).reshape(1, 1).contiguous()
                log_det_rinv = torch.log(rinv[0, 0])
            else:
                rinv = torch.cholesky_inverse(chol.reshape(m, m)).contiguous()
                log_det_rinv = -2.0 * torch.sum(torch.log(torch.diagonal(chol.reshape(m, m))))
            self._fused_cache = (key, sources, per_series, lam_t, var_t, chol, rinv, log_det_rinv)
        lib = _lib.load()
        ws_bytes = int(lib.mf_kf_loglik_workspace_bytes(bsz, n, d, t.element_size(), self._chunks))
        if ws_bytes == 0:
            return None
        ws = _lib.workspace(ws_bytes, dev)
        out = torch.empty(bsz, dtype=dtype, device=dev)
        info = _lib.pivot_info(dev)
        orders = (ctypes.c_int * len(comps))(*[c.order for c in comps])
        tail = (self._kernel._jitter, 0.0, _lib.ptr(out), _lib.ptr(ws), ws_bytes, info, self._chunks, self._prof_events[0],
                self._prof_events[1], _lib.stream_ptr(dev))
        head = (bsz, n, len(comps), orders, _lib.ptr(lam_t), _lib.ptr(var_t), int(per_series), _lib.ptr(t), _lib.ptr(y))
        if multi:
            rc = _lib.call_rc("mf_gpr_matern_multi_loglik", dtype, *head, m, _lib.ptr(rinv), *tail)
        else:
            rc = _lib.call_rc("mf_gpr_matern_loglik", dtype, *head, _lib.ptr(rinv), *tail)
        if rc == -101:
            return None                     # component signature not instantiated: materialise instead
        _lib.check(rc, "mf_gpr_matern_loglik")
        _lib.raise_on_info(info, "GaussianProcessRegression.log_likelihood", dev)
        const = -0.5 * math.log(2 * math.pi) * n * m + 0.5 * n * log_det_rinv
        return (out + const).reshape(batch)

    def log_likelihood(self) -> torch.Tensor:
        """``log p(y | ϑ)`` summed over the batch (gaussian_process_regression.py:150-160)."""
        per_series = self._fused_log_likelihood_per_series()
        if per_series is not None:
            return _lib.checked(torch.sum(per_series))
        return self._kalman.log_likelihood()

    def loss(self) -> torch.Tensor:
        return -self.log_likelihood()

    @property
    def posterior(self) -> AnalyticPosteriorProcess:
        """Posterior process for inference at new time points (gaussian_process_regression.py:130-148)."""
        return AnalyticPosteriorProcess(
            posterior_dist=self.posterior_state_space_model(),
            kernel=self._kernel,
            conditioning_time_points=self._time_points,
            chol_obs_covariance=self._chol_obs_covariance,
        )

    def _fused_posterior_chain(self) -> Optional[StateSpaceModel]:
        """The posterior chain with the kernel -> state space model step fused (``mf_gpr_matern_loglik_*`` on an explicit partition
        for its chunk summaries, then ``mf_gpr_matern_posterior_chain_*``): no prior tensors in memory.  ``None`` where the fused
        kernels do not apply (row signatures, several outputs, d > 6, short chains, gradients required)."""
        comps = self._kernel._components()
        d, m, n = self._kernel.state_dim, self._observations.shape[-1], self._time_points.shape[-1]
        if (not self.fused or not self.fused_backward or not self._observations.is_cuda or isinstance(self._kernel, IndependentMultiOutput)
                or len(comps) > 2 or m != 1 or d > 6 or n <= 64):
            return None
        if torch.is_grad_enabled() and (self._kernel._needs_grad() or self._chol_obs_covariance.requires_grad):
            return None
        dtype, dev = self._observations.dtype, self._observations.device
        batch = tuple(self._time_points.shape[:-1])
        t = self._time_points.reshape(-1, n).to(dtype).contiguous()
        y = self._observations.reshape(-1, n).contiguous()
        bsz, nt = t.shape[0], n - 1
        if bsz == 0 or bsz >= 2048:
            return None
        want, parts, length = _gpr_partition(bsz, nt, self._chunks, lanes=49152)                     # three wavefronts per CU
        if parts < 2:
            return None
        with torch.no_grad():
            lam = [c._lambda.to(dtype=dtype, device=dev) for c in comps]
            var = [c._variance_t.to(dtype=dtype, device=dev) for c in comps]
            per_series = any(x.dim() > 0 for x in lam + var)
            if per_series:
                lam_t = torch.stack([x.expand(batch).reshape(-1) for x in lam], dim=-1).contiguous()
                var_t = torch.stack([x.expand(batch).reshape(-1) for x in var], dim=-1).contiguous()
            else:
                lam_t, var_t = torch.stack(lam).contiguous(), torch.stack(var).contiguous()
            chol = self._chol_obs_covariance.to(dtype=dtype, device=dev)
            rinv = (1.0 / (chol * chol)).reshape(1, 1).contiguous()
            lib = _lib.load()
            esz = t.element_size()
            orders = (ctypes.c_int * len(comps))(*[c.order for c in comps])
            ws_f = _lib.workspace(int(lib.mf_kf_loglik_workspace_bytes(bsz, n, d, esz, want)), dev)
            val = torch.empty(bsz, dtype=dtype, device=dev)
            info = _lib.pivot_info(dev)
            rc = _lib.call_rc("mf_gpr_matern_loglik", dtype, bsz, n, len(comps), orders, _lib.ptr(lam_t), _lib.ptr(var_t),
                              int(per_series), _lib.ptr(t), _lib.ptr(y), _lib.ptr(rinv), self._kernel._jitter, 0.0, _lib.ptr(val),
                              _lib.ptr(ws_f), ws_f.numel(), info, want, None, None, _lib.stream_ptr(dev))
            if rc == -101:
                return None
            _lib.check(rc, "mf_gpr_matern_loglik")
            wsb = int(lib.mf_gpr_matern_posterior_chain_workspace_bytes(bsz, n, d, esz, parts))
            if wsb == 0:
                return None
            ws = _lib.workspace(wsb, dev)
            a_p = torch.empty((bsz, nt, d, d), dtype=dtype, device=dev)
            cq_p = torch.empty_like(a_p)
            b_p = torch.empty((bsz, nt, d), dtype=dtype, device=dev)
            mu0_p = torch.empty((bsz, d), dtype=dtype, device=dev)
            cp0_p = torch.empty((bsz, d, d), dtype=dtype, device=dev)
            rc = _lib.call_rc("mf_gpr_matern_posterior_chain", dtype, bsz, n, len(comps), orders, _lib.ptr(lam_t), _lib.ptr(var_t),
                              int(per_series), _lib.ptr(t), _lib.ptr(y), _lib.ptr(rinv), self._kernel._jitter, _lib.ptr(a_p),
                              _lib.ptr(mu0_p), _lib.ptr(b_p), _lib.ptr(cp0_p), _lib.ptr(cq_p), _lib.ptr(ws), wsb, info, _lib.ptr(ws_f),
                              parts, length, _lib.stream_ptr(dev))
            if rc == -101:
                return None
            _lib.check(rc, "mf_gpr_matern_posterior_chain")
            _lib.raise_on_info(info, "GaussianProcessRegression.posterior_state_space_model", dev)
        return StateSpaceModel(initial_mean=mu0_p.reshape(batch + (d,)), chol_initial_covariance=cp0_p.reshape(batch + (d, d)),
                               state_transitions=a_p.reshape(batch + (nt, d, d)), state_offsets=b_p.reshape(batch + (nt, d)),
                               chol_process_covariances=cq_p.reshape(batch + (nt, d, d)))

    def posterior_state_space_model(self) -> StateSpaceModel:
        """The smoothed chain on the training time points (what the reference's ``posterior`` is built from, :138-144)."""
        fused = self._fused_posterior_chain()
        return fused if fused is not None else self._kalman.posterior_state_space_model()
