"""
SDE kernels -> state space models on the MI355X: the step immediately before the Kalman path.

Mirror of the part of ``markovflow/kernels`` (reference) that generates the BASELINE configurations:
``SDEKernel`` / ``StationaryKernel`` (``kernels/sde_kernel.py:43-497``), ``Matern12`` / ``Matern32`` / ``Matern52``
(``kernels/matern.py``), ``ConcatKernel`` / ``Sum`` / ``IndependentMultiOutput`` (``sde_kernel.py:540-690,826-878``), with
the same method names and shapes.  The transition matrices ``A_k = exp(F Δt_k)`` and the Cholesky factors of
``Q_k = P∞ − A_k P∞ A_kᵀ`` are produced by one HIP kernel (``mf_sde_matern_transitions_*``) directly in the
``[batch, T−1, d, d]`` layout ``StateSpaceModel`` takes; nothing else of the reference's kernels package (Product, Stack,
Periodic, piecewise, latent-exp kernels) is mirrored.

Hyper-parameters are plain tensors / floats (the reference wraps them in ``gpflow.Parameter``); a hyper-parameter may also
carry ``batch_shape`` (one value per series), which the reference expresses with ``StackKernel``.
"""
import abc
import ctypes
import math
import weakref
from typing import List, Sequence, Tuple, Union

import torch

from . import _lib
from .emission_model import EmissionModel
from .gauss_markov import GaussMarkovDistribution
from .state_space_model import StateSpaceModel

Hyper = Union[float, torch.Tensor]


def to_delta_time(time_points: torch.Tensor) -> torch.Tensor:
    """``Δt_k = t_{k+1} − t_k`` (markovflow/utils.py ``to_delta_time``)."""
    return time_points[..., 1:] - time_points[..., :-1]


def _block_diag(blocks: Sequence[torch.Tensor]) -> torch.Tensor:
    """Block-diagonal of matrices that share their leading dims (markovflow/utils.py ``block_diag``)."""
    lead = torch.broadcast_shapes(*[tuple(b.shape[:-2]) for b in blocks])
    d = sum(b.shape[-1] for b in blocks)
    out = torch.zeros(tuple(lead) + (d, d), dtype=blocks[0].dtype, device=blocks[0].device)
    off = 0
    for b in blocks:
        k = b.shape[-1]
        out[..., off:off + k, off:off + k] = b
        off += k
    return out


_CHOL_INDEX = {}    # (component orders, device) -> flat positions of the non-zero entries of chol(P_inf + jitter)


class SDEKernel(abc.ABC):
    """Kernel defined by a linear SDE ``dx = F x dt + L dβ`` (sde_kernel.py:43-351)."""

    def __init__(self, output_dim: int = 1, jitter: float = 0.0) -> None:
        assert output_dim > 0, "The output dimension must be positive"     # sde_kernel.py:127-129
        assert jitter >= 0.0, "jitter must be a non-negative float number."
        self._output_dim = output_dim
        self._jitter = float(jitter)

    @property
    def output_dim(self) -> int:
        return self._output_dim

    @property
    @abc.abstractmethod
    def state_dim(self) -> int:
        """Dimension of the state."""

    # -- what a concrete kernel provides -----------------------------------------------------------------------------
    @abc.abstractmethod
    def _components(self) -> List["_MaternBase"]:
        """The Matérn components whose block-diagonal concatenation is this kernel's state."""

    @abc.abstractmethod
    def initial_mean(self, batch_shape) -> torch.Tensor:
        """``batch_shape + [state_dim]``."""

    @abc.abstractmethod
    def initial_covariance(self, initial_time_point: torch.Tensor) -> torch.Tensor:
        """``batch_shape + [state_dim, state_dim]``."""

    # -- generic machinery ---------------------------------------------------------------------------------------------
    @property
    def jitter_matrix(self) -> torch.Tensor:
        """``jitter · I`` (sde_kernel.py:332-340)."""
        ref = self._components()[0]._variance_t
        return self._jitter * torch.eye(self.state_dim, dtype=ref.dtype, device=ref.device)

    def _needs_grad(self) -> bool:
        return torch.is_grad_enabled() and any(
            c._lengthscale_t.requires_grad or c._variance_t.requires_grad for c in self._components())

    def _torch_transitions(self, time_deltas: torch.Tensor, want_chol: bool, want_cov: bool):
        """The same closed forms in differentiable torch ops: used only while a hyper-parameter requires a gradient (the
        HIP generator has no backward); the chain rule then runs  hyper-parameters -> (A, chol Q) -> log-likelihood, the
        last link being the Fisher-identity kernel of ``KalmanFilter``."""
        blocks_a, blocks_q = [], []
        dtm = time_deltas[..., None, None]
        for c in self._components():
            lam = c._lambda.to(dtype=time_deltas.dtype, device=time_deltas.device)
            lam_b = lam[..., None, None, None] if lam.dim() > 0 else lam
            k = c.state_dim
            f, pinf = c.feedback_matrix.to(time_deltas), c.steady_state_covariance.to(time_deltas)
            eye = torch.eye(k, dtype=time_deltas.dtype, device=time_deltas.device)
            nil = f + (lam[..., None, None] if lam.dim() > 0 else lam) * eye          # nilpotent: (F + lam I)^k = 0
            nil_b = nil[..., None, :, :] if nil.dim() > 2 else nil
            a = eye + nil_b * dtm
            if k == 3:
                a = a + (nil_b @ nil_b) * (0.5 * dtm ** 2)
            a = a * torch.exp(-lam_b * dtm)
            p_b = pinf[..., None, :, :] if pinf.dim() > 2 else pinf
            q = p_b - a @ p_b @ a.transpose(-1, -2)
            blocks_a.append(a)
            blocks_q.append(0.5 * (q + q.transpose(-1, -2)))
        a_s = _block_diag(blocks_a)
        q_s = _block_diag(blocks_q) + self.jitter_matrix.to(time_deltas)
        chol = torch.linalg.cholesky(q_s) if want_chol else None
        return a_s, chol, (q_s if want_cov else None)

    def _device_transitions(self, time_deltas: torch.Tensor, want_chol: bool, want_cov: bool):
        """(A, chol Q, Q) for ``time_deltas`` of shape ``batch_shape + [n]`` through the HIP kernel."""
        comps = self._components()
        batch = tuple(time_deltas.shape[:-1])
        n, d = time_deltas.shape[-1], self.state_dim
        needs_grad = self._needs_grad()
        if needs_grad and (want_cov or not time_deltas.is_cuda or (torch.is_grad_enabled() and time_deltas.requires_grad)
                           or time_deltas.numel() == 0):
            # Q itself, or a gradient with respect to the time points, is asked for: differentiable torch ops
            return self._torch_transitions(time_deltas, want_chol, want_cov)
        dt = time_deltas.reshape(-1, n).contiguous()
        bsz = dt.shape[0]
        lam = [c._lambda.to(dtype=dt.dtype, device=dt.device) for c in comps]
        var = [c._variance_t.to(dtype=dt.dtype, device=dt.device) for c in comps]
        per_series = any(x.dim() > 0 for x in lam + var)
        if per_series:
            lam_t = torch.stack([x.expand(batch).reshape(-1) for x in lam], dim=-1).contiguous()
            var_t = torch.stack([x.expand(batch).reshape(-1) for x in var], dim=-1).contiguous()
        else:
            lam_t, var_t = torch.stack(lam).contiguous(), torch.stack(var).contiguous()
        if needs_grad:
            # hyper-parameters under a gradient (the GPR training step): HIP generator forward AND backward
            # (_MaternTransitions); lam_t / var_t were assembled from the leaves by differentiable torch ops above
            a_s, chol = _MaternTransitions.apply(dt, lam_t, var_t, tuple(c.order for c in comps), bool(per_series), float(self._jitter),
                                                 bool(want_chol))
            shape = batch + (n, d, d)
            return a_s.reshape(shape), (chol.reshape(shape) if want_chol else None), None
        orders = (ctypes.c_int * len(comps))(*[c.order for c in comps])
        a_s = torch.empty((bsz, n, d, d), dtype=dt.dtype, device=dt.device)
        chol = torch.empty_like(a_s) if want_chol else None
        cov = torch.empty_like(a_s) if want_cov else None
        _lib.call("mf_sde_matern_transitions", dt.dtype, bsz, n, len(comps), orders, _lib.ptr(lam_t), _lib.ptr(var_t),
                  int(per_series), _lib.ptr(dt), self._jitter, _lib.ptr(a_s), _lib.ptr(chol), _lib.ptr(cov),
                  _lib.stream_ptr(dt.device))
        shape = batch + (n, d, d)
        return (a_s.reshape(shape), None if chol is None else chol.reshape(shape),
                None if cov is None else cov.reshape(shape))

    def state_transitions(self, transition_times: torch.Tensor, time_deltas: torch.Tensor) -> torch.Tensor:
        """``A_k = exp(F Δt_k)``, ``batch_shape + [num_transitions, state_dim, state_dim]`` (sde_kernel.py:299-310)."""
        return self._device_transitions(time_deltas, False, False)[0]

    def process_covariances(self, transition_times: torch.Tensor, time_deltas: torch.Tensor) -> torch.Tensor:
        """``Q_k`` (sde_kernel.py:312-330)."""
        return self._device_transitions(time_deltas, False, True)[2]

    def transition_statistics(self, transition_times: torch.Tensor, time_deltas: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """``(A_k, Q_k)`` with ``Q_k = P∞ − A_k P∞ A_kᵀ + jitter`` (sde_kernel.py:421-446)."""
        a_s, _, q_s = self._device_transitions(time_deltas, False, True)
        return a_s, q_s

    def transition_statistics_from_time_points(self, time_points: torch.Tensor):
        """sde_kernel.py:253-265."""
        return self.transition_statistics(time_points[..., :-1], to_delta_time(time_points))

    def state_offsets(self, transition_times: torch.Tensor, time_deltas: torch.Tensor) -> torch.Tensor:
        """``b_k = (I − A_k) m`` for a state mean ``m`` (sde_kernel.py:460-475); zero for a zero mean."""
        mean = self.initial_mean(tuple(time_deltas.shape[:-1]))
        if not bool(torch.any(mean != 0)):
            return torch.zeros(tuple(time_deltas.shape) + (self.state_dim,), dtype=time_deltas.dtype, device=time_deltas.device)
        a_s = self.state_transitions(transition_times, time_deltas)
        m = mean[..., None, :].to(dtype=a_s.dtype, device=a_s.device)
        return m - torch.matmul(a_s, m[..., None])[..., 0]

    def state_space_model(self, time_points: torch.Tensor) -> StateSpaceModel:
        """The chain this kernel induces on ``time_points`` (``batch_shape + [num_data]``, strictly increasing)
        (sde_kernel.py:153-171).  The Cholesky factors come straight from the device kernel (zero covariances pass
        through as zero, as in ``state_space_model_from_covariances``)."""
        batch = tuple(time_points.shape[:-1])
        deltas = to_delta_time(time_points)
        a_s, chol_q, _ = self._device_transitions(deltas, True, False)
        chol_p0 = self._initial_cholesky(batch, a_s.dtype, a_s.device)
        if chol_p0 is None:
            p0 = self.initial_covariance(time_points[..., 0:1]).to(dtype=a_s.dtype, device=a_s.device)
            p0 = p0.expand(batch + tuple(p0.shape[-2:])).contiguous()
            # (P-infinity + jitter is positive definite by construction: no info check, hence no host synchronisation)
            chol_p0 = _lib.checked_cholesky(p0, "SDEKernel.state_space_model (initial covariance)")
        return StateSpaceModel(
            initial_mean=self.initial_mean(batch).to(dtype=a_s.dtype, device=a_s.device).expand(batch + (self.state_dim,)).contiguous(),
            chol_initial_covariance=chol_p0,
            state_transitions=a_s,
            state_offsets=self.state_offsets(time_points[..., :-1], deltas).to(dtype=a_s.dtype),
            chol_process_covariances=chol_q,
        )

    def _initial_cholesky(self, batch, dtype, device):
        """``chol(P_inf + jitter I)`` of a concatenation of Matern components in closed form, ``batch + [d, d]`` - or None (the generic
        route: assemble P_inf, batched Cholesky).  The steady-state covariance of every Matern component is a 1 x 1, a diagonal 2 x 2 or
        an arrow-shaped 3 x 3 block (matern.py:27-518) whose factor is a handful of elementwise operations on the stacked
        hyper-parameters; assembling P_inf block by block by slice assignment and differentiating a batched Cholesky through torch
        is ~150 small launches of the ~320 of a training step at config 4's model (profiles/r05_config4_torch_ops.txt)."""
        comps = self._components()
        if not comps or not all(isinstance(c, _MaternBase) and c.order in (1, 3, 5) for c in comps):
            return None
        d, jit = self.state_dim, self._jitter
        rows, cols, vals = [], [], []
        offs, o = [], 0
        for c in comps:
            offs.append(o)
            o += c.state_dim
        for order in sorted({c.order for c in comps}):
            grp = [i for i, c in enumerate(comps) if c.order == order]
            lam = torch.stack([comps[i]._lambda.to(dtype=dtype, device=device).expand(batch) for i in grp], dim=-1)
            var = torch.stack([comps[i]._variance_t.to(dtype=dtype, device=device).expand(batch) for i in grp], dim=-1)
            base = [offs[i] for i in grp]
            l00 = torch.sqrt(var + jit)

            def put(dr, dc, v):
                rows.extend(b + dr for b in base)
                cols.extend(b + dc for b in base)
                vals.append(v)

            put(0, 0, l00)
            if order == 3:
                put(1, 1, torch.sqrt(var * lam ** 2 + jit))
            elif order == 5:
                kap = lam ** 2 / 3.0
                l20 = -(kap * var) / l00
                put(2, 0, l20)
                put(1, 1, torch.sqrt(kap * var + jit))
                put(2, 2, torch.sqrt(lam ** 4 * var + jit - l20 ** 2))
        flat = torch.zeros(tuple(batch) + (d * d,), dtype=dtype, device=device)
        key = (tuple(c.order for c in comps), str(device))
        index = _CHOL_INDEX.get(key)
        if index is None:       # (one small upload per kernel signature and device, not per model build)
            index = _CHOL_INDEX[key] = torch.tensor([r * d + c for r, c in zip(rows, cols)], dtype=torch.long, device=device)
        flat = flat.index_copy(-1, index, torch.cat(vals, dim=-1))
        return flat.reshape(tuple(batch) + (d, d))

    def build_finite_distribution(self, time_points: torch.Tensor) -> GaussMarkovDistribution:
        """sde_kernel.py:140-151."""
        return self.state_space_model(time_points)

    def generate_emission_model(self, time_points: torch.Tensor) -> EmissionModel:
        """``H = [1, 0, 0, …]`` tiled over the time points (sde_kernel.py:173-211)."""
        h = torch.zeros((self.output_dim, self.state_dim), dtype=time_points.dtype, device=time_points.device)
        h[:, 0] = 1.0
        return EmissionModel(h.expand(tuple(time_points.shape) + h.shape).contiguous())

    def __add__(self, other: "SDEKernel") -> "Sum":
        assert self.output_dim == other.output_dim                         # sde_kernel.py:342-345
        return Sum([self, other])


class _MaternTransitions(torch.autograd.Function):
    """``(A [B,n,d,d], chol Q [B,n,d,d])`` of a concatenation of Matern components as a differentiable function of the stacked
    hyper-parameters ``lam [B,ncomp] | [ncomp]`` (= sqrt(order) / lengthscale) and ``var`` - forward: the HIP generator
    (``mf_sde_matern_transitions_*``); backward: the same closed forms in forward mode inside one kernel per (series, transition)
    (``mf_sde_matern_transitions_grad_*``), summed over the transitions here.  The reference differentiates matern.py /
    sde_kernel.py:421-446 and a batched Cholesky through TensorFlow; the torch restatement of that (``_torch_transitions``) cost
    350 of the 373 ms of a GPR training step at B=1024, T=10000, d=6."""

    @staticmethod
    def forward(ctx, dt, lam_t, var_t, orders, per_series, jitter, want_chol):
        bsz, n = dt.shape
        d = sum((o + 1) // 2 for o in orders)
        c_orders = (ctypes.c_int * len(orders))(*orders)
        a_s = torch.empty((bsz, n, d, d), dtype=dt.dtype, device=dt.device)
        chol = torch.empty_like(a_s) if want_chol else None
        lam_c, var_c = lam_t.detach().contiguous(), var_t.detach().contiguous()
        _lib.call("mf_sde_matern_transitions", dt.dtype, bsz, n, len(orders), c_orders, _lib.ptr(lam_c), _lib.ptr(var_c),
                  int(per_series), _lib.ptr(dt.detach()), jitter, _lib.ptr(a_s), _lib.ptr(chol), None, _lib.stream_ptr(dt.device))
        ctx.save_for_backward(dt.detach(), lam_c, var_c)
        ctx.meta = (orders, per_series, jitter)
        if not want_chol:
            chol = a_s.new_zeros(())
            ctx.mark_non_differentiable(chol)
        return a_s, chol

    @staticmethod
    def backward(ctx, g_a, g_chol):
        dt, lam_c, var_c = ctx.saved_tensors
        orders, per_series, jitter = ctx.meta
        bsz, n = dt.shape
        c_orders = (ctypes.c_int * len(orders))(*orders)
        part = torch.empty((bsz, n, len(orders), 2), dtype=dt.dtype, device=dt.device)
        with torch.no_grad():
            ga = None if g_a is None else g_a.contiguous()
            gc = None if (g_chol is None or g_chol.dim() == 0) else g_chol.contiguous()
            _lib.call("mf_sde_matern_transitions_grad", dt.dtype, bsz, n, len(orders), c_orders, _lib.ptr(lam_c), _lib.ptr(var_c),
                      int(per_series), _lib.ptr(dt), jitter, _lib.ptr(ga), _lib.ptr(gc), _lib.ptr(part), _lib.stream_ptr(dt.device))
            g = torch.sum(part, dim=1)                                   # [B, ncomp, 2]
            if not per_series:
                g = torch.sum(g, dim=0)                                  # shared hyper-parameters: [ncomp, 2]
        return None, g[..., 0].contiguous(), g[..., 1].contiguous(), None, None, None, None


class StationaryKernel(SDEKernel, abc.ABC):
    """Stationary SDE kernel: constant feedback matrix ``F`` and steady state covariance ``P∞`` (sde_kernel.py:353-497)."""

    @property
    @abc.abstractmethod
    def feedback_matrix(self) -> torch.Tensor:
        """``F``, ``[state_dim, state_dim]`` (leading batch dims if a hyper-parameter has them)."""

    @property
    @abc.abstractmethod
    def steady_state_covariance(self) -> torch.Tensor:
        """``P∞``."""

    def initial_mean(self, batch_shape) -> torch.Tensor:
        ref = self._components()[0]._variance_t
        return torch.zeros(tuple(batch_shape) + (self.state_dim,), dtype=ref.dtype, device=ref.device)

    def state_offsets(self, transition_times: torch.Tensor, time_deltas: torch.Tensor) -> torch.Tensor:
        """Zero: a stationary kernel's state has zero mean (sde_kernel.py:460-475 evaluates ``(I - A_k) 0``) - known without
        looking at the device, where the generic form has to test the mean (a host synchronisation per model build)."""
        return torch.zeros(tuple(time_deltas.shape) + (self.state_dim,), dtype=time_deltas.dtype, device=time_deltas.device)

    def initial_covariance(self, initial_time_point: torch.Tensor) -> torch.Tensor:
        """``P∞ + jitter`` (sde_kernel.py:402-419)."""
        assert initial_time_point.shape[-1] == 1
        p = self.steady_state_covariance
        batch = tuple(initial_time_point.shape[:-1])
        return p.expand(torch.broadcast_shapes(batch + p.shape[-2:], p.shape)) + self.jitter_matrix


# (tensor, version) pairs whose positivity has been read back from the device: the check of matern.py:52-56 is a host
# synchronisation, and a training loop builds the kernel objects anew around the SAME leaf tensors every step (six synchronisations
# per step at config 4's model).  The entry holds a WEAK reference (a cache must not keep hyper-parameter tensors of a million series
# alive): a dead reference, or a live one to another object under a recycled id, is a miss; an in-place update (an optimiser step)
# changes the version and is checked again.
_POSITIVE = {}


def _version_key(t: torch.Tensor):
    """``(data pointer, version counter)``: what the caches below compare.  The pointer catches ``set_()`` / ``.data = `` re-seating
    (invisible to the version counter); inference tensors track no version at all - ``None``: never cached."""
    if t.is_inference():
        return None
    return (t.data_ptr(), t._version)


def _known_positive(*tensors: torch.Tensor) -> bool:
    fresh = []
    for t in tensors:
        key = _version_key(t)
        hit = _POSITIVE.get(id(t)) if key is not None else None
        if hit is None or hit[0]() is not t or hit[1] != key:
            fresh.append(t)
    if not fresh:
        return True
    with torch.no_grad():
        bad = [(t.detach() <= 0).any() for t in fresh]
        if len({t.device for t in fresh}) == 1:
            ok = not bool(torch.stack(bad).any())           # one read-back for all of them
        else:
            ok = not any(bool(b) for b in bad)
    if ok:
        if len(_POSITIVE) > 256:
            _POSITIVE.clear()
        for t in fresh:
            key = _version_key(t)
            if key is not None:
                _POSITIVE[id(t)] = (weakref.ref(t), key)
    return ok


class _MaternBase(StationaryKernel):
    order = 0   # Matérn-order/2

    def __init__(self, lengthscale: Hyper, variance: Hyper, output_dim: int = 1, jitter: float = 0.0, device=None,
                 dtype=torch.float64) -> None:
        super().__init__(output_dim, jitter)
        dev = device if device is not None else (lengthscale.device if isinstance(lengthscale, torch.Tensor) else "cpu")
        self._lengthscale_t = torch.as_tensor(lengthscale, dtype=dtype, device=dev)
        self._variance_t = torch.as_tensor(variance, dtype=dtype, device=dev)
        if not _known_positive(self._lengthscale_t, self._variance_t):
            raise ValueError("lengthscale and variance must be positive.")   # matern.py:52-56
        self._lambda_cached = None

    @property
    def state_dim(self) -> int:
        return (self.order + 1) // 2

    @property
    def lengthscale(self) -> torch.Tensor:
        return self._lengthscale_t

    @property
    def variance(self) -> torch.Tensor:
        return self._variance_t

    @property
    def _lambda(self) -> torch.Tensor:
        if torch.is_grad_enabled() and self._lengthscale_t.requires_grad:
            return math.sqrt(self.order) / self._lengthscale_t       # a node of its own per use: graphs built from it stay independent
        key = _version_key(self._lengthscale_t)                    # otherwise one division per kernel object, not one per use
        if key is None:                                            # (an inference tensor: no version to key a cache on)
            return math.sqrt(self.order) / self._lengthscale_t
        if self._lambda_cached is None or self._lambda_cached[0] != key:
            self._lambda_cached = (key, math.sqrt(self.order) / self._lengthscale_t.detach())
        return self._lambda_cached[1]

    def invalidate_cache(self) -> None:
        """Forget what was derived from the hyper-parameters (``√order/ℓ``, the positivity check).  The caches are keyed on the
        tensors' data pointer and version counter; a write THROUGH ``.data`` (``x.data.mul_()``) changes neither - call this
        after one (as ``KalmanFilter.invalidate_filter_cache`` / ``GaussianProcessRegression.invalidate_hyperparameter_cache``)."""
        self._lambda_cached = None
        for t in (self._lengthscale_t, self._variance_t):
            _POSITIVE.pop(id(t), None)

    def _components(self):
        return [self]


class Matern12(_MaternBase):
    """Matérn-1/2 (exponential / Ornstein–Uhlenbeck) kernel (matern.py:27-127): ``F = −1/ℓ``, ``P∞ = σ²``."""
    order = 1

    @property
    def feedback_matrix(self) -> torch.Tensor:
        return (-1.0 / self._lengthscale_t)[..., None, None]

    @property
    def steady_state_covariance(self) -> torch.Tensor:
        return self._variance_t[..., None, None].clone()


class Matern32(_MaternBase):
    """Matérn-3/2 kernel (matern.py:237-373): ``F = [[0, 1], [−λ², −2λ]]``, ``P∞ = σ² diag(1, λ²)``, ``λ = √3/ℓ``."""
    order = 3

    @property
    def feedback_matrix(self) -> torch.Tensor:
        lam = self._lambda
        f = torch.zeros(tuple(lam.shape) + (2, 2), dtype=lam.dtype, device=lam.device)
        f[..., 0, 1] = 1.0
        f[..., 1, 0] = -lam ** 2
        f[..., 1, 1] = -2 * lam
        return f

    @property
    def steady_state_covariance(self) -> torch.Tensor:
        lam, var = self._lambda, self._variance_t
        p = torch.zeros(tuple(torch.broadcast_shapes(lam.shape, var.shape)) + (2, 2), dtype=lam.dtype, device=lam.device)
        p[..., 0, 0] = var
        p[..., 1, 1] = var * lam ** 2
        return p


class Matern52(_MaternBase):
    """Matérn-5/2 kernel (matern.py:376-518): ``F = [[0,1,0],[0,0,1],[−λ³,−3λ²,−3λ]]``, ``λ = √5/ℓ``."""
    order = 5

    @property
    def feedback_matrix(self) -> torch.Tensor:
        lam = self._lambda
        f = torch.zeros(tuple(lam.shape) + (3, 3), dtype=lam.dtype, device=lam.device)
        f[..., 0, 1] = 1.0
        f[..., 1, 2] = 1.0
        f[..., 2, 0] = -lam ** 3
        f[..., 2, 1] = -3 * lam ** 2
        f[..., 2, 2] = -3 * lam
        return f

    @property
    def steady_state_covariance(self) -> torch.Tensor:
        lam, var = self._lambda, self._variance_t
        l23 = lam ** 2 / 3.0
        p = torch.zeros(tuple(torch.broadcast_shapes(lam.shape, var.shape)) + (3, 3), dtype=lam.dtype, device=lam.device)
        p[..., 0, 0] = var
        p[..., 0, 2] = -var * l23
        p[..., 2, 0] = -var * l23
        p[..., 1, 1] = var * l23
        p[..., 2, 2] = var * lam ** 4
        return p


class ConcatKernel(StationaryKernel, abc.ABC):
    """State = concatenation of the child states; ``A``, ``F``, ``P∞`` block diagonal (sde_kernel.py:540-658)."""

    def __init__(self, kernels: List[SDEKernel], jitter: float = 0.0):
        assert kernels, "There must be at least one child kernel."           # sde_kernel.py:566-571
        assert all(k.output_dim == kernels[0].output_dim for k in kernels), "All kernels must have the same output dimension"
        self._kernels = list(kernels)
        super().__init__(self._kernels[0].output_dim, jitter)

    @property
    def kernels(self) -> List[SDEKernel]:
        return self._kernels

    @property
    def state_dim(self) -> int:
        return sum(k.state_dim for k in self._kernels)

    def _components(self):
        return [c for k in self._kernels for c in k._components()]

    def initial_mean(self, batch_shape) -> torch.Tensor:
        return torch.cat([k.initial_mean(batch_shape) for k in self._kernels], dim=-1)

    @property
    def feedback_matrix(self) -> torch.Tensor:
        return _block_diag([k.feedback_matrix for k in self._kernels])

    @property
    def steady_state_covariance(self) -> torch.Tensor:
        return _block_diag([k.steady_state_covariance for k in self._kernels])


class Sum(ConcatKernel):
    """Sum of kernels: ``H = [H₁, H₂, …]`` (sde_kernel.py:660-688)."""

    def generate_emission_model(self, time_points: torch.Tensor) -> EmissionModel:
        return EmissionModel(torch.cat([k.generate_emission_model(time_points).emission_matrix for k in self._kernels], dim=-1))


class IndependentMultiOutput(ConcatKernel):
    """Independent outputs, one child kernel each: ``H = H₁ ⊕ H₂ ⊕ …`` (sde_kernel.py:826-878)."""

    def __init__(self, kernels: List[SDEKernel], jitter: float = 0.0):
        super().__init__(kernels, jitter)
        self._output_dim = sum(k.output_dim for k in kernels)                # sde_kernel.py:843-845

    def generate_emission_model(self, time_points: torch.Tensor) -> EmissionModel:
        mats = [k.generate_emission_model(time_points).emission_matrix for k in self._kernels]
        lead = tuple(time_points.shape)
        out = torch.zeros(lead + (self.output_dim, self.state_dim), dtype=mats[0].dtype, device=mats[0].device)
        r = c = 0
        for m in mats:
            out[..., r:r + m.shape[-2], c:c + m.shape[-1]] = m
            r += m.shape[-2]
            c += m.shape[-1]
        return EmissionModel(out)
