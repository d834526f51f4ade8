"""
State space model ``x_{k+1} = A_k x_k + b_k + q_k`` on the MI355X.

Mirror of ``markovflow/state_space_model.py`` (reference): same constructor, properties and
methods.  The sequential pieces (precision assembly, mean recursion, Cholesky, Takahashi
diagonal-of-inverse) are HIP kernels behind the C ABI; the remaining glue is element-wise torch
on the device, as it is element-wise TensorFlow in the reference.
"""
import math
from typing import Tuple

import torch

from . import _autograd_ops as _ag
from . import _lib
from .block_tri_diag import LowerTriangularBlockTriDiagonal, SymmetricBlockTriDiagonal, _flat
from .gauss_markov import GaussMarkovDistribution, SampleShape, check_compatible


class StateSpaceModel(GaussMarkovDistribution):
    """State space model (state_space_model.py:35-609)."""

    def __init__(
        self,
        initial_mean: torch.Tensor,
        chol_initial_covariance: torch.Tensor,
        state_transitions: torch.Tensor,
        state_offsets: torch.Tensor,
        chol_process_covariances: torch.Tensor,
    ) -> None:
        """
        :param initial_mean: ``batch_shape + [state_dim]``.
        :param chol_initial_covariance: ``batch_shape + [state_dim, state_dim]``.
        :param state_transitions: ``batch_shape + [num_transitions, state_dim, state_dim]``.
        :param state_offsets: ``batch_shape + [num_transitions, state_dim]``.
        :param chol_process_covariances: ``batch_shape + [num_transitions, state_dim, state_dim]``.
        """
        # shape checks of state_space_model.py:101-116 (InvalidArgumentError there, ValueError here)
        if initial_mean.dim() < 1 or chol_initial_covariance.dim() < 2 or state_transitions.dim() < 3 \
                or state_offsets.dim() < 2 or chol_process_covariances.dim() < 3:
            raise ValueError("StateSpaceModel: parameter ranks are too small")
        d = initial_mean.shape[-1]
        n = state_transitions.shape[-3]
        if d < 1:
            raise ValueError("StateSpaceModel: state_dim must be at least 1")
        if n < 1:
            raise ValueError("StateSpaceModel: num_transitions must be at least 1")   # test_state_space_model.py:58-60
        batch = tuple(initial_mean.shape[:-1])
        expect = {
            "chol_initial_covariance": (chol_initial_covariance, batch + (d, d)),
            "state_transitions": (state_transitions, batch + (n, d, d)),
            "state_offsets": (state_offsets, batch + (n, d)),
            "chol_process_covariances": (chol_process_covariances, batch + (n, d, d)),
        }
        for name, (t, shape) in expect.items():
            if tuple(t.shape) != shape:
                raise ValueError(f"StateSpaceModel: {name} has shape {tuple(t.shape)}, expected {shape}")
            if t.dtype != initial_mean.dtype or t.device != initial_mean.device:
                raise ValueError(f"StateSpaceModel: {name} must share dtype and device with initial_mean")
        self._mu_0 = initial_mean.contiguous()
        # held contiguous: every kernel call needs contiguous blocks, and a chain built from slices of larger tensors (the
        # posterior: chol_dinv[..., 1:, :, :]) would otherwise be copied again by every operator that touches it
        self._A_s = state_transitions.contiguous()
        self._chol_P_0 = chol_initial_covariance.contiguous()
        self._chol_Q_s = chol_process_covariances.contiguous()
        self._b_s = state_offsets.contiguous()

    # -- shapes / accessors (state_space_model.py:126-229) ------------------------------------------------
    @property
    def event_shape(self) -> Tuple[int, int]:
        return (self.num_transitions + 1, self.state_dim)

    @property
    def batch_shape(self) -> torch.Size:
        return self._A_s.shape[:-3]

    @property
    def state_dim(self) -> int:
        return self._A_s.shape[-2]

    @property
    def num_transitions(self) -> int:
        return self._A_s.shape[-3]

    @property
    def cholesky_process_covariances(self) -> torch.Tensor:
        return self._chol_Q_s

    @property
    def cholesky_initial_covariance(self) -> torch.Tensor:
        return self._chol_P_0

    @property
    def initial_covariance(self) -> torch.Tensor:
        return self._chol_P_0 @ self._chol_P_0.transpose(-1, -2)

    @property
    def concatenated_cholesky_process_covariance(self) -> torch.Tensor:
        return torch.cat([self._chol_P_0[..., None, :, :], self._chol_Q_s], dim=-3)

    @property
    def state_offsets(self) -> torch.Tensor:
        return self._b_s

    @property
    def initial_mean(self) -> torch.Tensor:
        return self._mu_0

    @property
    def concatenated_state_offsets(self) -> torch.Tensor:
        return torch.cat([self._mu_0[..., None, :], self._b_s], dim=-2)

    @property
    def state_transitions(self) -> torch.Tensor:
        return self._A_s

    # -- C-ABI plumbing ---------------------------------------------------------------------------------
    def _flat_params(self):
        # the five tensors are held contiguous and never re-bound (the constructor is the only place that sets them), so the
        # flattened forms are VIEWS that stay valid for the life of the object - built once (five reshapes cost ~8 us, a tenth
        # of a BASELINE config 2 evaluation)
        # (a view made under no_grad, or before `requires_grad_()` on a leaf, carries no graph: the key holds both)
        src = (self._mu_0, self._chol_P_0, self._A_s, self._b_s, self._chol_Q_s)
        key = (torch.is_grad_enabled(),) + tuple(t.requires_grad for t in src)
        cached = self.__dict__.get("_flat_cache")
        if cached is None or cached[0] != key:
            cached = self._flat_cache = (key, (_flat(src[0], 1), _flat(src[1], 2), _flat(src[2], 3), _flat(src[3], 2), _flat(src[4], 3)))
        return cached[1]

    def _propagate(self, offsets: torch.Tensor) -> torch.Tensor:
        """Solve ``A⁻¹ x = offsets`` (the a_inv_block.solve of state_space_model.py:251,322)."""
        lead = tuple(offsets.shape[:-2])
        a_f = _flat(self._A_s, 3)
        offs = _flat(offsets, 2)
        out = torch.empty_like(offs)
        ws_bytes = int(_lib.load().mf_btd_solve_workspace_bytes(a_f.shape[0], offs.shape[0], self.num_transitions + 1,
                                                                 self.state_dim, offs.element_size()))
        ws = _lib.workspace(ws_bytes, offs.device)
        _lib.call("mf_ssm_marginal_means", offs.dtype, a_f.shape[0], offs.shape[0], self.num_transitions + 1,
                  self.state_dim, _lib.ptr(a_f), _lib.ptr(offs), _lib.ptr(out), _lib.ptr(ws), ws_bytes,
                  _lib.stream_ptr(offs.device))
        return out.reshape(lead + (self.num_transitions + 1, self.state_dim))

    # -- marginals ----------------------------------------------------------------------------------------
    def _needs_grad(self) -> bool:
        return torch.is_grad_enabled() and any(
            t.requires_grad for t in (self._mu_0, self._chol_P_0, self._A_s, self._b_s, self._chol_Q_s))

    def _differentiable_marginals(self) -> Tuple[torch.Tensor, torch.Tensor]:
        means, covs = _Marginals.apply(*self._flat_params())
        batch, n, d = tuple(self.batch_shape), self.num_transitions + 1, self.state_dim
        return means.reshape(batch + (n, d)), covs.reshape(batch + (n, d, d))

    @property
    def marginal_means(self) -> torch.Tensor:
        """``μ_{k+1} = A_k μ_k + b_k`` (state_space_model.py:232-251)."""
        if self._needs_grad():
            return self._differentiable_marginals()[0]
        return self._propagate(self.concatenated_state_offsets)

    @property
    def marginal_covariances(self) -> torch.Tensor:
        """Diagonal blocks of the covariance (state_space_model.py:254-262)."""
        if self._needs_grad():
            return self._differentiable_marginals()[1]
        return self._covariance_scan(want_sub=False)[0]

    @property
    def marginals(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """Marginal means and covariances (gauss_markov.py:107-117); differentiable with respect to the chain's parameters
        (``_Marginals``: the expected log-likelihood of the variational models goes through here)."""
        if self._needs_grad():
            return self._differentiable_marginals()
        return self._moments(want_sub=False)[:2]

    def covariance_blocks(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """Marginal covariances and ``Cov(x_{k+1}, x_k)`` from ONE scan; with parameters that require a gradient the marginals'
        adjoint (``_Marginals``) and a differentiable block product ``A_k P_k`` (state_space_model.py:254-262,326-341 under a tape)."""
        if self._needs_grad():
            covs = self._differentiable_marginals()[1]
            return covs, _ag.block_matmul(self._A_s.contiguous(), covs[..., :-1, :, :].contiguous())
        return self._covariance_scan(want_sub=True)

    def _moments(self, want_sub: bool):
        """``(marginal_means, marginal_covariances, Cov(x_{k+1}, x_k) or None)`` without gradients.  Where one lane per series is
        the decomposition anyway (many series or a short chain) the means ride along the covariance sweep: one kernel that
        reads ``A`` once (``mf_ssm_marginals_*``); otherwise (-101 from the ABI) the two scans in time."""
        d, n = self.state_dim, self.num_transitions + 1
        mu0, cp0, a_f, b_f, cq = self._flat_params()
        bsz = a_f.shape[0]
        lib = _lib.load()
        d_max = lib.mf_max_state_dim_f32_loglik() if a_f.dtype == torch.float32 else lib.mf_max_state_dim_f64_tile_ops()
        if bsz > 0 and n > 1 and d <= d_max:
            means = torch.empty((bsz, n, d), dtype=a_f.dtype, device=a_f.device)
            covs = torch.empty((bsz, n, d, d), dtype=a_f.dtype, device=a_f.device)
            sub = torch.empty_like(a_f) if want_sub else None
            ws_bytes = int(lib.mf_ssm_marginals_workspace_bytes(bsz, n, d, a_f.element_size()))
            ws = _lib.workspace(ws_bytes, a_f.device)
            rc = _lib.call_rc("mf_ssm_marginals", a_f.dtype, bsz, n, d, _lib.ptr(mu0), _lib.ptr(cp0), _lib.ptr(a_f),
                              _lib.ptr(b_f), _lib.ptr(cq), _lib.ptr(means), _lib.ptr(covs), _lib.ptr(sub), _lib.ptr(ws),
                              ws_bytes, _lib.stream_ptr(a_f.device))
            if rc == 0:
                batch = tuple(self.batch_shape)
                return (means.reshape(batch + (n, d)), covs.reshape(batch + (n, d, d)),
                        sub.reshape(self._A_s.shape) if want_sub else None)
            if rc != -101:
                _lib.check(rc, "mf_ssm_marginals")
        means = self._propagate(self.concatenated_state_offsets)
        covs, sub = self._covariance_scan(want_sub=want_sub)
        return means, covs, sub

    def _covariance_scan(self, want_sub: bool):
        """``Σ_0 = P_0, Σ_{k+1} = A_kΣ_kA_kᵀ + Q_k`` (and ``A_kΣ_k``).  The reference takes the block diagonal of the inverse of
        the assembled precision (``self.precision.cholesky.block_diagonal_of_inverse()``, state_space_model.py:262); the
        forward recursion gives the same blocks without assembling or factorising anything - one kernel sweep instead of
        three (parallel in time for few series; for d > 9 on the LDS-tile / MFMA engine, partitioned in time).  State
        dimensions beyond that engine keep the reference's route."""
        d, n = self.state_dim, self.num_transitions + 1
        lib = _lib.load()
        d_max = lib.mf_max_state_dim_f32_loglik() if self._A_s.dtype == torch.float32 else lib.mf_max_state_dim_f64_tile_ops()
        if d > d_max or n < 2:
            covs = self.precision.cholesky.block_diagonal_of_inverse()
            return covs, (self.subsequent_covariances(covs) if want_sub else None)
        cp0, a_f, cq = _flat(self._chol_P_0, 2), _flat(self._A_s, 3), _flat(self._chol_Q_s, 3)
        bsz = a_f.shape[0]
        covs = torch.empty((bsz, n, d, d), dtype=a_f.dtype, device=a_f.device)
        sub = torch.empty_like(a_f) if want_sub else None
        if bsz > 0:
            ws_bytes = int(_lib.load().mf_btd_diag_of_inverse_workspace_bytes(bsz, n, d, a_f.element_size()))
            ws = _lib.workspace(ws_bytes, a_f.device)
            _lib.call("mf_ssm_marginal_covariances", a_f.dtype, bsz, n, d, _lib.ptr(cp0), _lib.ptr(a_f), _lib.ptr(cq),
                      _lib.ptr(covs), _lib.ptr(sub), _lib.ptr(ws), ws_bytes, _lib.stream_ptr(a_f.device))
        batch = tuple(self.batch_shape)
        return covs.reshape(batch + (n, d, d)), (sub.reshape(self._A_s.shape) if want_sub else None)

    @property
    def a_inv_block(self) -> LowerTriangularBlockTriDiagonal:
        """``A⁻¹`` as unit lower block-bidiagonal (state_space_model.py:277-296)."""
        identities = torch.eye(self.state_dim, dtype=self._A_s.dtype, device=self._A_s.device).expand(
            tuple(self.batch_shape) + (self.num_transitions + 1, self.state_dim, self.state_dim)).contiguous()
        return LowerTriangularBlockTriDiagonal(identities, -self._A_s)

    def sample(self, sample_shape: SampleShape) -> torch.Tensor:
        """Sample trajectories, ``sample_shape + batch_shape + event_shape`` (state_space_model.py:298-324)."""
        if isinstance(sample_shape, int):
            sample_shape = (sample_shape,)
        full = tuple(sample_shape) + tuple(self.batch_shape) + self.event_shape
        eps = torch.randn(full + (1,), dtype=self._A_s.dtype, device=self._A_s.device)
        z = torch.matmul(self.concatenated_cholesky_process_covariance, eps)[..., 0]
        cond = self.concatenated_state_offsets + z
        if cond.numel() == 0:
            return cond
        if self._needs_grad():
            # reparameterised: x = A^-1 (offsets + chol eps) as the reference writes it (state_space_model.py:307-322), through
            # the differentiable bidiagonal solve - gradients reach every parameter of the chain
            return self.a_inv_block.solve(cond)
        return self._propagate(cond)

    def subsequent_covariances(self, marginal_covariances: torch.Tensor) -> torch.Tensor:
        """``Cov(x_{k+1}, x_k) = A_k P_k`` (state_space_model.py:326-341)."""
        n, d = self.num_transitions, self.state_dim
        if tuple(marginal_covariances.shape) != tuple(self.batch_shape) + (n + 1, d, d):
            raise ValueError(f"marginal_covariances has shape {tuple(marginal_covariances.shape)}")
        if _ag.needs_grad(self._A_s, marginal_covariances):
            return _ag.block_matmul(self._A_s.contiguous(), marginal_covariances[..., :-1, :, :].contiguous())
        a_f, cov_f = _flat(self._A_s, 3), _flat(marginal_covariances, 3)
        out = torch.empty_like(a_f)
        # blocks 0..n-1 of the [n+1]-long covariance chain are read in place (series stride n+1): no slice copy
        _lib.call("mf_block_matmul", a_f.dtype, a_f.shape[0], n, d, _lib.ptr(a_f), n, _lib.ptr(cov_f), n + 1,
                  _lib.ptr(out), _lib.stream_ptr(a_f.device))
        return out.reshape(self._A_s.shape)

    def log_det_precision(self) -> torch.Tensor:
        """``-2 (log|chol P0| + Σ log|chol Q_k|)`` (state_space_model.py:343-373)."""
        d0 = torch.diagonal(self._chol_P_0, dim1=-2, dim2=-1)
        dq = torch.diagonal(self._chol_Q_s, dim1=-2, dim2=-1)
        return -(torch.sum(torch.log(torch.square(d0)), dim=-1) + torch.sum(torch.log(torch.square(dq)), dim=(-1, -2)))

    def create_non_trainable_copy(self) -> "StateSpaceModel":
        """Detached copy (state_space_model.py:375-394)."""
        return StateSpaceModel(self._mu_0.detach(), self._chol_P_0.detach(), self._A_s.detach(), self._b_s.detach(),
                               self._chol_Q_s.detach())

    def create_trainable_copy(self) -> "StateSpaceModel":
        """Copy with leaf parameters that require grad (state_space_model.py:396-429).

        The reference wraps the Choleskys in a FillTriangular bijector, evaluated on every access.  Here the chain holds the
        leaves themselves - an optimiser's in-place update is what the next evaluation reads - and the Cholesky leaves are
        lower-triangular matrices that stay so: a gradient hook keeps the lower triangle of whatever gradient reaches them.
        """
        def leaf(t):
            return t.detach().clone().requires_grad_(True)

        leaves = (leaf(self._mu_0), leaf(torch.tril(self._chol_P_0)), leaf(self._A_s), leaf(self._b_s),
                  leaf(torch.tril(self._chol_Q_s)))
        leaves[1].register_hook(torch.tril)
        leaves[4].register_hook(torch.tril)
        ssm = StateSpaceModel(*leaves)
        ssm._trainable = leaves
        check_compatible(ssm, self)
        return ssm

    @property
    def trainable_variables(self) -> Tuple[torch.Tensor, ...]:
        """The leaf tensors of a trainable copy, in constructor order (``tf.Module.trainable_variables`` in the reference);
        empty for a chain that was not made by ``create_trainable_copy``."""
        return getattr(self, "_trainable", ())

    def _build_precision(self) -> SymmetricBlockTriDiagonal:
        """``K⁻¹ = A⁻ᵀ Q⁻¹ A⁻¹`` in block form (state_space_model.py:431-483).  With parameters that require a gradient the
        blocks are formed by batched, differentiable d x d products (all blocks at once: the assembly is local in time) so
        that the reference's ``dist_p.precision -> naturals_to_ssm_params`` chain (models/variational_cvi.py:105-136) can be
        differentiated; otherwise by the fused kernel."""
        if torch.is_grad_enabled() and any(t.requires_grad for t in (self._chol_P_0, self._A_s, self._chol_Q_s)):
            chols = self.concatenated_cholesky_process_covariance                       # [cholP0, cholQ_1 ...]
            eye = torch.eye(self.state_dim, dtype=chols.dtype, device=chols.device).expand(chols.shape)
            q_inv = _ag.chol_solve_blocks(chols, eye)                                   # (HIP per-block solves and products, d <= 9)
            q_inv = 0.5 * (q_inv + q_inv.transpose(-1, -2))
            if self.num_transitions == 0:
                return SymmetricBlockTriDiagonal(q_inv)
            j = _ag.block_matmul(q_inv[..., 1:, :, :].contiguous(), self._A_s)           # Q_{k+1}^-1 A_{k+1}
            ata = _ag.block_matmul(self._A_s.transpose(-1, -2).contiguous(), j)
            diag = q_inv + torch.cat([ata, torch.zeros_like(ata[..., :1, :, :])], dim=-3)
            return SymmetricBlockTriDiagonal(diag, -j)
        diag, sub, _ = self._precision_and_eta(None, None, None, False, want_eta=False)
        return SymmetricBlockTriDiagonal(diag, sub)

    def _precision_and_eta(self, h, y, r_inv, per_step: bool, want_eta: bool):
        mu0, cp0, a_s, b_s, cq = self._flat_params()
        bsz, n, d = a_s.shape[0], self.num_transitions + 1, self.state_dim
        diag = torch.empty((bsz, n, d, d), dtype=a_s.dtype, device=a_s.device)
        sub = torch.empty((bsz, n - 1, d, d), dtype=a_s.dtype, device=a_s.device)
        eta = torch.empty((bsz, n, d), dtype=a_s.dtype, device=a_s.device) if want_eta else None
        m = 1 if h is None else h.shape[-2]
        _lib.call("mf_ssm_precision", a_s.dtype, bsz, n, d, m, _lib.ptr(mu0), _lib.ptr(cp0), _lib.ptr(a_s),
                  _lib.ptr(b_s), _lib.ptr(cq), _lib.ptr(h), _lib.ptr(y), _lib.ptr(r_inv), int(per_step),
                  _lib.ptr(diag), _lib.ptr(sub), _lib.ptr(eta), _lib.stream_ptr(a_s.device))
        batch = tuple(self.batch_shape)
        diag, sub = diag.reshape(batch + (n, d, d)), sub.reshape(batch + (n - 1, d, d))
        if eta is not None:
            eta = eta.reshape(batch + (n, d))
        return diag, sub, eta

    def _log_pdf_factors(self, states: torch.Tensor) -> torch.Tensor:
        """``[log p(x₀), log p(x₁|x₀), ...]`` (state_space_model.py:485-513)."""
        if tuple(states.shape[-2:]) != self.event_shape:
            raise ValueError(f"states has shape {tuple(states.shape)}, event shape is {self.event_shape}")
        d = self.state_dim

        def mvn_tril(loc, tril, x):
            diff = (x - loc)[..., None]
            tril_b = tril.expand(diff.shape[:-2] + tril.shape[-2:])
            z = torch.linalg.solve_triangular(tril_b, diff, upper=False)[..., 0]
            logdet = torch.sum(torch.log(torch.abs(torch.diagonal(tril, dim1=-2, dim2=-1))), dim=-1)
            return -0.5 * torch.sum(z * z, dim=-1) - logdet - 0.5 * d * math.log(2 * math.pi)

        init = mvn_tril(self._mu_0, self._chol_P_0, states[..., 0, :])
        cond = torch.matmul(self._A_s, states[..., :-1, :, None])[..., 0] + self._b_s
        rest = mvn_tril(cond, self._chol_Q_s, states[..., 1:, :])
        return torch.cat([init[..., None], rest], dim=-1)

    def log_pdf(self, states) -> torch.Tensor:
        """``log p(x)`` (state_space_model.py:515-526)."""
        return torch.sum(self._log_pdf_factors(states), dim=-1)

    def kl_divergence(self, dist: GaussMarkovDistribution) -> torch.Tensor:
        """``KL(self ∥ dist)`` with shape ``batch_shape`` (state_space_model.py:528-593).  Differentiable with respect to the
        parameters of both chains (``_KLDivergence``)."""
        check_compatible(self, dist)
        if torch.is_grad_enabled() and isinstance(dist, StateSpaceModel):
            tensors = self._flat_params() + dist._flat_params()
            if any(t.requires_grad for t in tensors):
                return _KLDivergence.apply(*tensors).reshape(tuple(self.batch_shape))
        return _lib.checked(self._kl_divergence_value(dist))

    def _kl_divergence_value(self, dist: GaussMarkovDistribution, keep_moments: bool = False):
        """The divergence from its local form (``mf_ssm_kl_divergence_*``): one sweep per series when the batch fills the chip,
        else the marginals of ``self`` by the scans in time + one lane per (series, step).  ``keep_moments`` (the forward of
        ``_KLDivergence``): also return what the backward can reuse - ``(means, covs, cross)`` of ``self`` and the inputs
        ``(N, n)`` of the adjoint recursion, by-products of either route.
        Other distributions and state dimensions beyond the register kernels take the reference's operator route."""
        bsz = int(math.prod(self.batch_shape))
        n, d = self.num_transitions + 1, self.state_dim
        if (isinstance(dist, StateSpaceModel) and bsz > 0 and 16 <= d <= 32 and n > 1 and self._A_s.is_cuda
                and dist._A_s.dtype == self._A_s.dtype and tuple(dist.batch_shape) == tuple(self.batch_shape)):
            # 16 <= d <= 32: q1's moment recursion, q2's means and the block terms in ONE walk per (series, chunk) on register
            # tiles (wave_kl_walk_kernel): neither the moments nor q2's precision exist in memory.  (No by-products for the
            # backward: above d = 9 it re-evaluates, `_dense_kl`.)
            dtype, dev = self._A_s.dtype, self._A_s.device
            out = torch.empty(bsz, dtype=dtype, device=dev)
            ws_bytes = int(_lib.load().mf_ssm_kl_workspace_bytes(bsz, n, d, out.element_size()))
            ws = _lib.workspace(ws_bytes, dev)
            info = _lib.pivot_info(dev)
            rc = _lib.call_rc("mf_ssm_kl_divergence", dtype, bsz, n, d, *[_lib.ptr(t) for t in self._flat_params()],
                              *[_lib.ptr(t) for t in dist._flat_params()], _lib.ptr(out), None, None, None, None, None, _lib.ptr(ws),
                              ws_bytes, info, _lib.stream_ptr(dev))
            if rc != -100:
                _lib.check(rc, "mf_ssm_kl_divergence")
                out = out.reshape(tuple(self.batch_shape))
                return (out, None, None) if keep_moments else out
        if isinstance(dist, StateSpaceModel) and bsz > 0 and _lib.small_state_dim(d, bsz, n, self._A_s.element_size()):
            dtype, dev = self._A_s.dtype, self._A_s.device
            out = torch.empty(bsz, dtype=dtype, device=dev)
            ws_bytes = int(_lib.load().mf_ssm_kl_workspace_bytes(bsz, n, d, out.element_size()))
            ws = _lib.workspace(ws_bytes, dev)
            moments = adjoint_inputs = None
            if keep_moments and n > 1:
                moments = (torch.empty((bsz, n, d), dtype=dtype, device=dev), torch.empty((bsz, n, d, d), dtype=dtype, device=dev),
                           torch.empty((bsz, n - 1, d, d), dtype=dtype, device=dev))
            if keep_moments:
                adjoint_inputs = (torch.empty((bsz, n, d, d), dtype=dtype, device=dev), torch.empty((bsz, n, d), dtype=dtype, device=dev))
            info = _lib.pivot_info(dev)
            _lib.call("mf_ssm_kl_divergence", dtype, bsz, n, d, *[_lib.ptr(t) for t in self._flat_params()],
                      *[_lib.ptr(t) for t in dist._flat_params()], _lib.ptr(out),
                      *([_lib.ptr(t) for t in moments] if moments else [None, None, None]),
                      *([_lib.ptr(t) for t in adjoint_inputs] if adjoint_inputs else [None, None]), _lib.ptr(ws), ws_bytes, info,
                      _lib.stream_ptr(dev))
            _lib.raise_on_info(info, "StateSpaceModel.kl_divergence", dev)
            out = out.reshape(tuple(self.batch_shape))
            return (out, moments, adjoint_inputs) if keep_moments else out
        out = self._kl_divergence_operators(dist)
        return (out, None, None) if keep_moments else out

    def _kl_divergence_operators(self, dist: GaussMarkovDistribution) -> torch.Tensor:
        """The reference's route (state_space_model.py:569-593) over the operator kernels."""
        means_1, marginal_covs_1, subsequent_covs_1 = self._moments(want_sub=True)
        d, n = self.state_dim, self.num_transitions + 1
        if (isinstance(dist, StateSpaceModel) and self._A_s.is_cuda and 16 <= d <= 32 and n > 1
                and dist._A_s.dtype == self._A_s.dtype and tuple(dist.batch_shape) == tuple(self.batch_shape)):
            # 16 <= d <= 32: q2's precision is formed block row by block row on register tiles and reduced on the spot against q1's
            # moments (mf_ssm_kl_from_moments_*): it never exists in memory and no element-wise / reduction launch of torch runs
            mean_diff = (dist.marginal_means - means_1).reshape(-1, n, d).contiguous()
            bsz = mean_diff.shape[0]
            cp0_1, cq_1 = _flat(self._chol_P_0, 2), _flat(self._chol_Q_s, 3)
            cp0_2, a_2, cq_2 = _flat(dist._chol_P_0, 2), _flat(dist._A_s, 3), _flat(dist._chol_Q_s, 3)
            out = torch.empty(bsz, dtype=a_2.dtype, device=a_2.device)
            ws_bytes = int(_lib.load().mf_ssm_kl_from_moments_workspace_bytes(bsz, n, d, a_2.element_size()))
            ws = _lib.workspace(ws_bytes, a_2.device)
            _lib.call("mf_ssm_kl_from_moments", a_2.dtype, bsz, n, d, _lib.ptr(cp0_1), _lib.ptr(cq_1), _lib.ptr(cp0_2), _lib.ptr(a_2),
                      _lib.ptr(cq_2), _lib.ptr(_flat(marginal_covs_1, 3)), _lib.ptr(_flat(subsequent_covs_1, 3)), _lib.ptr(mean_diff),
                      _lib.ptr(out), _lib.ptr(ws), ws_bytes, _lib.stream_ptr(a_2.device))
            return out.reshape(tuple(self.batch_shape))
        precision_2 = dist.precision
        # sums over the blocks first, then over time: a reduction of [.., T, d, d] straight to batch_shape runs on one
        # workgroup per series (0.42 ms each at B = 8, T = 2048, d = 64)
        def total(t):
            return torch.sum(torch.sum(t, dim=(-2, -1)), dim=-1)
        trace = total(precision_2.block_diagonal * marginal_covs_1) + 2.0 * total(precision_2.block_sub_diagonal * subsequent_covs_1)
        mean_diff = dist.marginal_means - means_1
        # the reference forms |L2^T (mu2 - mu1)|^2 with the Cholesky factor of P2 (state_space_model.py:575-583); the same
        # number is (mu2 - mu1)^T P2 (mu2 - mu1): one symmetric block-tridiagonal product, no factorisation
        mahalanobis = torch.sum(torch.sum(mean_diff * precision_2.dense_mult(mean_diff), dim=-1), dim=-1)
        dim = (self.num_transitions + 1) * self.state_dim
        return 0.5 * (trace + mahalanobis - dim - dist.log_det_precision() + self.log_det_precision())

    def normalizer(self):
        """Normaliser of the chain (state_space_model.py:595-609)."""
        dim = (self.num_transitions + 1) * self.state_dim
        cst = dim * math.log(2.0 * math.pi)
        log_det = -self.log_det_precision()
        l_mean = self.precision.cholesky.dense_mult(self.marginals[0], transpose_left=True)
        mahalanobis = torch.sum(l_mean * l_mean, dim=(-2, -1))
        return 0.5 * (cst + log_det + mahalanobis)


# ---- gradients beyond the register-resident kernels (10 <= d <= 64: BASELINE config 5's state dimension) ---------------------------
# The HIP kernels give the VALUES for every state dimension; the adjoint kernels (mf_ssm_kl_grad_*, mf_ssm_marginals_grad_*) stop at
# d = 9.  Above that the backward re-evaluates the quantity with differentiable batched products - the moment recursion as a
# parallel scan in time (log2 T rounds of [B, T, d, d] GEMMs: plain rocBLAS calls, no sequential loop over T) - and lets torch
# differentiate that.  Same numbers as the forward (checked), an order of magnitude slower than a dedicated adjoint kernel would be.
def _dense_moments(mu0, cp0, a_s, b_s, cq):
    """Differentiable ``(means [B,T,d], covs [B,T,d,d])`` of a chain: inclusive scan of the maps x -> A x + b, S -> A S A^T + Q."""
    tr = lambda t: t.transpose(-1, -2)                                    # noqa: E731
    n = a_s.shape[1] + 1
    if n == 1:
        return mu0[:, None], (cp0 @ tr(cp0))[:, None]
    phi, beta, sig = a_s, b_s, cq @ tr(cq)                                # element k: the map of transition k
    off = 1
    while off < n - 1:                                                    # Hillis-Steele: element k absorbs element k - off
        p2, b2, s2 = phi[:, off:], beta[:, off:], sig[:, off:]
        p1, b1, s1 = phi[:, :-off], beta[:, :-off], sig[:, :-off]
        phi = torch.cat([phi[:, :off], p2 @ p1], dim=1)
        beta = torch.cat([beta[:, :off], (p2 @ b1[..., None])[..., 0] + b2], dim=1)
        sig = torch.cat([sig[:, :off], p2 @ s1 @ tr(p2) + s2], dim=1)
        off *= 2
    p0 = cp0 @ tr(cp0)
    means = torch.cat([mu0[:, None], (phi @ mu0[:, None, :, None])[..., 0] + beta], dim=1)
    covs = torch.cat([p0[:, None], phi @ p0[:, None] @ tr(phi) + sig], dim=1)
    return means, covs


def _dense_kl(q1, q2):
    """Differentiable ``KL(q1 || q2)`` per series in its local form (mf_kl_grad.hpp): q1's marginals by ``_dense_moments``, then
    ``1/2 [|C2^-1 C1|^2 + tr(W S W^T) + |C2^-1 eps|^2] + log|C2| - log|C1|`` per transition (W = C2^-1 dA, eps = dA m + db) and the
    same for the initial state, minus ``T d / 2``."""
    mu1, c01, a1, b1, c1 = q1
    mu2, c02, a2, b2, c2 = q2
    tri = torch.linalg.solve_triangular
    logdiag = lambda c: torch.sum(torch.log(torch.abs(torch.diagonal(c, dim1=-2, dim2=-1))), dim=-1)   # noqa: E731
    means, covs = _dense_moments(mu1, c01, a1, b1, c1)
    n, d = a1.shape[1] + 1, mu1.shape[-1]
    u0 = tri(c02, (mu1 - mu2)[..., None], upper=False)
    out = 0.5 * (torch.sum(tri(c02, c01, upper=False) ** 2, dim=(-2, -1)) + torch.sum(u0 ** 2, dim=(-2, -1))) + logdiag(c02) - logdiag(c01)
    if n > 1:
        m_k, s_k = means[:, :-1], covs[:, :-1]
        d_a = a1 - a2
        eps = (d_a @ m_k[..., None])[..., 0] + (b1 - b2)
        w = tri(c2, d_a, upper=False)
        u = tri(c2, eps[..., None], upper=False)
        terms = 0.5 * (torch.sum(tri(c2, c1, upper=False) ** 2, dim=(-2, -1)) + torch.sum((w @ s_k) * w, dim=(-2, -1))
                       + torch.sum(u ** 2, dim=(-2, -1))) + logdiag(c2) - logdiag(c1)
        out = out + torch.sum(terms, dim=-1)
    return out - 0.5 * n * d


def _dense_route(mu0, a_s) -> bool:
    """The backward by re-evaluation: state dimensions (and, for 10 <= d <= 15, shapes) above the adjoint kernels', and the degenerate shapes they do not take
    (an empty local shard of a sharded batch, a chain without transitions)."""
    if mu0.shape[0] == 0 or a_s.shape[1] == 0:
        return True
    # (10 <= d <= 15 take the HIP adjoint sweeps where the row kernels run the operators: few series, long chains)
    return not _lib.small_state_dim(mu0.shape[-1], mu0.shape[0], a_s.shape[1] + 1, mu0.element_size())


def _dense_backward(fn, tensors, grad_outputs):
    """Gradients of ``fn(*tensors)`` (a tuple of tensors) for the incoming ``grad_outputs``, by re-evaluation under autograd."""
    with torch.enable_grad():
        leaves = [t.detach().requires_grad_(True) for t in tensors]
        outs = fn(*leaves)
        outs = outs if isinstance(outs, tuple) else (outs,)
        pairs = [(o, g) for o, g in zip(outs, grad_outputs) if g is not None]
        grads = torch.autograd.grad([o for o, _ in pairs], leaves, [g for _, g in pairs], allow_unused=True)
    return tuple(torch.zeros_like(t) if g is None else g for t, g in zip(tensors, grads))


class _KLDivergence(torch.autograd.Function):
    """
    ``KL(q1 ∥ q2)`` per series as a differentiable function of the flat parameters of both chains.  Forward: the operator
    kernels.  Backward: the divergence is a sum of terms local in time over q1's marginals, so
      * the gradient with respect to q2 is MINUS the expected complete-data score of q2 under q1's marginals - the same local
        kernel as the log-likelihood's backward, without an emission model (``mf_kf_loglik_grad_*`` with ``H = NULL``);
      * the gradient with respect to q1 is the local partial derivative plus the adjoint of q1's moment recursion: one backward
        sweep per series (``mf_ssm_kl_grad_*``, csrc/mf_kl_grad.hpp).
    The reference differentiates state_space_model.py:528-593 through TensorFlow (pinned by
    tests/integration/models/test_variational.py:123-132: the ELBO gradient vanishes at the optimum).
    """

    @staticmethod
    def forward(ctx, *tensors):
        with torch.no_grad():
            q1, q2 = StateSpaceModel(*tensors[:5]), StateSpaceModel(*tensors[5:])
            out, moments, adjoint_inputs = q1._kl_divergence_value(q2, keep_moments=True)
        ctx.has_moments, ctx.has_adjoint_inputs = moments is not None, adjoint_inputs is not None
        ctx.save_for_backward(*tensors, *(moments or ()), *(adjoint_inputs or ()))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        tensors = ctx.saved_tensors
        mu1, c01, a1, b1, c1, mu2, c02, a2, b2, c2 = tensors[:10]
        if _dense_route(mu1, a1):
            grads = _dense_backward(lambda *t: _dense_kl(t[:5], t[5:]), tensors[:10], (grad_out.reshape(mu1.shape[0]),))
            # the chain's factors are lower triangular by construction: only those entries vary
            return tuple(torch.tril(g) if i % 5 in (1, 4) else g for i, g in enumerate(grads))
        with torch.no_grad():
            extra = list(tensors[10:])
            if ctx.has_moments:           # few series: the forward's scans already produced q1's moments
                means, covs, cross = extra[:3]
                extra = extra[3:]
            else:
                means, covs, cross = StateSpaceModel(mu1, c01, a1, b1, c1)._moments(want_sub=True)
            adj_n_mat, adj_n_vec = extra if ctx.has_adjoint_inputs else (None, None)   # inputs of the adjoint recursion
            bsz, d = mu1.shape
            n = a1.shape[1] + 1
            dev, dtype = mu1.device, mu1.dtype
            w = grad_out.reshape(bsz).contiguous()
            c = lambda t: _lib.ptr(t.contiguous())  # noqa: E731
            info = _lib.pivot_info(dev)
            g1 = [torch.empty_like(t) for t in (mu1, c01, a1, b1, c1)]
            ws_bytes = int(_lib.load().mf_ssm_adjoint_workspace_bytes(bsz, n, d, mu1.element_size()))
            ws = _lib.workspace(ws_bytes, dev)
            _lib.call("mf_ssm_kl_grad", dtype, bsz, n, d, c(mu1), c(c01), c(a1), c(b1), c(c1), c(mu2), c(c02), c(a2), c(b2),
                      c(c2), c(means), c(covs), _lib.ptr(w), _lib.ptr(adj_n_mat), _lib.ptr(adj_n_vec),
                      *[_lib.ptr(g) for g in g1], _lib.ptr(ws), ws_bytes, info, _lib.stream_ptr(dev))
            g2 = [torch.empty_like(t) for t in (mu2, c02, a2, b2, c2)]
            neg_w = (-w).contiguous()
            _lib.call("mf_kf_loglik_grad", dtype, bsz, n, d, 1, c(mu2), c(c02), c(a2), c(b2), c(c2), None, None, None, 0,
                      c(means), c(covs), c(cross), *[_lib.ptr(g) for g in g2], None, None, None, _lib.ptr(neg_w), info,
                      _lib.stream_ptr(dev))
            _lib.raise_on_info(info, "kl_divergence (backward)", dev)
        return (*g1, *g2)


class _Marginals(torch.autograd.Function):
    """
    ``(marginal means [B,T,d], marginal covariances [B,T,d,d])`` as a differentiable function of the flat chain parameters.
    Forward: the mean recursion and the covariance scan (HIP).  Backward: the adjoint of both recursions in ONE backward sweep
    per series (``mf_ssm_marginals_grad_*``, csrc/mf_kl_grad.hpp).  The reference differentiates
    state_space_model.py:232-262 through TensorFlow (models/variational.py:150, models/sparse_variational.py:178-192).
    """

    @staticmethod
    def forward(ctx, mu0, cp0, a_s, b_s, cq):
        with torch.no_grad():
            ssm = StateSpaceModel(mu0, cp0, a_s, b_s, cq)
            means, covs, _ = ssm._moments(want_sub=False)
        if _dense_route(mu0, a_s):
            ctx.save_for_backward(mu0, cp0, a_s, b_s, cq)           # the dense backward re-evaluates the recursion
        else:
            ctx.save_for_backward(cp0, a_s, cq, means, covs)
        return means, covs

    @staticmethod
    def backward(ctx, g_means, g_covs):
        if len(ctx.saved_tensors) == 5 and ctx.saved_tensors[0].dim() == 2:       # (mu0, ...): state_dim > 9
            grads = _dense_backward(_dense_moments, ctx.saved_tensors, (g_means, g_covs))
            return tuple(torch.tril(g) if i in (1, 4) else g for i, g in enumerate(grads))
        cp0, a_s, cq, means, covs = ctx.saved_tensors
        bsz, nt, d = a_s.shape[0], a_s.shape[1], a_s.shape[-1]
        with torch.no_grad():
            g_mu0 = torch.empty((bsz, d), dtype=a_s.dtype, device=a_s.device)
            g_cp0, g_a, g_cq = torch.empty_like(cp0), torch.empty_like(a_s), torch.empty_like(cq)
            g_b = torch.empty((bsz, nt, d), dtype=a_s.dtype, device=a_s.device)
            c = lambda t: None if t is None else _lib.ptr(t.contiguous())  # noqa: E731
            if bsz > 0:
                ws_bytes = int(_lib.load().mf_ssm_adjoint_workspace_bytes(bsz, nt + 1, d, a_s.element_size()))
                ws = _lib.workspace(ws_bytes, a_s.device)
                _lib.call("mf_ssm_marginals_grad", a_s.dtype, bsz, nt + 1, d, c(cp0), c(a_s), c(cq), c(means), c(covs),
                          c(g_means), c(g_covs), _lib.ptr(g_mu0), _lib.ptr(g_cp0), _lib.ptr(g_a), _lib.ptr(g_b), _lib.ptr(g_cq),
                          _lib.ptr(ws), ws_bytes, _lib.stream_ptr(a_s.device))
        return g_mu0, g_cp0, g_a, g_b, g_cq


def state_space_model_from_covariances(
    initial_mean: torch.Tensor,
    initial_covariance: torch.Tensor,
    state_transitions: torch.Tensor,
    state_offsets: torch.Tensor,
    process_covariances: torch.Tensor,
) -> StateSpaceModel:
    """Build from full covariances; all-zero matrices pass through as zero (state_space_model.py:613-664)."""

    def cholesky_or_zero(covariance: torch.Tensor) -> torch.Tensor:
        if covariance.numel() < 1:
            raise ValueError("covariance must have at least one element")
        mask = torch.all(covariance == 0, dim=-1).all(dim=-1)[..., None, None]
        eye = torch.eye(covariance.shape[-1], dtype=covariance.dtype, device=covariance.device)
        fix = torch.where(mask, eye, torch.zeros_like(covariance))
        return torch.where(mask, torch.zeros_like(covariance), torch.linalg.cholesky(covariance + fix))

    return StateSpaceModel(
        initial_mean=initial_mean,
        chol_initial_covariance=cholesky_or_zero(initial_covariance),
        state_transitions=state_transitions,
        state_offsets=state_offsets,
        chol_process_covariances=cholesky_or_zero(process_covariances),
    )
