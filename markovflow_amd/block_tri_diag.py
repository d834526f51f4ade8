"""
Block-tridiagonal operators on the MI355X.

Mirror of ``markovflow/block_tri_diag.py`` (reference): same class names, constructor arguments,
method names and meaning, but tensors are ``torch`` HIP tensors and every method dispatches to a
hand-written HIP kernel through the C ABI (``include/markovflow_amd.h``).  The reference's band
layout and its block<->band conversions (block_tri_diag.py:206-237,549-592) do not exist here:
the native layout IS ``[..., outer_dim, inner_dim, inner_dim]``.
"""
import abc
from typing import Optional, Tuple

import torch

from . import _autograd_ops as _ag
from . import _lib


def _flat(t: torch.Tensor, tail: int) -> torch.Tensor:
    """Collapse leading batch dims: [..., <tail dims>] -> [B, <tail dims>] contiguous."""
    return t.reshape((-1,) + tuple(t.shape[-tail:])).contiguous()


class BlockTriDiagonal(abc.ABC):
    """Abstract block tridiagonal matrix (reference: block_tri_diag.py:37-288)."""

    def __init__(self, diagonal: torch.Tensor, symmetric: bool, sub_diagonal: Optional[torch.Tensor] = None) -> None:
        """
        :param diagonal: ``[... outer_dim, inner_dim, inner_dim]``.
        :param symmetric: whether the matrix is symmetric.
        :param sub_diagonal: ``[... outer_dim - 1, inner_dim, inner_dim]`` or None.
        """
        if diagonal.dim() < 3:
            raise ValueError(f"diagonal must be at least 3D but has shape {tuple(diagonal.shape)}")
        if diagonal.shape[-1] != diagonal.shape[-2]:
            raise ValueError("Last two dimensions of the block diagonal must match.")  # block_tri_diag.py:55-59
        self._diag = diagonal
        if sub_diagonal is not None:
            if self.outer_dim <= 1:
                raise ValueError("There is no sub-diagonal with outer dimension of one.")  # :68-74
            shape = tuple(self.batch_shape) + (self.outer_dim - 1, self.inner_dim, self.inner_dim)
            if tuple(sub_diagonal.shape) != shape:
                raise ValueError(f"Sub_diagonal has shape {tuple(sub_diagonal.shape)} but must have shape: {shape}")
            if sub_diagonal.dtype != diagonal.dtype or sub_diagonal.device != diagonal.device:
                raise ValueError("diagonal and sub_diagonal must share dtype and device")
        self._sub_diag = sub_diagonal
        self._symmetric = symmetric

    # -- shape properties (block_tri_diag.py:100-148) ------------------------------------------------
    @property
    def bandwidth(self) -> int:
        bandwidth = self.inner_dim - 1
        if self._sub_diag is not None:
            bandwidth += self.inner_dim
        return bandwidth

    @property
    def batch_shape(self) -> torch.Size:
        return self._diag.shape[:-3]

    @property
    def inner_dim(self) -> int:
        return self._diag.shape[-2]

    @property
    def outer_dim(self) -> int:
        return self._diag.shape[-3]

    @property
    def block_diagonal(self) -> torch.Tensor:
        return self._diag

    @property
    def block_sub_diagonal(self) -> Optional[torch.Tensor]:
        return self._sub_diag

    @property
    def as_band(self) -> torch.Tensor:
        """
        The lower band of the represented N x N matrix, ``batch_shape + [bandwidth + 1, N]`` with row k holding the
        k-th sub-diagonal aligned on columns (``band[k, j] = M[j + k, j]``, zero past the end) - the layout of the
        reference's ``BandedMatrixTensor`` (block_tri_diag.py:84-98,206-237).  Nothing here consumes it: the kernels work
        on the blocks; it exists for API parity and for interchange with band-oriented code.
        """
        return self._convert_to_band()

    def _convert_to_band(self) -> torch.Tensor:
        d, t = self.inner_dim, self.outer_dim
        n, width = d * t, self.bandwidth + 1
        dev = self._diag.device
        k = torch.arange(width, device=dev)[:, None]
        j = torch.arange(n, device=dev)[None, :]
        i = j + k                                          # row of the entry
        c, q = j // d, j % d
        r, p = torch.clamp(i // d, max=t - 1), i % d
        in_diag = (i < n) & (i // d == c)
        band = torch.where(in_diag, self._diag[..., c.expand_as(i), p, q.expand_as(i)], torch.zeros((), dtype=self._diag.dtype, device=dev))
        if self._sub_diag is not None:
            in_sub = (i < n) & (i // d == c + 1)
            cs = torch.clamp(c, max=max(t - 2, 0)).expand_as(i)
            band = torch.where(in_sub, self._sub_diag[..., cs, p, q.expand_as(i)], band)
        del r
        return band

    @property
    def _batch_numel(self) -> int:
        n = 1
        for s in self.batch_shape:
            n *= s
        return n

    # -- dense view, for tests (block_tri_diag.py:150-173) --------------------------------------------
    def to_dense(self) -> torch.Tensor:
        n, d = self.outer_dim, self.inner_dim
        dense = self._diag.new_zeros(tuple(self.batch_shape) + (n * d, n * d))
        lower_blocks = torch.tril(self._diag)
        for i in range(n):
            dense[..., i * d:(i + 1) * d, i * d:(i + 1) * d] = lower_blocks[..., i, :, :]
            if self._sub_diag is not None and i < n - 1:
                dense[..., (i + 1) * d:(i + 2) * d, i * d:(i + 1) * d] = self._sub_diag[..., i, :, :]
        if self._symmetric:
            dg = torch.diagonal(dense, dim1=-2, dim2=-1)
            dense = dense + dense.transpose(-1, -2) - torch.diag_embed(dg)
        return dense

    # -- products (block_tri_diag.py:175-199) ----------------------------------------------------------
    def dense_mult(self, right: torch.Tensor, transpose_left: bool = False) -> torch.Tensor:
        """``L x`` (or ``Lᵀ x``; symmetric objects ignore nothing: transposing them is a no-op)."""
        mode = 2 if self._symmetric else (1 if transpose_left else 0)
        if _ag.needs_grad(self._diag, self._sub_diag, right):
            return _ag.BtdMatvec.apply(_matvec_kernel, self._diag, self._sub_diag, right, mode)
        return _matvec_kernel(self._diag, self._sub_diag, right, mode)

    @abc.abstractmethod
    def __add__(self, other):
        raise NotImplementedError

    def _add_parts(self, other: "BlockTriDiagonal") -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        # block_tri_diag.py:368-380,412-421
        if self._sub_diag is not None:
            sub_diag = self._sub_diag
            if other.block_sub_diagonal is not None:
                sub_diag = sub_diag + other.block_sub_diagonal
        else:
            sub_diag = other.block_sub_diagonal
        return self._diag + other.block_diagonal, sub_diag

    # -- right-hand sides (block_tri_diag.py:239-287) ---------------------------------------------------
    def _assert_compatible_right_shape(self, right: torch.Tensor) -> None:
        if right.dim() < 2 or tuple(right.shape[-2:]) != (self.outer_dim, self.inner_dim):
            raise ValueError(
                f"right has shape {tuple(right.shape)} but its last two dims must be "
                f"({self.outer_dim}, {self.inner_dim})"
            )
        if right.dim() > 2:
            nb = len(self.batch_shape)
            rb = tuple(right.shape[-(2 + nb):-2]) if right.dim() >= 2 + nb else tuple(right.shape[:-2])
            mine = tuple(self.batch_shape)[len(self.batch_shape) - len(rb):]
            for a, b in zip(rb, mine):
                if not (a == b or a == 1 or b == 1):
                    raise ValueError(f"right batch shape {rb} is not compatible with {tuple(self.batch_shape)}")
        if right.dtype != self._diag.dtype or right.device != self._diag.device:
            raise ValueError("right must share dtype and device with the operator")

    def _broadcast_right(self, right: torch.Tensor):
        """Return ``right`` as [Br, T, d] with Br a multiple of the operator's flattened batch, + output shape."""
        self._assert_compatible_right_shape(right)
        batch = tuple(self.batch_shape)
        lead = tuple(right.shape[:-2])
        full = tuple(torch.broadcast_shapes(lead, batch))
        # the operator must not be broadcast (that would need copies of the factor): only leading extra dims
        if tuple(full[len(full) - len(batch):]) != batch:
            raise ValueError(
                f"right batch shape {lead} would broadcast the operator batch {batch}; expand the operator instead"
            )
        right_b = right.expand(full + tuple(right.shape[-2:])).reshape(-1, self.outer_dim, self.inner_dim).contiguous()
        return right_b, full + (self.outer_dim, self.inner_dim)


class LowerTriangularBlockTriDiagonal(BlockTriDiagonal):
    """Lower triangular block tridiagonal matrix (reference: block_tri_diag.py:291-380)."""

    def __init__(self, diagonal: torch.Tensor, sub_diagonal: Optional[torch.Tensor] = None) -> None:
        super().__init__(diagonal, symmetric=False, sub_diagonal=sub_diagonal)

    def block_diagonal_of_inverse(self) -> torch.Tensor:
        """Diagonal blocks of ``(L Lᵀ)⁻¹`` (block_tri_diag.py:318-337)."""
        return self._diag_and_sub_of_inverse(want_sub=False)[0]

    def _diag_and_sub_of_inverse(self, want_sub: bool):
        if _ag.needs_grad(self._diag, self._sub_diag):
            return _ag.BtdInverseBlocks.apply(_inverse_blocks_kernel, self._diag, self._sub_diag, want_sub)
        return _inverse_blocks_kernel(self._diag, self._sub_diag, want_sub)

    def _diag_and_sub_of_inverse_kernel(self, want_sub: bool):
        diag = _flat(self._diag, 3)
        sub = None if self._sub_diag is None else _flat(self._sub_diag, 3)
        odiag = torch.empty_like(diag)
        osub = torch.empty_like(sub) if (want_sub and sub is not None) else None
        ws_bytes = int(_lib.load().mf_btd_diag_of_inverse_workspace_bytes(diag.shape[0], self.outer_dim, self.inner_dim,
                                                                           diag.element_size()))
        ws = _lib.workspace(ws_bytes, diag.device)
        _lib.call("mf_btd_diag_of_inverse", diag.dtype, diag.shape[0], self.outer_dim, self.inner_dim,
                  _lib.ptr(diag), _lib.ptr(sub), _lib.ptr(odiag), _lib.ptr(osub), _lib.ptr(ws), ws_bytes,
                  _lib.stream_ptr(diag.device))
        odiag = odiag.reshape(self._diag.shape)
        if osub is not None:
            osub = osub.reshape(self._sub_diag.shape)
        return odiag, osub

    def solve(self, right: torch.Tensor, transpose_left: bool = False) -> torch.Tensor:
        """``L⁻¹ x`` or ``L⁻ᵀ x`` (block_tri_diag.py:339-351).  Differentiable w.r.t. the factor and the right-hand side."""
        if _ag.needs_grad(self._diag, self._sub_diag, right):
            return _ag.BtdSolve.apply(_solve_kernel, self._diag, self._sub_diag, right, bool(transpose_left))
        return self._solve_kernel(right, transpose_left)

    def _solve_kernel(self, right: torch.Tensor, transpose_left: bool = False) -> torch.Tensor:
        right_b, out_shape = self._broadcast_right(right)
        diag = _flat(self._diag, 3)
        sub = None if self._sub_diag is None else _flat(self._sub_diag, 3)
        out = torch.empty_like(right_b)
        ws_bytes = int(_lib.load().mf_btd_solve_workspace_bytes(diag.shape[0], right_b.shape[0], self.outer_dim,
                                                                 self.inner_dim, diag.element_size()))
        ws = _lib.workspace(ws_bytes, diag.device)
        _lib.call("mf_btd_solve", diag.dtype, diag.shape[0], right_b.shape[0], self.outer_dim, self.inner_dim,
                  _lib.ptr(diag), _lib.ptr(sub), _lib.ptr(right_b), _lib.ptr(out), int(transpose_left),
                  _lib.ptr(ws), ws_bytes, _lib.stream_ptr(diag.device))
        return out.reshape(out_shape)

    def abs_log_det(self) -> torch.Tensor:
        """``log |det L|`` with shape ``batch_shape`` (block_tri_diag.py:353-366)."""
        if _ag.needs_grad(self._diag):
            return torch.sum(torch.log(torch.abs(torch.diagonal(self._diag, dim1=-2, dim2=-1))), dim=(-1, -2))
        diag = _flat(self._diag, 3)
        out = torch.empty(diag.shape[0], dtype=diag.dtype, device=diag.device)
        _lib.call("mf_btd_logdet", diag.dtype, diag.shape[0], self.outer_dim, self.inner_dim, _lib.ptr(diag),
                  _lib.ptr(out), _lib.stream_ptr(diag.device))
        return out.reshape(tuple(self.batch_shape))

    def __add__(self, other: "LowerTriangularBlockTriDiagonal") -> "LowerTriangularBlockTriDiagonal":
        return LowerTriangularBlockTriDiagonal(*self._add_parts(other))


class SymmetricBlockTriDiagonal(BlockTriDiagonal):
    """Symmetric block tridiagonal matrix - the form of every precision (block_tri_diag.py:384-545)."""

    def __init__(self, diagonal: torch.Tensor, sub_diagonal: Optional[torch.Tensor] = None) -> None:
        super().__init__(diagonal, symmetric=True, sub_diagonal=sub_diagonal)

    def __add__(self, other: "SymmetricBlockTriDiagonal") -> "SymmetricBlockTriDiagonal":
        return SymmetricBlockTriDiagonal(*self._add_parts(other))

    @property
    def cholesky(self) -> LowerTriangularBlockTriDiagonal:
        """Natural-order Cholesky factor (block_tri_diag.py:423-436).  Differentiable (``_autograd_ops.BtdCholesky``)."""
        if _ag.needs_grad(self._diag, self._sub_diag):
            ldiag, lsub = _ag.BtdCholesky.apply(_cholesky_kernel, self._diag, self._sub_diag)
            return LowerTriangularBlockTriDiagonal(ldiag, lsub)
        return self._cholesky_kernel()

    def _cholesky_kernel(self) -> LowerTriangularBlockTriDiagonal:
        diag = _flat(self._diag, 3)
        sub = None if self._sub_diag is None else _flat(self._sub_diag, 3)
        ldiag = torch.empty_like(diag)
        lsub = None if sub is None else torch.empty_like(sub)
        info = _lib.pivot_info(diag.device)
        ws_bytes = int(_lib.load().mf_btd_cholesky_workspace_bytes(diag.shape[0], self.outer_dim, self.inner_dim,
                                                                    diag.element_size()))
        ws = _lib.workspace(ws_bytes, diag.device)
        _lib.call("mf_btd_cholesky", diag.dtype, diag.shape[0], self.outer_dim, self.inner_dim, _lib.ptr(diag),
                  _lib.ptr(sub), _lib.ptr(ldiag), _lib.ptr(lsub), _lib.ptr(ws), ws_bytes, info,
                  _lib.stream_ptr(diag.device))
        _lib.raise_on_info(info, "SymmetricBlockTriDiagonal.cholesky", diag.device, blocks=diag.shape[-3])
        return LowerTriangularBlockTriDiagonal(
            ldiag.reshape(self._diag.shape), None if lsub is None else lsub.reshape(self._sub_diag.shape)
        )

    def upper_diagonal_lower(self) -> Tuple[LowerTriangularBlockTriDiagonal, LowerTriangularBlockTriDiagonal]:
        """``U D Uᵀ`` factorisation; returns ``(Uᵀ, chol_D)`` (block_tri_diag.py:438-545)."""
        assert self._sub_diag is not None  # block_tri_diag.py:486
        if _ag.needs_grad(self._diag, self._sub_diag):
            u_t, chol_d = self._udl_differentiable()
        else:
            u_t, chol_d, _, _ = self._udl(None)
        identities = torch.eye(self.inner_dim, dtype=self._diag.dtype, device=self._diag.device).expand(
            self._diag.shape).contiguous()
        return LowerTriangularBlockTriDiagonal(identities, u_t), LowerTriangularBlockTriDiagonal(chol_d)

    def _udl_differentiable(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """``(U^T sub-diagonal blocks, chol_D)`` as differentiable functions of the blocks.  ``P = U D U^T`` with ``D = C C^T`` is
        ``P = (U C)(U C)^T``, an UPPER block-bidiagonal factor with lower-triangular diagonal blocks - the Cholesky factor ``L`` of
        the TIME-REVERSED matrix read backwards: ``C_k = L_{n-1-k, n-1-k}`` and ``U_{k,k+1} C_{k+1} = L_{n-1-k, n-2-k}``.  The
        reference differentiates its banded UDL through TensorFlow (block_tri_diag.py:438-545 under a tape); here the adjoint of
        ``cholesky`` (``mf_btd_cholesky_grad_*``) does the work."""
        diag_r = torch.flip(self._diag, dims=(-3,)).contiguous()
        sub_r = torch.flip(self._sub_diag, dims=(-3,)).transpose(-1, -2).contiguous()
        chol = SymmetricBlockTriDiagonal(diag_r, sub_r).cholesky
        chol_d = torch.flip(chol.block_diagonal, dims=(-3,))
        coupled = torch.flip(chol.block_sub_diagonal, dims=(-3,))                        # U_{k,k+1} C_{k+1}
        u_t = torch.linalg.solve_triangular(chol_d[..., 1:, :, :].transpose(-1, -2), coupled.transpose(-1, -2), upper=True)
        return u_t, chol_d

    def _udl(self, eta: Optional[torch.Tensor], chain: bool = False):
        """``chain=True`` (needs ``eta``): the outputs come in the layout of a posterior ``StateSpaceModel`` - returns
        ``(transitions = -Uᵀ, chol_d, (mu0', offsets'), (cholP0', cholQ'))`` with every tensor contiguous."""
        diag, sub = _flat(self._diag, 3), _flat(self._sub_diag, 3)
        u_t, chol_d = torch.empty_like(sub), (None if chain else torch.empty_like(diag))   # the chain does not contain chol_D
        m_post = chol_dinv = eta_f = None
        if eta is not None:
            eta_f = _flat(eta, 2)
            m_post, chol_dinv = torch.empty_like(eta_f), torch.empty_like(diag)
        info = _lib.pivot_info(diag.device)
        ws_bytes = int(_lib.load().mf_btd_udl_workspace_bytes(diag.shape[0], self.outer_dim, self.inner_dim,
                                                               diag.element_size()))
        ws = _lib.workspace(ws_bytes, diag.device)
        _lib.call("mf_btd_udl", diag.dtype, diag.shape[0], self.outer_dim, self.inner_dim, _lib.ptr(diag),
                  _lib.ptr(sub), _lib.ptr(u_t), _lib.ptr(chol_d), _lib.ptr(eta_f), _lib.ptr(m_post),
                  _lib.ptr(chol_dinv), int(chain), _lib.ptr(ws), ws_bytes, info, _lib.stream_ptr(diag.device))
        _lib.raise_on_info(info, "SymmetricBlockTriDiagonal.upper_diagonal_lower", diag.device)
        u_t = u_t.reshape(self._sub_diag.shape)
        chol_d = None if chol_d is None else chol_d.reshape(self._diag.shape)
        if chain:
            bsz, n, d = diag.shape[0], self.outer_dim, self.inner_dim
            batch = tuple(self.batch_shape)
            mf, cf = m_post.reshape(-1), chol_dinv.reshape(-1)
            means = (mf[:bsz * d].reshape(batch + (d,)), mf[bsz * d:].reshape(batch + (n - 1, d)))
            chols = (cf[:bsz * d * d].reshape(batch + (d, d)), cf[bsz * d * d:].reshape(batch + (n - 1, d, d)))
            return u_t, chol_d, means, chols
        if eta is not None:
            m_post, chol_dinv = m_post.reshape(eta.shape), chol_dinv.reshape(self._diag.shape)
        return u_t, chol_d, m_post, chol_dinv


# the kernels behind the differentiable operators, as plain functions of tensors (what _autograd_ops' Functions call)
def _cholesky_kernel(diag, sub):
    chol = SymmetricBlockTriDiagonal(diag.contiguous(), None if sub is None else sub.contiguous())._cholesky_kernel()
    return chol.block_diagonal, chol.block_sub_diagonal


def _solve_kernel(ldiag, lsub, rhs, transpose):
    return LowerTriangularBlockTriDiagonal(ldiag, lsub)._solve_kernel(rhs, transpose)


def _inverse_blocks_kernel(ldiag, lsub, want_sub):
    return LowerTriangularBlockTriDiagonal(ldiag, lsub)._diag_and_sub_of_inverse_kernel(want_sub)


def _matvec_kernel(diag, sub, right, mode):
    op = BlockTriDiagonal.__new__(SymmetricBlockTriDiagonal if mode == 2 else LowerTriangularBlockTriDiagonal)
    BlockTriDiagonal.__init__(op, diag, mode == 2, sub)
    right_b, out_shape = op._broadcast_right(right)
    out = torch.empty_like(right_b)
    sub_f = None if sub is None else _flat(sub, 3)
    _lib.call("mf_btd_matvec", diag.dtype, op._batch_numel, right_b.shape[0], op.outer_dim, op.inner_dim,
              _lib.ptr(_flat(diag, 3)), _lib.ptr(sub_f), _lib.ptr(right_b), _lib.ptr(out), mode, _lib.stream_ptr(diag.device))
    return out.reshape(out_shape)


def _banded_to_block_tri(banded: torch.Tensor, block_size: int) -> LowerTriangularBlockTriDiagonal:
    """
    Lower band ``[..., K, N]`` (K = d or 2 d rows, layout of :meth:`BlockTriDiagonal.as_band`) -> lower-triangular block
    tridiagonal (block_tri_diag.py:549-592 of the reference).
    """
    d = block_size
    width, n = banded.shape[-2], banded.shape[-1]
    if n % d != 0 or width not in (d, 2 * d):
        raise ValueError(f"band of shape {tuple(banded.shape)} is not block tridiagonal with block size {d}")
    t, dev = n // d, banded.device
    c = torch.arange(t, device=dev)[:, None, None]
    p = torch.arange(d, device=dev)[None, :, None]
    q = torch.arange(d, device=dev)[None, None, :]
    col = (c * d + q).expand(t, d, d)
    kd = (p - q).expand(t, d, d)
    zero = torch.zeros((), dtype=banded.dtype, device=dev)
    diag = torch.where(kd >= 0, banded[..., torch.clamp(kd, min=0), col], zero)
    sub = None
    if width == 2 * d:
        ks = (d + p - q).expand(t, d, d)
        sub = torch.where(ks < width, banded[..., torch.clamp(ks, max=width - 1), col], zero)[..., :-1, :, :]
    return LowerTriangularBlockTriDiagonal(diag, sub)
