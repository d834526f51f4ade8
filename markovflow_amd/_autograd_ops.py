"""
Reverse mode through the block-tridiagonal OPERATORS.

banded_matrices registers a gradient for every op (``/root/reference/markovflow/block_tri_diag.py:22-31``), so TensorFlow
differentiates through ``precision``, ``cholesky``, ``solve``, ``block_diagonal_of_inverse`` - the CVI models build their
posterior exactly that way (``models/variational_cvi.py:105-136``: ``dist_p.precision -> naturals_to_ssm_params``) and
differentiate the ELBO w.r.t. the kernel's hyper-parameters through it (``:402``).  Here: ``torch.autograd.Function``s whose
FORWARD is the HIP kernel behind the operator and whose backward is

* ``solve`` / ``dense_mult`` / ``abs_log_det``: closed forms on the operator's own kernels (one transposed solve / product +
  outer products that are local in time) - as fast as the forward;
* ``cholesky``: HIP (``mf_btd_cholesky_grad_*``, csrc/mf_btd_par.hpp).  The block Cholesky is a LOCAL map ``(P_k, S_k) -> (L_k, W_k)``
  behind the recursion ``P_k = D_k - S_{k-1} P_{k-1}^-1 S_{k-1}^T``, so its adjoint is a kernel that is local in time (the classic
  ``L^-T Phi(L^T Lbar) L^-1`` per pivot), the congruence recursion ``Z_k = C_k + G_k^T Z_{k+1} G_k`` with the block Takahashi coupling
  ``G_k = W_k L_k^-1`` (the scan of the marginals' adjoint: parallel in time for few series, a lane per series for many) and an axpy;
* ``block_diagonal_of_inverse`` (+ sub-diagonal blocks): HIP (``mf_btd_diag_of_inverse_grad_*``): the same recursion run forward
  (``A_{k+1} = Qbar_{k+1} + G_k A_k G_k^T``) between two local kernels.
  10 <= d <= 32: the same split on register MFMA tiles (csrc/mf_adj.hip).  Beyond the kernels (d > 32, or under ``create_graph``)
  a GPU tensor takes the same split in batched torch products with the congruence recursion as a Hillis-Steele scan in time
  (``_cholesky_backward_scan`` / ``_inverse_blocks_backward_scan``: log2 T rounds, no Python loop over the blocks); the
  block-by-block loops (``_cholesky_backward_torch`` / ``_inverse_blocks_backward_torch``) remain for CPU tensors and are what the
  tests compare everything against.

Values always come from the HIP kernels.
"""
from typing import Optional

import torch

from . import _lib


def _hip_grad_ws(ldiag: torch.Tensor):
    """``(B, n, d, workspace bytes)`` when the operator-adjoint kernels cover these blocks (HIP tensor; d <= 9 and 10 <= d <= 32,
    ``csrc/mf_adj.hip``: local kernels + a congruence scan in time), else None."""
    if not ldiag.is_cuda:
        return None
    n, d = ldiag.shape[-3], ldiag.shape[-1]
    bsz = 1
    for x in ldiag.shape[:-3]:
        bsz *= int(x)
    if bsz == 0:
        return None
    wsb = int(_lib.load().mf_btd_grad_workspace_bytes(bsz, n, d, ldiag.element_size()))
    return (bsz, n, d, wsb) if wsb else None


def _tr(x):
    return x.transpose(-1, -2)


def _chol_adjoint(chol: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """Adjoint of ``L = chol(P)`` for symmetric ``P`` (batched): ``sym(L^-T Phi(L^T Lbar) L^-1)``, ``Phi`` = lower triangle
    with the diagonal halved."""
    phi = torch.tril(_tr(chol) @ torch.tril(g))
    phi = phi - 0.5 * torch.diag_embed(torch.diagonal(phi, dim1=-2, dim2=-1))
    x = torch.linalg.solve_triangular(_tr(chol), phi, upper=True)               # L^-T Phi
    x = torch.linalg.solve_triangular(chol, x, upper=False, left=False)         # ... L^-1
    return 0.5 * (x + _tr(x))


def _scan_pays(t: torch.Tensor) -> bool:
    """A GPU tensor with more than a handful of blocks: the scan forms below instead of a Python loop over the blocks."""
    return t.is_cuda and t.shape[-3] > 8


def _congruence_scan(g: torch.Tensor, c: torch.Tensor, backward: bool) -> torch.Tensor:
    """``backward``: ``X_k = C_k + G_k^T X_{k+1} G_k`` (``X_{n-1} = C_{n-1}``); else ``X_{k+1} = C_{k+1} + G_k X_k G_k^T`` (``X_0 = C_0``).
    ``g [.., n-1, d, d]``, ``c [.., n, d, d]``.  Hillis-Steele over the maps ``X -> M^T X M + N`` (``M X M^T + N`` forwards): element k
    absorbs the composite ``off`` blocks further along; log2(n) rounds of batched products, differentiable."""
    n = c.shape[-3]
    if n == 1:
        return c
    if not backward:                                   # the forward recursion is the backward one on the reversed chain with G^T
        return torch.flip(_congruence_scan(torch.flip(_tr(g), dims=(-3,)), torch.flip(c, dims=(-3,)), True), dims=(-3,))
    m = torch.cat([g, torch.zeros_like(g[..., :1, :, :])], dim=-3)      # block n - 1 has nothing beyond it
    acc = c
    off = 1
    while off < n:
        m_far, n_far = m[..., off:, :, :], acc[..., off:, :, :]
        m_near = m[..., :-off, :, :]
        acc = torch.cat([acc[..., :-off, :, :] + _tr(m_near) @ n_far @ m_near, acc[..., -off:, :, :]], dim=-3)
        m = torch.cat([m_far @ m_near, torch.zeros_like(m[..., -off:, :, :])], dim=-3)
        off *= 2
    return acc


def _cholesky_backward_scan(ldiag, lsub, g_ldiag, g_lsub):
    """The adjoint of the block Cholesky as terms local in time + ONE congruence recursion (the split of csrc/mf_btd_par.hpp /
    mf_adj.hip): ``Sbar(loc) = Wbar L^-1``, ``Lbar(eff) = Lbar - tril(Sbar(loc)^T W)``, ``C = sym(L^-T Phi(L^T Lbar(eff)) L^-1)``,
    ``Z_k = C_k + G_k^T Z_{k+1} G_k`` with ``G = W L^-1``; ``Dbar = Z``, ``Sbar_k = Sbar_k(loc) - 2 Z_{k+1} G_k``."""
    lbar = torch.tril(g_ldiag) if g_ldiag is not None else torch.zeros_like(ldiag)
    if lsub is None:
        return _chol_adjoint(ldiag, lbar), None
    low = torch.tril(ldiag)
    eye = torch.eye(ldiag.shape[-1], dtype=ldiag.dtype, device=ldiag.device).expand(ldiag.shape)
    linv = torch.linalg.solve_triangular(low, eye, upper=False)
    g = lsub @ linv[..., :-1, :, :]
    if g_lsub is not None:
        s_loc = g_lsub @ linv[..., :-1, :, :]
        corr = torch.tril(_tr(s_loc) @ lsub)
        lbar = torch.cat([lbar[..., :-1, :, :] - corr, lbar[..., -1:, :, :]], dim=-3)
    else:
        s_loc = torch.zeros_like(lsub)
    z = _congruence_scan(g, _chol_adjoint(ldiag, lbar), True)
    return z, s_loc - 2.0 * z[..., 1:, :, :] @ g


def _inverse_blocks_backward_scan(ldiag, lsub, sigma, g_diag, g_sub):
    """The adjoint of the block Takahashi recursion in the same split (``sigma``: the diagonal blocks of the inverse, the forward's
    output): ``A_0 = sym(Sbar_0)``, ``A_{k+1} = sym(Sbar_{k+1}) - sym(subbar_k G_k^T) + G_k A_k G_k^T``;
    ``Lbar_k = -2 tril(L^-T (L^-1 A_k L^-T)) - tril(G_k^T Wbar_k)``, ``Wbar_k = (2 Sigma_{k+1} G_k A_k - Sigma_{k+1} subbar_k) L_k^-T``."""
    low = torch.tril(ldiag)
    eye = torch.eye(ldiag.shape[-1], dtype=ldiag.dtype, device=ldiag.device).expand(ldiag.shape)
    linv = torch.linalg.solve_triangular(low, eye, upper=False)
    q = 0.5 * (g_diag + _tr(g_diag)) if g_diag is not None else torch.zeros_like(ldiag)
    if lsub is None:
        a = q
        g = None
    else:
        g = lsub @ linv[..., :-1, :, :]
        if g_sub is not None:
            x = g_sub @ _tr(g)
            q = torch.cat([q[..., :1, :, :], q[..., 1:, :, :] - 0.5 * (x + _tr(x))], dim=-3)
        a = _congruence_scan(g, q, False)
    lbar = -2.0 * torch.tril(_tr(linv) @ (linv @ a @ _tr(linv)))
    if lsub is None:
        return lbar, None
    x = 2.0 * g @ a[..., :-1, :, :]
    if g_sub is not None:
        x = x - g_sub
    wbar = sigma[..., 1:, :, :] @ x @ _tr(linv[..., :-1, :, :])
    lbar = torch.cat([lbar[..., :-1, :, :] - torch.tril(_tr(g) @ wbar), lbar[..., -1:, :, :]], dim=-3)
    return lbar, wbar


class BtdCholesky(torch.autograd.Function):
    """``SymmetricBlockTriDiagonal.cholesky`` (block_tri_diag.py:423-436).  ``run(diag, sub) -> (ldiag, lsub)`` is the kernel."""

    @staticmethod
    def forward(ctx, run, diag, sub):
        ldiag, lsub = run(diag.detach(), None if sub is None else sub.detach())
        ctx.has_sub = sub is not None
        ctx.save_for_backward(ldiag, lsub if lsub is not None else ldiag.new_zeros(0))
        if lsub is None:
            return ldiag, None
        return ldiag, lsub

    @staticmethod
    def backward(ctx, g_ldiag, g_lsub):
        ldiag, lsub = ctx.saved_tensors
        lsub = lsub if ctx.has_sub else None
        # (under create_graph=True the backward runs with the tape ON: the raw adjoint kernels would drop the second-order terms
        # silently, so that case takes the torch expressions below, which are recorded)
        plan = None if torch.is_grad_enabled() else _hip_grad_ws(ldiag)
        if plan is None:
            if _scan_pays(ldiag):
                return (None,) + _cholesky_backward_scan(ldiag, lsub, g_ldiag, g_lsub)
            return (None,) + _cholesky_backward_torch(ldiag, lsub, g_ldiag, g_lsub)
        bsz, n, d, wsb = plan
        flat = lambda t: None if t is None else t.reshape((bsz, -1, d, d)).contiguous()           # noqa: E731
        g_diag = torch.empty_like(ldiag, memory_format=torch.contiguous_format)
        g_sub = torch.empty_like(lsub, memory_format=torch.contiguous_format) if lsub is not None else None
        ws = _lib.workspace(wsb, ldiag.device)
        with torch.no_grad():
            _lib.call("mf_btd_cholesky_grad", ldiag.dtype, bsz, n, d, _lib.ptr(flat(ldiag)), _lib.ptr(flat(lsub)),
                      _lib.ptr(flat(g_ldiag)), _lib.ptr(flat(g_lsub) if lsub is not None else None), _lib.ptr(g_diag),
                      _lib.ptr(g_sub), _lib.ptr(ws), wsb, _lib.stream_ptr(ldiag.device))
        return None, g_diag, g_sub


def _cholesky_backward_torch(ldiag, lsub, g_ldiag, g_lsub):
    """The adjoint of the block Cholesky as a backward sweep of batched torch products (d > 9, CPU tensors; the reference the HIP
    kernels are tested against)."""
    n = ldiag.shape[-3]
    lbar = torch.tril(g_ldiag).clone() if g_ldiag is not None else torch.zeros_like(ldiag)
    if lsub is None:
        return _chol_adjoint(ldiag, lbar), None                     # independent blocks: one batched adjoint
    wbar_in = g_lsub if g_lsub is not None else torch.zeros_like(lsub)
    dbar, sbar = torch.empty_like(ldiag), torch.empty_like(lsub)
    for k in range(n - 1, -1, -1):
        pbar = _chol_adjoint(ldiag[..., k, :, :], lbar[..., k, :, :])
        dbar[..., k, :, :] = pbar
        if k > 0:
            w = lsub[..., k - 1, :, :]
            wbar = wbar_in[..., k - 1, :, :] - 2.0 * pbar @ w                            # P_k = D_k - W W^T
            sb = torch.linalg.solve_triangular(ldiag[..., k - 1, :, :], wbar, upper=False, left=False)   # W = S L^-T
            sbar[..., k - 1, :, :] = sb
            lbar[..., k - 1, :, :] -= torch.tril(_tr(sb) @ w)
    return dbar, sbar


class BtdSolve(torch.autograd.Function):
    """``LowerTriangularBlockTriDiagonal.solve`` (block_tri_diag.py:339-351): ``x = L^-1 r`` or ``L^-T r``; ``r`` may carry
    extra leading dimensions (one factor, many right-hand sides).  ``run(ldiag, lsub, rhs, transpose)`` is the kernel."""

    @staticmethod
    def forward(ctx, run, ldiag, lsub, rhs, transpose):
        out = run(ldiag.detach(), None if lsub is None else lsub.detach(), rhs.detach(), transpose)
        ctx.run, ctx.transpose, ctx.has_sub = run, transpose, lsub is not None
        ctx.save_for_backward(ldiag, lsub if lsub is not None else ldiag.new_zeros(0), out)
        ctx.rhs_shape = rhs.shape
        return out

    @staticmethod
    def backward(ctx, g_out):
        ldiag, lsub, x = ctx.saved_tensors
        lsub = lsub if ctx.has_sub else None
        rbar = ctx.run(ldiag, lsub, g_out.contiguous(), not ctx.transpose)       # r-bar = L^-T x-bar (resp. L^-1 x-bar)
        # L-bar = -band(r-bar x^T) for L^-1, -band(x r-bar^T) for L^-T; extra leading dimensions of the right-hand side are summed
        a, b = (rbar, x) if not ctx.transpose else (x, rbar)
        extra = a.dim() - (ldiag.dim() - 1)
        red = tuple(range(extra))
        gd = -(a[..., :, :, None] * b[..., :, None, :])
        gd = torch.tril(gd.sum(dim=red) if red else gd)
        gs = None
        if ctx.has_sub:
            gs = -(a[..., 1:, :, None] * b[..., :-1, None, :])
            gs = gs.sum(dim=red) if red else gs
        g_rhs = rbar
        while g_rhs.dim() > len(ctx.rhs_shape):
            g_rhs = g_rhs.sum(dim=0)
        for i, (have, want) in enumerate(zip(g_rhs.shape, ctx.rhs_shape)):
            if want == 1 and have != 1:
                g_rhs = g_rhs.sum(dim=i, keepdim=True)
        return None, gd, gs, g_rhs, None


class BtdMatvec(torch.autograd.Function):
    """``BlockTriDiagonal.dense_mult`` (block_tri_diag.py:175-199); mode 0: ``L x``, 1: ``L^T x``, 2: symmetric ``M x``."""

    @staticmethod
    def forward(ctx, run, diag, sub, right, mode):
        out = run(diag.detach(), None if sub is None else sub.detach(), right.detach(), mode)
        ctx.run, ctx.mode, ctx.has_sub = run, mode, sub is not None
        ctx.save_for_backward(diag, sub if sub is not None else diag.new_zeros(0), right)
        return out

    @staticmethod
    def backward(ctx, g):
        diag, sub, x = ctx.saved_tensors
        sub = sub if ctx.has_sub else None
        mode = ctx.mode
        g = g.contiguous()
        gx = ctx.run(diag, sub, g, {0: 1, 1: 0, 2: 2}[mode])
        extra = g.dim() - (diag.dim() - 1)
        red = tuple(range(extra))
        outer = lambda u, v: (u[..., :, None] * v[..., None, :])            # noqa: E731
        if mode == 0:
            gd, gs = torch.tril(outer(g, x)), (outer(g[..., 1:, :], x[..., :-1, :]) if sub is not None else None)
        elif mode == 1:
            gd, gs = torch.tril(outer(x, g)), (outer(x[..., 1:, :], g[..., :-1, :]) if sub is not None else None)
        else:       # symmetric: the lower triangle of the blocks stands for both; the gradient is w.r.t. the symmetric blocks
            full = outer(g, x)
            gd = 0.5 * (full + _tr(full))
            gs = (outer(g[..., 1:, :], x[..., :-1, :]) + outer(x[..., 1:, :], g[..., :-1, :])) if sub is not None else None
        if red:
            gd = gd.sum(dim=red)
            gs = None if gs is None else gs.sum(dim=red)
        while gx.dim() > x.dim():
            gx = gx.sum(dim=0)
        for i, (have, want) in enumerate(zip(gx.shape, x.shape)):
            if want == 1 and have != 1:
                gx = gx.sum(dim=i, keepdim=True)
        return None, gd, gs, gx, None


class BtdInverseBlocks(torch.autograd.Function):
    """Diagonal (and sub-diagonal) blocks of ``(L L^T)^-1`` (block_tri_diag.py:318-337; block Takahashi)."""

    @staticmethod
    def forward(ctx, run, ldiag, lsub, want_sub):
        odiag, osub = run(ldiag.detach(), None if lsub is None else lsub.detach(), want_sub)
        ctx.has_sub, ctx.want_sub = lsub is not None, want_sub and lsub is not None
        ctx.save_for_backward(ldiag, lsub if lsub is not None else ldiag.new_zeros(0), odiag)
        return odiag, osub

    @staticmethod
    def backward(ctx, g_diag, g_sub):
        ldiag, lsub, odiag = ctx.saved_tensors
        lsub = lsub if ctx.has_sub else None
        if not ctx.want_sub:
            g_sub = None
        # (under create_graph=True the backward runs with the tape ON: the raw adjoint kernels would drop the second-order terms
        # silently, so that case takes the torch expressions below, which are recorded)
        plan = None if torch.is_grad_enabled() else _hip_grad_ws(ldiag)
        if plan is None:
            if _scan_pays(ldiag):
                g_chol, g_w = _inverse_blocks_backward_scan(ldiag, lsub, odiag, g_diag, g_sub)
            else:
                g_chol, g_w = _inverse_blocks_backward_torch(ldiag, lsub, g_diag, g_sub)
            return None, g_chol, g_w, None
        bsz, n, d, wsb = plan
        flat = lambda t: None if t is None else t.reshape((bsz, -1, d, d)).contiguous()           # noqa: E731
        g_ldiag = torch.empty_like(ldiag, memory_format=torch.contiguous_format)
        g_lsub = torch.empty_like(lsub, memory_format=torch.contiguous_format) if lsub is not None else None
        ws = _lib.workspace(wsb, ldiag.device)
        with torch.no_grad():
            _lib.call("mf_btd_diag_of_inverse_grad", ldiag.dtype, bsz, n, d, _lib.ptr(flat(ldiag)), _lib.ptr(flat(lsub)),
                      _lib.ptr(flat(odiag)), _lib.ptr(flat(g_diag)), _lib.ptr(flat(g_sub) if lsub is not None else None),
                      _lib.ptr(g_ldiag), _lib.ptr(g_lsub), _lib.ptr(ws), wsb, _lib.stream_ptr(ldiag.device))
        return None, g_ldiag, g_lsub, None


def _inverse_blocks_backward_torch(ldiag, lsub, g_diag, g_sub):
    """The block Takahashi recursion re-evaluated in differentiable torch ops and differentiated by torch (d > 9, CPU tensors; the
    reference the HIP kernels are tested against)."""
    with torch.enable_grad():
        chol = ldiag.detach().requires_grad_(True)
        w = lsub.detach().requires_grad_(True) if lsub is not None else None
        eye = torch.eye(chol.shape[-1], dtype=chol.dtype, device=chol.device).expand(chol.shape)
        linv = torch.linalg.solve_triangular(torch.tril(chol), eye, upper=False)
        base = _tr(linv) @ linv
        n = chol.shape[-3]
        if w is None:
            sig, subs = base, None
        else:
            gk = w @ linv[..., :-1, :, :]                               # G_k = W_k L_k^-1
            blocks, sblocks = [None] * n, [None] * (n - 1)
            blocks[n - 1] = base[..., n - 1, :, :]
            for k in range(n - 2, -1, -1):
                g = gk[..., k, :, :]
                blocks[k] = base[..., k, :, :] + _tr(g) @ blocks[k + 1] @ g
                sblocks[k] = -blocks[k + 1] @ g
            sig = torch.stack(blocks, dim=-3)
            subs = torch.stack(sblocks, dim=-3) if n > 1 else None
        outs, gouts = [sig], [g_diag if g_diag is not None else torch.zeros_like(sig)]
        if subs is not None and g_sub is not None:
            outs.append(subs)
            gouts.append(g_sub)
        ins = [chol] + ([w] if w is not None else [])
        grads = torch.autograd.grad(outs, ins, gouts, allow_unused=True)
    g_chol = torch.tril(grads[0]) if grads[0] is not None else None
    g_w = grads[1] if w is not None else None
    return g_chol, g_w


def _block_matmul_kernel(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """``x @ y`` over equally shaped stacks of ``d x d`` blocks: ``mf_block_matmul_*`` (a lane per block)."""
    d = x.shape[-1]
    xf, yf = x.reshape(-1, d, d).contiguous(), y.reshape(-1, d, d).contiguous()
    out = torch.empty_like(xf)
    n = xf.shape[0]
    if n:
        _lib.call("mf_block_matmul", x.dtype, 1, n, d, _lib.ptr(xf), n, _lib.ptr(yf), n, _lib.ptr(out), _lib.stream_ptr(x.device))
    return out.reshape(x.shape)


class BlockMatmul(torch.autograd.Function):
    """Products of stacks of small blocks, ``[..., d, d] @ [..., d, d]``, forward and backward on ``mf_block_matmul_*``: the
    batched GEMM torch dispatches for 640 000 blocks of 6 x 6 takes 1 - 2 ms a call (a tile kernel built for large matrices),
    the lane-per-block kernel moves the same bytes in 0.1 ms (profiles/r05_cvi_chain.txt)."""

    @staticmethod
    def forward(ctx, x, y):
        ctx.save_for_backward(x, y)
        return _block_matmul_kernel(x.detach(), y.detach())

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        # through block_matmul() itself: under create_graph=True the products are nodes again (Hessian-vector products)
        g = g.contiguous()
        gx = block_matmul(g, _tr(y).contiguous()) if ctx.needs_input_grad[0] else None
        gy = block_matmul(_tr(x).contiguous(), g) if ctx.needs_input_grad[1] else None
        return gx, gy


def block_matmul(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """``x @ y`` for equally shaped stacks of square blocks; the HIP kernel for HIP tensors of state dimension <= 9, torch else."""
    if (x.is_cuda and x.shape == y.shape and x.dim() >= 3 and x.shape[-1] == x.shape[-2] and x.dtype == y.dtype
            and x.shape[-1] <= _lib.load().mf_max_state_dim()):
        if needs_grad(x, y):
            return BlockMatmul.apply(x, y)
        return _block_matmul_kernel(x, y)
    return x @ y


def chol_solve_blocks(chol: torch.Tensor, rhs: torch.Tensor) -> torch.Tensor:
    """``(chol chol^T)^-1 rhs`` for stacks of blocks ``[..., d, d]`` against ``[..., d, k]``.  On the device (d <= 9): the blocks
    as ONE block-diagonal factor and the k columns as leading right-hand-side dimensions of ``LowerTriangularBlockTriDiagonal.solve``
    (a lane per (column, block); differentiable through its own adjoint) - rocBLAS' batched trsm takes 0.7 ms per call on
    640 000 blocks of 6 x 6, twelve calls per evaluation of the CVI chain (profiles/r05_cvi_chain_before.txt)."""
    d = chol.shape[-1]
    if not (chol.is_cuda and chol.dim() >= 3 and d <= _lib.load().mf_max_state_dim() and chol.shape[:-2] == rhs.shape[:-2]
            and chol.numel() > 0):
        return _lib.chol_solve(chol, rhs)
    from .block_tri_diag import LowerTriangularBlockTriDiagonal
    k = rhs.shape[-1]
    fac = LowerTriangularBlockTriDiagonal(chol.reshape(1, -1, d, d).contiguous())
    cols = rhs.reshape(-1, d, k).permute(2, 0, 1).reshape(k, 1, -1, d).contiguous()          # [k, 1, blocks, d]
    sol = fac.solve(fac.solve(cols), transpose_left=True)
    return sol.reshape(k, -1, d).permute(1, 2, 0).reshape(rhs.shape)


def needs_grad(*tensors: Optional[torch.Tensor]) -> bool:
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)
