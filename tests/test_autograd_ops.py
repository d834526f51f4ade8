"""
The adjoints behind the differentiable block-tridiagonal operators (markovflow_amd/_autograd_ops.py) WITHOUT a GPU: each
`torch.autograd.Function` takes the kernel it wraps as a callable, here a dense torch stand-in, and its backward is compared with
torch's own reverse mode through the dense matrix (fp64, rtol 1e-9).  The reference differentiates these operators through
banded_matrices' registered gradients (/root/reference/markovflow/block_tri_diag.py:22-31).  On the GPU the same Functions run
over the HIP kernels: tests/test_gpu_autograd_ops.py.
"""
import numpy as np
import pytest
import torch

from markovflow_amd import _autograd_ops as ag

F64 = torch.float64


def dense_of(diag, sub, symmetric):
    *batch, n, d, _ = diag.shape
    m = diag.new_zeros(tuple(batch) + (n * d, n * d))
    for k in range(n):
        blk = diag[..., k, :, :]
        m[..., k * d:(k + 1) * d, k * d:(k + 1) * d] = blk if symmetric else torch.tril(blk)
        if sub is not None and k + 1 < n:
            m[..., (k + 1) * d:(k + 2) * d, k * d:(k + 1) * d] = sub[..., k, :, :]
            if symmetric:
                m[..., k * d:(k + 1) * d, (k + 1) * d:(k + 2) * d] = sub[..., k, :, :].transpose(-1, -2)
    return m


def blocks_of(m, n, d, want_sub=True):
    diag = torch.stack([m[..., k * d:(k + 1) * d, k * d:(k + 1) * d] for k in range(n)], dim=-3)
    sub = torch.stack([m[..., (k + 1) * d:(k + 2) * d, k * d:(k + 1) * d] for k in range(n - 1)], dim=-3) if (want_sub and n > 1) else None
    return diag, sub


def random_spd(rng, batch, n, d):
    ld = np.tril(0.3 * rng.normal(size=batch + (n, d, d)), k=-1) + (1.0 + np.abs(0.3 * rng.normal(size=batch + (n, d))))[..., None] * np.eye(d)
    ls = 0.3 * rng.normal(size=batch + (n - 1, d, d))
    ldt, lst = torch.tensor(ld, dtype=F64), torch.tensor(ls, dtype=F64)
    full = dense_of(ldt, lst, False)
    return blocks_of(full @ full.transpose(-1, -2), n, d), (ldt, lst)


# ---- dense stand-ins for the kernels ---------------------------------------------------------------------------------------
def chol_run(diag, sub):
    n, d = diag.shape[-3], diag.shape[-1]
    chol = torch.linalg.cholesky(dense_of(diag, sub, True))
    ld, ls = blocks_of(chol, n, d, want_sub=sub is not None)
    return ld.contiguous(), None if ls is None else ls.contiguous()


def solve_run(ldiag, lsub, rhs, transpose):
    n, d = ldiag.shape[-3], ldiag.shape[-1]
    full = dense_of(ldiag, lsub, False)
    flat = rhs.reshape(rhs.shape[:-2] + (n * d, 1))
    out = torch.linalg.solve_triangular(full.transpose(-1, -2) if transpose else full, flat, upper=bool(transpose))
    return out.reshape(torch.broadcast_shapes(rhs.shape[:-2], ldiag.shape[:-3]) + (n, d))


def inv_run(ldiag, lsub, want_sub):
    n, d = ldiag.shape[-3], ldiag.shape[-1]
    full = dense_of(ldiag, lsub, False)
    inv = torch.linalg.inv(full @ full.transpose(-1, -2))
    dg, sb = blocks_of(inv, n, d, want_sub=want_sub and lsub is not None)
    return dg.contiguous(), None if sb is None else sb.contiguous()


def matvec_run(diag, sub, right, mode):
    n, d = diag.shape[-3], diag.shape[-1]
    full = dense_of(diag, sub, mode == 2)
    if mode == 1:
        full = full.transpose(-1, -2)
    flat = right.reshape(right.shape[:-2] + (n * d, 1))
    return (full @ flat).reshape(torch.broadcast_shapes(right.shape[:-2], diag.shape[:-3]) + (n, d))


@pytest.mark.parametrize("batch,n,d,with_sub", [((), 5, 3, True), ((2,), 4, 2, True), ((3,), 6, 3, False), ((), 1, 2, False)])
def test_cholesky_adjoint(batch, n, d, with_sub):
    rng = np.random.default_rng(0)
    (diag, sub), _ = random_spd(rng, batch, max(n, 2), d)
    diag, sub = diag[..., :n, :, :].clone(), (sub[..., :n - 1, :, :].clone() if (with_sub and n > 1) else None)
    if not with_sub:
        diag = diag + 2 * torch.eye(d, dtype=F64)
    gl = torch.tensor(rng.normal(size=diag.shape), dtype=F64)
    gw = None if sub is None else torch.tensor(rng.normal(size=sub.shape), dtype=F64)

    def loss(fn, dg, sb):
        ld, ls = fn(dg, sb)
        out = torch.sum(torch.tril(ld) * gl)
        return out if ls is None else out + torch.sum(ls * gw)

    d1 = diag.clone().requires_grad_(True)
    s1 = None if sub is None else sub.clone().requires_grad_(True)
    loss(lambda a, b: ag.BtdCholesky.apply(chol_run, a, b), d1, s1).backward()
    d2 = diag.clone().requires_grad_(True)
    s2 = None if sub is None else sub.clone().requires_grad_(True)
    # dense reverse mode; the symmetric blocks enter through their symmetrised form, as torch.linalg.cholesky's adjoint assumes
    loss(lambda a, b: chol_run(0.5 * (a + a.transpose(-1, -2)), b), d2, s2).backward()
    np.testing.assert_allclose(d1.grad.numpy(), d2.grad.numpy(), rtol=1e-9, atol=1e-11)
    if sub is not None:
        np.testing.assert_allclose(s1.grad.numpy(), s2.grad.numpy(), rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("transpose", [False, True])
@pytest.mark.parametrize("batch,lead,n,d,with_sub", [((), (), 5, 3, True), ((2,), (), 4, 2, True), ((2,), (3,), 4, 2, True), ((), (2,), 3, 2, False)])
def test_solve_adjoint(batch, lead, n, d, with_sub, transpose):
    rng = np.random.default_rng(1)
    _, (ld, ls) = random_spd(rng, batch, n, d)
    if not with_sub:
        ls = None
    rhs = torch.tensor(rng.normal(size=lead + batch + (n, d)), dtype=F64)
    gout = torch.tensor(rng.normal(size=lead + batch + (n, d)), dtype=F64)
    outs = []
    for fn in (lambda a, b, r: ag.BtdSolve.apply(solve_run, a, b, r, transpose), lambda a, b, r: solve_run(a, b, r, transpose)):
        a = ld.clone().requires_grad_(True)
        b = None if ls is None else ls.clone().requires_grad_(True)
        r = rhs.clone().requires_grad_(True)
        torch.sum(fn(a, b, r) * gout).backward()
        outs.append((torch.tril(a.grad), None if b is None else b.grad, r.grad))
    for g1, g2 in zip(*outs):
        if g1 is not None:
            np.testing.assert_allclose(g1.numpy(), g2.numpy(), rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_matvec_adjoint(mode):
    rng = np.random.default_rng(2)
    (dg, sb), (ld, ls) = random_spd(rng, (2,), 4, 3)
    diag, sub = (dg, sb) if mode == 2 else (ld, ls)
    x = torch.tensor(rng.normal(size=(3, 2, 4, 3)), dtype=F64)
    gout = torch.tensor(rng.normal(size=(3, 2, 4, 3)), dtype=F64)
    outs = []
    for fn in (lambda a, b, r: ag.BtdMatvec.apply(matvec_run, a, b, r, mode), lambda a, b, r: matvec_run(0.5 * (a + a.transpose(-1, -2)) if mode == 2 else a, b, r, mode)):
        a, b, r = diag.clone().requires_grad_(True), sub.clone().requires_grad_(True), x.clone().requires_grad_(True)
        torch.sum(fn(a, b, r) * gout).backward()
        outs.append((a.grad if mode == 2 else torch.tril(a.grad), b.grad, r.grad))
    for g1, g2 in zip(*outs):
        np.testing.assert_allclose(g1.numpy(), g2.numpy(), rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("batch,n,d,with_sub", [((), 5, 3, True), ((2,), 4, 2, True), ((2,), 3, 2, False)])
def test_inverse_blocks_adjoint(batch, n, d, with_sub):
    rng = np.random.default_rng(3)
    _, (ld, ls) = random_spd(rng, batch, n, d)
    if not with_sub:
        ls = None
    g1 = torch.tensor(rng.normal(size=ld.shape), dtype=F64)
    g2 = None if ls is None else torch.tensor(rng.normal(size=ls.shape), dtype=F64)
    outs = []
    for fn in (lambda a, b: ag.BtdInverseBlocks.apply(inv_run, a, b, True), lambda a, b: inv_run(a, b, True)):
        a = ld.clone().requires_grad_(True)
        b = None if ls is None else ls.clone().requires_grad_(True)
        od, osub = fn(a, b)
        val = torch.sum(od * g1) + (0 if osub is None else torch.sum(osub * g2))
        val.backward()
        outs.append((torch.tril(a.grad), None if b is None else b.grad))
    for x1, x2 in zip(*outs):
        if x1 is not None:
            np.testing.assert_allclose(x1.numpy(), x2.numpy(), rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("batch,n,d", [((), 1, 3), ((2,), 2, 4), ((3,), 9, 3), ((2,), 17, 5), ((1,), 64, 2), ((2,), 33, 6)])
@pytest.mark.parametrize("which", ["both", "diag", "sub"])
def test_scan_forms_of_the_operator_adjoints_against_the_block_loops(batch, n, d, which):
    """``_cholesky_backward_scan`` / ``_inverse_blocks_backward_scan`` (terms local in time + one congruence recursion as a
    Hillis-Steele scan: what a GPU tensor beyond the adjoint kernels takes instead of a Python loop over the blocks) against the
    block-by-block recursions the tests above pin on dense autograd."""
    gen = torch.Generator().manual_seed(7)
    f64 = torch.float64
    ld = torch.tril(0.3 * torch.randn(*batch, n, d, d, dtype=f64, generator=gen), -1) + torch.diag_embed(
        1 + torch.rand(*batch, n, d, dtype=f64, generator=gen))
    ls = 0.3 * torch.randn(*batch, n - 1, d, d, dtype=f64, generator=gen) if n > 1 else None
    g1 = torch.randn(ld.shape, dtype=f64, generator=gen) if which in ("both", "diag") else None
    g2 = torch.randn(ls.shape, dtype=f64, generator=gen) if (ls is not None and which in ("both", "sub")) else None
    a, b = ag._cholesky_backward_torch(ld, ls, g1, g2), ag._cholesky_backward_scan(ld, ls, g1, g2)
    torch.testing.assert_close(b[0], a[0], rtol=1e-11, atol=1e-12)
    if ls is not None:
        torch.testing.assert_close(b[1], a[1], rtol=1e-11, atol=1e-12)
    # the diagonal blocks of the inverse (the forward's output the scan form reuses): block Takahashi
    eye = torch.eye(d, dtype=f64).expand(ld.shape)
    linv = torch.linalg.solve_triangular(torch.tril(ld), eye, upper=False)
    base = linv.transpose(-1, -2) @ linv
    sig = [None] * n
    sig[n - 1] = base[..., n - 1, :, :]
    for k in range(n - 2, -1, -1):
        g = ls[..., k, :, :] @ linv[..., k, :, :]
        sig[k] = base[..., k, :, :] + g.transpose(-1, -2) @ sig[k + 1] @ g
    sigma = torch.stack(sig, dim=-3)
    c, e = ag._inverse_blocks_backward_torch(ld, ls, g1, g2), ag._inverse_blocks_backward_scan(ld, ls, sigma, g1, g2)
    torch.testing.assert_close(e[0], c[0], rtol=1e-11, atol=1e-12)
    if ls is not None:
        torch.testing.assert_close(e[1], c[1], rtol=1e-11, atol=1e-12)
