"""
Reverse mode through the block-tridiagonal OPERATORS on the GPU: the autograd Functions of
markovflow_amd/_autograd_ops.py over the HIP kernels, against torch's reverse mode through the dense matrices, and the chain the
reference's CVI models differentiate - ``dist_p.precision -> naturals_to_ssm_params -> kl_divergence``
(/root/reference/markovflow/models/variational_cvi.py:105-136,402; the operators' gradients there come from banded_matrices,
block_tri_diag.py:22-31) - against central differences of the same (forward-only, oracle-checked) kernels.  fp64.
"""
import numpy as np
import pytest
import torch

import markovflow_amd as mfa
from markovflow_amd import ssm_gaussian_transformations as G
from test_autograd_ops import blocks_of, dense_of, random_spd
from test_gpu_kalman import DEV, nn, random_ssm, tt

pytestmark = pytest.mark.gpu
F64 = torch.float64


def dev(x):
    return None if x is None else x.to(DEV)


@pytest.mark.parametrize("batch,n,d", [((), 7, 3), ((2,), 70, 6), ((3,), 5, 2), ((), 40, 9)])
def test_cholesky_solve_inverse_blocks_and_products_vs_dense_autograd(batch, n, d):
    rng = np.random.default_rng(5)
    (diag, sub), _ = random_spd(rng, batch, n, d)
    rhs = torch.tensor(rng.normal(size=(2,) + batch + (n, d)), dtype=F64)
    w1, w2, w3, w4 = (torch.tensor(rng.normal(size=s), dtype=F64, device=DEV) for s in
                      (batch + (n, d, d), batch + (n - 1, d, d), (2,) + batch + (n, d), (2,) + batch + (n, d)))

    def loss(dg, sb, r, native):
        if native:
            sym = mfa.SymmetricBlockTriDiagonal(dg, sb)
            chol = sym.cholesky
            inv_d, inv_s = chol._diag_and_sub_of_inverse(want_sub=True)
            x, xt = chol.solve(r), chol.solve(r, transpose_left=True)
            prod = sym.dense_mult(r) + chol.dense_mult(r) + chol.dense_mult(r, transpose_left=True)
            logdet = chol.abs_log_det()
        else:
            full = dense_of(0.5 * (dg + dg.transpose(-1, -2)), sb, True)
            cf = torch.linalg.cholesky(full)
            inv_d, inv_s = blocks_of(torch.linalg.inv(full), n, d)
            flat = r.reshape(r.shape[:-2] + (n * d, 1))
            x = torch.linalg.solve_triangular(cf, flat, upper=False).reshape(r.shape)
            xt = torch.linalg.solve_triangular(cf.transpose(-1, -2), flat, upper=True).reshape(r.shape)
            prod = ((full + cf + cf.transpose(-1, -2)) @ flat).reshape(r.shape)
            logdet = torch.sum(torch.log(torch.diagonal(cf, dim1=-2, dim2=-1)), dim=-1)
        return (torch.sum(inv_d * w1) + torch.sum(inv_s * w2) + torch.sum(x * w3) + torch.sum(xt * w4) + torch.sum(prod * w3)
                + torch.sum(logdet))

    grads = []
    for native in (True, False):
        dg, sb, r = (dev(t).clone().requires_grad_(True) for t in (diag, sub, rhs))
        val = loss(dg, sb, r, native)
        val.backward()
        grads.append((float(val.detach()), dg.grad, sb.grad, r.grad))
    assert grads[0][0] == pytest.approx(grads[1][0], rel=1e-9)
    for g1, g2 in zip(grads[0][1:], grads[1][1:]):
        scale = float(g2.abs().max())
        assert float((g1 - g2).abs().max()) <= 1e-8 * scale


def test_precision_is_differentiable(rng):
    kw = random_ssm(rng, (2,), 12, 3, 1, well=True)
    leaves = [tt(kw[k]).requires_grad_(True) for k in ("mu0", "chol_p0", "a_s", "b_s", "chol_q")]
    ssm = mfa.StateSpaceModel(*leaves)
    prec = ssm.precision
    plain = mfa.StateSpaceModel(*(x.detach() for x in leaves)).precision
    np.testing.assert_allclose(nn(prec.block_diagonal), nn(plain.block_diagonal), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(nn(prec.block_sub_diagonal), nn(plain.block_sub_diagonal), rtol=1e-10, atol=1e-12)
    (prec.block_diagonal.sum() + 2 * prec.block_sub_diagonal.sum()).backward()
    assert all(x.grad is not None and bool(torch.isfinite(x.grad).all()) for x in (leaves[1], leaves[2], leaves[4]))


def _cvi_kl(lengthscale, variance, t_pts, nat1, nat2):
    """KL(q || p) with p the Matern-3/2 prior and q = prior x Gaussian sites, built the way variational_cvi.py:105-136 builds it."""
    kern = mfa.Matern32(lengthscale, variance, jitter=1e-9)
    dist_p = kern.state_space_model(t_pts)
    prec = dist_p.precision
    h = kern.generate_emission_model(t_pts).emission_matrix                        # [.., T, 1, d]
    theta_lin = (h.transpose(-1, -2) @ nat1[..., None])[..., 0]
    theta_diag = -0.5 * prec.block_diagonal + h.transpose(-1, -2) @ nat2 @ h
    theta_sub = -prec.block_sub_diagonal
    a_s, offsets, chol_p0, chol_q, mu0 = G.naturals_to_ssm_params(theta_lin, theta_diag, theta_sub)
    dist_q = mfa.StateSpaceModel(mu0, chol_p0, a_s, offsets, chol_q)
    return torch.sum(dist_q.kl_divergence(dist_p))


def test_cvi_shaped_gradient_through_precision_and_naturals(rng):
    """d KL / d lengthscale, d KL / d variance and d KL / d sites through the whole chain against central differences."""
    bsz, n = 2, 60
    t_pts = torch.cumsum(0.05 + 0.1 * torch.rand(bsz, n, dtype=F64, device=DEV), dim=-1)
    nat1 = tt(rng.normal(size=(bsz, n, 1)))
    nat2 = tt(-0.5 * rng.uniform(0.5, 2.0, size=(bsz, n, 1, 1)))
    ls0 = tt(np.array([0.7, 1.3]))
    var0 = tt(np.array([1.2, 0.8]))
    ls, var = ls0.clone().requires_grad_(True), var0.clone().requires_grad_(True)
    n1, n2 = nat1.clone().requires_grad_(True), nat2.clone().requires_grad_(True)
    kl = _cvi_kl(ls, var, t_pts, n1, n2)
    kl.backward()
    with torch.no_grad():
        for which, base, grad in (("lengthscale", ls0, ls.grad), ("variance", var0, var.grad)):
            for i in range(bsz):
                hstep = 1e-6
                up, dn = base.clone(), base.clone()
                up[i] += hstep; dn[i] -= hstep
                args_up = (up, var0) if which == "lengthscale" else (ls0, up)
                args_dn = (dn, var0) if which == "lengthscale" else (ls0, dn)
                fd = (float(_cvi_kl(*args_up, t_pts, nat1, nat2)) - float(_cvi_kl(*args_dn, t_pts, nat1, nat2))) / (2 * hstep)
                assert float(grad[i]) == pytest.approx(fd, rel=2e-5, abs=1e-7), which
        d1 = tt(rng.normal(size=nat1.shape))
        hstep = 1e-6
        fd = (float(_cvi_kl(ls0, var0, t_pts, nat1 + hstep * d1, nat2)) - float(_cvi_kl(ls0, var0, t_pts, nat1 - hstep * d1, nat2))) / (2 * hstep)
        assert float(torch.sum(n1.grad * d1)) == pytest.approx(fd, rel=2e-5, abs=1e-7)
        assert bool(torch.isfinite(n2.grad).all())


# ---- the HIP adjoints of cholesky / block_diagonal_of_inverse (mf_btd_cholesky_grad_*, mf_btd_diag_of_inverse_grad_*) -------------
def _random_factor(rng, batch, n, d, dtype=F64):
    ld = np.tril(0.3 * rng.normal(size=batch + (n, d, d)), k=-1) + (1.0 + np.abs(0.3 * rng.normal(size=batch + (n, d))))[..., None] * np.eye(d)
    ls = 0.3 * rng.normal(size=batch + (n - 1, d, d))
    return torch.tensor(ld, dtype=dtype, device=DEV), torch.tensor(ls, dtype=dtype, device=DEV)


@pytest.mark.parametrize("batch,n,d,with_sub,which", [
    ((3,), 300, 6, True, "both"), ((2,), 1000, 4, True, "both"), ((4100,), 12, 4, True, "both"), ((2, 3), 40, 2, True, "both"),
    ((5,), 64, 9, True, "both"), ((3,), 50, 7, True, "both"), ((2,), 30, 5, False, "both"), ((3,), 1, 3, False, "both"),
    ((2,), 2, 6, True, "both"), ((2,), 200, 6, True, "diag"), ((2,), 200, 6, True, "sub"), ((1,), 3000, 1, True, "both"),
    # 10 <= d <= 32 (round 6, csrc/mf_adj.hip, register MFMA tiles): chains shorter than 32 blocks - one wavefront per series walks
    # the recurrence - and longer ones - local terms + congruence scans, parallel in time (ragged chunk partitions, one of the two
    # incoming gradients missing) - incl. the reference's largest tested operator shape, d = 30, T = 1001
    # (tests/unit/test_ssm_gaussian_transformations.py:40-46)
    ((3,), 40, 12, True, "both"), ((2,), 25, 16, True, "both"), ((2,), 30, 24, True, "diag"), ((2,), 30, 24, True, "sub"),
    ((1,), 12, 32, True, "both"), ((2,), 9, 17, False, "both"), ((3,), 1, 20, False, "both"), ((1,), 1001, 30, True, "both"),
    ((2,), 77, 16, True, "both"), ((3,), 100, 20, True, "diag"), ((2,), 64, 32, True, "sub"), ((5,), 33, 10, True, "both"),
    ((2,), 130, 24, True, "both"), ((70,), 45, 15, True, "both"),
])
def test_hip_operator_adjoints_against_the_torch_recursions(batch, n, d, with_sub, which):
    """The kernels against the block-by-block torch recursions (which tests/test_autograd_ops.py pins on dense autograd): few long
    series (recursion parallel in time), many short ones (a lane per series), no coupling, a single block, only one of the two
    output gradients given."""
    from markovflow_amd import _autograd_ops as ag
    rng = np.random.default_rng(11)
    ldiag, lsub = _random_factor(rng, batch, n, d)
    if d >= 10:
        # (G_k = W_k L_k^-1 has to stay a contraction for the recurrences to be evaluable over a thousand blocks at all - in any
        # implementation: scale the couplings and the strict lower triangles with the state dimension)
        lsub = lsub * (2.0 / d)
        ldiag = torch.tril(ldiag, -1) * (4.0 / d) + torch.diag_embed(torch.diagonal(ldiag, dim1=-2, dim2=-1))
    if not with_sub or n == 1:
        lsub = None
    g1 = torch.tensor(rng.normal(size=tuple(ldiag.shape)), dtype=F64, device=DEV) if which in ("both", "diag") else None
    g2 = (torch.tensor(rng.normal(size=tuple(lsub.shape)), dtype=F64, device=DEV)
          if (lsub is not None and which in ("both", "sub")) else None)
    assert ag._hip_grad_ws(ldiag) is not None
    # cholesky: gradients w.r.t. the factor's blocks -> gradients w.r.t. the matrix' blocks

    class Ctx:
        pass

    ctx = Ctx()
    ctx.has_sub = lsub is not None
    ctx.saved_tensors = (ldiag, lsub if lsub is not None else ldiag.new_zeros(0))
    _, gd, gs = ag.BtdCholesky.backward(ctx, g1, g2)
    want_d, want_s = ag._cholesky_backward_torch(ldiag, lsub, g1, g2)
    scale = float(want_d.abs().max())
    assert float((gd - want_d).abs().max()) <= 1e-9 * scale
    if lsub is not None:
        assert float((gs - want_s).abs().max()) <= 1e-9 * max(float(want_s.abs().max()), scale)
    # block_diagonal_of_inverse: gradients w.r.t. the blocks of the inverse -> gradients w.r.t. the factor's blocks
    chol = mfa.LowerTriangularBlockTriDiagonal(ldiag, lsub)
    odiag, _ = chol._diag_and_sub_of_inverse(want_sub=lsub is not None)
    ctx = Ctx()
    ctx.has_sub, ctx.want_sub = lsub is not None, lsub is not None
    ctx.saved_tensors = (ldiag, lsub if lsub is not None else ldiag.new_zeros(0), odiag.contiguous())
    _, gl, gw, _ = ag.BtdInverseBlocks.backward(ctx, g1, g2)
    want_l, want_w = ag._inverse_blocks_backward_torch(ldiag, lsub, g1, g2)
    if want_l is None:
        want_l = torch.zeros_like(ldiag)
    scale = max(float(want_l.abs().max()), 1e-30)
    assert float((gl - want_l).abs().max()) <= 1e-9 * scale
    if lsub is not None:
        if want_w is None:
            want_w = torch.zeros_like(lsub)
        assert float((gw - want_w).abs().max()) <= 1e-9 * max(float(want_w.abs().max()), scale)


def test_hip_operator_adjoints_fp32():
    from markovflow_amd import _autograd_ops as ag
    rng = np.random.default_rng(12)
    ldiag, lsub = _random_factor(rng, (3,), 120, 5, dtype=torch.float32)
    g1 = torch.tensor(rng.normal(size=tuple(ldiag.shape)), dtype=torch.float32, device=DEV)
    g2 = torch.tensor(rng.normal(size=tuple(lsub.shape)), dtype=torch.float32, device=DEV)

    class Ctx:
        pass

    ctx = Ctx()
    ctx.has_sub = True
    ctx.saved_tensors = (ldiag, lsub)
    _, gd, gs = ag.BtdCholesky.backward(ctx, g1, g2)
    want_d, want_s = ag._cholesky_backward_torch(ldiag.double(), lsub.double(), g1.double(), g2.double())
    assert float((gd.double() - want_d).abs().max()) <= 2e-4 * float(want_d.abs().max())
    assert float((gs.double() - want_s).abs().max()) <= 2e-4 * float(want_s.abs().max())


@pytest.mark.parametrize("d,n,bsz", [(40, 50, 2), (64, 130, 1), (33, 9, 3)])
def test_operator_adjoints_beyond_the_kernels_take_the_scan_forms(d, n, bsz):
    """d > 32 (fp32 operators on the panel / tile engine): no adjoint kernel; a GPU tensor takes the scan forms of
    markovflow_amd/_autograd_ops.py (local terms + a Hillis-Steele congruence scan, no Python loop over the blocks) - checked
    against the block loops in float32 arithmetic on the same device."""
    from markovflow_amd import _autograd_ops as ag
    gen = torch.Generator(device=DEV).manual_seed(5)
    f32 = torch.float32
    ld = torch.tril((4.0 / d) * 0.3 * torch.randn(bsz, n, d, d, dtype=f32, device=DEV, generator=gen), -1) + torch.diag_embed(
        1 + torch.rand(bsz, n, d, dtype=f32, device=DEV, generator=gen))
    ls = (2.0 / d) * 0.3 * torch.randn(bsz, n - 1, d, d, dtype=f32, device=DEV, generator=gen)
    assert ag._hip_grad_ws(ld) is None and ag._scan_pays(ld)
    dg = ld @ ld.transpose(-1, -2)
    dg[:, 1:] += ls @ ls.transpose(-1, -2)
    sb = ls @ ld[:, :-1].transpose(-1, -2)
    dg.requires_grad_(True); sb.requires_grad_(True)
    w = torch.randn(bsz, n, d, d, dtype=f32, device=DEV, generator=gen)
    inv_d, inv_s = mfa.SymmetricBlockTriDiagonal(dg, sb).cholesky._diag_and_sub_of_inverse(want_sub=True)
    (torch.sum(inv_d * w) + torch.sum(inv_s)).backward()
    got_d, got_s = dg.grad.clone(), sb.grad.clone()
    # the same chain with the block loops
    chol = mfa.SymmetricBlockTriDiagonal(dg.detach(), sb.detach()).cholesky
    cd, cs = chol.block_diagonal, chol.block_sub_diagonal
    gl, gw = ag._inverse_blocks_backward_torch(cd, cs, w, torch.ones_like(sb))
    want_d, want_s = ag._cholesky_backward_torch(cd, cs, gl, gw)
    scale = float(want_d.abs().max())
    assert float((got_d - want_d).abs().max()) <= 2e-3 * scale
    assert float((got_s - want_s).abs().max()) <= 2e-3 * max(float(want_s.abs().max()), scale)

