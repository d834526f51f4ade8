"""
GPU parity tests of the fused backward of GaussianProcessRegression.log_likelihood (csrc/mf_gpr_grad.hpp:
`mf_gpr_matern_loglik_grad_*` + `mf_sde_matern_transitions_grad_*`) - the training step of
/root/reference/markovflow/models/gaussian_process_regression.py:150-160 under a GradientTape (the reference checks it in
tests/integration/models/test_gaussian_process_regression.py:117-130).  Gradients with respect to lengthscales, variances and the
noise against torch autograd through the DENSE GP marginal likelihood (rtol 1e-6), and against the materialised route
(kernel tensors -> KalmanFilter -> streamed backward -> generator backward) at sizes beyond it.
"""
import numpy as np
import pytest
import torch

import markovflow_amd as mfa
from markovflow_amd import _lib
from test_gpu_kalman import DEV

pytestmark = pytest.mark.gpu
ORD = {1: mfa.Matern12, 3: mfa.Matern32, 5: mfa.Matern52}


def _dense_kernel(order, l, v, r):
    lam = np.sqrt(float(order)) / l
    if order == 1:
        return v * torch.exp(-lam * r)
    if order == 3:
        return v * (1 + lam * r) * torch.exp(-lam * r)
    return v * (1 + lam * r + (lam * r) ** 2 / 3) * torch.exp(-lam * r)


def _run(orders, t, y, vals, per_series, fused_backward, chunks=0, dtype=torch.float64, jitter=None):
    names = list(vals)
    leaves = {k: torch.tensor(v, dtype=dtype, device=DEV, requires_grad=True) for k, v in vals.items()}
    kw = {} if jitter is None else {"jitter": jitter}
    comps = [ORD[o](leaves[f"l{i}"], leaves[f"v{i}"], **kw) for i, o in enumerate(orders)]
    kern = mfa.Sum(comps, **kw) if len(comps) > 1 else comps[0]
    gpr = mfa.GaussianProcessRegression((torch.tensor(t, dtype=dtype, device=DEV), torch.tensor(y[..., None], dtype=dtype, device=DEV)),
                                        kern, chol_obs_covariance=leaves["s"].reshape(1, 1))
    gpr.fused_backward = fused_backward
    gpr._chunks = chunks
    ll = gpr.log_likelihood()
    ll.backward()
    return float(ll.detach()), {k: leaves[k].grad.detach().cpu().numpy() for k in names}


@pytest.mark.parametrize("orders,n,bsz,per_series,chunks", [
    ((5, 5), 90, 2, False, 0), ((5, 5), 101, 3, True, 7), ((5,), 80, 2, False, 0), ((3, 3), 120, 2, True, 0),
    ((5, 3), 70, 2, False, 5), ((3, 5), 66, 2, True, 3), ((3,), 150, 2, False, 40), ((1,), 100, 3, True, 0),
])
def test_fused_backward_vs_dense_gp(rng, monkeypatch, orders, n, bsz, per_series, chunks):
    t = np.cumsum(0.1 + rng.exponential(0.2, size=(bsz, n)), axis=-1)
    y = rng.normal(size=(bsz, n))
    shape = (bsz,) if per_series else ()
    vals = {"s": np.array(0.3)}
    for i in range(len(orders)):
        vals[f"l{i}"] = rng.uniform(0.6, 1.6, size=shape)
        vals[f"v{i}"] = rng.uniform(0.5, 1.5, size=shape)
    cpu = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in vals.items()}
    tt_, yt = torch.tensor(t), torch.tensor(y)
    total = 0.0
    for s in range(bsz):
        r = (tt_[s][:, None] - tt_[s][None, :]).abs()
        pick = (lambda x: x[s]) if per_series else (lambda x: x)
        kmat = sum(_dense_kernel(o, pick(cpu[f"l{i}"]), pick(cpu[f"v{i}"]), r) for i, o in enumerate(orders))
        kn = kmat + cpu["s"] ** 2 * torch.eye(n, dtype=torch.float64)
        total = total - 0.5 * (yt[s] @ torch.linalg.solve(kn, yt[s]) + torch.linalg.slogdet(kn)[1] + n * np.log(2 * np.pi))
    total.backward()
    seen = []
    real = _lib.call_rc

    def spy(name, *args):
        seen.append(name)
        return real(name, *args)

    monkeypatch.setattr(_lib, "call_rc", spy)
    ll, grads = _run(orders, t, y, vals, per_series, True, chunks)
    assert "mf_gpr_matern_loglik_grad" in seen, "the fused backward should have run"
    assert ll == pytest.approx(float(total.detach()), rel=1e-9)
    for k in vals:
        np.testing.assert_allclose(grads[k], cpu[k].grad.numpy(), rtol=1e-6, atol=1e-9, err_msg=k)


def test_fused_backward_agrees_with_the_materialised_route_on_long_chains(rng):
    bsz, n = 6, 1500
    t = np.cumsum(0.05 + rng.exponential(0.05, size=(bsz, n)), axis=-1)
    y = rng.normal(size=(bsz, n))
    vals = {"s": np.array(0.4), "l0": rng.uniform(0.6, 1.6, size=(bsz,)), "v0": rng.uniform(0.5, 1.5, size=(bsz,)),
            "l1": rng.uniform(0.6, 1.6, size=(bsz,)), "v1": rng.uniform(0.5, 1.5, size=(bsz,))}
    ll_f, g_f = _run((5, 5), t, y, vals, True, True)
    ll_m, g_m = _run((5, 5), t, y, vals, True, False)
    assert ll_f == pytest.approx(ll_m, rel=1e-11)
    for k in vals:
        np.testing.assert_allclose(g_f[k], g_m[k], rtol=1e-7, atol=1e-9 * (1 + np.abs(g_m[k]).max()), err_msg=k)


def test_fused_backward_twice_through_the_same_graph(rng, monkeypatch):
    """a second backward through the same node (``retain_graph=True``, per-parameter gradient loops) must work and
    give the same gradients - the forward's chunk summaries are read, not consumed."""
    bsz, n = 3, 200
    t = np.cumsum(0.1 + rng.exponential(0.2, size=(bsz, n)), axis=-1)
    y = rng.normal(size=(bsz, n))
    leaves = {k: torch.tensor(v, dtype=torch.float64, device=DEV, requires_grad=True)
              for k, v in {"s": 0.3, "l0": 0.9, "v0": 1.2, "l1": 1.4, "v1": 0.7}.items()}
    kern = mfa.Sum([mfa.Matern52(leaves["l0"], leaves["v0"]), mfa.Matern52(leaves["l1"], leaves["v1"])])
    gpr = mfa.GaussianProcessRegression((torch.tensor(t, device=DEV), torch.tensor(y[..., None], device=DEV)), kern,
                                        chol_obs_covariance=leaves["s"].reshape(1, 1))
    seen = []
    real = _lib.call_rc

    def spy(name, *args):
        seen.append(name)
        return real(name, *args)

    monkeypatch.setattr(_lib, "call_rc", spy)
    ll = gpr.log_likelihood()
    names = list(leaves)
    first = torch.autograd.grad(ll, [leaves[k] for k in names], retain_graph=True)
    per_parameter = [torch.autograd.grad(ll, leaves[k], retain_graph=True)[0] for k in names]
    assert seen.count("mf_gpr_matern_loglik_grad") == 1 + len(names)
    for k, a, b in zip(names, first, per_parameter):
        assert torch.equal(a, b), k


@pytest.mark.parametrize("orders", [(5, 5), (3,), (5, 3)])
def test_fused_backward_fp32(rng, orders):
    """fp32 instantiations: against the fp64 run of the same fused route (well-separated time points, jitter fp32 can carry)."""
    bsz, n = 3, 200
    t = np.cumsum(0.3 + rng.exponential(0.3, size=(bsz, n)), axis=-1).astype(np.float32).astype(np.float64)
    y = rng.normal(size=(bsz, n)).astype(np.float32).astype(np.float64)
    vals = {"s": np.array(0.5)}
    for i in range(len(orders)):
        vals[f"l{i}"] = rng.uniform(0.8, 1.6, size=(bsz,)).astype(np.float32).astype(np.float64)
        vals[f"v{i}"] = rng.uniform(0.5, 1.5, size=(bsz,)).astype(np.float32).astype(np.float64)
    ll64, g64 = _run(orders, t, y, vals, True, True, jitter=1e-4)
    ll32, g32 = _run(orders, t, y, vals, True, True, dtype=torch.float32, jitter=1e-4)
    assert ll32 == pytest.approx(ll64, rel=2e-4)
    for k in vals:
        np.testing.assert_allclose(g32[k], g64[k], rtol=2e-2, atol=2e-2 * (1 + np.abs(g64[k]).max()), err_msg=k)


# ---- posterior_state_space_model of the GPR model with the kernel -> state space model step fused ---------------------------------
@pytest.mark.parametrize("orders,n,bsz,per_series,chunks", [
    ((5, 5), 300, 3, True, 0), ((5, 5), 101, 2, False, 7), ((5,), 90, 2, True, 0), ((3, 3), 200, 3, False, 0),
    ((5, 3), 150, 2, True, 5), ((3, 5), 66, 2, True, 3), ((3,), 400, 1, False, 0), ((1,), 100, 3, True, 4),
])
def test_fused_posterior_chain_agrees_with_the_materialised_route(rng, monkeypatch, orders, n, bsz, per_series, chunks):
    t = np.cumsum(0.1 + rng.exponential(0.2, size=(bsz, n)), axis=-1)
    y = rng.normal(size=(bsz, n))
    shape = (bsz,) if per_series else ()

    def build():
        comps = [ORD[o](torch.tensor(rng2.uniform(0.6, 1.6, size=shape), device=DEV), torch.tensor(rng2.uniform(0.5, 1.5, size=shape), device=DEV))
                 for o in orders]
        kern = mfa.Sum(comps) if len(comps) > 1 else comps[0]
        return mfa.GaussianProcessRegression((torch.tensor(t, device=DEV), torch.tensor(y[..., None], device=DEV)), kern,
                                             chol_obs_covariance=torch.tensor([[0.4]], dtype=torch.float64, device=DEV))

    seen = []
    real = _lib.call_rc

    def spy(name, *args):
        seen.append(name)
        return real(name, *args)

    monkeypatch.setattr(_lib, "call_rc", spy)
    rng2 = np.random.default_rng(5)
    gpr = build()
    gpr._chunks = chunks
    fused = gpr.posterior_state_space_model()
    assert "mf_gpr_matern_posterior_chain" in seen
    rng2 = np.random.default_rng(5)
    ref_model = build()
    ref_model.fused_backward = False
    ref = ref_model.posterior_state_space_model()
    for name in ("initial_mean", "cholesky_initial_covariance", "state_transitions", "state_offsets", "cholesky_process_covariances"):
        g, w = getattr(fused, name).cpu().numpy(), getattr(ref, name).cpu().numpy()
        np.testing.assert_allclose(g, w, rtol=1e-8, atol=1e-10 * (1 + np.abs(w).max()), err_msg=name)
