"""
Synthetic state-space inputs of the shapes BASELINE.json names, generated on the device.

This is the generator SURVEY.md §8(d) describes: per-series Matérn hyper-parameters, closed-form
``A_k = exp(F Δt_k)`` and ``Q_k = P∞ − A_k P∞ A_kᵀ`` (the closed forms of the reference's
``markovflow/kernels/matern.py:299-356,434-501`` and ``kernels/sde_kernel.py:421-446``, restated -
checked against the reference's scipy-expm test kernels in tests/test_host.py::test_synthetic_closed_forms_match_reference_expm_kernels), every tensor
materialised at full ``[B, T-1, d, d]`` shape the way ``StateSpaceModel`` requires
(``state_space_model.py:111-116``).  It is input plumbing for tests and bench.py, not part of the
timed path.
"""
import math
from typing import Dict, Tuple

import torch

DEFAULT_SEED = 71892305  # the reference's tests/conftest.py:22


def _matern_block(order: int, lam: torch.Tensor, var: torch.Tensor, dt: torch.Tensor):
    """A [S, n, k, k], P∞ [S, k, k] for Matérn-(order/2), order in {1, 3, 5}; lam, var [S]; dt [S, n]."""
    k = (order + 1) // 2
    dev, dty = dt.device, dt.dtype
    f = torch.zeros(lam.shape + (k, k), dtype=dty, device=dev)
    if order == 1:
        f[..., 0, 0] = -lam
        pinf = var[..., None, None].clone()
    elif order == 3:
        f[..., 0, 1] = 1.0
        f[..., 1, 0] = -lam ** 2
        f[..., 1, 1] = -2 * lam
        pinf = torch.zeros_like(f)
        pinf[..., 0, 0] = var
        pinf[..., 1, 1] = var * lam ** 2
    elif order == 5:
        f[..., 0, 1] = 1.0
        f[..., 1, 2] = 1.0
        f[..., 2, 0] = -lam ** 3
        f[..., 2, 1] = -3 * lam ** 2
        f[..., 2, 2] = -3 * lam
        l23 = lam ** 2 / 3.0
        pinf = torch.zeros_like(f)
        pinf[..., 0, 0] = var
        pinf[..., 0, 2] = -var * l23
        pinf[..., 2, 0] = -var * l23
        pinf[..., 1, 1] = var * l23
        pinf[..., 2, 2] = var * lam ** 4
    else:
        raise ValueError(order)
    eye = torch.eye(k, dtype=dty, device=dev)
    nil = f + lam[..., None, None] * eye                       # nilpotent: (F + λI)^k = 0
    dtm = dt[..., None, None]
    a = eye + nil[:, None] * dtm
    if k == 3:
        a = a + (nil @ nil)[:, None] * (0.5 * dtm ** 2)
    a = a * torch.exp(-lam[:, None, None, None] * dtm)
    return a, pinf


def make_ssm(
    batch: int, num_points: int, components: Tuple[int, ...] = (5, 5), *, output_dim: int = 1,
    dtype=torch.float64, device="cuda", seed: int = DEFAULT_SEED, dt_min: float = 0.05, dt_scale: float = 0.05,
    noise_var: float = 0.1, jitter: float = 1e-9, slab: int = 256,
) -> Dict[str, torch.Tensor]:
    """
    A sum (``output_dim == 1``) or an independent multi-output stack (``output_dim == len(components)``)
    of Matérn components; ``components`` lists the orders (1, 3 or 5 for Matérn-1/2, 3/2, 5/2).

    Returns ``mu0 [B,d], cholP0 [B,d,d], A [B,T-1,d,d], b [B,T-1,d], cholQ [B,T-1,d,d], H [B,T,m,d],
    y [B,T,m], cholR [m,m]`` with ``y`` sampled from the model.
    """
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    sizes = [(o + 1) // 2 for o in components]
    d, n, m = sum(sizes), num_points, output_dim
    if m not in (1, len(components)):
        raise ValueError("output_dim must be 1 (sum kernel) or len(components) (independent outputs)")
    f64 = torch.float64
    out = {
        "mu0": torch.zeros(batch, d, dtype=dtype, device=device),
        "cholP0": torch.empty(batch, d, d, dtype=dtype, device=device),
        "A": torch.zeros(batch, n - 1, d, d, dtype=dtype, device=device),
        "b": torch.zeros(batch, n - 1, d, dtype=dtype, device=device),
        "cholQ": torch.zeros(batch, n - 1, d, d, dtype=dtype, device=device),
        "H": torch.zeros(batch, n, m, d, dtype=dtype, device=device),
        "y": torch.empty(batch, n, m, dtype=dtype, device=device),
        "cholR": math.sqrt(noise_var) * torch.eye(m, dtype=dtype, device=device),
    }
    for s0 in range(0, batch, slab):
        s1 = min(batch, s0 + slab)
        ns = s1 - s0
        dt = dt_min + dt_scale * torch.empty(ns, n - 1, dtype=f64, device=device).exponential_(1.0, generator=gen)
        a = torch.zeros(ns, n - 1, d, d, dtype=f64, device=device)
        pinf = torch.zeros(ns, d, d, dtype=f64, device=device)
        off = 0
        for j, (order, k) in enumerate(zip(components, sizes)):
            ell = 0.5 + 1.5 * torch.rand(ns, dtype=f64, device=device, generator=gen)
            var = 0.5 + 1.5 * torch.rand(ns, dtype=f64, device=device, generator=gen)
            lam = math.sqrt(order) / ell
            a_j, p_j = _matern_block(order, lam, var, dt)
            a[:, :, off:off + k, off:off + k] = a_j
            pinf[:, off:off + k, off:off + k] = p_j
            out["H"][s0:s1, :, (j if m > 1 else 0), off] = 1.0
            off += k
        q = pinf[:, None] - a @ pinf[:, None] @ a.transpose(-1, -2)
        q = 0.5 * (q + q.transpose(-1, -2)) + jitter * torch.eye(d, dtype=f64, device=device)
        chol_q = torch.linalg.cholesky(q)
        chol_p = torch.linalg.cholesky(pinf + jitter * torch.eye(d, dtype=f64, device=device))
        # sample the chain and the observations
        x = (chol_p @ torch.randn(ns, d, 1, dtype=f64, device=device, generator=gen))[..., 0]
        eps = torch.randn(ns, n - 1, d, 1, dtype=f64, device=device, generator=gen)
        noise = (chol_q @ eps)[..., 0]
        hs = out["H"][s0:s1].to(f64)
        f_vals = torch.empty(ns, n, m, dtype=f64, device=device)
        f_vals[:, 0] = (hs[:, 0] @ x[..., None])[..., 0]
        for k in range(n - 1):
            x = (a[:, k] @ x[..., None])[..., 0] + noise[:, k]
            f_vals[:, k + 1] = (hs[:, k + 1] @ x[..., None])[..., 0]
        y = f_vals + math.sqrt(noise_var) * torch.randn(ns, n, m, dtype=f64, device=device, generator=gen)
        out["A"][s0:s1] = a.to(dtype)
        out["cholQ"][s0:s1] = chol_q.to(dtype)
        out["cholP0"][s0:s1] = chol_p.to(dtype)
        out["y"][s0:s1] = y.to(dtype)
    return out


def make_dense_ssm(batch: int, num_points: int, state_dim: int, output_dim: int, *, dtype=torch.float32, device="cuda",
                   seed: int = DEFAULT_SEED, noise_std: float = 0.3) -> Dict[str, torch.Tensor]:
    """
    The shape of BASELINE config 5 (spatio-temporal model: ``state_dim`` = 64 latent states behind ``output_dim`` = 32 spatial
    outputs, ``models/spatio_temporal_variational.py:45-85`` of the reference): dense, well-conditioned random transitions
    ``A_k = 0.9 I + N(0, 0.09 / d)``, process factors ``tril(N(0, 0.09 / d)) + 0.5 I``, a dense emission matrix.  Same keys as
    :func:`make_ssm`.
    """
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    d, n, m = state_dim, num_points, output_dim
    kw = dict(dtype=dtype, device=device, generator=gen)
    eye = torch.eye(d, dtype=dtype, device=device)
    sc = 0.3 / d ** 0.5
    return {
        "A": 0.9 * eye + sc * torch.randn(batch, n - 1, d, d, **kw),
        "cholQ": torch.tril(sc * torch.randn(batch, n - 1, d, d, **kw)) + 0.5 * eye,
        "cholP0": torch.tril(0.1 * torch.randn(batch, d, d, **kw)) + eye,
        "mu0": torch.randn(batch, d, **kw),
        "b": 0.1 * torch.randn(batch, n - 1, d, **kw),
        "H": torch.randn(batch, n, m, d, **kw) / d ** 0.5,
        "y": torch.randn(batch, n, m, **kw),
        "cholR": noise_std * torch.eye(m, dtype=dtype, device=device),
    }


def kalman_filter_from(inputs: Dict[str, torch.Tensor]):
    """Build ``KalmanFilter`` from the dict returned by :func:`make_ssm`."""
    from . import EmissionModel, KalmanFilter, StateSpaceModel

    ssm = StateSpaceModel(inputs["mu0"], inputs["cholP0"], inputs["A"], inputs["b"], inputs["cholQ"])
    return KalmanFilter(ssm, EmissionModel(inputs["H"]), inputs["y"], inputs["cholR"])


def loglik_bytes_per_step(d: int, m: int, elem_size: int) -> int:
    """Algorithmic bytes per (series x time step) of log_likelihood: read A, cholQ (dense), b, H, y."""
    return (2 * d * d + d + m * d + m) * elem_size
