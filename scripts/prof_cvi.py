"""The chain the reference's CVI models differentiate - dist_p.precision -> naturals_to_ssm_params -> kl_divergence
(models/variational_cvi.py:105-136) - at B=64, T=10^4, Sum(M52, M52) (d = 6), fp64: forward and backward times; under
`rocprofv3 --kernel-trace --stats` the kernel-by-kernel breakdown (scripts/r05_*.sh).  Usage: python3 scripts/prof_cvi.py [B] [T]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import markovflow_amd as mfa  # noqa: E402
from markovflow_amd import ssm_gaussian_transformations as G  # noqa: E402

dev = torch.device("cuda:0")
bsz = int(sys.argv[1]) if len(sys.argv) > 1 else 64
tn = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
g = torch.Generator(device=dev)
g.manual_seed(3)
f64 = torch.float64
t_pts = torch.cumsum(0.05 + 0.05 * torch.empty(bsz, tn, dtype=f64, device=dev).exponential_(1.0, generator=g), dim=-1)
nat1 = torch.randn(bsz, tn, 1, dtype=f64, device=dev, generator=g).requires_grad_(True)
nat2 = (-0.5 * (0.5 + torch.rand(bsz, tn, 1, 1, dtype=f64, device=dev, generator=g))).requires_grad_(True)
ls_c = [(0.5 + 1.5 * torch.rand(bsz, dtype=f64, device=dev, generator=g)).requires_grad_(True) for _ in range(2)]
var_c = [(0.5 + 1.5 * torch.rand(bsz, dtype=f64, device=dev, generator=g)).requires_grad_(True) for _ in range(2)]


def stamp():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def cvi_kl(stamps=None):
    rec = (lambda n: stamps.append((n, stamp()))) if stamps is not None else (lambda n: None)
    rec("start")
    kern = mfa.Sum([mfa.Matern52(l, v, jitter=1e-9) for l, v in zip(ls_c, var_c)], jitter=1e-9)
    dist_p = kern.state_space_model(t_pts)
    rec("kernel -> ssm")
    prec = dist_p.precision
    rec("precision")
    h = kern.generate_emission_model(t_pts).emission_matrix
    theta_lin = (h.transpose(-1, -2) @ nat1[..., None])[..., 0]
    theta_diag = -0.5 * prec.block_diagonal + h.transpose(-1, -2) @ nat2 @ h
    rec("naturals")
    a_s, offsets, chol_p0, chol_q, mu0 = G.naturals_to_ssm_params(theta_lin, theta_diag, -prec.block_sub_diagonal)
    rec("naturals_to_ssm_params")
    kl = torch.sum(mfa.StateSpaceModel(mu0, chol_p0, a_s, offsets, chol_q).kl_divergence(dist_p))
    rec("kl_divergence")
    return kl


for i in range(4):
    for x in ls_c + var_c + [nat1, nat2]:
        x.grad = None
    stamps = []
    kl = cvi_kl(stamps)
    e1 = stamp()
    kl.backward()
    e2 = stamp()
    torch.cuda.synchronize()
    if i == 3:
        for (n0, s0), (n1, s1) in zip(stamps[:-1], stamps[1:]):
            print(f"  forward  {n1:26s} {s0.elapsed_time(s1):8.3f} ms")
        print(f"  forward total {stamps[0][1].elapsed_time(e1):8.3f} ms   backward {e1.elapsed_time(e2):8.3f} ms   KL {float(kl):.6f}")
