"""GaussianProcessRegression.log_likelihood() forward + backward with respect to the kernel hyper-parameters and the noise
(the reference's GPR training step, models/gaussian_process_regression.py:150-160 under a GradientTape) at BASELINE config 4's model.
    python3 scripts/bench_gpr_grad.py [--batch 512] [--T 1000] [--sig 5,5,5] [--multi]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import markovflow_amd as mfa

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=512); ap.add_argument("--T", type=int, default=1000); ap.add_argument("--sig", default="5,5,5")
ap.add_argument("--multi", action="store_true"); ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda:0"); dt = torch.float64
g = torch.Generator(device=dev); g.manual_seed(0)
B, T = a.batch, a.T
sig = [int(o) for o in a.sig.split(",")]
m = len(sig) if a.multi else 1
t = torch.cumsum(0.05 + 0.05 * torch.empty(B, T, dtype=dt, device=dev).exponential_(1.0, generator=g), dim=-1)
y = torch.randn(B, T, m, dtype=dt, device=dev, generator=g)
cls = {1: mfa.Matern12, 3: mfa.Matern32, 5: mfa.Matern52}
ls = [(0.5 + 1.5 * torch.rand(B, dtype=dt, device=dev, generator=g)).requires_grad_(True) for _ in sig]
var = [(0.5 + 1.5 * torch.rand(B, dtype=dt, device=dev, generator=g)).requires_grad_(True) for _ in sig]
chol_r = ((0.1 ** 0.5) * torch.eye(m, dtype=dt, device=dev)).requires_grad_(True)
leaves = ls + var + [chol_r]


def step():
    for x in leaves: x.grad = None
    parts = [cls[o](l, v, jitter=1e-9) for o, l, v in zip(sig, ls, var)]
    kern = mfa.IndependentMultiOutput(parts, jitter=1e-9) if a.multi else mfa.Sum(parts, jitter=1e-9)
    ll = mfa.GaussianProcessRegression((t, y), kern, chol_obs_covariance=chol_r).log_likelihood()
    ll.backward()
    return ll


for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.iters): ll = step()
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / a.iters * 1e3
print(f"B={B} T={T} sig=({a.sig}) multi={a.multi} d={sum((o + 1) // 2 for o in sig)}: log_likelihood forward + backward w.r.t. hyper-parameters {ms:.2f} ms "
      f"(ll = {float(ll):.6f}, |grad l0| = {float(ls[0].grad.abs().sum()):.4e})")
