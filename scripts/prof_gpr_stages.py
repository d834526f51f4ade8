"""Where the materialised GPR route spends its time (host glue vs kernels): wall-clock per stage, synchronised.
    python3 scripts/prof_gpr_stages.py [--batch 512] [--T 1000] [--sig 5,5,5]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import markovflow_amd as mfa

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=512); ap.add_argument("--T", type=int, default=1000); ap.add_argument("--sig", default="5,5,5")
ap.add_argument("--multi", action="store_true", help="IndependentMultiOutput (one output per component) instead of Sum")
a = ap.parse_args()
dev = torch.device("cuda:0"); dt = torch.float64
g = torch.Generator(device=dev); g.manual_seed(0)
B, T = a.batch, a.T
t = torch.cumsum(0.05 + 0.05 * torch.empty(B, T, dtype=dt, device=dev).exponential_(1.0, generator=g), dim=-1)
m = len(a.sig.split(",")) if a.multi else 1
y = torch.randn(B, T, m, dtype=dt, device=dev, generator=g)
cls = {1: mfa.Matern12, 3: mfa.Matern32, 5: mfa.Matern52}
parts = [cls[int(o)](0.5 + 1.5 * torch.rand(B, dtype=dt, device=dev, generator=g), 0.5 + 1.5 * torch.rand(B, dtype=dt, device=dev, generator=g))
         for o in a.sig.split(",")]
kern = mfa.IndependentMultiOutput(parts, jitter=1e-9) if a.multi else mfa.Sum(parts, jitter=1e-9)
chol_r = (0.1 ** 0.5) * torch.eye(m, dtype=dt, device=dev)


def wall(fn, it=20):
    for _ in range(3): r = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3, r


ms_ssm, ssm = wall(lambda: kern.state_space_model(t))
ms_em, em = wall(lambda: kern.generate_emission_model(t))
ms_kf, kf = wall(lambda: mfa.KalmanFilter(ssm, em, y, chol_r))
ms_ll, _ = wall(lambda: kf.log_likelihood())
def gpr_ll(fused):
    gp = mfa.GaussianProcessRegression((t, y), kern, chol_obs_covariance=chol_r); gp.fused = fused
    return gp.log_likelihood()
ms_all, ll_m = wall(lambda: gpr_ll(False))
ms_fused, ll_f = wall(lambda: gpr_ll(True))
print(f"fused route (kernel -> SSM inside the sweep) {ms_fused:.3f} ms, rel diff vs materialised {abs(float(ll_f) - float(ll_m)) / abs(float(ll_m)):.1e}")
print(f"B={B} T={T} sig=({a.sig}): state_space_model {ms_ssm:.3f} ms, emission model {ms_em:.3f} ms, KalmanFilter() {ms_kf:.3f} ms, "
      f"log_likelihood {ms_ll:.3f} ms; GaussianProcessRegression(...).log_likelihood() materialised {ms_all:.3f} ms")
