"""Posterior prediction at new time points: GPR posterior, predict_f."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import markovflow_amd as mfa
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
for bsz, n, npred in [(64, 10000, 10000), (1024, 2000, 2000), (1, 100000, 50000)]:
    t = torch.cumsum(0.05 + 0.05 * torch.empty(bsz, n, dtype=torch.float64, device=dev).exponential_(1.0, generator=g), dim=-1)
    y = torch.randn(bsz, n, 1, dtype=torch.float64, device=dev, generator=g)
    kern = mfa.Sum([mfa.Matern52(0.7, 1.3, device=dev), mfa.Matern32(1.1, 0.5, device=dev)], jitter=1e-9)
    gpr = mfa.GaussianProcessRegression((t, y), kern, chol_obs_covariance=(0.1 ** 0.5) * torch.eye(1, dtype=torch.float64, device=dev))
    tn = torch.sort(t[:, 0:1] + (t[:, -1:] - t[:, 0:1]) * torch.rand(bsz, npred, dtype=torch.float64, device=dev, generator=g), dim=-1)[0]
    def run():
        return gpr.posterior.predict_f(tn)
    for _ in range(2): run()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): out = run()
    e1.record(); torch.cuda.synchronize()
    print(f"B={bsz} N={n} new={npred} d=5: posterior + predict_f {e0.elapsed_time(e1) / 3:.2f} ms  mean[0,0]={float(out[0][0, 0, 0]):.6f}")
