"""GaussianProcessRegression.posterior_state_space_model at the headline shape: fused (mf_gpr_matern_posterior_chain) against the
materialised route (kernel tensors -> KalmanFilter.posterior_state_space_model)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import markovflow_amd as mfa
dev = torch.device("cuda:0"); dt = torch.float64
g = torch.Generator(device=dev); g.manual_seed(0)
B, T = 1024, 10000
t = torch.cumsum(0.05 + 0.05 * torch.empty(B, T, dtype=dt, device=dev).exponential_(1.0, generator=g), dim=-1)
y = torch.randn(B, T, 1, dtype=dt, device=dev, generator=g)
parts = [mfa.Matern52(0.5 + 1.5 * torch.rand(B, dtype=dt, device=dev, generator=g), 0.5 + 1.5 * torch.rand(B, dtype=dt, device=dev, generator=g),
                      jitter=1e-9) for _ in range(2)]
gpr = mfa.GaussianProcessRegression((t, y), mfa.Sum(parts, jitter=1e-9), chol_obs_covariance=(0.1 ** 0.5) * torch.eye(1, dtype=dt, device=dev))
def timed(fn):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts)
print(f"GPR posterior_state_space_model B={B} T={T} Sum(M52, M52): fused {timed(gpr.posterior_state_space_model):.2f} ms")
gpr.fused_backward = False
print(f"                                                  materialised {timed(gpr.posterior_state_space_model):.2f} ms")
