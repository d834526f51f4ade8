"""config 4's training step (3 x Matern-5/2, 3 outputs, d = 9, B = 512, T = 1000): the torch operators around the HIP kernels, with
shapes and the Python frames that issued them.   python3 scripts/prof_cfg4_ops.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import markovflow_amd as mfa

dev = torch.device("cuda:0"); dt = torch.float64
g = torch.Generator(device=dev); g.manual_seed(0)
B, T, m = 512, 1000, 3
t = torch.cumsum(0.05 + 0.05 * torch.empty(B, T, dtype=dt, device=dev).exponential_(1.0, generator=g), dim=-1)
y = torch.randn(B, T, m, dtype=dt, device=dev, generator=g)
ls = [(0.5 + 1.5 * torch.rand(B, dtype=dt, device=dev, generator=g)).requires_grad_(True) for _ in range(3)]
var = [(0.5 + 1.5 * torch.rand(B, dtype=dt, device=dev, generator=g)).requires_grad_(True) for _ in range(3)]
chol_r = ((0.1 ** 0.5) * torch.eye(m, dtype=dt, device=dev)).requires_grad_(True)
leaves = ls + var + [chol_r]


def step():
    for x in leaves: x.grad = None
    kern = mfa.IndependentMultiOutput([mfa.Matern52(l, v, jitter=1e-9) for l, v in zip(ls, var)], jitter=1e-9)
    ll = mfa.GaussianProcessRegression((t, y), kern, chol_obs_covariance=chol_r).log_likelihood()
    ll.backward()
    return ll


for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=60, max_shapes_column_width=70))
print(prof.key_averages(group_by_stack_n=6).table(sort_by="self_cuda_time_total", row_limit=25, max_name_column_width=50, max_src_column_width=110))
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=25, max_name_column_width=60))
