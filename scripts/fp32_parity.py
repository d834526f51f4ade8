"""Measured fp32 deviations behind the tolerances of tests/test_gpu_baseline_configs.py (config 5 full shape, fp32 headline
shape): python3 scripts/fp32_parity.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from markovflow_amd import synthetic  # noqa: E402
from oracle import c_oracle as C  # noqa: E402

dev = "cuda:0"


def host(inp, n=None):
    return {k: (v[:n] if n and v.shape[0] > n else v).detach().double().cpu().numpy() for k, v in inp.items()}


def dev_vs_oracle(inp, t, n=None):
    kf = synthetic.kalman_filter_from(inp)
    per = (kf._log_likelihood_per_series() + kf._constant_terms(t)).double().cpu().numpy()
    h = host(inp, n)
    r_inv = np.linalg.inv(h["cholR"] @ h["cholR"].T)
    ref = C.kf_loglik(h["mu0"], h["cholP0"], h["A"], h["b"], h["cholQ"], h["H"], h["y"], r_inv)
    return np.abs(per[: len(ref)] - ref) / np.abs(ref)


e = dev_vs_oracle(synthetic.make_dense_ssm(8, 2048, 64, 32, dtype=torch.float32, device=dev), 2048)
print(f"config 5 full shape (d=64 T=2048 m=32 B=8 fp32): per-series relative deviation max {e.max():.2e} median {np.median(e):.2e}")
e = dev_vs_oracle(synthetic.make_ssm(1024, 10000, (3, 3, 3), dtype=torch.float32, device=dev, dt_min=0.2, dt_scale=0.3, jitter=1e-6), 10000, 256)
print(f"headline shape fp32 (3 x Matern-3/2, B=1024 T=10000 d=6): 256 series, relative deviation max {e.max():.2e} median {np.median(e):.2e}")
