"""Batched block-tridiagonal Cholesky + solve (one lane per series) for rocprofv3: B=16384, T=500, d=6, fp64."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from markovflow_amd import synthetic
dev = torch.device("cuda", 0)
kf = synthetic.kalman_filter_from(synthetic.make_ssm(16384, 500, (5, 5), dtype=torch.float64, device=dev))
prec = kf._k_inv_post
rhs = torch.randn(16384, 500, 6, dtype=torch.float64, device=dev)
chol = prec.cholesky
for _ in range(5):
    chol = prec.cholesky
    chol.solve(rhs)
torch.cuda.synchronize()
