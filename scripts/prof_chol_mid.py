"""Block-tridiagonal Cholesky + solve + U D U^T at a MID shape (B=1024, T=2000, d=6, fp64: lane level 0, reduced levels) for rocprofv3."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import markovflow_amd as mfa
from markovflow_amd import synthetic
B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 2000)
dev = torch.device("cuda", 0)
kf = synthetic.kalman_filter_from(synthetic.make_ssm(B, T, (5, 5), dtype=torch.float64, device=dev))
prec = kf._k_inv_post
rhs = torch.randn(B, T, 6, dtype=torch.float64, device=dev)
for _ in range(6):
    sym = mfa.SymmetricBlockTriDiagonal(prec.block_diagonal, prec.block_sub_diagonal)
    chol = sym.cholesky
    chol.solve(rhs)
    sym.upper_diagonal_lower()
torch.cuda.synchronize()
