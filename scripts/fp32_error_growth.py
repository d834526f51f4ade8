"""fp32 error of the parallel-in-time Cholesky / solve (BASELINE config 3's path) as a function of the chain length: the
posterior precision of a sum of three Matern-3/2 components (the matrix of tests/test_gpu_baseline_configs.py), fp32 on the
GPU against the fp64 C oracle on the same fp32-rounded matrix.  Prints max |error| / block scale per length - the measured
basis of the tolerance that test uses.   python3 scripts/fp32_error_growth.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import markovflow_amd as mfa  # noqa: E402
from markovflow_amd import synthetic  # noqa: E402
from oracle import c_oracle as C  # noqa: E402

dev = "cuda:0"
print("    T   |  cholesky diag   sub-diag |  solve   solve^T   (max abs error / scale; fp32 eps = 6e-8)")
for n in (100, 1000, 10000, 100000):
    worst = np.zeros(4)
    for seed in range(3):
        inp = synthetic.make_ssm(1, n, (3, 3, 3), dtype=torch.float64, device=dev, dt_min=0.2, dt_scale=0.3, seed=100 + seed)
        prec = synthetic.kalman_filter_from(inp)._k_inv_post
        d32, s32 = prec.block_diagonal.float().contiguous(), prec.block_sub_diagonal.float().contiguous()
        chol = mfa.SymmetricBlockTriDiagonal(d32, s32).cholesky
        ld_ref, ls_ref = C.btd_cholesky(d32.double().cpu().numpy(), s32.double().cpu().numpy())
        ld, ls = chol.block_diagonal.double().cpu().numpy(), chol.block_sub_diagonal.double().cpu().numpy()
        sc = np.abs(ld_ref).max(axis=(-2, -1), keepdims=True)
        rhs = torch.randn(1, n, 6, dtype=torch.float32, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
        out = chol.solve(rhs).double().cpu().numpy()
        ref = C.btd_solve(ld_ref, ls_ref, rhs.double().cpu().numpy())
        out_t = chol.solve(rhs, transpose_left=True).double().cpu().numpy()
        ref_t = C.btd_solve(ld_ref, ls_ref, rhs.double().cpu().numpy(), transpose=True)
        worst = np.maximum(worst, [np.max(np.abs(ld - ld_ref) / sc), np.max(np.abs(ls - ls_ref) / sc[:, 1:]),
                                   np.max(np.abs(out - ref)) / np.abs(ref).max(), np.max(np.abs(out_t - ref_t)) / np.abs(ref_t).max()])
    print(f"{n:7d} |  {worst[0]:.2e}   {worst[1]:.2e} | {worst[2]:.2e}  {worst[3]:.2e}", flush=True)
