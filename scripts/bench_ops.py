"""Times every operator entry point of the C ABI at one shape and prints algorithmic GB/s (bytes per block as in
DESIGN.md section 4.2).  Usage: python3 scripts/bench_ops.py [--batch B --T T --d d --dtype f64]."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import markovflow_amd as mfa
from markovflow_amd import synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024); ap.add_argument("--T", type=int, default=2000)
ap.add_argument("--d", type=int, default=6); ap.add_argument("--dtype", default="f64"); ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda:0"); dt = torch.float64 if a.dtype == "f64" else torch.float32
comp = {2: (3,), 4: (3, 3), 6: (5, 5), 9: (5, 5, 5), 3: (5,), 1: (1,), 5: (3, 5), 8: (3, 5, 5), 7: (1, 5, 5)}[a.d]
inp = synthetic.make_ssm(a.batch, a.T, comp, dtype=dt, device=dev)
kf = synthetic.kalman_filter_from(inp)
ssm = kf.prior_ssm
B, T, d, s = a.batch, a.T, a.d, (8 if dt == torch.float64 else 4)

def timeit(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters

post_prec = kf._k_inv_post
chol = post_prec.cholesky
rhs = torch.randn(B, T, d, dtype=dt, device=dev)
post = kf.posterior_state_space_model()
rows = [
    ("KalmanFilter.log_likelihood", kf.log_likelihood, 2 * d * d + 3 * d + 1),
    ("_k_inv_post (ssm_precision)", lambda: kf._k_inv_post, 4 * d * d + d),
    ("Sym.cholesky", lambda: post_prec.cholesky, 4 * d * d),
    ("Lower.solve", lambda: chol.solve(rhs), 2 * d * d + 2 * d),
    ("Lower.solve^T", lambda: chol.solve(rhs, transpose_left=True), 2 * d * d + 2 * d),
    ("Sym.dense_mult", lambda: post_prec.dense_mult(rhs), 2 * d * d + 2 * d),
    ("Lower.abs_log_det", chol.abs_log_det, d),
    ("Lower.block_diagonal_of_inverse", chol.block_diagonal_of_inverse, 3 * d * d),
    ("Sym.upper_diagonal_lower", post_prec.upper_diagonal_lower, 4 * d * d),
    ("ssm.marginal_means", lambda: ssm.marginal_means, d * d + 2 * d),
    ("ssm.marginal_covariances", lambda: ssm.marginal_covariances, 7 * d * d),
    ("kf.posterior_state_space_model", kf.posterior_state_space_model, 6 * d * d + 4 * d),
    ("post.kl_divergence(prior)", lambda: post.kl_divergence(ssm), 8 * d * d),
]
print(f"B={B} T={T} d={d} {a.dtype}  (time includes Python-side output allocation; GB/s = algorithmic bytes / time)")
for name, fn, elems in rows:
    ms = timeit(fn)
    print(f"  {name:34s} {ms:9.3f} ms   {B * T * elems * s / ms / 1e6:9.1f} GB/s")
