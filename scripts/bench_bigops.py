"""Large-d operators at BASELINE config 5's shape (B=8, T=2048, d=64, m=32, fp32): HIP-event times of each operator entry point.
    python3 scripts/bench_bigops.py [--batch 8] [--T 2048] [--d 64] [--m 32] [--dtype f32] [--grad]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import markovflow_amd as mfa

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8); ap.add_argument("--T", type=int, default=2048)
ap.add_argument("--d", type=int, default=64); ap.add_argument("--m", type=int, default=32)
ap.add_argument("--dtype", default="f32"); ap.add_argument("--iters", type=int, default=3); ap.add_argument("--grad", action="store_true")
a = ap.parse_args()
dev = "cuda:0"; dt = torch.float32 if a.dtype == "f32" else torch.float64
B, T, d, m = a.batch, a.T, a.d, a.m
g = torch.Generator(device=dev); g.manual_seed(0)
rn = lambda *s: torch.randn(*s, dtype=dt, device=dev, generator=g)  # noqa: E731
eye = torch.eye(d, dtype=dt, device=dev)
A = 0.9 * eye + (0.3 / d ** 0.5) * rn(B, T - 1, d, d)
cq = torch.tril((0.3 / d ** 0.5) * rn(B, T - 1, d, d)) + 0.5 * eye
cp0 = torch.tril(0.1 * rn(B, d, d)) + eye
ssm = mfa.StateSpaceModel(rn(B, d), cp0, A, 0.1 * rn(B, T - 1, d), cq)
H = rn(B, T, m, d) / d ** 0.5
kf = mfa.KalmanFilter(ssm, mfa.EmissionModel(H), rn(B, T, m), 0.3 * torch.eye(m, dtype=dt, device=dev))


def timed(fn):
    fn(); torch.cuda.synchronize()
    best = 1e30
    for _ in range(a.iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


prec = ssm.precision
sym = mfa.SymmetricBlockTriDiagonal(prec.block_diagonal, prec.block_sub_diagonal)
chol = sym.cholesky
low = mfa.LowerTriangularBlockTriDiagonal(chol.block_diagonal, chol.block_sub_diagonal)
rhs = rn(B, T, d)
print(f"B={B} T={T} d={d} m={m} {a.dtype}")
for name, fn in [("ssm.precision", lambda: ssm.precision),
                 ("Sym.cholesky", lambda: mfa.SymmetricBlockTriDiagonal(prec.block_diagonal, prec.block_sub_diagonal).cholesky),
                 ("Sym.upper_diagonal_lower", lambda: mfa.SymmetricBlockTriDiagonal(prec.block_diagonal, prec.block_sub_diagonal).upper_diagonal_lower()),
                 ("Lower.solve", lambda: low.solve(rhs)),
                 ("Lower.solve^T", lambda: low.solve(rhs, transpose_left=True)),
                 ("Lower.block_diagonal_of_inverse", lambda: low.block_diagonal_of_inverse()),
                 ("kf.posterior_state_space_model", lambda: kf.posterior_state_space_model()),
                 ("kf.log_likelihood", lambda: kf.log_likelihood())]:
    print(f"  {name:34s} {timed(fn):9.3f} ms", flush=True)
post = kf.posterior_state_space_model()
print(f"  {'post.kl_divergence(prior)':34s} {timed(lambda: post.kl_divergence(ssm)):9.3f} ms")
print(f"  {'post._moments (marginals)':34s} {timed(lambda: post._moments(True)):9.3f} ms")
if a.grad:
    leaves = [t.detach().clone().requires_grad_(True) for t in ssm._flat_params()]
    kfg = mfa.KalmanFilter(mfa.StateSpaceModel(*leaves), kf.emission, kf.observations, kf._chol_obs_covariance)
    def fb():
        for l in leaves: l.grad = None
        kfg.log_likelihood().backward()
    print(f"  {'log_likelihood forward+backward':34s} {timed(fb):9.3f} ms")
