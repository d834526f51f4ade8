"""Large-d (config 5 shape) marginal covariances: time-partitioned forward recursion against the route over the serial
large-d operators (precision -> cholesky -> block_diagonal_of_inverse): python3 scripts/bench_cov_big.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import markovflow_amd as mfa
dev = "cuda:0"; dt = torch.float32
B, T, d = 8, 2048, 64
g = torch.Generator(device=dev); g.manual_seed(0)
eye = torch.eye(d, dtype=dt, device=dev)
A = 0.9 * eye + (0.3 / d ** 0.5) * torch.randn(B, T - 1, d, d, dtype=dt, device=dev, generator=g)
cq = torch.tril((0.3 / d ** 0.5) * torch.randn(B, T - 1, d, d, dtype=dt, device=dev, generator=g)) + 0.5 * eye
cp0 = torch.tril(0.1 * torch.randn(B, d, d, dtype=dt, device=dev, generator=g)) + eye
ssm = mfa.StateSpaceModel(torch.randn(B, d, dtype=dt, device=dev, generator=g), cp0, A, 0.1 * torch.randn(B, T - 1, d, dtype=dt, device=dev, generator=g), cq)
def timeit(fn, it=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
t_new = timeit(lambda: ssm.covariance_blocks())
t_mom = timeit(lambda: ssm._moments(True))
t_means = timeit(lambda: ssm.marginal_means)
print(f"marginals (means + covariances + cross) in the same three passes {t_mom:.2f} ms; marginal_means alone (one lane group per series) {t_means:.2f} ms")
def ref():
    c = ssm.precision.cholesky.block_diagonal_of_inverse(); return c, ssm.subsequent_covariances(c)
t_ref = timeit(ref, 1)
a, b = ssm.covariance_blocks(); c, e = ref()
print(f"d=64 T=2048 B=8 f32 covariance_blocks: forward recursion (partitioned) {t_new:.2f} ms, reference route over the large-d operators {t_ref:.2f} ms; max rel diff {float((a-c).abs().max()/c.abs().max()):.2e}")
H = torch.randn(B, T, 32, d, dtype=dt, device=dev, generator=g) / d ** 0.5
kf = mfa.KalmanFilter(ssm, mfa.EmissionModel(H), torch.randn(B, T, 32, dtype=dt, device=dev, generator=g), 0.3 * torch.eye(32, dtype=dt, device=dev))
t_post = timeit(lambda: kf.posterior_state_space_model(), 1)
post = kf.posterior_state_space_model()
t_kl = timeit(lambda: post.kl_divergence(ssm), 1)
t_chol = timeit(lambda: ssm.precision.cholesky, 1)
print(f"posterior_state_space_model {t_post:.1f} ms, kl_divergence(posterior || prior) {t_kl:.1f} ms, precision.cholesky {t_chol:.1f} ms, log_likelihood {timeit(lambda: kf.log_likelihood()):.2f} ms")
