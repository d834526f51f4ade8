"""kl_divergence alone at 16 <= d <= 32 (for rocprofv3 kernel stats): python scripts/prof_kl_only.py [d] [B] [T]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import markovflow_amd as mfa
from markovflow_amd import synthetic
d = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
T = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
dev = "cuda:0"
kf = synthetic.kalman_filter_from(synthetic.make_dense_ssm(B, T, d, 1, dtype=torch.float64, device=dev))
post = kf.posterior_state_space_model(); prior = kf.prior_ssm
torch.cuda.synchronize()
for _ in range(3): post.kl_divergence(prior)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): v = post.kl_divergence(prior)
e1.record(); torch.cuda.synchronize()
print(f"d={d} B={B} T={T} f64 kl_divergence {e0.elapsed_time(e1) / 10:.3f} ms  sum {float(v.sum()):.6f}")
