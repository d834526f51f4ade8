"""posterior_state_space_model() at few-series / long-chain shapes: time of the whole call and, by HIP events
on the launch stream, of its kernels; algorithmic GB/s on (4 d^2 + 3 d + m d + m) s bytes per step.
    python3 scripts/bench_post.py [--batch 1024 --T 10000 --iters 5 --route streamed|ops|serial]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import markovflow_amd as mfa
from markovflow_amd import synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024); ap.add_argument("--T", type=int, default=10000); ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--comps", default="5,5"); ap.add_argument("--outputs", type=int, default=1)
ap.add_argument("--route", default="streamed"); ap.add_argument("--chunks", type=int, default=0)
a = ap.parse_args()
dev = torch.device("cuda:0")
comps = tuple(int(c) for c in a.comps.split(","))
inp = synthetic.make_ssm(a.batch, a.T, comps, output_dim=a.outputs, dtype=torch.float64, device=dev)
kf = synthetic.kalman_filter_from(inp)
kf._chunks = a.chunks
d, m = inp["A"].shape[-1], inp["H"].shape[-2]
if a.route == "ops":
    mfa.BaseKalmanFilter._POST_STREAMED = False
elif a.route == "serial":
    mfa.BaseKalmanFilter._POST_FUSED_MIN_SERIES = 1
bytes_alg = a.batch * a.T * (4 * d * d + 3 * d + m * d + m) * 8
ts = []
for i in range(a.iters + 2):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); post = kf.posterior_state_space_model(); e1.record(); torch.cuda.synchronize()
    if i >= 2: ts.append(e0.elapsed_time(e1))
ts.sort()
t = ts[len(ts) // 2]
print(f"posterior_state_space_model route={a.route} B={a.batch} T={a.T} d={d} m={m} fp64: median {t:.3f} ms (min {ts[0]:.3f}), "
      f"{bytes_alg / t / 1e9:.2f} TB/s algorithmic = {bytes_alg / t / 1e9 / 8 * 100:.1f} % of 8 TB/s")
