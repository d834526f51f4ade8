"""log_likelihood at the hand-over shapes (B=512, T=1000, m=1) across d = 15 ... 32, both dtypes: the row kernels (d <= 15), the
wave kernels (16 <= d <= 32, csrc/mf_wave.hpp) and - with MF_WAVE=0 in an experiment build - the tile engine.
Usage: python3 scripts/bench_wave.py [--dims 15,16,24,32] [--dtype f64] [--B 512] [--T 1000] [--chunks 0]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import markovflow_amd as mfa  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dims", default="15,16,17,24,30,32")
ap.add_argument("--dtype", default="f64")
ap.add_argument("--B", type=int, default=512)
ap.add_argument("--T", type=int, default=1000)
ap.add_argument("--m", type=int, default=1)
ap.add_argument("--chunks", default="0")
args = ap.parse_args()
dev = torch.device("cuda:0")
dtype = torch.float64 if args.dtype == "f64" else torch.float32
esz = 8 if args.dtype == "f64" else 4
g = torch.Generator(device=dev)
g.manual_seed(5)
for d in [int(x) for x in args.dims.split(",")]:
    bsz, t, m = args.B, args.T, args.m
    rn = lambda *s: torch.randn(*s, dtype=dtype, device=dev, generator=g)                      # noqa: E731
    eye = torch.eye(d, dtype=dtype, device=dev)
    chol = lambda n: torch.tril(0.1 * rn(bsz, n, d, d)) + eye                                  # noqa: E731
    ssm = mfa.StateSpaceModel(rn(bsz, d), chol(1)[:, 0], (0.5 / d ** 0.5) * rn(bsz, t - 1, d, d), 0.3 * rn(bsz, t - 1, d), chol(t - 1))
    kf = mfa.KalmanFilter(ssm, mfa.EmissionModel(rn(bsz, t, m, d)), rn(bsz, t, m), 0.7 * torch.eye(m, dtype=dtype, device=dev))
    for _ in range(30):                      # (a process that has just started measures the clock ramp, not the kernel: 3.5 x)
        kf.log_likelihood()
    for chunks in [int(c) for c in args.chunks.split(",")]:
        kf._chunks = chunks
        for _ in range(3):
            val = kf.log_likelihood()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 10
        e0.record()
        for _ in range(iters):
            kf.log_likelihood()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        steps = bsz * t
        byt = (2 * d * d + d + m * d + m) * esz
        print(f"d={d:3d} m={m} B={bsz} T={t} {args.dtype} chunks={chunks}: log_likelihood {ms:8.3f} ms  {steps / ms / 1e3:8.1f} k steps/ms  "
              f"{steps * byt / ms / 1e6:8.1f} GB/s algorithmic  {steps * 15 * d ** 3 / ms / 1e9:7.2f} TFLOP/s (15 d^3)   ll {float(val):.6f}", flush=True)
