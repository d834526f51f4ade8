"""Small driver for profiling / A-B timing of the K0 pipeline: random well-conditioned inputs made with a
handful of torch ops (so rocprofv3 traces stay small), N evaluations, HIP-event timing of the dominant kernel."""
import argparse, ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import markovflow_amd as mfa

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024); ap.add_argument("--T", type=int, default=10000)
ap.add_argument("--d", type=int, default=6); ap.add_argument("--dtype", default="f64")
ap.add_argument("--chunks", type=int, default=0); ap.add_argument("--iters", type=int, default=5); ap.add_argument("--m", type=int, default=1)
ap.add_argument("--nan-ok", action="store_true", help="experiment builds that produce garbage on purpose: never raise on a pivot")
args = ap.parse_args()
dev = torch.device("cuda:0"); dt = torch.float64 if args.dtype == "f64" else torch.float32
B, T, d = args.batch, args.T, args.d
g = torch.Generator(device=dev); g.manual_seed(0)
eye = torch.eye(d, dtype=dt, device=dev)
A = 0.9 * eye + 0.05 * torch.randn(B, T - 1, d, d, dtype=dt, device=dev, generator=g)
cq = torch.tril(0.1 * torch.randn(B, T - 1, d, d, dtype=dt, device=dev, generator=g)) + 0.5 * eye
cp0 = torch.tril(0.1 * torch.randn(B, d, d, dtype=dt, device=dev, generator=g)) + eye
ssm = mfa.StateSpaceModel(torch.randn(B, d, dtype=dt, device=dev, generator=g), cp0, A,
                          0.1 * torch.randn(B, T - 1, d, dtype=dt, device=dev, generator=g), cq)
m = args.m
kf = mfa.KalmanFilter(ssm, mfa.EmissionModel(torch.randn(B, T, m, d, dtype=dt, device=dev, generator=g)),
                      torch.randn(B, T, m, dtype=dt, device=dev, generator=g), 0.3 * torch.eye(m, dtype=dt, device=dev))
kf._chunks = args.chunks
hip = ctypes.CDLL("libamdhip64.so")
e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
hip.hipEventCreate(ctypes.byref(e0)); hip.hipEventCreate(ctypes.byref(e1))
kf._prof_events = (e0, e1)
import contextlib
ms = []
ctx = mfa.errors_as_nan() if args.nan_ok else contextlib.nullcontext()
ctx.__enter__()
for i in range(args.iters + 2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ll = kf.log_likelihood(); torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
    f = ctypes.c_float(); hip.hipEventElapsedTime(ctypes.byref(f), e0, e1)
    if i >= 2: ms.append((f.value, wall))
k = sum(m[0] for m in ms) / len(ms); w = sum(m[1] for m in ms) / len(ms)
esz = 8 if dt == torch.float64 else 4
byts = B * T * (2 * d * d + d + args.m * d + args.m) * esz
print(f"dtype={args.dtype} B={B} T={T} d={d} chunks={args.chunks} impl={os.environ.get('MF_KF_IMPL','lds')}: "
      f"kernel {k:.3f} ms  wall {w:.3f} ms  {byts / k / 1e6:.0f} GB/s ({byts / k / 1e6 / 8000:.3f} of 8 TB/s)  ll={float(ll):.6f}")
