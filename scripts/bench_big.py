"""Times KalmanFilter.log_likelihood on BASELINE config 5 (state_dim=64, T=2048, fp32; m spatial outputs) through
the large-d LDS/MFMA path and prints steps/s and the MFMA-side rate.  Flop model per (series, step), DP = padded d:
products on the matrix cores: 2*DP^3 * (Ci*A 1/2 + Bm^T Bm 1 + Y 1/2 + W 1/2 + Ci^T Ci 1/2 + W W^T 1 + spike (V 1/2 +
V^T V 1 + W V 1)) + factor/inverse tiles ~ 2*DP^3 * (1/3 + 1/3 + 1/3)."""
import argparse, ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import markovflow_amd as mfa

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8); ap.add_argument("--T", type=int, default=2048)
ap.add_argument("--d", type=int, default=64); ap.add_argument("--m", type=int, default=32)
ap.add_argument("--chunks", type=int, default=0); ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda:0"); dt = torch.float32
B, T, d, m = a.batch, a.T, a.d, a.m
g = torch.Generator(device=dev); g.manual_seed(0)
eye = torch.eye(d, dtype=dt, device=dev)
A = 0.9 * eye + (0.3 / d ** 0.5) * torch.randn(B, T - 1, d, d, dtype=dt, device=dev, generator=g)
cq = torch.tril((0.3 / d ** 0.5) * torch.randn(B, T - 1, d, d, dtype=dt, device=dev, generator=g)) + 0.5 * eye
cp0 = torch.tril(0.1 * torch.randn(B, d, d, dtype=dt, device=dev, generator=g)) + eye
ssm = mfa.StateSpaceModel(torch.randn(B, d, dtype=dt, device=dev, generator=g), cp0, A,
                          0.1 * torch.randn(B, T - 1, d, dtype=dt, device=dev, generator=g), cq)
kf = mfa.KalmanFilter(ssm, mfa.EmissionModel(torch.randn(B, T, m, d, dtype=dt, device=dev, generator=g) / d ** 0.5),
                      torch.randn(B, T, m, dtype=dt, device=dev, generator=g), 0.3 * torch.eye(m, dtype=dt, device=dev))
kf._chunks = a.chunks
hip = ctypes.CDLL("libamdhip64.so")
e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
hip.hipEventCreate(ctypes.byref(e0)); hip.hipEventCreate(ctypes.byref(e1))
kf._prof_events = (e0, e1)
ms = []
for i in range(a.iters + 2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ll = kf.log_likelihood(); torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
    f = ctypes.c_float(); hip.hipEventElapsedTime(ctypes.byref(f), e0, e1)
    if i >= 2: ms.append((f.value, wall))
k = sum(x[0] for x in ms) / len(ms); w = sum(x[1] for x in ms) / len(ms)
dp = 16 * ((d + 15) // 16)
flop_step = 2 * dp ** 3 * (0.5 + 1 + 0.5 + 0.5 + 0.5 + 1 + 0.5 + 1 + 1 + 1.0)
byts = B * T * (2 * d * d + d + m * d + m) * 4
print(f"B={B} T={T} d={d} m={m} f32 chunks={a.chunks}: level-0 kernel {k:.3f} ms, log_likelihood wall {w:.3f} ms -> "
      f"{B * T / w * 1e3:.3e} steps/s | {B * T * flop_step / k / 1e9:.2f} TFLOP/s on the level-0 kernel (f32 MFMA peak 157) | "
      f"{byts / k / 1e6:.1f} GB/s algorithmic | ll={float(ll):.4f}")
