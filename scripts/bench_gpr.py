"""GaussianProcessRegression.log_likelihood from (time points, observations, hyper-parameters): the fused route
(mf_gpr_matern_loglik: kernel -> SSM generation inside the Kalman sweep) against the materialised route
(mf_sde_matern_transitions + mf_kf_loglik), B series x T points, Sum(Matern52, Matern52) (d = 6) by default."""
import argparse, ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import markovflow_amd as mfa

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024); ap.add_argument("--T", type=int, default=10000)
ap.add_argument("--sig", default="5,5"); ap.add_argument("--dtype", default="f64"); ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda:0"); dt = torch.float64 if a.dtype == "f64" else torch.float32
g = torch.Generator(device=dev); g.manual_seed(0)
B, T = a.batch, a.T
t = torch.cumsum(0.05 + 0.05 * torch.empty(B, T, dtype=dt, device=dev).exponential_(1.0, generator=g), dim=-1)
y = torch.randn(B, T, 1, dtype=dt, device=dev, generator=g)
cls = {1: mfa.Matern12, 3: mfa.Matern32, 5: mfa.Matern52}
parts = [cls[int(o)](0.5 + 1.5 * torch.rand(B, dtype=dt, device=dev, generator=g), 0.5 + 1.5 * torch.rand(B, dtype=dt, device=dev, generator=g))
         for o in a.sig.split(",")]
kern = parts[0] if len(parts) == 1 else mfa.Sum(parts, jitter=1e-9)
gpr = mfa.GaussianProcessRegression((t, y), kern, chol_obs_covariance=(0.1 ** 0.5) * torch.eye(1, dtype=dt, device=dev))
hip = ctypes.CDLL("libamdhip64.so"); e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
hip.hipEventCreate(ctypes.byref(e0)); hip.hipEventCreate(ctypes.byref(e1))
gpr._prof_events = (e0, e1)

def timeit(fn):
    for _ in range(2): r = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.iters): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / a.iters * 1e3, r
gpr.fused = True
ms_f, ll_f = timeit(gpr.log_likelihood)
f = ctypes.c_float()
if hip.hipEventElapsedTime(ctypes.byref(f), e0, e1) != 0:      # signature outside the fused kernel: the events were never recorded
    f.value = float('nan'); hip.hipGetLastError()
gpr.fused = False
ms_m, ll_m = timeit(gpr.log_likelihood)
d = kern.state_dim
print(f"B={B} T={T} sig=({a.sig}) d={d} {a.dtype}: fused {ms_f:.3f} ms (level-0 kernel {f.value:.3f} ms) = {B*T/ms_f*1e3:.3e} steps/s | "
      f"materialised (generate + filter) {ms_m:.3f} ms = {B*T/ms_m*1e3:.3e} steps/s | speed-up {ms_m/ms_f:.2f}x | "
      f"rel diff {abs(float(ll_f)-float(ll_m))/abs(float(ll_m)):.1e}")
