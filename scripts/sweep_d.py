"""log_likelihood time against the state dimension at BASELINE config 4's shape (B=512, T=1000, m=3): where the register /
row kernels (d <= 9) hand over to the LDS-tile / MFMA path (d >= 10).   python3 scripts/sweep_d.py [--dtype f64]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from markovflow_amd import synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="f64"); ap.add_argument("--batch", type=int, default=512); ap.add_argument("--T", type=int, default=1000)
ap.add_argument("--m", type=int, default=3); ap.add_argument("--dims", default="6,8,9,10,12,14,16,20,32")
a = ap.parse_args()
dt = torch.float64 if a.dtype == "f64" else torch.float32
dev = torch.device("cuda:0")
for d in [int(x) for x in a.dims.split(",")]:
    kf = synthetic.kalman_filter_from(synthetic.make_dense_ssm(a.batch, a.T, d, a.m, dtype=dt, device=dev))
    kf.log_likelihood(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); kf.log_likelihood(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    byts = a.batch * a.T * (2 * d * d + d + a.m * d + a.m) * (8 if a.dtype == "f64" else 4)
    print(f"d={d:3d} m={a.m} B={a.batch} T={a.T} {a.dtype}: log_likelihood {best:8.3f} ms  {a.batch * a.T / best / 1e3:8.1f} k steps/ms... "
          f"{byts / best / 1e6:7.1f} GB/s algorithmic", flush=True)
