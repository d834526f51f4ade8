"""Config 2 (KalmanFilter.log_likelihood, B=256, T=4096, d=4, fp64) on the tree named by MF_TREE (default: this one): GPU time per
call (HIP events around 200 back-to-back calls), host time per call (wall clock of the enqueue loop) and the same for the bare
C-ABI call - separates kernel time from the Python-side work of a 0.1-ms call.
Usage: MF_TREE=_prev python3 scripts/ab_config2.py ; python3 scripts/ab_config2.py"""
import os
import sys
import time

tree = os.path.abspath(os.environ.get("MF_TREE", os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, tree)
import torch  # noqa: E402

import markovflow_amd as mfa  # noqa: E402
from markovflow_amd import synthetic  # noqa: E402

dev = torch.device("cuda:0")
shapes = [(256, 4096, (3, 3), "config2 d=4"), (64, 10000, (5, 5), "B=64 T=1e4 d=6"), (1024, 10000, (5, 5), "headline")]
for bsz, tn, orders, name in shapes:
    inp = synthetic.make_ssm(bsz, tn, orders, dtype=torch.float64, device=dev)
    kf = synthetic.kalman_filter_from(inp)
    for _ in range(20):
        kf.log_likelihood()
    torch.cuda.synchronize()
    best_gpu, best_host = 1e9, 1e9
    iters = 200 if bsz * tn < 5e6 else 40
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(iters):
            kf.log_likelihood()
        e1.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        best_gpu = min(best_gpu, e0.elapsed_time(e1) / iters)
        best_host = min(best_host, (t1 - t0) / iters * 1e3)
    print(f"{os.path.basename(tree):10s} {name:16s} gpu {best_gpu:.4f} ms/call   host enqueue {best_host:.4f} ms/call   "
          f"ll {float(kf.log_likelihood()):.9f}", flush=True)
