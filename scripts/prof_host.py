"""Host-side profile (cProfile) of KalmanFilter.log_likelihood() at BASELINE config 2's shape: the evaluation is ~0.10 ms of kernels,
so the Python layer's per-call cost is what its wall time is made of.  python3 scripts/prof_host.py [calls]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from markovflow_amd import synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dev = torch.device("cuda:0")
kf = synthetic.kalman_filter_from(synthetic.make_ssm(256, 4096, (3, 3), dtype=torch.float64, device=dev))
for _ in range(50):
    kf.log_likelihood()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    kf.log_likelihood()
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"host enqueue {1e6 * (t1 - t0) / n:.1f} us per call (unprofiled)")
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    kf.log_likelihood()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime")
st.print_stats(28)
