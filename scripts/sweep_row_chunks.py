import sys; sys.path.insert(0, "/root/repo")
import torch
from markovflow_amd import synthetic
dev = torch.device("cuda:0")
for d in (12, 15):
    kf = synthetic.kalman_filter_from(synthetic.make_dense_ssm(512, 1000, d, 3, dtype=torch.float64, device=dev))
    for ch in (0, 4, 8, 12, 16, 24, 32):
        kf._chunks = ch
        kf.log_likelihood(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); kf.log_likelihood(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        print(f"d={d} chunks={ch:2d}: {best:.3f} ms", flush=True)
