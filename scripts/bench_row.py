"""Row kernels (csrc/mf_row.hpp): log-likelihood timing and parity at BASELINE config 4's shape and a d sweep.

    python3 scripts/bench_row.py [--chunks 0,8,16,24,32,48] [--dims 7,8,9] [--iters 20]

Times `KalmanFilter.log_likelihood()` (whole evaluation, HIP events on torch's stream) and the level-0 kernel (the library's
own events), checks every series against the C oracle, for each requested number of chunks per series."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from markovflow_amd import synthetic  # noqa: E402
from oracle import c_oracle as C  # noqa: E402


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(iters):
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[0], ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", default="0")
    ap.add_argument("--dims", default="9")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--T", type=int, default=1000)
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--no-check", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    dt = torch.float64 if a.dtype == "f64" else torch.float32
    for d in [int(x) for x in a.dims.split(",")]:
        comps, m = {9: ((5, 5, 5), 3), 8: ((5, 5, 3), 3), 7: ((5, 3, 3), 3), 6: ((5, 5), 1)}[d]
        inp = synthetic.make_ssm(a.batch, a.T, comps, output_dim=m, dtype=dt, device=dev)
        kf = synthetic.kalman_filter_from(inp)
        ref = None
        if not a.no_check:
            h = {k: (v.double().cpu().numpy() if torch.is_tensor(v) else v) for k, v in inp.items()}
            r_inv = np.linalg.inv(h["cholR"] @ h["cholR"].T)
            ref = C.kf_loglik(h["mu0"], h["cholP0"], h["A"], h["b"], h["cholQ"], h["H"], h["y"], r_inv)
        for ch in [int(x) for x in a.chunks.split(",")]:
            kf._chunks = ch
            tmin, tmed = timed(kf.log_likelihood, a.iters)
            msg = f"d={d} m={m} B={a.batch} T={a.T} {a.dtype} chunks={ch:3d}: log_likelihood min {tmin:.4f} ms, median {tmed:.4f} ms"
            if ref is not None:
                per = (kf._log_likelihood_per_series() + kf._constant_terms(a.T)).double().cpu().numpy()
                msg += f"   max rel. deviation from the C oracle {np.max(np.abs(per - ref) / np.abs(ref)):.2e}"
            print(msg, flush=True)


if __name__ == "__main__":
    main()
