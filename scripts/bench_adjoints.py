"""Reverse mode through cholesky -> block_diagonal_of_inverse at 10 <= d <= 32 (csrc/mf_adj.hip): forward and backward times.
    python scripts/bench_adjoints.py [--dims 12,16,30,32] [--batch 4,256] [--T 1001] [--dtype f64]"""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import markovflow_amd as mfa

ap = argparse.ArgumentParser()
ap.add_argument("--dims", default="12,16,30,32")
ap.add_argument("--batch", default="4,256")
ap.add_argument("--T", type=int, default=1001)
ap.add_argument("--dtype", default="f64")
a = ap.parse_args()
dt = torch.float64 if a.dtype == "f64" else torch.float32
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)
for d in map(int, a.dims.split(",")):
    for B in map(int, a.batch.split(",")):
        n = a.T
        ld = torch.tril((4.0 / d) * 0.3 * torch.randn(B, n, d, d, dtype=dt, device=dev, generator=g), -1) + torch.diag_embed(
            1 + torch.rand(B, n, d, dtype=dt, device=dev, generator=g))
        ls = (2.0 / d) * 0.3 * torch.randn(B, n - 1, d, d, dtype=dt, device=dev, generator=g)
        dg = ld @ ld.transpose(-1, -2)
        dg[:, 1:] += ls @ ls.transpose(-1, -2)
        sb = ls @ ld[:, :-1].transpose(-1, -2)
        dg.requires_grad_(True); sb.requires_grad_(True)
        wd = torch.randn(B, n, d, d, dtype=dt, device=dev, generator=g)
        f, b = [], []
        for i in range(4):
            dg.grad = sb.grad = None
            torch.cuda.synchronize()
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
            inv_d, inv_s = mfa.SymmetricBlockTriDiagonal(dg, sb).cholesky._diag_and_sub_of_inverse(want_sub=True)
            loss = torch.sum(inv_d * wd) + torch.sum(inv_s)
            e1.record(); loss.backward(); e2.record()
            torch.cuda.synchronize()
            if i:
                f.append(e0.elapsed_time(e1)); b.append(e1.elapsed_time(e2))
        fm, bm = sorted(f)[1], sorted(b)[1]
        print(f"d={d:2d} B={B:4d} T={n} {a.dtype}: forward {fm:8.3f} ms  backward {bm:8.3f} ms  ({bm / (n * 2):.4f} ms per block per adjoint... "
              f"{1e3 * bm / n / 2:.2f} us)  finite={bool(torch.isfinite(dg.grad).all())}")
        del ld, ls, dg, sb, wd
