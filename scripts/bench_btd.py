"""Times SymmetricBlockTriDiagonal.cholesky and LowerTriangularBlockTriDiagonal.solve on BASELINE config 3
(T=100000, d=6, fp32, one chain) and prints algorithmic GB/s (cholesky 4 d^2 s, solve (2 d^2 + 2 d) s
bytes per block).  Called through the C ABI directly so that allocation is outside the timed region."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from markovflow_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=100000); ap.add_argument("--d", type=int, default=6)
ap.add_argument("--batch", type=int, default=1); ap.add_argument("--dtype", default="f32"); ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--op", default="all", help="all | chol (the factorisation only: for kernel timelines)")
a = ap.parse_args()
dev = torch.device("cuda:0"); dt = torch.float32 if a.dtype == "f32" else torch.float64
B, n, d = a.batch, a.T, a.d
g = torch.Generator(device=dev); g.manual_seed(0)
ldiag = torch.tril(0.3 * torch.randn(B, n, d, d, dtype=torch.float64, device=dev, generator=g))
ldiag = ldiag - torch.diag_embed(torch.diagonal(ldiag, dim1=-2, dim2=-1)) + torch.diag_embed(1 + torch.rand(B, n, d, dtype=torch.float64, device=dev, generator=g))
lsub = 0.3 * torch.randn(B, n - 1, d, d, dtype=torch.float64, device=dev, generator=g)
diag = ldiag @ ldiag.transpose(-1, -2); diag[:, 1:] += lsub @ lsub.transpose(-1, -2)
sub = lsub @ ldiag[:, :-1].transpose(-1, -2)
diag, sub = diag.to(dt).contiguous(), sub.to(dt).contiguous()
rhs = torch.randn(B, n, d, dtype=dt, device=dev, generator=g)
lib = _lib.load(); esz = diag.element_size()
wsb = int(lib.mf_btd_cholesky_workspace_bytes(B, n, d, esz)); ws = _lib.workspace(wsb, dev)
wss = int(lib.mf_btd_solve_workspace_bytes(B, B, n, d, esz)); ws2 = _lib.workspace(wss, dev)
ld, ls, out = torch.empty_like(diag), torch.empty_like(sub), torch.empty_like(rhs)
info = _lib.new_info(dev); st = _lib.stream_ptr(dev)

def chol():
    _lib.call("mf_btd_cholesky", dt, B, n, d, _lib.ptr(diag), _lib.ptr(sub), _lib.ptr(ld), _lib.ptr(ls), _lib.ptr(ws), wsb, _lib.ptr(info), st)
def solve(tr=0):
    _lib.call("mf_btd_solve", dt, B, B, n, d, _lib.ptr(ld), _lib.ptr(ls), _lib.ptr(rhs), _lib.ptr(out), tr, _lib.ptr(ws2), wss, st)

def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters
tc = timeit(chol)
if a.op == "chol":
    print(f"B={B} T={n} d={d} {a.dtype}: cholesky {tc*1e3:.1f} us"); sys.exit(0)
ts, tst = timeit(solve), timeit(lambda: solve(1))
err = float((ld.double() - ldiag).abs().max())
bc, bs = B * n * 4 * d * d * esz, B * n * (2 * d * d + 2 * d) * esz
print(f"B={B} T={n} d={d} {a.dtype} parallel_ws={wsb>0}: cholesky {tc*1e3:.1f} us = {bc/tc/1e6:.1f} GB/s | solve {ts*1e3:.1f} us = {bs/ts/1e6:.1f} GB/s | "
      f"solve^T {tst*1e3:.1f} us | max|L-L_exact| {err:.2e}")
