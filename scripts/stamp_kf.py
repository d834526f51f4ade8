"""Diagnostic (MF_STAMP build): where does a loop iteration of kf_chunk_lds_kernel spend its cycles?"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from markovflow_amd import _lib
dev = torch.device("cuda:0"); dt = torch.float64
B, T, d = 1024, 10000, 6
g = torch.Generator(device=dev); g.manual_seed(0)
eye = torch.eye(d, dtype=dt, device=dev)
A = 0.9 * eye + 0.05 * torch.randn(B, T - 1, d, d, dtype=dt, device=dev, generator=g)
cq = torch.tril(0.1 * torch.randn(B, T - 1, d, d, dtype=dt, device=dev, generator=g)) + 0.5 * eye
cp0 = torch.tril(0.1 * torch.randn(B, d, d, dtype=dt, device=dev, generator=g)) + eye
mu0 = torch.randn(B, d, dtype=dt, device=dev, generator=g); b = 0.1 * torch.randn(B, T - 1, d, dtype=dt, device=dev, generator=g)
H = torch.randn(B, T, 1, d, dtype=dt, device=dev, generator=g); y = torch.randn(B, T, 1, dtype=dt, device=dev, generator=g)
ri = torch.tensor([[11.0]], dtype=dt, device=dev)
lib = _lib.load()
wsb = int(lib.mf_kf_loglik_workspace_bytes(B, T, d, 8, 0))
ws = torch.zeros(wsb, dtype=torch.uint8, device=dev); out = torch.empty(B, dtype=dt, device=dev)
for it in range(3):
    _lib.call("mf_kf_loglik", dt, B, T, d, 1, _lib.ptr(mu0), _lib.ptr(cp0), _lib.ptr(A), _lib.ptr(b), _lib.ptr(cq), _lib.ptr(H),
              _lib.ptr(y), _lib.ptr(ri), 0, 0.0, _lib.ptr(out), _lib.ptr(ws), wsb, None, 0, None, None, _lib.stream_ptr(dev))
torch.cuda.synchronize()
P = 64; nb = B * P
wsd = ws.view(torch.float64)
GU = wsd[nb * d * d: 2 * nb * d * d].reshape(nb, d * d)
st = GU[::64, :3].cpu()          # lane 0 of every wave
tot = st.sum(1)
print("per-wave mean cycles (s_memtime ticks): wait %.0f  io(LDS read + DMA issue) %.0f  compute %.0f  total %.0f" % (st[:,0].mean(), st[:,1].mean(), st[:,2].mean(), tot.mean()))
print("per step: wait %.0f io %.0f compute %.0f" % tuple((st.mean(0) / 157).tolist()))
print("min/max total over waves", tot.min().item(), tot.max().item())
