"""Diagnostic (MF_CHECKSUM build): compare what every lane pulled out of LDS with the input tensors."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from markovflow_amd import _lib
dev = torch.device("cuda:0"); dt = torch.float64
B, T, d, P = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
g = torch.Generator(device=dev); g.manual_seed(0)
eye = torch.eye(d, dtype=dt, device=dev)
A = 0.5 * eye + 0.05 * torch.randn(B, T - 1, d, d, dtype=dt, device=dev, generator=g)
cq = torch.tril(0.1 * torch.randn(B, T - 1, d, d, dtype=dt, device=dev, generator=g)) + eye
cp0 = torch.tril(0.1 * torch.randn(B, d, d, dtype=dt, device=dev, generator=g)) + eye
mu0 = torch.randn(B, d, dtype=dt, device=dev, generator=g); b = 0.1 * torch.randn(B, T - 1, d, dtype=dt, device=dev, generator=g)
H = torch.randn(B, T, 1, d, dtype=dt, device=dev, generator=g); y = torch.randn(B, T, 1, dtype=dt, device=dev, generator=g)
ri = torch.tensor([[2.0]], dtype=dt, device=dev)
lib = _lib.load()
wsb = int(lib.mf_kf_loglik_workspace_bytes(B, T, d, 8, P))
ws = torch.zeros(wsb, dtype=torch.uint8, device=dev); out = torch.empty(B, dtype=dt, device=dev)
_lib.call("mf_kf_loglik", dt, B, T, d, 1, _lib.ptr(mu0), _lib.ptr(cp0), _lib.ptr(A), _lib.ptr(b), _lib.ptr(cq), _lib.ptr(H),
          _lib.ptr(y), _lib.ptr(ri), 0, 0.0, _lib.ptr(out), _lib.ptr(ws), wsb, None, P, None, None, _lib.stream_ptr(dev))
torch.cuda.synchronize()
nt = T - 1; L = -(-nt // P); P = -(-nt // L); nb = B * P
wsd = ws.view(torch.float64)
GU = wsd[nb * d * d: 2 * nb * d * d].reshape(nb, d * d).cpu().numpy()
gU = wsd[3 * nb * d * d + nb * d: 3 * nb * d * d + 2 * nb * d].reshape(nb, d).cpu().numpy()
wA = torch.arange(1, d * d + 1, dtype=dt, device=dev).reshape(d, d); wC = torch.tril(wA); wH = torch.arange(1, d + 1, dtype=dt, device=dev)
for s in range(B):
    for c in range(P):
        t0, t1 = c * L, min((c + 1) * L, nt)
        exp = [float((A[s, t0:t1] * wA).sum()), float((cq[s, t0:t1] * wC).sum()), float(b[s, t0:t1].sum()),
               float((H[s, t0 + 1:t1 + 1, 0] * wH).sum()), float(y[s, t0 + 1:t1 + 1].sum())]
        got = [GU[s * P + c, 0], GU[s * P + c, 1], GU[s * P + c, 2], gU[s * P + c, 0], gU[s * P + c, 1]]
        bad = [n for n, e, gg in zip("ACbHy", exp, got) if abs(e - gg) > 1e-9 * (1 + abs(e))]
        if bad or (s < 2 and c < 2): print(f"series {s} chunk {c}: wrong streams {bad}  (A got {got[0]:.6f} exp {exp[0]:.6f})")
