"""Does the streamed backward's time depend on where its arrays start?  (All large tensors from torch's allocator are 2-MB aligned,
so a lane's reads of A, cholQ and its writes of g_A, g_cholQ at the same relative offset may land on the same HBM channel.)
Times mf_kf_loglik_grad_streamed at the headline shape with the workspace and the gradient tensors shifted by a few byte offsets."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from markovflow_amd import _lib, synthetic

dev = torch.device("cuda:0")
B, T, d, m = 1024, 10000, 6, 1
inp = synthetic.make_ssm(B, T, (5, 5), dtype=torch.float64, device=dev)
kf = synthetic.kalman_filter_from(inp)
mu0, cp0, a, b, cq = kf.prior_ssm._flat_params()
h, y, r_inv, per_step = kf._expanded()
lib = _lib.load()
fws = torch.empty(int(lib.mf_kf_loglik_workspace_bytes(B, T, d, 8, 0)), dtype=torch.uint8, device=dev)
val = torch.empty(B, dtype=torch.float64, device=dev)
info = _lib.new_info(dev)
ins = [mu0, cp0, a, b, cq, h, y, r_inv]
_lib.call("mf_kf_loglik", torch.float64, B, T, d, m, *[_lib.ptr(x) for x in ins], 0, 0.0, _lib.ptr(val), _lib.ptr(fws), fws.numel(),
          _lib.ptr(info), 0, None, None, _lib.stream_ptr(dev))
path, pf, lf = ctypes.c_int(0), ctypes.c_int64(0), ctypes.c_int64(0)
lib.mf_kf_loglik_plan(B, T, d, m, 0, 8, 0, 1, ctypes.byref(path), ctypes.byref(pf), ctypes.byref(lf))
wsb = int(lib.mf_kf_loglik_grad_streamed_workspace_bytes(B, T, d, m, 0, 8, 0))
w = torch.ones(B, dtype=torch.float64, device=dev)
PAD = 1 << 22


def shifted(shape, off):
    n = 1
    for s in shape:
        n *= s
    raw = torch.empty(n * 8 + PAD, dtype=torch.uint8, device=dev)
    return raw, raw[off:off + n * 8].view(torch.float64).reshape(shape)


def run(ws_off, out_offs, use_fwd=True):
    wsr = torch.empty(wsb + PAD, dtype=torch.uint8, device=dev)
    ws = wsr[ws_off:]
    keep, outs = [], []
    for shp, off in zip([mu0.shape, cp0.shape, a.shape, b.shape, cq.shape, h.shape, y.shape, (B, T, m, m)], out_offs):
        raw, t = shifted(tuple(shp), off)
        keep.append(raw); outs.append(t)
    ts = []
    for i in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.call("mf_kf_loglik_grad_streamed", torch.float64, B, T, d, m, *[_lib.ptr(x) for x in ins], 0, _lib.ptr(w),
                  *[_lib.ptr(x) for x in outs], _lib.ptr(ws), wsb, _lib.ptr(info), 0, _lib.ptr(fws) if use_fwd else None,
                  int(pf.value) if use_fwd else 0, int(lf.value) if use_fwd else 0, None, None, _lib.stream_ptr(dev))
        e1.record(); torch.cuda.synchronize()
        if i: ts.append(e0.elapsed_time(e1))
    return min(ts)


print("input bases mod 2 MiB:", [x.data_ptr() % (1 << 21) for x in (a, cq, b, h, y)])
for ws_off in (0, 256, 4096, 65536, 1 << 20):
    print(f"ws +{ws_off:8d}, outputs +0: {run(ws_off, [0] * 8):.2f} ms")
for k in (256, 1024, 4096, 16384, 65536, 262144):
    offs = [0, 0, 1 * k, 3 * k, 2 * k, 5 * k, 6 * k, 7 * k]
    print(f"ws +4096, outputs at multiples of {k:7d}: {run(4096, offs):.2f} ms")
