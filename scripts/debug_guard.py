import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from markovflow_amd import _lib
dev = "cuda:0"
G = 4096
SENT = 12345.678
class Guard:
    def __init__(self, shape, dtype=torch.float64, fill=None):
        n = int(np.prod(shape))
        self.buf = torch.full((n + 2 * G,), SENT, dtype=dtype, device=dev)
        self.view = self.buf[G:G + n].view(shape)
        if fill is not None:
            self.view.copy_(fill)
        self.n = n
    def ok(self):
        return bool((self.buf[:G] == SENT).all() and (self.buf[G + self.n:] == SENT).all())
def ws_guard(nbytes):
    if not nbytes:
        return None
    n = (nbytes + 7) // 8
    return Guard((n,))
lib = _lib.load()
st = _lib.stream_ptr(torch.device(dev))
for B in (1, 2, 5):
  for n in (64, 79, 80, 100, 257):
    for d in (1, 3, 6, 9):
        x = torch.randn(B, n, d, d, dtype=torch.float64, device=dev)
        spd = x @ x.transpose(-1, -2) + 4 * d * torch.eye(d, dtype=torch.float64, device=dev)
        diag = Guard((B, n, d, d), fill=spd)
        sub = Guard((B, n - 1, d, d), fill=0.3 * torch.randn(B, n - 1, d, d, dtype=torch.float64, device=dev))
        ld, ls = Guard((B, n, d, d)), Guard((B, n - 1, d, d))
        wb = int(lib.mf_btd_cholesky_workspace_bytes(B, n, d, 8)); ws = ws_guard(wb)
        info = _lib.new_info(torch.device(dev))
        _lib.call("mf_btd_cholesky", torch.float64, B, n, d, _lib.ptr(diag.view), _lib.ptr(sub.view), _lib.ptr(ld.view), _lib.ptr(ls.view),
                  _lib.ptr(ws.view) if ws else None, wb, _lib.ptr(info), st)
        torch.cuda.synchronize()
        bad = [nm for nm, g in (("diag", diag), ("sub", sub), ("ld", ld), ("ls", ls), ("ws", ws)) if g is not None and not g.ok()]
        if bad: print("cholesky", B, n, d, "OOB:", bad)
        od, os_ = Guard((B, n, d, d)), Guard((B, n - 1, d, d))
        wb = int(lib.mf_btd_diag_of_inverse_workspace_bytes(B, n, d, 8)); ws = ws_guard(wb)
        _lib.call("mf_btd_diag_of_inverse", torch.float64, B, n, d, _lib.ptr(ld.view), _lib.ptr(ls.view), _lib.ptr(od.view), _lib.ptr(os_.view),
                  _lib.ptr(ws.view) if ws else None, wb, st)
        torch.cuda.synchronize()
        bad = [nm for nm, g in (("ld", ld), ("ls", ls), ("od", od), ("os", os_), ("ws", ws)) if g is not None and not g.ok()]
        if bad: print("diag_of_inverse", B, n, d, "OOB:", bad)
        for tr in (0, 1):
            rhs, out = Guard((B, n, d), fill=torch.randn(B, n, d, dtype=torch.float64, device=dev)), Guard((B, n, d))
            wb = int(lib.mf_btd_solve_workspace_bytes(B, B, n, d, 8)); ws = ws_guard(wb)
            _lib.call("mf_btd_solve", torch.float64, B, B, n, d, _lib.ptr(ld.view), _lib.ptr(ls.view), _lib.ptr(rhs.view), _lib.ptr(out.view), tr,
                      _lib.ptr(ws.view) if ws else None, wb, st)
            torch.cuda.synchronize()
            bad = [nm for nm, g in (("rhs", rhs), ("out", out), ("ws", ws)) if g is not None and not g.ok()]
            if bad: print("solve", tr, B, n, d, "OOB:", bad)
        # no-sub cholesky
        wb = int(lib.mf_btd_cholesky_workspace_bytes(B, n, d, 8)); ws = ws_guard(wb)
        _lib.call("mf_btd_cholesky", torch.float64, B, n, d, _lib.ptr(diag.view), None, _lib.ptr(ld.view), None,
                  _lib.ptr(ws.view) if ws else None, wb, _lib.ptr(info), st)
        torch.cuda.synchronize()
        bad = [nm for nm, g in (("diag", diag), ("ld", ld), ("ws", ws)) if g is not None and not g.ok()]
        if bad: print("cholesky nosub", B, n, d, "OOB:", bad)
        u_t, cd, ef, mp, cdi = Guard((B, n - 1, d, d)), Guard((B, n, d, d)), Guard((B, n, d, d)), Guard((B, n, d, d)), Guard((B, n, d, d))
        wb = int(lib.mf_btd_udl_workspace_bytes(B, n, d, 8)); ws = ws_guard(wb)
        _lib.call("mf_btd_udl", torch.float64, B, n, d, _lib.ptr(diag.view), _lib.ptr(sub.view), _lib.ptr(u_t.view), _lib.ptr(cd.view), None, None, None, 0,
                  _lib.ptr(ws.view) if ws else None, wb, _lib.ptr(info), st)
        torch.cuda.synchronize()
        bad = [nm for nm, g in (("u_t", u_t), ("cd", cd), ("ws", ws)) if g is not None and not g.ok()]
        if bad: print("udl", B, n, d, "OOB:", bad)
print("done")
