import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from markovflow_amd.block_tri_diag import SymmetricBlockTriDiagonal as S
dev = "cuda:0"
d, n = int(sys.argv[1]), int(sys.argv[2]); dt = torch.float64 if sys.argv[3] == "f64" else torch.float32
g = torch.Generator(device=dev); g.manual_seed(0)
ld = torch.tril(0.3 * torch.randn(1, n, d, d, dtype=torch.float64, device=dev, generator=g))
ld = ld - torch.diag_embed(torch.diagonal(ld, dim1=-2, dim2=-1)) + torch.diag_embed(1 + torch.rand(1, n, d, dtype=torch.float64, device=dev, generator=g))
ls = 0.3 * torch.randn(1, n - 1, d, d, dtype=torch.float64, device=dev, generator=g)
diag = ld @ ld.transpose(-1, -2); diag[:, 1:] += ls @ ls.transpose(-1, -2)
sub = ls @ ld[:, :-1].transpose(-1, -2)
ch = S(diag.to(dt).contiguous(), sub.to(dt).contiguous()).cholesky
e = (ch.block_diagonal.double() - ld).abs().amax(dim=(-1, -2))[0]
bad = torch.nonzero(~(e < 1e-3)).flatten().tolist()
print(d, n, sys.argv[3], "PAR_LEN", os.environ.get("MF_BTD_PAR_LEN"), "max err", float(e[torch.isfinite(e)].max()) if torch.isfinite(e).any() else None, "bad blocks:", bad[:12], "count", len(bad))
