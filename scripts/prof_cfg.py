"""Run one configuration a few times (for rocprofv3 --kernel-trace): python3 scripts/prof_cfg.py head|c2|c4ll|c4kl [iters]"""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import markovflow_amd as mfa
from markovflow_amd import synthetic

which, iters = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
if which == "head":
    kf = synthetic.kalman_filter_from(synthetic.make_ssm(1024, 10000, (5, 5), dtype=torch.float64, device=dev))
    fn = kf.log_likelihood
elif which in ("c3chol", "c3solve"):
    n, d = 100000, 6
    g = torch.Generator(device=dev); g.manual_seed(3)
    ld = torch.tril(0.3 * torch.randn(1, n, d, d, dtype=torch.float64, device=dev, generator=g))
    ld = ld - torch.diag_embed(torch.diagonal(ld, dim1=-2, dim2=-1)) + torch.diag_embed(
        1 + torch.rand(1, n, d, dtype=torch.float64, device=dev, generator=g))
    ls = 0.3 * torch.randn(1, n - 1, d, d, dtype=torch.float64, device=dev, generator=g)
    diag = ld @ ld.transpose(-1, -2)
    diag[:, 1:] += ls @ ls.transpose(-1, -2)
    sub = ls @ ld[:, :-1].transpose(-1, -2)
    sym = mfa.SymmetricBlockTriDiagonal(diag.float().contiguous(), sub.float().contiguous())
    rhs = torch.randn(1, n, d, dtype=torch.float32, device=dev, generator=g)
    chol = sym.cholesky
    fn = (lambda: sym.cholesky) if which == "c3chol" else (lambda: chol.solve(rhs))
elif which == "c2":
    kf = synthetic.kalman_filter_from(synthetic.make_ssm(256, 4096, (3, 3), dtype=torch.float64, device=dev))
    fn = kf.log_likelihood
else:
    kf = synthetic.kalman_filter_from(synthetic.make_ssm(512, 1000, (5, 5, 5), output_dim=3, dtype=torch.float64, device=dev))
    if which == "c4ll":
        fn = kf.log_likelihood
    else:
        post = kf.posterior_state_space_model()
        fn = lambda: post.kl_divergence(kf.prior_ssm)
torch.cuda.synchronize()
marker = torch.zeros(1, device=dev)
for _ in range(iters):
    marker.add_(1.0)          # one recognisable tiny kernel between iterations
    fn()
torch.cuda.synchronize()
