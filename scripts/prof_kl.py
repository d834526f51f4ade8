import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from markovflow_amd import synthetic
dev = torch.device("cuda:0")
inp = synthetic.make_ssm(16384, 500, (5, 5), dtype=torch.float64, device=dev)
kf = synthetic.kalman_filter_from(inp); ssm = kf.prior_ssm
post = kf.posterior_state_space_model()
def t(name, fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); print(f"{name:40s} {(time.perf_counter()-t0)/n*1e3:8.3f} ms"); return r
mc = t("marginal_covariances", lambda: post.marginal_covariances)
p2 = t("precision", lambda: ssm.precision)
sc = t("subsequent_covariances", lambda: post.subsequent_covariances(mc))
t("trace terms", lambda: torch.sum(p2.block_diagonal * mc, dim=(-3, -2, -1)) + 2.0 * torch.sum(p2.block_sub_diagonal * sc, dim=(-3, -2, -1)))
md = t("mean diff", lambda: ssm.marginal_means - post.marginal_means)
ch = t("precision.cholesky", lambda: p2.cholesky)
t("dense_mult T", lambda: ch.dense_mult(md, transpose_left=True))
t("log_det_precision x2", lambda: ssm.log_det_precision() + post.log_det_precision())
t("kl total", lambda: post.kl_divergence(ssm))
