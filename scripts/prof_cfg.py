"""Run one configuration a few times (for rocprofv3 --kernel-trace): python3 scripts/prof_cfg.py head|c2|c4ll|c4kl [iters]"""
import sys
import torch
sys.path.insert(0, "/root/repo")
import markovflow_amd as mfa
from markovflow_amd import synthetic

which, iters = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
if which == "head":
    kf = synthetic.kalman_filter_from(synthetic.make_ssm(1024, 10000, (5, 5), dtype=torch.float64, device=dev))
    fn = kf.log_likelihood
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
