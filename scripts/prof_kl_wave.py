import sys, torch
sys.path.insert(0, '/root/repo')
import markovflow_amd as mfa
from markovflow_amd import synthetic
dev='cuda:0'
def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/it
for d in (16, 32):
    kf = synthetic.kalman_filter_from(synthetic.make_dense_ssm(512, 1000, d, 1, dtype=torch.float64, device=dev))
    post = kf.posterior_state_space_model(); prior = kf.prior_ssm
    print(d, 'kl', t(lambda: post.kl_divergence(prior)))
    print(d, 'moments(want_sub)', t(lambda: post._moments(want_sub=True)))
    print(d, 'prior.marginal_means', t(lambda: prior.marginal_means))
    m1 = post._moments(want_sub=True)
    print(d, 'sub', t(lambda: (prior.marginal_means - m1[0]).reshape(-1, 1000, d).contiguous()))
