"""Forward / backward time of KalmanFilter.log_likelihood with gradients w.r.t. every model tensor."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import markovflow_amd as mfa
from markovflow_amd import synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024); ap.add_argument("--T", type=int, default=10000); ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--chunks", type=int, default=0); ap.add_argument("--orders", default="5,5"); ap.add_argument("--dtype", default="float64")
a = ap.parse_args()
dev = torch.device("cuda:0")
orders = tuple(int(x) for x in a.orders.split(","))
dt = getattr(torch, a.dtype)
inp = synthetic.make_ssm(a.batch, a.T, orders, dtype=dt, device=dev)
kf0 = synthetic.kalman_filter_from(inp)
p = kf0.prior_ssm
leaves = [t.detach().clone().requires_grad_(True) for t in (p.initial_mean, p.cholesky_initial_covariance, p.state_transitions,
                                                            p.state_offsets, p.cholesky_process_covariances)]
ssm = mfa.StateSpaceModel(*leaves)
kf = mfa.KalmanFilter(ssm, kf0.emission, kf0.observations, kf0._chol_obs_covariance)
kf._chunks = a.chunks
def timed(fn):
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = fn(); e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1), out
for i in range(a.iters + 1):
    for l in leaves: l.grad = None
    tf, ll = timed(kf.log_likelihood)
    tb, _ = timed(ll.backward)
    if i: print(f"B={a.batch} T={a.T} Matern orders {orders} {a.dtype}: forward {tf:.2f} ms, backward {tb:.2f} ms")
