"""Times StateSpaceModel.kl_divergence forward + backward (both chains require gradients) and the marginals' backward at two
shapes: python3 scripts/bench_kl_grad.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import markovflow_amd as mfa
from markovflow_amd import synthetic

def timeit(fn, iters=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e3

SHAPES = [(16384, 500, (5, 5), 1), (512, 1000, (5, 5, 5), 3), (512, 1000, (5, 5, 5, 5), 4), (512, 1000, (5, 5, 5, 5, 5), 1)]
if len(sys.argv) > 1:          # python3 scripts/bench_kl_grad.py 0   -> the first shape only (for rocprofv3)
    SHAPES = [SHAPES[int(sys.argv[1])]]
for (b, t, comp, m) in SHAPES:
    inp = synthetic.make_ssm(b, t, comp, output_dim=m, dtype=torch.float64, device="cuda")
    kf = synthetic.kalman_filter_from(inp)
    with torch.no_grad():
        post = kf.posterior_state_space_model()
    q = post.create_trainable_copy()
    p = kf.prior_ssm
    d = sum((c + 1) // 2 for c in comp)
    fwd = timeit(lambda: q.kl_divergence(p).sum())
    def both():
        for v in q.trainable_variables: v.grad = None
        q.kl_divergence(p).sum().backward()
    tot = timeit(both)
    def marg():
        for v in q.trainable_variables: v.grad = None
        mm, cc = q.marginals
        (mm.sum() + cc.sum()).backward()
    tm = timeit(marg)
    print(f"B={b} T={t} d={d} fp64: kl_divergence forward {fwd:.2f} ms, forward+backward {tot:.2f} ms; marginals forward+backward {tm:.2f} ms")
