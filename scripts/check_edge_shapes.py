"""Sanity sweep of unusual shapes through the C ABI: many short series, P = 1 paths, partition invariance."""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from test_gpu_kalman import random_ssm, loglik_with_chunks
rng = np.random.default_rng(5)
for d, m, bsz, t in [(6, 1, 70000, 20), (9, 3, 70000, 12), (3, 1, 5000, 9), (7, 2, 300, 700), (8, 4, 33, 1500), (2, 1, 1, 50000), (9, 1, 1, 20000)]:
    kw = random_ssm(rng, (bsz,), t, d, m, well=True) if bsz <= 5000 else None
    if kw is None:      # big batches: tile a small set of series
        base = random_ssm(rng, (50,), t, d, m, well=True)
        kw = {k: np.tile(v, (bsz // 50,) + (1,) * (v.ndim - 1)) for k, v in base.items()}
    r_inv = np.eye(m) * 2.0
    ref = loglik_with_chunks(kw, r_inv, 1)
    for chunks in (0, 2, 5):
        got = loglik_with_chunks(kw, r_inv, chunks)
        err = np.max(np.abs(got - ref) / np.abs(ref))
        assert err < 1e-9, (d, m, bsz, t, chunks, err)
    print("ok", d, m, bsz, t, "max rel dev", err)
