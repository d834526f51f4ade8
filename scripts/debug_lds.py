import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_gpu_kalman import random_ssm, loglik_with_chunks
from oracle import numpy_oracle as O
rng = np.random.default_rng(5)
for (B, T, d) in [(1, 15, 2), (4, 15, 2), (4, 9, 2), (4, 41, 6)]:
    kw = random_ssm(rng, (B,), T, d, 1)
    ref = O.kf_log_likelihood(**kw, r_inv=np.array([[2.0]]), per_series=True)
    cst = -0.5*np.log(2*np.pi)*T + 0.5*T*np.log(2.0)
    for ch in (1, 2, 4):
        try:
            got = loglik_with_chunks(kw, np.array([[2.0]]), ch) + cst
        except AssertionError:
            print(f"B={B} T={T} d={d} chunks={ch}: info set"); continue
        err = np.abs(got-ref)/np.abs(ref)
        print(f"B={B} T={T} d={d} chunks={ch}: rel err per series {np.array2string(err, precision=1)}")
