import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_gpu_kalman import random_ssm, loglik_with_chunks
from oracle import numpy_oracle as O
rng = np.random.default_rng(71892305)
random_ssm(rng, (8,), 50, 2, 1)
kw = random_ssm(rng, (8,), 200, 6, 1)
print("min |diag cholQ|", np.abs(np.einsum("...ii->...i", kw["chol_q"])).min())
ref = O.kf_log_likelihood(**kw, r_inv=np.array([[2.0]]), per_series=True)
cst = -0.5*np.log(2*np.pi)*200 + 0.5*200*np.log(2.0)
for ch in (1, 2, 10, 50):
    for dt in (torch.float32, torch.float64):
        try:
            got = loglik_with_chunks(kw, np.array([[2.0]]), ch, dtype=dt) + cst
            print(ch, dt, np.abs(got-ref)/np.abs(ref))
        except AssertionError as e:
            print(ch, dt, "info set")
