"""Randomised log-likelihood parity for 10 <= d <= 64 in both dtypes (row, wave, panel and LDS-tile kernels; m up to 32; explicit
time partitions) against the numpy oracle; in fp32 at d > 32 (every third such case) also the operators that run on the panel
kernels since round 6: the precision assembly (prior and posterior), cholesky of the posterior precision, the posterior chain."""
import os, sys, time
import numpy as np
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import markovflow_amd as mfa
from oracle import numpy_oracle as O
from test_gpu_kalman import build_kf

n_cases, seed = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
worst = {"f64": 0.0, "f32": 0.0}
t0 = time.time()
for case in range(n_cases):
    f64 = bool(rng.integers(0, 2))
    d = int(rng.integers(10, 65)); m = int(rng.integers(1, 9)) if rng.random() < 0.7 else int(rng.integers(9, 33))
    bsz = int(rng.integers(1, 4)); t = int(rng.integers(2, 60)) if rng.random() < 0.8 else int(rng.integers(60, 400))
    kw = dict(mu0=rng.normal(size=(bsz, d)), chol_p0=np.tril(0.2 * rng.normal(size=(bsz, d, d))) / np.sqrt(d) + np.eye(d),
              a_s=0.6 * np.eye(d) + 0.3 * rng.normal(size=(bsz, t - 1, d, d)) / np.sqrt(d), b_s=0.3 * rng.normal(size=(bsz, t - 1, d)),
              chol_q=np.tril(0.2 * rng.normal(size=(bsz, t - 1, d, d))) / np.sqrt(d) + 0.7 * np.eye(d),
              h=rng.normal(size=(bsz, t, m, d)) / np.sqrt(d), y=rng.normal(size=(bsz, t, m)))
    r = rng.normal(size=(m, m)); cov = r @ r.T / m + np.eye(m)
    ref = sum(O.kf_log_likelihood(**{k: v[s] for k, v in kw.items()}, r_inv=np.linalg.inv(cov)) for s in range(bsz))
    kf = build_kf(kw, np.linalg.cholesky(cov), dtype=torch.float64 if f64 else torch.float32)
    kf._chunks = 0 if rng.random() < 0.5 else int(rng.integers(1, 12))
    got = float(kf.log_likelihood())
    key = "f64" if f64 else "f32"
    worst[key] = max(worst[key], abs(got - ref) / abs(ref))
    if not f64 and d > 32 and case % 3 == 0:
        rel = lambda a, b: float(np.max(np.abs(np.asarray(a) - b)) / (np.max(np.abs(b)) + 1e-300))   # noqa: E731
        r32 = {k: v.astype(np.float32).astype(np.float64) for k, v in kw.items()}
        chol_r = np.linalg.cholesky(cov).astype(np.float32).astype(np.float64)
        r_inv = np.linalg.inv(chol_r @ chol_r.T)
        kf32 = build_kf(r32, chol_r, dtype=torch.float32)
        errs = []
        pd, ps = O.ssm_precision(r32["chol_p0"], r32["a_s"], r32["chol_q"])
        prec = kf32.prior_ssm.precision
        errs += [rel(prec.block_diagonal.cpu().numpy(), pd), rel(prec.block_sub_diagonal.cpu().numpy(), ps)]
        qd, qs = O.kf_posterior_precision(r32["chol_p0"], r32["a_s"], r32["chol_q"], r32["h"], r_inv)
        post = kf32._k_inv_post
        errs += [rel(post.block_diagonal.cpu().numpy(), qd), rel(post.block_sub_diagonal.cpu().numpy(), qs)]
        ld, ls = O.btd_cholesky(qd, qs)
        tt = lambda x: torch.tensor(x, dtype=torch.float32, device="cuda:0")                           # noqa: E731
        chol = mfa.SymmetricBlockTriDiagonal(tt(qd), tt(qs)).cholesky
        errs += [rel(chol.block_diagonal.cpu().numpy(), np.tril(ld)), rel(chol.block_sub_diagonal.cpu().numpy(), ls)]
        chain = kf32.posterior_state_space_model()
        want = O.kf_posterior_ssm(**r32, r_inv=r_inv)
        for g_, w_ in zip((chain.initial_mean, chain.cholesky_initial_covariance, chain.state_transitions, chain.state_offsets,
                           chain.cholesky_process_covariances), want):
            errs.append(rel(g_.cpu().numpy(), w_))
        assert max(errs) < 2e-2, (case, dict(d=d, m=m, bsz=bsz, t=t), errs)
        worst["ops f32"] = max(worst.get("ops f32", 0.0), max(errs))
print(f"{n_cases} cases in {time.time() - t0:.0f} s; worst relative deviation:", {k: f"{v:.2e}" for k, v in worst.items()})
assert worst["f64"] < 1e-9 and worst["f32"] < 5e-3, worst
print("fuzz ok")
