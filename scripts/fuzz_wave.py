"""Randomised parity of everything the wave kernels run (16 <= d <= 32; csrc/mf_wave.hpp, mf_wave_ops.hpp) against the numpy oracle:
log-likelihood (m <= 4 outputs, shared precision), posterior chain, cholesky, solve (both orientations, broadcast right-hand sides),
upper_diagonal_lower, block diagonal / sub-diagonal of the inverse, marginals and covariance blocks; both dtypes, 1 ... 70 series,
1 ... 60 blocks.   python3 scripts/fuzz_wave.py <cases> <seed>"""
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
DEV = "cuda:0"
worst = {"f64": 0.0, "f32": 0.0}
t0 = time.time()


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - b)) / (np.max(np.abs(b)) + 1e-300))


for case in range(n_cases):
    f64 = bool(rng.integers(0, 2))
    dt = torch.float64 if f64 else torch.float32
    d = int(rng.choice([16, 32, int(rng.integers(16, 33))])); m = int(rng.integers(1, 5))
    bsz = int(rng.choice([1, 2, 3, 70])); t = int(rng.integers(2, 40 if bsz == 70 else 61))
    if case % 8 == 5 and bsz != 70:
        t = int(rng.integers(130, 400))          # long enough for the time partitions of solve / marginal_means / the factorisations
    rd = (lambda x: x) if f64 else (lambda x: x.astype(np.float32).astype(np.float64))
    kw = dict(mu0=rng.normal(size=(bsz, d)), chol_p0=np.tril(0.2 * rng.normal(size=(bsz, d, d))) / np.sqrt(d) + np.eye(d),
              a_s=0.6 * np.eye(d) + 0.3 * rng.normal(size=(bsz, t - 1, d, d)) / np.sqrt(d), b_s=0.3 * rng.normal(size=(bsz, t - 1, d)),
              chol_q=np.tril(0.2 * rng.normal(size=(bsz, t - 1, d, d))) / np.sqrt(d) + 0.7 * np.eye(d),
              h=rng.normal(size=(bsz, t, m, d)) / np.sqrt(d), y=rng.normal(size=(bsz, t, m)))
    kw = {k: rd(v) for k, v in kw.items()}
    r = rng.normal(size=(m, m)); chol_r = rd(np.linalg.cholesky(r @ r.T / m + np.eye(m))); r_inv = np.linalg.inv(chol_r @ chol_r.T)
    errs = []
    kf = build_kf(kw, chol_r, dtype=dt)
    ref = O.kf_log_likelihood(**kw, r_inv=r_inv, per_series=True)
    got = (kf._log_likelihood_per_series() + kf._constant_terms(t)).cpu().numpy().reshape(np.shape(ref))
    errs.append(("loglik", rel(got, ref)))
    post = kf.posterior_state_space_model()
    want = O.kf_posterior_ssm(**kw, r_inv=r_inv)
    for name, g_, w_ in zip(("mu0'", "cholP0'", "A'", "b'", "cholQ'"), (post.initial_mean, post.cholesky_initial_covariance, post.state_transitions,
                                                                         post.state_offsets, post.cholesky_process_covariances), want):
        errs.append((name, rel(g_.cpu().numpy(), w_)))
    tt = lambda x: None if x is None else torch.tensor(x, dtype=dt, device=DEV)      # noqa: E731
    diag, sub = O.kf_posterior_precision(kw["chol_p0"], kw["a_s"], kw["chol_q"], kw["h"], r_inv)
    diag, sub = rd(diag), rd(sub)
    ld, ls = O.btd_cholesky(diag, sub)
    chol = mfa.SymmetricBlockTriDiagonal(tt(diag), tt(sub)).cholesky
    errs.append(("chol diag", rel(chol.block_diagonal.cpu().numpy(), np.tril(ld)))); errs.append(("chol sub", rel(chol.block_sub_diagonal.cpu().numpy(), ls)))
    ld, ls = rd(np.tril(ld)), rd(ls)
    low = mfa.LowerTriangularBlockTriDiagonal(tt(ld), tt(ls))
    lead = (2,) if case % 3 == 0 else ()
    rhs = rd(rng.normal(size=lead + (bsz, t, d)))
    errs.append(("solve", rel(low.solve(tt(rhs)).cpu().numpy(), O.btd_solve(ld, ls, rhs))))
    errs.append(("solve^T", rel(low.solve(tt(rhs), transpose_left=True).cpu().numpy(), O.btd_solve(ld, ls, rhs, transpose_left=True))))
    inv_d, inv_s = O.btd_block_diagonal_of_inverse(ld, ls, return_sub=True)
    gd, gs = low._diag_and_sub_of_inverse(want_sub=True)
    errs.append(("inv diag", rel(gd.cpu().numpy(), inv_d))); errs.append(("inv sub", rel(gs.cpu().numpy(), inv_s)))
    u_t, chol_d = mfa.SymmetricBlockTriDiagonal(tt(diag), tt(sub)).upper_diagonal_lower()
    wu, wc = O.btd_upper_diagonal_lower(diag, sub)
    errs.append(("udl U", rel(u_t.block_sub_diagonal.cpu().numpy(), wu))); errs.append(("udl D", rel(chol_d.block_diagonal.cpu().numpy(), np.tril(wc))))
    means, covs = kf.prior_ssm.marginals
    covs2, cross = kf.prior_ssm.covariance_blocks()
    ec = [kw["chol_p0"] @ np.swapaxes(kw["chol_p0"], -1, -2)]
    for k in range(t - 1):
        a, c = kw["a_s"][:, k], kw["chol_q"][:, k]
        ec.append(a @ ec[-1] @ np.swapaxes(a, -1, -2) + c @ np.swapaxes(c, -1, -2))
    ec = np.stack(ec, axis=1)
    errs.append(("means", rel(means.cpu().numpy(), O.ssm_marginal_means(kw["mu0"], kw["a_s"], kw["b_s"]))))
    errs.append(("means alone", rel(kf.prior_ssm.marginal_means.cpu().numpy(), O.ssm_marginal_means(kw["mu0"], kw["a_s"], kw["b_s"]))))
    errs.append(("covs", rel(covs.cpu().numpy(), ec))); errs.append(("cross", rel(cross.cpu().numpy(), O.ssm_subsequent_covariances(kw["a_s"], ec))))
    key = "f64" if f64 else "f32"
    tol = 2e-8 if f64 else 2e-2
    bad = [(n_, e_) for n_, e_ in errs if not e_ < tol]
    assert not bad, (case, dict(d=d, m=m, bsz=bsz, t=t, dtype=key), bad)
    worst[key] = max(worst[key], max(e_ for _, e_ in errs))
mfa.check_errors()
print(f"{n_cases} cases in {time.time() - t0:.0f} s; worst relative deviation (max-norm per tensor):", {k: f"{v:.2e}" for k, v in worst.items()})
print("fuzz ok")
