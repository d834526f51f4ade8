"""Randomised differential campaign GPU (through the Python mirror / C ABI) vs the numpy oracle: random shapes across the
serial / parallel-in-time thresholds, ragged chunk tails, every state dimension 1..9 (or the range given), m 1..3 (1..4), explicit chunk counts."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import markovflow_amd as mfa
from oracle import numpy_oracle as O
from test_gpu_kalman import random_ssm, loglik_with_chunks, build_kf
from test_gpu_posterior_streamed import posterior_chain_abi
from test_gpu_grad_streamed import dense_autograd, grad_streamed_abi

tt = lambda x: torch.tensor(np.ascontiguousarray(x), dtype=torch.float64, device="cuda:0")   # noqa: E731
nn = lambda x: x.detach().cpu().numpy()                                                          # noqa: E731


def run(n_cases: int, seed: int, dmin: int = 1, dmax: int = 9) -> dict:
  rng = np.random.default_rng(seed)
  worst = dict(ll=0.0, post=0.0, post_chain=0.0, grad=0.0, chol=0.0, solve=0.0, covs=0.0, kl=0.0)
  for case in range(n_cases):
      d = int(rng.integers(dmin, dmax + 1)); m = int(rng.integers(1, 5 if dmax > 9 else 4)); bsz = int(rng.integers(1, 5))
      t = int(rng.choice([2, 3, 5, 8, 9, 17, 63, 64, 65, 71, 127, 128, 130, 200, 257, 400]))
      kw = random_ssm(rng, (bsz,), t, d, m, well=True)
      r = rng.normal(size=(m, m)); cov = r @ r.T + np.eye(m); r_inv = np.linalg.inv(cov)
      ref = np.array([O.kf_log_likelihood(**{k: v[s] for k, v in kw.items()}, r_inv=r_inv) for s in range(bsz)])
      cst = -0.5 * np.log(2 * np.pi) * m * t + 0.5 * t * np.linalg.slogdet(r_inv)[1]
      for chunks in (0, int(rng.integers(1, max(2, t)))):
          got = loglik_with_chunks(kw, r_inv, chunks) + cst
          worst["ll"] = max(worst["ll"], float(np.max(np.abs(got - ref) / np.abs(ref))))
      kf = build_kf(kw, np.linalg.cholesky(cov))
      post = kf.posterior_state_space_model()
      means, covs = post.marginals
      for s in range(bsz):
          o = O.kf_posterior_ssm(**{k: v[s] for k, v in kw.items()}, r_inv=r_inv)
          om = O.ssm_marginal_means(o[0], o[2], o[3]); oc = O.ssm_marginal_covariances(o[1], o[2], o[4])
          worst["post"] = max(worst["post"], float(np.max(np.abs(nn(means)[s] - om)) / (1 + np.max(np.abs(om)))),
                              float(np.max(np.abs(nn(covs)[s] - oc)) / (1 + np.max(np.abs(oc)))))
      if d <= 6 and m <= 3 and t >= 2:
          # the streamed, time-partitioned posterior chain (csrc/mf_post_lds.hpp) through the C ABI: automatic and a random
          # explicit partition, all five tensors of every series
          for chunks in (0, int(rng.integers(1, max(2, t)))):
              got5 = posterior_chain_abi(kw, r_inv, chunks)
              for s in range(bsz):
                  o = O.kf_posterior_ssm(**{k: v[s] for k, v in kw.items()}, r_inv=r_inv)
                  for g, w in zip(got5, o):
                      worst["post_chain"] = max(worst["post_chain"], float(np.max(np.abs(g[s] - w)) / (1 + np.max(np.abs(w)))))
      if d <= 6 and m <= 3 and 3 <= t <= 71:
          # the streamed backward of log_likelihood (csrc/mf_grad_lds.hpp) through the C ABI against dense autograd: its own five
          # passes on a random partition, and three passes from the summaries of a forward evaluation on another random partition
          w = rng.uniform(0.5, 1.5, size=bsz)
          want, _ = dense_autograd(kw, r_inv, w, False)
          for chunks, fwd in ((int(rng.integers(2, t)), None), (int(rng.integers(2, t)), int(rng.integers(2, t)))):
              got8 = grad_streamed_abi(kw, r_inv, w, chunks, fwd_chunks=fwd, strict=False)
              for g, ref in zip(got8[:7], want):
                  worst["grad"] = max(worst["grad"], float(np.max(np.abs(g - ref)) / (1 + np.max(np.abs(ref)))))
      prior = kf.prior_ssm
      pc, ps = prior.covariance_blocks()
      for s in range(bsz):                                   # every series, not only the first
          oc = O.ssm_marginal_covariances(kw["chol_p0"][s], kw["a_s"][s], kw["chol_q"][s])
          worst["covs"] = max(worst["covs"], float(np.max(np.abs(nn(pc)[s] - oc)) / (1 + np.max(np.abs(oc)))),
                              float(np.max(np.abs(nn(ps)[s] - O.ssm_subsequent_covariances(kw["a_s"][s], oc))) / (1 + np.max(np.abs(oc)))))
      prec = kf._k_inv_post
      ch = prec.cholesky
      dense = nn(prec.to_dense())
      lref = np.linalg.cholesky(dense)
      worst["chol"] = max(worst["chol"], float(np.max(np.abs(nn(ch.to_dense()) - lref)) / (1 + np.max(np.abs(lref)))))
      rhs = rng.normal(size=(bsz, t, d))
      for tr in (False, True):
          sol = nn(ch.solve(tt(rhs), transpose_left=tr)).reshape(bsz, -1)
          want = np.stack([np.linalg.solve(lref[s].T if tr else lref[s], rhs[s].reshape(-1)) for s in range(bsz)])
          worst["solve"] = max(worst["solve"], float(np.max(np.abs(sol - want)) / (1 + np.max(np.abs(want)))))
      kl = nn(post.kl_divergence(prior))
      for s in range(bsz):
          o = O.kf_posterior_ssm(**{k: v[s] for k, v in kw.items()}, r_inv=r_inv)
          klref = O.ssm_kl_divergence(o, (kw["mu0"][s], kw["chol_p0"][s], kw["a_s"][s], kw["b_s"][s], kw["chol_q"][s]))
          worst["kl"] = max(worst["kl"], float(abs(kl[s] - klref) / (1 + abs(klref))))
  return worst


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    t0 = time.time()
    # python3 scripts/fuzz_parity.py <cases> <seed> [dmin dmax]   (10 15: the row-kernel-only dimensions; 16 32: the tile engine)
    dmin, dmax = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1, 9)
    worst = run(n_cases, int(sys.argv[2]) if len(sys.argv) > 2 else 2024, dmin, dmax)
    print(f"{n_cases} random cases in {time.time() - t0:.0f} s; worst relative deviations vs the oracle:", {k: f"{v:.2e}" for k, v in worst.items()})
    # (the gradients are compared with autograd through a DENSE inverse of the chain's precision: 1e-6)
    assert all(v < (1e-6 if k == "grad" else 1e-7) for k, v in worst.items()), worst
    print("fuzz ok")
