"""
Generate the golden fixtures in this directory.

Run in the BUILD container only (``python tests/golden/make_golden.py``): it imports the
reference's own numpy-only test tools from ``/root/reference/tests/tools`` (the only part of the
reference that is importable here - TensorFlow / gpflow / banded_matrices are absent) and records
their inputs and outputs as small ``.npz`` files.  The fixtures are data (inputs + expected
outputs); nothing of the reference's source travels.

Fixtures:
  kf_T8_d3_m2_b{tag}.npz   NumpyKalmanFilter on the fixture of tests/integration/test_kalman_filter.py:31-102
                           (seed 71892305, tests/conftest.py:22): per-step log-liks, RTS means/covs.
  kf_sites_T7_d2_m1.npz    NumpyKalmanFilterWithSites on the fixture of
                           tests/integration/test_kalman_filter_with_sites.py:41-75 (seed 1 at :35).
  gpr_matern32_N{15,500}.npz  SSM tensors from the reference's Matern32Test (scipy expm) + dense GP
                           log marginal likelihood (the identity of
                           tests/integration/models/test_gaussian_process_regression.py:99-105).
  matern52_sum_d6_T64.npz  d=6 chain (two Matern-5/2 via Matern52Test) + NumpyKalmanFilter answers,
                           time-invariant step so the numpy filter applies.
  btd_d{d}_T{T}_sub{0,1}.npz  random SPD block-tridiagonal matrices built like
                           tests/unit/test_block_tri_diag.py:274-295 + dense numpy.linalg answers.
  kernels_matern_T24.npz   Matern12/32/52Test (closed form / scipy expm) of the reference's test tools on random, irregular
                           time points (batch (2,)): A_k, Q_k = Pinf - A Pinf A^T, Pinf per kernel; hyper-parameters of
                           tests/unit/kernels (variance, length scale drawn once).
  ssm_T5_d3.npz            random SSM pair + dense joint-Gaussian answers (means, covs, logdet, KL, log_pdf)
                           following tests/unit/test_state_space_model.py:40-235.
"""
import importlib.util
import os
import sys

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
DEFAULT_SEED = 71892305  # /root/reference/tests/conftest.py:22


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


nkf = _load("ref_numpy_kalman_filter", "tests/tools/numpy_kalman_filter.py")
gro = _load("ref_generate_random_objects", "tests/tools/generate_random_objects.py")
rkern = _load("ref_kernels", "tests/tools/kernels/kernels.py")


def make_kf_fixture(batch_shape, tag):
    # draws in the exact order of tests/integration/test_kalman_filter.py:34-47
    np.random.seed(DEFAULT_SEED)
    num_transitions, state_dim, output_dim = 7, 3, 2
    transition_matrix = np.random.normal(size=(state_dim, state_dim))
    chol_transition_noise = gro.generate_random_lower_triangular_matrix(state_dim)
    observation_matrix = np.random.normal(size=(output_dim, state_dim))
    observation_noise = gro.generate_random_pos_def_matrix(output_dim)
    chol_observation_noise = np.linalg.cholesky(observation_noise)
    initial_state_prior_mean = np.random.normal(size=state_dim)
    state_offsets = np.random.normal(size=state_dim)
    chol_initial_state_prior_cov = gro.generate_random_lower_triangular_matrix(state_dim)
    kf = nkf.NumpyKalmanFilter(
        num_timesteps=num_transitions + 1,
        transition_matrix=transition_matrix,
        transition_mean=state_offsets,
        transition_noise=chol_transition_noise @ chol_transition_noise.T,
        observation_matrix=observation_matrix,
        observation_noise=observation_noise,
        initial_state_prior_mean=initial_state_prior_mean,
        initial_state_prior_cov=chol_initial_state_prior_cov @ chol_initial_state_prior_cov.T,
    )
    y = kf.generate_trajectories(batch_shape)
    ll, f_mu, f_p, p_mu, p_p = kf.forward_filter(y)
    s_mu, s_p = kf.backward_smoothing_pass(f_mu, f_p, p_mu, p_p)
    np.savez(
        os.path.join(OUT, f"kf_T8_d3_m2_b{tag}.npz"),
        A=transition_matrix, cholQ=chol_transition_noise, H=observation_matrix,
        R=observation_noise, cholR=chol_observation_noise, mu0=initial_state_prior_mean,
        b=state_offsets, cholP0=chol_initial_state_prior_cov, y=y,
        log_liks=ll, smooth_means=s_mu, smooth_covs=s_p, filter_means=f_mu, filter_covs=f_p,
    )


def make_sites_fixture():
    # draws in the exact order of tests/integration/test_kalman_filter_with_sites.py:41-58
    np.random.seed(1)
    num_transitions, state_dim, output_dim = 6, 2, 1
    transition_matrix = np.random.normal(size=(state_dim, state_dim))
    chol_transition_noise = gro.generate_random_lower_triangular_matrix(state_dim)
    means = np.random.normal(size=(num_transitions + 1, output_dim))
    observation_matrix = np.random.normal(size=(output_dim, state_dim))
    covariances = gro.generate_random_pos_def_matrix(output_dim, (num_transitions + 1,))
    initial_state_prior_mean = np.random.normal(size=state_dim) * 0.0
    state_offsets = np.random.normal(size=state_dim) * 0.0
    chol_initial_state_prior_cov = gro.generate_random_lower_triangular_matrix(state_dim)
    kf = nkf.NumpyKalmanFilterWithSites(
        num_timesteps=num_transitions + 1,
        transition_matrix=transition_matrix,
        transition_mean=state_offsets,
        transition_noise=chol_transition_noise @ chol_transition_noise.T,
        observation_covariances=covariances,
        observation_means=means,
        observation_matrix=observation_matrix,
        initial_state_prior_mean=initial_state_prior_mean,
        initial_state_prior_cov=chol_initial_state_prior_cov @ chol_initial_state_prior_cov.T,
    )
    ll, f_mu, f_p, p_mu, p_p = kf.forward_filter(means)
    s_mu, s_p = kf.backward_smoothing_pass(f_mu, f_p, p_mu, p_p)
    np.savez(
        os.path.join(OUT, "kf_sites_T7_d2_m1.npz"),
        A=transition_matrix, cholQ=chol_transition_noise, H=observation_matrix,
        site_means=means, site_covs=covariances, mu0=initial_state_prior_mean, b=state_offsets,
        cholP0=chol_initial_state_prior_cov,
        nat1=means / covariances[..., 0], nat2=-0.5 / covariances,
        log_liks=ll, smooth_means=s_mu, smooth_covs=s_p,
    )


def make_gpr_fixture(n, seed_offset):
    # hyper-parameters of tests/integration/models/test_gaussian_process_regression.py:32-35
    length_scale, variance, noise = 0.9, 0.3, 1e-3
    np.random.seed(DEFAULT_SEED + seed_offset)
    t, obs = gro.generate_random_time_observations(obs_dim=1, num_data=n)
    # make the problem non-trivial for N=500 (the cos(100 t) signal of the tool is kept)
    dts = np.diff(t)
    kern = rkern.Matern32Test(variance, length_scale, rkern.DataShape((), n))
    a_s = kern.state_transitions(t[:-1], dts)
    q_s = kern.process_covariances(t[:-1], dts)
    p_inf = kern.steady_state_covariance()
    lam = np.sqrt(3.0) / length_scale
    r = np.abs(t[:, None] - t[None, :])
    k_dense = variance * (1.0 + lam * r) * np.exp(-lam * r)
    kn = k_dense + noise * np.eye(n)
    yv = obs[:, 0]
    lml = (-0.5 * yv @ np.linalg.solve(kn, yv) - 0.5 * np.linalg.slogdet(kn)[1]
           - 0.5 * n * np.log(2 * np.pi))
    # dense posterior at the training inputs
    post_mean = k_dense @ np.linalg.solve(kn, yv)
    post_cov = k_dense - k_dense @ np.linalg.solve(kn, k_dense)
    np.savez(
        os.path.join(OUT, f"gpr_matern32_N{n}.npz"),
        t=t, y=obs, A=a_s, Q=q_s, P0=p_inf, H=np.array([[1.0, 0.0]]), noise=noise,
        length_scale=length_scale, variance=variance, log_marginal_likelihood=lml,
        post_mean=post_mean, post_var=np.diag(post_cov),
    )


def make_matern52_sum_fixture():
    np.random.seed(DEFAULT_SEED + 52)
    n, dt = 64, 0.13
    blocks_a, blocks_p = [], []
    for ls, var in ((0.7, 1.3), (1.9, 0.6)):
        kern = rkern.Matern52Test(var, ls, rkern.DataShape((), n))
        blocks_a.append(kern.state_transitions(None, np.array(dt)))
        blocks_p.append(kern.steady_state_covariance())
    a = np.zeros((6, 6)); p = np.zeros((6, 6))
    a[:3, :3], a[3:, 3:] = blocks_a
    p[:3, :3], p[3:, 3:] = blocks_p
    q = p - a @ p @ a.T
    h = np.array([[1.0, 0, 0, 1.0, 0, 0]])
    rvar = np.array([[0.1]])
    kf = nkf.NumpyKalmanFilter(
        num_timesteps=n, transition_matrix=a, transition_mean=np.zeros(6), transition_noise=q,
        observation_matrix=h, observation_noise=rvar, initial_state_prior_mean=np.zeros(6),
        initial_state_prior_cov=p,
    )
    y = kf.generate_trajectories((4,))
    ll, f_mu, f_p, p_mu, p_p = kf.forward_filter(y)
    s_mu, s_p = kf.backward_smoothing_pass(f_mu, f_p, p_mu, p_p)
    np.savez(os.path.join(OUT, "matern52_sum_d6_T64.npz"), A=a, Q=q, P0=p, H=h, R=rvar, y=y,
             log_liks=ll, smooth_means=s_mu, smooth_covs=s_p)


def _to_dense_lower(diag, sub):
    *batch, n, d, _ = diag.shape
    dense = np.zeros(tuple(batch) + (n * d, n * d))
    for i in range(n):
        dense[..., i * d:(i + 1) * d, i * d:(i + 1) * d] = np.tril(diag[..., i, :, :])
        if sub is not None and i < n - 1:
            dense[..., (i + 1) * d:(i + 2) * d, i * d:(i + 1) * d] = sub[..., i, :, :]
    return dense


def make_btd_fixture(d, n, has_sub, batch_shape=(3,)):
    # generator restated from tests/unit/test_block_tri_diag.py:274-295 (tril(N(1,1)) diag, N(0,1) sub)
    np.random.seed(DEFAULT_SEED + 1000 * d + 10 * n + int(has_sub))
    ldiag = np.tril(np.random.normal(loc=1.0, size=batch_shape + (n, d, d)))
    lsub = np.random.normal(size=batch_shape + (n - 1, d, d)) if has_sub else None
    if n * d > 16:
        # long chains: keep the generating factor well conditioned (|diag| >= 1, milder coupling)
        idx = np.arange(d)
        dg = 1.0 + np.abs(ldiag[..., idx, idx])
        ldiag = 0.3 * (ldiag - 1.0)
        ldiag[..., idx, idx] = dg
        ldiag = np.tril(ldiag)
        lsub = 0.3 * lsub
    lower = _to_dense_lower(ldiag, lsub)
    dense = lower @ np.swapaxes(lower, -1, -2)
    diag = np.stack([dense[..., i * d:(i + 1) * d, i * d:(i + 1) * d] for i in range(n)], axis=-3)
    sub = (np.stack([dense[..., (i + 1) * d:(i + 2) * d, i * d:(i + 1) * d] for i in range(n - 1)], axis=-3)
           if has_sub else np.zeros(batch_shape + (0, d, d)))
    rhs = np.random.normal(size=batch_shape + (n, d))
    chol = np.linalg.cholesky(dense)
    inv = np.linalg.inv(dense)
    flat = rhs.reshape(batch_shape + (n * d,))
    def blocks(mat, off):
        if off and n == 1:
            return np.zeros(batch_shape + (0, d, d))
        return np.stack([mat[..., (i + off) * d:(i + off + 1) * d, i * d:(i + 1) * d]
                         for i in range(n - off)], axis=-3)

    small = n * d <= 16
    np.savez(
        os.path.join(OUT, f"btd_d{d}_T{n}_sub{int(has_sub)}.npz"),
        diag=diag, sub=sub, has_sub=has_sub, rhs=rhs, logdet=np.linalg.slogdet(dense)[1],
        # block-form answers cut out of the dense numpy.linalg results
        chol_diag=blocks(chol, 0), chol_sub=blocks(chol, 1),
        inv_diag=blocks(inv, 0), inv_sub=blocks(inv, 1),
        dense=dense if small else np.zeros(0), chol_dense=chol if small else np.zeros(0),
        solve_l=np.linalg.solve(chol, flat[..., None])[..., 0].reshape(rhs.shape),
        solve_lt=np.linalg.solve(np.swapaxes(chol, -1, -2), flat[..., None])[..., 0].reshape(rhs.shape),
        mult_sym=(dense @ flat[..., None])[..., 0].reshape(rhs.shape),
        mult_l=(chol @ flat[..., None])[..., 0].reshape(rhs.shape),
        mult_lt=(np.swapaxes(chol, -1, -2) @ flat[..., None])[..., 0].reshape(rhs.shape),
    )


def make_ssm_fixture():
    np.random.seed(DEFAULT_SEED + 7)
    batch, n, d = (3,), 5, 3

    def rand_ssm():
        return dict(
            mu0=np.random.normal(size=batch + (d,)),
            cholP0=gro.generate_random_lower_triangular_matrix(d, batch) + 1.5 * np.eye(d),
            A=0.6 * np.random.normal(size=batch + (n, d, d)),
            b=np.random.normal(size=batch + (n, d)),
            cholQ=gro.generate_random_lower_triangular_matrix(d, batch + (n,)) + 1.5 * np.eye(d),
        )

    def dense_joint(s):
        # explicit recursion of tests/unit/test_state_space_model.py:63-101 extended to the full joint
        nd = (n + 1) * d
        means = np.zeros(batch + (n + 1, d)); cov = np.zeros(batch + (nd, nd))
        means[..., 0, :] = s["mu0"]
        p = s["cholP0"] @ np.swapaxes(s["cholP0"], -1, -2)
        cov[..., :d, :d] = p
        for k in range(n):
            a = s["A"][..., k, :, :]
            q = s["cholQ"][..., k, :, :] @ np.swapaxes(s["cholQ"][..., k, :, :], -1, -2)
            means[..., k + 1, :] = (a @ means[..., k, :, None])[..., 0] + s["b"][..., k, :]
            # Cov(x_{k+1}, x_j) = A Cov(x_k, x_j)
            row = a @ cov[..., k * d:(k + 1) * d, :(k + 1) * d]
            cov[..., (k + 1) * d:(k + 2) * d, :(k + 1) * d] = row
            cov[..., :(k + 1) * d, (k + 1) * d:(k + 2) * d] = np.swapaxes(row, -1, -2)
            cov[..., (k + 1) * d:(k + 2) * d, (k + 1) * d:(k + 2) * d] = (
                a @ cov[..., k * d:(k + 1) * d, k * d:(k + 1) * d] @ np.swapaxes(a, -1, -2) + q)
        return means, cov

    s1, s2 = rand_ssm(), rand_ssm()
    m1, c1 = dense_joint(s1)
    m2, c2 = dense_joint(s2)
    nd = (n + 1) * d
    kl = np.zeros(batch)
    for i in np.ndindex(*batch):
        mm1, mm2 = m1[i].reshape(nd), m2[i].reshape(nd)
        sol = np.linalg.solve(c2[i], c1[i]); diff = mm2 - mm1
        kl[i] = 0.5 * (np.trace(sol) + diff @ np.linalg.solve(c2[i], diff) - nd
                       + np.linalg.slogdet(c2[i])[1] - np.linalg.slogdet(c1[i])[1])
    states = np.random.normal(size=(2,) + batch + (n + 1, d))
    logpdf = np.zeros((2,) + batch)
    for j in range(2):
        for i in np.ndindex(*batch):
            diff = states[(j,) + i].reshape(nd) - m1[i].reshape(nd)
            logpdf[(j,) + i] = (-0.5 * diff @ np.linalg.solve(c1[i], diff)
                                - 0.5 * np.linalg.slogdet(c1[i])[1] - 0.5 * nd * np.log(2 * np.pi))
    np.savez(
        os.path.join(OUT, "ssm_T5_d3.npz"),
        **{f"s1_{k}": v for k, v in s1.items()}, **{f"s2_{k}": v for k, v in s2.items()},
        means1=m1, cov1=c1, means2=m2, cov2=c2, precision1=np.linalg.inv(c1),
        logdet_precision1=-np.linalg.slogdet(c1)[1], kl_12=kl, states=states, log_pdf1=logpdf,
    )


def make_kernels_fixture():
    np.random.seed(DEFAULT_SEED + 7)
    batch, n = (2,), 24
    t = gro.generate_random_time_points(expected_range=6.0, shape=batch + (n,))
    dts = np.diff(t, axis=-1)
    out = {"t": t}
    for name, cls in (("m12", rkern.Matern12Test), ("m32", rkern.Matern32Test), ("m52", rkern.Matern52Test)):
        variance, length_scale = float(np.random.uniform(0.5, 2.0)), float(np.random.uniform(0.5, 2.0))
        kern = cls(variance, length_scale, rkern.DataShape(batch, n))
        out[f"{name}_variance"], out[f"{name}_length_scale"] = variance, length_scale
        out[f"{name}_A"] = kern.state_transitions(t[..., :-1], dts)
        out[f"{name}_Q"] = kern.process_covariances(t[..., :-1], dts)
        out[f"{name}_Pinf"] = kern.steady_state_covariance()
        out[f"{name}_P0"] = kern.initial_covariance()
    np.savez(os.path.join(OUT, "kernels_matern_T24.npz"), **out)


if __name__ == "__main__":
    for shape, tag in (((), "0"), ((3,), "3"), ((2, 1), "2x1")):
        make_kf_fixture(shape, tag)
    make_sites_fixture()
    make_gpr_fixture(15, 0)
    make_gpr_fixture(500, 1)
    make_matern52_sum_fixture()
    for d, n, s in ((1, 1, False), (1, 4, True), (3, 1, False), (3, 4, True), (3, 4, False),
                    (6, 64, True), (9, 64, True)):
        make_btd_fixture(d, n, s)
    make_ssm_fixture()
    make_kernels_fixture()
    print("golden fixtures written to", OUT)
