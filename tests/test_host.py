"""CPU-only tests: the C ABI library loads and exports every declared symbol; host-side logic and error behaviour."""
import os
import re

import numpy as np
import pytest
import torch

import markovflow_amd as mfa
from markovflow_amd import _lib
from conftest import ROOT


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "markovflow_amd.h")).read()
    declared = set(re.findall(r"\b(mf_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.exported_symbols())
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.mf_version() >= 1 and lib.mf_max_state_dim() == 9


def test_argument_errors_without_touching_the_gpu():
    lib = _lib.load()
    # invalid sizes / NULL pointers are rejected before any launch
    assert lib.mf_btd_cholesky_f64(1, 0, 3, None, None, None, None, None, 0, None, None) == -2
    assert lib.mf_btd_cholesky_f64(1, 4, 0, None, None, None, None, None, 0, None, None) == -3
    assert lib.mf_btd_cholesky_f64(1, 4, 40, None, None, None, None, None, 0, None, None) == -100   # fp64: d <= 32
    assert lib.mf_btd_cholesky_f32(1, 4, 65, None, None, None, None, None, 0, None, None) == -100   # fp32: d <= 64
    assert lib.mf_btd_cholesky_f64(1, 4, 3, None, None, None, None, None, 0, None, None) == -4
    assert lib.mf_btd_solve_f32(2, 3, 4, 3, None, None, None, None, 0, None, 0, None) == -1
    assert lib.mf_kf_loglik_workspace_bytes(1024, 10000, 6, 8, 0) > 0
    assert lib.mf_kf_loglik_workspace_bytes(1024, 10000, 12, 8, 0) > 0      # LDS-tiled f64 MFMA path (d <= 32)
    assert lib.mf_kf_loglik_workspace_bytes(1024, 10000, 40, 8, 0) > 0       # fp64 log-likelihood up to d = 64: the panel kernels
    assert lib.mf_kf_loglik_workspace_bytes(8, 100, 65, 8, 0) == 0
    assert lib.mf_max_state_dim_f64_loglik() == 64 and lib.mf_max_state_dim_f64_tile_ops() == 32
    assert lib.mf_kf_loglik_workspace_bytes(1024, 10000, 64, 4, 0) > 0 and lib.mf_kf_loglik_workspace_bytes(8, 100, 65, 4, 0) == 0
    assert lib.mf_btd_cholesky_f64(0, 4, 3, None, None, None, None, None, 0, None, None) == 0   # empty batch is a no-op


def test_workspace_queries_accept_degenerate_sizes():
    lib = _lib.load()
    for b, t in ((0, 10), (4, 0), (0, 0), (-1, 5)):
        assert lib.mf_kf_loglik_workspace_bytes(b, t, 6, 8, 0) == 0
        assert lib.mf_btd_cholesky_workspace_bytes(b, t, 6, 8) == 0
        assert lib.mf_btd_solve_workspace_bytes(max(b, 0), b, t, 6, 8) == 0
        assert lib.mf_btd_diag_of_inverse_workspace_bytes(b, t, 6, 8) == 0
        assert lib.mf_btd_udl_workspace_bytes(b, t, 6, 8) == 0
        assert lib.mf_btd_logdet_quad_workspace_bytes(b, t, 6, 8) == 0
    assert lib.mf_kf_loglik_workspace_bytes(3, 1, 6, 8, 0) > 0          # a chain of one block is legal


def test_workspace_queries_of_the_round_6_paths():
    """The time-partitioned wave solve / marginal_means and the one-walk kl_divergence at 16 <= d <= 32 size their own workspaces
    (csrc/mf_wave_inst.hip); no GPU needed for the queries."""
    lib = _lib.load()
    for esz in (4, 8):
        # B = 512, T = 1000, d = 16: six chunks of (M, v, z_in) per series at least
        assert lib.mf_btd_solve_workspace_bytes(512, 512, 1000, 16, esz) >= 512 * 6 * (16 * 16 + 2 * 16) * esz
        assert lib.mf_btd_solve_workspace_bytes(64, 64, 1000, 32, esz) > 0
        assert lib.mf_ssm_kl_workspace_bytes(512, 1000, 16, esz) > 512 * 1000 * esz          # per-block terms + the chunk states
        assert lib.mf_ssm_kl_workspace_bytes(512, 1000, 32, esz) > 0
        assert lib.mf_ssm_kl_workspace_bytes(8, 1, 16, esz) == 0                              # one block: the operator route
    assert lib.mf_ssm_kl_workspace_bytes(8, 100, 40, 4) == 0                                   # d > 32: the operator route
    # operator adjoints at 10 <= d <= 32: three arrays of the factor's size + the chunk states; a token beyond 8 GiB (sequential form)
    assert lib.mf_btd_grad_workspace_bytes(4, 1001, 30, 8) >= 3 * 4 * 1001 * 30 * 30 * 8
    assert lib.mf_btd_grad_workspace_bytes(8192, 1001, 32, 8) == 16
    assert lib.mf_info_flat_index(0) == -1 and lib.mf_info_flat_index(1) == -1                 # no error / block unknown
    assert lib.mf_info_flat_index(0x7fffffff - 137) == 137


def test_cpu_tensors_fail_loudly():
    d = torch.eye(3, dtype=torch.float64).expand(2, 4, 3, 3).contiguous()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mfa.SymmetricBlockTriDiagonal(d).cholesky


def test_constructor_shape_checks():
    z = torch.zeros
    with pytest.raises(ValueError):
        mfa.EmissionModel(z(3, 2))                                         # emission_model.py:46-49
    with pytest.raises(ValueError):
        mfa.SymmetricBlockTriDiagonal(z(4, 3, 2))
    with pytest.raises(ValueError):
        mfa.SymmetricBlockTriDiagonal(z(1, 3, 3), z(0, 3, 3))              # no sub-diagonal with outer dim 1
    with pytest.raises(ValueError):
        mfa.SymmetricBlockTriDiagonal(z(4, 3, 3), z(4, 3, 3))
    with pytest.raises(ValueError):                                        # zero transitions, test_state_space_model.py:58-60
        mfa.StateSpaceModel(z(3), z(3, 3), z(0, 3, 3), z(0, 3), z(0, 3, 3))
    with pytest.raises(ValueError):                                        # batch shapes must match exactly (:111-116)
        mfa.StateSpaceModel(z(2, 3), z(2, 3, 3), z(1, 5, 3, 3), z(2, 5, 3), z(2, 5, 3, 3))
    ssm = mfa.StateSpaceModel(z(2, 3), z(2, 3, 3), z(2, 5, 3, 3), z(2, 5, 3), z(2, 5, 3, 3))
    assert ssm.event_shape == (6, 3) and ssm.num_transitions == 5 and ssm.state_dim == 3
    assert tuple(ssm.concatenated_state_offsets.shape) == (2, 6, 3)
    em = mfa.EmissionModel(z(2, 6, 1, 3))
    with pytest.raises(ValueError):
        mfa.KalmanFilter(ssm, em, z(2, 5, 1), torch.eye(1))               # kalman_filter.py:326-336
    with pytest.raises(ValueError):
        mfa.KalmanFilter(ssm, em, z(2, 6, 1), torch.eye(2))               # kalman_filter.py:320-324
    with pytest.raises(ValueError):
        mfa.UnivariateGaussianSitesNat(z(6, 2), z(6, 1, 1))               # kalman_filter.py:403-409
    kf = mfa.KalmanFilter(ssm, em, z(2, 6, 1), torch.eye(1))
    assert kf.prior_ssm is ssm and kf.emission is em


def test_sparse_sites_bookkeeping_on_cpu():
    ssm = mfa.StateSpaceModel(torch.zeros(2), torch.eye(2), torch.zeros(9, 2, 2), torch.zeros(9, 2), torch.eye(2).expand(9, 2, 2))
    em = mfa.EmissionModel(torch.ones(10, 1, 2))
    idx = torch.tensor([[1], [4], [7]])
    sites = mfa.UnivariateGaussianSitesNat(nat1=torch.tensor([[1.0], [2.0], [3.0]]), nat2=-0.5 * torch.ones(3, 1, 1))
    kf = mfa.KalmanFilterWithSparseSites(ssm, em, sites, 10, idx, torch.tensor([[[1.0], [2.0], [3.0]]]))
    assert tuple(kf.observations.shape) == (10, 1) and float(kf.observations[4, 0]) == 2.0
    assert tuple(kf._r_inv.shape) == (10, 1, 1) and float(kf._r_inv[7, 0, 0]) == 1.0 and float(kf._r_inv[0, 0, 0]) == 0.0
    np.testing.assert_allclose(kf.dense_to_sparse(kf.observations).numpy(), [[1.0], [2.0], [3.0]])
    np.testing.assert_allclose(sites.means.numpy(), [[1.0], [2.0], [3.0]])
    with pytest.raises(Exception):
        mfa.KalmanFilterWithSparseSites(ssm, em, sites, 10, idx, torch.zeros(2, 3, 1))   # batches unsupported (:531-539)


def test_synthetic_closed_forms_match_reference_expm_kernels():
    """Closed-form Matérn transitions == the reference's scipy-expm test kernels (fixtures hold their output)."""
    from conftest import golden
    from markovflow_amd import synthetic
    g = golden("gpr_matern32_N15.npz")
    dt = torch.tensor(np.diff(g["t"]))[None]
    f64 = torch.float64
    lam = torch.tensor([np.sqrt(3.0) / float(g["length_scale"])], dtype=f64)
    a, pinf = synthetic._matern_block(3, lam, torch.tensor([float(g["variance"])], dtype=f64), dt)
    np.testing.assert_allclose(a[0].numpy(), g["A"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(pinf[0].numpy(), g["P0"], rtol=1e-12)
    g5 = golden("matern52_sum_d6_T64.npz")
    a5, p5 = synthetic._matern_block(5, torch.tensor([np.sqrt(5.0) / 0.7], dtype=f64), torch.tensor([1.3], dtype=f64),
                                       torch.tensor([[0.13]], dtype=f64))
    np.testing.assert_allclose(a5[0, 0].numpy(), g5["A"][:3, :3], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(p5[0].numpy(), g5["P0"][:3, :3], rtol=1e-12)
    inp = synthetic.make_ssm(3, 12, (5, 5), device="cpu")
    assert tuple(inp["A"].shape) == (3, 11, 6, 6) and tuple(inp["H"].shape) == (3, 12, 1, 6)
    inp9 = synthetic.make_ssm(2, 5, (5, 5, 5), output_dim=3, device="cpu")
    assert tuple(inp9["H"].shape) == (2, 5, 3, 9) and tuple(inp9["cholR"].shape) == (3, 3)


@pytest.mark.parametrize("d,t,has_sub,batch", [(1, 1, False, ()), (3, 4, True, ()), (3, 4, False, (2,)), (2, 5, True, (2, 1))])
def test_as_band_layout_and_round_trip(d, t, has_sub, batch):
    """band[k, j] = M[j + k, j] (the BandedMatrixTensor layout, block_tri_diag.py:84-98) and its inverse - host-side torch."""
    from markovflow_amd.block_tri_diag import _banded_to_block_tri
    rng = np.random.default_rng(3)
    diag = torch.tensor(np.tril(rng.normal(size=batch + (t, d, d))))
    sub = torch.tensor(rng.normal(size=batch + (t - 1, d, d))) if has_sub else None
    low = mfa.LowerTriangularBlockTriDiagonal(diag, sub)
    band = low.as_band
    n = d * t
    assert tuple(band.shape) == batch + (low.bandwidth + 1, n)
    dense = np.zeros(batch + (n, n))
    for k in range(t):
        dense[..., k * d:(k + 1) * d, k * d:(k + 1) * d] = diag[..., k, :, :].numpy()
        if has_sub and k + 1 < t:
            dense[..., (k + 1) * d:(k + 2) * d, k * d:(k + 1) * d] = sub[..., k, :, :].numpy()
    for kk in range(low.bandwidth + 1):
        for j in range(n):
            want = dense[..., j + kk, j] if j + kk < n else np.zeros(batch)
            np.testing.assert_allclose(band[..., kk, j].numpy(), want)
    back = _banded_to_block_tri(band, d)
    np.testing.assert_allclose(back.block_diagonal.numpy(), diag.numpy())
    if has_sub:
        np.testing.assert_allclose(back.block_sub_diagonal.numpy(), sub.numpy())
    else:
        assert back.block_sub_diagonal is None


def test_mixed_dtypes_are_rejected_before_any_launch():
    """A float32 observation buffer handed to a float64 chain would be read past its end by the kernels: ValueError, as the
    reference's dtype check."""
    import markovflow_amd as mfa
    d, n = 2, 5
    f64 = dict(dtype=torch.float64)
    ssm = mfa.StateSpaceModel(torch.zeros(d, **f64), torch.eye(d, **f64), torch.eye(d, **f64).expand(n - 1, d, d).contiguous(),
                              torch.zeros(n - 1, d, **f64), torch.eye(d, **f64).expand(n - 1, d, d).contiguous())
    em = mfa.EmissionModel(torch.ones(n, 1, d, **f64))
    with pytest.raises(ValueError, match="observations"):
        mfa.KalmanFilter(ssm, em, torch.zeros(n, 1, dtype=torch.float32), torch.eye(1, **f64))
    with pytest.raises(ValueError, match="chol_obs_covariance"):
        mfa.KalmanFilter(ssm, em, torch.zeros(n, 1, **f64), torch.eye(1, dtype=torch.float32))
    with pytest.raises(ValueError, match="emission_matrix"):
        mfa.KalmanFilter(ssm, mfa.EmissionModel(torch.ones(n, 1, d, dtype=torch.float32)), torch.zeros(n, 1, **f64), torch.eye(1, **f64))
    with pytest.raises(ValueError, match="time_points"):
        mfa.GaussianProcessRegression((torch.arange(n, **f64), torch.zeros(n, 1, dtype=torch.float32)), mfa.Matern32(1.0, 1.0))


def test_bench_refuses_more_gpus_than_visible():
    """`bench.py --gpus N` starts its own ranks; with fewer GPUs than asked it exits non-zero instead of reporting n_gpus=1."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    proc = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64"], capture_output=True, text=True, timeout=300)
    assert proc.returncode == 2 and "--gpus 64" in proc.stderr


@pytest.mark.parametrize("gpus", [2, 4, 8])
def test_bench_dry_launch_runs_the_rank_plumbing_without_a_gpu(gpus):
    """`bench.py --gpus N --dry-launch`: the GPU-free parent (it asserts that it never imported torch) spawns N ranks through
    torch.distributed.run; they join a gloo group, shard the batch and all-reduce a probe.  Port / argv / environment /
    stdout discipline / return code of the real multi-GPU launch, in CI."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    proc = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus), "--dry-launch", "--batch", "1000"],
                          capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, proc.stdout                                    # rank 0's stdout carries exactly one JSON line
    out = json.loads(lines[0])
    assert out["dry_launch"] and out["n_gpus"] == gpus and out["ranks_seen"] == gpus
    assert out["series_total"] == 1000 * gpus and out["ipc_mode_legacy"] == "0"


def test_bench_dry_launch_of_the_config4_workload_at_world_8():
    """BASELINE config 4 (4096 series over 8 GPUs, log-likelihood + KL combined by ONE scalar all-reduce): the same launch path
    with `--workload config4`, eight ranks, the shard size of the real run and the sharded ELBO reduction on gloo (VERDICT r04
    next 10)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    proc = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--dry-launch", "--workload", "config4"],
                          capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, proc.stdout
    out = json.loads(lines[0])
    assert out["dry_launch"] and out["n_gpus"] == 8 and out["ranks_seen"] == 8 and out["workload"] == "config4"
    assert out["series_total"] == 4096 and out["series_per_gpu"] == 512 and out["sharded_total"] == 4096.0


def test_visible_gpus_opens_no_runtime():
    """The launcher counts GPUs from sysfs: no torch, no HIP library in the process afterwards."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import bench; n = bench.visible_gpus(); "
            "assert 'torch' not in sys.modules; "
            "maps = open('/proc/self/maps').read(); assert 'libamdhip64' not in maps and 'libhsa-runtime' not in maps; print(n)" % root)
    proc = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert proc.returncode == 0, proc.stderr
    assert int(proc.stdout.strip()) >= 0


def test_deferred_pivot_failures_keep_their_names_until_a_synchronised_look():
    """an unsynchronised clean look at the pivot flag must not forget the factorisations on record; the
    flag is read whether or not a name is recorded; a synchronised clean look clears the record.  (CPU: a stand-in flag.)"""
    import ctypes
    from markovflow_amd import _lib
    flag = torch.zeros(1, dtype=torch.int32)

    class _HostStandIn:                 # what _lib._Flag offers, without a device: the mirror word, a stream, clear()
        view = ctypes.c_int.from_address(flag.data_ptr())

        class stream:
            cuda_stream = 0
            synchronize = staticmethod(lambda: None)

        def clear(self):
            self.view.value = 0

    saved_flags, saved_issued = dict(_lib._flags), list(_lib._issued)
    try:
        _lib._flags.clear()
        _lib._issued.clear()
        _lib._flags[(0, 0)] = _HostStandIn()
        _lib._issued.append("SymmetricBlockTriDiagonal.cholesky")
        _lib.raise_pending()                                   # e.g. the look at the start of a later solve: kernel still running
        assert _lib._issued == ["SymmetricBlockTriDiagonal.cholesky"]
        flag[0] = 1                                            # ... the kernel finishes and raises the flag
        with pytest.raises(_lib.MarkovflowAmdError, match="SymmetricBlockTriDiagonal.cholesky"):
            _lib.raise_pending(synced=True)
        assert int(flag[0]) == 0 and _lib._issued == []
        flag[0] = 1                                            # a flag without any name on record is still reported
        with pytest.raises(_lib.MarkovflowAmdError, match="a factorisation"):
            _lib.raise_pending()
        _lib._issued.append("x")
        _lib.raise_pending(synced=True)                        # synchronised and clean: the record is dropped
        assert _lib._issued == []
        # results that reach the host look at the flag
        val = _lib.checked(torch.tensor(2.0, dtype=torch.float64))
        assert float(val + 1) == 3.0
        flag[0] = 1
        with pytest.raises(_lib.MarkovflowAmdError):
            (val * 2).item()
    finally:
        _lib._flags.clear()
        _lib._flags.update(saved_flags)
        _lib._issued[:] = saved_issued


def test_dense_local_gradients_match_autograd_of_the_dense_joint():
    """The closed forms behind the log-likelihood gradients for d > 9 (kalman_filter._local_gradients_dense: Fisher's identity on
    smoothed moments) against torch autograd through the dense joint Gaussian - on the CPU, with the moments from the numpy
    oracle's posterior chain; per-series weights included.  (The GPU path differs only in where the moments come from.)"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import numpy_oracle as O
    from markovflow_amd.kalman_filter import _local_gradients_dense
    from test_gpu_gradients import dense_log_likelihood
    rng = np.random.default_rng(3)
    bsz, t, d, m = 2, 5, 12, 3
    vals = dict(mu0=rng.normal(size=(bsz, d)), cp0=np.tril(0.2 * rng.normal(size=(bsz, d, d))) + np.eye(d),
                a=0.7 * np.eye(d) + 0.1 * rng.normal(size=(bsz, t - 1, d, d)), b=0.1 * rng.normal(size=(bsz, t - 1, d)),
                cq=np.tril(0.1 * rng.normal(size=(bsz, t - 1, d, d))) + 0.5 * np.eye(d), h=rng.normal(size=(bsz, t, m, d)),
                y=rng.normal(size=(bsz, t, m)))
    chol_r = np.linalg.cholesky(0.4 * np.eye(m) + 0.1 * np.ones((m, m)))
    r_inv = np.linalg.inv(chol_r @ chol_r.T)
    mu0p, cp0p, ap, bp, cqp = O.kf_posterior_ssm(*vals.values(), r_inv)
    means, covs = O.ssm_marginal_means(mu0p, ap, bp), O.ssm_marginal_covariances(cp0p, ap, cqp)
    cross = ap @ covs[:, :-1]
    w = np.array([0.7, -1.3])
    cpu = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in vals.items()}
    cr = torch.tensor(chol_r, requires_grad=True)
    sum(w[s] * dense_log_likelihood(*(cpu[k][s] for k in vals), cr) for s in range(bsz)).backward()
    tt = lambda a: torch.tensor(a, dtype=torch.float64)   # noqa: E731
    got = _local_gradients_dense(*(tt(v) for v in vals.values()), tt(r_inv), tt(means), tt(covs), tt(cross), tt(w))
    for k, g in zip(vals, got[:7]):
        want = cpu[k].grad.numpy()
        want = np.tril(want) if k in ("cp0", "cq") else want
        np.testing.assert_allclose(g.numpy(), want, rtol=1e-9, atol=1e-11, err_msg=k)


@pytest.mark.parametrize("t", [1, 2, 3, 7, 12])
def test_dense_scan_of_the_moments_and_local_kl_match_the_oracle(t):
    """What the backward of `marginals` / `kl_divergence` re-evaluates for state dimensions above 9 (state_space_model._dense_moments:
    the recursion as a parallel scan in batched products; _dense_kl: the local form of the divergence) against the numpy oracle."""
    from oracle import numpy_oracle as O
    from markovflow_amd.state_space_model import _dense_kl, _dense_moments

    def chain(seed, bsz=2, d=11):
        r = np.random.default_rng(seed)
        return (r.normal(size=(bsz, d)), np.tril(0.2 * r.normal(size=(bsz, d, d))) + np.eye(d),
                0.7 * np.eye(d) + 0.1 * r.normal(size=(bsz, t - 1, d, d)), 0.1 * r.normal(size=(bsz, t - 1, d)),
                np.tril(0.1 * r.normal(size=(bsz, t - 1, d, d))) + 0.5 * np.eye(d))
    q1, q2 = chain(3), chain(4)
    t1, t2 = [torch.tensor(x) for x in q1], [torch.tensor(x) for x in q2]
    means, covs = _dense_moments(*t1)
    np.testing.assert_allclose(means.numpy(), O.ssm_marginal_means(q1[0], q1[2], q1[3]), rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(covs.numpy(), O.ssm_marginal_covariances(q1[1], q1[2], q1[4]), rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(_dense_kl(t1, t2).numpy(), O.ssm_kl_divergence(q1, q2), rtol=1e-11)


def test_row_kernel_emulation_matches_the_oracle():
    """scripts/row_sim.py executes the op sequence of the row kernels (csrc/mf_row.hpp) register by register with the one
    cross-lane primitive they use - on the CPU, against the numpy oracle: the layout / sign bookkeeping of the kernel's design."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("row_sim", os.path.join(ROOT, "scripts", "row_sim.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main()          # asserts the worst relative deviation over its cases < 1e-10


def test_dense_prediction_closed_forms_against_joint_gaussian_conditioning():
    """posterior._predict_state_dense (the route of state dimensions beyond the lane-per-point kernel) against a direct numpy
    evaluation of conditionals.py:29-83,122-203 of the reference: x_t | x_-, x_+ ~ N(D x_- + E x_+, T), marginalised over the
    pairwise posterior of the neighbours (the prior beyond the ends)."""
    import numpy as np
    import torch
    from markovflow_amd.posterior import _predict_state_dense

    rng = np.random.default_rng(5)
    b, n, npts, d = 2, 5, 7, 11

    def spd(*lead):
        a = rng.normal(size=lead + (d, d))
        return a @ np.swapaxes(a, -1, -2) + 0.5 * np.eye(d)

    idx = np.array([[0, 1, 2, 3, 4, 5, 5], [0, 0, 2, 2, 3, 5, 1]])
    a_mt, a_tp = 0.5 * rng.normal(size=(b, npts, d, d)), 0.5 * rng.normal(size=(b, npts, d, d))
    q_mt, q_tp = spd(b, npts), spd(b, npts)
    means, m0 = rng.normal(size=(b, n, d)), rng.normal(size=(b, d))
    p0 = spd(b)
    # a consistent joint over the training points is not needed: any pairwise (P-, P+, C) with a PSD joint will do
    joint = spd(b, n)                                                       # marginal covariances
    sub = 0.1 * rng.normal(size=(b, n - 1, d, d))                           # Cov(x_{k+1}, x_k)
    t = lambda x: torch.tensor(x)                                            # noqa: E731
    mean, cov = _predict_state_dense(t(idx), t(a_mt), t(q_mt), t(a_tp), t(q_tp), t(means), t(joint), t(sub), t(m0), t(p0))
    for s in range(b):
        for j in range(npts):
            i = idx[s, j]
            am, ap, qm, qp = a_mt[s, j], a_tp[s, j], q_mt[s, j], q_tp[s, j]
            qpm = qp + ap @ qm @ ap.T
            e = qm @ ap.T @ np.linalg.inv(qpm)
            dm = am - e @ ap @ am
            tm = qm - qm @ ap.T @ np.linalg.inv(qpm) @ ap @ qm
            mu_m, pm = (means[s, i - 1], joint[s, i - 1]) if i > 0 else (m0[s], p0[s])
            mu_p, pp = (means[s, i], joint[s, i]) if i < n else (m0[s], p0[s])
            c = sub[s, i - 1] if 0 < i < n else np.zeros((d, d))
            want_mean = dm @ mu_m + e @ mu_p
            want_cov = tm + dm @ pm @ dm.T + e @ pp @ e.T + e @ c @ dm.T + dm @ c.T @ e.T
            np.testing.assert_allclose(mean[s, j].numpy(), want_mean, rtol=1e-10, atol=1e-12)
            np.testing.assert_allclose(cov[s, j].numpy(), want_cov, rtol=1e-9, atol=1e-11)


def test_plans_and_workspace_queries_of_the_streamed_backward_need_no_gpu():
    """mf_kf_loglik_plan (which level-0 kernel, which time partition), the workspace queries of the streamed backward and of the
    fused GPR backward are host code: the headline shape takes the streaming kernel on 64 chunks of 157 transitions, and the
    partition the fused GPR route computes in Python (models._GprFusedLogLik) is the one mf_gpr_matern_loglik makes of an
    explicit chunk count."""
    import ctypes
    lib = _lib.load()
    path, p, length = ctypes.c_int(-1), ctypes.c_int64(0), ctypes.c_int64(0)
    assert lib.mf_kf_loglik_plan(1024, 10000, 6, 1, 0, 8, 0, 1, ctypes.byref(path), ctypes.byref(p), ctypes.byref(length)) == 0
    assert (path.value, p.value, length.value) == (2, 64, 157)
    # unaligned tensors cannot take the LDS-DMA kernel
    assert lib.mf_kf_loglik_plan(1024, 10000, 6, 1, 0, 8, 0, 0, ctypes.byref(path), ctypes.byref(p), ctypes.byref(length)) == 0
    assert path.value != 2
    # so many series that one chunk each fills the chip: a single chunk (the autograd forward then asks for two)
    assert lib.mf_kf_loglik_plan(100000, 128, 6, 1, 0, 8, 0, 1, ctypes.byref(path), ctypes.byref(p), ctypes.byref(length)) == 0
    assert (path.value, p.value) == (2, 1)
    assert lib.mf_kf_loglik_plan(100000, 128, 6, 1, 0, 8, 2, 1, ctypes.byref(path), ctypes.byref(p), ctypes.byref(length)) == 0
    assert (path.value, p.value, length.value) == (2, 2, 64)
    assert lib.mf_kf_loglik_grad_streamed_workspace_bytes(1024, 10000, 6, 1, 0, 8, 0) > 1024 * 9999 * 224
    assert lib.mf_kf_loglik_grad_streamed_workspace_bytes(1024, 10000, 7, 1, 0, 8, 0) == 0        # d = 7: not this route's
    assert lib.mf_kf_loglik_grad_streamed_workspace_bytes(1024, 10000, 6, 4, 0, 8, 0) == 0        # four outputs neither
    assert lib.mf_gpr_matern_loglik_grad_workspace_bytes(1024, 10000, 6, 8, 64) > 1024 * 9999 * 224
    assert lib.mf_gpr_matern_loglik_grad_workspace_bytes(1024, 10000, 6, 8, 1) == 0               # no summaries to start from
    assert lib.mf_version() == 8


def test_filter_cache_key_follows_the_source_tensors_not_their_flattened_copies():
    """VERDICT r04 weak 1 / ADVICE r04: the smoother-after-the-filter cache used to be keyed on the flattened, broadcast copies
    handed to the kernels - temporaries whose address the allocator hands back on the next call with version 0, so an in-place
    write to the shared source went unseen (20 / 20 collisions on the CPU allocator).  The key is now built from the tensors
    the filter HOLDS and must change on every in-place write, for broadcast and non-contiguous sources alike; an unchanged
    model must keep its key.  (No kernel runs: CPU tensors.)"""
    rng = np.random.default_rng(5)
    for bsz, t, d, m in ((3, 40, 6, 1), (2, 17, 4, 2), (5, 9, 2, 1), (1, 100, 3, 1)):
        a_store = torch.tensor(rng.normal(size=(bsz, t - 1, d, d)))
        ssm = mfa.StateSpaceModel(torch.zeros(bsz, d, dtype=torch.float64), torch.eye(d, dtype=torch.float64).expand(bsz, d, d),
                                  a_store.transpose(-1, -2), torch.zeros(bsz, t - 1, d, dtype=torch.float64),
                                  torch.eye(d, dtype=torch.float64).expand(bsz, t - 1, d, d))
        h = torch.tensor(rng.normal(size=(t, m, d)))                      # shared by the batch: expanded on every call
        y_store = torch.tensor(rng.normal(size=(bsz, t, 2 * m)))
        y = y_store[..., ::2]
        chol_r = torch.eye(m, dtype=torch.float64)
        kf = mfa.KalmanFilter(ssm, mfa.EmissionModel(h), y, chol_r)
        for _ in range(5):
            key0, sources = kf._cache_key()
            assert key0 is not None and kf._cache_key()[0] == key0         # stable while nothing changes
            # the OLD key, for the record: built from the expanded temporaries it cannot see the write below
            old = [tuple((x.data_ptr(), x._version) for x in kf._expanded()[:2])]
            h.mul_(2.0)
            old.append(tuple((x.data_ptr(), x._version) for x in kf._expanded()[:2]))
            key1 = kf._cache_key()[0]
            assert key1 != key0, ("emission matrix written in place, key unchanged", old)
            ssm.state_transitions.mul_(0.5)                               # (the chain holds a contiguous snapshot of the view)
            key2 = kf._cache_key()[0]
            assert key2 != key1
            y_store.add_(1.0)                                             # through the BASE of the strided view the filter holds
            key3 = kf._cache_key()[0]
            assert key3 != key2
            chol_r.mul_(1.1)
            assert kf._cache_key()[0] != key3
            del sources
    # a filter whose inputs are derived on the fly from tensors the base class cannot name caches nothing
    sites = mfa.UnivariateGaussianSitesNat(torch.ones(t, 1, dtype=torch.float64), -torch.ones(t, 1, 1, dtype=torch.float64))
    ssm1 = mfa.StateSpaceModel(torch.zeros(d, dtype=torch.float64), torch.eye(d, dtype=torch.float64),
                               torch.zeros(t - 1, d, d, dtype=torch.float64), torch.zeros(t - 1, d, dtype=torch.float64),
                               torch.eye(d, dtype=torch.float64).expand(t - 1, d, d))
    kfs = mfa.KalmanFilterWithSites(ssm1, mfa.EmissionModel(torch.ones(t, 1, d, dtype=torch.float64)), sites)
    assert kfs._cache_key() == (None, None)


def test_kernels_can_be_built_and_evaluated_under_inference_mode():
    """the positivity / sqrt(order)/lengthscale caches are keyed on the autograd version counter, which inference tensors
    do not have - building or evaluating a kernel under torch.inference_mode() must not raise."""
    import markovflow_amd as mfa
    with torch.inference_mode():
        k = mfa.Matern32(torch.tensor(1.0, dtype=torch.float64), torch.tensor(2.0, dtype=torch.float64))
        assert float(k._lambda) == pytest.approx(3 ** 0.5) and k.steady_state_covariance.shape == (2, 2)
        s = mfa.Sum([k, mfa.Matern12(torch.tensor(0.5, dtype=torch.float64), torch.tensor(1.0, dtype=torch.float64))])
        assert s.state_dim == 3
    with pytest.raises(ValueError):
        with torch.inference_mode():
            mfa.Matern32(torch.tensor(-1.0, dtype=torch.float64), torch.tensor(2.0, dtype=torch.float64))


def test_kernel_cache_follows_reseated_data_and_can_be_invalidated():
    """a `set_()` re-seats the storage without touching the version counter - the cache key holds the data pointer
    too; a write through `.data` is invisible to both and is what invalidate_cache() is for."""
    import markovflow_amd as mfa
    ls = torch.tensor(2.0, dtype=torch.float64)
    k = mfa.Matern32(ls, torch.tensor(1.0, dtype=torch.float64))
    lam0 = float(k._lambda)
    ls.set_(torch.tensor(4.0, dtype=torch.float64))
    assert float(k._lambda) == pytest.approx(lam0 / 2)
    ls.data.mul_(2.0)
    k.invalidate_cache()
    assert float(k._lambda) == pytest.approx(lam0 / 4)
