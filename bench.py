#!/usr/bin/env python3
"""
bench.py - headline benchmark of the Kalman log-likelihood hot path on MI355X.

Metric (BASELINE.json): Kalman log-lik steps/s = (series x time points) / wall time of one
``KalmanFilter.log_likelihood()`` at d=6, on the north-star target configuration
B=1024, T=10000, d=6, m=1 (per GPU; weak scaling: every rank owns its own 1024 series and the only
collective is one RCCL all-reduce of the scalar log-likelihood).

A "step" of the driver contract = one full ``log_likelihood()`` evaluation over the resident batch.
Prints ONE JSON line on rank 0 with `roofline` (HBM roofline of the dominant kernel, timed live with
HIP events on the launch stream) and `cpu_baseline` (the C restatement of the reference algorithm,
oracle/c/mf_oracle.c, timed on the host cores on a bounded sample of the same workload).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1024, help="series per GPU")
    ap.add_argument("--time-points", type=int, default=10000)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--chunks", type=int, default=0, help="time partitions per series (0 = automatic)")
    ap.add_argument("--workload", default="target", choices=["target", "config4"],
                    help="target: the headline (KalmanFilter.log_likelihood, B=1024/GPU, T=10000, d=6); config4: BASELINE config 4's "
                         "per-GPU shard (512 series, T=1000, 3 x Matern-5/2 with 3 outputs, d=9) - log-likelihood + KL(q || prior), "
                         "one scalar all-reduce (sharded_elbo)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the brief timing of BASELINE configs 2-5")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline sample")
    ap.add_argument("--dry-launch", action="store_true",
                    help="exercise the multi-rank launch only: ranks join a gloo group, shard the batch, all-reduce a probe and "
                         "print n_gpus / ranks_seen; no GPU is touched (runs in CI without one)")
    return ap.parse_args()


def visible_gpus() -> int:
    """GPUs a rank could use, counted WITHOUT opening the HIP / HSA runtime (the launching process must stay GPU-free: a
    process that has initialised the GPU must never be the parent that gets replaced or forked into ranks): KFD topology
    nodes that have SIMDs, cut down by the *_VISIBLE_DEVICES lists."""
    import glob

    n = 0
    for path in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            with open(path) as fh:
                for line in fh:
                    if line.startswith("simd_count"):
                        n += int(line.split()[1]) > 0
        except (OSError, ValueError, IndexError):
            pass
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        lst = os.environ.get(var)
        if lst is not None:
            n = min(n, len([x for x in lst.split(",") if x.strip()]))
    return n


class HipEvents:
    """A pair of hipEvent_t created directly through the HIP runtime (timing the dominant kernel alone)."""

    def __init__(self):
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.start, self.stop = ctypes.c_void_p(), ctypes.c_void_p()
        assert self.hip.hipEventCreate(ctypes.byref(self.start)) == 0
        assert self.hip.hipEventCreate(ctypes.byref(self.stop)) == 0

    def elapsed_ms(self) -> float:
        ms = ctypes.c_float()
        assert self.hip.hipEventSynchronize(self.stop) == 0
        assert self.hip.hipEventElapsedTime(ctypes.byref(ms), self.start, self.stop) == 0
        return float(ms.value)


def _cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _native_oracle():
    """Compile oracle/c/mf_oracle.c for THIS host (-march=native) into a scratch directory; None if that fails (the
    -march=x86-64-v3 library shipped with the repo is used then).  Building the checker is not using it."""
    import subprocess
    import tempfile

    src = os.path.join(ROOT, "oracle", "c", "mf_oracle.c")
    out = os.path.join(tempfile.mkdtemp(prefix="mf_oracle_"), "libmf_oracle_native.so")
    cmd = ["gcc", "-O3", "-march=native", "-fopenmp", "-fPIC", "-std=gnu11", "-shared", "-o", out, src, "-lm"]
    try:
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120)
        return out
    except Exception:
        return None


def cpu_baseline(inputs, seconds: float):
    """Time the C restatement of the reference algorithm (oracle/c/mf_oracle.c, dimensions d=6 / m=1 fixed at compile time,
    built for this host) on the same workload: the WHOLE batch on all host threads, and a bounded sample on one thread."""
    import numpy as np

    from oracle import c_oracle as C

    native = _native_oracle()
    if native is not None:
        C.use_library(native)
    cores = C.num_threads()
    bsz, t = inputs["H"].shape[0], inputs["H"].shape[1]

    def host(k):
        return inputs[k].detach().cpu().numpy().astype(np.float64, copy=False)

    arrs = {k: host(k) for k in ("mu0", "cholP0", "A", "b", "cholQ", "H", "y")}
    r_inv = np.linalg.inv((inputs["cholR"] @ inputs["cholR"].T).cpu().numpy().astype(np.float64))

    def run(n):
        t0 = time.perf_counter()
        out = C.kf_loglik(arrs["mu0"][:n], arrs["cholP0"][:n], arrs["A"][:n], arrs["b"][:n], arrs["cholQ"][:n],
                          arrs["H"][:n], arrs["y"][:n], r_inv)
        return time.perf_counter() - t0, out

    def best_of(n, budget):
        dt, out = run(n)                      # warm-up + calibration
        best, total, reps = dt, 0.0, 0
        while total < budget and reps < 50:
            dt, out = run(n)
            best, total, reps = min(best, dt), total + dt, reps + 1
        return best, reps, out

    C.set_num_threads(cores)
    best_all, reps_all, out = best_of(bsz, 0.6 * seconds)
    C.set_num_threads(1)
    n1 = max(1, min(bsz, 32))
    best_1, reps_1, _ = best_of(n1, 0.4 * seconds)
    C.set_num_threads(cores)
    return {
        "value": bsz * t / best_all, "unit": "steps/s", "cores": cores, "kind": "port",
        "value_1_thread": n1 * t / best_1, "cpu_model": _cpu_model(),
        "build": "gcc -O3 -march=native -fopenmp" if native else "gcc -O3 -march=x86-64-v3 -fopenmp (shipped build)",
        "sample": f"all threads: the whole batch, {bsz} series x {t} time points (d={arrs['A'].shape[-1]}, m=1, fp64), best of "
                  f"{reps_all} passes; 1 thread: {n1} series x {t}, best of {reps_1}; C restatement of the reference algorithm "
                  "(oracle/c/mf_oracle.c: precision assembly, banded Cholesky, solve - state and output dimensions fixed "
                  "at compile time for d=6, m=1), OpenMP over series",
    }, out


def _time_gpu(fn, iters=10, warm=2):
    import torch

    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def other_configs(dev):
    """The remaining BASELINE.json configs, timed briefly on rank 0 (parity for them lives in tests/): reported beside
    the headline, never as `value`."""
    import torch

    from markovflow_amd import _lib, synthetic
    import markovflow_amd as mfa

    out = {}
    # config 2: KalmanFilter.log_likelihood B=256 T=4096 d=4 fp64
    inp = synthetic.make_ssm(256, 4096, (3, 3), dtype=torch.float64, device=dev)
    kf = synthetic.kalman_filter_from(inp)
    ms = _time_gpu(kf.log_likelihood)
    out["config2_loglik_B256_T4096_d4_f64"] = {
        "ms": ms, "steps_per_s": 256 * 4096 / ms * 1e3,
        "algorithmic_GBps": 256 * 4096 * synthetic.loglik_bytes_per_step(4, 1, 8) / ms / 1e6}
    # the headline shape in FLOAT32: B=1024, T=10000, d=6, m=1 on a chain fp32 can represent (three Matern-3/2
    # components at gaps 0.2 + Exp(0.3): the headline's Matern-5/2 process covariances are below fp32 resolution at its gaps;
    # tests/test_gpu_baseline_configs.py::test_headline_shape_fp32_on_a_chain_fp32_can_represent holds its parity)
    inp = synthetic.make_ssm(1024, 10000, (3, 3, 3), dtype=torch.float32, device=dev, dt_min=0.2, dt_scale=0.3, jitter=1e-6)
    kf = synthetic.kalman_filter_from(inp)
    ms = _time_gpu(kf.log_likelihood)
    b32 = synthetic.loglik_bytes_per_step(6, 1, 4)
    out["headline_f32"] = {
        "ms": ms, "steps_per_s": 1024 * 10000 / ms * 1e3, "bytes_per_step": b32,
        "algorithmic_GBps": 1024 * 10000 * b32 / ms / 1e6, "frac_of_hbm_peak": 1024 * 10000 * b32 / ms / 1e6 / HBM_PEAK_GBS,
        "note": "KalmanFilter.log_likelihood B=1024 T=10000 d=6 m=1 float32, whole call; 340 B/step algorithmic.  The fp32 rows are "
                "144 B: three 128-B lines are touched for 2.25 lines of data on the two wide streams, and the step is issue-bound at "
                "the same instruction count as fp64 with half the bytes (DESIGN.md, K0)"}
    del inp, kf
    # config 3: SymmetricBlockTriDiagonal.cholesky + solve, T=100000 d=6 fp32, one chain (parallel-in-time path)
    n, d = 100000, 6
    g = torch.Generator(device=dev); g.manual_seed(3)
    ld = torch.tril(0.3 * torch.randn(1, n, d, d, dtype=torch.float64, device=dev, generator=g))
    ld = ld - torch.diag_embed(torch.diagonal(ld, dim1=-2, dim2=-1)) + torch.diag_embed(
        1 + torch.rand(1, n, d, dtype=torch.float64, device=dev, generator=g))
    ls = 0.3 * torch.randn(1, n - 1, d, d, dtype=torch.float64, device=dev, generator=g)
    diag = ld @ ld.transpose(-1, -2)
    diag[:, 1:] += ls @ ls.transpose(-1, -2)
    sub = ls @ ld[:, :-1].transpose(-1, -2)
    sym = mfa.SymmetricBlockTriDiagonal(diag.float().contiguous(), sub.float().contiguous())
    rhs = torch.randn(1, n, d, dtype=torch.float32, device=dev, generator=g)
    chol = sym.cholesky
    t_c = _time_gpu(lambda: sym.cholesky)
    t_s = _time_gpu(lambda: chol.solve(rhs))
    out["config3_btd_T100000_d6_f32_B1"] = {
        "cholesky_us": t_c * 1e3, "solve_us": t_s * 1e3,
        "cholesky_algorithmic_GBps": n * 4 * d * d * 4 / t_c / 1e6, "solve_algorithmic_GBps": n * (2 * d * d + 2 * d) * 4 / t_s / 1e6,
        "frac_of_hbm_peak_cholesky": n * 4 * d * d * 4 / t_c / 1e6 / HBM_PEAK_GBS,
        "note": "one chain: bound by the dependent block steps of the multi-level elimination (13 launches of 5-8 dependent "
                "block steps each, ~1.5 k instructions per step on one lane), not bytes; the figures include the Python-side "
                "allocation of outputs and workspace",
        "max_abs_err_vs_exact_factor": float((chol.block_diagonal.double() - ld).abs().max())}
    # the same operator where it IS bandwidth-bound: many series, one lane per series (B >= 4096), d=6 fp64
    bb, tb = 16384, 500
    inp = synthetic.make_ssm(bb, tb, (5, 5), dtype=torch.float64, device=dev)
    prec = synthetic.kalman_filter_from(inp)._k_inv_post
    t_b = _time_gpu(lambda: prec.cholesky, iters=5)
    chol_b = prec.cholesky
    rhs_b = torch.randn(bb, tb, 6, dtype=torch.float64, device=dev, generator=g)
    t_bs = _time_gpu(lambda: chol_b.solve(rhs_b), iters=5)
    out["btd_cholesky_solve_B16384_T500_d6_f64"] = {
        "cholesky_ms": t_b, "cholesky_algorithmic_GBps": bb * tb * 4 * 36 * 8 / t_b / 1e6,
        "cholesky_frac_of_hbm_peak": bb * tb * 4 * 36 * 8 / t_b / 1e6 / HBM_PEAK_GBS,
        "solve_ms": t_bs, "solve_algorithmic_GBps": bb * tb * (2 * 36 + 12) * 8 / t_bs / 1e6,
        "solve_frac_of_hbm_peak": bb * tb * (2 * 36 + 12) * 8 / t_bs / 1e6 / HBM_PEAK_GBS,
        "note": "SURVEY 8d bytes per block: cholesky 4 d^2 s, solve (2 d^2 + 2 d) s; one lane per series, natural order"}
    # the composite entry points at the same shape: fused one-lane-per-series sweeps since round 2
    kf_b = synthetic.kalman_filter_from(inp)
    t_post = _time_gpu(kf_b.posterior_state_space_model, iters=5)
    post_b = kf_b.posterior_state_space_model()
    t_kl = _time_gpu(lambda: post_b.kl_divergence(kf_b.prior_ssm), iters=5)
    out["composites_B16384_T500_d6_f64"] = {
        "posterior_state_space_model_ms": t_post, "kl_divergence_ms": t_kl,
        "posterior_algorithmic_GBps": bb * tb * (4 * 36 + 3 * 6 + 2) * 8 / t_post / 1e6,
        "kl_algorithmic_GBps": bb * tb * (4 * 36 + 2 * 6) * 8 / t_kl / 1e6,
        "note": "bytes per block: posterior reads A, cholQ, b, H, y and writes A', cholQ', b' ((4 d^2 + 3 d + 2) s); "
                "kl_divergence reads both chains ((4 d^2 + 2 d) s) and writes one scalar per series"}
    del inp, prec, chol_b, rhs_b, kf_b, post_b
    # config 4 shape: d=9 (3 x Matern-5/2, 3 outputs), 512 series per GPU (4096 over 8 GPUs), fp64
    inp = synthetic.make_ssm(512, 1000, (5, 5, 5), output_dim=3, dtype=torch.float64, device=dev)
    kf = synthetic.kalman_filter_from(inp)
    ms = _time_gpu(kf.log_likelihood, iters=5)
    post = kf.posterior_state_space_model()
    ms_kl = _time_gpu(lambda: post.kl_divergence(kf.prior_ssm), iters=3, warm=1)
    out["config4_d9_m3_B512_T1000_f64"] = {"loglik_ms": ms, "loglik_steps_per_s": 512 * 1000 / ms * 1e3,
                                          "kl_divergence_ms": ms_kl}
    # GPR log-likelihood with the kernel -> SSM generation fused into the sweep, at the headline shape
    # (B=1024, T=10000, Sum of two Matern-5/2 = d 6, fp64); input is (t, y, hyper-parameters): 16 B per step
    bsz, tn = 1024, 10000
    t_pts = torch.cumsum(0.05 + 0.05 * torch.empty(bsz, tn, dtype=torch.float64, device=dev).exponential_(1.0, generator=g), dim=-1)
    y_obs = torch.randn(bsz, tn, 1, dtype=torch.float64, device=dev, generator=g)
    parts = [mfa.Matern52(0.5 + 1.5 * torch.rand(bsz, dtype=torch.float64, device=dev, generator=g),
                          0.5 + 1.5 * torch.rand(bsz, dtype=torch.float64, device=dev, generator=g)) for _ in range(2)]
    gpr = mfa.GaussianProcessRegression((t_pts, y_obs), mfa.Sum(parts, jitter=1e-9),
                                        chol_obs_covariance=(0.1 ** 0.5) * torch.eye(1, dtype=torch.float64, device=dev))
    ms = _time_gpu(gpr.log_likelihood, iters=10)
    # the training step of the same model: forward + backward w.r.t. lengthscales, variances and the noise factor - both
    # directions with the kernel -> SSM step fused since round 4 (csrc/mf_gpr_grad.hpp)
    ls_t = [(0.5 + 1.5 * torch.rand(bsz, dtype=torch.float64, device=dev, generator=g)).requires_grad_(True) for _ in range(2)]
    var_t = [(0.5 + 1.5 * torch.rand(bsz, dtype=torch.float64, device=dev, generator=g)).requires_grad_(True) for _ in range(2)]
    chol_t = ((0.1 ** 0.5) * torch.eye(1, dtype=torch.float64, device=dev)).requires_grad_(True)

    def train_step_tgt(fused_backward=True):
        for x in ls_t + var_t + [chol_t]:
            x.grad = None
        kern = mfa.Sum([mfa.Matern52(l, v, jitter=1e-9) for l, v in zip(ls_t, var_t)], jitter=1e-9)
        model = mfa.GaussianProcessRegression((t_pts, y_obs), kern, chol_obs_covariance=chol_t)
        model.fused_backward = fused_backward
        model.log_likelihood().backward()

    ms_train = _time_gpu(train_step_tgt, iters=5, warm=2)
    ms_train_mat = _time_gpu(lambda: train_step_tgt(False), iters=3, warm=1)
    # the smoother of the same model: posterior chain with the kernel -> SSM step fused, against the materialised route
    ms_gpost = _time_gpu(gpr.posterior_state_space_model, iters=5, warm=2)
    gpr.fused_backward = False
    ms_gpost_mat = _time_gpu(gpr.posterior_state_space_model, iters=3, warm=1)
    gpr.fused_backward = True
    out["gpr_fused_matern52x2_B1024_T10000_d6_f64"] = {
        "ms": ms, "steps_per_s": bsz * tn / ms * 1e3, "training_step_ms": ms_train, "training_step_materialised_ms": ms_train_mat,
        "posterior_ms": ms_gpost, "posterior_materialised_ms": ms_gpost_mat,
        "note": "GaussianProcessRegression.log_likelihood through mf_gpr_matern_loglik (A_k, chol Q_k generated in registers); "
                "NOT the headline metric: the boundary differs (time points + hyper-parameters instead of SSM tensors).  "
                "training_step_ms: forward + backward w.r.t. every hyper-parameter, both directions fused "
                "(mf_gpr_matern_loglik_grad + mf_sde_matern_transitions_grad; round 3: 28.1 ms); training_step_materialised_ms: "
                "kernel tensors -> KalmanFilter -> streamed backward -> generator backward; posterior_ms: "
                "posterior_state_space_model through mf_gpr_matern_posterior_chain (fused forward for its summaries + emit pass "
                "generating the transitions) against kernel tensors -> KalmanFilter.posterior_state_space_model"}
    del t_pts, y_obs, gpr
    # config 4's MODEL: IndependentMultiOutput of three Matern-5/2 kernels (d = 9, 3 outputs), 512 series x 1000 points, from
    # (t, y, hyper-parameters): fused into the row kernel since round 3 (csrc/mf_row_gpr.hpp) against the materialised route
    bsz, tn = 512, 1000
    t_pts = torch.cumsum(0.05 + 0.05 * torch.empty(bsz, tn, dtype=torch.float64, device=dev).exponential_(1.0, generator=g), dim=-1)
    y_obs = torch.randn(bsz, tn, 3, dtype=torch.float64, device=dev, generator=g)
    parts = [mfa.Matern52(0.5 + 1.5 * torch.rand(bsz, dtype=torch.float64, device=dev, generator=g),
                          0.5 + 1.5 * torch.rand(bsz, dtype=torch.float64, device=dev, generator=g)) for _ in range(3)]
    gpr = mfa.GaussianProcessRegression((t_pts, y_obs), mfa.IndependentMultiOutput(parts, jitter=1e-9),
                                        chol_obs_covariance=(0.1 ** 0.5) * torch.eye(3, dtype=torch.float64, device=dev))
    ms = _time_gpu(gpr.log_likelihood, iters=10)
    gpr.fused = False
    ms_mat = _time_gpu(gpr.log_likelihood, iters=5)
    # the training step of the same model: forward + backward w.r.t. lengthscales, variances and the noise factor
    ls = [(0.5 + 1.5 * torch.rand(bsz, dtype=torch.float64, device=dev, generator=g)).requires_grad_(True) for _ in range(3)]
    vs = [(0.5 + 1.5 * torch.rand(bsz, dtype=torch.float64, device=dev, generator=g)).requires_grad_(True) for _ in range(3)]
    chol_r = ((0.1 ** 0.5) * torch.eye(3, dtype=torch.float64, device=dev)).requires_grad_(True)

    def train_step():
        for x in ls + vs + [chol_r]:
            x.grad = None
        kern = mfa.IndependentMultiOutput([mfa.Matern52(l, v, jitter=1e-9) for l, v in zip(ls, vs)], jitter=1e-9)
        mfa.GaussianProcessRegression((t_pts, y_obs), kern, chol_obs_covariance=chol_r).log_likelihood().backward()
    ms_train = _time_gpu(train_step, iters=5)
    out["config4_gpr_3xMatern52_3outputs_B512_T1000_d9_f64"] = {
        "fused_ms": ms, "materialised_ms": ms_mat, "fused_steps_per_s": bsz * tn / ms * 1e3, "training_step_ms": ms_train,
        "note": "GaussianProcessRegression.log_likelihood from (t, y, hyper-parameters): mf_gpr_matern_multi_loglik (every lane of the "
                "row kernel generates its row of chol Q_k / column of A_k) against mf_sde_matern_transitions + mf_kf_loglik; "
                "training_step_ms: log_likelihood forward + backward w.r.t. every hyper-parameter (HIP generator backward + "
                "Fisher-identity backward of the filter)"}
    del t_pts, y_obs, gpr
    # config 5: state_dim 64, T=2048, fp32, 32 spatial outputs, 8 series - the panel kernels (csrc/mf_panel.hpp, round 6: a workgroup of
    # four wavefronts per chunk, register-resident column panels, A operand through LDS, two workgroups per CU)
    bsz, tn, d, m = 8, 2048, 64, 32
    kf = synthetic.kalman_filter_from(synthetic.make_dense_ssm(bsz, tn, d, m, dtype=torch.float32, device=dev))
    ms = _time_gpu(kf.log_likelihood, iters=5)
    # the level-0 kernel alone: HIP events around its launch, on the stream it is launched on
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    ev = (ctypes.c_void_p(), ctypes.c_void_p())
    hip.hipEventCreate(ctypes.byref(ev[0])); hip.hipEventCreate(ctypes.byref(ev[1]))
    kf._prof_events = ev
    lvl0 = []
    for _ in range(6):
        kf.log_likelihood(); torch.cuda.synchronize()
        f = ctypes.c_float(); hip.hipEventElapsedTime(ctypes.byref(f), ev[0], ev[1]); lvl0.append(f.value)
    kf._prof_events = (None, None)
    ms0 = sorted(lvl0[1:])[len(lvl0[1:]) // 2]
    # two flop models, both printed: ALGORITHMIC 9 d^3 per step (what the plain natural-order recursion needs) and
    # EXECUTED 15 d^3 (what the time-partitioned elimination performs, incl. the spike's three extra products)
    alg, exe = 9.0 * d ** 3, 15.0 * d ** 3
    out["config5_loglik_d64_T2048_m32_B8_f32"] = {
        "ms": ms, "steps_per_s": bsz * tn / ms * 1e3, "level0_kernel_ms": ms0,
        "algorithmic_TFLOPs": bsz * tn * alg / ms / 1e9, "frac_of_f32_mfma_peak_algorithmic": bsz * tn * alg / ms / 1e9 / 157.3,
        "executed_TFLOPs": bsz * tn * exe / ms / 1e9, "frac_of_f32_mfma_peak_executed": bsz * tn * exe / ms / 1e9 / 157.3,
        "level0_algorithmic_TFLOPs": bsz * tn * alg / ms0 / 1e9, "level0_frac_of_f32_mfma_peak_algorithmic": bsz * tn * alg / ms0 / 1e9 / 157.3,
        "note": "whole log_likelihood() incl. reduction levels, and its level-0 kernel alone (HIP events); flop models 9 d^3 (algorithmic, "
                "SURVEY 8d) and 15 d^3 (executed by the partitioned elimination); rounds 2-5 (LDS-tile engine): 2.7 ms / level 0 2.25 ms, "
                "matrix pipe 20.5 % busy; MFMA busy counters of this kernel: profiles/r06_panel_*_pmc_summary.txt"}
    # the same shape in float64 - the reference's default float (state_space_model.py:294, models/spatio_temporal_variational.py:45-85):
    # the panel kernels on v_mfma_f64_16x16x4_f64, one workgroup per CU (fp64 beyond d = 32 did not exist before round 6)
    kf64 = synthetic.kalman_filter_from(synthetic.make_dense_ssm(bsz, tn, d, m, dtype=torch.float64, device=dev))
    ms64 = _time_gpu(kf64.log_likelihood, iters=5)
    out["config5_loglik_d64_T2048_m32_B8_f64"] = {
        "ms": ms64, "steps_per_s": bsz * tn / ms64 * 1e3, "algorithmic_TFLOPs": bsz * tn * alg / ms64 / 1e9,
        "frac_of_f64_mfma_peak_algorithmic": bsz * tn * alg / ms64 / 1e9 / 78.6,
        "note": "float64 at config 5's shape; peak 78.6 TFLOP/s (v_mfma_f64_16x16x4_f64, nominal)"}
    del kf64
    # the operators around it at the same shape: partitioned in time on the same tile engine since round 3 (csrc/mf_bigpar_impl.hpp)
    prec = kf.prior_ssm.precision
    t_chol = _time_gpu(lambda: mfa.SymmetricBlockTriDiagonal(prec.block_diagonal, prec.block_sub_diagonal).cholesky, iters=3, warm=1)
    t_post = _time_gpu(kf.posterior_state_space_model, iters=3, warm=1)
    out["config5_operators_d64_T2048_m32_B8_f32"] = {
        "cholesky_ms": t_chol, "posterior_state_space_model_ms": t_post,
        "note": "SymmetricBlockTriDiagonal.cholesky of the prior precision and KalmanFilter.posterior_state_space_model (precision "
                "assembly + U D U^T + chain means), time-partitioned (round 2: 54 ms / 81 ms with one workgroup per series)"}
    del kf, prec
    # between the row kernels (d <= 15) and the tile engine: the wave kernels (csrc/mf_wave.hpp, round 5) at the shape VERDICT r04
    # names - B = 512, T = 1000, fp64, m = 1.  Flop model 15 d^3 (executed by the partitioned elimination); the fp64 matrix rate
    # this chip sustains is 63.8 TFLOP/s (scripts/micro/mfma_rate.hip: v_mfma_f64_16x16x4_f64 at 4 wavefronts per SIMD)
    wave = {}
    for dd in (15, 16, 24, 32):
        kfw = synthetic.kalman_filter_from(synthetic.make_dense_ssm(512, 1000, dd, 1, dtype=torch.float64, device=dev))
        msw = _time_gpu(kfw.log_likelihood, iters=5, warm=2)
        wave[f"d{dd}"] = {"ms": msw, "steps_per_s": 512 * 1000 / msw * 1e3, "executed_TFLOPs_15d3": 512 * 1000 * 15.0 * dd ** 3 / msw / 1e9,
                          "algorithmic_GBps": 512 * 1000 * synthetic.loglik_bytes_per_step(dd, 1, 8) / msw / 1e6}
        if dd in (16, 32):
            # the operators of the same dimensions (csrc/mf_wave_ops.hpp: one wavefront per series walks the chain)
            precw = kfw.prior_ssm.precision
            symw = mfa.SymmetricBlockTriDiagonal(precw.block_diagonal, precw.block_sub_diagonal)
            loww = symw.cholesky
            rhsw = torch.randn(512, 1000, dd, dtype=torch.float64, device=dev, generator=g)
            wave[f"d{dd}"]["operators_ms"] = {
                "precision": _time_gpu(lambda: kfw.prior_ssm.precision, iters=3, warm=1),
                "cholesky": _time_gpu(lambda: mfa.SymmetricBlockTriDiagonal(precw.block_diagonal, precw.block_sub_diagonal).cholesky, iters=3, warm=1),
                "solve": _time_gpu(lambda: loww.solve(rhsw), iters=3, warm=1),
                "block_diagonal_of_inverse": _time_gpu(loww.block_diagonal_of_inverse, iters=3, warm=1),
                "posterior_state_space_model": _time_gpu(kfw.posterior_state_space_model, iters=3, warm=1)}
            # round 6: solve^T, the prior's marginal means and KL(posterior || prior) (one walk per (series, chunk) on the tiles)
            postw = kfw.posterior_state_space_model()
            wave[f"d{dd}"]["operators_ms"].update({
                "solve_transposed": _time_gpu(lambda: loww.solve(rhsw, transpose_left=True), iters=3, warm=1),
                "marginal_means": _time_gpu(lambda: kfw.prior_ssm.marginal_means, iters=3, warm=1),
                "kl_divergence": _time_gpu(lambda: postw.kl_divergence(kfw.prior_ssm), iters=3, warm=1)})
            del precw, symw, loww, rhsw, postw
        del kfw
    wave["note"] = ("KalmanFilter.log_likelihood B=512 T=1000 m=1 fp64: d = 15 row kernels (mf_row.hpp), d >= 16 wave kernels "
                    "(one wavefront per chunk, register tiles in the MFMA accumulator layout; the pivot's Cholesky factor and the next "
                    "chol(Q)'s inverse share one DPP pass); round 4: d=16 18.1 ms, d=32 57.9 ms.  operators_ms: one wavefront per series "
                    "(round 4 at d=16 / d=32: precision 10.9 / 43.9, cholesky 5.4 / 26.6, solve 8.1 / 22.9, inverse blocks 8.3 / 25.4, "
                    "posterior_state_space_model 25.4 / 102.8 ms; round 5: solve 2.0 / 3.4, kl_divergence 6.8 / 12.0)")
    out["wave_kernels_B512_T1000_m1_f64"] = wave
    # reverse mode through the OPERATORS: the chain the reference's CVI models differentiate,
    # dist_p.precision -> naturals_to_ssm_params -> kl_divergence (models/variational_cvi.py:105-136), few long series
    from markovflow_amd import ssm_gaussian_transformations as G

    bsz, tn = 64, 10000
    t_pts = torch.cumsum(0.05 + 0.05 * torch.empty(bsz, tn, dtype=torch.float64, device=dev).exponential_(1.0, generator=g), dim=-1)
    nat1 = torch.randn(bsz, tn, 1, dtype=torch.float64, device=dev, generator=g).requires_grad_(True)
    nat2 = (-0.5 * (0.5 + torch.rand(bsz, tn, 1, 1, dtype=torch.float64, device=dev, generator=g))).requires_grad_(True)
    ls_c = [(0.5 + 1.5 * torch.rand(bsz, dtype=torch.float64, device=dev, generator=g)).requires_grad_(True) for _ in range(2)]
    var_c = [(0.5 + 1.5 * torch.rand(bsz, dtype=torch.float64, device=dev, generator=g)).requires_grad_(True) for _ in range(2)]

    def cvi_kl():
        kern = mfa.Sum([mfa.Matern52(l, v, jitter=1e-9) for l, v in zip(ls_c, var_c)], jitter=1e-9)
        dist_p = kern.state_space_model(t_pts)
        prec = dist_p.precision
        h = kern.generate_emission_model(t_pts).emission_matrix
        theta_lin = (h.transpose(-1, -2) @ nat1[..., None])[..., 0]
        theta_diag = -0.5 * prec.block_diagonal + h.transpose(-1, -2) @ nat2 @ h
        a_s, offsets, chol_p0, chol_q, mu0 = G.naturals_to_ssm_params(theta_lin, theta_diag, -prec.block_sub_diagonal)
        return torch.sum(mfa.StateSpaceModel(mu0, chol_p0, a_s, offsets, chol_q).kl_divergence(dist_p))

    fwd, bwd = [], []
    for i in range(4):
        for x in ls_c + var_c + [nat1, nat2]:
            x.grad = None
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record(); kl = cvi_kl(); e1.record(); kl.backward(); e2.record()
        torch.cuda.synchronize()
        if i:
            fwd.append(e0.elapsed_time(e1)); bwd.append(e1.elapsed_time(e2))
    f_ms, b_ms = sorted(fwd)[len(fwd) // 2], sorted(bwd)[len(bwd) // 2]
    out["cvi_chain_precision_naturals_kl_B64_T10000_d6_f64"] = {
        "forward_ms": f_ms, "backward_ms": b_ms, "backward_over_forward": b_ms / f_ms,
        "note": "KL(q || p) with q built as variational_cvi.py:105-136 builds it (dist_p.precision -> naturals_to_ssm_params), "
                "gradients w.r.t. lengthscales, variances and sites.  The adjoints of cholesky and block_diagonal_of_inverse are HIP "
                "(mf_btd_cholesky_grad / mf_btd_diag_of_inverse_grad: local kernels + the congruence scan, parallel in time); round 4: a "
                "Python loop over the T blocks"}
    # reverse mode through the OPERATORS at the reference's largest tested shape (d = 30, T = 1001, one chain;
    # tests/unit/test_ssm_gaussian_transformations.py:40-46): cholesky -> block_diagonal_of_inverse (+ sub-diagonal blocks) and back.
    # Round 6 (csrc/mf_adj.hip, register MFMA tiles): local terms per block + congruence scans, parallel in time; rounds 4-5: a Python
    # loop over the blocks
    n30, d30 = 1001, 30
    ld = torch.tril((4.0 / d30) * 0.3 * torch.randn(4, n30, d30, d30, dtype=torch.float64, device=dev, generator=g), -1) + torch.diag_embed(
        1 + torch.rand(4, n30, d30, dtype=torch.float64, device=dev, generator=g))
    ls = (2.0 / d30) * 0.3 * torch.randn(4, n30 - 1, d30, d30, dtype=torch.float64, device=dev, generator=g)
    dg = ld @ ld.transpose(-1, -2)
    dg[:, 1:] += ls @ ls.transpose(-1, -2)
    sb = ls @ ld[:, :-1].transpose(-1, -2)
    dg.requires_grad_(True); sb.requires_grad_(True)
    wd = torch.randn(4, n30, d30, d30, dtype=torch.float64, device=dev, generator=g)
    fwd30, bwd30 = [], []
    for i in range(4):
        dg.grad = sb.grad = None
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        inv_d, inv_s = mfa.SymmetricBlockTriDiagonal(dg, sb).cholesky._diag_and_sub_of_inverse(want_sub=True)
        loss = torch.sum(inv_d * wd) + torch.sum(inv_s)
        e1.record(); loss.backward(); e2.record()
        torch.cuda.synchronize()
        if i:
            fwd30.append(e0.elapsed_time(e1)); bwd30.append(e1.elapsed_time(e2))
    f30, b30 = sorted(fwd30)[len(fwd30) // 2], sorted(bwd30)[len(bwd30) // 2]
    out["operator_adjoints_d30_T1001_B4_f64"] = {
        "forward_ms": f30, "backward_ms": b30, "backward_over_forward": b30 / f30,
        "note": "SymmetricBlockTriDiagonal.cholesky -> block_diagonal_of_inverse (with sub-diagonal blocks), forward and reverse mode, "
                "four chains of the reference's largest tested operator shape"}
    del ld, ls, dg, sb, wd
    return out


def smoother_and_backward(kf, inputs, bsz, tn, d, m, esz):
    """The other half of "filter / smoother" at the headline shape: posterior_state_space_model() and the
    backward of log_likelihood() on the SAME resident inputs, with algorithmic GB/s - bytes every implementation must move:
    posterior reads A, cholQ, b, H, y and writes A', cholQ', b' ((4 d^2 + 3 d + m d + m) s per step); the backward reads the
    inputs and writes one gradient per input element ((4 d^2 + 2 d + 2 m d + 2 m) s)."""
    import torch

    import markovflow_amd as mfa

    out = {}
    ev = HipEvents()
    kf._post_prof_events = (ev.start, ev.stop)
    # on its own: no filter pass to start from
    kf._POST_FROM_FILTER = False
    kf.invalidate_filter_cache()
    ms = _time_gpu(kf.posterior_state_space_model, iters=5)
    kern = []
    for _ in range(3):
        kf.posterior_state_space_model()
        kern.append(ev.elapsed_ms())
    # the smoother after the filter: log_likelihood() on the same tensors left its chunk summaries behind
    kf._POST_FROM_FILTER = True
    kf.log_likelihood()
    ms_after = _time_gpu(kf.posterior_state_space_model, iters=5)
    kf._post_prof_events = (None, None)
    b_post = bsz * tn * (4 * d * d + 3 * d + m * d + m) * esz
    out["posterior_TGT"] = {
        "ms": ms, "kernels_ms": sum(kern) / len(kern), "algorithmic_GBps": b_post / ms / 1e6,
        "frac_of_hbm_peak": b_post / ms / 1e6 / HBM_PEAK_GBS,
        "ms_after_log_likelihood": ms_after, "algorithmic_GBps_after_log_likelihood": b_post / ms_after / 1e6,
        "frac_of_hbm_peak_after_log_likelihood": b_post / ms_after / 1e6 / HBM_PEAK_GBS,
        "kernels": "mf::post_lds_kernel<MODE=0> (reversed elimination per chunk) + mf::post_scan_kernel + "
                   "mf::post_lds_kernel<MODE=1>; round 3: mf_ssm_precision + parallel-in-time U D U^T + affine scan, 11.2 ms.  "
                   "ms_after_log_likelihood: mf::k0_scan_kernel over the summaries the log-likelihood left + the emit pass alone "
                   "(mf_kf_posterior_chain_from_filter)",
        "note": f"KalmanFilter.posterior_state_space_model B={bsz} T={tn} d={d} m={m}: all five tensors of the posterior chain"}
    p = kf.prior_ssm
    leaves = [t.detach().clone().requires_grad_(True) for t in (p.initial_mean, p.cholesky_initial_covariance, p.state_transitions,
                                                                p.state_offsets, p.cholesky_process_covariances)]
    kfg = mfa.KalmanFilter(mfa.StateSpaceModel(*leaves), kf.emission, kf.observations, inputs["cholR"])
    fwd, bwd = [], []
    for i in range(4):
        for x in leaves:
            x.grad = None
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record(); ll = kfg.log_likelihood(); e1.record(); ll.backward(); e2.record()
        torch.cuda.synchronize()
        if i:
            fwd.append(e0.elapsed_time(e1)); bwd.append(e1.elapsed_time(e2))
    b_bwd = bsz * tn * (4 * d * d + 2 * d + 2 * m * d + 2 * m) * esz
    f_ms, b_ms = sorted(fwd)[len(fwd) // 2], sorted(bwd)[len(bwd) // 2]
    # the kernels of the backward alone (hipEvents recorded by the library around them)
    from markovflow_amd import kalman_filter as kfm
    evg = HipEvents()
    kfm._grad_prof_events = (evg.start, evg.stop)
    kern = []
    for _ in range(3):
        for x in leaves:
            x.grad = None
        kfg.log_likelihood().backward()
        kern.append(evg.elapsed_ms())
    kfm._grad_prof_events = (None, None)
    out["loglik_backward_TGT"] = {
        "forward_ms": f_ms, "backward_ms": b_ms, "backward_kernels_ms": sum(kern) / len(kern), "backward_over_forward": b_ms / f_ms,
        "backward_algorithmic_GBps": b_bwd / b_ms / 1e6, "backward_frac_of_hbm_peak": b_bwd / b_ms / 1e6 / HBM_PEAK_GBS,
        "kernels": "mf::k0_scan_kernel x2 (compositions of the forward evaluation's chunk summaries) + mf::post_lds_kernel<MODE=2> "
                   "(emit pass: chol(Q'), b') + mf::grad_lds_kernel (forward pass, smoothed marginals in registers)",
        "note": "KalmanFilter.log_likelihood().backward() w.r.t. mu0, cholP0, A, b, cholQ (Fisher's identity, streamed: "
                "csrc/mf_grad_lds.hpp); round 3: 21.8 ms, start of round 4: 17.1 ms"}
    return out


def config4_rank(args, dev, dist, rank, world, dtype) -> int:
    """BASELINE config 4's per-GPU shard as the timed step (so that a scaling run measures it too): 512 series x 1000 points,
    IndependentMultiOutput(3 x Matern-5/2) with 3 outputs (d = 9); step = log-likelihood of the shard + KL(q || prior) of the
    shard with q the exact posterior chain, combined by sharded_elbo - ONE scalar all-reduce, as models/sparse_variational.py:178-192
    of the reference sums its terms.  Rank 0 prints one JSON line with its own metric name (never the headline's)."""
    import torch

    from markovflow_amd import distributed as mfd
    from markovflow_amd import synthetic

    bsz, tn, d, m = 512, 1000, 9, 3
    inputs = synthetic.make_ssm(bsz, tn, (5, 5, 5), output_dim=3, dtype=dtype, device=dev, seed=synthetic.DEFAULT_SEED + rank)
    kf = synthetic.kalman_filter_from(inputs)
    q = kf.posterior_state_space_model()

    def step():
        return mfd.sharded_elbo(kf.log_likelihood(), q.kl_divergence(kf.prior_ssm))

    for _ in range(args.warmup):
        val = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        val = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ranks_seen = 1
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        ones = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        ranks_seen = int(ones.item())
    if rank == 0:
        print(json.dumps({
            "metric": "config 4 shard: log-lik + KL steps/sec (BxT) at d=9, m=3", "value": world * bsz * tn * args.steps / elapsed,
            "unit": "steps/s", "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"KalmanFilter.log_likelihood + StateSpaceModel.kl_divergence, B={bsz}/GPU T={tn} d={d} m={m} "
                                   f"{args.dtype} (BASELINE config 4: 4096 series over 8 GPUs), sharded_elbo",
                       "series_per_gpu": bsz, "time_points": tn, "state_dim": d, "output_dim": m,
                       "parallelism": f"batch-sharded x{world}, one scalar RCCL all-reduce"},
            "elbo_like_value": float(val.item())}))
    if dist is not None:
        dist.destroy_process_group()
    return 0


def launch_ranks(args) -> int:
    """`bench.py --gpus N` without a launcher: THIS process stays GPU-free - it imports neither torch nor any HIP library and
    counts GPUs from sysfs - and starts N fresh rank processes through torch.distributed.run, one per GPU, rendezvous on
    127.0.0.1; rank 0's JSON line goes straight to our stdout and we exit with the children's return code (child processes
    only: never exec over, or fork from, a process that touched the GPU)."""
    import socket
    import subprocess

    if not args.dry_launch:
        have = visible_gpus()
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} requested but only {have} GPU(s) are visible", file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    # dmabuf IPC: this image's host driver supports no legacy IPC handles - without it RCCL's communicator setup (and any
    # CUDA-tensor sharing across processes) fails with `hipIpcGetMemHandle: invalid argument` (environment notes of the pool)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8" if not args.dry_launch else "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    assert "torch" not in sys.modules, "the launching process must not import torch (it has to stay GPU-free)"
    return subprocess.call(cmd, env=env)


def dry_rank(args) -> int:
    """A rank of `--dry-launch`: the launch plumbing (spawn, port, argv, environment, stdout discipline, return code) and the
    sharding arithmetic with a gloo group instead of RCCL and no kernels."""
    import torch
    import torch.distributed as dist

    from markovflow_amd import distributed as mfd

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} does not match the launcher's WORLD_SIZE={world}", file=sys.stderr)
        return 2
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    with _StdoutToStderr():
        dist.init_process_group(backend="gloo")
    ones = torch.ones(1, dtype=torch.float64)
    dist.all_reduce(ones)
    per_gpu = 512 if args.workload == "config4" else args.batch        # (config 4: 4096 series over 8 GPUs = 512 per shard)
    lo, hi = mfd.shard_bounds(per_gpu * world, rank, world)             # weak scaling: every rank owns per_gpu series
    owned = torch.tensor([float(hi - lo)], dtype=torch.float64)
    dist.all_reduce(owned)
    # the sharded reductions the timed step ends in, on the rank's share of a known total (gloo instead of RCCL)
    share = torch.full((), float(hi - lo), dtype=torch.float64)
    total = (mfd.sharded_elbo(share, torch.zeros((), dtype=torch.float64)) if args.workload == "config4"
             else mfd.all_reduce_sum(share.clone()))
    if rank == 0:
        print(json.dumps({"dry_launch": True, "n_gpus": dist.get_world_size(), "ranks_seen": int(ones.item()),
                          "workload": args.workload, "sharded_total": float(total),
                          "series_total": int(owned.item()), "series_per_gpu": per_gpu,
                          "ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}))
    dist.destroy_process_group()
    return 0


class _StdoutToStderr:
    """File descriptor 1 points at stderr while this is active: RCCL prints a version banner to STDOUT when its communicator
    is created, and rank 0's stdout must carry exactly one JSON line."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def main():
    args = parse()
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ      # under torch.distributed.run
    if args.gpus > 1 and not launched:
        sys.exit(launch_ranks(args))
    if args.dry_launch:
        if not launched:
            print("bench.py: --dry-launch needs --gpus N > 1 (or a torch.distributed.run launcher)", file=sys.stderr)
            sys.exit(2)
        sys.exit(dry_rank(args))
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if launched and world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} does not match the launcher's WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if launched:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        with _StdoutToStderr():
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))   # RCCL on ROCm
            probe = torch.ones(1, device=torch.device("cuda", local_rank))
            dist.all_reduce(probe)                                # the communicator (and RCCL's banner) is created here
            torch.cuda.synchronize()
        world = dist.get_world_size()
    else:
        dist = None
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if dist is not None else 0)

    from markovflow_amd import _lib
    from markovflow_amd import distributed as mfd
    from markovflow_amd import synthetic

    dtype = torch.float64 if args.dtype == "f64" else torch.float32
    if args.workload == "config4":
        sys.exit(config4_rank(args, dev, dist, rank, world, dtype))
    bsz, tn, d, m = args.batch, args.time_points, 6, 1
    # every rank generates its own series (never replicated): weak scaling over the batch axis
    inputs = synthetic.make_ssm(bsz, tn, (5, 5), dtype=dtype, device=dev, seed=synthetic.DEFAULT_SEED + rank)
    kf = synthetic.kalman_filter_from(inputs)
    kf._chunks = args.chunks
    ev = HipEvents()
    kf._prof_events = (ev.start, ev.stop)

    def step():
        # this rank's series + the path's only exchange: one all-reduce of the scalar (a no-op at N=1)
        return mfd.sharded_log_likelihood(kf)

    for _ in range(args.warmup):
        ll = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    # per-step durations from events recorded on the launch stream inside the timed region (no synchronisation between steps):
    # the JSON line carries min / median beside the wall-clock mean, 20-100 steps of ~1.5 ms are a small sample
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        marks[i].record()
        ll = step()
    marks[args.steps].record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    # dominant-kernel duration: separate short loop so that reading the events does not perturb the timed region
    kernel_ms = []
    for _ in range(min(args.steps, 10)):
        step()
        kernel_ms.append(ev.elapsed_ms())
    kern_avg_ms = sum(kernel_ms) / len(kernel_ms)

    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        ones = torch.ones(1, dtype=torch.float64, device=dev)      # proof that RCCL saw every rank
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        ranks_seen = int(ones.item())
    else:
        ranks_seen = 1

    esz = 8 if dtype == torch.float64 else 4
    bytes_per_step = synthetic.loglik_bytes_per_step(d, m, esz)
    units_per_launch = bsz * tn
    achieved_gbs = units_per_launch * bytes_per_step / (kern_avg_ms * 1e-3) / 1e9
    value = world * bsz * tn * args.steps / elapsed

    # HBM-side bytes per launch measured with rocprofv3 PMC passes (cannot be collected from inside this
    # process): quoted from profiles/hbm_traffic.json when it holds this exact workload, else null
    traffic, traffic_src = None, None
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as fh:
            entry = json.load(fh).get(f"kf_loglik B={bsz} T={tn} d={d} m={m} {args.dtype}")
        # only when the counters were collected on exactly THESE library sources (sha256 over markovflow_amd/csrc + the header,
        # scripts/csrc_hash.py): a stale figure is worse than none
        from scripts.csrc_hash import csrc_hash

        if entry and args.chunks == 0 and entry.get("csrc_sha256") == csrc_hash(ROOT):
            traffic, traffic_src = entry["traffic_bytes"], entry["source"]
    except OSError:
        pass

    result = {
        "metric": "Kalman log-lik steps/sec (BxT) at d=6",
        "value": value,
        "unit": "steps/s",
        "n_gpus": world,
        "ranks_seen": ranks_seen,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "ms_per_step_min": step_ms[0], "ms_per_step_median": step_ms[len(step_ms) // 2],
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {
            "workload": f"KalmanFilter.log_likelihood B={bsz}/GPU T={tn} d={d} m={m} {args.dtype} "
                        "(sum of two Matern-5/2, north-star target config)",
            "series_per_gpu": bsz, "time_points": tn, "state_dim": d, "output_dim": m,
            "parallelism": f"batch-sharded x{world}, one scalar RCCL all-reduce",
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
            "kernel": "mf::kf_chunk_lds_kernel (level 0 of the partitioned elimination)", "kernel_ms": kern_avg_ms,
            "algorithmic_bytes_per_launch": units_per_launch * bytes_per_step,
        },
        "log_likelihood": float(ll.item()),
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        base, cpu_out = cpu_baseline(inputs, args.cpu_seconds)
        result["cpu_baseline"] = base
        # the checker, not the thing measured: GPU per-series values agree with the CPU port on the sample
        import numpy as np

        per = kf._log_likelihood_per_series()[: cpu_out.shape[0]].cpu().numpy().astype(np.float64)
        cst = -0.5 * np.log(2 * np.pi) * tn + 0.5 * tn * np.log(1.0 / 0.1)
        rel = float(np.max(np.abs(per + cst - cpu_out) / np.abs(cpu_out)))
        result["cpu_baseline"]["max_rel_diff_vs_gpu"] = rel
    if rank == 0 and world == 1 and not args.no_other_configs:
        try:
            result["smoother_and_backward"] = smoother_and_backward(kf, inputs, bsz, tn, d, m, esz)
        except Exception as exc:
            result["smoother_and_backward"] = {"error": repr(exc)}
        del kf, inputs
        try:
            result["other_configs"] = other_configs(dev)
        except Exception as exc:   # the headline line must still be printed
            result["other_configs"] = {"error": repr(exc)}
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
