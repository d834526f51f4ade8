"""
The batch-sharded path on real GPUs: ranks started by torch.distributed.run, backend "nccl" (RCCL), HIP kernels on every
rank's shard, one scalar all-reduce (markovflow/kalman_filter.py:255 is the sum being distributed).  World size 1 always;
world size 2 when the box has two GPUs.  Also: `bench.py --gpus N` starts its own ranks and reports the world RCCL saw.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import numpy_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "helpers"))
from nccl_worker import inputs  # noqa: E402
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["OMP_NUM_THREADS"] = "4"
    return env


@pytest.mark.parametrize("world", [1, 2])
@pytest.mark.parametrize("bsz,t,d,m", [(5, 40, 3, 1), (64, 300, 6, 1), (3, 33, 9, 3)])
def test_sharded_log_likelihood_nccl(tmp_path, world, bsz, t, d, m):
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs")
    out = tmp_path / "res.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "helpers", "nccl_worker.py"),
           str(out), str(bsz), str(t), str(d), str(m), "11"]
    subprocess.run(cmd, check=True, env=_env(), cwd=ROOT, timeout=600)
    res = json.loads(out.read_text())
    assert res["world"] == world and res["ranks_seen"] == world and res["backend"] == "nccl"
    full = inputs(bsz, t, d, m, 11)
    r_inv = np.linalg.inv(0.25 * np.eye(m))
    expect = float(O.kf_log_likelihood(r_inv=r_inv, **full))
    assert res["total"] == pytest.approx(expect, rel=1e-9)
    assert all(v == res["total"] for v in res["per_rank_totals"])          # identical on every rank
    post = O.kf_posterior_ssm(r_inv=r_inv, **full)
    prior = tuple(full[k] for k in ("mu0", "chol_p0", "a_s", "b_s", "chol_q"))
    assert res["kl_total"] == pytest.approx(float(np.sum(O.ssm_kl_divergence(post, prior))), rel=1e-8)


@pytest.mark.parametrize("gpus", [1, 2])
def test_bench_starts_its_own_ranks(gpus):
    """`python bench.py --gpus N` (no launcher): N rank processes, n_gpus == ranks_seen == N on the JSON line."""
    if torch.cuda.device_count() < gpus:
        pytest.skip(f"needs {gpus} GPUs")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "3", "--warmup", "1", "--batch", "64",
           "--time-points", "500", "--no-cpu-baseline", "--no-other-configs"]
    proc = subprocess.run(cmd, check=True, env=_env(), cwd=ROOT, timeout=900, capture_output=True, text=True)
    line = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == gpus and line["ranks_seen"] == gpus
    assert line["value"] > 0 and line["scaling"] == "weak"


def test_bench_under_the_drivers_launcher_world1():
    """The driver's own command line for N > 1, at N = 1: torch.distributed.run -> RCCL process group of one rank."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--batch", "64", "--time-points", "500", "--no-cpu-baseline", "--no-other-configs"]
    proc = subprocess.run(cmd, check=True, env=_env(), cwd=ROOT, timeout=900, capture_output=True, text=True)
    line = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["ranks_seen"] == 1
