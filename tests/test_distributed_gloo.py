"""
Multi-process (world sizes 2, 4 and 8, gloo, CPU) test of the batch-sharded path of markovflow_amd.distributed.

The HIP kernels cannot run here, so every rank's LOCAL log-likelihood comes from the numpy oracle (the
checker); what is under test is the host logic of the sharded path: contiguous, non-overlapping, exhaustive
shards (including uneven and empty ones) and the single scalar all-reduce that every rank must agree on.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from markovflow_amd import distributed as mfd
from oracle import numpy_oracle as O


def _inputs(bsz, t=9, d=3, m=1, seed=5):
    rng = np.random.default_rng(seed)
    return dict(
        mu0=rng.normal(size=(bsz, d)),
        chol_p0=np.tril(0.1 * rng.normal(size=(bsz, d, d))) + np.eye(d),
        a_s=0.8 * np.eye(d) + 0.05 * rng.normal(size=(bsz, t - 1, d, d)),
        b_s=0.1 * rng.normal(size=(bsz, t - 1, d)),
        chol_q=np.tril(0.1 * rng.normal(size=(bsz, t - 1, d, d))) + 0.5 * np.eye(d),
        h=rng.normal(size=(bsz, t, m, d)),
        y=rng.normal(size=(bsz, t, m)),
    )


class _OracleBackedFilter:
    """Stands in for a KalmanFilter that holds one rank's series: log_likelihood() of the local shard."""

    def __init__(self, arrays):
        self.arrays = arrays

    def log_likelihood(self):
        if self.arrays["mu0"].shape[0] == 0:
            return torch.zeros((), dtype=torch.float64)
        return torch.tensor(O.kf_log_likelihood(r_inv=np.array([[4.0]]), **self.arrays), dtype=torch.float64)


def _worker(rank, world, port, bsz, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = _inputs(bsz)
        lo, hi = mfd.shard_bounds(bsz, rank, world)
        local = {k: v[lo:hi] for k, v in full.items()}
        assert torch.equal(mfd.shard_batch(torch.arange(bsz), rank, world), torch.arange(lo, hi))
        total = mfd.sharded_log_likelihood(_OracleBackedFilter(local))
        results[rank] = (lo, hi, float(total))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,bsz", [(2, 6), (2, 5), (2, 1), (4, 5), (4, 3), (8, 11), (8, 3)])
def test_sharded_log_likelihood_gloo(world, bsz):
    """world sizes 2, 4, 8 with even, uneven and EMPTY shards (bsz < world): the 8-GPU node's layout, on CPU"""
    with mp.Manager() as manager:
        results = manager.dict()
        mp.spawn(_worker, args=(world, _free_port(), bsz, results), nprocs=world, join=True)
        results = dict(results)
    expect = float(O.kf_log_likelihood(r_inv=np.array([[4.0]]), **_inputs(bsz)))
    assert sorted(results) == list(range(world))
    bounds = [results[r][:2] for r in range(world)]                      # contiguous, in rank order, exhaustive
    assert bounds[0][0] == 0 and bounds[-1][1] == bsz
    assert all(bounds[r][1] == bounds[r + 1][0] for r in range(world - 1))
    for rank in range(world):
        assert results[rank][2] == pytest.approx(expect, rel=1e-12)
    assert len({results[r][2] for r in range(world)}) == 1               # bit-identical on every rank


@pytest.mark.parametrize("n,world", [(0, 1), (1, 8), (7, 8), (8, 8), (1024, 8), (4099, 8), (5, 2)])
def test_shard_bounds_partition_the_batch(n, world):
    cover = []
    for r in range(world):
        lo, hi = mfd.shard_bounds(n, r, world)
        assert 0 <= lo <= hi <= n and hi - lo in (n // world, n // world + 1)
        cover += list(range(lo, hi))
    assert cover == list(range(n))
    with pytest.raises(ValueError):
        mfd.shard_bounds(n, world, world)


def test_all_reduce_is_identity_without_process_group():
    x = torch.tensor(3.5, dtype=torch.float64)
    assert mfd.all_reduce_sum(x) is x and float(x) == 3.5


class _OracleBackedChain:
    """Stands in for a StateSpaceModel that holds one rank's series: kl_divergence() per local series from the numpy oracle."""

    def __init__(self, params):
        self.params = params

    def kl_divergence(self, other):
        if self.params[0].shape[0] == 0:
            return torch.zeros(0, dtype=torch.float64)
        return torch.tensor(O.ssm_kl_divergence(self.params, other.params), dtype=torch.float64)


def _elbo_worker(rank, world, port, bsz, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q, p = _inputs(bsz, seed=7), _inputs(bsz, seed=8)
        lo, hi = mfd.shard_bounds(bsz, rank, world)
        names = ("mu0", "chol_p0", "a_s", "b_s", "chol_q")
        ql = _OracleBackedChain(tuple(q[k][lo:hi] for k in names))
        pl = _OracleBackedChain(tuple(p[k][lo:hi] for k in names))
        kl = mfd.sharded_kl_divergence(ql, pl)
        ell = torch.tensor(np.arange(bsz, dtype=np.float64)[lo:hi] * 0.25)        # any per-series expectation
        elbo = mfd.sharded_elbo(ell, ql.kl_divergence(pl))
        # with a differentiable local term the total keeps THIS rank's graph: same value, local gradient
        scale = torch.ones(hi - lo, dtype=torch.float64, requires_grad=True)
        elbo_g = mfd.sharded_elbo(ell * scale, ql.kl_divergence(pl))
        elbo_g.backward()
        assert float(elbo_g) == float(elbo) and torch.equal(scale.grad, ell)
        # awkward magnitudes: `local + (total - local)` would round differently on every rank
        big = torch.tensor([1e8 / 3.0 * (rank + 1)], dtype=torch.float64, requires_grad=True)
        elbo_big = mfd.sharded_elbo(big * 1.0000001, torch.zeros(1, dtype=torch.float64))
        results[rank] = (float(kl), float(elbo), float(elbo_g), float(elbo_big))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,bsz", [(2, 6), (2, 3), (2, 1), (4, 6), (4, 2), (8, 12), (8, 5)])
def test_sharded_kl_and_elbo_gloo(world, bsz):
    """BASELINE config 4's exchange step: the ELBO of a batch sharded over the ranks is one scalar all-reduce (world sizes
    2, 4, 8; uneven and empty shards)."""
    with mp.Manager() as manager:
        results = manager.dict()
        mp.spawn(_elbo_worker, args=(world, _free_port(), bsz, results), nprocs=world, join=True)
        results = dict(results)
    names = ("mu0", "chol_p0", "a_s", "b_s", "chol_q")
    q, p = _inputs(bsz, seed=7), _inputs(bsz, seed=8)
    kl = float(np.sum(O.ssm_kl_divergence(tuple(q[k] for k in names), tuple(p[k] for k in names))))
    ell = float(np.sum(np.arange(bsz) * 0.25))
    for rank in range(world):
        assert results[rank][0] == pytest.approx(kl, rel=1e-12)
        assert results[rank][1] == pytest.approx(ell - kl, rel=1e-12)
    # bit-identical on every rank: the plain totals AND the totals that carry a local graph
    assert len({results[r] for r in range(world)}) == 1
