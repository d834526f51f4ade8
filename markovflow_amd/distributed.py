"""
Batch sharding of independent series over the GPUs of one node (one process per GPU).

The reference is single-process: its only "parallel" axis is the leading batch shape, and
``log_likelihood`` / ``kl_divergence`` end in a sum over it (``markovflow/kalman_filter.py:255``).  That sum is
the single exchange step of the sharded path: every rank evaluates its own contiguous slice of series with
the HIP kernels and ONE all-reduce of a scalar (RCCL over xGMI with backend ``"nccl"``; ``gloo`` on CPU in the
tests) yields the total.  No tensor of the path is ever replicated or moved between ranks.
"""
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(num_series: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous slice ``[lo, hi)`` of the batch axis owned by ``rank``; sizes differ by at most one."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError(f"invalid rank {rank} for world size {world_size}")
    if num_series < 0:
        raise ValueError("num_series must be non-negative")
    base, extra = divmod(num_series, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(tensor: torch.Tensor, rank: int, world_size: int) -> torch.Tensor:
    """This rank's slice of a tensor whose leading axis is the series axis (a view, nothing is copied)."""
    lo, hi = shard_bounds(tensor.shape[0], rank, world_size)
    return tensor[lo:hi]


def all_reduce_sum(value: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """Sum a (scalar) tensor over the ranks in place; a no-op when torch.distributed is not initialised."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(value, op=dist.ReduceOp.SUM, group=group)
    return value


def sharded_log_likelihood(kalman_filter, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """
    Total log marginal likelihood of a batch that is sharded over the ranks: ``kalman_filter`` holds THIS
    rank's series only; the result is identical on every rank.  An empty local shard contributes zero.
    """
    local = kalman_filter.log_likelihood().reshape(())     # a fresh tensor: reduced in place (no copy kernel)
    return all_reduce_sum(local.clone() if local.requires_grad else local, group)


def sharded_kl_divergence(dist_q, dist_p, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """
    ``sum over ALL series of KL(q_s || p_s)`` when both chains hold THIS rank's series only (the KL term of a variational
    model is a sum over the batch: ``models/sparse_variational.py:178-182`` of the reference).  One scalar all-reduce.
    """
    local = torch.sum(dist_q.kl_divergence(dist_p)).reshape(())
    return all_reduce_sum(local.clone() if local.requires_grad else local, group)


def sharded_elbo(expected_log_likelihood: torch.Tensor, kl_divergence: torch.Tensor,
                 group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """
    Evidence lower bound of a batch sharded over the ranks (BASELINE config 4): every rank passes the variational expectations
    and the KL terms of ITS series (any shape; summed here), the result - identical on every rank - is
    ``sum_all E_q[log p(y|f)] - sum_all KL`` (``models/sparse_variational.py:192``).  The two partial sums travel as ONE
    all-reduce of a single scalar.  Gradients: the returned value carries the graph of THIS rank's terms (backward() gives the
    gradients of the local series' parameters; shared hyper-parameters need the usual gradient all-reduce on top).
    """
    local = (torch.sum(expected_log_likelihood) - torch.sum(kl_divergence)).reshape(())
    if not local.requires_grad:
        return all_reduce_sum(local.clone(), group)
    # keep the LOCAL graph: value = the all-reduced total, gradient = that of this rank's terms (each rank owns its series'
    # parameters; shared hyper-parameters need the usual gradient all-reduce on top)
    # `total + 0`: the added term is exactly zero, so the VALUE is the all-reduced total bit for bit on every rank (ranks that
    # branch on it - early stopping, a line search - stay together; `local + (total - local)` rounds differently per rank)
    total = all_reduce_sum(local.detach().clone(), group)
    return total + (local - local.detach())
