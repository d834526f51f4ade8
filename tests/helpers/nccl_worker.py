"""
Rank process of tests/test_gpu_distributed.py (started by torch.distributed.run, one per GPU, backend "nccl" = RCCL):
every rank builds the SAME seeded batch on the host, keeps only its contiguous shard on its GPU, evaluates it with the HIP
kernels and joins the single scalar all-reduce of markovflow_amd.distributed.  Rank 0 writes what it saw as JSON.
"""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import markovflow_amd as mfa  # noqa: E402
from markovflow_amd import distributed as mfd  # noqa: E402


def inputs(bsz, t, d, m, seed):
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


def main():
    out_path, bsz, t, d, m, seed = sys.argv[1], *map(int, sys.argv[2:7])
    local_rank = int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist.init_process_group(backend="nccl", device_id=dev)
    rank, world = dist.get_rank(), dist.get_world_size()
    full = inputs(bsz, t, d, m, seed)
    lo, hi = mfd.shard_bounds(bsz, rank, world)
    loc = {k: torch.tensor(v[lo:hi], dtype=torch.float64, device=dev) for k, v in full.items()}
    ssm = mfa.StateSpaceModel(loc["mu0"], loc["chol_p0"], loc["a_s"], loc["b_s"], loc["chol_q"])
    kf = mfa.KalmanFilter(ssm, mfa.EmissionModel(loc["h"]), loc["y"], torch.tensor([[0.5]] if m == 1 else 0.5 * np.eye(m),
                                                                                    dtype=torch.float64, device=dev))
    total = mfd.sharded_log_likelihood(kf)
    # the variational exchange step of BASELINE config 4: sum over all series of KL(posterior || prior), one scalar all-reduce
    kl_total = mfd.sharded_kl_divergence(kf.posterior_state_space_model(), ssm) if hi > lo else mfd.all_reduce_sum(
        torch.zeros((), dtype=torch.float64, device=dev))
    ones = torch.ones(1, dtype=torch.float64, device=dev)
    dist.all_reduce(ones)
    every = [torch.zeros((), dtype=torch.float64, device=dev) for _ in range(world)]
    dist.all_gather(every, total)
    if rank == 0:
        with open(out_path, "w") as fh:
            json.dump({"world": world, "ranks_seen": int(ones.item()), "total": float(total), "kl_total": float(kl_total),
                       "per_rank_totals": [float(x) for x in every], "backend": dist.get_backend()}, fh)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
