"""Walker sharding across the GPUs of one node (one process per GPU, torch.distributed).

The path shards by independent units: walkers / ensemble members share only the
observation data.  Rank r evaluates the contiguous block ``shard_range(nb, r, world)`` on its own GPU and
the per-walker log-likelihoods are exchanged with ONE all-gather (backend "nccl" = RCCL over xGMI on the GPU
box, "gloo" in the CPU tests): 8 bytes per walker, latency bound, no ring all-reduce.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous block [lo, hi) of n units owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def sharded_loglike(samples, local_eval, group=None, device=None):
    """Evaluate ln L of samples[nb, ndim] with walkers block-sharded over the process group.

    local_eval(samples_block) -> float64[len(block)] runs on this rank's GPU (Fitter.loglike_batch).
    Every rank passes the same `samples` (emcee runs replicated) and gets the full [nb] vector back.
    """
    samples = np.ascontiguousarray(samples, dtype=np.float64)
    nb = samples.shape[0]
    if not (dist.is_available() and dist.is_initialized()):
        return np.asarray(local_eval(samples), dtype=np.float64)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_range(nb, rank, world)
    per = (nb + world - 1) // world  # padded block so all_gather_into_tensor has equal shapes
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    mine = torch.full((per,), float("nan"), dtype=torch.float64, device=device)
    if hi > lo:
        vals = np.asarray(local_eval(samples[lo:hi]), dtype=np.float64)
        mine[: hi - lo] = torch.from_numpy(vals).to(device)
    gathered = torch.empty(per * world, dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(gathered, mine, group=group)
    g = gathered.cpu().numpy().reshape(world, per)
    out = np.empty(nb)
    for r in range(world):
        a, b = shard_range(nb, r, world)
        out[a:b] = g[r, : b - a]
    return out
