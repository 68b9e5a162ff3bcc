"""Walker / ensemble sharding across the GPUs of one node (one process per GPU, torch.distributed).

The path shards by independent units: walkers / ensemble members share only the observation data
(fitter.py:503-533 builds a fresh Model per walker).  There is no data-path collective: every rank evaluates its units on
its own GPU and the per-walker results are exchanged with ONE all-gather per call (backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests): 16 bytes per walker -- ln L and the walker's cost --, latency bound, no ring all-reduce.

Walker cost varies ~8x over a prior box (every walker builds its own adaptive grid), so blocks of equal COUNT are not blocks
of equal WORK: ``WalkerSharder`` deals the walkers to the ranks by the cost the engine reported for the same batch position in
the previous call (theta x phi x t cells, ``vag_last_model_costs_dev``) -- equal counts per rank (one fixed-shape all-gather),
near-equal cost sums.  Everything stays on the device: theta comes in as a tensor on the rank's GPU, ln L goes out as one.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous block [lo, hi) of n units owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def balanced_assignment(costs, world):
    """Deal n units to `world` ranks, ceil(n / world) slots each (-1 = padding): units in order of decreasing cost, ranks in
    boustrophedon order (0..w-1, w-1..0, ...), so every rank gets the same count and the cost sums differ by at most about one
    unit's cost per sweep.  Deterministic for identical `costs` (ties keep the original order): every rank computes the same
    table without talking.  Returns int64[world, per]."""
    costs = np.asarray(costs, dtype=np.float64)
    n = costs.size
    per = (n + world - 1) // world
    table = np.full((world, per), -1, dtype=np.int64)
    order = np.argsort(-costs, kind="stable")
    for pos, unit in enumerate(order):
        sweep, k = divmod(pos, world)
        table[k if sweep % 2 == 0 else world - 1 - k, sweep] = unit
    return table


def _default_device(group):
    if dist.is_available() and dist.is_initialized() and dist.get_backend(group) == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


class WalkerSharder:
    """Sharded evaluation of a per-walker function, device-resident and cost-balanced.

    ``eval_dev(theta_block) -> (values, costs)`` evaluates a block on this rank's device: ``theta_block`` is a float64 tensor
    [k, ndim] on ``device``; ``values`` float64[k] (ln L, or ln L + ln prior); ``costs`` float64[k] relative cost of each
    walker (None: unknown, the blocks then stay balanced by count).  Every rank calls the sharder with the SAME theta
    (samplers run replicated, or rank 0 broadcasts its proposals) and gets the full [nb] vector back, on the device.
    Without an initialised process group it is a plain call.
    """

    def __init__(self, eval_dev, group=None, device=None):
        self.eval_dev, self.group = eval_dev, group
        self.device = device if device is not None else _default_device(group)
        self.costs = None  # float64[nb] of the previous call with the same batch size (host copy: it only orders indices)
        self._pending = None  # the last call's gathered costs, still on the device: read when the next call needs them
        self.last_table = None

    def _take_pending_costs(self):
        if self._pending is None:
            return
        nc = self._pending.cpu().numpy()  # finished long ago: the caller has consumed that call's ln L
        self._pending = None
        pos = nc > 0  # walkers that were not evaluated (invalid parameters: cost 0) are assumed average next time
        self.costs = np.where(pos, nc, nc[pos].mean() if pos.any() else 1.0)

    def costs_per_rank(self):
        """Sum of the reported walker costs on every rank in the last call (how well the deal balanced the work)."""
        self._take_pending_costs()
        if self.last_table is None or self.costs is None:
            return None
        return np.array([self.costs[row[row >= 0]].sum() for row in self.last_table])

    def __call__(self, theta):
        theta = torch.as_tensor(theta, dtype=torch.float64, device=self.device)
        if theta.dim() != 2:
            raise ValueError("theta must be [nb, ndim]")
        nb = theta.shape[0]
        if not (dist.is_available() and dist.is_initialized()):
            values, _ = self.eval_dev(theta)
            return values
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        self._take_pending_costs()
        if self.costs is None or self.costs.size != nb:
            self.costs = np.ones(nb)
        table = balanced_assignment(self.costs, world)
        per = table.shape[1]
        mine = table[rank][table[rank] >= 0]
        # [ln L | cost] per slot; padding slots carry NaN / 0 and are never read back
        block = torch.full((per, 2), float("nan"), dtype=torch.float64, device=self.device)
        block[:, 1] = 0.0
        if mine.size:
            idx = torch.as_tensor(mine, device=self.device)
            values, costs = self.eval_dev(theta.index_select(0, idx))
            block[: mine.size, 0] = values
            block[: mine.size, 1] = costs if costs is not None else 1.0
        gathered = torch.empty((world * per, 2), dtype=torch.float64, device=self.device)  # rank-major concatenation
        dist.all_gather_into_tensor(gathered, block, group=self.group)  # the path's only collective: 16 B per walker
        flat_idx = torch.as_tensor(table.reshape(-1), device=self.device)
        keep = flat_idx >= 0
        out = torch.empty((nb,), dtype=torch.float64, device=self.device)
        out[flat_idx[keep]] = gathered[..., 0].reshape(-1)[keep]
        new_costs = torch.ones((nb,), dtype=torch.float64, device=self.device)
        new_costs[flat_idx[keep]] = gathered[..., 1].reshape(-1)[keep]
        self._pending = new_costs
        self.last_table = table
        return out


def sharded_loglike(samples, local_eval, group=None, device=None):
    """Host-array form: ln L of samples[nb, ndim] with walkers block-sharded by count over the process group.

    local_eval(samples_block) -> float64[len(block)] runs on this rank's GPU (Fitter.loglike_batch).  Every rank passes the
    same `samples` and gets the full [nb] vector back.  Kept for callers that hold numpy arrays; the device-resident,
    cost-balanced path is ``WalkerSharder``.
    """
    samples = np.ascontiguousarray(samples, dtype=np.float64)
    nb = samples.shape[0]
    if not (dist.is_available() and dist.is_initialized()):
        return np.asarray(local_eval(samples), dtype=np.float64)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_range(nb, rank, world)
    per = (nb + world - 1) // world  # padded block so all_gather_into_tensor has equal shapes
    if device is None:
        device = _default_device(group)
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


def sharded_flux_density_grid(params, eval_dev, n_out, group=None, device=None, gather=True):
    """Prior-predictive ensembles (BASELINE configs[4]): `params` is a sequence of nb model parameter structs, the same on every
    rank; rank r evaluates the contiguous block shard_range(nb, r, world) with ``eval_dev(block) -> float64 tensor [k, *n_out]``
    on its GPU.  gather=True returns the full [nb, *n_out] tensor on every rank (one all-gather, ~3 KB per model for 4 x 100
    fluxes); gather=False returns (lo, hi, block): each rank keeps (or writes out) its own members, no collective at all."""
    nb = len(params)
    if not (dist.is_available() and dist.is_initialized()):
        block = eval_dev(params)
        return block if gather else (0, nb, block)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_range(nb, rank, world)
    if device is None:
        device = _default_device(group)
    per = (nb + world - 1) // world
    mine = torch.full((per, *n_out), float("nan"), dtype=torch.float64, device=device)
    if hi > lo:
        mine[: hi - lo] = eval_dev(params[lo:hi])
    if not gather:
        return lo, hi, mine[: hi - lo]
    gathered = torch.empty((world * per, *n_out), dtype=torch.float64, device=device)  # rank-major concatenation
    dist.all_gather_into_tensor(gathered, mine, group=group)
    out = torch.empty((nb, *n_out), dtype=torch.float64, device=device)
    for r in range(world):
        a, b = shard_range(nb, r, world)
        out[a:b] = gathered[r * per: r * per + (b - a)]
    return out
