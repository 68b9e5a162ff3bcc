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


def deal_positions(n, world):
    """The fixed part of the deal: position q (0 = most expensive unit) of every (rank, slot), int64[world, per] with -1 for
    padding.  Sweep s hands positions s*world .. s*world + world-1 to the ranks in boustrophedon order (0..w-1, w-1..0, ...),
    so every rank gets the same count and the cost sums differ by at most about one unit's cost per sweep."""
    per = (n + world - 1) // world
    s = np.arange(per, dtype=np.int64)[None, :]
    r = np.arange(world, dtype=np.int64)[:, None]
    q = s * world + np.where(s % 2 == 0, r, world - 1 - r)
    return np.where(q < n, q, -1)


def balanced_assignment(costs, world):
    """Deal n units to `world` ranks, ceil(n / world) slots each (-1 = padding): units in order of decreasing cost (ties keep
    the original order) onto ``deal_positions``.  Deterministic for identical `costs`: every rank computes the same table
    without talking.  Returns int64[world, per].  Host (numpy) statement of the deal; the sharder does the same on the device."""
    costs = np.asarray(costs, dtype=np.float64)
    order = np.argsort(-costs, kind="stable")
    q = deal_positions(costs.size, world)
    return np.where(q >= 0, order[np.maximum(q, 0)] if costs.size else q, -1)


def _default_device(group):
    if dist.is_available() and dist.is_initialized() and dist.get_backend(group) == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


class _Deal:
    """What is fixed for a (batch size, world): slot positions, the flat indices of the slots that hold a walker, buffers."""

    def __init__(self, nb, world, rank, device):
        q = deal_positions(nb, world)
        self.per = q.shape[1]
        self.n_mine = int((q[rank] >= 0).sum())
        flat = q.reshape(-1)
        self.pos = torch.from_numpy(np.maximum(flat, 0)).to(device)          # [world * per] position of every slot (padding: 0)
        self.keep = torch.from_numpy(np.nonzero(flat >= 0)[0]).to(device)    # flat slots that hold a walker
        self.pad = torch.from_numpy(flat < 0).to(device)
        self.block = torch.empty((self.per, 2), dtype=torch.float64, device=device)
        self.gathered = torch.empty((world * self.per, 2), dtype=torch.float64, device=device)  # rank-major concatenation


class WalkerSharder:
    """Sharded evaluation of a per-walker function, device-resident and cost-balanced.

    ``eval_dev(theta_block) -> (values, costs)`` evaluates a block on this rank's device: ``theta_block`` is a float64 tensor
    [k, ndim] on ``device``; ``values`` float64[k] (ln L, or ln L + ln prior); ``costs`` float64[k] relative cost of each
    walker (None: unknown, the blocks then stay balanced by count).  Every rank calls the sharder with the SAME theta
    (samplers run replicated, or rank 0 broadcasts its proposals) and gets the full [nb] vector back, on the device.
    Without an initialised process group it is a plain call.

    Nothing in a call touches the host beyond launching: the walkers are ranked by the previous call's gathered costs ON THE
    DEVICE, the slot positions of a (batch size, world) are cached tensors, and there is no ``.cpu()`` / ``.item()`` anywhere.
    An evaluator made by ``Fitter.device_evaluator`` carries ``eval_dev.native``: the deal, the rank's block and the scatter are
    then three launches inside the engine (``vag_loglike_shard_dev`` / ``vag_loglike_shard_finish_dev``) around the one
    all-gather; any other evaluator gets the same deal from a handful of torch operations.
    """

    def __init__(self, eval_dev, group=None, device=None):
        self.eval_dev, self.group = eval_dev, group
        self.device = device if device is not None else _default_device(group)
        self.native = getattr(eval_dev, "native", None)
        self._deals = {}
        self._costs = None       # float64[nb] tensor on the device: gathered costs of the previous call (generic path)
        self._table = None       # int64[world * per] tensor: walker of every (rank, slot) in the last call, -1 = padding
        self._shape = None       # (nb, world, per) of the last call

    @property
    def costs(self):
        """Gathered per-walker costs the NEXT deal ranks by (numpy, diagnostic: synchronises)."""
        if self.native is not None:
            return None if self._shape is None else self.native.state(self._shape[0], self._shape[1], self._shape[2])[1]
        return None if self._costs is None else self._costs.cpu().numpy()

    @property
    def last_table(self):
        """int64[world, per] walker of every (rank, slot) in the last call, -1 = padding (numpy, diagnostic: synchronises)."""
        if self._shape is None:
            return None
        nb, world, per = self._shape
        if self.native is not None:
            return self.native.state(nb, world, per)[0]
        return self._table.cpu().numpy().reshape(world, per)

    def costs_per_rank(self):
        """Sum of the reported walker costs on every rank in the last call (how well the deal balanced the work)."""
        table, costs = self.last_table, self.costs
        if table is None or costs is None:
            return None
        return np.array([costs[row[row >= 0]].sum() for row in table])

    def _deal(self, nb, world, rank):
        key = (nb, world, rank)
        d = self._deals.get(key)
        if d is None:
            d = self._deals[key] = _Deal(nb, world, rank, self.device)
        return d

    def __call__(self, theta):
        theta = torch.as_tensor(theta, dtype=torch.float64, device=self.device)
        if theta.dim() != 2:
            raise ValueError("theta must be [nb, ndim]")
        nb = theta.shape[0]
        if not (dist.is_available() and dist.is_initialized()):  # one process: nothing to deal, so no cost report is asked for
            values, _ = self.eval_dev(theta, want_costs=False) if getattr(self.eval_dev, "optional_costs", False) else self.eval_dev(theta)
            return values
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        d = self._deal(nb, world, rank)
        out = torch.empty((nb,), dtype=torch.float64, device=self.device)
        self._shape = (nb, world, d.per)
        if self.native is not None:
            # The context's lock is held inside shard() and inside finish(), NOT across the collective: a process-local lock held
            # while waiting for other ranks can deadlock two sharders that take it in different orders on different ranks.  The
            # engine keeps the deal of every call in flight under a ticket (ABI v13), so another sharder of this process may deal --
            # and finish, in any order -- on the same context in between.
            with self.native.lock:
                ticket = self.native.shard(theta.contiguous(), nb, rank, world, d.block)
            dist.all_gather_into_tensor(d.gathered, d.block, group=self.group)  # the path's only collective: 16 B per walker
            with self.native.lock:
                self.native.finish(ticket, d.gathered, nb, world, out)
            return out
        if self._costs is None or self._costs.shape[0] != nb:
            self._costs = torch.ones((nb,), dtype=torch.float64, device=self.device)
        order = torch.argsort(self._costs, descending=True, stable=True)  # position -> walker
        table = order.index_select(0, d.pos).masked_fill(d.pad, -1)       # walker of every (rank, slot)
        mine = table[rank * d.per: rank * d.per + d.n_mine]
        # [ln L | cost] per slot; padding slots carry NaN / 0 and are never read back
        d.block[:, 0] = float("nan")
        d.block[:, 1] = 0.0
        if d.n_mine:
            values, costs = self.eval_dev(theta.index_select(0, mine))
            d.block[: d.n_mine, 0] = values
            d.block[: d.n_mine, 1] = costs if costs is not None else 1.0
        dist.all_gather_into_tensor(d.gathered, d.block, group=self.group)
        walkers = table.index_select(0, d.keep)
        got = d.gathered.index_select(0, d.keep)
        out.index_copy_(0, walkers, got[:, 0])
        nc = torch.ones((nb,), dtype=torch.float64, device=self.device).index_copy_(0, walkers, got[:, 1])
        pos = nc > 0  # walkers that were not evaluated (invalid parameters: cost 0) are assumed average next time
        npos = pos.sum()
        mean = torch.where(npos > 0, (nc * pos).sum() / npos.clamp(min=1), torch.ones((), dtype=torch.float64, device=self.device))
        self._costs = torch.where(pos, nc, mean)
        self._table = table
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
