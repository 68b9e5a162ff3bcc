"""world_size-2 gloo test of the walker sharding + all-gather used for N > 1 GPUs (CPU only).

The per-rank evaluator here is the CPU oracle's log-likelihood (the checker), standing in for the rank's GPU:
what is under test is the partition, the padding and the single all-gather of vegasafterglow_amd.dist.
"""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import _abi
import configs
from vegasafterglow_amd import dist as vdist
from vegasafterglow_amd import _lib


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _c4_spec_and_data():
    """C4 mock: truth fluxes from the oracle with 5 % multiplicative noise, 10 % errors."""
    orc = _abi.load_oracle()
    t, nu = configs.c4_mock_data()
    truth = orc.flux_density(_abi.make_params(**configs.C4_TRUTH), t, nu)
    rng = np.random.default_rng(42)
    f_obs = truth * (1 + 0.05 * rng.standard_normal(t.size))
    err = 0.1 * f_obs
    return t, nu, np.log(f_obs), err / f_obs, np.ones_like(t)


def _oracle_loglike(samples, data):
    """Fitter._evaluate through the oracle (vag_oracle_loglike_batch)."""
    t, nu, lnf, lne, w = data
    orc = _abi.load_oracle()
    fn = orc.lib.vag_oracle_loglike_batch
    fn.argtypes = [C.POINTER(_lib.FitSpec), C.POINTER(C.c_double), C.c_int, C.c_int, C.POINTER(C.c_double)]
    spec = _lib.FitSpec()
    base = _abi.make_params(**dict(configs.C4_TRUTH))
    spec.base = _lib.ModelParams.from_buffer_copy(bytes(base))
    free = configs.C4_FREE
    spec.ndim = len(free)
    for d, (name, is_log, _, _) in enumerate(free):
        spec.slot[d] = _lib.PARAM_SLOTS[name]
        spec.is_log[d] = is_log
    spec.n_data = t.size
    dp = C.POINTER(C.c_double)
    arrs = [np.ascontiguousarray(a) for a in (t, nu, lnf, lne, w)]
    spec.t, spec.nu, spec.ln_flux, spec.ln_err, spec.weight = [a.ctypes.data_as(dp) for a in arrs]
    samples = np.ascontiguousarray(samples)
    out = np.empty(len(samples))
    assert fn(C.byref(spec), samples.ctypes.data_as(dp), len(samples), spec.ndim, out.ctypes.data_as(dp)) == 0
    return out


def _worker(rank, world, port, samples, data, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []

    def local_eval(block):
        calls.append(len(block))
        return _oracle_loglike(block, data)

    full = vdist.sharded_loglike(samples, local_eval)
    q.put((rank, full, calls))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 1024, 1027):
        for world in (1, 2, 3, 8):
            blocks = [vdist.shard_range(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


def test_sharded_loglike_world2_gloo_matches_single_process():
    data = _c4_spec_and_data()
    rng = np.random.default_rng(0)
    lo = np.array([f[2] for f in configs.C4_FREE])
    hi = np.array([f[3] for f in configs.C4_FREE])
    samples = lo + (hi - lo) * rng.random((7, len(lo)))  # odd count: ragged shards 4 + 3
    samples[3, 2] = -1.0  # invalid theta_c -> the walker must come back as -inf, not crash its rank
    want = _oracle_loglike(samples, data)
    assert want[3] == -np.inf and np.isfinite(np.delete(want, 3)).all()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, samples, data, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, full, calls in results:
        assert calls == [4] if rank == 0 else calls == [3]
        np.testing.assert_array_equal(full, want)  # every rank holds the full, identical vector


# ---- the device-resident, cost-balanced sharder (the path bench.py and a sharded sampler use) ----

def test_balanced_assignment_equal_counts_near_equal_costs():
    rng = np.random.default_rng(1)
    for n, world in ((1024, 8), (1027, 8), (7, 2), (3, 8), (512, 4)):
        costs = np.exp(rng.uniform(np.log(1.0), np.log(8.0), n))  # the 8x spread of walker cost over a prior box
        table = vdist.balanced_assignment(costs, world)
        assert table.shape == (world, -(-n // world))
        used = table[table >= 0]
        assert np.array_equal(np.sort(used), np.arange(n))  # every unit exactly once
        counts = (table >= 0).sum(axis=1)
        assert counts.max() - counts.min() <= 1
        if n >= 16 * world:
            sums = np.array([costs[row[row >= 0]].sum() for row in table])
            by_count = np.array([costs[a:b].sum() for a, b in (vdist.shard_range(n, r, world) for r in range(world))])
            assert sums.max() / sums.mean() < 1.01  # dealt by cost: within 1 % of perfect
            assert sums.max() / sums.mean() <= by_count.max() / by_count.mean()


def test_deal_is_the_boustrophedon_deal_of_the_cost_ranking():
    """balanced_assignment / deal_positions (vectorised; what the device kernels and the torch path compute) against the plain
    statement of the deal: units in order of decreasing cost, ranks 0..w-1, w-1..0, ... sweep by sweep."""
    rng = np.random.default_rng(3)
    for n, world in ((1, 1), (5, 1), (9, 2), (203, 8), (1024, 8), (3, 8), (64, 3)):
        costs = np.round(np.exp(rng.uniform(0, 2, n)), 1)  # rounded: ties must keep their original order
        per = -(-n // world)
        want = np.full((world, per), -1, dtype=np.int64)
        for pos, unit in enumerate(np.argsort(-costs, kind="stable")):
            sweep, k = divmod(pos, world)
            want[k if sweep % 2 == 0 else world - 1 - k, sweep] = unit
        np.testing.assert_array_equal(vdist.balanced_assignment(costs, world), want)
        q = vdist.deal_positions(n, world)
        assert np.array_equal(np.sort(q[q >= 0]), np.arange(n))


def test_sharder_makes_no_host_round_trip_per_call(monkeypatch):
    """Nothing in WalkerSharder.__call__ may read a tensor back (the r02 sharder spent 0.7 ms per call in a Python deal and a
    .cpu() of the costs): with a 1-rank gloo group and .cpu / .item / .numpy / .tolist disabled, two calls must still run."""
    import torch
    port = _free_port()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        theta = torch.from_numpy(np.random.default_rng(0).random((37, 3)))
        sharder = vdist.WalkerSharder(lambda th: (th.sum(1), 1.0 + th[:, 0]))
        first = sharder(theta)  # builds the cached slot positions (host work, once per (nb, world))

        def forbidden(*a, **k):
            raise AssertionError("host read inside WalkerSharder.__call__")
        for name in ("cpu", "item", "numpy", "tolist"):
            monkeypatch.setattr(torch.Tensor, name, forbidden)
        second = sharder(theta)
        third = sharder(theta)
        monkeypatch.undo()
        assert torch.equal(first, theta.sum(1)) and torch.equal(second, first) and torch.equal(third, first)
        np.testing.assert_array_equal(sharder.last_table[0], np.argsort(-(1.0 + theta[:, 0].numpy()), kind="stable"))
    finally:
        dist.destroy_process_group()


def _sharder_worker(rank, world, port, samples, data, q):
    import torch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []

    def eval_dev(theta):  # the oracle stands in for this rank's GPU; cost = a strong function of the viewing angle
        block = theta.numpy()
        calls.append(len(block))
        return torch.from_numpy(_oracle_loglike(block, data)), torch.from_numpy(1.0 + 50.0 * block[:, 3] ** 2)

    sharder = vdist.WalkerSharder(eval_dev)
    first = sharder(torch.from_numpy(samples)).numpy().copy()
    second = sharder(torch.from_numpy(samples)).numpy().copy()  # now dealt by the costs the first call reported
    cpr = sharder.costs_per_rank()
    grid = vdist.sharded_flux_density_grid(list(range(5)), lambda blk: torch.tensor([[float(i), 2.0 * i] for i in blk],
                                                                                 dtype=torch.float64), (2,))
    lo, hi, mine = vdist.sharded_flux_density_grid(list(range(5)), lambda blk: torch.tensor([[float(i), 2.0 * i] for i in blk],
                                                                                             dtype=torch.float64), (2,), gather=False)
    q.put((rank, first, second, calls, cpr, sharder.last_table.copy(), grid.numpy(), (lo, hi, mine.numpy())))
    dist.barrier()
    dist.destroy_process_group()


def test_walker_sharder_world2_gloo_device_path_and_cost_balance():
    data = _c4_spec_and_data()
    rng = np.random.default_rng(0)
    lo = np.array([f[2] for f in configs.C4_FREE])
    hi = np.array([f[3] for f in configs.C4_FREE])
    samples = lo + (hi - lo) * rng.random((9, len(lo)))  # odd count: ragged shards 5 + 4
    samples[3, 2] = -1.0  # invalid theta_c -> -inf, not a crashed rank
    want = _oracle_loglike(samples, data)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharder_worker, args=(r, 2, port, samples, data, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    tables = []
    for rank, first, second, calls, cpr, table, grid, (glo, ghi, gmine) in results:
        np.testing.assert_array_equal(first, want)   # full, identical vector on every rank ...
        np.testing.assert_array_equal(second, want)  # ... whatever the deal
        assert sorted(calls) == sorted([5, 5] if rank == 0 else [4, 4])
        costs = 1.0 + 50.0 * samples[:, 3] ** 2
        by_count = np.array([costs[:5].sum(), costs[5:].sum()])
        assert cpr.max() / cpr.mean() <= by_count.max() / by_count.mean() + 1e-12
        tables.append(table)
        np.testing.assert_array_equal(grid, np.array([[i, 2.0 * i] for i in range(5)]))
        assert (glo, ghi) == ((0, 3) if rank == 0 else (3, 5))
        np.testing.assert_array_equal(gmine, np.array([[i, 2.0 * i] for i in range(glo, ghi)]))
    np.testing.assert_array_equal(tables[0], tables[1])  # both ranks computed the same deal without talking


def _bench_logic_worker(rank, world, port, q):
    """bench.py's N > 1 branch on CPU: the timing contract (warm-up, barrier, EXACT step count, max over ranks) and the
    per-step collective, with a sleep standing in for the GPU step."""
    import time
    import torch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, _abi.ROOT)
    import bench
    dev = torch.device("cpu")
    counted = {"steps": 0, "recorded": 0}
    gathered = torch.empty((world * 4,), dtype=torch.float64)

    def step(record):
        time.sleep(0.01 * (1 + rank))  # rank 1 is the slow one
        dist.all_gather_into_tensor(gathered, torch.full((4,), float(rank), dtype=torch.float64))
        counted["steps"] += 1
        counted["recorded"] += 1 if record else 0

    sync, barrier, max_over_ranks = bench._dist_helpers(world, dev)
    elapsed = bench.timed_steps(step, 5, 2, world, sync, barrier, max_over_ranks)
    line = None
    if rank == 0:  # what rank 0 of `torchrun --nproc-per-node N bench.py --gpus N` writes: the same compact line as at N = 1
        import types
        plan = types.SimpleNamespace(n_cells=1000, spec_evals=10 ** 6, interps=10 ** 6, n_rows=10, total_pairs=100, flux_blocks=4, pairs_per_block=25)
        rec = bench.headline_record(world, dist.get_world_size(), 512, 5, 2, elapsed, np.array([0.1, 0.2, 0.1, 9.0, 0.05, 9.5]), plan, 10, 200)
        r, w = os.pipe()
        os.environ["VAG_BENCH_DETAIL"] = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"vag_bench_detail_test_{os.getpid()}.json")
        bench.emit(rec, w, world)
        os.close(w)
        line = os.read(r, 1 << 16).decode()
        os.close(r)
        os.remove(os.environ["VAG_BENCH_DETAIL"])
    q.put((rank, elapsed, dict(counted), gathered.numpy().copy(), line))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_multi_rank_timing_contract_under_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_logic_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, e0, c0, g0, line), (r1, e1, c1, g1, none) = results
    assert none is None and line.endswith("\n") and line.count("\n") == 1 and len(line) < 4096  # rank 0 only, ONE short line
    import json
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["extra"]["rccl_world"] == 2 and rec["scaling"] == "weak"
    assert rec["value"] == pytest.approx(2 * 512 * 5 / e0, rel=1e-5)  # whole-job aggregate over the MAX-over-ranks time
    assert rec["config"]["global_batch"] == 1024 and rec["roofline"]["ms_per_launch"] == 9.0 and rec["cpu_baseline"] is None
    assert c0 == c1 == {"steps": 7, "recorded": 5}  # 2 warm-up + exactly 5 timed
    assert e0 == e1 and e0 >= 5 * 0.02  # the MAX over ranks (the slow rank's 5 x 20 ms), identical on both
    np.testing.assert_array_equal(g0, [0, 0, 0, 0, 1, 1, 1, 1])


def _ensemble_leg_worker(rank, world, port, q):
    """bench.py's configs[4] leg (sharded_ensemble_leg over dist.sharded_flux_density_grid) on CPU: a list of member ids stands
    in for the resident parameter structs, a sleep + closed form for the rank's GPU."""
    import time
    import torch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, _abi.ROOT)
    import bench
    dev = torch.device("cpu")
    members = list(range(11))  # ragged over two ranks: 6 + 5
    calls = []

    def eval_dev(block):
        calls.append(list(block))
        time.sleep(0.005 * (1 + rank))
        return torch.tensor([[[float(m), 2.0 * m, 3.0 * m]] * 2 for m in block], dtype=torch.float64).reshape(len(block), 2, 3)

    sync, barrier, max_over_ranks = bench._dist_helpers(world, dev)
    out = {}
    for gather in (True, False):
        calls.clear()
        res, got = bench.sharded_ensemble_leg(members, eval_dev, (2, 3), world, dev, 4, 1, gather, sync, barrier, max_over_ranks)
        out[gather] = (res, got if gather else (got[0], got[1], got[2].numpy().copy()), [list(c) for c in calls])
    q.put((rank, out[True][0], out[True][1].numpy().copy(), out[True][2], out[False][0], out[False][1], out[False][2]))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_sharded_ensemble_leg_contract_under_gloo():
    """The leg the N > 1 bench prints for BASELINE configs[4]: every rank evaluates ONLY its contiguous block, exactly
    warm-up + steps times; gather=True hands every rank the full ensemble in member order, gather=False its own block and no
    collective; the rate counts ALL members over the MAX-over-ranks time."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ensemble_leg_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=300) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.array([[[m, 2.0 * m, 3.0 * m]] * 2 for m in range(11)], dtype=np.float64)
    blocks = {0: list(range(0, 6)), 1: list(range(6, 11))}
    for rank, res_g, full, calls_g, res_n, (lo, hi, mine), calls_n in results:
        assert res_g["members"] == res_n["members"] == 11 and res_g["members_per_rank"] == 6 and res_g["steps"] == 4
        assert res_g["gather"] is True and res_n["gather"] is False and res_g["scaling"] == "strong"
        np.testing.assert_array_equal(full, want)                       # every rank holds the whole ensemble, in member order
        assert (lo, hi) == (blocks[rank][0], blocks[rank][-1] + 1)
        np.testing.assert_array_equal(mine, want[lo:hi])                # gather=False: the rank's own block
        assert calls_g == [blocks[rank]] * 5 and calls_n == [blocks[rank]] * 5  # 1 warm-up + exactly 4 timed, own block only
        assert res_g["value"] == pytest.approx(11 * 4 / (res_g["ms_per_step"] * 4e-3))
        assert res_g["ms_per_step"] >= 10.0 - 1e-9                      # the slow rank's 10 ms per step: MAX over ranks
    assert results[0][1]["ms_per_step"] == results[1][1]["ms_per_step"]  # identical on both ranks
