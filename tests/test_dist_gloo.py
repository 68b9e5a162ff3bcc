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
