#!/usr/bin/env python3
"""bench.py -- light-curves/s of the MI355X afterglow engine on BASELINE.json's configs[1].

One "step" = one pass of the hot path (adaptive grid -> blast-wave ODE -> per-cell synchrotron -> EAT flux
integration) over one batch of synthetic models: `--batch` Gaussian-jet off-axis models of the C2
configuration (GaussianJet(0.1, 1e52, 300) + ISM(1), theta_obs = 0.3,
resolutions (0.355, 0.31, 20.5) -> ~(64, 64, 199) grid, 200 times x 10 bands) with +-10 % jitter on the
physical parameters.  Inputs (parameter structs, t, nu) are resident in HBM before the timed region and the
fluxes stay in HBM.  `python bench.py --gpus N --steps K --warmup W`; for N > 1 launch with torch.distributed.run
(one rank per GPU): models are block-sharded (weak scaling: --batch models per rank), and each step ends with the
all-gather of a per-model summary (8 B/model) that a sampler would consume.

Prints ONE JSON line (see README / DESIGN.md for the roofline conventions).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

F_SPEC = 210.0   # FP64 flop-equivalents per spectrum evaluation
F_INTERP = 26.0  # per log-log interpolation + exp2 + accumulate
PEAK_HBM_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PEAK_FP64_TFLOPS = 78.6  # FP64 vector = half the 157.3 TF FP32 vector peak of MI355X_MICROARCH.md


def c2_batch(nb, seed):
    """C2 models with +-10 % log-uniform jitter (synthetic, seeded)."""
    import _abi
    import configs
    rng = np.random.default_rng(seed)
    arr = (_abi.ModelParams * nb)()
    for i in range(nb):
        kw = dict(configs.C2)
        j = lambda: float(np.exp(rng.uniform(np.log(0.9), np.log(1.1))))
        kw["E_iso"] *= j()
        kw["Gamma0"] *= j()
        kw["n_ism"] *= j()
        kw["eps_e"] *= j()
        kw["eps_B"] *= j()
        kw["p"] = 2.3 + rng.uniform(-0.1, 0.1)
        kw["theta_c"] *= j()
        arr[i] = _abi.make_params(**kw)
    return arr


def cpu_baseline(arr, t, nu, budget_s=12.0):
    """Reference CPU path timed on this box's host, one thread, on a bounded sample of the same workload."""
    import _abi
    lib, kind = _abi.load_ref(), "reference"
    if lib is None:
        lib, kind = _abi.load_oracle(fast=True), "port"
        if lib is None:
            import subprocess
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle_fast.so"])
            lib = _abi.load_oracle(fast=True)
    n, t0 = 0, time.perf_counter()
    while n < len(arr) and (time.perf_counter() - t0 < budget_s or n < 2):
        lib.flux_density_grid(arr[n], t, nu)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "light-curves/s", "cores": 1, "kind": kind,
            "sample": f"{n} models of the timed batch (C2: 200 t x 10 nu), single thread, {dt:.1f} s"}


def cpu_baseline_all_cores(arr, t, nu, budget_s=4.0):
    """Same reference path on many host cores, one model per thread (the reference's own scheme:
    ThreadPoolExecutor over walkers with the GIL released, fitting/samplers.py:59-91; ctypes drops the GIL too).
    The box may expose more logical CPUs than its CPU quota grants, so a few thread counts are tried and the best
    throughput is reported with the thread count that produced it."""
    import _abi
    from concurrent.futures import ThreadPoolExecutor
    lib = _abi.load_ref() or _abi.load_oracle(fast=True)
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    best = None
    for nthreads in sorted({min(8, ncpu), min(32, ncpu), ncpu}):
        deadline = time.perf_counter() + budget_s

        def worker(w):
            n, i = 0, w
            while time.perf_counter() < deadline:
                lib.flux_density_grid(arr[i % len(arr)], t, nu)
                n += 1
                i += nthreads
            return n

        t0 = time.perf_counter()
        with ThreadPoolExecutor(nthreads) as ex:
            total = sum(ex.map(worker, range(nthreads)))
        dt = time.perf_counter() - t0
        r = {"value": total / dt, "unit": "light-curves/s", "cores": nthreads,
             "sample": f"{total} models, {nthreads} threads, {dt:.1f} s (host exposes {ncpu} logical CPUs)"}
        if best is None or r["value"] > best["value"]:
            best = r
    return best


def tophat_sweep(lib, h, _lib, dev):
    """BASELINE configs[0] (the >= 100x target config): top-hat + ISM, resolutions (0.089, 0.05, 12),
    100 times x 3 bands; single-call latency and batched throughput, on-axis (C1a) and theta_obs = 0.05 (C1b)."""
    import torch
    import _abi
    import configs
    out = {}
    t, nu = configs.C1_T, configs.C1_NU
    d_t, d_nu = torch.from_numpy(t).to(dev), torch.from_numpy(nu).to(dev)
    for name, kw, batches in (("C1a_onaxis", configs.C1A, (1, 64, 1024, 4096)), ("C1b_theta_obs_0.05", configs.C1B, (1, 64, 1024))):
        res = {}
        for nb in batches:
            rng = np.random.default_rng(1)
            arr = (_abi.ModelParams * nb)()
            for i in range(nb):
                k = dict(kw)
                for key in ("E_iso", "n_ism", "eps_B"):
                    k[key] *= float(np.exp(rng.uniform(-0.1, 0.1)))
                arr[i] = _abi.make_params(**k)
            d_p = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
            d_o = torch.empty((nb, nu.size, t.size), dtype=torch.float64, device=dev)
            call = lambda: _lib.check(lib.vag_flux_density_grid_batch_dev(h, d_p.data_ptr(), nb, d_t.data_ptr(), t.size,
                                                                            d_nu.data_ptr(), nu.size, d_o.data_ptr()))
            call()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                call()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 5
            res[f"batch_{nb}"] = {"ms_per_call": 1e3 * dt, "light_curves_per_s": nb / dt}
        out[name] = res
    return out


def ensemble_bench(lib, h, _lib, dev):
    """BASELINE configs[2] and [4] as batched ensembles on one GPU: C3 = power-law jet in a wind,
    forward + reverse shock with SSC + Klein-Nishina on both (jittered parameters); C5 = prior-predictive sweep of
    two-component SSC jets.  100 times x 4 bands (incl. 2.4e26 Hz), device-resident inputs."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    from ssc_ensemble import c3_batch, c5_batch
    t, nu = np.logspace(2, 8, 100), np.array([1e9, 4.84e14, 1e18, 2.4e26])
    d_t, d_nu = torch.from_numpy(t).to(dev), torch.from_numpy(nu).to(dev)
    out = {}
    for name, prms, ref_ms in (("C3_fs_rs_ssc_kn", c3_batch(128), 1197.0), ("C5_two_component_ssc", c5_batch(256), 1120.0)):
        nb = len(prms)
        import _abi
        arr = (_abi.ModelParams * nb)(*prms)
        d_p = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        d_o = torch.empty((nb, nu.size, t.size), dtype=torch.float64, device=dev)
        call = lambda: _lib.check(lib.vag_flux_density_grid_batch_dev(h, d_p.data_ptr(), nb, d_t.data_ptr(), t.size,
                                                                        d_nu.data_ptr(), nu.size, d_o.data_ptr()))
        call()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        plan = _lib.Plan()
        lib.vag_last_plan(h, C.byref(plan))
        out[name] = {"batch": nb, "ms_per_batch": 1e3 * dt, "light_curves_per_s": nb / dt, "finite": bool(torch.isfinite(d_o).all()),
                     "ode_rows": plan.n_rows, "cells": plan.n_cells,
                     "reference_cpu_ms_per_model_survey": ref_ms}
    return out


def walker_bench(lib, h, _lib, dev, rank, world, steps=5, nwalkers=1024):
    """Secondary metric of BASELINE.json: MCMC walker-steps/s on the C4 problem: GW170817-like mock,
    60 data points (3 bands x 20 epochs), 8 free parameters, default resolutions, 1024 walkers drawn uniformly
    from the prior box, block-sharded over the ranks with one all-gather of ln L per step."""
    import torch
    import torch.distributed as dist
    import configs
    from vegasafterglow_amd import fitting
    from vegasafterglow_amd.dist import shard_range
    t, nu = configs.c4_mock_data()
    kw = configs.C4_TRUTH
    # mock data from the engine's own truth model (+5 % noise, 10 % errors), built through the C-ABI
    truth = np.empty(t.size)
    p = _lib.ModelParams()
    lib.vag_params_default(C.byref(p))
    p.jet_type = _lib.JET_GAUSSIAN
    p.theta_c, p.E_iso, p.Gamma0, p.n_ism = kw["theta_c"], kw["E_iso"], kw["Gamma0"], kw["n_ism"]
    p.lumi_dist, p.z, p.theta_obs, p.eps_e, p.eps_B, p.p = kw["lumi_dist"], kw["z"], kw["theta_obs"], kw["eps_e"], kw["eps_B"], kw["p"]
    dp = C.POINTER(C.c_double)
    _lib.check(lib.vag_flux_density_batch(h, C.byref(p), 1, t.ctypes.data_as(dp), nu.ctypes.data_as(dp), t.size,
                                          truth.ctypes.data_as(dp)))
    f_obs = truth * (1 + 0.05 * np.random.default_rng(42).standard_normal(t.size))
    fit = fitting.Fitter(z=kw["z"], lumi_dist=kw["lumi_dist"], jet="gaussian", medium="ism")
    for b in configs.C4_BANDS:
        sel = nu == b
        fit.add_flux_density(b, t[sel], f_obs[sel], 0.1 * f_obs[sel])
    defs = [fitting.ParamDef(n, 10.0 ** lo if lg else lo, 10.0 ** hi if lg else hi,
                             fitting.Scale.log if lg else fitting.Scale.linear) for n, lg, lo, hi in configs.C4_FREE]
    spec, lo, hi = fit.build_spec(defs)
    theta = lo + (hi - lo) * np.random.default_rng(0).random((nwalkers, len(defs)))
    a, b = shard_range(nwalkers, rank, world)
    per = (nwalkers + world - 1) // world
    d_theta = torch.from_numpy(np.ascontiguousarray(theta[a:b])).to(dev)
    d_ll = torch.full((per,), float("nan"), dtype=torch.float64, device=dev)
    d_all = torch.empty((per * world,), dtype=torch.float64, device=dev) if world > 1 else None

    def step():
        _lib.check(lib.vag_loglike_batch_dev(h, C.byref(spec), d_theta.data_ptr(), b - a, spec.ndim, d_ll.data_ptr()))
        if world > 1:
            dist.all_gather_into_tensor(d_all, d_ll)

    step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        el = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dt = float(el.item())
    st = _lib.StageTimes()
    lib.vag_last_stage_times(h, C.byref(st))
    finite = int(torch.isfinite(d_ll[: b - a]).sum().item())
    return {"value": nwalkers * steps / dt, "unit": "walker-steps/s", "walkers": nwalkers, "steps": steps,
            "ms_per_step": 1e3 * dt / steps, "scaling": "strong", "finite_on_rank0": finite, "walkers_on_rank0": b - a,
            "rank0_stage_ms": {"grid": st.grid_ms, "dynamics": st.dynamics_ms, "syn_cells": st.cells_ms,
                               "series_flux": st.flux_ms, "reduce": st.reduce_ms}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=512, help="models per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-walkers", action="store_true", help="skip the secondary walker-steps/s measurement")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import configs
    from vegasafterglow_amd import _lib

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    lib = _lib.load()
    h = C.c_void_p()
    _lib.check(lib.vag_ctx_create(local_rank, C.byref(h)))
    stream = torch.cuda.current_stream()
    _lib.check(lib.vag_ctx_set_stream(h, C.c_void_p(stream.cuda_stream)))

    nb = args.batch
    t_np, nu_np = configs.C2_T, configs.C2_NU
    nt, nnu = t_np.size, nu_np.size
    arr = c2_batch(nb, seed=1234 + rank)
    dev = torch.device("cuda", local_rank)
    d_params = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    d_t = torch.from_numpy(t_np).to(dev)
    d_nu = torch.from_numpy(nu_np).to(dev)
    d_out = torch.empty((nb, nnu, nt), dtype=torch.float64, device=dev)
    gathered = torch.empty((world * nb,), dtype=torch.float64, device=dev) if world > 1 else None

    flux_ms = []

    def step(record):
        _lib.check(lib.vag_flux_density_grid_batch_dev(h, d_params.data_ptr(), nb, d_t.data_ptr(), nt, d_nu.data_ptr(), nnu,
                                                       d_out.data_ptr()))
        if world > 1:
            # what a sampler consumes per model (here: the band-summed fluence proxy), 8 B/model over RCCL
            dist.all_gather_into_tensor(gathered, d_out.sum(dim=(1, 2)))
        if record:
            st = _lib.StageTimes()
            _lib.check(lib.vag_last_stage_times(h, C.byref(st)))  # HIP events on the kernel's own stream
            flux_ms.append((st.grid_ms, st.dynamics_ms, st.cells_ms, st.flux_ms, st.reduce_ms, st.total_ms))

    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())

    # one extra UNTIMED pass with the kernel's work tallies on: exact spectrum-evaluation / interpolation counts
    _lib.check(lib.vag_ctx_count_work(h, 1))
    step(False)
    torch.cuda.synchronize()
    _lib.check(lib.vag_ctx_count_work(h, 0))
    plan = _lib.Plan()
    lib.vag_last_plan(h, C.byref(plan))
    if not bool(torch.isfinite(d_out).all()) or plan.n_models_ok != nb:
        raise SystemExit("bench produced non-finite fluxes or rejected models")
    walkers = None if args.no_walkers else walker_bench(lib, h, _lib, dev, rank, world)
    walkers_half = None if args.no_walkers else walker_bench(lib, h, _lib, dev, rank, world, nwalkers=512)
    # an ensemble sized to the node (1024 walkers per GPU): the weak-scaling counterpart of the 1024-walker run above
    walkers_weak = walker_bench(lib, h, _lib, dev, rank, world, nwalkers=1024 * world) if (world > 1 and not args.no_walkers) else None
    tophat = tophat_sweep(lib, h, _lib, dev) if (not args.no_walkers and world == 1) else None
    ensembles = ensemble_bench(lib, h, _lib, dev) if (not args.no_walkers and world == 1) else None

    if rank == 0:
        st = np.mean(np.array(flux_ms), axis=0)
        flux_s = st[3] * 1e-3
        # algorithmic work of ONE flux-kernel launch (DESIGN.md "Roofline accounting")
        alg_bytes = plan.n_cells * 18 * 8 + nb * nnu * nt * 8 + nb * (64 + 64) * 8
        alg_flops = plan.spec_evals * F_SPEC + plan.interps * F_INTERP
        out = {
            "metric": "light-curves/sec (single model) and MCMC walker-steps/sec at 1/2/4/8 MI355X",
            "value": world * nb * args.steps / elapsed,
            "unit": "light-curves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: GaussianJet off-axis (theta_obs=0.3) + ISM, synchrotron+SSA, "
                                   "resolutions (0.355,0.31,20.5) -> ~64x64x199 cells/model, 200 time bins x 10 bands",
                       "models_per_gpu_per_step": nb, "global_batch": world * nb, "parallelism": f"walker-shard x{world}"},
            "roofline": {"bound": "hbm", "kernel": "vag_flux_grid_kernel",
                         "achieved": alg_bytes / flux_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": alg_bytes / flux_s / 1e9 / PEAK_HBM_GBS, "traffic": None,
                         "bytes_per_launch": alg_bytes, "ms_per_launch": st[3]},
            "roofline_fp64": {"bound": "fp64_valu", "kernel": "vag_flux_grid_kernel",
                              "achieved": alg_flops / flux_s / 1e12, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                              "frac": alg_flops / flux_s / 1e12 / PEAK_FP64_TFLOPS,
                              "spec_evals_per_launch": plan.spec_evals, "interps_per_launch": plan.interps},
            "stage_ms": {"grid": st[0], "dynamics": st[1], "syn_cells": st[2], "sync_flux": st[3], "reduce": st[4],
                         "total_device": st[5]},
            "plan": {"ode_rows": plan.n_rows, "cells": plan.n_cells, "theta_phi_rows": plan.total_pairs,
                     "flux_workgroups": plan.flux_blocks, "rows_per_workgroup": plan.pairs_per_block},
        }
        # HBM traffic of the dominant kernel from the committed rocprofv3 PMC passes of this same command
        # (profiles/run_profile.sh; FETCH_SIZE x2 per the gfx950 note, calibrated on a kernel with known bytes)
        tp = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(tp) and nb == 512 and world == 1:
            out["roofline"]["traffic"] = json.load(open(tp))["traffic_bytes_per_launch"]
        if walkers is not None:
            out["walker_steps"] = walkers
            out["walker_steps_redblue_half"] = walkers_half  # emcee red-blue moves evaluate nwalkers/2 per call
            if walkers_weak:
                walkers_weak["scaling"] = "weak"
                out["walker_steps_1024_per_gpu"] = walkers_weak
        if tophat is not None:
            out["tophat_config0"] = tophat
        if ensembles:
            out["ensembles_config2_config4"] = ensembles
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(arr, t_np, nu_np)
            out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(arr, t_np, nu_np)
        print(json.dumps(out))
    lib.vag_ctx_destroy(h)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
