#!/usr/bin/env python3
"""bench.py -- light-curves/s and MCMC walker-steps/s of the MI355X afterglow engine on BASELINE.json's configs.

Headline (`value`): one "step" = one pass of the hot path (adaptive grid -> blast-wave ODE -> per-cell synchrotron -> EAT flux
integration) over one batch of synthetic models: `--batch` Gaussian-jet off-axis models of BASELINE configs[1] (SURVEY.md 8d C2:
GaussianJet(0.1, 1e52, 300) + ISM(1), theta_obs = 0.3, resolutions (0.355, 0.31, 20.5) -> ~(64, 64, 199) grid, 200 times x 10
bands), every physical parameter jittered log-uniformly by +-10 % so that the batch is ragged like a real ensemble.  Inputs
(parameter structs, t, nu) are resident in HBM before the timed region and the fluxes stay in HBM.

`python bench.py --gpus N --steps K --warmup W`; for N > 1 launch with torch.distributed.run (one rank per GPU): models are
block-sharded (weak scaling: --batch models per rank) and each step ends with the all-gather of a per-model summary (8 B/model).

The same JSON line carries what BASELINE.json's north_star asks for, each next to the reference CPU path timed on this box:
  walker_steps*     MCMC walker-steps/s on configs[3] through the product's sharded evaluator (dist.WalkerSharder over
                    Fitter.device_evaluator): 1024 and 512 (red-blue half) walkers over all ranks, ln L read by the host after
                    every call as a sampler does; and, on one GPU, the per-rank shares of an 8-GPU run (128 / 64 walkers) with the
                    strong-scaling ceiling they imply;
  tophat_config0    configs[0] (the >= 100x target): single-call latency and batched throughput, on-axis and theta_obs = 0.05,
                    with the reference on 1 core and on all host cores;
  ensembles         configs[2] (FS + RS + SSC + KN) and configs[4] (two-component SSC ensemble, 1024 members) with tallied
                    FP64 rooflines of their flux passes;
  ensemble_c5_4096  configs[4] at its full 4096 members through dist.sharded_flux_density_grid, with and without the all-gather
                    of the fluxes, at EVERY N (N = 1: also the 512-member share of one rank of eight and the ratio it implies);
  walker_steps_8192_total / walker_steps_1024_per_gpu   the strong- and weak-scaling walker legs, at every N as well, so that the
                    driver's N = 1, 2, 4, 8 lines give each curve its origin;
  single_model_latency   one model per call (the reference's own protocol) for configs[1] / [2] / [4] beside the reference's
                    one-core time.
Prints ONE JSON line of < 4 KB on stdout: the driver's contract keys + a flat `extra` of headline scalars (`compact_line`); every leg's
full record goes to stderr and to the side file the line names (`bench_detail.json`, `bench_detail_nN.json` under torchrun).  See
DESIGN.md for the roofline conventions.
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
F_IC_TERM = 14.0  # FP64 flops per (electron energy, seed frequency) term of the SSC table build: CDF difference, bin integral, accumulate
F_IC_NODE = 240.0 # per lattice node of its set-up: one synchrotron spectrum / electron distribution / output evaluation
F_RHS = 100.0    # per right-hand side of the forward-shock ODE
F_TABLE = 30.0   # per tabulated SSC spectrum evaluation (ICPhoton::compute_log2_I_nu: index + linear interpolation)
PEAK_HBM_GBS = 8000.0    # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
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


def _cpu_lib():
    import _abi
    lib, kind = _abi.load_ref(), "reference"
    if lib is None:
        lib, kind = _abi.load_oracle(fast=True), "port"
        if lib is None:
            import subprocess
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle_fast.so"])
            lib = _abi.load_oracle(fast=True)
    return lib, kind


def cpu_rate(call, n_items, budget_s, unit, what, min_calls=2):
    """`call(i)` on ONE host thread for about budget_s seconds, cycling over n_items inputs."""
    lib_kind = _cpu_lib()[1]
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s or n < min_calls:
        call(n % n_items)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": unit, "cores": 1, "kind": lib_kind, "sample": f"{n} {what}, single thread, {dt:.1f} s"}


def cpu_rate_all_cores(call, n_items, budget_s, unit, what, counts=(8, 32, None)):
    """Same on many host cores, one model per thread (the reference's own scheme: ThreadPoolExecutor over walkers with the GIL
    released, fitting/samplers.py:59-91; ctypes drops the GIL too).  The box may expose more logical CPUs than its quota
    grants, so a few thread counts are tried and the best throughput is reported with the count that produced it."""
    from concurrent.futures import ThreadPoolExecutor
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    best, tried = None, {}
    for nthreads in sorted({min(k or ncpu, ncpu) for k in counts}):
        deadline = time.perf_counter() + budget_s

        def worker(w):
            n, i = 0, w
            while time.perf_counter() < deadline:
                call(i % n_items)
                n += 1
                i += nthreads
            return n

        t0 = time.perf_counter()
        with ThreadPoolExecutor(nthreads) as ex:
            total = sum(ex.map(worker, range(nthreads)))
        dt = time.perf_counter() - t0
        r = {"value": total / dt, "unit": unit, "cores": nthreads, "kind": _cpu_lib()[1],
             "sample": f"{total} {what}, {nthreads} threads, {dt:.1f} s (host exposes {ncpu} logical CPUs)"}
        tried[str(nthreads)] = r["value"]
        if best is None or r["value"] > best["value"]:
            best = r
    best["threads_tried"] = tried  # rate per thread count: the best one is reported as `value` with its count in `cores`
    return best


def timed_steps(step, steps, warmup, world, sync, barrier, max_over_ranks):
    """The timing contract of the driver: `warmup` untimed steps, then EXACTLY `steps` steps bracketed by a barrier and a device
    synchronisation on both sides; the elapsed time is the MAX over ranks.  Backend-agnostic (sync / barrier / max_over_ranks
    are callables) so that the N > 1 logic runs under gloo in the CPU tests."""
    for _ in range(warmup):
        step(False)
    sync()
    if world > 1:
        barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
    sync()
    if world > 1:
        barrier()
    sync()
    elapsed = time.perf_counter() - t0
    return max_over_ranks(elapsed) if world > 1 else elapsed


def _dist_helpers(world, dev):
    import torch
    import torch.distributed as dist

    def max_over_ranks(x):
        el = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el.item())

    sync = torch.cuda.synchronize if dev.type == "cuda" else (lambda: None)
    return sync, (dist.barrier if world > 1 else (lambda: None)), max_over_ranks


def _grid_call(lib, h, _lib, dev, prms, t, nu):
    """Device-resident flux_density_grid call over a list / ctypes array of parameter structs."""
    import torch
    import _abi
    nb = len(prms)
    arr = prms if isinstance(prms, C.Array) else (_abi.ModelParams * nb)(*prms)
    d_p = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    d_t, d_nu = torch.from_numpy(np.ascontiguousarray(t)).to(dev), torch.from_numpy(np.ascontiguousarray(nu)).to(dev)
    d_o = torch.empty((nb, nu.size, t.size), dtype=torch.float64, device=dev)
    keep = (d_p, d_t, d_nu)

    def call():
        _lib.check(lib.vag_flux_density_grid_batch_dev(h, d_p.data_ptr(), nb, d_t.data_ptr(), t.size, d_nu.data_ptr(), nu.size,
                                                       d_o.data_ptr()))
    call.out, call.keep, call.arr = d_o, keep, arr
    return call


def tophat_sweep(lib, h, _lib, dev, with_cpu):
    """BASELINE configs[0] (the >= 100x target config): top-hat + ISM, resolutions (0.089, 0.05, 12),
    100 times x 3 bands; single-call latency and batched throughput, on-axis (C1a) and theta_obs = 0.05 (C1b), next to the
    reference CPU path on this box (1 core and all cores)."""
    import torch
    import _abi
    import configs
    out = {}
    t, nu = configs.C1_T, configs.C1_NU
    for name, kw, batches in (("C1a_onaxis", configs.C1A, (1, 64, 1024, 4096, 16384)), ("C1b_theta_obs_0.05", configs.C1B, (1, 64, 1024))):
        res = {}
        for nb in batches:
            rng = np.random.default_rng(1)
            prms = []
            for i in range(nb):
                k = dict(kw)
                for key in ("E_iso", "n_ism", "eps_B"):
                    k[key] *= float(np.exp(rng.uniform(-0.1, 0.1)))
                prms.append(_abi.make_params(**k))
            call = _grid_call(lib, h, _lib, dev, prms, t, nu)
            call()
            torch.cuda.synchronize()
            reps = 20 if nb == 1 else 5
            t0 = time.perf_counter()
            for _ in range(reps):
                call()
                if nb == 1:
                    torch.cuda.synchronize()  # latency of ONE light curve: the caller waits for each
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            res[f"batch_{nb}"] = {"ms_per_call": 1e3 * dt, "light_curves_per_s": nb / dt}
        if with_cpu:
            cpu, _ = _cpu_lib()
            rng = np.random.default_rng(1)
            cprm = []
            for i in range(64):
                k = dict(kw)
                for key in ("E_iso", "n_ism", "eps_B"):
                    k[key] *= float(np.exp(rng.uniform(-0.1, 0.1)))
                cprm.append(_abi.make_params(**k))
            f = lambda i: cpu.flux_density_grid(cprm[i], t, nu)
            one = cpu_rate(f, 64, 2.0, "light-curves/s", "models of this config")
            allc = cpu_rate_all_cores(f, 64, 2.0, "light-curves/s", "models")
            res["cpu_baseline"], res["cpu_baseline_all_cores"] = one, allc
            best = max(v["light_curves_per_s"] for k2, v in res.items() if k2.startswith("batch_"))
            res["speedup_vs_reference"] = {"batched_vs_1_core": best / one["value"], "batched_vs_all_cores": best / allc["value"],
                                           "single_call_vs_1_core": res["batch_1"]["light_curves_per_s"] / one["value"]}
        out[name] = res
    return out


def ensemble_bench(lib, h, _lib, dev, with_cpu=True, c3_models=512):
    """BASELINE configs[2] and [4] as batched ensembles on one GPU: C3 = power-law jet in a wind, forward +
    reverse shock with SSC + Klein-Nishina on both (jittered parameters), 512 models per call (128 leave a third of a launch
    to the tail: 9.7 k wavefronts are < 4 rounds of the chip); C5 = prior-predictive sweep of two-component SSC
    jets, 1024 members.  100 times x 4 bands (incl. 2.4e26 Hz), device-resident inputs.  The flux passes' FP64 roofline uses the
    kernels' own work tallies (one extra untimed pass with vag_ctx_count_work): spectrum evaluations x 210 (synchrotron) or
    x 30 (tabulated SSC spectrum) + interpolations x 26, over the flux stage's HIP-event time."""
    import torch
    from configs import c3_batch, c5_batch
    t, nu = np.logspace(2, 8, 100), np.array([1e9, 4.84e14, 1e18, 2.4e26])
    out = {}
    for name, prms in (("C3_fs_rs_ssc_kn", c3_batch(c3_models)), ("C5_two_component_ssc", c5_batch(1024))):
        nb = len(prms)
        call = _grid_call(lib, h, _lib, dev, prms, t, nu)
        call()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        st = _lib.StageTimes()
        lib.vag_last_stage_times(h, C.byref(st))
        _lib.check(lib.vag_ctx_profile(h, 1))  # one more pass under the named-stage profiler (reference stage names)
        call()
        prof = _lib.Profile()
        _lib.check(lib.vag_last_profile(h, C.byref(prof)))
        _lib.check(lib.vag_ctx_profile(h, 0))
        _lib.check(lib.vag_ctx_count_work(h, 1))
        call()
        torch.cuda.synchronize()
        _lib.check(lib.vag_ctx_count_work(h, 0))
        plan = _lib.Plan()
        lib.vag_last_plan(h, C.byref(plan))
        # the tallies cover every flux pass of the call (synchrotron and SSC of each shock); half of the evaluations are the
        # tabulated SSC spectrum on these SSC-on configurations
        flops = plan.spec_evals * 0.5 * (F_SPEC + F_TABLE) + plan.interps * F_INTERP
        out[name] = {"batch": nb, "ms_per_batch": 1e3 * dt, "light_curves_per_s": nb / dt,
                     "finite": bool(torch.isfinite(call.out).all()), "ode_rows": plan.n_rows, "cells": plan.n_cells,
                     "ssc_table_pool_mb": plan.ic_pool_bytes / 1e6,  # per shock; the fixed 192-node layout took cells x 1584 B
                     "ssc_table_pool_mb_fixed_192_node_layout": plan.n_cells * 198 * 8 / 1e6,
                     "stage_ms": {"grid": st.grid_ms, "dynamics": st.dynamics_ms, "cells_cooling_tables": st.cells_ms,
                                  "flux_passes": st.flux_ms, "reduce": st.reduce_ms},
                     "stage_ms_reference_names": {n: getattr(prof, n) for n, _ in _lib.Profile._fields_},
                     "roofline_fp64_flux_passes": {"tally_note": "unit counts from one extra UNTIMED pass with vag_ctx_count_work, which runs the workgroup "
                                                                 "flux kernel (the row-per-lane kernels that serve the timed passes carry no tallies); "
                                                                 "the counts -- window-clamped evaluations and interpolations -- do not depend on the kernel",
                                                   "valu_busy_see": "valu_busy_from_pmc (vag_flux_grid_rows_kernel<1> / <2>)",
                                                   "spec_evals": plan.spec_evals, "interps": plan.interps,
                                                   "ms": prof.sync_flux + prof.ssc_flux,
                                                   "ms_note": "sync_flux + ssc_flux of the reference-named profile: the flux kernels alone (stage_ms.flux_passes "
                                                              "also holds the SSC table build, which has its own line below)",
                                                   "achieved": flops / ((prof.sync_flux + prof.ssc_flux) * 1e-3) / 1e12, "peak": PEAK_FP64_TFLOPS,
                                                   "unit": "TFLOP/s", "frac": flops / ((prof.sync_flux + prof.ssc_flux) * 1e-3) / 1e12 / PEAK_FP64_TFLOPS},
                     "roofline_fp64_ic_photons": {"kernel": "vag_ic_photon_kernel (+ seed band)", "ic_terms": plan.ic_terms, "ic_nodes": plan.ic_nodes,
                                                  "ms": prof.ic_photons,
                                                  "achieved": (plan.ic_terms * F_IC_TERM + plan.ic_nodes * F_IC_NODE) / max(prof.ic_photons * 1e-3, 1e-12) / 1e12,
                                                  "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                                                  "frac": (plan.ic_terms * F_IC_TERM + plan.ic_nodes * F_IC_NODE) / max(prof.ic_photons * 1e-3, 1e-12) / 1e12 / PEAK_FP64_TFLOPS},
                     }
        if with_cpu:  # the reference's own code on THIS box's host cores, same models, same request (about 1 s per model and core)
            cpu, _ = _cpu_lib()
            f = lambda i, prms=prms: cpu.flux_density_grid(prms[i], t, nu)
            one = cpu_rate(f, 8, 3.0, "light-curves/s", "models of the timed batch", min_calls=3)
            allc = cpu_rate_all_cores(f, 64, 4.0, "light-curves/s", "models", counts=(32, None))
            out[name]["cpu_baseline"], out[name]["cpu_baseline_all_cores"] = one, allc
            out[name]["speedup_vs_reference"] = {"vs_1_core": (nb / dt) / one["value"], "vs_all_cores": (nb / dt) / allc["value"]}
    return out


class ResidentMembers:
    """The parameter structs of an ensemble, resident in HBM; a slice is a view of the same device array (what
    dist.sharded_flux_density_grid hands a rank's evaluator), never a copy."""

    def __init__(self, arr, d_params, lo=0, hi=None):
        self.arr, self.d_params, self.lo, self.hi = arr, d_params, lo, len(arr) if hi is None else hi

    def __len__(self):
        return self.hi - self.lo

    def __getitem__(self, sl):
        if not isinstance(sl, slice):
            raise TypeError("slices only")
        a, b, _ = sl.indices(len(self))
        return ResidentMembers(self.arr, self.d_params, self.lo + a, self.lo + b)


def resident_members(prms, dev):
    import torch
    import _abi
    arr = prms if isinstance(prms, C.Array) else (_abi.ModelParams * len(prms))(*prms)
    return ResidentMembers(arr, torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev))


def grid_evaluator(lib, h, _lib, dev, t, nu, chunk=1024):
    """eval_dev(block: ResidentMembers) -> float64 tensor [k, n_nu, n_t] on the device, in calls of at most `chunk` members (a
    1024-member call already fills the GPU; the engine itself has no batch limit)."""
    import torch
    d_t, d_nu = torch.from_numpy(np.ascontiguousarray(t)).to(dev), torch.from_numpy(np.ascontiguousarray(nu)).to(dev)
    outs = {}
    psize = C.sizeof(_lib.ModelParams)

    def eval_dev(block):
        k = len(block)
        out = outs.get(k)
        if out is None:
            out = outs[k] = torch.empty((k, nu.size, t.size), dtype=torch.float64, device=dev)
        for a in range(0, k, chunk):
            n = min(chunk, k - a)
            _lib.check(lib.vag_flux_density_grid_batch_dev(h, block.d_params.data_ptr() + (block.lo + a) * psize, n, d_t.data_ptr(), t.size,
                                                           d_nu.data_ptr(), nu.size, out[a:].data_ptr()))
        return out
    eval_dev.keep = (d_t, d_nu)
    return eval_dev


def sharded_ensemble_leg(members, eval_dev, n_out, world, dev, steps, warmup, gather, sync, barrier, max_over_ranks):
    """One leg of the configs[4] measurement: `members` (the same object on every rank) through dist.sharded_flux_density_grid
    -- rank r evaluates shard_range(len(members), r, world), then one all-gather (gather=True) or none (gather=False: every rank
    keeps its own block) -- under the driver's timing contract.  Backend-agnostic (the CPU tests run it under gloo)."""
    from vegasafterglow_amd import dist as vdist
    last = {}

    def step(_record):
        last["res"] = vdist.sharded_flux_density_grid(members, eval_dev, n_out, device=dev, gather=gather)

    dt = timed_steps(step, steps, warmup, world, sync, barrier, max_over_ranks)
    n = len(members)
    return {"value": n * steps / dt, "unit": "light-curves/s", "members": n, "members_per_rank": -(-n // world), "steps": steps,
            "ms_per_step": 1e3 * dt / steps, "scaling": "strong", "gather": bool(gather)}, last["res"]


def ensemble_c5_sharded(lib, h, _lib, dev, world, n_members=4096, steps=3):
    """BASELINE configs[4] at its full size -- 4096 two-component SSC members, 100 times x 4 bands -- block-sharded over the
    ranks by dist.sharded_flux_density_grid, with and without the all-gather of the fluxes; at N = 1 also the 512-member share
    ONE rank of an 8-GPU run evaluates, timed alone, and the strong-scaling ratio it implies."""
    import torch
    from configs import c5_batch
    t, nu = np.logspace(2, 8, 100), np.array([1e9, 4.84e14, 1e18, 2.4e26])
    members = resident_members(c5_batch(n_members), dev)  # seeded: every rank builds the same ensemble
    ev = grid_evaluator(lib, h, _lib, dev, t, nu)
    sync, barrier, max_over_ranks = _dist_helpers(world, dev)
    out = {"workload": f"BASELINE configs[4]: {n_members} TwoComponentJet + ISM members (prior-predictive sweep), SSC on, "
                       "resolutions (0.59, 0.98, 12) -> ~128x128x111 cells/member, 100 times x 4 bands (incl. 2.4e26 Hz)"}
    for key, gather in (("all_gather_of_fluxes", True), ("no_collective", False)):
        res, got = sharded_ensemble_leg(members, ev, (nu.size, t.size), world, dev, steps, 1, gather, sync, barrier, max_over_ranks)
        blk = got if gather else got[2]
        res["finite"] = bool(torch.isfinite(blk).all())
        out[key] = res
    if world == 1:
        share = resident_members(members.arr[: n_members // 8], dev)
        res, _ = sharded_ensemble_leg(share, ev, (nu.size, t.size), 1, dev, steps, 1, False, sync, barrier, max_over_ranks)
        out["per_rank_share_of_8gpu"] = dict(res, note=f"the {n_members // 8} members rank 0 of an 8-GPU run evaluates, timed alone on one GPU")
        out["implied_8gpu_speedup_over_1gpu"] = out["no_collective"]["ms_per_step"] / res["ms_per_step"]
        out["implied_8gpu_light_curves_per_s"] = n_members / (res["ms_per_step"] * 1e-3)
    return out


def single_model_latency(lib, h, _lib, dev, cases, reps=10):
    """The reference's own protocol (pybind/pymodel.cpp:498-514): ONE model per call, the caller waits for each.  `cases` maps a
    name to (params, t, nu, reference light-curves/s on one core of this box or None)."""
    import torch
    out = {}
    for name, (prm, t, nu, ref_lc_s) in cases.items():
        call = _grid_call(lib, h, _lib, dev, [prm], t, nu)
        for _ in range(2):
            call()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            call()
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out[name] = {"ms_per_call": 1e3 * dt, "light_curves_per_s": 1.0 / dt, "grid": f"{nu.size} nu x {t.size} t"}
        if ref_lc_s:
            out[name]["reference_1_core_ms_per_call"] = 1e3 / ref_lc_s
            out[name]["speedup_vs_reference_1_core"] = (1.0 / dt) / ref_lc_s
    return out


def c4_fitter(lib, h, _lib):
    """The C4 problem: GW170817-like mock from the engine's own truth model (+5 % noise, 10 % errors), 3 bands x 20
    epochs, 8 free parameters with the prior box of configs.C4_FREE."""
    import configs
    from vegasafterglow_amd import fitting
    t, nu = configs.c4_mock_data()
    kw = configs.C4_TRUTH
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
    return fit, defs, (t, nu, f_obs)


def walker_bench(lib, h, _lib, dev, rank, world, steps=10, nwalkers=1024, sharder_cls=None, host_consumes=True, tally=False):
    """MCMC walker-steps/s on the C4 problem: `nwalkers` drawn uniformly from the prior box, evaluated through the product's
    sharded evaluator (dist.WalkerSharder: walkers dealt to the ranks by the engine's cost report, one all-gather of
    [ln L | cost] per call).  host_consumes: the host reads ln L after every call (pinned copy + stream synchronisation), as a
    sampler must before it can propose the next step; False queues the calls back to back."""
    import torch
    import torch.distributed as dist
    from vegasafterglow_amd.dist import WalkerSharder
    fit, defs, _ = c4_fitter(lib, h, _lib)
    spec, lo, hi = fit.build_spec(defs)
    theta = lo + (hi - lo) * np.random.default_rng(0).random((nwalkers, len(defs)))
    d_theta = torch.from_numpy(np.ascontiguousarray(theta)).to(dev)
    sharder = (sharder_cls or WalkerSharder)(fit.device_evaluator(defs, context=(h, _NullLock())), device=dev)
    h_ll = torch.empty((nwalkers,), dtype=torch.float64).pin_memory()
    last = {}

    def step(_record):
        ll = sharder(d_theta)
        if host_consumes:
            h_ll.copy_(ll, non_blocking=True)
            torch.cuda.current_stream().synchronize()
        last["ll"] = ll

    sync, barrier, max_over_ranks = _dist_helpers(world, dev)
    dt = timed_steps(step, steps, 2, world, sync, barrier, max_over_ranks)
    st = _lib.StageTimes()
    lib.vag_last_stage_times(h, C.byref(st))
    finite = int(torch.isfinite(last["ll"]).sum().item())
    res = {"value": nwalkers * steps / dt, "unit": "walker-steps/s", "walkers": nwalkers, "steps": steps, "ms_per_step": 1e3 * dt / steps,
           "scaling": "strong", "ln_L_read_by_host_every_call": host_consumes, "finite_walkers": finite,
           "rank0_stage_ms": {"grid": st.grid_ms, "dynamics": st.dynamics_ms, "syn_cells": st.cells_ms, "series_flux": st.flux_ms,
                              "reduce": st.reduce_ms}}
    if world > 1:
        cpr = sharder.costs_per_rank()
        if cpr is not None:
            res["cost_per_rank_max_over_mean"] = float(cpr.max() / cpr.mean())
    if tally and world == 1:
        # metric M2's roofline (FP64 VALU): one more UNTIMED step with vag_ctx_count_work -- the likelihood's flux kernel
        # tallies the boundary spectra it forms (x 210) and the (row, data point) interpolations inside a row's lattice (x 26), the
        # forward-shock solver its right-hand sides (x 100) -- over the timed step and over the two kernels' own stage times
        _lib.check(lib.vag_ctx_count_work(h, 1))
        step(False)
        torch.cuda.synchronize()
        plan = _lib.Plan()
        lib.vag_last_plan(h, C.byref(plan))
        _lib.check(lib.vag_ctx_count_work(h, 0))
        f_flux, f_ode = plan.spec_evals * F_SPEC + plan.interps * F_INTERP, plan.ode_rhs * F_RHS
        tf = lambda flops, ms: flops / max(ms * 1e-3, 1e-12) / 1e12
        res["roofline_fp64"] = {
            "bound": "fp64_valu", "unit": "TFLOP/s", "peak": PEAK_FP64_TFLOPS,
            "spec_evals": plan.spec_evals, "interps": plan.interps, "ode_rhs": plan.ode_rhs, "ode_rows": plan.n_rows,
            # live lanes over the ODE solver's step attempts / the lane slots those attempts occupied (a wavefront of the plain kernel runs
            # until its slowest row is done; the refill kernel of large batches takes new rows into finished lanes)
            "ode_lane_utilisation": plan.ode_lane_attempts / max(plan.ode_lane_slots, 1),
            "flop_eq_per_walker": (f_flux + f_ode) / nwalkers,
            "achieved": tf(f_flux + f_ode, 1e3 * dt / steps), "frac": tf(f_flux + f_ode, 1e3 * dt / steps) / PEAK_FP64_TFLOPS,
            "note": "whole step (grid, ODE, cells, flux, chi^2 and the host's read of ln L) against the flux + ODE work",
            "vag_flux_fit_rows_kernel": {"ms": st.flux_ms, "achieved": tf(f_flux, st.flux_ms), "frac": tf(f_flux, st.flux_ms) / PEAK_FP64_TFLOPS},
            # the ODE stage: vag_dynamics_fast_kernel (a latency chain: one lane per row, ~110 dependent steps -- on a 1024-walker batch the
            # fraction says how little of the chip 53.9 k rows can occupy, not how the kernel issues), or from 98 k rows on the persistent
            # lane-refill kernel with its preparation and queue-sorting launches
            "ode_stage": {"kernel": "vag_dynamics_refill_kernel (+ prep / scan / file)" if plan.n_rows >= 96 * 4 * 256 else "vag_dynamics_fast_kernel",
                          "ms": st.dynamics_ms, "achieved": tf(f_ode, st.dynamics_ms), "frac": tf(f_ode, st.dynamics_ms) / PEAK_FP64_TFLOPS}}
    return res


class _NullLock:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def walker_cpu_baseline(lib, h, _lib, budget_s=5.0):
    """The reference's per-walker likelihood on one host core: Model.flux_density at the walker's parameters over the 60 C4 data
    points (the chi^2 itself is negligible), prior draws of the same box."""
    import _abi
    import configs
    cpu, _ = _cpu_lib()
    fit, defs, (t, nu, _) = c4_fitter(lib, h, _lib)
    _, lo, hi = fit.build_spec(defs)
    theta = lo + (hi - lo) * np.random.default_rng(0).random((64, len(defs)))
    prms = []
    for s in theta:
        kw = dict(configs.C4_TRUTH)
        for (name, lg, _, _), v in zip(configs.C4_FREE, s):
            kw[{"theta_v": "theta_obs"}.get(name, name)] = 10 ** v if lg else v
        prms.append(_abi.make_params(**kw))
    order = np.argsort(t)
    ts, nus = np.ascontiguousarray(t[order]), np.ascontiguousarray(nu[order])
    f = lambda i: cpu.flux_density(prms[i], ts, nus)
    one = cpu_rate(f, 64, budget_s, "walker-steps/s", "prior draws of the C4 box (60 data points each)")
    allc = cpu_rate_all_cores(f, 64, 2.0, "walker-steps/s", "prior draws of the C4 box")  # the reference's own scheme: a thread pool over walkers
    return one, allc


def threadpool_bench(lib, h, _lib, n_threads=32, n_walkers=1024, rounds=3, wait_us=50):
    """The reference's own calling pattern, unmodified (fitting/samplers.py:59-91): eval_one per walker mapped over a
    ThreadPoolExecutor -- a Model per walker (Fitter._build_model), Model.flux_density at the C4 data's 60 (t, nu) points with the GIL
    released inside the engine, chi^2 in numpy -- on ONE per-device context.  Serialised (every call its own launch chain) against
    va.set_coalescing(True): the engine gathers the calls that wait at the same time into batch calls."""
    from concurrent.futures import ThreadPoolExecutor
    import _abi
    import configs
    import vegasafterglow_amd as va
    fit, defs, (t, nu, f_obs) = c4_fitter(lib, h, _lib)
    _, lo, hi = fit.build_spec(defs)
    theta = lo + (hi - lo) * np.random.default_rng(0).random((n_walkers, len(defs)))
    prms = []
    for s in theta:
        kw = dict(configs.C4_TRUTH)
        for (name, lg, _, _), v in zip(configs.C4_FREE, s):
            kw[{"theta_v": "theta_obs"}.get(name, name)] = 10 ** v if lg else v
        prms.append(_abi.make_params(**kw))
    order = np.argsort(t)
    ts, nus, fo = np.ascontiguousarray(t[order]), np.ascontiguousarray(nu[order]), f_obs[order]
    ln_fo, sig = np.log(fo), 0.1

    def eval_one(p):
        try:
            F = va.Model.from_params(p).flux_density(ts, nus).total
        except Exception:  # noqa: BLE001  (samplers.py:61-70: any failure scores -inf)
            return -np.inf
        r = (ln_fo - np.log(np.maximum(F, 1e-300))) / sig
        return -0.5 * float(np.dot(r, r))

    out = {"threads": n_threads, "walkers": n_walkers, "data_points": int(ts.size),
           "what": "eval_one (Model.from_params + flux_density + chi^2) over ThreadPoolExecutor(%d), the reference samplers' loop" % n_threads}
    for label, on in (("serialised", False), ("coalesced", True)):
        va.set_coalescing(on, max_batch=64, wait_us=wait_us)
        c0, b0 = va.coalescing_stats()
        with ThreadPoolExecutor(n_threads) as ex:
            list(ex.map(eval_one, prms[:64]))  # warm-up
            t0 = time.perf_counter()
            for _ in range(rounds):
                ll = list(ex.map(eval_one, prms))
            dt = (time.perf_counter() - t0) / rounds
        c1, b1 = va.coalescing_stats()
        out[label] = {"walker_steps_per_s": n_walkers / dt, "ms_per_1024_walkers": 1e3 * dt, "finite_walkers": int(np.isfinite(ll).sum())}
        if on:
            out[label]["mean_batch"] = (c1 - c0) / max(b1 - b0, 1)
    va.set_coalescing(False)
    out["coalesced_over_serialised"] = out["coalesced"]["walker_steps_per_s"] / out["serialised"]["walker_steps_per_s"]
    return out


CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config", "roofline", "cpu_baseline")
LINE_LIMIT = 4096  # the driver's parser gave up on the 21 KB line of round 5: the line is now short BY CONSTRUCTION


def _dig(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def _sig(x, n=5):
    """Numbers of the line at n significant digits (the side file keeps full precision)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    try:
        return float(f"{float(x):.{n}g}")
    except (TypeError, ValueError):
        return x


def compact_line(detail, detail_file=None):
    """The ONE line of stdout: exactly the driver's contract keys plus a flat `extra` of scalars; everything else bench.py measures
    lives in `detail` (stderr + the side file whose name the line carries).  Pure function of `detail`, so a CPU test can hold it to
    the length limit and to the contract keys on a recorded run (tests/test_abi_and_host.py)."""
    rf = detail.get("roofline") or {}
    cb = detail.get("cpu_baseline")
    line = {k: detail.get(k) for k in CONTRACT_KEYS if k not in ("config", "roofline", "cpu_baseline")}
    line["value"], line["ms_per_step"] = _sig(line.get("value"), 7), _sig(line.get("ms_per_step"), 6)
    cfg = detail.get("config") or {}
    line["config"] = {"workload": str(cfg.get("workload", ""))[:400], **{k: v for k, v in cfg.items() if k != "workload"}}
    line["roofline"] = {k: _sig(rf.get(k)) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "ms_per_launch", "valu_busy")}
    line["roofline"]["traffic"] = rf.get("traffic")
    line["cpu_baseline"] = None if not cb else {"value": _sig(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"),
                                                "kind": cb.get("kind"), "sample": str(cb.get("sample", ""))[:160]}
    ens = detail.get("ensembles_config2_config4") or {}
    c3, c5 = ens.get("C3_fs_rs_ssc_kn") or {}, ens.get("C5_two_component_ssc") or {}
    extra = {
        "hbm_GBps": _dig(detail, "roofline_hbm", "achieved"), "hbm_frac": _dig(detail, "roofline_hbm", "frac"),
        "cpu_all_cores_lc_per_s": _dig(detail, "cpu_baseline_all_cores", "value"), "cpu_all_cores_threads": _dig(detail, "cpu_baseline_all_cores", "cores"),
        "vs_reference_1_core": detail.get("vs_reference_1_core_same_box"), "vs_reference_all_cores": detail.get("vs_reference_all_cores_same_box"),
        "walker_steps_per_s": _dig(detail, "walker_steps", "value"), "walker_steps_fp64_frac": _dig(detail, "walker_steps", "roofline_fp64", "frac"),
        "walker_steps_8192_per_s": _dig(detail, "walker_steps_8192_total", "value"),
        "walker_8192_ode_ms": _dig(detail, "walker_steps_8192_total", "rank0_stage_ms", "dynamics"),
        "walker_8192_ode_lane_util": _dig(detail, "walker_steps_8192_total", "roofline_fp64", "ode_lane_utilisation"),
        "walker_cpu_1_core_per_s": _dig(detail, "walker_steps", "cpu_baseline", "value"),
        "c1a_batched_vs_all_cores": _dig(detail, "tophat_config0", "C1a_onaxis", "speedup_vs_reference", "batched_vs_all_cores"),
        "c1a_single_call_vs_1_core": _dig(detail, "tophat_config0", "C1a_onaxis", "speedup_vs_reference", "single_call_vs_1_core"),
        "c1a_single_call_ms": _dig(detail, "tophat_config0", "C1a_onaxis", "batch_1", "ms_per_call"),
        "c3_lc_per_s": c3.get("light_curves_per_s"), "c3_flux_frac": _dig(c3, "roofline_fp64_flux_passes", "frac"),
        "c3_table_frac": _dig(c3, "roofline_fp64_ic_photons", "frac"),
        "c5_lc_per_s": c5.get("light_curves_per_s"), "c5_flux_frac": _dig(c5, "roofline_fp64_flux_passes", "frac"),
        "c5_table_frac": _dig(c5, "roofline_fp64_ic_photons", "frac"),
        "implied_8gpu_ensemble_c5": _dig(detail, "ensemble_c5_4096", "implied_8gpu_speedup_over_1gpu"),
        "implied_8gpu_walkers_8192": _dig(detail, "walker_steps_per_rank_share_of_8gpu", "implied_8gpu_speedup_8192_walkers"),
        "implied_8gpu_walkers_1024": _dig(detail, "walker_steps_per_rank_share_of_8gpu", "implied_8gpu_speedup_over_1gpu"),
        "rank_share_128_walkers_ms": _dig(detail, "walker_steps_per_rank_share_of_8gpu", "128_walkers_per_rank", "ms_per_step"),
        "rccl_world": detail.get("rccl_world"),
    }
    line["extra"] = {k: _sig(v) for k, v in extra.items() if v is not None}
    line["detail"] = detail_file
    text = json.dumps(line, separators=(",", ":"))
    while len(text) >= LINE_LIMIT and line["extra"]:  # cannot happen with the keys above; the limit holds whatever a leg returns
        line["extra"].popitem()
        text = json.dumps(line, separators=(",", ":"))
    if len(text) >= LINE_LIMIT:
        line["config"]["workload"] = line["config"]["workload"][:120]
        text = json.dumps(line, separators=(",", ":"))
    return text


def emit(detail, real_stdout, world):
    """Detail -> stderr and the side file; the compact contract line -> the real stdout (rank 0 only calls this)."""
    name = os.environ.get("VAG_BENCH_DETAIL") or os.path.join(ROOT, "bench_detail.json" if world == 1 else f"bench_detail_n{world}.json")
    try:
        with open(name, "w") as f:
            json.dump(detail, f, indent=1)
        shown = os.path.relpath(name, ROOT)
    except OSError as e:  # a read-only tree must not cost the line
        shown = f"(not written: {e})"
    sys.stderr.write("[bench detail] " + json.dumps(detail) + "\n")
    sys.stderr.flush()
    os.write(real_stdout, (compact_line(detail, shown) + "\n").encode())


def headline_record(world, rccl_world, nb, steps, warmup, elapsed, st, plan, nnu, nt):
    """The contract part of the record (same keys at every N; rank 0 prints it).  `st` = mean stage times of the timed steps
    (grid, dynamics, cells, flux, reduce, total; HIP events on the kernels' own stream), `plan` = the tallied plan of one launch."""
    flux_s = st[3] * 1e-3
    # algorithmic work of ONE flux-kernel launch (DESIGN.md "Roofline accounting")
    alg_bytes = plan.n_cells * 18 * 8 + nb * nnu * nt * 8 + nb * (64 + 64) * 8
    alg_flops = plan.spec_evals * F_SPEC + plan.interps * F_INTERP
    return {
        "metric": "light-curves/sec (single model) and MCMC walker-steps/sec at 1/2/4/8 MI355X",
        "value": world * nb * steps / elapsed,
        "unit": "light-curves/s",
        "n_gpus": world, "rccl_world": rccl_world,  # the second is the RCCL group's own rank count (dist.get_world_size())
        "steps": steps, "warmup": warmup,
        "ms_per_step": 1e3 * elapsed / steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: GaussianJet off-axis (theta_obs=0.3) + ISM, synchrotron+SSA, "
                               "resolutions (0.355,0.31,20.5) -> ~64x64x199 cells/model, 200 time bins x 10 bands; "
                               "every physical parameter jittered +-10 % log-uniformly (ragged batch)",
                   "models_per_gpu_per_step": nb, "global_batch": world * nb, "parallelism": f"walker-shard x{world}"},
        # the kernel's bound is the FP64 vector ALU: flop-equivalents of ONE launch (210 per spectrum evaluation,
        # 26 per interpolation, tallied by the kernel in an untimed pass) over its HIP-event time; `traffic` = its HBM bytes by counters
        "roofline": {"bound": "fp64_valu", "kernel": "vag_flux_grid_kernel",
                     "achieved": alg_flops / flux_s / 1e12, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                     "frac": alg_flops / flux_s / 1e12 / PEAK_FP64_TFLOPS, "traffic": None,
                     "spec_evals_per_launch": plan.spec_evals, "interps_per_launch": plan.interps,
                     "flop_eq_per_launch": alg_flops, "ms_per_launch": st[3]},
        # BASELINE.json asks for HBM GB/s vs peak as well: the same launch's algorithmic bytes -- reported, not the bound
        "roofline_hbm": {"bound": "hbm (reported because BASELINE.json names it; the kernel is FP64-VALU bound)", "kernel": "vag_flux_grid_kernel",
                         "achieved": alg_bytes / flux_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": alg_bytes / flux_s / 1e9 / PEAK_HBM_GBS, "traffic": None,
                         "bytes_per_launch": alg_bytes, "ms_per_launch": st[3]},
        "stage_ms": {"grid": st[0], "dynamics": st[1], "syn_cells": st[2], "sync_flux": st[3], "reduce": st[4],
                     "total_device": st[5]},
        "plan": {"ode_rows": plan.n_rows, "cells": plan.n_cells, "theta_phi_rows": plan.total_pairs,
                 "flux_workgroups": plan.flux_blocks, "rows_per_workgroup": plan.pairs_per_block},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=512, help="models per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-walkers", action="store_true", help="skip everything but the headline measurement")
    args = ap.parse_args()

    # stdout carries ONE JSON line and nothing else: RCCL prints its version banner to the C stdout when a communicator is
    # created, so fd 1 points at stderr while the bench runs and the line is written to the real stdout at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

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
    _lib.check(lib.vag_ctx_set_stream(h, _lib.torch_stream_handle(stream)))

    nb = args.batch
    t_np, nu_np = configs.C2_T, configs.C2_NU
    nt, nnu = t_np.size, nu_np.size
    arr = c2_batch(nb, seed=1234 + rank)
    dev = torch.device("cuda", local_rank)
    call = _grid_call(lib, h, _lib, dev, arr, t_np, nu_np)
    d_out = call.out
    gathered = torch.empty((world * nb,), dtype=torch.float64, device=dev) if world > 1 else None
    flux_ms = []

    def step(record):
        call()
        if world > 1:
            # what a sampler consumes per model (here: the band-summed fluence proxy), 8 B/model over RCCL
            dist.all_gather_into_tensor(gathered, d_out.sum(dim=(1, 2)))
        if record:
            st = _lib.StageTimes()
            _lib.check(lib.vag_last_stage_times(h, C.byref(st)))  # HIP events on the kernel's own stream
            flux_ms.append((st.grid_ms, st.dynamics_ms, st.cells_ms, st.flux_ms, st.reduce_ms, st.total_ms))

    sync, barrier, max_over_ranks = _dist_helpers(world, dev)
    elapsed = timed_steps(step, args.steps, args.warmup, world, sync, barrier, max_over_ranks)

    # one extra UNTIMED pass with the kernel's work tallies on: exact spectrum-evaluation / interpolation counts
    _lib.check(lib.vag_ctx_count_work(h, 1))
    step(False)
    torch.cuda.synchronize()
    _lib.check(lib.vag_ctx_count_work(h, 0))
    plan = _lib.Plan()
    lib.vag_last_plan(h, C.byref(plan))
    if not bool(torch.isfinite(d_out).all()) or plan.n_models_ok != nb:
        raise SystemExit("bench produced non-finite fluxes or rejected models")
    extra = not args.no_walkers
    walkers = walker_bench(lib, h, _lib, dev, rank, world, tally=True) if extra else None
    walkers_half = walker_bench(lib, h, _lib, dev, rank, world, nwalkers=512) if extra else None
    walkers_queued = walker_bench(lib, h, _lib, dev, rank, world, host_consumes=False) if extra else None
    # an ensemble sized to the node (1024 walkers per GPU): the weak-scaling counterpart of the 1024-walker run above; and a
    # 8192-walker ensemble over all ranks (what a nested sampler's live-point pool or a large emcee ensemble hands over)
    # Both run at EVERY N, N = 1 included: the driver computes scaling from the per-N lines, so each curve needs its origin.
    walkers_weak = walker_bench(lib, h, _lib, dev, rank, world, nwalkers=1024 * world) if extra else None
    walkers_8192 = walker_bench(lib, h, _lib, dev, rank, world, nwalkers=8192, steps=5, tally=True) if extra else None
    # configs[4] ("sharded 8xMI355X") at its full 4096 members through dist.sharded_flux_density_grid, at every N as well
    ensemble_c5 = ensemble_c5_sharded(lib, h, _lib, dev, world) if extra else None
    shares = sharded1 = None
    if extra and world == 1:  # what ONE rank of an 8-GPU run evaluates per call, measured here: the strong-scaling ceiling
        s128 = walker_bench(lib, h, _lib, dev, 0, 1, nwalkers=128)
        s64 = walker_bench(lib, h, _lib, dev, 0, 1, nwalkers=64)
        s1024of8192 = walker_bench(lib, h, _lib, dev, 0, 1, nwalkers=1024, steps=5)  # a rank's share of the 8192-walker step
        shares = {"128_walkers_per_rank": s128, "64_walkers_per_rank_redblue": s64,
                  "implied_8gpu_speedup_8192_walkers": walkers_8192["ms_per_step"] / s1024of8192["ms_per_step"],
                  "implied_8gpu_walker_steps_per_s_8192_walkers": 8192.0 / (s1024of8192["ms_per_step"] * 1e-3),
                  "implied_note": "strong scaling of a 1024-walker step is bounded by one model's grid + ODE chain (~0.5 ms of the 0.7 ms "
                                  "a 128-walker call takes); the 8192-walker step gives every rank a full 1024-walker call; weak "
                                  "scaling (1024 walkers per GPU) has no shared work at all besides the 16 B/walker all-gather",
                  "implied_8gpu_walker_steps_per_s": 1024.0 / (s128["ms_per_step"] * 1e-3),
                  "implied_8gpu_redblue_walker_steps_per_s": 512.0 / (s64["ms_per_step"] * 1e-3),
                  "implied_8gpu_speedup_over_1gpu": walkers["ms_per_step"] / s128["ms_per_step"],
                  "note": "per-rank block of a 1024-walker (512 red-blue) step at 8 GPUs, timed on one GPU without the all-gather"}
        # The code an N-GPU run executes, timed on one GPU: a world-size-1 RCCL group makes WalkerSharder take its sharded branch
        # (device-side deal, the rank's block, all_gather_into_tensor over RCCL, scatter) instead of the plain call above.
        import socket
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        # a private tcp:// rendezvous; under torch.distributed.run (N = 1 launched through it) c10d would otherwise connect to the
        # launcher's store as a client -- on a port nobody serves -- and wait for its timeout
        agent_store = os.environ.pop("TORCHELASTIC_USE_AGENT_STORE", None)
        try:
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                    device_id=torch.device("cuda", local_rank))
        finally:
            if agent_store is not None:
                os.environ["TORCHELASTIC_USE_AGENT_STORE"] = agent_store
        try:
            plain = {1024: walkers, 512: walkers_half, 128: s128, 64: s64}
            sharded1 = {"note": "dist.WalkerSharder's sharded branch over a world-size-1 nccl (RCCL) group on this GPU vs the plain "
                                "evaluator call; ln L read by the host after every call in both"}
            for n in (1024, 512, 128, 64):
                r = walker_bench(lib, h, _lib, dev, 0, 1, nwalkers=n)
                sharded1[f"{n}_walkers"] = {"ms_per_step_sharded": r["ms_per_step"], "ms_per_step_plain": plain[n]["ms_per_step"],
                                            "sharder_overhead_ms": r["ms_per_step"] - plain[n]["ms_per_step"],
                                            "walker_steps_per_s_sharded": r["value"], "finite_walkers": r["finite_walkers"]}
        finally:
            dist.destroy_process_group()
    with_cpu = not args.no_cpu_baseline and world == 1
    tophat = tophat_sweep(lib, h, _lib, dev, with_cpu) if (extra and world == 1) else None
    ensembles = ensemble_bench(lib, h, _lib, dev, with_cpu) if (extra and world == 1) else None
    threadpool = threadpool_bench(lib, h, _lib) if (extra and world == 1) else None
    single = None
    if extra and world == 1:  # the metric reads "light-curves/sec (single model)": one model per call for configs[1] / [2] / [4]
        from configs import c3_batch, c5_batch
        t_e, nu_e = np.logspace(2, 8, 100), np.array([1e9, 4.84e14, 1e18, 2.4e26])
        ref = lambda k: (ensembles or {}).get(k, {}).get("cpu_baseline", {}).get("value")
        single = single_model_latency(lib, h, _lib, dev, {
            "C2_gaussian_offaxis": (arr[0], t_np, nu_np, None),  # the reference's one-core rate is filled in below
            "C3_fs_rs_ssc_kn": (c3_batch(1)[0], t_e, nu_e, ref("C3_fs_rs_ssc_kn")),
            "C5_two_component_ssc": (c5_batch(1)[0], t_e, nu_e, ref("C5_two_component_ssc"))})

    if rank == 0:
        out = headline_record(world, dist.get_world_size() if world > 1 else 1, nb, args.steps, args.warmup, elapsed,
                              np.mean(np.array(flux_ms), axis=0), plan, nnu, nt)
        # HBM traffic of the dominant kernel from the committed rocprofv3 PMC passes of this same command
        # (profiles/run_profile.sh; FETCH_SIZE x2 per the gfx950 note, calibrated on a kernel with known bytes)
        for name in ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
            tp = os.path.join(ROOT, "profiles", name)
            if os.path.exists(tp) and nb == 512 and world == 1:
                out["roofline"]["traffic"] = out["roofline_hbm"]["traffic"] = json.load(open(tp))["traffic_bytes_per_launch"]
                out["roofline"]["traffic_unit"] = "HBM bytes per launch (FETCH_SIZE + WRITE_SIZE)"
                out["roofline"]["traffic_source"] = f"profiles/{name}: static file from a separate rocprofv3 --pmc pass of this command (not measured in this run)"
                break
        # VALU-pipe busy fractions of the kernels the rooflines name, from committed rocprofv3 --pmc passes (SQ_INSTS_VALU x 4 /
        # (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)): the machine's own view next to the convention-based `frac`s
        for name in ("r06_valu_busy.json", "r05_valu_busy.json", "r04_valu_busy.json", "r03_valu_busy.json"):
            vp = os.path.join(ROOT, "profiles", name)
            if os.path.exists(vp):
                vb = json.load(open(vp))
                out["roofline"]["valu_busy"] = vb.get("vag_flux_grid_kernel<C2>")
                out["valu_busy_from_pmc"] = dict(vb, source=f"profiles/{name} (static, separate --pmc passes)")
                break
        if walkers is not None:
            out["walker_steps"] = walkers
            out["walker_steps_redblue_half"] = walkers_half  # emcee red-blue moves evaluate nwalkers/2 per call
            out["walker_steps_queued_back_to_back"] = walkers_queued
            if walkers_weak:
                walkers_weak["scaling"] = "weak"
                out["walker_steps_1024_per_gpu"] = walkers_weak
            if walkers_8192:
                out["walker_steps_8192_total"] = walkers_8192
            if ensemble_c5:
                out["ensemble_c5_4096"] = ensemble_c5
            if sharded1:
                out["walker_steps_sharded_branch_world1"] = sharded1
            if shares:
                out["walker_steps_per_rank_share_of_8gpu"] = shares
        if tophat is not None:
            out["tophat_config0"] = tophat
            c1a = tophat.get("C1a_onaxis", {}).get("speedup_vs_reference")
            if c1a:
                allc = tophat["C1a_onaxis"]["cpu_baseline_all_cores"]
                out["north_star_100x_on_config0"] = {
                    "batched_vs_reference_1_core": c1a["batched_vs_1_core"],
                    "batched_vs_reference_all_cores": c1a["batched_vs_all_cores"],
                    "met_vs_all_cores": bool(c1a["batched_vs_all_cores"] >= 100.0),
                    "caveat": f"'all cores' = the best of {sorted(int(k) for k in allc.get('threads_tried', {}))} host threads "
                              f"({allc['cores']} won: {allc['value'] / tophat['C1a_onaxis']['cpu_baseline']['value']:.1f}x one core) on a box that "
                              "exposes more logical CPUs than its quota delivers; against a host that scales with its core count the "
                              "same batched rate would be a correspondingly smaller multiple, and the single-call rate is only "
                              f"{c1a['single_call_vs_1_core']:.2f}x one core"}
        if ensembles:
            out["ensembles_config2_config4"] = ensembles
        if threadpool:
            out["walker_steps_threadpool_32"] = threadpool
        if single:
            out["single_model_latency"] = single
        if with_cpu:
            cpu, _ = _cpu_lib()
            f = lambda i: cpu.flux_density_grid(arr[i], t_np, nu_np)
            out["cpu_baseline"] = cpu_rate(f, len(arr), 12.0, "light-curves/s", "models of the timed batch (C2: 200 t x 10 nu)")
            out["cpu_baseline_all_cores"] = cpu_rate_all_cores(f, len(arr), 4.0, "light-curves/s", "models")
            # `vs_baseline` stays null: BASELINE.md holds no published number for this metric on this config (its table is a
            # different protocol on an Apple M2).  The same-box ratio against the reference's own thread-pool scheme is this:
            out["vs_reference_all_cores_same_box"] = out["value"] / out["cpu_baseline_all_cores"]["value"]
            out["vs_reference_1_core_same_box"] = out["value"] / out["cpu_baseline"]["value"]
            if single and "C2_gaussian_offaxis" in single:
                c2s = single["C2_gaussian_offaxis"]
                c2s["reference_1_core_ms_per_call"] = 1e3 / out["cpu_baseline"]["value"]
                c2s["speedup_vs_reference_1_core"] = c2s["light_curves_per_s"] / out["cpu_baseline"]["value"]
            if walkers is not None:
                wc, wall = walker_cpu_baseline(lib, h, _lib)
                out["walker_steps"]["cpu_baseline"] = wc
                out["walker_steps"]["cpu_baseline_all_cores"] = wall
                out["walker_steps"]["speedup_vs_reference_1_core"] = walkers["value"] / wc["value"]
                out["walker_steps"]["speedup_vs_reference_all_cores"] = walkers["value"] / wall["value"]
        emit(out, real_stdout, world)
    lib.vag_ctx_destroy(h)
    if world > 1:
        dist.destroy_process_group()
    sys.stdout.flush()
    C.CDLL(None).fflush(None)  # whatever C stdio still buffers goes to stderr too
    os.close(real_stdout)


if __name__ == "__main__":
    main()
