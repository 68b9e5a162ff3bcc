"""Can the per-call chain (grid + ODE: 0.5 ms that do not shrink with the batch) be hidden behind another call's flux pass?
Two contexts (each its own HIP stream), two host threads, each evaluating its own walker block of the C4 likelihood and reading
ln L after every call -- e.g. two independent ensembles or two temperatures of one sampler -- against ONE context doing the same
walkers in one call.  (One emcee ensemble cannot do this: its two half-steps depend on each other.)
usage: python profiles/two_contexts_probe.py  > profiles/r03_two_contexts_probe.txt"""
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import bench  # noqa: E402
from vegasafterglow_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")


def make(n):
    h = C.c_void_p()
    _lib.check(lib.vag_ctx_create(0, C.byref(h)))
    fit, defs, _ = bench.c4_fitter(lib, h, _lib)
    _, lo, hi = fit.build_spec(defs)
    ev = fit.device_evaluator(defs, context=(h, bench._NullLock()))
    stream = torch.cuda.Stream(dev)
    theta = torch.from_numpy(np.ascontiguousarray(lo + (hi - lo) * np.random.default_rng(n).random((n, len(defs))))).to(dev)
    pinned = torch.empty((n,), dtype=torch.float64).pin_memory()
    return ev, stream, theta, pinned, fit  # (the evaluator's spec points into the fitter's arrays)


def run(ctx, steps, out, k):
    ev, stream, theta, pinned, _ = ctx
    with torch.cuda.stream(stream):
        for _ in range(3):
            pinned.copy_(ev(theta)[0], non_blocking=True)
            stream.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            pinned.copy_(ev(theta)[0], non_blocking=True)
            stream.synchronize()
        out[k] = time.perf_counter() - t0


STEPS = 200
for total in (2048, 1024, 512, 256):
    one = make(total)
    res = [0.0]
    run(one, STEPS, res, 0)
    single = total * STEPS / res[0]
    pair = [make(total // 2), make(total // 2)]
    res = [0.0, 0.0]
    th = [threading.Thread(target=run, args=(pair[k], STEPS, res, k)) for k in range(2)]
    t0 = time.perf_counter()
    [t.start() for t in th]
    [t.join() for t in th]
    wall = time.perf_counter() - t0
    both = total * STEPS / max(res)
    print(f"{total:5d} walkers: one context, one call per step {1e3 * total / single:.3f} ms -> {single / 1e3:7.1f} k walker-steps/s;  "
          f"two contexts x {total // 2} walkers on two threads {1e3 * max(res) / STEPS:.3f} ms per step each -> {both / 1e3:7.1f} k walker-steps/s  ({both / single:.2f}x)")
