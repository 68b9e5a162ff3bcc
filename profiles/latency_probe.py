"""Developer probe: per-stage device times and wall time per call of the latency-bound shapes --
one C1a / C1b light curve, and the C4 log-likelihood at the walker counts one rank sees at 1..8 GPUs."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import bench  # noqa: E402
import _abi  # noqa: E402
import configs  # noqa: E402
from vegasafterglow_amd import _lib  # noqa: E402

lib = _lib.load()
h = C.c_void_p()
_lib.check(lib.vag_ctx_create(0, C.byref(h)))
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream()
_lib.check(lib.vag_ctx_set_stream(h, _lib.torch_stream_handle(stream)))
out = {}
t, nu = configs.C1_T, configs.C1_NU
d_t, d_nu = torch.from_numpy(t).to(dev), torch.from_numpy(nu).to(dev)
for name, kw in (("C1a", configs.C1A), ("C1b", configs.C1B)):
    arr = (_abi.ModelParams * 1)(_abi.make_params(**kw))
    d_p = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    d_o = torch.empty((1, nu.size, t.size), dtype=torch.float64, device=dev)
    call = lambda: _lib.check(lib.vag_flux_density_grid_batch_dev(h, d_p.data_ptr(), 1, d_t.data_ptr(), t.size, d_nu.data_ptr(),
                                                                    nu.size, d_o.data_ptr()))
    call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        call()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 20
    st = _lib.StageTimes()
    lib.vag_last_stage_times(h, C.byref(st))
    out[name] = {"wall_ms": 1e3 * wall, "grid": st.grid_ms, "dyn": st.dynamics_ms, "cells": st.cells_ms, "flux": st.flux_ms,
                 "reduce": st.reduce_ms, "device_total": st.total_ms}
    print(name, json.dumps(out[name]), flush=True)
for n in (1, 16, 64, 128, 256, 512, 1024):
    r = bench.walker_bench(lib, h, _lib, dev, 0, 1, steps=10, nwalkers=n)
    out[f"C4_{n}"] = {"ms_per_call": r["ms_per_step"], **r["rank0_stage_ms"]}
    print(f"C4 walkers {n}:", json.dumps(out[f"C4_{n}"]), flush=True)
print(json.dumps(out))
