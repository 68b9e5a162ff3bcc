"""Developer probe: plan and stage times of the C4 walker likelihood at a given walker count (rows, pairs, lattice length, LDS per
series wavefront), for sizing the series kernel.  With a -DVAG_SERIES_STAMPS library it also prints one wavefront's phase cycles."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import bench  # noqa: E402
from vegasafterglow_amd import _lib  # noqa: E402

lib = _lib.load()
h = C.c_void_p()
_lib.check(lib.vag_ctx_create(0, C.byref(h)))
dev = torch.device("cuda", 0)
_lib.check(lib.vag_ctx_set_stream(h, _lib.torch_stream_handle(torch.cuda.current_stream())))
for n in [int(x) for x in (sys.argv[1:] or ["1024"])]:
    r = bench.walker_bench(lib, h, _lib, dev, 0, 1, steps=10, nwalkers=n)
    pl = _lib.Plan()
    lib.vag_last_plan(h, C.byref(pl))
    print(json.dumps({"walkers": n, "ms": r["ms_per_step"], **r["rank0_stage_ms"], "rows": pl.n_rows, "cells": pl.n_cells,
                      "pairs": pl.total_pairs, "K_mean": pl.n_cells / max(pl.n_rows, 1), "flux_blocks": pl.flux_blocks,
                      "pairs_per_block": pl.pairs_per_block}), flush=True)
