"""Stage times of the configs[0] top-hat batches the bench times (device-resident call), C1a on-axis and C1b."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _abi, configs, bench
from vegasafterglow_amd import _lib
lib = _lib.load(); h = C.c_void_p(); _lib.check(lib.vag_ctx_create(0, C.byref(h)))
dev = torch.device("cuda", 0)
_lib.check(lib.vag_ctx_set_stream(h, _lib.torch_stream_handle(torch.cuda.current_stream())))
t, nu = configs.C1_T, configs.C1_NU
for name, kw in (("C1a", configs.C1A), ("C1b", configs.C1B)):
    for nb in (1024, 4096):
        rng = np.random.default_rng(1)
        prms = []
        for i in range(nb):
            k = dict(kw)
            for key in ("E_iso", "n_ism", "eps_B"):
                k[key] *= float(np.exp(rng.uniform(-0.1, 0.1)))
            prms.append(_abi.make_params(**k))
        call = bench._grid_call(lib, h, _lib, dev, prms, t, nu)
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        for _ in range(5):
            call()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        st = _lib.StageTimes(); lib.vag_last_stage_times(h, C.byref(st))
        pl = _lib.Plan(); lib.vag_last_plan(h, C.byref(pl))
        print(name, nb, "ms/call %.3f" % (1e3 * dt), {f: round(getattr(st, f), 3) for f, _ in _lib.StageTimes._fields_},
              "rows", pl.n_rows, "pairs", pl.total_pairs, "blocks", pl.flux_blocks, "ppb", pl.pairs_per_block, flush=True)
