"""The per-cell radiation following the ODE solve row by row on a second stream (VAG_CELLS_FOLLOW=1) against the separate cells launch:
ln L bitwise, step and stage times at several walker counts.  usage: python3 profiles/debug/follow_probe.py [nwalkers ...]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from vegasafterglow_amd import _lib  # noqa: E402

lib = _lib.load()
h = C.c_void_p()
_lib.check(lib.vag_ctx_create(0, C.byref(h)))
_lib.check(lib.vag_ctx_set_stream(h, _lib.torch_stream_handle(torch.cuda.current_stream())))
dev = torch.device("cuda", 0)
fit, defs, _ = bench.c4_fitter(lib, h, _lib)
spec, lo, hi = fit.build_spec(defs)
ev = fit.device_evaluator(defs, context=(h, bench._NullLock()))
for nw in [int(a) for a in sys.argv[1:]] or [128, 1024, 8192]:
    theta = torch.from_numpy(lo + (hi - lo) * np.random.default_rng(0).random((nw, len(defs)))).to(dev)
    ref = None
    for mode in ("0", "1", "0", "1"):
        _lib.hooks["VAG_CELLS_FOLLOW"] = mode
        ll, _ = ev(theta)
        ll = ll.cpu().numpy()
        if ref is None:
            ref = ll
        same = np.array_equal(ll, ref, equal_nan=True)
        r = bench.walker_bench(lib, h, _lib, dev, 0, 1, nwalkers=nw, steps=20 if nw <= 1024 else 5)
        _lib.hooks.pop("VAG_CELLS_FOLLOW")
        print(f"walkers {nw} follow {mode}: ln L same bits {same} (finite {int(np.isfinite(ll).sum())}); step {r['ms_per_step']:.3f} ms, stages {({k: round(v, 3) for k, v in r['rank0_stage_ms'].items()})}", flush=True)
        assert same
