"""Developer probe (GPU box): wall time of the C4 walker step (device evaluator + pinned host copy of ln L), no stage times."""
import ctypes as C, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench
from vegasafterglow_amd import _lib
lib = _lib.load(); h = C.c_void_p(); _lib.check(lib.vag_ctx_create(0, C.byref(h)))
dev = torch.device("cuda", 0)
_lib.check(lib.vag_ctx_set_stream(h, _lib.torch_stream_handle(torch.cuda.current_stream())))
fit, defs, _ = bench.c4_fitter(lib, h, _lib)
spec, lo, hi = fit.build_spec(defs)
ev = fit.device_evaluator(defs, context=(h, bench._NullLock()))
for n in [int(x) for x in (sys.argv[1:] or ["1", "128", "1024"])]:
    theta = lo + (hi - lo) * np.random.default_rng(0).random((n, len(defs)))
    d_theta = torch.from_numpy(np.ascontiguousarray(theta)).to(dev)
    h_ll = torch.empty((n,), dtype=torch.float64).pin_memory()
    best = 1e9
    for rep in range(5):
        for _ in range(3):
            r = ev(d_theta); h_ll.copy_(r[0] if isinstance(r, tuple) else r, non_blocking=True); torch.cuda.current_stream().synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            r = ev(d_theta); h_ll.copy_(r[0] if isinstance(r, tuple) else r, non_blocking=True); torch.cuda.current_stream().synchronize()
        best = min(best, (time.perf_counter() - t0) / 20)
    print("walkers %5d  %.4f ms per step (best of 5 x 20)" % (n, best * 1e3), flush=True)
