"""Developer probe: does the order of the walkers in the batch matter to the likelihood's flux pass?  (dispatch is in batch order:
long blocks last leave the GPU draining)  Times the C4 call with the walkers as drawn, sorted by descending / ascending cost."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
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
fit, defs, _ = bench.c4_fitter(lib, h, _lib)
_, lo, hi = fit.build_spec(defs)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
theta = lo + (hi - lo) * np.random.default_rng(0).random((n, len(defs)))
ev = fit.device_evaluator(defs, context=(h, bench._NullLock()))


def timed(th):
    d = torch.from_numpy(np.ascontiguousarray(th)).to(dev)
    for _ in range(3):
        vals, costs = ev(d)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(10):
        vals, costs = ev(d)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    st = _lib.StageTimes()
    lib.vag_last_stage_times(h, C.byref(st))
    return 1e3 * dt, st.flux_ms, st.dynamics_ms, costs.cpu().numpy()


ms, flux, dyn, costs = timed(theta)
print("as drawn:        call %.3f ms  series flux %.3f  dynamics %.3f" % (ms, flux, dyn))
for name, order in (("descending cost", np.argsort(-costs)), ("ascending cost", np.argsort(costs))):
    ms, flux, dyn, _ = timed(theta[order])
    print("%-16s call %.3f ms  series flux %.3f  dynamics %.3f" % (name + ":", ms, flux, dyn))
