"""Developer probe (GPU box): the C4 walker step with N walkers that are all the SAME prior draw (every model the same number of
64-row blocks: no empty workgroups in the fit kernel's launch, no imbalance) against N different draws.  argv: N [tile_index]"""
import ctypes as C, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import bench
from vegasafterglow_amd import _lib
lib = _lib.load()
h = C.c_void_p()
_lib.check(lib.vag_ctx_create(0, C.byref(h)))
dev = torch.device("cuda", 0)
_lib.check(lib.vag_ctx_set_stream(h, _lib.torch_stream_handle(torch.cuda.current_stream())))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
tile = int(sys.argv[2]) if len(sys.argv) > 2 else -1
fit, defs, _ = bench.c4_fitter(lib, h, _lib)
spec, lo, hi = fit.build_spec(defs)
theta = lo + (hi - lo) * np.random.default_rng(0).random((n, len(defs)))
if tile >= 0:
    theta[:] = theta[tile]
d_theta = torch.from_numpy(np.ascontiguousarray(theta)).to(dev)
ev = fit.device_evaluator(defs, context=(h, bench._NullLock()))
for rep in range(4):
    ll = ev(d_theta)
    torch.cuda.synchronize()
st = _lib.StageTimes()
lib.vag_last_stage_times(h, C.byref(st))
pl = _lib.Plan()
lib.vag_last_plan(h, C.byref(pl))
print("walkers %d tile %d: flux %.3f ms  grid %.3f dyn %.3f cells %.3f  pairs %d blocks launched %d (x64 = %d rows)" % (
    n, tile, st.flux_ms, st.grid_ms, st.dynamics_ms, st.cells_ms, pl.total_pairs, pl.flux_blocks, pl.flux_blocks * 64))
