"""Developer probe (GPU box): the C4 walker step's stage times at a few batch sizes for the library VAG_LIB_PATH names."""
import ctypes as C, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from vegasafterglow_amd import _lib
lib = _lib.load()
h = C.c_void_p()
_lib.check(lib.vag_ctx_create(0, C.byref(h)))
dev = torch.device("cuda", 0)
_lib.check(lib.vag_ctx_set_stream(h, _lib.torch_stream_handle(torch.cuda.current_stream())))
for n in [int(x) for x in (sys.argv[1:] or ["128", "1024", "8192"])]:
    best = None
    for rep in range(3):
        r = bench.walker_bench(lib, h, _lib, dev, 0, 1, steps=10 if n <= 1024 else 4, nwalkers=n)
        if best is None or r["ms_per_step"] < best["ms_per_step"]:
            best = r
    print("walkers %5d  %.4f ms per step  series_flux %.4f  (best of 3)" % (n, best["ms_per_step"], best["rank0_stage_ms"]["series_flux"]))
