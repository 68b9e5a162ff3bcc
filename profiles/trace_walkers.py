"""Developer probe: run the C4 log-likelihood at N walkers a few times (for rocprofv3 --kernel-trace timelines)."""
import ctypes as C, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from vegasafterglow_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
lib = _lib.load()
h = C.c_void_p()
_lib.check(lib.vag_ctx_create(0, C.byref(h)))
dev = torch.device("cuda", 0)
_lib.check(lib.vag_ctx_set_stream(h, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
r = bench.walker_bench(lib, h, _lib, dev, 0, 1, steps=10, nwalkers=n)
print(r["ms_per_step"], r["rank0_stage_ms"])
