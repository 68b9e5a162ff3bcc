"""Where do the integrator / saver wavefronts of the persistent ODE kernel land?  Needs a -DVAG_DYN_PLACEMENT build (profiles/build_variant.sh
placement -DVAG_DYN_PLACEMENT; VAG_LIB_PATH=variants/libvag_placement.so).  Prints, per (xcc, se, sh, cu), the SIMDs of the integrators."""
import collections
import ctypes as C
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import bench
    from vegasafterglow_amd import _lib
    lib = _lib.load()
    h = C.c_void_p()
    _lib.check(lib.vag_ctx_create(0, C.byref(h)))
    _lib.check(lib.vag_ctx_set_stream(h, _lib.torch_stream_handle(torch.cuda.current_stream())))
    _lib.hooks["VAG_DYN_REFILL"] = "1"
    fit, defs, _ = bench.c4_fitter(lib, h, _lib)
    import numpy as np
    spec, lo, hi = fit.build_spec(defs)
    theta = lo + (hi - lo) * np.random.default_rng(0).random((8192, len(defs)))
    ev = fit.device_evaluator(defs, context=(h, bench._NullLock()))
    ll, _ = ev(torch.from_numpy(theta).to("cuda:0"))
    torch.cuda.synchronize()
    sys.exit(0)
out = subprocess.run([sys.executable, __file__, "child"], capture_output=True, text=True).stdout
cus = collections.defaultdict(lambda: {0: [], 1: []})
for m in re.finditer(r"P (\d+) (\d) xcc (\d+) se (\d+) sh (\d+) cu (\d+) simd (\d+) wave (\d+)", out):
    b, role, xcc, se, sh, cu, simd, wave = map(int, m.groups())
    cus[(xcc, se, sh, cu)][role].append((b, simd))
hist = collections.Counter()
for key in sorted(cus):
    ints = sorted(s for _, s in cus[key][0])
    hist[tuple(ints)] += 1
print("CUs seen:", len(cus))
print("integrator SIMDs per CU -> number of CUs:")
for k, v in sorted(hist.items(), key=lambda kv: -kv[1]):
    print("  ", k, v)
for key in sorted(cus)[:6]:
    print(key, "integrators (block, simd):", sorted(cus[key][0]), "savers:", sorted(cus[key][1]))
