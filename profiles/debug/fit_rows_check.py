"""Developer check: the row-per-lane fit kernel against the row-per-wavefront series kernel on C4 prior draws: agreement,
run-to-run and batch / sub-batch bitwise equality."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402

import bench  # noqa: E402
from vegasafterglow_amd import _lib  # noqa: E402

lib = _lib.load()
h = C.c_void_p()
_lib.check(lib.vag_ctx_create(0, C.byref(h)))
fit, defs, _ = bench.c4_fitter(lib, h, _lib)
_, lo, hi = fit.build_spec(defs)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
theta = lo + (hi - lo) * np.random.default_rng(1).random((n, len(defs)))
theta[3, 3] = 0.0  # an on-axis walker
_lib.hooks.pop("VAG_SERIES_ROW_PER_WAVE", None)
a = fit.loglike_batch(theta, defs)
a2 = fit.loglike_batch(theta, defs)
c = fit.loglike_batch(theta[:7], defs)
one = fit.loglike_batch(theta[5:6], defs)
_lib.hooks["VAG_SERIES_ROW_PER_WAVE"] = "1"
b = fit.loglike_batch(theta, defs)
_lib.hooks.pop("VAG_SERIES_ROW_PER_WAVE", None)
fin = np.isfinite(b)
print("finite", fin.sum(), "of", n, " same finite set:", np.array_equal(np.isfinite(a), fin), " max rel diff vs row-per-wave",
      np.max(np.abs(a[fin] - b[fin]) / np.abs(b[fin])), " run-to-run bitwise:", np.array_equal(a, a2), " sub-batch bitwise:",
      np.array_equal(c, a[:7]), " single bitwise:", np.array_equal(one, a[5:6]))
res = {}
for w in ("1", "2", "4"):
    _lib.hooks["VAG_FIT_WAVES_PER_BLOCK"] = w
    res[w] = fit.loglike_batch(theta, defs)
_lib.hooks.pop("VAG_FIT_WAVES_PER_BLOCK", None)
print("wavefronts per block 1 / 2 / 4 bitwise equal:", np.array_equal(res["1"], res["2"]) and np.array_equal(res["1"], res["4"]),
      " and equal to the default choice:", np.array_equal(res["1"], a))
