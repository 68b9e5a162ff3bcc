"""Developer stress check: the row-per-lane kernels against the row-per-wavefront / workgroup kernels on many random models
(C4 prior box with several seeds; C5 / C3 draws with other seeds): finite sets, worst relative difference."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "profiles"))
import torch  # noqa: F401,E402

import bench  # noqa: E402
from ssc_ensemble import c3_batch, c5_batch  # noqa: E402
from vegasafterglow_amd import _lib  # noqa: E402

lib = _lib.load()
h = C.c_void_p()
_lib.check(lib.vag_ctx_create(0, C.byref(h)))
dp = C.POINTER(C.c_double)
fit, defs, _ = bench.c4_fitter(lib, h, _lib)
_, lo, hi = fit.build_spec(defs)
worst = 0.0
for seed in range(4):
    theta = lo + (hi - lo) * np.random.default_rng(100 + seed).random((2048, len(defs)))
    _lib.hooks.pop("VAG_SERIES_ROW_PER_WAVE", None)
    a = fit.loglike_batch(theta, defs)
    _lib.hooks["VAG_SERIES_ROW_PER_WAVE"] = "1"
    b = fit.loglike_batch(theta, defs)
    _lib.hooks.pop("VAG_SERIES_ROW_PER_WAVE", None)
    assert np.array_equal(np.isfinite(a), np.isfinite(b)), seed
    f = np.isfinite(b)
    worst = max(worst, float(np.max(np.abs(a[f] - b[f]) / np.maximum(np.abs(b[f]), 1e-300))))
    print("C4 seed", seed, "finite", int(f.sum()), "worst so far %.2e" % worst, flush=True)
t, nu = np.logspace(2, 8, 100), np.array([1e9, 4.84e14, 1e18, 2.4e26])
for name, prms in (("C5", c5_batch(64, seed=7)), ("C5b", c5_batch(64, seed=8)), ("C3", c3_batch(96, seed=9))):
    nb = len(prms)
    arr = (_lib.ModelParams * nb)(*[_lib.ModelParams.from_buffer_copy(bytes(p)) for p in prms])
    res = []
    for mode in (None, "1"):
        if mode:
            _lib.hooks["VAG_GRID_ROW_PER_WORKGROUP"] = mode
        out = np.empty((nb, nu.size, t.size))
        _lib.check(lib.vag_flux_density_grid_batch(h, arr, nb, t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp), nu.size, out.ctypes.data_as(dp)))
        _lib.hooks.pop("VAG_GRID_ROW_PER_WORKGROUP", None)
        res.append(out)
    a, b = res
    assert np.all(np.isfinite(a)) and np.all(np.isfinite(b))
    m = b > 1e-12 * b.max(axis=(1, 2), keepdims=True)
    print(name, "worst rel diff %.2e" % np.max(np.abs(a - b)[m] / b[m]), flush=True)
