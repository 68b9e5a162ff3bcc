"""Developer check: the row-per-lane grid kernel against the row-per-workgroup kernel on the C5 / C3 ensembles (agreement per
component through the total), and timing of both."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "profiles"))
import _abi  # noqa: E402,F401
from ssc_ensemble import c3_batch, c5_batch  # noqa: E402
from vegasafterglow_amd import _lib  # noqa: E402
from vegasafterglow_amd.model import get_context  # noqa: E402

lib = _lib.load()
h, _ = get_context(0)
dp = C.POINTER(C.c_double)
t, nu = np.logspace(2, 8, 100), np.array([1e9, 4.84e14, 1e18, 2.4e26])
for name, prms in (("C5", c5_batch(256)), ("C3", c3_batch(128))):
    nb = len(prms)
    arr = (_lib.ModelParams * nb)(*[_lib.ModelParams.from_buffer_copy(bytes(p)) for p in prms])
    res = {}
    for mode in ("rows", "workgroup"):
        if mode == "workgroup":
            _lib.hooks["VAG_GRID_ROW_PER_WORKGROUP"] = "1"
        else:
            _lib.hooks.pop("VAG_GRID_ROW_PER_WORKGROUP", None)
        out = np.empty((nb, nu.size, t.size))
        for rep in range(3):
            t0 = time.time()
            _lib.check(lib.vag_flux_density_grid_batch(h, arr, nb, t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp), nu.size,
                                                       out.ctypes.data_as(dp)))
            dt = time.time() - t0
        st = _lib.StageTimes()
        lib.vag_last_stage_times(h, C.byref(st))
        res[mode] = (out, dt, st.flux_ms)
    _lib.hooks.pop("VAG_GRID_ROW_PER_WORKGROUP", None)
    a, b = res["rows"][0], res["workgroup"][0]
    m = b > 1e-12 * b.max(axis=(1, 2), keepdims=True)
    print(name, "max rel diff", np.max(np.abs(a - b)[m] / b[m]), "finite", np.isfinite(a).all(),
          " ms per batch rows / workgroup: %.1f / %.1f  (flux stage %.1f / %.1f)" % (1e3 * res["rows"][1], 1e3 * res["workgroup"][1],
                                                                                    res["rows"][2], res["workgroup"][2]))
