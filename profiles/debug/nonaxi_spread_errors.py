"""Developer aid (GPU box): per-component agreement of Model(axisymmetric=False) + spreading jets with the reference vectors."""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
sys.path.insert(0, "tests")
import _abi
import vegasafterglow_amd as va
from vegasafterglow_amd import _lib

lib = _lib.load()
h, _ = va.get_context(0)
dp = C.POINTER(C.c_double)
fx = np.load("tests/golden/reference_nonaxi_spread.npz")
meta = json.loads(str(fx["meta"]))
t, nu = fx["t"], fx["nu"]
for case, mt in meta.items():
    kw = dict(mt["kw"])
    if "resolutions" in kw:
        kw["resolutions"] = tuple(kw["resolutions"])
    prm = _abi.make_params(**kw)
    arr = (_lib.ModelParams * 1)(_lib.ModelParams.from_buffer_copy(bytes(prm)))
    comps = [np.empty((1, nu.size, t.size)) for _ in range(4)]
    out4 = (dp * 4)(*[a.ctypes.data_as(dp) for a in comps])
    _lib.check(lib.vag_flux_density_grid_components4_batch(h, arr, 1, t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp), nu.size, out4))

    def err(g, w):
        m = w > 1e-9 * w.max()
        return float(np.max(np.abs(g - w)[m] / w[m])) if m.any() else 0.0
    print(case, " ".join(f"{n} {err(c[0], fx[case + '__' + n]):.2e}" for c, n in zip(comps, ("sync", "ssc", "rvs_sync", "rvs_ssc"))))
    # the same model with axisymmetric=True against the oracle restatement gives the scale of the coupled solver's own agreement
