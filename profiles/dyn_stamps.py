"""Developer probe: run a variant built with -DVAG_DYN_STAMPS on one C1a model and one C4 walker (prints cycle stamps)."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import _abi, configs
from vegasafterglow_amd import _lib
import vegasafterglow_amd as va
lib = _lib.load()
h, lock = va.get_context(0)
dp = C.POINTER(C.c_double)
t, nu = configs.C1_T, configs.C1_NU
for name, kw in (("C1a", configs.C1A), ("C4truth", dict(configs.C4_TRUTH))):
    print("==", name, flush=True)
    prm = _abi.make_params(**kw)
    arr = (_lib.ModelParams * 1)(_lib.ModelParams.from_buffer_copy(bytes(prm)))
    out = np.empty((1, 3, 100))
    for rep in range(2):
        _lib.check(lib.vag_flux_density_grid_batch(h, arr, 1, t.ctypes.data_as(dp), 100, nu.ctypes.data_as(dp), 3, out.ctypes.data_as(dp)))
