"""Developer probe: section cycle counts of vag_grid_kernel (library built with -DVAG_GRID_STAMPS)."""
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
cases = {"C1a": (dict(theta_obs=0.0, resolutions=(0.089, 0.05, 12.0)), np.logspace(2, 8, 100), np.array([1e9, 4.84e14, 1e18])),
         "C1b": (dict(theta_obs=0.05, resolutions=(0.089, 0.05, 12.0)), np.logspace(2, 8, 100), np.array([1e9, 4.84e14, 1e18])),
         "C2": (dict(jet="GaussianJet", theta_obs=0.3, resolutions=(0.355, 0.31, 20.5)), configs.C2_T, configs.C2_NU),
         "C4": (dict(configs.C4_TRUTH, jet="GaussianJet"), configs.C4_EPOCHS, configs.C4_BANDS)}
for name, (kw, t, nu) in cases.items():
    prm = _abi.make_params(**kw)
    arr = (_lib.ModelParams * 1)(_lib.ModelParams.from_buffer_copy(bytes(prm)))
    out = np.empty((1, nu.size, t.size))
    print(name, flush=True)
    for rep in range(2):
        _lib.check(lib.vag_flux_density_grid_batch(h, arr, 1, t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp), nu.size, out.ctypes.data_as(dp)))
    lib.vag_ctx_synchronize(h)
