"""Where one model's call goes (the reference's stage names, vag_ctx_profile): configs[1], [2], [4] and a top hat, ONE model per call."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "profiles"))
import _abi, configs
from vegasafterglow_amd import _lib
from vegasafterglow_amd.model import get_context
from ssc_ensemble import c3_batch, c5_batch
lib = _lib.load(); h, _ = get_context(0); dp = C.POINTER(C.c_double)
te, nue = np.logspace(2, 8, 100), np.array([1e9, 4.84e14, 1e18, 2.4e26])
cases = {"C2": (_abi.make_params(**configs.C2), configs.C2_T, configs.C2_NU), "C3": (c3_batch(1)[0], te, nue), "C5": (c5_batch(1)[0], te, nue),
         "C1a": (_abi.make_params(**configs.C1A), configs.C1_T, configs.C1_NU)}
for name, (prm, t, nu) in cases.items():
    arr = (_lib.ModelParams * 1)(_lib.ModelParams.from_buffer_copy(bytes(prm)))
    out = np.empty((1, nu.size, t.size))
    call = lambda: _lib.check(lib.vag_flux_density_grid_batch(h, arr, 1, t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp), nu.size, out.ctypes.data_as(dp)))
    for _ in range(3):
        call()
    st = _lib.StageTimes(); lib.vag_last_stage_times(h, C.byref(st))
    _lib.check(lib.vag_ctx_profile(h, 1)); call()
    prof = _lib.Profile(); _lib.check(lib.vag_last_profile(h, C.byref(prof))); _lib.check(lib.vag_ctx_profile(h, 0))
    print(name, "stage ms:", {f: round(getattr(st, f), 3) for f, _ in _lib.StageTimes._fields_}, "| reference names:",
          {n: round(getattr(prof, n), 3) for n, _ in _lib.Profile._fields_}, flush=True)
