"""Developer probe: stage times of ONE C1a light curve (single-call latency of Model.flux_density_grid)."""
import ctypes as C, sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import _abi, configs
from vegasafterglow_amd import _lib
import vegasafterglow_amd as va

lib = _lib.load()
h, lock = va.get_context(0)
t, nu = np.logspace(2, 8, 100), np.array([1e9, 4.84e14, 1e18])
for theta_obs in (0.0, 0.05):
    prm = _abi.make_params(theta_obs=theta_obs, resolutions=(0.089, 0.05, 12.0))
    arr = (_lib.ModelParams * 1)(_lib.ModelParams.from_buffer_copy(bytes(prm)))
    out = np.empty((1, 3, 100))
    dp = C.POINTER(C.c_double)
    for rep in range(5):
        t0 = time.perf_counter()
        _lib.check(lib.vag_flux_density_grid_batch(h, arr, 1, t.ctypes.data_as(dp), 100, nu.ctypes.data_as(dp), 3, out.ctypes.data_as(dp)))
        wall = (time.perf_counter() - t0) * 1e3
        st = _lib.StageTimes()
        lib.vag_last_stage_times(h, C.byref(st))
    print(f"theta_obs={theta_obs}: wall {wall:.3f} ms  grid {st.grid_ms:.3f} dyn {st.dynamics_ms:.3f} cells {st.cells_ms:.3f} "
          f"flux {st.flux_ms:.3f} reduce {st.reduce_ms:.3f} device total {st.total_ms:.3f}")
