"""Developer probe: SSC light curves at the shipped default resolution (small grids), batch 1024."""
import ctypes as C, sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import _abi
from vegasafterglow_amd import _lib
import vegasafterglow_amd as va
lib = _lib.load()
h, lock = va.get_context(0)
dp = C.POINTER(C.c_double)
nb = 1024
rng = np.random.default_rng(3)
prms = [_abi.make_params(jet="GaussianJet", theta_obs=float(rng.uniform(0.05, 0.3)), E_iso=10 ** rng.uniform(51, 53), ssc=True) for _ in range(nb)]
arr = (_lib.ModelParams * nb)(*[_lib.ModelParams.from_buffer_copy(bytes(p)) for p in prms])
t, nu = np.logspace(2, 8, 100), np.array([1e9, 4.84e14, 1e18, 2.4e26])
out = np.empty((nb, nu.size, t.size))
for rep in range(4):
    t0 = time.perf_counter()
    _lib.check(lib.vag_flux_density_grid_batch(h, arr, nb, t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp), nu.size, out.ctypes.data_as(dp)))
    dt = time.perf_counter() - t0
print(f"{1e3 * dt:.2f} ms per {nb} models -> {nb / dt:.0f} LC/s, finite={np.isfinite(out).all()}")
