"""Developer probe: the fast ODE kernel against the general one (VAG_DYN_GENERAL=1) on random C4 walkers and C2 / C1 models."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _abi, configs, bench
from vegasafterglow_amd import _lib
import vegasafterglow_amd as va
lib = _lib.load(); h, lock = va.get_context(0); dp = C.POINTER(C.c_double)
def grid(prms, t, nu):
    n = len(prms)
    arr = (_lib.ModelParams * n)(*[_lib.ModelParams.from_buffer_copy(bytes(p)) for p in prms])
    out = np.empty((n, nu.size, t.size))
    _lib.check(lib.vag_flux_density_grid_batch(h, arr, n, t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp), nu.size, out.ctypes.data_as(dp)))
    return out
rng = np.random.default_rng(5)
for name, n in (("C4 box", 256), ("C4 box", 16), ("C4 box", 1)):
    prms = []
    for _ in range(n):
        kw = dict(configs.C4_TRUTH, jet="GaussianJet")
        kw.update(E_iso=10 ** rng.uniform(50, 54), Gamma0=10 ** rng.uniform(1.5, 3), theta_c=rng.uniform(0.02, 0.3), theta_obs=rng.uniform(0, 0.8),
                  n_ism=10 ** rng.uniform(-4, 1), p=rng.uniform(2.05, 2.8), eps_e=10 ** rng.uniform(-3, -0.5), eps_B=10 ** rng.uniform(-5, -1))
        prms.append(_abi.make_params(**kw))
    t, nu = np.logspace(4.5, 8, 40), np.array([3e9, 5.06e14, 2.41e17])
    a = grid(prms, t, nu)
    _lib.hooks["VAG_DYN_GENERAL"] = "1"
    b = grid(prms, t, nu)
    _lib.hooks.pop("VAG_DYN_GENERAL")
    sel = a > 1e-3 * a.max(axis=(1, 2), keepdims=True)
    print(name, n, "max rel diff fast vs general", float(np.max(np.abs(a - b)[sel] / a[sel])), "finite", bool(np.isfinite(a).all() and np.isfinite(b).all()))
