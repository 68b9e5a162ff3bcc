"""Property sweep of the lazily built SSC tables: random SSC models x random NARROW request windows (1 ... 12 times inside 0.05 ... 2 decades
anywhere between 30 s and 3e8 s) -- the requests that leave most cells without a table -- evaluated twice, with the lazy tables and with
every table (VAG_IC_ALL_CELLS=1): the four components must be the same bits and no call may meet a cell without a table.  No checker
involved, so hundreds of draws take seconds.  usage: python profiles/debug/prior_sweep_window.py [n_models] [n_windows]"""
import os, sys
import numpy as np
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(_ROOT, "tests")); sys.path.insert(0, _ROOT)
import ctypes as C
import _abi
from vegasafterglow_amd import _lib
import vegasafterglow_amd as va

n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 40
lib = _lib.load(); h, lock = va.get_context(0); dp = C.POINTER(C.c_double)
import sweeps
seed = int(os.environ.get('SWEEP_SEED', '31337'))
prms, tags = sweeps.ssc_window_models(n, seed)
arr = (_lib.ModelParams * n)(*[_lib.ModelParams.from_buffer_copy(bytes(p)) for p in prms])
nu = sweeps.WINDOW_NU
windows = sweeps.narrow_windows(nw, seed + 1)

def run(t, series):
    if series:
        tt, nn = np.repeat(t, nu.size), np.tile(nu, t.size)
        comps = [np.empty((n, tt.size)) for _ in range(4)]
        out4 = (dp * 4)(*[a.ctypes.data_as(dp) for a in comps])
        _lib.check(lib.vag_flux_density_components4_batch(h, arr, n, tt.ctypes.data_as(dp), nn.ctypes.data_as(dp), tt.size, out4))
    else:
        comps = [np.empty((n, nu.size, t.size)) for _ in range(4)]
        out4 = (dp * 4)(*[a.ctypes.data_as(dp) for a in comps])
        _lib.check(lib.vag_flux_density_grid_components4_batch(h, arr, n, t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp), nu.size, out4))
    return comps

bad = 0
pool_lazy = pool_all = 0
for w in range(nw):
    t = windows[w]; k = t.size
    series = bool(w % 3 == 2)
    try:
        got = run(t, series)
        pl = _lib.Plan(); lib.vag_last_plan(h, C.byref(pl)); pool_lazy += pl.ic_pool_bytes
        _lib.hooks["VAG_IC_ALL_CELLS"] = "1"
        want = run(t, series)
        lib.vag_last_plan(h, C.byref(pl)); pool_all += pl.ic_pool_bytes
    except RuntimeError as e:
        print(f"window {w} [{t.min():.3g}, {t.max():.3g}] x{k} {'series' if series else 'grid'}: FAILED {str(e)[-100:]}", flush=True)
        bad += 1
        continue
    finally:
        _lib.hooks.pop("VAG_IC_ALL_CELLS", None)
    for c, (g, x) in enumerate(zip(got, want)):
        same = np.array_equal(g, x, equal_nan=True)
        if not same:
            mism = [i for i in range(n) if not np.array_equal(g[i], x[i], equal_nan=True)]
            print(f"window {w} [{t.min():.3g}, {t.max():.3g}] x{k} {'series' if series else 'grid'} component {c}: differs on models {mism[:8]} ({[tags[i] for i in mism[:4]]})", flush=True)
            bad += 1
print(f"{n} models x {nw} windows: {bad} failures; SSC table pool (last shock built) lazy / all tables: {pool_lazy / max(pool_all, 1):.3f}")
