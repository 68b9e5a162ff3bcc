"""The lazily-built-SSC-table property (tests/test_gpu_parity.py::test_lazily_built_ssc_tables_are_the_bits_of_every_table_on_random_
narrow_windows) on model / window draws no test has seen: lazy tables == every table, bit for bit, and how often the lazy selection
had a hole (vag_plan.n_ssc_all_cell_fallbacks: the pass is then repeated with every table -- correct, but a hole worth knowing).
usage: python profiles/debug/lazy_tables_seeds.py <seed> [<seed> ...]"""
import ctypes as C, os, sys
import numpy as np
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(_ROOT, "tests")); sys.path.insert(0, _ROOT)
import sweeps
from vegasafterglow_amd import _lib
import vegasafterglow_amd as va

lib = _lib.load(); h, lock = va.get_context(0); dp = C.POINTER(C.c_double)
nu = sweeps.WINDOW_NU
for seed in [int(a) for a in sys.argv[1:]] or [1]:
    prms, tags = sweeps.ssc_window_models(48, seed=seed)
    n = len(prms)
    arr = (_lib.ModelParams * n)(*[_lib.ModelParams.from_buffer_copy(bytes(p)) for p in prms])

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
        pl = _lib.Plan()
        lib.vag_last_plan(h, C.byref(pl))
        return comps, pl
    fallbacks = mismatches = 0
    lazy_bytes = all_bytes = 0
    for w, t in enumerate(sweeps.narrow_windows(24, seed=seed + 1000)):
        series = w % 3 == 2
        got, pl = run(t, series)
        fallbacks += pl.n_ssc_all_cell_fallbacks
        lazy_bytes += pl.ic_pool_bytes
        _lib.hooks["VAG_IC_ALL_CELLS"] = "1"
        try:
            want, pl = run(t, series)
        finally:
            _lib.hooks.pop("VAG_IC_ALL_CELLS")
        all_bytes += pl.ic_pool_bytes
        for c, (g, x) in enumerate(zip(got, want)):
            bad = [i for i in range(n) if not np.array_equal(g[i], x[i], equal_nan=True)]
            if bad:
                mismatches += len(bad)
                print("  seed", seed, "window", w, "component", c, "differs:", [tags[i] for i in bad[:4]], flush=True)
    print(f"seed {seed}: 24 windows x {n} models: mismatches {mismatches}, all-cell fallbacks {fallbacks}, lazy / all table bytes {lazy_bytes / all_bytes:.2f}", flush=True)
