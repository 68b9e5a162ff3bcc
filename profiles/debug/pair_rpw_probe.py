"""Rows per wavefront of vag_dynamics_pair_kernel (VAG_PAIR_RPW): dynamics stage time of one configs[2] model and of the 512-model batch.
usage: python profiles/debug/pair_rpw_probe.py"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import os, sys, numpy as np, ctypes as C
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests")); sys.path.insert(0, os.path.join(%r, "profiles"))
import _abi
from vegasafterglow_amd import _lib
from vegasafterglow_amd.model import get_context
from ssc_ensemble import c3_batch
lib = _lib.load(); h, _ = get_context(0); dp = C.POINTER(C.c_double)
t = np.logspace(2, 8, 100); nu = np.array([1e9, 4.84e14, 1e18, 2.4e26])
for nb in (1, 512):
    prms = c3_batch(nb)
    arr = (_lib.ModelParams * nb)(*[_lib.ModelParams.from_buffer_copy(bytes(p)) for p in prms])
    out = np.empty((nb, nu.size, t.size))
    best = None
    for r in range(4):
        lib.vag_flux_density_grid_batch(h, arr, nb, t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp), nu.size, out.ctypes.data_as(dp))
        st = _lib.StageTimes(); lib.vag_last_stage_times(h, C.byref(st))
        best = st.dynamics_ms if best is None else min(best, st.dynamics_ms)
        tot = st.total_ms
    print("  nb %%4d: dynamics %%.3f ms (best of 4), call %%.2f ms, checksum %%.17g" %% (nb, best, tot, float(out.sum())))
''' % (ROOT, ROOT, ROOT)
for rpw in ("64", "32", "16", "8"):
    print("VAG_PAIR_RPW=" + rpw, flush=True)
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, VAG_PAIR_RPW=rpw))
