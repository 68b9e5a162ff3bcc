"""Rows per wavefront of the general dynamics kernel (spreading jets; VAG_DYN_RPW): dynamics stage of one model and of 256 models."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import os, sys, numpy as np, ctypes as C
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import _abi
from vegasafterglow_amd import _lib
from vegasafterglow_amd.model import get_context
lib = _lib.load(); h, _ = get_context(0); dp = C.POINTER(C.c_double)
t = np.logspace(2, 8, 100); nu = np.array([1e9, 4.84e14, 1e18])
rng = np.random.default_rng(5)
for nb in (1, 256):
    prms = [_abi.make_params(jet="GaussianJet", theta_obs=0.3, spreading=True, E_iso=1e52 * (1 + 0.1 * rng.random())) for _ in range(nb)]
    arr = (_lib.ModelParams * nb)(*[_lib.ModelParams.from_buffer_copy(bytes(p)) for p in prms])
    out = np.empty((nb, nu.size, t.size))
    best = None
    for r in range(4):
        lib.vag_flux_density_grid_batch(h, arr, nb, t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp), nu.size, out.ctypes.data_as(dp))
        st = _lib.StageTimes(); lib.vag_last_stage_times(h, C.byref(st))
        best = st.dynamics_ms if best is None else min(best, st.dynamics_ms)
        pl = _lib.Plan(); lib.vag_last_plan(h, C.byref(pl))
    print("  nb %%4d (%%d rows): dynamics %%.3f ms (best of 4), call %%.2f ms, checksum %%.17g" %% (nb, pl.n_rows, best, st.total_ms, float(out.sum())))
''' % (ROOT, ROOT)
for rpw in ("default", "64", "32", "16", "8"):
    print("VAG_DYN_RPW=" + rpw, flush=True)
    env = dict(os.environ)
    if rpw != "default":
        env["VAG_DYN_RPW"] = rpw
    subprocess.run([sys.executable, "-c", code], env=env)
