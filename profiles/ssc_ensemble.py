"""C5 / C3 workload probes, timed through the C-ABI.
  C5 (default): ensemble of two-component SSC models;  C3 (env ENSEMBLE=c3): power-law jet in a wind, forward +
  reverse shock, SSC + Klein-Nishina on both, with +-10 % jitter on the jet and microphysics parameters.
usage: python3 profiles/ssc_ensemble.py [nb] [reps] [check]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import _abi
from vegasafterglow_amd import _lib
from vegasafterglow_amd.model import get_context


from configs import c3_batch, c5_batch  # the workloads live in tests/configs.py (round 6)


if __name__ == "__main__":
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    lib = _lib.load(); h, _ = get_context(0); dp = C.POINTER(C.c_double)
    which = os.environ.get("ENSEMBLE", "c5")
    prms = c3_batch(nb) if which == "c3" else c5_batch(nb)
    if os.environ.get("C5_NOSSC"):
        for q in prms: q.flags = 0
    arr = (_lib.ModelParams * nb)(*[_lib.ModelParams.from_buffer_copy(bytes(p)) for p in prms])
    t = np.logspace(2, 8, 100); nu = np.array([1e9, 4.84e14, 1e18, 2.4e26])
    out = np.empty((nb, nu.size, t.size))
    for r in range(reps + 1):
        t0 = time.time()
        rc = lib.vag_flux_density_grid_batch(h, arr, nb, t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp), nu.size, out.ctypes.data_as(dp))
        dt = time.time() - t0
        if rc: raise RuntimeError(lib.vag_last_error().decode())
        st = _lib.StageTimes(); lib.vag_last_stage_times(h, C.byref(st))
        pl = _lib.Plan(); lib.vag_last_plan(h, C.byref(pl))
        if r == 0: print({f: getattr(pl, f) for f, _ in _lib.Plan._fields_})
        print(f"rep {r}: {dt*1e3:.1f} ms wall -> {nb/dt:.1f} LC/s  [grid {st.grid_ms:.2f} dyn {st.dynamics_ms:.2f} cells+cool {st.cells_ms:.2f} "
              f"flux(sync+ic+ssc) {st.flux_ms:.2f} red {st.reduce_ms:.2f} tot {st.total_ms:.2f}] nan={np.isnan(out).sum()}", flush=True)
    if len(sys.argv) > 3:  # parity of a few members against the CPU checker
        orc = _abi.load_oracle()
        for i in range(0, nb, max(1, nb // 4)):
            w = orc.flux_density_grid(prms[i], t, nu)
            m = w > 1e-12 * w.max()
            print(f" member {i}: rel {np.abs(out[i] - w)[m].max() if False else (np.abs(out[i]-w)/np.where(m,w,1))[m].max():.3e}")
