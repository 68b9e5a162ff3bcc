"""Ad-hoc GPU parity probe (not a pytest file): python profiles/debug/gpu_debug.py"""
import json, os, sys, time
import numpy as np
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(_ROOT, "tests"))
sys.path.insert(0, _ROOT)
import _abi, configs
import ctypes as C
from vegasafterglow_amd import _lib
from vegasafterglow_amd.model import get_context

lib = _lib.load()
orc = _abi.load_oracle()
h, _ = get_context(0)
dp = C.POINTER(C.c_double)

def gpu_grid(prm, t, nu):
    t = np.ascontiguousarray(t, dtype=np.float64); nu = np.ascontiguousarray(nu, dtype=np.float64)
    out = np.zeros((nu.size, t.size))
    q = _lib.ModelParams.from_buffer_copy(bytes(prm))
    rc = lib.vag_flux_density_grid_batch(h, C.byref(q), 1, t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp), nu.size, out.ctypes.data_as(dp))
    if rc: raise RuntimeError(lib.vag_last_error().decode())
    st = _lib.StageTimes(); lib.vag_last_stage_times(h, C.byref(st))
    return out, st

def rel(a, b):
    m = b > 1e-12 * b.max()
    return (np.abs(a - b) / np.where(m, b, 1))[m].max()

cases = {"C1a": (configs.C1A, configs.C1_T, configs.C1_NU), "C1b": (configs.C1B, configs.C1_T, configs.C1_NU),
         "C2": (configs.C2, configs.C2_T, configs.C2_NU)}
cases.update({k: v for k, v in configs.EXTRA.items()})
for name, (kw, t, nu) in cases.items():
    prm = _abi.make_params(**kw)
    t0 = time.time(); O = orc.flux_density_grid(prm, t, nu); to = time.time() - t0
    t0 = time.time(); G, st = gpu_grid(prm, t, nu); tg = time.time() - t0
    print(f"{name:20s} rel={rel(G, O):.3e} nan={np.isnan(G).sum()} oracle {to*1e3:8.1f} ms  gpu wall {tg*1e3:8.1f} ms  "
          f"[grid {st.grid_ms:.3f} dyn {st.dynamics_ms:.3f} cells {st.cells_ms:.3f} flux {st.flux_ms:.3f} red {st.reduce_ms:.3f} tot {st.total_ms:.3f}]", flush=True)
for name in ["tophat_ism", "tophat_ism_adiabatic", "two_component_ism"]:
    g = np.load(os.path.join(_abi.ROOT, "tests", "golden", name + ".npz"))
    prm = _abi.params_from_golden_config(json.loads(str(g["config"])))
    G, st = gpu_grid(prm, g["t"], g["nus"])
    T = g["total"]
    ok = np.all(np.abs(G - T) <= 2e-3 * np.abs(T) + 1e-2 * np.abs(T).max())
    print(f"golden {name:24s} contract={ok} rel(bright)={rel(np.where(T>1e-2*T.max(),G,T), T):.3e}")
