"""Ad-hoc GPU parity probe for spreading jets (not a pytest file)."""
import os, sys, time
import numpy as np
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(_ROOT, "tests"))
sys.path.insert(0, _ROOT)
import _abi, configs
import ctypes as C
from vegasafterglow_amd import _lib
from vegasafterglow_amd.model import get_context
lib = _lib.load(); orc = _abi.load_oracle(); h, _ = get_context(0); dp = C.POINTER(C.c_double)
def gpu_comp4(prm, t, nu):
    comps = [np.zeros((nu.size, t.size)) for _ in range(4)]
    arr = (dp * 4)(*[a.ctypes.data_as(dp) for a in comps])
    q = _lib.ModelParams.from_buffer_copy(bytes(prm))
    rc = lib.vag_flux_density_grid_components4_batch(h, C.byref(q), 1, t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp), nu.size, arr)
    if rc: raise RuntimeError(lib.vag_last_error().decode())
    return comps
def rel(a, b):
    m = b > 1e-12 * b.max()
    return (np.abs(a - b) / np.where(m, b, 1))[m].max() if m.any() else 0.0
for name, kw in configs.SPREAD_CASES.items():
    prm = _abi.make_params(**kw); t, nu = configs.SPREAD_T, configs.SPREAD_NU
    t0 = time.time(); O = orc.flux_components4(prm, t, nu); to = time.time() - t0
    t0 = time.time(); G = gpu_comp4(prm, t, nu); tg = time.time() - t0
    print(f"{name:24s} " + " ".join(f"{rel(g, o):.2e}" for g, o in zip(G, O)) + f" oracle {to*1e3:7.1f} ms gpu {tg*1e3:7.1f} ms", flush=True)
