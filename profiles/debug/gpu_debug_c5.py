"""Ad-hoc: parity of individual C5 ensemble members, per component and with SSC off (python profiles/debug/gpu_debug_c5.py)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "profiles"))
import ctypes as C
import _abi
from ssc_ensemble import c5_batch
from vegasafterglow_amd import _lib
from vegasafterglow_amd.model import get_context
lib = _lib.load(); h, _ = get_context(0); dp = C.POINTER(C.c_double)
orc = _abi.load_oracle()
t = np.logspace(2, 8, 100); nu = np.array([1e9, 4.84e14, 1e18, 2.4e26])
prms = c5_batch(256)

def comp(p):
    s = np.zeros((nu.size, t.size)); c = np.zeros((nu.size, t.size))
    q = _lib.ModelParams.from_buffer_copy(bytes(p))
    rc = lib.vag_flux_density_grid_components_batch(h, C.byref(q), 1, t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp), nu.size, s.ctypes.data_as(dp), c.ctypes.data_as(dp))
    if rc: raise RuntimeError(lib.vag_last_error().decode())
    return s, c

def rel(a, b):
    m = b > 1e-12 * b.max()
    e = np.abs(a - b) / np.where(m, b, 1) * m
    i = np.unravel_index(np.argmax(e), e.shape)
    return e.max(), i

idx = [int(a) for a in sys.argv[1:]] or list(range(0, 256, 8))
for i in idx:
    s, c = comp(prms[i]); os_, oc = orc.flux_components(prms[i], t, nu)
    p0 = _abi.ModelParams.from_buffer_copy(bytes(prms[i])); p0.flags = 0
    s0, _ = comp(p0); o0 = orc.flux_density_grid(p0, t, nu)
    print(i, "sync", rel(s, os_), "ssc", rel(c, oc), "nossc", rel(s0, o0), "theta_c %.4f theta_w %.4f" % (prms[i].theta_c, prms[i].theta_w), flush=True)
