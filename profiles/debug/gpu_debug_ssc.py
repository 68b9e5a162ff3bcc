"""Ad-hoc GPU parity probe for the SSC tier (not a pytest file): python profiles/debug/gpu_debug_ssc.py"""
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


def gpu_comp(prm, t, nu):
    t = np.ascontiguousarray(t, dtype=np.float64); nu = np.ascontiguousarray(nu, dtype=np.float64)
    s = np.zeros((nu.size, t.size)); c = np.zeros((nu.size, t.size))
    q = _lib.ModelParams.from_buffer_copy(bytes(prm))
    rc = lib.vag_flux_density_grid_components_batch(h, C.byref(q), 1, t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp),
                                                    nu.size, s.ctypes.data_as(dp), c.ctypes.data_as(dp))
    if rc: raise RuntimeError(lib.vag_last_error().decode())
    return s, c


def rel(a, b):
    m = b > 1e-12 * b.max()
    if not m.any(): return 0.0
    return (np.abs(a - b) / np.where(m, b, 1))[m].max()


cases = {}
for name in ["gauss_wind_ssc", "dense_ism_ssa_ssc", "ism_absorbed_slow_ssc"]:
    g = np.load(os.path.join(_abi.ROOT, "tests", "golden", name + ".npz"))
    cases[name] = (_abi.params_from_golden_config(json.loads(str(g["config"]))), g["t"], g["nus"])
t = np.logspace(2, 7, 24)
nu = np.array([1e9, 1e14, 1e17, 1e20, 1e23])
cases["gauss_ssc"] = (_abi.make_params(jet="GaussianJet", theta_c=0.1, E_iso=1e52, Gamma0=300, n_ism=1.0, theta_obs=0.2, ssc=True), t, nu)
cases["gauss_ssc_kn"] = (_abi.make_params(jet="GaussianJet", theta_c=0.1, E_iso=1e52, Gamma0=300, n_ism=1.0, theta_obs=0.2, ssc=True, kn=True), t, nu)
cases["tophat_ssc_kn"] = (_abi.make_params(theta_c=0.1, E_iso=1e53, Gamma0=300, n_ism=0.1, theta_obs=0.0, eps_B=1e-4, ssc=True, kn=True), t, nu)
for name, (prm, t, nu) in cases.items():
    t0 = time.time(); Os, Oc = orc.flux_components(prm, t, nu); to = time.time() - t0
    try:
        t0 = time.time(); Gs, Gc = gpu_comp(prm, t, nu); tg = time.time() - t0
    except RuntimeError as e:
        print(f"{name:24s} ERROR {e}"); continue
    print(f"{name:24s} sync rel={rel(Gs, Os):.3e} ssc rel={rel(Gc, Oc):.3e} nan={np.isnan(Gs).sum()+np.isnan(Gc).sum()} "
          f"oracle {to*1e3:8.1f} ms gpu {tg*1e3:8.1f} ms", flush=True)
    if rel(Gc, Oc) > 1e-5 or rel(Gs, Os) > 1e-5:
        np.set_printoptions(linewidth=200, precision=4)
        print(" sync ratio:\n", Gs / Os)
        print(" ssc ratio:\n", Gc / np.where(Oc > 0, Oc, 1))
