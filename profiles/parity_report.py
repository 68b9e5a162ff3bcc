"""Achieved GPU-vs-oracle parity per configuration (max relative error over bins above 1e-2 of the peak and over
all bins above 1e-12 of the peak) + the reference golden contract.  Run on the GPU box:
    python profiles/parity_report.py > gpurun_out/parity_report.txt"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _abi  # noqa: E402
import configs  # noqa: E402
from vegasafterglow_amd import _lib  # noqa: E402
from vegasafterglow_amd.model import get_context  # noqa: E402

lib = _lib.load()
orc = _abi.load_oracle()
h, _ = get_context(0)
dp = C.POINTER(C.c_double)


def gpu_grid(prm, t, nu):
    t = np.ascontiguousarray(t, dtype=np.float64)
    nu = np.ascontiguousarray(nu, dtype=np.float64)
    out = np.zeros((nu.size, t.size))
    q = _lib.ModelParams.from_buffer_copy(bytes(prm))
    _lib.check(lib.vag_flux_density_grid_batch(h, C.byref(q), 1, t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp),
                                               nu.size, out.ctypes.data_as(dp)))
    return out


def rel(a, b, floor):
    m = b > floor * b.max()
    return float((np.abs(a - b) / np.where(m, b, 1))[m].max())


cases = {"C1a": (configs.C1A, configs.C1_T, configs.C1_NU), "C1b": (configs.C1B, configs.C1_T, configs.C1_NU),
         "C2": (configs.C2, configs.C2_T, configs.C2_NU), "C4 truth": (configs.C4_TRUTH, configs.C4_EPOCHS, configs.C4_BANDS)}
cases.update(configs.EXTRA)
print(f"{'config':22s} {'grid (phi,theta,t)':>20s} {'max rel err >1e-2 peak':>24s} {'max rel err >1e-12 peak':>26s}")
for name, (kw, t, nu) in cases.items():
    prm = _abi.make_params(**kw)
    O, G = orc.flux_density_grid(prm, t, nu), gpu_grid(prm, t, nu)
    s = orc.details(prm, t.min(), t.max())["shape"]
    print(f"{name:22s} {str((s['n_phi'], s['n_theta'], s['n_t'])):>20s} {rel(G, O, 1e-2):24.3e} {rel(G, O, 1e-12):26.3e}")
print("\nreference golden baselines (contract: |d| <= 2e-3 |ref| + 1e-2 max|ref|)")
for name in ("tophat_ism", "tophat_ism_adiabatic", "two_component_ism"):
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    prm = _abi.params_from_golden_config(json.loads(str(g["config"])))
    G, T = gpu_grid(prm, g["t"], g["nus"]), g["total"]
    ok = bool(np.all(np.abs(G - T) <= 2e-3 * np.abs(T) + 1e-2 * np.abs(T).max()))
    print(f"{name:24s} contract={'PASS' if ok else 'FAIL'}  max rel err >1e-2 peak = {rel(G, T, 1e-2):.3e}")
