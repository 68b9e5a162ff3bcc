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


def gpu_comp4(prm, t, nu):
    t = np.ascontiguousarray(t, dtype=np.float64)
    nu = np.ascontiguousarray(nu, dtype=np.float64)
    comps = [np.zeros((nu.size, t.size)) for _ in range(4)]
    arr = (dp * 4)(*[a.ctypes.data_as(dp) for a in comps])
    q = _lib.ModelParams.from_buffer_copy(bytes(prm))
    _lib.check(lib.vag_flux_density_grid_components4_batch(h, C.byref(q), 1, t.ctypes.data_as(dp), t.size,
                                                           nu.ctypes.data_as(dp), nu.size, arr))
    return comps


print("\nper-component parity of the widened tiers (max rel err over bins > 1e-12 of the component's peak; '-' = component off)")
print(f"{'config':28s} {'fwd.sync':>10s} {'fwd.ssc':>10s} {'rvs.sync':>10s} {'rvs.ssc':>10s}")
tiers = {"C3 (FS+RS, SSC+KN)": (configs.C3, configs.C3_T, configs.C3_NU)}
tiers.update({k: v for k, v in configs.RS_CASES.items()})
tiers.update({k: (kw, configs.SPREAD_T, configs.SPREAD_NU) for k, kw in configs.SPREAD_CASES.items()})
tiers.update({k: (kw, configs.SPREAD_T, configs.SPREAD_NU) for k, kw in configs.PROFILE_CASES.items()})
for name, (kw, t, nu) in tiers.items():
    prm = _abi.make_params(**kw)
    O, G = orc.flux_components4(prm, t, nu), gpu_comp4(prm, t, nu)
    print(f"{name:28s} " + " ".join(f"{rel(g, o, 1e-12):10.2e}" if o.max() > 0 else f"{'-':>10s}" for g, o in zip(G, O)))

print("\nall 12 golden baselines of the reference test-suite (contract: |d| <= 2e-3 |ref| + 1e-2 max|ref| per component)")
import glob
for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz"))):
    name = os.path.basename(path)[:-4]
    if name.startswith("reference_vectors"):
        continue
    g = np.load(path)
    if "config" not in g.files:
        continue  # not a light-curve golden (e.g. the extinction-law vectors)
    prm = _abi.params_from_golden_config(json.loads(str(g["config"])))
    G = dict(zip(("fwd_sync", "fwd_ssc", "rvs_sync", "rvs_ssc"), gpu_comp4(prm, g["t"], g["nus"])))
    line = []
    for comp in ("fwd_sync", "fwd_ssc", "rvs_sync", "rvs_ssc"):
        T = g[comp]
        if T.ndim == 0:
            continue
        ok = bool(np.all(np.abs(G[comp] - T) <= 2e-3 * np.abs(T) + 1e-2 * np.abs(T).max()))
        line.append(f"{comp} contract={ok} rel(>1e-2 peak)={rel(G[comp], T, 1e-2):.1e}")
    print(f"{name:24s} " + "; ".join(line))
