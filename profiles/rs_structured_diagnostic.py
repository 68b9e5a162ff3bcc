"""Diagnostic for the three reverse-shock-on-structured-jet cases whose GPU-vs-oracle agreement is looser than 2e-6 at the default
ODE tolerance: are they the coupled ODE's step-sequence sensitivity, or a defect?

For each case: (1) per-component flux agreement at rtol = 1e-6 (default) and at rtol = 1e-9 on BOTH sides -- a step-sequence effect
collapses with the tolerance, a defect does not; (2) per theta row, the largest relative difference of the reverse shock's
Gamma / Gamma_th / B / N_p arrays (vag_details_rvs vs the oracle) at both tolerances and the rows' initial Lorentz factors;
(3) when oracle/_ref is present (dev container), the same row table between the reference's own -O3 and strict builds.
usage: python profiles/rs_structured_diagnostic.py  > profiles/r02_rs_structured_diagnostic.txt"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _abi  # noqa: E402
import configs  # noqa: E402
import vegasafterglow_amd as va  # noqa: E402
from vegasafterglow_amd import _lib  # noqa: E402

dp = C.POINTER(C.c_double)
lib = _lib.load()
h, _ = va.get_context(0)
orc = _abi.load_oracle()
COMP = ("fwd.sync", "fwd.ssc", "rvs.sync", "rvs.ssc")


def gpu_components4(prm, t, nu):
    arr = (_lib.ModelParams * 1)(_lib.ModelParams.from_buffer_copy(bytes(prm)))
    outs = [np.zeros((1, nu.size, t.size)) for _ in range(4)]
    ptrs = (dp * 4)(*[o.ctypes.data_as(dp) for o in outs])
    _lib.check(lib.vag_flux_density_grid_components4_batch(h, arr, 1, t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp), nu.size, ptrs))
    return [o[0] for o in outs]


def gpu_details_rvs(prm, t_min, t_max):
    sh = _lib.DetailsShape()
    p = _lib.ModelParams.from_buffer_copy(bytes(prm))
    _lib.check(lib.vag_details_rvs(h, C.byref(p), t_min, t_max, C.byref(sh), None))
    d = {"phi": np.zeros(sh.n_phi), "theta": np.zeros(sh.n_theta)}
    for n in ("t_src", "Gamma", "r", "t_comv", "B", "N_p", "Gamma_th"):
        d[n] = np.zeros((sh.n_theta, sh.n_t))
    out = _lib.DetailsOut(*[d[n].ctypes.data_as(dp) for n, _ in _lib.DetailsOut._fields_])
    _lib.check(lib.vag_details_rvs(h, C.byref(p), t_min, t_max, C.byref(sh), C.byref(out)))
    return d


def rel(a, b, floor=1e-9):
    m = np.abs(b) > floor * np.abs(b).max()
    return float(np.max(np.abs(a - b)[m] / np.abs(b)[m])) if m.any() else 0.0


def golden_case(name):
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    return _abi.params_from_golden_config(json.loads(str(g["config"]))), np.ascontiguousarray(g["t"]), np.ascontiguousarray(g["nus"])


cases = {}
kw, t, nu = configs.RS_CASES["rs_gaussian_adiabatic"]
cases["rs_gaussian_adiabatic"] = (_abi.make_params(**kw), t, nu)
cases["gauss_ism_rs (reference golden)"] = golden_case("gauss_ism_rs")
kw = configs.PROFILE_CASES["step_powerlaw_rs_spread"]
cases["step_powerlaw_rs_spread"] = (_abi.make_params(**kw), configs.SPREAD_T, configs.SPREAD_NU)
ref_fast, ref_strict = _abi.load_ref(), None
sp = os.path.join(ROOT, "oracle", "_ref", "libvag_ref_strict.so")
if os.path.exists(sp):
    ref_strict = _abi.CpuLib(sp, "vag_ref")

for name, (prm0, t, nu) in cases.items():
    print(f"==== {name}")
    for rtol in (1e-6, 1e-9):
        prm = _abi.ModelParams.from_buffer_copy(bytes(prm0))
        prm.rtol = rtol
        want = orc.flux_components4(prm, t, nu)
        got = gpu_components4(prm, t, nu)
        line = ", ".join(f"{c} {rel(g, w, 1e-2):.2e}" for g, w, c in zip(got, want, COMP) if w.max() > 0)
        print(f"  flux components, GPU vs oracle, rtol {rtol:g} (bins above 1e-2 of the peak): {line}")
        dg = gpu_details_rvs(prm, float(t.min()), float(t.max()))
        do = orc.details(prm, float(t.min()), float(t.max()), rvs=True)
        rows = []
        for j in range(dg["Gamma"].shape[0]):
            e = max(rel(dg[k][j], do[k][j], 1e-6) for k in ("Gamma", "Gamma_th", "B", "N_p"))
            rows.append(e)
        rows = np.array(rows)
        worst = np.argsort(-rows)[:5]
        print(f"  reverse-shock arrays per theta row, rtol {rtol:g}: median {np.median(rows):.2e}, rows above 1e-5: {int((rows > 1e-5).sum())} of {rows.size}; "
              f"worst rows (index: theta, max rel): " + ", ".join(f"{j}: {dg['theta'][j]:.4f}, {rows[j]:.2e}" for j in worst))
        if ref_strict is not None and ref_fast is not None and rtol == 1e-6:
            a = ref_fast.flux_components4(prm, t, nu)
            b = ref_strict.flux_components4(prm, t, nu)
            line = ", ".join(f"{c} {rel(x, y, 1e-2):.2e}" for x, y, c in zip(a, b, COMP) if y.max() > 0)
            print(f"  the reference's OWN two builds (-O3 flags vs strict FP), rtol {rtol:g}: {line}")
            da = ref_fast.details(prm, float(t.min()), float(t.max()), rvs=True)
            db = ref_strict.details(prm, float(t.min()), float(t.max()), rvs=True)
            rr = np.array([max(rel(da[k][j], db[k][j], 1e-6) for k in ("Gamma", "Gamma_th", "B", "N_p")) for j in range(da["Gamma"].shape[0])])
            print(f"    their reverse-shock arrays per theta row: median {np.median(rr):.2e}, rows above 1e-5: {int((rr > 1e-5).sum())} of {rr.size}; "
                  f"overlap of their 10 worst rows with GPU-vs-oracle's 10 worst: {len(set(np.argsort(-rr)[:10]) & set(np.argsort(-rows)[:10]))}")
