"""How well-conditioned are the three reverse-shock-on-structured-jet cases in the REFERENCE itself?  (dev container: oracle/_ref)

For each case, at the default ODE tolerance: the spread between the reference's two builds (its own -O3 flags vs strict FP), and the
change of each build's components when ONE input (theta_c, Gamma0, E_iso, theta_obs) moves by ONE ulp either way -- rel. change over
the bins above 1e-2 of the component's peak, the same measure profiles/rs_structured_diagnostic.py uses for GPU vs checker.
usage: python profiles/rs_one_ulp_sensitivity.py > profiles/r03_rs_one_ulp_sensitivity.txt
Also writes tests/golden/rs_one_ulp_sensitivity.json (per case and component: the larger of the build spread and the one-ulp responses),
which the GPU parity tests use as the gate for these cases."""
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _abi  # noqa: E402
import configs  # noqa: E402

COMP = ("fwd.sync", "fwd.ssc", "rvs.sync", "rvs.ssc")
FIELDS = ("theta_c", "Gamma0", "E_iso", "theta_obs")


def rel(a, b, floor=1e-2):
    m = np.abs(b) > floor * np.abs(b).max()
    return float(np.max(np.abs(a - b)[m] / np.abs(b)[m])) if m.any() else 0.0


cases = {}
kw, t, nu = configs.RS_CASES["rs_gaussian_adiabatic"]
cases["rs_gaussian_adiabatic"] = (_abi.make_params(**kw), t, nu)
g = np.load(os.path.join(ROOT, "tests", "golden", "gauss_ism_rs.npz"))
cases["gauss_ism_rs (reference golden)"] = (_abi.params_from_golden_config(json.loads(str(g["config"]))), np.ascontiguousarray(g["t"]), np.ascontiguousarray(g["nus"]))
cases["step_powerlaw_rs_spread"] = (_abi.make_params(**configs.PROFILE_CASES["step_powerlaw_rs_spread"]), configs.SPREAD_T, configs.SPREAD_NU)
kw, t, nu = configs.RS_CASES["rs_thin_tophat"]
cases["rs_thin_tophat (control: a top-hat jet)"] = (_abi.make_params(**kw), t, nu)

builds = {"-O3 build": _abi.load_ref(), "strict build": _abi.CpuLib(os.path.join(ROOT, "oracle", "_ref", "libvag_ref_strict.so"), "vag_ref")}
demonstrated = {}
for name, (prm, t, nu) in cases.items():
    print(f"==== {name}")
    base = {b: lib.flux_components4(prm, t, nu) for b, lib in builds.items()}
    live = [k for k in (0, 2) if base["strict build"][k].max() > 0]
    dem = {COMP[k]: rel(base["-O3 build"][k], base["strict build"][k]) for k in live}
    print("  spread between the two builds: " + ", ".join(f"{COMP[k]} {rel(base['-O3 build'][k], base['strict build'][k]):.2e}" for k in live))
    for b, lib in builds.items():
        worst = {k: (0.0, "") for k in live}
        for field in FIELDS:
            for up in (True, False):
                q = _abi.ModelParams.from_buffer_copy(bytes(prm))
                setattr(q, field, float(np.nextafter(getattr(q, field), np.inf if up else -np.inf)))
                o = lib.flux_components4(q, t, nu)
                for k in live:
                    e = rel(o[k], base[b][k])
                    if e > worst[k][0]:
                        worst[k] = (e, f"{field} {'+' if up else '-'}1 ulp")
        for k in live:
            dem[COMP[k]] = max(dem[COMP[k]], worst[k][0])
        print(f"  {b}, largest change under a one-ulp move of one input: " + ", ".join(f"{COMP[k]} {worst[k][0]:.2e} ({worst[k][1]})" for k in live))
    demonstrated[name.split(" ")[0]] = dem
with open(os.path.join(ROOT, "tests", "golden", "rs_one_ulp_sensitivity.json"), "w") as f:
    json.dump(demonstrated, f, indent=1, sort_keys=True)
