"""Developer aid (GPU box): the three reverse-shock-on-structured-jet cases through several builds of the library (paths on the
command line, e.g. libraries built from earlier commits in a worktree), each against the checker and against the first build.
Raw ctypes on the entry points every ABI version has, so that an older library loads.
usage: python profiles/debug/rs_lib_history.py variants/libvag_a.so variants/libvag_b.so ..."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _abi  # noqa: E402
import configs  # noqa: E402

dp = C.POINTER(C.c_double)
COMP = ("fwd.sync", "fwd.ssc", "rvs.sync", "rvs.ssc")


def rel(a, b, floor=1e-2):
    m = np.abs(b) > floor * np.abs(b).max()
    return float(np.max(np.abs(a - b)[m] / np.abs(b)[m])) if m.any() else 0.0


def components4(lib, h, prm, t, nu):
    outs = [np.zeros((1, nu.size, t.size)) for _ in range(4)]
    ptrs = (dp * 4)(*[o.ctypes.data_as(dp) for o in outs])
    p = _abi.ModelParams.from_buffer_copy(bytes(prm))
    rc = lib.vag_flux_density_grid_components4_batch(h, C.byref(p), 1, t.ctypes.data_as(dp), C.c_int(t.size), nu.ctypes.data_as(dp), C.c_int(nu.size), ptrs)
    assert rc == 0, rc
    return [o[0] for o in outs]


cases = {}
kw, t, nu = configs.RS_CASES["rs_gaussian_adiabatic"]
cases["rs_gaussian_adiabatic"] = (_abi.make_params(**kw), t, nu)
g = np.load(os.path.join(ROOT, "tests", "golden", "gauss_ism_rs.npz"))
cases["gauss_ism_rs"] = (_abi.params_from_golden_config(json.loads(str(g["config"]))), np.ascontiguousarray(g["t"]), np.ascontiguousarray(g["nus"]))
cases["step_powerlaw_rs_spread"] = (_abi.make_params(**configs.PROFILE_CASES["step_powerlaw_rs_spread"]), configs.SPREAD_T, configs.SPREAD_NU)
kw, t, nu = configs.RS_CASES["rs_thin_tophat"]
cases["rs_thin_tophat (control)"] = (_abi.make_params(**kw), t, nu)

orc = _abi.load_oracle()
first = {}
for path in sys.argv[1:]:
    lib = C.CDLL(os.path.abspath(path))
    lib.vag_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.vag_flux_density_grid_components4_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, dp, C.c_int, dp, C.c_int, C.POINTER(dp)]
    h = C.c_void_p()
    assert lib.vag_ctx_create(0, C.byref(h)) == 0
    print(f"######## {os.path.basename(path)}")
    for name, (prm0, t, nu) in cases.items():
        for rtol in (1e-6, 1e-9):
            prm = _abi.ModelParams.from_buffer_copy(bytes(prm0))
            prm.rtol = rtol
            key = (name, rtol)
            if key not in first:
                first[key] = (orc.flux_components4(prm, t, nu), None)
            want, base = first[key]
            got = components4(lib, h, prm, t, nu)
            if base is None:
                first[key] = (want, got)
                base = got
            print(f"  {name:28s} rtol {rtol:g}: vs checker " + ", ".join(f"{c} {rel(a, w):.2e}" for a, w, c in zip(got, want, COMP) if w.max() > 0)
                  + "   vs first build " + ", ".join(f"{c} {rel(a, b):.2e}" for a, b, w, c in zip(got, base, want, COMP) if w.max() > 0))
