"""Ad-hoc GPU parity probe for the reverse-shock tier (not a pytest file): python profiles/debug/gpu_debug_rs.py"""
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


def gpu_comp4(prm, t, nu):
    t = np.ascontiguousarray(t, dtype=np.float64); nu = np.ascontiguousarray(nu, dtype=np.float64)
    comps = [np.zeros((nu.size, t.size)) for _ in range(4)]
    arr = (dp * 4)(*[a.ctypes.data_as(dp) for a in comps])
    q = _lib.ModelParams.from_buffer_copy(bytes(prm))
    rc = lib.vag_flux_density_grid_components4_batch(h, C.byref(q), 1, t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp), nu.size, arr)
    if rc: raise RuntimeError(lib.vag_last_error().decode())
    return comps


def rel(a, b):
    m = b > 1e-12 * b.max()
    if not m.any(): return 0.0
    return (np.abs(a - b) / np.where(m, b, 1))[m].max()


cases = {}
for name in ["rs_thick", "gauss_ism_rs", "powerlaw_wind_rs"]:
    g = np.load(os.path.join(_abi.ROOT, "tests", "golden", name + ".npz"))
    cases[name] = (_abi.params_from_golden_config(json.loads(str(g["config"]))), g["t"], g["nus"])
for k, (kw, t, nu) in configs.RS_CASES.items():
    cases[k] = (_abi.make_params(**kw), t, nu)
cases["C3"] = (_abi.make_params(**configs.C3), configs.C3_T[::4], configs.C3_NU)
for name, (prm, t, nu) in cases.items():
    t0 = time.time(); O = orc.flux_components4(prm, t, nu); to = time.time() - t0
    try:
        t0 = time.time(); G = gpu_comp4(prm, t, nu); tg = time.time() - t0
    except RuntimeError as e:
        print(f"{name:24s} ERROR {e}"); continue
    print(f"{name:24s} " + " ".join(f"{rel(g, o):.2e}" for g, o in zip(G, O)) + f" nan={sum(np.isnan(g).sum() for g in G)} oracle {to*1e3:7.1f} ms gpu {tg*1e3:7.1f} ms", flush=True)
    if "-v" in sys.argv:
        np.set_printoptions(linewidth=220, precision=4)
        for g, o, nm in zip(G, O, ("fs", "fssc", "rs", "rssc")):
            if o.max() > 0 and rel(g, o) > 1e-5: print(nm, "ratio\n", g / np.where(o > 0, o, 1))


def gpu_details(prm, tmin, tmax, rvs):
    sh = _lib.DetailsShape()
    q = _lib.ModelParams.from_buffer_copy(bytes(prm))
    fn = lib.vag_details_rvs if rvs else lib.vag_details
    _lib.check(fn(h, C.byref(q), tmin, tmax, C.byref(sh), None))
    d = {"phi": np.zeros(sh.n_phi), "theta": np.zeros(sh.n_theta)}
    for n in ("t_src", "Gamma", "r", "t_comv", "B", "N_p", "Gamma_th"):
        d[n] = np.zeros((sh.n_theta, sh.n_t))
    out = _lib.DetailsOut(*[d[n].ctypes.data_as(dp) for n, _ in _lib.DetailsOut._fields_])
    _lib.check(fn(h, C.byref(q), tmin, tmax, C.byref(sh), C.byref(out)))
    return d


if "-d" in sys.argv:
    name = sys.argv[sys.argv.index("-d") + 1]
    prm, t, nu = cases[name]
    for rvs in (False, True):
        g = gpu_details(prm, t.min(), t.max(), rvs); o = orc.details(prm, t.min(), t.max(), rvs=rvs)
        print("rvs" if rvs else "fwd", "shape", g["t_src"].shape, o["t_src"].shape)
        for k in ("theta", "t_src", "Gamma", "r", "t_comv", "B", "N_p", "Gamma_th"):
            a, b = g[k], o[k]
            with np.errstate(all="ignore"):
                e = np.abs(a - b) / np.where(b != 0, np.abs(b), 1)
            e = np.where(np.isfinite(e), e, 0)
            rows = np.argsort(e.max(axis=-1))[-3:] if e.ndim == 2 else None
            print(f"  {k:9s} max rel {e.max():.3e}", "worst rows", rows, "" if rows is None else e.max(axis=-1)[rows])
        if rvs:
            e = np.abs(g["B"] - o["B"]) / np.where(o["B"] != 0, np.abs(o["B"]), 1)
            j = int(np.argmax(e.max(axis=1)))
            np.set_printoptions(linewidth=220, precision=5)
            print("  worst row", j, "theta", o["theta"][j], "inj(oracle)", o["injection_idx"][j, 0])
            for k in ("Gamma", "Gamma_th", "B", "N_p"):
                print("   ", k, "gpu", g[k][j][::12]); print("   ", k, "orc", o[k][j][::12])
