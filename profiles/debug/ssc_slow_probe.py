"""Which models send SSC cells to vag_ic_photon_slow_kernel by themselves (lattices beyond the fast kernel's on-chip limits)?
A hunt over corners of the parameter space; prints vag_plan.n_ssc_slow_cells per model and compares with the CPU checker where some do.
Usage (GPU box): python profiles/debug/ssc_slow_probe.py"""
import ctypes as C
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import _abi  # noqa: E402
from vegasafterglow_amd import _lib  # noqa: E402

lib = _lib.load()
h = C.c_void_p()
_lib.check(lib.vag_ctx_create(0, C.byref(h)))
dp = C.POINTER(C.c_double)
t, nu = np.logspace(1, 8.5, 24), np.array([1e8, 1e12, 1e16, 1e20, 1e24, 1e27])


def run(prm):
    arr = (_lib.ModelParams * 1)(_lib.ModelParams.from_buffer_copy(bytes(prm)))
    comps = [np.empty((1, nu.size, t.size)) for _ in range(4)]
    out4 = (dp * 4)(*[a.ctypes.data_as(dp) for a in comps])
    rc = lib.vag_flux_density_grid_components4_batch(h, arr, 1, t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp), nu.size, out4)
    pl = _lib.Plan()
    lib.vag_last_plan(h, C.byref(pl))
    return rc, comps, pl.n_ssc_slow_cells


found = []
for eps_B, n_ism, E_iso, eps_e, G0, p in itertools.product([1e-9, 1e-6, 0.3], [1e-6, 1e-3, 1e3], [1e50, 1e55], [1e-3, 0.3], [20.0, 1000.0], [2.05, 2.9]):
    kw = dict(jet="TophatJet", theta_obs=0.0, ssc=True, kn=True, eps_B=eps_B, n_ism=n_ism, E_iso=E_iso, eps_e=eps_e, Gamma0=G0, p=p,
              resolutions=(0.06, 0.15, 3.0))
    rc, comps, n_slow = run(_abi.make_params(**kw))
    if rc or n_slow:
        print(f"eps_B {eps_B:g} n {n_ism:g} E {E_iso:g} eps_e {eps_e:g} G0 {G0:g} p {p}: rc {rc} slow cells {n_slow}", flush=True)
        if n_slow:
            found.append(kw)
print(f"{len(found)} models with cells on the slow path")
if found and "--check" in sys.argv:
    oracle = _abi.load_oracle()  # test infrastructure: the CPU checker
    for kw in found[:4]:
        prm = _abi.make_params(**kw)
        rc, comps, n_slow = run(prm)
        want = oracle.flux_components4(prm, t, nu)
        for c in range(4):
            w = np.asarray(want[c])
            if w.max() > 0:
                print(kw, c, float(np.max(np.abs(comps[c][0] - w) / (2e-3 * np.abs(w) + 1e-2 * w.max()))))
