"""Developer probe (GPU box): do two contexts, each serving half of a C3 / C5 batch from its own host thread and stream, overlap one
half's latency chains (pair solver, cooling) with the other half's SSC spectra and flux passes?
usage: python profiles/debug/two_contexts_c3.py [c3|c5] [n_models]"""
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "profiles"))
from ssc_ensemble import c3_batch, c5_batch  # noqa: E402
from vegasafterglow_amd import _lib  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "c3"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 512
lib = _lib.load()
dp = C.POINTER(C.c_double)
t, nu = np.logspace(2, 8, 100), np.array([1e9, 4.84e14, 1e18, 2.4e26])
prms = c3_batch(nb) if which == "c3" else c5_batch(nb)


def ctx():
    h = C.c_void_p()
    _lib.check(lib.vag_ctx_create(0, C.byref(h)))
    return h


def run(h, sub, reps, out, k):
    n = len(sub)
    arr = (_lib.ModelParams * n)(*[_lib.ModelParams.from_buffer_copy(bytes(p)) for p in sub])
    res = np.empty((n, nu.size, t.size))
    for r in range(reps + 1):
        if r == 1:
            t0 = time.perf_counter()
        _lib.check(lib.vag_flux_density_grid_batch(h, arr, n, t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp), nu.size, res.ctypes.data_as(dp)))
    out[k] = (time.perf_counter() - t0) / reps
    out[k + 2] = float(res.sum())


REPS = 4
h0, h1 = ctx(), ctx()
o = [0.0] * 4
run(h0, prms, REPS, o, 0)
one = o[0]
th = [threading.Thread(target=run, args=(h, sub, REPS, o, k)) for k, (h, sub) in enumerate(((h0, prms[: nb // 2]), (h1, prms[nb // 2:])))]
[x.start() for x in th]
[x.join() for x in th]
print(f"{which} {nb} models: one context {1e3 * one:.1f} ms per call -> {nb / one:.0f} LC/s;  two contexts x {nb // 2} models concurrently "
      f"{1e3 * max(o[0], o[1]):.1f} ms per call each -> {nb / max(o[0], o[1]):.0f} LC/s")
