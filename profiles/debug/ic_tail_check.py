"""Developer aid (GPU box): the packed tail pass of vag_ic_photon_kernel against its main pass.  A library built with
-DVAG_IC_MAIN_BINS=24 sends every seed bin beyond the 24th through the tail pass (W = 8 ... 64 lanes per bin set, each lane at
its own electron energy); the product sends the bins beyond the 64th.  Both must give the same spectra up to the order of the
sums: the C3 batch (Klein-Nishina, forward + reverse shock) and a slice of the C5 batch (Thomson), per component.
usage: profiles/build_variant.sh ictail24 -DVAG_IC_MAIN_BINS=24; python profiles/debug/ic_tail_check.py [n_models]"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    import ctypes as C
    from ssc_ensemble import c3_batch, c5_batch
    from vegasafterglow_amd import _lib
    lib = _lib.load()
    h = C.c_void_p()
    _lib.check(lib.vag_ctx_create(0, C.byref(h)))
    nb = int(sys.argv[3])
    dp = C.POINTER(C.c_double)
    t, nu = np.logspace(2, 8, 60), np.array([1e9, 4.84e14, 1e18, 2.4e26])
    res = {}
    for name, prms in (("c3", c3_batch(nb)), ("c5", c5_batch(nb))):
        arr = (_lib.ModelParams * nb)(*[_lib.ModelParams.from_buffer_copy(bytes(p)) for p in prms])
        outs = [np.zeros((nb, nu.size, t.size)) for _ in range(4)]
        ptrs = (dp * 4)(*[o.ctypes.data_as(dp) for o in outs])
        _lib.check(lib.vag_flux_density_grid_components4_batch(h, arr, nb, t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp), nu.size, ptrs))
        for c, o in zip(("fwd_sync", "fwd_ssc", "rvs_sync", "rvs_ssc"), outs):
            res[name + "_" + c] = o
    np.savez(sys.argv[2], **res)
    sys.exit(0)
nb = sys.argv[1] if len(sys.argv) > 1 else "48"
outs = []
for tag, path in (("product", None), ("tail24", os.path.join(ROOT, "variants", "libvag_ictail24.so"))):
    env = dict(os.environ)
    if path:
        env["VAG_LIB_PATH"] = path
    f = f"/tmp/ic_tail_{tag}.npz"
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "child", f, nb], env=env)
    outs.append(np.load(f))
a, b = outs
for k in a.files:
    x, y = a[k], b[k]
    if x.max() == 0:
        print(f"{k:14s} all zero in both: {bool(np.all(y == 0))}")
        continue
    m = x > 1e-12 * x.max()
    print(f"{k:14s} max rel. difference {float(np.max(np.abs(x - y)[m] / x[m])):.2e}   finite {bool(np.isfinite(y).all())}")
