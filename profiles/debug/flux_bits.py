"""Developer aid (GPU box): the C2 bench batch through two builds of the library (VAG_LIB_A, VAG_LIB_B) or two settings of one
(VAG_ENV_A / VAG_ENV_B = "NAME=value ..."): the fluxes must be the same bits when a change only re-schedules the flux kernel (same
boundary values, same sum order).  Each side runs in its own process."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import bench
    import configs
    from vegasafterglow_amd import _lib
    lib = _lib.load()
    h = C.c_void_p()
    _lib.check(lib.vag_ctx_create(0, C.byref(h)))
    nb = int(sys.argv[3])
    arr = bench.c2_batch(nb, seed=1234)
    t, nu = configs.C2_T, configs.C2_NU
    out = np.empty((nb, nu.size, t.size))
    dp = C.POINTER(C.c_double)
    _lib.check(lib.vag_flux_density_grid_batch(h, C.cast(arr, C.POINTER(_lib.ModelParams)), nb, t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp), nu.size, out.ctypes.data_as(dp)))
    np.save(sys.argv[2], out)
    sys.exit(0)
nb = sys.argv[1] if len(sys.argv) > 1 else "64"
outs = []
for tag in ("A", "B"):
    path = f"/tmp/flux_bits_{tag}.npy"
    env = dict(os.environ)
    if os.environ.get(f"VAG_LIB_{tag}"):
        env["VAG_LIB_PATH"] = os.environ[f"VAG_LIB_{tag}"]
    for kv in os.environ.get(f"VAG_ENV_{tag}", "").split():  # e.g. VAG_ENV_A="VAG_FLUX_NO_WIDE=1": a switch of the same library
        k, v = kv.split("=", 1)
        env[k] = v
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "child", path, nb], env=env)
    outs.append(np.load(path))
a, b = outs
print("bitwise equal:", np.array_equal(a, b), " max rel diff:", float(np.max(np.abs(a - b) / np.maximum(np.abs(a), 1e-300))), " finite:", bool(np.isfinite(a).all()))
