"""What predicts the number of step attempts of a forward-shock row?  Needs a -DVAG_DYN_ROWSTATS build (profiles/build_variant.sh rowstats
-DVAG_DYN_ROWSTATS; VAG_LIB_PATH=variants/libvag_rowstats.so): every 61st row of an 8192-walker C4 step prints its accepted steps and start
record.  Prints the correlation of the step count with a few candidate predictors and the spread of the residual."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import bench
    from vegasafterglow_amd import _lib
    lib = _lib.load()
    h = C.c_void_p()
    _lib.check(lib.vag_ctx_create(0, C.byref(h)))
    _lib.check(lib.vag_ctx_set_stream(h, _lib.torch_stream_handle(torch.cuda.current_stream())))
    _lib.hooks["VAG_DYN_REFILL"] = "1"
    fit, defs, _ = bench.c4_fitter(lib, h, _lib)
    spec, lo, hi = fit.build_spec(defs)
    theta = lo + (hi - lo) * np.random.default_rng(0).random((8192, len(defs)))
    ev = fit.device_evaluator(defs, context=(h, bench._NullLock()))
    ll, _ = ev(torch.from_numpy(theta).to("cuda:0"))
    torch.cuda.synchronize()
    sys.exit(0)
out = subprocess.run([sys.executable, __file__, "child"], capture_output=True, text=True).stdout
rows = np.array([[float(v) for v in m.groups()] for m in re.finditer(
    r"R (\d+) steps (\d+) G0 (\S+) t0 (\S+) tlast (\S+) mjet (\S+) rho (\S+) A (\S+) nt (\d+)", out)])
print("rows sampled:", len(rows))
steps, G0, t0, tl, mjet, rho = rows[:, 1], rows[:, 2], rows[:, 3], rows[:, 4], rows[:, 5], rows[:, 6]
print("steps: min %d median %d mean %.1f p90 %d max %d" % (steps.min(), np.median(steps), steps.mean(), np.percentile(steps, 90), steps.max()))
cands = {"log(tlast/t0)": np.log(tl / t0), "log G0": np.log(G0), "log(G0-1)": np.log(np.maximum(G0 - 1, 1e-9)), "log rho": np.log(rho), "log mjet": np.log(mjet)}
for k, x in cands.items():
    print(f"  corr(steps, {k}) = {np.corrcoef(steps, x)[0, 1]:.3f}")
X = np.stack([np.ones_like(steps)] + list(cands.values()), 1)
coef, *_ = np.linalg.lstsq(X, steps, rcond=None)
res = steps - X @ coef
print("linear fit on all:", dict(zip(["1"] + list(cands), np.round(coef, 3))), "residual std %.1f (steps std %.1f)" % (res.std(), steps.std()))
X2 = np.stack([np.ones_like(steps), cands["log(tlast/t0)"]], 1)
c2, *_ = np.linalg.lstsq(X2, steps, rcond=None)
print("fit on log(tlast/t0) alone:", np.round(c2, 3), "residual std %.1f" % (steps - X2 @ c2).std())
