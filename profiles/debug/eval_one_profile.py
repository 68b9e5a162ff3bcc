"""Where does the interpreter's share of one thread-pool evaluation go?  cProfile of bench.threadpool_bench's eval_one on one thread
(the engine call shows as the ctypes function; everything else is GIL-holding time that bounds a thread pool's rate).
usage: python3 profiles/debug/eval_one_profile.py"""
import cProfile, ctypes as C, os, pstats, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, _abi, configs
import vegasafterglow_amd as va
from vegasafterglow_amd import _lib
lib = _lib.load(); h = C.c_void_p(); _lib.check(lib.vag_ctx_create(0, C.byref(h)))
fit, defs, (t, nu, f_obs) = bench.c4_fitter(lib, h, _lib)
_, lo, hi = fit.build_spec(defs)
theta = lo + (hi - lo) * np.random.default_rng(0).random((512, len(defs)))
prms = []
for s in theta:
    kw = dict(configs.C4_TRUTH)
    for (name, lg, _, _), v in zip(configs.C4_FREE, s):
        kw[{"theta_v": "theta_obs"}.get(name, name)] = 10 ** v if lg else v
    prms.append(_abi.make_params(**kw))
order = np.argsort(t)
ts, nus, fo = np.ascontiguousarray(t[order]), np.ascontiguousarray(nu[order]), f_obs[order]
ln_fo, sig = np.log(fo), 0.1
def eval_one(p):
    F = va.Model.from_params(p).flux_density(ts, nus).total
    r = (ln_fo - np.log(np.maximum(F, 1e-300))) / sig
    return -0.5 * float(np.dot(r, r))
for p in prms[:32]: eval_one(p)
t0 = time.perf_counter()
for p in prms: eval_one(p)
print("plain: %.1f us per evaluation" % (1e6 * (time.perf_counter() - t0) / len(prms)))
pr = cProfile.Profile(); pr.enable()
for p in prms: eval_one(p)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
