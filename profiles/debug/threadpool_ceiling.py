"""The interpreter's own ceiling for the reference's thread-pool loop: bench.threadpool_bench's eval_one with the engine call replaced by
a ctypes call that returns at once (vag_abi_version: the GIL is released and re-taken exactly as around the real call), everything
else -- Model.from_params, the numpy chi^2, ThreadPoolExecutor.map -- unchanged.  What this loop reaches is what ANY engine behind the
unmodified samplers.py:59-91 loop can reach on this host.
usage: python3 profiles/debug/threadpool_ceiling.py [threads ...]"""
import ctypes as C, os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _abi, configs
import vegasafterglow_amd as va
from vegasafterglow_amd import _lib
lib = _lib.load()
t, nu = configs.c4_mock_data()
order = np.argsort(t)
ts, nus = np.ascontiguousarray(t[order]), np.ascontiguousarray(nu[order])
ln_fo, sig = np.log(np.full(ts.size, 1e-28)), 0.1
rng = np.random.default_rng(0)
prms = []
for _ in range(1024):
    kw = dict(configs.C4_TRUTH)
    for (name, lg, lo, hi) in configs.C4_FREE:
        v = rng.uniform(lo, hi)
        kw[{"theta_v": "theta_obs"}.get(name, name)] = 10 ** v if lg else v
    prms.append(_abi.make_params(**kw))
F_stub = np.full(ts.size, 1e-28)


def eval_one(p):
    m = va.Model.from_params(p)          # as the real loop
    lib.vag_abi_version()                # the engine call's GIL release / re-take, without the engine
    F = F_stub.copy()
    r = (ln_fo - np.log(np.maximum(F, 1e-300))) / sig
    return -0.5 * float(np.dot(r, r))


for n in [int(a) for a in sys.argv[1:]] or [32]:
    with ThreadPoolExecutor(n) as ex:
        list(ex.map(eval_one, prms[:64]))
        t0 = time.perf_counter()
        for _ in range(3):
            list(ex.map(eval_one, prms))
        dt = (time.perf_counter() - t0) / 3
    print("threads %d: %.0f evaluations/s with the engine call stubbed out (%.1f us each)" % (n, len(prms) / dt, 1e6 * dt / len(prms)), flush=True)
t0 = time.perf_counter()
for p in prms:
    eval_one(p)
dt = time.perf_counter() - t0
print("no pool, one thread: %.0f evaluations/s (%.1f us each)" % (len(prms) / dt, 1e6 * dt / len(prms)))
