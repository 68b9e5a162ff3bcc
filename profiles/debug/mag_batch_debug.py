"""Developer probe: the SWEEP_MODE=magnetar batch of 30 (kn) in which model 19 met an SSC cell without a table; subsets of it."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _abi
from vegasafterglow_amd import _lib
import vegasafterglow_amd as va
rng = np.random.default_rng(4242)
prms = []
KN = os.environ.get("KN", "0") == "1"  # the failing batch is the Thomson one, drawn after the 30 Klein-Nishina models
for i in range(60):
    jet = ["TophatJet", "GaussianJet", "PowerLawJet"][(i % 30) % 3]
    kw = dict(jet=jet, E_iso=10 ** rng.uniform(50.5, 54), Gamma0=10 ** rng.uniform(1.5, 2.9), theta_c=rng.uniform(0.03, 0.3),
              theta_obs=rng.uniform(0, 0.5), p=rng.uniform(2.05, 2.9), eps_e=10 ** rng.uniform(-2.5, -0.5),
              eps_B=10 ** rng.uniform(-6, -1), ssc=True, kn=i < 30)
    if (i % 30) % 2:
        kw.update(medium="Wind", A_star=10 ** rng.uniform(-2, 0.5))
    else:
        kw.update(n_ism=10 ** rng.uniform(-3, 2))
    if jet == "PowerLawJet":
        kw.update(k_e=rng.uniform(1.5, 3.0), k_g=rng.uniform(1.5, 3.0))
    kw["magnetar"] = (10 ** rng.uniform(45, 48), 10 ** rng.uniform(2, 4), rng.uniform(1.5, 2.5))
    prms.append(_abi.make_params(**kw))
    if i == 49: print("draw 19 of the Thomson set:", kw)
prms = prms[:30] if KN else prms[30:]
lib = _lib.load(); h, lock = va.get_context(0); dp = C.POINTER(C.c_double)
t, nu = np.logspace(1.5, 7.5, 30), np.array([1e9, 4.84e14, 1e18, 2.4e22, 1e26])
def run(tag, idx, **env):
    for k, v in env.items(): _lib.hooks[k] = v
    n = len(idx)
    arr = (_lib.ModelParams * n)(*[_lib.ModelParams.from_buffer_copy(bytes(prms[i])) for i in idx])
    s, c = np.empty((n, nu.size, t.size)), np.empty((n, nu.size, t.size))
    try:
        _lib.check(lib.vag_flux_density_grid_components_batch(h, arr, n, t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp), nu.size, s.ctypes.data_as(dp), c.ctypes.data_as(dp)))
        print(tag, "ok", flush=True)
        ok = True
    except RuntimeError as e:
        print(tag, "FAILED:", str(e)[-90:], flush=True)
        ok = False
    for k in env: _lib.hooks.pop(k)
    return ok
run("all 30", list(range(30)))
run("all 30 again", list(range(30)))
run("all 30, every table", list(range(30)), VAG_IC_ALL_CELLS="1")
run("all 30, no fused", list(range(30)), VAG_NO_FUSED="1")
run("0..19", list(range(20)))
run("19..29", list(range(19, 30)))
for j in range(30):
    if j != 19 and not run(f"pair ({j}, 19)", [j, 19]):
        break
for j in range(30):
    if j != 19 and not run(f"pair (19, {j})", [19, j]):
        break
