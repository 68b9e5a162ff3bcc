"""Developer probe: the magnetar draw that met an SSC cell without a table (SWEEP_MODE=magnetar, draw 19 of 30)."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _abi
from vegasafterglow_amd import _lib
import vegasafterglow_amd as va
kw = {'jet': 'GaussianJet', 'E_iso': 4.460426649397416e+51, 'Gamma0': 427.30930019677726, 'theta_c': 0.07602877048106377, 'theta_obs': 0.015165954752948074,
      'p': 2.1137269063496933, 'eps_e': 0.06343644661487786, 'eps_B': 0.0024324771035933146, 'ssc': True, 'kn': False, 'medium': 'Wind',
      'A_star': 1.865321778529919, 'magnetar': (4.722996518123433e+46, 669.4503306793644, 2.0487730053290845)}
prm = _abi.make_params(**kw)
lib = _lib.load(); h, lock = va.get_context(0); dp = C.POINTER(C.c_double)
t, nu = np.logspace(1.5, 7.5, 30), np.array([1e9, 4.84e14, 1e18, 2.4e22, 1e26])
arr = (_lib.ModelParams * 1)(_lib.ModelParams.from_buffer_copy(bytes(prm)))
def run(tag, **env):
    for k, v in env.items(): _lib.hooks[k] = v
    s, c = np.empty((1, nu.size, t.size)), np.empty((1, nu.size, t.size))
    try:
        _lib.check(lib.vag_flux_density_grid_components_batch(h, arr, 1, t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp), nu.size, s.ctypes.data_as(dp), c.ctypes.data_as(dp)))
        print(tag, "ok", s.sum(), c.sum())
    except RuntimeError as e:
        print(tag, "FAILED:", str(e)[-80:])
    for k in env: _lib.hooks.pop(k)
    return s, c
run("default")
run("all cells", VAG_IC_ALL_CELLS="1")
run("no fused", VAG_NO_FUSED="1")
_lib.hooks["VAG_IC_ALL_CELLS"] = "1"
m = va.Model.from_params(prm)
d = m.details(t.min(), t.max())
t_obs = d.fwd.t_obs
print("t_obs", t_obs.shape, "ascending along k:", bool(np.all(np.diff(t_obs, axis=2) > 0)), "finite", bool(np.isfinite(t_obs).all()))
idx = np.searchsorted(t, t_obs)
has = np.zeros(t_obs.shape, bool); has[:, :, :-1] = idx[:, :, 1:] > idx[:, :, :-1]
need = has.copy(); need[:, :, 1:] |= has[:, :, :-1]
lo = np.concatenate([t_obs[:, :, :1], t_obs[:, :, :-1]], axis=2); hi = np.concatenate([t_obs[:, :, 1:], t_obs[:, :, -1:]], axis=2)
cons = ((lo <= t.max()) & (hi >= t.min()))
print("exact needed cells", need.any(axis=0).sum(), "range test", cons.any(axis=0).sum(), "exact but not range:", (need.any(axis=0) & ~cons.any(axis=0)).sum())
print("t_obs min/max", t_obs.min(), t_obs.max(), "request", t.min(), t.max())
bad = np.argwhere(~np.isfinite(t_obs)); print("non-finite t_obs cells", len(bad), bad[:5])

nd = need.any(axis=0)
G = d.fwd.Gamma if hasattr(d.fwd, "Gamma") else None
print("rows x nodes", nd.shape, "needed per row (first 12 rows):", nd.sum(axis=1)[:12], "theta[:6]", getattr(d, "theta", None)[:6] if hasattr(d, "theta") else "")
print("t_obs[0, :3, :4]", t_obs[0, :3, :4]); print("t_obs[-1, :3, :4]", t_obs[-1, :3, :4])
