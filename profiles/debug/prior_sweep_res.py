"""Ad-hoc robustness sweep over the RESOLUTIONS a user may pass (grid-refinement.h:639-706 sizes the grids freely): random (phi, theta, t)
resolutions around and beyond the grid kernel's small layout (256 theta / 208 phi nodes) and the 512-node staged row, random jets and
viewing angles, every fourth draw with SSC, every fifth with a reverse shock; evaluated as ONE ragged batch (models of both layouts in
one call) and one by one, against the checker.  usage: python profiles/debug/prior_sweep_res.py [n]"""
import os, sys
import numpy as np
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(_ROOT, "tests")); sys.path.insert(0, _ROOT)
import ctypes as C
import _abi
from vegasafterglow_amd import _lib
import vegasafterglow_amd as va

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lib = _lib.load(); h, lock = va.get_context(0); orc = _abi.load_oracle(); dp = C.POINTER(C.c_double)
rng = np.random.default_rng(90210)
t, nu = np.logspace(2, 7.5, 40), np.array([1e9, 4.84e14, 1e18])
prms, kws = [], []
for i in range(n):
    jet = ["TophatJet", "GaussianJet", "PowerLawJet", "TwoComponentJet"][i % 4]
    kw = dict(jet=jet, E_iso=10 ** rng.uniform(51, 53.5), Gamma0=10 ** rng.uniform(1.7, 2.7), theta_c=rng.uniform(0.04, 0.25),
              theta_obs=rng.uniform(0, 0.5) if i % 3 else 0.0, n_ism=10 ** rng.uniform(-2, 1), p=rng.uniform(2.1, 2.8), eps_e=10 ** rng.uniform(-2, -0.7),
              eps_B=10 ** rng.uniform(-4, -1.5),
              resolutions=(float(10 ** rng.uniform(-1.2, 0.1)), float(10 ** rng.uniform(-0.7, 0.5)), float(10 ** rng.uniform(0.8, 1.7))))
    if jet == "PowerLawJet":
        kw.update(k_e=rng.uniform(1.5, 3.0), k_g=rng.uniform(1.5, 3.0))
    if jet == "TwoComponentJet":
        kw.update(theta_w=kw["theta_c"] * rng.uniform(1.5, 3.0), E_iso_w=kw["E_iso"] * 10 ** rng.uniform(-2, -0.5), Gamma0_w=max(20.0, kw["Gamma0"] * rng.uniform(0.1, 0.5)))
    if i % 4 == 1:
        kw.update(ssc=True, kn=bool(i % 8 == 1))
    if i % 5 == 2:
        kw.update(duration=10 ** rng.uniform(0.5, 2.5), rvs=dict(eps_e=10 ** rng.uniform(-2, -0.7), eps_B=10 ** rng.uniform(-3, -1), p=rng.uniform(2.1, 2.7)))
    kws.append(kw); prms.append(_abi.make_params(**kw))

def gpu(idx):
    arr = (_lib.ModelParams * len(idx))(*[_lib.ModelParams.from_buffer_copy(bytes(prms[i])) for i in idx])
    out = np.empty((len(idx), nu.size, t.size))
    _lib.check(lib.vag_flux_density_grid_batch(h, arr, len(idx), t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp), nu.size, out.ctypes.data_as(dp)))
    return out
batch = gpu(list(range(n)))
worst = []
for i in range(n):
    one = gpu([i])[0]
    d = orc.details(prms[i], float(t.min()), float(t.max()))["shape"]
    w = orc.flux_density_grid(prms[i], t, nu)
    sel = w > 1e-3 * w.max()
    e1 = float(np.max(np.abs(one - w)[sel] / w[sel])); eb = float(np.max(np.abs(batch[i] - w)[sel] / w[sel]))
    same = bool(np.array_equal(one, batch[i]))
    worst.append((max(e1, eb), i))
    print(f"#{i:2d} {kws[i]['jet']:16s} grid ({d['n_phi']:4d}, {d['n_theta']:4d}, {d['n_t']:4d}) ssc {int(bool(kws[i].get('ssc')))} rvs {int('rvs' in kws[i])}"
          f"  alone {e1:.2e}  in the batch {eb:.2e}  {'same bits' if same else 'max rel. diff alone/batch %.1e' % float(np.max(np.abs(one - batch[i])[sel] / w[sel]))}", flush=True)
worst.sort(reverse=True)
print("worst:", ["%.2e (#%d)" % x for x in worst[:5]], "non-finite:", int((~np.isfinite(batch)).sum()))
