"""Ad-hoc robustness sweep: forward + reverse shock with SSC / KN on random draws, every component against the checker."""
import os, sys
import numpy as np
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(_ROOT, "tests")); sys.path.insert(0, _ROOT)
import ctypes as C
import _abi, configs
from vegasafterglow_amd import _lib
import vegasafterglow_amd as va

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
lib = _lib.load(); h, lock = va.get_context(0); orc = _abi.load_oracle(); dp = C.POINTER(C.c_double)
rng = np.random.default_rng(int(os.environ.get("SWEEP_SEED", 777)))
t, nu = np.logspace(1.5, 7.5, 36), np.array([1e9, 4.84e14, 1e18, 2.4e24])
prms = []
for i in range(n):
    jet = ["TophatJet", "GaussianJet", "PowerLawJet"][i % 3]
    kw = dict(jet=jet, E_iso=10 ** rng.uniform(51, 53.5), Gamma0=10 ** rng.uniform(1.7, 2.7), theta_c=rng.uniform(0.04, 0.2),
              theta_obs=rng.uniform(0, 0.3), n_ism=10 ** rng.uniform(-2, 1), p=rng.uniform(2.1, 2.7), eps_e=10 ** rng.uniform(-2, -0.7),
              eps_B=10 ** rng.uniform(-4, -1.5), duration=10 ** rng.uniform(0, 3), ssc=True, kn=True,
              rvs=dict(eps_e=10 ** rng.uniform(-2, -0.7), eps_B=10 ** rng.uniform(-3, -1), p=rng.uniform(2.1, 2.7), ssc=True, kn=True))
    if jet == "PowerLawJet":
        kw.update(k_e=2.0, k_g=2.0)
    prms.append(_abi.make_params(**kw))
arr = (_lib.ModelParams * n)(*[_lib.ModelParams.from_buffer_copy(bytes(p)) for p in prms])
comps = [np.empty((n, nu.size, t.size)) for _ in range(4)]
out4 = (dp * 4)(*[a.ctypes.data_as(dp) for a in comps])
_lib.check(lib.vag_flux_density_grid_components4_batch(h, arr, n, t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp), nu.size, out4))
names = ["fwd.sync", "fwd.ssc", "rvs.sync", "rvs.ssc"]
worst = {k: (0.0, -1) for k in names}
allerr = {k: [] for k in names}
outside = []  # draws outside the reference's golden contract
for i, p in enumerate(prms):
    want = orc.flux_components4(p, t, nu)
    for c, k in enumerate(names):
        w, g = want[c], comps[c][i]
        if w.max() <= 0:
            continue
        if not np.all(np.abs(g - w) <= 2e-3 * np.abs(w) + 1e-2 * np.max(np.abs(w))):
            outside.append((k, i))
        sel = w > 1e-2 * w.max()
        e = float(np.max(np.abs(g[sel] / w[sel] - 1)))
        allerr[k].append((e, i))
        if e > worst[k][0]:
            worst[k] = (e, i)
for k in names:
    print(k, ["%.1e (#%d)" % v for v in sorted(allerr[k], reverse=True)[:5]])
print("models", n, {k: "%.2e (#%d)" % v for k, v in worst.items()}, "non-finite", sum(int(np.sum(~np.isfinite(c))) for c in comps),
      "outside the golden contract", outside)
