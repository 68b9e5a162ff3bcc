"""Ad-hoc robustness sweep: GPU vs checker on random draws of the C4 prior box (series + grid forms), worst cases printed."""
import os, sys
import numpy as np
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(_ROOT, "tests")); sys.path.insert(0, _ROOT)
import ctypes as C
import _abi, configs
from vegasafterglow_amd import _lib
import vegasafterglow_amd as va

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib = _lib.load(); h, lock = va.get_context(0); orc = _abi.load_oracle(); dp = C.POINTER(C.c_double)
rng = np.random.default_rng(int(os.environ.get("SWEEP_SEED", 12345)))
t, nu = configs.c4_mock_data()
tg, nug = np.logspace(4.5, 8, 40), np.array([3e9, 5.06e14, 2.41e17])
prms = []
for _ in range(n):
    kw = dict(configs.C4_TRUTH, jet="GaussianJet")
    kw.update(E_iso=10 ** rng.uniform(50, 54), Gamma0=10 ** rng.uniform(1.5, 3), theta_c=rng.uniform(0.02, 0.3), theta_obs=rng.uniform(0, 0.8),
              n_ism=10 ** rng.uniform(-4, 1), p=rng.uniform(2.05, 2.8), eps_e=10 ** rng.uniform(-3, -0.5), eps_B=10 ** rng.uniform(-5, -1))
    prms.append(_abi.make_params(**kw))
arr = (_lib.ModelParams * n)(*[_lib.ModelParams.from_buffer_copy(bytes(p)) for p in prms])
out = np.empty((n, t.size)); outg = np.empty((n, nug.size, tg.size))
_lib.check(lib.vag_flux_density_batch(h, arr, n, t.ctypes.data_as(dp), nu.ctypes.data_as(dp), t.size, out.ctypes.data_as(dp)))
_lib.check(lib.vag_flux_density_grid_batch(h, arr, n, tg.ctypes.data_as(dp), tg.size, nug.ctypes.data_as(dp), nug.size, outg.ctypes.data_as(dp)))
worst = []
contract = lambda g, w: bool(np.all(np.abs(g - w) <= 2e-3 * np.abs(w) + 1e-2 * np.max(np.abs(w))))  # the reference's golden contract
outside = []
for i, p in enumerate(prms):
    ws = orc.flux_density(p, t, nu); wg = orc.flux_density_grid(p, tg, nug)
    def rel(g, w):
        sel = w > 1e-3 * w.max()
        return np.max(np.abs(g[sel] / w[sel] - 1)) if sel.any() else 0.0
    worst.append((max(rel(out[i], ws), rel(outg[i], wg)), i))
    if not (contract(out[i], ws) and contract(outg[i], wg)):
        outside.append(i)
worst.sort(reverse=True)
print("models", n, "worst rel err (bins > 1e-3 peak):", ["%.2e (#%d)" % w for w in worst[:6]], "median %.2e" % np.median([w[0] for w in worst]))
print("outside the golden contract:", outside)
print("non-finite:", int(np.sum(~np.isfinite(out))), int(np.sum(~np.isfinite(outg))))
