"""Ad-hoc robustness sweep: forward shock with SSC (Thomson and Klein-Nishina) on random draws of a wide box, both components
against the checker.  usage: [SWEEP_MODE=spread|magnetar|nonaxi] python profiles/debug/prior_sweep_ssc.py [n]
(SWEEP_MODE widens the jets: two-component / step-power-law / wing profiles with lateral spreading, a magnetar, or
axisymmetric=False)"""
import os, sys
import numpy as np
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(_ROOT, "tests")); sys.path.insert(0, _ROOT)
import ctypes as C
import _abi
from vegasafterglow_amd import _lib
import vegasafterglow_amd as va

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
lib = _lib.load(); h, lock = va.get_context(0); orc = _abi.load_oracle(); dp = C.POINTER(C.c_double)
rng = np.random.default_rng(int(os.environ.get("SWEEP_SEED", 4242)))
t, nu = np.logspace(1.5, 7.5, 30), np.array([1e9, 4.84e14, 1e18, 2.4e22, 1e26])
for kn in (True, False):
    worst = {"sync": (0.0, -1), "ssc": (0.0, -1)}  # per pass (until round 5 the Thomson line repeated the Klein-Nishina maximum: VERDICT r05)
    bad = 0
    outside = []  # draws outside the reference's golden contract
    prms = []
    for i in range(n):
        jet = ["TophatJet", "GaussianJet", "PowerLawJet"][i % 3]
        kw = dict(jet=jet, E_iso=10 ** rng.uniform(50.5, 54), Gamma0=10 ** rng.uniform(1.5, 2.9), theta_c=rng.uniform(0.03, 0.3),
                  theta_obs=rng.uniform(0, 0.5), p=rng.uniform(2.05, 2.9), eps_e=10 ** rng.uniform(-2.5, -0.5),
                  eps_B=10 ** rng.uniform(-6, -1), ssc=True, kn=kn)
        if i % 2:
            kw.update(medium="Wind", A_star=10 ** rng.uniform(-2, 0.5))
        else:
            kw.update(n_ism=10 ** rng.uniform(-3, 2))
        if jet == "PowerLawJet":
            kw.update(k_e=rng.uniform(1.5, 3.0), k_g=rng.uniform(1.5, 3.0))
        mode = os.environ.get("SWEEP_MODE", "")
        if mode == "spread":
            kw["spreading"] = True
            kw["jet"] = ["TophatJet", "GaussianJet", "PowerLawJet", "TwoComponentJet", "StepPowerLawJet", "PowerLawWing"][i % 6]
            if kw["jet"] in ("TwoComponentJet", "StepPowerLawJet", "PowerLawWing"):
                kw.update(theta_w=kw["theta_c"] * rng.uniform(1.5, 3.0), E_iso_w=kw["E_iso"] * 10 ** rng.uniform(-2, -0.5),
                          Gamma0_w=max(20.0, kw["Gamma0"] * rng.uniform(0.1, 0.5)), k_e=rng.uniform(1.5, 3.0), k_g=rng.uniform(1.5, 3.0))
        elif mode == "magnetar":
            kw["magnetar"] = (10 ** rng.uniform(45, 48), 10 ** rng.uniform(2, 4), rng.uniform(1.5, 2.5))
        elif mode == "nonaxi":
            kw["axisymmetric"] = False
        prms.append(_abi.make_params(**kw))
    arr = (_lib.ModelParams * n)(*[_lib.ModelParams.from_buffer_copy(bytes(p)) for p in prms])
    s, c = np.empty((n, nu.size, t.size)), np.empty((n, nu.size, t.size))
    _lib.check(lib.vag_flux_density_grid_components_batch(h, arr, n, t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp), nu.size,
                                                          s.ctypes.data_as(dp), c.ctypes.data_as(dp)))
    for i, prm in enumerate(prms):
        o_s, o_c = orc.flux_components(prm, t, nu)
        for name, g, o in (("sync", s[i], o_s), ("ssc", c[i], o_c)):
            if not (np.all(np.isfinite(g)) and np.all(np.isfinite(o))):
                bad += 1
                continue
            if not np.all(np.abs(g - o) <= 2e-3 * np.abs(o) + 1e-2 * np.max(np.abs(o))):
                outside.append((name, i + (0 if kn else 1000)))
            m = o > 1e-6 * o.max() if o.max() > 0 else np.zeros_like(o, bool)
            if m.any():
                e = float(np.max(np.abs(g - o)[m] / o[m]))
                if e > worst[name][0]:
                    worst[name] = (e, i + (0 if kn else 1000))
    print("kn" if kn else "thomson", {k: "%.2e (#%d)" % v for k, v in worst.items()}, "non-finite", bad, "outside the golden contract", outside, flush=True)
    if os.environ.get("SWEEP_DETAIL"):
        for i in [int(x) for x in os.environ["SWEEP_DETAIL"].split(",")]:
            o_s, o_c = orc.flux_components(prms[i], t, nu)
            for name, g, o in (("sync", s[i], o_s), ("ssc", c[i], o_c)):
                e = np.abs(g - o) / np.where(o > 0, o, 1)
                k = np.unravel_index(np.argmax(np.where(o > 1e-6 * o.max(), e, 0)), e.shape)
                print(f"  #{i} {name}: worst bin nu[{k[0]}] t[{k[1]}] rel {e[k]:.2e} at {o[k] / o.max():.2e} of the peak; "
                      f"worst over bins > 1e-3 peak {np.max(np.where(o > 1e-3 * o.max(), e, 0)):.2e}")
