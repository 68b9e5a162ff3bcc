"""Seeded random draws of wide parameter boxes for the randomised parity tests (tests/test_gpu_parity.py) and for the script that
measures what the reference itself demonstrates on them (tests/golden/make_sweep_fixture.py).  The streams are those of
profiles/debug/prior_sweep_ssc.py and prior_sweep_rs_ssc.py, so draw #i here is draw #i of the sweeps recorded under profiles/."""
import numpy as np

import _abi

SSC_T, SSC_NU = np.logspace(1.5, 7.5, 30), np.array([1e9, 4.84e14, 1e18, 2.4e22, 1e26])
RS_T, RS_NU = np.logspace(1.5, 7.5, 36), np.array([1e9, 4.84e14, 1e18, 2.4e24])


def ssc_draws(n, kn, seed=4242):
    """Forward shock with SSC: the n Klein-Nishina draws (kn=True) or the n Thomson draws that follow them in the stream."""
    rng = np.random.default_rng(seed)
    sets = {}
    for flag in (True, False):
        prms = []
        for i in range(n):
            jet = ["TophatJet", "GaussianJet", "PowerLawJet"][i % 3]
            kw = dict(jet=jet, E_iso=10 ** rng.uniform(50.5, 54), Gamma0=10 ** rng.uniform(1.5, 2.9), theta_c=rng.uniform(0.03, 0.3),
                      theta_obs=rng.uniform(0, 0.5), p=rng.uniform(2.05, 2.9), eps_e=10 ** rng.uniform(-2.5, -0.5),
                      eps_B=10 ** rng.uniform(-6, -1), ssc=True, kn=flag)
            if i % 2:
                kw.update(medium="Wind", A_star=10 ** rng.uniform(-2, 0.5))
            else:
                kw.update(n_ism=10 ** rng.uniform(-3, 2))
            if jet == "PowerLawJet":
                kw.update(k_e=rng.uniform(1.5, 3.0), k_g=rng.uniform(1.5, 3.0))
            prms.append(_abi.make_params(**kw))
        sets[flag] = prms
    return sets[bool(kn)]


def rs_ssc_draws(n, seed=777):
    """Forward + reverse shock, SSC with Klein-Nishina on both."""
    rng = np.random.default_rng(seed)
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
    return prms


NONAXI_T, NONAXI_NU = np.logspace(2.5, 7.5, 28), np.array([1e9, 4.84e14, 1e18])


def nonaxi_spread_draws(n, seed=2024):
    """Model(axisymmetric=False) with a spreading jet -- one lattice and one blast-wave solve per (phi, theta) node -- over the jet
    profiles; every fourth draw carries a reverse shock."""
    rng = np.random.default_rng(seed)
    prms = []
    for i in range(n):
        jet = ["TophatJet", "GaussianJet", "PowerLawJet", "TwoComponentJet"][i % 4]
        kw = dict(jet=jet, E_iso=10 ** rng.uniform(51, 53.5), Gamma0=10 ** rng.uniform(1.7, 2.7), theta_c=rng.uniform(0.05, 0.2),
                  theta_obs=rng.uniform(0, 0.4), n_ism=10 ** rng.uniform(-2, 1), p=rng.uniform(2.1, 2.8), eps_e=10 ** rng.uniform(-2, -0.7),
                  eps_B=10 ** rng.uniform(-4, -1.5), spreading=True, axisymmetric=False)
        if jet == "PowerLawJet":
            kw.update(k_e=rng.uniform(1.5, 3.0), k_g=rng.uniform(1.5, 3.0))
        if jet == "TwoComponentJet":
            kw.update(theta_w=kw["theta_c"] * rng.uniform(1.5, 3.0), E_iso_w=kw["E_iso"] * 10 ** rng.uniform(-2, -0.5),
                      Gamma0_w=max(20.0, kw["Gamma0"] * rng.uniform(0.1, 0.5)))
        if i % 4 == 3:
            kw.update(duration=10 ** rng.uniform(0.5, 2.5), rvs=dict(eps_e=10 ** rng.uniform(-2, -0.7), eps_B=10 ** rng.uniform(-3, -1), p=rng.uniform(2.1, 2.7)))
        prms.append(_abi.make_params(**kw))
    return prms


def spread_ssc_draws(n, seed=4242):
    """Axisymmetric jets with lateral spreading and SSC (Klein-Nishina) over all six jet profiles: the first n draws of
    `SWEEP_MODE=spread python profiles/debug/prior_sweep_ssc.py n` (draw 13 of the 30-draw stream is the Gaussian jet in a dense wind
    whose rows arrive earlier from later nodes: tests/test_gpu_parity.py pins it by name as well)."""
    rng = np.random.default_rng(seed)
    prms = []
    for i in range(n):
        jet = ["TophatJet", "GaussianJet", "PowerLawJet"][i % 3]
        kw = dict(jet=jet, E_iso=10 ** rng.uniform(50.5, 54), Gamma0=10 ** rng.uniform(1.5, 2.9), theta_c=rng.uniform(0.03, 0.3),
                  theta_obs=rng.uniform(0, 0.5), p=rng.uniform(2.05, 2.9), eps_e=10 ** rng.uniform(-2.5, -0.5),
                  eps_B=10 ** rng.uniform(-6, -1), ssc=True, kn=True)
        if i % 2:
            kw.update(medium="Wind", A_star=10 ** rng.uniform(-2, 0.5))
        else:
            kw.update(n_ism=10 ** rng.uniform(-3, 2))
        if jet == "PowerLawJet":
            kw.update(k_e=rng.uniform(1.5, 3.0), k_g=rng.uniform(1.5, 3.0))
        kw["spreading"] = True
        kw["jet"] = ["TophatJet", "GaussianJet", "PowerLawJet", "TwoComponentJet", "StepPowerLawJet", "PowerLawWing"][i % 6]
        if kw["jet"] in ("TwoComponentJet", "StepPowerLawJet", "PowerLawWing"):
            kw.update(theta_w=kw["theta_c"] * rng.uniform(1.5, 3.0), E_iso_w=kw["E_iso"] * 10 ** rng.uniform(-2, -0.5),
                      Gamma0_w=max(20.0, kw["Gamma0"] * rng.uniform(0.1, 0.5)), k_e=rng.uniform(1.5, 3.0), k_g=rng.uniform(1.5, 3.0))
        prms.append(_abi.make_params(**kw))
    return prms


WINDOW_NU = np.array([1e9, 4.84e14, 1e18, 2.4e24])


def ssc_window_models(n, seed=31337):
    """SSC models of every kind the engine serves in one ragged, mixed-flag batch: six jet profiles, ISM / wind, KN / Thomson, every
    fourth with a reverse shock (SSC on both), some with a magnetar, lateral spreading or axisymmetric=False.  Returns (params, tags)."""
    rng = np.random.default_rng(seed)
    prms, tags = [], []
    for i in range(n):
        jet = ["TophatJet", "GaussianJet", "PowerLawJet", "TwoComponentJet", "StepPowerLawJet", "PowerLawWing"][i % 6]
        kw = dict(jet=jet, E_iso=10 ** rng.uniform(50.5, 54), Gamma0=10 ** rng.uniform(1.5, 2.9), theta_c=rng.uniform(0.03, 0.3),
                  theta_obs=rng.uniform(0, 0.6) if i % 5 else 0.0, p=rng.uniform(2.05, 2.9), eps_e=10 ** rng.uniform(-2.5, -0.5),
                  eps_B=10 ** rng.uniform(-6, -1), ssc=True, kn=bool(i % 2))
        if i % 3 == 1:
            kw.update(medium="Wind", A_star=10 ** rng.uniform(-2, 0.5))
        else:
            kw.update(n_ism=10 ** rng.uniform(-3, 2))
        if jet in ("PowerLawJet", "StepPowerLawJet", "PowerLawWing"):
            kw.update(k_e=rng.uniform(1.5, 3.0), k_g=rng.uniform(1.5, 3.0))
        if jet in ("TwoComponentJet", "StepPowerLawJet", "PowerLawWing"):
            kw.update(theta_w=kw["theta_c"] * rng.uniform(1.5, 3.0), E_iso_w=kw["E_iso"] * 10 ** rng.uniform(-2, -0.5),
                      Gamma0_w=max(20.0, kw["Gamma0"] * rng.uniform(0.1, 0.5)))
        tag = jet
        if i % 4 == 2:
            kw.update(duration=10 ** rng.uniform(0, 3), rvs=dict(eps_e=10 ** rng.uniform(-2, -0.7), eps_B=10 ** rng.uniform(-3, -1), p=rng.uniform(2.1, 2.7),
                                                                 ssc=True, kn=bool(i % 2)))
            tag += "+rvs"
        if i % 7 == 3 and jet in ("TophatJet", "GaussianJet", "PowerLawJet"):
            kw["magnetar"] = (10 ** rng.uniform(45, 48), 10 ** rng.uniform(2, 4), rng.uniform(1.5, 2.5))
            tag += "+magnetar"
        if i % 9 == 4:
            kw["spreading"] = True
            tag += "+spreading"
        if i % 11 == 5:
            kw["axisymmetric"] = False
            tag += "+nonaxi"
        prms.append(_abi.make_params(**kw))
        tags.append(tag)
    return prms, tags


def narrow_windows(nw, seed=4711):
    """Request windows that leave most cells of a model unqueried: 1 ... 12 times inside 0.05 ... 2 decades anywhere between 30 s and 3e8 s."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(nw):
        lo = 10 ** rng.uniform(1.5, 8.0)
        span = 10 ** rng.uniform(-1.3, 0.3)
        k = int(rng.integers(1, 13))
        out.append(np.sort(lo * 10 ** (span * rng.random(k))) if k > 1 else np.array([lo]))
    return out
