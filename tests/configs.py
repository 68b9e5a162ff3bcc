"""The BASELINE.json / SURVEY.md section 8(d) configurations as plain keyword dictionaries."""
import numpy as np

DAY = 86400.0

C1_T = np.logspace(2, 8, 100)
C1_NU = np.array([1e9, 4.84e14, 1e18])
C1A = dict(jet="TophatJet", theta_c=0.1, E_iso=1e52, Gamma0=300.0, medium="ISM", n_ism=1.0, lumi_dist=1e28, z=1.0,
           theta_obs=0.0, eps_e=0.1, eps_B=0.01, p=2.3, resolutions=(0.089, 0.05, 12.0))
C1B = dict(C1A, theta_obs=0.05)

C2_T = np.logspace(2, 8, 200)
C2_NU = np.logspace(9, 18, 10)
C2 = dict(jet="GaussianJet", theta_c=0.1, E_iso=1e52, Gamma0=300.0, medium="ISM", n_ism=1.0, lumi_dist=1e28, z=1.0,
          theta_obs=0.3, eps_e=0.1, eps_B=0.01, p=2.3, resolutions=(0.355, 0.31, 20.5))

# C4: GW170817-like mock: truth parameters, 3 bands x 20 epochs
C4_TRUTH = dict(jet="GaussianJet", theta_c=0.07, E_iso=10 ** 52.4, Gamma0=500.0, medium="ISM", n_ism=1e-2,
                lumi_dist=1.23e26, z=0.0098, theta_obs=0.4, eps_e=10 ** -1.5, eps_B=10 ** -3.5, p=2.15)
C4_BANDS = np.array([3e9, 5.06e14, 2.41e17])
C4_EPOCHS = np.geomspace(9 * DAY, 1000 * DAY, 20)
# free parameters: (name, is_log, lower, upper)
C4_FREE = [("E_iso", 1, 50.0, 54.0), ("Gamma0", 1, 1.5, 3.0), ("theta_c", 0, 0.02, 0.3), ("theta_v", 0, 0.0, 0.8),
           ("n_ism", 1, -4.0, 1.0), ("p", 0, 2.05, 2.8), ("eps_e", 1, -3.0, -0.5), ("eps_B", 1, -5.0, -1.0)]

EXTRA = {
    "powerlaw_wind": (dict(jet="PowerLawJet", medium="Wind", theta_c=0.1, E_iso=1e52, Gamma0=300.0, k_e=2.0, k_g=2.0,
                           A_star=0.1, n_ism=0.0, theta_obs=0.2, resolutions=(0.29, 0.16, 10.0)),
                      np.logspace(2, 8, 60), np.array([1e9, 4.84e14, 1e18])),
    "tophat_wind_offaxis": (dict(jet="TophatJet", medium="Wind", A_star=0.1, n_ism=0.0, theta_obs=0.2),
                            np.logspace(2, 8, 30), np.array([4.84e14])),
    "two_component": (dict(jet="TwoComponentJet", theta_c=0.05, theta_w=0.3, E_iso_w=1e50, Gamma0_w=50.0, theta_obs=0.15,
                           resolutions=(0.2, 0.3, 8.0)), np.logspace(2, 8, 60), np.array([1e9, 4.84e14, 1e18])),
    "gaussian_p_below_2": (dict(jet="GaussianJet", p=1.8, theta_obs=0.1), np.logspace(2, 8, 30), np.array([1e9, 1e15])),
    "adiabatic": (dict(jet="TophatJet", radiative_fireball=False, theta_obs=0.0), np.logspace(2, 8, 30), np.array([1e9, 1e15])),
    "narrow_window": (dict(jet="GaussianJet", theta_obs=0.3), np.geomspace(5 * DAY, 50 * DAY, 12), np.array([3e9, 5e14])),
}


def c4_mock_data():
    """Mock data set of SURVEY 8(d) C4: returns sorted (t, nu, flux placeholder built by caller)."""
    t = np.concatenate([C4_EPOCHS] * 3)
    nu = np.concatenate([np.full(20, b) for b in C4_BANDS])
    order = np.argsort(t, kind="stable")
    return t[order], nu[order]


# BASELINE configs[2] as pinned by SURVEY section 8(d): forward + reverse shock, both with SSC + Klein-Nishina
C3 = dict(jet="PowerLawJet", medium="Wind", theta_c=0.1, E_iso=1e52, Gamma0=300.0, k_e=2.0, k_g=2.0, duration=1.0,
          A_star=0.1, n_ism=0.0, lumi_dist=1e28, z=1.0, theta_obs=0.2, eps_e=0.1, eps_B=0.01, p=2.3, ssc=True, kn=True,
          rvs=dict(eps_e=0.1, eps_B=0.01, p=2.3, ssc=True, kn=True), resolutions=(0.29, 0.16, 10.0))
C3_T = np.logspace(2, 8, 100)
C3_NU = np.array([1e9, 4.84e14, 1e18, 2.4e26])

# reverse-shock cases for live/fixture comparisons (thin and thick shells, structured jets, RS with its own SSC)
RS_CASES = {
    "rs_thin_tophat": (dict(jet="TophatJet", theta_obs=0.0, rvs=dict(eps_e=0.1, eps_B=0.01, p=2.3)),
                       np.logspace(0, 7, 48), np.array([1e9, 4.84e14, 1e18])),
    "rs_thick_offaxis": (dict(jet="TophatJet", E_iso=1e53, Gamma0=100.0, duration=1000.0, theta_obs=0.15, z=0.5,
                              lumi_dist=3e28, eps_B=1e-3, rvs=dict(eps_e=0.1, eps_B=0.01, p=2.5)),
                         np.logspace(1, 7, 40), np.array([1e9, 4.84e14, 1e18])),
    "rs_two_component": (dict(jet="TwoComponentJet", theta_c=0.05, theta_w=0.3, E_iso_w=1e50, Gamma0_w=50.0,
                              theta_obs=0.15, rvs=dict(eps_e=0.1, eps_B=0.01, p=2.3)),
                         np.logspace(1, 7, 40), np.array([1e9, 1e14, 1e17])),
    "rs_gaussian_adiabatic": (dict(jet="GaussianJet", theta_obs=0.2, radiative_fireball=False, duration=10.0,
                                   rvs=dict(eps_e=0.05, eps_B=0.1, p=2.2, xi_e=0.5), resolutions=(0.1, 0.3, 8.0)),
                              np.logspace(1, 7, 40), np.array([1e9, 1e14, 1e17])),
    "rs_tophat_both_ssc_kn": (dict(jet="TophatJet", duration=100.0, ssc=True, kn=True,
                                   rvs=dict(eps_e=0.1, eps_B=0.01, p=2.3, ssc=True, kn=True)),
                              np.logspace(1, 7, 40), np.array([1e9, 1e14, 1e17, 1e23])),
}


# spreading jets (jet(..., spreading=True)): Symmetry::structured grids, lateral expansion, per-cell solid angles
SPREAD_CASES = {
    "tophat_spread_onaxis": dict(spreading=True),
    "tophat_spread_offaxis": dict(spreading=True, theta_obs=0.2),
    "gauss_spread": dict(jet="GaussianJet", spreading=True, theta_obs=0.15),
    "powerlaw_wind_spread": dict(jet="PowerLawJet", medium="Wind", A_star=0.1, n_ism=0.0, spreading=True, theta_obs=0.3),
    "two_comp_spread": dict(jet="TwoComponentJet", theta_c=0.05, theta_w=0.3, E_iso_w=1e50, Gamma0_w=50.0, theta_obs=0.15,
                            spreading=True),
    "tophat_spread_rs": dict(spreading=True, theta_obs=0.1, rvs=dict(eps_e=0.1, eps_B=0.01, p=2.3)),
    "gauss_spread_ssc": dict(jet="GaussianJet", spreading=True, theta_obs=0.15, ssc=True, kn=True),
}
SPREAD_T = np.logspace(2, 8, 40)
SPREAD_NU = np.array([1e9, 4.84e14, 1e18])


# the remaining closed-form profiles of the reference's registry (fitting/config.py:99-137): step power law, power-law wing,
# and a wind with a general density slope k_m (a generic Medium in the reference)
PROFILE_CASES = {
    "step_powerlaw": dict(jet="StepPowerLawJet", theta_c=0.05, E_iso_w=3e51, Gamma0_w=100.0, k_e=3.0, k_g=2.0, theta_obs=0.2),
    "powerlaw_wing": dict(jet="PowerLawWing", theta_c=0.05, E_iso_w=3e51, Gamma0_w=100.0, k_e=3.0, k_g=2.0, theta_obs=0.2),
    "step_powerlaw_rs_spread": dict(jet="StepPowerLawJet", theta_c=0.05, E_iso_w=3e51, Gamma0_w=100.0, k_e=3.0, k_g=2.0,
                                    theta_obs=0.1, spreading=True, rvs=dict(eps_e=0.1, eps_B=0.01, p=2.3)),
    "wind_k1.5": dict(medium="Wind", A_star=0.3, n_ism=0.0, k_m=1.5, theta_obs=0.1),
    "wind_k2.5_floor": dict(jet="GaussianJet", medium="Wind", A_star=0.1, n_ism=1e-3, n0=1e3, k_m=2.5, theta_obs=0.2),
    "wind_k1_rs_ssc": dict(medium="Wind", A_star=0.1, n_ism=0.0, k_m=1.0, ssc=True, rvs=dict(eps_e=0.1, eps_B=0.01, p=2.3)),
}

MAG = (1e47, 1e4, 2.0)
MAGNETAR_CASES = {
    "tophat_mag": dict(magnetar=MAG),
    "gauss_mag_offaxis": dict(jet="GaussianJet", theta_obs=0.2, magnetar=MAG),
    "powerlaw_mag_wind": dict(jet="PowerLawJet", medium="Wind", A_star=0.1, n_ism=0.0, theta_obs=0.2, magnetar=(3e46, 3e3, 1.5)),
    "two_comp_mag": dict(jet="TwoComponentJet", theta_c=0.05, theta_w=0.3, E_iso_w=1e50, Gamma0_w=50.0, theta_obs=0.15, magnetar=MAG),
    "step_mag_spread": dict(jet="StepPowerLawJet", theta_c=0.05, E_iso_w=3e51, Gamma0_w=100.0, k_e=3.0, k_g=2.0, theta_obs=0.1,
                            spreading=True, magnetar=MAG),
    "tophat_mag_rs": dict(magnetar=MAG, duration=100.0, rvs=dict(eps_e=0.1, eps_B=0.01, p=2.3)),
    "tophat_mag_ssc": dict(magnetar=MAG, ssc=True, kn=True),
}

# Model(axisymmetric=False): full-circle phi grid with every phi node observed
NONAXI_CASES = {
    "tophat_onaxis_3d": dict(axisymmetric=False),
    "tophat_offaxis_3d": dict(theta_obs=0.3, axisymmetric=False),
    "gauss_offaxis_3d": dict(jet="GaussianJet", theta_obs=0.2, axisymmetric=False),
    "two_comp_onaxis_3d": dict(jet="TwoComponentJet", theta_w=0.3, E_iso_w=1e51, Gamma0_w=50.0, axisymmetric=False),
    "tophat_ssc_3d": dict(theta_obs=0.1, ssc=True, axisymmetric=False),
    "tophat_rs_3d": dict(theta_obs=0.1, duration=100.0, rvs=dict(eps_e=0.1, eps_B=0.01, p=2.3), axisymmetric=False),
    "powerlaw_wind_one_phi_3d": dict(jet="PowerLawJet", medium="Wind", A_star=0.1, n_ism=0.0, theta_obs=0.02,
                                     resolutions=(0.005, 0.5, 5.0), axisymmetric=False),
}


# The timed ensemble workloads of bench.py (BASELINE configs[2] / configs[4]); also what the full-size parity tests and the
# reference-spread fixtures (tests/golden/make_spread_fixture.py) evaluate.  Seeded: every caller, and every rank, builds the same models.
def c5_batch(nb, seed=1):
    """BASELINE configs[4]: prior-predictive sweep of two-component jets, SSC on, resolutions (0.59, 0.98, 12) -> ~128 x 128 x 111 cells."""
    import _abi
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(nb):
        out.append(_abi.make_params(jet="TwoComponentJet", theta_c=rng.uniform(0.03, 0.1), E_iso=10 ** rng.uniform(51, 53),
                                    Gamma0=rng.uniform(100, 500), theta_w=rng.uniform(0.2, 0.5),
                                    E_iso_w=10 ** rng.uniform(49, 51), Gamma0_w=rng.uniform(20, 100), n_ism=1.0,
                                    lumi_dist=1e28, z=1.0, theta_obs=0.15, eps_e=0.1, eps_B=0.01, p=2.3, ssc=True,
                                    resolutions=(0.59, 0.98, 12.0)))
    return out


def c3_batch(nb, seed=3):
    """BASELINE configs[2] (C3 above) with +-10 % log-uniform jitter on the jet and microphysics parameters of both shocks."""
    import _abi
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(nb):
        kw = dict(C3)
        j = lambda: float(np.exp(rng.uniform(np.log(0.9), np.log(1.1))))
        kw.update(theta_c=kw["theta_c"] * j(), E_iso=kw["E_iso"] * j(), Gamma0=kw["Gamma0"] * j(), A_star=kw["A_star"] * j(),
                  eps_e=kw["eps_e"] * j(), eps_B=kw["eps_B"] * j(), p=2.3 + rng.uniform(-0.1, 0.1))
        kw["rvs"] = dict(kw["rvs"], eps_B=kw["rvs"]["eps_B"] * j(), p=2.3 + rng.uniform(-0.1, 0.1))
        out.append(_abi.make_params(**kw))
    return out


# Grids beyond what the grid kernel's two LDS layouts hold (round 6): the reference sizes its grids freely (grid-refinement.h:639-706).
# tests/golden/make_big_grid_fixture.py runs the reference itself on them; the third needs ~2 minutes of CPU, hence a fixture.
BIG_GRID_T = np.logspace(3, 7, 6)
BIG_GRID_NU = np.array([1e9, 1e15])
BIG_GRID_CASES = {
    "theta_2000": dict(jet="GaussianJet", theta_c=0.1, E_iso=1e52, Gamma0=300.0, medium="ISM", n_ism=1.0, theta_obs=0.0, resolutions=(0.1, 25.0, 5.0)),
    "time_10000": dict(C1A, resolutions=(0.1, 0.5, 1600.0)),
    "theta_2000_time_10000": dict(jet="GaussianJet", theta_c=0.1, E_iso=1e52, Gamma0=300.0, medium="ISM", n_ism=1.0, theta_obs=0.0,
                                  resolutions=(0.1, 25.0, 1600.0)),
    "phi_3000_offaxis": dict(jet="GaussianJet", theta_c=0.1, E_iso=1e52, Gamma0=300.0, medium="ISM", n_ism=1.0, theta_obs=0.3, resolutions=(17.0, 0.31, 5.0)),
}
