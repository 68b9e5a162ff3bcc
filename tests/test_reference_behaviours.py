"""Behaviours the reference's own Python suites assert through its user API, asserted here through the mirrored API on the
device engine (no checker involved: these are invariants and closure relations, not parity numbers).

Sources of the properties: tests/python/test_physics_invariants.py:37-108 (exact invariants),
tests/python/test_closure_relations.py:37-175 (temporal / spectral indices), tests/python/test_features.py:309-375
(radiative fireball), tests/python/test_advanced.py:23-55,84-130 (exposure averaging, Model properties).  The engine
evaluates logs / softplus / exp2 through tables and polynomials of ~1e-10 relative accuracy, so the "exact" invariants
are held to 1e-8 here (the reference allows 1e-6 for its own fast-math build, 1e-9 for exact libm).
"""
import numpy as np
import pytest

import vegasafterglow_amd as va
from vegasafterglow_amd import ISM, Magnetar, Model, Observer, Radiation, TophatJet, Wind

pytestmark = pytest.mark.gpu

EXACT_RTOL = 1e-8
P = 2.5
T = np.logspace(3.5, 5.5, 16)
NU = np.full_like(T, 1e15)


def _model(lumi_dist=3e28, z=0.5, axisymmetric=True, ssc=False, rvs=False):
    kw = {"rvs_rad": Radiation(eps_e=0.1, eps_B=0.01, p=P)} if rvs else {}
    return Model(jet=TophatJet(theta_c=0.3, E_iso=1e53, Gamma0=300), medium=ISM(n_ism=1.0),
                 observer=Observer(lumi_dist=lumi_dist, z=z, theta_obs=0.0),
                 fwd_rad=Radiation(eps_e=0.1, eps_B=1e-3, p=P, ssc=ssc), axisymmetric=axisymmetric, **kw)


def _closure(medium, p=2.5, eps_B=1e-3, theta_c=0.3, jet_kw=None, **rad_kw):
    jet_args = {"theta_c": theta_c, "E_iso": 1e53, "Gamma0": 300, **(jet_kw or {})}
    return Model(jet=TophatJet(**jet_args), medium=medium, observer=Observer(lumi_dist=3e28, z=0.5, theta_obs=0.0),
                 fwd_rad=Radiation(eps_e=0.1, eps_B=eps_B, p=p, **rad_kw))


def _slope(x, y):
    return np.polyfit(np.log10(x), np.log10(np.asarray(y)), 1)[0]


# ---------------------------------------------------------------------------------------------------------------
# exact invariants
# ---------------------------------------------------------------------------------------------------------------
def test_inverse_square_law_and_redshift_transformation():
    f1 = _model(lumi_dist=3e28).flux_density(T, NU).total
    f2 = _model(lumi_dist=6e28).flux_density(T, NU).total
    np.testing.assert_allclose(f1 / f2, 4.0, rtol=EXACT_RTOL)
    z1, z2 = 0.2, 1.4
    scale = (1 + z2) / (1 + z1)
    moved = _model(z=z1).flux_density(T / scale, NU * scale).total * scale
    np.testing.assert_allclose(moved, _model(z=z2).flux_density(T, NU).total, rtol=EXACT_RTOL)


def test_total_is_the_sum_of_components_and_disabled_components_are_zero():
    f = _model(ssc=True, rvs=True).flux_density(T, NU)
    parts = f.fwd.sync + f.fwd.ssc + f.rvs.sync + f.rvs.ssc
    np.testing.assert_allclose(f.total, parts, rtol=1e-12)
    assert np.all(np.isfinite(f.total)) and np.all(f.total > 0)
    g = _model().flux_density(T, NU)
    assert np.all(g.fwd.ssc == 0) and np.all(g.rvs.sync == 0) and np.all(g.rvs.ssc == 0) and np.all(g.fwd.sync > 0)


def test_full_3d_integration_equals_the_axisymmetric_path_on_axis():
    fa = _model(axisymmetric=True).flux_density(T, NU).total
    fb = _model(axisymmetric=False).flux_density(T, NU).total
    np.testing.assert_allclose(fa, fb, rtol=EXACT_RTOL)


def test_series_grid_and_band_evaluations_agree():
    m = _model()
    a = m.flux_density(T, NU).total
    b = m.flux_density_grid(T, np.array([1e15])).total[0]
    np.testing.assert_allclose(a, b, rtol=EXACT_RTOL)
    tb = np.logspace(3, 5, 8)
    band = m.flux(tb, 1e14, 1e15, 16).total
    nu_fine = np.logspace(14, 15, 60)
    grid = m.flux_density_grid(tb, nu_fine).total
    np.testing.assert_allclose(band, np.trapezoid(grid, nu_fine, axis=0), rtol=1e-2)


# ---------------------------------------------------------------------------------------------------------------
# closure relations (tolerances as calibrated in the reference suite)
# ---------------------------------------------------------------------------------------------------------------
T_MID = np.logspace(4.0, 5.5, 24)
NU_OPT = np.full_like(T_MID, 1e15)


def test_temporal_indices_follow_the_closure_relations():
    alpha = -_slope(T_MID, _closure(ISM(n_ism=1.0)).flux_density(T_MID, NU_OPT).total)
    assert abs(alpha - 3 * (P - 1) / 4) < 0.08
    alpha = -_slope(T_MID, _closure(ISM(n_ism=1.0), p=2.2).flux_density(T_MID, NU_OPT).total)
    assert abs(alpha - 0.9) < 0.1
    m = _closure(ISM(n_ism=1.0), eps_B=0.1)
    alpha = -_slope(T_MID, m.flux_density(T_MID, np.full_like(T_MID, 1e19)).total)
    assert abs(alpha - (3 * P - 2) / 4) < 0.15
    alpha = -_slope(T_MID, _closure(Wind(A_star=0.1)).flux_density(T_MID, NU_OPT).total)
    assert abs(alpha - (3 * P - 1) / 4) < 0.08


def test_spectral_indices_follow_the_closure_relations():
    m = _closure(ISM(n_ism=1.0))
    nu = np.logspace(14, 16, 10)
    beta = -_slope(nu, m.flux_density_grid(np.array([3e4]), nu).total[:, 0])
    assert abs(beta - (P - 1) / 2) < 0.08
    nu = np.logspace(18, 20, 8)
    beta = -_slope(nu, _closure(ISM(n_ism=1.0), eps_B=0.1).flux_density_grid(np.array([3e4]), nu).total[:, 0])
    assert abs(beta - P / 2) < 0.1
    nu = np.logspace(10.4, 13.5, 30)        # between nu_a and past nu_m: rises near nu^(1/3), flattens across nu_m
    F = m.flux_density_grid(np.array([1e5]), nu).total[:, 0]
    local = np.gradient(np.log10(F), np.log10(nu))
    assert 0.15 < np.max(local) < 0.45 and local[0] > local[-1]


def test_jet_break_magnetar_and_off_axis_trends():
    m = _closure(ISM(n_ism=1.0), theta_c=0.05)
    t_pre, t_post = np.logspace(3.3, 3.8, 10), np.logspace(6.0, 6.7, 10)
    a_pre = -_slope(t_pre, m.flux_density(t_pre, np.full(10, 1e15)).total)
    a_post = -_slope(t_post, m.flux_density(t_post, np.full(10, 1e15)).total)
    assert a_post - a_pre > 0.7
    t = np.logspace(3.5, 4.5, 12)
    nu = np.full_like(t, 1e15)
    a0 = -_slope(t, _closure(ISM(n_ism=1.0), jet_kw={"E_iso": 1e52}).flux_density(t, nu).total)
    am = -_slope(t, _closure(ISM(n_ism=1.0), jet_kw={"E_iso": 1e52, "magnetar": Magnetar(L0=1e48, t0=1e4, q=2)})
                 .flux_density(t, nu).total)
    assert am < a0 - 0.1
    t = np.logspace(3, 4, 8)
    nu = np.full_like(t, 1e15)
    on = _closure(ISM(n_ism=1.0), theta_c=0.1).flux_density(t, nu).total
    off = Model(jet=TophatJet(theta_c=0.1, E_iso=1e53, Gamma0=300), medium=ISM(n_ism=1.0),
                observer=Observer(lumi_dist=3e28, z=0.5, theta_obs=0.4),
                fwd_rad=Radiation(eps_e=0.1, eps_B=1e-3, p=2.5)).flux_density(t, nu).total
    assert np.all(off < on)


def test_thick_shell_reverse_shock_peak_and_ssc_fraction():
    T_dur, z = 1000.0, 0.5
    m = Model(jet=TophatJet(theta_c=0.3, E_iso=1e53, Gamma0=100, duration=T_dur), medium=ISM(n_ism=1.0),
              observer=Observer(lumi_dist=3e28, z=z, theta_obs=0.0), fwd_rad=Radiation(eps_e=0.1, eps_B=1e-3, p=2.5),
              rvs_rad=Radiation(eps_e=0.1, eps_B=1e-2, p=2.5))
    t = np.logspace(1, 6, 60)
    rvs = m.flux_density(t, np.full_like(t, 1e14)).rvs.sync
    assert 0.3 < t[int(np.argmax(rvs))] / (T_dur * (1 + z)) < 3.0
    t = np.logspace(4, 5, 8)
    nu = np.full_like(t, 1e24)

    def ssc_ratio(eps_B):
        f = _closure(ISM(n_ism=1.0), eps_B=eps_B, ssc=True).flux_density(t, nu)
        return np.max(f.fwd.ssc) / np.max(f.fwd.sync)

    lo, hi = ssc_ratio(1e-2), ssc_ratio(1e-4)
    assert hi > 10 * lo and lo > 1


# ---------------------------------------------------------------------------------------------------------------
# radiative fireball switch, exposure averaging, Model properties
# ---------------------------------------------------------------------------------------------------------------
def _fireball(radiative, eps_e=0.3, **kw):
    return Model(TophatJet(theta_c=0.1, E_iso=1e53, Gamma0=300), ISM(n_ism=1.0), Observer(lumi_dist=3e28, z=0.5, theta_obs=0.0),
                 Radiation(eps_e=eps_e, eps_B=0.01, p=2.3), **({} if radiative is None else {"radiative_fireball": radiative}), **kw)


def test_radiative_fireball_switch():
    t = np.logspace(4, 8, 15)
    nu = np.full_like(t, 1e15)
    f_rad, f_ad = _fireball(True).flux_density(t, nu).total, _fireball(False).flux_density(t, nu).total
    assert np.all(f_ad >= f_rad * 0.999) and f_ad[-1] > 1.2 * f_rad[-1]
    assert np.array_equal(_fireball(None).flux_density(t, nu).total, f_rad)   # the default is radiative

    def gamma_slope(radiative, eps_e, p):
        m = Model(TophatJet(theta_c=0.1, E_iso=1e53, Gamma0=300), ISM(n_ism=100.0), Observer(lumi_dist=1e28, z=0.0, theta_obs=0.0),
                  Radiation(eps_e=eps_e, eps_B=0.01, p=p, xi_e=1.0), radiative_fireball=radiative)
        d = m.details(1e-1, 1e9)
        G, r = np.asarray(d.fwd.Gamma)[0, 0, :], np.asarray(d.fwd.r)[0, 0, :]
        w = (G > 15) & (G < 60)
        return float(np.median(np.gradient(np.log(G), np.log(r))[w]))

    assert -1.6 < gamma_slope(False, 0.9, 2.3) < -1.3      # Blandford-McKee
    assert gamma_slope(True, 1.0, 1.9) < -2.2              # fully radiative limit is much steeper


def test_exposure_average_is_close_to_the_instantaneous_flux():
    m = Model(TophatJet(theta_c=0.1, E_iso=1e52, Gamma0=300), ISM(n_ism=1.0), Observer(lumi_dist=1e28, z=1.0, theta_obs=0.0),
              Radiation(eps_e=0.1, eps_B=0.01, p=2.2))
    t = np.logspace(3, 6, 10)
    nu = np.full_like(t, 4.84e14)
    avg = m.flux_density_exposures(t, nu, np.full_like(t, 1.0), num_points=10).total
    inst = m.flux_density(t, nu).total
    assert np.all(np.isfinite(avg)) and np.all(avg > 0)
    np.testing.assert_allclose(avg, inst, rtol=0.01)


def test_model_properties_and_single_point_requests():
    m = Model(TophatJet(theta_c=0.1, E_iso=1e52, Gamma0=300), ISM(n_ism=1.0), Observer(lumi_dist=1e28, z=1.0, theta_obs=0.0),
              Radiation(eps_e=0.1, eps_B=0.01, p=2.2))
    assert m.observer.z == 1.0 and m.observer.theta_obs == 0.0
    assert m.fwd_rad.eps_e == pytest.approx(0.1) and m.fwd_rad.p == pytest.approx(2.2) and m.rvs_rad is None
    assert len(m.resolutions) == 3 and m.rtol > 0 and m.axisymmetric is True and "Model" in repr(m)
    f = m.flux_density(np.array([1e4]), np.array([1e14]))
    assert f.total.shape == (1,) and f.total[0] > 0
    assert m.flux_density_grid(np.logspace(3, 6, 7), np.array([1e14])).total.shape == (1, 7)
    assert m.flux_density_grid(np.array([1e4]), np.logspace(9, 18, 5)).total.shape == (5, 1)
    idx = va.logscale_screen(np.logspace(1, 5, 1000), 10)
    assert 0 < len(idx) < 1000 and idx[0] == 0 and idx[-1] == 999


# ---------------------------------------------------------------------------------------------------------------
# Blandford-McKee phase scalings of the forward shock through Model.details (tests/python/test_shock_scalings.py:35-62
# with the tables of tests/validation/regression/run_regression.py:20-47: on-axis top-hat, E_iso 1e52, theta_c 0.1,
# eps_e = eps_B = 0.01, p 2.2, resolutions (0.3, 2, 15); ISM n 0.1 / Gamma0 300 over t_obs 5e2..5e3 s, wind
# A_star 0.3 / Gamma0 70 over 1e4..1e5 s; slope tolerance 0.1)
# ---------------------------------------------------------------------------------------------------------------
_BM = {"ISM": (dict(medium=ISM(n_ism=0.1), Gamma0=300), (5e2, 5e3), {"u": -3 / 8, "r": 1 / 4, "B_comv": -3 / 8, "N_p": 3 / 4}),
       "wind": (dict(medium=Wind(A_star=0.3), Gamma0=70), (1e4, 1e5), {"u": -1 / 4, "r": 1 / 2, "B_comv": -3 / 4, "N_p": 1 / 2})}


@pytest.mark.parametrize("medium", ["ISM", "wind"])
def test_blandford_mckee_phase_scalings(medium):
    cfg, (t_lo, t_hi), expected = _BM[medium]
    m = Model(TophatJet(theta_c=0.1, E_iso=1e52, Gamma0=cfg["Gamma0"]), cfg["medium"], Observer(lumi_dist=1e28, z=1.0, theta_obs=0.0),
              Radiation(eps_e=0.01, eps_B=0.01, p=2.2, xi_e=1.0), resolutions=(0.3, 2, 15))
    d = m.details(t_lo, t_hi)
    t = np.asarray(d.fwd.t_obs)[0, 0, :]
    order = np.argsort(t)
    t = t[order]
    G = np.asarray(d.fwd.Gamma)[0, 0, :][order]
    series = {"u": G * np.sqrt(1.0 - 1.0 / (G * G))}
    for key in ("r", "B_comv", "N_p"):
        series[key] = np.asarray(getattr(d.fwd, key))[0, 0, :][order]
    w = (t >= t_lo) & (t <= t_hi)
    assert w.sum() >= 5
    for key, want in expected.items():
        y = series[key]
        ok = w & (y > 0) & np.isfinite(y)
        assert abs(_slope(t[ok], y[ok]) - want) < 0.1, (medium, key, _slope(t[ok], y[ok]), want)
