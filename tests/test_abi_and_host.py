"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol include/*.h declares
(no compute without a GPU), struct layouts agree between C and ctypes, the host-side mirror validates like the
reference's pybind layer, and the product never falls back to a CPU path."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import _abi
import vegasafterglow_amd as va
from vegasafterglow_amd import _lib, fitting

HEADER = os.path.join(_abi.ROOT, "include", "vegasafterglow_amd.h")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.load()


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vag_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(lib):
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/vegasafterglow_amd.h but not exported"
    assert set(_lib.EXPORTS) <= set(names)
    assert lib.vag_abi_version() == 13
    assert b"gfx950" in lib.vag_version()


def test_struct_layouts_match_the_header():
    assert C.sizeof(_lib.ModelParams) == 272 == C.sizeof(_abi.ModelParams)
    assert _lib.ModelParams.theta_c.offset == 8 and _lib.ModelParams.rtol.offset == 184
    # VAG_P_* slots index the doubles that follow the two int32 tags
    for key, slot in _lib.PARAM_SLOTS.items():
        field = {"tau": "duration", "theta_v": "theta_obs", "eps_e_r": "rvs_eps_e", "eps_B_r": "rvs_eps_B", "p_r": "rvs_p",
                 "xi_e_r": "rvs_xi_e", "L0": "mag_L0", "t0": "mag_t0", "q": "mag_q"}.get(key, key)
        assert getattr(_lib.ModelParams, field).offset == 8 + 8 * slot, key
    v6 = 272 + 4 + 64 + 64 + 8 + 5 * 8 + 4 + 8 + 8 + 8 + 8  # ... + ext_kernel, a_v_fixed, n_bands+pad, bands
    assert _lib.FitSpec.use_priors.offset == v6  # ABI v7 appends: use_priors+pad, lower, upper, prior_kind, prior_a, prior_b
    assert C.sizeof(_lib.FitSpec) == v6 + 8 + 128 + 128 + 64 + 128 + 128
    # ABI v8 appends n_models_ssc_rebuilt + pad (v11: the pad is n_ssc_all_cell_fallbacks), v10 ic_pool_bytes, v11 ode_rhs, v12 n_ssc_slow_cells,
    # v13 ode_lane_attempts + ode_lane_slots
    assert C.sizeof(_lib.Plan) == 4 + 4 + 5 * 8 + 6 * 4 + 2 * 4 + 2 * 8 + 2 * 4 + 8 + 8 + 8 + 16
    assert _lib.Plan.ode_lane_slots.offset == _lib.Plan.n_ssc_slow_cells.offset + 16
    assert _lib.Plan.n_ssc_all_cell_fallbacks.offset == _lib.Plan.n_models_ssc_rebuilt.offset + 4
    assert _lib.Plan.ode_rhs.offset == _lib.Plan.ic_pool_bytes.offset + 8
    assert _lib.Plan.n_ssc_slow_cells.offset == _lib.Plan.ode_rhs.offset + 8


def test_defaults_and_validation_through_the_c_abi(lib):
    p = _lib.ModelParams()
    lib.vag_params_default(C.byref(p))
    assert (p.phi_resol, p.theta_resol, p.t_resol, p.rtol) == (0.06, 0.15, 6.0, 1e-6)  # simulation-defaults.h:58-68
    assert p.xi_e == 1.0 and p.radiative_fireball == 1 and p.n0 == float("inf")
    assert lib.vag_params_validate(C.byref(p)) == 0
    p.eps_e = 1.5
    assert lib.vag_params_validate(C.byref(p)) == _lib.VAG_E_INVALID
    assert b"eps_e" in lib.vag_last_error()


def test_thread_pool_entry_points_reject_bad_arguments_without_touching_a_device(lib):
    """ABI v11: the coalesced single-model entry points and their configuration validate before they queue anything (no GPU here: a
    null context / buffer is VAG_E_INVALID with a message, never a crash), and the Python switch is off by default."""
    dp = C.POINTER(C.c_double)
    p = _lib.ModelParams()
    lib.vag_params_default(C.byref(p))
    t = np.array([1e3, 1e4])
    out = np.empty(2)
    assert lib.vag_flux_density_coalesced(None, C.byref(p), t.ctypes.data_as(dp), t.ctypes.data_as(dp), 2, out.ctypes.data_as(dp), None) == _lib.VAG_E_INVALID
    assert b"null" in lib.vag_last_error()
    assert lib.vag_flux_density_grid_coalesced(None, C.byref(p), t.ctypes.data_as(dp), 2, t.ctypes.data_as(dp), 2, out.ctypes.data_as(dp), None) == _lib.VAG_E_INVALID
    assert lib.vag_flux_coalesced(None, C.byref(p), t.ctypes.data_as(dp), 2, 1e14, 1e15, 4, out.ctypes.data_as(dp), None) == _lib.VAG_E_INVALID
    assert lib.vag_ctx_coalesce(None, 64, 50) == _lib.VAG_E_INVALID
    assert lib.vag_ctx_coalesce_stats(None, None, None) == _lib.VAG_E_INVALID
    from vegasafterglow_amd import model
    assert not any(model._coalescing.values())


def test_no_cpu_fallback_without_a_device(lib):
    if lib.vag_device_count() > 0:
        pytest.skip("a HIP device is present")
    h = C.c_void_p()
    assert lib.vag_ctx_create(0, C.byref(h)) == _lib.VAG_E_NO_DEVICE
    m = va.Model(va.TophatJet(0.1, 1e52, 300), va.ISM(1.0), va.Observer(1e28, 1.0, 0.0), va.Radiation(0.1, 0.01, 2.3))
    with pytest.raises(RuntimeError, match="no CPU path"):
        m.flux_density_grid(np.logspace(3, 5, 4), np.array([1e9]))


def test_product_package_never_references_the_oracle():
    pkg = os.path.join(_abi.ROOT, "vegasafterglow_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower(), f"{f} mentions the oracle"


# ---- host-side mirror of the pybind objects (pybind/pybind.cpp:205-223,347-377,384-422) ----
def test_constructor_validation_matches_reference_error_types():
    with pytest.raises(ValueError):
        va.TophatJet(0.0, 1e52, 300)
    with pytest.raises(ValueError):
        va.TophatJet(0.1, 1e52, 1.0)
    with pytest.raises(ValueError):
        va.TwoComponentJet(0.1, 1e52, 300, 0.05, 1e50, 50)  # theta_w <= theta_c
    with pytest.raises(ValueError):
        va.ISM(-1.0)
    with pytest.raises(ValueError):
        va.Wind(0.0)
    with pytest.raises(ValueError):
        va.Observer(1e28, -0.1, 0.0)
    with pytest.raises(ValueError):
        va.Observer(1e28, 0.0, 4.0)
    with pytest.raises(ValueError):
        va.Radiation(0.1, 0.01, 1.0)
    with pytest.raises(ValueError):
        va.Radiation(0.0, 0.01, 2.3)
    obs, rad = va.Observer(1e28, 1.0, 0.0), va.Radiation(0.1, 0.01, 2.3)
    with pytest.raises(TypeError):
        va.Model("tophat", va.ISM(1.0), obs, rad)
    with pytest.raises(TypeError):
        va.Model(va.TophatJet(0.1, 1e52, 300), "ism", obs, rad)
    with pytest.raises(ValueError):
        va.Model(va.TophatJet(0.1, 1e52, 300), va.ISM(1.0), obs, rad, rtol=1.0)
    with pytest.raises(ValueError):
        va.Model(va.TophatJet(0.1, 1e52, 300), va.ISM(1.0), obs, rad, resolutions=(0.0, 0.15, 6))
    msp = va.Model(va.TophatJet(0.1, 1e52, 300, spreading=True), va.ISM(1.0), obs, rad, axisymmetric=False)
    assert msp.params.flags == 128 | 32  # (phi, theta) pair rows on the device
    m3d = va.Model(va.TophatJet(0.1, 1e52, 300), va.ISM(1.0), obs, rad, axisymmetric=False)
    assert m3d.params.flags == 128 and not m3d.axisymmetric
    with pytest.raises(TypeError):
        va.Model(va.TophatJet(0.1, 1e52, 300), va.ISM(1.0), obs, rad, rvs_rad=(0.1, 0.01, 2.3))
    m = va.Model(va.TophatJet(0.1, 1e52, 300), va.ISM(1.0), obs, rad, rvs_rad=va.Radiation(0.2, 0.02, 2.6, ssc=True))
    assert m.params.flags == 4 | 8 and (m.params.rvs_eps_e, m.params.rvs_eps_B, m.params.rvs_p) == (0.2, 0.02, 2.6)
    assert m.resolutions == (0.06, 0.2, 10.0)  # reverse-shock runs default to the denser grid (pymodel.h:630-637)
    m = va.Model(va.TophatJet(0.1, 1e52, 300), va.ISM(1.0), obs, va.Radiation(0.1, 0.01, 2.3, ssc=True, kn=True))
    assert m.params.flags == 3  # VAG_FLAG_SSC | VAG_FLAG_KN


def test_model_flattens_to_the_same_struct_as_the_test_helper():
    m = va.Model(va.PowerLawJet(0.1, 1e52, 300, 2.0, 3.0), va.Wind(0.1, n_ism=1e-3), va.Observer(1e28, 1.0, 0.2),
                 va.Radiation(0.1, 0.01, 2.3, xi_e=0.5), resolutions=(0.29, 0.16, 10.0), rtol=1e-5,
                 radiative_fireball=False)
    want = _abi.make_params(jet="PowerLawJet", medium="Wind", k_e=2.0, k_g=3.0, A_star=0.1, n_ism=1e-3, theta_obs=0.2,
                            xi_e=0.5, resolutions=(0.29, 0.16, 10.0), rtol=1e-5, radiative_fireball=False,
                            theta_w=np.pi / 2, E_iso_w=1e52, Gamma0_w=300.0)
    assert bytes(m.params) == bytes(want)
    assert m.resolutions == (0.29, 0.16, 10.0) and m.rtol == 1e-5 and m.axisymmetric and not m.radiative_fireball


def test_argument_checks_of_flux_calls_happen_before_the_device():
    m = va.Model(va.TophatJet(0.1, 1e52, 300), va.ISM(1.0), va.Observer(1e28, 1.0, 0.0), va.Radiation(0.1, 0.01, 2.3))
    with pytest.raises(ValueError, match="non-empty"):
        m.flux_density_grid([], [1e9])
    with pytest.raises(ValueError, match="same size"):
        m.flux_density([1e3, 1e4], [1e9])


# ---- Fitter host logic (fitter.py:407-451, utils.py:110-135, samplers.py:72-91) ----
def _fitter():
    f = fitting.Fitter(z=0.0098, lumi_dist=1.23e26, jet="gaussian", medium="ism")
    f.add_flux_density(3e9, [3e6, 1e6, 2e6], [1e-27, 2e-27, 3e-27], [1e-28, 2e-28, 3e-28], weights=[1, 2, 1])
    f.add_flux_density(5e14, [1.5e6], [4e-29], [4e-30])
    return f


def test_consolidate_sorts_by_time_and_normalises_weights():
    f = _fitter()
    f._consolidate_data()
    assert list(f._all_t) == [1e6, 1.5e6, 2e6, 3e6]
    assert list(f._all_nu) == [3e9, 5e14, 3e9, 3e9]
    np.testing.assert_allclose(f._all_weights.sum(), 4.0)
    np.testing.assert_allclose(f._all_weights, np.array([2, 1, 1, 1]) * 4 / 5)
    np.testing.assert_allclose(f._all_log_err, 0.1)
    with pytest.raises(ValueError):
        g = fitting.Fitter(z=0.1, lumi_dist=1e27)
        g.add_flux_density(1e9, [1e5], [-1.0], [0.1])
        g._consolidate_data()


def test_spec_is_the_transformer_as_a_slot_map():
    f = _fitter()
    defs = [fitting.ParamDef("E_iso", 1e50, 1e54, fitting.Scale.log), fitting.ParamDef("theta_v", 0.0, 0.8),
            fitting.ParamDef("p", 2.05, 2.8), fitting.ParamDef("n_ism", 1e-2, 1e-2, fitting.Scale.fixed),
            fitting.ParamDef("eps_B", 1e-3, 1e-3, fitting.Scale.fixed, initial=2e-3)]
    spec, lo, hi = f.build_spec(defs)
    assert spec.ndim == 3 and list(spec.slot[:3]) == [1, 14, 17] and list(spec.is_log[:3]) == [1, 0, 0]
    assert spec.base.n_ism == 1e-2 and spec.base.eps_B == 2e-3 and spec.base.jet_type == _lib.JET_GAUSSIAN
    assert spec.base.Gamma0 == 300.0 and spec.base.eps_e == 0.1  # ModelParams defaults, types.py:37-77
    assert spec.base.lumi_dist == 1.23e26 and spec.base.z == 0.0098 and spec.n_data == 4
    assert list(lo) == [50, 0.0, 2.05] and list(hi) == [54, 0.8, 2.8]


def test_log_prob_batch_bounds_prior_and_nonfinite_handling():
    f = _fitter()
    defs = [fitting.ParamDef("E_iso", 1e50, 1e54, fitting.Scale.log), fitting.ParamDef("theta_v", 0.0, 0.8)]
    calls = []

    def fake_loglike(s):
        calls.append(s.copy())
        out = -np.arange(len(s), dtype=float)
        out[0] = np.nan
        return out

    lp = f.make_log_prob_batch(defs, loglike_fn=fake_loglike)
    samples = np.array([[52.0, 0.1], [49.0, 0.1], [53.0, 0.2], [52.0, 0.9], [51.0, 0.3]])
    got = lp(samples)
    assert calls[0].shape == (3, 2)  # only in-bounds walkers are evaluated
    ln_prior = -np.log(4.0) - np.log(0.8)
    assert got[0] == -np.inf and got[1] == -np.inf and got[3] == -np.inf
    np.testing.assert_allclose(got[[2, 4]], np.array([-1.0, -2.0]) + ln_prior)


def test_priors_are_kept_when_a_sharded_loglike_is_substituted():
    """make_log_prob_batch(loglike_fn=..., priors=...) (the multi-GPU form): the reference's log_prob_batch always adds
    prior_dict[name].ln_prob (samplers.py:72-91), so Gaussian / LogUniform / Uniform-with-own-support / object priors must all
    reach the result, and a sample outside the bounds or a prior's support is never evaluated."""
    f = fitting.Fitter(z=0.1, lumi_dist=1e27, jet="gaussian", medium="ism")
    f.add_flux_density(1e9, np.array([1e4, 1e5]), np.array([1e-27, 2e-27]), np.array([1e-28, 2e-28]))
    defs = [fitting.ParamDef("E_iso", 1e50, 1e54, fitting.Scale.log), fitting.ParamDef("theta_c", 0.02, 0.5),
            fitting.ParamDef("p", 2.05, 2.9), fitting.ParamDef("eps_e", 1e-3, 0.5, fitting.Scale.log)]

    class Triangle:  # an object the device does not know: only ln_prob
        def ln_prob(self, x):
            return np.log(np.clip(1 - np.abs(x - 2.5) / 0.5, 1e-300, None))

    class Uniform:  # shaped like bilby.core.prior.Uniform, narrower than the ParamDef
        minimum, maximum = -2.5, -0.7

    priors = {"E_iso": ("gaussian", 52.0, 0.7), "theta_c": ("log_uniform", 0.01, 1.0), "p": Triangle(), "eps_e": Uniform()}
    seen = []

    def loglike(block):
        seen.append(np.array(block))
        return -0.5 * np.sum(block ** 2, axis=1)
    lpb = f.make_log_prob_batch(defs, loglike_fn=loglike, priors=priors)
    x = np.array([[52.3, 0.1, 2.4, -1.0], [49.0, 0.1, 2.4, -1.0], [52.3, 0.1, 2.4, -2.8], [51.0, 0.3, 2.7, -2.0]])
    got = lpb(x)
    ok = [0, 3]
    assert np.all(got[[1, 2]] == -np.inf) and len(seen) == 1 and np.array_equal(seen[0], x[ok])  # out of bounds / support: not evaluated
    want = (-0.5 * np.sum(x[ok] ** 2, axis=1)
            + (-0.5 * ((x[ok, 0] - 52.0) / 0.7) ** 2 - np.log(0.7 * np.sqrt(2 * np.pi)))
            + (-np.log(x[ok, 1] * np.log(1.0 / 0.01)))
            + priors["p"].ln_prob(x[ok, 2])
            + (-np.log(-0.7 + 2.5)))
    np.testing.assert_allclose(got[ok], want, rtol=1e-14)
    # the device spec carries the same kinds: Uniform-with-own-support is VAG_PRIOR_UNIFORM_RANGE, the object stays on the host
    spec, _, _ = f.build_spec(defs, priors=priors, use_priors=True)
    assert list(spec.prior_kind[:4]) == [_lib.PRIOR_GAUSSIAN, _lib.PRIOR_LOG_UNIFORM, _lib.PRIOR_NONE, _lib.PRIOR_UNIFORM_RANGE]
    assert (spec.prior_a[3], spec.prior_b[3]) == (-2.5, -0.7) and [d for d, _ in f._host_priors] == [2]
    with pytest.raises(ValueError, match="host-only priors"):
        f.device_evaluator(defs, priors=priors, use_priors=True)


def test_stretch_move_sampler_recovers_a_gaussian_posterior():
    """The dependency-free ensemble sampler (vegasafterglow_amd/sampling.py) on an analytic target: it must call the
    batched log-probability with half the walkers at a time and recover the mean / covariance of a correlated Gaussian."""
    from vegasafterglow_amd import sampling
    mean = np.array([1.0, -2.0, 0.5])
    cov = np.array([[1.0, 0.6, 0.0], [0.6, 2.0, -0.3], [0.0, -0.3, 0.5]])
    icov = np.linalg.inv(cov)
    calls = []

    def log_prob_batch(x):
        calls.append(len(x))
        d = x - mean
        return -0.5 * np.einsum("ij,jk,ik->i", d, icov, d)

    rng = np.random.default_rng(3)
    pos0 = sampling.initial_positions(np.full(3, -10.0), np.full(3, 10.0), 32, rng)
    chain, logp, acc = sampling.run_stretch_move(log_prob_batch, pos0, 1500, rng=rng)
    assert calls[0] == 32 and set(calls[1:]) == {16}
    flat = chain[300:].reshape(-1, 3)
    assert np.all(np.abs(flat.mean(axis=0) - mean) < 0.15)
    assert np.all(np.abs(np.cov(flat.T) - cov) < 0.35)
    assert 0.2 < acc.mean() < 0.8 and np.all(np.isfinite(logp))
    with pytest.raises(ValueError):
        sampling.run_stretch_move(log_prob_batch, pos0[:5], 10)


def test_batch_pool_routes_point_queues_through_one_batched_call():
    """BatchPool (the bilby `pool=` adapter): a queue of sampler-space points is ONE batch_fn call; anything else falls
    back to the per-point callable, and non-finite likelihoods come back as -inf."""
    from vegasafterglow_amd import sampling
    seen = []

    def batch_fn(v):
        seen.append(v.shape)
        out = -0.5 * np.sum(v * v, axis=1)
        out[0] = np.nan
        return out

    pool = sampling.BatchPool(batch_fn, ndim=3, size=64)
    pts = [np.full(3, 0.1 * i) for i in range(10)]
    got = pool.map(lambda x: 123.0, pts)
    assert seen == [(10, 3)] and pool.calls == 1 and got[0] == -np.inf
    assert np.allclose(got[1:], [-0.5 * 3 * (0.1 * i) ** 2 for i in range(1, 10)])
    assert pool.map(lambda x: x + 1, [1, 2, 3]) == [2, 3, 4] and pool.calls == 1  # not a point queue
    assert pool.map(lambda x: x, []) == []
    with pool as p:
        assert p is pool
    pool.close(), pool.join(), pool.shutdown(wait=True)


def test_fitter_add_spectrum_and_validate_parameters():
    """Fitter.add_spectrum (fitter.py:284-314) files one point per frequency at a fixed time; validate_parameters rejects
    what the reference's fitting/params.py rejects before any sampler runs."""
    f = fitting.Fitter(z=0.1, lumi_dist=1e27, jet="tophat", medium="ism")
    nu = np.array([1e9, 1e14, 1e18])
    f.add_spectrum(3e5, nu, np.array([1e-27, 2e-28, 3e-31]), np.array([1e-28, 2e-29, 3e-32]))
    f.add_flux_density(5e14, np.array([1e4, 1e6]), np.array([1e-27, 1e-28]), np.array([1e-28, 1e-29]))
    f._consolidate_data()
    assert f._all_t.size == 5 and np.all(np.diff(f._all_t) >= 0)
    assert sorted(f._all_nu[f._all_t == 3e5].tolist()) == nu.tolist()
    with pytest.raises(ValueError):
        f.add_spectrum(-1.0, nu, nu, nu)
    with pytest.raises(ValueError):
        f.add_spectrum(1e5, np.array([1e9, -1.0]), np.ones(2), np.ones(2))
    P, S = fitting.ParamDef, fitting.Scale
    f.validate_parameters([P("E_iso", 1e50, 1e54, S.log), P("theta_v", 0.0, 0.5, S.linear), P("p", 2.3, 2.3, S.fixed)])
    for bad in ([P("E_iso", 1e50, 1e54, S.log), P("E_iso", 1e50, 1e54, S.log)],      # duplicate
                [P("E_iso", 0.0, 1e54, S.log)],                                        # log scale from zero
                [P("theta_v", 0.5, 0.1, S.linear)],                                    # inverted bounds
                [P("not_a_parameter", 0.0, 1.0, S.linear)],
                [P("A_V", 0.0, 1.0, S.linear)]):                                       # no extinction law configured
        with pytest.raises(ValueError):
            f.validate_parameters(bad)


def test_reprs_follow_the_reference_format():
    """__repr__ of the value classes (pymodel.h:55-58,262-272,316-334)."""
    assert repr(va.Observer(1e28, 1.0, 0.0)) == "Observer(lumi_dist=1e+28, z=1, theta_obs=0)"
    assert repr(va.Observer(1e28, 1.0, 0.1, 0.5)) == "Observer(lumi_dist=1e+28, z=1, theta_obs=0.1, phi_obs=0.5)"
    assert repr(va.Radiation(0.1, 0.01, 2.2)) == "Radiation(eps_e=0.1, eps_B=0.01, p=2.2)"
    assert repr(va.Radiation(0.1, 0.01, 2.2, xi_e=0.5, ssc=True, kn=True)) == "Radiation(eps_e=0.1, eps_B=0.01, p=2.2, xi_e=0.5, ssc=True, kn=True)"
    assert "Magnetar(L0=1e+47" in repr(va.Magnetar(1e47, 1e3, 2))


def test_fitter_observation_validation_rules():
    """The add_* boundary checks of the reference Fitter (fitter.py:212-282,316-377): clear ValueErrors for empty data, shape
    mismatches, non-finite fluxes, non-positive errors, bad weights, bad frequencies and bad bands."""
    f = fitting.Fitter(z=0.5, lumi_dist=1e28)
    t, fl, er = np.array([1e3, 1e4, 1e5]), np.array([1e-26, 5e-27, 1e-27]), np.array([1e-27, 5e-28, 1e-28])
    f.add_flux_density(4.84e14, t, fl, er, weights=[1.0, 0.5, 0.0], label="r")
    for bad_nu in (0, -1e14, np.nan, np.inf):
        with pytest.raises(ValueError, match="nu must be finite and > 0"):
            f.add_flux_density(bad_nu, t, fl, er)
    with pytest.raises(ValueError, match="same shape"):
        f.add_flux_density(1e14, t, fl[:2], er)
    with pytest.raises(ValueError, match="empty"):
        f.add_flux_density(1e14, [], [], [])
    with pytest.raises(ValueError, match="non-finite"):
        f.add_flux_density(1e14, t, np.array([1e-26, np.nan, 1e-27]), er)
    for bad_err in ([1e-27, 0.0, 1e-28], [1e-27, -1e-28, 1e-28], [1e-27, np.nan, 1e-28], [1e-27, np.inf, 1e-28]):
        with pytest.raises(ValueError, match="err must be finite and > 0"):
            f.add_flux_density(1e14, t, fl, np.array(bad_err))
    with pytest.raises(ValueError, match="weights.shape"):
        f.add_flux_density(1e14, t, fl, er, weights=[1.0, 1.0])
    with pytest.raises(ValueError, match="weights must be finite and >= 0"):
        f.add_flux_density(1e14, t, fl, er, weights=[1.0, -1.0, 1.0])
    with pytest.raises(ValueError, match="same shape"):
        f.add_flux((1e17, 1e18), t, fl[:2], er)
    for bad_band in (1e17, (1e17,), (1e18, 1e17), (0.0, 1e17), (1e17, np.inf)):
        with pytest.raises(ValueError):
            f.add_flux(bad_band, t, fl, er)
    with pytest.raises(ValueError, match="num_points"):
        f.add_flux((1e17, 1e18), t, fl, er, num_points=1)


def test_logscale_screen_thins_to_a_density_per_decade():
    """pybind.h:40-107: end points kept, log-uniform targets snapped to the nearest interior sample, no duplicates."""
    import vegasafterglow_amd as va
    assert va.logscale_screen(np.array([1.0, 2.0, 5.0, 10.0, 20.0, 50.0, 100.0]), 1) == [0, 3, 6]
    assert va.logscale_screen(np.array([42.0]), 10) == [0] and va.logscale_screen(np.array([]), 10) == []
    assert va.logscale_screen(np.array([1.0, 2.0, 3.0]), 0) == [0, 1, 2]
    assert va.logscale_screen(np.array([1.0, 1.5]), 10) == [0, 1]
    idx = va.logscale_screen(np.logspace(1, 5, 1000), 10)
    assert len(idx) == 41 and idx[0] == 0 and idx[-1] == 999 and idx == sorted(set(idx))


@pytest.mark.parametrize("law", ["smc", "lmc", "mw"])
def test_pei92_extinction_laws(law):
    """The properties the reference asserts of its laws (tests/python/test_extinction.py:11-55) and three values computed
    by hand from Pei (1992) Table 4."""
    from vegasafterglow_amd import extinction as ex
    assert ex.pei92(5.5e-5, law) == pytest.approx(1.0, rel=1e-12)
    assert np.array_equal(ex.pei92(np.array([0.5, 0.9, 0.99]) * 912e-8, law), np.zeros(3))
    assert ex.pei92(2000e-8, law) > ex.pei92(7000e-8, law) > 0
    lam = np.geomspace(1000e-8, 3e-4, 25).reshape(5, 5)
    out = ex.pei92(lam, law)
    assert out.shape == lam.shape and np.all(np.isfinite(out)) and np.all(out >= 0)
    assert np.array_equal(ex.BUILTIN_LAWS[law](lam), out) and set(ex.BUILTIN_LAWS) == {"smc", "lmc", "mw"}
    # independent scalar evaluation of the six-term sum at 1500 A
    r_v, terms = ex._TERMS[law]

    def raw(l_um):
        return (1 + 1 / r_v) * sum(a / ((l_um / li) ** n + (li / l_um) ** n + b) for a, li, b, n in terms)

    assert ex.pei92(1500e-8, law) == pytest.approx(raw(0.15) / raw(0.55), rel=1e-13)


def test_fitter_accepts_named_extinction_laws():
    from vegasafterglow_amd import fitting, extinction as ex
    f = fitting.Fitter(z=1.0, lumi_dist=1e28, extinction="smc")
    assert f.extinction is ex.smc
    with pytest.raises(ValueError, match="Unknown extinction law"):
        fitting.Fitter(z=1.0, lumi_dist=1e28, extinction="calzetti")
    with pytest.raises(ValueError):
        fitting.Fitter(z=1.0, lumi_dist=1e28, extinction=3.1)


def test_pei92_laws_equal_the_reference_vectors():
    """k(lambda) on 43 wavelengths from the reference's own module (tests/golden/make_extinction_vectors.py)."""
    from vegasafterglow_amd import extinction as ex
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "extinction_pei92.npz"))
    for law in ("smc", "lmc", "mw"):
        np.testing.assert_allclose(ex.pei92(g["lam_cm"], law), g[law], rtol=1e-14, atol=0)


def test_units_multipliers_magnitudes_and_named_bands():
    """units.py:36-88,172-205,339-367 of the reference: engine conventions are s, Hz, cm, rad, erg/cm^2/s/Hz."""
    from vegasafterglow_amd import units as u
    assert u.day == 86400.0 and u.yr == pytest.approx(3.15569e7, rel=1e-5) and u.GHz == 1e9 and u.mJy == 1e-26
    assert u.keV == pytest.approx(2.417989e17, rel=1e-6) and u.Mpc == pytest.approx(3.0857e24, rel=1e-4)
    assert u.deg * 180 == pytest.approx(np.pi) and u.arcsec * 3600 == pytest.approx(u.deg)
    assert u.ABmag_to_cgs(0.0) == pytest.approx(3.631e-20) and u.ABmag_to_cgs(23.9) == pytest.approx(1e-29, rel=2e-3)
    mags = np.array([15.0, 20.0, 25.0])
    np.testing.assert_allclose(u.cgs_to_ABmag(u.ABmag_to_cgs(mags)), mags, rtol=1e-14)
    lo, hi = u.band("XRT")
    assert lo == pytest.approx(0.3 * u.keV) and hi == pytest.approx(10 * u.keV) and u.band("LAT")[1] == pytest.approx(300 * u.GeV)
    with pytest.raises(ValueError, match="Unknown band"):
        u.band("nope")


def test_custom_extinction_callables_in_either_signature():
    """k(lambda) and the reference's (lambda, params) form both work; a law that needs the sampled params fails loudly."""
    from vegasafterglow_amd import fitting
    lam = np.array([3e-5, 6e-5])
    one = fitting.Fitter(z=1.0, lumi_dist=1e28, extinction=lambda l: 5.5e-5 / l)
    two = fitting.Fitter(z=1.0, lumi_dist=1e28, extinction=lambda l, params: 5.5e-5 / l)
    assert np.array_equal(one._k_lambda(lam), two._k_lambda(lam))
    needs = fitting.Fitter(z=1.0, lumi_dist=1e28, extinction=lambda l, params: params["R_V"] / l)
    with pytest.raises(TypeError):
        needs._k_lambda(lam)


def test_extinction_kernel_follows_a_fixed_z_and_outlives_later_specs():
    """A fixed 'z' ParamDef moves the law's rest-frame wavelengths; a later spec WITHOUT a fixed z goes back to Fitter.z (no
    KeyError), every spec keeps pointing at a live array of its own z, and a free 'z' next to an extinction law is refused."""
    from vegasafterglow_amd import fitting
    P, S = fitting.ParamDef, fitting.Scale
    f = fitting.Fitter(z=0.5, lumi_dist=1e28, jet="tophat", medium="ism", extinction=lambda l: 5.5e-5 / l)
    nu = np.array([3e14, 5e14, 8e14])
    f.add_flux_density(nu, np.array([1e4, 1e5, 1e6]), np.array([1e-26, 1e-27, 1e-28]), np.array([1e-27, 1e-28, 1e-29]))
    free = [P("E_iso", 1e50, 1e54, S.log), P("A_V", 0.0, 1.0, S.linear)]
    kern = lambda z: 0.4 * np.log(10.0) * 5.5e-5 / ((2.99792458e10 / nu) / (1.0 + z))
    read = lambda spec: np.array([spec.ext_kernel[i] for i in range(3)])
    s_own, _, _ = f.build_spec(free)
    np.testing.assert_allclose(read(s_own), kern(0.5), rtol=1e-15)
    s_fix, _, _ = f.build_spec(free + [P("z", 1.5, 1.5, S.fixed)])
    np.testing.assert_allclose(read(s_fix), kern(1.5), rtol=1e-15)
    assert s_fix.base.z == 1.5
    s_back, _, _ = f.build_spec(free)            # used to raise KeyError('z')
    np.testing.assert_allclose(read(s_back), kern(0.5), rtol=1e-15)
    s_other, _, _ = f.build_spec(free + [P("z", 2.5, 2.5, S.fixed)])
    import gc
    gc.collect()
    np.testing.assert_allclose(read(s_fix), kern(1.5), rtol=1e-15)   # earlier specs still read THEIR kernel
    np.testing.assert_allclose(read(s_own), kern(0.5), rtol=1e-15)
    np.testing.assert_allclose(read(s_other), kern(2.5), rtol=1e-15)
    with pytest.raises(ValueError, match="free 'z'"):
        f.build_spec(free + [P("z", 0.1, 2.0, S.linear)])


def _toy_fitter():
    f = fitting.Fitter(z=0.1, lumi_dist=1e27, jet="tophat", medium="ism")
    f.add_flux_density(5e14, np.array([1e4, 1e5, 1e6]), np.array([1e-26, 1e-27, 1e-28]), np.array([1e-27, 1e-28, 1e-29]))
    P, S = fitting.ParamDef, fitting.Scale
    return f, [P("E_iso", 1e50, 1e54, S.log), P("theta_v", 0.0, 0.8, S.linear), P("p", 2.05, 2.8, S.linear)]


def test_emcee_adapter_follows_the_vectorized_ensemble_sampler_protocol(monkeypatch):
    """sampling.emcee_sampler against a stand-in with emcee 3's call protocol for vectorize=True (ensemble.py compute_log_prob):
    the log-probability function receives a float64 [n, ndim] block -- the whole ensemble first, then one red-blue half per
    move --, must return n values, NaN raises "Probability function returned NaN", -inf is a legal rejection.  The closure
    built by Fitter.make_log_prob_batch has to satisfy that contract without emcee being installed here."""
    import sys
    import types
    from vegasafterglow_amd import sampling
    calls = []

    class EnsembleSampler:  # the part of emcee.EnsembleSampler the reference uses (fitting/samplers.py:100-130)
        def __init__(self, nwalkers, ndim, log_prob_fn, vectorize=False, moves=None, **kw):
            assert vectorize is True, "the reference passes vectorize=True"
            self.nwalkers, self.ndim, self.fn, self.moves = nwalkers, ndim, log_prob_fn, moves
            self.chain, self.lnprob = [], []

        def compute_log_prob(self, coords):
            p = np.asarray(coords, dtype=np.float64)
            assert p.ndim == 2 and p.shape[1] == self.ndim
            results = self.fn(p)  # vectorize: ONE call with the block
            calls.append(p.shape[0])
            try:
                log_prob = np.array([float(r[0]) for r in results])
            except (IndexError, TypeError):
                log_prob = np.array([float(r) for r in results])
            assert log_prob.shape == (p.shape[0],)
            if np.any(np.isnan(log_prob)):
                raise ValueError("Probability function returned NaN")
            return log_prob

        def run_mcmc(self, initial_state, nsteps, rng=np.random.default_rng(0)):
            pos = np.array(initial_state, dtype=np.float64)
            lp = self.compute_log_prob(pos)
            half = self.nwalkers // 2
            for _ in range(nsteps):
                for s, c in ((slice(0, half), slice(half, None)), (slice(half, None), slice(0, half))):
                    z = ((2.0 - 1.0) * rng.random(half) + 1.0) ** 2 / 2.0
                    partner = pos[c][rng.integers(half, size=half)]
                    prop = partner + z[:, None] * (pos[s] - partner)
                    new = self.compute_log_prob(prop)
                    take = np.log(rng.random(half)) < (self.ndim - 1) * np.log(z) + new - lp[s]
                    pos[s] = np.where(take[:, None], prop, pos[s])
                    lp[s] = np.where(take, new, lp[s])
                self.chain.append(pos.copy())
                self.lnprob.append(lp.copy())
            return pos

    monkeypatch.setitem(sys.modules, "emcee", types.SimpleNamespace(EnsembleSampler=EnsembleSampler))
    f, defs = _toy_fitter()
    _, lo, hi = f.build_spec(defs)
    centre = 0.5 * (lo + hi)

    def fake_loglike(s):  # stands in for the device: a Gaussian in sampler space; one walker per call comes back NaN
        out = -0.5 * np.sum(((s - centre) / (0.05 * (hi - lo))) ** 2, axis=1)
        out[0] = np.nan  # the closure must turn a non-finite likelihood into -inf, never hand NaN to the sampler
        return out

    sampler = sampling.emcee_sampler(f, defs, nwalkers=16, loglike_fn=fake_loglike)
    assert isinstance(sampler, EnsembleSampler) and sampler.ndim == 3
    rng = np.random.default_rng(1)
    pos0 = sampling.initial_positions(lo, hi, 16, rng)
    pos0[3, 0] = hi[0] + 1.0  # an out-of-bounds walker: -inf, not an exception
    sampler.run_mcmc(pos0, 40)
    assert calls[0] == 16 and set(calls[1:]) == {8}  # whole ensemble, then red-blue halves
    lnp = np.array(sampler.lnprob)
    assert not np.any(np.isnan(lnp)) and np.isfinite(lnp[-1]).sum() >= 14
    chain = np.array(sampler.chain)[:, np.isfinite(lnp[-1])]
    assert np.all(chain >= lo) and np.all(chain <= hi)  # accepted positions never leave the prior box


def test_batch_pool_serves_a_nested_sampler_queue_like_bilby_drives_it():
    """BatchPool against the way bilby hands a pool to dynesty (bilby/core/sampler/dynesty.py: `pool.map(loglikelihood, points)`
    with queue_size points per call, `use_pool={"loglikelihood": True}`, then pool.close() / pool.join() when the run ends;
    fitting/samplers.py:157-188 builds exactly that): every queue is ONE batched evaluation, the per-point callable is never
    needed for point queues, results keep the queue's order, and the pool object survives close/join like a thread pool."""
    from vegasafterglow_amd import sampling
    f, defs = _toy_fitter()
    _, lo, hi = f.build_spec(defs)
    evaluated = []

    def fake_loglike(s):
        evaluated.append(len(s))
        return -np.sum((s - lo) / (hi - lo), axis=1)

    log_prob = f.make_log_prob_batch(defs, loglike_fn=fake_loglike)
    pool = sampling.BatchPool(log_prob, ndim=3, size=24)

    def per_point(v):  # bilby's _log_likelihood_wrapper: must not be called for a point queue
        raise AssertionError("point queues go through the batched call")

    rng = np.random.default_rng(2)
    live = lo + (hi - lo) * rng.random((100, 3))
    logl = np.array(pool.map(per_point, list(live)))        # initial live points: one call
    assert logl.shape == (100,) and evaluated == [100]
    for it in range(5):                                      # each iteration fills a queue of `size` proposals
        queue = [lo + (hi - lo) * rng.random(3) for _ in range(pool.size)]
        queue[0] = hi + 1.0                                   # a proposal outside the prior box: -inf, still one call
        out = pool.map(per_point, queue)
        assert len(out) == pool.size and out[0] == -np.inf and np.all(np.isfinite(out[1:]))
        want = -np.sum((np.array(queue[1:]) - lo) / (hi - lo), axis=1) - np.sum(np.log(hi - lo))
        np.testing.assert_allclose(out[1:], want, rtol=1e-13)  # order preserved
    assert evaluated == [100] + [pool.size - 1] * 5 and pool.calls == 6
    pool.close()
    pool.join()
    assert pool.map(per_point, [live[0]])[0] == logl[0]      # like ThreadPoolWithClose, still usable by a resumed run


def test_every_device_buffer_of_a_context_is_released_by_destroy():
    """vag_ctx_destroy frees the context's buffers from a hand-written list; a member missing from it is a leak per destroyed context
    (round 5 found one this way).  Static check of the source: every DevBuf member of vag_ctx (arrays and the members of its nested
    records included) appears in vag_ctx_destroy.  The GPU soak test holds the same to the byte through vag_device_bytes_in_use."""
    import re
    src = open(os.path.join(_abi.ROOT, "vegasafterglow_amd", "csrc", "vag_capi.hip")).read()
    a = src.index("struct vag_ctx {")
    body = src[a:src.index("\n};", a)]
    members = set()
    for m in re.finditer(r"^\s*DevBuf\s+([^;]+);", body, re.M):
        for name in re.sub(r"/\*.*?\*/", "", m.group(1)).split(","):
            name = re.sub(r"\[.*\]", "", name.strip())
            if name:
                members.add(name)
    d0 = src.index("void vag_ctx_destroy")
    released = set(re.findall(r"(?:c->|\.)(\w+)", src[d0:src.index("delete c;", d0)]))
    assert len(members) > 60
    assert not (members - released), sorted(members - released)


def test_null_arguments_without_a_device_are_error_codes(lib):
    """The entry points reject null pointers before they touch the device (the GPU suite covers the rest with a live context)."""
    assert lib.vag_params_validate(None) == _lib.VAG_E_INVALID
    assert lib.vag_ctx_create(0, None) == _lib.VAG_E_INVALID
    assert lib.vag_ctx_synchronize(None) == _lib.VAG_E_INVALID
    assert lib.vag_ctx_count_work(None, 1) == _lib.VAG_E_INVALID
    assert lib.vag_last_plan(None, C.byref(_lib.Plan())) == _lib.VAG_E_INVALID
    assert lib.vag_flux_density_grid_batch(None, None, 1, None, 1, None, 1, None) == _lib.VAG_E_INVALID
    assert lib.vag_loglike_batch(None, None, None, 1, 1, None) == _lib.VAG_E_INVALID
    lib.vag_params_default(None)
    assert lib.vag_device_bytes_in_use() == 0 or lib.vag_device_count() > 0


def test_bench_line_is_short_and_complete_by_construction():
    """The driver parses ONE JSON line of stdout.  Round 5's line had grown to 21 KB and was not parsed (BENCH_r05.json: parsed null);
    bench.compact_line now makes the line from the full record: < 4 KB, a JSON round trip, every contract key, roofline and
    cpu_baseline with their members -- checked here on a recorded full run (tests/golden/bench_detail_recorded.json = round 5's record)
    and on the same record with every leg blown up or missing."""
    import json
    import sys
    sys.path.insert(0, _abi.ROOT)
    import bench
    detail = json.load(open(os.path.join(_abi.ROOT, "tests", "golden", "bench_detail_recorded.json")))
    assert len(json.dumps(detail)) > 15000  # the record itself is what no longer fitted
    for variant in ("recorded", "bloated", "headline_only"):
        d = json.loads(json.dumps(detail))
        if variant == "bloated":
            d["config"]["workload"] = d["config"]["workload"] * 40
            d["cpu_baseline"]["sample"] = "x" * 5000
            for k in list(d):
                if isinstance(d[k], dict) and k not in bench.CONTRACT_KEYS:
                    d[k]["note"] = "y" * 20000
        if variant == "headline_only":  # --no-walkers --no-cpu-baseline, or any N > 1 rank-0 record
            d = {k: d[k] for k in bench.CONTRACT_KEYS if k != "cpu_baseline"}
        text = bench.compact_line(d, "bench_detail.json")
        assert "\n" not in text and len(text) < bench.LINE_LIMIT
        line = json.loads(text)
        assert all(k in line for k in bench.CONTRACT_KEYS)
        assert line["value"] == pytest.approx(detail["value"], rel=1e-6) and line["ms_per_step"] == pytest.approx(detail["ms_per_step"], rel=1e-5)
        assert line["n_gpus"] == 1 and line["higher_is_better"] is True and line["vs_baseline"] is None and line["dtype"] == "f64"
        assert set(line["roofline"]) == {"bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "ms_per_launch", "valu_busy"}
        assert line["roofline"]["frac"] == pytest.approx(line["roofline"]["achieved"] / line["roofline"]["peak"], rel=1e-3)
        assert line["config"]["workload"].startswith("BASELINE configs[1]") and "model" not in line["config"]
        if variant == "headline_only":
            assert line["cpu_baseline"] is None and line["extra"] == {}
        else:
            assert set(line["cpu_baseline"]) == {"value", "unit", "cores", "kind", "sample"} and line["cpu_baseline"]["kind"] == "reference"
            assert all(isinstance(v, (int, float)) for v in line["extra"].values())  # scalars only
            for k in ("walker_steps_per_s", "walker_steps_fp64_frac", "c1a_batched_vs_all_cores", "c1a_single_call_vs_1_core", "c3_lc_per_s",
                      "c5_lc_per_s", "c3_table_frac", "implied_8gpu_walkers_8192", "implied_8gpu_walkers_1024", "implied_8gpu_ensemble_c5"):
                assert k in line["extra"], k
        assert line["detail"] == "bench_detail.json"
