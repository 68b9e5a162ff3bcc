"""GPU parity tests (run with -m gpu on an MI355X): every call goes through the C-ABI of
vegasafterglow_amd/libvegasafterglow_amd.so; the CPU oracle is the checker.

Tolerance.  All arithmetic is FP64.  Against the oracle on identical inputs the measured agreement is ~1e-10
relative (transcendental rounding, FMA contraction, summation order); jets with a sharp edge (top-hat,
two-component) reach 2e-7 because the reference's adaptive theta grid integrates a discontinuous CDF whose
step sequence flips with last-bit changes (the reference's own -O1 and -O3 builds differ by 7e-7 there).
Gate: |gpu - oracle| <= 2e-6 * oracle on every bin above 1e-12 of the peak; and the reference's own golden
contract (rtol 2e-3 + atol 1e-2 peak, tests/python/golden/regenerate.py:29-30) on its golden files.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

import _abi
import configs
import vegasafterglow_amd as va
from vegasafterglow_amd import _lib, fitting

pytestmark = pytest.mark.gpu
RTOL = 2e-6
GOLDEN = os.path.join(_abi.ROOT, "tests", "golden")
dp = C.POINTER(C.c_double)


@pytest.fixture(scope="module")
def eng():
    lib = _lib.load()  # raises if the HIP library is missing: no silent fallback
    h, lock = va.get_context(0)
    return lib, h


def gpu_grid(eng, prms, t, nu):
    lib, h = eng
    prms = prms if isinstance(prms, (list, tuple)) else [prms]
    arr = (_lib.ModelParams * len(prms))(*[_lib.ModelParams.from_buffer_copy(bytes(p)) for p in prms])
    t = np.ascontiguousarray(t, dtype=np.float64)
    nu = np.ascontiguousarray(nu, dtype=np.float64)
    out = np.empty((len(prms), nu.size, t.size))
    _lib.check(lib.vag_flux_density_grid_batch(h, arr, len(prms), t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp),
                                               nu.size, out.ctypes.data_as(dp)))
    return out


def gpu_series(eng, prms, t, nu):
    lib, h = eng
    prms = prms if isinstance(prms, (list, tuple)) else [prms]
    arr = (_lib.ModelParams * len(prms))(*[_lib.ModelParams.from_buffer_copy(bytes(p)) for p in prms])
    t = np.ascontiguousarray(t, dtype=np.float64)
    nu = np.ascontiguousarray(nu, dtype=np.float64)
    out = np.empty((len(prms), t.size))
    _lib.check(lib.vag_flux_density_batch(h, arr, len(prms), t.ctypes.data_as(dp), nu.ctypes.data_as(dp), t.size,
                                          out.ctypes.data_as(dp)))
    return out


def assert_close(got, want, rtol=RTOL, floor=1e-12):
    assert np.all(np.isfinite(got))
    m = want > floor * want.max()
    err = np.abs(got - want)[m] / want[m]
    assert err.max() <= rtol, f"max rel err {err.max():.3e}"
    assert np.all(np.abs(got - want)[~m] <= rtol * want.max())


CASES = {"C1a": (configs.C1A, configs.C1_T, configs.C1_NU), "C1b": (configs.C1B, configs.C1_T, configs.C1_NU),
         "C2": (configs.C2, configs.C2_T, configs.C2_NU), "C4": (configs.C4_TRUTH, configs.C4_EPOCHS, configs.C4_BANDS)}
CASES.update(configs.EXTRA)


@pytest.mark.parametrize("name", list(CASES))
def test_grid_matches_oracle(eng, oracle, name):
    kw, t, nu = CASES[name]
    prm = _abi.make_params(**kw)
    assert_close(gpu_grid(eng, prm, t, nu)[0], oracle.flux_density_grid(prm, t, nu))


@pytest.mark.parametrize("name", ["tophat_ism", "tophat_ism_adiabatic", "two_component_ism"])
def test_reference_golden_contract(eng, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    prm = _abi.params_from_golden_config(json.loads(str(g["config"])))
    got = gpu_grid(eng, prm, g["t"], g["nus"])[0]
    want = g["total"]
    assert np.all(np.abs(got - want) <= 2e-3 * np.abs(want) + 1e-2 * np.abs(want).max())
    assert_close(got, want, rtol=1e-6, floor=1e-2)


def test_committed_reference_vectors(eng):
    v = np.load(os.path.join(GOLDEN, "reference_vectors.npz"))
    meta = json.loads(str(v["meta"]))
    for name in ("C1a", "C1b", "C2", "powerlaw_wind", "two_component", "gaussian_p_below_2"):
        kw = dict(meta[name])
        if "resolutions" in kw:
            kw["resolutions"] = tuple(kw["resolutions"])
        assert_close(gpu_grid(eng, _abi.make_params(**kw), v[f"{name}__t"], v[f"{name}__nu"])[0], v[f"{name}__grid"])
    prm = _abi.make_params(**configs.C4_TRUTH)
    assert_close(gpu_series(eng, prm, v["C4__t"], v["C4__nu"])[0], v["C4__series"])


def test_series_band_and_model_api(eng, oracle):
    t, nu = configs.c4_mock_data()
    prm = _abi.make_params(**configs.C4_TRUTH)
    assert_close(gpu_series(eng, prm, t, nu)[0], oracle.flux_density(prm, t, nu))
    kw = configs.C4_TRUTH
    m = va.Model(va.GaussianJet(kw["theta_c"], kw["E_iso"], kw["Gamma0"]), va.ISM(kw["n_ism"]),
                 va.Observer(kw["lumi_dist"], kw["z"], kw["theta_obs"]), va.Radiation(kw["eps_e"], kw["eps_B"], kw["p"]))
    f = m.flux_density(t, nu)
    assert f.total.shape == (t.size,) and f.fwd.ssc.shape == () and f.rvs.sync.shape == () and float(f.fwd.ssc) == 0.0
    assert_close(f.total, oracle.flux_density(prm, t, nu))
    g = m.flux_density_grid(configs.C4_EPOCHS, configs.C4_BANDS)
    assert g.total.shape == (3, 20)  # (n_nu, n_t): pybind.cpp:472-483
    assert_close(g.total, oracle.flux_density_grid(prm, configs.C4_EPOCHS, configs.C4_BANDS))
    b = m.flux(configs.C4_EPOCHS, 1e14, 1e15, 16)
    assert_close(b.total, oracle.flux(prm, configs.C4_EPOCHS, 1e14, 1e15, 16))
    with pytest.raises(ValueError):
        m.flux_density_grid(configs.C4_EPOCHS[::-1].copy(), configs.C4_BANDS)
    with pytest.raises(ValueError):
        m.flux(configs.C4_EPOCHS, 1e15, 1e14, 16)
    d = m.details(t.min(), t.max())
    od = oracle.details(prm, t.min(), t.max())
    assert {k: d["shape"][k] for k in d["shape"]} == {k: od["shape"][k] for k in d["shape"]}
    # angular grid = quantiles of a CDF that both sides integrate adaptively to rtol = 1e-6 (same sensitivity);
    # the time lattice follows theta through t_dec(theta)
    for k in ("phi", "theta", "t_src"):
        np.testing.assert_allclose(d[k], od[k], rtol=2e-6, err_msg=k)
    # blast-wave state: both sides integrate to rtol = 1e-6 with an adaptive step sequence that last-bit
    # differences (FMA contraction, libm vs OCML pow) can shift; agreement is bounded by the ODE tolerance
    for k in ("Gamma", "r", "t_comv", "B", "N_p", "Gamma_th"):
        np.testing.assert_allclose(d[k], od[k], rtol=2e-6, err_msg=k)


def test_exposure_averaged_flux_and_long_series(eng, oracle):
    """Model.flux_density_exposures (pymodel.cpp:412-496): 80 exposures x 10 samples = 800 sorted series points,
    i.e. more than one series launch holds -> chunked on the same grid."""
    kw = configs.C4_TRUTH
    m = va.Model(va.GaussianJet(kw["theta_c"], kw["E_iso"], kw["Gamma0"]), va.ISM(kw["n_ism"]),
                 va.Observer(kw["lumi_dist"], kw["z"], kw["theta_obs"]), va.Radiation(kw["eps_e"], kw["eps_B"], kw["p"]))
    rng = np.random.default_rng(3)
    t = np.sort(10 ** rng.uniform(5.5, 8.0, 80))
    nu = rng.choice(configs.C4_BANDS, 80)
    expo = 10 ** rng.uniform(3, 5.5, 80)
    got = m.flux_density_exposures(t, nu, expo, num_points=10).total
    want = oracle.flux_density_exposures(_abi.make_params(**kw), t, nu, expo, 10)
    assert_close(got, want)
    with pytest.raises(ValueError):
        m.flux_density_exposures(t, nu, expo, num_points=1)
    with pytest.raises(ValueError):
        m.flux_density_exposures(t, nu, -expo)


def test_ragged_batch_equals_individual_calls(eng, oracle):
    """Models with different jets, symmetries and grid sizes in ONE batch give bit-identical results to
    one-at-a-time calls (compact ragged layout, no cross-talk), and match the oracle."""
    t, nu = np.logspace(2.5, 7.5, 40), np.array([1e9, 1e14, 1e17])
    kws = [configs.C1A, configs.C1B, dict(configs.C2, resolutions=(0.1, 0.2, 8.0)), configs.C4_TRUTH,
           configs.EXTRA["two_component"][0], configs.EXTRA["powerlaw_wind"][0], configs.EXTRA["tophat_wind_offaxis"][0]]
    prms = [_abi.make_params(**k) for k in kws]
    batch = gpu_grid(eng, prms, t, nu)
    for i, p in enumerate(prms):
        assert np.array_equal(batch[i], gpu_grid(eng, p, t, nu)[0])
        assert_close(batch[i], oracle.flux_density_grid(p, t, nu))


def test_long_time_axis_is_chunked(eng, oracle):
    prm = _abi.make_params(**configs.C1B)
    t, nu = np.logspace(2, 8, 700), np.logspace(9, 18, 8)  # 5600 slots > one launch's 4096
    assert_close(gpu_grid(eng, prm, t, nu)[0], oracle.flux_density_grid(prm, t, nu))


def _c4_fitter(oracle):
    t, nu = configs.c4_mock_data()
    truth = oracle.flux_density(_abi.make_params(**configs.C4_TRUTH), t, nu)
    rng = np.random.default_rng(42)
    f_obs = truth * (1 + 0.05 * rng.standard_normal(t.size))
    f = fitting.Fitter(z=configs.C4_TRUTH["z"], lumi_dist=configs.C4_TRUTH["lumi_dist"], jet="gaussian", medium="ism")
    for b in configs.C4_BANDS:
        sel = nu == b
        f.add_flux_density(b, t[sel], f_obs[sel], 0.1 * f_obs[sel])
    defs = [fitting.ParamDef(n, 10.0 ** lo if lg else lo, 10.0 ** hi if lg else hi,
                             fitting.Scale.log if lg else fitting.Scale.linear) for n, lg, lo, hi in configs.C4_FREE]
    return f, defs


def test_loglike_batch_matches_fitter_formula_on_oracle_fluxes(eng, oracle):
    """64 prior draws of the C4 problem: ln L from vag_loglike_batch vs -chi2/2 computed with numpy from the
    oracle's Model.flux_density (fitter.py:497-533); invalid / out-of-domain walkers -> -inf."""
    f, defs = _c4_fitter(oracle)
    rng = np.random.default_rng(0)
    _, lo, hi = f.build_spec(defs)  # sampler-space bounds (log10 for LOG-scale parameters)
    samples = lo + (hi - lo) * rng.random((64, len(defs)))
    samples[5, 2] = -0.5  # theta_c < 0: Model construction raises in the reference -> eval_one returns -inf
    got = f.loglike_batch(samples, defs)
    f._consolidate_data()
    want = np.empty(64)
    for i, s in enumerate(samples):
        kw = dict(configs.C4_TRUTH)
        for (name, lg, _, _), v in zip(configs.C4_FREE, s):
            kw[{"theta_v": "theta_obs"}.get(name, name)] = 10 ** v if lg else v
        try:
            F = oracle.flux_density(_abi.make_params(**kw), f._all_t, f._all_nu)
            chi2 = np.sum(f._all_weights * ((f._all_log_flux - np.log(np.maximum(F, 1e-300))) / f._all_log_err) ** 2)
            want[i] = -0.5 * chi2 if np.isfinite(chi2) else -np.inf
        except ValueError:
            want[i] = -np.inf
    assert got[5] == -np.inf and want[5] == -np.inf
    ok = np.isfinite(want)
    draws = []
    for smp in samples[ok]:
        kw = dict(configs.C4_TRUTH)
        for (name, lg, _, _), v in zip(configs.C4_FREE, smp):
            kw[{"theta_v": "theta_obs"}.get(name, name)] = 10 ** v if lg else v
        draws.append(_abi.make_params(**kw))
    _assert_same_grid_shapes(eng, oracle, draws, f._all_t, "C4 prior")  # the adaptive grid's integers, every valid draw
    assert ok.sum() >= 60 and np.array_equal(np.isfinite(got), ok)
    np.testing.assert_allclose(got[ok], want[ok], rtol=1e-5, atol=1e-6)  # chi2 amplifies flux error by |chi|/sigma
    lp = f.make_log_prob_batch(defs)
    out_of_bounds = samples.copy()
    out_of_bounds[0, 0] = 60.0
    assert lp(out_of_bounds)[0] == -np.inf


def test_full_size_properties_on_the_bench_workload(eng):
    """Size-independent properties at BASELINE's full size (C2, 64 x 64 x 199 cells, 200 x 10 outputs), batch of 24:
    run-to-run determinism (bitwise), exact 1/d_L^2 scaling, redshift transformation, series == grid column."""
    import bench
    arr = bench.c2_batch(24, seed=7)
    prms = [arr[i] for i in range(24)]
    t, nu = configs.C2_T, configs.C2_NU
    a = gpu_grid(eng, prms, t, nu)
    assert np.array_equal(a, gpu_grid(eng, prms, t, nu))  # fixed reduction order: bitwise reproducible
    assert np.all(np.isfinite(a)) and np.all(a > 0)
    far = []
    for p in prms:
        q = _lib.ModelParams.from_buffer_copy(bytes(p))
        q.lumi_dist *= 2
        far.append(q)
    np.testing.assert_allclose(a / gpu_grid(eng, far, t, nu), 4.0, rtol=1e-12)
    one = prms[0]
    col = gpu_series(eng, one, t, np.full_like(t, nu[3]))[0]
    np.testing.assert_allclose(col, a[0, 3], rtol=1e-12)
    z1, z2 = 0.2, 1.4
    s = (1 + z2) / (1 + z1)
    p1 = _lib.ModelParams.from_buffer_copy(bytes(one)); p1.z = z1
    p2 = _lib.ModelParams.from_buffer_copy(bytes(one)); p2.z = z2
    np.testing.assert_allclose(gpu_grid(eng, p1, t / s, nu * s)[0] * s, gpu_grid(eng, p2, t, nu)[0], rtol=1e-9)


def test_device_pointer_entry_points_match_host_entry_points(eng):
    import torch
    lib, h = eng
    prm = _abi.make_params(**configs.C1B)
    t, nu = configs.C1_T, configs.C1_NU
    want = gpu_grid(eng, prm, t, nu)[0]
    dev = torch.device("cuda", 0)
    d_p = torch.frombuffer(bytearray(bytes(prm)), dtype=torch.uint8).to(dev)
    d_t, d_nu = torch.from_numpy(t).to(dev), torch.from_numpy(nu).to(dev)
    d_out = torch.zeros((1, nu.size, t.size), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    _lib.check(lib.vag_flux_density_grid_batch_dev(h, d_p.data_ptr(), 1, d_t.data_ptr(), t.size, d_nu.data_ptr(), nu.size,
                                                   d_out.data_ptr()))
    _lib.check(lib.vag_ctx_synchronize(h))
    assert np.array_equal(d_out.cpu().numpy()[0], want)
    plan = _lib.Plan()
    lib.vag_last_plan(h, C.byref(plan))
    assert plan.n_models_ok == 1 and plan.total_pairs == 40 * 26 and plan.n_rows == 1


def test_capacity_and_argument_errors_are_loud(eng):
    prm = _abi.make_params(**dict(configs.C1A, resolutions=(5.0, 0.05, 12.0)))
    with pytest.raises(ValueError, match="capacity"):  # 72 000 phi nodes: beyond even the third layout (32768; 2560 until round 5)
        gpu_grid(eng, _abi.make_params(**dict(configs.C1B, resolutions=(200.0, 0.05, 12.0))), configs.C1_T, configs.C1_NU)
    for _ in range(9):  # (back to the small layout)
        gpu_grid(eng, prm if False else _abi.make_params(**configs.C1A), configs.C1_T, configs.C1_NU)
    with pytest.raises(ValueError):
        gpu_grid(eng, _abi.make_params(theta_c=-1.0), configs.C1_T, configs.C1_NU)
    assert prm.phi_resol == 5.0


# ---------------------------------------------------------------------------------------------------------------
# SSC / inverse-Compton tier: Radiation(ssc=True[, kn=True])
# ---------------------------------------------------------------------------------------------------------------
def gpu_components(eng, prms, t, nu):
    lib, h = eng
    prms = prms if isinstance(prms, (list, tuple)) else [prms]
    arr = (_lib.ModelParams * len(prms))(*[_lib.ModelParams.from_buffer_copy(bytes(p)) for p in prms])
    t = np.ascontiguousarray(t, dtype=np.float64)
    nu = np.ascontiguousarray(nu, dtype=np.float64)
    sync, ssc = np.empty((len(prms), nu.size, t.size)), np.empty((len(prms), nu.size, t.size))
    _lib.check(lib.vag_flux_density_grid_components_batch(h, arr, len(prms), t.ctypes.data_as(dp), t.size,
                                                          nu.ctypes.data_as(dp), nu.size, sync.ctypes.data_as(dp),
                                                          ssc.ctypes.data_as(dp)))
    return sync, ssc


SSC_T = np.logspace(2, 7, 24)
SSC_NU = np.array([1e9, 1e14, 1e17, 1e20, 1e23])
SSC_CASES = {
    "gauss_thomson": dict(jet="GaussianJet", theta_obs=0.2, ssc=True),
    "gauss_kn": dict(jet="GaussianJet", theta_obs=0.2, ssc=True, kn=True),
    "tophat_kn_weakB": dict(E_iso=1e53, n_ism=0.1, eps_B=1e-4, ssc=True, kn=True),
    "powerlaw_wind_kn": dict(jet="PowerLawJet", medium="Wind", A_star=0.1, n_ism=0.0, k_e=2.0, k_g=2.0, theta_obs=0.3,
                             eps_B=1e-3, ssc=True, kn=True),
    "two_component_thomson": dict(jet="TwoComponentJet", theta_c=0.065, theta_w=0.35, E_iso_w=1e50, Gamma0_w=60.0,
                                  theta_obs=0.15, ssc=True, resolutions=(0.59, 0.98, 12.0)),
    "kn_flag_without_ssc": dict(jet="GaussianJet", theta_obs=0.2, kn=True),
}


@pytest.mark.parametrize("name", list(SSC_CASES))
def test_ssc_components_match_oracle(eng, oracle, name):
    prm = _abi.make_params(**SSC_CASES[name])
    want_sync, want_ssc = oracle.flux_components(prm, SSC_T, SSC_NU)
    sync, ssc = gpu_components(eng, prm, SSC_T, SSC_NU)
    assert_close(sync[0], want_sync)
    if want_ssc.max() > 0:
        assert_close(ssc[0], want_ssc)
    else:
        assert np.all(ssc[0] == 0)
    total = gpu_grid(eng, prm, SSC_T, SSC_NU)[0]
    assert_close(total, want_sync + want_ssc)


@pytest.mark.parametrize("name", ["gauss_wind_ssc", "dense_ism_ssa_ssc", "ism_absorbed_slow_ssc"])
def test_ssc_reference_golden_contract(eng, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    prm = _abi.params_from_golden_config(json.loads(str(g["config"])))
    sync, ssc = gpu_components(eng, prm, g["t"], g["nus"])
    for got, want in ((sync[0], g["fwd_sync"]), (ssc[0], g["fwd_ssc"]), (sync[0] + ssc[0], g["total"])):
        assert np.all(np.abs(got - want) <= 2e-3 * np.abs(want) + 1e-2 * np.abs(want).max())
        assert_close(got, want, rtol=2e-6, floor=1e-2)


def test_ssc_committed_reference_vectors(eng):
    v = np.load(os.path.join(GOLDEN, "reference_vectors.npz"))
    meta = json.loads(str(v["meta"]))
    for name in ("C5_central", "ssc_kn_gaussian"):
        kw = dict(meta[name])
        if "resolutions" in kw:
            kw["resolutions"] = tuple(kw["resolutions"])
        t, nu = v[f"{name}__t"], v[f"{name}__nu"]
        sync, ssc = gpu_components(eng, _abi.make_params(**kw), t, nu)
        assert_close(sync[0], v[f"{name}__sync"])
        assert_close(ssc[0], v[f"{name}__ssc"], floor=1e-9)


def test_ssc_batch_band_model_api_and_loud_limits(eng, oracle):
    lib, h = eng
    names = ["gauss_kn", "tophat_kn_weakB", "two_component_thomson"]
    prms = [_abi.make_params(**{**SSC_CASES[n], "ssc": True, "kn": True}) for n in names]
    sync, ssc = gpu_components(eng, prms, SSC_T, SSC_NU)
    for i in range(3):
        s1, c1 = gpu_components(eng, prms[i], SSC_T, SSC_NU)
        assert np.array_equal(sync[i], s1[0]) and np.array_equal(ssc[i], c1[0])  # ragged batch == single calls, bitwise
    # Model.flux (band integral) with SSC: total and components
    t = np.logspace(3, 6, 12)
    m = va.Model(va.GaussianJet(0.1, 1e52, 300), va.ISM(1.0), va.Observer(1e28, 1.0, 0.2),
                 va.Radiation(0.1, 0.01, 2.3, ssc=True, kn=True))
    prm = _abi.make_params(jet="GaussianJet", theta_obs=0.2, ssc=True, kn=True)
    band = m.flux(t, 1e17, 1e19, 9)
    assert_close(band.total, oracle.flux(prm, t, 1e17, 1e19, 9))
    assert np.all(band.fwd.ssc > 0) and np.allclose(band.total, band.fwd.sync + band.fwd.ssc, rtol=1e-15)
    fd = m.flux_density_grid(SSC_T, SSC_NU)
    o_sync, o_ssc = oracle.flux_components(prm, SSC_T, SSC_NU)
    assert_close(fd.fwd.sync, o_sync)
    assert_close(fd.fwd.ssc, o_ssc)
    assert_close(fd.total, o_sync + o_ssc)
    # mixed Radiation flags in one batch: split inside the call (test_mixed_flag_batches_are_split_inside_the_call)
    import torch
    mixed = (_lib.ModelParams * 2)(_lib.ModelParams.from_buffer_copy(bytes(prm)),
                                   _lib.ModelParams.from_buffer_copy(bytes(_abi.make_params(jet="GaussianJet"))))
    out = np.empty((2, SSC_NU.size, SSC_T.size))
    assert lib.vag_flux_density_grid_batch(h, mixed, 2, SSC_T.ctypes.data_as(dp), SSC_T.size, SSC_NU.ctypes.data_as(dp),
                                           SSC_NU.size, out.ctypes.data_as(dp)) == 0
    assert_close(out[0], o_sync + o_ssc)
    dev = torch.device("cuda", 0)
    d_p = torch.frombuffer(bytearray(bytes(mixed)), dtype=torch.uint8).to(dev)
    d_t, d_nu = torch.from_numpy(SSC_T).to(dev), torch.from_numpy(SSC_NU).to(dev)
    d_o = torch.empty((2, SSC_NU.size, SSC_T.size), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    # the device-pointer entry point regroups such a batch by flags itself (test_mixed_flag_batches_on_the_device_pointer_entry_points)
    _lib.check(lib.vag_flux_density_grid_batch_dev(h, d_p.data_ptr(), 2, d_t.data_ptr(), SSC_T.size, d_nu.data_ptr(), SSC_NU.size, d_o.data_ptr()))
    torch.cuda.synchronize()
    assert np.array_equal(d_o.cpu().numpy(), out)


def test_ssc_series_and_loglike(eng, oracle):
    """Model.flux_density and the Fitter log-likelihood with Radiation(ssc=True, kn=True)."""
    rng = np.random.default_rng(5)
    t = np.sort(10 ** rng.uniform(2.5, 6.5, 90))
    nu = 10 ** rng.choice([9.0, 14.7, 17.5, 22.0, 25.0], size=t.size)
    prms = [_abi.make_params(**SSC_CASES[n]) for n in ("gauss_kn", "tophat_kn_weakB", "powerlaw_wind_kn")]
    got = gpu_series(eng, prms, t, nu)
    for i, prm in enumerate(prms):
        assert_close(got[i], oracle.flux_density(prm, t, nu))
    # Fitter(fwd_ssc=True, kn=True): ln L from the device == the fitter formula on the checker's fluxes
    f = fitting.Fitter(z=1.0, lumi_dist=1e28, jet="gaussian", medium="ism", fwd_ssc=True, kn=True)
    truth = _abi.make_params(jet="GaussianJet", theta_obs=0.2, ssc=True, kn=True)
    f_obs = oracle.flux_density(truth, t, nu) * (1 + 0.05 * rng.standard_normal(t.size))
    f.add_flux_density(nu, t, f_obs, 0.05 * f_obs)
    P, S = fitting.ParamDef, fitting.Scale
    defs = [P("E_iso", 1e51, 1e53, S.log), P("Gamma0", 100, 500, S.log), P("theta_c", 0.05, 0.2, S.linear),
            P("theta_v", 0.0, 0.4, S.linear), P("n_ism", 0.1, 10, S.log), P("p", 2.1, 2.6, S.linear),
            P("eps_e", 0.03, 0.3, S.log), P("eps_B", 1e-3, 1e-1, S.log), P("xi_e", 1.0, 1.0, S.fixed, 1.0)]
    lo = np.array([np.log10(d.lower) if d.scale is S.log else d.lower for d in defs[:8]])
    hi = np.array([np.log10(d.upper) if d.scale is S.log else d.upper for d in defs[:8]])
    theta = lo + (hi - lo) * rng.random((6, 8))
    ll = f.loglike_batch(theta, defs)
    f._consolidate_data()
    for w in range(theta.shape[0]):
        v = [10 ** x if d.scale is S.log else x for x, d in zip(theta[w], defs[:8])]
        prm = _abi.make_params(jet="GaussianJet", E_iso=v[0], Gamma0=v[1], theta_c=v[2], theta_obs=v[3], n_ism=v[4], p=v[5],
                               eps_e=v[6], eps_B=v[7], ssc=True, kn=True)
        model = oracle.flux_density(prm, f._all_t, f._all_nu)
        chi2 = np.sum(f._all_weights * ((np.log(model) - f._all_log_flux) / f._all_log_err) ** 2)
        assert abs(ll[w] - (-0.5 * chi2)) <= 1e-5 * max(1.0, abs(chi2)), (w, ll[w], -0.5 * chi2)


def test_ssc_loglike_with_a_rejected_walker_between_valid_ones(eng, oracle):
    """A walker outside the model's domain (theta_c < 0: the reference's eval_one returns -inf, samplers.py:61-70) in the middle
    of an SSC batch: its cells get no lattice plan and no table (vag_ic_plan_kernel finishes them), the table memory still
    holds a larger, earlier call's rows, and the walkers on either side must score exactly what they score alone."""
    rng = np.random.default_rng(11)
    t = np.sort(10 ** rng.uniform(3.0, 6.0, 40))
    nu = 10 ** rng.choice([14.7, 17.5, 23.0], size=t.size)
    f = fitting.Fitter(z=1.0, lumi_dist=1e28, jet="gaussian", medium="ism", fwd_ssc=True, kn=True)
    truth = _abi.make_params(jet="GaussianJet", theta_obs=0.2, ssc=True, kn=True)
    f_obs = oracle.flux_density(truth, t, nu) * (1 + 0.05 * rng.standard_normal(t.size))
    f.add_flux_density(nu, t, f_obs, 0.05 * f_obs)
    P, S = fitting.ParamDef, fitting.Scale
    defs = [P("E_iso", 1e51, 1e53, S.log), P("Gamma0", 100, 500, S.log), P("theta_c", -1.0, 0.2, S.linear),
            P("theta_v", 0.0, 0.4, S.linear), P("n_ism", 0.1, 10, S.log), P("p", 2.1, 2.6, S.linear),
            P("eps_e", 0.03, 0.3, S.log), P("eps_B", 1e-3, 1e-1, S.log), P("xi_e", 1.0, 1.0, S.fixed, 1.0)]
    lo = np.array([51, 2.0, 0.05, 0.0, -1, 2.1, np.log10(0.03), -3])
    hi = np.array([53, np.log10(500), 0.2, 0.4, 1, 2.6, np.log10(0.3), -1])
    big = lo + (hi - lo) * rng.random((12, 8))
    assert np.all(np.isfinite(f.loglike_batch(big, defs)))  # leaves 12 walkers' tables behind
    theta = big[:5].copy()
    theta[2, 2] = -0.5  # theta_c < 0
    ll = f.loglike_batch(theta, defs)
    assert np.isneginf(ll[2]) and f.last_plan.n_walkers_rejected == 1
    for w in (0, 1, 3, 4):
        alone = f.loglike_batch(theta[w:w + 1], defs)
        assert np.isfinite(alone[0]) and ll[w] == alone[0], (w, ll[w], alone[0])


@pytest.mark.parametrize("jet,medium,kn", [("gaussian", "ism", True), ("tophat", "wind", True), ("powerlaw", "ism", False)])
def test_no_walker_of_a_wide_ssc_box_leaves_its_clamped_table_band(eng, jet, medium, kn):
    """Inside a likelihood call a walker whose SSC flux pass asks for a frequency outside its cells' clamped table band (but inside the
    theoretical range) is not rebuilt unclamped as Model's entry points and the reference do (inverse-compton.h:626-635): it scores
    -inf and is counted in n_walkers_ssc_failed.  The band is the extrema of the rows' Doppler factors widened by two octaves either
    way (pymodel.h:896-909), the row-per-lane kernels' table logarithms are good to 1e-15, so no walker should ever get there: 512
    draws of a wide box, data from radio to TeV over five decades of time, none fails and every ln L is finite."""
    rng = np.random.default_rng(2024)
    t = np.sort(10 ** rng.uniform(2.0, 7.0, 48))
    nu = 10 ** rng.choice([9.0, 11.0, 14.7, 17.5, 20.0, 23.0, 26.0], size=t.size)
    f = fitting.Fitter(z=0.5, lumi_dist=8e27, jet=jet, medium=medium, fwd_ssc=True, kn=kn)
    f.add_flux_density(nu, t, np.full(t.size, 1e-28), np.full(t.size, 1e-29))
    P, S = fitting.ParamDef, fitting.Scale
    defs = [P("E_iso", 1e50, 1e54, S.log), P("Gamma0", 30, 800, S.log), P("theta_c", 0.03, 0.3, S.linear),
            P("theta_v", 0.0, 0.6, S.linear), P("p", 2.05, 2.9, S.linear), P("eps_e", 3e-3, 0.4, S.log), P("eps_B", 1e-6, 1e-1, S.log),
            P("xi_e", 1.0, 1.0, S.fixed, 1.0)]
    defs.append(P("A_star", 1e-2, 3, S.log) if medium == "wind" else P("n_ism", 1e-3, 100, S.log))
    if jet == "powerlaw":
        defs += [P("k_e", 2.0, 2.0, S.fixed, 2.0), P("k_g", 2.0, 2.0, S.fixed, 2.0)]
    free = [d for d in defs if d.scale is not S.fixed]
    lo = np.array([np.log10(d.lower) if d.scale is S.log else d.lower for d in free])
    hi = np.array([np.log10(d.upper) if d.scale is S.log else d.upper for d in free])
    theta = lo + (hi - lo) * rng.random((512, len(free)))
    ll = f.loglike_batch(theta, defs)
    assert f.last_plan.n_walkers_ssc_failed == 0 and f.last_plan.n_walkers_rejected == 0
    assert np.all(np.isfinite(ll))


# ---------------------------------------------------------------------------------------------------------------
# Reverse-shock tier: Model(rvs_rad=Radiation(...))
#
# Tolerances.  Sharp-edged and power-law jets agree with the oracle like the forward-only path (<= 2e-6, measured
# 1e-12 .. 3e-7).  Gaussian jets with a reverse shock are the reference's own sensitive case: its -O3 and strict builds
# differ by 7e-3 in the reverse-shock flux of gauss_ism_rs (low-Gamma wing rows amplify last-bit noise in the coupled
# ODE; tests/python/test_golden.py:94-95 of the reference expects ~1 %), so those are held to the reference's golden
# contract (rtol 2e-3 + atol 1e-2 peak) AND to what the reference itself demonstrates on the same input
# (tests/golden/rs_one_ulp_sensitivity.json, written by profiles/rs_one_ulp_sensitivity.py: the larger of the spread between
# its two builds and the response of either build to a ONE-ulp move of theta_c / Gamma0 / E_iso / theta_obs -- 1.8e-3, 6.4e-3
# and 1.1e-2 in the reverse-shock flux of the three cases, 3e-8 on a top-hat jet).  Those numbers are the maximum of a handful of
# draws of a heavy-tailed quantity, so the gate is 3 x them; this engine sits at 1.1 x ... 1.3 x (profiles/r03_rs_lib_history.txt).
# ---------------------------------------------------------------------------------------------------------------
COMPONENTS = ("fwd_sync", "fwd_ssc", "rvs_sync", "rvs_ssc")
with open(os.path.join(GOLDEN, "rs_one_ulp_sensitivity.json")) as _f:
    RS_DEMONSTRATED = json.load(_f)


def within_demonstrated(case, comp, got, want):
    """rel. error over the bins above 1e-2 of the peak <= 3 x the reference's own demonstrated sensitivity on this case."""
    m = want > 1e-2 * want.max()
    err = float(np.max(np.abs(got - want)[m] / want[m]))
    return err <= 3 * RS_DEMONSTRATED[case][comp.replace("_", ".")], err


def gpu_components4(eng, prms, t, nu):
    lib, h = eng
    prms = prms if isinstance(prms, (list, tuple)) else [prms]
    arr = (_lib.ModelParams * len(prms))(*[_lib.ModelParams.from_buffer_copy(bytes(p)) for p in prms])
    t = np.ascontiguousarray(t, dtype=np.float64)
    nu = np.ascontiguousarray(nu, dtype=np.float64)
    comps = [np.empty((len(prms), nu.size, t.size)) for _ in range(4)]
    out4 = (dp * 4)(*[a.ctypes.data_as(dp) for a in comps])
    _lib.check(lib.vag_flux_density_grid_components4_batch(h, arr, len(prms), t.ctypes.data_as(dp), t.size,
                                                           nu.ctypes.data_as(dp), nu.size, out4))
    return comps


def within_contract(got, want):
    return np.all(np.abs(got - want) <= 2e-3 * np.abs(want) + 1e-2 * np.abs(want).max())


RS_TIGHT = ["rs_thin_tophat", "rs_thick_offaxis", "rs_two_component", "rs_tophat_both_ssc_kn"]


@pytest.mark.parametrize("name", ["C3"] + list(configs.RS_CASES))
def test_rs_components_match_oracle(eng, oracle, name):
    kw, t, nu = {"C3": (configs.C3, configs.C3_T, configs.C3_NU)}.get(name) or configs.RS_CASES[name]
    prm = _abi.make_params(**kw)
    want = oracle.flux_components4(prm, t, nu)
    got = gpu_components4(eng, prm, t, nu)
    total = gpu_grid(eng, prm, t, nu)[0]
    for g, w, comp in zip(got, want, COMPONENTS):
        if w.max() == 0:
            assert np.all(g[0] == 0), comp
        elif name == "rs_gaussian_adiabatic":
            assert within_contract(g[0], w), comp
            ok, err = within_demonstrated(name, comp, g[0], w)
            assert ok, (comp, err)
        else:
            assert_close(g[0], w)
    w_total = want[0] + want[1] + want[2] + want[3]
    assert within_contract(total, w_total)
    if name != "rs_gaussian_adiabatic":
        assert_close(total, w_total)


@pytest.mark.parametrize("name", ["rs_thick", "gauss_ism_rs", "powerlaw_wind_rs", "tophat_sigma_rs", "tophat_sigma1_rs",
                                  "tophat_sigma10_rs"])
def test_rs_reference_golden_contract(eng, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    prm = _abi.params_from_golden_config(json.loads(str(g["config"])))
    got = dict(zip(COMPONENTS, (c[0] for c in gpu_components4(eng, prm, g["t"], g["nus"]))))
    for comp in ("fwd_sync", "rvs_sync"):
        assert within_contract(got[comp], g[comp]), comp
        if name != "gauss_ism_rs":
            assert_close(got[comp], g[comp], rtol=2e-6, floor=1e-2)
        else:
            ok, err = within_demonstrated(name, comp, got[comp], g[comp])
            assert ok, (comp, err)
    assert within_contract(gpu_grid(eng, prm, g["t"], g["nus"])[0], g["total"])
    assert np.all(got["fwd_ssc"] == 0) and np.all(got["rvs_ssc"] == 0)


@pytest.mark.parametrize("sigma0", [0.1, 1.0, 10.0])
def test_magnetized_tophat_matches_oracle(eng, oracle, sigma0):
    """sigma > 0: cubic jump condition, magnetosonic crossing cap, 10x tighter ODE tolerance (reverse-shock.tpp:529-537)."""
    prm = _abi.make_params(jet="MagnetizedTophat", sigma0=sigma0, theta_obs=0.05, z=0.5, lumi_dist=3e28, eps_B=1e-3,
                           rvs=dict(eps_e=0.1, eps_B=0.01, p=2.5))
    t, nu = np.logspace(0, 7, 36), np.array([1e9, 4.84e14, 1e18])
    want = oracle.flux_components4(prm, t, nu)
    got = gpu_components4(eng, prm, t, nu)
    assert_close(got[0][0], want[0])
    assert_close(got[2][0], want[2], rtol=2e-5)  # the magnetised solve is the reference's tolerance-sensitive case
    m = va.Model(va.MagnetizedTophatJet(0.1, 1e52, 300.0, sigma0), va.ISM(1.0), va.Observer(3e28, 0.5, 0.05),
                 va.Radiation(0.1, 1e-3, 2.3), rvs_rad=va.Radiation(0.1, 0.01, 2.5))
    assert_close(m.flux_density_grid(t, nu).rvs.sync, want[2], rtol=2e-5)


def test_rs_committed_reference_vectors(eng):
    v = np.load(os.path.join(GOLDEN, "reference_vectors_rs.npz"))
    meta = json.loads(str(v["meta"]))
    for name in ["C3"] + list(configs.RS_CASES):
        kw = dict(meta[name])
        if "resolutions" in kw:
            kw["resolutions"] = tuple(kw["resolutions"])
        prm = _abi.make_params(**kw)
        got = gpu_components4(eng, prm, v[f"{name}__t"], v[f"{name}__nu"])
        for g, comp in zip(got, COMPONENTS):
            want = v[f"{name}__{comp}"]
            if want.max() == 0:
                assert np.all(g[0] == 0)
            else:
                assert within_contract(g[0], want), (name, comp)
                if name in RS_TIGHT:
                    assert_close(g[0], want, rtol=2e-4, floor=1e-3)  # vectors come from the reference-flag build
    prm = _abi.make_params(**{**meta["rs_thick_offaxis"]})
    assert_close(gpu_series(eng, prm, v["series__t"], v["series__nu"])[0], v["series__flux"], rtol=1e-5, floor=1e-3)


def test_rs_series_band_loglike_model_api_and_batch(eng, oracle):
    lib, h = eng
    kw, t, nu = configs.RS_CASES["rs_thick_offaxis"]
    prm = _abi.make_params(**kw)
    ts, nus = np.repeat(t, 2), np.tile(nu[[0, 2]], t.size)
    assert_close(gpu_series(eng, prm, ts, nus)[0], oracle.flux_density(prm, ts, nus))
    # Model mirror: FluxDict.rvs components, band integral, reverse-shock details
    m = va.Model(va.TophatJet(0.1, 1e53, 100.0, duration=1000.0), va.ISM(1.0), va.Observer(3e28, 0.5, 0.15),
                 va.Radiation(0.1, 1e-3, 2.3), rvs_rad=va.Radiation(0.1, 0.01, 2.5))
    assert m.resolutions == (0.06, 0.2, 10.0) and m.params.flags == prm.flags == 4
    want = oracle.flux_components4(prm, t, nu)
    fd = m.flux_density_grid(t, nu)
    assert_close(fd.fwd.sync, want[0])
    assert_close(fd.rvs.sync, want[2])
    assert fd.fwd.ssc.shape == () and fd.rvs.ssc.shape == ()  # disabled components are 0-d zeros like the reference's
    assert_close(fd.total, want[0] + want[2])
    band = m.flux(t, 1e17, 1e19, 9)
    assert_close(band.total, oracle.flux(prm, t, 1e17, 1e19, 9))
    assert np.allclose(band.total, band.fwd.sync + band.rvs.sync, rtol=1e-15) and band.rvs.sync.max() > 0
    d, o = m.details(t.min(), t.max(), rvs=True), oracle.details(prm, t.min(), t.max(), rvs=True)
    for k in ("t_src", "Gamma", "r", "B", "N_p", "Gamma_th"):
        np.testing.assert_allclose(d[k], o[k], rtol=2e-6, atol=1e-300, err_msg=k)
    # ragged batch of reverse-shock models == single calls, bitwise
    prms = [_abi.make_params(**configs.RS_CASES[n][0]) for n in ("rs_thin_tophat", "rs_thick_offaxis", "rs_two_component")]
    tb, nub = np.logspace(1, 7, 30), np.array([1e9, 1e14, 1e17])
    got = gpu_components4(eng, prms, tb, nub)
    for i, p in enumerate(prms):
        one = gpu_components4(eng, p, tb, nub)
        assert all(np.array_equal(got[c][i], one[c][0]) for c in range(4))
    # Fitter(rvs_shock=True): ln L from the device == the fitter formula on the checker's fluxes
    f = fitting.Fitter(z=0.5, lumi_dist=3e28, jet="tophat", medium="ism", rvs_shock=True)
    rng = np.random.default_rng(11)
    f_obs = oracle.flux_density(prm, ts, nus) * (1 + 0.05 * rng.standard_normal(ts.size))
    f.add_flux_density(nus, ts, f_obs, 0.05 * f_obs)
    P, S = fitting.ParamDef, fitting.Scale
    defs = [P("E_iso", 1e52, 1e54, S.log), P("Gamma0", 50, 300, S.log), P("theta_c", 0.05, 0.2, S.linear),
            P("theta_v", 0.0, 0.3, S.linear), P("n_ism", 0.1, 10, S.log), P("eps_B", 1e-4, 1e-2, S.log),
            P("eps_B_r", 1e-3, 1e-1, S.log), P("p_r", 2.1, 2.8, S.linear), P("tau", 1000.0, 1000.0, S.fixed, 1000.0)]
    _, lo, hi = f.build_spec(defs)
    theta = lo + (hi - lo) * rng.random((5, 8))
    ll = f.loglike_batch(theta, defs)
    f._consolidate_data()
    for w in range(theta.shape[0]):
        v = [10 ** x if d.scale is S.log else x for x, d in zip(theta[w], defs[:8])]
        q = _abi.make_params(E_iso=v[0], Gamma0=v[1], theta_c=v[2], theta_obs=v[3], n_ism=v[4], eps_B=v[5], duration=1000.0,
                             z=0.5, lumi_dist=3e28, rvs=dict(eps_e=0.1, eps_B=v[6], p=v[7]))
        model = oracle.flux_density(q, f._all_t, f._all_nu)
        chi2 = np.sum(f._all_weights * ((np.log(model) - f._all_log_flux) / f._all_log_err) ** 2)
        assert abs(ll[w] - (-0.5 * chi2)) <= 1e-5 * max(1.0, abs(chi2)), (w, ll[w], -0.5 * chi2)


def test_end_to_end_mcmc_on_the_device_likelihood(eng, oracle):
    """Affine-invariant MCMC (vegasafterglow_amd/sampling.py) on the C4 mock problem: every half-step is one
    vag_loglike_batch call; the chain must climb to the truth's likelihood level and stay inside the prior box."""
    from vegasafterglow_amd import sampling
    f, defs = _c4_fitter(oracle)
    _, lo, hi = f.build_spec(defs)
    truth = np.array([np.log10(configs.C4_TRUTH[{"theta_v": "theta_obs"}.get(n, n)]) if lg
                      else configs.C4_TRUTH[{"theta_v": "theta_obs"}.get(n, n)] for n, lg, _, _ in configs.C4_FREE])
    ll_truth = f.loglike_batch(truth[None, :], defs)[0]
    res = sampling.fit(f, defs, nwalkers=32, nsteps=40, nburn=20, seed=1, center=truth + 0.05 * (hi - lo), spread=0.02)
    assert res["samples"].shape == (20 * 32, len(defs)) and np.all(np.isfinite(res["log_prob"]))
    assert np.all(res["samples"] >= lo) and np.all(res["samples"] <= hi)
    assert res["log_prob"].max() > ll_truth - 40  # started ~5 % of the box away; walked back toward the truth
    assert res["chain"][-1].std(axis=0).min() > 0 and 0.05 < res["acceptance"].mean() < 0.95


# ---------------------------------------------------------------------------------------------------------------
# Spreading jets: jet(..., spreading=True)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(configs.SPREAD_CASES))
def test_spreading_components_match_oracle(eng, oracle, name):
    prm = _abi.make_params(**configs.SPREAD_CASES[name])
    t, nu = configs.SPREAD_T, configs.SPREAD_NU
    want = oracle.flux_components4(prm, t, nu)
    got = gpu_components4(eng, prm, t, nu)
    for g, w, comp in zip(got, want, COMPONENTS):
        if w.max() == 0:
            assert np.all(g[0] == 0), comp
        else:
            assert_close(g[0], w, rtol=5e-6)
    ts, nus = np.repeat(t, 2), np.tile(nu[[0, 2]], t.size)
    assert_close(gpu_series(eng, prm, ts, nus)[0], oracle.flux_density(prm, ts, nus), rtol=5e-6)


def test_spreading_model_api_batch_and_vectors(eng, oracle):
    v = np.load(os.path.join(GOLDEN, "reference_vectors_rs.npz"))
    t, nu = configs.SPREAD_T, configs.SPREAD_NU
    m = va.Model(va.GaussianJet(0.1, 1e52, 300.0, spreading=True), va.ISM(1.0), va.Observer(1e28, 1.0, 0.15),
                 va.Radiation(0.1, 0.01, 2.3))
    assert m.params.flags == 32
    got = m.flux_density_grid(t, nu).total
    assert within_contract(got, v["gauss_spread__total"])
    assert_close(got, v["gauss_spread__total"], rtol=1e-4, floor=1e-3)
    d = m.details(t.min(), t.max())
    assert d["shape"]["symmetry"] == 0 and d["shape"]["n_reps"] == d["shape"]["n_theta"]
    names = ["tophat_spread_onaxis", "tophat_spread_offaxis", "two_comp_spread"]
    prms = [_abi.make_params(**configs.SPREAD_CASES[n]) for n in names]
    batch = gpu_grid(eng, prms, t, nu)
    for i, p in enumerate(prms):
        assert np.array_equal(batch[i], gpu_grid(eng, p, t, nu)[0])
        assert_close(batch[i], oracle.flux_density_grid(p, t, nu), rtol=5e-6)


@pytest.mark.parametrize("name", list(configs.PROFILE_CASES))
def test_remaining_profiles_match_oracle(eng, oracle, name):
    """StepPowerLawJet, PowerLawWing, Wind(k_m != 2) -- the rest of the reference's jet / medium registry."""
    prm = _abi.make_params(**configs.PROFILE_CASES[name])
    t, nu = configs.SPREAD_T, configs.SPREAD_NU
    want = oracle.flux_components4(prm, t, nu)
    got = gpu_components4(eng, prm, t, nu)
    for g_, w, comp in zip(got, want, COMPONENTS):
        if w.max() == 0:
            assert np.all(g_[0] == 0), comp
        elif name == "step_powerlaw_rs_spread":
            # structured jet + reverse shock: the reference's own builds differ by 4e-4 (fwd) / 5e-3 (rvs) here
            assert within_contract(g_[0], w), comp
            ok, err = within_demonstrated(name, comp, g_[0], w)
            assert ok, (comp, err)
        else:
            assert_close(g_[0], w, rtol=5e-6)


def test_remaining_profiles_python_mirror(eng, oracle):
    t, nu = configs.SPREAD_T, configs.SPREAD_NU
    obs, rad = va.Observer(1e28, 1.0, 0.2), va.Radiation(0.1, 0.01, 2.3)
    m = va.Model(va.StepPowerLawJet(0.05, 1e52, 300.0, 3e51, 100.0, 3.0, 2.0), va.ISM(1.0), obs, rad)
    assert_close(m.flux_density_grid(t, nu).total, oracle.flux_density_grid(_abi.make_params(**configs.PROFILE_CASES["step_powerlaw"]), t, nu))
    m = va.Model(va.PowerLawWing(0.05, 3e51, 100.0, 3.0, 2.0), va.ISM(1.0), obs, rad)
    assert_close(m.flux_density_grid(t, nu).total, oracle.flux_density_grid(_abi.make_params(**configs.PROFILE_CASES["powerlaw_wing"]), t, nu))
    m = va.Model(va.TophatJet(0.1, 1e52, 300.0), va.Wind(0.3, k_m=1.5), va.Observer(1e28, 1.0, 0.1), rad)
    assert_close(m.flux_density_grid(t, nu).total, oracle.flux_density_grid(_abi.make_params(**configs.PROFILE_CASES["wind_k1.5"]), t, nu))
    f = fitting.Fitter(z=1.0, lumi_dist=1e28, jet="uniform", medium="wind")
    assert f._base_params({"A_star": 0.1, "k_m": 1.5}).theta_c == np.pi / 2 and f._base_params({"k_m": 1.5}).k_m == 1.5


def test_long_time_axis_is_chunked_for_every_tier(eng, oracle):
    """nt * nnu > 4096 with SSC and a reverse shock: the time axis is processed in chunks on the same grid."""
    prm = _abi.make_params(theta_obs=0.1, duration=100.0, ssc=True, kn=True, rvs=dict(eps_e=0.1, eps_B=0.01, p=2.3, ssc=True))
    t, nu = np.logspace(1, 7, 1500), np.array([1e9, 1e14, 1e17, 1e22])
    assert_close(gpu_grid(eng, prm, t, nu)[0], oracle.flux_density_grid(prm, t, nu))


@pytest.mark.parametrize("name", list(configs.MAGNETAR_CASES))
def test_magnetar_matches_oracle(eng, oracle, name):
    prm = _abi.make_params(**configs.MAGNETAR_CASES[name])
    t, nu = configs.SPREAD_T, configs.SPREAD_NU
    want = oracle.flux_components4(prm, t, nu)
    got = gpu_components4(eng, prm, t, nu)
    for g_, w, comp in zip(got, want, COMPONENTS):
        if w.max() == 0:
            assert np.all(g_[0] == 0), comp
        else:
            assert_close(g_[0], w, rtol=5e-6)
    if name == "gauss_mag_offaxis":
        m = va.Model(va.GaussianJet(0.1, 1e52, 300.0, magnetar=va.Magnetar(*configs.MAG)), va.ISM(1.0), va.Observer(1e28, 1.0, 0.2),
                     va.Radiation(0.1, 0.01, 2.3))
        assert_close(m.flux_density_grid(t, nu).total, want[0], rtol=5e-6)


def test_loglike_with_band_groups_extinction_and_python_fitter(eng, oracle):
    """vag_loglike_batch with band-integrated groups (own grid per group), extinction and a free A_V against the checker,
    then the same through Fitter.add_flux / extinction= / ParamDef("A_V")."""
    import test_oracle
    lib, h = eng
    kw = dict(jet="GaussianJet", z=0.5, lumi_dist=3e27)
    rng = np.random.default_rng(2)
    for with_points in (True, False):
        spec, keep = test_oracle._band_spec(kw, with_points=with_points)
        theta = np.column_stack([rng.uniform(51.5, 52.8, 6), rng.uniform(0.0, 0.3, 6), rng.uniform(0.0, 1.0, 6)])
        want = test_oracle.oracle_loglike(spec, theta)
        got = np.empty(6)
        _lib.check(lib.vag_loglike_batch(h, C.byref(spec), theta.ctypes.data_as(dp), 6, 3, got.ctypes.data_as(dp)))
        np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-6)
    # Python mirror
    f = fitting.Fitter(z=0.5, lumi_dist=3e27, jet="gaussian", medium="ism", extinction=lambda lam: (5.5e-5 / lam) ** 1.1)
    spec, keep = test_oracle._band_spec(kw)
    t, nu, lnf, lne, w, ext = keep["pts"]
    f.add_flux_density(nu, t, np.exp(lnf), lne * np.exp(lnf))
    for g in range(2):
        tb, lnfb, lneb, wb = keep[f"b{g}"]
        bd = keep["bands"][g]
        f.add_flux((bd.nu_min, bd.nu_max), tb, np.exp(lnfb), lneb * np.exp(lnfb), num_points=bd.num_points)
    P, S = fitting.ParamDef, fitting.Scale
    defs = [P("E_iso", 1e51, 1e53, S.log), P("theta_v", 0.0, 0.4, S.linear), P("A_V", 0.0, 2.0, S.linear),
            P("n_ism", 1.0, 1.0, S.fixed, 1.0)]
    theta = np.array([[52.3, 0.15, 0.4], [51.8, 0.05, 0.0]])
    ll = f.loglike_batch(theta, defs)
    f._consolidate_data()
    for b, (lgE, thv, av) in enumerate(theta):
        prm = _abi.make_params(**{**kw, "E_iso": 10 ** lgE, "theta_obs": thv})
        F = oracle.flux_density(prm, f._all_t, f._all_nu) * np.exp(-av * f._ext_kernel)
        chi2 = np.sum(f._all_weights * ((f._all_log_flux - np.log(np.maximum(F, 1e-300))) / f._all_log_err) ** 2)  # fitter.py:499
        for bd in f._band_obs:
            Fb = oracle.flux(prm, bd["t"], bd["nu_min"], bd["nu_max"], bd["num_points"])
            chi2 += np.sum(bd["weights"] * ((bd["ln_flux"] - np.log(Fb)) / bd["ln_err"]) ** 2)
        assert abs(ll[b] + 0.5 * chi2) <= 2e-6 * abs(chi2), (b, ll[b], -0.5 * chi2)


_U_SEC = 3e10 / 1.5e13
_U_HZ = 1 / _U_SEC
_U_FLUX_DEN = ((1 / 2e33) * (1 / 1.5e13) ** 2 / _U_SEC ** 2) / (1 / 1.5e13) ** 2 / _U_SEC / _U_HZ


def _check_radiation_details(d, o, rtol):
    """ShockDetails electron / photon arrays (pybind.cpp:522-546) against the checker's per-cell spectra."""
    for k in ("gamma_m", "gamma_c", "gamma_a", "gamma_M", "N_e"):
        np.testing.assert_allclose(d[k], o[k], rtol=rtol, atol=1e-300, equal_nan=True, err_msg=k)
    for k in ("nu_m", "nu_c", "nu_a", "nu_M"):
        np.testing.assert_allclose(d[k], o[k] / _U_HZ, rtol=rtol, atol=1e-300, equal_nan=True, err_msg=k)
    np.testing.assert_allclose(d["I_nu_max"], o["I_nu_max"] / _U_FLUX_DEN, rtol=rtol, atol=1e-300, equal_nan=True)


def test_details_radiation_arrays(eng, oracle):
    kw = configs.C4_TRUTH
    prm = _abi.make_params(**kw)
    m = va.Model(va.GaussianJet(kw["theta_c"], kw["E_iso"], kw["Gamma0"]), va.ISM(kw["n_ism"]),
                 va.Observer(kw["lumi_dist"], kw["z"], kw["theta_obs"]), va.Radiation(kw["eps_e"], kw["eps_B"], kw["p"]))
    d, o = m.details(1e2, 1e7), oracle.details(prm, 1e2, 1e7)
    _check_radiation_details(d, o, 5e-6)
    assert np.array_equal(d["theta_cell"], np.repeat(d["theta"][:, None], d["shape"]["n_t"], axis=1))
    # ShockDetails.t_obs / .Doppler: the linear forms of the equal-arrival-time logs, every (phi, theta, t) cell
    assert d["t_obs"].shape == o["lg2_t"].shape == (o["shape"]["n_phi_eff"], d["shape"]["n_theta"], d["shape"]["n_t"])
    np.testing.assert_allclose(d["t_obs"], np.exp2(o["lg2_t"]) / _U_SEC, rtol=5e-6)
    np.testing.assert_allclose(d["Doppler"], np.exp2(o["lg2_doppler"]), rtol=5e-6)
    # inverse-Compton-cooled electrons of a Radiation(ssc=True, kn=True) model
    m3 = va.Model(va.TophatJet(0.1, 1e52, 300.0), va.ISM(1.0), va.Observer(1e28, 1.0, 0.05),
                  va.Radiation(0.1, 1e-4, 2.3, ssc=True, kn=True))
    prm3 = _abi.ModelParams.from_buffer_copy(bytes(m3.params))
    _check_radiation_details(m3.details(1e2, 1e7), oracle.details(prm3, 1e2, 1e7), 5e-6)
    # reverse shock of a thick shell; the spreading jet reports its evolved polar angle per cell
    kwr, t, _ = configs.RS_CASES["rs_thick_offaxis"]
    mr = va.Model(va.TophatJet(0.1, 1e53, 100.0, duration=1000.0), va.ISM(1.0), va.Observer(3e28, 0.5, 0.15),
                  va.Radiation(0.1, 1e-3, 2.3), rvs_rad=va.Radiation(0.1, 0.01, 2.5))
    pr = _abi.make_params(**kwr)
    _check_radiation_details(mr.details(t.min(), t.max(), rvs=True), oracle.details(pr, t.min(), t.max(), rvs=True), 5e-6)
    _check_radiation_details(mr.details(t.min(), t.max()), oracle.details(pr, t.min(), t.max()), 5e-6)
    ms = va.Model(va.GaussianJet(0.1, 1e52, 300.0, spreading=True), va.ISM(1.0), va.Observer(1e28, 1.0, 0.15),
                  va.Radiation(0.1, 0.01, 2.3))
    ds = ms.details(1e2, 1e8)
    assert np.all(np.diff(ds["theta_cell"], axis=1) >= 0) and ds["theta_cell"][:, -1].max() > ds["theta"].max()
    os_ = oracle.details(_abi.ModelParams.from_buffer_copy(bytes(ms.params)), 1e2, 1e8)
    np.testing.assert_allclose(ds["t_obs"], np.exp2(os_["lg2_t"]) / _U_SEC, rtol=5e-6)
    np.testing.assert_allclose(ds["Doppler"], np.exp2(os_["lg2_doppler"]), rtol=5e-6)


@pytest.mark.parametrize("name", list(configs.NONAXI_CASES))
def test_non_axisymmetric_models_match_oracle(eng, oracle, name):
    """Model(axisymmetric=False) on the named jets: full-circle phi grid, every phi node observed."""
    prm = _abi.make_params(**configs.NONAXI_CASES[name])
    t, nu = configs.SPREAD_T, configs.SPREAD_NU
    got, want = gpu_components4(eng, prm, t, nu), oracle.flux_components4(prm, t, nu)
    for c in range(4):
        if want[c].max() > 0:
            assert_close(got[c][0], want[c], rtol=5e-6)
    if name == "tophat_offaxis_3d":
        m = va.Model(va.TophatJet(0.1, 1e52, 300.0), va.ISM(1.0), va.Observer(1e28, 1.0, 0.3), va.Radiation(0.1, 0.01, 2.3),
                     resolutions=(prm.phi_resol, prm.theta_resol, prm.t_resol), axisymmetric=False)
        assert m.params.flags == prm.flags == 128
        d, o = m.details(t.min(), t.max()), oracle.details(prm, t.min(), t.max())
        assert d["shape"] == {k: o["shape"][k] for k in d["shape"]} and not d["shape"]["phi_mirrored"]
        np.testing.assert_allclose(d["phi"], o["phi"], rtol=2e-6)
        assert_close(m.flux_density_grid(t, nu).total, want[0], rtol=5e-6)
        # a spreading jet takes (phi, theta) pair rows, with or without a reverse shock (test_non_axisymmetric_spreading_jets_match_the_reference)
        m3 = va.Model(va.GaussianJet(0.1, 1e52, 300, spreading=True), va.ISM(1.0), va.Observer(1e28, 1.0, 0.2), va.Radiation(0.1, 0.01, 2.3),
                      axisymmetric=False)
        assert np.all(np.isfinite(m3.flux_density_grid(t, nu).total))


@pytest.mark.parametrize("kw", [dict(configs.C4_TRUTH, jet="GaussianJet"),
                                dict(jet="TophatJet", theta_obs=0.1, ssc=True),
                                dict(jet="TophatJet", theta_obs=0.1, duration=100.0, rvs=dict(eps_e=0.1, eps_B=0.01, p=2.3)),
                                dict(jet="GaussianJet", theta_obs=0.2, spreading=True)],
                         ids=["c4", "ssc", "rs", "spread"])
def test_shared_node_series_path_is_bitwise_the_per_point_path(eng, oracle, kw):
    """A fit's few bands (host-pointer series call, n <= 64): the spectrum is evaluated once per (band, lattice node) and
    shared by the points (observer.h:447-538); the device-pointer entry does not know the frequencies and evaluates per
    point in vag_flux_series_kernel.  The host-pointer call goes through vag_flux_fit_rows_kernel (a row per lane): same
    evaluators and interpolation arithmetic, another summation order -- the two agree to rounding."""
    import torch
    lib, h = eng
    t, nu = configs.c4_mock_data()
    prm = _abi.make_params(**kw)
    shared = gpu_series(eng, prm, t, nu)[0]
    dev = torch.device("cuda", 0)
    d_p = torch.frombuffer(bytearray(bytes(prm)), dtype=torch.uint8).to(dev)
    d_t, d_nu = torch.from_numpy(np.ascontiguousarray(t)).to(dev), torch.from_numpy(np.ascontiguousarray(nu)).to(dev)
    d_out = torch.zeros((1, t.size), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    _lib.check(lib.vag_flux_density_batch_dev(h, d_p.data_ptr(), 1, d_t.data_ptr(), d_nu.data_ptr(), t.size, d_out.data_ptr()))
    _lib.check(lib.vag_ctx_synchronize(h))
    per_point = d_out.cpu().numpy()[0]
    np.testing.assert_allclose(shared, per_point, rtol=1e-13)
    assert shared.max() > 0
    assert_close(shared, oracle.flux_density(prm, t, nu), rtol=5e-6)
    # all-distinct frequencies: nothing to share, the call silently takes the per-point path
    nu2 = nu * (1 + 1e-3 * np.arange(nu.size))
    assert_close(gpu_series(eng, prm, t, nu2)[0], oracle.flux_density(prm, t, nu2), rtol=5e-6)


@pytest.mark.parametrize("case", ["c5_like", "rs_ssc", "spread_ssc"])
def test_fused_sync_and_ssc_pass_is_bitwise_the_two_pass_form(eng, case):
    """Radiation(ssc=True): the synchrotron and SSC components normally come out of ONE flux pass (shared EAT logs, bracket
    search and barriers); VAG_NO_FUSED=1 restores the two-pass form.  Same per-component arithmetic: bitwise equal."""
    kw = {"c5_like": dict(jet="TwoComponentJet", theta_c=0.06, theta_w=0.3, E_iso_w=1e50, Gamma0_w=50.0, theta_obs=0.15, ssc=True),
          "rs_ssc": dict(configs.C3),
          "spread_ssc": dict(jet="GaussianJet", theta_obs=0.2, spreading=True, ssc=True, kn=True)}[case]
    prm = _abi.make_params(**kw)
    t, nu = configs.C3_T[::4], configs.C3_NU
    fused = gpu_components4(eng, prm, t, nu)
    _lib.hooks["VAG_NO_FUSED"] = "1"
    try:
        two_pass = gpu_components4(eng, prm, t, nu)
    finally:
        del _lib.hooks["VAG_NO_FUSED"]
    for a, b in zip(fused, two_pass):
        assert np.array_equal(a, b, equal_nan=True)
    assert fused[1].max() > 0
    _lib.hooks["VAG_GRID_ROWWISE"] = "1"  # the wavefront-per-row kernel on the same small grid (measured slower: not the default)
    try:
        rowwise = gpu_components4(eng, prm, t, nu)  # same algorithm, another summation order
    finally:
        del _lib.hooks["VAG_GRID_ROWWISE"]
    for a, b in zip(rowwise, two_pass):
        m = b > 1e-12 * b.max() if b.max() > 0 else np.zeros_like(b, dtype=bool)
        assert np.all(np.abs(a - b)[m] <= 1e-11 * b[m]) and np.all(a[~m] <= 1e-11 * max(b.max(), 1e-300))
    lib, h = eng
    band_f = np.empty((1, t.size))
    arr = (_lib.ModelParams * 1)(_lib.ModelParams.from_buffer_copy(bytes(prm)))
    _lib.check(lib.vag_flux_batch(h, arr, 1, t.ctypes.data_as(dp), t.size, 1e17, 1e19, 7, band_f.ctypes.data_as(dp)))
    _lib.hooks["VAG_NO_FUSED"] = "1"
    try:
        band_t = np.empty((1, t.size))
        _lib.check(lib.vag_flux_batch(h, arr, 1, t.ctypes.data_as(dp), t.size, 1e17, 1e19, 7, band_t.ctypes.data_as(dp)))
    finally:
        del _lib.hooks["VAG_NO_FUSED"]
    assert np.array_equal(band_f, band_t)


def test_spreading_jet_whose_arrival_times_do_not_ascend_keeps_every_ssc_table(eng, oracle):
    """A spreading jet's polar angle evolves along the lattice: a row that swings towards the line of sight reaches the observer EARLIER
    from a later node, and the flux kernels place the observation window by counting nodes like the reference (observed_window,
    observer.h:324-338) -- so which cells a request queries is not the range test of the non-spreading case.  Draw 13 of the random
    sweep of spreading SSC jets (profiles/debug/prior_sweep_ssc.py, SWEEP_MODE=spread: a Gaussian jet in a dense wind seen from
    outside the core) queried cells that test had skipped and failed loudly (status bit 4) in round 4; spreading jets keep every
    table now.  Against the checker, both components."""
    kw = dict(jet="GaussianJet", E_iso=5.439833428348364e+50, Gamma0=387.31924795103055, theta_c=0.16449474595309396,
              theta_obs=0.4047201203232014, p=2.5802629713712104, eps_e=0.06605189948979821, eps_B=0.08481166776782337, ssc=True, kn=True,
              medium="Wind", A_star=2.3251434233117934, spreading=True)  # (n_ism stays at make_params' 1.0: the draw's wind has that floor)
    t, nu = np.logspace(1.5, 7.5, 30), np.array([1e9, 4.84e14, 1e18, 2.4e22, 1e26])
    prm = _abi.make_params(**kw)
    got = gpu_components4(eng, [prm], t, nu)
    want = oracle.flux_components(prm, t, nu)
    for g, w in zip((got[0][0], got[1][0]), want):
        assert np.all(np.isfinite(g)) and w.max() > 0
        m = w > 1e-3 * w.max()
        assert np.max(np.abs(g - w)[m] / w[m]) <= 2e-6  # (measured 1e-10)


def test_rows_whose_state_goes_non_finite_keep_their_ssc_tables(eng, oracle):
    """A Gaussian jet with a magnetar in a dense wind, seen almost on axis: the blast wave of the outermost theta row leaves the solver's
    range and its state -- observer times included -- is NaN from node 13 on (in the reference as well: those intervals have no finite
    slope and add nothing).  The flux kernels still visit the cells, because they count the nodes before the window like the reference;
    the test that decides which cells get an SSC table compared NaN times and skipped them (draw 19 of the Thomson half of
    `SWEEP_MODE=magnetar profiles/debug/prior_sweep_ssc.py 30`: status bit 4, loudly, in round 4).  It is written as "not excluded"
    now.  Against the checker, both components."""
    kw = dict(jet="GaussianJet", E_iso=4.460426649397416e+51, Gamma0=427.30930019677726, theta_c=0.07602877048106377,
              theta_obs=0.015165954752948074, p=2.1137269063496933, eps_e=0.06343644661487786, eps_B=0.0024324771035933146, ssc=True,
              kn=False, medium="Wind", A_star=1.865321778529919, magnetar=(4.722996518123433e+46, 669.4503306793644, 2.0487730053290845))
    t, nu = np.logspace(1.5, 7.5, 30), np.array([1e9, 4.84e14, 1e18, 2.4e22, 1e26])
    prm = _abi.make_params(**kw)
    got = gpu_components4(eng, [prm], t, nu)
    want = oracle.flux_components(prm, t, nu)
    for g, w in zip((got[0][0], got[1][0]), want):
        assert np.all(np.isfinite(g)) and w.max() > 0
        m = w > 1e-3 * w.max()
        assert np.max(np.abs(g - w)[m] / w[m]) <= 2e-6


def test_fast_and_general_ode_kernels_agree_on_random_walkers(eng):
    """The common case (ISM / analytic wind, no spreading, no injection, no reverse shock) is integrated by vag_dynamics_fast_kernel (flat
    attempt loop + saver wavefront, reciprocal / rsqrt + one Newton step in the right-hand side); VAG_DYN_GENERAL=1 sends the same rows
    through the general kernel (library divisions, rows per wavefront chosen from the batch size since round 4).  Same controller, same
    step sequences: the fluxes of 256 random C4-box walkers, of 16 and of one agree to 1e-10 (measured 5e-13)."""
    rng = np.random.default_rng(5)
    t, nu = np.logspace(4.5, 8, 40), np.array([3e9, 5.06e14, 2.41e17])
    for n in (256, 16, 1):
        prms = []
        for _ in range(n):
            kw = dict(configs.C4_TRUTH, jet="GaussianJet")
            kw.update(E_iso=10 ** rng.uniform(50, 54), Gamma0=10 ** rng.uniform(1.5, 3), theta_c=rng.uniform(0.02, 0.3), theta_obs=rng.uniform(0, 0.8),
                      n_ism=10 ** rng.uniform(-4, 1), p=rng.uniform(2.05, 2.8), eps_e=10 ** rng.uniform(-3, -0.5), eps_B=10 ** rng.uniform(-5, -1))
            prms.append(_abi.make_params(**kw))
        fast = gpu_grid(eng, prms, t, nu)
        _lib.hooks["VAG_DYN_GENERAL"] = "1"
        try:
            general = gpu_grid(eng, prms, t, nu)
        finally:
            _lib.hooks.pop("VAG_DYN_GENERAL")
        assert np.all(np.isfinite(fast)) and np.all(np.isfinite(general))
        sel = fast > 1e-3 * fast.max(axis=(1, 2), keepdims=True)
        assert np.max(np.abs(fast - general)[sel] / fast[sel]) <= 1e-10, n


def test_lane_refill_ode_kernel_gives_the_bits_of_the_plain_kernel(eng):
    """Round 6 ("persistent-lane work queue"): batches that fill the GPU's integrator slots twice over are integrated by
    vag_dynamics_refill_kernel -- persistent wavefronts whose finished lanes take the next row from a device counter and which evaluate the
    dense output inline (two integrators per SIMD instead of an integrator and its saver), every row's start prepared beforehand by
    vag_dyn_prep_kernel -- instead of one wavefront per 64 rows that runs until its slowest row is done.  Rows are
    independent and the arithmetic is the plain kernel's, expression for expression: the fluxes of 300 random walkers (ISM and wind
    media mixed in one batch, narrow jets whose wings stop, one invalid model) are the SAME BITS whichever kernel runs and however
    many finished lanes a wavefront collects before it refills (1, 8, 64); the solver's own tallies agree too (right-hand sides equal,
    row failures equal), and the refill kernel's lanes are busier."""
    rng = np.random.default_rng(66)
    t, nu = np.logspace(4.0, 8, 40), np.array([3e9, 5.06e14, 2.41e17])
    prms = []
    for i in range(300):
        kw = dict(configs.C4_TRUTH, jet="GaussianJet" if i % 3 else "PowerLawJet")
        kw.update(E_iso=10 ** rng.uniform(50, 54), Gamma0=10 ** rng.uniform(0.5 if i % 3 == 0 else 1.5, 3), theta_c=rng.uniform(0.02, 0.3),
                  theta_obs=rng.uniform(0, 0.8), p=rng.uniform(2.05, 2.8), eps_e=10 ** rng.uniform(-3, -0.5), eps_B=10 ** rng.uniform(-5, -1))
        if i % 4 == 1:
            kw.update(medium="Wind", A_star=10 ** rng.uniform(-2, 0.5), n_ism=0.0)
        else:
            kw.update(n_ism=10 ** rng.uniform(-4, 1))
        if i % 7 == 3:
            kw["radiative_fireball"] = False
        prms.append(_abi.make_params(**kw))
    prms[17].eps_e = 2.0  # rejected by validation: its rows do not exist
    lib, h = eng
    import torch
    dev = torch.device("cuda", 0)
    arr = (_lib.ModelParams * len(prms))(*[_lib.ModelParams.from_buffer_copy(bytes(p)) for p in prms])
    d_p = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    d_t, d_nu = torch.from_numpy(t).to(dev), torch.from_numpy(nu).to(dev)

    def gpu_grid_dev():  # (the device-pointer form reports an invalid model as a NaN row instead of failing the call)
        d_o = torch.full((len(prms), nu.size, t.size), -1.0, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        _lib.check(lib.vag_flux_density_grid_batch_dev(h, d_p.data_ptr(), len(prms), d_t.data_ptr(), t.size, d_nu.data_ptr(), nu.size, d_o.data_ptr()))
        _lib.check(lib.vag_ctx_synchronize(h))
        return d_o.cpu().numpy()

    def run(refill, refill_min=None):
        _lib.hooks["VAG_DYN_REFILL"] = refill
        if refill_min is not None:
            _lib.hooks["VAG_DYN_REFILL_MIN"] = str(refill_min)
            _lib.hooks["VAG_DYN_REFILL_WGS"] = "24"  # 24 persistent wavefronts for ~16 k rows: every lane refills ~10 times
        try:
            flux = gpu_grid_dev()
            _lib.check(lib.vag_ctx_count_work(h, 1))
            flux_tallied = gpu_grid_dev()
            plan = _lib.Plan()
            _lib.check(lib.vag_last_plan(h, C.byref(plan)))
        finally:
            _lib.check(lib.vag_ctx_count_work(h, 0))
            _lib.hooks.pop("VAG_DYN_REFILL")
            _lib.hooks.pop("VAG_DYN_REFILL_MIN", None)
            _lib.hooks.pop("VAG_DYN_REFILL_WGS", None)
        assert np.array_equal(flux, flux_tallied, equal_nan=True)  # the tallying instantiation changes no bit either
        return flux, plan

    plain, p0 = run("0")
    ok = np.isfinite(plain).all(axis=(1, 2))
    assert ok.sum() == 299 and not ok[17] and plain[ok].max() > 0
    assert p0.ode_rhs > 0 and 0 < p0.ode_lane_attempts < p0.ode_lane_slots
    busy = {}
    for refill_min in (1, 8, 64):
        got, p1 = run("1", refill_min)
        assert np.array_equal(got, plain, equal_nan=True), refill_min
        assert (p1.n_rows, p1.ode_rhs, p1.n_rows_failed, p1.n_rows_gave_up) == (p0.n_rows, p0.ode_rhs, p0.n_rows_failed, p0.n_rows_gave_up)
        assert p1.ode_lane_attempts == p0.ode_lane_attempts  # the same attempts of the same rows ...
        busy[refill_min] = p1.ode_lane_attempts / p1.ode_lane_slots  # ... packed into fewer wavefront trips
    assert busy[1] >= busy[8] >= busy[64] - 1e-12 and busy[1] > p0.ode_lane_attempts / p0.ode_lane_slots


def test_the_grid_kernels_two_layouts_give_the_same_bits(eng):
    """vag_grid_kernel has a small LDS layout (256 theta / 208 phi nodes, eight models per CU) and a large one (1280 / 2560, one per CU); a
    batch in which some model outgrows the small one is laid out again with the large one, and every array downstream is strided by the
    layout in use (VagGridMeta::th_stride / ph_stride).  The layout must not enter the numbers: 48 mixed models (all jet profiles, reverse
    shocks, SSC, spreading, axisymmetric=False) on a grid and on a series request, with the layout their sizes call for and with the large
    one forced (VAG_GRID_FORCE_LARGE) -- same bits; and back again afterwards."""
    import sweeps
    lib, h = eng
    prms, tags = sweeps.ssc_window_models(48, seed=2718)
    n, nu = len(prms), sweeps.WINDOW_NU
    arr = (_lib.ModelParams * n)(*[_lib.ModelParams.from_buffer_copy(bytes(p)) for p in prms])
    t = np.logspace(2.5, 7, 14)

    def run(series):
        if series:
            tt, nn = np.repeat(t, nu.size), np.tile(nu, t.size)
            comps = [np.empty((n, tt.size)) for _ in range(4)]
            out4 = (dp * 4)(*[a.ctypes.data_as(dp) for a in comps])
            _lib.check(lib.vag_flux_density_components4_batch(h, arr, n, tt.ctypes.data_as(dp), nn.ctypes.data_as(dp), tt.size, out4))
        else:
            comps = [np.empty((n, nu.size, t.size)) for _ in range(4)]
            out4 = (dp * 4)(*[a.ctypes.data_as(dp) for a in comps])
            _lib.check(lib.vag_flux_density_grid_components4_batch(h, arr, n, t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp), nu.size, out4))
        return comps

    for series in (False, True):
        small = run(series)
        _lib.hooks["VAG_GRID_FORCE_LARGE"] = "1"
        try:
            large = run(series)
        finally:
            _lib.hooks.pop("VAG_GRID_FORCE_LARGE")
        for c, (a, b) in enumerate(zip(small, large)):
            bad = [i for i in range(n) if not np.array_equal(a[i], b[i], equal_nan=True)]
            assert not bad, (series, c, [tags[i] for i in bad[:4]])
        assert max(float(np.nanmax(x)) for x in small) > 0
    for _ in range(9):  # eight fitting batches in a row take the context back to the small layout
        again = run(False)
    for a, b in zip(again, run(False)):
        assert np.array_equal(a, b, equal_nan=True)


def test_lazily_built_ssc_tables_are_the_bits_of_every_table_on_random_narrow_windows(eng):
    """Property test of the table-per-queried-cell logic (vag_ic_band_kernel's range test): 48 SSC models of every kind -- six jet
    profiles, ISM / wind, KN / Thomson, reverse shocks, magnetars, spreading, axisymmetric=False, one ragged mixed-flag batch -- times 24
    random NARROW request windows (1 ... 12 times inside 0.05 ... 2 decades anywhere between 30 s and 3e8 s: the requests that leave
    80-90 % of the cells without a table), grid and series requests alternating.  Each is evaluated with the lazy tables and with every
    table (VAG_IC_ALL_CELLS=1): the four components must be the same bits, and no call may meet a cell without a table (that would
    raise).  The two holes round 4's oracle sweeps found in that logic would both have failed here."""
    import sweeps
    lib, h = eng
    prms, tags = sweeps.ssc_window_models(48)
    n, nu = len(prms), sweeps.WINDOW_NU
    arr = (_lib.ModelParams * n)(*[_lib.ModelParams.from_buffer_copy(bytes(p)) for p in prms])

    def run(t, series):
        if series:
            tt, nn = np.repeat(t, nu.size), np.tile(nu, t.size)
            comps = [np.empty((n, tt.size)) for _ in range(4)]
            out4 = (dp * 4)(*[a.ctypes.data_as(dp) for a in comps])
            _lib.check(lib.vag_flux_density_components4_batch(h, arr, n, tt.ctypes.data_as(dp), nn.ctypes.data_as(dp), tt.size, out4))
        else:
            comps = [np.empty((n, nu.size, t.size)) for _ in range(4)]
            out4 = (dp * 4)(*[a.ctypes.data_as(dp) for a in comps])
            _lib.check(lib.vag_flux_density_grid_components4_batch(h, arr, n, t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp), nu.size, out4))
        return comps

    lazy_bytes = all_bytes = 0
    for w, t in enumerate(sweeps.narrow_windows(24)):
        series = w % 3 == 2
        got = run(t, series)
        pl = _lib.Plan()
        lib.vag_last_plan(h, C.byref(pl))
        lazy_bytes += pl.ic_pool_bytes
        _lib.hooks["VAG_IC_ALL_CELLS"] = "1"
        try:
            want = run(t, series)
        finally:
            _lib.hooks.pop("VAG_IC_ALL_CELLS")
        lib.vag_last_plan(h, C.byref(pl))
        all_bytes += pl.ic_pool_bytes
        for c, (g, x) in enumerate(zip(got, want)):
            bad = [i for i in range(n) if not np.array_equal(g[i], x[i], equal_nan=True)]
            assert not bad, (w, c, [tags[i] for i in bad[:4]])
    assert lazy_bytes < 0.6 * all_bytes  # (the windows do leave most cells without a table: the property is exercised)


@pytest.mark.parametrize("case", ["grid", "series", "fused", "rows_batch"])
def test_ssc_tables_only_for_the_cells_a_request_queries(eng, case):
    """The reference builds a cell's SSC spectrum on its first query (ICPhoton::compute_log2_I_nu, inverse-compton.h:614-620); the
    engine gives a table to the cells some (theta, phi) row's observation window touches (vag_ic_band_kernel: ~80 % of the cells of
    the configs[2] shape) and leaves the others without.  (1) The fluxes are the bits of a pass that builds every table
    (VAG_IC_ALL_CELLS=1), on every kernel family.  (2) A query of a skipped cell is never a silent zero: VAG_DEBUG_IC_NEED_SHRINK cuts
    the window short so that the flux pass meets such cells; the engine then builds every table and repeats the pass (counted in
    vag_plan.n_ssc_all_cell_fallbacks), and raises VAG_E_INTERNAL only if that cannot help."""
    lib, h = eng
    kw = dict(jet="PowerLawJet", medium="Wind", A_star=0.1, n_ism=0.0, theta_obs=0.3, duration=50.0, ssc=True, kn=True,
              rvs=dict(eps_e=0.1, eps_B=0.01, p=2.3, ssc=True, kn=True))
    t, nu = np.logspace(3, 6, 24), np.array([1e9, 1e15, 1e18, 1e24])  # a window well inside the lattice: cells at both ends unqueried
    prms = [_abi.make_params(**kw)]
    if case == "rows_batch":  # enough rows for the row-per-lane grid kernels (>= 4096 blocks of 64 rows)
        prms = [_abi.make_params(**dict(kw, E_iso=1e52 * (1 + 0.01 * i), resolutions=(0.3, 1.0, 10.0))) for i in range(48)]

    def run():
        if case == "series":
            tt, nn = np.repeat(t, nu.size), np.tile(nu, t.size)
            comps = [np.empty((len(prms), tt.size)) for _ in range(4)]
            out4 = (dp * 4)(*[a.ctypes.data_as(dp) for a in comps])
            arr = (_lib.ModelParams * len(prms))(*[_lib.ModelParams.from_buffer_copy(bytes(p)) for p in prms])
            _lib.check(lib.vag_flux_density_components4_batch(h, arr, len(prms), tt.ctypes.data_as(dp), nn.ctypes.data_as(dp), tt.size, out4))
            return comps
        if case == "grid":
            _lib.hooks["VAG_NO_FUSED"] = "1"
        try:
            return gpu_components4(eng, prms, t, nu)
        finally:
            _lib.hooks.pop("VAG_NO_FUSED", None)

    got = run()
    _lib.hooks["VAG_IC_ALL_CELLS"] = "1"
    try:
        want = run()
    finally:
        _lib.hooks.pop("VAG_IC_ALL_CELLS")
    assert want[1].max() > 0 and want[3].max() > 0
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    # (2) a hole in the selection: the pass is repeated with every cell's table -- the reference's answer, bit for bit -- and counted;
    #     with the fallback switched off the fault is loud (what a second miss, on the all-cells pass, would raise)
    _lib.hooks["VAG_DEBUG_IC_NEED_SHRINK"] = "1e-2"
    try:
        got_fb = run()
        plan = _lib.Plan()
        lib.vag_last_plan(h, C.byref(plan))
        assert plan.n_ssc_all_cell_fallbacks >= 1
        for g, w in zip(got_fb, want):
            assert np.array_equal(g, w)
        _lib.hooks["VAG_DEBUG_IC_NO_FALLBACK"] = "1"
        with pytest.raises(RuntimeError, match="queried an SSC cell that was given no table"):
            run()
    finally:
        _lib.hooks.pop("VAG_DEBUG_IC_NEED_SHRINK")
        _lib.hooks.pop("VAG_DEBUG_IC_NO_FALLBACK", None)
    for g, w in zip(run(), want):  # and the context is in order afterwards
        assert np.array_equal(g, w)
    plan = _lib.Plan()
    lib.vag_last_plan(h, C.byref(plan))
    assert plan.n_ssc_all_cell_fallbacks == 0


@pytest.mark.parametrize("case", ["kn_rvs_wind", "thomson_ism", "kn_ism_offaxis"])
def test_ssc_cells_beyond_the_on_chip_lattice_limits_take_the_general_kernel(eng, oracle, case):
    """The reference sizes an SSC cell's lattices per cell (ICPhoton::initialize_grids, inverse-compton.h:340-369); the wavefront-per-cell
    kernel holds at most 128 seed frequencies, 64 electron energies and 192 output nodes on chip.  A cell beyond that takes
    vag_ic_photon_slow_kernel -- ICPhoton::generate_spectrum in the reference's own CDF form, arrays in HBM -- instead of
    VAG_E_CAPACITY.  No cell of the reference's configurations is that long, so VAG_DEBUG_IC_FAST_NU_MAX lowers the fast kernel's limit:
    0 sends EVERY cell through the general kernel (which makes this also a check of the fast kernel's diagonal-histogram form against a
    second, reference-shaped restatement on the device), a limit inside the batch's range mixes the two.  Same components to 1e-11,
    the count is reported in vag_plan.n_ssc_slow_cells, and the all-slow run is compared with the CPU checker as well."""
    lib, h = eng
    kw = {"kn_rvs_wind": dict(jet="PowerLawJet", medium="Wind", A_star=0.1, n_ism=0.0, theta_obs=0.3, duration=50.0, ssc=True, kn=True,
                              rvs=dict(eps_e=0.1, eps_B=0.01, p=2.3, ssc=True, kn=True)),
          "thomson_ism": dict(jet="TophatJet", theta_obs=0.0, ssc=True, kn=False),
          "kn_ism_offaxis": dict(jet="GaussianJet", theta_obs=0.25, ssc=True, kn=True, eps_B=1e-4)}[case]
    prm = _abi.make_params(**kw)
    t, nu = np.logspace(2.5, 7, 20), np.array([1e9, 1e14, 1e17, 1e20, 1e23, 1e26])

    def run(limit):
        if limit is not None:
            _lib.hooks["VAG_DEBUG_IC_FAST_NU_MAX"] = str(limit)
        try:
            comps = gpu_components4(eng, prm, t, nu)
        finally:
            _lib.hooks.pop("VAG_DEBUG_IC_FAST_NU_MAX", None)
        pl = _lib.Plan()
        lib.vag_last_plan(h, C.byref(pl))
        return comps, pl.n_ssc_slow_cells
    want, n0 = run(None)
    assert n0 == 0 and want[1].max() > 0
    got, n_all = run(0)
    assert n_all > 20
    for g, w in zip(got, want):
        np.testing.assert_allclose(g, w, rtol=1e-11, atol=0)
    # a limit that splits the cells: walk down from the fast kernel's own until some, not all, cells are over it
    n_mix = 0
    for limit in range(120, 8, -8):
        mixed, n_mix = run(limit)
        if 0 < n_mix < n_all:
            break
    assert 0 < n_mix < n_all
    for g, w in zip(mixed, want):
        np.testing.assert_allclose(g, w, rtol=1e-11, atol=0)
    # and against the CPU checker, like any other SSC result
    ref = oracle.flux_components4(prm, t, nu)  # [4 comps][nu][t]
    for g, r in zip(got, ref):
        assert _within_golden_contract(g[0], np.asarray(r))


def test_ssc_loglike_with_cells_on_the_general_kernel(eng, oracle):
    """A likelihood call never waits for the table build's counters: the general kernel is launched over a list it reads from HBM and
    its cells draw on a fixed reserve of the pool (64 MB).  With some cells over the (lowered) limit the walkers score what they score
    otherwise; with EVERY cell over it the reserve runs out and the walkers concerned score -inf, counted as SSC failures -- loud, not
    wrong."""
    rng = np.random.default_rng(17)
    t = np.sort(10 ** rng.uniform(3.0, 6.0, 40))
    nu = 10 ** rng.choice([14.7, 17.5, 23.0], size=t.size)
    f = fitting.Fitter(z=1.0, lumi_dist=1e28, jet="gaussian", medium="ism", fwd_ssc=True, kn=True)
    truth = _abi.make_params(jet="GaussianJet", theta_obs=0.2, ssc=True, kn=True)
    f_obs = oracle.flux_density(truth, t, nu) * (1 + 0.05 * rng.standard_normal(t.size))
    f.add_flux_density(nu, t, f_obs, 0.05 * f_obs)
    P, S = fitting.ParamDef, fitting.Scale
    defs = [P("E_iso", 1e51, 1e53, S.log), P("Gamma0", 100, 500, S.log), P("theta_c", 0.05, 0.2, S.linear),
            P("theta_v", 0.0, 0.4, S.linear), P("n_ism", 0.1, 10, S.log), P("p", 2.1, 2.6, S.linear),
            P("eps_e", 0.03, 0.3, S.log), P("eps_B", 1e-3, 1e-1, S.log), P("xi_e", 1.0, 1.0, S.fixed, 1.0)]
    lo = np.array([51, 2.0, 0.05, 0.0, -1, 2.1, np.log10(0.03), -3])
    hi = np.array([53, np.log10(500), 0.2, 0.4, 1, 2.6, np.log10(0.3), -1])
    theta = lo + (hi - lo) * rng.random((8, 8))
    want = f.loglike_batch(theta, defs)
    assert np.all(np.isfinite(want)) and f.last_plan.n_ssc_slow_cells == 0

    def run(limit):
        _lib.hooks["VAG_DEBUG_IC_FAST_NU_MAX"] = str(limit)
        try:
            ll = f.loglike_batch(theta, defs)
        finally:
            _lib.hooks.pop("VAG_DEBUG_IC_FAST_NU_MAX")
        return ll, f.last_plan
    n_slow = 0
    for limit in range(120, 8, -4):  # the highest limit that leaves some cells over it: a handful of them
        ll, plan = run(limit)
        n_slow = plan.n_ssc_slow_cells
        if n_slow > 0:
            break
    assert 0 < n_slow < 3000 and plan.n_walkers_ssc_failed == 0
    np.testing.assert_allclose(ll, want, rtol=1e-10, atol=0)
    ll, plan = run(0)  # every cell: far beyond the reserve
    assert plan.n_walkers_ssc_failed >= 1 and plan.n_walkers_rejected == plan.n_walkers_ssc_failed
    assert np.sum(np.isneginf(ll)) == plan.n_walkers_ssc_failed
    ok = np.isfinite(ll)
    np.testing.assert_allclose(ll[ok], want[ok], rtol=1e-10, atol=0)
    assert np.array_equal(f.loglike_batch(theta, defs), want)  # and the context is in order afterwards


@pytest.mark.parametrize("case", ["grid", "series", "fused"])
def test_ssc_band_breach_rebuilds_the_tables_unclamped(eng, case):
    """ICPhoton::compute_log2_I_nu drops a cell's band clamp and rebuilds its spectrum when a query falls outside the clamped
    band but inside the theoretical range (inverse-compton.h:626-635).  The engine derives the clamp from the very Doppler
    extremes the flux pass uses, so the case does not arise by itself; VAG_DEBUG_IC_NARROW shrinks the clamp's upper edge so that it
    does.  The pass must then rebuild those models' tables over the full range, repeat, and return what the clamped tables
    return (the lattice is phase-locked: shared nodes carry the same values) -- not raise."""
    lib, h = eng
    prm = _abi.make_params(jet="GaussianJet", theta_obs=0.2, ssc=True, kn=(case != "fused"))
    t, nu = np.logspace(3, 7, 24), np.array([1e9, 1e15, 1e18, 1e24])

    def run():
        if case == "series":
            tt, nn = np.repeat(t, nu.size), np.tile(nu, t.size)
            comps = [np.empty((1, tt.size)) for _ in range(4)]
            out4 = (dp * 4)(*[a.ctypes.data_as(dp) for a in comps])
            arr = (_lib.ModelParams * 1)(_lib.ModelParams.from_buffer_copy(bytes(prm)))
            _lib.check(lib.vag_flux_density_components4_batch(h, arr, 1, tt.ctypes.data_as(dp), nn.ctypes.data_as(dp), tt.size, out4))
        else:
            if case == "grid":
                _lib.hooks["VAG_NO_FUSED"] = "1"
            try:
                comps = gpu_components4(eng, prm, t, nu)
            finally:
                _lib.hooks.pop("VAG_NO_FUSED", None)
        pl = _lib.Plan()
        lib.vag_last_plan(h, C.byref(pl))
        return comps, pl.n_models_ssc_rebuilt
    want, rebuilt0 = run()
    assert rebuilt0 == 0 and want[1].max() > 0
    _lib.hooks["VAG_DEBUG_IC_NARROW"] = "1e-9"
    try:
        got, rebuilt = run()
    finally:
        del _lib.hooks["VAG_DEBUG_IC_NARROW"]
    assert rebuilt >= 1
    for a, b in zip(got, want):
        np.testing.assert_allclose(a, b, rtol=1e-12, atol=0)


@pytest.mark.parametrize("case", ["tophat", "gaussian_offaxis", "ssc_kn", "rs", "spreading"])
def test_lattices_longer_than_the_staged_row_are_taken_in_pieces(eng, case):
    """The flux kernels stage at most 512 lattice nodes of a row at a time and take a longer lattice in pieces that overlap by
    one node (each requested time falls into exactly one piece).  VAG_FLUX_K_CAP makes ordinary models take that path with
    pieces of 12 nodes: grid, series and band results must come back as from the one-piece form, per component, to summation
    rounding."""
    lib, h = eng
    kw = {"tophat": configs.C1A, "gaussian_offaxis": configs.C2, "ssc_kn": dict(configs.C1B, ssc=True, kn=True),
          "rs": dict(configs.C3, ssc=False, kn=False, rvs=dict(configs.C3["rvs"], ssc=False, kn=False)),
          "spreading": dict(jet="GaussianJet", theta_obs=0.25, spreading=True)}[case]
    prm = _abi.make_params(**kw)
    t, nu = np.logspace(2.5, 7.5, 37), np.array([1e9, 4.84e14, 1e18])
    tt, nn = np.repeat(t, nu.size)[::2], np.tile(nu, t.size)[::2]
    arr = (_lib.ModelParams * 1)(_lib.ModelParams.from_buffer_copy(bytes(prm)))

    def run():
        grid = gpu_components4(eng, prm, t, nu)
        series = gpu_series(eng, prm, tt, nn)
        band = np.empty((1, t.size))
        _lib.check(lib.vag_flux_batch(h, arr, 1, t.ctypes.data_as(dp), t.size, 1e17, 1e19, 7, band.ctypes.data_as(dp)))
        pl = _lib.Plan()
        lib.vag_last_plan(h, C.byref(pl))
        return grid, series, band, pl.n_cells // max(pl.n_rows, 1)
    g1, s1, b1, k_nodes = run()
    assert k_nodes > 40  # several pieces of 12
    _lib.hooks["VAG_FLUX_K_CAP"] = "12"
    try:
        g2, s2, b2, _ = run()
    finally:
        del _lib.hooks["VAG_FLUX_K_CAP"]
    for a, b in zip(g2 + [s2, b2], g1 + [s1, b1]):
        assert np.all(np.isfinite(a))
        np.testing.assert_allclose(a, b, rtol=2e-13, atol=1e-300)
    assert g1[0].max() > 0


def test_a_lattice_of_more_than_512_nodes_matches_the_oracle(eng, oracle):
    """resolutions = (0.3, 1, 70): 70 time nodes per decade give a lattice of more than 512 nodes (VAG_E_CAPACITY before the flux
    kernels took lattices in pieces).  Grid and series against the oracle at the tolerance of the default-resolution cases."""
    lib, h = eng
    prm = _abi.make_params(**dict(configs.C1B, resolutions=(0.3, 1.0, 70.0)))
    t, nu = np.logspace(2, 7, 40), np.array([1e9, 1e14, 1e18])
    got = gpu_grid(eng, prm, t, nu)[0]
    pl = _lib.Plan()
    lib.vag_last_plan(h, C.byref(pl))
    assert pl.n_models_ok == 1 and pl.n_cells // pl.n_rows > 512
    want = oracle.flux_density_grid(prm, t, nu)
    m = want > 1e-9 * want.max()
    assert np.max(np.abs(got - want)[m] / want[m]) < 2e-6
    tt, nn = np.repeat(t, nu.size), np.tile(nu, t.size)
    ser = gpu_series(eng, prm, tt, nn)[0]
    np.testing.assert_allclose(ser.reshape(t.size, nu.size).T[m], got[m], rtol=1e-9)


@pytest.mark.parametrize("res, n_theta, n_phi", [((2.0, 4.0, 5.0), 413, 360), ((4.0, 1.0, 5.0), None, 720)])
def test_angular_grids_beyond_the_small_layout_of_the_grid_kernel(eng, oracle, res, n_theta, n_phi):
    """resolutions a user may pass to the reference (grid-refinement.h:639-706 sizes the grids freely): more than 320 theta nodes,
    more than 640 phi nodes.  The grid kernel's default LDS layout does not hold them (VAG_E_CAPACITY before); the batch is laid out
    again with the large layout.  Grid against the oracle; a default-resolution model in the same batch comes back as in a batch
    of its own (to summation rounding: a grid request groups its rows by the batch total)."""
    lib, h = eng
    big = _abi.make_params(jet="GaussianJet", theta_c=0.1, E_iso=1e52, Gamma0=300.0, medium="ISM", n_ism=1.0, theta_obs=0.3,
                           resolutions=res)
    small = _abi.make_params(**configs.C1B)
    t, nu = np.logspace(3, 7, 12), np.array([1e9, 1e15])
    got = gpu_grid(eng, [big, small], t, nu)
    pl = _lib.Plan()
    lib.vag_last_plan(h, C.byref(pl))
    assert pl.n_models_ok == 2 and pl.n_models_capacity == 0
    d = oracle.details(big, t.min(), t.max())["shape"]
    assert d["n_phi"] == n_phi and (n_theta is None or d["n_theta"] == n_theta)
    assert d["n_theta"] > 320 or d["n_phi"] > 640
    want = oracle.flux_density_grid(big, t, nu)
    m = want > 1e-9 * want.max()
    assert np.max(np.abs(got[0] - want)[m] / want[m]) < 2e-6
    for _ in range(9):  # the context goes back to the small layout after a run of batches that fit it
        alone = gpu_grid(eng, small, t, nu)[0]
    np.testing.assert_allclose(alone, got[1], rtol=1e-12)


@pytest.mark.parametrize("case", list(configs.BIG_GRID_CASES))
def test_grids_beyond_the_lds_layouts_match_the_reference(eng, oracle, case):
    """Round 6: the reference sizes its grids freely (grid-refinement.h:639-706); until round 5 a model beyond 1280 theta / 2560 phi / 8192
    lattice nodes was VAG_E_CAPACITY (-inf in a fit).  The grid kernel now has a third layout -- the same wavefront program with its
    scratch arrays in HBM, up to 16384 / 32768 angular nodes -- and the lattice length is no layout limit at all (no kernel holds a row
    at once).  More than 2000 theta nodes, more than 10 000 lattice times, BOTH AT ONCE (17 M cells: two minutes of CPU for the
    reference, hence tests/golden/reference_big_grids.npz from the reference's own build), more than 2560 phi nodes off axis: grid
    integers equal, fluxes <= 2e-6 from the nearer reference build; the cheap ones also live against the checker.  A
    default-resolution model in the same batch comes back as in a batch of its own."""
    lib, h = eng
    fx = np.load(os.path.join(_abi.ROOT, "tests", "golden", "reference_big_grids.npz"))
    t, nu = fx["t"], fx["nu"]
    assert np.array_equal(t, configs.BIG_GRID_T) and np.array_equal(nu, configs.BIG_GRID_NU)
    big = _abi.make_params(**configs.BIG_GRID_CASES[case])
    small = _abi.make_params(**configs.C1B)
    got = gpu_grid(eng, [big, small], t, nu)
    pl = _lib.Plan()
    lib.vag_last_plan(h, C.byref(pl))
    assert pl.n_models_ok == 2 and pl.n_models_capacity == 0
    n_phi, n_theta, n_t = (int(v) for v in fx[case + "_shape"][:3])
    assert {"theta_2000": n_theta > 2000, "time_10000": n_t > 10000, "theta_2000_time_10000": n_theta > 2000 and n_t > 10000,
            "phi_3000_offaxis": n_phi > 2560}[case], (n_phi, n_theta, n_t)
    assert tuple(_engine_shape(eng, big, t)) == tuple(int(v) for v in fx[case + "_shape"])  # the six grid integers of the reference
    if case != "theta_2000_time_10000":  # (the checker's own grid takes minutes on that case)
        assert oracle.grid_shape(big, float(t.min()), float(t.max())) == tuple(int(v) for v in fx[case + "_shape"])
    fast, strict = fx[case + "_fast"], fx[case + "_strict"]
    err = np.minimum(np.abs(got[0] - fast) / fast, np.abs(got[0] - strict) / strict)
    assert np.all(np.isfinite(got[0])) and err.max() <= 2e-6, err.max()
    if case != "theta_2000_time_10000":
        want = oracle.flux_density_grid(big, t, nu)
        assert np.max(np.abs(got[0] - want) / want) <= 2e-6
    tt, nn = np.repeat(t, nu.size), np.tile(nu, t.size)
    ser = gpu_series(eng, big, tt, nn)[0]  # the series kernels walk the same rows
    np.testing.assert_allclose(ser.reshape(t.size, nu.size).T, got[0], rtol=1e-9)
    for _ in range(9):  # the context goes back to the small layout after a run of batches that fit it
        alone = gpu_grid(eng, small, t, nu)[0]
    np.testing.assert_allclose(alone, got[1], rtol=1e-12)


def test_a_fit_whose_walkers_need_the_third_grid_layout_scores_them(eng, oracle):
    """Until round 5 a walker whose adaptive grid outgrew the grid kernel's LDS layouts scored -inf (VAG_E_CAPACITY inside a fit).  A
    Fitter at resolution (0.06, 17, 6) gives every Gaussian-jet walker ~1400 theta nodes: the likelihood call lays the batch out with
    the third layout itself (scratch in HBM) and every walker gets ln L -- the fitter's formula on the checker's fluxes."""
    t, nu = configs.c4_mock_data()
    truth = oracle.flux_density(_abi.make_params(**configs.C4_TRUTH), t, nu)
    f = fitting.Fitter(z=configs.C4_TRUTH["z"], lumi_dist=configs.C4_TRUTH["lumi_dist"], jet="gaussian", medium="ism", resolution=(0.06, 17.0, 6.0))
    for b in configs.C4_BANDS:
        sel = nu == b
        f.add_flux_density(b, t[sel], truth[sel], 0.1 * truth[sel])
    defs = [fitting.ParamDef(n, 10.0 ** lo if lg else lo, 10.0 ** hi if lg else hi,
                             fitting.Scale.log if lg else fitting.Scale.linear) for n, lg, lo, hi in configs.C4_FREE]
    _, lo, hi = f.build_spec(defs)
    samples = lo + (hi - lo) * np.random.default_rng(3).random((6, len(defs)))
    got = f.loglike_batch(samples, defs)
    lib, h = eng
    pl = _lib.Plan()
    lib.vag_last_plan(h, C.byref(pl))
    assert pl.n_models_capacity == 0 and pl.n_models_ok == 6 and pl.n_rows > 6 * 1280 and np.all(np.isfinite(got))
    f._consolidate_data()
    for i, s in enumerate(samples):
        kw = dict(configs.C4_TRUTH, resolutions=(0.06, 17.0, 6.0))
        for (name, lg, _, _), v in zip(configs.C4_FREE, s):
            kw[{"theta_v": "theta_obs"}.get(name, name)] = 10 ** v if lg else v
        F = oracle.flux_density(_abi.make_params(**kw), f._all_t, f._all_nu)
        want = -0.5 * np.sum(f._all_weights * ((f._all_log_flux - np.log(np.maximum(F, 1e-300))) / f._all_log_err) ** 2)
        assert abs(got[i] - want) <= 1e-5 * max(1.0, abs(want)), (i, got[i], want)
    for _ in range(9):  # (back to the small layout)
        gpu_grid(eng, _abi.make_params(**configs.C1A), configs.C1_T, configs.C1_NU)


def test_the_grid_kernels_third_layout_gives_the_bits_of_the_lds_layouts(eng):
    """The grid kernel's scratch arrays in HBM (VAG_GRID_FORCE_LARGE=2) instead of LDS: the same program, the same numbers -- every
    grid array and the fluxes of a mixed batch (all six jets, on and off axis) bit for bit; and, downstream of it, everything that is
    strided by the layout: an SSC + Klein-Nishina batch with a reverse shock (the SSC band kernel then keeps its per-theta extrema in
    HBM) and a spreading SSC jet."""
    rng = np.random.default_rng(8)
    t, nu = np.logspace(3, 7, 10), np.array([1e9, 1e15])
    prms = []
    for i, jet in enumerate(["TophatJet", "GaussianJet", "PowerLawJet", "TwoComponentJet", "StepPowerLawJet", "PowerLawWing"] * 2):
        kw = dict(jet=jet, theta_c=rng.uniform(0.04, 0.2), E_iso=10 ** rng.uniform(51, 53), Gamma0=rng.uniform(50, 400), theta_obs=rng.uniform(0, 0.4) * (i % 2))
        if jet in ("TwoComponentJet", "StepPowerLawJet", "PowerLawWing"):
            kw.update(theta_w=kw["theta_c"] * 2.5, E_iso_w=kw["E_iso"] * 0.05, Gamma0_w=30.0)
        prms.append(_abi.make_params(**kw))
    ssc = [_abi.make_params(**dict(configs.C3, theta_obs=0.1 * k)) for k in range(3)]
    spread = [_abi.make_params(jet="GaussianJet", theta_obs=0.15, spreading=True, ssc=True)]

    def everything():
        out = [gpu_grid(eng, prms, t, nu)]
        out += list(gpu_components4(eng, ssc, configs.C3_T[::5], configs.C3_NU))
        out += list(gpu_components(eng, spread, t, nu))
        out += [a for p in prms for a in _grid_arrays(eng, p, t)]
        return out

    small = everything()
    _lib.hooks["VAG_GRID_FORCE_LARGE"] = "2"
    try:
        huge = everything()
    finally:
        _lib.hooks.pop("VAG_GRID_FORCE_LARGE")
        for _ in range(9):
            gpu_grid(eng, prms[0], t, nu)  # back to the small layout
    assert len(small) == len(huge) and small[1].max() > 0 and small[2].max() > 0 and small[4].max() > 0
    for a, b in zip(small, huge):
        assert np.array_equal(a, b, equal_nan=True)


def _engine_shape(eng, prm, t):
    lib, h = eng
    sh = _lib.DetailsShape()
    arr = (_lib.ModelParams * 1)(_lib.ModelParams.from_buffer_copy(bytes(prm)))
    _lib.check(lib.vag_details(h, arr, float(np.min(t)), float(np.max(t)), C.byref(sh), None))
    return tuple(getattr(sh, k) for k in _SHAPE_KEYS)


def _grid_arrays(eng, prm, t):
    """phi, theta and the source-frame lattice times of every row, as the grid kernel left them."""
    lib, h = eng
    sh = _lib.DetailsShape()
    arr = (_lib.ModelParams * 1)(_lib.ModelParams.from_buffer_copy(bytes(prm)))
    t_min, t_max = float(np.min(t)), float(np.max(t))
    _lib.check(lib.vag_details(h, arr, t_min, t_max, C.byref(sh), None))
    phi, th, t_src = np.zeros(sh.n_phi), np.zeros(sh.n_theta), np.zeros((sh.n_theta, sh.n_t))
    out = _lib.DetailsOut()
    out.phi, out.theta, out.t_src = phi.ctypes.data_as(dp), th.ctypes.data_as(dp), t_src.ctypes.data_as(dp)
    _lib.check(lib.vag_details(h, arr, t_min, t_max, C.byref(sh), C.byref(out)))
    return phi, th, t_src


def test_profile_evaluators_match_the_checker(eng, oracle):
    """Model.jet_E_iso / jet_Gamma0 / medium (pybind.cpp:441-448) for every named profile family."""
    theta = np.linspace(1e-4, 1.5, 97)
    r = np.logspace(14, 20, 61)
    jets = [va.TophatJet(0.1, 1e52, 300.0), va.GaussianJet(0.1, 1e52, 300.0), va.PowerLawJet(0.1, 1e52, 300.0, 2.0, 1.5),
            va.TwoComponentJet(0.05, 1e52, 300.0, 0.3, 1e50, 30.0), va.StepPowerLawJet(0.05, 1e52, 300.0, 3e51, 100.0, 3.0, 2.0),
            va.PowerLawWing(0.05, 3e51, 100.0, 3.0, 2.0), va.GaussianJet(0.1, 1e52, 300.0, magnetar=va.Magnetar(1e47, 1e3, 2.0))]
    media = [va.ISM(0.3), va.Wind(0.1), va.Wind(0.1, n_ism=1e-3, n0=10.0), va.Wind(0.1, k_m=1.5)]
    for jet in jets:
        m = va.Model(jet, media[0], va.Observer(1e28, 1.0, 0.1), va.Radiation(0.1, 0.01, 2.3))
        prm = _abi.ModelParams.from_buffer_copy(bytes(m.params))
        np.testing.assert_allclose(m.jet_E_iso(0.0, theta), oracle.profile(prm, 0, theta), rtol=1e-13, atol=0)
        np.testing.assert_allclose(m.jet_Gamma0(0.0, theta), oracle.profile(prm, 1, theta), rtol=1e-13, atol=0)
    for med in media:
        m = va.Model(jets[0], med, va.Observer(1e28, 1.0, 0.1), va.Radiation(0.1, 0.01, 2.3))
        prm = _abi.ModelParams.from_buffer_copy(bytes(m.params))
        np.testing.assert_allclose(m.medium(0.0, 0.0, r), oracle.profile(prm, 2, r), rtol=1e-12, atol=0)
    m = va.Model(jets[0], media[0], va.Observer(1e28, 1.0, 0.1), va.Radiation(0.1, 0.01, 2.3))
    assert m.jet_E_iso(0.0, [0.05, 0.2]).tolist() == [1e52, 0.0] and m.jet_Gamma0(0.0, [0.05, 0.2]).tolist() == [300.0, 1.0]
    assert abs(m.medium(0.0, 0.0, [1e17])[0] / (0.3 * 1.67e-24) - 1) < 1e-12


def test_fitter_predictions_at_a_sample(eng, oracle):
    """Fitter.model / flux_density_grid / flux (fitter.py:779-833,1089-1099): the Model at a point of sampler space, with
    the host-galaxy extinction applied per frequency to the grid."""
    k_law = lambda lam_cm: (5.5e-5 / np.asarray(lam_cm)) ** 1.1  # a smooth stand-in for a named extinction law
    f = fitting.Fitter(z=0.5, lumi_dist=3e28, jet="gaussian", medium="ism", extinction=k_law, fwd_ssc=True)
    t, nu = np.logspace(3, 7, 12), np.array([1e9, 4.84e14, 1e18])
    f.add_flux_density(4.84e14, t, np.full(t.size, 1e-28), np.full(t.size, 1e-29))
    P, S = fitting.ParamDef, fitting.Scale
    defs = [P("E_iso", 1e50, 1e54, S.log), P("theta_v", 0.0, 0.5, S.linear), P("A_V", 0.0, 2.0, S.linear),
            P("theta_c", 0.1, 0.1, S.fixed), P("Gamma0", 300.0, 300.0, S.fixed), P("n_ism", 1.0, 1.0, S.fixed),
            P("eps_e", 0.1, 0.1, S.fixed), P("eps_B", 0.01, 0.01, S.fixed), P("p", 2.3, 2.3, S.fixed)]
    f.validate_parameters(defs)
    sample = np.array([52.3, 0.17, 0.8])
    m = f.model(sample, defs)
    assert (m.params.E_iso, m.params.theta_obs, m.params.flags & 1) == (10 ** 52.3, 0.17, 1)
    prm = _abi.ModelParams.from_buffer_copy(bytes(m.params))
    want = oracle.flux_components4(prm, t, nu)
    raw = m.flux_density_grid(t, nu)
    assert_close(raw.fwd.sync, want[0])
    assert_close(raw.fwd.ssc, want[1], rtol=5e-6)
    att = np.exp(-0.8 * 0.4 * np.log(10.0) * k_law((2.99792458e10 / nu) / 1.5))[:, None]
    got = f.flux_density_grid(sample, t, nu, defs)
    assert np.array_equal(got.fwd.sync, raw.fwd.sync * att) and np.array_equal(got.fwd.ssc, raw.fwd.ssc * att)
    assert np.allclose(got.total, (raw.fwd.sync + raw.fwd.ssc) * att, rtol=1e-15) and got.rvs.sync.shape == ()
    assert_close(f.flux(sample, t, (1e17, 1e19), defs, num_points=9).total, oracle.flux(prm, t, 1e17, 1e19, 9))
    coarse = f.model(sample, defs, resolution=(0.1, 0.3, 8.0))
    assert coarse.resolutions == (0.1, 0.3, 8.0)
    with pytest.raises(ValueError):
        f.model(sample[:2], defs)


# ---- the reference's corner sweep (tests/python/test_parameter_corners.py) on the named profiles, against the checker ----
_CT = np.logspace(2, 7, 20)
_CNU = np.full_like(_CT, 1e14)
_CNUS = np.array([1e9, 1e14, 1e17])
_CJETS = {
    "tophat": lambda: va.TophatJet(0.1, 1e53, 300),
    "tophat_spread": lambda: va.TophatJet(0.1, 1e53, 300, spreading=True),
    "tophat_thick": lambda: va.TophatJet(0.1, 1e53, 300, duration=1000),
    "tophat_magnetar": lambda: va.TophatJet(0.1, 1e53, 300, magnetar=va.Magnetar(1e47, 1e3, 2)),
    "gaussian": lambda: va.GaussianJet(0.05, 1e53, 300),
    "powerlaw": lambda: va.PowerLawJet(0.05, 1e53, 300, 2, 1),
    "two_component": lambda: va.TwoComponentJet(0.05, 1e53, 300, 0.3, 1e51, 30),
    "step_powerlaw": lambda: va.StepPowerLawJet(0.05, 1e53, 300, 1e51, 30, 2, 1),
    "powerlaw_wing": lambda: va.PowerLawWing(0.05, 1e52, 100, 2, 1),
}
_CMEDIA = {"ism": lambda: va.ISM(1.0), "ism_thin": lambda: va.ISM(1e-4), "wind": lambda: va.Wind(0.1),
           "wind_full": lambda: va.Wind(0.5, n_ism=1.0, n0=1e6, k_m=1.5)}
_CRADS = {"plain": lambda: va.Radiation(0.1, 0.01, 2.3), "ssc": lambda: va.Radiation(0.1, 1e-4, 2.3, ssc=True),
          "ssc_kn": lambda: va.Radiation(0.1, 1e-4, 2.3, ssc=True, kn=True), "p_near2": lambda: va.Radiation(0.1, 0.01, 2.02),
          "p_steep": lambda: va.Radiation(0.3, 0.3, 2.9), "xi_e": lambda: va.Radiation(0.1, 0.01, 2.3, xi_e=0.1)}


def _corner_model(jet="tophat", medium="ism", rad="plain", off_axis=False, rvs=None, **kw):
    obs = va.Observer(3e28, 0.5, 0.4) if off_axis else va.Observer(3e28, 1.0, 0.0)
    return va.Model(_CJETS[jet](), _CMEDIA[medium](), obs, _CRADS[rad](), rvs_rad=_CRADS[rvs]() if rvs else None, **kw)


def _corner_check(m, oracle, rtol=5e-6, contract=False):
    prm = _abi.ModelParams.from_buffer_copy(bytes(m.params))
    f = m.flux_density(_CT, _CNU)
    assert f.total.shape == _CT.shape and np.all(np.isfinite(f.total)) and np.all(f.total > 0)
    want = oracle.flux_density(prm, _CT, _CNU)
    if contract:  # structured jets with a reverse shock: the reference's own builds differ by up to 7e-3
        assert within_contract(f.total, want)
    else:
        assert_close(f.total, want, rtol=rtol)
    return f, prm


@pytest.mark.parametrize("jet_name", sorted(_CJETS))
def test_corner_jets(eng, oracle, jet_name):
    _corner_check(_corner_model(jet=jet_name), oracle)


@pytest.mark.parametrize("medium_name", sorted(_CMEDIA))
def test_corner_media(eng, oracle, medium_name):
    _corner_check(_corner_model(medium=medium_name), oracle)


@pytest.mark.parametrize("rad_name", sorted(_CRADS))
def test_corner_radiation(eng, oracle, rad_name):
    _corner_check(_corner_model(rad=rad_name), oracle)


@pytest.mark.parametrize("rad_name", sorted(_CRADS))
def test_corner_reverse_shock(eng, oracle, rad_name):
    f, _ = _corner_check(_corner_model(jet="tophat_thick", rvs=rad_name), oracle)
    assert f.rvs.sync.shape == _CT.shape and np.all(np.isfinite(f.rvs.sync)) and np.all(f.rvs.sync >= 0)


@pytest.mark.parametrize("jet_name", ["tophat", "gaussian", "two_component"])
def test_corner_off_axis(eng, oracle, jet_name):
    m = _corner_model(jet=jet_name, off_axis=True)
    _, prm = _corner_check(m, oracle)
    grid = m.flux_density_grid(_CT, _CNUS).total
    assert grid.shape == (_CNUS.size, _CT.size) and np.all(np.isfinite(grid)) and np.all(grid > 0)
    assert_close(grid, oracle.flux_density_grid(prm, _CT, _CNUS), rtol=5e-6)


def test_corner_resolution_rtol_and_methods(eng, oracle):
    _corner_check(_corner_model(resolutions=(0.3, 1.0, 20), rtol=1e-4), oracle, rtol=2e-4)  # both sides integrate to 1e-4
    m = _corner_model()
    prm = _abi.ModelParams.from_buffer_copy(bytes(m.params))
    band = m.flux(_CT, 1e17, 1e19, 8).total
    assert band.shape == _CT.shape and np.all(band > 0)
    assert_close(band, oracle.flux(prm, _CT, 1e17, 1e19, 8))
    d = m.details(1e2, 1e6)
    assert np.all(d["Gamma"] >= 1) and np.all(d["r"] > 0) and np.all(np.isfinite(d["t_obs"]))
    fe = m.flux_density_exposures(_CT[:5], _CNU[:5], np.full(5, 10.0), 4).total
    assert fe.shape == (5,) and np.all(np.isfinite(fe)) and np.all(fe > 0)
    g = _corner_model(jet="gaussian")
    theta = np.array([0.0, 0.02, 0.05])
    assert np.all(g.jet_E_iso(0.0, theta) > 0) and np.all(g.jet_Gamma0(0.0, theta) > 1)


def test_series_components_of_ssc_and_reverse_shock_models(eng, oracle):
    """Model.flux_density returns FluxDict's components for a series too (pymodel.cpp:373-389): fwd.sync / fwd.ssc /
    rvs.sync / rvs.ssc of paired (t, nu) points == the diagonal of the component grids, short and long (chunked) series,
    and the exposure average keeps them apart."""
    m = va.Model(va.TophatJet(0.1, 1e53, 300, duration=1000), va.ISM(1.0), va.Observer(3e28, 1.0, 0.05),
                 va.Radiation(0.1, 1e-3, 2.3, ssc=True), rvs_rad=va.Radiation(0.1, 0.01, 2.5, ssc=True, kn=True))
    prm = _abi.ModelParams.from_buffer_copy(bytes(m.params))
    t = np.logspace(2, 7, 20)
    nu = np.tile([1e9, 1e14, 1e18, 1e24], 5)
    f = m.flux_density(t, nu)
    want = oracle.flux_components4(prm, t, np.array([1e9, 1e14, 1e18, 1e24]))  # [4 comps][nu][t]
    pick = np.tile(np.arange(4), 5)
    for got, w in zip((f.fwd.sync, f.fwd.ssc, f.rvs.sync, f.rvs.ssc), want):
        assert got.shape == t.shape
        assert_close(got, w[pick, np.arange(t.size)], rtol=5e-6)
    assert np.allclose(f.total, f.fwd.sync + f.fwd.ssc + f.rvs.sync + f.rvs.ssc, rtol=1e-15)
    assert_close(f.total, oracle.flux_density(prm, t, nu), rtol=5e-6)
    # a forward-only model keeps the 0-d placeholders
    plain = va.Model(va.TophatJet(0.1, 1e53, 300), va.ISM(1.0), va.Observer(3e28, 1.0, 0.05), va.Radiation(0.1, 1e-3, 2.3))
    assert plain.flux_density(t, nu).rvs.sync.shape == () and plain.flux_density(t, nu).fwd.ssc.shape == ()
    # long series: 700 sorted points go through the engine in chunks on one grid
    tl = np.logspace(2, 7, 700)
    nul = np.full(700, 1e14)
    fl = m.flux_density(tl, nul)
    wl = oracle.flux_components4(prm, tl, np.array([1e14]))
    for got, w in zip((fl.fwd.sync, fl.fwd.ssc, fl.rvs.sync, fl.rvs.ssc), wl):
        assert_close(got, w[0], rtol=5e-6)
    fe = m.flux_density_exposures(t[:6], nu[:6], np.full(6, 50.0), 5)
    assert fe.rvs.sync.shape == (6,) and np.all(fe.total > 0)
    assert np.allclose(fe.total, fe.fwd.sync + fe.fwd.ssc + fe.rvs.sync + fe.rvs.ssc, rtol=1e-14)


# ---- degenerate observation windows (tests/python/test_features.py:148-240): every time-lattice construction path ----
_GRID_PATHS = {
    "fwd_ism": dict(medium=("ism", 0.1), rvs=False, theta_obs=0.0),
    "fwd_wind": dict(medium=("wind", 0.1), rvs=False, theta_obs=0.0),
    "fwd_offaxis": dict(medium=("ism", 0.1), rvs=False, theta_obs=0.3),
    "rvs_ism": dict(medium=("ism", 0.1), rvs=True, theta_obs=0.0),
    "rvs_wind": dict(medium=("wind", 0.1), rvs=True, theta_obs=0.0),
}
_WINDOWS = [[10.0], [1e6], [10.0, 11.0], [1e3, 1.1e3], [1e6, 1.1e6]]


@pytest.mark.parametrize("path", sorted(_GRID_PATHS))
def test_degenerate_windows_all_grid_paths(eng, oracle, path):
    cfg = _GRID_PATHS[path]
    kind, value = cfg["medium"]
    medium = va.ISM(value) if kind == "ism" else va.Wind(value)
    rad = va.Radiation(0.1, 0.01, 2.3)
    m = va.Model(va.TophatJet(0.1, 1e52, 300, duration=1.0), medium, va.Observer(1e26, 0.1, cfg["theta_obs"]), rad,
                 rvs_rad=va.Radiation(0.1, 0.01, 2.3) if cfg["rvs"] else None)
    prm = _abi.ModelParams.from_buffer_copy(bytes(m.params))
    for window in _WINDOWS:
        t = np.array(window)
        total = m.flux_density_grid(t, np.array([1e14])).total
        assert np.all(np.isfinite(total)) and np.all(total > 0), (path, window)
        assert_close(total, oracle.flux_density_grid(prm, t, np.array([1e14])), rtol=5e-6)


def test_single_epochs_and_eat_nodes(eng, oracle):
    m = va.Model(va.TophatJet(0.1, 1e52, 300), va.ISM(0.1), va.Observer(1e26, 0.1, 0.0), va.Radiation(0.1, 0.5, 2.5))
    prm = _abi.ModelParams.from_buffer_copy(bytes(m.params))
    for t in np.logspace(0, 7, 15):
        total = m.flux_density_grid(np.array([t]), np.array([1e14, 1e17])).total
        assert np.all(total > 0), f"zero flux at single epoch t={t:g}s"
        assert_close(total, oracle.flux_density_grid(prm, np.array([t]), np.array([1e14, 1e17])), rtol=5e-6)
    for t_min, t_max in [(1.0, 2.0), (9.0, 11.0), (5.0, 1e3), (1e4, 1e6)]:
        d = m.details(t_min, t_max)
        assert np.all(np.isfinite(d["t_obs"])) and np.all(np.isfinite(d["Doppler"])) and np.all(d["Doppler"] > 0)


def test_details_object_matches_reference_layout(eng):
    """Model.details() is used like the reference's SimulationDetails (tests/python/test_features.py:100-110,
    test_advanced.py, test_parameter_corners.py:166-177): .fwd / .rvs ShockDetails with 3-D arrays."""
    m = va.Model(va.TophatJet(0.1, 1e52, 300), va.ISM(1.0), va.Observer(1e28, 1.0, 0.0), va.Radiation(0.1, 0.01, 2.2),
                 rvs_rad=va.Radiation(0.1, 0.01, 2.2))
    det = m.details(t_min=1e2, t_max=1e5)
    nth, nt = det["Gamma"].shape
    assert det.rvs is not None and det.rvs.Gamma.size > 0 and det.fwd.Gamma.shape == det.rvs.Gamma.shape == (1, nth, nt)
    assert det.t_src.shape == (1, nth, nt) and det.phi.ndim == 1 and det.theta.shape == (nth,)
    assert np.all(det.fwd.Gamma >= 1) and np.all(det.fwd.r > 0) and np.all(np.isfinite(det.fwd.t_obs))
    assert det.fwd.t_obs.shape[1:] == (nth, nt) and np.array_equal(det.fwd.t_obs, det.rvs.t_obs)  # one contact discontinuity
    assert np.array_equal(det.fwd.B_comv[0], det["B"]) and np.array_equal(det.rvs.Gamma[0], m.details(1e2, 1e5, rvs=True)["Gamma"])
    plain = va.Model(va.TophatJet(0.1, 1e52, 300), va.ISM(1.0), va.Observer(1e28, 1.0, 0.0), va.Radiation(0.1, 0.01, 2.2))
    assert plain.details(1e2, 1e5).rvs is None


def test_series_components_batch_equals_single_calls(eng):
    """vag_flux_density_components4_batch on a ragged batch == one call per model, bitwise, short and chunked series."""
    lib, h = eng
    kws = [dict(theta_obs=0.05, duration=1000.0, ssc=True, rvs=dict(eps_e=0.1, eps_B=0.01, p=2.5, ssc=True)),
           dict(jet="GaussianJet", theta_obs=0.2, duration=10.0, ssc=True, rvs=dict(eps_e=0.05, eps_B=0.02, p=2.3, ssc=True)),
           dict(theta_obs=0.0, E_iso=1e53, duration=300.0, ssc=True, rvs=dict(eps_e=0.2, eps_B=0.005, p=2.2, ssc=True))]
    prms = [_abi.make_params(**kw) for kw in kws]

    def run(ps, t, nu):
        arr = (_lib.ModelParams * len(ps))(*[_lib.ModelParams.from_buffer_copy(bytes(p)) for p in ps])
        comps = [np.empty((len(ps), t.size)) for _ in range(4)]
        out4 = (dp * 4)(*[a.ctypes.data_as(dp) for a in comps])
        _lib.check(lib.vag_flux_density_components4_batch(h, arr, len(ps), t.ctypes.data_as(dp), nu.ctypes.data_as(dp), t.size, out4))
        return comps

    for n in (30, 600):
        t = np.logspace(2, 7, n)
        nu = np.tile([1e9, 1e14, 1e18], n // 3)
        batch = run(prms, t, nu)
        for i, p in enumerate(prms):
            one = run([p], t, nu)
            assert all(np.array_equal(batch[c][i], one[c][0]) for c in range(4))
        assert all(np.all(np.isfinite(c)) for c in batch) and batch[2].max() > 0 and batch[3].max() > 0


def test_series_on_long_lattices_uses_fewer_wavefronts_per_workgroup(eng, oracle):
    """Long time lattices (t_resol = 25: K ~ 250) with SSC: the series kernel's private rows no longer fit four times into
    LDS, so the launch falls back to two or one wavefront per workgroup instead of failing."""
    for kw in (dict(jet="GaussianJet", theta_obs=0.2, ssc=True, resolutions=(0.1, 0.2, 25.0)),
               dict(theta_obs=0.05, duration=1000.0, rvs=dict(eps_e=0.1, eps_B=0.01, p=2.5), resolutions=(0.1, 0.2, 25.0))):
        prm = _abi.make_params(**kw)
        t = np.logspace(2, 7.5, 45)
        nu = np.tile([1e9, 4.84e14, 1e18], 15)
        assert_close(gpu_series(eng, prm, t, nu)[0], oracle.flux_density(prm, t, nu), rtol=5e-6)
        tl = np.logspace(2, 7.5, 300)
        assert_close(gpu_series(eng, prm, tl, np.full(300, 1e14))[0], oracle.flux_density(prm, tl, np.full(300, 1e14)), rtol=5e-6)


def test_long_time_axes_are_chunked_for_every_request_kind(eng, oracle):
    """nt * nnu beyond one launch's accumulator (and long lattices): total grids, component grids and band fluxes are
    computed in time chunks on ONE model grid and assembled; each equals the checker's un-chunked answer."""
    lib, h = eng
    prm = _abi.make_params(theta_obs=0.05, duration=300.0, ssc=True, rvs=dict(eps_e=0.1, eps_B=0.01, p=2.4, ssc=True))
    t = np.logspace(2, 7.5, 700)
    nu = np.logspace(9, 19, 8)
    got = gpu_components4(eng, prm, t, nu)
    want = oracle.flux_components4(prm, t, nu)
    for c in range(4):
        assert_close(got[c][0], want[c], rtol=5e-6)
    assert_close(gpu_grid(eng, prm, t, nu)[0], sum(want), rtol=5e-6)
    tb = np.logspace(2, 7.5, 500)
    band = np.empty((1, tb.size))
    arr = (_lib.ModelParams * 1)(_lib.ModelParams.from_buffer_copy(bytes(prm)))
    _lib.check(lib.vag_flux_batch(h, arr, 1, tb.ctypes.data_as(dp), tb.size, 1e17, 1e19, 17, band.ctypes.data_as(dp)))
    assert_close(band[0], oracle.flux(prm, tb, 1e17, 1e19, 17), rtol=5e-6)
    # a long lattice (t_resol 40 -> K ~ 390) with 16 frequencies: the chunk shrinks until the workgroup's LDS fits
    big = _abi.make_params(jet="GaussianJet", theta_obs=0.25, resolutions=(0.15, 0.5, 40.0))
    tg, nug = np.logspace(2, 8, 300), np.logspace(9, 19, 16)
    assert_close(gpu_grid(eng, big, tg, nug)[0], oracle.flux_density_grid(big, tg, nug), rtol=5e-6)


def test_wide_spectra_are_chunked_along_frequency(eng, oracle):
    """More frequencies than one launch carries (broad-band SEDs): the frequency axis is cut into chunks, the SSC seed
    band stays clamped over ALL requested frequencies (as one reference call does), and every component equals the
    checker's un-chunked answer; with a long time axis both cuts combine."""
    nu = np.logspace(8, 27, 150)
    t = np.logspace(3, 6.5, 6)
    syn = _abi.make_params(jet="GaussianJet", theta_obs=0.2)
    assert_close(gpu_grid(eng, syn, t, nu)[0], oracle.flux_density_grid(syn, t, nu), rtol=2e-6)
    prm = _abi.make_params(theta_obs=0.05, duration=300.0, ssc=True, kn=True, rvs=dict(eps_e=0.1, eps_B=0.01, p=2.4, ssc=True))
    got = gpu_components4(eng, prm, t, nu)
    want = oracle.flux_components4(prm, t, nu)
    for c in range(4):
        assert_close(got[c][0], want[c], rtol=5e-6)
    assert_close(gpu_grid(eng, prm, t, nu)[0], sum(want), rtol=5e-6)
    tl, nul = np.logspace(2, 7.5, 130), np.logspace(9, 24, 70)   # 70 x 130 slots: two frequency x three time chunks
    got = gpu_components4(eng, prm, tl, nul)
    want = oracle.flux_components4(prm, tl, nul)
    for c in range(4):
        assert_close(got[c][0], want[c], rtol=5e-6)


def test_loglike_with_more_data_points_than_one_series_launch(eng, oracle):
    """A fit with 700 point data (> 512 per series launch): ln L from the device == the fitter formula on the checker's
    fluxes (fitter.py:497-522)."""
    rng = np.random.default_rng(5)
    t = np.sort(10 ** rng.uniform(3, 7, 700))
    nu = rng.choice([3e9, 5.06e14, 2.41e17], size=700)
    truth = _abi.make_params(jet="GaussianJet", theta_obs=0.2)
    f_obs = oracle.flux_density(truth, t, nu) * (1 + 0.05 * rng.standard_normal(700))
    err = 0.1 * f_obs
    f = fitting.Fitter(z=truth.z, lumi_dist=truth.lumi_dist, jet="gaussian", medium="ism")
    f.add_flux_density(nu, t, f_obs, err)
    P, S = fitting.ParamDef, fitting.Scale
    defs = [P("E_iso", 1e51, 1e53, S.log), P("theta_v", 0.0, 0.5, S.linear), P("theta_c", 0.1, 0.1, S.fixed),
            P("Gamma0", 300.0, 300.0, S.fixed), P("n_ism", 1.0, 1.0, S.fixed), P("eps_e", 0.1, 0.1, S.fixed),
            P("eps_B", 0.01, 0.01, S.fixed), P("p", 2.3, 2.3, S.fixed)]
    samples = np.array([[52.0, 0.2], [52.3, 0.1], [51.6, 0.35]])
    got = f.loglike_batch(samples, defs)
    for s, g in zip(samples, got):
        prm = _abi.make_params(jet="GaussianJet", E_iso=10 ** s[0], theta_obs=s[1])
        model = np.maximum(oracle.flux_density(prm, t, nu), 1e-300)
        chi2 = np.sum(((np.log(f_obs) - np.log(model)) / (err / f_obs)) ** 2)
        assert abs(g - (-0.5 * chi2)) <= 2e-5 * abs(0.5 * chi2) + 1e-6


def test_bounds_mask_and_priors_run_on_the_device(eng, oracle):
    """log_prob_batch of fitting/samplers.py:72-91 in one device call: out-of-bounds walkers score -inf without being evaluated;
    the others ln L + sum ln prior with Uniform (default), Gaussian and LogUniform priors on the device and an arbitrary
    ``ln_prob`` object on the host.  The data cache must follow a change of the observations."""
    f, defs = _c4_fitter(oracle)
    rng = np.random.default_rng(3)
    _, lo, hi = f.build_spec(defs)
    samples = lo + (hi - lo) * rng.random((32, len(defs)))
    samples[3, 0] = hi[0] + 0.5   # above the upper bound
    samples[7, 5] = lo[5] - 1e-9  # just below the lower bound
    ll = f.loglike_batch(samples, defs)
    inside = np.all((samples >= lo) & (samples <= hi), axis=1)
    assert (~inside).sum() == 2 and np.isfinite(ll[~inside]).all()  # ln L alone does not mask
    got = f.log_prob_batch(samples, defs)
    want = np.where(inside, ll - np.sum(np.log(hi - lo)), -np.inf)
    assert np.array_equal(np.isfinite(got), np.isfinite(want))
    np.testing.assert_allclose(got[inside], want[inside], rtol=1e-13)
    assert f.last_plan.n_walkers_rejected == int((~np.isfinite(want)).sum())

    class Gaussian:  # shaped like bilby.core.prior.Gaussian
        def __init__(self, mu, sigma):
            self.mu, self.sigma = mu, sigma

    class LogUniform:
        def __init__(self, minimum, maximum):
            self.minimum, self.maximum = minimum, maximum

    class Triangle:  # anything with ln_prob stays on the host
        def ln_prob(self, x):
            return np.log(np.maximum(1e-300, 1 - np.abs(x - 0.4) / 0.5))

    names = [d.name for d in defs]
    i_p, i_tc, i_tv = names.index("p"), names.index("theta_c"), names.index("theta_v")
    priors = {"p": Gaussian(2.3, 0.2), "theta_c": LogUniform(0.01, 0.5), "theta_v": Triangle()}
    got2 = f.log_prob_batch(samples, defs, priors=priors)
    uni = [d for d in range(len(defs)) if d not in (i_p, i_tc, i_tv)]
    lp = -np.sum(np.log(hi[uni] - lo[uni]))
    lp = lp - 0.5 * ((samples[:, i_p] - 2.3) / 0.2) ** 2 - np.log(0.2 * np.sqrt(2 * np.pi))
    lp = lp - np.log(samples[:, i_tc] * np.log(0.5 / 0.01)) + Triangle().ln_prob(samples[:, i_tv])
    want2 = np.where(inside, ll + lp, -np.inf)
    np.testing.assert_allclose(got2[inside], want2[inside], rtol=1e-12)
    assert np.all(got2[~inside] == -np.inf)
    # same spec, new observations: the resident copy must be replaced
    f2, _ = _c4_fitter(oracle)
    f2._point_flux = [x * 1.3 for x in f2._point_flux]
    f2._all_t = None
    ll2 = f2.loglike_batch(samples[:4], defs)
    assert not np.allclose(ll2, ll[:4]) and np.allclose(f.loglike_batch(samples[:4], defs), ll[:4], rtol=1e-13)


@pytest.mark.parametrize("name", ["rs_gaussian_adiabatic", "gauss_ism_rs", "step_powerlaw_rs_spread"])
def test_rs_structured_jet_deviations_collapse_with_ode_tolerance(eng, oracle, name):
    """The three reverse-shock-on-structured-jet cases that are only held to the reference's golden contract at the default ODE
    tolerance (their low-Gamma wing rows amplify last-bit differences of the coupled 11-variable solve into a different step
    sequence; the reference's own -O3 and strict builds differ by 1e-3 ... 5e-3 on the same rows,
    profiles/r02_rs_structured_diagnostic.txt).  If that is the whole story the disagreement must vanish when BOTH sides
    integrate to rtol = 1e-9; a defect in the pair solver or the relic cooling would stay.  Measured (round 3): 2.0e-3 -> 3.6e-6,
    6.4e-3 -> 1.0e-5, 1.4e-2 -> 2.6e-6 (rvs.sync); forward shock <= 4e-7."""
    if name == "gauss_ism_rs":
        g = np.load(os.path.join(GOLDEN, name + ".npz"))
        prm, t, nu = _abi.params_from_golden_config(json.loads(str(g["config"]))), np.ascontiguousarray(g["t"]), np.ascontiguousarray(g["nus"])
    elif name == "step_powerlaw_rs_spread":
        prm, t, nu = _abi.make_params(**configs.PROFILE_CASES[name]), configs.SPREAD_T, configs.SPREAD_NU
    else:
        kw, t, nu = configs.RS_CASES[name]
        prm = _abi.make_params(**kw)
    prm.rtol = 1e-9
    want = oracle.flux_components4(prm, t, nu)
    got = gpu_components4(eng, prm, t, nu)
    for g_, w, comp in zip(got, want, COMPONENTS):
        if w.max() == 0:
            assert np.all(g_[0] == 0), comp
        else:
            assert_close(g_[0], w, rtol=5e-5 if comp == "rvs_sync" else 2e-6, floor=1e-2)


def test_profile_data_uses_the_reference_stage_names(eng):
    """Model.profile_data() (pybind.cpp:458-459): the stage names of AFTERGLOW_PROFILE_SCOPE (pymodel.h:877-953), filled from
    HIP events around each stage's kernels while the run-time profiler switch is on."""
    t, nu = configs.C3_T[::4], configs.C3_NU
    names = {"dynamics", "EAT_grid", "syn_electrons", "syn_photons", "cooling", "sync_flux", "ic_photons", "ssc_flux", "total"}
    plain = va.Model(va.GaussianJet(0.1, 1e52, 300), va.ISM(1.0), va.Observer(1e28, 1.0, 0.2), va.Radiation(0.1, 0.01, 2.3))
    full = va.Model(va.PowerLawJet(0.1, 1e52, 300, 2.0, 2.0, duration=1.0), va.Wind(0.1), va.Observer(1e28, 1.0, 0.2),
                    va.Radiation(0.1, 0.01, 2.3, ssc=True, kn=True), rvs_rad=va.Radiation(0.1, 0.01, 2.3, ssc=True, kn=True))
    va.Model.profile_enable(True)
    try:
        want = plain.flux_density_grid(t, nu).total
        p0 = va.Model.profile_data()
        _lib.hooks["VAG_NO_FUSED"] = "1"  # the fused synchrotron + SSC pass is booked under sync_flux: ask for separate passes
        full.flux_density_grid(t, nu)
        p1 = va.Model.profile_data()
    finally:
        _lib.hooks.pop("VAG_NO_FUSED", None)
        va.Model.profile_enable(False)
    assert set(p0) == names == set(p1)
    assert all(v >= 0 for v in p0.values()) and all(v >= 0 for v in p1.values())
    for k in ("dynamics", "syn_electrons", "sync_flux"):
        assert p0[k] > 0 and p1[k] > 0
    assert p0["cooling"] == p0["ic_photons"] == p0["ssc_flux"] == p0["syn_photons"] == 0.0  # no SSC: those stages never run
    assert p1["cooling"] > 0 and p1["ic_photons"] > 0 and p1["ssc_flux"] > 0 and p1["syn_photons"] > 0
    for p in (p0, p1):
        assert p["EAT_grid"] == 0.0  # fused into the flux kernels here
        assert sum(v for k, v in p.items() if k != "total") <= p["total"] * 1.001
    assert np.array_equal(plain.flux_density_grid(t, nu).total, want)  # the switch changes nothing but the timing records


def test_mixed_flag_batches_are_split_inside_the_call(eng, oracle):
    """The reference evaluates any mix of models side by side (samplers.py:59-91).  One launch sequence here serves one flag set,
    so a batch mixing plain, SSC, reverse-shock and spreading models is split by flags inside the host-pointer call and each
    model must come back exactly as it does in a call of its own -- grid, series, band and component outputs alike."""
    t, nu = np.logspace(3, 7, 12), np.array([1e9, 4.84e14, 1e18])
    kws = [dict(jet="GaussianJet", theta_obs=0.2), dict(jet="TophatJet", theta_obs=0.05, ssc=True),
           dict(jet="GaussianJet", theta_obs=0.3, n_ism=0.3), dict(jet="TophatJet", theta_obs=0.1, duration=50.0, rvs=dict(eps_e=0.1, eps_B=0.01, p=2.3)),
           dict(jet="TophatJet", theta_obs=0.05, ssc=True, kn=True), dict(jet="GaussianJet", theta_obs=0.15, spreading=True),
           dict(jet="TophatJet", theta_obs=0.05, ssc=True, eps_e=0.05)]
    prms = [_abi.make_params(**kw) for kw in kws]
    assert len({p.flags for p in prms}) == 5
    mixed = gpu_grid(eng, prms, t, nu)
    for i, p in enumerate(prms):
        assert np.array_equal(mixed[i], gpu_grid(eng, p, t, nu)[0]), i
        assert_close(mixed[i], oracle.flux_density_grid(p, t, nu), rtol=5e-5)  # (this test is about the split; the per-tier gates live above)
    ts, nus = np.repeat(t, 3), np.tile(nu, t.size)
    ser = gpu_series(eng, prms, ts, nus)
    for i, p in enumerate(prms):
        assert np.array_equal(ser[i], gpu_series(eng, p, ts, nus)[0]), i
    lib, h = eng
    arr = (_lib.ModelParams * len(prms))(*[_lib.ModelParams.from_buffer_copy(bytes(p)) for p in prms])
    band = np.empty((len(prms), t.size))
    _lib.check(lib.vag_flux_batch(h, arr, len(prms), t.ctypes.data_as(dp), t.size, 1e14, 1e15, 8, band.ctypes.data_as(dp)))
    one = np.empty((1, t.size))
    for i in (1, 3, 5):
        a1 = (_lib.ModelParams * 1)(_lib.ModelParams.from_buffer_copy(bytes(prms[i])))
        _lib.check(lib.vag_flux_batch(h, a1, 1, t.ctypes.data_as(dp), t.size, 1e14, 1e15, 8, one.ctypes.data_as(dp)))
        assert np.array_equal(band[i], one[0]), i
    comps = gpu_components4(eng, prms, t, nu)
    assert comps[1][0].max() == 0 and comps[1][1].max() > 0 and comps[2][3].max() > 0 and comps[2][0].max() == 0
    np.testing.assert_allclose(comps[0] + comps[1] + comps[2] + comps[3], mixed, rtol=1e-12)


def test_mixed_flag_batches_on_the_device_pointer_entry_points(eng):
    """vag_flux_density_grid_batch_dev / vag_flux_density_batch_dev with parameters, times and outputs resident in HBM and five
    different flag sets in one batch (what dist.sharded_flux_density_grid and the bench hand over): the models are sorted by flags on
    the way in and scattered back, each one bitwise as in a device call of its own flag group; the plan counts the whole batch."""
    import torch
    lib, h = eng
    t, nu = np.logspace(3, 7, 12), np.array([1e9, 4.84e14, 1e18])
    kws = [dict(jet="GaussianJet", theta_obs=0.2), dict(jet="TophatJet", theta_obs=0.05, ssc=True),
           dict(jet="GaussianJet", theta_obs=0.3, n_ism=0.3), dict(jet="TophatJet", theta_obs=0.1, duration=50.0, rvs=dict(eps_e=0.1, eps_B=0.01, p=2.3)),
           dict(jet="TophatJet", theta_obs=0.05, ssc=True, kn=True), dict(jet="GaussianJet", theta_obs=0.15, spreading=True),
           dict(jet="TophatJet", theta_obs=0.05, ssc=True, eps_e=0.05), dict(jet="GaussianJet", theta_c=-1.0)]  # the last one is invalid
    prms = [_abi.make_params(**kw) for kw in kws]
    dev = torch.device("cuda", 0)
    d_t, d_nu = torch.from_numpy(t).to(dev), torch.from_numpy(nu).to(dev)
    ts, nus = np.repeat(t, 3), np.tile(nu, t.size)
    d_ts, d_nus = torch.from_numpy(ts).to(dev), torch.from_numpy(nus).to(dev)

    def dev_params(ps):
        arr = (_lib.ModelParams * len(ps))(*[_lib.ModelParams.from_buffer_copy(bytes(p)) for p in ps])
        return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)

    def grid(ps):
        d_p = dev_params(ps)
        d_o = torch.full((len(ps), nu.size, t.size), -1.0, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        _lib.check(lib.vag_flux_density_grid_batch_dev(h, d_p.data_ptr(), len(ps), d_t.data_ptr(), t.size, d_nu.data_ptr(), nu.size, d_o.data_ptr()))
        _lib.check(lib.vag_ctx_synchronize(h))
        return d_o.cpu().numpy()

    def series(ps):
        d_p = dev_params(ps)
        d_o = torch.full((len(ps), ts.size), -1.0, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        _lib.check(lib.vag_flux_density_batch_dev(h, d_p.data_ptr(), len(ps), d_ts.data_ptr(), d_nus.data_ptr(), ts.size, d_o.data_ptr()))
        _lib.check(lib.vag_ctx_synchronize(h))
        return d_o.cpu().numpy()

    mixed = grid(prms)
    pl = _lib.Plan()
    lib.vag_last_plan(h, C.byref(pl))
    assert pl.n_models_ok == len(prms) - 1 and pl.n_models_invalid == 1
    assert np.isnan(mixed[-1]).all() and np.isfinite(mixed[:-1]).all() and mixed[:-1].min() >= 0
    mixed_s = series(prms)
    groups = {}
    for i, p in enumerate(prms):
        groups.setdefault(p.flags, []).append(i)
    assert len(groups) == 5
    for idx in groups.values():
        own, own_s = grid([prms[i] for i in idx]), series([prms[i] for i in idx])
        for q, i in enumerate(idx):
            assert np.array_equal(mixed[i], own[q], equal_nan=True), i
            assert np.array_equal(mixed_s[i], own_s[q], equal_nan=True), i
    assert np.array_equal(grid(prms), mixed, equal_nan=True)  # run to run


NONAXI_SPREAD = np.load(os.path.join(_abi.ROOT, "tests", "golden", "reference_nonaxi_spread.npz"))


@pytest.mark.parametrize("case", ["gauss_offaxis", "tophat_offaxis", "gauss_onaxis", "powerlaw_wind_ssc", "two_component_fine", "tophat_rs",
                                  "gauss_rs_ssc"])
def test_non_axisymmetric_spreading_jets_match_the_reference(eng, case):
    """Model(axisymmetric=False) with a spreading jet: one time lattice and one blast-wave solve per (phi, theta) node
    (grid-refinement.h:619-625, observer.cpp:51-141) -- the ODE rows are (phi, theta) pairs here.  Checked against vectors from the
    reference's own build (tests/golden/make_nonaxi_spread_fixture.py; the C restatement does not cover this combination): grid
    components, the paired series and the band integral, and the grid shape."""
    lib, h = eng
    fx = NONAXI_SPREAD
    meta = json.loads(str(fx["meta"]))[case]
    kw = dict(meta["kw"])
    if "resolutions" in kw:
        kw["resolutions"] = tuple(kw["resolutions"])
    prm = _abi.make_params(**kw)
    t, nu = fx["t"], fx["nu"]
    arr = (_lib.ModelParams * 1)(_lib.ModelParams.from_buffer_copy(bytes(prm)))
    comps = [np.empty((1, nu.size, t.size)) for _ in range(4)]
    out4 = (dp * 4)(*[a.ctypes.data_as(dp) for a in comps])
    _lib.check(lib.vag_flux_density_grid_components4_batch(h, arr, 1, t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp), nu.size, out4))
    sh = _lib.DetailsShape()
    _lib.check(lib.vag_details(h, arr, float(t.min()), float(t.max()), C.byref(sh), None))
    want_sh = meta["shape"]
    assert (sh.n_phi, sh.n_theta, sh.n_t, sh.n_reps) == (want_sh["n_phi"], want_sh["n_theta"], want_sh["n_t"], want_sh["n_reps"])

    def close(got, want, tol):
        m = want > 1e-9 * want.max()
        assert np.all(np.isfinite(got)) and m.any()
        err = np.max(np.abs(got - want)[m] / want[m])
        assert err < tol, err

    for got, name in zip(comps, ("sync", "ssc", "rvs_sync", "rvs_ssc")):
        want = fx[f"{case}__{name}"]
        if want.max() == 0:
            assert np.all(got == 0), name
        elif name.startswith("rvs") and kw["jet"] == "GaussianJet":
            # the reverse shock of a structured jet: low-Gamma wing rows of the coupled solver amplify last-bit differences -- the
            # reference's own two builds differ by 7e-3 there (DESIGN.md parity note 1) -- so the reference's golden contract applies
            assert np.all(np.abs(got[0] - want) <= 2e-3 * np.abs(want) + 1e-2 * want.max()), name
            # ... AND 3 x what the reference demonstrates on this very input (its two builds, one-ulp moves of an input:
            # tests/golden/sweep_sensitivity.json "nonaxi_rs"), like every other structured-jet reverse shock
            dem = _sweep_gate("nonaxi_rs")[case][name.replace("_", ".")]
            err = _sweep_err(got[0], want)
            assert err <= max(2e-6, 3 * dem), (name, err, dem)
        else:  # measured: forward solver 2e-11 ... 1e-8, coupled solver 1e-9 (top hat) ... 6e-6 (Gaussian)
            close(got[0], want, 2e-5 if kw.get("rvs") else 1e-6)
    sync, ssc = comps[0] + comps[2], comps[1] + comps[3]
    ts, nus = np.repeat(t, 3), np.tile(nu, t.size)
    tol_sum = 2e-4 if (kw.get("rvs") and kw["jet"] == "GaussianJet") else 2e-5  # (sums that hold the structured-jet reverse shock)
    close(gpu_series(eng, prm, ts, nus)[0], fx[f"{case}__series"], tol_sum)
    band = np.empty((1, t.size))
    _lib.check(lib.vag_flux_batch(h, arr, 1, t.ctypes.data_as(dp), t.size, 1e14, 1e15, 8, band.ctypes.data_as(dp)))
    close(band[0], fx[f"{case}__band"], tol_sum)
    # a batch of such models next to each other, and next to an axisymmetric spreading one, equals the single calls bit for bit
    other = _abi.make_params(**dict(kw, theta_obs=kw["theta_obs"] + 0.05))
    both = gpu_grid(eng, [prm, other], t, nu)
    assert np.array_equal(both[0], gpu_grid(eng, prm, t, nu)[0]) and np.array_equal(both[1], gpu_grid(eng, other, t, nu)[0])
    np.testing.assert_allclose(both[0], sync[0] + ssc[0], rtol=1e-12)


# ---------------------------------------------------------------------------------------------------------------
# Randomised sweeps against the checker (the first 16 draws of the boxes profiles/debug/prior_sweep_ssc.py and
# prior_sweep_rs_ssc.py walk; tests/sweeps.py regenerates them from the seeds).  Every draw and component is held to
# max(2e-6, 3 x what the REFERENCE demonstrates on that very input): tests/golden/sweep_sensitivity.json, written in the dev container
# by tests/golden/make_sweep_fixture.py from the reference's two builds and its response to a one-ulp move of one input.  Most draws
# demonstrate < 1e-8 and are therefore held to 2e-6; the few ill-conditioned ones (a Klein-Nishina cooling fixed point that stops at a
# 1e-3 change, reverse shocks on structured jets) get the allowance the reference itself needs, draw by draw.
# ---------------------------------------------------------------------------------------------------------------
def _sweep_gate(group):
    with open(os.path.join(GOLDEN, "sweep_sensitivity.json")) as f:
        return json.load(f)[group]


def _sweep_err(got, want):
    m = want > 1e-2 * want.max()
    return float(np.max(np.abs(got - want)[m] / want[m]))


def _within_golden_contract(got, want):
    """The reference's own golden contract (tests/python/golden/regenerate.py:29-30, test_golden.py:106): the outer cap of every
    per-draw gate below, whatever allowance the reference demonstrates on the draw."""
    return bool(np.all(np.abs(got - want) <= 2e-3 * np.abs(want) + 1e-2 * np.max(np.abs(want))))


_SHAPE_KEYS = ("n_phi", "n_theta", "n_t", "n_reps", "symmetry", "phi_mirrored")


def _grid_shapes(eng, oracle, prm, t):
    """(engine, checker) integers of the adaptive grid of one model for the request's time range: n_phi, n_theta, n_t, the number of
    representative rows, the symmetry level and the phi-mirror flag (auto_grid + Coord::detect_symmetry, grid-refinement.h:639-706,
    mesh.h:121-187) -- index work: equal or wrong."""
    lib, h = eng
    sh = _lib.DetailsShape()
    arr = (_lib.ModelParams * 1)(_lib.ModelParams.from_buffer_copy(bytes(prm)))
    _lib.check(lib.vag_details(h, arr, float(np.min(t)), float(np.max(t)), C.byref(sh), None))
    return tuple(getattr(sh, k) for k in _SHAPE_KEYS), oracle.grid_shape(prm, float(np.min(t)), float(np.max(t)))


def _theta_nodes(eng, oracle, prm, t):
    lib, h = eng
    sh = _lib.DetailsShape()
    arr = (_lib.ModelParams * 1)(_lib.ModelParams.from_buffer_copy(bytes(prm)))
    t_min, t_max = float(np.min(t)), float(np.max(t))
    _lib.check(lib.vag_details(h, arr, t_min, t_max, C.byref(sh), None))
    th = np.zeros(sh.n_theta)
    out = _lib.DetailsOut()
    out.theta = th.ctypes.data_as(dp)
    _lib.check(lib.vag_details(h, arr, t_min, t_max, C.byref(sh), C.byref(out)))
    return th, oracle.details(prm, t_min, t_max)["theta"]


def _assert_same_grid_shapes(eng, oracle, draws, t, what, known_duplicate_draws=()):
    """Every draw: the six grid integers equal to the checker's.  ONE difference is legitimate and is checked for what it is: the
    reference's theta grid can carry a node one ulp away from its neighbour -- merge_grids keeps values that are not bit-equal
    (grid-refinement.h:362-393), and the last quantile of inverse_CFD_sampling, interpolated towards pow(10, log10(theta_max)), need not
    land on theta_max itself (:138-189) -- a zero-width bin that the engine's own last-bit arithmetic may or may not reproduce.  Such a
    draw must agree in every other integer and in the node count once nodes closer than 1e-12 (relative) to their neighbour are
    counted once.  The draws it is KNOWN to happen on are named by the caller.  They are TOP-HAT jets, every one (draws i = 0 mod 3 of
    the forward-shock and reverse-shock sweeps, i = 0 mod 6 of the spreading one: 3 of 6 in the Klein-Nishina sweep, 5 of 6 in the Thomson
    and reverse-shock sweeps, 3 of 4 in the spreading sweep; draw 3 of the first has 38 nodes against 37): a top hat's theta grid ends on
    the jet edge, where the reference's last quantile pow(10, log10(theta_c)) is or is not theta_c to the last bit of glibc's pow and
    log10 -- which the engine's exp2 / log2 polynomials do not reproduce.  The twin nodes span a bin of zero solid angle.  At most one
    draw beyond the named ones may show it -- a library rebuilt with other last bits can move the coincidence to
    another draw -- and the message names every draw that did."""
    dup = []
    for i, p in enumerate(draws):
        got, want = _grid_shapes(eng, oracle, p, t)
        if got == want:
            continue
        g, w = dict(zip(_SHAPE_KEYS, got)), dict(zip(_SHAPE_KEYS, want))
        msg = f"{what} draw {i}: grid integers {g} != checker's {w}"
        assert all(g[k] == w[k] for k in ("n_phi", "n_t", "symmetry", "phi_mirrored")), msg
        th_g, th_w = _theta_nodes(eng, oracle, p, t)
        distinct = lambda th: 1 + int(np.sum(np.diff(th) > 1e-12 * th[1:]))
        assert distinct(th_g) == distinct(th_w), msg + f" (distinct theta nodes {distinct(th_g)} vs {distinct(th_w)})"
        assert g["n_theta"] - distinct(th_g) + w["n_theta"] - distinct(th_w) == abs(g["n_theta"] - w["n_theta"]), msg
        assert g["n_reps"] - w["n_reps"] in (0, g["n_theta"] - w["n_theta"]), msg  # (representative rows: one per theta node, or unaffected)
        dup.append(i)
    new = sorted(set(dup) - set(known_duplicate_draws))
    assert len(new) <= 1 and len(dup) <= len(known_duplicate_draws) + 1, \
        f"{what}: duplicate-node differences on draws {dup} (known: {list(known_duplicate_draws)}, not known: {new})"


@pytest.mark.parametrize("kn", [True, False], ids=["klein_nishina", "thomson"])
def test_random_forward_shock_ssc_draws_match_the_checker(eng, oracle, kn):
    import sweeps
    prms = sweeps.ssc_draws(16, kn)
    gate = _sweep_gate("sweep_ssc")
    _assert_same_grid_shapes(eng, oracle, prms, sweeps.SSC_T, "ssc " + ("kn" if kn else "thomson"), known_duplicate_draws=(3, 6, 9) if kn else (0, 3, 6, 9, 12))
    sync, ssc = gpu_components(eng, prms, sweeps.SSC_T, sweeps.SSC_NU)
    report = []
    for i, p in enumerate(prms):
        want = oracle.flux_components(p, sweeps.SSC_T, sweeps.SSC_NU)
        dem = gate[f"{'kn' if kn else 'thomson'}_{i}"]
        for name, g, w in (("fwd.sync", sync[i], want[0]), ("fwd.ssc", ssc[i], want[1])):
            assert np.all(np.isfinite(g)) and w.max() > 0, (i, name)
            assert _within_golden_contract(g, w), (i, name)
            err, tol = _sweep_err(g, w), max(2e-6, 3 * dem[name])
            report.append((err / tol, err, tol, i, name))
    worst = max(report)
    assert worst[0] <= 1.0, f"draw {worst[3]} {worst[4]}: rel. err {worst[1]:.2e} > gate {worst[2]:.2e}"
    assert np.median([r[1] for r in report]) < 1e-8  # the bulk agrees far below the gate


@pytest.mark.parametrize("seed", [60001, 60002, 60003])
def test_fresh_seed_draws_of_every_sweep_stay_inside_the_reference_contract(eng, oracle, seed):
    """The sweeps above are held to what the reference demonstrates on their sixteen named draws (a fixture from its own builds).  Here:
    three further seeds of each of the four generators -- 24 Klein-Nishina + 24 Thomson forward-shock SSC draws, 12 forward + reverse
    shock draws with SSC + KN on both, 12 spreading SSC draws over all six jets, 6 (phi, theta) pair-row draws: 234 models no earlier test
    or recorded sweep has seen -- against the checker under the reference's own golden contract, every component finite, the grid
    integers equal (up to the one-ulp twin node of top-hat jets: distinct theta nodes equal), and the forward-shock synchrotron
    component of the 144 non-spreading forward-shock draws to max(1e-6, 3 x what the REFERENCE demonstrates on that draw) over the bins
    above 1e-3 of the peak (tests/golden/fresh_seed_sensitivity.json from its own two builds: build spread and one-ulp response; 138 of
    the 144 demonstrate < 1e-7, a top hat seen from inside its cone 1.3e-4 -- the engine is at 3.1e-4 there), the median below 1e-8."""
    import json
    import sweeps
    errs = []
    dem = json.load(open(os.path.join(_abi.ROOT, "tests", "golden", "fresh_seed_sensitivity.json")))["demonstrated"]

    def check(name, got, want, tight):
        assert np.all(np.isfinite(got)), name
        if want.max() <= 0:
            assert got.max() <= 0, name
            return
        assert _within_golden_contract(got, want), name
        m = want > 1e-3 * want.max()
        e = float(np.max(np.abs(got - want)[m] / want[m]))
        if tight is not None:
            assert e <= max(1e-6, 3 * dem[tight]), (name, e, dem[tight])
            errs.append(e)

    def shapes(prm, t, name):
        g, w = _grid_shapes(eng, oracle, prm, t)
        if g != w:  # the twin node (see _assert_same_grid_shapes)
            th_g, th_w = _theta_nodes(eng, oracle, prm, t)
            distinct = lambda th: 1 + int(np.sum(np.diff(th) > 1e-12 * th[1:]))
            assert distinct(th_g) == distinct(th_w) and g[0] == w[0] and g[2] == w[2] and g[4:] == w[4:], (name, g, w)

    for kn in (True, False):
        prms = sweeps.ssc_draws(24, kn, seed=seed)
        sync, ssc = gpu_components(eng, prms, sweeps.SSC_T, sweeps.SSC_NU)
        for i, p in enumerate(prms):
            want = oracle.flux_components(p, sweeps.SSC_T, sweeps.SSC_NU)
            check(f"ssc kn={kn} #{i} sync", sync[i], want[0], f"{seed}_{'kn' if kn else 'thomson'}_{i}")
            check(f"ssc kn={kn} #{i} ssc", ssc[i], want[1], None)
            shapes(p, sweeps.SSC_T, f"ssc kn={kn} #{i}")
    prms = sweeps.rs_ssc_draws(12, seed=seed)
    comps = gpu_components4(eng, prms, sweeps.RS_T, sweeps.RS_NU)
    for i, p in enumerate(prms):
        want = oracle.flux_components4(p, sweeps.RS_T, sweeps.RS_NU)
        for c, name in enumerate(("fwd.sync", "fwd.ssc", "rvs.sync", "rvs.ssc")):
            check(f"rs #{i} {name}", comps[c][i], want[c], None)
        shapes(p, sweeps.RS_T, f"rs #{i}")
    prms = sweeps.spread_ssc_draws(12, seed=seed)
    sync, ssc = gpu_components(eng, prms, sweeps.SSC_T, sweeps.SSC_NU)
    for i, p in enumerate(prms):
        want = oracle.flux_components(p, sweeps.SSC_T, sweeps.SSC_NU)
        check(f"spread #{i} sync", sync[i], want[0], None)
        check(f"spread #{i} ssc", ssc[i], want[1], None)
    prms = sweeps.nonaxi_spread_draws(6, seed=seed)
    comps = gpu_components4(eng, prms, sweeps.NONAXI_T, sweeps.NONAXI_NU)
    for i, p in enumerate(prms):
        want = oracle.flux_components4(p, sweeps.NONAXI_T, sweeps.NONAXI_NU)
        for c in range(4):
            check(f"nonaxi #{i} component {c}", comps[c][i], want[c], None)
    assert len(errs) == 48 and np.median(errs) < 1e-8


def test_random_forward_reverse_shock_ssc_draws_match_the_checker(eng, oracle):
    import sweeps
    prms = sweeps.rs_ssc_draws(16)
    gate = _sweep_gate("sweep_rs_ssc")
    _assert_same_grid_shapes(eng, oracle, prms, sweeps.RS_T, "rs + ssc", known_duplicate_draws=(3, 6, 9, 12, 15))
    comps = gpu_components4(eng, prms, sweeps.RS_T, sweeps.RS_NU)
    names = ("fwd.sync", "fwd.ssc", "rvs.sync", "rvs.ssc")
    report = []
    for i, p in enumerate(prms):
        want = oracle.flux_components4(p, sweeps.RS_T, sweeps.RS_NU)
        for c, name in enumerate(names):
            g, w = comps[c][i], want[c]
            assert np.all(np.isfinite(g)), (i, name)
            if w.max() <= 0:
                continue
            assert _within_golden_contract(g, w), (i, name)
            err, tol = _sweep_err(g, w), max(2e-6, 3 * gate[str(i)][name])
            report.append((err / tol, err, tol, i, name))
    worst = max(report)
    assert worst[0] <= 1.0, f"draw {worst[3]} {worst[4]}: rel. err {worst[1]:.2e} > gate {worst[2]:.2e}"
    assert np.median([r[1] for r in report]) < 1e-6


def test_random_spreading_ssc_draws_match_the_checker(eng, oracle):
    """Spreading jets of all six profiles with SSC + Klein-Nishina: the first 24 draws of `SWEEP_MODE=spread
    profiles/debug/prior_sweep_ssc.py` (draw 13: rows whose observer times do not ascend along the lattice; draw 22: the spreading
    StepPowerLawJet with eps_B = 2e-6 on which the reference's own two builds differ by 4e-5 and one ulp of Gamma0 moves it by 4e-4),
    both components, each draw held to max(2e-6, 3 x what the reference demonstrates on it)."""
    import sweeps
    prms = sweeps.spread_ssc_draws(24)
    gate = _sweep_gate("sweep_spread_ssc")
    _assert_same_grid_shapes(eng, oracle, prms, sweeps.SSC_T, "spreading ssc", known_duplicate_draws=(3, 12, 18))
    sync, ssc = gpu_components(eng, prms, sweeps.SSC_T, sweeps.SSC_NU)
    report = []
    for i, p in enumerate(prms):
        want = oracle.flux_components(p, sweeps.SSC_T, sweeps.SSC_NU)
        for name, g, w in (("fwd.sync", sync[i], want[0]), ("fwd.ssc", ssc[i], want[1])):
            assert np.all(np.isfinite(g)) and w.max() > 0, (i, name)
            assert _within_golden_contract(g, w), (i, name)
            err, tol = _sweep_err(g, w), max(2e-6, 3 * gate[str(i)][name])
            report.append((err / tol, err, tol, i, name))
    worst = max(report)
    assert worst[0] <= 1.0, f"draw {worst[3]} {worst[4]}: rel. err {worst[1]:.2e} > gate {worst[2]:.2e}"
    assert np.median([r[1] for r in report]) < 1e-7


def test_random_non_axisymmetric_spreading_draws_match_the_checker(eng, oracle):
    """(phi, theta) pair rows against the C checker (which restates the mode since round 4 and equals the reference's strict build
    to the last digits on the seven named cases, tests/test_oracle.py): 8 draws over the jet profiles, every fourth with a reverse
    shock, each held to max(2e-6, 3 x what the reference demonstrates on that draw)."""
    import sweeps
    prms = sweeps.nonaxi_spread_draws(8)
    gate = _sweep_gate("sweep_nonaxi_spread")
    _assert_same_grid_shapes(eng, oracle, prms, sweeps.NONAXI_T, "non-axisymmetric spreading")
    comps = gpu_components4(eng, prms, sweeps.NONAXI_T, sweeps.NONAXI_NU)
    names = ("fwd.sync", "fwd.ssc", "rvs.sync", "rvs.ssc")
    report = []
    for i, p in enumerate(prms):
        want = oracle.flux_components4(p, sweeps.NONAXI_T, sweeps.NONAXI_NU)
        for c, name in enumerate(names):
            g, w = comps[c][i], want[c]
            assert np.all(np.isfinite(g)), (i, name)
            if w.max() <= 0:
                assert np.all(g == 0), (i, name)
                continue
            assert _within_golden_contract(g, w), (i, name)
            err, tol = _sweep_err(g, w), max(2e-6, 3 * gate[str(i)][name])
            report.append((err / tol, err, tol, i, name))
    worst = max(report)
    assert worst[0] <= 1.0, f"draw {worst[3]} {worst[4]}: rel. err {worst[1]:.2e} > gate {worst[2]:.2e}"
    assert np.median([r[1] for r in report]) < 1e-7


def _regimes(gamma_a, gamma_c, gamma_m):
    """determine_regime (src/radiation/synchrotron.cpp:45-60) on arrays: 1 ... 6 by the ordering of the three energies, 0 = none."""
    a, c, m = gamma_a, gamma_c, gamma_m
    out = np.zeros(a.shape, dtype=np.int32)
    for tag, cond in ((6, (c <= m) & (m <= a)), (5, (m <= c) & (c <= a)), (4, (c <= a) & (a <= m)), (3, (a <= c) & (c <= m)),
                      (2, (m <= a) & (a <= c)), (1, (a <= m) & (m <= c))):  # (the first match wins: assigned last)
        out[cond] = tag
    return out


@pytest.mark.parametrize("name", ["C2", "C3", "C3_rvs", "C4", "ssc_kn_ism"])
def test_regime_tags_match_the_checker_exactly(eng, oracle, name):
    """SynElectrons::regime per (theta, t) cell -- an integer tag, so equal or wrong: the engine's (vag_details_regime) against
    determine_regime applied to the checker's gamma_a / gamma_c / gamma_m, on configs[1], configs[2] (both shocks, IC-cooled
    electrons), the C4 truth model and an SSC + Klein-Nishina jet in a uniform medium.  A cell whose two closest energies agree to
    1e-9 in the checker is a tie that last-bit differences may order either way; such cells are counted and must stay rare."""
    lib, h = eng
    rvs = name.endswith("_rvs")
    kw = {"C2": configs.C2, "C3": configs.C3, "C3_rvs": configs.C3, "C4": configs.C4_TRUTH,
          "ssc_kn_ism": dict(configs.C2, theta_obs=0.2, ssc=True, kn=True)}[name]
    kw = dict(kw)
    if "resolutions" in kw:
        kw["resolutions"] = tuple(kw["resolutions"])
    prm = _abi.make_params(**kw)
    t = configs.C3_T if name.startswith("C3") else (configs.c4_mock_data()[0] if name == "C4" else configs.C2_T)
    od = oracle.details(prm, float(t.min()), float(t.max()), rvs=rvs)
    want = _regimes(od["gamma_a"], od["gamma_c"], od["gamma_m"])
    got = np.zeros(want.shape, dtype=np.int32)
    arr = (_lib.ModelParams * 1)(_lib.ModelParams.from_buffer_copy(bytes(prm)))
    _lib.check(lib.vag_details_regime(h, arr, float(t.min()), float(t.max()), 1 if rvs else 0, got.ctypes.data_as(C.POINTER(C.c_int32))))
    g = np.sort(np.stack([od["gamma_a"], od["gamma_c"], od["gamma_m"]]), axis=0)
    with np.errstate(divide="ignore", invalid="ignore"):
        tie = (g[1] - g[0] <= 1e-9 * g[1]) | (g[2] - g[1] <= 1e-9 * g[2])  # (inf - x = inf, inf - inf = nan: both compare False)
    # (cells of the reverse shock that holds no shocked matter yet carry NaN energies on both sides: every comparison fails, tag 0)
    assert set(np.unique(want)) <= {0, 1, 2, 3, 4, 5, 6}
    assert np.array_equal(got[~tie], want[~tie]), f"{int(np.sum(got[~tie] != want[~tie]))} of {int(np.sum(~tie))} cells differ"
    assert tie.mean() < 0.01, (int(tie.sum()), tie.size)


def test_null_and_empty_arguments_are_error_codes_not_crashes(eng):
    """The C-ABI is called by hand-written bindings: a null pointer or an empty array is VAG_E_INVALID with a
    message, never a fault, on every family of entry points; and the context works afterwards."""
    lib, h = eng
    prm = _abi.make_params()
    arr = (_lib.ModelParams * 1)(_lib.ModelParams.from_buffer_copy(bytes(prm)))
    t, nu, out = np.logspace(2, 6, 8), np.array([1e9, 1e15]), np.empty((1, 2, 8))
    T, N, O = t.ctypes.data_as(dp), nu.ctypes.data_as(dp), out.ctypes.data_as(dp)
    null, nullp = C.cast(None, dp), C.cast(None, C.POINTER(_lib.ModelParams))
    tt, nn, so = np.repeat(t, 2), np.tile(nu, 8), np.empty((1, 16))
    TT, NN, SO = tt.ctypes.data_as(dp), nn.ctypes.data_as(dp), so.ctypes.data_as(dp)
    o4 = (dp * 4)(O, null, null, null)
    bad = {
        "grid: null ctx": lambda: lib.vag_flux_density_grid_batch(None, arr, 1, T, 8, N, 2, O),
        "grid: null models": lambda: lib.vag_flux_density_grid_batch(h, nullp, 1, T, 8, N, 2, O),
        "grid: null t": lambda: lib.vag_flux_density_grid_batch(h, arr, 1, null, 8, N, 2, O),
        "grid: null nu": lambda: lib.vag_flux_density_grid_batch(h, arr, 1, T, 8, null, 2, O),
        "grid: null out": lambda: lib.vag_flux_density_grid_batch(h, arr, 1, T, 8, N, 2, null),
        "grid: nb 0": lambda: lib.vag_flux_density_grid_batch(h, arr, 0, T, 8, N, 2, O),
        "grid: nb < 0": lambda: lib.vag_flux_density_grid_batch(h, arr, -1, T, 8, N, 2, O),
        "grid: nt 0": lambda: lib.vag_flux_density_grid_batch(h, arr, 1, T, 0, N, 2, O),
        "grid: nnu 0": lambda: lib.vag_flux_density_grid_batch(h, arr, 1, T, 8, N, 0, O),
        "grid4: null nu": lambda: lib.vag_flux_density_grid_components4_batch(h, arr, 1, T, 8, null, 2, o4),
        "grid4: null out4": lambda: lib.vag_flux_density_grid_components4_batch(h, arr, 1, T, 8, N, 2, None),
        "series: null t": lambda: lib.vag_flux_density_batch(h, arr, 1, null, NN, 16, SO),
        "series: null nu": lambda: lib.vag_flux_density_batch(h, arr, 1, TT, null, 16, SO),
        "series: null out": lambda: lib.vag_flux_density_batch(h, arr, 1, TT, NN, 16, null),
        "series: n 0": lambda: lib.vag_flux_density_batch(h, arr, 1, TT, NN, 0, SO),
        "band: null out": lambda: lib.vag_flux_batch(h, arr, 1, T, 8, C.c_double(1e17), C.c_double(1e18), 5, null),
        "band: null t": lambda: lib.vag_flux_batch(h, arr, 1, null, 8, C.c_double(1e17), C.c_double(1e18), 5, SO),
        "grid_dev: null": lambda: lib.vag_flux_density_grid_batch_dev(h, None, 1, None, 8, None, 2, None),
        "series_dev: null": lambda: lib.vag_flux_density_batch_dev(h, None, 1, None, None, 16, None),
        "loglike: null spec": lambda: lib.vag_loglike_batch(h, None, T, 1, 8, O),
        "loglike_dev: null": lambda: lib.vag_loglike_batch_dev(h, None, None, 1, 8, None),
        "last_plan: null out": lambda: lib.vag_last_plan(h, None),
        "last_plan: null ctx": lambda: lib.vag_last_plan(None, C.byref(_lib.Plan())),
        "synchronize: null ctx": lambda: lib.vag_ctx_synchronize(None),
        "count_work: null ctx": lambda: lib.vag_ctx_count_work(None, 1),
        "set_stream: null ctx": lambda: lib.vag_ctx_set_stream(None, None),
        "coalesce: null ctx": lambda: lib.vag_ctx_coalesce(None, 8, 50),
        "details: null model": lambda: lib.vag_details(h, None, C.c_double(1e2), C.c_double(1e6), None, None),
        "validate: null model": lambda: lib.vag_params_validate(None),
        "ctx_create: null out": lambda: lib.vag_ctx_create(0, None),
    }
    for name, call in bad.items():
        rc = call()
        assert rc == _lib.VAG_E_INVALID, (name, rc)
        assert lib.vag_last_error(), name
    lib.vag_params_default(None)  # (returns nothing; must not fault)
    _lib.check(lib.vag_flux_density_grid_batch(h, arr, 1, T, 8, N, 2, O))
    assert np.all(np.isfinite(out)) and out.max() > 0
