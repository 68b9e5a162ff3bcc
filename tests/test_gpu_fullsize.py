"""Full-size GPU tests (-m gpu): BASELINE configs[3] (1024 walkers) and configs[4] (two-component SSC ensemble, 512 members) at the
sizes the bench runs them, checked against the oracle on sub-samples and through size-independent properties; and the sharded
evaluator over a world-size-1 RCCL group (the code path the multi-GPU bench and a sharded sampler use)."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest

import _abi
import configs
import vegasafterglow_amd as va
from vegasafterglow_amd import _lib, fitting

pytestmark = pytest.mark.gpu
dp = C.POINTER(C.c_double)


def gpu_series(eng, prms, t, nu):
    lib, h = eng
    arr = (_lib.ModelParams * len(prms))(*[_lib.ModelParams.from_buffer_copy(bytes(p)) for p in prms])
    t = np.ascontiguousarray(t, dtype=np.float64)
    nu = np.ascontiguousarray(nu, dtype=np.float64)
    out = np.empty((len(prms), t.size))
    _lib.check(lib.vag_flux_density_batch(h, arr, len(prms), t.ctypes.data_as(dp), nu.ctypes.data_as(dp), t.size, out.ctypes.data_as(dp)))
    return out


@pytest.fixture(scope="module")
def eng():
    lib = _lib.load()
    h, lock = va.get_context(0)
    return lib, h


def _c4_fitter(oracle):
    t, nu = configs.c4_mock_data()
    truth = oracle.flux_density(_abi.make_params(**configs.C4_TRUTH), t, nu)
    f_obs = truth * (1 + 0.05 * np.random.default_rng(42).standard_normal(t.size))
    f = fitting.Fitter(z=configs.C4_TRUTH["z"], lumi_dist=configs.C4_TRUTH["lumi_dist"], jet="gaussian", medium="ism")
    for b in configs.C4_BANDS:
        sel = nu == b
        f.add_flux_density(b, t[sel], f_obs[sel], 0.1 * f_obs[sel])
    defs = [fitting.ParamDef(n, 10.0 ** lo if lg else lo, 10.0 ** hi if lg else hi,
                             fitting.Scale.log if lg else fitting.Scale.linear) for n, lg, lo, hi in configs.C4_FREE]
    return f, defs


def _oracle_loglike(oracle, f, samples):
    f._consolidate_data()
    want = np.empty(len(samples))
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
    return want


def test_config3_1024_walkers_full_batch(eng, oracle):
    """configs[3] at full size: 1024 walkers drawn from the prior box in ONE vag_loglike_batch call.  ln L of a 64-walker
    sub-sample against the oracle; the batch is bitwise the same numbers as smaller batches and single-walker calls (walkers are
    independent units: batching must not change a bit); walkers outside the model's domain score -inf and are counted."""
    f, defs = _c4_fitter(oracle)
    _, lo, hi = f.build_spec(defs)
    rng = np.random.default_rng(0)
    samples = lo + (hi - lo) * rng.random((1024, len(defs)))
    bad = [17, 400, 1023]
    samples[bad[0], 2] = -0.1   # theta_c < 0
    samples[bad[1], 5] = 0.9    # p < 1
    samples[bad[2], 2] = 2.0    # theta_c > pi/2
    ll = f.loglike_batch(samples, defs)
    assert ll.shape == (1024,) and np.all(ll[bad] == -np.inf)
    assert np.isfinite(np.delete(ll, bad)).all()
    assert f.last_plan.n_walkers_rejected == len(bad) and f.last_plan.n_models_invalid == len(bad)
    sub = rng.choice(1024, 64, replace=False)
    want = _oracle_loglike(oracle, f, samples[sub])
    ok = np.isfinite(want)
    assert np.array_equal(np.isfinite(ll[sub]), ok)
    np.testing.assert_allclose(ll[sub][ok], want[ok], rtol=1e-5, atol=1e-6)
    assert np.array_equal(f.loglike_batch(samples, defs), ll)                  # run to run
    assert np.array_equal(f.loglike_batch(samples[256:512], defs), ll[256:512])  # a quarter of the batch
    for i in (0, 511, bad[0], 1000):
        assert np.array_equal(f.loglike_batch(samples[i:i + 1], defs), ll[i:i + 1])  # single-walker calls


def test_fit_kernel_partial_sums_do_not_depend_on_the_launch_shape(eng, oracle):
    """The likelihood's flux pass (vag_flux_fit_rows_kernel: one (theta, phi) row per lane) cuts every block of 64 rows into four
    lattice segments with a partial sum each and lets 1, 2 or 4 wavefronts walk them, chosen from the batch size: ln L must be
    the same bits for every choice (a wavefront arriving at a cut holds exactly what one starting there computes), and agree
    with the row-per-wavefront series kernel (another summation order) to rounding."""
    f, defs = _c4_fitter(oracle)
    _, lo, hi = f.build_spec(defs)
    samples = lo + (hi - lo) * np.random.default_rng(7).random((192, len(defs)))
    samples[11, 3] = 0.0  # an on-axis walker: one phi row per theta row
    base = f.loglike_batch(samples, defs)
    try:
        for w in ("1", "2", "4"):
            _lib.hooks["VAG_FIT_WAVES_PER_BLOCK"] = w
            assert np.array_equal(f.loglike_batch(samples, defs), base), w
        _lib.hooks.pop("VAG_FIT_WAVES_PER_BLOCK")
        _lib.hooks["VAG_SERIES_ROW_PER_WAVE"] = "1"
        other = f.loglike_batch(samples, defs)
    finally:
        _lib.hooks.pop("VAG_FIT_WAVES_PER_BLOCK", None)
        _lib.hooks.pop("VAG_SERIES_ROW_PER_WAVE", None)
    ok = np.isfinite(base)
    assert np.array_equal(np.isfinite(other), ok) and ok.sum() > 150
    np.testing.assert_allclose(base[ok], other[ok], rtol=1e-13)


def test_fit_kernel_items_beyond_the_first_come_from_the_counter_and_change_nothing(eng, oracle):
    """The likelihood's flux pass is a persistent launch: at most three workgroups per CU, every wavefront takes (model, block of 64
    rows, lattice segment) items -- its own number first, then from a device-wide counter -- until none is left.  4096 walkers are
    ~50 k items for ~3 k wavefronts, so nearly all of them come from the counter, in an order that differs from run to run: ln L must
    not (an item's partial sum does not depend on who serves it), and must be the bits the same walkers get in a batch small enough
    that no wavefront takes a second item.  The counter is back at zero after every call (a smaller batch right behind a larger one)."""
    f, defs = _c4_fitter(oracle)
    _, lo, hi = f.build_spec(defs)
    samples = lo + (hi - lo) * np.random.default_rng(11).random((4096, len(defs)))
    samples[100, 2] = -0.1  # a rejected walker in the middle: a model without blocks
    big = f.loglike_batch(samples, defs)
    assert big[100] == -np.inf and np.isfinite(np.delete(big, 100)).all()
    for _ in range(2):
        assert np.array_equal(f.loglike_batch(samples, defs), big)
    for a, b in ((0, 64), (90, 130), (4000, 4096)):
        assert np.array_equal(f.loglike_batch(samples[a:b], defs), big[a:b]), (a, b)
    assert np.array_equal(f.loglike_batch(samples, defs), big)


def test_cost_ordering_of_the_walkers_changes_no_bit(eng, oracle):
    """A likelihood call evaluates its walkers in descending order of the cost the same batch position had in the previous call
    (vag_order_kernel; VAG_NO_ORDER switches it off): theta is gathered, ln L scattered back.  Walkers are independent units, so ln L
    must be the same bits ordered or not, from the first call (no ranking yet) through calls ranked by a DIFFERENT batch's costs."""
    f, defs = _c4_fitter(oracle)
    _, lo, hi = f.build_spec(defs)
    rng = np.random.default_rng(23)
    a = lo + (hi - lo) * rng.random((512, len(defs)))
    b = lo + (hi - lo) * rng.random((512, len(defs)))
    _lib.hooks["VAG_NO_ORDER"] = "1"
    try:
        want_a, want_b = f.loglike_batch(a, defs), f.loglike_batch(b, defs)
    finally:
        _lib.hooks.pop("VAG_NO_ORDER")
    assert np.isfinite(want_a).sum() > 400
    for _ in range(2):  # first call of a batch size, then ranked by the other batch's costs, then by its own
        assert np.array_equal(f.loglike_batch(a, defs), want_a)
        assert np.array_equal(f.loglike_batch(b, defs), want_b)
        assert np.array_equal(f.loglike_batch(b, defs), want_b)


def test_fit_kernel_with_300_points_in_six_bands(eng, oracle):
    """A larger data set than the C4 mock (300 points, six bands: the eight-band instantiation, several slots per lane in the
    flush): the row-per-lane kernel against the row-per-wavefront kernel on 48 models, and against the oracle on two."""
    lib, h = eng
    rng = np.random.default_rng(11)
    bands = np.array([1.4e9, 1e10, 4.6e14, 8e14, 2.4e17, 1.2e18])
    t = np.sort(10 ** rng.uniform(3.5, 7.5, 300))
    nu = bands[rng.integers(0, bands.size, t.size)]
    prms = [_abi.make_params(**dict(configs.C4_TRUTH, jet="GaussianJet", E_iso=10 ** rng.uniform(51.5, 53), theta_obs=rng.uniform(0.0, 0.5),
                                    theta_c=rng.uniform(0.04, 0.2))) for _ in range(48)]
    got = gpu_series(eng, prms, t, nu)
    again = gpu_series(eng, prms, t, nu)
    assert np.array_equal(got, again) and np.all(np.isfinite(got)) and got.max() > 0
    _lib.hooks["VAG_SERIES_ROW_PER_WAVE"] = "1"
    try:
        other = gpu_series(eng, prms, t, nu)
    finally:
        del _lib.hooks["VAG_SERIES_ROW_PER_WAVE"]
    np.testing.assert_allclose(got, other, rtol=1e-12, atol=1e-300)
    for i in (0, 47):
        want = oracle.flux_density(prms[i], t, nu)
        m = want > 1e-9 * want.max()
        assert np.max(np.abs(got[i] - want)[m] / want[m]) < 5e-6


@pytest.mark.parametrize("case", ["two_component_ssc", "gaussian_offaxis", "rs_ssc_kn"])
def test_row_per_lane_grid_kernel_agrees_with_the_workgroup_kernel(eng, case):
    """Large batches with small (nu, t) grids go through vag_flux_grid_rows_kernel (a (theta, phi) row per lane, per-slot sums by
    LDS atomics in lane order); VAG_GRID_ROW_PER_WORKGROUP=1 keeps them on vag_flux_grid_kernel.  Same boundary values, same
    interpolation arithmetic, another summation order: every component agrees to rounding, and the row-per-lane result is the
    same bits from run to run."""
    from configs import c3_batch, c5_batch
    lib, h = eng
    if case == "two_component_ssc":
        prms = c5_batch(32)
    elif case == "rs_ssc_kn":
        prms = c3_batch(128)
    else:
        rng = np.random.default_rng(5)
        prms = [_abi.make_params(**dict(configs.C2, E_iso=10 ** rng.uniform(51, 53), theta_obs=rng.uniform(0.1, 0.4))) for _ in range(80)]
    t, nu = np.logspace(2, 8, 100), np.array([1e9, 4.84e14, 1e18, 2.4e26])
    nb = len(prms)
    arr = (_lib.ModelParams * nb)(*[_lib.ModelParams.from_buffer_copy(bytes(p)) for p in prms])

    def run():
        comps = [np.empty((nb, nu.size, t.size)) for _ in range(4)]
        out4 = (dp * 4)(*[a.ctypes.data_as(dp) for a in comps])
        _lib.check(lib.vag_flux_density_grid_components4_batch(h, arr, nb, t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp), nu.size, out4))
        pl = _lib.Plan()
        lib.vag_last_plan(h, C.byref(pl))
        return comps, pl.total_pairs, pl.pairs_per_block
    rows, pairs, ppb = run()
    assert pairs >= 4096 * 64 and ppb == 64  # the batch is large enough for the row-per-lane kernel (vag_capi.hip: run_flux_grid)
    again, _, _ = run()
    _lib.hooks["VAG_GRID_ROW_PER_WORKGROUP"] = "1"
    try:
        wg, _, _ = run()
    finally:
        del _lib.hooks["VAG_GRID_ROW_PER_WORKGROUP"]
    for a, a2, b in zip(rows, again, wg):
        assert np.array_equal(a, a2)
        assert np.all(np.isfinite(a))
        m = b > 1e-12 * np.maximum(b.max(axis=(1, 2), keepdims=True), 1e-300)
        assert np.all(np.abs(a - b)[m] <= 1e-11 * b[m])
        # the row-per-lane kernel skips a frequency far beyond the synchrotron cut-off when a whole wavefront is there (band_is_dead,
        # vag_grid_rows.h: the 2.4e26 Hz column of the synchrotron components): its zeros are the workgroup kernel's exact zeros,
        # and nothing it skipped was anything but zero
        assert np.array_equal(a == 0, b == 0)
    assert rows[0].max() > 0


def test_config4_two_component_ssc_ensemble_512_members(eng, oracle):
    """configs[4]: 512 members of the prior-predictive two-component SSC sweep (128 x 128 grids, 100 t x 4 nu incl. 2.4e26 Hz) in
    one call: sub-sample against the oracle per member, run-to-run determinism (bitwise), batch == sub-batch to rounding, exact
    1/d_L^2 scaling."""
    from configs import c5_batch
    lib, h = eng
    prms = c5_batch(512)
    t, nu = np.logspace(2, 8, 100), np.array([1e9, 4.84e14, 1e18, 2.4e26])
    arr = (_lib.ModelParams * 512)(*[_lib.ModelParams.from_buffer_copy(bytes(p)) for p in prms])
    out = np.empty((512, nu.size, t.size))

    def run(a, n, o):
        _lib.check(lib.vag_flux_density_grid_batch(h, a, n, t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp), nu.size,
                                                   o.ctypes.data_as(dp)))
    run(arr, 512, out)
    assert np.all(np.isfinite(out)) and np.all(out >= 0) and out[:, :3].min() > 0
    again = np.empty_like(out)
    run(arr, 512, again)
    assert np.array_equal(out, again)
    sub = np.empty((64, nu.size, t.size))
    sub_arr = (_lib.ModelParams * 64)(*[_lib.ModelParams.from_buffer_copy(bytes(p)) for p in prms[128:192]])
    run(sub_arr, 64, sub)
    # grid requests sum (theta, phi) rows in workgroup-sized groups chosen from the batch total (the partial grids are too large to
    # keep per fixed chunk as the series path does): a member's value is reproducible across batch sizes to rounding, not bitwise
    np.testing.assert_allclose(sub, out[128:192], rtol=1e-12)
    # against the REFERENCE itself (tests/golden/reference_spread.npz: both of its builds on these members).  Two-component jets
    # inherit the reference's own theta-grid sensitivity at the core edge -- its -O3 and strict builds differ by 1.7e-5 on member
    # 192 and 3.4e-6 on member 311 of this draw, by <= 1.5e-7 on the others -- so every member is held to max(2e-6, 3 x the spread
    # the reference shows ON THAT MEMBER), measured against the nearer of its two builds
    errs = _spread_gate("c5", out, [3, 77, 192, 200, 311, 480])
    assert np.median(errs) < 2e-6, errs
    far = (_lib.ModelParams * 8)(*[_lib.ModelParams.from_buffer_copy(bytes(p)) for p in prms[:8]])
    for q in far:
        q.lumi_dist *= 2
    o8 = np.empty((8, nu.size, t.size))
    run(far, 8, o8)
    np.testing.assert_allclose(out[:8] / o8, 4.0, rtol=1e-12)


def test_config4_at_the_full_4096_members(eng, oracle):
    """configs[4] at the size BASELINE.json names: 4096 members of the two-component SSC sweep in one call (0.9 M SSC tables, 67 M
    (theta, phi) rows).  Size-independent properties: every flux finite and non-negative, the first 64 members equal a 64-member
    call to summation rounding, members at the far end of the batch against the oracle."""
    from configs import c5_batch
    lib, h = eng
    n = 4096
    prms = c5_batch(n)
    t, nu = np.logspace(2, 8, 100), np.array([1e9, 4.84e14, 1e18, 2.4e26])
    arr = (_lib.ModelParams * n)(*[_lib.ModelParams.from_buffer_copy(bytes(p)) for p in prms])
    out = np.empty((n, nu.size, t.size))
    _lib.check(lib.vag_flux_density_grid_batch(h, arr, n, t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp), nu.size,
                                               out.ctypes.data_as(dp)))
    pl = _lib.Plan()
    lib.vag_last_plan(h, C.byref(pl))
    assert pl.n_models_ok == n and pl.n_models_ssc_rebuilt == 0
    assert np.all(np.isfinite(out)) and np.all(out >= 0) and out[:, :3].min() > 0
    head = np.empty((64, nu.size, t.size))
    sub = (_lib.ModelParams * 64)(*[_lib.ModelParams.from_buffer_copy(bytes(p)) for p in prms[:64]])
    _lib.check(lib.vag_flux_density_grid_batch(h, sub, 64, t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp), nu.size,
                                               head.ctypes.data_as(dp)))
    np.testing.assert_allclose(head, out[:64], rtol=1e-12)
    _spread_gate("c5", out, [4000, 4095])  # the reference's builds agree to 1.5e-9 on these two and do not move under one ulp: gate 2e-6


def _spread_gate(name, got, members, comps=None):
    """Per-member gate from tests/golden/reference_spread.npz (make_spread_fixture.py): rel. error of `got[member]` against the
    nearer of the reference's two builds (bins above 1e-9 of the member's peak) <= max(2e-6, what the reference itself demonstrates
    on that member: the spread between its builds, and the change of its strict build when theta_obs or Gamma0 moves by one ulp).
    Returns the errors."""
    fx = np.load(os.path.join(_abi.ROOT, "tests", "golden", "reference_spread.npz"))
    idx = {int(m): q for q, m in enumerate(fx[name + "_members"])}
    errs = []
    for i in members:
        fast, strict, ulp = fx[name + "_fast"][idx[i]], fx[name + "_strict"][idx[i]], fx[name + "_ulp"][idx[i]]
        g = got[i] if comps is None else np.stack([c[i] for c in comps])
        axes = tuple(range(g.ndim)) if comps is None else (1, 2)
        mask = strict > 1e-9 * np.maximum(strict.max(axis=axes, keepdims=True), 1e-300)
        rel = lambda a, b: np.where(mask, np.abs(a - b) / np.where(mask, b, 1.0), 0.0).max(axis=axes)
        demonstrated = np.maximum(rel(fast, strict), ulp)
        err = np.minimum(rel(g, fast), rel(g, strict))
        assert np.all(err <= np.maximum(2e-6, demonstrated)), (name, i, err, demonstrated)
        errs.append(float(np.max(err)))
    return errs


def test_config2_jittered_ensemble_members_against_both_reference_builds(eng):
    """The batch the bench times for configs[2] (c3_batch(128): forward + reverse shock, SSC + Klein-Nishina on both, jittered
    parameters) against the reference itself, per FluxDict component, on 16 of its members: each within max(2e-6, the reference's
    own sensitivity on that member and component: the spread between its two builds, 1e-11 ... 1.3e-6 here, and its response to a
    one-ulp change of theta_obs / Gamma0, up to 4.6e-4 in the reverse-shock components of single members)."""
    from configs import c3_batch
    lib, h = eng
    prms = c3_batch(128)
    fx = np.load(os.path.join(_abi.ROOT, "tests", "golden", "reference_spread.npz"))
    t, nu = fx["t"], fx["nu"]
    nb = len(prms)
    arr = (_lib.ModelParams * nb)(*[_lib.ModelParams.from_buffer_copy(bytes(p)) for p in prms])
    comps = [np.empty((nb, nu.size, t.size)) for _ in range(4)]
    out4 = (dp * 4)(*[a.ctypes.data_as(dp) for a in comps])
    _lib.check(lib.vag_flux_density_grid_components4_batch(h, arr, nb, t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp), nu.size, out4))
    assert all(np.all(np.isfinite(c)) for c in comps)
    errs = _spread_gate("c3", None, [int(m) for m in fx["c3_members"]], comps=comps)
    assert np.median(errs) < 1e-6, errs


def test_sharded_evaluator_over_rccl_world_size_1(eng, oracle):
    """The multi-GPU code path end to end on one GPU: a world-size-1 `nccl` (= RCCL) process group, dist.WalkerSharder over
    Fitter.device_evaluator (theta and ln L stay in HBM, one all-gather of [ln L | cost]), cost feedback from
    vag_last_model_costs_dev, and the ensemble all-gather of sharded_flux_density_grid."""
    import torch
    import torch.distributed as dist
    from vegasafterglow_amd.dist import WalkerSharder, sharded_flux_density_grid
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        f, defs = _c4_fitter(oracle)
        _, lo, hi = f.build_spec(defs)
        samples = lo + (hi - lo) * np.random.default_rng(4).random((96, len(defs)))
        samples[5, 2] = -0.5
        want = f.loglike_batch(samples, defs)
        dev = torch.device("cuda", 0)
        sharder = WalkerSharder(f.device_evaluator(defs), device=dev)
        d_theta = torch.from_numpy(samples).to(dev)
        got = sharder(d_theta)
        assert got.device.type == "cuda"
        assert np.array_equal(got.cpu().numpy(), want)
        assert sharder.native is not None  # the engine's own deal / scatter (vag_loglike_shard_dev), not the torch statement of it
        got2 = sharder(d_theta)  # second call: dealt by the engine's cost report
        assert np.array_equal(got2.cpu().numpy(), want)
        plain = f.device_evaluator(defs)
        generic = WalkerSharder(lambda th: plain(th), device=dev)  # no .native: the same deal from torch operations
        assert generic.native is None
        for _ in range(2):
            assert np.array_equal(generic(d_theta).cpu().numpy(), want)
        assert np.array_equal(generic.last_table, sharder.last_table)
        np.testing.assert_allclose(generic.costs, sharder.costs, rtol=1e-15)
        costs = sharder.costs_per_rank()
        assert costs is not None and costs.shape == (1,) and costs[0] > 0
        assert sharder.costs.min() > 0 and sharder.costs.max() / sharder.costs.min() > 1.5  # ragged grids: cost varies
        lp = WalkerSharder(f.device_evaluator(defs, use_priors=True), device=dev)(d_theta).cpu().numpy()
        np.testing.assert_allclose(lp[np.isfinite(lp)], (want - np.sum(np.log(hi - lo)))[np.isfinite(lp)], rtol=1e-13)
        assert lp[5] == -np.inf
        # ensemble path: 6 C1b models, gathered
        lib, h = eng
        prms = [_abi.make_params(**dict(configs.C1B, E_iso=1e52 * (1 + 0.1 * i))) for i in range(6)]
        t, nu = configs.C1_T, configs.C1_NU

        def eval_dev(block):
            arr = (_lib.ModelParams * len(block))(*[_lib.ModelParams.from_buffer_copy(bytes(p)) for p in block])
            o = np.empty((len(block), nu.size, t.size))
            _lib.check(lib.vag_flux_density_grid_batch(h, arr, len(block), t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp),
                                                       nu.size, o.ctypes.data_as(dp)))
            return torch.from_numpy(o).to(dev)
        full = sharded_flux_density_grid(prms, eval_dev, (nu.size, t.size), device=dev)
        assert full.shape == (6, nu.size, t.size)
        np.testing.assert_array_equal(full.cpu().numpy(), eval_dev(prms).cpu().numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_native_shard_entry_points_reproduce_the_plain_call_for_every_world(eng, oracle, world):
    """The N > 1 deal without N GPUs: vag_loglike_shard_dev is called once per (simulated) rank on this GPU, the blocks are
    concatenated in rank order (what the all-gather does) and vag_loglike_shard_finish_dev scatters them.  ln L must be the bits
    of one plain vag_loglike_batch_dev call whatever the deal; the device's table must be the numpy statement of the deal
    (dist.balanced_assignment) on the gathered costs; the second deal balances the ranks' cost sums."""
    import torch
    from vegasafterglow_amd.dist import balanced_assignment, shard_range
    lib, h = eng
    f, defs = _c4_fitter(oracle)
    spec, lo, hi = f.build_spec(defs)
    nb, ndim = 203, len(defs)  # ragged: the last sweep has padding slots for every world here
    samples = lo + (hi - lo) * np.random.default_rng(21).random((nb, ndim))
    samples[7, 2] = -0.5  # invalid walker: -inf, cost 0 -> assumed average in the next deal
    want = f.loglike_batch(samples, defs)
    dev = torch.device("cuda", 0)
    d_theta = torch.from_numpy(samples).to(dev)
    per = -(-nb // world)
    _lib.check(lib.vag_ctx_set_stream(h, _lib.torch_stream_handle(torch.cuda.current_stream(dev))))
    try:
        prev_costs = np.ones(nb)
        for call in range(3):
            blocks, tables = [], []
            for rank in range(world):
                blk = torch.empty((per, 2), dtype=torch.float64, device=dev)
                _lib.check(lib.vag_loglike_shard_dev(h, C.byref(spec), d_theta.data_ptr(), nb, ndim, rank, world, blk.data_ptr()))
                tab = torch.empty((world * per,), dtype=torch.int32, device=dev)
                _lib.check(lib.vag_loglike_shard_state_dev(h, nb, world, tab.data_ptr(), None))
                blocks.append(blk)
                tables.append(tab.cpu().numpy().reshape(world, per))
            for tab in tables[1:]:
                assert np.array_equal(tab, tables[0])  # every rank computes the same deal
            assert np.array_equal(tables[0], balanced_assignment(prev_costs, world))
            gathered = torch.cat(blocks, 0).contiguous()
            out = torch.full((nb,), 123.0, dtype=torch.float64, device=dev)
            _lib.check(lib.vag_loglike_shard_finish_dev(h, gathered.data_ptr(), nb, world, out.data_ptr()))
            cost = torch.empty((nb,), dtype=torch.float64, device=dev)
            _lib.check(lib.vag_loglike_shard_state_dev(h, nb, world, None, cost.data_ptr()))
            got, g = out.cpu().numpy(), gathered.cpu().numpy()
            assert np.array_equal(got, want), call
            pad = tables[0].reshape(-1) < 0
            assert np.isnan(g[pad, 0]).all() and np.all(g[pad, 1] == 0) and pad.sum() == world * per - nb
            raw = np.ones(nb)
            raw[tables[0].reshape(-1)[~pad]] = g[~pad, 1]
            assert raw[7] == 0 and (np.delete(raw, 7) > 0).all()
            prev_costs = np.where(raw > 0, raw, raw[raw > 0].mean())
            np.testing.assert_allclose(cost.cpu().numpy(), prev_costs, rtol=1e-15)
            sums = np.array([prev_costs[row[row >= 0]].sum() for row in tables[0]])
            if call >= 1:  # dealt by the previous call's costs: near-equal cost sums, better than blocks of equal count
                by_count = np.array([prev_costs[a:b].sum() for a, b in (shard_range(nb, r, world) for r in range(world))])
                assert sums.max() / sums.mean() < 1.03 and sums.max() / sums.mean() <= by_count.max() / by_count.mean()
        # a finish without a matching shard call, and a rank outside the world, are refused
        assert lib.vag_loglike_shard_finish_dev(h, gathered.data_ptr(), nb, world, out.data_ptr()) == _lib.VAG_E_INVALID
        assert lib.vag_loglike_shard_dev(h, C.byref(spec), d_theta.data_ptr(), nb, ndim, world, world, blk.data_ptr()) == _lib.VAG_E_INVALID
    finally:
        torch.cuda.synchronize()
        _lib.check(lib.vag_ctx_set_stream(h, None))


def test_two_fitters_alternating_on_one_context_keep_their_own_cost_ranking(eng, oracle):
    """The gathered costs a deal ranks by belong to (fit data, batch size, world): two fitters of equal batch size that take
    turns on the shared context must each be dealt by THEIR OWN previous call, and a finish must belong to the shard call
    before it.  (Second fitter: the same bands over a longer campaign -> longer lattices -> other per-walker costs.)"""
    import torch
    from vegasafterglow_amd.dist import balanced_assignment
    lib, h = eng
    fa, defs = _c4_fitter(oracle)
    t, nu = configs.c4_mock_data()
    tb = t * (t / t.min()) ** 0.7  # a longer campaign: more lattice nodes per walker
    truth = oracle.flux_density(_abi.make_params(**configs.C4_TRUTH), tb, nu)
    fb = fitting.Fitter(z=configs.C4_TRUTH["z"], lumi_dist=configs.C4_TRUTH["lumi_dist"], jet="gaussian", medium="ism")
    for b in configs.C4_BANDS:
        sel = nu == b
        fb.add_flux_density(b, tb[sel], truth[sel], 0.1 * truth[sel])
    sa, lo, hi = fa.build_spec(defs)
    sb, _, _ = fb.build_spec(defs)
    nb, ndim, world = 96, len(defs), 2
    per = nb // world
    samples = lo + (hi - lo) * np.random.default_rng(33).random((nb, ndim))
    want = {"a": fa.loglike_batch(samples, defs), "b": fb.loglike_batch(samples, defs)}
    dev = torch.device("cuda", 0)
    d_theta = torch.from_numpy(samples).to(dev)
    _lib.check(lib.vag_ctx_set_stream(h, _lib.torch_stream_handle(torch.cuda.current_stream(dev))))

    def sharded_call(spec):
        blocks = []
        for rank in range(world):
            blk = torch.empty((per, 2), dtype=torch.float64, device=dev)
            _lib.check(lib.vag_loglike_shard_dev(h, C.byref(spec), d_theta.data_ptr(), nb, ndim, rank, world, blk.data_ptr()))
            blocks.append(blk)
        tab = torch.empty((world * per,), dtype=torch.int32, device=dev)
        _lib.check(lib.vag_loglike_shard_state_dev(h, nb, world, tab.data_ptr(), None))
        out = torch.empty((nb,), dtype=torch.float64, device=dev)
        gathered = torch.cat(blocks, 0).contiguous()
        _lib.check(lib.vag_loglike_shard_finish_dev(h, gathered.data_ptr(), nb, world, out.data_ptr()))
        cost = torch.empty((nb,), dtype=torch.float64, device=dev)
        _lib.check(lib.vag_loglike_shard_state_dev(h, nb, world, None, cost.data_ptr()))
        return out.cpu().numpy(), tab.cpu().numpy().reshape(world, per), cost.cpu().numpy()

    try:
        prev = {"a": np.ones(nb), "b": np.ones(nb)}
        for turn, key in enumerate("ababab"):
            got, table, cost = sharded_call(sa if key == "a" else sb)
            assert np.array_equal(got, want[key]), (turn, key)
            assert np.array_equal(table, balanced_assignment(prev[key], world)), (turn, key)  # ranked by ITS OWN last call
            prev[key] = cost
        assert not np.array_equal(prev["a"], prev["b"])  # the two problems do have different cost profiles
    finally:
        torch.cuda.synchronize()
        _lib.check(lib.vag_ctx_set_stream(h, None))


def test_two_sharded_calls_in_flight_on_one_context_finish_with_their_own_deals(eng, oracle):
    """ABI v11: the deal of a call in flight is kept per (batch size, world, spec), so another sharder of the same process may deal on
    the shared context between a call's shard and its finish (the Python side no longer holds the context lock across the all-gather:
    a process-local lock held while waiting for other ranks can deadlock).  Two fitters, same batch shape: A deals, B deals, A
    finishes, B finishes -- and the other way round -- each ln L vector the bits of its plain call; finishes of equal shape are served in
    the order the calls were dealt."""
    import torch
    lib, h = eng
    fa, defs = _c4_fitter(oracle)
    t, nu = configs.c4_mock_data()
    tb = t * (t / t.min()) ** 0.7
    truth = oracle.flux_density(_abi.make_params(**configs.C4_TRUTH), tb, nu)
    fb = fitting.Fitter(z=configs.C4_TRUTH["z"], lumi_dist=configs.C4_TRUTH["lumi_dist"], jet="gaussian", medium="ism")
    for b in configs.C4_BANDS:
        sel = nu == b
        fb.add_flux_density(b, tb[sel], truth[sel], 0.1 * truth[sel])
    sa, lo, hi = fa.build_spec(defs)
    sb, _, _ = fb.build_spec(defs)
    nb, ndim, world = 80, len(defs), 2
    per = nb // world
    samples = lo + (hi - lo) * np.random.default_rng(34).random((nb, ndim))
    want = {"a": fa.loglike_batch(samples, defs), "b": fb.loglike_batch(samples, defs)}
    assert not np.array_equal(want["a"], want["b"])
    dev = torch.device("cuda", 0)
    d_theta = torch.from_numpy(samples).to(dev)
    _lib.check(lib.vag_ctx_set_stream(h, _lib.torch_stream_handle(torch.cuda.current_stream(dev))))

    def shard(spec):
        blocks = []
        for rank in range(world):
            blk = torch.empty((per, 2), dtype=torch.float64, device=dev)
            _lib.check(lib.vag_loglike_shard_dev(h, C.byref(spec), d_theta.data_ptr(), nb, ndim, rank, world, blk.data_ptr()))
            blocks.append(blk)
        return torch.cat(blocks, 0).contiguous()

    def finish(gathered):
        out = torch.empty((nb,), dtype=torch.float64, device=dev)
        _lib.check(lib.vag_loglike_shard_finish_dev(h, gathered.data_ptr(), nb, world, out.data_ptr()))
        return out.cpu().numpy()

    try:
        for first, second in (("a", "b"), ("b", "a"), ("a", "b")):
            specs = {"a": sa, "b": sb}
            g1 = shard(specs[first])
            g2 = shard(specs[second])  # deals on the same context while the first call waits for its "all-gather"
            assert np.array_equal(finish(g1), want[first])
            assert np.array_equal(finish(g2), want[second])
        assert lib.vag_loglike_shard_finish_dev(h, g1.data_ptr(), nb, world, torch.empty((nb,), dtype=torch.float64, device=dev).data_ptr()) == _lib.VAG_E_INVALID
    finally:
        torch.cuda.synchronize()
        _lib.check(lib.vag_ctx_set_stream(h, None))


def test_ticketed_sharded_calls_finish_in_any_order_and_twice_for_one_fit(eng, oracle):
    """ABI v13: a call in flight is named by the ticket vag_loglike_shard_begin_dev returns.  Two fits of EQUAL shape dealt A,
    B and finished B, A -- the order the unticketed finish ("oldest deal of this shape") gets wrong: it would apply B's deal to A's
    block --; two calls of the SAME fit in flight at once (different samples), both finished, second first; a ticket finished twice, a
    ticket with the wrong shape and an abandoned ticket are handled; after all of it the fit's cost cache ranks the next deal."""
    import torch
    lib, h = eng
    fa, defs = _c4_fitter(oracle)
    t, nu = configs.c4_mock_data()
    tb = t * (t / t.min()) ** 0.7
    truth = oracle.flux_density(_abi.make_params(**configs.C4_TRUTH), tb, nu)
    fb = fitting.Fitter(z=configs.C4_TRUTH["z"], lumi_dist=configs.C4_TRUTH["lumi_dist"], jet="gaussian", medium="ism")
    for b in configs.C4_BANDS:
        sel = nu == b
        fb.add_flux_density(b, tb[sel], truth[sel], 0.1 * truth[sel])
    sa, lo, hi = fa.build_spec(defs)
    sb, _, _ = fb.build_spec(defs)
    nb, ndim, world = 96, len(defs), 3
    per = nb // world
    rng = np.random.default_rng(77)
    s1, s2 = (lo + (hi - lo) * rng.random((nb, ndim)) for _ in range(2))
    want = {("a", 1): fa.loglike_batch(s1, defs), ("b", 1): fb.loglike_batch(s1, defs), ("a", 2): fa.loglike_batch(s2, defs)}
    dev = torch.device("cuda", 0)
    d1, d2 = torch.from_numpy(s1).to(dev), torch.from_numpy(s2).to(dev)
    _lib.check(lib.vag_ctx_set_stream(h, _lib.torch_stream_handle(torch.cuda.current_stream(dev))))
    out = torch.empty((nb,), dtype=torch.float64, device=dev)

    def begin(spec, d_theta):
        """One call as `world` simulated ranks on this GPU; rank 0's ticket names the call, the others' are abandoned (on their own
        GPUs each rank would hold one)."""
        blocks, tickets = [], []
        for rank in range(world):
            blk = torch.empty((per, 2), dtype=torch.float64, device=dev)
            tk = C.c_uint64(0)
            _lib.check(lib.vag_loglike_shard_begin_dev(h, C.byref(spec), d_theta.data_ptr(), nb, ndim, rank, world, blk.data_ptr(), C.byref(tk)))
            assert tk.value != 0 and tk.value not in tickets
            blocks.append(blk)
            tickets.append(tk.value)
        for tk in tickets[1:]:
            _lib.check(lib.vag_loglike_shard_end_dev(h, tk, None, nb, world, None))
        return tickets[0], torch.cat(blocks, 0).contiguous()

    def end(ticket, gathered):
        _lib.check(lib.vag_loglike_shard_end_dev(h, ticket, gathered.data_ptr(), nb, world, out.data_ptr()))
        return out.cpu().numpy()

    try:
        for _ in range(2):  # the second round is dealt by the costs the first one gathered: A's and B's tables then differ
            ta, ga = begin(sa, d1)
            tb_, gb = begin(sb, d1)
            assert np.array_equal(end(tb_, gb), want[("b", 1)])  # B first: the opposite order to the deals
            assert np.array_equal(end(ta, ga), want[("a", 1)])
        # the same fit twice in flight, finished in the opposite order
        t1, g1 = begin(sa, d1)
        t2, g2 = begin(sa, d2)
        assert np.array_equal(end(t2, g2), want[("a", 2)])
        assert np.array_equal(end(t1, g1), want[("a", 1)])
        # a ticket serves once; the shape is part of the call; ticket 0 names nothing
        assert lib.vag_loglike_shard_end_dev(h, t1, g1.data_ptr(), nb, world, out.data_ptr()) == _lib.VAG_E_INVALID
        t3, g3 = begin(sa, d1)
        assert lib.vag_loglike_shard_end_dev(h, t3, g3.data_ptr(), nb + 1, world, out.data_ptr()) == _lib.VAG_E_INVALID
        assert lib.vag_loglike_shard_end_dev(h, 0, g3.data_ptr(), nb, world, out.data_ptr()) == _lib.VAG_E_INVALID
        assert np.array_equal(end(t3, g3), want[("a", 1)])
        # nine calls in flight exceed the context's eight slots: refused, and the eight still finish
        held = []
        for i in range(8):
            tk = C.c_uint64(0)
            blk = torch.empty((per, 2), dtype=torch.float64, device=dev)
            _lib.check(lib.vag_loglike_shard_begin_dev(h, C.byref(sa), d1.data_ptr(), nb, ndim, 0, world, blk.data_ptr(), C.byref(tk)))
            held.append(tk.value)
        tk = C.c_uint64(0)
        assert lib.vag_loglike_shard_begin_dev(h, C.byref(sa), d1.data_ptr(), nb, ndim, 0, world, blk.data_ptr(), C.byref(tk)) == _lib.VAG_E_INVALID
        for tk in held:
            _lib.check(lib.vag_loglike_shard_end_dev(h, tk, None, nb, world, None))
        # the unticketed pair still serves (oldest deal of the shape; a second deal of the fit replaces the first)
        blocks = []
        for rank in range(world):
            blk = torch.empty((per, 2), dtype=torch.float64, device=dev)
            _lib.check(lib.vag_loglike_shard_dev(h, C.byref(sa), d2.data_ptr(), nb, ndim, rank, world, blk.data_ptr()))
            blocks.append(blk)
        _lib.check(lib.vag_loglike_shard_finish_dev(h, torch.cat(blocks, 0).contiguous().data_ptr(), nb, world, out.data_ptr()))
        assert np.array_equal(out.cpu().numpy(), want[("a", 2)])
        assert lib.vag_loglike_shard_finish_dev(h, g1.data_ptr(), nb, world, out.data_ptr()) == _lib.VAG_E_INVALID  # nothing left in flight
    finally:
        torch.cuda.synchronize()
        _lib.check(lib.vag_ctx_set_stream(h, None))


def test_context_buffers_are_ordered_across_a_change_of_stream(eng, oracle):
    """vag_ctx_set_stream chains the streams (event at the tail of the one it leaves, waited for by the one it takes): a
    likelihood call queued on a torch side stream through the shared context and a grid request right behind it on the
    context's own stream -- no host synchronisation in between -- both give the bits of the same calls made one at a time."""
    import torch
    import vegasafterglow_amd as va
    f, defs = _c4_fitter(oracle)
    _, lo, hi = f.build_spec(defs)
    samples = lo + (hi - lo) * np.random.default_rng(5).random((1024, len(defs)))
    want_ll = f.loglike_batch(samples, defs)
    m = va.Model(va.GaussianJet(0.1, 1e52, 300), va.ISM(1.0), va.Observer(1e28, 1.0, 0.25), va.Radiation(0.1, 0.01, 2.3))
    t, nu = np.logspace(3, 7, 40), np.array([1e9, 5e14, 1e18])
    want_grid = m.flux_density_grid(t, nu).total
    dev = torch.device("cuda", 0)
    ev = f.device_evaluator(defs)
    d_theta = torch.from_numpy(samples).to(dev)
    side = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    for _ in range(3):
        with torch.cuda.stream(side):
            ll, _ = ev(d_theta)                  # ~1.4 ms of kernels queued on the side stream; returns at once
        got_grid = m.flux_density_grid(t, nu).total  # same context, its own stream, queued while those still run
        side.synchronize()
        assert np.array_equal(ll.cpu().numpy(), want_ll)
        assert np.array_equal(got_grid, want_grid)


def _mixed_models(n):
    """n Models of mixed jets / media / radiation switches (what a pool of threads holds in the reference's samplers: one Model each)."""
    rng = np.random.default_rng(7)
    out = []
    for i in range(n):
        th_c, E, G0 = rng.uniform(0.05, 0.15), 10 ** rng.uniform(51.5, 52.5), rng.uniform(150, 400)
        obs = va.Observer(1e28, 1.0, rng.uniform(0.0, 0.3))
        kind = i % 4
        if kind == 0:
            m = va.Model(va.TophatJet(th_c, E, G0), va.ISM(1.0), obs, va.Radiation(0.1, 0.01, 2.3))
        elif kind == 1:
            m = va.Model(va.GaussianJet(th_c, E, G0), va.ISM(0.5), obs, va.Radiation(0.1, 0.01, 2.2))
        elif kind == 2:
            m = va.Model(va.PowerLawJet(th_c, E, G0, 2.0, 2.0), va.Wind(0.1), obs, va.Radiation(0.1, 0.01, 2.3, ssc=True))
        else:
            m = va.Model(va.GaussianJet(th_c, E, G0, duration=10.0), va.ISM(1.0), obs, va.Radiation(0.1, 0.01, 2.3),
                         rvs_rad=va.Radiation(0.1, 0.01, 2.3))
        out.append(m)
    return out


def _flux_parts(fd):
    return [np.array(a) for a in (fd.total, fd.fwd.sync, fd.fwd.ssc, fd.rvs.sync, fd.rvs.ssc)]


def test_models_driven_from_a_thread_pool_give_the_bits_of_their_single_threaded_calls(eng):
    """The reference's calling pattern (pybind/pybind.cpp:424-448 releases the GIL in every compute method; fitting/samplers.py:59-70
    maps eval_one over a ThreadPoolExecutor, one Model per thread): 8 threads x 16 Models of mixed jets call flux_density_grid,
    flux_density and flux concurrently on the one per-device context.  The library serialises the calls itself (every entry point locks
    the context); each result must be bit for bit what the same call returns single-threaded."""
    from concurrent.futures import ThreadPoolExecutor
    models = _mixed_models(128)
    t, nu = np.logspace(3, 7, 24), np.array([1e9, 4.84e14, 1e18])
    ts, nus = np.repeat(t, nu.size), np.tile(nu, t.size)

    def work(m):
        return (_flux_parts(m.flux_density_grid(t, nu)), _flux_parts(m.flux_density(ts, nus)), _flux_parts(m.flux(t, 1e17, 1e18, 5)))

    want = [work(m) for m in models]
    va.set_coalescing(False)
    with ThreadPoolExecutor(8) as ex:
        got = list(ex.map(work, models))
    for g, w in zip(got, want):
        for gm, wm in zip(g, w):
            for a, b in zip(gm, wm):
                assert a.shape == b.shape and np.array_equal(a, b)
    assert all(np.all(np.isfinite(w[0][0])) and w[0][0].max() > 0 for w in want)


def test_coalesced_thread_pool_calls_are_served_as_batches_with_each_callers_own_result(eng):
    """vag_*_coalesced (opt-in, va.set_coalescing): concurrent single-model calls with the same request run as one batch call.  32
    threads x 128 mixed Models: far fewer batch calls than requests; flux_density / flux are the bits of the single calls (their
    summation trees do not depend on the batch), flux_density_grid agrees to 1e-12 (its fixed-order sums are laid out per batch); a model
    the engine rejects raises for its own caller only, and callers with a different request are not mixed in."""
    from concurrent.futures import ThreadPoolExecutor
    models = _mixed_models(128)
    t, nu = np.logspace(3, 7, 24), np.array([1e9, 4.84e14, 1e18])
    t2 = np.logspace(3.5, 6.5, 10)  # a second request in the same pool
    ts, nus = np.repeat(t, nu.size), np.tile(nu, t.size)

    def work(im):
        i, m = im
        tt = t2 if i % 5 == 0 else t
        return (_flux_parts(m.flux_density_grid(tt, nu)), _flux_parts(m.flux_density(ts, nus)), _flux_parts(m.flux(tt, 1e17, 1e18, 5)))

    va.set_coalescing(False)
    want = [work(im) for im in enumerate(models)]
    calls0, batches0 = va.coalescing_stats()
    prev = va.set_coalescing(True, max_batch=64, wait_us=200)
    try:
        with ThreadPoolExecutor(32) as ex:
            got = list(ex.map(work, enumerate(models)))
        calls, batches = va.coalescing_stats()
        assert calls - calls0 == 3 * len(models)
        assert batches - batches0 < (calls - calls0) // 3  # batches did form
        for g, w in zip(got, want):
            for a, b in zip(g[0], w[0]):  # grid
                assert a.shape == b.shape
                if a.ndim:
                    assert np.allclose(a, b, rtol=1e-12, atol=0)
            for part in (1, 2):  # series, band: the same bits
                for a, b in zip(g[part], w[part]):
                    assert a.shape == b.shape and np.array_equal(a, b)
        # an over-capacity grid among valid ones: its caller gets the error, the others their fluxes
        bad = va.Model(va.GaussianJet(0.1, 1e52, 300), va.ISM(1.0), va.Observer(1e28, 1.0, 0.2), va.Radiation(0.1, 0.01, 2.3),
                       resolutions=(200.0, 200.0, 5.0))  # 72 000 phi nodes: beyond the grid kernel's third layout
        mix = models[:15] + [bad] + models[15:30]

        def guarded(m):
            try:
                return np.array(m.flux_density(ts, nus).total)
            except Exception as e:  # noqa: BLE001
                return e

        with ThreadPoolExecutor(31) as ex:
            res = list(ex.map(guarded, mix))
        assert isinstance(res[15], Exception)
        for r, im in zip(res[:15] + res[16:], list(range(15)) + list(range(15, 30))):
            assert np.array_equal(r, want[im][1][0])
    finally:
        va.set_coalescing(prev)


def test_a_long_mixed_loop_neither_grows_device_memory_nor_the_process(eng, oracle):
    """A sampler runs for hours on one context: grid, series and band requests of mixed models, likelihood calls of varying batch size,
    SSC tables (pooled), reverse shocks, a thread pool with coalescing -- 120 rounds of all of it.  The context's buffers are grow-only and
    must stop growing once the largest request has been seen: after the first rounds the library holds exactly the bytes it held
    (vag_device_bytes_in_use) and the process's resident set stays within 64 MB (numpy temporaries, allocator slack).  And a context
    that is destroyed gives ALL its device memory back: create / use / destroy cycles leave the account where it was, to the byte."""
    import psutil
    from concurrent.futures import ThreadPoolExecutor
    lib, h = eng
    proc = psutil.Process()
    models = _mixed_models(32)
    f, defs = _c4_fitter(oracle)
    rng = np.random.default_rng(3)
    lo = np.array([l for _, _, l, _ in configs.C4_FREE])
    hi = np.array([u for _, _, _, u in configs.C4_FREE])
    t, nu = np.logspace(3, 7, 24), np.array([1e9, 4.84e14, 1e18])
    ts, nus = np.repeat(t, nu.size), np.tile(nu, t.size)

    def one_round(r):
        for m in models[(r % 4)::4]:
            m.flux_density_grid(t, nu)
            m.flux_density(ts, nus)
            m.flux(t, 1e17, 1e18, 5)
        for nb in (1024, 37, 512):
            theta = lo + (hi - lo) * rng.random((nb, len(defs)))
            f.loglike_batch(theta, defs)
        va.set_coalescing(True, max_batch=64, wait_us=100)
        try:
            with ThreadPoolExecutor(16) as ex:
                list(ex.map(lambda m: m.flux_density(ts, nus).total, models))
        finally:
            va.set_coalescing(False)

    # the library's own account of its device memory (vag_device_bytes_in_use: exact, this process only) -- plus, loosely, the
    # process's resident set
    lib.vag_ctx_synchronize(h)
    for r in range(6):  # every request shape has been seen: buffers at their final size
        one_round(r)
    held0, rss0 = lib.vag_device_bytes_in_use(), proc.memory_info().rss / 2 ** 20
    assert held0 > 0
    for r in range(6, 120):
        one_round(r)
    assert lib.vag_device_bytes_in_use() == held0
    assert proc.memory_info().rss / 2 ** 20 - rss0 < 64.0
    # contexts give ALL their memory back: grid and series requests of four kinds of model on a context of its own (SSC + KN tables,
    # forward + reverse shock, a spreading jet, axisymmetric=False), then destroy
    prms = [_abi.make_params(jet="GaussianJet", theta_obs=0.2, ssc=True, kn=True),
            _abi.make_params(jet="PowerLawJet", medium="Wind", A_star=0.1, n_ism=0.0, theta_obs=0.3, duration=50.0, ssc=True, kn=True,
                             rvs=dict(eps_e=0.1, eps_B=0.01, p=2.3, ssc=True, kn=True)),
            _abi.make_params(jet="TophatJet", theta_obs=0.1, spreading=True),
            _abi.make_params(jet="GaussianJet", theta_obs=0.3, axisymmetric=False)]
    out = np.empty((1, nu.size, t.size))
    series = np.empty((1, ts.size))

    def cycle():
        hh = C.c_void_p()
        _lib.check(lib.vag_ctx_create(0, C.byref(hh)))
        for prm in prms:
            arr = (_lib.ModelParams * 1)(_lib.ModelParams.from_buffer_copy(bytes(prm)))
            _lib.check(lib.vag_flux_density_grid_batch(hh, arr, 1, t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp), nu.size,
                                                       out.ctypes.data_as(dp)))
            _lib.check(lib.vag_flux_density_batch(hh, arr, 1, ts.ctypes.data_as(dp), nus.ctypes.data_as(dp), ts.size,
                                                  series.ctypes.data_as(dp)))
            assert np.all(np.isfinite(out)) and out.max() > 0
        lib.vag_ctx_destroy(hh)
    before = lib.vag_device_bytes_in_use()
    for _ in range(5):
        cycle()
        assert lib.vag_device_bytes_in_use() == before
