"""Property sweep of the lane-refill ODE kernel without a checker: wide random batches of every forward-shock kind the fast solver serves
(all six jets, ISM and wind media of every kind the solver accepts, SSC on / off, adiabatic / radiative, narrow and wide time ranges, an
invalid model now and then) through VAG_DYN_REFILL=0 (one wavefront per 64 rows) and =1 (persistent wavefronts, ordered row queue, few
wavefronts so that every lane refills many times): the fluxes must be the same bits, the solver's tallies equal.
usage: python3 profiles/debug/refill_property_sweep.py [n_seeds] [models_per_batch]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import _abi  # noqa: E402
import vegasafterglow_amd as va  # noqa: E402
from vegasafterglow_amd import _lib  # noqa: E402

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 384
lib = _lib.load()
h, _ = va.get_context(0)
dev = torch.device("cuda", 0)
JETS = ["TophatJet", "GaussianJet", "PowerLawJet", "TwoComponentJet", "StepPowerLawJet", "PowerLawWing"]
worst_util = []
for seed in range(9000, 9000 + n_seeds):
    rng = np.random.default_rng(seed)
    ssc = bool(seed % 2)
    prms = []
    for i in range(nb):
        jet = JETS[int(rng.integers(6))]
        kw = dict(jet=jet, E_iso=10 ** rng.uniform(49.5, 54.5), Gamma0=10 ** rng.uniform(0.3, 3.0), theta_c=rng.uniform(0.02, 0.5),
                  theta_obs=rng.uniform(0, 1.0), p=rng.uniform(1.6, 2.9), eps_e=10 ** rng.uniform(-3, -0.3), eps_B=10 ** rng.uniform(-6, -0.5),
                  radiative_fireball=bool(rng.integers(2)), z=10 ** rng.uniform(-2, 0.7), ssc=ssc, kn=ssc and bool(seed % 4 == 1))
        kw["lumi_dist"] = 10 ** rng.uniform(26, 28.5)
        if rng.random() < 0.4:
            kw.update(medium="Wind", A_star=10 ** rng.uniform(-3, 1), n_ism=0.0 if rng.random() < 0.5 else 10 ** rng.uniform(-4, 0))
            if rng.random() < 0.3:
                kw.update(n0=10 ** rng.uniform(0, 4))
        else:
            kw.update(n_ism=10 ** rng.uniform(-5, 3))
        if jet in ("TwoComponentJet", "StepPowerLawJet", "PowerLawWing"):
            kw.update(theta_w=min(1.5, kw["theta_c"] * rng.uniform(1.5, 4.0)), E_iso_w=kw["E_iso"] * 10 ** rng.uniform(-3, -0.3),
                      Gamma0_w=max(1.5, kw["Gamma0"] * rng.uniform(0.05, 0.6)))
        if jet in ("PowerLawJet", "StepPowerLawJet", "PowerLawWing"):
            kw.update(k_e=rng.uniform(1.0, 4.0), k_g=rng.uniform(1.0, 4.0))
        prms.append(_abi.make_params(**kw))
    for i in rng.integers(0, nb, 3):
        prms[int(i)].eps_e = 3.0  # invalid
    lo = rng.uniform(0.5, 4.0)
    t, nu = np.logspace(lo, lo + rng.uniform(1.0, 6.0), 24), np.array([1e9, 4.84e14, 1e18] + ([1e24] if ssc else []))
    arr = (_lib.ModelParams * nb)(*[_lib.ModelParams.from_buffer_copy(bytes(p)) for p in prms])
    d_p = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    d_t, d_nu = torch.from_numpy(t).to(dev), torch.from_numpy(nu).to(dev)

    def run(refill):
        _lib.hooks["VAG_DYN_REFILL"] = refill
        if refill == "1":
            _lib.hooks["VAG_DYN_REFILL_WGS"] = str(int(rng.integers(3, 200)))
            _lib.hooks["VAG_DYN_REFILL_MIN"] = str(int(rng.choice([1, 2, 8, 16, 64])))
        try:
            _lib.check(lib.vag_ctx_count_work(h, 1))
            d_o = torch.full((nb, nu.size, t.size), -1.0, dtype=torch.float64, device=dev)
            _lib.check(lib.vag_flux_density_grid_batch_dev(h, d_p.data_ptr(), nb, d_t.data_ptr(), t.size, d_nu.data_ptr(), nu.size, d_o.data_ptr()))
            _lib.check(lib.vag_ctx_synchronize(h))
            plan = _lib.Plan()
            _lib.check(lib.vag_last_plan(h, C.byref(plan)))
        finally:
            _lib.check(lib.vag_ctx_count_work(h, 0))
            for k in ("VAG_DYN_REFILL", "VAG_DYN_REFILL_WGS", "VAG_DYN_REFILL_MIN"):
                _lib.hooks.pop(k, None)
        return d_o.cpu().numpy(), plan

    a, pa = run("0")
    b, pb = run("1")
    same = np.array_equal(a, b, equal_nan=True)
    tall = (pa.n_rows, pa.ode_rhs, pa.n_rows_failed, pa.n_rows_gave_up, pa.ode_lane_attempts) == (pb.n_rows, pb.ode_rhs, pb.n_rows_failed, pb.n_rows_gave_up,
                                                                                                 pb.ode_lane_attempts)
    fin = int(np.isfinite(a).all(axis=(1, 2)).sum())
    print(f"seed {seed}: {nb} models ({'ssc' if ssc else 'syn'}), rows {pa.n_rows}, finite models {fin}, rhs {pa.ode_rhs}, rows failed / gave up {pa.n_rows_failed} / "
          f"{pa.n_rows_gave_up}; lane utilisation plain {pa.ode_lane_attempts / max(pa.ode_lane_slots, 1):.3f} refill {pb.ode_lane_attempts / max(pb.ode_lane_slots, 1):.3f}; "
          f"same bits {same}, same tallies {tall}", flush=True)
    assert same and tall
print("all seeds: same bits")
