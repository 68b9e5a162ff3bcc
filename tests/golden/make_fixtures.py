"""Generate golden vectors from the REAL reference (oracle/_ref/libvag_ref.so, the reference's own C++ sources
compiled in place with the reference's flags) for the BASELINE configs.  Run in the dev container only
(needs /root/reference); the resulting small .npz files are committed and travel to the GPU box.

    python tests/golden/make_fixtures.py

The three *_ism.npz files next to this script are the reference's own golden baselines
(tests/python/golden/*.npz of the reference tree: data files held by its test-suite), copied verbatim.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _abi  # noqa: E402
import configs  # noqa: E402


def main_rs(ref):
    """Reverse-shock tier: all four FluxDict components of C3 and five RS cases, a (t, nu) series and
    a band integral, plus the reverse shock's own arrays (Model.details().rvs) for one case."""
    out, meta = {}, {}
    cases = {"C3": (configs.C3, configs.C3_T, configs.C3_NU)}
    cases.update(configs.RS_CASES)
    for name, (kw, t, nu) in cases.items():
        prm = _abi.make_params(**kw)
        comps = ref.flux_components4(prm, t, nu)
        out[f"{name}__t"], out[f"{name}__nu"] = t, nu
        for cname, a in zip(("fwd_sync", "fwd_ssc", "rvs_sync", "rvs_ssc"), comps):
            out[f"{name}__{cname}"] = a
        out[f"{name}__total"] = ref.flux_density_grid(prm, t, nu)
        meta[name] = json.loads(json.dumps(kw, default=list))
    for name in ("tophat_spread_offaxis", "gauss_spread"):  # spreading jets
        prm = _abi.make_params(**configs.SPREAD_CASES[name])
        out[f"{name}__total"] = ref.flux_density_grid(prm, configs.SPREAD_T, configs.SPREAD_NU)
    kw, t, nu = configs.RS_CASES["rs_thick_offaxis"]
    prm = _abi.make_params(**kw)
    ts, nus = np.repeat(t, 2), np.tile(nu[[0, 2]], t.size)
    out["series__t"], out["series__nu"] = ts, nus
    out["series__flux"] = ref.flux_density(prm, ts, nus)
    out["band__flux"] = ref.flux(prm, t, 1e17, 1e19, 9)
    d = ref.details(prm, t.min(), t.max(), rvs=True)
    meta["rs_thick_offaxis__shape"] = d["shape"]
    for k in ("t_src", "Gamma", "r", "B", "N_p", "Gamma_th", "gamma_c", "gamma_M", "injection_idx"):
        out[f"rs_thick_offaxis__rvs_{k}"] = np.asarray(d[k])
    out["meta"] = json.dumps(meta)
    path = os.path.join(HERE, "reference_vectors_rs.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def main():
    ref = _abi.load_ref()
    if ref is None:
        raise SystemExit("oracle/_ref/libvag_ref.so missing: run `make -C oracle ref` in the dev container")
    out = {}
    meta = {}
    cases = {"C1a": (configs.C1A, configs.C1_T, configs.C1_NU), "C1b": (configs.C1B, configs.C1_T, configs.C1_NU),
             "C2": (configs.C2, configs.C2_T, configs.C2_NU)}
    cases.update(configs.EXTRA)
    for name, (kw, t, nu) in cases.items():
        prm = _abi.make_params(**kw)
        out[f"{name}__t"] = t
        out[f"{name}__nu"] = nu
        out[f"{name}__grid"] = ref.flux_density_grid(prm, t, nu)
        meta[name] = {k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()}
    # C4: series + band + grid shape for the truth model
    prm = _abi.make_params(**configs.C4_TRUTH)
    t, nu = configs.c4_mock_data()
    out["C4__t"], out["C4__nu"] = t, nu
    out["C4__series"] = ref.flux_density(prm, t, nu)
    out["C4__band_t"] = configs.C4_EPOCHS
    out["C4__band"] = ref.flux(prm, configs.C4_EPOCHS, 1e14, 1e15, 16)
    meta["C4"] = {k: v for k, v in configs.C4_TRUTH.items()}
    # grid shapes + stage intermediates (Model.details) for two configs
    for name, kw, tmin, tmax in (("C1b", configs.C1B, 1e2, 1e8), ("C4", configs.C4_TRUTH, t.min(), t.max())):
        d = ref.details(_abi.make_params(**kw), tmin, tmax)
        meta[name + "__shape"] = d["shape"]
        for k in ("phi", "theta", "t_src", "Gamma", "r", "B", "N_p", "Gamma_th", "nu_m", "nu_c", "nu_a", "I_nu_max"):
            out[f"{name}__details_{k}"] = np.asarray(d[k])
    # SSC tier: components of a C5-like two-component SSC model and a KN Gaussian model
    import numpy as _np
    ssc_cases = {
        "C5_central": (dict(jet="TwoComponentJet", theta_c=0.065, E_iso=1e52, Gamma0=300.0, theta_w=0.35, E_iso_w=1e50,
                            Gamma0_w=60.0, medium="ISM", n_ism=1.0, lumi_dist=1e28, z=1.0, theta_obs=0.15, eps_e=0.1,
                            eps_B=0.01, p=2.3, ssc=True, kn=False, resolutions=(0.59, 0.98, 12.0)),
                       _np.logspace(2, 8, 100), _np.array([1e9, 4.84e14, 1e18, 2.4e26])),
        "ssc_kn_gaussian": (dict(jet="GaussianJet", theta_obs=0.2, eps_B=1e-4, ssc=True, kn=True),
                            _np.logspace(2, 8, 40), _np.array([1e9, 1e14, 1e18, 1e22, 2.4e26])),
    }
    for name, (kw, t, nu) in ssc_cases.items():
        prm = _abi.make_params(**kw)
        sync, ssc = ref.flux_components(prm, t, nu)
        out[f"{name}__t"], out[f"{name}__nu"] = t, nu
        out[f"{name}__sync"], out[f"{name}__ssc"] = sync, ssc
        meta[name] = {k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()}
    out["meta"] = json.dumps(meta)
    path = os.path.join(HERE, "reference_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")
    main_rs(ref)


if __name__ == "__main__":
    main()
