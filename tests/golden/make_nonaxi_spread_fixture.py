"""Golden vectors for Model(axisymmetric=False) with a SPREADING jet from the REAL reference (oracle/_ref/libvag_ref.so): one time
lattice and one blast-wave solve per (phi, theta) node (grid-refinement.h:619-625, observer.cpp:51-141).  The C restatement under
oracle/ does not cover this combination, so the GPU path is checked against these vectors.  Dev container only.

    python tests/golden/make_nonaxi_spread_fixture.py   ->  tests/golden/reference_nonaxi_spread.npz
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _abi  # noqa: E402

CASES = {
    "gauss_offaxis": dict(jet="GaussianJet", theta_obs=0.2, spreading=True, axisymmetric=False),
    "tophat_offaxis": dict(jet="TophatJet", theta_obs=0.15, spreading=True, axisymmetric=False),
    "gauss_onaxis": dict(jet="GaussianJet", theta_obs=0.0, spreading=True, axisymmetric=False),
    "powerlaw_wind_ssc": dict(jet="PowerLawJet", medium="Wind", A_star=0.1, n_ism=0.0, theta_obs=0.25, spreading=True, axisymmetric=False,
                              ssc=True, resolutions=(0.1, 0.3, 8.0)),
    "tophat_rs": dict(jet="TophatJet", theta_obs=0.1, duration=100.0, spreading=True, axisymmetric=False,
                      rvs=dict(eps_e=0.1, eps_B=0.01, p=2.3)),
    "gauss_rs_ssc": dict(jet="GaussianJet", theta_obs=0.25, duration=30.0, spreading=True, axisymmetric=False, ssc=True,
                         rvs=dict(eps_e=0.1, eps_B=0.01, p=2.3, ssc=True)),
    "two_component_fine": dict(jet="TwoComponentJet", theta_c=0.05, theta_w=0.3, theta_obs=0.1, spreading=True, axisymmetric=False,
                               resolutions=(0.15, 0.3, 10.0)),
}
T = np.logspace(3, 7.5, 24)
NU = np.array([1e9, 4.84e14, 1e18])


def main():
    ref = _abi.load_ref()
    if ref is None:
        raise SystemExit("oracle/_ref/libvag_ref.so missing: run `make -C oracle ref` in the dev container")
    out, meta = {"t": T, "nu": NU}, {}
    for name, kw in CASES.items():
        prm = _abi.make_params(**kw)
        sync, ssc, rsync, rssc = ref.flux_components4(prm, T, NU)
        out[f"{name}__sync"], out[f"{name}__ssc"], out[f"{name}__rvs_sync"], out[f"{name}__rvs_ssc"] = sync, ssc, rsync, rssc
        ts, nus = np.repeat(T, 3), np.tile(NU, T.size)
        out[f"{name}__series"] = ref.flux_density(prm, ts, nus)
        out[f"{name}__band"] = ref.flux(prm, T, 1e14, 1e15, 8)
        d = ref.details(prm, T.min(), T.max())
        meta[name] = dict(kw=json.loads(json.dumps(kw, default=list)), shape=d["shape"])
        print(name, d["shape"], "peak", sync.max(), rsync.max())
    out["meta"] = json.dumps(meta)
    path = os.path.join(HERE, "reference_nonaxi_spread.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
