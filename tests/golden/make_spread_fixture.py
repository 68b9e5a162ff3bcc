"""Both builds of the REAL reference (oracle/_ref/libvag_ref.so: the reference's own flags, -O3 -ffp-contract=fast; and
libvag_ref_strict.so: -O2 -ffp-contract=off) on the ensemble members the full-size GPU tests sample: the members of the configs[4]
draw (tests/configs.py c5_batch) that tests/test_gpu_fullsize.py checks, and 16 members of the jittered configs[2] batch
(c3_batch(128)) the bench times.  The spread between the two builds is the reference's own compile-flag sensitivity on that input.
Its conditioning is measured as well: the strict build is re-run with theta_obs and Gamma0 moved by ONE ulp either way (`*_ulp`: the
largest relative change of the fluxes) -- on member 192 of the configs[4] draw, a two-component jet whose adaptive theta grid sits on the
core edge, one ulp of Gamma0 moves the reference's light curve by 1.1e-4.  The tests hold every member to max(2e-6, the sensitivity the
reference itself demonstrates on that member) instead of a blanket allowance.  Dev container only (needs /root/reference).

    python tests/golden/make_spread_fixture.py    ->  tests/golden/reference_spread.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
import _abi  # noqa: E402
from configs import c3_batch, c5_batch  # noqa: E402

C5_MEMBERS = [3, 77, 192, 200, 311, 480, 4000, 4095]
C3_MEMBERS = list(range(0, 128, 8))


def main():
    fast = _abi.load_ref()
    strict = _abi.CpuLib(os.path.join(ROOT, "oracle", "_ref", "libvag_ref_strict.so"), "vag_ref")
    t, nu = np.logspace(2, 8, 100), np.array([1e9, 4.84e14, 1e18, 2.4e26])
    out = {"t": t, "nu": nu, "c5_members": np.array(C5_MEMBERS), "c3_members": np.array(C3_MEMBERS)}
    c5 = c5_batch(4096)
    out["c5_fast"] = np.stack([fast.flux_density_grid(c5[i], t, nu) for i in C5_MEMBERS])
    out["c5_strict"] = np.stack([strict.flux_density_grid(c5[i], t, nu) for i in C5_MEMBERS])
    c3 = c3_batch(128)
    out["c3_fast"] = np.stack([np.stack(fast.flux_components4(c3[i], t, nu)) for i in C3_MEMBERS])      # [16][4 components][nu][t]
    out["c3_strict"] = np.stack([np.stack(strict.flux_components4(c3[i], t, nu)) for i in C3_MEMBERS])
    def rel(a, b, axes):
        peak = b.max(axis=axes, keepdims=True)
        m = b > 1e-9 * np.maximum(peak, 1e-300)
        return np.where(m, np.abs(a - b) / np.where(m, b, 1.0), 0.0).max(axis=axes)

    def one_ulp(prm, field, up):
        q = _abi.ModelParams.from_buffer_copy(bytes(prm))
        v = getattr(q, field)
        setattr(q, field, float(np.nextafter(v, np.inf if up else -np.inf)))
        return q

    c5_ulp = np.zeros(len(C5_MEMBERS))
    for n, i in enumerate(C5_MEMBERS):
        for field in ("theta_obs", "Gamma0"):
            for up in (True, False):
                o = strict.flux_density_grid(one_ulp(c5[i], field, up), t, nu)
                c5_ulp[n] = max(c5_ulp[n], float(rel(o, out["c5_strict"][n], (0, 1))))
    c3_ulp = np.zeros((len(C3_MEMBERS), 4))
    for n, i in enumerate(C3_MEMBERS):
        for field in ("theta_obs", "Gamma0"):
            for up in (True, False):
                o = np.stack(strict.flux_components4(one_ulp(c3[i], field, up), t, nu))
                c3_ulp[n] = np.maximum(c3_ulp[n], rel(o, out["c3_strict"][n], (1, 2)))
    out["c5_ulp"], out["c3_ulp"] = c5_ulp, c3_ulp
    print("c5: build spread", np.array2string(rel(out["c5_fast"], out["c5_strict"], (1, 2)), precision=2, max_line_width=200))
    print("c5: one-ulp sensitivity", np.array2string(c5_ulp, precision=2, max_line_width=200))
    print("c3: build spread (max over members)", np.array2string(rel(out["c3_fast"], out["c3_strict"], (2, 3)).max(axis=0), precision=2))
    print("c3: one-ulp sensitivity (max over members)", np.array2string(c3_ulp.max(axis=0), precision=2))
    path = os.path.join(HERE, "reference_spread.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
