"""The REAL reference (oracle/_ref/libvag_ref.so, the reference's own flags, and the strict build) on models whose adaptive grids are
larger than the grid kernel's LDS layouts hold -- more than 2000 theta nodes, more than 10 000 lattice times, both at once, more than
2560 phi nodes (tests/configs.py BIG_GRID_CASES).  The reference sizes its grids freely (grid-refinement.h:639-706); the engine's third
layout keeps the grid kernel's scratch arrays in HBM for them.  Dev container only (needs /root/reference).

    python tests/golden/make_big_grid_fixture.py    ->  tests/golden/reference_big_grids.npz
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
import _abi  # noqa: E402
import configs  # noqa: E402


def main():
    fast = _abi.load_ref()
    strict = _abi.CpuLib(os.path.join(ROOT, "oracle", "_ref", "libvag_ref_strict.so"), "vag_ref")
    t, nu = configs.BIG_GRID_T, configs.BIG_GRID_NU
    out = {"t": t, "nu": nu}
    for name, kw in configs.BIG_GRID_CASES.items():
        prm = _abi.make_params(**kw)
        t0 = time.time()
        out[name + "_fast"] = fast.flux_density_grid(prm, t, nu)
        out[name + "_strict"] = strict.flux_density_grid(prm, t, nu)
        shape = strict.grid_shape(prm, t.min(), t.max())
        out[name + "_shape"] = np.array(shape)  # (n_phi, n_theta, n_t, n_reps, symmetry, phi_mirrored)
        print(name, shape, f"{time.time() - t0:.1f} s; builds differ by",
              float(np.max(np.abs(out[name + '_fast'] - out[name + '_strict']) / out[name + '_strict'])), flush=True)
    np.savez_compressed(os.path.join(HERE, "reference_big_grids.npz"), **out)


if __name__ == "__main__":
    main()
