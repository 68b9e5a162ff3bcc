"""DESIGN 4g's open question: what fraction of the headline workload's boundary evaluations B[l][k] enter only terms
below 2^-60 of the final flux of their (nu, t) bin?  Counted on the CPU checker (oracle/vag_oracle.c: g_flux_tally), which walks the same
rows, windows and intervals as the flux kernels, on members of bench.py's configs[1] batch.
usage: python3 profiles/debug/negligible_work.py [n_models]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _abi  # noqa: E402
import bench  # noqa: E402
import configs  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
lib = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
dp = C.POINTER(C.c_double)
t, nu = configs.C2_T, configs.C2_NU
arr = bench.c2_batch(512, seed=1234)
for bits in (60.0, 53.0, 40.0, 30.0):
    tot = np.zeros(4, dtype=np.int64)
    for i in np.linspace(0, 511, n).astype(int):
        counts = (C.c_longlong * 4)()
        rc = lib.vag_oracle_flux_tally(C.byref(arr[i]), t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp), nu.size, C.c_double(bits), counts)
        assert rc == 0
        tot += np.array(list(counts))
    print(f"threshold 2^-{bits:.0f} of the bin's final flux, {n} members of the configs[1] batch: boundary evaluations {tot[0]}, not needed {tot[1]} "
          f"({tot[1] / tot[0]:.3%}); interpolated terms {tot[2]}, negligible {tot[3]} ({tot[3] / tot[2]:.3%})", flush=True)

# ---- the SSC table build (vag_ic_photon_kernel's unit of work: (electron energy, seed bin) terms), members of the configs[2] / [4] batches
for name, prms in (("configs[2] (FS + RS, SSC + KN)", configs.c3_batch(2)), ("configs[4] (two-component SSC, Thomson)", configs.c5_batch(2))):
    t_e, nu_e = np.logspace(2, 8, 100), np.array([1e9, 4.84e14, 1e18, 2.4e26])
    for bits in (60.0, 40.0):
        tot = np.zeros(5, dtype=np.int64)
        for prm in prms[:2]:
            counts = (C.c_longlong * 5)()
            rc = lib.vag_oracle_ssc_tally(C.byref(prm), t_e.ctypes.data_as(dp), t_e.size, nu_e.ctypes.data_as(dp), nu_e.size, C.c_double(bits), counts)
            assert rc == 0, rc
            tot += np.array(list(counts))
        print(f"{name}, threshold 2^-{bits:.0f} of the smallest output node a term enters: tables {tot[0]}, (energy, bin) terms {tot[1]}, not needed "
              f"{tot[2]} ({tot[2] / tot[1]:.3%}); (table, energy) walks {tot[3]}, wholly not needed {tot[4]} ({tot[4] / tot[3]:.3%})", flush=True)
