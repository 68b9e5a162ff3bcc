"""Developer probe: is a walker's ln L independent of its position in the batch?"""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _abi, configs
from test_gpu_fullsize import _c4_fitter
orc = _abi.load_oracle()
f, defs = _c4_fitter(orc)
_, lo, hi = f.build_spec(defs)
samples = lo + (hi - lo) * np.random.default_rng(4).random((96, len(defs)))
samples[5, 2] = -0.5
want = f.loglike_batch(samples, defs)
rng = np.random.default_rng(1)
for trial in range(3):
    perm = rng.permutation(96)
    got = f.loglike_batch(samples[perm], defs)
    d = got - want[perm]
    bad = np.where((d != 0) & np.isfinite(d))[0]
    print("trial", trial, "differing walkers", len(bad), [(int(perm[i]), float(d[i] / abs(want[perm][i]))) for i in bad[:6]])
rev = f.loglike_batch(samples[::-1].copy(), defs)
print("reversed: differing", int(np.sum((rev != want[::-1]) & np.isfinite(rev))))
for k in (1, 2, 48):
    part = f.loglike_batch(samples[:k], defs)
    print("first", k, "differing", int(np.sum((part != want[:k]) & np.isfinite(part))))
