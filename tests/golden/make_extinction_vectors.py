"""Generates tests/golden/extinction_pei92.npz by importing the reference's pure-Python extinction module in the dev
container (python tests/golden/make_extinction_vectors.py; needs /root/reference).  The fixture holds wavelengths and the
reference's k(lambda) for the three named laws -- data only."""
import importlib.util
import os

import numpy as np

spec = importlib.util.spec_from_file_location("ref_extinction", "/root/reference/VegasAfterglow/extinction.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

lam_cm = np.concatenate([np.geomspace(5e-6, 5e-3, 40), [5.5e-5, 9.12e-6, 2.175e-5]])
out = {"lam_cm": lam_cm}
for law in ("smc", "lmc", "mw"):
    out[law] = ref.pei92(lam_cm, law)
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "extinction_pei92.npz"), **out)
print({k: v.shape for k, v in out.items()})
