"""Host-galaxy dust laws k(lambda) = A_lambda / A_V for Fitter(extinction="smc" | "lmc" | "mw").

Mirror of the reference's VegasAfterglow/extinction.py:17-102: the Pei (1992, ApJ 395, 130) Table 4 six-term fits
(background, far-UV, 2175 A bump and the 9.7 / 18 / 25 um features), each term a / ((l/l_i)^n + (l_i/l)^n + b), scaled by
(1 + R_V) / R_V, divided by its own value at 5500 A so that k(V) == 1, and set to zero short of the Lyman limit (912 A)
where the fit has no meaning.  Host-side data preparation: the fitter multiplies the result into a per-datum kernel once.
"""
import numpy as np

_V_BAND_CM = 5.5e-5
_LYMAN_LIMIT_CM = 9.12e-6

# rows: (a_i, lambda_i [um], b_i, n_i), Pei (1992) Table 4
_TERMS = {
    "smc": (2.93, ((185.0, 0.042, 90.0, 2.0), (27.0, 0.08, 5.50, 4.0), (0.005, 0.22, -1.95, 2.0),
                   (0.010, 9.7, -1.95, 2.0), (0.012, 18.0, -1.80, 2.0), (0.030, 25.0, 0.0, 2.0))),
    "lmc": (3.16, ((175.0, 0.046, 90.0, 2.0), (19.0, 0.08, 5.50, 4.5), (0.023, 0.22, -1.95, 2.0),
                   (0.005, 9.7, -1.95, 2.0), (0.006, 18.0, -1.80, 2.0), (0.020, 25.0, 0.0, 2.0))),
    "mw": (3.08, ((165.0, 0.047, 90.0, 2.0), (14.0, 0.08, 4.0, 6.5), (0.045, 0.22, -1.95, 2.0),
                  (0.002, 9.7, -1.95, 2.0), (0.002, 18.0, -1.80, 2.0), (0.012, 25.0, 0.0, 2.0))),
}


def _unnormalised(lam_cm, profile):
    r_v, terms = _TERMS[profile]
    a, lam_i, b, n = (np.array(col) for col in zip(*terms))
    lam = np.asarray(lam_cm, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        x = (lam * 1e4)[..., None] / lam_i
        total = (a / (x ** n + (1.0 / x) ** n + b)).sum(axis=-1)
    return np.where(lam < _LYMAN_LIMIT_CM, 0.0, ((r_v + 1.0) / r_v) * total)


_AT_V = {name: float(_unnormalised(_V_BAND_CM, name)) for name in _TERMS}


def pei92(lam_cm, profile="smc"):
    """k(lambda_rest [cm]) of the named Pei92 profile; same shape as the input, exactly 1 at 5500 A, 0 below 912 A."""
    if profile not in _TERMS:
        raise ValueError(f"Unknown extinction law: {profile!r}. Expected one of {sorted(_TERMS)}")
    return _unnormalised(lam_cm, profile) / _AT_V[profile]


def smc(lam_cm):
    return pei92(lam_cm, "smc")


def lmc(lam_cm):
    return pei92(lam_cm, "lmc")


def mw(lam_cm):
    return pei92(lam_cm, "mw")


BUILTIN_LAWS = {"smc": smc, "lmc": lmc, "mw": mw}
