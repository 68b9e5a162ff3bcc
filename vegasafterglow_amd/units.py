"""Unit multipliers into the engine's input conventions (seconds, Hz, cm, radians, erg/cm^2/s/Hz), AB magnitudes and the
named instrument bands -- the part of the reference's VegasAfterglow/units.py:36-88,172-205,339-367 that data preparation
for `Model.flux_density*` / `Fitter.add_*` calls needs.  Write `t = 3 * units.day`, `nu = 5 * units.keV`,
`f.add_flux(units.band("XRT"), ...)`.  The per-filter Vega / ST zero-point tables of the reference are not shipped.
"""
import math

import numpy as np

# time [s]
sec, ms, minute, hr, day = 1.0, 1e-3, 60.0, 3600.0, 86400.0
yr = 365.2425 * day
# frequency [Hz]; photon energies as frequencies E / h
Hz, kHz, MHz, GHz = 1.0, 1e3, 1e6, 1e9
_PLANCK_ERG_S = 6.62607015e-27
_EV_ERG = 1.602176634e-12
eV = _EV_ERG / _PLANCK_ERG_S
keV, MeV, GeV = 1e3 * eV, 1e6 * eV, 1e9 * eV
# flux density [erg / cm^2 / s / Hz]
Jy, mJy, uJy = 1e-23, 1e-26, 1e-29
# length [cm]
cm, m, km = 1.0, 1e2, 1e5
pc = 3.0856775814913673e18
kpc, Mpc, Gpc = 1e3 * pc, 1e6 * pc, 1e9 * pc
# angle [rad]
rad = 1.0
deg = math.pi / 180
arcmin, arcsec = deg / 60, deg / 3600
mas, uas = arcsec / 1e3, arcsec / 1e6

_AB_ZERO_CGS = 3631.0 * Jy


def ABmag_to_cgs(mag):
    """AB magnitude(s) -> flux density [erg/cm^2/s/Hz] (zero point 3631 Jy)."""
    return _AB_ZERO_CGS * np.power(10.0, -0.4 * np.asarray(mag, dtype=float))


def cgs_to_ABmag(f_nu):
    """Flux density [erg/cm^2/s/Hz] -> AB magnitude(s)."""
    return -2.5 * np.log10(np.asarray(f_nu, dtype=float) / _AB_ZERO_CGS)


# (nu_min, nu_max) [Hz] of instrument bands, for Model.flux / Fitter.add_flux
_BANDS = {
    "XRT": (0.3 * keV, 10 * keV), "BAT": (15 * keV, 150 * keV), "FXT": (0.3 * keV, 10 * keV), "WXT": (0.5 * keV, 4 * keV),
    "MXT": (0.2 * keV, 10 * keV), "ECLAIRs": (4 * keV, 150 * keV), "LAT": (100 * MeV, 300 * GeV), "GBM": (8 * keV, 40 * MeV),
}


def band(name):
    """(nu_min, nu_max) in Hz of a named instrument band (Swift XRT / BAT, EP FXT / WXT, SVOM MXT / ECLAIRs, Fermi LAT / GBM)."""
    if name not in _BANDS:
        raise ValueError(f"Unknown band '{name}'. Available: {', '.join(sorted(_BANDS))}")
    return _BANDS[name]
