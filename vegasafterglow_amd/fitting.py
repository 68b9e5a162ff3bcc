"""Batched log-likelihood: host-side mirror of the reference's Fitter evaluation path.

Mirrors ``VegasAfterglow/fitting/fitter.py:407-451`` (_consolidate_data), ``:497-533`` (_chi2_sum,
_evaluate), ``fitting/utils.py:110-135`` (transformer), ``fitting/samplers.py:61-91`` (eval_one,
log_prob_batch) for point flux-density data, with the per-walker ``Model`` evaluations replaced by ONE
batched call into the HIP engine (``vag_loglike_batch``).  Sampler drivers (emcee/bilby) stay
third-party: hand ``Fitter.log_prob_batch`` to ``emcee.EnsembleSampler(..., vectorize=True)``.
"""
import ctypes as C
import logging
import math
from dataclasses import dataclass
from enum import Enum
from typing import Optional, Sequence

import numpy as np

from . import _lib
from .model import get_context

logger = logging.getLogger(__name__)
logger.addHandler(logging.NullHandler())

# the registry of VegasAfterglow/fitting/config.py:99-121 ("uniform" = TophatJet with theta_c fixed to pi/2)
JET_TYPES = {"tophat": _lib.JET_TOPHAT, "gaussian": _lib.JET_GAUSSIAN, "powerlaw": _lib.JET_POWERLAW,
             "two_component": _lib.JET_TWO_COMPONENT, "step_powerlaw": _lib.JET_STEP_POWERLAW,
             "powerlaw_wing": _lib.JET_POWERLAW_WING, "uniform": _lib.JET_TOPHAT}
MEDIUM_TYPES = {"ism": _lib.MEDIUM_ISM, "wind": _lib.MEDIUM_WIND}


def logscale_screen(data, data_density):
    """Indices that thin a sorted positive array to about `data_density` points per decade, end points always kept
    (pybind.h:40-107): the targets are log-uniform between data[0] and data[-1], each replaced by the interior sample
    nearest to it in LINEAR distance (first one on ties), duplicates dropped.  Host-side data preparation for a fit."""
    x = np.asarray(data, dtype=np.float64).ravel()
    n, density = x.size, int(data_density)
    if n <= 1:
        return list(range(n))
    if density == 0:
        return list(range(n))
    lo, hi = np.log10(x[0]), np.log10(x[-1])
    n_targets = int(np.ceil((hi - lo) * density)) + 1
    keep = {0, n - 1}
    if n_targets > 2 and n > 2:
        step = (hi - lo) / (n_targets - 1)
        interior = x[1:n - 1]
        for i in range(1, n_targets - 1):
            keep.add(1 + int(np.argmin(np.abs(interior - 10.0 ** (lo + i * step)))))
    elif n_targets > 2:
        keep.add(1)
    return sorted(keep)


class Scale(Enum):
    linear = "linear"
    log = "log"
    fixed = "fixed"


@dataclass
class ParamDef:
    """ParamDef(name, lower, upper, scale, initial) -- VegasAfterglow/types.py."""
    name: str
    lower: float
    upper: float
    scale: Scale = Scale.linear
    initial: Optional[float] = None


# ModelParams defaults, VegasAfterglow/types.py:37-77
MODEL_PARAM_DEFAULTS = dict(theta_v=0.0, n_ism=0.0, n0=math.inf, A_star=0.0, k_m=2.0, E_iso=1e52, Gamma0=300.0,
                            theta_c=0.1, k_e=2.0, k_g=2.0, tau=1.0, E_iso_w=1e52, Gamma0_w=300.0,
                            theta_w=math.pi / 2, p=2.3, eps_e=0.1, eps_B=0.01, xi_e=1.0,
                            p_r=2.3, eps_e_r=0.1, eps_B_r=0.01, xi_e_r=1.0, L0=0.0, t0=1.0, q=2.0)

_dp = C.POINTER(C.c_double)


def _device_prior(prior):
    """(VAG_PRIOR_* kind, a, b) of a prior the device evaluates itself; PRIOR_NONE hands an unknown object's ln_prob to the host.
    Recognised by shape, not by import: bilby.core.prior.Uniform(minimum, maximum) (its own support: -ln(maximum - minimum)
    inside, -inf outside), Gaussian(mu, sigma), LogUniform(minimum, maximum), or tuples ("uniform",) = the ParamDef's own
    Uniform(lower, upper), ("uniform", lo, hi), ("gaussian", mu, sigma), ("log_uniform", lo, hi)."""
    if prior is None:
        return _lib.PRIOR_UNIFORM, 0.0, 0.0
    if isinstance(prior, (tuple, list)):
        name = str(prior[0]).lower()
        if name == "uniform":
            if len(prior) >= 3:
                return _lib.PRIOR_UNIFORM_RANGE, float(prior[1]), float(prior[2])
            return _lib.PRIOR_UNIFORM, 0.0, 0.0
        if name in ("gaussian", "normal"):
            return _lib.PRIOR_GAUSSIAN, float(prior[1]), float(prior[2])
        if name in ("log_uniform", "loguniform"):
            return _lib.PRIOR_LOG_UNIFORM, float(prior[1]), float(prior[2])
        raise ValueError(f"unknown prior {prior!r}")
    cls = type(prior).__name__
    if cls in ("Gaussian", "Normal") and hasattr(prior, "mu") and hasattr(prior, "sigma"):
        return _lib.PRIOR_GAUSSIAN, float(prior.mu), float(prior.sigma)
    if cls == "LogUniform" and hasattr(prior, "minimum") and hasattr(prior, "maximum"):
        return _lib.PRIOR_LOG_UNIFORM, float(prior.minimum), float(prior.maximum)
    if cls == "Uniform" and hasattr(prior, "minimum") and hasattr(prior, "maximum"):
        return _lib.PRIOR_UNIFORM_RANGE, float(prior.minimum), float(prior.maximum)
    if not hasattr(prior, "ln_prob"):
        raise ValueError(f"prior {prior!r} has no ln_prob")
    return _lib.PRIOR_NONE, 0.0, 0.0


def host_ln_prior(samples, lower, upper, prior_specs):
    """sum_d ln prior_d(samples[:, d]) with the bounds mask of fitting/samplers.py:72-91 on the host: -inf outside
    [lower, upper], else the closed forms of bilby's Uniform / Gaussian / LogUniform (the same expressions
    vag_fit_front_kernel evaluates) or the prior object's own ln_prob.  prior_specs[d] = (kind, a, b, obj)."""
    samples = np.atleast_2d(np.asarray(samples, dtype=np.float64))
    lp = np.zeros(samples.shape[0])
    with np.errstate(divide="ignore", invalid="ignore"):
        for d, (kind, a, b, obj) in enumerate(prior_specs):
            x = samples[:, d]
            if kind == _lib.PRIOR_UNIFORM:
                lp += -np.log(upper[d] - lower[d])
            elif kind == _lib.PRIOR_UNIFORM_RANGE:
                lp += np.where((x >= a) & (x <= b), -np.log(b - a), -np.inf)
            elif kind == _lib.PRIOR_GAUSSIAN:
                lp += -0.5 * ((x - a) / b) ** 2 - np.log(b * 2.5066282746310002)
            elif kind == _lib.PRIOR_LOG_UNIFORM:
                lp += np.where((x >= a) & (x <= b), -np.log(x * np.log(b / a)), -np.inf)
            else:
                lp += np.asarray(obj.ln_prob(x), dtype=np.float64)
    inside = np.all((samples >= lower) & (samples <= upper), axis=1)
    lp[~inside] = -np.inf
    lp[~np.isfinite(lp)] = -np.inf
    return lp


class Fitter:
    """Fitter(*, z=0.0, lumi_dist=1e26, jet=..., medium=..., resolution=..., rtol=...): keyword-only with the reference's
    defaults (fitter.py:96-135)."""

    def __init__(self, *, z=0.0, lumi_dist=1e26, jet="tophat", medium="ism", resolution=None, rtol=1e-6,
                 radiative_fireball=True, device=0, fwd_ssc=False, kn=False, rvs_shock=False, rvs_ssc=False,
                 magnetar=False, extinction=None):
        # extinction: k(lambda_rest [cm]) -> A_lambda / A_V of the host-galaxy law (a callable; fitter.py:379-397).  The
        # point-data model fluxes are scaled by exp(-A_V * 0.4 ln10 * k) with A_V a (free or fixed) parameter.
        # A name selects a built-in Pei92 law; a custom callable is evaluated ONCE per data set (it must not depend on the
        # sampled parameters: the device likelihood applies one fixed kernel per datum).
        if isinstance(extinction, str):
            from .extinction import BUILTIN_LAWS
            if extinction not in BUILTIN_LAWS:
                raise ValueError(f"Unknown extinction law: {extinction!r}. Expected one of {sorted(BUILTIN_LAWS)} or a callable.")
            self.extinction_name, extinction = extinction, BUILTIN_LAWS[extinction]
        elif extinction is not None and not callable(extinction):
            raise ValueError("extinction must be None, 'smc' / 'lmc' / 'mw', or a callable k(lambda_rest_cm)")
        self.extinction = extinction
        self.magnetar = bool(magnetar)
        if rvs_ssc and not rvs_shock:
            rvs_ssc = False  # the reference only builds rvs_rad when rvs_shock is on (fitter.py:476-484)
        self.fwd_ssc, self.kn, self.rvs_shock, self.rvs_ssc = bool(fwd_ssc), bool(kn), bool(rvs_shock), bool(rvs_ssc)
        if jet not in JET_TYPES:
            raise ValueError(f"Unknown jet type: {jet}")
        if medium not in MEDIUM_TYPES:
            raise ValueError(f"Unknown medium type: {medium}")
        self.z, self.lumi_dist, self.jet, self.medium = float(z), float(lumi_dist), jet, medium
        # None -> the Model ctor's mode-aware default (fitter.py:120-123, pymodel.h:630-637)
        if resolution is None:
            resolution = (0.06, 0.2, 10.0) if self.rvs_shock else (0.06, 0.15, 6.0)
        self.resolution, self.rtol, self.radiative_fireball = tuple(resolution), float(rtol), bool(radiative_fireball)
        self.device = device
        self._point_t, self._point_nu, self._point_flux, self._point_err, self._point_weights = [], [], [], [], []
        self._band_obs = []
        self._ext_kernel = None
        self._ext_kernels = {}  # z -> 0.4 ln10 k(lambda_rest) over the consolidated point data
        self._ext_z = float(z)
        self._all_t = None

    @staticmethod
    def _checked_observations(t, f_nu, err, weights, who):
        """The boundary checks of every add_* method (fitter.py:212-254): same shapes, non-empty, finite fluxes, finite
        positive errors, finite non-negative weights.  Returns float64 arrays (weights default to ones)."""
        t, f_nu, err = (np.asarray(a, dtype=np.float64) for a in (t, f_nu, err))
        if t.size == 0:
            raise ValueError(f"{who}: time array is empty")
        if not (t.shape == f_nu.shape == err.shape):
            raise ValueError(f"{who}: t, f_nu, err must have the same shape; got t.shape={t.shape}, "
                             f"f_nu.shape={f_nu.shape}, err.shape={err.shape}")
        if not np.isfinite(f_nu).all():
            raise ValueError(f"{who}: f_nu contains {int((~np.isfinite(f_nu)).sum())} non-finite (NaN or inf) values")
        if not np.isfinite(err).all() or (err <= 0).any():
            raise ValueError(f"{who}: err must be finite and > 0 at every point (got min={float(err.min())}, "
                             f"max={float(err.max())})")
        if weights is None:
            w = np.ones_like(t)
        else:
            w = np.asarray(weights, dtype=np.float64)
            if w.shape != t.shape:
                raise ValueError(f"{who}: weights.shape={w.shape} must match t.shape={t.shape}")
            if not np.isfinite(w).all() or (w < 0).any():
                raise ValueError(f"{who}: weights must be finite and >= 0 at every point")
        return t, f_nu, err, w

    def _add_points(self, t, nu, f_nu, err, w):
        self._point_t.append(t)
        self._point_nu.append(nu)
        self._point_flux.append(f_nu)
        self._point_err.append(err)
        self._point_weights.append(w)
        self._all_t = None

    # fitter.py:256-282
    def add_flux_density(self, nu, t, f_nu, err, weights=None, label=None):
        """Light-curve data at one frequency nu [Hz] (`label` is accepted for API compatibility; it only names plot legends)."""
        nu_arr = np.asarray(nu, dtype=np.float64)
        if not np.isfinite(nu_arr).all() or (nu_arr <= 0).any():
            raise ValueError(f"add_flux_density: nu must be finite and > 0, got {nu}")
        t, f_nu, err, w = self._checked_observations(t, f_nu, err, weights, "add_flux_density")
        if nu_arr.ndim != 0 and nu_arr.shape != t.shape:  # extension: one frequency per point
            raise ValueError(f"add_flux_density: an array nu must have the shape of t, got {nu_arr.shape} vs {t.shape}")
        self._add_points(t, np.full_like(t, float(nu_arr)) if nu_arr.ndim == 0 else nu_arr.copy(), f_nu, err, w)

    # fitter.py:284-314
    def add_spectrum(self, t, nu, f_nu, err, weights=None):
        """A broadband spectrum at one time t [s]: one point-data row per frequency."""
        if np.ndim(t) != 0 or not np.isfinite(t) or t <= 0:
            raise ValueError(f"add_spectrum: t must be finite and > 0, got {t}")
        nu = np.asarray(nu, dtype=np.float64)
        if nu.size and (not np.isfinite(nu).all() or (nu <= 0).any()):
            raise ValueError(f"add_spectrum: nu must be finite and > 0 at every point (got min={float(nu.min())}, "
                             f"max={float(nu.max())})")
        nu, f_nu, err, w = self._checked_observations(nu, f_nu, err, weights, "add_spectrum")  # nu is the axis array here
        self._add_points(np.full_like(nu, float(t)), nu, f_nu, err, w)

    # fitter.py:316-377
    def add_flux(self, band, t, flux, err, num_points=5, weights=None):
        """Band-integrated fluxes [erg/cm^2/s] over band = (nu_min, nu_max) [Hz]; each group is one Model.flux request."""
        try:
            nu_min, nu_max = band
        except (TypeError, ValueError):
            raise ValueError(f"add_flux: band must be a (nu_min, nu_max) tuple in Hz, got {band!r}") from None
        if not (np.isfinite(nu_min) and np.isfinite(nu_max) and 0 < nu_min < nu_max):
            raise ValueError(f"add_flux: band must satisfy 0 < nu_min < nu_max with both finite; got nu_min={nu_min}, "
                             f"nu_max={nu_max}")
        if num_points < 2:
            raise ValueError(f"add_flux: num_points must be >= 2 for band integration, got {num_points}")
        t, flux, err, w = self._checked_observations(t, flux, err, weights, "add_flux")
        if t.ndim != 1:
            raise ValueError("add_flux: t, flux and err must be 1-D arrays")
        if np.any(flux <= 0):
            raise ValueError("add_flux: the log-flux likelihood requires strictly positive fluxes")
        order = np.argsort(t)
        self._band_obs.append(dict(nu_min=float(nu_min), nu_max=float(nu_max), num_points=int(num_points),
                                   t=np.ascontiguousarray(t[order]), ln_flux=np.ascontiguousarray(np.log(flux[order])),
                                   ln_err=np.ascontiguousarray(err[order] / flux[order]),
                                   weights=np.ascontiguousarray(w[order])))

    # fitter.py:407-451
    def _consolidate_data(self):
        if self._all_t is not None:
            return
        if not self._point_t:
            if not self._band_obs:
                raise ValueError("no data: call add_flux_density or add_flux first")
            self._all_t = self._all_nu = self._all_log_flux = self._all_log_err = self._all_weights = np.array([])
            return
        t = np.concatenate(self._point_t)
        nu = np.concatenate(self._point_nu)
        f = np.concatenate(self._point_flux)
        e = np.concatenate(self._point_err)
        w = np.concatenate(self._point_weights)
        order = np.argsort(t)
        t, nu, f, e, w = t[order], nu[order], f[order], e[order], w[order].copy()
        s = w.sum()
        if s > 0:
            w *= len(w) / s
        if np.any(f <= 0) or np.any(e <= 0):
            raise ValueError("the log-flux likelihood requires strictly positive fluxes and errors")
        self._all_t, self._all_nu = np.ascontiguousarray(t), np.ascontiguousarray(nu)
        self._all_log_flux = np.ascontiguousarray(np.log(f))
        self._all_log_err = np.ascontiguousarray(e / f)
        self._all_weights = np.ascontiguousarray(w)
        if self.extinction is not None:  # fitter.py:439-449: rest-frame wavelengths, kernel = 0.4 ln10 k(lambda)
            self._ext_z = self.z
            lam_rest_cm = (2.99792458e10 / self._all_nu) / (1.0 + self.z)
            self._ext_kernel = np.ascontiguousarray(0.4 * np.log(10.0) * np.asarray(self._k_lambda(lam_rest_cm), dtype=np.float64))
            self._ext_kernels = {self._ext_z: self._ext_kernel}

    def _k_lambda(self, lam_rest_cm):
        """k(lambda) of the configured law.  The reference hands custom callables (lam_rest_cm, params) (fitter.py:445-449);
        such a two-argument law is called with params=None here, so one that really depends on the sampled parameters
        fails loudly instead of being frozen silently."""
        import inspect
        try:
            n_pos = sum(1 for q in inspect.signature(self.extinction).parameters.values()
                        if q.kind in (q.POSITIONAL_ONLY, q.POSITIONAL_OR_KEYWORD) and q.default is q.empty)
        except (TypeError, ValueError):
            n_pos = 1
        return self.extinction(lam_rest_cm, None) if n_pos >= 2 else self.extinction(lam_rest_cm)

    def _base_params(self, fixed):
        vals = dict(MODEL_PARAM_DEFAULTS)
        vals.update(fixed)
        p = _lib.ModelParams()
        _lib.load().vag_params_default(C.byref(p))
        p.jet_type, p.medium_type = JET_TYPES[self.jet], MEDIUM_TYPES[self.medium]
        p.theta_c, p.E_iso, p.Gamma0, p.k_e, p.k_g = vals["theta_c"], vals["E_iso"], vals["Gamma0"], vals["k_e"], vals["k_g"]
        p.theta_w, p.E_iso_w, p.Gamma0_w, p.duration = vals["theta_w"], vals["E_iso_w"], vals["Gamma0_w"], vals["tau"]
        p.n_ism, p.A_star, p.n0, p.k_m = vals["n_ism"], vals["A_star"], vals["n0"], vals["k_m"]
        if self.jet == "uniform":
            p.theta_c = math.pi / 2
        p.lumi_dist, p.z, p.theta_obs = self.lumi_dist, self.z, vals["theta_v"]
        p.eps_e, p.eps_B, p.p, p.xi_e = vals["eps_e"], vals["eps_B"], vals["p"], vals["xi_e"]
        p.phi_resol, p.theta_resol, p.t_resol = self.resolution
        p.rtol = self.rtol
        p.radiative_fireball = 1 if self.radiative_fireball else 0
        p.flags = (_lib.FLAG_SSC if self.fwd_ssc else 0) | (_lib.FLAG_KN if self.kn else 0)  # fitter.py:466-473
        if self.magnetar and self.jet != "powerlaw_wing":  # fitting/utils.py:47-52: Magnetar(L0, t0, q) on the jets that take one
            p.flags |= _lib.FLAG_MAGNETAR                      # (config.py:99-121: powerlaw_wing has supports_magnetar=False: it is left out)
            p.mag_L0, p.mag_t0, p.mag_q = vals["L0"], vals["t0"], vals["q"]
        if self.rvs_shock:  # fitter.py:476-484: rvs_rad = Radiation(eps_e_r, eps_B_r, p_r, xi_e_r, ssc=rvs_ssc, kn=kn)
            p.flags |= _lib.FLAG_RVS | (_lib.FLAG_RVS_SSC if self.rvs_ssc else 0) | (_lib.FLAG_RVS_KN if self.kn else 0)
            p.rvs_eps_e, p.rvs_eps_B, p.rvs_p, p.rvs_xi_e = vals["eps_e_r"], vals["eps_B_r"], vals["p_r"], vals["xi_e_r"]
        return p

    def build_spec(self, param_defs: Sequence[ParamDef], priors=None, use_priors=False):
        """The transformer of fitting/utils.py:110-135 as a C-ABI slot map (vag_fit_spec).  With ``use_priors`` the spec also
        carries the sampler-space bounds and the priors of fitting/params.py:209-227 (Uniform(lower, upper) unless ``priors``
        names another one), so that the device applies the bounds mask and adds sum ln prior (samplers.py:72-91)."""
        self._consolidate_data()
        fixed = {pd.name: (pd.initial if pd.initial is not None else pd.lower) for pd in param_defs if pd.scale is Scale.fixed}
        free = [pd for pd in param_defs if pd.scale is not Scale.fixed]
        if len(free) > 16:
            raise ValueError("at most 16 free parameters")
        spec = _lib.FitSpec()
        spec.base = self._base_params(fixed)
        # fixed parameters that are Model / Observer fields rather than ModelParams entries (z, lumi_dist, sigma0, the theta_obs /
        # duration aliases ...) go straight into their slot, exactly like a free parameter would
        base_fields = (C.c_double * 40).from_address(C.addressof(spec.base) + _lib.ModelParams.theta_c.offset)
        for name, value in fixed.items():
            if name in MODEL_PARAM_DEFAULTS or name == "A_V":
                continue
            if name not in _lib.PARAM_SLOTS:
                raise ValueError(f"parameter {name} is not accepted by the accelerated path")
            base_fields[_lib.PARAM_SLOTS[name]] = float(value)
        spec.ndim = len(free)
        for d, pd in enumerate(free):
            if pd.name == "A_V":
                spec.slot[d] = _lib.P_A_V
            elif pd.name not in _lib.PARAM_SLOTS:
                raise ValueError(f"parameter {pd.name} is not accepted by the accelerated path")
            else:
                spec.slot[d] = _lib.PARAM_SLOTS[pd.name]
            spec.is_log[d] = 1 if pd.scale is Scale.log else 0
        spec.a_v_fixed = float(fixed.get("A_V", 0.0))
        if self.extinction is not None and any(pd.name == "z" for pd in free):
            raise ValueError("a free 'z' cannot be combined with Fitter(extinction=...): the law's rest-frame wavelengths are fixed per fit")
        z_eff = float(fixed.get("z", self.z))
        if self.extinction is not None and self._all_t.size and z_eff != self._ext_z:
            # a fixed 'z' ParamDef overrides Fitter.z in the model: the rest-frame wavelengths of the law must follow it.  One
            # kernel per z, all kept for the Fitter's lifetime: earlier specs (a device_evaluator's closure) still point at theirs
            ext = self._ext_kernels.get(z_eff)
            if ext is None:
                lam_rest_cm = (2.99792458e10 / self._all_nu) / (1.0 + z_eff)
                ext = self._ext_kernels[z_eff] = np.ascontiguousarray(
                    0.4 * np.log(10.0) * np.asarray(self._k_lambda(lam_rest_cm), dtype=np.float64))
            self._ext_z, self._ext_kernel = z_eff, ext
        spec.ext_kernel = self._ext_kernel.ctypes.data_as(_dp) if self._ext_kernel is not None else None
        self._band_structs = (_lib.BandObs * max(len(self._band_obs), 1))()
        for g, bd in enumerate(self._band_obs):
            b = self._band_structs[g]
            b.nu_min, b.nu_max, b.num_points, b.n = bd["nu_min"], bd["nu_max"], bd["num_points"], bd["t"].size
            b.t, b.ln_flux = bd["t"].ctypes.data_as(_dp), bd["ln_flux"].ctypes.data_as(_dp)
            b.ln_err, b.weight = bd["ln_err"].ctypes.data_as(_dp), bd["weights"].ctypes.data_as(_dp)
        spec.n_bands = len(self._band_obs)
        spec.bands = self._band_structs
        spec.n_data = self._all_t.size
        spec.t = self._all_t.ctypes.data_as(_dp)
        spec.nu = self._all_nu.ctypes.data_as(_dp)
        spec.ln_flux = self._all_log_flux.ctypes.data_as(_dp)
        spec.ln_err = self._all_log_err.ctypes.data_as(_dp)
        spec.weight = self._all_weights.ctypes.data_as(_dp)
        # the struct holds raw pointers: what they point at lives as long as the spec (a device_evaluator closure keeps its spec
        # while the Fitter may consolidate new data or another z)
        spec._keep_alive = (self._ext_kernel, self._band_structs, list(self._band_obs), self._all_t, self._all_nu,
                            self._all_log_flux, self._all_log_err, self._all_weights)
        # sampler-space bounds: log10 of the ParamDef bounds for LOG-scale parameters (fitting/params.py:196-201)
        lower = np.array([np.log10(pd.lower) if pd.scale is Scale.log else pd.lower for pd in free], dtype=np.float64)
        upper = np.array([np.log10(pd.upper) if pd.scale is Scale.log else pd.upper for pd in free], dtype=np.float64)
        self._host_priors, self._prior_specs = [], []
        spec.use_priors = 1 if use_priors else 0
        for d, pd in enumerate(free):
            spec.lower[d], spec.upper[d] = lower[d], upper[d]
            kind, a, b = _device_prior((priors or {}).get(pd.name))
            if kind == _lib.PRIOR_NONE:
                self._host_priors.append((d, priors[pd.name]))
            self._prior_specs.append((kind, a, b, (priors or {}).get(pd.name)))
            spec.prior_kind[d], spec.prior_a[d], spec.prior_b[d] = kind, a, b
        return spec, lower, upper

    # fitting/params.py validate_parameters: the checks that do not depend on the sampler
    def validate_parameters(self, param_defs: Sequence[ParamDef]) -> None:
        names = [pd.name for pd in param_defs]
        if len(set(names)) != len(names):
            raise ValueError("duplicate parameter names")
        for pd in param_defs:
            if pd.name != "A_V" and pd.name not in _lib.PARAM_SLOTS:
                raise ValueError(f"parameter {pd.name} is not accepted by the accelerated path")
            if pd.scale is Scale.fixed:
                continue
            if not (np.isfinite(pd.lower) and np.isfinite(pd.upper) and pd.lower < pd.upper):
                raise ValueError(f"{pd.name}: need finite lower < upper, got [{pd.lower}, {pd.upper}]")
            if pd.scale is Scale.log and pd.lower <= 0:
                raise ValueError(f"{pd.name}: log-scale parameters need lower > 0")
        if "A_V" in names and self.extinction is None:
            raise ValueError("A_V needs Fitter(extinction=...)")

    def _params_at(self, sample, param_defs, resolution=None):
        """vag_model_params and A_V of one point of sampler space (the transformer of fitting/utils.py:110-135)."""
        spec, _, _ = self.build_spec(param_defs)
        sample = np.asarray(sample, dtype=np.float64).reshape(-1)
        if sample.size != spec.ndim:
            raise ValueError(f"expected {spec.ndim} free parameters, got {sample.size}")
        p = _lib.ModelParams.from_buffer_copy(bytes(spec.base))
        fields = (C.c_double * 40).from_address(C.addressof(p) + _lib.ModelParams.theta_c.offset)
        a_v = spec.a_v_fixed
        for d in range(spec.ndim):
            val = 10.0 ** sample[d] if spec.is_log[d] else sample[d]
            if spec.slot[d] == _lib.P_A_V:
                a_v = val
            else:
                fields[spec.slot[d]] = val
        if resolution is not None:
            p.phi_resol, p.theta_resol, p.t_resol = (float(x) for x in resolution)
        return p, float(a_v)

    # fitter.py:1089-1099
    def model(self, best_params, param_defs, resolution=None):
        """The underlying Model at a point of sampler space (no extinction applied)."""
        from .model import Model
        return Model.from_params(self._params_at(best_params, param_defs, resolution)[0], device=self.device)

    # fitter.py:779-804
    def flux_density_grid(self, best_params, t, nu, param_defs, resolution=None):
        """FluxDict on a (t, nu) grid at a point of sampler space; host-galaxy extinction (Fitter(extinction=...), A_V != 0) is
        applied per frequency to every component, like the fitter's own chi-squared path."""
        from .model import Model, FluxDict
        p, a_v = self._params_at(best_params, param_defs, resolution)
        res = Model.from_params(p, device=self.device).flux_density_grid(t, nu)
        if self.extinction is None or a_v == 0.0:
            return res
        lam_rest_cm = (2.99792458e10 / np.asarray(nu, dtype=np.float64)) / (1.0 + self.z)
        att = np.exp(-a_v * 0.4 * np.log(10.0) * np.asarray(self._k_lambda(lam_rest_cm), dtype=np.float64))[:, None]
        comps = [c * att if np.ndim(c) == 2 else None for c in (res.fwd.sync, res.fwd.ssc, res.rvs.sync, res.rvs.ssc)]
        return FluxDict(*comps)

    # fitter.py:806-833
    def flux(self, best_params, t, band, param_defs, num_points=5, resolution=None):
        """Band-integrated flux [erg/cm^2/s] over band = (nu_min, nu_max) at a point of sampler space (no extinction)."""
        nu_min, nu_max = band
        return self.model(best_params, param_defs, resolution).flux(t, nu_min, nu_max, num_points)

    # fitter.py:545-674 (the emcee branch; the stretch-move sampler of vegasafterglow_amd.sampling needs no third-party package)
    def fit(self, param_defs, nwalkers=None, nsteps=1000, nburn=0, seed=0, **kw):
        from . import sampling
        self.validate_parameters(param_defs)
        ndim = sum(1 for pd in param_defs if pd.scale is not Scale.fixed)
        return sampling.fit(self, param_defs, nwalkers=nwalkers or max(2 * ndim + 2, 32), nsteps=nsteps, nburn=nburn, seed=seed, **kw)

    def loglike_batch(self, samples, param_defs):
        """ln L for each row of samples[nb, ndim] (one batched device call)."""
        spec, _, _ = self.build_spec(param_defs)
        samples = np.ascontiguousarray(samples, dtype=np.float64)
        if samples.ndim != 2 or samples.shape[1] != spec.ndim:
            raise ValueError("samples must be [nb, ndim]")
        return self._device_loglike(spec, samples)

    def device_evaluator(self, param_defs, priors=None, use_priors=False, context=None):
        """eval_dev(theta) -> (values, costs) for vegasafterglow_amd.dist.WalkerSharder: theta is a float64 torch tensor [k, ndim]
        on this fitter's GPU; values = ln L (use_priors=False) or ln L + ln prior with the bounds mask (use_priors=True),
        costs = the engine's per-walker cell counts -- both float64[k] tensors on the device.  The call runs on torch's current
        stream through vag_loglike_batch_dev: nothing crosses PCIe.  Priors that only exist as host objects are not accepted
        here (the device path must be self-contained)."""
        import torch
        spec, _, _ = self.build_spec(param_defs, priors=priors, use_priors=use_priors)
        param_defs_free = [pd.name for pd in param_defs if pd.scale is not Scale.fixed]
        if self._host_priors:
            raise ValueError("device_evaluator: only Uniform / Gaussian / LogUniform priors (or their tuple forms) run on the device; "
                             f"got host-only priors for {[param_defs_free[d] for d, _ in self._host_priors]}")
        lib = _lib.load()
        if context is None:
            h, lock = get_context(self.device)
        else:
            h, lock = context
        dev = torch.device("cuda", self.device)
        keep = [spec]  # the spec's arrays are owned by this Fitter; the struct itself by this closure

        def _on_current_stream(fn):
            # the context follows torch's current stream for the duration of the call and then goes back to the stream it was on:
            # a stream the caller destroys later must not stay bound to the (shared) context
            with lock:
                prev = C.c_void_p()
                _lib.check(lib.vag_ctx_get_stream(h, C.byref(prev)))
                _lib.check(lib.vag_ctx_set_stream(h, _lib.torch_stream_handle(torch.cuda.current_stream(dev))))
                try:
                    return fn()
                finally:
                    lib.vag_ctx_set_stream(h, prev)

        def eval_dev(theta, want_costs=True):
            theta = theta.contiguous()
            k = theta.shape[0]
            values = torch.empty((k,), dtype=torch.float64, device=dev)
            costs = torch.empty((k,), dtype=torch.float64, device=dev) if want_costs else None

            def run():
                _lib.check(lib.vag_loglike_batch_dev(h, C.byref(keep[0]), theta.data_ptr(), k, keep[0].ndim, values.data_ptr()))
                if want_costs:  # (one more launch: only a sharder that deals by cost asks for it)
                    _lib.check(lib.vag_last_model_costs_dev(h, k, costs.data_ptr()))
            _on_current_stream(run)
            return values, costs
        eval_dev.optional_costs = True

        class _Native:
            """The engine's own sharded call for dist.WalkerSharder: deal + this rank's block, then the scatter after the
            all-gather (vag_loglike_shard_begin_dev / vag_loglike_shard_end_dev); no host work besides the launches."""
            parts = (lib, h, lock, keep)
            check = staticmethod(_lib.check)

            @staticmethod
            def shard(theta_all, nb, rank, world, block):
                """Returns the ticket that names this call in flight; finish() takes it (ABI v13)."""
                ticket = C.c_uint64(0)
                _on_current_stream(lambda: _lib.check(lib.vag_loglike_shard_begin_dev(
                    h, C.byref(keep[0]), theta_all.data_ptr(), nb, keep[0].ndim, rank, world, block.data_ptr(), C.byref(ticket))))
                return ticket.value

            @staticmethod
            def finish(ticket, gathered, nb, world, out):
                _on_current_stream(lambda: _lib.check(lib.vag_loglike_shard_end_dev(h, ticket, gathered.data_ptr(), nb, world, out.data_ptr())))

            @staticmethod
            def state(nb, world, per):
                tab = torch.empty((world * per,), dtype=torch.int32, device=dev)
                cost = torch.empty((nb,), dtype=torch.float64, device=dev)
                _on_current_stream(lambda: _lib.check(lib.vag_loglike_shard_state_dev(h, nb, world, tab.data_ptr(), cost.data_ptr())))
                return tab.cpu().numpy().astype(np.int64).reshape(world, per), cost.cpu().numpy()

        _Native.lock = lock  # dist.WalkerSharder takes it around shard() and around finish() (not across the all-gather between them)
        eval_dev.native = _Native
        return eval_dev

    def _device_loglike(self, spec, samples):
        out = np.empty(samples.shape[0])
        h, lock = get_context(self.device)
        plan = _lib.Plan()
        with lock:
            _lib.check(_lib.load().vag_loglike_batch(h, C.byref(spec), samples.ctypes.data_as(_dp), samples.shape[0],
                                                     spec.ndim, out.ctypes.data_as(_dp)))
            _lib.load().vag_last_plan(h, C.byref(plan))
        if plan.n_models_capacity:
            # never silent: these walkers were NOT evaluated (their adaptive grid exceeds the engine's static limits)
            logger.warning("%d of %d walkers exceeded the engine grid limits and were assigned -inf",
                           plan.n_models_capacity, samples.shape[0])
        if plan.n_walkers_ssc_failed:
            logger.warning("%d of %d walkers had SSC tables outside the engine's capacity and were assigned -inf",
                           plan.n_walkers_ssc_failed, samples.shape[0])
        self.last_plan = plan
        return out

    def log_prob_batch(self, samples, param_defs, priors=None):
        """log_prob_batch of fitting/samplers.py:72-91 in ONE device call: walkers outside the ParamDef bounds are not evaluated
        and score -inf, the others ln L + sum ln prior.  ``priors`` maps parameter names to priors acting on the SAMPLER-space
        value: objects shaped like bilby.core.prior.Uniform / Gaussian / LogUniform (or ("gaussian", mu, sigma),
        ("log_uniform", minimum, maximum), ("uniform",)) run on the device; any other object with ``ln_prob`` is added on the host.
        Parameters without an entry get Uniform(lower, upper) (params.py:209-227)."""
        spec, _, _ = self.build_spec(param_defs, priors=priors, use_priors=True)
        samples = np.ascontiguousarray(np.atleast_2d(np.asarray(samples, dtype=np.float64)))
        if samples.shape[1] != spec.ndim:
            raise ValueError("samples must be [nb, ndim]")
        out = self._device_loglike(spec, samples)
        for d, prior in self._host_priors:
            out = out + np.asarray(prior.ln_prob(samples[:, d]), dtype=np.float64)
        out[~np.isfinite(out)] = -np.inf
        return out

    def make_log_prob_batch(self, param_defs, loglike_fn=None, priors=None):
        """log_prob_batch(samples) of fitting/samplers.py:72-91: out-of-bounds -> -inf; otherwise ln L + sum ln prior, evaluated
        by the device in one call (Fitter.log_prob_batch).  ``loglike_fn`` substitutes an evaluator of ln L alone (e.g. one that
        shards walkers over ranks); the bounds mask and EVERY prior of ``priors`` are then applied here on the host
        (host_ln_prior: the same closed forms as the device, or the prior object's ln_prob)."""
        if loglike_fn is None:
            return lambda samples: self.log_prob_batch(samples, param_defs, priors=priors)
        _, lower, upper = self.build_spec(param_defs, priors=priors, use_priors=True)
        prior_specs = list(self._prior_specs)
        fn = loglike_fn

        def log_prob_batch(samples):
            samples = np.atleast_2d(np.asarray(samples, dtype=np.float64))
            ln_prior = host_ln_prior(samples, lower, upper, prior_specs)  # -inf outside the bounds or a prior's support
            log_probs = np.full(samples.shape[0], -np.inf)
            idx = np.where(ln_prior > -np.inf)[0]
            if idx.size:
                ll = np.asarray(fn(samples[idx]), dtype=np.float64)
                ll[~np.isfinite(ll)] = -np.inf
                log_probs[idx] = ll + ln_prior[idx]
            return log_probs

        return log_prob_batch
