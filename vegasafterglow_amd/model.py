"""Host-side mirror of the reference's ``VegasAfterglowC`` objects for the accelerated path.

Same names, argument meaning, defaults and error behaviour as the pybind layer
(``/root/reference/pybind/pybind.cpp:205-223,347-377,384-483``) for what the MI355X engine covers:
``TophatJet / GaussianJet / PowerLawJet / TwoComponentJet``, ``ISM / Wind(k_m=2)``, ``Observer``,
``Radiation`` (forward shock, synchrotron), ``Model.flux_density_grid / flux_density / flux``.
Everything numerical happens in the HIP library through the C-ABI.
"""
import ctypes as C
import math
import threading

import numpy as np

from . import _lib
from ._lib import ModelParams


def _req(cond, msg):
    if not cond:
        raise ValueError(msg)


def _finite_pos(name, x):
    _req(math.isfinite(x) and x > 0, f"{name} must be positive and finite, got {x}")


class _Jet:
    jet_type = None
    spreading = False

    def _fill(self, p):
        raise NotImplementedError


class Magnetar:
    """Magnetar(L0, t0, q=2): energy injection L0 (1 + t / t0)^-q inside theta_c -- pybind.cpp:198-203, pymodel.h:34-54."""

    def __init__(self, L0, t0, q=2.0):
        _finite_pos("L0", L0)
        _finite_pos("t0", t0)
        _finite_pos("q", q)
        self.L0, self.t0, self.q = float(L0), float(t0), float(q)

    def __repr__(self):
        return f"Magnetar(L0={self.L0:.6g}, t0={self.t0:.6g}, q={self.q:.6g})"


class TophatJet(_Jet):
    """TophatJet(theta_c, E_iso, Gamma0, spreading=False, duration=1, magnetar=None) -- pybind.cpp:205."""
    jet_type = _lib.JET_TOPHAT

    def __init__(self, theta_c, E_iso, Gamma0, spreading=False, duration=1.0, magnetar=None):
        _req(math.isfinite(theta_c) and 0 < theta_c <= math.pi / 2, f"theta_c must be in (0, pi/2], got {theta_c}")
        _finite_pos("E_iso", E_iso)
        _req(math.isfinite(Gamma0) and Gamma0 > 1, f"Gamma0 must be > 1, got {Gamma0}")
        _finite_pos("duration", duration)
        if magnetar is not None and not isinstance(magnetar, Magnetar):
            raise TypeError("magnetar must be a Magnetar")
        self.theta_c, self.E_iso, self.Gamma0, self.duration = float(theta_c), float(E_iso), float(Gamma0), float(duration)
        self.spreading, self.magnetar = bool(spreading), magnetar

    def _fill(self, p):
        p.jet_type = self.jet_type
        p.theta_c, p.E_iso, p.Gamma0, p.duration = self.theta_c, self.E_iso, self.Gamma0, self.duration


class GaussianJet(TophatJet):
    """GaussianJet(theta_c, E_iso, Gamma0, ...) -- pybind.cpp:208."""
    jet_type = _lib.JET_GAUSSIAN


class PowerLawJet(TophatJet):
    """PowerLawJet(theta_c, E_iso, Gamma0, k_e, k_g, ...) -- pybind.cpp:211."""
    jet_type = _lib.JET_POWERLAW

    def __init__(self, theta_c, E_iso, Gamma0, k_e, k_g, spreading=False, duration=1.0, magnetar=None):
        super().__init__(theta_c, E_iso, Gamma0, spreading, duration, magnetar)
        _finite_pos("k_e", k_e)
        _finite_pos("k_g", k_g)
        self.k_e, self.k_g = float(k_e), float(k_g)

    def _fill(self, p):
        super()._fill(p)
        p.k_e, p.k_g = self.k_e, self.k_g


class TwoComponentJet(TophatJet):
    """TwoComponentJet(theta_c, E_iso, Gamma0, theta_w, E_iso_w, Gamma0_w, ...) -- pybind.cpp:214."""
    jet_type = _lib.JET_TWO_COMPONENT

    def __init__(self, theta_c, E_iso, Gamma0, theta_w, E_iso_w, Gamma0_w, spreading=False, duration=1.0,
                 magnetar=None):
        super().__init__(theta_c, E_iso, Gamma0, spreading, duration, magnetar)
        _req(math.isfinite(theta_w) and 0 < theta_w <= math.pi / 2, f"theta_w must be in (0, pi/2], got {theta_w}")
        _req(theta_w > theta_c, "theta_w (wing angle) must be greater than theta_c (core angle), "
                                f"got theta_w={theta_w}, theta_c={theta_c}")
        _finite_pos("E_iso_w", E_iso_w)
        _req(math.isfinite(Gamma0_w) and Gamma0_w > 1, f"Gamma0_w must be > 1, got {Gamma0_w}")
        self.theta_w, self.E_iso_w, self.Gamma0_w = float(theta_w), float(E_iso_w), float(Gamma0_w)

    def _fill(self, p):
        super()._fill(p)
        p.theta_w, p.E_iso_w, p.Gamma0_w = self.theta_w, self.E_iso_w, self.Gamma0_w


class StepPowerLawJet(TophatJet):
    """StepPowerLawJet(theta_c, E_iso, Gamma0, E_iso_w, Gamma0_w, k_e, k_g, spreading=False, duration=1, magnetar=None)
    -- pybind.cpp:221, pymodel.cpp:112-128."""
    jet_type = _lib.JET_STEP_POWERLAW

    def __init__(self, theta_c, E_iso, Gamma0, E_iso_w, Gamma0_w, k_e, k_g, spreading=False, duration=1.0, magnetar=None):
        super().__init__(theta_c, E_iso, Gamma0, spreading, duration, magnetar)
        _finite_pos("E_iso_w", E_iso_w)
        _req(math.isfinite(Gamma0_w) and Gamma0_w > 1, f"Gamma0_w must be > 1, got {Gamma0_w}")
        _finite_pos("k_e", k_e)
        _finite_pos("k_g", k_g)
        self.E_iso_w, self.Gamma0_w, self.k_e, self.k_g = float(E_iso_w), float(Gamma0_w), float(k_e), float(k_g)

    def _fill(self, p):
        super()._fill(p)
        p.E_iso_w, p.Gamma0_w, p.k_e, p.k_g = self.E_iso_w, self.Gamma0_w, self.k_e, self.k_g


class PowerLawWing(_Jet):
    """PowerLawWing(theta_c, E_iso_w, Gamma0_w, k_e, k_g, spreading=False, duration=1) -- pybind.cpp:214,
    pymodel.cpp:90-110: a hollow-core wing, eps ~ (theta / theta_c)^-k_e outside theta_c."""
    jet_type = _lib.JET_POWERLAW_WING

    def __init__(self, theta_c, E_iso_w, Gamma0_w, k_e, k_g, spreading=False, duration=1.0):
        _req(math.isfinite(theta_c) and 0 < theta_c <= math.pi / 2, f"theta_c must be in (0, pi/2], got {theta_c}")
        _finite_pos("E_iso_w", E_iso_w)
        _req(math.isfinite(Gamma0_w) and Gamma0_w > 1, f"Gamma0_w must be > 1, got {Gamma0_w}")
        _finite_pos("k_e", k_e)
        _finite_pos("k_g", k_g)
        _finite_pos("duration", duration)
        self.theta_c, self.E_iso_w, self.Gamma0_w = float(theta_c), float(E_iso_w), float(Gamma0_w)
        self.k_e, self.k_g, self.duration, self.spreading = float(k_e), float(k_g), float(duration), bool(spreading)

    def _fill(self, p):
        p.jet_type = self.jet_type
        p.theta_c, p.E_iso_w, p.Gamma0_w, p.k_e, p.k_g, p.duration = (self.theta_c, self.E_iso_w, self.Gamma0_w, self.k_e,
                                                                        self.k_g, self.duration)


class MagnetizedTophatJet(TophatJet):
    """Top-hat profile on the generic Ejecta with a constant magnetisation sigma0 -- what the reference's test-suite
    builds as Ejecta(E_iso=lambda phi, theta: E_iso if theta <= theta_c else 0, Gamma0=..., sigma0=lambda ...: sigma0)
    (tests/python/golden/regenerate.py:141-149).  Arbitrary python-callback Ejecta profiles are not on the device."""
    jet_type = _lib.JET_MAGNETIZED_TOPHAT

    def __init__(self, theta_c, E_iso, Gamma0, sigma0, duration=1.0):
        super().__init__(theta_c, E_iso, Gamma0, duration=duration)
        _req(math.isfinite(sigma0) and sigma0 >= 0, f"sigma0 must be finite and non-negative, got {sigma0}")
        self.sigma0 = float(sigma0)

    def _fill(self, p):
        super()._fill(p)
        p.sigma0 = self.sigma0


class ISM:
    """ISM(n_ism) -- pybind.cpp:347, pymodel.cpp:148-151."""

    def __init__(self, n_ism):
        _req(math.isfinite(n_ism) and n_ism >= 0, f"n_ism must be non-negative and finite, got {n_ism}")
        self.n_ism = float(n_ism)

    def _fill(self, p):
        p.medium_type = _lib.MEDIUM_ISM
        p.n_ism, p.A_star, p.n0 = self.n_ism, 0.0, math.inf


class Wind:
    """Wind(A_star, n_ism=None, n0=None, k_m=2) -- pybind.cpp:350-355, pymodel.cpp:153-186."""

    def __init__(self, A_star, n_ism=None, n0=None, k_m=2):
        _finite_pos("A_star", A_star)
        _finite_pos("k_m", k_m)
        if n_ism is not None:
            _req(math.isfinite(n_ism) and n_ism >= 0, f"n_ism must be non-negative and finite, got {n_ism}")
        if n0 is not None:
            _req(n0 > 0, f"n0 must be > 0 (or +inf for no floor), got {n0}")
        self.k_m = float(k_m)  # k_m != 2 is the closed-form generic Medium of pymodel.cpp:167-185
        self.A_star = float(A_star)
        self.n_ism = 0.0 if n_ism is None else float(n_ism)
        self.n0 = math.inf if n0 is None else float(n0)

    def _fill(self, p):
        p.medium_type = _lib.MEDIUM_WIND
        p.n_ism, p.A_star, p.n0, p.k_m = self.n_ism, self.A_star, self.n0, self.k_m


class Observer:
    """Observer(lumi_dist, z, theta_obs, phi_obs=0) -- pybind.cpp:358-365, pymodel.h:208-222."""

    def __init__(self, lumi_dist, z, theta_obs, phi_obs=0.0):
        _finite_pos("lumi_dist", lumi_dist)
        _req(math.isfinite(z) and z >= 0, f"z must be non-negative and finite, got {z}")
        _req(math.isfinite(theta_obs) and 0 <= theta_obs <= math.pi, f"theta_obs must be in [0, pi], got {theta_obs}")
        _req(math.isfinite(phi_obs), f"phi_obs must be finite, got {phi_obs}")
        self.lumi_dist, self.z, self.theta_obs, self.phi_obs = float(lumi_dist), float(z), float(theta_obs), float(phi_obs)

    def __repr__(self):  # pymodel.h:262-272
        extra = f", phi_obs={self.phi_obs:.6g}" if self.phi_obs != 0 else ""
        return f"Observer(lumi_dist={self.lumi_dist:.6g}, z={self.z:.6g}, theta_obs={self.theta_obs:.6g}{extra})"


class Radiation:
    """Radiation(eps_e, eps_B, p, xi_e=1, ssc=False, kn=False) -- pybind.cpp:368-377, pymodel.h:241-260."""

    def __init__(self, eps_e, eps_B, p, xi_e=1.0, ssc=False, kn=False):
        for name, x in (("eps_e", eps_e), ("eps_B", eps_B), ("xi_e", xi_e)):
            _req(math.isfinite(x) and 0 < x <= 1, f"{name} must be in (0, 1], got {x}")
        _req(math.isfinite(p) and p > 1, f"p must be > 1, got {p}")
        self.eps_e, self.eps_B, self.p, self.xi_e, self.ssc, self.kn = float(eps_e), float(eps_B), float(p), float(xi_e), bool(ssc), bool(kn)

    def __repr__(self):  # pymodel.h:316-334
        s = f"Radiation(eps_e={self.eps_e:.6g}, eps_B={self.eps_B:.6g}, p={self.p:.6g}"
        if self.xi_e != 1:
            s += f", xi_e={self.xi_e:.6g}"
        return s + (", ssc=True" if self.ssc else "") + (", kn=True" if self.kn else "") + ")"


class Flux:
    """Flux{sync, ssc} (pybind.cpp:472-477); disabled components are 0-d zeros like the reference's."""

    def __init__(self, sync=None, ssc=None):
        self.sync = np.zeros(()) if sync is None else sync
        self.ssc = np.zeros(()) if ssc is None else ssc


class ShockDetails:
    """One shock of Model.details() with the reference's attribute names and array ranks (pybind.cpp:522-576)."""

    _MAP = {"t_comv": "t_comv", "r": "r", "theta": "theta_cell", "Gamma": "Gamma", "Gamma_th": "Gamma_th", "B_comv": "B",
            "N_p": "N_p", "gamma_m": "gamma_m", "gamma_c": "gamma_c", "gamma_a": "gamma_a", "gamma_M": "gamma_M", "N_e": "N_e",
            "nu_m": "nu_m", "nu_c": "nu_c", "nu_a": "nu_a", "nu_M": "nu_M", "I_nu_max": "I_nu_max"}

    def __init__(self, d):
        for attr, key in self._MAP.items():
            setattr(self, attr, d[key][None, :, :])
        self.t_obs, self.Doppler = d["t_obs"], d["Doppler"]

    def __repr__(self):
        return f"ShockDetails(shape={self.Gamma.shape})"


class SimulationDetails(dict):
    """Model.details(): attribute access like the reference's SimulationDetails, dict access to one shock's 2-D arrays."""

    def __repr__(self):
        return f"SimulationDetails(phi={self['phi'].size}, theta={self['theta'].size}, t_src={self['t_src'].shape[-1]})"


class FluxDict:
    """FluxDict{total, fwd, rvs} (pybind.cpp:478-483, pymodel.cpp:350-364)."""

    def __init__(self, fwd_sync, fwd_ssc=None, rvs_sync=None, rvs_ssc=None):
        self.fwd = Flux(sync=fwd_sync, ssc=fwd_ssc)
        self.rvs = Flux(sync=rvs_sync, ssc=rvs_ssc)
        total = fwd_sync.copy()  # PyFlux::calc_total order, pymodel.cpp:350-364
        for extra in (fwd_ssc, rvs_sync, rvs_ssc):
            if extra is not None:
                total = total + extra
        self.total = total
        for a in (self.total, self.fwd.sync, self.fwd.ssc, self.rvs.sync, self.rvs.ssc):
            a.setflags(write=False)


_ctx_lock = threading.Lock()
_ctx = {}


def get_context(device=0):
    """Process-wide engine context per device (the C-ABI context serialises its own stream)."""
    lib = _lib.load()
    with _ctx_lock:
        if device not in _ctx:
            h = C.c_void_p()
            _lib.check(lib.vag_ctx_create(device, C.byref(h)))
            _ctx[device] = (h, threading.RLock())  # re-entrant: a sharded call holds it across deal, all-gather and scatter
        return _ctx[device]


_coalescing = {}  # device -> True while concurrent single-model calls are gathered into batch calls


def set_coalescing(enabled=True, max_batch=64, wait_us=50, device=0):
    """Serve concurrent ``Model.flux_density_grid / flux_density / flux`` calls of a thread pool as batch calls (opt-in).

    The reference's samplers map ``eval_one`` over a ``ThreadPoolExecutor`` with one Model per thread and the GIL released inside the
    compute methods (pybind.cpp:424-448, fitting/samplers.py:59-70).  On one GPU context such calls are served one after the other, at
    one call latency each.  With coalescing on, a call blocks in the engine (GIL released) and the calls that wait at the same time with
    the same request -- times, frequencies / band -- run as ONE batch call (``vag_*_coalesced``, at most ``max_batch`` members; the first
    caller waits ``wait_us`` for company, later ones queue up while the GPU is busy).  Each caller gets its own model's result and its own
    model's error.  ``flux_density`` / ``flux`` results are the bits of an uncoalesced call; ``flux_density_grid`` may differ in the last
    bits (its sums are laid out per batch).  Returns the previous setting."""
    h, _ = get_context(device)
    prev = _coalescing.get(device, False)
    if enabled:
        _lib.check(_lib.load().vag_ctx_coalesce(h, int(max_batch), int(wait_us)))
    _coalescing[device] = bool(enabled)
    return prev


def coalescing_stats(device=0):
    """(calls served, batch calls issued) by the coalescer of this device's context so far."""
    h, _ = get_context(device)
    a, b = C.c_longlong(0), C.c_longlong(0)
    _lib.check(_lib.load().vag_ctx_coalesce_stats(h, C.byref(a), C.byref(b)))
    return a.value, b.value


def _as_f64(a, name):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    _req(a.ndim == 1, f"{name} must be one-dimensional")
    return a


_dp = C.POINTER(C.c_double)


class Model:
    """Model(jet, medium, observer, fwd_rad, rvs_rad=None, resolutions=None, rtol=1e-6, axisymmetric=True,
    radiative_fireball=True) -- pybind.cpp:384-422, pymodel.h:613-649."""

    def __init__(self, jet, medium, observer, fwd_rad, rvs_rad=None, resolutions=None, rtol=1e-6, axisymmetric=True,
                 radiative_fireball=True, device=0):
        if not isinstance(jet, _Jet):
            raise TypeError("jet must be TophatJet, GaussianJet, PowerLawJet, TwoComponentJet, StepPowerLawJet, PowerLawWing "
                            "or MagnetizedTophatJet")
        if not isinstance(medium, (ISM, Wind)):
            raise TypeError("medium must be ISM or Wind")
        if rvs_rad is not None and not isinstance(rvs_rad, Radiation):
            raise TypeError("rvs_rad must be a Radiation")
        if fwd_rad.kn and not fwd_rad.ssc:
            pass  # Klein-Nishina corrections only act through the IC cooling enabled by ssc (pymodel.h:567-577)
        _req(math.isfinite(rtol) and 0 < rtol < 1, f"rtol must be in (0, 1), got {rtol}")
        # forward-only runs default to the coarser calibrated grid, reverse-shock runs to the denser one (pymodel.h:630-637)
        default_res = (0.06, 0.2, 10.0) if rvs_rad is not None else (0.06, 0.15, 6.0)
        res = default_res if resolutions is None else tuple(float(x) for x in resolutions)
        for n, x in zip(("phi_resol", "theta_resol", "t_resol"), res):
            _finite_pos(n, x)
        self._jet, self._medium, self.observer, self.fwd_rad, self.rvs_rad = jet, medium, observer, fwd_rad, rvs_rad
        self.resolutions, self.rtol, self.axisymmetric, self.radiative_fireball = res, float(rtol), bool(axisymmetric), bool(radiative_fireball)
        self._device = device
        p = ModelParams()
        _lib.load().vag_params_default(C.byref(p))
        jet._fill(p)
        medium._fill(p)
        p.lumi_dist, p.z, p.theta_obs = observer.lumi_dist, observer.z, observer.theta_obs
        p.eps_e, p.eps_B, p.p, p.xi_e = fwd_rad.eps_e, fwd_rad.eps_B, fwd_rad.p, fwd_rad.xi_e
        p.phi_resol, p.theta_resol, p.t_resol = res
        p.rtol = self.rtol
        p.radiative_fireball = 1 if radiative_fireball else 0
        p.flags = (_lib.FLAG_SSC if fwd_rad.ssc else 0) | (_lib.FLAG_KN if fwd_rad.kn else 0)
        if getattr(jet, "spreading", False):
            p.flags |= _lib.FLAG_SPREADING
        if not axisymmetric:  # full-circle phi grid, no mirror / on-axis shortcut (grid-refinement.h:671-689)
            p.flags |= _lib.FLAG_NON_AXISYMMETRIC
        mag = getattr(jet, "magnetar", None)
        if mag is not None:
            p.flags |= _lib.FLAG_MAGNETAR
            p.mag_L0, p.mag_t0, p.mag_q = mag.L0, mag.t0, mag.q
        if rvs_rad is not None:
            p.flags |= _lib.FLAG_RVS | (_lib.FLAG_RVS_SSC if rvs_rad.ssc else 0) | (_lib.FLAG_RVS_KN if rvs_rad.kn else 0)
            p.rvs_eps_e, p.rvs_eps_B, p.rvs_p, p.rvs_xi_e = rvs_rad.eps_e, rvs_rad.eps_B, rvs_rad.p, rvs_rad.xi_e
        _lib.check(_lib.load().vag_params_validate(C.byref(p)))
        self.params = p

    @classmethod
    def from_params(cls, params, device=0):
        """A Model around an already filled vag_model_params (what Fitter._build_model hands to the engine,
        fitter.py:455-495); validated like the constructor's."""
        self = cls.__new__(cls)
        p = ModelParams.from_buffer_copy(bytes(params))
        _lib.check(_lib.load().vag_params_validate(C.byref(p)))
        self.params, self._device = p, device
        self._jet = self._medium = None
        self.observer = Observer(p.lumi_dist, p.z, p.theta_obs)
        self.fwd_rad = Radiation(p.eps_e, p.eps_B, p.p, p.xi_e, ssc=bool(p.flags & _lib.FLAG_SSC), kn=bool(p.flags & _lib.FLAG_KN))
        self.rvs_rad = None
        if p.flags & _lib.FLAG_RVS:
            self.rvs_rad = Radiation(p.rvs_eps_e, p.rvs_eps_B, p.rvs_p, p.rvs_xi_e, ssc=bool(p.flags & _lib.FLAG_RVS_SSC),
                                     kn=bool(p.flags & _lib.FLAG_RVS_KN))
        self.resolutions = (p.phi_resol, p.theta_resol, p.t_resol)
        self.rtol, self.radiative_fireball = p.rtol, bool(p.radiative_fireball)
        self.axisymmetric = not (p.flags & _lib.FLAG_NON_AXISYMMETRIC)
        return self

    def _has_components(self):
        return self.fwd_rad.ssc or self.rvs_rad is not None

    def _component_buffers(self, shape):
        """Output arrays for the enabled FluxDict components (None = disabled) and the matching double*[4]."""
        want = (True, self.fwd_rad.ssc, self.rvs_rad is not None, self.rvs_rad is not None and self.rvs_rad.ssc)
        comps = [np.empty(shape) if w else None for w in want]
        arr = (_dp * 4)(*[a.ctypes.data_as(_dp) if a is not None else None for a in comps])
        return comps, arr

    # -- Model.flux_density_grid: pybind.cpp:424, pymodel.cpp:498-514 --
    def flux_density_grid(self, t, nu):
        t, nu = _as_f64(t, "t"), _as_f64(nu, "nu")
        _req(t.size > 0, "time array must be non-empty")
        _req(nu.size > 0, "frequency array must be non-empty")
        out = np.empty((nu.size, t.size))
        h, lock = get_context(self._device)
        if _coalescing.get(self._device):  # no Python-side lock: the call blocks in the engine while its batch forms
            comps, arr = self._component_buffers((nu.size, t.size)) if self._has_components() else (None, None)
            _lib.check(_lib.load().vag_flux_density_grid_coalesced(
                h, C.byref(self.params), t.ctypes.data_as(_dp), t.size, nu.ctypes.data_as(_dp), nu.size,
                None if comps else out.ctypes.data_as(_dp), arr))
            return FluxDict(*comps) if comps else FluxDict(out)
        if self._has_components():
            comps, arr = self._component_buffers((nu.size, t.size))
            with lock:
                _lib.check(_lib.load().vag_flux_density_grid_components4_batch(
                    h, C.byref(self.params), 1, t.ctypes.data_as(_dp), t.size, nu.ctypes.data_as(_dp), nu.size, arr))
            return FluxDict(*comps)
        with lock:
            _lib.check(_lib.load().vag_flux_density_grid_batch(
                h, C.byref(self.params), 1, t.ctypes.data_as(_dp), t.size, nu.ctypes.data_as(_dp), nu.size,
                out.ctypes.data_as(_dp)))
        return FluxDict(out)

    # -- Model.flux_density: pybind.cpp:427, pymodel.cpp:373-389 --
    def flux_density(self, t, nu):
        t, nu = _as_f64(t, "t"), _as_f64(nu, "nu")
        _req(t.size > 0, "time array must be non-empty")
        _req(nu.size > 0, "frequency array must be non-empty")
        _req(t.size == nu.size, "time and frequency arrays must have the same size\nIf you intend to get grid-like "
                                "output, use the generic `flux_density_grid` instead")
        out = np.empty(t.size)
        h, lock = get_context(self._device)
        if _coalescing.get(self._device):
            comps, arr = self._component_buffers((t.size,)) if self._has_components() else (None, None)
            _lib.check(_lib.load().vag_flux_density_coalesced(
                h, C.byref(self.params), t.ctypes.data_as(_dp), nu.ctypes.data_as(_dp), t.size,
                None if comps else out.ctypes.data_as(_dp), arr))
            return FluxDict(*comps) if comps else FluxDict(out)
        if self._has_components():
            comps, arr = self._component_buffers((t.size,))
            with lock:
                _lib.check(_lib.load().vag_flux_density_components4_batch(
                    h, C.byref(self.params), 1, t.ctypes.data_as(_dp), nu.ctypes.data_as(_dp), t.size, arr))
            return FluxDict(*comps)
        with lock:
            _lib.check(_lib.load().vag_flux_density_batch(
                h, C.byref(self.params), 1, t.ctypes.data_as(_dp), nu.ctypes.data_as(_dp), t.size,
                out.ctypes.data_as(_dp)))
        return FluxDict(out)

    # -- Model.flux: pybind.cpp:430, pymodel.cpp:391-410 --
    def flux(self, t, nu_min, nu_max, num_nu):
        t = _as_f64(t, "t")
        _req(t.size > 0, "time array must be non-empty")
        out = np.empty(t.size)
        h, lock = get_context(self._device)
        if _coalescing.get(self._device):
            comps, arr = self._component_buffers((t.size,)) if self._has_components() else (None, None)
            _lib.check(_lib.load().vag_flux_coalesced(
                h, C.byref(self.params), t.ctypes.data_as(_dp), t.size, float(nu_min), float(nu_max), int(num_nu),
                None if comps else out.ctypes.data_as(_dp), arr))
            return FluxDict(*comps) if comps else FluxDict(out)
        if self._has_components():
            comps, arr = self._component_buffers((t.size,))
            with lock:
                _lib.check(_lib.load().vag_flux_components4_batch(
                    h, C.byref(self.params), 1, t.ctypes.data_as(_dp), t.size, float(nu_min), float(nu_max), int(num_nu), arr))
            return FluxDict(*comps)
        with lock:
            _lib.check(_lib.load().vag_flux_batch(h, C.byref(self.params), 1, t.ctypes.data_as(_dp), t.size,
                                                  float(nu_min), float(nu_max), int(num_nu), out.ctypes.data_as(_dp)))
        return FluxDict(out)

    # -- Model.flux_density_exposures: pybind.cpp:433-434, pymodel.cpp:412-496 --
    def _profile(self, kind, x):
        x = np.ascontiguousarray(np.atleast_1d(x), dtype=np.float64)
        out = np.empty_like(x)
        lib = _lib.load()
        h, lock = get_context(self._device)
        with lock:
            _lib.check(lib.vag_profile_eval(h, C.byref(self.params), kind, x.ctypes.data_as(_dp), x.size, out.ctypes.data_as(_dp)))
        return out

    def jet_E_iso(self, phi, theta):
        """Isotropic-equivalent energy [erg] of the jet at theta (pybind.cpp:444; the named profiles do not depend on phi)."""
        return self._profile(0, theta)

    def jet_Gamma0(self, phi, theta):
        """Initial Lorentz factor of the jet at theta (pybind.cpp:447)."""
        return self._profile(1, theta)

    def medium(self, phi, theta, r):
        """Mass density [g/cm^3] of the circumburst medium at radius r [cm] (pybind.cpp:441)."""
        return self._profile(2, r)

    def flux_density_exposures(self, t, nu, expo_time, num_points=10):
        """Exposure-averaged flux density: `num_points` samples across each [t_i, t_i + expo_time_i] window are
        evaluated as one sorted (t, nu) series on the device and averaged per exposure."""
        t, nu, expo = _as_f64(t, "t"), _as_f64(nu, "nu"), _as_f64(expo_time, "expo_time")
        _req(t.size == nu.size == expo.size, "time, frequency, and exposure time arrays must have the same size")
        _req(int(num_points) >= 2, "num_points must be at least 2 to sample within each exposure time")
        for i, e in enumerate(expo):
            _req(math.isfinite(e) and e > 0, f"expo_time[{i}] must be finite and > 0, got {e}")
        num_points = int(num_points)
        k = np.arange(num_points, dtype=np.float64)
        dt = expo / float(num_points - 1)
        t_s = (t[:, None] + k[None, :] * dt[:, None]).ravel()       # generate_exposure_sampling, pymodel.cpp:412-431
        nu_s = np.repeat(nu, num_points)
        idx = np.repeat(np.arange(t.size), num_points)
        order = np.argsort(t_s, kind="stable")
        series = self.flux_density(t_s[order], nu_s[order])

        def avg(a):                                                  # average_exposure_flux, pymodel.cpp:452-470
            if np.ndim(a) == 0:
                return None                                          # disabled component
            summed = np.zeros(t.size)
            np.add.at(summed, idx[order], a)
            return summed / float(num_points)

        return FluxDict(avg(series.fwd.sync), avg(series.fwd.ssc), avg(series.rvs.sync), avg(series.rvs.ssc))

    # -- Model.details (shock part): pybind.cpp:448, pymodel.cpp:315-348 --
    def details(self, t_min, t_max, rvs=False):
        """Model.details(t_min, t_max) (pybind.cpp:448,522-584): the returned object is used like the reference's
        SimulationDetails -- `.phi`, `.theta`, `.t_src`, `.fwd` and `.rvs` (ShockDetails with t_comv, r, theta, Gamma,
        Gamma_th, B_comv, N_p, gamma_m/c/a/M, N_e, nu_m/c/a/M, I_nu_max as (1, n_theta, n_t) arrays -- one solved phi
        slice -- and t_obs, Doppler as (n_phi_eff, n_theta, n_t)); `.rvs` is None without rvs_rad.  It is also a dict of
        the 2-D [n_theta][n_t] arrays of one shock (the forward one, or with rvs=True the reverse one)."""
        d = SimulationDetails(self._details_of(t_min, t_max, rvs))
        fwd = d if not rvs else self._details_of(t_min, t_max, False)
        d.fwd = ShockDetails(fwd)
        d.rvs = None
        if self.rvs_rad is not None:
            d.rvs = ShockDetails(d if rvs else self._details_of(t_min, t_max, True))
        d.phi, d.theta, d.t_src = fwd["phi"], fwd["theta"], fwd["t_src"][None, :, :]
        return d

    def _details_of(self, t_min, t_max, rvs):
        lib = _lib.load()
        h, lock = get_context(self._device)
        sh = _lib.DetailsShape()
        fn = lib.vag_details_rvs if rvs else lib.vag_details
        with lock:
            _lib.check(fn(h, C.byref(self.params), float(t_min), float(t_max), C.byref(sh), None))
            d = {"phi": np.zeros(sh.n_phi), "theta": np.zeros(sh.n_theta)}
            for n in ("t_src", "Gamma", "r", "t_comv", "B", "N_p", "Gamma_th"):
                d[n] = np.zeros((sh.n_theta, sh.n_t))
            out = _lib.DetailsOut(*[d[n].ctypes.data_as(_dp) for n, _ in _lib.DetailsOut._fields_])
            _lib.check(fn(h, C.byref(self.params), float(t_min), float(t_max), C.byref(sh), C.byref(out)))
        d["shape"] = dict(n_phi=sh.n_phi, n_theta=sh.n_theta, n_t=sh.n_t, n_reps=sh.n_reps, symmetry=sh.symmetry,
                          phi_mirrored=sh.phi_mirrored)
        # electron / photon break arrays of ShockDetails (pybind.cpp:522-546), reference units
        names = ("gamma_m", "gamma_c", "gamma_a", "gamma_M", "N_e", "nu_m", "nu_c", "nu_a", "nu_M", "I_nu_max", "theta_cell")
        for n in names:
            d[n] = np.zeros((sh.n_theta, sh.n_t))
        arr = (_dp * 11)(*[d[n].ctypes.data_as(_dp) for n in names])
        with lock:
            _lib.check(lib.vag_details_radiation(h, C.byref(self.params), float(t_min), float(t_max), 1 if rvs else 0, arr))
            # observer-frame time and Doppler factor of every (phi, theta, t) cell (ShockDetails.t_obs / .Doppler)
            npe = C.c_int(0)
            _lib.check(lib.vag_details_eat(h, C.byref(self.params), float(t_min), float(t_max), C.byref(npe), None, None))
            d["t_obs"] = np.zeros((npe.value, sh.n_theta, sh.n_t))
            d["Doppler"] = np.zeros((npe.value, sh.n_theta, sh.n_t))
            _lib.check(lib.vag_details_eat(h, C.byref(self.params), float(t_min), float(t_max), C.byref(npe),
                                           d["t_obs"].ctypes.data_as(_dp), d["Doppler"].ctypes.data_as(_dp)))
        return d

    @staticmethod
    def profile_enable(on=True, device=0):
        """Turn the per-stage profiler on / off for the following calls on this device (the reference needs a build with
        -DAFTERGLOW_PROFILE; here it is a run-time switch because every stage costs two event records)."""
        h, _ = get_context(device)
        _lib.check(_lib.load().vag_ctx_profile(h, 1 if on else 0))

    @staticmethod
    def profile_data(device=0):
        """Model.profile_data() (pybind.cpp:458-459): per-stage device time [ms] of the last computation under the reference
        profiler's stage names (pymodel.h:877-953); see vag_profile in the C header for what each covers here."""
        h, _ = get_context(device)
        p = _lib.Profile()
        _lib.check(_lib.load().vag_last_profile(h, C.byref(p)))
        return {n: getattr(p, n) for n, _ in _lib.Profile._fields_}

    def stage_times(self):
        """Per-stage device milliseconds of the last call (names follow pybind/pymodel.h:877-953)."""
        st = _lib.StageTimes()
        h, _ = get_context(self._device)
        _lib.load().vag_last_stage_times(h, C.byref(st))
        return {"EAT_grid+grid": st.grid_ms, "dynamics": st.dynamics_ms, "syn_electrons+syn_photons": st.cells_ms,
                "sync_flux": st.flux_ms, "reduce": st.reduce_ms, "total": st.total_ms}

    def __repr__(self):
        o, r = self.observer, self.fwd_rad
        return (f"Model(observer=Observer(lumi_dist={o.lumi_dist:.6g}, z={o.z:.6g}, theta_obs={o.theta_obs:.6g}),\n"
                f"      fwd_rad=Radiation(eps_e={r.eps_e:.6g}, eps_B={r.eps_B:.6g}, p={r.p:.6g}),\n"
                f"      resolutions=({self.resolutions[0]:.6g}, {self.resolutions[1]:.6g}, {self.resolutions[2]:.6g}), "
                f"rtol={self.rtol:.6g})")
