"""ctypes mirror of include/vegasafterglow_amd.h shared by the tests, bench.py and smoke().

Only structure layouts and loaders live here; no arithmetic.  The oracle libraries
(oracle/liboracle.so, oracle/_ref/libvag_ref.so) are CHECKERS: they are loaded from tests,
__graft_entry__.smoke() and bench.py's cpu_baseline leg only.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

JET_TOPHAT, JET_GAUSSIAN, JET_POWERLAW, JET_TWO_COMPONENT, JET_MAGNETIZED_TOPHAT = 0, 1, 2, 3, 4
JET_STEP_POWERLAW, JET_POWERLAW_WING = 5, 6
MEDIUM_ISM, MEDIUM_WIND = 0, 1
JET_IDS = {"TophatJet": 0, "GaussianJet": 1, "PowerLawJet": 2, "TwoComponentJet": 3, "MagnetizedTophat": 4,
           "StepPowerLawJet": 5, "PowerLawWing": 6}
MEDIUM_IDS = {"ISM": 0, "Wind": 1}


class ModelParams(C.Structure):
    _fields_ = [
        ("jet_type", C.c_int32), ("medium_type", C.c_int32),
        ("theta_c", C.c_double), ("E_iso", C.c_double), ("Gamma0", C.c_double),
        ("k_e", C.c_double), ("k_g", C.c_double), ("theta_w", C.c_double),
        ("E_iso_w", C.c_double), ("Gamma0_w", C.c_double), ("duration", C.c_double),
        ("n_ism", C.c_double), ("A_star", C.c_double), ("n0", C.c_double),
        ("lumi_dist", C.c_double), ("z", C.c_double), ("theta_obs", C.c_double),
        ("eps_e", C.c_double), ("eps_B", C.c_double), ("p", C.c_double), ("xi_e", C.c_double),
        ("phi_resol", C.c_double), ("theta_resol", C.c_double), ("t_resol", C.c_double), ("rtol", C.c_double),
        ("radiative_fireball", C.c_int32), ("flags", C.c_int32),
        ("rvs_eps_e", C.c_double), ("rvs_eps_B", C.c_double), ("rvs_p", C.c_double), ("rvs_xi_e", C.c_double),
        ("sigma0", C.c_double), ("k_m", C.c_double),
        ("mag_L0", C.c_double), ("mag_t0", C.c_double), ("mag_q", C.c_double),
    ]


assert C.sizeof(ModelParams) == 272


class DetailsShape(C.Structure):
    _fields_ = [("n_phi", C.c_int32), ("n_theta", C.c_int32), ("n_t", C.c_int32), ("n_reps", C.c_int32),
                ("symmetry", C.c_int32), ("phi_mirrored", C.c_int32)]


class DetailsOut(C.Structure):
    _fields_ = [(n, C.POINTER(C.c_double)) for n in
                ("phi", "theta", "t_src", "Gamma", "r", "t_comv", "B", "N_p", "Gamma_th")]


def make_params(jet="TophatJet", medium="ISM", theta_c=0.1, E_iso=1e52, Gamma0=300.0, k_e=2.0, k_g=2.0,
                theta_w=0.3, E_iso_w=1e50, Gamma0_w=50.0, duration=1.0, n_ism=1.0, A_star=0.0,
                n0=float("inf"), lumi_dist=1e28, z=1.0, theta_obs=0.0, eps_e=0.1, eps_B=0.01, p=2.3,
                xi_e=1.0, resolutions=None, rtol=1e-6, radiative_fireball=True, ssc=False, kn=False, rvs=None, sigma0=0.0,
                spreading=False, k_m=2.0, magnetar=None, axisymmetric=True):
    # magnetar = (L0, t0, q) mirrors jet(..., magnetar=Magnetar(L0, t0, q))
    # rvs = dict(eps_e, eps_B, p[, xi_e, ssc, kn]) mirrors Model(rvs_rad=Radiation(...)); the default resolutions are
    # mode-aware like the reference's Model ctor (pybind/pymodel.h:630-637)
    """Flatten Model(jet, medium, Observer, Radiation, resolutions, rtol) keyword arguments."""
    q = ModelParams()
    q.jet_type = JET_IDS[jet] if isinstance(jet, str) else int(jet)
    q.medium_type = MEDIUM_IDS[medium] if isinstance(medium, str) else int(medium)
    q.theta_c, q.E_iso, q.Gamma0, q.k_e, q.k_g = theta_c, E_iso, Gamma0, k_e, k_g
    q.theta_w, q.E_iso_w, q.Gamma0_w, q.duration = theta_w, E_iso_w, Gamma0_w, duration
    q.n_ism, q.A_star, q.n0 = n_ism, A_star, n0
    q.lumi_dist, q.z, q.theta_obs = lumi_dist, z, theta_obs
    q.eps_e, q.eps_B, q.p, q.xi_e = eps_e, eps_B, p, xi_e
    if resolutions is None:
        resolutions = (0.06, 0.2, 10.0) if rvs else (0.06, 0.15, 6.0)
    q.phi_resol, q.theta_resol, q.t_resol = resolutions
    q.rtol = rtol
    q.sigma0 = sigma0
    q.k_m = k_m
    q.radiative_fireball = 1 if radiative_fireball else 0
    q.flags = (1 if ssc else 0) | (2 if kn else 0) | (32 if spreading else 0) | (0 if axisymmetric else 128)
    if magnetar:
        q.flags |= 64
        q.mag_L0, q.mag_t0, q.mag_q = magnetar
    if rvs:
        q.flags |= 4 | (8 if rvs.get("ssc") else 0) | (16 if rvs.get("kn") else 0)
        q.rvs_eps_e, q.rvs_eps_B, q.rvs_p, q.rvs_xi_e = rvs["eps_e"], rvs["eps_B"], rvs["p"], rvs.get("xi_e", 1.0)
    return q


def params_from_golden_config(cfg):
    """Map a tests/golden/*.npz `config` JSON (reference tests/python/golden/regenerate.py) to params."""
    jet = dict(cfg["jet"])
    med = dict(cfg["medium"])
    kw = dict(jet=jet.pop("type"), medium=med.pop("type"))
    if kw["medium"] == "Wind":
        kw["n_ism"] = 0.0  # Wind(A_star, n_ism=None, n0=None): pybind/pymodel.cpp:153-166
    kw.update(jet)  # MagnetizedTophat carries sigma0
    kw.update(med)
    kw.update(cfg["observer"])
    kw.update(cfg["fwd_rad"])  # includes ssc / kn
    if "rvs_rad" in cfg:
        kw["rvs"] = dict(cfg["rvs_rad"])
    if "resolutions" in cfg:
        kw["resolutions"] = tuple(cfg["resolutions"])
    if "radiative_fireball" in cfg:
        kw["radiative_fireball"] = cfg["radiative_fireball"]
    return make_params(**kw)


_dp = C.POINTER(C.c_double)


def _p(a):
    return a.ctypes.data_as(_dp)


class CpuLib:
    """Loader for a CPU checker exporting <prefix>_flux_density_grid / _flux_density / _flux / _details."""

    def __init__(self, path, prefix):
        self.lib = C.CDLL(path)
        self.prefix = prefix
        f = getattr(self.lib, prefix + "_flux_density_grid")
        f.argtypes = [C.POINTER(ModelParams), _dp, C.c_int, _dp, C.c_int, _dp]
        f.restype = C.c_int
        f = getattr(self.lib, prefix + "_flux_density")
        f.argtypes = [C.POINTER(ModelParams), _dp, _dp, C.c_int, _dp]
        f.restype = C.c_int
        f = getattr(self.lib, prefix + "_flux")
        f.argtypes = [C.POINTER(ModelParams), _dp, C.c_int, C.c_double, C.c_double, C.c_int, _dp]
        f.restype = C.c_int
        f = getattr(self.lib, prefix + "_details")
        f.argtypes = [C.POINTER(ModelParams), C.c_double, C.c_double, C.POINTER(DetailsShape),
                      C.POINTER(DetailsOut), C.POINTER(_dp), C.c_int, C.POINTER(C.c_int), _dp, C.c_int]
        f.restype = C.c_int
        e = getattr(self.lib, prefix + "_last_error")
        e.restype = C.c_char_p

    def _check(self, rc):
        if rc != 0:
            raise ValueError(getattr(self.lib, self.prefix + "_last_error")().decode())

    def flux_density_grid(self, prm, t, nu):
        t = np.ascontiguousarray(t, dtype=np.float64)
        nu = np.ascontiguousarray(nu, dtype=np.float64)
        out = np.zeros((nu.size, t.size))
        self._check(getattr(self.lib, self.prefix + "_flux_density_grid")(
            C.byref(prm), _p(t), t.size, _p(nu), nu.size, _p(out)))
        return out

    def flux_density(self, prm, t, nu):
        t = np.ascontiguousarray(t, dtype=np.float64)
        nu = np.ascontiguousarray(nu, dtype=np.float64)
        out = np.zeros(t.size)
        self._check(getattr(self.lib, self.prefix + "_flux_density")(C.byref(prm), _p(t), _p(nu), t.size, _p(out)))
        return out

    def flux(self, prm, t, nu_min, nu_max, num_nu):
        t = np.ascontiguousarray(t, dtype=np.float64)
        out = np.zeros(t.size)
        self._check(getattr(self.lib, self.prefix + "_flux")(C.byref(prm), _p(t), t.size, nu_min, nu_max,
                                                              num_nu, _p(out)))
        return out

    def flux_components(self, prm, t, nu):
        """(fwd_sync, fwd_ssc) grids [nnu][nt]; fwd_ssc is zeros when Radiation.ssc is off."""
        fn = getattr(self.lib, self.prefix + "_flux_density_grid_components")
        fn.argtypes = [C.POINTER(ModelParams), _dp, C.c_int, _dp, C.c_int, _dp, _dp]
        fn.restype = C.c_int
        t = np.ascontiguousarray(t, dtype=np.float64)
        nu = np.ascontiguousarray(nu, dtype=np.float64)
        sync, ssc = np.zeros((nu.size, t.size)), np.zeros((nu.size, t.size))
        self._check(fn(C.byref(prm), _p(t), t.size, _p(nu), nu.size, _p(sync), _p(ssc)))
        return sync, ssc

    def flux_components4(self, prm, t, nu):
        """(fwd_sync, fwd_ssc, rvs_sync, rvs_ssc) grids [nnu][nt]; disabled components are zeros."""
        fn = getattr(self.lib, self.prefix + "_flux_density_grid_components4")
        fn.argtypes = [C.POINTER(ModelParams), _dp, C.c_int, _dp, C.c_int, C.POINTER(_dp)]
        fn.restype = C.c_int
        t = np.ascontiguousarray(t, dtype=np.float64)
        nu = np.ascontiguousarray(nu, dtype=np.float64)
        comps = [np.zeros((nu.size, t.size)) for _ in range(4)]
        arr = (_dp * 4)(*[_p(a) for a in comps])
        self._check(fn(C.byref(prm), _p(t), t.size, _p(nu), nu.size, arr))
        return tuple(comps)

    def flux_density_exposures(self, prm, t, nu, expo, num_points=10):
        fn = getattr(self.lib, self.prefix + "_flux_density_exposures")
        fn.argtypes = [C.POINTER(ModelParams), _dp, _dp, _dp, C.c_int, C.c_int, _dp]
        fn.restype = C.c_int
        t, nu, expo = (np.ascontiguousarray(a, dtype=np.float64) for a in (t, nu, expo))
        out = np.zeros(t.size)
        self._check(fn(C.byref(prm), _p(t), _p(nu), _p(expo), t.size, num_points, _p(out)))
        return out

    def profile(self, prm, kind, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.empty_like(x)
        fn = getattr(self.lib, self.prefix + "_profile")
        fn.argtypes = [C.POINTER(ModelParams), C.c_int, _dp, C.c_int, _dp]
        fn.restype = C.c_int
        self._check(fn(C.byref(prm), kind, _p(x), x.size, _p(out)))
        return out

    def grid_shape(self, prm, t_min, t_max):
        """(n_phi, n_theta, n_t, n_reps, symmetry, phi_mirrored) of the adaptive grid: the shape-only form of details()."""
        fn = getattr(self.lib, self.prefix + "_details")
        fn.argtypes = [C.POINTER(ModelParams), C.c_double, C.c_double, C.POINTER(DetailsShape),
                       C.POINTER(DetailsOut), C.POINTER(_dp), C.c_int, C.POINTER(C.c_int), _dp, C.c_int]
        fn.restype = C.c_int
        sh, npe = DetailsShape(), C.c_int(0)
        self._check(fn(C.byref(prm), t_min, t_max, C.byref(sh), None, None, 0, C.byref(npe), None, 0))
        return (sh.n_phi, sh.n_theta, sh.n_t, sh.n_reps, sh.symmetry, sh.phi_mirrored)

    EXTRA_NAMES = ["gamma_m", "gamma_c", "gamma_a", "gamma_M", "N_e", "column_den", "nu_m", "nu_c", "nu_a",
                   "nu_M", "I_nu_max", "lg2_t", "lg2_doppler", "lg2_geom", "lg2_I_probe", "injection_idx"]

    def details(self, prm, t_min, t_max, probe_lg2_nu=None, rvs=False):
        """Intermediates of the forward shock (or, rvs=True, of the reverse shock of a Model(rvs_rad=...))."""
        fn = getattr(self.lib, self.prefix + ("_details_rvs" if rvs else "_details"))
        fn.argtypes = [C.POINTER(ModelParams), C.c_double, C.c_double, C.POINTER(DetailsShape),
                       C.POINTER(DetailsOut), C.POINTER(_dp), C.c_int, C.POINTER(C.c_int), _dp, C.c_int]
        fn.restype = C.c_int
        sh = DetailsShape()
        nphi_eff = C.c_int(0)
        self._check(fn(C.byref(prm), t_min, t_max, C.byref(sh), None, None, 0, C.byref(nphi_eff), None, 0))
        nth, nt, nphi, npe = sh.n_theta, sh.n_t, sh.n_phi, nphi_eff.value
        probe = np.ascontiguousarray(probe_lg2_nu if probe_lg2_nu is not None else [], dtype=np.float64)
        d = {"phi": np.zeros(nphi), "theta": np.zeros(nth)}
        for n in ("t_src", "Gamma", "r", "t_comv", "B", "N_p", "Gamma_th"):
            d[n] = np.zeros((nth, nt))
        out = DetailsOut(*[_p(d[n]) for n, _ in DetailsOut._fields_])
        ex = {}
        for n in self.EXTRA_NAMES[:11]:
            ex[n] = np.zeros((nth, nt))
        for n in self.EXTRA_NAMES[11:14]:
            ex[n] = np.zeros((npe, nth, nt))
        ex["lg2_I_probe"] = np.zeros((nth, nt, max(probe.size, 1)))
        ex["injection_idx"] = np.zeros((nth, nt))
        arr = (_dp * 16)(*[_p(ex[n]) for n in self.EXTRA_NAMES])
        self._check(fn(C.byref(prm), t_min, t_max, C.byref(sh), C.byref(out), arr, 16, C.byref(nphi_eff),
                       _p(probe) if probe.size else None, probe.size))
        d.update(ex)
        d["shape"] = dict(n_phi=nphi, n_theta=nth, n_t=nt, n_reps=sh.n_reps, symmetry=sh.symmetry,
                          phi_mirrored=sh.phi_mirrored, n_phi_eff=npe)
        return d


def load_ref():
    path = os.path.join(ROOT, "oracle", "_ref", "libvag_ref.so")
    return CpuLib(path, "vag_ref") if os.path.exists(path) else None


def load_oracle(fast=False):
    path = os.path.join(ROOT, "oracle", "liboracle_fast.so" if fast else "liboracle.so")
    return CpuLib(path, "vag_oracle") if os.path.exists(path) else None
