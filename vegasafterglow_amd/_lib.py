"""ctypes binding of the C-ABI in include/vegasafterglow_amd.h.

The shared library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).  There is no
fallback of any kind: a missing library or a machine without a HIP device raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VAG_LIB_PATH") or os.path.join(_HERE, "libvegasafterglow_amd.so")  # override: kernel experiments

JET_TOPHAT, JET_GAUSSIAN, JET_POWERLAW, JET_TWO_COMPONENT, JET_MAGNETIZED_TOPHAT = 0, 1, 2, 3, 4
JET_STEP_POWERLAW, JET_POWERLAW_WING = 5, 6
MEDIUM_ISM, MEDIUM_WIND = 0, 1

VAG_OK, VAG_E_INVALID, VAG_E_NO_DEVICE, VAG_E_HIP, VAG_E_UNSUPPORTED, VAG_E_CAPACITY, VAG_E_NUMERIC, VAG_E_INTERNAL = 0, -1, -2, -3, -4, -5, -6, -7
FLAG_SSC, FLAG_KN, FLAG_RVS, FLAG_RVS_SSC, FLAG_RVS_KN, FLAG_SPREADING, FLAG_MAGNETAR = 1, 2, 4, 8, 16, 32, 64  # VAG_FLAG_* of include/vegasafterglow_amd.h
FLAG_NON_AXISYMMETRIC = 128

# VAG_P_* slots of the fit transformer (include/vegasafterglow_amd.h)
PARAM_SLOTS = {
    "theta_c": 0, "E_iso": 1, "Gamma0": 2, "k_e": 3, "k_g": 4, "theta_w": 5, "E_iso_w": 6, "Gamma0_w": 7,
    "tau": 8, "duration": 8, "n_ism": 9, "A_star": 10, "n0": 11, "lumi_dist": 12, "z": 13, "theta_v": 14,
    "theta_obs": 14, "eps_e": 15, "eps_B": 16, "p": 17, "xi_e": 18,
    "eps_e_r": 24, "eps_B_r": 25, "p_r": 26, "xi_e_r": 27,  # VAG_P_RVS_*: rvs_rad of Fitter(rvs_shock=True)
    "sigma0": 28, "k_m": 29, "L0": 30, "t0": 31, "q": 32,
}


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


class BandObs(C.Structure):
    _fields_ = [("nu_min", C.c_double), ("nu_max", C.c_double), ("num_points", C.c_int32), ("n", C.c_int32),
                ("t", C.POINTER(C.c_double)), ("ln_flux", C.POINTER(C.c_double)), ("ln_err", C.POINTER(C.c_double)),
                ("weight", C.POINTER(C.c_double))]


P_A_V = 1000  # VAG_P_A_V


class FitSpec(C.Structure):
    _fields_ = [
        ("base", ModelParams), ("ndim", C.c_int32), ("slot", C.c_int32 * 16), ("is_log", C.c_int32 * 16),
        ("n_data", C.c_int32), ("pad", C.c_int32),
        ("t", C.POINTER(C.c_double)), ("nu", C.POINTER(C.c_double)), ("ln_flux", C.POINTER(C.c_double)),
        ("ln_err", C.POINTER(C.c_double)), ("weight", C.POINTER(C.c_double)),
        ("ext_kernel", C.POINTER(C.c_double)), ("a_v_fixed", C.c_double), ("n_bands", C.c_int32), ("pad2", C.c_int32),
        ("bands", C.POINTER(BandObs)),
        # ABI v7: bounds mask + priors on the device
        ("use_priors", C.c_int32), ("pad3", C.c_int32), ("lower", C.c_double * 16), ("upper", C.c_double * 16),
        ("prior_kind", C.c_int32 * 16), ("prior_a", C.c_double * 16), ("prior_b", C.c_double * 16),
    ]


PRIOR_UNIFORM, PRIOR_GAUSSIAN, PRIOR_LOG_UNIFORM, PRIOR_NONE, PRIOR_UNIFORM_RANGE = 0, 1, 2, 3, 4


class DetailsShape(C.Structure):
    _fields_ = [("n_phi", C.c_int32), ("n_theta", C.c_int32), ("n_t", C.c_int32), ("n_reps", C.c_int32),
                ("symmetry", C.c_int32), ("phi_mirrored", C.c_int32)]


class DetailsOut(C.Structure):
    _fields_ = [(n, C.POINTER(C.c_double)) for n in
                ("phi", "theta", "t_src", "Gamma", "r", "t_comv", "B", "N_p", "Gamma_th")]


class StageTimes(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("grid_ms", "dynamics_ms", "cells_ms", "flux_ms", "reduce_ms", "total_ms")]


class Profile(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("dynamics", "EAT_grid", "syn_electrons", "syn_photons", "cooling", "sync_flux", "ic_photons",
                                          "ssc_flux", "total")]


class Plan(C.Structure):
    _fields_ = [("n_models_ok", C.c_int32), ("n_rows", C.c_int32), ("n_cells", C.c_int64), ("total_pairs", C.c_int64),
                ("eat_cells", C.c_int64), ("spec_evals", C.c_int64), ("interps", C.c_int64),
                ("flux_blocks", C.c_int32), ("pairs_per_block", C.c_int32),
                ("n_models_invalid", C.c_int32), ("n_models_capacity", C.c_int32),
                ("n_rows_failed", C.c_int32), ("n_rows_gave_up", C.c_int32),
                ("n_walkers_rejected", C.c_int32), ("n_walkers_ssc_failed", C.c_int32),
                ("ic_terms", C.c_int64), ("ic_nodes", C.c_int64), ("n_models_ssc_rebuilt", C.c_int32), ("n_ssc_all_cell_fallbacks", C.c_int32),
                ("ic_pool_bytes", C.c_int64), ("ode_rhs", C.c_int64), ("n_ssc_slow_cells", C.c_int64),
                ("ode_lane_attempts", C.c_int64), ("ode_lane_slots", C.c_int64)]


class Limits(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("max_theta", "max_phi", "max_time", "max_nu")]


EXPORTS = [
    "vag_params_default", "vag_params_validate", "vag_last_error", "vag_version", "vag_abi_version", "vag_reload_env_hooks",
    "vag_device_count", "vag_device_bytes_in_use", "vag_ctx_create", "vag_ctx_destroy", "vag_ctx_set_stream", "vag_ctx_get_stream", "vag_ctx_synchronize",
    "vag_get_limits", "vag_flux_density_grid_batch", "vag_flux_density_grid_components_batch", "vag_flux_components_batch",
    "vag_flux_density_grid_components4_batch", "vag_flux_components4_batch", "vag_flux_density_batch", "vag_flux_batch",
    "vag_flux_density_components4_batch", "vag_flux_density_grid_batch_dev", "vag_flux_density_batch_dev", "vag_loglike_batch", "vag_loglike_batch_dev",
    "vag_last_model_costs_dev", "vag_loglike_shard_dev", "vag_loglike_shard_finish_dev", "vag_loglike_shard_begin_dev", "vag_loglike_shard_end_dev", "vag_loglike_shard_state_dev", "vag_ctx_profile", "vag_last_profile", "vag_details", "vag_details_rvs", "vag_details_radiation", "vag_details_regime", "vag_details_eat", "vag_profile_eval", "vag_last_stage_times", "vag_last_plan", "vag_ctx_count_work",
    "vag_ctx_coalesce", "vag_ctx_coalesce_stats", "vag_flux_density_grid_coalesced", "vag_flux_density_coalesced", "vag_flux_coalesced",
]

_lib = None
_dp = C.POINTER(C.c_double)
_pp = C.POINTER(ModelParams)


def load():
    """Load the HIP engine; raises ImportError (no silent CPU path) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so); the engine links the system one (/opt/rocm/lib).
    # Both carry the same SONAME, so whichever is mapped first serves both -- and torch does not initialise on the system
    # runtime ("No HIP GPUs are available").  Streams and device pointers are shared with torch (vag_ctx_set_stream, the *_dev
    # entry points), so the process must run ONE runtime: torch's, mapped before the engine.
    try:
        import torch  # noqa: F401
    except ImportError:  # a torch-free process uses the system runtime
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the gfx950 engine first (python -c 'import __graft_entry__ as g; g.build()')")
    lib = C.CDLL(LIB_PATH)
    v = C.c_void_p
    lib.vag_last_error.restype = C.c_char_p
    lib.vag_version.restype = C.c_char_p
    lib.vag_reload_env_hooks.restype = None
    lib.vag_params_default.argtypes = [_pp]
    lib.vag_params_default.restype = None
    lib.vag_params_validate.argtypes = [_pp]
    lib.vag_device_bytes_in_use.restype = C.c_longlong
    lib.vag_ctx_create.argtypes = [C.c_int, C.POINTER(v)]
    lib.vag_ctx_destroy.argtypes = [v]
    lib.vag_ctx_destroy.restype = None
    lib.vag_ctx_set_stream.argtypes = [v, v]
    lib.vag_ctx_get_stream.argtypes = [v, C.POINTER(v)]
    lib.vag_ctx_synchronize.argtypes = [v]
    lib.vag_get_limits.argtypes = [C.POINTER(Limits)]
    lib.vag_get_limits.restype = None
    lib.vag_flux_density_grid_batch.argtypes = [v, _pp, C.c_int, _dp, C.c_int, _dp, C.c_int, _dp]
    lib.vag_flux_density_grid_components_batch.argtypes = [v, _pp, C.c_int, _dp, C.c_int, _dp, C.c_int, _dp, _dp]
    lib.vag_flux_density_grid_components4_batch.argtypes = [v, _pp, C.c_int, _dp, C.c_int, _dp, C.c_int, C.POINTER(_dp)]
    lib.vag_flux_density_components4_batch.argtypes = [v, _pp, C.c_int, _dp, _dp, C.c_int, C.POINTER(_dp)]
    lib.vag_flux_components4_batch.argtypes = [v, _pp, C.c_int, _dp, C.c_int, C.c_double, C.c_double, C.c_int, C.POINTER(_dp)]
    lib.vag_flux_density_batch.argtypes = [v, _pp, C.c_int, _dp, _dp, C.c_int, _dp]
    lib.vag_flux_batch.argtypes = [v, _pp, C.c_int, _dp, C.c_int, C.c_double, C.c_double, C.c_int, _dp]
    lib.vag_ctx_coalesce.argtypes = [v, C.c_int, C.c_int]
    lib.vag_ctx_coalesce_stats.argtypes = [v, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    lib.vag_flux_density_grid_coalesced.argtypes = [v, _pp, _dp, C.c_int, _dp, C.c_int, _dp, C.POINTER(_dp)]
    lib.vag_flux_density_coalesced.argtypes = [v, _pp, _dp, _dp, C.c_int, _dp, C.POINTER(_dp)]
    lib.vag_flux_coalesced.argtypes = [v, _pp, _dp, C.c_int, C.c_double, C.c_double, C.c_int, _dp, C.POINTER(_dp)]
    lib.vag_flux_components_batch.argtypes = [v, _pp, C.c_int, _dp, C.c_int, C.c_double, C.c_double, C.c_int, _dp, _dp]
    lib.vag_flux_density_grid_batch_dev.argtypes = [v, v, C.c_int, v, C.c_int, v, C.c_int, v]
    lib.vag_flux_density_batch_dev.argtypes = [v, v, C.c_int, v, v, C.c_int, v]
    lib.vag_loglike_batch.argtypes = [v, C.POINTER(FitSpec), _dp, C.c_int, C.c_int, _dp]
    lib.vag_loglike_batch_dev.argtypes = [v, C.POINTER(FitSpec), v, C.c_int, C.c_int, v]
    lib.vag_last_model_costs_dev.argtypes = [v, C.c_int, v]
    lib.vag_loglike_shard_dev.argtypes = [v, C.POINTER(FitSpec), v, C.c_int, C.c_int, C.c_int, C.c_int, v]
    lib.vag_loglike_shard_finish_dev.argtypes = [v, v, C.c_int, C.c_int, v]
    lib.vag_loglike_shard_begin_dev.argtypes = [v, C.POINTER(FitSpec), v, C.c_int, C.c_int, C.c_int, C.c_int, v, C.POINTER(C.c_uint64)]
    lib.vag_loglike_shard_end_dev.argtypes = [v, C.c_uint64, v, C.c_int, C.c_int, v]
    lib.vag_loglike_shard_state_dev.argtypes = [v, C.c_int, C.c_int, v, v]
    lib.vag_ctx_profile.argtypes = [v, C.c_int]
    lib.vag_last_profile.argtypes = [v, C.POINTER(Profile)]
    lib.vag_details.argtypes = [v, _pp, C.c_double, C.c_double, C.POINTER(DetailsShape), C.POINTER(DetailsOut)]
    lib.vag_details_rvs.argtypes = [v, _pp, C.c_double, C.c_double, C.POINTER(DetailsShape), C.POINTER(DetailsOut)]
    lib.vag_details_radiation.argtypes = [v, _pp, C.c_double, C.c_double, C.c_int, C.POINTER(_dp)]
    lib.vag_details_regime.argtypes = [v, _pp, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_int32)]
    lib.vag_details_eat.argtypes = [v, _pp, C.c_double, C.c_double, C.POINTER(C.c_int), _dp, _dp]
    lib.vag_profile_eval.argtypes = [v, _pp, C.c_int, _dp, C.c_int, _dp]
    lib.vag_last_stage_times.argtypes = [v, C.POINTER(StageTimes)]
    lib.vag_last_plan.argtypes = [v, C.POINTER(Plan)]
    lib.vag_ctx_count_work.argtypes = [v, C.c_int]
    _lib = lib
    return lib


def torch_stream_handle(stream):
    """The void* vag_ctx_set_stream takes for a torch.cuda.Stream: its HIP handle, or VAG_STREAM_LEGACY_DEFAULT for torch's default
    stream (handle 0 = the legacy default stream; a plain 0 would select the context's own stream and the engine would race
    with torch's kernels)."""
    return C.c_void_p(stream.cuda_stream or 1)


def last_error():
    return load().vag_last_error().decode()


def check(rc):
    """Map C-ABI error codes onto the reference's exception types (pybind/error_handling.h:12-27)."""
    if rc == VAG_OK:
        return
    msg = last_error()
    if rc == VAG_E_INVALID:
        raise ValueError(msg)
    if rc == VAG_E_NO_DEVICE:
        raise RuntimeError("vegasafterglow_amd needs a HIP device (no CPU path): " + msg)
    if rc == VAG_E_CAPACITY:
        raise ValueError("engine capacity exceeded: " + msg)
    if rc == VAG_E_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == VAG_E_NUMERIC:
        raise RuntimeError("ODE integration failed: " + msg)  # odeint's step_adjustment_error surfaces as RuntimeError
    if rc == VAG_E_INTERNAL:
        raise RuntimeError("vegasafterglow_amd internal error (please report): " + msg)
    raise RuntimeError(f"vegasafterglow_amd error {rc}: {msg}")


class _Hooks:
    """The engine's developer / test switches are VAG_* environment variables which the library reads ONCE (inside its first call) --
    never on a call path, where getenv would race with another thread's setenv.  A process that flips one later does it through this
    mapping: the variable is set (or removed) in os.environ and the library is told to read its environment again.  Not for use while
    another thread is inside the engine."""

    @staticmethod
    def _reload():
        if os.path.exists(LIB_PATH):
            load().vag_reload_env_hooks()

    def __setitem__(self, name, value):
        os.environ[name] = value
        self._reload()

    def __delitem__(self, name):
        del os.environ[name]
        self._reload()

    def pop(self, name, *default):
        value = os.environ.pop(name, *default)
        self._reload()
        return value

    def get(self, name, default=None):
        return os.environ.get(name, default)

    def __contains__(self, name):
        return name in os.environ


hooks = _Hooks()
