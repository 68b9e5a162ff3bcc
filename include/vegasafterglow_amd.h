/*
 * vegasafterglow_amd.h -- C-ABI of the MI355X-native afterglow forward-model engine.
 *
 * This is the drop-in boundary for ONE path of VegasAfterglow: the forward-shock
 * synchrotron light-curve model behind Model.flux_density_grid / Model.flux_density /
 * Model.flux and the per-walker log-likelihood built on it.  Every entry point states
 * the reference interface (file:line under the VegasAfterglow tree) it replaces.
 *
 * Conventions
 *  - plain C, no exceptions cross the ABI: every call returns 0 on success or a negative
 *    VAG_E_* code; vag_last_error() returns a thread-local message for the last failure.
 *  - all physical inputs are in the reference's user units (CGS: erg, cm, s, Hz, rad);
 *    flux densities come back in erg cm^-2 s^-1 Hz^-1, band fluxes in erg cm^-2 s^-1
 *    (pybind/pymodel.cpp:368-371,506-508).
 *  - the *_dev entry points take DEVICE pointers (HBM resident inputs/outputs) and are
 *    ordered on the context's HIP stream.  They are not fire-and-forget: the host waits
 *    (spinning on a pinned, coherent summary the grid kernel's last wavefront publishes --
 *    no copy, no stream synchronisation) for the batch's layout before it can size the
 *    later launches, and returns with those launches queued; results are complete when
 *    the stream reaches that point.  The host-pointer forms stage through the context's
 *    buffers and synchronise before returning.
 *  - there is no CPU fallback: without a HIP device vag_ctx_create fails with
 *    VAG_E_NO_DEVICE.
 */
#ifndef VEGASAFTERGLOW_AMD_H
#define VEGASAFTERGLOW_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VAG_ABI_VERSION 13  /* v13: ticketed sharded calls (vag_loglike_shard_begin_dev / _end_dev); v12: vag_plan.n_ssc_slow_cells; v11: vag_plan.ode_rhs; a likelihood call's work tallies under vag_ctx_count_work (v10: VAG_E_INTERNAL; vag_ctx_set_stream orders the context's buffers across a change of stream) */

/* error codes */
#define VAG_OK 0
#define VAG_E_INVALID (-1)   /* bad argument: the reference raises ValueError (pybind/error_handling.h:31-69) */
#define VAG_E_NO_DEVICE (-2) /* no HIP device / HIP runtime failure at context creation */
#define VAG_E_HIP (-3)       /* a HIP call failed; message carries hipGetErrorString */
#define VAG_E_UNSUPPORTED (-4) /* configuration outside the accelerated path (today: a likelihood batch whose models differ in their flags) */
#define VAG_E_CAPACITY (-5)  /* grid larger than the engine's static limits */
#define VAG_E_NUMERIC (-6)   /* an ODE row could not find a step size (the reference throws odeint's step_adjustment_error) */
#define VAG_E_INTERNAL (-7)  /* an invariant of the engine failed (a defect, reported instead of a silently wrong flux) */

/* jet profiles: src/environment/jet.h:84-259 (TophatJet, GaussianJet, PowerLawJet),
 * math::two_component jet.h:421-437 via PyTwoComponentJet pybind/pymodel.cpp:130-146 */
#define VAG_JET_TOPHAT 0
#define VAG_JET_GAUSSIAN 1
#define VAG_JET_POWERLAW 2
#define VAG_JET_TWO_COMPONENT 3
/* Top-hat profile on the generic Ejecta with a constant magnetisation sigma0 (eps = E_iso, Gamma0 for theta <= theta_c):
 * the jet of the reference's tophat_sigma*_rs goldens (tests/python/golden/regenerate.py:141-149). */
#define VAG_JET_MAGNETIZED_TOPHAT 4
/* StepPowerLawJet(theta_c, E_iso, Gamma0, E_iso_w, Gamma0_w, k_e, k_g) and PowerLawWing(theta_c, E_iso_w, Gamma0_w, k_e, k_g):
 * pybind/pymodel.cpp:90-125, src/environment/jet.h:403-429 */
#define VAG_JET_STEP_POWERLAW 5
#define VAG_JET_POWERLAW_WING 6

/* Radiation flags.  VAG_FLAG_SSC / VAG_FLAG_KN = fwd_rad Radiation(ssc=, kn=): inverse-Compton cooling + SSC emission
 * (src/radiation/inverse-compton.*).  VAG_FLAG_RVS = Model(rvs_rad=Radiation(...)) i.e. the coupled forward+reverse
 * shock solve (src/dynamics/reverse-shock.tpp); VAG_FLAG_RVS_SSC / VAG_FLAG_RVS_KN = rvs_rad's ssc / kn. */
#define VAG_FLAG_SSC 1
#define VAG_FLAG_KN 2
#define VAG_FLAG_RVS 4
#define VAG_FLAG_RVS_SSC 8
#define VAG_FLAG_RVS_KN 16
/* jet(..., spreading=True): lateral expansion of the forward shock (forward-shock.tpp:36-40,78-84,110-116), per-row time
 * lattices and per-cell solid angles (observer.cpp:51-141). */
#define VAG_FLAG_SPREADING 32
/* jet(..., magnetar=Magnetar(L0, t0, q)): energy injection L0 (1 + t/t0)^-q inside theta_c (src/environment/jet.h:518-527,
 * pybind/pymodel.cpp:38-45); the jet then runs on the generic Ejecta profile forms of the reference. */
#define VAG_FLAG_MAGNETAR 64
/* Model(..., axisymmetric=False): full-circle phi grid, no mirror / on-axis shortcut (grid-refinement.h:671-689,
 * observer.cpp:215-222).  The named jets stay phi-independent, so every phi slice of the dynamics is the same solve. */
#define VAG_FLAG_NON_AXISYMMETRIC 128

/* media: src/environment/medium.h:50-133 (ISM, Wind with k_m = 2) */
#define VAG_MEDIUM_ISM 0
#define VAG_MEDIUM_WIND 1

/*
 * One forward model = the arguments of
 *   Model(jet, medium, Observer(lumi_dist, z, theta_obs), Radiation(eps_e, eps_B, p, xi_e),
 *         resolutions=(phi, theta, t), rtol, axisymmetric=True, radiative_fireball)
 * (pybind/pybind.cpp:384-422, pybind/pymodel.h:613-649), flattened to plain scalars.
 * All doubles; the two tags are int32.  Layout is fixed (272 bytes) and is what the
 * device kernels read straight from HBM.
 */
typedef struct vag_model_params {
    int32_t jet_type;    /* VAG_JET_* */
    int32_t medium_type; /* VAG_MEDIUM_* */
    /* jet (unused fields ignored by the profile) */
    double theta_c;  /* core half-opening angle [rad] */
    double E_iso;    /* isotropic-equivalent energy (core) [erg] */
    double Gamma0;   /* initial Lorentz factor (core) */
    double k_e;      /* PowerLawJet energy index */
    double k_g;      /* PowerLawJet Lorentz-factor index */
    double theta_w;  /* TwoComponentJet wing angle [rad] */
    double E_iso_w;  /* TwoComponentJet wing energy [erg] */
    double Gamma0_w; /* TwoComponentJet wing Lorentz factor */
    double duration; /* engine duration T0 [s] (shapes the reverse shock and its time lattice) */
    /* medium */
    double n_ism;  /* ISM number density [cm^-3]; Wind: ISM floor */
    double A_star; /* Wind parameter */
    double n0;     /* Wind inner plateau density [cm^-3]; +inf = none */
    /* observer */
    double lumi_dist; /* [cm] */
    double z;
    double theta_obs; /* [rad] */
    /* forward-shock radiation */
    double eps_e;
    double eps_B;
    double p;
    double xi_e;
    /* numerics */
    double phi_resol;   /* points per degree */
    double theta_resol; /* points per degree */
    double t_resol;     /* points per decade */
    double rtol;        /* ODE tolerance, (0,1) */
    int32_t radiative_fireball; /* 1 = radiative losses feed back on dynamics (default) */
    int32_t flags;              /* VAG_FLAG_* (Radiation(ssc=, kn=), pybind/pybind.cpp:368-377); other bits must be 0 */
    /* reverse-shock radiation (Model(rvs_rad=...), pybind/pymodel.h:613-629); read only when VAG_FLAG_RVS is set */
    double rvs_eps_e;
    double rvs_eps_B;
    double rvs_p;
    double rvs_xi_e;
    double sigma0; /* ejecta magnetisation, VAG_JET_MAGNETIZED_TOPHAT only (Ejecta(sigma0=...), pybind/pybind.cpp:224-272) */
    double k_m;    /* Wind density slope rho ~ r^-k_m (pybind/pymodel.cpp:153-186); 2 = the analytic Wind class */
    /* Magnetar(L0 [erg/s], t0 [s], q), read only with VAG_FLAG_MAGNETAR (pybind/pymodel.h:34-54) */
    double mag_L0;
    double mag_t0;
    double mag_q;
} vag_model_params;

/* Fill a params struct with the reference's defaults: Radiation xi_e = 1,
 * resolutions (0.06, 0.15, 6) (src/config/simulation-defaults.h:58-68), rtol 1e-6,
 * duration 1 s, n0 = +inf, radiative_fireball = 1, k_e = k_g = 2. */
void vag_params_default(vag_model_params* p);

/* Validate one params struct exactly like the reference's factories and Model ctor
 * (pybind/pymodel.cpp:47-186, pybind/pymodel.h:205-260,613-649).  0 or VAG_E_INVALID. */
int vag_params_validate(const vag_model_params* p);

const char* vag_last_error(void);
const char* vag_version(void);
/* Developer / test hooks are VAG_* environment variables (DESIGN.md names them).  The library reads the process environment once, inside
 * the first API call, and never on a call path afterwards (ABI v13; a thread pool may drive one context while another thread calls
 * setenv).  A process that changes such a variable later -- the test-suite does -- calls this to have it read again; not to be called
 * while another thread is inside the library. */
void vag_reload_env_hooks(void);
int vag_abi_version(void);

/* Number of visible HIP devices (0 when none / runtime missing). */
int vag_device_count(void);

/* ABI v12: bytes of device memory the library holds in this process, over all contexts.  A context's buffers only grow, to what the
 * largest request so far needed (+25 %), and are freed by vag_ctx_destroy: in a sampler's loop the figure stops moving after the first
 * calls (tests/test_gpu_fullsize.py holds it to that). */
long long vag_device_bytes_in_use(void);

/* ---- engine context: one per (process, device); owns stream + workspace in HBM ---- */
typedef struct vag_ctx vag_ctx;

int vag_ctx_create(int device, vag_ctx** out);
void vag_ctx_destroy(vag_ctx* ctx);
/* Use an external HIP stream (hipStream_t passed as void*); NULL = the context's own (non-blocking) stream.
 * The legacy default stream has the handle 0 and cannot be told apart from NULL: pass VAG_STREAM_LEGACY_DEFAULT for it
 * (PyTorch's default `torch.cuda.current_stream()` IS that stream: its `.cuda_stream` is 0). */
#define VAG_STREAM_LEGACY_DEFAULT ((void*)1)
/* Stream lifetime (ABI v11): a caller-owned stream must stay alive for the duration of every engine call made on it; it may be destroyed
 * between calls, before it is handed back -- the engine never touches the stream it leaves (the hand-off event that orders the
 * context's scratch buffers across streams is recorded at the end of each device-resident call, while the stream is known to be
 * alive).  VAG_E_HIP if the new stream cannot be made to wait for that event. */
int vag_ctx_set_stream(vag_ctx* ctx, void* hip_stream);
/* The value vag_ctx_set_stream would take to select the stream the context is on now (NULL = its own): lets a caller that
 * borrows the context for one call on another stream put the previous one back (ABI v9). */
int vag_ctx_get_stream(vag_ctx* ctx, void** out);
int vag_ctx_synchronize(vag_ctx* ctx);

/* Static per-model capacity limits of the device grids (rows/time nodes). */
typedef struct vag_limits {
    int32_t max_theta; /* theta nodes per model */
    int32_t max_phi;   /* phi nodes per model */
    int32_t max_time;  /* time-lattice nodes per row */
    int32_t max_nu;    /* frequencies per launch (grids with more are chunked inside the engine; a band integrates at most this many) */
} vag_limits;
void vag_get_limits(vag_limits* out);

/*
 * Model.flux_density_grid(t[nt] ascending, nu[nnu]) -> total[nnu][nt]
 * (pybind/pybind.cpp:424, pybind/pymodel.cpp:498-514, src/core/observer.h:355-445),
 * batched over nb independent models sharing (t, nu).  out is [nb][nnu][nt] row-major.
 */
int vag_flux_density_grid_batch(vag_ctx* ctx, const vag_model_params* params, int nb, const double* t, int nt,
                                const double* nu, int nnu, double* out);

/* Same request with the components kept apart: out_sync = FluxDict.fwd.sync, out_ssc = FluxDict.fwd.ssc (zeros when
 * Radiation.ssc is off), each [nb][nnu][nt] (pybind/pybind.cpp:472-483).  nt * nnu <= 4096. */
int vag_flux_density_grid_components_batch(vag_ctx* ctx, const vag_model_params* params, int nb, const double* t, int nt,
                                           const double* nu, int nnu, double* out_sync, double* out_ssc);

/* All four FluxDict components {fwd.sync, fwd.ssc, rvs.sync, rvs.ssc} (pybind/pybind.cpp:472-483), each [nb][nnu][nt];
 * NULL entries of out4 are skipped, disabled components come back as zeros.  nt * nnu <= 4096. */
int vag_flux_density_grid_components4_batch(vag_ctx* ctx, const vag_model_params* params, int nb, const double* t, int nt,
                                            const double* nu, int nnu, double* const* out4);

/*
 * Model.flux_density(t[n] ascending, nu[n]) -> total[n]
 * (pybind/pybind.cpp:427, pybind/pymodel.cpp:373-389, src/core/observer.h:447-538),
 * batched: out is [nb][n].
 */
int vag_flux_density_batch(vag_ctx* ctx, const vag_model_params* params, int nb, const double* t, const double* nu,
                           int n, double* out);
/* The same series with FluxDict's components apart (pymodel.cpp:373-389): out4[i] != NULL receives component i of
 * {fwd.sync, fwd.ssc, rvs.sync, rvs.ssc} as [nb][n]; components the model does not enable come back as zeros. */
int vag_flux_density_components4_batch(vag_ctx* ctx, const vag_model_params* params, int nb, const double* t, const double* nu,
                                       int n, double* const* out4);

/*
 * Model.flux(t[nt], nu_min, nu_max, num_nu) -> band flux[nt]
 * (pybind/pybind.cpp:430, pybind/pymodel.cpp:391-410, src/core/observer.h:555-567,
 * Boole weights src/core/quadrature.h:153-196); out is [nb][nt].
 */
int vag_flux_batch(vag_ctx* ctx, const vag_model_params* params, int nb, const double* t, int nt, double nu_min,
                   double nu_max, int num_nu, double* out);
/* Same with the components apart: out_sync = fwd.sync, out_ssc = fwd.ssc, each [nb][nt]. */
int vag_flux_components_batch(vag_ctx* ctx, const vag_model_params* params, int nb, const double* t, int nt,
                              double nu_min, double nu_max, int num_nu, double* out_sync, double* out_ssc);
/* ... and all four components, each [nb][nt] (NULL entries skipped). */
int vag_flux_components4_batch(vag_ctx* ctx, const vag_model_params* params, int nb, const double* t, int nt,
                               double nu_min, double nu_max, int num_nu, double* const* out4);

/* Device-pointer forms: params/t/nu/out are HBM addresses; stream-ordered on the context stream (see the note at the top). */
int vag_flux_density_grid_batch_dev(vag_ctx* ctx, const vag_model_params* d_params, int nb, const double* d_t,
                                    int nt, const double* d_nu, int nnu, double* d_out);
int vag_flux_density_batch_dev(vag_ctx* ctx, const vag_model_params* d_params, int nb, const double* d_t,
                               const double* d_nu, int n, double* d_out);

/*
 * Batched log-likelihood: the seam emcee's vectorized log_prob_batch calls
 * (VegasAfterglow/fitting/samplers.py:61-91 -> fitter.py:503-533).
 *
 * Point data (sorted by time, weights normalised to sum N as in fitter.py:407-451):
 *   chi2 = sum_i w_i ((ln F_obs_i - ln max(F_mod_i, 1e-300)) / (err_i / F_obs_i))^2
 *   loglike = -0.5 chi2; non-finite chi2 -> -inf (samplers.py:61-70).
 * The transformer (fitting/utils.py:110-135) is expressed as a slot map: free
 * parameter d of theta writes field slot[d] (a VAG_P_* index) of a copy of `base`,
 * as 10**theta when is_log[d] != 0.
 */
#define VAG_P_THETA_C 0
#define VAG_P_E_ISO 1
#define VAG_P_GAMMA0 2
#define VAG_P_K_E 3
#define VAG_P_K_G 4
#define VAG_P_THETA_W 5
#define VAG_P_E_ISO_W 6
#define VAG_P_GAMMA0_W 7
#define VAG_P_DURATION 8
#define VAG_P_N_ISM 9
#define VAG_P_A_STAR 10
#define VAG_P_N0 11
#define VAG_P_LUMI_DIST 12
#define VAG_P_Z 13
#define VAG_P_THETA_OBS 14
#define VAG_P_EPS_E 15
#define VAG_P_EPS_B 16
#define VAG_P_P 17
#define VAG_P_XI_E 18
#define VAG_P_COUNT 19 /* slots 0..18 are contiguous; the reverse-shock radiation parameters follow the numerics block */
#define VAG_P_RVS_EPS_E 24
#define VAG_P_RVS_EPS_B 25
#define VAG_P_RVS_P 26
#define VAG_P_RVS_XI_E 27
#define VAG_P_SIGMA0 28
#define VAG_P_K_M 29
#define VAG_P_MAG_L0 30
#define VAG_P_MAG_T0 31
#define VAG_P_MAG_Q 32

/* Not a Model field: the host-galaxy extinction A_V of ModelParams (types.py:77).  A free parameter with this slot only
 * scales the point-data model fluxes by exp(-A_V * ext_kernel[i]) (fitter.py:512-519). */
#define VAG_P_A_V 1000

/* One band-integrated data group of Fitter.add_flux (fitter.py:316-377): Model.flux(t, nu_min, nu_max, num_points) is
 * evaluated as its own request (own grid from its own time range), exactly like the reference's loop (fitter.py:524-531). */
typedef struct vag_band_obs {
    double nu_min, nu_max;  /* [Hz] */
    int32_t num_points;     /* Boole nodes across the band */
    int32_t n;              /* observations */
    const double* t;        /* [n] ascending [s] */
    const double* ln_flux;  /* [n] ln F_obs [erg/cm^2/s] */
    const double* ln_err;   /* [n] err / F_obs */
    const double* weight;   /* [n] */
} vag_band_obs;

typedef struct vag_fit_spec {
    vag_model_params base; /* fixed parameters + numerics */
    int32_t ndim;          /* number of free parameters (<= 16) */
    int32_t slot[16];      /* VAG_P_* target of each free parameter */
    int32_t is_log[16];    /* 1: value = 10**theta */
    int32_t n_data;        /* number of point observations (may be 0 when only band data is fitted) */
    int32_t pad;
    const double* t;        /* [n_data] observer times, ascending [s] */
    const double* nu;       /* [n_data] frequencies [Hz] */
    const double* ln_flux;  /* [n_data] ln F_obs */
    const double* ln_err;   /* [n_data] err/F_obs */
    const double* weight;   /* [n_data] normalised weights */
    /* ABI v6 */
    const double* ext_kernel;   /* [n_data] 0.4 ln10 k(lambda_rest) of the extinction law, or NULL (fitter.py:439-449) */
    double a_v_fixed;           /* A_V when it is not a free parameter (0 = no extinction) */
    int32_t n_bands;            /* band-integrated groups */
    int32_t pad2;
    const vag_band_obs* bands;  /* [n_bands] */
    /* ABI v7: the bounds mask and the priors of log_prob_batch (fitting/samplers.py:72-91) on the device.
     * use_priors = 0: the call returns ln L for every walker (ABI v6 behaviour).
     * use_priors = 1: a walker with any theta[d] outside [lower[d], upper[d]] is NOT evaluated and scores -inf; the others
     * score ln L + sum_d ln prior_d(theta[d]) with prior_kind[d] one of VAG_PRIOR_* acting on the SAMPLER-space value
     * (bilby.core.prior.Uniform / Gaussian / LogUniform.ln_prob; params.py:209-227 builds Uniform(lower, upper) by default). */
    int32_t use_priors;
    int32_t pad3;
    double lower[16], upper[16];
    int32_t prior_kind[16];
    double prior_a[16]; /* GAUSSIAN: mu;    LOG_UNIFORM: minimum (> 0); UNIFORM: unused (the bounds are the support) */
    double prior_b[16]; /* GAUSSIAN: sigma; LOG_UNIFORM: maximum */
} vag_fit_spec;
#define VAG_PRIOR_UNIFORM 0     /* -ln(upper - lower) */
#define VAG_PRIOR_GAUSSIAN 1    /* -(x - mu)^2 / (2 sigma^2) - ln(sigma sqrt(2 pi)) */
#define VAG_PRIOR_LOG_UNIFORM 2 /* -ln(x ln(max / min)) for min <= x <= max, else -inf */
#define VAG_PRIOR_NONE 3        /* 0 inside the bounds: the caller adds its own ln prior for this parameter */
#define VAG_PRIOR_UNIFORM_RANGE 4 /* ABI v9: bilby Uniform(minimum = prior_a, maximum = prior_b) with its OWN support:
                                   * -ln(max - min) for min <= x <= max, else -inf (narrower or wider than the bounds) */

/* theta is [nb][ndim] (host); out is [nb] log-likelihoods (host).  Walkers whose
 * transformed parameters fail validation get -inf, like eval_one's except branch. */
int vag_loglike_batch(vag_ctx* ctx, const vag_fit_spec* spec, const double* theta, int nb, int ndim, double* out);

/* Same with theta/out in HBM.  The data arrays of spec are host pointers: their CONTENT is hashed on every call and they are
 * uploaded (one pinned staging copy) only when it differs from the previous call's, so a sampler loop moves no data.
 * Stream-ordered on the context stream, but host-blocking: the call returns after the batch's device plan has been read back
 * (see DESIGN.md "host synchronisation"). */
int vag_loglike_batch_dev(vag_ctx* ctx, const vag_fit_spec* spec, const double* d_theta, int nb, int ndim,
                          double* d_out);

/* Relative cost of every model of the last batch call on this context, into HBM: d_cost[nb] = n_theta * n_phi_eff * n_t, the
 * (theta, phi, t) cell count the equal-arrival-time integration walks (0 for a model that was not evaluated).  Stream-ordered,
 * no host synchronisation.  A sharded sampler balances the next call's walker blocks with it: walker cost varies 8x over a
 * prior box because every walker builds its own adaptive grid (fitter.py:503-533). */
int vag_last_model_costs_dev(vag_ctx* ctx, int nb, double* d_cost);

/*
 * ABI v9 -- one rank's share of a sharded log_prob_batch (the reference spreads eval_one over a thread pool,
 * VegasAfterglow/fitting/samplers.py:72-91; here the walkers are spread over the GPUs of a node, one process per GPU).
 * Every rank holds the SAME d_theta_all[nb_all][ndim] and calls, in this order and without any host work in between:
 *
 *   vag_loglike_shard_dev        deals the walkers on the device -- ranked by the cost the engine reported for them in the previous
 *                                finished call of this size (stable, descending), position q = sweep * world + k goes to rank k on
 *                                even sweeps and world-1-k on odd ones, so every rank gets ceil(nb_all / world) slots (`per`) of
 *                                near-equal total cost --, evaluates this rank's walkers and writes d_block[per][2] =
 *                                {ln L (+ ln prior with use_priors), cost}; padding slots carry {NaN, 0};
 *   <the caller's all-gather>    d_gathered[world * per][2] = the ranks' blocks in rank order (RCCL / any transport);
 *   vag_loglike_shard_finish_dev scatters d_gathered back into walker order, d_out[nb_all], and keeps the gathered costs for the
 *                                next deal (a walker that was not evaluated, cost 0, is assumed average).
 *
 * The deal is a function of the gathered costs only, so all ranks compute the same one without talking.  Stream-ordered on the
 * context stream like vag_loglike_batch_dev; rank / world are the caller's (no communicator is touched here).
 * ABI v11: the deal of a call in flight is kept per (nb_all, world, spec content), so other sharded calls may run on the context between
 * a call's two halves -- a caller need not, and should not, hold a process-local lock across its collective.
 * ABI v13: a call in flight is NAMED.  vag_loglike_shard_begin_dev is vag_loglike_shard_dev plus a ticket (never 0), and
 * vag_loglike_shard_end_dev finishes exactly that call: calls of equal shape may finish in any order, and several calls of the SAME
 * fit may be in flight (each holds its own deal; up to eight in flight on a context, of up to four different (fit, nb_all, world) keys).
 * vag_loglike_shard_end_dev(ctx, ticket, NULL, nb_all, world, NULL) abandons a call (the caller's collective failed): its slot is
 * released and the fit's costs stay as they were.  The unticketed pair stays: its finish takes the OLDEST call in flight of that
 * shape, which is right only while calls of equal shape finish in the order they were dealt, and an unticketed deal of a fit that
 * already has one in flight replaces it.
 */
int vag_loglike_shard_dev(vag_ctx* ctx, const vag_fit_spec* spec, const double* d_theta_all, int nb_all, int ndim, int rank,
                          int world, double* d_block);
int vag_loglike_shard_finish_dev(vag_ctx* ctx, const double* d_gathered, int nb_all, int world, double* d_out);
int vag_loglike_shard_begin_dev(vag_ctx* ctx, const vag_fit_spec* spec, const double* d_theta_all, int nb_all, int ndim, int rank,
                                int world, double* d_block, uint64_t* ticket);
int vag_loglike_shard_end_dev(vag_ctx* ctx, uint64_t ticket, const double* d_gathered, int nb_all, int world, double* d_out);
/* For inspection (either pointer may be NULL): d_table[world * per] = walker of every (rank, slot) in the last deal, -1 = padding;
 * d_cost[nb_all] = the gathered costs the NEXT deal will rank by (after a finished call). */
int vag_loglike_shard_state_dev(vag_ctx* ctx, int nb_all, int world, int32_t* d_table, double* d_cost);

/*
 * Model.details(t_min, t_max) intermediates for ONE model (pybind/pymodel.cpp:315-348):
 * grid sizes first, then arrays copied into caller buffers (any pointer may be NULL).
 *   phi[n_phi], theta[n_theta], t_src[n_theta][n_t] (engine frame, s),
 *   Gamma, r (cm), t_comv (s), B (G), N_p, Gamma_th : [n_theta][n_t]
 */
typedef struct vag_details_shape {
    int32_t n_phi, n_theta, n_t, n_reps;
    int32_t symmetry;     /* 0 structured, 1 phi_symmetric, 2 piecewise, 3 isotropic (src/core/mesh.h:55-60) */
    int32_t phi_mirrored; /* src/core/mesh.h:74-79 */
} vag_details_shape;

typedef struct vag_details_out {
    double* phi;
    double* theta;
    double* t_src;
    double* Gamma;
    double* r;
    double* t_comv;
    double* B;
    double* N_p;
    double* Gamma_th;
} vag_details_out;

int vag_details(vag_ctx* ctx, const vag_model_params* params, double t_min, double t_max, vag_details_shape* shape,
                const vag_details_out* out);
/* Same protocol for the reverse shock of a Model(rvs_rad=...) (Model.details().rvs, pybind/pymodel.cpp:315-348). */
int vag_details_rvs(vag_ctx* ctx, const vag_model_params* params, double t_min, double t_max, vag_details_shape* shape,
                    const vag_details_out* out);
/* ShockDetails' electron / photon arrays of the forward (rvs = 0) or reverse (rvs = 1) shock, each [n_theta][n_t] with the
 * shape vag_details reports (save_electron_details / save_photon_details, pybind/pymodel.cpp:236-290):
 *   arrays[0..10] = gamma_m, gamma_c, gamma_a, gamma_M, N_e, nu_m [Hz], nu_c, nu_a, nu_M, I_nu_max [erg/cm^2/s/Hz], theta.
 * NULL entries are skipped.  With Radiation(ssc=True) these are the inverse-Compton-cooled values. */
int vag_details_radiation(vag_ctx* ctx, const vag_model_params* params, double t_min, double t_max, int rvs,
                          double* const* arrays);
/* ABI v11: SynElectrons::regime of every (theta, t) cell -- determine_regime (src/radiation/synchrotron.cpp:45-60): 1 ... 6 by the
 * ordering of gamma_a, gamma_c, gamma_m, 0 = none -- regime[n_theta][n_t] (shape as vag_details reports it). */
int vag_details_regime(vag_ctx* ctx, const vag_model_params* params, double t_min, double t_max, int rvs, int32_t* regime);
/* ShockDetails.t_obs [s] and .Doppler of every (phi, theta, k) cell (pybind/pymodel.cpp:296-298; shared by the forward
 * and the reverse shock, which ride the same contact discontinuity): [n_phi_eff][n_theta][n_t] with the shape
 * vag_details reports and n_phi_eff = Observer::eff_phi_grid (1 for an on-axis axisymmetric model).  Call with
 * t_obs = doppler = NULL to query n_phi_eff. */
int vag_details_eat(vag_ctx* ctx, const vag_model_params* params, double t_min, double t_max, int* n_phi_eff, double* t_obs,
                    double* doppler);
/* Model.jet_E_iso(phi, theta), Model.jet_Gamma0(phi, theta), Model.medium(phi, theta, r) (pybind.cpp:441-448,
 * pymodel.cpp:572-594) for the named, phi-independent profiles: kind 0 -> isotropic-equivalent energy [erg] at theta[n],
 * 1 -> initial Lorentz factor at theta[n], 2 -> mass density [g/cm^3] at radius r[n] [cm]. */
int vag_profile_eval(vag_ctx* ctx, const vag_model_params* params, int kind, const double* x, int n, double* out);

/* Per-stage device timings (ms) of the last batch call, stage names follow the reference's
 * profiler (pybind/pymodel.h:877-953): grid, dynamics, syn_cells, sync_flux, reduce, total. */
typedef struct vag_stage_times {
    float grid_ms, dynamics_ms, cells_ms, flux_ms, reduce_ms, total_ms;
} vag_stage_times;
int vag_last_stage_times(vag_ctx* ctx, vag_stage_times* out);

/* Per-stage device time of the last call under the reference profiler's stage names (AFTERGLOW_PROFILE_SCOPE in
 * pybind/pymodel.h:877-953; Model.profile_data(), pybind.cpp:458-459), measured with HIP events around the kernels of each
 * stage.  Off by default (every scope costs two event records): vag_ctx_profile(ctx, 1) turns it on for the following calls.
 * What each name covers here, where kernels are fused differently from the reference's loops:
 *   dynamics      adaptive grid + blast-wave ODE (both shocks);
 *   EAT_grid      always 0: the equal-arrival-time logs are recomputed inside the flux kernels (counted in *_flux);
 *   syn_electrons vag_cells_kernel: electrons AND photons of every cell in one pass;
 *   cooling       inverse-Compton cooling recurrence (vag_ic_cooling_kernel);
 *   syn_photons   the photon rebuild from the cooled electrons (vag_photons_ic_kernel; 0 without SSC);
 *   sync_flux     synchrotron flux passes (both shocks) incl. their reductions; the fused synchrotron + SSC pass counts here;
 *   ic_photons    seed band + per-cell SSC spectrum tables;   ssc_flux  SSC flux passes;   total  first to last kernel. */
typedef struct vag_profile {
    double dynamics, EAT_grid, syn_electrons, syn_photons, cooling, sync_flux, ic_photons, ssc_flux, total; /* ms */
} vag_profile;
int vag_ctx_profile(vag_ctx* ctx, int enable);
int vag_last_profile(vag_ctx* ctx, vag_profile* out);

/* Work done by the last batch call, for roofline accounting (SURVEY.md section 8d units):
 *   eat_cells  = sum over models of (theta x phi_eff pairs) x n_t  -- (phi, theta, k) cells of Observer::observe
 *   spec_evals = eat_cells x nnu (grid) or 2 x pairs x n (series)  -- calls of SmoothPowerLawSyn::compute_log2_I_nu
 *   interps    = pairs x nt x nnu (grid) or pairs x n (series)     -- log-log interpolations + exp2 */
typedef struct vag_plan {
    int32_t n_models_ok; /* models whose grid fit the engine limits */
    int32_t n_rows;      /* ODE rows solved (representative theta rows) */
    int64_t n_cells;     /* (row, k) cells = photon parameter blocks */
    int64_t total_pairs; /* (theta, phi_eff) rows integrated by the flux kernel */
    int64_t eat_cells;
    int64_t spec_evals;
    int64_t interps;
    int32_t flux_blocks; /* workgroups of the flux kernel */
    int32_t pairs_per_block;
    int32_t n_models_invalid;  /* parameters rejected by validation (ValueError in the reference) */
    int32_t n_models_capacity; /* adaptive grid larger than the engine limits: NOT evaluated (NaN / -inf) */
    int32_t n_rows_failed;     /* ODE rows without an acceptable step after 500 rejections (error in the reference) */
    int32_t n_rows_gave_up;    /* ODE rows that hit the 100000-step cap or stalled (warning in the reference; row kept) */
    /* ABI v7, likelihood calls only, tallied over ALL passes (point data + every band group) of the last call */
    int32_t n_walkers_rejected;   /* walkers scored -inf: out of bounds, invalid parameters, grid over capacity, failed ODE row, SSC failure, non-finite chi2 */
    int32_t n_walkers_ssc_failed; /* of those: SSC tables over capacity or queried outside their clamped band */
    /* with vag_ctx_count_work(1): the SSC table build's work (ICPhoton::generate_spectrum, inverse-compton.h:529-607), summed over
     * both shocks: ic_terms = sum over cells of (electron-energy nodes x seed-frequency nodes), the accumulation's unit;
     * ic_nodes = sum over cells of (electron + seed + output lattice nodes), the set-up's unit */
    int64_t ic_terms, ic_nodes;
    /* ABI v8: models whose SSC tables were rebuilt over their full theoretical range because a flux pass queried them outside the
     * clamped band (ICPhoton::compute_log2_I_nu's self-healing path, inverse-compton.h:626-635); the pass was then repeated */
    int32_t n_models_ssc_rebuilt;
    /* ABI v11 (the v8 padding word): SSC passes of the last call that had to be repeated with EVERY cell's table because a flux pass
     * queried a cell the lazy selection (tables only for the cells a request's observation window touches) had skipped */
    int32_t n_ssc_all_cell_fallbacks;
    /* ABI v10: bytes of the pool that holds the SSC tables of one shock of the batch (each table as long as its own output lattice;
     * cells no (theta, phi) row queries have none) -- the largest table build of the last call */
    int64_t ic_pool_bytes;
    /* ABI v11, with vag_ctx_count_work(1): right-hand sides the forward-shock solver evaluated for the batch (ForwardShockEqn::operator(),
     * forward-shock.tpp:10-118; FSAL: six per step attempt + one per row); 0 for the solvers that carry no tally (reverse shock,
     * spreading, injection).  In the same mode a likelihood call's spec_evals / interps are tallied by the flux kernel itself
     * (boundary-spectrum evaluations actually formed, data points inside a row's lattice) instead of the 2 / 1 per (row, point) bound. */
    int64_t ode_rhs;
    /* ABI v12: SSC cells of the last call whose lattices exceed the wavefront-per-cell kernel's on-chip layout (more than 128 seed
     * frequencies, 64 electron energies or 192 output nodes) and took the general kernel with its arrays in HBM -- same algorithm, same
     * table.  A likelihood call does not wait for the count: vag_last_plan reads it then, for the call's last table build only */
    int64_t n_ssc_slow_cells;
    /* ABI v13, with vag_ctx_count_work(1): lane utilisation of the forward-shock solver's attempt loop -- live lanes summed over the
     * step attempts of every wavefront / lane slots those attempts occupied (64 per attempt).  A wavefront runs until its slowest row
     * is done (forward-shock.tpp:194-207 loops over rows one by one); the persistent kernel refills finished lanes from a row queue. */
    int64_t ode_lane_attempts, ode_lane_slots;
} vag_plan;
int vag_last_plan(vag_ctx* ctx, vag_plan* out); /* synchronises the stream to read the ODE row counters */
/* Instrumentation: when enabled, grid-flux launches tally the exact spec_evals / interps (window-clamped) with
 * one atomic per workgroup and a host sync; leave disabled in timed runs. */
int vag_ctx_count_work(vag_ctx* ctx, int enable);

/*
 * ABI v11 -- thread pools.  Every entry point that takes a context locks it for its own duration, so a context may be shared by the
 * threads of a pool (the reference releases the GIL in every compute method and its samplers map eval_one over a ThreadPoolExecutor with
 * one Model per thread, pybind/pybind.cpp:424-448, VegasAfterglow/fitting/samplers.py:59-70); calls are served one after the other.
 * The *_coalesced forms take ONE model, block the calling thread, and serve the calls that wait at the same time with the same request
 * (times, frequencies / band) as ONE batch call of the corresponding *_batch entry point: an unmodified thread-pool sampler gets batched
 * throughput.  `out` receives the total ([nnu][nt] / [n] / [nt]); or, with out == NULL, out4[i] != NULL receive the components as in the
 * *_components4_batch forms.  The caller's buffers must stay valid until its call returns; errors are per caller (a batch that fails as
 * a whole is repeated member by member).  Results: the series and band forms are the bits of a single call (their summation tree does
 * not depend on the batch); a grid call's fixed-order sums are laid out per batch, so its values may differ from a single call's in the
 * last bits (~1e-15 relative).
 */
int vag_ctx_coalesce(vag_ctx* ctx, int max_batch /* default 64 */, int wait_us /* the first caller waits this long for company: default 50 */);
int vag_ctx_coalesce_stats(vag_ctx* ctx, long long* calls, long long* batches); /* requests served / batch calls issued so far */
int vag_flux_density_grid_coalesced(vag_ctx* ctx, const vag_model_params* p, const double* t, int nt, const double* nu, int nnu,
                                    double* out, double* const* out4);
int vag_flux_density_coalesced(vag_ctx* ctx, const vag_model_params* p, const double* t, const double* nu, int n, double* out,
                               double* const* out4);
int vag_flux_coalesced(vag_ctx* ctx, const vag_model_params* p, const double* t, int nt, double nu_min, double nu_max, int num_nu,
                       double* out, double* const* out4);

#ifdef __cplusplus
}
#endif
#endif /* VEGASAFTERGLOW_AMD_H */
