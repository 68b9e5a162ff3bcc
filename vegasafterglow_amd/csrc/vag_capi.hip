// vag_capi.hip -- C-ABI of the MI355X afterglow engine (include/vegasafterglow_amd.h).
// Host orchestration only: every number the library returns is produced by the gfx950 kernels in
// vag_grid_kernel.h / vag_kernels.h.  There is no CPU compute path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <map>
#include <mutex>
#include <shared_mutex>

#include "vag_ic_kernels.h"
#include "vag_kernels.h"
#include "vag_rs_kernels.h"
#include "vag_fit_rows.h"
#include "vag_grid_rows.h"

using namespace vag;

namespace {

thread_local std::string g_err;

int set_err(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}


// Developer / test hooks (VAG_* environment variables).  The process environment is read ONCE -- when the first hook is asked for, i.e.
// inside the first API call -- and again only by vag_reload_env_hooks(): no getenv on any call path afterwards (getenv races with
// setenv / putenv of other threads, and a thread pool drives one context from many threads).  A value's storage stays where
// it is until the next reload.
extern "C" char** environ;
struct EnvHooks {
    std::shared_mutex mu;
    std::map<std::string, std::string> kv;
    bool loaded = false;
    void load_locked() {
        kv.clear();
        for (char** e = environ; e && *e; ++e) {
            if (std::strncmp(*e, "VAG_", 4) != 0) continue;
            const char* eq = std::strchr(*e, '=');
            if (eq) kv.emplace(std::string(*e, eq - *e), std::string(eq + 1));
        }
        loaded = true;
    }
    const char* get(const char* name) {
        {
            std::shared_lock<std::shared_mutex> rd(mu);
            if (loaded) {
                auto it = kv.find(name);
                return it == kv.end() ? nullptr : it->second.c_str();
            }
        }
        std::unique_lock<std::shared_mutex> wr(mu);
        if (!loaded) load_locked();
        auto it = kv.find(name);
        return it == kv.end() ? nullptr : it->second.c_str();
    }
    void reload() {
        std::unique_lock<std::shared_mutex> wr(mu);
        load_locked();
    }
};
EnvHooks g_hooks;
inline const char* vag_hook(const char* name) { return g_hooks.get(name); }

#define HIPCHK(call)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return set_err(VAG_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// grow-only device buffer
// the grid kernel's three layouts (vag_common.h): 0 small (LDS, eight models per CU), 1 large (LDS, one per CU), 2 huge (scratch in HBM)
static int layout_theta(int level) { return level == 2 ? VAG_HUGE_THETA : (level == 1 ? VAG_MAX_THETA : VAG_GRID_THETA); }
static int layout_phi(int level) { return level == 2 ? VAG_HUGE_PHI : (level == 1 ? VAG_MAX_PHI : VAG_GRID_PHI); }
static int rowgeo_stride(int level) {  // doubles per model of the row-geometry records (vag_grid_kernel.h)
    return VAG_ROWGEO_HDR + 2 * layout_phi(level) + 4 * layout_theta(level);
}

// device memory this library holds in this process, all contexts (vag_device_bytes_in_use): every buffer in HBM is a DevBuf
static std::atomic<long long> g_device_bytes{0};
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) HIPCHK(hipFree(p));
        g_device_bytes -= (long long)cap;
        p = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 4 + 256;
        HIPCHK(hipMalloc(&p, want));
        cap = want;
        g_device_bytes += (long long)cap;
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        g_device_bytes -= (long long)cap;
        p = nullptr;
        cap = 0;
    }
    template <class T>
    T* as() const {
        return reinterpret_cast<T*>(p);
    }
};

struct HostBuf {  // pinned
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) HIPCHK(hipHostFree(p));
        p = nullptr;
        cap = 0;
        HIPCHK(hipHostMalloc(&p, bytes + bytes / 4 + 256, hipHostMallocDefault));
        cap = bytes + bytes / 4 + 256;
        return 0;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T>
    T* as() const {
        return reinterpret_cast<T*>(p);
    }
};

enum { PS_DYNAMICS = 0, PS_SYN_ELECTRONS, PS_SYN_PHOTONS, PS_COOLING, PS_SYNC_FLUX, PS_IC_PHOTONS, PS_SSC_FLUX, PS_COUNT };

// prepares log2 arrays and the observer-time extrema on the device (single workgroup)
__global__ void vag_prep_kernel(const double* __restrict__ t, int nt, const double* __restrict__ nu, int nnu,
                                double* __restrict__ lg2_t, double* __restrict__ lg2_nu, double* __restrict__ tminmax) {
    __shared__ double s_min[256], s_max[256];
    double lo = INFINITY, hi = -INFINITY;
    for (int i = threadIdx.x; i < nt; i += blockDim.x) {
        const double v = t[i];
        lo = fmin(lo, v);
        hi = fmax(hi, v);
        lg2_t[i] = log2(v * U_SEC);  // xt::log2(t_obs), observer.h:359
    }
    for (int i = threadIdx.x; i < nnu; i += blockDim.x) lg2_nu[i] = log2(nu[i] * U_HZ);
    s_min[threadIdx.x] = lo;
    s_max[threadIdx.x] = hi;
    __syncthreads();
    for (int off = blockDim.x / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            s_min[threadIdx.x] = fmin(s_min[threadIdx.x], s_min[threadIdx.x + off]);
            s_max[threadIdx.x] = fmax(s_max[threadIdx.x], s_max[threadIdx.x + off]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        tminmax[0] = s_min[0];
        tminmax[1] = s_max[0];
    }
}

}  // namespace

// ---- validation shared by host entry points; rules of pybind/pymodel.cpp:47-186, pymodel.h:205-260,613-649 ----
static bool finite_pos(double x) { return std::isfinite(x) && x > 0; }
static bool range_oi(double x, double lo, double hi) { return std::isfinite(x) && x > lo && x <= hi; }

static const char* validate_msg(const vag_model_params* p) {
    const double pi = 3.14159265358979323846;
    if (p->jet_type < 0 || p->jet_type > VAG_JET_POWERLAW_WING) return "unknown jet_type";
    if (p->jet_type == VAG_JET_MAGNETIZED_TOPHAT && !(std::isfinite(p->sigma0) && p->sigma0 >= 0))
        return "sigma0 must be finite and non-negative";
    if (p->medium_type < 0 || p->medium_type > VAG_MEDIUM_WIND) return "unknown medium_type";
    if (!range_oi(p->theta_c, 0.0, pi / 2)) return "theta_c must be in (0, pi/2]";
    if (p->jet_type != VAG_JET_POWERLAW_WING) {  // PowerLawWing has no core (pymodel.cpp:90-110)
        if (!finite_pos(p->E_iso)) return "E_iso must be positive and finite";
        if (!(std::isfinite(p->Gamma0) && p->Gamma0 > 1.0)) return "Gamma0 must be > 1";
    }
    if (!finite_pos(p->duration)) return "duration must be positive and finite";
    if (p->jet_type == VAG_JET_STEP_POWERLAW || p->jet_type == VAG_JET_POWERLAW_WING) {
        if (!finite_pos(p->E_iso_w)) return "E_iso_w must be positive and finite";
        if (!(std::isfinite(p->Gamma0_w) && p->Gamma0_w > 1.0)) return "Gamma0_w must be > 1";
    }
    if (p->jet_type == VAG_JET_POWERLAW || p->jet_type == VAG_JET_STEP_POWERLAW || p->jet_type == VAG_JET_POWERLAW_WING) {
        if (!finite_pos(p->k_e)) return "k_e must be positive and finite";
        if (!finite_pos(p->k_g)) return "k_g must be positive and finite";
    }
    if (p->jet_type == VAG_JET_TWO_COMPONENT) {
        if (!range_oi(p->theta_w, 0.0, pi / 2)) return "theta_w must be in (0, pi/2]";
        if (!(p->theta_w > p->theta_c)) return "theta_w (wing angle) must be greater than theta_c (core angle)";
        if (!finite_pos(p->E_iso_w)) return "E_iso_w must be positive and finite";
        if (!(std::isfinite(p->Gamma0_w) && p->Gamma0_w > 1.0)) return "Gamma0_w must be > 1";
    }
    if (p->medium_type == VAG_MEDIUM_ISM) {
        if (!(std::isfinite(p->n_ism) && p->n_ism >= 0)) return "n_ism must be non-negative and finite";
    } else {
        if (!finite_pos(p->A_star)) return "A_star must be positive and finite";
        if (!(std::isfinite(p->n_ism) && p->n_ism >= 0)) return "n_ism must be non-negative and finite";
        if (!(p->n0 > 0)) return "n0 must be > 0 (or +inf for no floor)";
        if (!finite_pos(p->k_m)) return "k_m must be positive and finite";
    }
    if (!finite_pos(p->lumi_dist)) return "lumi_dist must be positive and finite";
    if (!(std::isfinite(p->z) && p->z >= 0)) return "z must be non-negative and finite";
    if (!(std::isfinite(p->theta_obs) && p->theta_obs >= 0 && p->theta_obs <= pi)) return "theta_obs must be in [0, pi]";
    if (!range_oi(p->eps_e, 0.0, 1.0)) return "eps_e must be in (0, 1]";
    if (!range_oi(p->eps_B, 0.0, 1.0)) return "eps_B must be in (0, 1]";
    if (!range_oi(p->xi_e, 0.0, 1.0)) return "xi_e must be in (0, 1]";
    if (!(std::isfinite(p->p) && p->p > 1.0)) return "p must be > 1";
    if (!(std::isfinite(p->rtol) && p->rtol > 0 && p->rtol < 1)) return "rtol must be in (0, 1)";
    if (!finite_pos(p->phi_resol) || !finite_pos(p->theta_resol) || !finite_pos(p->t_resol))
        return "resolutions must be positive and finite";
    if (p->flags & VAG_FLAG_MAGNETAR) {  // PyMagnetar ctor, pymodel.h:45-49
        if (!finite_pos(p->mag_L0) || !finite_pos(p->mag_t0) || !finite_pos(p->mag_q)) return "magnetar L0, t0, q must be positive and finite";
        if (p->jet_type == VAG_JET_POWERLAW_WING || p->jet_type == VAG_JET_MAGNETIZED_TOPHAT) return "this jet type takes no magnetar";
    }
    if (p->flags & ~(VAG_FLAG_SSC | VAG_FLAG_KN | VAG_FLAG_RVS | VAG_FLAG_RVS_SSC | VAG_FLAG_RVS_KN | VAG_FLAG_SPREADING |
                     VAG_FLAG_MAGNETAR | VAG_FLAG_NON_AXISYMMETRIC))
        return "unknown bits set in flags";
    if (p->flags & VAG_FLAG_RVS) {  // rvs_rad is a Radiation too (pymodel.h:241-260)
        if (!range_oi(p->rvs_eps_e, 0.0, 1.0)) return "rvs eps_e must be in (0, 1]";
        if (!range_oi(p->rvs_eps_B, 0.0, 1.0)) return "rvs eps_B must be in (0, 1]";
        if (!range_oi(p->rvs_xi_e, 0.0, 1.0)) return "rvs xi_e must be in (0, 1]";
        if (!(std::isfinite(p->rvs_p) && p->rvs_p > 1.0)) return "rvs p must be > 1";
    }
    return nullptr;
}

// g(a) = log2(1 + 2^-a) interpolation table for sp_fast (vag_device.h): interval i is centred on the node a = i/SP_PER_UNIT
// and holds the degree-(SP_NCOEF-1) interpolant at Chebyshev nodes, converted to monomials in
// tau = a*SP_PER_UNIT - i in [-1/2, 1/2], all in long double.
// Returns the max abs error measured on a dense check grid.
static double build_softplus_table(std::vector<double>& tab) {
    const int n = SP_NCOEF, NI = SP_INTERVALS, per = SP_PER_UNIT;
    const long double PI = 3.141592653589793238462643383279502884L;
    tab.assign((size_t)NI * n, 0.0);
    long double T[SP_NCOEF][SP_NCOEF] = {};
    T[0][0] = 1;
    T[1][1] = 1;
    for (int k = 2; k < n; ++k)
        for (int q = 0; q < n; ++q) T[k][q] = (q > 0 ? 2 * T[k - 1][q - 1] : 0) - T[k - 2][q];
    for (int i = 0; i < NI; ++i) {
        const long double ac = (long double)i / per, h = 1.0L / per;
        long double f[SP_NCOEF], c[SP_NCOEF], mono[SP_NCOEF] = {};
        for (int j = 0; j < n; ++j) {
            const long double t = cosl(PI * (2 * j + 1) / (2 * n));
            f[j] = log2l(1 + exp2l(-(ac + h / 2 * t)));
        }
        for (int k = 0; k < n; ++k) {
            long double s = 0;
            for (int j = 0; j < n; ++j) s += f[j] * cosl(k * PI * (2 * j + 1) / (2 * n));
            c[k] = 2 * s / n;
        }
        c[0] /= 2;
        for (int k = 0; k < n; ++k)
            for (int q = 0; q < n; ++q) mono[q] += c[k] * T[k][q];
        long double sc = 1;
        for (int q = 0; q < n; ++q) {
            tab[(size_t)i * n + q] = (double)(mono[q] * sc);
            sc *= 2;
        }
    }
    double maxerr = 0;
    for (int i = 0; i < NI - 1; ++i)
        for (int s = 0; s <= 32; ++s) {
            const double a = (i + s / 32.0) / per;
            const double t = std::fma(a, (double)per, SP_MAGIC);  // same node selection as sp_fast
            const int idx = std::min((int)(t - SP_MAGIC), NI - 1);
            const double tau = std::fma(a, (double)per, -(t - SP_MAGIC));
            const double* c = &tab[(size_t)idx * n];
            double p = c[n - 1];
            for (int q = n - 2; q >= 0; --q) p = std::fma(p, tau, c[q]);
            const long double truth = log2l(1 + exp2l(-(long double)a));
            maxerr = std::max(maxerr, std::fabs((double)(p - truth)));
        }
    return maxerr;
}

// log2_tab (vag_device.h): entry i = {rc = double(1 / c_i), -log2(rc)} with c_i the centre of the i-th 1/64 slice of [1, 2).
// Returns the max error of the same arithmetic on the host relative to max(1, |log2 x|).
static double append_log2_table(std::vector<double>& tab) {
    for (int i = 0; i < LOG_TAB_N; ++i) {
        const double rc = (double)(1.0L / (1.0L + (i + 0.5L) / LOG_TAB_N));
        tab.push_back(rc);
        tab.push_back((double)(-log2l((long double)rc)));
    }
    const double* lt = tab.data() + tab.size() - LOG_TAB_DOUBLES;
    double maxerr = 0;
    for (int s = 0; s < 200000; ++s) {
        const double x = std::ldexp(1.0 + (s + 0.37) / 200000.0, (s % 41) - 20);
        int e;
        const double m = 2 * std::frexp(x, &e);  // [1, 2)
        e -= 1;
        const int idx = std::min((int)((m - 1) * LOG_TAB_N), LOG_TAB_N - 1);
        const double r = std::fma(m, lt[2 * idx], -1.0);
        double q = std::fma(1.0 / 7, r, -1.0 / 6);
        q = std::fma(q, r, 0.2);
        q = std::fma(q, r, -0.25);
        q = std::fma(q, r, 1.0 / 3);
        q = std::fma(q, r, -0.5);
        const double ln_m = std::fma(q * r, r, r);
        const double got = std::fma(ln_m, LOG2E, (double)e + lt[2 * idx + 1]);
        const long double want = log2l((long double)x);
        maxerr = std::max(maxerr, (double)(fabsl(got - want) / std::max(1.0L, fabsl(want))));
    }
    return maxerr;
}

struct SeriesOcc {
    int waves;
    size_t lds;
    int wg;
};
struct CoalesceRequest;
struct vag_ctx {
    int device = 0;
    // Every entry point that takes the context locks it for its duration (ApiLock): Model methods release the GIL and are called
    // from thread pools, one Model per thread (pybind.cpp:424-448, samplers.py:59-70), so the library serialises GPU access itself.
    std::recursive_mutex api_mutex;
    // Concurrent single-model calls gathered into batch calls (vag_*_coalesced, opt-in): see the coalescer below
    std::mutex co_mutex;
    std::condition_variable co_cv;  // the leader's: arrivals wake the leader alone (a waiting member sleeps on its own request's)
    std::vector<CoalesceRequest*> co_queue;
    bool co_leader = false;
    int co_max_batch = 64, co_wait_us = 50;
    long long co_calls = 0, co_batches = 0;  // requests served / batch calls issued (vag_ctx_coalesce_stats)
    // Likelihood calls evaluate the walkers in descending order of the cost the SAME batch position had in the previous call
    // (dispatch is in batch order: expensive walkers last leave the GPU draining, and neighbours of similar cost diverge less):
    // d_order[cur] maps evaluation slot -> walker; ln L, costs and counters come back in walker order.  A walker's ln L does
    // not depend on its slot, so the result is the same bits.
    DevBuf d_order[2], d_cost_f;  // d_cost_f: per-slot cost of the last batch (written with the batch plan by the grid kernel)
    int order_cur = 0, order_nb = 0;  // order_nb: batch size d_order[order_cur] was computed for (0: none)
    // sharded likelihood calls (vag_loglike_shard_dev): the gathered per-walker costs of the last finished call (what the next
    // deal ranks by), the current deal (walker of every (rank, slot)), this rank's gathered theta rows / ln L
    // The costs belong to a (fit data, batch size, world) triple: two fitters that alternate on one context must not rank each
    // other's walkers, so a few of them are kept (least recently used is replaced), keyed by the fit spec's content hash.
    struct ShardCosts {
        DevBuf cost;
        uint64_t hash = 0;
        int nb = 0, world = 0;
        bool valid = false;     // a call of this key has finished: `cost` holds its gathered costs
        int in_flight = 0;      // calls of this key that wait for their finish (the entry is not replaced meanwhile)
        unsigned long long used = 0;  // clock of the last use
    };
    // A call in flight (between its deal and its finish) owns ONE flight slot: the deal's table and the ticket that names it (ABI v13).
    // Until v12 the table hung on the key and a finish took "the oldest pending deal of this shape": two calls of equal shape that
    // finished in the opposite order to their deals applied each other's tables, and two calls of the SAME key could not both finish.
    struct ShardFlight {
        DevBuf table;           // walker of every (rank, slot); kept after the finish for inspection until the slot is dealt again
        uint64_t ticket = 0;    // the deal's clock value: unique per context, never 0
        int key = -1, nb = 0, world = 0;
        bool in_use = false;
        bool unticketed = false;  // opened by vag_loglike_shard_dev: a new deal of the same key replaces it (the v9 behaviour)
    };
    ShardCosts shard_costs[4];
    ShardFlight shard_flights[8];
    unsigned long long shard_clock = 0;
    int shard_last = -1;                    // key entry of the last finished call (vag_loglike_shard_state_dev reports its costs)
    int shard_last_dealt = -1;              // flight slot of the last deal (... and its table)
    DevBuf d_shard_theta, d_shard_ll;
    // device-resident batches whose models differ in their Radiation / shock flags: regrouped by flags (flux_dev_by_flags)
    bool mixed_flags_seen = false;  // the last grid pass stopped on such a batch
    DevBuf d_mix_flags, d_mix_perm, d_mix_params, d_mix_out;
    bool order_next = false, order_active = false;  // the next / the last model-stage run is in evaluation-slot order
    const int* last_order = nullptr;                // ... and the order it used
    int grid_large = 0;       // layout level of the grid kernel in use: 1 / 2 after a recent batch needed more than the small / large layout holds
    int grid_large_idle = 0;  // consecutive batches that would have fitted the small one
    DevBuf d_gridscratch;     // level 2: the grid kernel's scratch arrays, one GridSharedHuge per model
    std::vector<SeriesOcc> series_occ;  // occupancy-query results of series launches seen so far
    DevBuf d_partial2, d_ssc2;  // fused synchrotron + SSC flux pass: second partial-grid buffer / second scratch output
    DevBuf d_bandidx;  // [512 band index per point | 8 first point of each band] for the shared-node / row-per-lane series paths
    int h_bandbuf[512 + 8] = {};  // host mirror of d_bandidx (skips the upload while a fit keeps its data)
    int h_bands_n = -1;
    int pending_bands = 0;  // set by the host-pointer entry points that know the frequencies; consumed by the next series call
    DevBuf d_sptab, d_workcount, d_knlut, d_icy, d_cellq, d_band, d_icstatus, d_icunclamp, d_ssc;
    // SSC tables: a header per cell, the lattice plan between the plan and the spectrum kernel, and the pool of tables (each as long
    // as its own output lattice; offsets handed out by vag_ic_plan_kernel, d_icused counts the doubles in use)
    DevBuf d_ichdr, d_icplan, d_icpool, d_icused, d_icslow /* records of the cells on the spectrum kernel's slow path */;
    DevBuf d_icneed;  // [cells] bytes: 1 = some (theta, phi) row's observation window touches the cell (vag_ic_band_kernel)
    bool count_work = false;
    bool ic_all_cells = false;    // this request's SSC tables are built for every cell (the lazy selection was caught with a hole, see check_ic_status)
    int batch_flags = 0;  // VAG_FLAG_* shared by every model of the current batch
    // reverse shock (VAG_FLAG_RVS): its own shock / electron / photon arrays and radiation parameters.  The radiation and
    // flux passes always read d_shock, d_cellpar, ...; select_emitter() swaps the reverse shock's buffers in and out.
    DevBuf d_shock_r, d_cellpar_r, d_celldet_r, d_icy_r, d_cellq_r, d_params_rvs, d_inj, d_comp;
    DevBuf d_fail;     // int[16], reset per batch by the grid kernel: ODE rows per status ([1] step underflow, [2] step cap, [3] stalled); three 64-bit tallies at
                       // [8..13] (vag_ctx_count_work: right-hand sides, live lanes over the attempts, lane slots of the attempts); [14] the row queue of the refill kernel
    DevBuf d_dynrec;   // [rows][DYN_ROWREC] start records of vag_dynamics_refill_kernel (vag_dyn_prep_kernel)
    int dyn_refill_wg_per_cu = 0;  // resident wavefronts of that kernel per CU (occupancy query, once)
    DevBuf d_cellgeo;  // spreading jets (VAG_FLAG_SPREADING): per-cell cos/sin(theta), log2|dcos|, shared by both shocks
    int cur_emitter = 0;                            // 0 forward, 1 reverse
    bool cur_ssc = false;                           // SSC switch of the selected emitter
    const vag_model_params* cur_params = nullptr;   // parameters its radiation / flux kernels read
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    hipEvent_t ev[8] = {};
    hipEvent_t ev_handoff = nullptr;  // orders the context's buffers across a change of stream (vag_ctx_set_stream)
    bool handoff_ready = false;       // ev_handoff marks the tail of the work queued on the current (caller-owned) stream
    bool handoff_dirty = false;       // a compute call has queued work since the event was last recorded (ApiLock records it on the way out)
    // inputs
    DevBuf d_params, d_t, d_nu, d_lg2t, d_lg2nu, d_tminmax, d_bandw, d_out;
    // grid results
    DevBuf d_meta, d_phi, d_theta, d_rep_of, d_rep_start, d_tdec, d_geo_th, d_geo_ph;
    HostBuf h_meta, h_off, h_plan;
    DevBuf d_plan;                // VagDevPlan of the current batch (vag_plan_kernel)
    DevBuf d_chunk;               // staging of chunked requests
    DevBuf d_icwork;              // work tallies of the SSC tables (vag_ic_plan_kernel, vag_ctx_count_work)
    VagDevPlan hint{};            // the last plan the host read back: sizes the next call of the same batch size in advance
    int hint_nb = 0;
    bool hint_valid = false;
    bool allow_spec = false;      // set by the entry points that end with finish_speculation()
    bool spec_pending = false;    // the current call was planned from the hint and has not been verified yet
    VagDevPlan* d_hplan = nullptr;  // device address of h_plan
    int plan_seq = 0;
    int layout_large = 0;  // vag_grid_kernel's layout level of the batch at hand
    DevBuf d_rowgeo;            // row-geometry records for the flux grid kernel, written by the grid kernel
    bool plan_counter_ready = false;
    int n_cus = 256;              // compute units of the device (persistent launches size themselves by it)
    // named-stage profiler (vag_ctx_profile): spans of (stage id, begin event, end event) recorded during a call
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev;
    struct ProfSpan { int id, e0, e1; };
    std::vector<ProfSpan> prof_spans;
    int prof_used = 0;
    int spec_cap_k = 0;
    int spec_margin_k = 2;        // lattice nodes of head room in a planned-ahead call; grows when a call had to be repeated
    bool meta_on_host = false;    // h_meta holds the current batch's grid results (copied on demand)
    // compact per-row / per-cell storage
    DevBuf d_row_off, d_cell_off, d_shock, d_cellpar, d_row_status, d_celldet, d_partial;
    // fit spec cache (upload_fit_spec): content hash of what d_fit holds, its size, where the prior block starts
    DevBuf d_fit, d_theta_in, d_valid, d_series_flux, d_chi2, d_bandobs, d_fitstat;
    HostBuf h_fit;
    uint64_t fit_hash = 0;
    size_t fit_doubles = 0, fit_prior_off = 0;
    bool fit_hash_valid = false;
    bool fit_stats_pending = false;  // d_fitstat of the last likelihood call not read back yet
    bool ic_need_reset = true;       // first SSC table build of a pass clears d_icstatus
    bool ic_soft_fail = false;       // likelihood calls: SSC table failures invalidate the walker instead of raising
    bool ic_slow_unread = false;     // the last table build did not wait for its counters: vag_last_plan reads the slow path's from HBM
    // plan of the last grid pass
    int nb = 0, n_rows = 0, max_k = 0, max_pairs = 0;
    long long n_cells = 0, total_pairs = 0, eat_cells = 0;
    int n_ok = 0;
    vag_plan plan{};
    vag_stage_times times{};
};

// One profiled stage: begin / end events on the context stream while the profiler is on (vag_ctx_profile)
struct StageScope {
    vag_ctx* c;
    int span = -1;
    StageScope(vag_ctx* ctx, int id) : c(ctx) {
        if (!c->prof_on) return;
        if (c->prof_used + 2 > (int)c->prof_ev.size()) {
            for (int i = 0; i < 16; ++i) {
                hipEvent_t e;
                if (hipEventCreate(&e) != hipSuccess) return;
                c->prof_ev.push_back(e);
            }
        }
        span = (int)c->prof_spans.size();
        c->prof_spans.push_back({id, c->prof_used, c->prof_used + 1});
        (void)hipEventRecord(c->prof_ev[c->prof_used], c->stream);
        c->prof_used += 2;
    }
    ~StageScope() {
        if (span >= 0) (void)hipEventRecord(c->prof_ev[c->prof_spans[span].e1], c->stream);
    }
};

extern "C" {

const char* vag_last_error(void) { return g_err.c_str(); }
struct ApiLock {  // (a null context is rejected by the entry point itself)
    std::recursive_mutex* m;
    vag_ctx* ctx;
    explicit ApiLock(vag_ctx* c);
    ~ApiLock();
    ApiLock(const ApiLock&) = delete;
};
const char* vag_version(void) { return "vegasafterglow_amd 0.1 (gfx950)"; }
void vag_reload_env_hooks(void) { g_hooks.reload(); }
int vag_abi_version(void) { return VAG_ABI_VERSION; }

void vag_params_default(vag_model_params* p) {
    if (!p) return;
    std::memset(p, 0, sizeof *p);
    p->jet_type = VAG_JET_TOPHAT;
    p->medium_type = VAG_MEDIUM_ISM;
    p->theta_c = 0.1;
    p->E_iso = 1e52;
    p->Gamma0 = 300;
    p->k_e = 2;
    p->k_g = 2;
    p->theta_w = 3.14159265358979323846 / 2;
    p->E_iso_w = 1e52;
    p->Gamma0_w = 300;
    p->duration = 1;
    p->n_ism = 1;
    p->A_star = 0;
    p->n0 = INFINITY;
    p->k_m = 2;
    p->lumi_dist = 1e28;
    p->z = 0;
    p->theta_obs = 0;
    p->eps_e = 0.1;
    p->eps_B = 0.01;
    p->p = 2.3;
    p->xi_e = 1;
    p->phi_resol = 0.06;
    p->theta_resol = 0.15;
    p->t_resol = 6;
    p->rtol = 1e-6;
    p->radiative_fireball = 1;
}

int vag_params_validate(const vag_model_params* p) {
    if (!p) return set_err(VAG_E_INVALID, "null model");
    const char* msg = validate_msg(p);
    return msg ? set_err(VAG_E_INVALID, "%s", msg) : VAG_OK;
}

int vag_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

long long vag_device_bytes_in_use(void) { return g_device_bytes.load(); }

void vag_get_limits(vag_limits* out) {
    out->max_theta = VAG_HUGE_THETA;
    out->max_phi = VAG_HUGE_PHI;
    out->max_time = VAG_MAX_TIME;
    out->max_nu = VAG_MAX_NU;
}

}  // extern "C"

// streams, events, kernel attributes and the request-independent tables of a fresh context
static int ctx_init(vag_ctx* c) {
    HIPCHK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    for (auto& e : c->ev) HIPCHK(hipEventCreate(&e));
    HIPCHK(hipEventCreateWithFlags(&c->ev_handoff, hipEventDisableTiming));
    HIPCHK(hipDeviceGetAttribute(&c->n_cus, hipDeviceAttributeMultiprocessorCount, c->device));
    // allow the flux kernels the full 160 KiB LDS of a gfx950 CU
    for (const void* fn : {reinterpret_cast<const void*>(vag_flux_grid_kernel<false, FLUX_SYN>),
                           reinterpret_cast<const void*>(vag_flux_grid_kernel<true, FLUX_SYN>),
                           reinterpret_cast<const void*>(vag_flux_grid_kernel<false, FLUX_SYN_IC>),
                           reinterpret_cast<const void*>(vag_flux_grid_kernel<false, FLUX_SSC>),
                           reinterpret_cast<const void*>(vag_flux_grid_kernel<false, FLUX_SYN, false, 256>),
                           reinterpret_cast<const void*>(vag_flux_grid_kernel<false, FLUX_SYN_IC, false, 256>),
                           reinterpret_cast<const void*>(vag_flux_grid_kernel<false, FLUX_SSC, false, 256>),
                           reinterpret_cast<const void*>(vag_flux_grid_kernel<false, FLUX_SYN, true>),
                           reinterpret_cast<const void*>(vag_flux_grid_kernel<false, FLUX_SYN_IC, true>),
                           reinterpret_cast<const void*>(vag_flux_grid_kernel<false, FLUX_SSC, true>),
                           reinterpret_cast<const void*>(vag_flux_grid_kernel<false, FLUX_FUSED>),
                           reinterpret_cast<const void*>(vag_flux_grid_kernel<false, FLUX_FUSED, false, 256>),
                           reinterpret_cast<const void*>(vag_flux_grid_kernel<false, FLUX_FUSED, true>),
                           reinterpret_cast<const void*>(vag_flux_series_kernel<FLUX_SYN, true>),
                           reinterpret_cast<const void*>(vag_flux_series_kernel<FLUX_SYN_IC, true>),
                           reinterpret_cast<const void*>(vag_flux_series_kernel<FLUX_SSC, true>)})
        HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    {   // Klein-Nishina cross-section table (ComptonSigmaLUT, src/radiation/inverse-compton.cpp:285-300): request-independent
        std::vector<double> lut(2 * KN_LUT_N);
        const double step = (KN_LG2_XMAX - KN_LG2_XMIN) / (double)(KN_LUT_N - 1);
        for (int i = 0; i < KN_LUT_N; ++i) {
            const double x = std::exp2(KN_LG2_XMIN + step * (double)i);
            double ratio;
            if (x < 1e-2) {
                ratio = 1 - 2 * x;
            } else if (x > 1e2) {
                ratio = 3. / 8 * (std::log(2 * x) + 0.5) / x;
            } else {
                const double l = std::log1p(2.0 * x), invx = 1.0 / x, invx2 = invx * invx;
                const double invt1 = 1.0 / (1.0 + 2.0 * x), invt1_2 = invt1 * invt1;
                ratio = 0.75 * ((1.0 + x) * invx2 * invx * (2.0 * x * (1.0 + x) * invt1 - l) + 0.5 * l * invx -
                                (1.0 + 3.0 * x) * invt1_2);
            }
            lut[i] = ratio;
            lut[KN_LUT_N + i] = std::log2(ratio);
        }
        if (c->d_knlut.ensure(sizeof(double) * lut.size())) return VAG_E_HIP;
        HIPCHK(hipMemcpy(c->d_knlut.p, lut.data(), sizeof(double) * lut.size(), hipMemcpyHostToDevice));
    }
    for (const void* fn : {reinterpret_cast<const void*>(vag_flux_series_kernel<FLUX_SYN>),
                           reinterpret_cast<const void*>(vag_flux_series_kernel<FLUX_SYN_IC>),
                           reinterpret_cast<const void*>(vag_flux_series_kernel<FLUX_SSC>),
                           reinterpret_cast<const void*>(vag_flux_series_kernel<FLUX_SYN, false, 1>),
                           reinterpret_cast<const void*>(vag_flux_series_kernel<FLUX_SYN_IC, false, 1>),
                           reinterpret_cast<const void*>(vag_flux_series_kernel<FLUX_SSC, false, 1>),
                           reinterpret_cast<const void*>(vag_flux_series_kernel<FLUX_SYN, true, 1>),
                           reinterpret_cast<const void*>(vag_flux_series_kernel<FLUX_SYN_IC, true, 1>),
                           reinterpret_cast<const void*>(vag_flux_series_kernel<FLUX_SSC, true, 1>),
                           reinterpret_cast<const void*>(vag_flux_series_kernel<FLUX_SYN, false, SERIES_MAX_SLOTS, true>),
                           reinterpret_cast<const void*>(vag_flux_series_kernel<FLUX_SYN_IC, false, SERIES_MAX_SLOTS, true>),
                           reinterpret_cast<const void*>(vag_flux_series_kernel<FLUX_SSC, false, SERIES_MAX_SLOTS, true>)})
        HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    {
        std::vector<double> tab;
        const double err = build_softplus_table(tab);
        if (!(err < 5e-14)) return set_err(VAG_E_HIP, "softplus table accuracy check failed: %.3e", err);
        const double lerr = append_log2_table(tab);  // [SP_TABLE_DOUBLES, SP_LDS_DOUBLES): log2_tab's {1/c, -log2(1/c)} pairs
        if (!(lerr < 4e-16)) return set_err(VAG_E_HIP, "log2 table accuracy check failed: %.3e", lerr);
        if (c->d_sptab.ensure(sizeof(double) * tab.size())) return VAG_E_HIP;
        HIPCHK(hipMemcpy(c->d_sptab.p, tab.data(), sizeof(double) * tab.size(), hipMemcpyHostToDevice));
    }
    return VAG_OK;
}

extern "C" {

ApiLock::ApiLock(vag_ctx* c) : m(c ? &c->api_mutex : nullptr), ctx(c) {
    if (m) m->lock();
}
// EVERY entry point leaves the hand-off event at the tail of a caller-owned stream, whichever way it returns (only the
// device-pointer forms did, so a host-buffer call that failed with kernels still queued left `handoff_ready` stale and the next
// vag_ctx_set_stream ordered the new stream behind an older event -- scratch buffers could be reused while the old stream wrote them).
// (Only on the way out of a call that queued work -- `handoff_dirty`, set where a call's model stage starts: the getters may be called
// after the caller destroyed its stream and must not touch it.)
ApiLock::~ApiLock() {
    if (ctx && ctx->handoff_dirty && ctx->stream != nullptr && ctx->stream != ctx->own_stream && ctx->ev_handoff) {
        ctx->handoff_ready = hipEventRecord(ctx->ev_handoff, ctx->stream) == hipSuccess;
        ctx->handoff_dirty = false;
    }
    if (m) m->unlock();
}

int vag_ctx_create(int device, vag_ctx** out) {
    if (!out) return set_err(VAG_E_INVALID, "null output pointer");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return set_err(VAG_E_NO_DEVICE, "no HIP device available: the engine has no CPU path");
    if (device < 0 || device >= n) return set_err(VAG_E_INVALID, "device %d out of range (0..%d)", device, n - 1);
    HIPCHK(hipSetDevice(device));
    vag_ctx* c = new vag_ctx();
    c->device = device;
    const int rc = ctx_init(c);
    if (rc) {  // nothing half-built survives a failed creation
        vag_ctx_destroy(c);
        return rc;
    }
    *out = c;
    return VAG_OK;
}

void vag_ctx_destroy(vag_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (DevBuf* b : {&c->d_partial2, &c->d_ssc2, &c->d_bandidx, &c->d_sptab, &c->d_workcount, &c->d_knlut, &c->d_icy, &c->d_cellq, &c->d_band, &c->d_ichdr, &c->d_icplan, &c->d_icpool, &c->d_icused, &c->d_icslow,
                      &c->d_icstatus, &c->d_icunclamp, &c->d_ssc, &c->d_shock_r, &c->d_cellpar_r, &c->d_celldet_r, &c->d_icy_r,
                      &c->d_cellq_r, &c->d_params_rvs, &c->d_inj, &c->d_comp, &c->d_cellgeo, &c->d_fail, &c->d_dynrec, &c->d_gridscratch, &c->d_chi2, &c->d_bandobs, &c->d_params, &c->d_t, &c->d_nu, &c->d_lg2t, &c->d_lg2nu, &c->d_tminmax, &c->d_bandw, &c->d_out,
                      &c->d_meta, &c->d_phi, &c->d_theta, &c->d_rep_of, &c->d_rep_start, &c->d_tdec, &c->d_geo_th, &c->d_geo_ph, &c->d_row_off,
                      &c->d_cell_off, &c->d_shock, &c->d_cellpar, &c->d_row_status, &c->d_celldet, &c->d_partial,
                      &c->d_fit, &c->d_theta_in, &c->d_valid, &c->d_series_flux})
        b->release();
    c->h_meta.release();
    c->h_off.release();
    c->h_plan.release();
    c->d_plan.release();
    c->d_chunk.release();
    c->d_icwork.release();
    c->h_fit.release();
    c->d_fitstat.release();
    for (DevBuf* b : {&c->d_mix_flags, &c->d_mix_perm, &c->d_mix_params, &c->d_mix_out, &c->shard_costs[0].cost, &c->shard_costs[1].cost, &c->shard_costs[2].cost, &c->shard_costs[3].cost, &c->shard_flights[0].table, &c->shard_flights[1].table, &c->shard_flights[2].table, &c->shard_flights[3].table, &c->shard_flights[4].table, &c->shard_flights[5].table, &c->shard_flights[6].table, &c->shard_flights[7].table, &c->d_shard_theta, &c->d_shard_ll, &c->d_order[0], &c->d_order[1], &c->d_cost_f, &c->d_rowgeo, &c->d_icneed})
        b->release();
    for (auto& e : c->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : c->prof_ev) (void)hipEventDestroy(e);
    if (c->ev_handoff) (void)hipEventDestroy(c->ev_handoff);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int vag_ctx_set_stream(vag_ctx* c, void* s) {
    ApiLock api_lock(c);
    if (!c) return set_err(VAG_E_INVALID, "null context");
    hipStream_t next;
    if (s == VAG_STREAM_LEGACY_DEFAULT)
        next = nullptr;  // the legacy default stream
    else
        next = s ? reinterpret_cast<hipStream_t>(s) : c->own_stream;
    if (next != c->stream) {
        // The context's scratch buffers (parameters, grid results, shock arrays, deal tables ...) are reused by every call, so
        // work queued on the stream the context leaves must be ordered before anything the next stream does with them: an event
        // at the tail of the old stream, waited for by the new one.  No host synchronisation.
        (void)hipSetDevice(c->device);
        const bool foreign = c->stream != nullptr && c->stream != c->own_stream;
        if (!foreign) {  // the context's own stream / the legacy default stream: always alive, the event is recorded here
            if (hipEventRecord(c->ev_handoff, c->stream) != hipSuccess) return set_err(VAG_E_HIP, "vag_ctx_set_stream: event record failed");
            c->handoff_ready = true;
        }
        // a caller-owned stream may have been destroyed since its last call (legal: nothing can be in flight on it then), so it is
        // never touched here: every device-resident entry point records the event at its own end, while the stream is known to be
        // alive (HandoffScope), and the host-buffer entry points end with a synchronisation
        if (c->handoff_ready && hipStreamWaitEvent(next, c->ev_handoff, 0) != hipSuccess)
            return set_err(VAG_E_HIP, "vag_ctx_set_stream: the new stream cannot wait for the work queued on the old one");
        c->handoff_ready = false;
        c->stream = next;
    }
    return VAG_OK;
}

// At the end of an entry point that leaves work in flight on a caller-owned stream: the hand-off event of vag_ctx_set_stream
struct HandoffScope {
    vag_ctx* c;
    explicit HandoffScope(vag_ctx* ctx) : c(ctx) {}
    ~HandoffScope() {
        if (c && c->stream != nullptr && c->stream != c->own_stream)
            c->handoff_ready = hipEventRecord(c->ev_handoff, c->stream) == hipSuccess;
    }
};

int vag_ctx_get_stream(vag_ctx* c, void** out) {
    ApiLock api_lock(c);
    if (!c || !out) return set_err(VAG_E_INVALID, "null context or pointer");
    if (c->stream == c->own_stream)
        *out = nullptr;
    else
        *out = c->stream ? reinterpret_cast<void*>(c->stream) : VAG_STREAM_LEGACY_DEFAULT;
    return VAG_OK;
}

int vag_ctx_count_work(vag_ctx* c, int enable) {
    ApiLock api_lock(c);
    if (!c) return set_err(VAG_E_INVALID, "null context");
    c->count_work = enable != 0;
    return VAG_OK;
}

static int read_row_failures(vag_ctx* c) {
    int f[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (c->d_fail.p && c->n_rows > 0) {
        HIPCHK(hipMemcpyAsync(f, c->d_fail.p, sizeof f, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    c->plan.n_rows_failed = f[1];
    c->plan.n_rows_gave_up = f[2] + f[3];
    {   // 64-bit tallies (vag_ctx_count_work; ADVICE r05: the int32 counter wrapped beyond ~3 M rows)
        unsigned long long t64[3];
        std::memcpy(t64, f + 8, sizeof t64);
        c->plan.ode_rhs = (long long)t64[0];
        c->plan.ode_lane_attempts = (long long)t64[1];
        c->plan.ode_lane_slots = (long long)t64[2];
    }
    if (c->fit_stats_pending && c->d_fitstat.p) {  // the last likelihood call's tallies over ALL of its passes
        int fs[4] = {0, 0, 0, 0};
        HIPCHK(hipMemcpyAsync(fs, c->d_fitstat.p, sizeof fs, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        c->plan.n_walkers_rejected = fs[0];
        c->plan.n_walkers_ssc_failed = fs[1];
        c->fit_stats_pending = false;
    }
    if (c->ic_slow_unread && c->d_icused.p) {  // a likelihood call's LAST table build (the builds before it reused the counters)
        unsigned long long n = 0;
        HIPCHK(hipMemcpyAsync(&n, c->d_icused.as<unsigned long long>() + 3, sizeof n, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        c->plan.n_ssc_slow_cells += (long long)n;
        c->ic_slow_unread = false;
    }
    return VAG_OK;
}

int vag_last_plan(vag_ctx* c, vag_plan* out) {
    ApiLock api_lock(c);
    if (!c || !out) return set_err(VAG_E_INVALID, "null context or output");
    const int rc = read_row_failures(c);
    *out = c->plan;
    return rc;
}

int vag_ctx_synchronize(vag_ctx* c) {
    ApiLock api_lock(c);
    if (!c) return set_err(VAG_E_INVALID, "null context");
    HIPCHK(hipStreamSynchronize(c->stream));
    return VAG_OK;
}

static int collect_times_fwd(vag_ctx* c);
int vag_ctx_profile(vag_ctx* c, int enable) {
    ApiLock api_lock(c);
    if (!c) return set_err(VAG_E_INVALID, "null context");
    c->prof_on = enable != 0;
    return VAG_OK;
}

int vag_last_profile(vag_ctx* c, vag_profile* out) {
    ApiLock api_lock(c);
    if (!c || !out) return set_err(VAG_E_INVALID, "null context or output");
    HIPCHK(hipStreamSynchronize(c->stream));
    double acc[PS_COUNT] = {0};
    float first = 0, last = 0;
    for (size_t i = 0; i < c->prof_spans.size(); ++i) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, c->prof_ev[c->prof_spans[i].e0], c->prof_ev[c->prof_spans[i].e1]));
        acc[c->prof_spans[i].id] += ms;
        float to_end = 0;
        HIPCHK(hipEventElapsedTime(&to_end, c->prof_ev[c->prof_spans[0].e0], c->prof_ev[c->prof_spans[i].e1]));
        last = std::max(last, to_end);
    }
    (void)first;
    *out = vag_profile{acc[PS_DYNAMICS], 0.0, acc[PS_SYN_ELECTRONS], acc[PS_SYN_PHOTONS], acc[PS_COOLING], acc[PS_SYNC_FLUX],
                       acc[PS_IC_PHOTONS], acc[PS_SSC_FLUX], (double)last};
    return VAG_OK;
}

int vag_last_stage_times(vag_ctx* c, vag_stage_times* out) {
    ApiLock api_lock(c);
    if (!c || !out) return set_err(VAG_E_INVALID, "null context or output");
    // measured with HIP events recorded on the context stream around each kernel of the last batch call
    const int rc = collect_times_fwd(c);
    *out = c->times;
    return rc;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// pipeline stages (host orchestration)
// ------------------------------------------------------------------------------------------------
namespace {

// Swap the reverse shock's arrays into (e = 1) or out of (e = 0) the slots every radiation / flux pass reads, and
// point the passes at the matching radiation parameters (single_shock_emission is the same code for both shocks,
// pybind/pymodel.h:877-920).
void select_emitter(vag_ctx* c, int e, const vag_model_params* d_params) {
    if (e != c->cur_emitter) {
        std::swap(c->d_shock, c->d_shock_r);
        std::swap(c->d_cellpar, c->d_cellpar_r);
        std::swap(c->d_celldet, c->d_celldet_r);
        std::swap(c->d_icy, c->d_icy_r);
        std::swap(c->d_cellq, c->d_cellq_r);
        c->cur_emitter = e;
    }
    if (e == 0) {
        c->cur_params = d_params;
        c->cur_ssc = (c->batch_flags & VAG_FLAG_SSC) != 0;
    } else {
        c->cur_params = c->d_params_rvs.as<vag_model_params>();
        c->cur_ssc = (c->batch_flags & VAG_FLAG_RVS_SSC) != 0;
    }
}

// Stage 3 for the selected emitter: electrons + photons per cell (generate_syn_electrons / generate_syn_photons),
// then apply_ic_cooling (pybind/pymodel.h:567-577) when its Radiation has ssc.
int run_radiation(vag_ctx* c, const vag_model_params* d_rad_params, int nb, const int* d_inj, bool ssc, bool want_details,
                  bool raw_shock = false) {
    hipStream_t st = c->stream;
    const long long cells = c->n_cells;
    const int rows = c->n_rows;
    if (c->d_cellpar.ensure(sizeof(double) * (size_t)cells * VAG_NPAR)) return VAG_E_HIP;
    if (ssc) want_details = true;  // the cooling pass works on the electron arrays
    if (want_details && c->d_celldet.ensure(sizeof(double) * (size_t)cells * VAG_NDET)) return VAG_E_HIP;
    Layout lay{c->d_row_off.as<int>(), c->d_cell_off.as<long long>()};
    {
        StageScope ps(c, PS_SYN_ELECTRONS);
        hipLaunchKernelGGL(vag_cells_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, st, d_rad_params, nb,
                           c->d_meta.as<VagGridMeta>(), lay, c->d_shock.as<double>(), cells, c->d_cellpar.as<double>(),
                           want_details ? c->d_celldet.as<double>() : nullptr, d_inj, raw_shock ? c->d_shock.as<double>() : nullptr,
                           want_details || vag_hook("VAG_CELLS_WRITE_BACK") != nullptr);
    }
    HIPCHK(hipGetLastError());
    if (ssc) {  // cool the electrons row by row, then rebuild the photons
        if (c->d_icy.ensure(sizeof(double) * (size_t)cells * VAG_NICY)) return VAG_E_HIP;
        if (c->d_cellq.ensure(sizeof(double) * (size_t)cells * VAG_NQ)) return VAG_E_HIP;
        {
            StageScope ps(c, PS_COOLING);
            hipLaunchKernelGGL(vag_ic_cooling_kernel, dim3((rows + 63) / 64), dim3(64), 0, st, d_rad_params, nb,
                               c->d_meta.as<VagGridMeta>(), lay, rows, c->d_shock.as<double>(), cells, c->d_celldet.as<double>(),
                               d_inj);
        }
        HIPCHK(hipGetLastError());
        {
            StageScope ps(c, PS_SYN_PHOTONS);
            hipLaunchKernelGGL(vag_photons_ic_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, st, d_rad_params, nb,
                               c->d_meta.as<VagGridMeta>(), lay, c->d_shock.as<double>(), cells, c->d_celldet.as<double>(),
                               c->d_icy.as<double>(), c->d_cellpar.as<double>(), c->d_cellq.as<double>(), d_inj);
        }
        HIPCHK(hipGetLastError());
    }
    return VAG_OK;
}

// Wait until the grid kernel of the current call has published its batch summary in pinned host memory (plan_scan_wave writes
// the sequence number last).  A spin on host memory: no copy, no stream synchronisation, the queue behind the grid kernel keeps
// running.  Falls back to a stream synchronisation if the number does not arrive (a failed launch).
int wait_plan(vag_ctx* c) {
    volatile VagDevPlan* hp = c->h_plan.as<VagDevPlan>();
    const auto t0 = std::chrono::steady_clock::now();
    for (long long spins = 0; hp->seq != c->plan_seq; ++spins) {
        __builtin_ia32_pause();
        if ((spins & 0xfff) == 0xfff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
            HIPCHK(hipStreamSynchronize(c->stream));
            if (hp->seq != c->plan_seq) return set_err(VAG_E_HIP, "the grid stage did not publish its batch summary");
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return VAG_OK;
}

// ODE rows per wavefront of the dynamics kernels (one lane integrates one row).  Measured (profiles/r02_rpw.txt): fewer
// rows per wavefront do NOT pay for the general kernel -- at 256 VGPRs only one wavefront fits a SIMD and the dispatcher
// does not spread single-wavefront workgroups evenly -- so full wavefronts stay the default; the knob remains for tuning.
// The coupled forward + reverse shock solver keeps its retry loop inside the step (a wavefront repeats an attempt while ANY of its
// lanes rejects) and one wavefront fills a SIMD (256 VGPRs): fewer rows per wavefront mean fewer repeated attempts and -- while the
// batch has fewer wavefronts than the chip has SIMDs -- more SIMDs at work.  VAG_PAIR_RPW overrides (tuning).
int pair_rows_per_wave(int rows, const char* env = "VAG_PAIR_RPW") {
    if (const char* e = vag_hook(env)) {
        const int v = std::atoi(e);
        if (v > 0) return std::min(v, 64);
    }
    // Measured (profiles/debug/pair_rpw_probe.py: one configs[2] model = 65 rows / 512 models = 33 k rows; general_rpw_probe.py: one
    // spreading Gaussian jet = 49 rows / 256 of them = 12.5 k rows; same bits whatever the choice): 64 rows per wavefront 3.18 / 3.48 ms
    // and 1.85 / 1.98 ms, 32: 2.83 / 5.35 and 1.57 / 2.28, 16: 2.68 / 5.89 and 1.77 / 2.30, 8: 2.32 / 16.3 and 1.41 / 4.36 -- fewer rows pay
    // for a handful of models and cost as soon as there are a few hundred wavefronts (a 256-VGPR wavefront owns its SIMD and the
    // dispatcher does not spread them evenly), so only small batches are split up.
    for (int rpw = 8; rpw < 64; rpw *= 2)
        if ((rows + rpw - 1) / rpw <= 64) return rpw;
    return 64;
}
int dyn_rows_per_wave(int rows) {
    (void)rows;
    if (const char* e = vag_hook("VAG_DYN_RPW")) {
        const int v = std::atoi(e);
        if (v > 0) return std::min(v, 64);
    }
    return 64;
}

// Stage 1-3: adaptive grid -> blast-wave dynamics -> per-cell radiation, for nb models whose
// parameters are already in HBM.  d_tminmax holds the observer-time extrema [s].
int run_model_stages(vag_ctx* c, const vag_model_params* d_params, int nb, bool want_details) {
    c->handoff_dirty = true;  // work is about to be queued on the context stream (ApiLock::~ApiLock)
    c->order_active = c->order_next;  // only a likelihood call that permuted its walkers sets order_next (vag_ctx::d_order)
    c->order_next = false;
    hipStream_t st = c->stream;
    if (c->d_meta.ensure(sizeof(VagGridMeta) * nb)) return VAG_E_HIP;
    if (c->d_cost_f.ensure(sizeof(float) * nb)) return VAG_E_HIP;
    // per-model angular arrays at the stride of the layout the grid kernel runs with (VagGridMeta::th_stride / ph_stride): 17 KB per
    // model for default-resolution batches (256 theta / 208 phi slots), 140 KB when a batch needed the large layout
    auto ensure_angular = [&](int large) -> bool {
        const size_t ts = layout_theta(large), ps = layout_phi(large);
        if (large == 2 && c->d_gridscratch.ensure(sizeof(GridSharedHuge) * (size_t)nb)) return true;
        return c->d_phi.ensure(sizeof(double) * (size_t)nb * ps) || c->d_theta.ensure(sizeof(double) * (size_t)nb * ts) ||
               c->d_tdec.ensure(sizeof(double) * (size_t)nb * 3 * ts) || c->d_geo_th.ensure(sizeof(double) * (size_t)nb * 3 * ts) ||
               c->d_geo_ph.ensure(sizeof(double) * (size_t)nb * 2 * ps) || c->d_rep_of.ensure(sizeof(int) * (size_t)nb * ts) ||
               c->d_rep_start.ensure(sizeof(int) * (size_t)nb * ts) || c->d_rowgeo.ensure(sizeof(double) * (size_t)nb * rowgeo_stride(large));
    };
    if (const char* e = vag_hook("VAG_GRID_FORCE_LARGE"))  // test hook: the large (1) / huge (2) layout for batches that fit the small one
        c->grid_large = std::max(c->grid_large, std::atoi(e) == 2 ? 2 : 1), c->grid_large_idle = 0;
    if (ensure_angular(c->grid_large)) return VAG_E_HIP;
    if (c->d_row_off.ensure(sizeof(int) * 2 * (size_t)(nb + 1))) return VAG_E_HIP;  // [nb + 1] row offsets, [nb + 1] offsets of the 64-row blocks
    if (c->d_cell_off.ensure(sizeof(long long) * (size_t)(nb + 1))) return VAG_E_HIP;
    if (c->d_fail.ensure(sizeof(int) * 16)) return VAG_E_HIP;

    // Grid shapes decide the compact layout and the launch geometry of everything downstream.  The last wavefront of
    // vag_grid_kernel scans them on the device and publishes an 80-byte summary in pinned host memory; the host spins on its
    // sequence number (wait_plan: ~6 us after the kernel's end, no copy, no stream synchronisation) and launches the rest.
    // VAG_PLAN_AHEAD=1 instead plans from the previous call's summary with a margin and verifies afterwards
    // (finish_speculation): no host wait in the middle of the pipeline at all.  Measured equal within 3 % in a sampler loop
    // (0.716 vs 0.715 ms per 128-walker call) and slower when calls are queued back to back without reading ln L
    // (0.775 vs 0.693 ms), because its flux launches are sized for the margin -- so waiting is the default.
    static const bool plan_ahead = vag_hook("VAG_PLAN_AHEAD") != nullptr;
    const bool spec = plan_ahead && c->allow_spec && !want_details && c->hint_valid && c->hint_nb == nb;
    int cap_rows = INT32_MAX, cap_k = INT32_MAX, cap_pairs = INT32_MAX;
    long long cap_cells = INT64_MAX;
    if (spec) {
        cap_rows = c->hint.rows + c->hint.rows / 4 + 64;
        cap_cells = c->hint.cells + c->hint.cells / 4 + 4096;
        cap_k = std::min(VAG_MAX_TIME, c->hint.max_k + c->spec_margin_k);  // LDS of the flux kernels scales with it: keep it tight
        cap_pairs = c->hint.max_pairs + c->hint.max_pairs / 2 + 64;
        c->spec_cap_k = cap_k;
    }
    if (c->d_plan.ensure(sizeof(VagDevPlan) + 64)) return VAG_E_HIP;
    if (!c->h_plan.p) {  // pinned, host-mapped, COHERENT (fine-grained: a store from a running kernel reaches the host after a
                         // system-scope fence, not only at the end of the queue): the grid kernel's last wavefront writes the
                         // batch summary straight into it
        HIPCHK(hipHostMalloc(&c->h_plan.p, 256, hipHostMallocMapped | hipHostMallocCoherent));
        c->h_plan.cap = 256;
        std::memset(c->h_plan.p, 0, sizeof(VagDevPlan));
        void* dp = nullptr;
        HIPCHK(hipHostGetDevicePointer(&dp, c->h_plan.p, 0));
        c->d_hplan = static_cast<VagDevPlan*>(dp);
    }
    if (!c->plan_counter_ready) {  // the ticket counter behind the plan: zeroed once, the kernel leaves it at zero
        HIPCHK(hipMemsetAsync(reinterpret_cast<char*>(c->d_plan.p) + sizeof(VagDevPlan), 0, 64, st));
        c->plan_counter_ready = true;
    }
    c->ic_need_reset = true;
    c->prof_spans.clear();
    c->prof_used = 0;
    HIPCHK(hipEventRecord(c->ev[0], st));
    std::unique_ptr<StageScope> ps_grid(new StageScope(c, PS_DYNAMICS));  // closed right after the launch below
    auto launch_grid = [&](int large) {
        c->layout_large = large;  // the layout THIS batch is laid out with (c->grid_large may change below for the next one)
        auto kern = large == 2 ? vag_grid_kernel<2> : (large == 1 ? vag_grid_kernel<1> : vag_grid_kernel<0>);
        hipLaunchKernelGGL(kern, dim3(nb), dim3(WAVE), 0, st, d_params, nb, c->d_tminmax.as<double>(),
                           c->d_meta.as<VagGridMeta>(), c->d_phi.as<double>(), c->d_theta.as<double>(), c->d_rep_of.as<int>(),
                           c->d_rep_start.as<int>(), c->d_tdec.as<double>(), c->d_geo_th.as<double>(), c->d_geo_ph.as<double>(),
                           c->d_fail.as<int>(), reinterpret_cast<int*>(reinterpret_cast<char*>(c->d_plan.p) + sizeof(VagDevPlan)),
                           c->d_row_off.as<int>(), c->d_cell_off.as<long long>(), c->d_plan.as<VagDevPlan>(), c->d_hplan, ++c->plan_seq,
                           cap_rows, cap_cells, cap_k, cap_pairs, spec ? (c->hint.flags_first < 0 ? 0 : c->hint.flags_first) : -1,
                           spec ? c->hint.dyn_class : 0, c->d_cost_f.as<float>(), c->d_rowgeo.as<double>(),
                           large == 2 ? c->d_gridscratch.as<GridSharedHuge>() : nullptr);
    };
    launch_grid(c->grid_large);
    ps_grid.reset();
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(c->ev[1], st));
    long long cells, pairs, eat;
    int rows, max_k, max_pairs, n_ok, n_invalid = 0, n_capacity = 0, dyn_class, flags;
    c->spec_pending = false;
    if (!spec) {
        VagDevPlan* hp = c->h_plan.as<VagDevPlan>();
        if (int rcw = wait_plan(c)) return rcw;
        // some model's angular grid outgrew the layout of the grid kernel: lay the batch out again with the next one (the same grids
        // for every model that fitted), and keep using it while the caller keeps sending such models
        while (hp->n_capacity > 0 && c->grid_large < 2) {
            ++c->grid_large;
            HIPCHK(hipStreamSynchronize(st));  // (the arrays are about to be re-allocated at the larger stride; nothing may still write them)
            if (ensure_angular(c->grid_large)) return VAG_E_HIP;
            if (vag_hook("VAG_DEBUG_LAUNCH"))
                std::fprintf(stderr, "[vag] grid: %d of %d models over the capacity of layout %d, laying the batch out again\n", hp->n_capacity, nb,
                             c->grid_large - 1);
            launch_grid(c->grid_large);
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventRecord(c->ev[1], st));
            if (int rcw = wait_plan(c)) return rcw;
        }
        if (hp->n_capacity == 0 && c->grid_large && c->layout_large == c->grid_large && ++c->grid_large_idle >= 8) {
            c->grid_large = 0;  // eight batches in a row fitted: back to the small layout (several models per CU)
            c->grid_large_idle = 0;
        }
        if (hp->n_capacity > 0) c->grid_large_idle = 0;
        c->mixed_flags_seen = hp->flags_mixed != 0;
        if (hp->flags_mixed)
            return set_err(VAG_E_UNSUPPORTED, "models with different Radiation / shock flags in one device-resident batch: split it by flags "
                                              "(the flux entry points do that themselves)");
        rows = hp->rows, cells = hp->cells, pairs = hp->pairs, eat = hp->eat, max_k = hp->max_k, max_pairs = hp->max_pairs;
        n_ok = hp->n_ok, n_invalid = hp->n_invalid, n_capacity = hp->n_capacity, dyn_class = hp->dyn_class;
        flags = hp->flags_first < 0 ? 0 : hp->flags_first;
        c->hint = *hp;
        c->hint_nb = nb;
        c->hint_valid = hp->n_ok > 0;
    } else {  // launch geometry and strides from the capacities, kernel choice from the previous call's flags
        rows = cap_rows, cells = cap_cells, max_k = cap_k, max_pairs = cap_pairs;
        pairs = c->hint.pairs, eat = c->hint.eat, n_ok = c->hint.n_ok, dyn_class = c->hint.dyn_class;
        flags = c->hint.flags_first < 0 ? 0 : c->hint.flags_first;
        c->spec_pending = true;
    }
    c->batch_flags = flags;
    // (A spreading jet under axisymmetric=False has one time lattice and one ODE solve per (phi, theta) node, grid-refinement.h:619-625:
    // the ODE rows are (phi, theta) pairs, VagGridMeta::rep_phi_stride.)
    c->nb = nb;
    c->n_rows = rows;
    c->n_cells = cells;
    c->max_k = max_k;
    c->max_pairs = max_pairs;
    c->total_pairs = pairs;
    c->eat_cells = eat;
    c->n_ok = n_ok;
    c->plan = vag_plan{};
    c->fit_stats_pending = false;
    c->ic_slow_unread = false;
    c->plan.n_models_ok = n_ok;
    c->plan.n_rows = rows;
    c->plan.n_cells = cells;
    c->plan.total_pairs = pairs;
    c->plan.eat_cells = eat;
    c->plan.n_models_invalid = n_invalid;
    c->plan.n_models_capacity = n_capacity;
    c->meta_on_host = false;
    if (rows == 0) {
        HIPCHK(hipEventRecord(c->ev[2], st));
        HIPCHK(hipEventRecord(c->ev[3], st));
        return VAG_OK;
    }
    const bool rvs = (c->batch_flags & VAG_FLAG_RVS) != 0;
    const bool spreading = (c->batch_flags & VAG_FLAG_SPREADING) != 0;
    if (c->d_shock.ensure(sizeof(double) * (size_t)cells * VAG_NSHOCK)) return VAG_E_HIP;
    if (c->d_row_status.ensure(sizeof(int) * (size_t)rows)) return VAG_E_HIP;
    Layout lay{c->d_row_off.as<int>(), c->d_cell_off.as<long long>()};
    bool raw_shock = false;
    std::unique_ptr<StageScope> ps_dyn(new StageScope(c, PS_DYNAMICS));
    if (rvs) {  // generate_shock_pair (reverse-shock.tpp:592-614): both shocks from one ODE state per row
        if (c->d_shock_r.ensure(sizeof(double) * (size_t)cells * VAG_NSHOCK)) return VAG_E_HIP;
        if (c->d_inj.ensure(sizeof(int) * (size_t)rows)) return VAG_E_HIP;
        if (c->d_params_rvs.ensure(sizeof(vag_model_params) * (size_t)nb)) return VAG_E_HIP;
        hipLaunchKernelGGL(vag_rvs_params_kernel, dim3((nb + 127) / 128), dim3(128), 0, st, d_params, nb,
                           c->d_params_rvs.as<vag_model_params>());
        const int prw = pair_rows_per_wave(rows);
        hipLaunchKernelGGL(vag_dynamics_pair_kernel, dim3((rows + prw - 1) / prw), dim3(64), 0, st, d_params, nb,
                           c->d_meta.as<VagGridMeta>(), c->d_theta.as<double>(), c->d_rep_start.as<int>(),
                           c->d_tdec.as<double>(), lay, rows, c->d_shock.as<double>(), c->d_shock_r.as<double>(), cells,
                           c->d_inj.as<int>(), c->d_row_status.as<int>(), c->d_fail.as<int>(), c->d_phi.as<double>(),
                           c->d_tminmax.as<double>(), prw);
    } else if (dyn_class == 0 && !vag_hook("VAG_DYN_GENERAL")) {  // the common case: flat attempt loop, raw saves
        raw_shock = true;
        // Batches beyond one and a half rounds of the plain kernel's integrator slots (four two-wavefront workgroups per CU) take the
        // persistent kernel whose lanes refill from a row queue and save inline (two integrators per SIMD instead of one integrator and
        // its saver); smaller ones keep the plain kernel, which is built for the latency of one row (no preparation pass on the chain).
        // Measured (profiles/r06_refill_probe.txt, C4 walkers, ODE stage plain / refill): 53.9 k rows 0.37 / 0.41 ms, 108.6 k rows
        // 0.72 / 0.62, 218 k rows 1.22 / 0.89, 433.8 k rows 2.28 / 1.73.  The rows are queued longest first (vag_dyn_fast.h: a row's step
        // count is predictable from its start record), by a counting sort of three small kernels.  VAG_DYN_REFILL=0 / 1 forces either; same bits (tests/test_gpu_parity.py).
        bool refill = rows >= 96 * 4LL * c->n_cus;
        if (const char* e = vag_hook("VAG_DYN_REFILL")) refill = std::atoi(e) != 0;
        if (refill) {
            int refill_min = 16;  // finished lanes a wavefront collects before it takes new rows (a refill is ~200 instructions for the whole wavefront; 1 / 4 / 8 / 16: 1.97 / 1.80 / 1.80 / 1.73 ms per 8192 walkers)
            if (const char* e = vag_hook("VAG_DYN_REFILL_MIN")) refill_min = std::max(1, std::min(64, std::atoi(e)));
            // records | the sorted queue (rows ints) | the queue's two counters | histogram / offsets [DYN_BUCKETS][chunks] | length class per row
            const size_t n_chunks = ((size_t)rows + DYN_CHUNK - 1) / DYN_CHUNK;
            const size_t rec_bytes = sizeof(double) * (size_t)rows * DYN_ROWREC, ord_bytes = sizeof(int) * (((size_t)rows + 1) & ~(size_t)1);
            const size_t hist_bytes = sizeof(unsigned) * DYN_BUCKETS * n_chunks;
            if (c->d_dynrec.ensure(rec_bytes + ord_bytes + 16 + hist_bytes + (size_t)rows)) return VAG_E_HIP;
            char* dyn_base = static_cast<char*>(c->d_dynrec.p);
            int* d_dynorder = reinterpret_cast<int*>(dyn_base + rec_bytes);
            unsigned* d_queue = reinterpret_cast<unsigned*>(dyn_base + rec_bytes + ord_bytes);
            unsigned* d_hist = d_queue + 4;
            signed char* d_cls = reinterpret_cast<signed char*>(dyn_base + rec_bytes + ord_bytes + 16 + hist_bytes);
            hipLaunchKernelGGL(vag_dyn_prep_kernel, dim3((unsigned)n_chunks), dim3(DYN_CHUNK), 0, st, d_params, nb, c->d_meta.as<VagGridMeta>(),
                               c->d_theta.as<double>(), c->d_rep_start.as<int>(), c->d_tdec.as<double>(), lay, rows, c->d_shock.as<double>(), cells,
                               c->d_row_status.as<int>(), c->d_dynrec.as<double>(), d_cls, d_hist);
            hipLaunchKernelGGL(vag_dyn_scan_kernel, dim3(1), dim3(1024), 0, st, d_hist, (int)n_chunks, d_queue);
            hipLaunchKernelGGL(vag_dyn_file_kernel, dim3((unsigned)n_chunks), dim3(DYN_CHUNK), 0, st, d_cls, d_hist, rows, d_dynorder);
            auto kern = c->count_work ? vag_dynamics_refill_kernel<true> : vag_dynamics_refill_kernel<false>;
            if (c->dyn_refill_wg_per_cu <= 0) {  // what the registers allow (two wavefronts per SIMD at <= 256 VGPRs)
                int per_cu = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, vag_dynamics_refill_kernel<false>, 64, 0) != hipSuccess || per_cu <= 0) per_cu = 8;
                c->dyn_refill_wg_per_cu = per_cu;
            }
            long long slots = (long long)c->dyn_refill_wg_per_cu * c->n_cus;
            if (const char* e = vag_hook("VAG_DYN_REFILL_WGS")) slots = std::max(1, std::atoi(e));
            const unsigned wgs = (unsigned)std::max<long long>(1, std::min<long long>((rows + 63) / 64, slots));
            hipLaunchKernelGGL(kern, dim3(wgs), dim3(64), 0, st, c->d_dynrec.as<double>(), rows, c->d_shock.as<double>(), cells,
                               c->d_row_status.as<int>(), c->d_sptab.as<double>(), refill_min, c->d_fail.as<int>(), d_queue, d_dynorder);
        } else {
        const int rpw = dyn_rows_per_wave(rows);
        hipLaunchKernelGGL(c->count_work ? vag_dynamics_fast_kernel<true> : vag_dynamics_fast_kernel<false>, dim3((rows + rpw - 1) / rpw),
                           dim3(128), 0, st, d_params, nb, c->d_meta.as<VagGridMeta>(), c->d_theta.as<double>(), c->d_rep_start.as<int>(),
                           c->d_tdec.as<double>(), lay, rows, c->d_shock.as<double>(), cells, c->d_row_status.as<int>(),
                           c->d_sptab.as<double>(), rpw, c->d_fail.as<int>());
        }
    } else {
        const bool inject = (c->batch_flags & VAG_FLAG_MAGNETAR) != 0;
        auto kern = spreading ? (inject ? vag_dynamics_kernel<true, true> : vag_dynamics_kernel<true, false>)
                              : (inject ? vag_dynamics_kernel<false, true> : vag_dynamics_kernel<false, false>);
        const int rpw = pair_rows_per_wave(rows, "VAG_DYN_RPW");  // (the general solver retries inside the step like the pair solver)
        hipLaunchKernelGGL(kern, dim3((rows + rpw - 1) / rpw), dim3(64), 0, st, d_params, nb, c->d_meta.as<VagGridMeta>(),
                           c->d_theta.as<double>(), c->d_rep_start.as<int>(), c->d_tdec.as<double>(), lay, rows,
                           c->d_shock.as<double>(), cells, c->d_row_status.as<int>(), c->d_sptab.as<double>(), rpw,
                           c->d_fail.as<int>(), c->d_phi.as<double>(), c->d_tminmax.as<double>());
    }
    HIPCHK(hipGetLastError());
    if (spreading) {  // per-cell viewing geometry from the evolved theta (both shocks ride the same contact discontinuity)
        if (c->d_cellgeo.ensure(sizeof(double) * (size_t)cells * 3)) return VAG_E_HIP;
        hipLaunchKernelGGL(vag_spread_geo_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, st, nb,
                           c->d_meta.as<VagGridMeta>(), lay, c->d_shock.as<double>(), cells, c->d_cellgeo.as<double>());
        HIPCHK(hipGetLastError());
    }
    ps_dyn.reset();
    HIPCHK(hipEventRecord(c->ev[2], st));
    c->cur_emitter = 0;
    c->cur_params = d_params;
    c->cur_ssc = (c->batch_flags & VAG_FLAG_SSC) != 0;
    int rc = run_radiation(c, d_params, nb, nullptr, c->cur_ssc, want_details, raw_shock);
    if (rc) return rc;
    if (rvs) {
        select_emitter(c, 1, d_params);
        rc = run_radiation(c, c->cur_params, nb, c->d_inj.as<int>(), c->cur_ssc, want_details);
        select_emitter(c, 0, d_params);
        if (rc) return rc;
    }
    HIPCHK(hipEventRecord(c->ev[3], st));
    return VAG_OK;
}

int choose_pairs_per_block(const vag_ctx* c) {
    if (const char* e = vag_hook("VAG_PAIRS_PER_BLOCK")) {  // tuning/debug override
        const int v = std::atoi(e);
        if (v > 0) return std::min(v, std::max(1, c->max_pairs));
    }
    // enough workgroups to fill 256 CUs several times over, but not so many that staging the photon rows
    // and the partial grids dominate
    long long ppb = (c->total_pairs + 16383) / 16384;
    long long lo = 4;
    if (c->total_pairs / 16 >= 2048) lo = 16;
    if (c->total_pairs / 64 >= 4096) lo = 64;
    ppb = std::max(ppb, lo);
    ppb = std::min<long long>(ppb, std::max(1, c->max_pairs));
    return (int)std::max<long long>(1, ppb);
}

// Lattice nodes the flux kernels stage at a time (their LDS scales with it).  Longer lattices are taken in overlapping pieces
// inside the kernels; VAG_FLUX_K_CAP is a test hook that makes ordinary models take that path.
static int flux_ks(const vag_ctx* c) {
    int cap = 512;
    if (const char* e = vag_hook("VAG_FLUX_K_CAP")) cap = std::max(4, std::atoi(e));
    return std::max(2, std::min(c->max_k, cap));
}

// the work-item counter of a persistent launch (SeriesArgs::work), in the zeroed words behind the device plan: the kernel behind every
// such launch puts it back to zero, and kernels of one context run on one stream
static int* work_counters(vag_ctx* c) { return reinterpret_cast<int*>(reinterpret_cast<char*>(c->d_plan.p) + sizeof(VagDevPlan)) + 4; }

// dynamic LDS of vag_flux_grid_kernel (layout at the top of the kernel)
static size_t flux_grid_lds_bytes(int mode, int ks, int nt, int nnu) {
    const size_t slots = (size_t)nt * nnu;
    size_t d = (size_t)(VAG_NPAR + 4) * ks + (size_t)ks * nnu + 2 * (size_t)nt + nnu + SP_LDS_DOUBLES + slots;
    if (mode == FLUX_FUSED) d += 6 * (size_t)ks + (size_t)ks * nnu + slots;
    return sizeof(double) * d + sizeof(int) * nt + 16;  // + s_win: [2][2] observation-window counts (WinCount)
}

// Stage 4-5 for a (t, nu) grid request: d_lg2t/d_lg2nu are log2 of code-unit times / frequencies.
int run_flux_series(vag_ctx* c, const vag_model_params* d_params, int nb, const double* d_lg2t, const double* d_lg2nu, int n,
                    double* d_out, int mode = FLUX_SYN, int n_bands = 0, int grid_nt = 0);

// fixed-order sum of a flux pass's workgroup partials + normalisation: plain grids through the grouped kernel, band-weighted
// requests through the serial one
static void launch_reduce(hipStream_t st, const vag_model_params* d_params, const VagGridMeta* meta, const double* partial, int max_blocks,
                          int ppb, int nt, int nnu, const double* d_bandw, double* d_out, int nb,
                          int* work_counter = nullptr /* of a persistent flux launch before this one: the reduction puts it back to zero */) {
    if (d_bandw) {
        hipLaunchKernelGGL(vag_reduce_kernel, dim3((nt + 255) / 256, nb), dim3(256), 0, st, d_params, meta, partial, max_blocks, ppb, nt, nnu,
                           d_bandw, d_out, work_counter);
    } else {
        const int slots = nt * nnu;
        hipLaunchKernelGGL(vag_reduce_grid_kernel, dim3((slots + REDUCE_SLOTS - 1) / REDUCE_SLOTS, nb), dim3(REDUCE_GROUPS * REDUCE_SLOTS), 0, st,
                           d_params, meta, partial, max_blocks, ppb, slots, d_out, work_counter);
    }
}

int run_flux_grid(vag_ctx* c, const vag_model_params* d_params, int nb, const double* d_lg2t, int nt,
                  const double* d_lg2nu, int nnu, const double* d_bandw, double* d_out, int mode = FLUX_SYN,
                  double* d_out2 = nullptr /* FLUX_FUSED: the SSC component */) {
    hipStream_t st = c->stream;
    StageScope ps(c, mode == FLUX_SSC ? PS_SSC_FLUX : PS_SYNC_FLUX);
    const int slots = nt * nnu;
    // Rows of a few hundred (nu, t) slots keep a 256-lane workgroup 22-78 % busy between its two barriers (C3, C5, C1: four or
    // three frequencies).  The wavefront-per-row kernel can serve them (VAG_GRID_ROWWISE=1: same algorithm -- boundary spectra
    // per (nu, lattice node of the window), log-log interpolation --, no workgroup barrier, accumulators in registers), but
    // MEASURED SLOWER on every such shape (C5 2.2 k vs 4.5 k light curves/s, C3 1.27 k vs 1.79 k, C1a 1.28 M vs 2.58 M: eight
    // points per lane cost the occupancy, and every lane walks its own bracket searches), so the workgroup kernel stays.
    if (slots <= SERIES_THREADS * SERIES_MAX_SLOTS && nnu <= SERIES_MAX_BANDS && !d_bandw && mode != FLUX_FUSED && !c->count_work &&
        vag_hook("VAG_GRID_ROWWISE") && !(c->batch_flags & VAG_FLAG_SPREADING))
        return run_flux_series(c, d_params, nb, d_lg2t, d_lg2nu, slots, d_out, mode, nnu, nt);
    if (slots > FLUX_MAX_SLOTS)
        return set_err(VAG_E_CAPACITY, "nt*nnu = %d exceeds %d per launch", slots, FLUX_MAX_SLOTS);
    // Small grids of large batches: a (theta, phi) row per lane (vag_grid_rows.h).  A lone model would walk its lattice as one
    // sequential chain there, so the batch must bring blocks of 64 rows enough to fill the GPU; models of a few rows (on-axis
    // top hats: 32 rows) would leave lanes empty.
    {
        const long long blocks = (c->total_pairs + FITROWS_ROWS - 1) / FITROWS_ROWS;
        if (mode != FLUX_FUSED && !(c->batch_flags & VAG_FLAG_SPREADING) && !c->count_work && slots <= GRIDROWS_MAX_SLOTS &&
            nnu <= GRIDROWS_BANDS && nt <= GRIDROWS_MAX_NT && blocks >= 4096 && c->total_pairs >= 128LL * nb && c->n_rows > 0 &&
            !vag_hook("VAG_GRID_ROW_PER_WORKGROUP")) {
            const int max_blocks = std::max(1, (c->max_pairs + FITROWS_ROWS - 1) / FITROWS_ROWS);
            if (c->d_partial.ensure(sizeof(double) * (size_t)nb * max_blocks * slots)) return VAG_E_HIP;
            SeriesArgs a{};
            a.params = d_params;
            a.meta = c->d_meta.as<VagGridMeta>();
            a.geo_th = c->d_geo_th.as<double>();
            a.geo_ph = c->d_geo_ph.as<double>();
            a.g_rep_of = c->d_rep_of.as<int>();
            a.lay = Layout{c->d_row_off.as<int>(), c->d_cell_off.as<long long>()};
            a.cellpar = c->d_cellpar.as<double>();
            a.lg2_t_obs = d_lg2t;
            a.lg2_nu_obs = d_lg2nu;
            a.n = slots;
            a.grid_nt = nt;
            a.n_bands = nnu;
            a.max_chunks = max_blocks;
            a.partial = c->d_partial.as<double>();
            a.sp_table = c->d_sptab.as<double>();
            a.cellq = c->d_cellq.as<double>();
            a.ichdr = c->d_ichdr.as<double>();
            a.icpool = c->d_icpool.as<double>();
            a.ic_status = c->d_icstatus.as<int>();
            c->plan.spec_evals += c->eat_cells * nnu;
            c->plan.interps += c->total_pairs * (long long)nt * nnu;
            c->plan.flux_blocks = max_blocks * nb;
            c->plan.pairs_per_block = FITROWS_ROWS;
            // The plain-synchrotron pass is a persistent launch like vag_flux_fit_rows_kernel (three workgroups per CU that take blocks
            // until none is left: vag_grid_rows.h); the IC-corrected and the tabulated-SSC pass keep one workgroup per four blocks (the
            // first has no registers for the loop -- 61 against 47 ms per 1024 C5 members, measured -- the second nothing to gain).
            const bool persistent = mode != FLUX_SYN_IC && mode != FLUX_SSC;
            a.nb = nb;
            a.work = work_counters(c);
            const long long wg_need = (blocks + nb + GRIDROWS_WAVES - 1) / GRIDROWS_WAVES;  // (total_pairs may be the previous call's)
            const dim3 g_all((max_blocks + GRIDROWS_WAVES - 1) / GRIDROWS_WAVES, nb), b(SERIES_THREADS * GRIDROWS_WAVES);
            const dim3 g_pers((unsigned)std::max<long long>(1, std::min<long long>(wg_need, 3LL * c->n_cus)));
            const size_t lds = grid_rows_lds_bytes(slots);
            if (mode == FLUX_SYN_IC)
                hipLaunchKernelGGL((vag_flux_grid_rows_kernel<FLUX_SYN_IC>), g_all, b, lds, st, a);
            else if (mode == FLUX_SSC)
                hipLaunchKernelGGL((vag_flux_grid_rows_kernel<FLUX_SSC>), g_all, b, lds, st, a);
            else
                hipLaunchKernelGGL((vag_flux_grid_rows_kernel<FLUX_SYN>), g_pers, b, lds, st, a);
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventRecord(c->ev[4], st));
            launch_reduce(st, d_params, c->d_meta.as<VagGridMeta>(), c->d_partial.as<double>(), max_blocks, FITROWS_ROWS, nt, nnu, d_bandw, d_out, nb,
                          persistent ? work_counters(c) : nullptr);
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventRecord(c->ev[5], st));
            return VAG_OK;
        }
    }
    const int ppb = choose_pairs_per_block(c);
    const int max_blocks = std::max(1, (c->max_pairs + ppb - 1) / ppb);
    if (c->d_partial.ensure(sizeof(double) * (size_t)nb * max_blocks * slots)) return VAG_E_HIP;
    const int ks = flux_ks(c);
    const size_t lds = flux_grid_lds_bytes(mode, ks, nt, nnu);
    if (lds > 160 * 1024) return set_err(VAG_E_CAPACITY, "LDS request %zu B exceeds 160 KiB (n_t=%d, nnu=%d)", lds, ks, nnu);
    if (mode == FLUX_FUSED && c->d_partial2.ensure(sizeof(double) * (size_t)nb * max_blocks * slots)) return VAG_E_HIP;
    FluxArgs a;
    a.params = d_params;
    a.meta = c->d_meta.as<VagGridMeta>();
    a.geo_th = c->d_geo_th.as<double>();
    a.geo_ph = c->d_geo_ph.as<double>();
    a.g_rep_of = c->d_rep_of.as<int>();
    a.cell_off = c->d_cell_off.as<long long>();
    a.cellpar = c->d_cellpar.as<double>();
    a.lg2_t_obs = d_lg2t;
    a.lg2_nu_obs = d_lg2nu;
    a.nt = nt;
    a.nnu = nnu;
    a.pairs_per_block = ppb;
    a.max_blocks = max_blocks;
    a.k_stride = ks;
    a.partial = c->d_partial.as<double>();
    a.partial2 = c->d_partial2.as<double>();
    a.sp_table = c->d_sptab.as<double>();
    a.work_count = nullptr;
    a.cellq = c->d_cellq.as<double>();
    a.ichdr = c->d_ichdr.as<double>();
    a.icpool = c->d_icpool.as<double>();
    a.ic_status = c->d_icstatus.as<int>();
    a.cellgeo = c->d_cellgeo.as<double>();
    const bool spreading = (c->batch_flags & VAG_FLAG_SPREADING) != 0;
    a.rowgeo = nullptr;
    a.rowgeo_stride = 0;
    if (!spreading) {  // the models' row-geometry records, written by the grid kernel with the layout it ran with
        a.rowgeo = c->d_rowgeo.as<double>();
        a.rowgeo_stride = rowgeo_stride(c->layout_large);
    }
    if (c->count_work && mode == FLUX_SYN && !spreading) {
        if (c->d_workcount.ensure(2 * sizeof(unsigned long long))) return VAG_E_HIP;
        HIPCHK(hipMemsetAsync(c->d_workcount.p, 0, 2 * sizeof(unsigned long long), st));
        a.work_count = c->d_workcount.as<unsigned long long>();
    }
    // upper bounds (every lattice node inside the observation window); vag_ctx_count_work(1) replaces them by
    // the exact tallies of the kernel
    c->plan.spec_evals += c->eat_cells * nnu;
    c->plan.interps += c->total_pairs * (long long)nt * nnu;
    c->plan.flux_blocks = max_blocks * nb;
    c->plan.pairs_per_block = ppb;
    if (c->n_rows > 0) {
        // requests with few (nu, t) slots and short rows keep less than half of a 512-lane workgroup busy: use 256 lanes
        // (+47 % on the C5 / C1b shapes; a 128-lane variant measured slower)
        const bool small = !spreading && !a.work_count && (long long)nt * nnu <= 512 && (long long)ks * ((nnu + 1) / 2) <= 512 &&
                           !vag_hook("VAG_FLUX_WIDE");
        if (vag_hook("VAG_DEBUG_LAUNCH")) {
            int occ = -1;
            if (small && mode == FLUX_SSC)
                (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, vag_flux_grid_kernel<false, FLUX_SSC, false, 256>, 256, lds);
            else if (small && mode == FLUX_SYN_IC)
                (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, vag_flux_grid_kernel<false, FLUX_SYN_IC, false, 256>, 256, lds);
            else if (small)
                (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, vag_flux_grid_kernel<false, FLUX_SYN, false, 256>, 256, lds);
            else if (mode == FLUX_SYN && !spreading)
                (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, vag_flux_grid_kernel<false, FLUX_SYN>, FLUX_THREADS, lds);
            std::fprintf(stderr, "[vag] grid flux launch: mode %d nt=%d nnu=%d ks=%d rows/wg=%d lanes=%d lds=%zu B wg/CU=%d\n", mode, nt, nnu, ks,
                         ppb, small ? 256 : FLUX_THREADS, lds, occ);
        }
        const bool pieces = c->max_k > ks;  // some lattice is longer than the staged row: the instantiations with the piece loop
        // (Measured and rejected, profiles/rejected/vag_flux_wide.h: ONE 1024-lane workgroup per CU sharing the staged row and tables, a
        // second boundary block in the LDS that frees, one barrier per row and balanced slot ownership -- bitwise the same fluxes,
        // 27.0-27.7 ms against 21.7 per 512 C2 models: sixteen wavefronts in lockstep lose the overlap two independent workgroups have.)
        if (pieces) {
            a.work_count = nullptr;  // the tallying instantiation has no piece loop: the plan keeps the upper bounds
            const dim3 g(max_blocks, nb), b(FLUX_THREADS);
#define VAG_PIECES_LAUNCH(M_)                                                                                          \
    do {                                                                                                               \
        if (spreading)                                                                                                 \
            hipLaunchKernelGGL((vag_flux_grid_kernel<false, M_, true, FLUX_THREADS, true>), g, b, lds, st, a);          \
        else                                                                                                           \
            hipLaunchKernelGGL((vag_flux_grid_kernel<false, M_, false, FLUX_THREADS, true>), g, b, lds, st, a);         \
    } while (0)
            if (mode == FLUX_FUSED)
                VAG_PIECES_LAUNCH(FLUX_FUSED);
            else if (mode == FLUX_SYN_IC)
                VAG_PIECES_LAUNCH(FLUX_SYN_IC);
            else if (mode == FLUX_SSC)
                VAG_PIECES_LAUNCH(FLUX_SSC);
            else
                VAG_PIECES_LAUNCH(FLUX_SYN);
#undef VAG_PIECES_LAUNCH
        } else if (small && mode == FLUX_FUSED)
            hipLaunchKernelGGL((vag_flux_grid_kernel<false, FLUX_FUSED, false, 256>), dim3(max_blocks, nb), dim3(256), lds, st, a);
        else if (spreading && mode == FLUX_FUSED)
            hipLaunchKernelGGL((vag_flux_grid_kernel<false, FLUX_FUSED, true>), dim3(max_blocks, nb), dim3(FLUX_THREADS), lds, st, a);
        else if (mode == FLUX_FUSED)
            hipLaunchKernelGGL((vag_flux_grid_kernel<false, FLUX_FUSED>), dim3(max_blocks, nb), dim3(FLUX_THREADS), lds, st, a);
        else if (small && mode == FLUX_SYN_IC)
            hipLaunchKernelGGL((vag_flux_grid_kernel<false, FLUX_SYN_IC, false, 256>), dim3(max_blocks, nb), dim3(256), lds, st, a);
        else if (small && mode == FLUX_SSC)
            hipLaunchKernelGGL((vag_flux_grid_kernel<false, FLUX_SSC, false, 256>), dim3(max_blocks, nb), dim3(256), lds, st, a);
        else if (small)
            hipLaunchKernelGGL((vag_flux_grid_kernel<false, FLUX_SYN, false, 256>), dim3(max_blocks, nb), dim3(256), lds, st, a);
        else if (spreading && mode == FLUX_SYN_IC)
            hipLaunchKernelGGL((vag_flux_grid_kernel<false, FLUX_SYN_IC, true>), dim3(max_blocks, nb), dim3(FLUX_THREADS), lds, st, a);
        else if (spreading && mode == FLUX_SSC)
            hipLaunchKernelGGL((vag_flux_grid_kernel<false, FLUX_SSC, true>), dim3(max_blocks, nb), dim3(FLUX_THREADS), lds, st, a);
        else if (spreading)
            hipLaunchKernelGGL((vag_flux_grid_kernel<false, FLUX_SYN, true>), dim3(max_blocks, nb), dim3(FLUX_THREADS), lds, st, a);
        else if (mode == FLUX_SYN_IC)
            hipLaunchKernelGGL((vag_flux_grid_kernel<false, FLUX_SYN_IC>), dim3(max_blocks, nb), dim3(FLUX_THREADS), lds, st, a);
        else if (mode == FLUX_SSC)
            hipLaunchKernelGGL((vag_flux_grid_kernel<false, FLUX_SSC>), dim3(max_blocks, nb), dim3(FLUX_THREADS), lds, st, a);
        else if (a.work_count)
            hipLaunchKernelGGL((vag_flux_grid_kernel<true, FLUX_SYN>), dim3(max_blocks, nb), dim3(FLUX_THREADS), lds, st, a);
        else
            hipLaunchKernelGGL((vag_flux_grid_kernel<false, FLUX_SYN>), dim3(max_blocks, nb), dim3(FLUX_THREADS), lds, st, a);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(c->ev[4], st));
    if (a.work_count) {
        unsigned long long h[2];
        HIPCHK(hipMemcpyAsync(h, c->d_workcount.p, sizeof h, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        c->plan.spec_evals += (long long)h[0] - c->eat_cells * nnu;
        c->plan.interps += (long long)h[1] - c->total_pairs * (long long)nt * nnu;
    }
    launch_reduce(st, d_params, c->d_meta.as<VagGridMeta>(), c->d_partial.as<double>(), max_blocks, ppb, nt, nnu, d_bandw, d_out, nb);
    if (mode == FLUX_FUSED)
        launch_reduce(st, d_params, c->d_meta.as<VagGridMeta>(), c->d_partial2.as<double>(), max_blocks, ppb, nt, nnu, d_bandw, d_out2, nb);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(c->ev[5], st));
    return VAG_OK;
}

__global__ void vag_add_kernel(double* __restrict__ out, const double* __restrict__ add, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] += add[i];
}

// SSC emission of the whole batch on a (t, nu) grid (single_shock_emission, pybind/pymodel.h:896-919):
// observation band per k -> SSC table per representative cell -> EAT flux integration over the tables.
// rebuild = the pass before this one ended with band breaches (status bit 2): those models' tables are rebuilt unclamped
int build_ssc_tables(vag_ctx* c, const vag_model_params* d_params, int nb, const double* d_lg2nu, int nnu, bool rebuild = false) {
    hipStream_t st = c->stream;
    StageScope ps(c, PS_IC_PHOTONS);
    const int band_stride = std::max(c->max_k, 1);  // [nb][2][band_stride]
    if (c->d_band.ensure(sizeof(double) * (size_t)nb * 2 * band_stride)) return VAG_E_HIP;
    if (c->d_ichdr.ensure(sizeof(double) * (size_t)std::max<long long>(c->n_cells, 1) * IC_HDR)) return VAG_E_HIP;
    if (c->d_icplan.ensure(sizeof(double) * (size_t)std::max<long long>(c->n_cells, 1) * IC_PLAN)) return VAG_E_HIP;
    constexpr int IC_SLOW_LIST = 1 << 18;  // cells of one table build that may take the slow path (more: VAG_E_CAPACITY)
    if (c->d_icused.ensure(4 * sizeof(unsigned long long))) return VAG_E_HIP;
    if (c->d_icslow.ensure(sizeof(int) * (size_t)IC_SLOW_LIST)) return VAG_E_HIP;
    if (c->d_icpool.ensure(sizeof(double) * 1024)) return VAG_E_HIP;  // (never null: the empty tables point at its first words)
    if (c->d_icstatus.ensure(sizeof(int) * (size_t)nb)) return VAG_E_HIP;
    if (c->d_icunclamp.ensure(sizeof(int) * (size_t)nb)) return VAG_E_HIP;
    if (rebuild) {
        hipLaunchKernelGGL(vag_ic_unclamp_kernel, dim3((nb + 255) / 256), dim3(256), 0, st, c->d_icstatus.as<int>(),
                           c->d_icunclamp.as<int>(), nb);
    } else {
        if (!c->ic_soft_fail || c->ic_need_reset)  // a likelihood pass ORs the failures of both shocks' tables into one status per walker
            HIPCHK(hipMemsetAsync(c->d_icstatus.p, 0, sizeof(int) * (size_t)nb, st));
        HIPCHK(hipMemsetAsync(c->d_icunclamp.p, 0, sizeof(int) * (size_t)nb, st));
    }
    c->ic_need_reset = false;
    if (c->n_rows > 0) {
        Layout lay{c->d_row_off.as<int>(), c->d_cell_off.as<long long>()};
        double narrow = 1.0;  // test hook: VAG_DEBUG_IC_NARROW=<factor> shrinks the clamp so that the flux pass breaches it
        if (const char* e = vag_hook("VAG_DEBUG_IC_NARROW")) narrow = std::atof(e);
        // tables only for the cells some row's observation window touches (the reference builds a cell's spectrum on its first
        // query); VAG_IC_ALL_CELLS=1 builds every cell's table (developer aid: A/B timing, and the loud-fault test)
        unsigned char* d_need = nullptr;
        double need_shrink = 1.0;  // test hook: VAG_DEBUG_IC_NEED_SHRINK=<factor> cuts the window short, so that a flux pass meets a skipped cell
        if (const char* e = vag_hook("VAG_DEBUG_IC_NEED_SHRINK")) need_shrink = std::atof(e);
        // (A likelihood call keeps every table: there a model's SSC status folds into the walker's score -- ic_soft_fail, -inf -- so a cell
        // wrongly left without a table would be a silent wrong answer instead of VAG_E_INTERNAL; round 4's sweeps found two such holes
        // in the range test, both on grid requests, both loud.)
        if (!vag_hook("VAG_IC_ALL_CELLS") && !c->ic_all_cells && c->d_tminmax.p && !c->ic_soft_fail) {
            if (c->d_icneed.ensure((size_t)std::max<long long>(c->n_cells, 1))) return VAG_E_HIP;
            HIPCHK(hipMemsetAsync(c->d_icneed.p, 0, (size_t)std::max<long long>(c->n_cells, 1), st));
            d_need = c->d_icneed.as<unsigned char>();
        }
        const bool huge = c->layout_large == 2;  // (the rows' viewing-cosine extrema then go through HBM: the scratch of the grid kernel is free by now)
        hipLaunchKernelGGL(huge ? vag_ic_band_kernel<true> : vag_ic_band_kernel<false>, dim3(nb), dim3(64), 0, st, d_params, c->d_meta.as<VagGridMeta>(),
                           c->d_geo_th.as<double>(), c->d_geo_ph.as<double>(), c->d_rep_of.as<int>(),
                           c->d_cell_off.as<long long>(), c->d_cellpar.as<double>(), d_lg2nu, nnu, c->d_band.as<double>(),
                           (c->batch_flags & VAG_FLAG_SPREADING) ? c->d_cellgeo.as<double>() : nullptr,
                           c->d_icunclamp.as<int>(), narrow, band_stride, c->d_tminmax.as<double>(), d_need, need_shrink,
                           huge ? c->d_gridscratch.as<double>() : nullptr);
        HIPCHK(hipGetLastError());
        if (c->count_work) {
            if (c->d_icwork.ensure(2 * sizeof(unsigned long long))) return VAG_E_HIP;
            HIPCHK(hipMemsetAsync(c->d_icwork.p, 0, 2 * sizeof(unsigned long long), st));
        }
        // {doubles of the pool in use: its first two words serve the gathers of the cells without a table; plan records written; doubles
        // and records of the cells on the slow path}
        static const unsigned long long ic_counters_start[4] = {2, 0, 0, 0};
        HIPCHK(hipMemcpyAsync(c->d_icused.p, ic_counters_start, sizeof ic_counters_start, hipMemcpyHostToDevice, st));
        // The pool grows to what the plan handed out (grow-only: in a sampler's loop it stops growing after a few calls).  The host has
        // to see the total before the spectrum kernel may write: one 32-byte copy and a wait per table build (~20 us against the
        // milliseconds of the build; a grid / series SSC pass ends with such a wait anyway, check_ic_status).  A LIKELIHOOD call has no
        // host wait anywhere in its SSC stage (its table status folds into the walker's score on the device), so it does not get one
        // here either: the pool is sized for the worst case -- every cell a table of IC_MAX_OUT nodes, plus a fixed reserve for the
        // cells of the slow path -- once, the launch covers every cell and the kernels read the record counts from HBM.  That worst case
        // is ~1.5 KB per cell where the tables use ~0.5 KB, and the pool is grow-only and per context: beyond 4 GB of worst case (~2.7 M
        // cells, e.g. 1500 walkers of the configs[3] grid) the wait is taken after all and the pool sized by what the plan handed out
        // (32 GB until round 5 -- several contexts, or torch allocations next to one, could run out of HBM to save 20 us).
        constexpr unsigned long long SLOW_RESERVE_NO_WAIT = 8ull << 20;  // doubles (64 MB): ~2000 cells of twice the fast kernel's lattices
        const unsigned long long worst = (unsigned long long)std::max<long long>(c->n_cells, 1) * IC_MAX_OUT + 1024 + SLOW_RESERVE_NO_WAIT;
        const bool no_wait = c->ic_soft_fail && worst * sizeof(double) <= (4ull << 30) && !vag_hook("VAG_IC_POOL_READBACK");
        int fast_nu_max = IC_MAX_NU;  // test hook: VAG_DEBUG_IC_FAST_NU_MAX=<n> sends the cells with longer seed lattices through the slow path (0: all)
        if (const char* e = vag_hook("VAG_DEBUG_IC_FAST_NU_MAX")) fast_nu_max = std::min(std::atoi(e), IC_MAX_NU);
        hipLaunchKernelGGL(vag_ic_plan_kernel, dim3((unsigned)((c->n_cells + 255) / 256)), dim3(256), 0, st, d_params, nb,
                           c->d_meta.as<VagGridMeta>(), lay, c->n_cells, c->d_celldet.as<double>(), c->d_band.as<double>(),
                           c->d_ichdr.as<double>(), c->d_icplan.as<double>(), c->d_icused.as<unsigned long long>(),
                           c->d_icstatus.as<int>(), c->count_work ? c->d_icwork.as<unsigned long long>() : nullptr, band_stride, d_need,
                           fast_nu_max, no_wait ? SLOW_RESERVE_NO_WAIT : ~0ull >> 1, c->d_icslow.as<int>(), IC_SLOW_LIST);
        HIPCHK(hipGetLastError());
        long long n_run = c->n_cells;  // launch size (an upper bound when the host has not seen the count)
        long long n_slow = -1;         // records on the slow path (-1: not seen)
        c->ic_slow_unread = no_wait;
        if (no_wait) {
            if (c->d_icpool.ensure(sizeof(double) * (size_t)worst)) return VAG_E_HIP;
        } else {
            unsigned long long ic_counters[4] = {0, 0, 0, 0};
            HIPCHK(hipMemcpyAsync(ic_counters, c->d_icused.p, sizeof ic_counters, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            const unsigned long long pool_used = ic_counters[0];
            n_run = (long long)ic_counters[1];  // cells that get a table: one plan record each
            n_slow = (long long)std::min<unsigned long long>(ic_counters[3], IC_SLOW_LIST);
            c->plan.n_ssc_slow_cells += n_slow;
            if (c->d_icpool.ensure(sizeof(double) * (size_t)(pool_used + pool_used / 16 + 1024))) return VAG_E_HIP;
            c->plan.ic_pool_bytes = std::max<long long>(c->plan.ic_pool_bytes, (long long)(sizeof(double) * pool_used));
        }
        // one wavefront per plan record (= per cell that gets a table); -DVAG_IC_PERSISTENT=1 developer builds: as many wavefronts as the
        // device holds at the kernel's four per SIMD, each taking the records w, w + G, ...
        long long ic_waves = n_run;
#if VAG_IC_PERSISTENT
        ic_waves = (long long)c->n_cus * 4 * VAG_IC_WAVES;
        if (const char* e = vag_hook("VAG_IC_GRID")) ic_waves = std::max(1, std::atoi(e));  // developer aid
#endif
        if (n_run > 0)
        hipLaunchKernelGGL(vag_ic_photon_kernel, dim3((unsigned)std::min<long long>(n_run, ic_waves)), dim3(64), 0, st,
                           IcPhotonArgs{n_run, no_wait ? c->d_icused.as<unsigned long long>() + 1 : nullptr, c->n_cells, c->d_icy.as<double>(),
                                        c->d_cellpar.as<double>(), c->d_cellq.as<double>(), c->d_sptab.as<double>(), c->d_knlut.as<double>(),
                                        c->d_icplan.as<double>(), c->d_icpool.as<double>()});
        HIPCHK(hipGetLastError());
        // the cells beyond the fast kernel's on-chip layout: one wavefront per listed record (a call that did not wait for the count
        // launches a small fixed grid that finds the list empty)
        if (n_slow != 0) {
            const IcPhotonArgs a{n_run, nullptr, c->n_cells, c->d_icy.as<double>(), c->d_cellpar.as<double>(), c->d_cellq.as<double>(),
                                 c->d_sptab.as<double>(), c->d_knlut.as<double>(), c->d_icplan.as<double>(), c->d_icpool.as<double>()};
            hipLaunchKernelGGL(vag_ic_photon_slow_kernel, dim3((unsigned)(n_slow > 0 ? std::min<long long>(n_slow, 8192) : 128)), dim3(64), 0, st,
                               a, c->d_icslow.as<int>(), c->d_icused.as<unsigned long long>() + 3, IC_SLOW_LIST);
            HIPCHK(hipGetLastError());
        }
        if (c->count_work) {
            unsigned long long h[2] = {0, 0};
            HIPCHK(hipMemcpyAsync(h, c->d_icwork.p, sizeof h, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            c->plan.ic_terms += (long long)h[0];
            c->plan.ic_nodes += (long long)h[1];
        }
    }
    return VAG_OK;
}

// > 0: not an error code of the C-ABI -- a flux pass breached a clamped band, the caller rebuilds those tables unclamped and repeats
constexpr int VAG_IC_REBUILD = 1000;
// > 0 as well: a flux pass queried a cell the lazy selection (vag_ic_band_kernel's range test: a superset of the queried cells by
// construction, but round 4's sweeps met two holes in it) had left without a table.  The caller builds EVERY cell's table and repeats the
// pass -- the answer the reference gives -- and the event is counted (vag_plan.n_ssc_all_cell_fallbacks); VAG_E_INTERNAL only if the
// all-cells pass reports a missing table too.
constexpr int VAG_IC_REBUILD_ALL = 1001;
int check_ic_status(vag_ctx* c, int nb) {
    if (c->ic_soft_fail) return VAG_OK;  // vag_fit_back_kernel folds d_icstatus into the walker's validity (-inf), samplers.py:61-70
    hipStream_t st = c->stream;
    std::vector<int> h(nb);
    HIPCHK(hipMemcpyAsync(h.data(), c->d_icstatus.p, sizeof(int) * (size_t)nb, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    bool breach = false, hole = false;
    for (int m = 0; m < nb; ++m) {
        if (h[m] & 1) return set_err(VAG_E_CAPACITY, "model %d: SSC lattices exceed the engine limits", m);
        if (h[m] & 4) {
            if (c->ic_all_cells) return set_err(VAG_E_INTERNAL, "model %d: a flux pass queried an SSC cell that was given no table", m);
            hole = true;
        }
    }
    if (hole) {
        ++c->plan.n_ssc_all_cell_fallbacks;
        if (vag_hook("VAG_DEBUG_IC_NO_FALLBACK"))  // test hook: the loud answer itself
            return set_err(VAG_E_INTERNAL, "a flux pass queried an SSC cell that was given no table (fallback disabled)");
        return VAG_IC_REBUILD_ALL;
    }
    for (int m = 0; m < nb; ++m)
        if (h[m] & 2) {
            breach = true;
            ++c->plan.n_models_ssc_rebuilt;
        }
    return breach ? VAG_IC_REBUILD : VAG_OK;
}

// The attempts of one SSC flux pass: lazily selected tables -> (a hole in the selection: every cell's table) -> (a breach of a
// clamped band: those models' tables unclamped).  `pass(rebuild)` builds the tables and runs the flux pass; returns the C-ABI code.
template <class Pass>
int ssc_attempts(vag_ctx* c, int nb, Pass pass) {
    struct Reset {
        vag_ctx* c;
        ~Reset() { c->ic_all_cells = false; }
    } reset{c};
    bool rebuild = false;
    int rc = VAG_OK;
    for (int attempt = 0; attempt < 3; ++attempt) {
        rc = pass(rebuild);
        if (rc) return rc;
        rc = check_ic_status(c, nb);
        if (rc == VAG_IC_REBUILD_ALL) {
            c->ic_all_cells = true;
            rebuild = false;
        } else if (rc == VAG_IC_REBUILD) {
            rebuild = true;
        } else {
            return rc;
        }
    }
    return set_err(VAG_E_NUMERIC, "SSC query outside the band of an unclamped table");
}

// d_lg2nu_all / nnu_all: every frequency of the request (the seed band is clamped over all of them, also when the
// frequency axis is evaluated in chunks)
int run_flux_ssc(vag_ctx* c, const vag_model_params* d_params, int nb, const double* d_lg2t, int nt, const double* d_lg2nu,
                 int nnu, const double* d_bandw, double* d_ssc, const double* d_lg2nu_all, int nnu_all) {
    return ssc_attempts(c, nb, [&](bool rebuild) {
        const int rc = build_ssc_tables(c, d_params, nb, d_lg2nu_all, nnu_all, rebuild);
        return rc ? rc : run_flux_grid(c, d_params, nb, d_lg2t, nt, d_lg2nu, nnu, d_bandw, d_ssc, FLUX_SSC);
    });
}

// Synchrotron (IC-cooled) and SSC components of one emitter in ONE flux pass: the tables are built first, then both
// spectra ride the same EAT logs, bracket search and barriers (the two-pass form pays that row skeleton twice -- 35-45 %
// of a pass on the C5 / C3 shapes).  Falls back to two passes when the doubled buffers would not fit in LDS twice.
static bool fused_fits(vag_ctx* c, int nt, int nnu) {
    if (vag_hook("VAG_NO_FUSED")) return false;
    if (vag_hook("VAG_GRID_ROWWISE") && nt * nnu <= SERIES_THREADS * SERIES_MAX_SLOTS && nnu <= SERIES_MAX_BANDS && !c->count_work)
        return false;  // experiment switch of run_flux_grid: two passes of the wavefront-per-row kernel
    // the second set of buffers must not cost a resident workgroup: on the C5 / C3 shapes it does (63 vs 51 KB: two
    // workgroups per CU instead of three) and the fused pass measured 18 % SLOWER than two passes there
    const size_t cu = 160 * 1024, fused = flux_grid_lds_bytes(FLUX_FUSED, flux_ks(c), nt, nnu),
                 two = flux_grid_lds_bytes(FLUX_SYN_IC, flux_ks(c), nt, nnu);
    if (vag_hook("VAG_FORCE_FUSED")) return fused <= cu;
    return fused <= cu && std::min<size_t>(cu / fused, 4) >= std::min<size_t>(cu / two, 4);
}
int run_flux_fused(vag_ctx* c, const vag_model_params* d_params, int nb, const double* d_lg2t, int nt, const double* d_lg2nu,
                   int nnu, const double* d_bandw, double* d_syn, double* d_ssc, const double* d_lg2nu_all, int nnu_all) {
    return ssc_attempts(c, nb, [&](bool rebuild) {
        const int rc = build_ssc_tables(c, d_params, nb, d_lg2nu_all, nnu_all, rebuild);
        return rc ? rc : run_flux_grid(c, d_params, nb, d_lg2t, nt, d_lg2nu, nnu, d_bandw, d_syn, FLUX_FUSED, d_ssc);
    });
}

__global__ void vag_copy_kernel(double* __restrict__ out, const double* __restrict__ src, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = src[i];
}

// One request on a (t, nu) grid (or a band when d_bandw != nullptr) for the whole batch; model stages already run.
// Components in FluxDict order {fwd.sync, fwd.ssc, rvs.sync, rvs.ssc} (pybind/pybind.cpp:472-483):
// d_comp (optional) -> d_comp[i] != nullptr receives component i, zeros when that component is disabled;
// d_total (optional) receives the sum of the enabled components in PyFlux::calc_total order (pymodel.cpp:350-364).
int grid_request(vag_ctx* c, const vag_model_params* d_params, int nb, int nt, int nnu, const double* d_bandw, double* d_total,
                 double* const* d_comp, int t_off = 0, int nu_off = 0, int nnu_all = 0) {
    // t_off / nu_off: first requested time / frequency of this call, nnu_all: frequencies of the whole request (chunking)
    if (nnu_all == 0) nnu_all = nnu;
    const double* lg2nu_all = c->d_lg2nu.as<double>();
    const double* lg2nu = lg2nu_all + nu_off;
    const size_t n_out = (size_t)nb * (d_bandw ? nt : (size_t)nt * nnu);
    const int n_em = (c->batch_flags & VAG_FLAG_RVS) ? 2 : 1;
    bool first = true;
    int rc = VAG_OK;
    for (int e = 0; e < 2 && rc == VAG_OK; ++e) {
        if (e < n_em) select_emitter(c, e, d_params);
        double* fused_ssc = nullptr;  // set once the fused pass has produced this emitter's SSC component
        for (int pass = 0; pass < 2 && rc == VAG_OK; ++pass) {
            double* dst_comp = d_comp ? d_comp[2 * e + pass] : nullptr;
            const bool enabled = e < n_em && (pass == 0 || c->cur_ssc);
            if (!enabled) {
                if (dst_comp && hipMemsetAsync(dst_comp, 0, sizeof(double) * n_out, c->stream) != hipSuccess) rc = VAG_E_HIP;
                continue;
            }
            double* dst = dst_comp;
            if (pass == 1 && fused_ssc) {
                dst = fused_ssc;
            } else {
                if (!dst) {
                    if (!d_total) continue;  // nobody wants this component
                    if (first) {
                        dst = d_total;
                    } else {
                        if (c->d_ssc.ensure(sizeof(double) * n_out)) {
                            rc = VAG_E_HIP;
                            break;
                        }
                        dst = c->d_ssc.as<double>();
                    }
                }
                const double* lg2t = c->d_lg2t.as<double>() + t_off;
                double* want_ssc = d_comp ? d_comp[2 * e + 1] : nullptr;
                if (pass == 0 && c->cur_ssc && (want_ssc || d_total) && fused_fits(c, nt, nnu)) {
                    if (!want_ssc) {
                        if (c->d_ssc2.ensure(sizeof(double) * n_out)) {
                            rc = VAG_E_HIP;
                            break;
                        }
                        want_ssc = c->d_ssc2.as<double>();
                    }
                    rc = run_flux_fused(c, c->cur_params, nb, lg2t, nt, lg2nu, nnu, d_bandw, dst, want_ssc, lg2nu_all, nnu_all);
                    fused_ssc = want_ssc;
                } else if (pass == 0) {
                    rc = run_flux_grid(c, c->cur_params, nb, lg2t, nt, lg2nu, nnu, d_bandw, dst, c->cur_ssc ? FLUX_SYN_IC : FLUX_SYN);
                } else {
                    rc = run_flux_ssc(c, c->cur_params, nb, lg2t, nt, lg2nu, nnu, d_bandw, dst, lg2nu_all, nnu_all);
                }
                if (rc) break;
            }
            if (d_total) {
                if (first) {
                    if (dst != d_total) hipLaunchKernelGGL(vag_copy_kernel, dim3(256), dim3(256), 0, c->stream, d_total, dst, n_out);
                    first = false;
                } else {
                    hipLaunchKernelGGL(vag_add_kernel, dim3(256), dim3(256), 0, c->stream, d_total, dst, n_out);
                }
                if (hipGetLastError() != hipSuccess) rc = VAG_E_HIP;
            }
        }
    }
    select_emitter(c, 0, d_params);
    return rc;
}

// [nb][rows][n] chunk results -> their place in the [nb][rows_all][nt] output
__global__ void vag_place_kernel(double* __restrict__ dst, const double* __restrict__ src, int rows, int n, int rows_all, int nt,
                                 int row0, int t0, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % n);
        const size_t r = i / n;
        const int l = (int)(r % rows);
        const size_t m = r / rows;
        dst[(m * rows_all + row0 + l) * nt + t0 + j] = src[i];
    }
}

// grid_request with the frequency axis cut into chunks of <= VAG_MAX_NU and the time axis into chunks whose (nu, t)
// accumulator and boundary values fit the workgroup's LDS (<= 4096 slots, fewer when the lattice is long): outputs are
// assembled into [nb][rows][nt], rows = nnu (1 for a band).
int grid_request_chunked(vag_ctx* c, const vag_model_params* d_params, int nb, int nt, int nnu, const double* d_bandw,
                         double* d_total, double* const* d_comp) {
    const bool any_ssc = (c->batch_flags & (VAG_FLAG_SSC | VAG_FLAG_RVS_SSC)) != 0;
    // a band integrates all of its frequencies in one launch; a grid is cut into near-equal frequency chunks
    const int nu_parts = d_bandw ? 1 : (nnu + VAG_MAX_NU - 1) / VAG_MAX_NU;
    const int nu_chunk = (nnu + nu_parts - 1) / nu_parts;
    int chunk = std::max(1, 4096 / nu_chunk);
    while (chunk > 8 &&
           flux_grid_lds_bytes(any_ssc ? FLUX_SYN_IC : FLUX_SYN, flux_ks(c), std::min(chunk, nt), nu_chunk) > 160 * 1024)
        chunk >>= 1;
    if (nt <= chunk && nu_parts == 1) return grid_request(c, d_params, nb, nt, nnu, d_bandw, d_total, d_comp);
    chunk = std::min(chunk, nt);
    const int rows_all = d_bandw ? 1 : nnu;
    const size_t cap = (size_t)nb * (d_bandw ? 1 : nu_chunk) * chunk;
    DevBuf& tmp = c->d_chunk;  // grow-only context scratch: the stream orders its reuse
    if (tmp.ensure(sizeof(double) * 5 * cap)) return VAG_E_HIP;
    double* t_total = d_total ? tmp.as<double>() : nullptr;
    double* t_comp[4];
    for (int i = 0; i < 4; ++i) t_comp[i] = (d_comp && d_comp[i]) ? tmp.as<double>() + (size_t)(i + 1) * cap : nullptr;
    int rc = VAG_OK;
    for (int l0 = 0; l0 < nnu && rc == VAG_OK; l0 += nu_chunk) {
        const int nl = std::min(nu_chunk, nnu - l0);
        const int rows = d_bandw ? 1 : nl;
        for (int t0 = 0; t0 < nt && rc == VAG_OK; t0 += chunk) {
            const int n = std::min(chunk, nt - t0);
            rc = grid_request(c, d_params, nb, n, d_bandw ? nnu : nl, d_bandw, t_total, d_comp ? t_comp : nullptr, t0, l0, nnu);
            const size_t total = (size_t)nb * rows * n;
            auto put = [&](double* dst, const double* src) {
                if (rc != VAG_OK || !dst) return;
                hipLaunchKernelGGL(vag_place_kernel, dim3(256), dim3(256), 0, c->stream, dst, src, rows, n, rows_all, nt,
                                   d_bandw ? 0 : l0, t0, total);
                if (hipGetLastError() != hipSuccess) rc = VAG_E_HIP;
            };
            put(d_total, t_total);
            for (int i = 0; i < 4; ++i) put(d_comp ? d_comp[i] : nullptr, t_comp[i]);
        }
        if (d_bandw) break;
    }
    return rc;
}

// Fixed-order sum of the workgroup partials of a series request.  A workgroup is 64 points x 4 block groups: group g adds the
// partial blocks b = g, g + 4, ... with four loads in flight, then the four group sums are added in order -- the same
// result run to run, and the walk over up to a few hundred blocks is no longer one dependent load after another
// (63 us per 128-walker call before).
__global__ void __launch_bounds__(256)
vag_series_reduce_kernel(const vag_model_params* __restrict__ params, const VagGridMeta* __restrict__ meta,
                         const double* __restrict__ partial, int max_blocks, int pairs_per_block, int n,
                         double* __restrict__ out, int parts /* partial sums per block of rows */,
                         int* __restrict__ work_counter = nullptr /* of the persistent launch before this one: back to zero */) {
    __shared__ double s_part[4][64];
    const int m = blockIdx.y;
    if (work_counter && blockIdx.x == 0 && m == 0 && threadIdx.x == 0) *work_counter = 0;
    const VagGridMeta M = meta[m];
    const vag_model_params P = params[m];
    const int nblk = (M.status == 0) ? (M.n_theta * M.n_phi_eff + pairs_per_block - 1) / pairs_per_block * parts : 0;
    const double d_L = P.lumi_dist * U_CM;
    const double norm = (1 + P.z) / (d_L * d_L);
    const double* src = partial + (size_t)m * max_blocks * n;
    const int g = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int s = blockIdx.x * 64 + lane;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    if (s < n) {
        int b = g;
        for (; b + 12 < nblk; b += 16) {
            a0 += src[(size_t)b * n + s];
            a1 += src[(size_t)(b + 4) * n + s];
            a2 += src[(size_t)(b + 8) * n + s];
            a3 += src[(size_t)(b + 12) * n + s];
        }
        for (; b < nblk; b += 4) a0 += src[(size_t)b * n + s];
    }
    s_part[g][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (g == 0 && s < n) {
        const double v = (s_part[0][lane] + s_part[1][lane]) + (s_part[2][lane] + s_part[3][lane]);
        out[(size_t)m * n + s] = (M.status == 0) ? (v * norm) / U_FLUX_DEN_CGS : NAN;
    }
}

// resident workgroups per CU of a series launch with `waves` wavefronts and `lds` bytes (occupancy query, remembered per shape)
static int series_wg_per_cu(vag_ctx* c, int waves, size_t lds) {
    for (const auto& e : c->series_occ)
        if (e.waves == waves && e.lds == lds) return e.wg;
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, vag_flux_series_kernel<FLUX_SYN, false, 1>, SERIES_THREADS * waves, lds) !=
        hipSuccess)
        occ = (int)((144 * 1024) / std::max<size_t>(lds, 1));
    if (c->series_occ.size() > 256) c->series_occ.clear();
    c->series_occ.push_back({waves, lds, occ});
    return occ;
}

int run_flux_series(vag_ctx* c, const vag_model_params* d_params, int nb, const double* d_lg2t, const double* d_lg2nu,
                    int n, double* d_out, int mode, int n_bands, int grid_nt) {
    hipStream_t st = c->stream;
    StageScope ps(c, mode == FLUX_SSC ? PS_SSC_FLUX : PS_SYNC_FLUX);
    if (n > SERIES_THREADS * SERIES_MAX_SLOTS)
        return set_err(VAG_E_CAPACITY, "series length %d exceeds %d", n, SERIES_THREADS * SERIES_MAX_SLOTS);
    // A fit's shape (plain synchrotron, <= 64 points in a few bands): the row-per-lane kernel (vag_fit_rows.h)
    if (mode != FLUX_FUSED && grid_nt == 0 && n <= FITROWS_MAX_POINTS && n_bands > 0 &&
        n_bands <= FITROWS_BANDS && !vag_hook("VAG_SERIES_ROW_PER_WAVE")) {
        const int max_blocks = std::max(1, (c->max_pairs + FITROWS_ROWS - 1) / FITROWS_ROWS);
        if (c->d_partial.ensure(sizeof(double) * (size_t)nb * max_blocks * FITROWS_SEGS * n)) return VAG_E_HIP;
        SeriesArgs a{};
        a.params = d_params;
        a.meta = c->d_meta.as<VagGridMeta>();
        a.geo_th = c->d_geo_th.as<double>();
        a.geo_ph = c->d_geo_ph.as<double>();
        a.g_rep_of = c->d_rep_of.as<int>();
        a.lay = Layout{c->d_row_off.as<int>(), c->d_cell_off.as<long long>()};
        a.cellpar = c->d_cellpar.as<double>();
        a.lg2_t_obs = d_lg2t;
        a.lg2_nu_obs = d_lg2nu;
        a.n = n;
        a.pairs_per_block = FITROWS_ROWS;
        a.max_blocks = max_blocks;
        a.max_chunks = max_blocks * FITROWS_SEGS;  // one partial sum per (block of 64 rows, lattice segment)
        a.chunk = FITROWS_ROWS;
        a.partial = c->d_partial.as<double>();
        a.sp_table = c->d_sptab.as<double>();
        a.n_bands = n_bands;
        a.band_idx = c->d_bandidx.as<int>();
        a.band_first = c->d_bandidx.as<int>() + FITROWS_MAX_POINTS;
        a.cellq = c->d_cellq.as<double>();
        a.ichdr = c->d_ichdr.as<double>();
        a.icpool = c->d_icpool.as<double>();
        a.ic_status = c->d_icstatus.as<int>();
        // (+=: a call with several series passes or chunks reports all of them; the plan is reset by the model stage of the call)
        const long long upper_evals = 2 * c->total_pairs * (long long)n, upper_interps = c->total_pairs * (long long)n;
        c->plan.spec_evals += upper_evals;
        c->plan.interps += upper_interps;
        c->plan.flux_blocks = max_blocks * nb;
        c->plan.pairs_per_block = FITROWS_ROWS;
        if (c->n_rows > 0) {
            // wavefronts per block of 64 rows: four while the batch leaves the GPU room (each walks a quarter of the lattice),
            // one when there are blocks enough to fill it (one prologue per block).  The partial sums are the same either way.
            const long long blocks = (c->total_pairs + FITROWS_ROWS - 1) / FITROWS_ROWS;
            int wpb = blocks <= 1024 ? 4 : (blocks <= 4608 ? 2 : 1);  // measured (persistent launch, 4 / 2 / 1): 0.8 k blocks 0.072 / 0.088 / 0.140 ms, 1.6 k 0.115 / 0.105 / 0.142, 6.2 k 0.333 / 0.267 / 0.269
            if (const char* e = vag_hook("VAG_FIT_WAVES_PER_BLOCK")) wpb = std::atoi(e) == 4 ? 4 : (std::atoi(e) == 2 ? 2 : 1);
            a.grid_nt = wpb;
            a.nb = nb;
            a.work = work_counters(c);
            // persistent workgroups: as many as the GPU holds (three per CU), fewer when there are fewer items (blocks x wavefronts per block)
            const long long items = (blocks + nb) * wpb;  // (total_pairs may be the previous call's: every model can own one block more)
            const int wgs = (int)std::max<long long>(1, std::min<long long>((items + FITROWS_WAVES - 1) / FITROWS_WAVES, 3LL * c->n_cus));
            const dim3 g(wgs), b(SERIES_THREADS * FITROWS_WAVES);
            const size_t lds = fit_rows_lds_bytes(n);
            const bool spread = (c->batch_flags & VAG_FLAG_SPREADING) != 0;
            a.cellgeo = c->d_cellgeo.as<double>();
            // vag_ctx_count_work: the tallying instantiation (plain synchrotron, <= 4 bands, no spreading: the walker metric's case)
            const bool tally = c->count_work && mode == FLUX_SYN && !spread && n_bands <= 4;
            if (tally) {
                if (c->d_workcount.ensure(2 * sizeof(unsigned long long))) return VAG_E_HIP;
                HIPCHK(hipMemsetAsync(c->d_workcount.p, 0, 2 * sizeof(unsigned long long), st));
                a.tally = c->d_workcount.as<unsigned long long>();
                hipLaunchKernelGGL((vag_flux_fit_rows_kernel<FLUX_SYN, 4, false, true>), g, b, lds, st, a);
                HIPCHK(hipGetLastError());
                unsigned long long hcount[2] = {0, 0};
                HIPCHK(hipMemcpyAsync(hcount, c->d_workcount.p, sizeof hcount, hipMemcpyDeviceToHost, st));
                HIPCHK(hipStreamSynchronize(st));
                c->plan.spec_evals += (long long)hcount[0] - upper_evals;  // the tallied counts replace this pass's upper bounds
                c->plan.interps += (long long)hcount[1] - upper_interps;
            } else
#define VAG_FIT_LAUNCH(M_)                                                                      \
    do {                                                                                        \
        if (spread && n_bands <= 4)                                                             \
            hipLaunchKernelGGL((vag_flux_fit_rows_kernel<M_, 4, true>), g, b, lds, st, a);       \
        else if (spread)                                                                        \
            hipLaunchKernelGGL((vag_flux_fit_rows_kernel<M_, 8, true>), g, b, lds, st, a);       \
        else if (n_bands <= 4)                                                                  \
            hipLaunchKernelGGL((vag_flux_fit_rows_kernel<M_, 4>), g, b, lds, st, a);             \
        else                                                                                    \
            hipLaunchKernelGGL((vag_flux_fit_rows_kernel<M_, 8>), g, b, lds, st, a);             \
    } while (0)
            {
            if (mode == FLUX_SYN_IC)
                VAG_FIT_LAUNCH(FLUX_SYN_IC);
            else if (mode == FLUX_SYN)
                VAG_FIT_LAUNCH(FLUX_SYN);
            else
                VAG_FIT_LAUNCH(FLUX_SSC);
            }
#undef VAG_FIT_LAUNCH
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipEventRecord(c->ev[4], st));
        hipLaunchKernelGGL(vag_series_reduce_kernel, dim3((n + 63) / 64, nb), dim3(256), 0, st, d_params, c->d_meta.as<VagGridMeta>(),
                           c->d_partial.as<double>(), max_blocks * FITROWS_SEGS, FITROWS_ROWS, n, d_out, FITROWS_SEGS, work_counters(c));
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(c->ev[5], st));
        return VAG_OK;
    }
    // series work per (theta, phi) row is small: fewer, longer wavefronts as the batch grows.  The partial sums are kept per
    // SERIES_CHUNK rows whatever this choice is, so a model's result does not depend on what else is in the batch (a walker
    // scores the same bits alone, in a block of 64, or among 1024).
    long long ppb = std::max<long long>(SERIES_CHUNK, (c->total_pairs + 32767) / 32768);
    ppb = (ppb + SERIES_CHUNK - 1) / SERIES_CHUNK * SERIES_CHUNK;
    if (const char* e = vag_hook("VAG_SERIES_PPB")) ppb = std::max(1, std::atoi(e)) * SERIES_CHUNK;  // tuning override (in chunks)
    const int max_blocks = std::max(1, (int)((c->max_pairs + ppb - 1) / ppb));
    // a small (nu, t) grid served by this kernel keeps one partial per wavefront (its 128 x 128 rows would need thousands of
    // 8-row chunks per model); a grid request makes no batch-independence promise, as the workgroup kernel does not either
    const int chunk = grid_nt > 0 ? (int)ppb : SERIES_CHUNK;
    const int max_chunks = std::max(1, (c->max_pairs + chunk - 1) / chunk);
    if (c->d_partial.ensure(sizeof(double) * (size_t)nb * max_chunks * n)) return VAG_E_HIP;
    const int ks = flux_ks(c);
    if (n > SERIES_THREADS && grid_nt == 0) n_bands = 0;  // a fit's shared-node path keeps one point per lane
    // wavefronts per workgroup: four when their private rows fit next to the shared tables, else two or one
    auto lds_for = [&](int w) {
        return sizeof(double) * ((size_t)w * series_region_doubles(ks, mode == FLUX_SYN_IC, n_bands) + SP_LDS_DOUBLES + SERIES_MAX_BANDS);
    };
    // the workgroup shape that keeps the most wavefronts resident on a CU (each workgroup carries one copy of the tables);
    // ties go to the smaller workgroup
    // (the runtime's occupancy query decides what fits: two 80 KB workgroups do not share a CU's 160 KB)
    int waves = 1, best = 0;
    for (int w = 1; w <= SERIES_WAVES; w <<= 1) {
        if (lds_for(w) > 160 * 1024) break;
        const int resident = std::min(series_wg_per_cu(c, w, lds_for(w)) * w, 16);
        if (resident >= best) best = resident, waves = w;
    }
    if (const char* e = vag_hook("VAG_SERIES_WAVES")) waves = std::max(1, std::min(SERIES_WAVES, std::atoi(e)));
    const size_t lds = lds_for(waves);
    if (lds > 160 * 1024) return set_err(VAG_E_CAPACITY, "LDS request %zu B exceeds 160 KiB (n_t=%d)", lds, ks);
    const dim3 sgrid((max_blocks + waves - 1) / waves, nb), sblock(SERIES_THREADS * waves);
    if (vag_hook("VAG_DEBUG_LAUNCH")) {
        int occ = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, vag_flux_series_kernel<FLUX_SYN, false, 1>, SERIES_THREADS * waves, lds);
        std::fprintf(stderr, "[vag] series launch: n=%d bands=%d max_k=%d rows/wave=%lld waves/wg=%d lds=%zu B grid=(%u,%u) wg/CU=%d\n", n,
                     n_bands, ks, ppb, waves, lds, sgrid.x, sgrid.y, occ);
    }
    SeriesArgs a{};
    a.cellq = c->d_cellq.as<double>();
    a.ichdr = c->d_ichdr.as<double>();
    a.icpool = c->d_icpool.as<double>();
    a.ic_status = c->d_icstatus.as<int>();
    a.cellgeo = c->d_cellgeo.as<double>();
    a.n_bands = n_bands;
    a.grid_nt = grid_nt;
    a.chunk = chunk;
    a.band_idx = c->d_bandidx.as<int>();
    a.band_first = c->d_bandidx.as<int>() + FITROWS_MAX_POINTS;
    const bool spreading = (c->batch_flags & VAG_FLAG_SPREADING) != 0;
    a.params = d_params;
    a.meta = c->d_meta.as<VagGridMeta>();
    a.geo_th = c->d_geo_th.as<double>();
    a.geo_ph = c->d_geo_ph.as<double>();
    a.g_rep_of = c->d_rep_of.as<int>();
    a.lay = Layout{c->d_row_off.as<int>(), c->d_cell_off.as<long long>()};
    a.cellpar = c->d_cellpar.as<double>();
    a.lg2_t_obs = d_lg2t;
    a.lg2_nu_obs = d_lg2nu;
    a.n = n;
    a.pairs_per_block = (int)ppb;
    a.max_blocks = max_blocks;
    a.max_chunks = max_chunks;
    a.k_stride = ks;
    a.partial = c->d_partial.as<double>();
    a.sp_table = c->d_sptab.as<double>();
    c->plan.spec_evals = 2 * c->total_pairs * (long long)n;
    c->plan.interps = c->total_pairs * (long long)n;
    c->plan.flux_blocks = max_blocks * nb;
    c->plan.pairs_per_block = (int)ppb;
    if (c->n_rows > 0) {
        const bool one = n <= SERIES_THREADS;  // one data point per lane
#define VAG_SERIES_LAUNCH(M_, S_)                                                                              \
    do {                                                                                                       \
        if (grid_nt > 0)                                                                                       \
            hipLaunchKernelGGL((vag_flux_series_kernel<M_, false, SERIES_MAX_SLOTS, true>), sgrid, sblock, lds, st, a); \
        else if (one)                                                                                          \
            hipLaunchKernelGGL((vag_flux_series_kernel<M_, S_, 1>), sgrid, sblock, lds, st, a);                 \
        else                                                                                                   \
            hipLaunchKernelGGL((vag_flux_series_kernel<M_, S_, SERIES_MAX_SLOTS>), sgrid, sblock, lds, st, a);  \
    } while (0)
        if (spreading && mode == FLUX_SYN_IC)
            VAG_SERIES_LAUNCH(FLUX_SYN_IC, true);
        else if (spreading && mode == FLUX_SSC)
            VAG_SERIES_LAUNCH(FLUX_SSC, true);
        else if (spreading)
            VAG_SERIES_LAUNCH(FLUX_SYN, true);
        else if (mode == FLUX_SYN_IC)
            VAG_SERIES_LAUNCH(FLUX_SYN_IC, false);
        else if (mode == FLUX_SSC)
            VAG_SERIES_LAUNCH(FLUX_SSC, false);
        else
            VAG_SERIES_LAUNCH(FLUX_SYN, false);
#undef VAG_SERIES_LAUNCH
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(c->ev[4], st));
    hipLaunchKernelGGL(vag_series_reduce_kernel, dim3((n + 63) / 64, nb), dim3(256), 0, st, d_params,
                       c->d_meta.as<VagGridMeta>(), c->d_partial.as<double>(), max_chunks, chunk, n, d_out, 1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(c->ev[5], st));
    return VAG_OK;
}

// One chunk of a (t, nu) series for the batch: the sum of every enabled component -> d_out[nb][n].  The comoving band
// of an SSC table spans ALL requested frequencies (d_lg2nu_all[n_all], pymodel.h:896-909), not only this chunk's.
// One chunk of a (t, nu) series for the batch.  d_out (optional) [nb][n] receives the sum of every enabled component in
// PyFlux::calc_total order; d_comp (optional) -> d_comp[i] != nullptr receives component i of {fwd.sync, fwd.ssc, rvs.sync,
// rvs.ssc} [nb][n], zeros when that component is disabled.
int series_chunk(vag_ctx* c, const vag_model_params* d_params, int nb, const double* d_lg2t, const double* d_lg2nu, int n,
                 const double* d_lg2nu_all, int n_all, double* d_out, int n_bands = 0, double* const* d_comp = nullptr) {
    const int n_em = (c->batch_flags & VAG_FLAG_RVS) ? 2 : 1;
    const size_t n_out = (size_t)nb * n;
    int rc = VAG_OK;
    bool first = true;
    for (int e = 0; e < 2 && rc == VAG_OK; ++e) {
        if (e < n_em) select_emitter(c, e, d_params);
        for (int pass = 0; pass < 2 && rc == VAG_OK; ++pass) {
            double* dst_comp = d_comp ? d_comp[2 * e + pass] : nullptr;
            const bool enabled = e < n_em && (pass == 0 || c->cur_ssc);
            if (!enabled) {
                if (dst_comp && hipMemsetAsync(dst_comp, 0, sizeof(double) * n_out, c->stream) != hipSuccess) rc = VAG_E_HIP;
                continue;
            }
            double* dst = dst_comp;
            if (!dst) {
                if (!d_out) continue;  // nobody wants this component
                if (first) {
                    dst = d_out;
                } else {
                    if (c->d_ssc.ensure(sizeof(double) * n_out)) {
                        rc = VAG_E_HIP;
                        break;
                    }
                    dst = c->d_ssc.as<double>();
                }
            }
            if (pass == 0) {
                rc = run_flux_series(c, c->cur_params, nb, d_lg2t, d_lg2nu, n, dst, c->cur_ssc ? FLUX_SYN_IC : FLUX_SYN, n_bands);
            } else {
                rc = ssc_attempts(c, nb, [&](bool rebuild) {
                    const int rb = build_ssc_tables(c, c->cur_params, nb, d_lg2nu_all, n_all, rebuild);
                    return rb ? rb : run_flux_series(c, c->cur_params, nb, d_lg2t, d_lg2nu, n, dst, FLUX_SSC, n_bands);
                });
            }
            if (rc) break;
            if (d_out) {
                if (first) {
                    if (dst != d_out) hipLaunchKernelGGL(vag_copy_kernel, dim3(256), dim3(256), 0, c->stream, d_out, dst, n_out);
                } else {
                    hipLaunchKernelGGL(vag_add_kernel, dim3(256), dim3(256), 0, c->stream, d_out, dst, n_out);
                }
                if (hipGetLastError() != hipSuccess) rc = VAG_E_HIP;
            }
            first = false;
        }
    }
    select_emitter(c, 0, d_params);
    return rc;
}

int collect_times(vag_ctx* c) {
    HIPCHK(hipEventSynchronize(c->ev[5]));
    float v;
    HIPCHK(hipEventElapsedTime(&v, c->ev[0], c->ev[1]));
    c->times.grid_ms = v;
    HIPCHK(hipEventElapsedTime(&v, c->ev[1], c->ev[2]));
    c->times.dynamics_ms = v;
    HIPCHK(hipEventElapsedTime(&v, c->ev[2], c->ev[3]));
    c->times.cells_ms = v;
    HIPCHK(hipEventElapsedTime(&v, c->ev[3], c->ev[4]));
    c->times.flux_ms = v;
    HIPCHK(hipEventElapsedTime(&v, c->ev[4], c->ev[5]));
    c->times.reduce_ms = v;
    HIPCHK(hipEventElapsedTime(&v, c->ev[0], c->ev[5]));
    c->times.total_ms = v;
    return VAG_OK;
}

}  // namespace
static int collect_times_fwd(vag_ctx* c) { return collect_times(c); }
namespace {

int prep_times(vag_ctx* c, const double* d_t, int nt, const double* d_nu, int nnu) {
    if (c->d_lg2t.ensure(sizeof(double) * nt)) return VAG_E_HIP;
    if (c->d_lg2nu.ensure(sizeof(double) * std::max(1, nnu))) return VAG_E_HIP;
    if (c->d_tminmax.ensure(sizeof(double) * 2)) return VAG_E_HIP;
    hipLaunchKernelGGL(vag_prep_kernel, dim3(1), dim3(256), 0, c->stream, d_t, nt, d_nu, nnu, c->d_lg2t.as<double>(),
                       c->d_lg2nu.as<double>(), c->d_tminmax.as<double>());
    HIPCHK(hipGetLastError());
    return VAG_OK;
}

int check_host_inputs(const vag_model_params* params, int nb, const double* t, int nt) {
    if (!params || !t) return set_err(VAG_E_INVALID, "null model or time array");
    if (nb <= 0) return set_err(VAG_E_INVALID, "batch must be non-empty");
    if (nt <= 0) return set_err(VAG_E_INVALID, "time array must be non-empty");
    for (int i = 0; i < nt; ++i)
        if (!(t[i] > 0) || !std::isfinite(t[i])) return set_err(VAG_E_INVALID, "times must be positive and finite");
    for (int i = 1; i < nt; ++i)
        if (t[i] < t[i - 1]) return set_err(VAG_E_INVALID, "time array must be in ascending order");
    for (int m = 0; m < nb; ++m) {
        const char* msg = validate_msg(&params[m]);
        if (msg) return set_err(VAG_E_INVALID, "model %d: %s", m, msg);
    }
    return VAG_OK;
}

// the batch's grid results on the host, copied only when someone needs more than the plan's totals
int fetch_meta(vag_ctx* c) {
    if (c->meta_on_host) return VAG_OK;
    if (c->h_meta.ensure(sizeof(VagGridMeta) * (size_t)c->nb)) return VAG_E_HIP;
    HIPCHK(hipMemcpyAsync(c->h_meta.p, c->d_meta.p, sizeof(VagGridMeta) * (size_t)c->nb, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->meta_on_host = true;
    return VAG_OK;
}

// End of a call that was planned from the previous call's summary (run_model_stages, spec): read this batch's own summary and
// verify the assumptions.  VAG_RETRY: a capacity was exceeded (the device then skipped every model) or the flags / kernel class
// differ -- the caller repeats the call on the waiting path, which also refreshes the hint.
constexpr int VAG_RETRY = 1;
int finish_speculation(vag_ctx* c) {
    if (!c->spec_pending) return VAG_OK;
    c->spec_pending = false;
    VagDevPlan* hp = c->h_plan.as<VagDevPlan>();
    if (int rcw = wait_plan(c)) return rcw;  // the grid stage finished long ago: the later stages keep running behind this check
    const int flags = hp->flags_first < 0 ? 0 : hp->flags_first, hflags = c->hint.flags_first < 0 ? 0 : c->hint.flags_first;
    if (hp->overflow || hp->flags_mixed || (hp->n_ok > 0 && (flags != hflags || hp->dyn_class != c->hint.dyn_class))) {
        if (vag_hook("VAG_DEBUG_SPEC"))
            std::fprintf(stderr, "[vag] planned-ahead call repeated: overflow %d mixed %d flags %d/%d dyn %d/%d rows %d cells %lld max_k %d (cap %d) max_pairs %d; hint rows %d cells %lld max_k %d max_pairs %d\n",
                         hp->overflow, hp->flags_mixed, flags, hflags, hp->dyn_class, c->hint.dyn_class, hp->rows, (long long)hp->cells,
                         hp->max_k, c->spec_cap_k, hp->max_pairs, c->hint.rows, (long long)c->hint.cells, c->hint.max_k, c->hint.max_pairs);
        if (hp->overflow && hp->max_k > c->spec_cap_k)  // the lattice outgrew its head room: plan the next calls with more
            c->spec_margin_k = std::min(32, 2 * c->spec_margin_k + 2);
        c->hint_valid = false;
        return VAG_RETRY;
    }
    c->hint = *hp;
    c->hint_valid = hp->n_ok > 0;
    c->n_ok = hp->n_ok;
    c->plan.n_models_ok = hp->n_ok;
    c->plan.n_rows = hp->rows;
    c->plan.n_cells = hp->cells;
    c->plan.total_pairs = hp->pairs;
    c->plan.eat_cells = hp->eat;
    c->plan.n_models_invalid = hp->n_invalid;
    c->plan.n_models_capacity = hp->n_capacity;
    return VAG_OK;
}

int check_status(vag_ctx* c, int nb) {
    const int rf = read_row_failures(c);
    if (rf) return rf;
    if (c->plan.n_rows_failed > 0)
        return set_err(VAG_E_NUMERIC, "%d ODE row(s): no acceptable step size after 500 rejections", c->plan.n_rows_failed);
    if (c->plan.n_models_capacity == 0) return VAG_OK;
    if (fetch_meta(c)) return VAG_E_HIP;
    const VagGridMeta* hm = c->h_meta.as<VagGridMeta>();
    for (int m = 0; m < nb; ++m)
        if (hm[m].status == VAG_E_CAPACITY)
            return set_err(VAG_E_CAPACITY, "model %d: adaptive grid exceeds engine limits (theta %d, phi %d, time %d)", m,
                           VAG_HUGE_THETA, VAG_HUGE_PHI, VAG_MAX_TIME);
    return VAG_OK;
}

}  // namespace

// ---- batches whose models carry different Radiation / shock flags (ssc, kn, rvs, spreading, magnetar, non-axisymmetric) ----
// One launch sequence serves one flag set (the flags pick kernels and passes).  The reference evaluates any mix of models
// side by side (samplers.py:59-91), so a mixed batch handed over the host-pointer API is split by flags, every group runs as
// its own batch -- same entry point, same arithmetic as if the caller had split it --, and the groups' rows are put back in
// place.  OutRows: one output array of `stride` doubles per model (NULL: not requested).
struct OutRows {
    double* p;
    size_t stride;
};
static bool uniform_flags(const vag_model_params* params, int nb) {
    for (int m = 1; m < nb; ++m)
        if (params[m].flags != params[0].flags) return false;
    return true;
}
template <class Call>
static int run_flag_groups(const vag_model_params* params, int nb, const std::vector<OutRows>& outs, Call call) {
    std::vector<int> keys;
    std::vector<std::vector<int>> groups;
    for (int m = 0; m < nb; ++m) {
        size_t g = 0;
        while (g < keys.size() && keys[g] != params[m].flags) ++g;
        if (g == keys.size()) {
            keys.push_back(params[m].flags);
            groups.emplace_back();
        }
        groups[g].push_back(m);
    }
    for (const std::vector<int>& idx : groups) {
        const int ng = (int)idx.size();
        std::vector<vag_model_params> gp(ng);
        for (int i = 0; i < ng; ++i) gp[i] = params[idx[i]];
        std::vector<std::vector<double>> bufs(outs.size());
        std::vector<double*> ptrs(outs.size(), nullptr);
        for (size_t q = 0; q < outs.size(); ++q)
            if (outs[q].p) {
                bufs[q].resize((size_t)ng * outs[q].stride);
                ptrs[q] = bufs[q].data();
            }
        const int rc = call(gp.data(), ng, ptrs);
        if (rc) return rc;
        for (size_t q = 0; q < outs.size(); ++q)
            if (outs[q].p)
                for (int i = 0; i < ng; ++i)
                    std::memcpy(outs[q].p + (size_t)idx[i] * outs[q].stride, bufs[q].data() + (size_t)i * outs[q].stride,
                                sizeof(double) * outs[q].stride);
    }
    return VAG_OK;
}

extern "C" {

}  // extern "C"

// ---- device-resident batches with mixed Radiation / shock flags (samplers.py:59-91 evaluates any mix of models side by side) ----
__global__ void vag_flags_kernel(const vag_model_params* __restrict__ params, int nb, int* __restrict__ flags) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m < nb) flags[m] = params[m].flags;
}
__global__ void vag_gather_params_kernel(const vag_model_params* __restrict__ params, const int* __restrict__ perm, int nb,
                                         vag_model_params* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nb) out[i] = params[perm[i]];
}
__global__ void vag_scatter_rows_kernel(const double* __restrict__ src, const int* __restrict__ perm, int nb, size_t stride,
                                        double* __restrict__ dst) {
    // gridDim.y is capped (65535): a workgroup row takes every gridDim.y-th model
    for (int i = blockIdx.y; i < nb; i += gridDim.y) {
        const double* s = src + (size_t)i * stride;
        double* d = dst + (size_t)perm[i] * stride;
        for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < stride; q += (size_t)gridDim.x * blockDim.x) d[q] = s[q];
    }
}

// A batch the grid pass refused for its mixed flags: sort the models by flags (stable: the flags travel to the host, 4 bytes per
// model, the permutation comes back), run every group as a batch of its own on the gathered parameters -- each model then comes
// out bitwise as in a call of its own group -- and scatter the rows back into the caller's order.  `body(d_params, n, d_out)` is
// the single-flag path of the entry point.
template <class Body>
static int flux_dev_by_flags(vag_ctx* c, const vag_model_params* d_params, int nb, size_t stride, double* d_out, Body body) {
    if (c->d_mix_flags.ensure(sizeof(int) * (size_t)nb) || c->d_mix_perm.ensure(sizeof(int) * (size_t)nb) ||
        c->d_mix_params.ensure(sizeof(vag_model_params) * (size_t)nb) || c->d_mix_out.ensure(sizeof(double) * (size_t)nb * stride))
        return VAG_E_HIP;
    hipStream_t st = c->stream;
    hipLaunchKernelGGL(vag_flags_kernel, dim3((nb + 255) / 256), dim3(256), 0, st, d_params, nb, c->d_mix_flags.as<int>());
    HIPCHK(hipGetLastError());
    std::vector<int> flags(nb), perm;
    HIPCHK(hipMemcpyAsync(flags.data(), c->d_mix_flags.p, sizeof(int) * (size_t)nb, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    std::vector<int> keys;
    std::vector<std::vector<int>> groups;
    for (int m = 0; m < nb; ++m) {
        size_t g = 0;
        while (g < keys.size() && keys[g] != flags[m]) ++g;
        if (g == keys.size()) {
            keys.push_back(flags[m]);
            groups.emplace_back();
        }
        groups[g].push_back(m);
    }
    perm.reserve(nb);
    for (const auto& g : groups) perm.insert(perm.end(), g.begin(), g.end());
    HIPCHK(hipMemcpyAsync(c->d_mix_perm.p, perm.data(), sizeof(int) * (size_t)nb, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));  // perm is a local: the copy must have left it
    hipLaunchKernelGGL(vag_gather_params_kernel, dim3((nb + 127) / 128), dim3(128), 0, st, d_params, c->d_mix_perm.as<int>(), nb,
                       c->d_mix_params.as<vag_model_params>());
    HIPCHK(hipGetLastError());
    size_t off = 0;
    int n_ok = 0, n_inv = 0, n_cap = 0;
    for (const auto& g : groups) {
        const int ng = (int)g.size();
        const int rc = body(c->d_mix_params.as<vag_model_params>() + off, ng, c->d_mix_out.as<double>() + off * stride);
        if (rc) return rc;
        n_ok += c->plan.n_models_ok, n_inv += c->plan.n_models_invalid, n_cap += c->plan.n_models_capacity;
        off += (size_t)ng;
    }
    c->plan.n_models_ok = n_ok, c->plan.n_models_invalid = n_inv, c->plan.n_models_capacity = n_cap;  // the batch's, not the last group's
    const unsigned gx = (unsigned)std::min<size_t>((stride + 255) / 256, 64);
    hipLaunchKernelGGL(vag_scatter_rows_kernel, dim3(gx, (unsigned)std::min(nb, 65535)), dim3(256), 0, st, c->d_mix_out.as<double>(),
                       c->d_mix_perm.as<int>(), nb, stride, d_out);
    HIPCHK(hipGetLastError());
    c->hint_valid = false;  // the hint is the LAST GROUP's plan: the next mixed batch of this size must not be planned from it
    return VAG_OK;
}

extern "C" {

static int grid_dev_body(vag_ctx* c, const vag_model_params* d_params, int nb, const double* d_t, int nt, const double* d_nu, int nnu,
                         double* d_out) {
    int rc = VAG_OK;
    for (int attempt = 0; attempt < 2; ++attempt) {  // planned ahead from the previous call's summary, verified at the end
        rc = prep_times(c, d_t, nt, d_nu, nnu);
        if (rc) return rc;
        c->allow_spec = attempt == 0 && !c->count_work;
        rc = run_model_stages(c, d_params, nb, false);
        c->allow_spec = false;
        if (rc == VAG_OK) rc = grid_request_chunked(c, d_params, nb, nt, nnu, nullptr, d_out, nullptr);
        if (rc) {
            if (!c->spec_pending) return rc;
            c->spec_pending = false;  // an error while planning from stale sizes is not the caller's: repeat on the waiting path
            c->hint_valid = false;
            if (vag_hook("VAG_DEBUG_SPEC")) std::fprintf(stderr, "[vag] planned-ahead grid call failed (%d: %s): repeating\n", rc, g_err.c_str());
            continue;
        }
        rc = finish_speculation(c);
        if (rc != VAG_RETRY) break;
    }
    return rc;
}

int vag_flux_density_grid_batch_dev(vag_ctx* c, const vag_model_params* d_params, int nb, const double* d_t, int nt,
                                    const double* d_nu, int nnu, double* d_out) {
    ApiLock api_lock(c);
    HandoffScope handoff(c);
    if (!c) return set_err(VAG_E_INVALID, "null context");
    if (!d_params || !d_t || !d_nu || !d_out) return set_err(VAG_E_INVALID, "null device pointer");
    if (nb <= 0 || nt <= 0 || nnu <= 0) return set_err(VAG_E_INVALID, "empty batch, time or frequency array");
    HIPCHK(hipSetDevice(c->device));
    c->mixed_flags_seen = false;
    int rc = grid_dev_body(c, d_params, nb, d_t, nt, d_nu, nnu, d_out);
    if (rc == VAG_E_UNSUPPORTED && c->mixed_flags_seen)
        rc = flux_dev_by_flags(c, d_params, nb, (size_t)nnu * nt, d_out, [&](const vag_model_params* gp, int ng, double* o) {
            return grid_dev_body(c, gp, ng, d_t, nt, d_nu, nnu, o);
        });
    return rc;
}

static int upload_series_bands(vag_ctx* c, const double* nu, int n);

// The whole (t, nu) series already in c->d_lg2t / c->d_lg2nu (prep_times) -> d_out[nb][n], the sum of every enabled
// component.  Longer series than one launch holds (exposure sampling, large data sets) go through in chunks of sorted
// points on the same model grid.
static int series_request(vag_ctx* c, const vag_model_params* d_params, int nb, int n, double* d_out, int n_bands) {
    const int chunk = SERIES_THREADS * SERIES_MAX_SLOTS;
    if (n <= chunk)
        return series_chunk(c, d_params, nb, c->d_lg2t.as<double>(), c->d_lg2nu.as<double>(), n, c->d_lg2nu.as<double>(), n,
                            d_out, n_bands);
    DevBuf& tmp = c->d_chunk;  // grow-only context scratch: the stream orders its reuse
    if (tmp.ensure(sizeof(double) * (size_t)nb * chunk)) return VAG_E_HIP;
    int rc = VAG_OK;
    for (int s0 = 0; s0 < n && rc == VAG_OK; s0 += chunk) {
        const int m = std::min(chunk, n - s0);
        rc = series_chunk(c, d_params, nb, c->d_lg2t.as<double>() + s0, c->d_lg2nu.as<double>() + s0, m,
                          c->d_lg2nu.as<double>(), n, tmp.as<double>());
        if (rc == VAG_OK && hipMemcpy2DAsync(d_out + s0, sizeof(double) * n, tmp.p, sizeof(double) * m, sizeof(double) * m,
                                             (size_t)nb, hipMemcpyDeviceToDevice, c->stream) != hipSuccess)
            rc = VAG_E_HIP;
    }
    return rc;
}

static int series_dev_body(vag_ctx* c, const vag_model_params* d_params, int nb, const double* d_t, const double* d_nu, int n,
                           double* d_out, int n_bands) {
    int rc = VAG_OK;
    for (int attempt = 0; attempt < 2; ++attempt) {
        rc = prep_times(c, d_t, n, d_nu, n);  // the grid sees the extrema of ALL requested times
        if (rc) return rc;
        c->allow_spec = attempt == 0 && !c->count_work;
        rc = run_model_stages(c, d_params, nb, false);
        c->allow_spec = false;
        if (rc == VAG_OK) rc = series_request(c, d_params, nb, n, d_out, n_bands);
        if (rc) {
            if (!c->spec_pending) return rc;
            c->spec_pending = false;
            c->hint_valid = false;
            if (vag_hook("VAG_DEBUG_SPEC")) std::fprintf(stderr, "[vag] planned-ahead series call failed (%d: %s): repeating\n", rc, g_err.c_str());
            continue;
        }
        rc = finish_speculation(c);
        if (rc != VAG_RETRY) break;
    }
    return rc;
}

int vag_flux_density_batch_dev(vag_ctx* c, const vag_model_params* d_params, int nb, const double* d_t,
                               const double* d_nu, int n, double* d_out) {
    ApiLock api_lock(c);
    HandoffScope handoff(c);
    if (!c) return set_err(VAG_E_INVALID, "null context");
    if (!d_params || !d_t || !d_nu || !d_out) return set_err(VAG_E_INVALID, "null device pointer");
    if (nb <= 0 || n <= 0) return set_err(VAG_E_INVALID, "empty batch or data array");
    HIPCHK(hipSetDevice(c->device));
    const int n_bands = c->pending_bands;  // only the host-pointer wrapper below knows the frequencies
    c->pending_bands = 0;
    c->mixed_flags_seen = false;
    int rc = series_dev_body(c, d_params, nb, d_t, d_nu, n, d_out, n_bands);
    if (rc == VAG_E_UNSUPPORTED && c->mixed_flags_seen)
        rc = flux_dev_by_flags(c, d_params, nb, (size_t)n, d_out, [&](const vag_model_params* gp, int ng, double* o) {
            return series_dev_body(c, gp, ng, d_t, d_nu, n, o, n_bands);
        });
    return rc;
}

int vag_flux_density_grid_batch(vag_ctx* c, const vag_model_params* params, int nb, const double* t, int nt,
                                const double* nu, int nnu, double* out) {
    ApiLock api_lock(c);
    if (!c) return set_err(VAG_E_INVALID, "null context");
    if (!nu || !out) return set_err(VAG_E_INVALID, "null frequency or output array");
    if (nnu <= 0) return set_err(VAG_E_INVALID, "frequency array must be non-empty");
    int rc = check_host_inputs(params, nb, t, nt);
    if (rc) return rc;
    if (!uniform_flags(params, nb))
        return run_flag_groups(params, nb, {{out, (size_t)nnu * nt}}, [&](const vag_model_params* gp, int ng, const std::vector<double*>& o) {
            return vag_flux_density_grid_batch(c, gp, ng, t, nt, nu, nnu, o[0]);
        });
    HIPCHK(hipSetDevice(c->device));
    if (c->d_params.ensure(sizeof(vag_model_params) * nb)) return VAG_E_HIP;
    if (c->d_t.ensure(sizeof(double) * nt)) return VAG_E_HIP;
    if (c->d_nu.ensure(sizeof(double) * nnu)) return VAG_E_HIP;
    if (c->d_out.ensure(sizeof(double) * (size_t)nb * nnu * nt)) return VAG_E_HIP;
    HIPCHK(hipMemcpyAsync(c->d_params.p, params, sizeof(vag_model_params) * nb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_t.p, t, sizeof(double) * nt, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_nu.p, nu, sizeof(double) * nnu, hipMemcpyHostToDevice, c->stream));
    rc = vag_flux_density_grid_batch_dev(c, c->d_params.as<vag_model_params>(), nb, c->d_t.as<double>(), nt,
                                         c->d_nu.as<double>(), nnu, c->d_out.as<double>());
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(out, c->d_out.p, sizeof(double) * (size_t)nb * nnu * nt, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    (void)collect_times(c);
    return check_status(c, nb);
}

// out4 = {fwd.sync, fwd.ssc, rvs.sync, rvs.ssc}; NULL entries are skipped
static int grid_components_impl(vag_ctx* c, const vag_model_params* params, int nb, const double* t, int nt, const double* nu,
                                int nnu, double* const* out4) {
    if (!c) return set_err(VAG_E_INVALID, "null context");
    if (nnu <= 0) return set_err(VAG_E_INVALID, "frequency array must be non-empty");
    int rc = check_host_inputs(params, nb, t, nt);
    if (rc) return rc;
    if (!uniform_flags(params, nb)) {
        const size_t st = (size_t)nnu * nt;
        return run_flag_groups(params, nb, {{out4[0], st}, {out4[1], st}, {out4[2], st}, {out4[3], st}},
                               [&](const vag_model_params* gp, int ng, const std::vector<double*>& o) {
                                   double* o4[4] = {o[0], o[1], o[2], o[3]};
                                   return grid_components_impl(c, gp, ng, t, nt, nu, nnu, o4);
                               });
    }
    HIPCHK(hipSetDevice(c->device));
    const size_t n_out = (size_t)nb * nnu * nt;
    if (c->d_params.ensure(sizeof(vag_model_params) * nb)) return VAG_E_HIP;
    if (c->d_t.ensure(sizeof(double) * nt)) return VAG_E_HIP;
    if (c->d_nu.ensure(sizeof(double) * nnu)) return VAG_E_HIP;
    if (c->d_comp.ensure(sizeof(double) * 4 * n_out)) return VAG_E_HIP;
    HIPCHK(hipMemcpyAsync(c->d_params.p, params, sizeof(vag_model_params) * nb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_t.p, t, sizeof(double) * nt, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_nu.p, nu, sizeof(double) * nnu, hipMemcpyHostToDevice, c->stream));
    rc = prep_times(c, c->d_t.as<double>(), nt, c->d_nu.as<double>(), nnu);
    if (rc) return rc;
    rc = run_model_stages(c, c->d_params.as<vag_model_params>(), nb, false);
    if (rc) return rc;
    double* d4[4];
    for (int q = 0; q < 4; ++q) d4[q] = out4[q] ? c->d_comp.as<double>() + q * n_out : nullptr;
    rc = grid_request_chunked(c, c->d_params.as<vag_model_params>(), nb, nt, nnu, nullptr, nullptr, d4);
    if (rc) return rc;
    for (int q = 0; q < 4; ++q)
        if (out4[q]) HIPCHK(hipMemcpyAsync(out4[q], d4[q], sizeof(double) * n_out, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    (void)collect_times(c);
    return check_status(c, nb);
}

int vag_flux_density_grid_components_batch(vag_ctx* c, const vag_model_params* params, int nb, const double* t, int nt,
                                           const double* nu, int nnu, double* out_sync, double* out_ssc) {
    ApiLock api_lock(c);
    double* out4[4] = {out_sync, out_ssc, nullptr, nullptr};
    return grid_components_impl(c, params, nb, t, nt, nu, nnu, out4);
}

int vag_flux_density_grid_components4_batch(vag_ctx* c, const vag_model_params* params, int nb, const double* t, int nt,
                                            const double* nu, int nnu, double* const* out4) {
    ApiLock api_lock(c);
    if (!out4) return set_err(VAG_E_INVALID, "out4 must not be null");
    if (!nu) return set_err(VAG_E_INVALID, "null frequency array");
    return grid_components_impl(c, params, nb, t, nt, nu, nnu, out4);
}

int vag_flux_density_batch(vag_ctx* c, const vag_model_params* params, int nb, const double* t, const double* nu, int n,
                           double* out) {
    ApiLock api_lock(c);
    if (!c) return set_err(VAG_E_INVALID, "null context");
    if (!nu || !out) return set_err(VAG_E_INVALID, "null frequency or output array");
    int rc = check_host_inputs(params, nb, t, n);
    if (rc) return rc;
    if (!uniform_flags(params, nb))
        return run_flag_groups(params, nb, {{out, (size_t)n}}, [&](const vag_model_params* gp, int ng, const std::vector<double*>& o) {
            return vag_flux_density_batch(c, gp, ng, t, nu, n, o[0]);
        });
    HIPCHK(hipSetDevice(c->device));
    if (c->d_params.ensure(sizeof(vag_model_params) * nb)) return VAG_E_HIP;
    if (c->d_t.ensure(sizeof(double) * n)) return VAG_E_HIP;
    if (c->d_nu.ensure(sizeof(double) * n)) return VAG_E_HIP;
    if (c->d_out.ensure(sizeof(double) * (size_t)nb * n)) return VAG_E_HIP;
    HIPCHK(hipMemcpyAsync(c->d_params.p, params, sizeof(vag_model_params) * nb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_t.p, t, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_nu.p, nu, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    c->pending_bands = upload_series_bands(c, nu, n);
    rc = vag_flux_density_batch_dev(c, c->d_params.as<vag_model_params>(), nb, c->d_t.as<double>(), c->d_nu.as<double>(), n,
                                    c->d_out.as<double>());
    c->pending_bands = 0;
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(out, c->d_out.p, sizeof(double) * (size_t)nb * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    (void)collect_times(c);
    return check_status(c, nb);
}

// Model.flux_density with its components apart (FluxDict.fwd / .rvs of a series, pymodel.cpp:373-389): out4[i] != NULL
// receives component i of {fwd.sync, fwd.ssc, rvs.sync, rvs.ssc} [nb][n]; disabled components come back as zeros.
int vag_flux_density_components4_batch(vag_ctx* c, const vag_model_params* params, int nb, const double* t, const double* nu,
                                       int n, double* const* out4) {
    ApiLock api_lock(c);
    if (!c) return set_err(VAG_E_INVALID, "null context");
    if (!out4) return set_err(VAG_E_INVALID, "out4 must not be null");
    if (!nu) return set_err(VAG_E_INVALID, "null frequency array");
    int rc = check_host_inputs(params, nb, t, n);
    if (rc) return rc;
    if (!uniform_flags(params, nb))
        return run_flag_groups(params, nb, {{out4[0], (size_t)n}, {out4[1], (size_t)n}, {out4[2], (size_t)n}, {out4[3], (size_t)n}},
                               [&](const vag_model_params* gp, int ng, const std::vector<double*>& o) {
                                   double* o4[4] = {o[0], o[1], o[2], o[3]};
                                   return vag_flux_density_components4_batch(c, gp, ng, t, nu, n, o4);
                               });
    HIPCHK(hipSetDevice(c->device));
    const size_t n_out = (size_t)nb * n;
    if (c->d_params.ensure(sizeof(vag_model_params) * nb)) return VAG_E_HIP;
    if (c->d_t.ensure(sizeof(double) * n)) return VAG_E_HIP;
    if (c->d_nu.ensure(sizeof(double) * n)) return VAG_E_HIP;
    if (c->d_comp.ensure(sizeof(double) * 4 * n_out)) return VAG_E_HIP;
    HIPCHK(hipMemcpyAsync(c->d_params.p, params, sizeof(vag_model_params) * nb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_t.p, t, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_nu.p, nu, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    const int n_bands = upload_series_bands(c, nu, n);
    rc = prep_times(c, c->d_t.as<double>(), n, c->d_nu.as<double>(), n);  // the grid sees the extrema of ALL requested times
    if (rc) return rc;
    rc = run_model_stages(c, c->d_params.as<vag_model_params>(), nb, false);
    if (rc) return rc;
    double* d4[4];
    for (int i = 0; i < 4; ++i) d4[i] = out4[i] ? c->d_comp.as<double>() + (size_t)i * n_out : nullptr;
    const int chunk = SERIES_THREADS * SERIES_MAX_SLOTS;
    if (n <= chunk) {
        rc = series_chunk(c, c->d_params.as<vag_model_params>(), nb, c->d_lg2t.as<double>(), c->d_lg2nu.as<double>(), n,
                          c->d_lg2nu.as<double>(), n, nullptr, n_bands, d4);
        if (rc) return rc;
    } else {  // long series (exposure sampling): chunks of sorted points on the same grid
        DevBuf& tmp = c->d_chunk;  // grow-only context scratch: the stream orders its reuse
        if (tmp.ensure(sizeof(double) * 4 * (size_t)nb * chunk)) return VAG_E_HIP;
        for (int s0 = 0; s0 < n && rc == VAG_OK; s0 += chunk) {
            const int mlen = std::min(chunk, n - s0);
            double* t4[4];
            for (int i = 0; i < 4; ++i) t4[i] = d4[i] ? tmp.as<double>() + (size_t)i * nb * chunk : nullptr;
            rc = series_chunk(c, c->d_params.as<vag_model_params>(), nb, c->d_lg2t.as<double>() + s0, c->d_lg2nu.as<double>() + s0,
                              mlen, c->d_lg2nu.as<double>(), n, nullptr, 0, t4);
            for (int i = 0; i < 4 && rc == VAG_OK; ++i)
                if (d4[i] && hipMemcpy2DAsync(d4[i] + s0, sizeof(double) * n, t4[i], sizeof(double) * mlen, sizeof(double) * mlen,
                                              (size_t)nb, hipMemcpyDeviceToDevice, c->stream) != hipSuccess)
                    rc = VAG_E_HIP;
        }
        if (rc) return rc;
    }
    for (int i = 0; i < 4; ++i)
        if (out4[i]) HIPCHK(hipMemcpyAsync(out4[i], d4[i], sizeof(double) * n_out, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    (void)collect_times(c);
    return check_status(c, nb);
}

// Band request with parameters and times already in HBM: nu = logspace(nu_min, nu_max, num_nu) nodes with Boole weights
// (Observer::flux, src/core/quadrature.h:153-196, pymodel.cpp:391-410), own grid from the request's own time range.
// d_total (optional) [nb][nt] receives the sum of the enabled components, d4 (optional) the components apart.
static int band_request_dev(vag_ctx* c, const vag_model_params* d_params, int nb, const double* d_t, int nt, double nu_min,
                            double nu_max, int num_nu, double* d_total, double* const* d4) {
    if (!(nu_min > 0)) return set_err(VAG_E_INVALID, "nu_min must be positive");
    if (!(nu_max > nu_min)) return set_err(VAG_E_INVALID, "nu_max must be greater than nu_min");
    if (num_nu < 2) return set_err(VAG_E_INVALID, "num_nu must be at least 2");
    if (num_nu > VAG_MAX_NU) return set_err(VAG_E_CAPACITY, "at most %d band frequencies", VAG_MAX_NU);
    // nu = xt::logspace(log10(nu_min Hz), log10(nu_max Hz), num_nu) in code units
    std::vector<double> nu_code(num_nu), nu_cgs(num_nu), w(num_nu, 0.0);
    {
        const double a = std::log10(nu_min * U_HZ), b = std::log10(nu_max * U_HZ);
        const double step = (b - a) / std::fmax(1.0, (double)(num_nu - 1));
        for (int i = 0; i < num_nu; ++i) {
            nu_code[i] = std::pow(10.0, (i == num_nu - 1) ? b : a + step * (double)i);
            nu_cgs[i] = nu_code[i] / U_HZ;
        }
        const int n = num_nu;
        const double h = std::log(nu_code[1] / nu_code[0]);
        const double cb = 2.0 * h / 45.0;
        int j = 0;
        for (; j + 4 < n; j += 4) {
            w[j] += cb * 7;
            w[j + 1] += cb * 32;
            w[j + 2] += cb * 12;
            w[j + 3] += cb * 32;
            w[j + 4] += cb * 7;
        }
        const int rem = n - 1 - j;
        if (rem == 3) {
            const double c38 = 3.0 * h / 8.0;
            w[j] += c38;
            w[j + 1] += c38 * 3;
            w[j + 2] += c38 * 3;
            w[j + 3] += c38;
        } else if (rem == 2) {
            const double c13 = h / 3.0;
            w[j] += c13;
            w[j + 1] += c13 * 4;
            w[j + 2] += c13;
        } else if (rem == 1) {
            w[j] += 0.5 * h;
            w[j + 1] += 0.5 * h;
        }
        for (int i = 0; i < n; ++i) w[i] *= nu_code[i];
    }
    if (c->d_nu.ensure(sizeof(double) * num_nu)) return VAG_E_HIP;
    if (c->d_bandw.ensure(sizeof(double) * num_nu)) return VAG_E_HIP;
    HIPCHK(hipMemcpyAsync(c->d_nu.p, nu_cgs.data(), sizeof(double) * num_nu, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_bandw.p, w.data(), sizeof(double) * num_nu, hipMemcpyHostToDevice, c->stream));
    int rc = prep_times(c, d_t, nt, c->d_nu.as<double>(), num_nu);
    if (rc) return rc;
    // the band nodes are exact code-unit values: overwrite log2(nu_cgs * Hz) with log2(nu_code) to avoid a round trip
    std::vector<double> lg2nu(num_nu);
    for (int i = 0; i < num_nu; ++i) lg2nu[i] = std::log2(nu_code[i]);
    HIPCHK(hipMemcpyAsync(c->d_lg2nu.p, lg2nu.data(), sizeof(double) * num_nu, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));  // the staging vectors above are stack-local
    rc = run_model_stages(c, d_params, nb, false);
    if (rc) return rc;
    return grid_request_chunked(c, d_params, nb, nt, num_nu, c->d_bandw.as<double>(), d_total, d4);
}

// Model.flux for a batch: out_total (optional) receives the sum of the enabled components, out4 (optional) the
// components {fwd.sync, fwd.ssc, rvs.sync, rvs.ssc} apart (NULL entries skipped).
static int flux_band_impl(vag_ctx* c, const vag_model_params* params, int nb, const double* t, int nt, double nu_min,
                          double nu_max, int num_nu, double* out_total, double* const* out4) {
    if (!c) return set_err(VAG_E_INVALID, "null context");
    if (!out_total && !out4) return set_err(VAG_E_INVALID, "null output array");
    int rc = check_host_inputs(params, nb, t, nt);
    if (rc) return rc;
    if (!uniform_flags(params, nb)) {
        const size_t st = (size_t)nt;
        return run_flag_groups(params, nb,
                               {{out_total, st}, {out4 ? out4[0] : nullptr, st}, {out4 ? out4[1] : nullptr, st},
                                {out4 ? out4[2] : nullptr, st}, {out4 ? out4[3] : nullptr, st}},
                               [&](const vag_model_params* gp, int ng, const std::vector<double*>& o) {
                                   double* o4[4] = {o[1], o[2], o[3], o[4]};
                                   return flux_band_impl(c, gp, ng, t, nt, nu_min, nu_max, num_nu, o[0], out4 ? o4 : nullptr);
                               });
    }
    HIPCHK(hipSetDevice(c->device));
    if (c->d_params.ensure(sizeof(vag_model_params) * nb)) return VAG_E_HIP;
    if (c->d_t.ensure(sizeof(double) * nt)) return VAG_E_HIP;
    if (c->d_out.ensure(sizeof(double) * (size_t)nb * nt)) return VAG_E_HIP;
    if (c->d_comp.ensure(sizeof(double) * (size_t)nb * nt * 4)) return VAG_E_HIP;
    HIPCHK(hipMemcpyAsync(c->d_params.p, params, sizeof(vag_model_params) * nb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_t.p, t, sizeof(double) * nt, hipMemcpyHostToDevice, c->stream));
    const size_t n_out = (size_t)nb * nt;
    double* d4[4] = {nullptr, nullptr, nullptr, nullptr};
    if (out4)
        for (int q = 0; q < 4; ++q) d4[q] = out4[q] ? c->d_comp.as<double>() + q * n_out : nullptr;
    rc = band_request_dev(c, c->d_params.as<vag_model_params>(), nb, c->d_t.as<double>(), nt, nu_min, nu_max, num_nu,
                          out_total ? c->d_out.as<double>() : nullptr, out4 ? d4 : nullptr);
    if (rc) return rc;
    if (out_total) HIPCHK(hipMemcpyAsync(out_total, c->d_out.p, sizeof(double) * n_out, hipMemcpyDeviceToHost, c->stream));
    for (int q = 0; q < 4; ++q)
        if (d4[q]) HIPCHK(hipMemcpyAsync(out4[q], d4[q], sizeof(double) * n_out, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    (void)collect_times(c);
    return check_status(c, nb);
}

int vag_flux_batch(vag_ctx* c, const vag_model_params* params, int nb, const double* t, int nt, double nu_min,
                   double nu_max, int num_nu, double* out) {
    ApiLock api_lock(c);
    return flux_band_impl(c, params, nb, t, nt, nu_min, nu_max, num_nu, out, nullptr);
}

int vag_flux_components_batch(vag_ctx* c, const vag_model_params* params, int nb, const double* t, int nt, double nu_min,
                              double nu_max, int num_nu, double* out_sync, double* out_ssc) {
    ApiLock api_lock(c);
    double* out4[4] = {out_sync, out_ssc, nullptr, nullptr};
    return flux_band_impl(c, params, nb, t, nt, nu_min, nu_max, num_nu, nullptr, out4);
}

int vag_flux_components4_batch(vag_ctx* c, const vag_model_params* params, int nb, const double* t, int nt, double nu_min,
                               double nu_max, int num_nu, double* const* out4) {
    ApiLock api_lock(c);
    if (!out4) return set_err(VAG_E_INVALID, "out4 must not be null");
    return flux_band_impl(c, params, nb, t, nt, nu_min, nu_max, num_nu, nullptr, out4);
}

// Distinct frequencies of a short series (n <= 64, at most 8 of them): uploads [band of point s | first point of band b]
// and returns the number of bands, 0 when the shared-node path does not apply.
static int upload_series_bands(vag_ctx* c, const double* nu, int n) {
    constexpr int MAXB = SERIES_MAX_BANDS;
    if (n <= 0 || n > FITROWS_MAX_POINTS) return 0;
    int buf[FITROWS_MAX_POINTS + MAXB] = {};
    double vals[MAXB];
    int nbands = 0;
    for (int s = 0; s < n; ++s) {
        int b = 0;
        while (b < nbands && vals[b] != nu[s]) ++b;
        if (b == nbands) {
            if (nbands == MAXB) return 0;
            vals[nbands] = nu[s];
            buf[FITROWS_MAX_POINTS + nbands] = s;
            ++nbands;
        }
        buf[s] = b;
    }
    if (2 * nbands > n) return 0;  // nothing to share
    if (c->h_bands_n == nbands && std::memcmp(c->h_bandbuf, buf, sizeof buf) == 0) return nbands;  // already resident
    if (c->d_bandidx.ensure(sizeof buf)) return 0;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return 0;  // an earlier copy may still read h_bandbuf
    std::memcpy(c->h_bandbuf, buf, sizeof buf);
    c->h_bands_n = nbands;
    if (hipMemcpyAsync(c->d_bandidx.p, c->h_bandbuf, sizeof buf, hipMemcpyHostToDevice, c->stream) != hipSuccess) {
        c->h_bands_n = -1;
        return 0;
    }
    return nbands;
}

// ---- fit data cache: the observation arrays, the slot map and the priors of a vag_fit_spec live in ONE device buffer, uploaded
//      through one pinned staging copy only when their content hash changes (a sampler loop calls with the same spec) ----
// layout in doubles: [t | nu | ln_flux | ln_err | weight | ext] (n each), then per band group [t | ln_flux | ln_err | weight]
// (bd.n each), then [lower | upper | prior_a | prior_b] (16 each), then int32 [slot | is_log | prior_kind] (16 each).
static uint64_t fnv1a(uint64_t h, const void* p, size_t n) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
    return h;
}

static int upload_fit_spec(vag_ctx* c, const vag_fit_spec* spec, int ndim) {
    if (ndim != spec->ndim || ndim <= 0 || ndim > 16) return set_err(VAG_E_INVALID, "ndim must match spec and be in 1..16");
    const int n = spec->n_data;
    if (n < 0 || spec->n_bands < 0 || (n == 0 && spec->n_bands == 0)) return set_err(VAG_E_INVALID, "fit spec has no data");
    for (int d = 0; d < ndim; ++d) {
        const int s = spec->slot[d];
        if (s == VAG_P_A_V) continue;
        if (s < 0 || (s >= VAG_P_COUNT && (s < VAG_P_RVS_EPS_E || s > VAG_P_MAG_Q))) return set_err(VAG_E_INVALID, "bad parameter slot");
    }
    if (spec->use_priors)
        for (int d = 0; d < ndim; ++d) {
            if (!(spec->lower[d] < spec->upper[d])) return set_err(VAG_E_INVALID, "prior bounds of parameter %d: need lower < upper", d);
            const int k = spec->prior_kind[d];
            if (k < VAG_PRIOR_UNIFORM || k > VAG_PRIOR_UNIFORM_RANGE) return set_err(VAG_E_INVALID, "unknown prior kind %d", k);
            if (k == VAG_PRIOR_UNIFORM_RANGE && !(spec->prior_b[d] > spec->prior_a[d])) return set_err(VAG_E_INVALID, "Uniform prior needs maximum > minimum");
            if (k == VAG_PRIOR_GAUSSIAN && !(spec->prior_b[d] > 0)) return set_err(VAG_E_INVALID, "Gaussian prior needs sigma > 0");
            if (k == VAG_PRIOR_LOG_UNIFORM && !(spec->prior_a[d] > 0 && spec->prior_b[d] > spec->prior_a[d]))
                return set_err(VAG_E_INVALID, "LogUniform prior needs 0 < minimum < maximum");
        }
    uint64_t h = 1469598103934665603ull;
    const int head[4] = {n, spec->n_bands, ndim, spec->ext_kernel ? 1 : 0};
    h = fnv1a(h, head, sizeof head);
    size_t total = 6 * (size_t)std::max(n, 1);
    for (const double* arr : {spec->t, spec->nu, spec->ln_flux, spec->ln_err, spec->weight, spec->ext_kernel})
        if (arr && n > 0) h = fnv1a(h, arr, sizeof(double) * n);
    for (int g = 0; g < spec->n_bands; ++g) {
        const vag_band_obs& bd = spec->bands[g];
        if (bd.n <= 0) return set_err(VAG_E_INVALID, "band group %d has no observations", g);
        h = fnv1a(h, &bd.n, sizeof bd.n);
        for (const double* arr : {bd.t, bd.ln_flux, bd.ln_err, bd.weight}) h = fnv1a(h, arr, sizeof(double) * bd.n);
        total += 4 * (size_t)bd.n;
    }
    h = fnv1a(h, spec->slot, sizeof spec->slot);
    h = fnv1a(h, spec->is_log, sizeof spec->is_log);
    h = fnv1a(h, &spec->use_priors, sizeof spec->use_priors);
    if (spec->use_priors) {
        h = fnv1a(h, spec->lower, sizeof spec->lower);
        h = fnv1a(h, spec->upper, sizeof spec->upper);
        h = fnv1a(h, spec->prior_kind, sizeof spec->prior_kind);
        h = fnv1a(h, spec->prior_a, sizeof spec->prior_a);
        h = fnv1a(h, spec->prior_b, sizeof spec->prior_b);
    }
    const size_t prior_off = total;
    total += 4 * 16 + 3 * 16 / 2;  // four double[16] + three int32[16]
    c->fit_prior_off = prior_off;
    if (c->fit_hash_valid && c->fit_hash == h && c->fit_doubles == total) return VAG_OK;  // resident already

    // content changed: validate it (once), stage and upload
    for (int i = 0; i < n; ++i)
        if (!(spec->t[i] > 0)) return set_err(VAG_E_INVALID, "data times must be positive");
    for (int i = 1; i < n; ++i)
        if (spec->t[i] < spec->t[i - 1]) return set_err(VAG_E_INVALID, "data times must be ascending (fitter.py:420-428)");
    for (int g = 0; g < spec->n_bands; ++g) {
        const vag_band_obs& bd = spec->bands[g];
        for (int i = 0; i < bd.n; ++i)
            if (!(bd.t[i] > 0) || (i > 0 && bd.t[i] < bd.t[i - 1]))
                return set_err(VAG_E_INVALID, "band group %d: times must be positive and ascending", g);
    }
    c->fit_hash_valid = false;
    HIPCHK(hipStreamSynchronize(c->stream));  // an earlier staging copy may still be in flight
    if (c->h_fit.ensure(sizeof(double) * total)) return VAG_E_HIP;
    if (c->d_fit.ensure(sizeof(double) * total)) return VAG_E_HIP;
    double* hp = c->h_fit.as<double>();
    std::memset(hp, 0, sizeof(double) * total);
    if (n > 0) {
        const double* src[6] = {spec->t, spec->nu, spec->ln_flux, spec->ln_err, spec->weight, spec->ext_kernel};
        for (int q = 0; q < 6; ++q)
            if (src[q]) std::memcpy(hp + (size_t)q * n, src[q], sizeof(double) * n);
    }
    size_t off = 6 * (size_t)std::max(n, 1);
    for (int g = 0; g < spec->n_bands; ++g) {
        const vag_band_obs& bd = spec->bands[g];
        const double* src[4] = {bd.t, bd.ln_flux, bd.ln_err, bd.weight};
        for (int q = 0; q < 4; ++q) std::memcpy(hp + off + (size_t)q * bd.n, src[q], sizeof(double) * bd.n);
        off += 4 * (size_t)bd.n;
    }
    std::memcpy(hp + prior_off, spec->lower, sizeof spec->lower);
    std::memcpy(hp + prior_off + 16, spec->upper, sizeof spec->upper);
    std::memcpy(hp + prior_off + 32, spec->prior_a, sizeof spec->prior_a);
    std::memcpy(hp + prior_off + 48, spec->prior_b, sizeof spec->prior_b);
    int* hi = reinterpret_cast<int*>(hp + prior_off + 64);
    std::memcpy(hi, spec->slot, sizeof spec->slot);
    std::memcpy(hi + 16, spec->is_log, sizeof spec->is_log);
    std::memcpy(hi + 32, spec->prior_kind, sizeof spec->prior_kind);
    HIPCHK(hipMemcpyAsync(c->d_fit.p, hp, sizeof(double) * total, hipMemcpyHostToDevice, c->stream));
    c->fit_hash = h;
    c->fit_doubles = total;
    c->fit_hash_valid = true;
    return VAG_OK;
}

// The front of a likelihood call, one launch: bounds mask and ln prior (log_prob_batch, fitting/samplers.py:72-91), the
// transformer of fitting/utils.py:110-135 (theta[nb][ndim] -> params[nb], 10^theta for log-scale parameters), A_V per walker,
// and -- block 0 -- log2 of the point data's times / frequencies and their time extrema for the grid stage.
__global__ void __launch_bounds__(128)
vag_fit_front_kernel(vag_model_params base, const double* __restrict__ theta, int nb, int ndim, const double* __restrict__ prior,
                     int use_priors, double a_v_fixed, vag_model_params* __restrict__ out, double* __restrict__ a_v,
                     double* __restrict__ ln_prior, int* __restrict__ fitstat, const double* __restrict__ t, int n,
                     const double* __restrict__ nu, double* __restrict__ lg2_t, double* __restrict__ lg2_nu,
                     double* __restrict__ tminmax, const int* __restrict__ order /* evaluation slot -> walker, or null */) {
    const int* slot = reinterpret_cast<const int*>(prior + 64);
    const int* is_log = slot + 16;
    const int* kind = slot + 32;
    if (blockIdx.x == 0) {
        if (threadIdx.x < 4) fitstat[threadIdx.x] = 0;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            lg2_t[i] = log2(t[i] * U_SEC);  // xt::log2(t_obs), observer.h:359
            lg2_nu[i] = log2(nu[i] * U_HZ);
        }
        if (threadIdx.x == 0 && n > 0) {  // ascending data (fitter.py:420-428)
            tminmax[0] = t[0];
            tminmax[1] = t[n - 1];
        }
    }
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    out[b] = base;  // the sampled fields are patched in place (a private copy indexed by slot would live in scratch)
    double* f = &out[b].theta_c;
    double av = a_v_fixed, lp = 0;
    bool inside = true;
    const int walker = order ? order[b] : b;
    for (int d = 0; d < ndim; ++d) {
        const double v = theta[(size_t)walker * ndim + d];
        if (use_priors) {
            const double lo = prior[d], hi = prior[16 + d];
            inside = inside && (v >= lo) && (v <= hi);
            if (kind[d] == VAG_PRIOR_GAUSSIAN) {
                const double z = (v - prior[32 + d]) / prior[48 + d];
                lp += -0.5 * z * z - log(prior[48 + d] * 2.5066282746310002);
            } else if (kind[d] == VAG_PRIOR_LOG_UNIFORM) {
                const double mn = prior[32 + d], mx = prior[48 + d];
                lp += (v >= mn && v <= mx) ? -log(v * log(mx / mn)) : -INFINITY;
            } else if (kind[d] == VAG_PRIOR_UNIFORM) {
                lp += -log(hi - lo);
            } else if (kind[d] == VAG_PRIOR_UNIFORM_RANGE) {
                const double mn = prior[32 + d], mx = prior[48 + d];
                lp += (v >= mn && v <= mx) ? -log(mx - mn) : -INFINITY;
            }
        }
        const double val = is_log[d] ? pow(10.0, v) : v;
        if (slot[d] == VAG_P_A_V)
            av = val;  // not a Model field: scales the point-data fluxes (fitter.py:512-519)
        else
            f[slot[d]] = val;
    }
    if (!inside) {  // never evaluated by the reference either: an invalid parameter set stops at the grid stage with no work
        out[b].theta_c = NAN;
        lp = -INFINITY;
    }
    a_v[b] = av;
    ln_prior[b] = use_priors ? lp : 0.0;
}

// The back of one pass of a likelihood call, one wavefront per walker:
//   chi2[m] (+)= sum_i w_i ((ln F_obs,i - ln max(F_model,i e^{-A_V k_i}, 1e-300)) / sigma_i)^2   (Fitter._chi2_sum, fitter.py:497-501,
//   with the extinction factor of fitter.py:512-519);
//   valid[m] &= this pass evaluated the walker -- parameters valid, grid within the engine limits, no ODE row without an acceptable
//   step, SSC tables within their capacity -- else the walker scores -inf like eval_one's except branch (samplers.py:61-70);
//   last pass: out[m] = valid ? -chi2 / 2 + ln prior : -inf.
__global__ void __launch_bounds__(64)
vag_fit_back_kernel(const double* __restrict__ flux /* [nb][n] */, int n, const double* __restrict__ ln_flux,
                    const double* __restrict__ ln_err, const double* __restrict__ weight, const double* __restrict__ ext /* or null */,
                    const double* __restrict__ a_v, const VagGridMeta* __restrict__ meta, const int* __restrict__ row_status,
                    const int* __restrict__ row_off, const int* __restrict__ ic_status /* or null */, double* __restrict__ chi2,
                    int* __restrict__ valid, const double* __restrict__ ln_prior, int first, int last, double* __restrict__ out,
                    int* __restrict__ fitstat /* [0] walkers scored -inf, [1] of those: SSC table failures */,
                    const int* __restrict__ order /* evaluation slot -> walker, or null */,
                    const float* __restrict__ cost /* with next_order: the slots' costs of THIS call (the grid kernel's plan scan leaves them) */,
                    int nb, int* __restrict__ next_order /* or null: [rank] = walker, descending cost */) {
    const int m = blockIdx.x, lane = threadIdx.x;
    // next_order[rank] = walker, ranks by descending cost of the slot in THIS call (cost[] is in evaluation-slot order: `order` maps a slot
    // back to its walker; null = identity).  Ranking by counting: the lanes compare this wavefront's slot with all others.  (Until round 5
    // a launch of its own, vag_order_kernel, behind this kernel.)
    if (next_order) {
        const float mine = cost[m];
        int rank = 0;
        for (int i0 = 0; i0 < nb; i0 += 64) {
            const int i = i0 + lane;
            const float c = i < nb ? cost[i] : -1.0f;
            rank += __popcll(__ballot(c > mine || (c == mine && i < m)));
        }
        if (lane == 0) next_order[rank] = order ? order[m] : m;
    }
    const double av = (ext != nullptr) ? a_v[m] : 0.0;
    const bool grid_ok = meta[m].status == 0;
    double s = 0;
    if (grid_ok)
        for (int i = lane; i < n; i += 64) {
            double f = flux[(size_t)m * n + i];
            if (av != 0.0) f = f * exp(-av * ext[i]);
            const double fm = (f != f) ? f : (f > 1e-300 ? f : 1e-300);
            const double q = (ln_flux[i] - log(fm)) / ln_err[i];
            s += weight[i] * (q * q);
        }
    s = vag::wave_sum(s);
    bool bad_row = false;
    if (grid_ok)
        for (int r = row_off[m] + lane; r < row_off[m + 1]; r += 64) bad_row = bad_row || row_status[r] == 1;
    const bool any_bad = __any(bad_row);
    if (lane == 0) {
        const bool ic_bad = grid_ok && ic_status && ic_status[m] != 0;
        const int ok = (first ? 1 : valid[m]) && grid_ok && !any_bad && !ic_bad;
        const double acc = first ? s : chi2[m] + s;
        valid[m] = ok;
        chi2[m] = acc;
        if (ic_bad) atomicAdd(fitstat + 1, 1);
        if (last) {
            const double lp = ln_prior[m];
            const bool fin = ok && isfinite(acc) && lp > -INFINITY;
            out[order ? order[m] : m] = fin ? -0.5 * acc + lp : -INFINITY;
            if (!fin) atomicAdd(fitstat, 1);
        }
    }
}

static int loglike_body(vag_ctx* c, const vag_fit_spec* spec, const double* d_theta, int nb, int ndim, double* d_out, bool try_spec) {
    int rc = VAG_OK;
    const int n = spec->n_data;
    hipStream_t st = c->stream;
    if (c->d_params.ensure(sizeof(vag_model_params) * nb)) return VAG_E_HIP;
    if (c->d_valid.ensure(sizeof(int) * nb)) return VAG_E_HIP;
    if (c->d_chi2.ensure(sizeof(double) * 3 * (size_t)nb)) return VAG_E_HIP;  // [chi2 | A_V | ln prior]
    if (c->d_fitstat.ensure(sizeof(int) * 4)) return VAG_E_HIP;
    if (c->d_lg2t.ensure(sizeof(double) * std::max(n, 1))) return VAG_E_HIP;
    if (c->d_lg2nu.ensure(sizeof(double) * std::max(n, 1))) return VAG_E_HIP;
    if (c->d_tminmax.ensure(sizeof(double) * 2)) return VAG_E_HIP;
    double* d_chi2 = c->d_chi2.as<double>();
    double* d_av = d_chi2 + nb;
    double* d_lp = d_chi2 + 2 * (size_t)nb;
    double* d = c->d_fit.as<double>();
    const double* d_prior = d + c->fit_prior_off;
    vag_model_params* d_params = c->d_params.as<vag_model_params>();
    // evaluation order: by the costs of the previous call with this batch size (see vag_ctx::d_order)
    const bool can_order = nb >= 64 && nb <= 8192 && !vag_hook("VAG_NO_ORDER");
    const int* d_order = (can_order && c->order_nb == nb) ? c->d_order[c->order_cur].as<int>() : nullptr;
    hipLaunchKernelGGL(vag_fit_front_kernel, dim3((nb + 127) / 128), dim3(128), 0, st, spec->base, d_theta, nb, ndim, d_prior,
                       spec->use_priors, spec->a_v_fixed, d_params, d_av, d_lp, c->d_fitstat.as<int>(), d, n, d + n,
                       c->d_lg2t.as<double>(), c->d_lg2nu.as<double>(), c->d_tminmax.as<double>(), d_order);
    HIPCHK(hipGetLastError());
    bool order_made = false;
    // once per call, from the first pass's grids: the order the NEXT call evaluates in -- made by the first back kernel (its wavefronts are
    // one per evaluation slot: no launch of its own on the call's critical path since round 6)
    auto next_order = [&]() -> int* {
        if (!can_order || order_made) return nullptr;
        DevBuf& nxt = c->d_order[c->order_cur ^ 1];
        if (nxt.ensure(sizeof(int) * (size_t)nb)) return nullptr;
        order_made = true;
        return nxt.as<int>();
    };
    const int n_pass = (n > 0 ? 1 : 0) + spec->n_bands;
    int pass = 0, n_cap = 0, n_inv = 0;  // per-pass rejection counts: the call reports the worst pass
    // the SSC tables of a pass report per-model failures in d_icstatus: in a fit they invalidate the walker, they do not raise
    auto back = [&](const double* flux, int npts, const double* lnf, const double* lne, const double* w, const double* ext) -> int {
        const bool ssc = (c->batch_flags & (VAG_FLAG_SSC | VAG_FLAG_RVS_SSC)) != 0;
        hipLaunchKernelGGL(vag_fit_back_kernel, dim3(nb), dim3(64), 0, st, flux, npts, lnf, lne, w, ext, d_av,
                           c->d_meta.as<VagGridMeta>(), c->d_row_status.as<int>(), c->d_row_off.as<int>(),
                           (ssc && c->d_icstatus.p) ? c->d_icstatus.as<int>() : nullptr, d_chi2, c->d_valid.as<int>(), d_lp,
                           pass == 0 ? 1 : 0, pass == n_pass - 1 ? 1 : 0, d_out, c->d_fitstat.as<int>(), d_order, c->d_cost_f.as<float>(), nb,
                           next_order());
        HIPCHK(hipGetLastError());
        ++pass;
        n_cap = std::max(n_cap, c->plan.n_models_capacity);
        n_inv = std::max(n_inv, c->plan.n_models_invalid);
        return VAG_OK;
    };
    c->ic_soft_fail = true;
    if (n > 0) {  // point data: one (t, nu) series per walker (fitter.py:510-522)
        if (c->d_series_flux.ensure(sizeof(double) * (size_t)nb * n)) return VAG_E_HIP;
        c->allow_spec = try_spec;
        c->order_next = d_order != nullptr;
        c->last_order = d_order;
        rc = run_model_stages(c, d_params, nb, false);
        c->allow_spec = false;
        if (rc == VAG_OK) rc = series_request(c, d_params, nb, n, c->d_series_flux.as<double>(), upload_series_bands(c, spec->nu, n));
        if (rc == VAG_OK)
            rc = back(c->d_series_flux.as<double>(), n, d + 2 * (size_t)n, d + 3 * (size_t)n, d + 4 * (size_t)n,
                      spec->ext_kernel ? d + 5 * (size_t)n : nullptr);
        if (rc == VAG_OK) {
            rc = finish_speculation(c);  // before a band group's own grid pass reuses the plan buffers
            n_cap = std::max(n_cap, c->plan.n_models_capacity);
            n_inv = std::max(n_inv, c->plan.n_models_invalid);
        } else if (c->spec_pending) {  // an error while planning from stale sizes: repeat the call on the waiting path
            if (vag_hook("VAG_DEBUG_SPEC")) std::fprintf(stderr, "[vag] planned-ahead likelihood call failed (%d: %s): repeating\n", rc, g_err.c_str());
            c->spec_pending = false;
            c->hint_valid = false;
            rc = VAG_RETRY;
        }
    }
    size_t off = 6 * (size_t)std::max(n, 1);
    for (int g = 0; g < spec->n_bands && rc == VAG_OK; ++g) {  // band-integrated groups: one Model.flux request each (fitter.py:524-531)
        const vag_band_obs& bd = spec->bands[g];
        if (c->d_series_flux.ensure(sizeof(double) * (size_t)nb * std::max(bd.n, n))) return VAG_E_HIP;
        double* db = d + off;
        off += 4 * (size_t)bd.n;
        c->order_next = d_order != nullptr;
        c->last_order = d_order;
        rc = band_request_dev(c, d_params, nb, db, bd.n, bd.nu_min, bd.nu_max, bd.num_points, c->d_series_flux.as<double>(), nullptr);
        if (rc == VAG_OK) rc = back(c->d_series_flux.as<double>(), bd.n, db + bd.n, db + 2 * (size_t)bd.n, db + 3 * (size_t)bd.n, nullptr);
    }
    c->ic_soft_fail = false;
    c->plan.n_models_capacity = n_cap;
    c->plan.n_models_invalid = n_inv;
    c->fit_stats_pending = true;
    if (rc == VAG_OK && order_made) {  // the order computed from this call's grids serves the next call of this size
        c->order_cur ^= 1;
        c->order_nb = nb;
    } else if (rc != VAG_OK) {
        c->order_nb = 0;
    }
    return rc;
}

int vag_loglike_batch_dev(vag_ctx* c, const vag_fit_spec* spec, const double* d_theta, int nb, int ndim, double* d_out) {
    ApiLock api_lock(c);
    HandoffScope handoff(c);
    if (!c) return set_err(VAG_E_INVALID, "null context");
    if (!spec || !d_theta || !d_out) return set_err(VAG_E_INVALID, "null spec or device pointer");
    if (nb <= 0) return set_err(VAG_E_INVALID, "batch must be non-empty");
    HIPCHK(hipSetDevice(c->device));
    int rc = upload_fit_spec(c, spec, ndim);
    if (rc) return rc;
    rc = loglike_body(c, spec, d_theta, nb, ndim, d_out, !c->count_work);
    if (rc == VAG_RETRY) rc = loglike_body(c, spec, d_theta, nb, ndim, d_out, false);
    return rc;
}

__global__ void vag_model_cost_kernel(const VagGridMeta* __restrict__ meta, int nb, double* __restrict__ cost,
                                      const int* __restrict__ order /* evaluation slot -> walker of the last batch, or null */) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= nb) return;
    const VagGridMeta M = meta[m];
    cost[order ? order[m] : m] = M.status == 0 ? (double)M.n_theta * (double)M.n_phi_eff * (double)M.n_t : 0.0;
}

int vag_last_model_costs_dev(vag_ctx* c, int nb, double* d_cost) {
    ApiLock api_lock(c);
    HandoffScope handoff(c);
    if (!c || !d_cost) return set_err(VAG_E_INVALID, "null context or buffer");
    if (nb <= 0 || nb != c->nb || !c->d_meta.p) return set_err(VAG_E_INVALID, "no batch of %d models has been evaluated on this context", nb);
    HIPCHK(hipSetDevice(c->device));
    hipLaunchKernelGGL(vag_model_cost_kernel, dim3((nb + 127) / 128), dim3(128), 0, c->stream, c->d_meta.as<VagGridMeta>(), nb, d_cost,
                       c->order_active ? c->last_order : nullptr);
    HIPCHK(hipGetLastError());
    return VAG_OK;
}

// ---- sharded log_prob_batch: the deal, this rank's block, the scatter after the caller's all-gather (ABI v9) ----

// One wavefront per position-or-walker m.  m < nb_all: walker m's rank q by counting (stable, descending cost; without costs
// every walker counts 1 and q = m), its (rank r, slot s) of the boustrophedon deal, table[r * per + s] = m, and -- when r is this
// process's rank -- its theta row into the rank's compact block.  m >= nb_all: padding positions of the last sweep, table = -1.
__global__ void __launch_bounds__(256)
vag_shard_deal_kernel(const double* __restrict__ cost /* [nb_all] or null */, const double* __restrict__ theta_all, int nb_all,
                      int ndim, int rank, int world, int per, int* __restrict__ table, double* __restrict__ theta_mine) {
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= world * per) return;
    int q = m;
    if (m < nb_all && cost) {
        const double mine = cost[m];
        q = 0;
        for (int i0 = 0; i0 < nb_all; i0 += 64) {
            const int i = i0 + lane;
            const double c = i < nb_all ? cost[i] : -1.0;
            q += __popcll(__ballot(c > mine || (c == mine && i < m)));
        }
    }
    const int s = q / world, k = q - s * world;
    const int r = (s & 1) ? world - 1 - k : k;
    if (lane == 0) table[r * per + s] = m < nb_all ? m : -1;
    if (m < nb_all && r == rank && lane < ndim) theta_mine[(size_t)s * ndim + lane] = theta_all[(size_t)m * ndim + lane];
}

// block[s] = {ln L, cost} of this rank's slot s (cost = the (theta, phi, t) cell count of vag_last_model_costs_dev), {NaN, 0} for padding
__global__ void vag_shard_pack_kernel(const double* __restrict__ ll, const VagGridMeta* __restrict__ meta,
                                      const int* __restrict__ order /* evaluation slot -> block slot, or null */, int n_mine, int per,
                                      double* __restrict__ block) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= per) return;
    if (s < n_mine) {
        const VagGridMeta M = meta[s];
        block[2 * (size_t)s] = ll[s];
        block[2 * (size_t)(order ? order[s] : s) + 1] = M.status == 0 ? (double)M.n_theta * (double)M.n_phi_eff * (double)M.n_t : 0.0;
    } else {
        block[2 * (size_t)s] = NAN;
        block[2 * (size_t)s + 1] = 0.0;
    }
}

// One workgroup: ln L back into walker order, the gathered costs kept for the next deal (a walker that was not evaluated is
// assumed average: the mean of the positive costs, or 1).
__global__ void __launch_bounds__(1024)
vag_shard_finish_kernel(const double* __restrict__ gathered, const int* __restrict__ table, int n_slots, double* __restrict__ out,
                        double* __restrict__ cost) {
    __shared__ double s_sum[16];
    __shared__ double s_cnt[16];
    double sum = 0, cnt = 0;
    for (int i = threadIdx.x; i < n_slots; i += blockDim.x) {
        const int w = table[i];
        if (w < 0) continue;
        out[w] = gathered[2 * (size_t)i];
        const double cw = gathered[2 * (size_t)i + 1];
        if (cw > 0) sum += cw, cnt += 1;
    }
    sum = vag::wave_sum(sum);
    cnt = vag::wave_sum(cnt);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = sum, s_cnt[threadIdx.x >> 6] = cnt;
    __syncthreads();
    sum = 0, cnt = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) sum += s_sum[i], cnt += s_cnt[i];
    const double mean = cnt > 0 ? sum / cnt : 1.0;
    for (int i = threadIdx.x; i < n_slots; i += blockDim.x) {
        const int w = table[i];
        if (w < 0) continue;
        const double cw = gathered[2 * (size_t)i + 1];
        cost[w] = cw > 0 ? cw : mean;
    }
}

static int shard_begin(vag_ctx* c, const vag_fit_spec* spec, const double* d_theta_all, int nb_all, int ndim, int rank, int world,
                       double* d_block, uint64_t* ticket) {
    if (!c || !spec || !d_theta_all || !d_block) return set_err(VAG_E_INVALID, "null context, spec or buffer");
    if (nb_all <= 0) return set_err(VAG_E_INVALID, "batch must be non-empty");
    if (world <= 0 || rank < 0 || rank >= world) return set_err(VAG_E_INVALID, "rank %d outside a world of %d", rank, world);
    if (ndim <= 0 || ndim > 16) return set_err(VAG_E_INVALID, "ndim must be 1..16");
    HIPCHK(hipSetDevice(c->device));
    const int per = (nb_all + world - 1) / world;
    // slots of this rank that hold a walker: positions q = s * world + k(s) < nb_all
    int n_mine = 0;
    for (int s = 0; s < per; ++s) {
        const int k = (s & 1) ? world - 1 - rank : rank;
        if ((long long)s * world + k < nb_all) ++n_mine;
    }
    if (c->d_shard_theta.ensure(sizeof(double) * (size_t)per * ndim)) return VAG_E_HIP;
    if (c->d_shard_ll.ensure(sizeof(double) * (size_t)per)) return VAG_E_HIP;
    // every rank needs the same deal, whether or not it holds a walker: the key of the costs is the spec's content hash
    {
        const int rc = upload_fit_spec(c, spec, ndim);
        if (rc) return rc;
    }
    int entry = -1, victim = -1;
    for (int i = 0; i < 4; ++i) {
        const vag_ctx::ShardCosts& e = c->shard_costs[i];
        if (e.nb == nb_all && e.world == world && e.hash == c->fit_hash) entry = i;
        if (e.in_flight == 0 && (victim < 0 || e.used < c->shard_costs[victim].used)) victim = i;
    }
    int slot = -1;  // a free flight slot, the least recently dealt one
    for (int i = 0; i < 8; ++i)
        if (!c->shard_flights[i].in_use && (slot < 0 || c->shard_flights[i].ticket < c->shard_flights[slot].ticket)) slot = i;
    // the unticketed form deals a key anew when it is dealt again before its finish (the earlier block never goes through a finish)
    bool replaces = false;
    for (int i = 0; i < 8 && !ticket && entry >= 0; ++i)
        if (c->shard_flights[i].in_use && c->shard_flights[i].unticketed && c->shard_flights[i].key == entry) slot = i, replaces = true;
    if (slot < 0) return set_err(VAG_E_INVALID, "eight sharded calls wait for their finish on this context");
    if (entry < 0) {
        if (victim < 0) return set_err(VAG_E_INVALID, "sharded calls of four other fits wait for their finish on this context");
        entry = victim;
        c->shard_costs[entry].nb = nb_all;
        c->shard_costs[entry].world = world;
        c->shard_costs[entry].hash = c->fit_hash;
        c->shard_costs[entry].valid = false;  // until its first call finishes
        if (c->shard_last == entry) c->shard_last = -1;
    }
    // ranking by counting is O(nb_all^2 / 64) per call: beyond 16384 walkers the deal stays by position (equal counts)
    const bool ranked = c->shard_costs[entry].valid && nb_all <= 16384;
    c->shard_costs[entry].used = ++c->shard_clock;
    vag_ctx::ShardFlight& fl = c->shard_flights[slot];
    if (c->shard_costs[entry].cost.ensure(sizeof(double) * (size_t)nb_all)) return VAG_E_HIP;
    if (fl.table.ensure(sizeof(int) * (size_t)world * per)) return VAG_E_HIP;
    hipLaunchKernelGGL(vag_shard_deal_kernel, dim3((world * per + 3) / 4), dim3(256), 0, c->stream,
                       ranked ? c->shard_costs[entry].cost.as<double>() : nullptr, d_theta_all, nb_all, ndim, rank, world, per,
                       fl.table.as<int>(), c->d_shard_theta.as<double>());
    HIPCHK(hipGetLastError());
    fl.ticket = c->shard_clock;
    fl.key = entry;
    fl.nb = nb_all;
    fl.world = world;
    c->shard_last_dealt = slot;  // (kept for inspection even if the evaluation below fails)
    const int* d_order = nullptr;
    if (n_mine > 0) {
        int rc = loglike_body(c, spec, c->d_shard_theta.as<double>(), n_mine, ndim, c->d_shard_ll.as<double>(), !c->count_work);
        if (rc == VAG_RETRY) rc = loglike_body(c, spec, c->d_shard_theta.as<double>(), n_mine, ndim, c->d_shard_ll.as<double>(), false);
        if (rc) {  // (no block went out: there is nothing to finish, the slot is free)
            if (replaces) fl.in_use = false, --c->shard_costs[entry].in_flight;
            return rc;
        }
        d_order = c->order_active ? c->last_order : nullptr;
    }
    hipLaunchKernelGGL(vag_shard_pack_kernel, dim3((per + 127) / 128), dim3(128), 0, c->stream, c->d_shard_ll.as<double>(),
                       c->d_meta.as<VagGridMeta>(), d_order, n_mine, per, d_block);
    HIPCHK(hipGetLastError());
    fl.in_use = true;
    fl.unticketed = !ticket;
    if (!replaces) ++c->shard_costs[entry].in_flight;
    if (ticket) *ticket = fl.ticket;
    return VAG_OK;
}

// ticket 0: the oldest call in flight of this shape (the v9 form; right when calls of equal shape finish in the order of their deals)
static int shard_end(vag_ctx* c, uint64_t ticket, const double* d_gathered, int nb_all, int world, double* d_out) {
    if (!c) return set_err(VAG_E_INVALID, "null context");
    const bool abandon = ticket && !d_gathered && !d_out;  // the caller's collective failed: release the call, keep the costs as they are
    if (!abandon && (!d_gathered || !d_out)) return set_err(VAG_E_INVALID, "null context or buffer");
    int slot = -1;
    for (int i = 0; i < 8 && nb_all > 0; ++i) {
        const vag_ctx::ShardFlight& f = c->shard_flights[i];
        if (!f.in_use) continue;
        if (ticket ? f.ticket == ticket : (f.nb == nb_all && f.world == world && (slot < 0 || f.ticket < c->shard_flights[slot].ticket))) slot = i;
    }
    if (slot < 0) {
        if (ticket) return set_err(VAG_E_INVALID, "no sharded call with ticket %llu is waiting for its finish", (unsigned long long)ticket);
        return set_err(VAG_E_INVALID, "no vag_loglike_shard_dev call of %d walkers over %d ranks is waiting for its finish", nb_all, world);
    }
    vag_ctx::ShardFlight& fl = c->shard_flights[slot];
    if (fl.nb != nb_all || fl.world != world)
        return set_err(VAG_E_INVALID, "ticket %llu was dealt for %d walkers over %d ranks, not %d over %d", (unsigned long long)ticket, fl.nb, fl.world,
                       nb_all, world);
    vag_ctx::ShardCosts& key = c->shard_costs[fl.key];
    if (abandon) {
        --key.in_flight;
        fl.in_use = false;
        return VAG_OK;
    }
    HIPCHK(hipSetDevice(c->device));
    const int per = (nb_all + world - 1) / world;
    hipLaunchKernelGGL(vag_shard_finish_kernel, dim3(1), dim3(1024), 0, c->stream, d_gathered, fl.table.as<int>(), world * per, d_out,
                       key.cost.as<double>());
    HIPCHK(hipGetLastError());
    key.valid = true;
    --key.in_flight;
    fl.in_use = false;
    c->shard_last = fl.key;
    return VAG_OK;
}

int vag_loglike_shard_dev(vag_ctx* c, const vag_fit_spec* spec, const double* d_theta_all, int nb_all, int ndim, int rank, int world,
                          double* d_block) {
    ApiLock api_lock(c);
    HandoffScope handoff(c);
    return shard_begin(c, spec, d_theta_all, nb_all, ndim, rank, world, d_block, nullptr);
}

int vag_loglike_shard_finish_dev(vag_ctx* c, const double* d_gathered, int nb_all, int world, double* d_out) {
    ApiLock api_lock(c);
    HandoffScope handoff(c);
    return shard_end(c, 0, d_gathered, nb_all, world, d_out);
}

int vag_loglike_shard_begin_dev(vag_ctx* c, const vag_fit_spec* spec, const double* d_theta_all, int nb_all, int ndim, int rank, int world,
                                double* d_block, uint64_t* ticket) {
    ApiLock api_lock(c);
    HandoffScope handoff(c);
    if (!ticket) return set_err(VAG_E_INVALID, "null ticket");
    *ticket = 0;
    return shard_begin(c, spec, d_theta_all, nb_all, ndim, rank, world, d_block, ticket);
}

int vag_loglike_shard_end_dev(vag_ctx* c, uint64_t ticket, const double* d_gathered, int nb_all, int world, double* d_out) {
    ApiLock api_lock(c);
    HandoffScope handoff(c);
    if (!ticket) return set_err(VAG_E_INVALID, "ticket 0 names no call");
    return shard_end(c, ticket, d_gathered, nb_all, world, d_out);
}

int vag_loglike_shard_state_dev(vag_ctx* c, int nb_all, int world, int32_t* d_table, double* d_cost) {
    ApiLock api_lock(c);
    HandoffScope handoff(c);
    if (!c) return set_err(VAG_E_INVALID, "null context");
    const int per = world > 0 ? (nb_all + world - 1) / world : 0;
    const int ld = c->shard_last_dealt;
    if (nb_all <= 0 || world <= 0 || ld < 0 || c->shard_flights[ld].nb != nb_all || c->shard_flights[ld].world != world ||
        c->shard_flights[ld].table.cap < sizeof(int) * (size_t)world * per)
        return set_err(VAG_E_INVALID, "no deal of %d walkers over %d ranks on this context", nb_all, world);
    HIPCHK(hipSetDevice(c->device));
    if (d_table)
        HIPCHK(hipMemcpyAsync(d_table, c->shard_flights[ld].table.p, sizeof(int) * (size_t)world * per, hipMemcpyDeviceToDevice, c->stream));
    if (d_cost) {
        if (c->shard_last < 0 || !c->shard_costs[c->shard_last].valid || c->shard_costs[c->shard_last].nb != nb_all || c->shard_costs[c->shard_last].world != world)
            return set_err(VAG_E_INVALID, "no finished sharded call of %d walkers over %d ranks on this context", nb_all, world);
        HIPCHK(hipMemcpyAsync(d_cost, c->shard_costs[c->shard_last].cost.p, sizeof(double) * (size_t)nb_all, hipMemcpyDeviceToDevice, c->stream));
    }
    return VAG_OK;
}

int vag_loglike_batch(vag_ctx* c, const vag_fit_spec* spec, const double* theta, int nb, int ndim, double* out) {
    ApiLock api_lock(c);
    if (!c) return set_err(VAG_E_INVALID, "null context");
    if (!spec || !theta || !out || ndim <= 0) return set_err(VAG_E_INVALID, "null spec, sample or output array");
    if (nb <= 0) return set_err(VAG_E_INVALID, "batch must be non-empty");
    HIPCHK(hipSetDevice(c->device));
    if (c->d_theta_in.ensure(sizeof(double) * (size_t)nb * (ndim + 1))) return VAG_E_HIP;
    double* d_theta = c->d_theta_in.as<double>();
    double* d_out = d_theta + (size_t)nb * ndim;
    HIPCHK(hipMemcpyAsync(d_theta, theta, sizeof(double) * (size_t)nb * ndim, hipMemcpyHostToDevice, c->stream));
    int rc = vag_loglike_batch_dev(c, spec, d_theta, nb, ndim, d_out);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(out, d_out, sizeof(double) * nb, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    (void)collect_times(c);
    return VAG_OK;
}

// ------------------------------------------------------------------------------------------------
// Coalescer (ABI v11, opt-in): the reference's own calling pattern is a thread pool with one Model per thread and the GIL released
// inside the compute methods (pybind/pybind.cpp:424-448; VegasAfterglow/fitting/samplers.py:59-70 maps eval_one over a
// ThreadPoolExecutor).  Serialised on one GPU context that pattern runs at one light curve per call latency.  The *_coalesced entry
// points take ONE model each, block the calling thread, and gather the calls that are waiting at the same time -- same request
// (method, times, frequencies / band) -- into one batch call of the ordinary entry point; every caller gets its own model's slice.
// Group commit: the first caller to find no batch forming becomes the leader, waits up to co_wait_us for company (none needed while
// the GPU is busy: whoever arrives during a batch call is queued for the next one), runs the batch under the context lock, hands
// the results out and promotes the next leader.  A batch that fails as a whole (one member's grid over capacity, say) is repeated
// member by member, so a caller only ever sees the error of its own model.
// ------------------------------------------------------------------------------------------------
struct CoalesceRequest {
    int kind;  // 0 grid, 1 series, 2 band; +4: components form (out4) instead of the total (out)
    const vag_model_params* p;
    const double *t, *nu;
    int nt, nnu;  // series: nnu == nt
    double nu_min, nu_max;
    int num_nu;
    double* out;
    double* const* out4;
    int rc = VAG_OK;
    std::string err;
    bool done = false, promoted = false;
    std::condition_variable cv;  // this member's own wake-up (done / promoted): an arrival or a finished batch wakes who it concerns,
                                 // not every sleeping thread of the pool (r05: one shared condition variable with notify_all made
                                 // each arrival wake ~20 sleepers, each of which took the mutex to find nothing for it)
    bool same_request(const CoalesceRequest& o) const {
        if (kind != o.kind || nt != o.nt || nnu != o.nnu) return false;
        if ((kind & 3) == 2) return nu_min == o.nu_min && nu_max == o.nu_max && num_nu == o.num_nu && (t == o.t || !std::memcmp(t, o.t, sizeof(double) * nt));
        return (t == o.t || !std::memcmp(t, o.t, sizeof(double) * nt)) && (nu == o.nu || !std::memcmp(nu, o.nu, sizeof(double) * nnu));
    }
    size_t out_len() const { return (kind & 3) == 0 ? (size_t)nt * nnu : (size_t)nt; }
};

static int coalesce_run_one(vag_ctx* c, CoalesceRequest& r, const vag_model_params* params, int nb, double* out, double* const* out4) {
    switch (r.kind) {
        case 0: return vag_flux_density_grid_batch(c, params, nb, r.t, r.nt, r.nu, r.nnu, out);
        case 4: return vag_flux_density_grid_components4_batch(c, params, nb, r.t, r.nt, r.nu, r.nnu, out4);
        case 1: return vag_flux_density_batch(c, params, nb, r.t, r.nu, r.nt, out);
        case 5: return vag_flux_density_components4_batch(c, params, nb, r.t, r.nu, r.nt, out4);
        case 2: return vag_flux_batch(c, params, nb, r.t, r.nt, r.nu_min, r.nu_max, r.num_nu, out);
        default: return vag_flux_components4_batch(c, params, nb, r.t, r.nt, r.nu_min, r.nu_max, r.num_nu, out4);
    }
}

static void coalesce_serve(vag_ctx* c, std::vector<CoalesceRequest*>& batch) {
    const int nb = (int)batch.size();
    CoalesceRequest& lead = *batch[0];
    const size_t len = lead.out_len();
    bool ok = false;
    if (nb > 1) try {
        std::vector<vag_model_params> params(nb);
        for (int i = 0; i < nb; ++i) params[i] = *batch[i]->p;
        const bool comps = (lead.kind & 4) != 0;
        bool want[4] = {!comps, false, false, false};
        if (comps)
            for (int q = 0; q < 4; ++q)
                for (int i = 0; i < nb; ++i) want[q] = want[q] || batch[i]->out4[q] != nullptr;
        std::vector<double> buf[4];
        double* o4[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int q = 0; q < 4; ++q)
            if (want[q]) {
                buf[q].resize((size_t)nb * len);
                o4[q] = buf[q].data();
            }
        const int rc = coalesce_run_one(c, lead, params.data(), nb, o4[0], o4);
        if (rc == VAG_OK) {
            for (int i = 0; i < nb; ++i) {
                if (comps) {
                    for (int q = 0; q < 4; ++q)
                        if (batch[i]->out4[q]) std::memcpy(batch[i]->out4[q], o4[q] + (size_t)i * len, sizeof(double) * len);
                } else {
                    std::memcpy(batch[i]->out, o4[0] + (size_t)i * len, sizeof(double) * len);
                }
                batch[i]->rc = VAG_OK;
            }
            ok = true;
        }
    } catch (const std::exception&) {
        // the staging buffers could not be allocated: no exception crosses the C ABI -- the members are served one by one below,
        // straight into their own buffers (that path allocates nothing on the host)
        ok = false;
    }
    if (!ok)  // alone, or the batch failed as a whole: every member on its own, with its own error
        for (int i = 0; i < nb; ++i) {
            CoalesceRequest& r = *batch[i];
            r.rc = coalesce_run_one(c, r, r.p, 1, r.out, r.out4);
            if (r.rc) r.err = g_err;
        }
}

static int coalesce_submit(vag_ctx* c, CoalesceRequest& req) {
    std::unique_lock<std::mutex> lk(c->co_mutex);
    c->co_queue.push_back(&req);
    ++c->co_calls;
    if (!c->co_leader) {
        c->co_leader = true;
    } else {
        c->co_cv.notify_one();  // (the leader counts the queue)
        req.cv.wait(lk, [&] { return req.done || req.promoted; });
        if (req.done) {
            if (req.rc) g_err = req.err;
            return req.rc;
        }
        // promoted: this request heads the next batch
    }
    {   // the leader waits up to co_wait_us for company (callers of a pool arrive one interpreter hand-off apart)
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(c->co_wait_us);
        while ((int)c->co_queue.size() < c->co_max_batch && c->co_cv.wait_until(lk, deadline) != std::cv_status::timeout) {
        }
    }
    // the leader's batch: its own request and every queued one that asks the same thing, in arrival order
    std::vector<CoalesceRequest*> batch{&req};
    std::vector<CoalesceRequest*> rest;
    for (CoalesceRequest* r : c->co_queue) {
        if (r == &req) continue;
        if ((int)batch.size() < c->co_max_batch && r->same_request(req))
            batch.push_back(r);
        else
            rest.push_back(r);
    }
    c->co_queue.swap(rest);
    ++c->co_batches;
    lk.unlock();
    coalesce_serve(c, batch);  // (takes the context lock inside the entry points)
    lk.lock();
    for (CoalesceRequest* r : batch)
        if (r != &req) {
            r->done = true;
            r->cv.notify_one();  // (under the mutex: the member cannot return -- and its request leave the stack -- before this)
        }
    if (!c->co_queue.empty()) {
        c->co_queue.front()->promoted = true;  // stays the leader-in-waiting: co_leader remains set
        c->co_queue.front()->cv.notify_one();
    } else {
        c->co_leader = false;
    }
    lk.unlock();
    if (req.rc) g_err = req.err;
    return req.rc;
}

int vag_ctx_coalesce(vag_ctx* c, int max_batch, int wait_us) {
    if (!c) return set_err(VAG_E_INVALID, "null context");
    if (max_batch < 1 || wait_us < 0) return set_err(VAG_E_INVALID, "max_batch must be >= 1 and wait_us >= 0");
    std::lock_guard<std::mutex> lk(c->co_mutex);
    c->co_max_batch = max_batch;
    c->co_wait_us = wait_us;
    return VAG_OK;
}

int vag_ctx_coalesce_stats(vag_ctx* c, long long* calls, long long* batches) {
    if (!c) return set_err(VAG_E_INVALID, "null context");
    std::lock_guard<std::mutex> lk(c->co_mutex);
    if (calls) *calls = c->co_calls;
    if (batches) *batches = c->co_batches;
    return VAG_OK;
}

int vag_flux_density_grid_coalesced(vag_ctx* c, const vag_model_params* p, const double* t, int nt, const double* nu, int nnu, double* out,
                                    double* const* out4) {
    if (!c || !p || !t || !nu || (!out && !out4)) return set_err(VAG_E_INVALID, "null context, model or buffer");
    if (nt <= 0 || nnu <= 0) return set_err(VAG_E_INVALID, "time and frequency arrays must be non-empty");
    CoalesceRequest r{out4 ? 4 : 0, p, t, nu, nt, nnu, 0, 0, 0, out, out4};
    return coalesce_submit(c, r);
}

int vag_flux_density_coalesced(vag_ctx* c, const vag_model_params* p, const double* t, const double* nu, int n, double* out,
                               double* const* out4) {
    if (!c || !p || !t || !nu || (!out && !out4)) return set_err(VAG_E_INVALID, "null context, model or buffer");
    if (n <= 0) return set_err(VAG_E_INVALID, "time array must be non-empty");
    CoalesceRequest r{out4 ? 5 : 1, p, t, nu, n, n, 0, 0, 0, out, out4};
    return coalesce_submit(c, r);
}

int vag_flux_coalesced(vag_ctx* c, const vag_model_params* p, const double* t, int nt, double nu_min, double nu_max, int num_nu, double* out,
                       double* const* out4) {
    if (!c || !p || !t || (!out && !out4)) return set_err(VAG_E_INVALID, "null context, model or buffer");
    if (nt <= 0) return set_err(VAG_E_INVALID, "time array must be non-empty");
    CoalesceRequest r{out4 ? 6 : 2, p, t, nullptr, nt, 0, nu_min, nu_max, num_nu, out, out4};
    return coalesce_submit(c, r);
}


static int details_impl(vag_ctx* c, const vag_model_params* params, double t_min, double t_max, vag_details_shape* shape,
                        const vag_details_out* out, bool want_rvs) {
    if (!c) return set_err(VAG_E_INVALID, "null context");
    if (!params || !shape) return set_err(VAG_E_INVALID, "null model or shape");
    if (want_rvs && !(params->flags & VAG_FLAG_RVS)) return set_err(VAG_E_INVALID, "model has no reverse shock");
    const char* msg = validate_msg(params);
    if (msg) return set_err(VAG_E_INVALID, "%s", msg);
    if (!(t_min > 0) || !(t_max >= t_min)) return set_err(VAG_E_INVALID, "need 0 < t_min <= t_max");
    HIPCHK(hipSetDevice(c->device));
    if (c->d_params.ensure(sizeof(vag_model_params))) return VAG_E_HIP;
    if (c->d_tminmax.ensure(sizeof(double) * 2)) return VAG_E_HIP;
    const double tmm[2] = {t_min, t_max};
    HIPCHK(hipMemcpyAsync(c->d_params.p, params, sizeof(vag_model_params), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_tminmax.p, tmm, sizeof tmm, hipMemcpyHostToDevice, c->stream));
    int rc = run_model_stages(c, c->d_params.as<vag_model_params>(), 1, true);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    rc = check_status(c, 1);
    if (rc) return rc;
    if (fetch_meta(c)) return VAG_E_HIP;
    const VagGridMeta M = c->h_meta.as<VagGridMeta>()[0];
    shape->n_phi = M.n_phi;
    shape->n_theta = M.n_theta;
    shape->n_t = M.n_t;
    shape->n_reps = M.rep_phi_stride ? M.n_theta : M.n_reps;  // ((phi, theta) pair rows: the arrays below are the phi[0] slice)
    shape->symmetry = M.symmetry;
    shape->phi_mirrored = M.phi_mirrored;
    if (!out) return VAG_OK;
    const int nth = M.n_theta, nt = M.n_t;
    if (vag_hook("VAG_DEBUG_ROWS")) {  // developer aid: ODE status (and injection cutoff) per representative row
        std::vector<int> st(c->n_rows), inj(c->n_rows, -1);
        HIPCHK(hipMemcpy(st.data(), c->d_row_status.p, sizeof(int) * c->n_rows, hipMemcpyDeviceToHost));
        if (params->flags & VAG_FLAG_RVS) HIPCHK(hipMemcpy(inj.data(), c->d_inj.p, sizeof(int) * c->n_rows, hipMemcpyDeviceToHost));
        for (int r = 0; r < c->n_rows; ++r) std::fprintf(stderr, "row %d status %d inj %d\n", r, st[r], inj[r]);
    }
    std::vector<double> theta(nth), buf((size_t)c->n_cells * VAG_NSHOCK);
    std::vector<int> rep_of(nth);
    HIPCHK(hipMemcpy(theta.data(), c->d_theta.p, sizeof(double) * nth, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(rep_of.data(), c->d_rep_of.p, sizeof(int) * nth, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(buf.data(), want_rvs ? c->d_shock_r.p : c->d_shock.p, sizeof(double) * buf.size(), hipMemcpyDeviceToHost));
    if (out->phi) HIPCHK(hipMemcpy(out->phi, c->d_phi.p, sizeof(double) * M.n_phi, hipMemcpyDeviceToHost));
    if (out->theta) std::memcpy(out->theta, theta.data(), sizeof(double) * nth);
    // Shock::broadcast_groups (src/dynamics/shock.cpp:42-91): every theta row shows its representative's state
    auto expand = [&](double* dst, int which, double scale) {
        if (!dst) return;
        const double* src = buf.data() + (size_t)which * c->n_cells;
        for (int j = 0; j < nth; ++j)
            for (int k = 0; k < nt; ++k) dst[(size_t)j * nt + k] = src[(size_t)rep_of[j] * nt + k] * scale;
    };
    expand(out->t_src, VS_TENG, 1 / U_SEC);
    expand(out->Gamma, VS_GAMMA, 1);
    expand(out->r, VS_R, 1 / U_CM);
    expand(out->t_comv, VS_TCOMV, 1 / U_SEC);
    expand(out->B, VS_B, 1 / U_GAUSS);
    expand(out->N_p, VS_NP, 1);
    expand(out->Gamma_th, VS_GAMMA_TH, 1);
    return VAG_OK;
}

int vag_details(vag_ctx* c, const vag_model_params* params, double t_min, double t_max, vag_details_shape* shape,
                const vag_details_out* out) {
    ApiLock api_lock(c);
    return details_impl(c, params, t_min, t_max, shape, out, false);
}

int vag_profile_eval(vag_ctx* c, const vag_model_params* params, int kind, const double* x, int n, double* out) {
    ApiLock api_lock(c);
    if (!c) return set_err(VAG_E_INVALID, "null context");
    if (kind < 0 || kind > 2) return set_err(VAG_E_INVALID, "profile kind must be 0 (E_iso), 1 (Gamma0) or 2 (rho)");
    if (n <= 0 || !x || !out) return set_err(VAG_E_INVALID, "empty abscissa array");
    const char* msg = validate_msg(params);
    if (msg) return set_err(VAG_E_INVALID, "%s", msg);
    HIPCHK(hipSetDevice(c->device));
    if (c->d_params.ensure(sizeof(vag_model_params))) return VAG_E_HIP;
    DevBuf tmp;
    if (tmp.ensure(sizeof(double) * 2 * (size_t)n)) return VAG_E_HIP;
    double* d_x = tmp.as<double>();
    hipError_t e = hipMemcpyAsync(c->d_params.p, params, sizeof(vag_model_params), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_x, x, sizeof(double) * n, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(vag_profile_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->d_params.as<vag_model_params>(),
                           kind, d_x, n, d_x + n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_x + n, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    tmp.release();
    if (e != hipSuccess) return set_err(VAG_E_HIP, "profile evaluation failed: %s", hipGetErrorString(e));
    return VAG_OK;
}

int vag_details_eat(vag_ctx* c, const vag_model_params* params, double t_min, double t_max, int* n_phi_eff, double* t_obs,
                    double* doppler) {
    ApiLock api_lock(c);
    vag_details_shape sh;
    int rc = details_impl(c, params, t_min, t_max, &sh, nullptr, false);
    if (rc) return rc;
    if (fetch_meta(c)) return VAG_E_HIP;
    const VagGridMeta M = c->h_meta.as<VagGridMeta>()[0];
    if (n_phi_eff) *n_phi_eff = M.n_phi_eff;
    if (!t_obs && !doppler) return VAG_OK;  // shape query
    if (!t_obs || !doppler) return set_err(VAG_E_INVALID, "t_obs and doppler must be given together");
    const size_t total = (size_t)M.n_phi_eff * M.n_theta * M.n_t;
    DevBuf tmp;
    if (tmp.ensure(sizeof(double) * 2 * total)) return VAG_E_HIP;
    double* d_t = tmp.as<double>();
    hipLaunchKernelGGL(vag_eat_details_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream,
                       c->d_params.as<vag_model_params>(), c->d_meta.as<VagGridMeta>(), c->d_geo_th.as<double>(),
                       c->d_geo_ph.as<double>(), c->d_rep_of.as<int>(), c->d_cellpar.as<double>(),
                       (params->flags & VAG_FLAG_SPREADING) ? c->d_cellgeo.as<double>() : nullptr, d_t, d_t + total);
    if (hipGetLastError() != hipSuccess) {
        tmp.release();
        return set_err(VAG_E_HIP, "vag_eat_details_kernel launch failed");
    }
    hipError_t e1 = hipMemcpyAsync(t_obs, d_t, sizeof(double) * total, hipMemcpyDeviceToHost, c->stream);
    hipError_t e2 = hipMemcpyAsync(doppler, d_t + total, sizeof(double) * total, hipMemcpyDeviceToHost, c->stream);
    hipError_t e3 = hipStreamSynchronize(c->stream);
    tmp.release();
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) return set_err(VAG_E_HIP, "copying the EAT details failed");
    return VAG_OK;
}

int vag_details_radiation(vag_ctx* c, const vag_model_params* params, double t_min, double t_max, int rvs, double* const* arrays) {
    ApiLock api_lock(c);
    vag_details_shape sh;
    int rc = details_impl(c, params, t_min, t_max, &sh, nullptr, rvs != 0);  // runs the stages with the electron arrays kept
    if (rc) return rc;
    if (!arrays) return set_err(VAG_E_INVALID, "arrays must not be null");
    const int nth = sh.n_theta, nt = sh.n_t;
    const long long cells = c->n_cells;
    std::vector<double> buf((size_t)cells * 11), th((size_t)cells, 0.0);
    std::vector<int> rep_of(nth);
    const DevBuf& det = rvs ? c->d_celldet_r : c->d_celldet;
    const DevBuf& shock = rvs ? c->d_shock_r : c->d_shock;
    HIPCHK(hipMemcpy(buf.data(), det.p, sizeof(double) * buf.size(), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(rep_of.data(), c->d_rep_of.p, sizeof(int) * nth, hipMemcpyDeviceToHost));
    const bool per_cell_theta = (params->flags & VAG_FLAG_SPREADING) != 0;
    if (per_cell_theta)
        HIPCHK(hipMemcpy(th.data(), static_cast<const double*>(shock.p) + (size_t)VS_THETA * cells, sizeof(double) * cells,
                         hipMemcpyDeviceToHost));
    std::vector<double> theta(nth);
    HIPCHK(hipMemcpy(theta.data(), c->d_theta.p, sizeof(double) * nth, hipMemcpyDeviceToHost));
    // VD_* rows: gamma_m, gamma_c, gamma_a, gamma_M, N_e, column_den, nu_m, nu_c, nu_a, nu_M, I_nu_max (code units)
    const int src_row[10] = {0, 1, 2, 3, 4, 6, 7, 8, 9, 10};
    const double scale[10] = {1, 1, 1, 1, 1, 1 / U_HZ, 1 / U_HZ, 1 / U_HZ, 1 / U_HZ, 1 / U_FLUX_DEN_CGS};
    for (int a = 0; a < 10; ++a) {
        if (!arrays[a]) continue;
        const double* src = buf.data() + (size_t)src_row[a] * cells;
        for (int j = 0; j < nth; ++j)
            for (int k = 0; k < nt; ++k) arrays[a][(size_t)j * nt + k] = src[(size_t)rep_of[j] * nt + k] * scale[a];
    }
    if (arrays[10])  // ShockDetails.theta: the evolved polar angle of a spreading jet, the grid angle otherwise
        for (int j = 0; j < nth; ++j)
            for (int k = 0; k < nt; ++k)
                arrays[10][(size_t)j * nt + k] = (params->flags & VAG_FLAG_SPREADING) ? th[(size_t)rep_of[j] * nt + k] : theta[j];
    return VAG_OK;
}

// SynElectrons::regime of every (theta, t) cell (determine_regime, synchrotron.cpp:45-60: the ordering of gamma_a, gamma_c, gamma_m,
// 1 ... 6, 0 = none), an integer the reference's parity contract covers exactly: from the electron rows the stages left -- and, where
// the SSC stages ran, checked against the tag vag_photons_ic_kernel stored for its IC kernels (VD_REGIME).
int vag_details_regime(vag_ctx* c, const vag_model_params* params, double t_min, double t_max, int rvs, int32_t* regime) {
    ApiLock api_lock(c);
    vag_details_shape sh;
    int rc = details_impl(c, params, t_min, t_max, &sh, nullptr, rvs != 0);
    if (rc) return rc;
    if (!regime) return set_err(VAG_E_INVALID, "regime must not be null");
    const int nth = sh.n_theta, nt = sh.n_t;
    const long long cells = c->n_cells;
    std::vector<double> buf((size_t)cells * VAG_NDET);
    std::vector<int> rep_of(nth);
    const DevBuf& det = rvs ? c->d_celldet_r : c->d_celldet;
    HIPCHK(hipMemcpy(buf.data(), det.p, sizeof(double) * buf.size(), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(rep_of.data(), c->d_rep_of.p, sizeof(int) * nth, hipMemcpyDeviceToHost));
    const bool ssc = rvs ? (params->flags & VAG_FLAG_RVS_SSC) != 0 : (params->flags & VAG_FLAG_SSC) != 0;
    auto order = [](double a, double cc, double m) {  // determine_regime
        if (a <= m && m <= cc) return 1;
        if (m <= a && a <= cc) return 2;
        if (a <= cc && cc <= m) return 3;
        if (cc <= a && a <= m) return 4;
        if (m <= cc && cc <= a) return 5;
        if (cc <= m && m <= a) return 6;
        return 0;
    };
    for (int j = 0; j < nth; ++j)
        for (int k = 0; k < nt; ++k) {
            const size_t q = (size_t)rep_of[j] * nt + k;
            const int tag = order(buf[(size_t)VD_GAMMA_A * cells + q], buf[(size_t)VD_GAMMA_C * cells + q], buf[(size_t)VD_GAMMA_M * cells + q]);
            if (ssc && tag != (int)buf[(size_t)VD_REGIME * cells + q])
                return set_err(VAG_E_INTERNAL, "cell (%d, %d): the regime tag of the IC kernels (%d) is not the ordering of the stored electrons (%d)",
                               j, k, (int)buf[(size_t)VD_REGIME * cells + q], tag);
            regime[(size_t)j * nt + k] = tag;
        }
    return VAG_OK;
}

int vag_details_rvs(vag_ctx* c, const vag_model_params* params, double t_min, double t_max, vag_details_shape* shape,
                    const vag_details_out* out) {
    ApiLock api_lock(c);
    return details_impl(c, params, t_min, t_max, shape, out, true);
}

}  // extern "C"
