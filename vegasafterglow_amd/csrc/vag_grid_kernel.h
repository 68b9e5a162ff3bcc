// vag_grid_kernel.h -- adaptive (phi, theta, t) grid on the device, one wavefront per model.
//
// Restates auto_grid (src/core/grid-refinement.h:639-706) for axisymmetric, non-spreading
// named jets.  The sequential skeleton (CDF integration, merges) is executed uniformly by all
// 64 lanes of the model's wavefront; every loop whose iterations are independent (profile scans,
// the theta sum inside the phi weight, per-theta deceleration times, inverse-CDF lookups) is
// spread over the lanes, with __shfl_xor butterflies for the sums/minima.
#pragma once
#include <type_traits>

#include "vag_device.h"

namespace vag {

constexpr int WAVE = 64;
constexpr int N_SCAN = 512;      // find_jet_jumps / find_theta_range scans
constexpr int N_SAMPLES = 200;   // defaults::sampling::theta_samples
constexpr int QUAD_REC = 24;     // accepted CDF-quadrature steps buffered before their samples are interpolated (a quadrature takes 23-33 steps: one or
                                 // two flushes).  With this and 16-bit merge flags the small layout is 27.0 KB: SIX models resident per CU instead of
                                 // four (33 KB at 96 records) -- the grid stage of batches larger than the chip holds at once

#ifdef VAG_GRID_STAMPS  // developer aid: cycle stamps of model 0 at the section boundaries, printed by lane 0
#define VAG_GRID_STAMP(i) do { if (m == 0 && lane == 0) stamps_[i] = __builtin_readcyclecounter(); } while (0)
#else
#define VAG_GRID_STAMP(i) do { } while (0)
#endif

// Sum over the wavefront, the same value in every lane: DPP prefix scan (VALU only; the __shfl_xor butterfly makes six
// round trips through the LDS crossbar) + a broadcast of lane 63's total.
VAG_DEV double wave_sum(double v) {
    v = wave_prefix_sum(v);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
VAG_DEV double wave_min(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off, WAVE));
    return v;
}
VAG_DEV int wave_min_int(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, WAVE));
    return v;
}

// xt::linspace element (external/xtensor/generators/xbuilder.hpp:231-280,460-471)
VAG_DEV double linspace_at(double start, double stop, int n, int i) {
    const double step = (stop - start) / fmax(1.0, (double)(n - 1));
    return (n > 1 && i == n - 1) ? stop : start + step * (double)i;
}

// LDS of one wavefront (= one model).  Arrays with disjoint lifetimes share memory (what decides how many models a CU holds when a
// batch is larger than the chip):
//   profile scans + theta quantiles (find_jet_jumps ... merge_grids)   |  phi-weight constants (adaptive_phi_grid)
//   CDF samples (both inverse_CFD_sampling calls)                       |  t_dec per row (build_time_grid, after the phi grid)
// The wavefront's scratch arrays.  Two sizes exist: the angular grids of every default-resolution model fit (VAG_GRID_THETA,
// VAG_GRID_PHI) = (256, 208) -- 20 KB, eight models resident per CU --, and a batch in which some model asks for more is laid out
// again with (VAG_MAX_THETA, VAG_MAX_PHI) -- 97 KB, one model per CU.  The arrays in HBM are strided by the layout in use.
template <int MAXTH, int MAXPH>
struct GridSharedT {
    static constexpr int max_theta = MAXTH, max_phi = MAXPH;
    union {
        struct {
            double scan_g[N_SCAN + 8];   // Gamma0 at the scan abscissae (find_jet_jumps)
            double base[MAXTH];  // theta quantiles before the jump nodes are merged in
        };
        struct {  // per-theta constants of the phi weight (adaptive_phi_grid, grid-refinement.h:296-331)
            double pj_beta[MAXTH], pj_sw[MAXTH], pj_dcos[MAXTH], pj_ct[MAXTH], pj_st[MAXTH];
        };
    };
    union {
        struct {
            double xs[N_SAMPLES];   // CDF sample abscissae
            double cdf[N_SAMPLES];  // CDF at xs
        };
        double tdec[MAXTH];
    };
    double theta[MAXTH + 64];
    double phi[MAXPH];
    short flag[MAXTH];  // merge bookkeeping: node indices < MAXTH, -1; group-start marks
    double rec[9][QUAD_REC];          // accepted quadrature steps awaiting their dense-output pass (integrate_cdf)
    double jumps[VAG_MAX_JUMPS];      // find_jet_jumps results (dynamically indexed: LDS, not scratch)
    double feat[3 * VAG_MAX_JUMPS];   // jump_refinement_grid nodes
};

// cos(x) for |x| <= pi + a bit: 1 - 2 sin^2(x/2) with sin's odd series on [-pi/2, pi/2] to x^21 (truncation 3e-16 at the end of
// the range).  The CDF integrands (theta - theta_v, phi in [0, 2 pi] folded below) only ever need this range.
VAG_DEV double cos_small(double x) {
    const double h = 0.5 * x, z = h * h;
    double p = fma(z, -1.9572941063391263e-20, 8.22063524662433e-18);
    p = fma(p, z, -2.8114572543455206e-15);
    p = fma(p, z, 7.647163731819816e-13);
    p = fma(p, z, -1.6059043836821613e-10);
    p = fma(p, z, 2.505210838544172e-08);
    p = fma(p, z, -2.7557319223985893e-06);
    p = fma(p, z, 0.0001984126984126984);
    p = fma(p, z, -0.008333333333333333);
    p = fma(p, z, 0.16666666666666666);
    const double sn = fma(-(p * z), h, h);  // h - h^3/6 + ...
    return fma(-2.0 * sn, sn, 1.0);
}
// cos(phi) for phi in [0, 2 pi]: cos(phi) = -cos(phi - pi)
VAG_DEV double cos_0_2pi(double phi) { return -cos_small(phi - C_PI); }

// jet.Gamma0(theta) through the fast exp2 / log2 (used by the CDF integrands only: every other use -- cuts, symmetry classes,
// initial conditions -- keeps jet_Gamma0)
VAG_DEV double jet_Gamma0_fast(const Jet& j, double theta) {
    if (j.magnetar) return jet_Gamma0(j, theta);
    switch (j.type) {
        case VAG_JET_GAUSSIAN: return (j.Gamma0 - 1) * exp2_sat(theta * theta * (j.norm * LOG2E)) + 1;
        case VAG_JET_POWERLAW: return (j.Gamma0 - 1) * rcp_fast(1 + exp2_sat(j.k_g * log2_fast(theta * j.inv_theta_c))) + 1;
        case VAG_JET_STEP_POWERLAW:
            return (theta <= j.theta_c ? j.Gm1 : j.Gm1_w * exp2_sat(-j.k_g * log2_fast(theta * j.inv_theta_c))) + 1;
        case VAG_JET_POWERLAW_WING:
            return (theta <= j.theta_c ? 0. : j.Gm1_w * exp2_sat(-j.k_g * log2_fast(theta * j.inv_theta_c))) + 1;
        default: return jet_Gamma0(j, theta);  // piecewise-constant profiles: comparisons only
    }
}
VAG_DEV double beta_fast(double g) { return sqrt_fast((g - 1) * (g + 1)) * rcp_fast(g); }
VAG_DEV double structure_weight_fast(double G) { return G * sqrt_fast(dmax((G - 1) * G, 0.0)); }

VAG_DEV double lane_value(double v, int src_lane) {  // wave-uniform copy of lane src_lane's v
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}

// One accepted DOPRI5 step of d(cdf)/dx = pdf(x): the arithmetic of Dopri5<1>::step (vag_device.h) to the last bit, but
// the right-hand side does not depend on cdf, so the stage abscissae x + {1/5, 3/10, 4/5, 8/9, 1} h are known up front and
// their five pdf values are evaluated side by side (`stages(tx, kv)`; stages 6 and 7 share x + h) -- one pdf latency per
// attempted step instead of six.
#ifdef VAG_GRID_STAMPS
__device__ long long g_stage_cycles, g_dense_cycles;
__device__ int g_quad_steps;
#endif
template <class Stages>
VAG_DEV bool quad_step(Dopri5<1>& s, Stages& stages) {
    constexpr double a2 = 1.0 / 5, a3 = 3.0 / 10, a4 = 4.0 / 5, a5 = 8.0 / 9;
    constexpr double c1 = 35.0 / 384, c3 = 500.0 / 1113, c4 = 125.0 / 192, c5 = -2187.0 / 6784, c6 = 11.0 / 84;
    constexpr double dc1 = c1 - 5179.0 / 57600, dc3 = c3 - 7571.0 / 16695, dc4 = c4 - 393.0 / 640,
                     dc5 = c5 - -92097.0 / 339200, dc6 = c6 - 187.0 / 2100, dc7 = -1.0 / 40;
    s.t_old = s.t;
    for (int fails = 0; fails < 500; ++fails) {
        const double h = s.dt;
        const double tx[5] = {s.t + h * a2, s.t + h * a3, s.t + h * a4, s.t + h * a5, s.t + h};
        double kv[5];
#ifdef VAG_GRID_STAMPS
        const long long q0_ = __builtin_readcyclecounter();
#endif
        stages(tx, kv);
#ifdef VAG_GRID_STAMPS
        g_stage_cycles += __builtin_readcyclecounter() - q0_;
#endif
        const double k3 = kv[1], k4 = kv[2], k5 = kv[3], k6 = kv[4], k7 = kv[4];
        const double x = s.x[0], dx = s.dx[0];
        const double xn = x + (h * c1) * dx + (h * c3) * k3 + (h * c4) * k4 + (h * c5) * k5 + (h * c6) * k6;
        const double xe = (h * dc1) * dx + (h * dc3) * k3 + (h * dc4) * k4 + (h * dc5) * k5 + (h * dc6) * k6 + (h * dc7) * k7;
        double err = dmax(0.0, fabs(xe) * rcp_fast(s.eps + s.eps * (fabs(x) + fabs(h) * fabs(dx))));
        if (err > 1.0) {
            s.dt = h * dmax(9.0 / 10.0 * exp2_sat(log2_fast(err) * (-1.0 / 3)), 1.0 / 5.0);
            continue;
        }
        s.xo[0] = x;
        s.dxo[0] = dx;
        s.x[0] = xn;
        s.dx[0] = k7;
        s.k3[0] = k3;
        s.k4[0] = k4;
        s.k5[0] = k5;
        s.k6[0] = k6;
        s.t = s.t + h;
        if (err < 0.5) {
            err = dmax(3.2e-4, err);  // 5^-5
            s.dt = h * (9.0 / 10.0 * exp2_fast(log2_fast(err) * (-1.0 / 5)));
        }
        return true;
    }
    return false;
}

// Dense output of the buffered steps at the samples they passed: sample kk belongs to the first step whose end exceeds
// xs[kk] (`current_time() > x`, grid-refinement.h:150-158); one sample per lane, same arithmetic per sample as the serial sweep.
// Returns the index of the first sample not yet passed.
template <class SH>
VAG_DEV int flush_quad_steps(SH& sh, int n_rec, int k) {
    const int lane = threadIdx.x;
    if (n_rec == 0) return k;
    const double t_end = sh.rec[1][n_rec - 1];
    for (;;) {
        const int kk = k + lane;
        const bool in = kk < N_SAMPLES && t_end > sh.xs[kk];
        if (in) {
            const double x = sh.xs[kk];
            int lo = 0, hi = n_rec - 1;  // first record with t > x
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (sh.rec[1][mid] > x)
                    hi = mid;
                else
                    lo = mid + 1;
            }
            Dopri5<1> st;
            st.t_old = sh.rec[0][lo];
            st.t = sh.rec[1][lo];
            st.xo[0] = sh.rec[2][lo];
            st.dxo[0] = sh.rec[3][lo];
            st.k3[0] = sh.rec[4][lo];
            st.k4[0] = sh.rec[5][lo];
            st.k5[0] = sh.rec[6][lo];
            st.k6[0] = sh.rec[7][lo];
            st.dx[0] = sh.rec[8][lo];
            double v;
            st.interp(x, &v);
            sh.cdf[kk] = v;
        }
        const int cnt = __popcll(__ballot(in));  // xs ascends: the passed samples are a prefix of the lanes
        k += cnt;
        if (cnt < WAVE) break;
    }
    return k;
}

// Integrate d(cdf)/dx = pdf(x) from lo to hi with boost's dense-output DOPRI5 at rtol=atol=1e-6 and
// sample it at sh.xs[1..N_SAMPLES) (inverse_CFD_sampling, grid-refinement.h:138-161).
// pdf0 = pdf(lo) (wave-uniform); stages(tx[5], kv[5]) returns the wave-uniform pdf at five abscissae.
// The dense output is off the stepping chain: accepted steps are buffered in LDS (nine doubles each) and interpolated at the
// samples they passed afterwards, one sample per lane.
template <class SH, class Stages>
VAG_DEV void integrate_cdf(SH& sh, double pdf0, Stages& stages, double lo, double hi) {
    const int lane = threadIdx.x;
    for (int k = lane; k < N_SAMPLES; k += WAVE) sh.cdf[k] = 0;
    __syncthreads();
    Dopri5<1> st;
    st.x[0] = 0;
    st.dx[0] = pdf0;
    st.t = lo;
    st.dt = (hi - lo) / 1e3;
    st.eps = 1e-6;
    int k = 1, n_rec = 0;
    for (int steps = 0; st.t <= hi;) {
        if (!quad_step(st, stages)) break;
        if (++steps > 100000) break;
#ifdef VAG_GRID_STAMPS
        ++g_quad_steps;
#endif
        if (lane == 0) {
            sh.rec[0][n_rec] = st.t_old;
            sh.rec[1][n_rec] = st.t;
            sh.rec[2][n_rec] = st.xo[0];
            sh.rec[3][n_rec] = st.dxo[0];
            sh.rec[4][n_rec] = st.k3[0];
            sh.rec[5][n_rec] = st.k4[0];
            sh.rec[6][n_rec] = st.k5[0];
            sh.rec[7][n_rec] = st.k6[0];
            sh.rec[8][n_rec] = st.dx[0];
        }
        if (++n_rec == QUAD_REC) {
            __syncthreads();
            k = flush_quad_steps(sh, n_rec, k);
            __syncthreads();
            n_rec = 0;
        }
    }
    __syncthreads();
#ifdef VAG_GRID_STAMPS
    const long long d0_ = __builtin_readcyclecounter();
#endif
    k = flush_quad_steps(sh, n_rec, k);
#ifdef VAG_GRID_STAMPS
    g_dense_cycles += __builtin_readcyclecounter() - d0_;
#endif
    __syncthreads();
}

// x_out[k] for CDF quantiles (grid-refinement.h:163-188); lanes split k.  out may alias nothing in sh.xs/cdf.
template <class SH>
VAG_DEV void invert_cdf(const SH& sh, int num, bool midpoint, double* out) {
    const double front = sh.cdf[0], back = sh.cdf[N_SAMPLES - 1];
    for (int k = threadIdx.x; k < num; k += WAVE) {
        const double target = midpoint ? front + (back - front) * ((double)k + 0.5) / num : linspace_at(front, back, num, k);
        double x = 0;
        // first j with target <= cdf[j]; the CDF rises strictly (every pdf carries a positive floor)
        int lo = -1, hi = N_SAMPLES;  // cdf[lo] < target <= cdf[hi], with virtual ends
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (target <= sh.cdf[mid])
                hi = mid;
            else
                lo = mid;
        }
        if (hi < N_SAMPLES) {
            const int j = hi;
            if (j == 0) {
                x = sh.xs[0];
            } else {
                const double denom = sh.cdf[j] - sh.cdf[j - 1];
                x = denom > 0 ? sh.xs[j - 1] + (sh.xs[j] - sh.xs[j - 1]) / denom * (target - sh.cdf[j - 1]) : sh.xs[j - 1];
            }
        }
        out[k] = x;
    }
    __syncthreads();
}

// Compact layout of a batch from its grid results, by ONE wavefront (the last one of vag_grid_kernel to finish): exclusive
// scans of rows (n_reps) and cells (n_reps x n_t) over the models -> row_off / cell_off [nb + 1], of the blocks of 64 (theta, phi)
// rows the row-per-lane flux kernels deal out (row_off[nb + 1 ...]: a second [nb + 1] array behind the first), and the totals / maxima /
// flag summary the host plans the later stages with (VagDevPlan).  When a total exceeds the capacity the host sized its
// buffers and launches for (it plans ahead from the previous call of the same batch size instead of waiting for this
// summary), every model is marked VAG_E_CAPACITY and the offsets are zeroed: the later kernels find no work and write
// nothing; the host sees `overflow` at the end of the call and repeats it.
VAG_DEV void plan_scan_wave(VagGridMeta* meta, int nb, int* __restrict__ row_off, long long* __restrict__ cell_off,
                            VagDevPlan* __restrict__ plan, VagDevPlan* host_plan /* pinned, host-mapped */, int seq, int cap_rows,
                            long long cap_cells, int cap_k, int cap_pairs, int expect_flags, int expect_dyn,
                            float* __restrict__ cost /* [nb]: (theta, phi, t) cells of each model, 0 for one not evaluated */) {
    const int t = threadIdx.x;
    const int per = (nb + WAVE - 1) / WAVE, m0 = min(nb, t * per), m1 = min(nb, m0 + per);
    long long cells = 0, pairs = 0, eat = 0;
    int rows = 0, blks = 0, max_k = 2, max_pairs = 0, n_ok = 0, n_inv = 0, n_cap = 0, first = -1, mixed = 0, dyn = 0;
    // a lane's models four at a time: their records are requested together (one trip to memory per four models instead of one per
    // model -- this wavefront runs after every other one of the launch has left, so its chain of trips is pure latency of the call:
    // ~1 us per model and lane, i.e. 16 / 128 trips per lane at 1024 / 8192 models)
    struct MetaHead {  // the leading members of VagGridMeta the scan reads
        int32_t status, n_phi, n_theta, n_t, n_reps, symmetry, phi_mirrored, n_phi_eff, t_num_tot, has_early, flags, t_num_base, dyn_class;
    };
    static_assert(offsetof(VagGridMeta, dyn_class) == offsetof(MetaHead, dyn_class), "MetaHead mirrors the head of VagGridMeta");
    constexpr int SCAN_U = 4;
    auto load_heads = [&](int m, MetaHead (&H)[SCAN_U]) {
#pragma unroll
        for (int u = 0; u < SCAN_U; ++u) H[u] = *reinterpret_cast<const MetaHead*>(meta + min(m + u, nb - 1));
    };
    for (int mb = m0; mb < m1; mb += SCAN_U) {
        MetaHead H[SCAN_U];
        load_heads(mb, H);
#pragma unroll
        for (int u = 0; u < SCAN_U; ++u) {
            const int m = mb + u;
            if (m >= m1) break;
            const MetaHead& M = H[u];
            cost[m] = M.status == 0 ? (float)M.n_theta * (float)M.n_phi_eff * (float)M.n_t : 0.0f;
            if (M.status == 0) {
                rows += M.n_reps;
                cells += (long long)M.n_reps * M.n_t;
                max_k = max(max_k, M.n_t);
                const int pr = M.n_theta * M.n_phi_eff;
                max_pairs = max(max_pairs, pr);
                pairs += pr;
                blks += (pr + 63) >> 6;
                eat += (long long)pr * M.n_t;
                dyn |= M.dyn_class;
                if (first < 0) first = M.flags;
                mixed |= (M.flags != first) ? 1 : 0;
                ++n_ok;
            } else if (M.status == VAG_E_CAPACITY) {
                ++n_cap;
            } else {
                ++n_inv;
            }
        }
    }
    // inclusive scans of rows / cells over the lanes, totals and summaries by butterflies
    int r_inc = rows, b_inc = blks;
    long long c_inc = cells;
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
        const int ro = __shfl_up(r_inc, off, WAVE), bo = __shfl_up(b_inc, off, WAVE);
        const long long co = __shfl_up(c_inc, off, WAVE);
        if (t >= off) {
            r_inc += ro;
            b_inc += bo;
            c_inc += co;
        }
    }
    const int tot_rows = __shfl(r_inc, WAVE - 1, WAVE), tot_blks = __shfl(b_inc, WAVE - 1, WAVE);
    const long long tot_cells = __shfl(c_inc, WAVE - 1, WAVE);
    int first_all = first < 0 ? INT32_MAX : t;  // lane of the first valid model
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        pairs += __shfl_xor(pairs, off, WAVE);
        eat += __shfl_xor(eat, off, WAVE);
        max_k = max(max_k, __shfl_xor(max_k, off, WAVE));
        max_pairs = max(max_pairs, __shfl_xor(max_pairs, off, WAVE));
        n_ok += __shfl_xor(n_ok, off, WAVE);
        n_inv += __shfl_xor(n_inv, off, WAVE);
        n_cap += __shfl_xor(n_cap, off, WAVE);
        dyn |= __shfl_xor(dyn, off, WAVE);
        mixed |= __shfl_xor(mixed, off, WAVE);
        first_all = min(first_all, __shfl_xor(first_all, off, WAVE));
    }
    const int flags_first = first_all == INT32_MAX ? -1 : __shfl(first, first_all, WAVE);
    mixed |= __any(first >= 0 && first != flags_first) ? 1 : 0;
    // a planned-ahead call also fixed the kernels of the later stages from the flags it expected (expect_flags >= 0): any
    // difference is treated like an exceeded capacity, so a call that must be repeated never leaves plausible numbers behind
    bool overflow = tot_rows > cap_rows || tot_cells > cap_cells || max_k > cap_k || max_pairs > cap_pairs;
    if (expect_flags >= 0 && n_ok > 0) overflow = overflow || mixed || flags_first != expect_flags || dyn != expect_dyn;
    if (t == 0) {
        VagDevPlan P;
        P.rows = tot_rows, P.cells = tot_cells, P.pairs = pairs, P.eat = eat, P.max_k = max_k, P.max_pairs = max_pairs;
        P.n_ok = n_ok, P.n_invalid = n_inv, P.n_capacity = n_cap, P.flags_first = flags_first, P.flags_mixed = mixed;
        P.dyn_class = dyn, P.overflow = overflow ? 1 : 0;
        P.seq = seq, P.pad_ = 0;
        *plan = P;
        // the host's copy: written straight into pinned host memory while the later stages are still queued, summary first,
        // sequence number last -- the host spins on the number instead of paying a copy + stream synchronisation
        VagDevPlan H = P;
        H.seq = host_plan->seq;  // not yet
        *host_plan = H;
        __threadfence_system();
        __hip_atomic_store(&host_plan->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();  // push the number itself out of the L2: nothing else in this kernel would
    }
    int r = r_inc - rows, b = b_inc - blks;  // exclusive prefixes of this lane's chunk
    long long c = c_inc - cells;
    int* __restrict__ blk_off = row_off + nb + 1;
    for (int mb = m0; mb < m1; mb += SCAN_U) {
        MetaHead H[SCAN_U];
        load_heads(mb, H);
#pragma unroll
        for (int u = 0; u < SCAN_U; ++u) {
            const int m = mb + u;
            if (m >= m1) break;
            const MetaHead& M = H[u];
            row_off[m] = overflow ? 0 : r;
            blk_off[m] = overflow ? 0 : b;
            cell_off[m] = overflow ? 0 : c;
            if (M.status == 0) {
                r += M.n_reps;
                b += (M.n_theta * M.n_phi_eff + 63) >> 6;
                c += (long long)M.n_reps * M.n_t;
                if (overflow) meta[m].status = VAG_E_CAPACITY;
            }
        }
    }
    if (t == 0) {
        row_off[nb] = overflow ? 0 : tot_rows;
        blk_off[nb] = overflow ? 0 : tot_blks;
        cell_off[nb] = overflow ? 0 : tot_cells;
    }
}

// First regular lattice node and early node of the row (theta, phi) of a structured (spreading) grid: TimeScanResult::t_start /
// early_t (grid-refinement.h:462-469,497-498).  t_min = 0.99-scaled first requested time is applied here; cut = the row's start
// cut-off (min(0.01 t_dec, 0.01 s) [, 0.01 T0]).  Used by the grid kernel for the phi = phi[0] slice it stores per theta row and by
// the dynamics / geometry kernels for the (phi, theta) pair rows of Model(axisymmetric=False).
VAG_DEV void row_time_start(double beta0, double cos_th, double sin_th, double cos_phi, double cos_tv, double sin_tv, double t_min, double z,
                            double cut, double& t_start, double& t_early, double& ts_raw) {
    const double cos_a = cos_th * cos_tv + sin_th * sin_tv * cos_phi;
    const double ts = 0.99 * t_min * (1 - beta0) / (1 - cos_a * beta0) / (1 + z);
    ts_raw = ts;
    t_start = dmax(ts, cut);
    t_early = 0.99 * dmin(ts, cut);
}

// The adaptive grid of model m = blockIdx.x, by one wavefront.
// tminmax[0..1]: min / max of the requested observer times [s] (device memory).
template <class SH>
VAG_DEV void
grid_model(SH& sh, const vag_model_params* __restrict__ params, int nb, const double* __restrict__ tminmax,
                VagGridMeta* __restrict__ meta, double* __restrict__ g_phi, double* __restrict__ g_theta,
                int* __restrict__ g_rep_of, int* __restrict__ g_rep_start, double* __restrict__ g_tdec,
                double* __restrict__ g_geo_th /* [nb][3][VAG_MAX_THETA]: cos, sin, log2|dcos| */,
                double* __restrict__ g_geo_ph /* [nb][2][VAG_MAX_PHI]: cos(phi), log2(dphi) */,
                int* __restrict__ fail /* [16] ODE-row failure counters, work tallies and row queue of the dynamics stage, reset here */,
                double* __restrict__ g_rowgeo /* [nb][VAG_ROWGEO_HDR + 2 SH::max_phi + 4 SH::max_theta]: the same geometry as records */) {
    const int m = blockIdx.x;
    const int lane = threadIdx.x;
    if (m == 0 && lane < 16) fail[lane] = 0;  // (ODE row tallies and the refill kernel's row queue, vag_capi.hip: d_fail)
    const vag_model_params P = params[m];
    const double t_min_s = tminmax[0], t_max_s = tminmax[1];
    if (!params_valid(P)) {  // the reference raises ValueError; batched walkers get status != 0 (-> NaN / -inf)
        if (lane == 0) {
            VagGridMeta bad = {};
            bad.status = VAG_E_INVALID;
            meta[m] = bad;
        }
        return;
    }
    Jet jet;
    Medium med;
    jet_init(jet, P);
    medium_init(med, P);
    const double theta_v = P.theta_obs, z = P.z;
    VagGridMeta M;
    M.status = 0;
    M.flags = P.flags;
    M.t_num_base = 0;
    M.dyn_class = (med.generic || (P.flags & (VAG_FLAG_SPREADING | VAG_FLAG_MAGNETAR | VAG_FLAG_RVS))) ? 1 : 0;
    M.rep_phi_stride = 0;
    M.th_stride = SH::max_theta;
    M.ph_stride = SH::max_phi;
#ifdef VAG_GRID_STAMPS
    long long stamps_[10] = {};
#endif
    VAG_GRID_STAMP(0);

    // ---- find_jet_jumps (grid-refinement.h:41-86): parallel profile scan, sequential jump logic ----
    const double th_lo = 1e-6, th_hi = C_PI / 2;
    double* jumps = sh.jumps;  // every lane writes the same value: wave-uniform list
    int n_jumps = 0;
    if (jet_Gamma0(jet, th_hi) >= GAMMA_CUT) {
        jumps[n_jumps++] = th_hi;
    } else {
        const double dth = (th_hi - th_lo) / (N_SCAN - 1);
        for (int s = lane; s < N_SCAN; s += WAVE) sh.scan_g[s] = jet_Gamma0(jet, th_lo + dth * (double)s);
        __syncthreads();
        // candidate intervals (s-1, s) are flagged in parallel; the rare flagged ones are bisected in scan order
        for (int base = 1; base < N_SCAN; base += WAVE) {
            const int s = base + lane;
            bool cand = false;
            double prev_G = 0, cur_G = 0;
            if (s < N_SCAN) {
                prev_G = sh.scan_g[s - 1];
                cur_G = sh.scan_g[s];
                if (prev_G >= GAMMA_CUT || cur_G >= GAMMA_CUT) {
                    const double dG = fabs(cur_G - prev_G);
                    const double scale = dmax(prev_G - 1, cur_G - 1);
                    cand = scale > 0 && dG > 0.5 * scale;
                }
            }
            unsigned long long mask = __ballot(cand);
            while (mask) {
                const int src = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const int sj = base + src;
                const double pG = sh.scan_g[sj - 1], cG = sh.scan_g[sj];
                double lo = th_lo + dth * (double)(sj - 1), hi = th_lo + dth * (double)sj;
                while (hi - lo > 1e-9) {
                    const double mid = 0.5 * (lo + hi);
                    const double Gm = jet_Gamma0(jet, mid);
                    if (fabs(Gm - pG) < fabs(Gm - cG))
                        lo = mid;
                    else
                        hi = mid;
                }
                if (n_jumps < VAG_MAX_JUMPS) jumps[n_jumps++] = pG > cG ? lo : hi;
            }
        }
        __syncthreads();
    }

    VAG_GRID_STAMP(1);
    // ---- find_theta_range (grid-refinement.h:89-111).  The abscissae come from a running
    //      subtraction/addition (kept sequential, it is only adds); the profile evaluations are parallel.
    double inner_edge = th_lo, outer_edge = th_hi;
    {
        const double step = (th_hi - th_lo) / N_SCAN;
        // The reference walks th -= step down from pi/2 and its rounding accumulates, so abscissa s is the result of s dependent
        // subtractions.  Every lane runs the chain once over the nine block starts (s = 64 b, static indices: registers), then
        // walks its own `lane` further steps on all nine chains side by side: bit-identical abscissae, ~7 k cycles instead of
        // the 40 k of a single lane filling LDS.
        constexpr int NBLK = (N_SCAN + 8 + WAVE - 1) / WAVE;  // 9 blocks of 64 candidates
        double mine[NBLK];
        {
            double th = th_hi;
#pragma unroll
            for (int b = 0; b < NBLK; ++b) {
                mine[b] = th;
#pragma unroll 8
                for (int u = 0; u < WAVE; ++u) th -= step;
            }
            for (int u = 0; u < lane; ++u) {
#pragma unroll
                for (int b = 0; b < NBLK; ++b) mine[b] -= step;
            }
        }
        // first abscissa (descending) inside [th_lo, pi/2] with Gamma0 >= cut: blocks in order, lanes in parallel
        int first = 1 << 30;
#pragma unroll
        for (int b = 0; b < NBLK; ++b) {
            if (first == (1 << 30)) {  // wave-uniform
                const bool hit = mine[b] >= th_lo && jet_Gamma0(jet, mine[b]) >= GAMMA_CUT;
                const unsigned long long mask = __ballot(hit);
                if (mask) {
                    const int src = __ffsll((long long)mask) - 1;
                    first = b * WAVE + src;
                    outer_edge = lane_value(mine[b], src);
                }
            }
        }
        // upward scan: the first abscissa almost always qualifies; stay sequential with early exit
        for (double th = th_lo; th <= th_hi; th += step) {
            if (jet_Gamma0(jet, th) >= GAMMA_CUT) {
                inner_edge = th;
                break;
            }
        }
    }
    for (int i = 0; i < n_jumps; ++i) outer_edge = dmax(outer_edge, jumps[i]);
    const double theta_min = dmax(1e-6, inner_edge);
    const double theta_max = dmin(outer_edge, C_PI / 2);
    const size_t base_pts = 36 + (size_t)((theta_max - theta_min) * 180 / C_PI * P.theta_resol);

    VAG_GRID_STAMP(2);
    // ---- adaptive_theta_grid (grid-refinement.h:200-291) ----
    int n_base = 0;
    {
        constexpr int scan_pts = 100;
        const double extent = theta_max - theta_min;
        // 101 scan points, two per lane (i = lane, lane + 64).  The reference's sequential pass -- running sum, running peak
        // (strict >, so the first maximum wins) and last_bright = last i whose weight exceeds 1 % of the peak SO FAR -- becomes a
        // wave prefix-maximum: last_bright = max{i : w_i > 0.01 max_{j<i} w_j} (a new peak satisfies that too).
        double Gamma_v = 1.0;
        double wv[2], gv[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = lane + q * WAVE;
            wv[q] = -1.0;
            gv[q] = 1.0;
            if (i <= scan_pts) {
                const double theta = theta_min + extent * i / scan_pts;
                const double G = jet_Gamma0(jet, theta);
                gv[q] = G;
                wv[q] = structure_weight(G);
                const double d = theta - theta_v;
                Gamma_v = dmax(Gamma_v, G / sqrt(1.0 + G * G * d * d));
            }
        }
        Gamma_v = -wave_min(-Gamma_v);  // a maximum: order-free
        double struct_sum = 0;
        {   // the sum in scan order: lanes 0..63 hold i = 0..63, then 64..100
            double acc = 0;
            for (int src = 0; src < WAVE; ++src) acc += lane_value(dmax(wv[0], 0.0), src);
            for (int src = 0; src <= scan_pts - WAVE; ++src) acc += lane_value(dmax(wv[1], 0.0), src);
            struct_sum = acc;
        }
        // exclusive prefix maximum over i (weights are >= 0; -1 marks "no point")
        double pm0 = wv[0];  // inclusive scan of the first 64
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            const double o = __shfl_up(pm0, off, WAVE);
            if (lane >= off) pm0 = dmax(pm0, o);
        }
        const double tot0 = lane_value(pm0, WAVE - 1);
        double ex0 = __shfl_up(pm0, 1, WAVE);
        if (lane == 0) ex0 = 0.0;
        double pm1 = wv[1];
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            const double o = __shfl_up(pm1, off, WAVE);
            if (lane >= off) pm1 = dmax(pm1, o);
        }
        double ex1 = __shfl_up(pm1, 1, WAVE);
        ex1 = lane == 0 ? tot0 : dmax(ex1, tot0);
        const double peak_weight = dmax(dmax(tot0, lane_value(pm1, WAVE - 1)), 0.0);
        int lb = -1;
        if (wv[0] > 0.01 * dmax(ex0, 0.0) && wv[0] >= 0) lb = lane;
        if (lane + WAVE <= scan_pts && wv[1] > 0.01 * dmax(ex1, 0.0)) lb = lane + WAVE;
        const int last_bright = max(0, -wave_min_int(-lb));
        // Gamma at the FIRST maximum of the weight
        int arg = 1 << 30;
        if (wv[0] == peak_weight) arg = lane;
        else if (lane + WAVE <= scan_pts && wv[1] == peak_weight) arg = lane + WAVE;
        arg = wave_min_int(arg);
        double Gamma_peak = 1.0;
        if (peak_weight > 0) Gamma_peak = arg < WAVE ? lane_value(gv[0], arg) : lane_value(gv[1], arg - WAVE);
        const double floor_weight = 0.25 * peak_weight;
        const double CDF_est = (struct_sum / scan_pts + floor_weight) * extent;
        const double theta_bright = theta_min + extent * last_bright / scan_pts;
        Gamma_peak = dmax(Gamma_peak, Gamma_v);
        const double doppler_alpha = 12.0 * sqrt(peak_weight / dmax(structure_weight(Gamma_v), 1.0));
        const double Gp2 = Gamma_peak * Gamma_peak, Gv2 = Gamma_v * Gamma_v;
        auto beam_pts = [&](double log_decades, double coeff, double offset) -> size_t {
            return (size_t)(dmax(0.0, log_decades - offset) * P.theta_resol * coeff);
        };
        const size_t core_pts = beam_pts(log10(dmax(1.0, Gamma_peak * (theta_bright - theta_min))), 55.0, 1.0);
        const size_t view_pts =
            (theta_v * Gamma_peak > 3.0)
                ? beam_pts(log10(dmax(1.0, Gamma_v * dmax(theta_v - theta_min, theta_max - theta_v))), 25.0, 0.0)
                : 0;
        const size_t total_pts = base_pts + core_pts + view_pts;
        const double core_cdf = 0.5 * log((1.0 + Gp2 * theta_max * theta_max) / (1.0 + Gp2 * theta_min * theta_min));
        const double core_weight = (core_pts > 0 && core_cdf > 0) ? (double)core_pts / base_pts * CDF_est / core_cdf : 0.0;
        const double tl = theta_v - theta_min, tr = theta_max - theta_v;
        const double view_cdf = 0.5 * (log(1.0 + Gv2 * tl * tl) + log(1.0 + Gv2 * tr * tr));
        const double view_weight = (view_pts > 0 && view_cdf > 0) ? (double)view_pts / base_pts * CDF_est / view_cdf : 0.0;
        if (total_pts > SH::max_theta - 3 * VAG_MAX_JUMPS) {
            M.status = VAG_E_CAPACITY;
            if (lane == 0) meta[m] = M;
            return;
        }
        n_base = (int)total_pts;
        // sample abscissae: xt::logspace(log10(min), log10(max), 200)
        {
            const double a = log10(theta_min), b = log10(theta_max);
            for (int k = lane; k < N_SAMPLES; k += WAVE)  // 10^x through the fast exp2 (3e-16)
                sh.xs[k] = exp2_fast(linspace_at(a, b, N_SAMPLES, k) * 3.321928094887362347870319429489390175865);
        }
        // the integrand on the fast scalar kernels (exp2 / log2 / rsq / rcp forms, 1e-16 apart from the library calls)
        auto pdf = [&](double theta) -> double {
            const double G = jet_Gamma0_fast(jet, theta);
            const double beta = beta_fast(G);
            const double d = theta - theta_v;
            const double doppler = (1 - beta) * rcp_fast(1 - beta * cos_small(d));
            const double structure = structure_weight_fast(G);
            return core_weight * Gp2 * theta * rcp_fast(1.0 + Gp2 * theta * theta) +
                   view_weight * Gv2 * fabs(d) * rcp_fast(1.0 + Gv2 * d * d) + (1 + doppler_alpha * doppler) * structure +
                   floor_weight;
        };
        // the five stage abscissae of a step on lanes 0..4 (the pdf is scalar code: every lane may take its own theta)
        auto stages = [&](const double* tx, double* kv) {
            const double v = pdf(lane == 0 ? tx[0] : lane == 1 ? tx[1] : lane == 2 ? tx[2] : lane == 3 ? tx[3] : tx[4]);
#pragma unroll
            for (int i = 0; i < 5; ++i) kv[i] = lane_value(v, i);
        };
#ifdef VAG_GRID_STAMPS
        const long long s_a = __builtin_readcyclecounter();
        if (m == 0 && lane == 0) g_stage_cycles = g_dense_cycles = 0, g_quad_steps = 0;
#endif
        integrate_cdf(sh, pdf(theta_min), stages, theta_min, theta_max);
#ifdef VAG_GRID_STAMPS
        const long long s_b = __builtin_readcyclecounter();
#endif
        invert_cdf(sh, n_base, false, sh.base);
#ifdef VAG_GRID_STAMPS
        if (m == 0 && lane == 0)
            printf("  theta: preamble %lld integrate %lld (steps %d: stage evaluations %lld, dense output %lld) invert %lld (n_base %d)\n",
                   s_a - stamps_[2], s_b - s_a, g_quad_steps, g_stage_cycles, g_dense_cycles,
                   (long long)__builtin_readcyclecounter() - s_b, n_base);
#endif
    }

    VAG_GRID_STAMP(3);
    // ---- jump_refinement_grid (grid-refinement.cpp:136-160) + merge_grids (grid-refinement.h:362-393) ----
    int n_theta = 0;
    {
        double* feat = sh.feat;  // wave-uniform list in LDS (every lane writes the same values)
        int nf = 0;
        const double tight = ((theta_max - theta_min) / n_base) / 8;
        for (int q = 0; q < n_jumps; ++q) {
            const double jt = jumps[q];
            if (jt >= C_PI / 2 - 0.01) continue;
            if (jt - tight >= theta_min) feat[nf++] = jt - tight;
            if (jt + tight <= theta_max) feat[nf++] = jt + tight;
            if (jt >= theta_min && jt <= theta_max) feat[nf++] = jt;
        }
        for (int a = 1; a < nf; ++a) {  // insertion sort (tiny)
            const double v = feat[a];
            int b = a - 1;
            while (b >= 0 && feat[b] > v) {
                feat[b + 1] = feat[b];
                --b;
            }
            feat[b + 1] = v;
        }
        int nu = 0;
        for (int a = 0; a < nf; ++a)
            if (nu == 0 || feat[nu - 1] != feat[a]) feat[nu++] = feat[a];
        nf = nu;
        __syncthreads();
        // merge_grids (grid-refinement.h:362-393): sorted union without repeats.  Both lists ascend (the base grid strictly:
        // its CDF rises strictly), so an element's slot in the union is its own index plus the number of elements of the other
        // list before it; a feature equal to a base node is dropped.  One element per lane and round.
        n_theta = 0;
        {
            int n_dup = 0;
            for (int j = lane; j < nf; j += WAVE) {  // features: rank among the base nodes (binary search), skipped when equal to one
                const double v = feat[j];
                int lo = 0, hi = n_base;  // first base index with base[idx] >= v
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (sh.base[mid] < v)
                        lo = mid + 1;
                    else
                        hi = mid;
                }
                const bool dup = lo < n_base && sh.base[lo] == v;
                sh.flag[j] = dup ? -1 : lo;  // base nodes before this feature
                n_dup += dup ? 1 : 0;
            }
            __syncthreads();
            for (int off = 32; off > 0; off >>= 1) n_dup += __shfl_xor(n_dup, off, WAVE);
            n_theta = n_base + nf - n_dup;
            for (int i = lane; i < n_base; i += WAVE) {  // base nodes: shifted by the kept features strictly below them
                const double v = sh.base[i];
                int before = 0;
                for (int j = 0; j < nf; ++j) before += (sh.flag[j] >= 0 && feat[j] < v) ? 1 : 0;
                sh.theta[i + before] = v;
            }
            for (int j = lane; j < nf; j += WAVE) {
                if (sh.flag[j] < 0) continue;
                int kept_before = 0;
                for (int q = 0; q < j; ++q) kept_before += sh.flag[q] >= 0 ? 1 : 0;
                sh.theta[sh.flag[j] + kept_before] = feat[j];
            }
        }
        __syncthreads();
    }

    VAG_GRID_STAMP(4);
    // ---- phi grid (grid-refinement.h:664-695, adaptive_phi_grid 296-360) ----
    int n_phi = 0, phi_mirrored = 0;
    {
        size_t phi_base = (size_t)(360 * P.phi_resol);
        if (phi_base < 1) phi_base = 1;
        const bool axisym = !(P.flags & VAG_FLAG_NON_AXISYMMETRIC);  // Model(axisymmetric=...)
        const bool mirror = axisym && theta_v != 0 && phi_base > 4;
        size_t phi_num;
        double phi_max, boost_cap;
        if (mirror) {
            phi_num = (phi_base + 1) / 2;
            phi_max = C_PI;
            boost_cap = 5.0;
            phi_mirrored = 1;
        } else {
            const double sharp = jet_Gamma0(jet, theta_v) * sin(theta_v);
            const double boost = sqrt(dmax(sharp / (2 * C_PI), 1.0));
            phi_num = (size_t)(phi_base * boost);
            if (phi_num < 1) phi_num = 1;
            if (phi_num > phi_base * 5) phi_num = phi_base * 5;
            phi_max = 2 * C_PI;
            boost_cap = 0;
        }
        const bool uniform = (!mirror && phi_num <= 2) || (theta_v == 0 && axisym);
        if (uniform) {
            if (phi_num > SH::max_phi) {
                M.status = VAG_E_CAPACITY;
                if (lane == 0) meta[m] = M;
                return;
            }
            n_phi = (int)phi_num;
            for (int i = lane; i < n_phi; i += WAVE) sh.phi[i] = linspace_at(0., 2 * C_PI, n_phi, i);
            __syncthreads();
        } else {
            const bool half_range = phi_max < 2 * C_PI;
            const double cos_tv = cos(theta_v), sin_tv = sin(theta_v);
            for (int j = lane; j < n_theta; j += WAVE) {
                const double th = sh.theta[j];
                const double left = (j == 0) ? 0.0 : 0.5 * (sh.theta[j - 1] + th);
                const double right = (j == n_theta - 1) ? th : 0.5 * (th + sh.theta[j + 1]);
                const double G = jet_Gamma0(jet, th);
                sh.pj_dcos[j] = fabs(cos(left) - cos(right));
                sh.pj_beta[j] = gamma_to_beta(G);
                sh.pj_sw[j] = structure_weight(G);
                sh.pj_ct[j] = cos(th) * cos_tv;
                sh.pj_st[j] = sin(th) * sin_tv;
            }
            __syncthreads();
            // per-bin term a_j sw_j dcos_j with a_j = (1 - beta_j) / (1 - beta_j cos_alpha_j): the reciprocal through rcp_fast
            // (two Newton steps, <= 2 ulp) instead of the IEEE division sequence, cos(phi) through cos_0_2pi
            auto bin_term = [&](int j, double cos_phi) -> double {
                const double beta = sh.pj_beta[j];
                const double cos_alpha = sh.pj_ct[j] + sh.pj_st[j] * cos_phi;
                const double a = (1 - beta) * rcp_fast(1 - beta * cos_alpha);
                return a * sh.pj_sw[j] * sh.pj_dcos[j];
            };
            auto phi_weight = [&](double phi) -> double {
                const double cos_phi = cos_0_2pi(phi);
                double w = 0;
                for (int j = lane; j < n_theta; j += WAVE) w += bin_term(j, cos_phi);
                return wave_sum(w);
            };
            constexpr int scan_pts = 100;
            double peak = 0, sum = 0;
            for (int s = lane; s <= scan_pts; s += WAVE) {  // one scan point per lane, theta bins summed in grid order
                const double cos_phi = cos_0_2pi(phi_max * (double)s / scan_pts);
                double w = 0;
                for (int j = 0; j < n_theta; ++j) w += bin_term(j, cos_phi);
                peak = dmax(peak, w);
                sum += w;
            }
            peak = -wave_min(-peak);
            sum = wave_sum(sum);
            const double floor_w = 0.05 * peak;
            if (boost_cap > 0 && peak > 0) {
                const double mean_pdf = sum / (scan_pts + 1) + floor_w;
                const double conc = (peak + floor_w) / mean_pdf;
                double boost = conc / 5;
                boost = boost < 1.0 ? 1.0 : (boost_cap < boost ? boost_cap : boost);
                phi_num = (size_t)((double)phi_num * boost);
            }
            if (phi_num > SH::max_phi) {
                M.status = VAG_E_CAPACITY;
                if (lane == 0) meta[m] = M;
                return;
            }
            n_phi = (int)phi_num;
            for (int k = lane; k < N_SAMPLES; k += WAVE) sh.xs[k] = linspace_at(0, phi_max, N_SAMPLES, k);
            // five stage abscissae at once: their cosines on lanes 0..4, then every lane adds its theta bins to five
            // running sums (same per-term arithmetic and the same wave reduction as phi_weight)
            auto stages = [&](const double* tx, double* kv) {
                const double cv = cos_0_2pi(lane == 0 ? tx[0] : lane == 1 ? tx[1] : lane == 2 ? tx[2] : lane == 3 ? tx[3] : tx[4]);
                double cp[5], w[5];
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    cp[i] = lane_value(cv, i);
                    w[i] = 0;
                }
                for (int j = lane; j < n_theta; j += WAVE) {
                    const double beta = sh.pj_beta[j], ct = sh.pj_ct[j], st = sh.pj_st[j], sw = sh.pj_sw[j], dc = sh.pj_dcos[j];
                    const double omb = 1 - beta;
#pragma unroll
                    for (int i = 0; i < 5; ++i) {
                        const double cos_alpha = ct + st * cp[i];
                        const double a = omb * rcp_fast(1 - beta * cos_alpha);
                        w[i] += a * sw * dc;
                    }
                }
#pragma unroll
                for (int i = 0; i < 5; ++i) kv[i] = wave_sum(w[i]) + floor_w;
            };
            integrate_cdf(sh, phi_weight(0.0) + floor_w, stages, 0, phi_max);
            invert_cdf(sh, n_phi, half_range, sh.phi);
        }
        if (!mirror && phi_num >= 2) {
            const double shift = 0.5 * (sh.phi[1] - sh.phi[0]);
            __syncthreads();
            for (int i = lane; i < n_phi; i += WAVE) sh.phi[i] += shift;
            __syncthreads();
        }
    }

    VAG_GRID_STAMP(5);
    // ---- Coord::detect_symmetry (src/core/mesh.h:121-187): contiguous groups of identical rows; a spreading jet
    //      evolves every row on its own (Symmetry::structured) ----
    const bool spreading = (P.flags & VAG_FLAG_SPREADING) != 0;
    int n_reps = 0;
    {
        for (int j = lane; j < n_theta; j += WAVE) {
            int differs = 1;
            if (j > 0 && !spreading) {
                const double a = sh.theta[j - 1], b = sh.theta[j];
                differs = (jet_eps_k(jet, a) != jet_eps_k(jet, b)) || (jet_Gamma0(jet, a) != jet_Gamma0(jet, b));
            }
            sh.flag[j] = differs;
        }
        __syncthreads();
        int* rep_of = g_rep_of + (size_t)m * SH::max_theta;
        int* rep_start = g_rep_start + (size_t)m * SH::max_theta;
        // group index of row j = (number of group starts in [0, j]) - 1: ballot prefix counts, 64 rows per round
        for (int base = 0; base < n_theta; base += WAVE) {
            const int j = base + lane;
            const bool start = j < n_theta && sh.flag[j] != 0;
            const unsigned long long mask = __ballot(start);
            const int before = __popcll(mask & ((1ull << lane) - 1ull));  // starts among the lower lanes of this round
            if (j < n_theta) {
                const int g = n_reps + before + (start ? 1 : 0) - 1;
                rep_of[j] = g;
                if (start) rep_start[g] = j;
            }
            n_reps += __popcll(mask);
        }
        __syncthreads();
    }
    const int symmetry = spreading ? VAG_SYM_STRUCTURED
                                   : (n_reps == 1 ? VAG_SYM_ISOTROPIC : (n_reps < n_theta ? VAG_SYM_PIECEWISE : VAG_SYM_PHI_SYMMETRIC));

    VAG_GRID_STAMP(6);
    // ---- build_time_grid scalars (grid-refinement.h:472-528,594-636); is_rvs = Model(rvs_rad=...) ----
    {
        const bool is_rvs = (P.flags & VAG_FLAG_RVS) != 0;
        const double T0 = P.duration * U_SEC;
        double max_ref = 0;
        const double t_min = t_min_s * U_SEC, t_max = t_max_s * U_SEC;
        const double t_end = 1.01 * t_max / (1 + z);
        const double cos_tv = cos(theta_v), sin_tv = sin(theta_v);
        const double cos_phi0 = cos(sh.phi[0]);
        double min_raw = t_end, min_guarded = t_end, min_cut = t_end;
        for (int j = lane; j < n_theta; j += WAVE) {
            const double th = sh.theta[j];
            const double b = gamma_to_beta(jet_Gamma0(jet, th));
            const double td = estimate_t_dec(jet, med, th);
            sh.tdec[j] = td;
            double cut = dmin(0.01 * td, 1e-2 * U_SEC);
            if (is_rvs) {
                cut = dmin(cut, 0.01 * T0);
                max_ref = dmax(max_ref, 10.0 * dmax(td, T0));
            }
            // TimeScanResult::t_start / early_t (grid-refinement.h:462-469,497-498), used per row by structured (spreading)
            // grids; symmetric grids overwrite them below with the shared start / early node
            double ts, t_start_row, t_early_row;
            row_time_start(b, cos(th), sin(th), cos_phi0, cos_tv, sin_tv, t_min, z, cut, t_start_row, t_early_row, ts);
            g_tdec[((size_t)m * 3 + 1) * SH::max_theta + j] = t_start_row;
            g_tdec[((size_t)m * 3 + 2) * SH::max_theta + j] = t_early_row;
            min_raw = dmin(min_raw, ts);
            min_guarded = dmin(min_guarded, t_start_row);
            min_cut = dmin(min_cut, cut);
            if (P.flags & VAG_FLAG_NON_AXISYMMETRIC)  // phi_size = |phi|: the global bounds scan every phi node (:484-507)
                for (int i = 1; i < n_phi; ++i) {
                    double tsi, tsg, tse;
                    row_time_start(b, cos(th), sin(th), cos(sh.phi[i]), cos_tv, sin_tv, t_min, z, cut, tsg, tse, tsi);
                    min_raw = dmin(min_raw, tsi);
                    min_guarded = dmin(min_guarded, tsg);
                }
        }
        min_raw = wave_min(min_raw);
        min_guarded = wave_min(min_guarded);
        min_cut = wave_min(min_cut);
        max_ref = -wave_min(-max_ref);
        const int has_early = min_raw < min_cut;
        const size_t t_num_base = (size_t)(dmax(log10(t_end / min_guarded), 1.0) * P.t_resol);
        size_t t_num_extra = 0;  // compute_time_grid_size: pre-crossing lattice twice as dense
        if (is_rvs && max_ref > min_guarded)
            t_num_extra = (size_t)((2.0 - 1.0) * log10(dmin(max_ref, t_end) / min_guarded) * P.t_resol);
        const size_t t_num_tot = t_num_base + t_num_extra;
        const size_t t_num = t_num_tot + (has_early ? 1 : 0);
        if (t_num > VAG_MAX_TIME || t_num_tot < 2) M.status = VAG_E_CAPACITY;
        M.n_t = (int)t_num;
        M.t_num_tot = (int)t_num_tot;
        M.t_num_base = (int)t_num_base;
        M.has_early = has_early;
        M.t_early = min_raw;
        M.t_start = min_guarded;
        M.t_end = t_end;
    }
    // Model(axisymmetric=False) with a spreading jet: Symmetry::structured with phi_size = |phi| -- every (phi, theta) node has its own
    // lattice start (the viewing cosine enters it) and its own solve (grid-refinement.h:619-625, forward-shock.tpp:175-208)
    if (spreading && (P.flags & VAG_FLAG_NON_AXISYMMETRIC) && n_phi > 1) {
        M.rep_phi_stride = n_theta;
        n_reps = n_phi * n_theta;
    }
    M.n_phi = n_phi;
    M.n_theta = n_theta;
    M.n_reps = n_reps;
    M.symmetry = symmetry;
    M.phi_mirrored = phi_mirrored;
    // Observer::build_time_grid, observer.cpp:215-222 (jet_3d = non-axisymmetric model with more than one phi node)
    M.n_phi_eff = (theta_v == 0 && !((P.flags & VAG_FLAG_NON_AXISYMMETRIC) && n_phi > 1)) ? 1 : n_phi;
    __syncthreads();
    for (int i = lane; i < n_phi; i += WAVE) g_phi[(size_t)m * SH::max_phi + i] = sh.phi[i];
    for (int j = lane; j < n_theta; j += WAVE) {
        g_theta[(size_t)m * SH::max_theta + j] = sh.theta[j];
        // lattice scalars per row: [0] t_dec, [1] first regular node, [2] early node.  Symmetric grids share the global
        // start / early point (build_time_grid, grid-refinement.h:609-626)
        g_tdec[((size_t)m * 3 + 0) * SH::max_theta + j] = sh.tdec[j];
        if (!spreading) {
            g_tdec[((size_t)m * 3 + 1) * SH::max_theta + j] = M.t_start;
            g_tdec[((size_t)m * 3 + 2) * SH::max_theta + j] = M.t_early;
        }
    }
    VAG_GRID_STAMP(7);
    // Geometry factors of the equal-arrival-time step that depend on the angular grid only
    // (calc_eat_non_spreading + compute_dphi, src/core/observer.cpp:17-37,143-188): computed once per model
    // here instead of once per (theta, phi) row in the flux kernel.
    M.cos_obs = cos(theta_v);
    M.sin_obs = sin(theta_v);
    {
        double* gth = g_geo_th + (size_t)m * 3 * SH::max_theta;
        double* gph = g_geo_ph + (size_t)m * 2 * SH::max_phi;
        // The same numbers once more as ROW-GEOMETRY RECORDS behind one base address per model, for the workgroup flux kernel's
        // scalar loads (it is out of scalar registers for five plane pointers, the rep_of pointer and the observer constants):
        // header {cos theta_obs, sin theta_obs, byte offset of the theta records, -}, phi records {cos phi, log2 dphi} from
        // double VAG_ROWGEO_HDR on, theta records {cos theta, sin theta, log2 |dcos theta|, representative row (int)} behind them.
        double* rg = g_rowgeo + (size_t)m * (VAG_ROWGEO_HDR + 2 * SH::max_phi + 4 * SH::max_theta);
        const int npe = M.n_phi_eff;
        const int rg_th = VAG_ROWGEO_HDR + 2 * npe;
        const int* rep_of = g_rep_of + (size_t)m * SH::max_theta;  // written above, a barrier ago
        if (lane == 0) {
            rg[0] = M.cos_obs;
            rg[1] = M.sin_obs;
            rg[2] = __hiloint2double(0, rg_th * 8);
            rg[3] = 0;
        }
        const int last = n_theta - 1;
        for (int j = lane; j < n_theta; j += WAVE) {
            const double th = sh.theta[j];
            const double ct = cos(th);
            const double cos_lo = (j == 0) ? ct : cos(0.5 * (sh.theta[j - 1] + th));
            const double cos_hi = (j == last) ? ct : cos(0.5 * (th + sh.theta[j + 1]));
            const double st = sin(th), ld = log2(fabs(cos_hi - cos_lo));
            gth[j] = ct;
            gth[SH::max_theta + j] = st;
            gth[2 * SH::max_theta + j] = ld;
            double* r = rg + rg_th + 4 * j;
            r[0] = ct, r[1] = st, r[2] = ld, r[3] = __hiloint2double(0, rep_of[j]);
        }
        for (int i = lane; i < npe; i += WAVE) {
            double dphi;
            if (npe == 1) {
                dphi = 2 * C_PI;
            } else if (phi_mirrored) {
                const double left = (i > 0) ? 0.5 * (sh.phi[i - 1] + sh.phi[i]) : 0.0;
                const double right = (i < npe - 1) ? 0.5 * (sh.phi[i] + sh.phi[i + 1]) : C_PI;
                dphi = 2 * (right - left);
            } else {
                dphi = 0.5 * (sh.phi[min(i + 1, npe - 1)] - sh.phi[i > 0 ? i - 1 : 0]);
            }
            const double cp = cos(sh.phi[i]), ldp = log2(fabs(dphi));
            gph[i] = cp;
            gph[SH::max_phi + i] = ldp;
            rg[VAG_ROWGEO_HDR + 2 * i] = cp;
            rg[VAG_ROWGEO_HDR + 2 * i + 1] = ldp;
        }
    }
    M.lg2_1pz = log2(1 + z);
    if (lane == 0) meta[m] = M;
    VAG_GRID_STAMP(8);
#ifdef VAG_GRID_STAMPS
    if (m == 0 && lane == 0)
        printf("grid cycles: jumps %lld  theta_range %lld  theta_cdf %lld  merge %lld  phi %lld  symmetry %lld  time %lld  geometry %lld  total %lld\n",
               stamps_[1] - stamps_[0], stamps_[2] - stamps_[1], stamps_[3] - stamps_[2], stamps_[4] - stamps_[3],
               stamps_[5] - stamps_[4], stamps_[6] - stamps_[5], stamps_[7] - stamps_[6], stamps_[8] - stamps_[7],
               stamps_[8] - stamps_[0]);
#endif
}

// One wavefront (blockDim.x == 64) per model; the last wavefront to finish also lays the batch out (plan_scan_wave): no
// separate launch, no host round trip between the grids and the stages that depend on their sizes.
// Two wavefronts per SIMD (second launch bound): left alone the compiler allocates 208 VGPRs + 49 AGPRs = 257 registers -- ONE more than
// two resident wavefronts allow -- and a batch larger than the chip's 1024 SIMDs then runs its models one per SIMD, round after round
// (4096 top-hat models: 0.52 ms of grid stage at 0.87 resident wavefronts per SIMD, profiles/debug/grid_occupancy.sh).
// LEVEL 0 / 1: the small / large scratch layout in LDS.  LEVEL 2 (round 6): the same wavefront program with its scratch arrays in HBM
// (`scratch`, one GridSharedT<VAG_HUGE_THETA, VAG_HUGE_PHI> per model) -- for a batch in which some model outgrew the large layout; the
// barriers between the program's phases order global memory within the workgroup as they order LDS.
using GridSharedHuge = GridSharedT<VAG_HUGE_THETA, VAG_HUGE_PHI>;
template <int LEVEL>
__global__ void __launch_bounds__(WAVE, LEVEL == 0 ? 2 : 1)
vag_grid_kernel(const vag_model_params* __restrict__ params, int nb, const double* __restrict__ tminmax,
                VagGridMeta* meta, double* __restrict__ g_phi, double* __restrict__ g_theta,
                int* __restrict__ g_rep_of, int* __restrict__ g_rep_start, double* __restrict__ g_tdec,
                double* __restrict__ g_geo_th, double* __restrict__ g_geo_ph, int* __restrict__ fail,
                int* __restrict__ done_counter /* zero between launches */, int* __restrict__ row_off,
                long long* __restrict__ cell_off, VagDevPlan* __restrict__ plan, VagDevPlan* host_plan, int seq, int cap_rows,
                long long cap_cells, int cap_k, int cap_pairs, int expect_flags, int expect_dyn, float* __restrict__ cost,
                double* __restrict__ g_rowgeo, GridSharedHuge* scratch /* LEVEL 2: [nb] */) {
    if constexpr (LEVEL == 2) {
        if ((int)blockIdx.x < nb)
            grid_model<GridSharedHuge>(scratch[blockIdx.x], params, nb, tminmax, meta, g_phi, g_theta, g_rep_of, g_rep_start, g_tdec, g_geo_th, g_geo_ph,
                                       fail, g_rowgeo);
    } else {
        using Shared = typename std::conditional<LEVEL == 1, GridSharedT<VAG_MAX_THETA, VAG_MAX_PHI>, GridSharedT<VAG_GRID_THETA, VAG_GRID_PHI>>::type;
        __shared__ Shared sh;  // declared here, not in grid_model: LDS of a device function is charged to every kernel of the module
        if ((int)blockIdx.x < nb)
            grid_model<Shared>(sh, params, nb, tminmax, meta, g_phi, g_theta, g_rep_of, g_rep_start, g_tdec, g_geo_th, g_geo_ph, fail, g_rowgeo);
    }
    __shared__ int s_last;
    __threadfence();  // this model's results are visible device-wide before the ticket is taken
    __syncthreads();
    if (threadIdx.x == 0) s_last = (atomicAdd(done_counter, 1) == (int)gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    if (threadIdx.x == 0) *done_counter = 0, done_counter[4] = 0;  // (+4: the work-item counter of the persistent flux launches -- a call that
                                                                    //  died half-way must not leave its count to the next one)
    plan_scan_wave(meta, nb, row_off, cell_off, plan, host_plan, seq, cap_rows, cap_cells, cap_k, cap_pairs, expect_flags, expect_dyn, cost);
}

}  // namespace vag
