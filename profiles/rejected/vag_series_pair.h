// NOT PART OF THE BUILD -- a measured-and-rejected experiment, kept for the record ("measured and rejected").
// To try it again: copy next to vag_kernels.h, include it from vag_capi.hip and launch vag_flux_series_pair_kernel<FLUX_SYN> from
// run_flux_series with series_pair_region_doubles() for the LDS size.
// Result on MI355X (1024 C4 walkers, 787 k rows): 1.13 ms against 0.81 ms for the single-row kernel.  The dual evaluator does come
// out as one interleaved basic block (356 instructions for two evaluations), a lone wavefront needs 14 % fewer cycles per row, but
// 247 VGPRs and 28 K doubles of LDS per wavefront leave two wavefronts per SIMD instead of three, and the bracket phase diverges
// into the searching form whenever one point enters a row's time range.  Values agree with the single-row kernel to 7e-16.
//
// vag_series_pair.h -- the likelihood's flux pass: a fit's few bands, at most one data point per lane, TWO (theta, phi) rows per
// wavefront iteration.
//
// vag_flux_series_kernel (vag_kernels.h) gives a (theta, phi) row to a wavefront, and a row is four short phases that each
// wait on the one before: EAT logs of the lattice nodes, bracket search per point, boundary spectra per (band, node of the
// points' window), interpolation per point.  LDS holds three such wavefronts per SIMD and each of them issues an
// instruction every ~13 cycles (the chain, not the VALU, is the limit: profiles/series_probe.py with -DVAG_SERIES_STAMPS),
// so the VALU idles half the time.  Here the wavefront takes rows p and p+1 together -- they share the staged photon
// block -- and every phase is ONE basic block that carries both rows' dependency chains: branch-free evaluators
// (`sp_flat`, `log2_I_nu_rows`), a windowed bracket look-up whose common case has no loop, and selects instead of guarded
// updates.  The arithmetic of each value is that of the single-row kernel to the last bit (same expressions in the same
// order; a skipped shortcut computes the same number the shortcut returns), and so is the summation order, so the two
// kernels are interchangeable: tests/test_gpu_parity.py compares them bitwise.
//
// Reference: Observer::specific_flux_series (src/core/observer.h:447-538) over the rows of calc_eat_non_spreading
// (src/core/observer.cpp:143-205), SmoothPowerLawSyn::compute_log2_I_nu (src/radiation/smooth-power-law-syn.cpp:15-46).
#pragma once
#include <type_traits>

#include "vag_kernels.h"

namespace vag {

// doubles of LDS one wavefront of the pair kernel owns: the staged photon block, and per row slot the lattice times, Doppler
// logs and boundary values of every band
__host__ __device__ inline int series_pair_region_doubles(int ks, int n_bands) { return ((VAG_NPAR + 2 * (2 + n_bands)) * ks + 1) & ~1; }

// log2_softplus as sp_fast computes it, without the |z| > 20 branch: the shortcut's value max(z, 0) is selected at the end
VAG_DEV double sp_flat(double z, LdsTab tab) {
    const double a = fabs(z);
    const double t = fma(a, (double)SP_PER_UNIT, SP_MAGIC);
    const int idx = (int)min((unsigned)__double2loint(t), (unsigned)(SP_INTERVALS - 1));
    const double tau = fma(a, (double)SP_PER_UNIT, -(t - SP_MAGIC));
    const auto c2 = sp_row(tab, idx);
    const vdouble2 c01 = c2[0], c23 = c2[1], c45 = c2[2];
    double p = fma(c45.y, tau, c45.x);
    p = fma(p, tau, c23.y);
    p = fma(p, tau, c23.x);
    p = fma(p, tau, c01.y);
    p = fma(p, tau, c01.x);
    const double r = fma(0.5, z + a, p);
    return a > 20.0 ? fmax(z, 0.0) : r;
}

// log2_I_nu_fast (vag_device.h) for NR arguments on one cell, straight-line: the compiler interleaves the NR chains
template <int NR>
VAG_DEV void log2_I_nu_rows(const SpecRegs& c, const SpecConst& sc, const double (&x)[NR], LdsTab sp, double (&out)[NR]) {
    double thin[NR], lb[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const double d_lo = x[r] - c[VP_LG2_LO], d_hi = x[r] - c[VP_LG2_HI];
        thin[r] = d_lo * (1.0 / 3.0) - sp_flat(c[VP_DLO] * d_lo, sp) * c[VP_INV_SLO] - sp_flat(c[VP_DHI] * d_hi, sp) * c[VP_INV_SHI];
        const double lx = x[r] - c[VP_LG2_NUM];
        const bool far = lx > sc.log2_x_far;  // beyond it the softplus term of the optically thick branch is dropped
        const double lxc = far ? sc.log2_x_far : lx;
        const double s = -sc.smooth_thick * exp2_fast(2. / 3 * lxc);
        const double soft = sp_flat(-0.5 * lxc + s, sp);
        const double thick = far ? 2.5 * lx : 2.5 * lx + soft;
        lb[r] = thick + c[VP_TNORM];
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const double smooth_one = thin[r] - sp_flat(c[VP_SAB] * (thin[r] - lb[r]), sp) * c[VP_INV_SAB];
        const double spec = c[VP_LG2_I] + (c[VP_INV_SLO] + smooth_one);
        const double cut = c[VP_INV_NUMAX] * exp2_fast(x[r]);
        out[r] = (x[r] - c[VP_LG2_NUMAX] < -20) ? spec : spec - cut;
    }
}

// series_bracket (vag_kernels.h) whose common case -- the point stays within a node of where the previous row had it -- is
// four independent LDS reads and two compares; anything else takes the searching form.  Same result: the k with
// s_t[k] < t <= s_t[k+1] is unique on a strictly ascending lattice.
VAG_DEV int series_bracket_near(const double* __restrict__ s_t, int K, double t, int hint) {
    if (K >= 4) {
        const int h = hint < 1 ? 1 : (hint > K - 3 ? K - 3 : hint);
        const double a = s_t[h - 1], b = s_t[h], c = s_t[h + 1], d = s_t[h + 2];
        if (a < t && t <= d) return h - 1 + (b < t ? 1 : 0) + (c < t ? 1 : 0);
    }
    return series_bracket(s_t, K, t, hint);
}

template <int MODE>
__global__ void __launch_bounds__(SERIES_THREADS * 4)
vag_flux_series_pair_kernel(SeriesArgs a) {
    static_assert(MODE == FLUX_SYN, "the pair kernel evaluates the plain synchrotron spectrum");
    const int m = blockIdx.y;
    const int wave = threadIdx.x >> 6, tid = threadIdx.x & 63;
    const int KS = a.k_stride;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* s_sp = lds;
    {
        const VagGridMeta* Mp = a.meta + m;
        if (Mp->status != 0 || (long long)blockIdx.x * (blockDim.x >> 6) * a.pairs_per_block >= (long long)Mp->n_theta * Mp->n_phi_eff)
            return;
    }
    for (int i = threadIdx.x; i < SP_LDS_DOUBLES; i += blockDim.x) s_sp[i] = a.sp_table[i];
    double* s_band = s_sp + SP_LDS_DOUBLES;
    const int NB = a.n_bands;
    if (threadIdx.x < NB) s_band[threadIdx.x] = a.lg2_nu_obs[a.band_first[threadIdx.x]];
    __syncthreads();  // the only workgroup-wide barrier
    const int vb = blockIdx.x * (blockDim.x >> 6) + wave;
    if (vb >= a.max_blocks) return;
    const VagGridMeta M = a.meta[m];
    double* chunk_partial = a.partial + (size_t)m * a.max_chunks * a.n;
    const int n_phi_eff = M.n_phi_eff;
    const int n_pairs = M.n_theta * n_phi_eff;
    const int p0 = vb * a.pairs_per_block;
    if (p0 >= n_pairs) return;
    const int p1 = min(n_pairs, p0 + a.pairs_per_block);
    const int K = M.n_t;
    double* s_par = s_band + SERIES_MAX_BANDS + (size_t)wave * series_pair_region_doubles(KS, NB);
    double* s_t = s_par + VAG_NPAR * KS;  // [2][KS]
    double* s_dop = s_t + 2 * KS;         // [2][KS]
    double* s_Bw = s_dop + 2 * KS;        // [2][NB][KS]
    const LdsTab sp_tab = lds_tab(s_sp), lg_tab = lds_tab(s_sp + SP_TABLE_DOUBLES);

    const double one_plus_z = 1 + a.params[m].z;
    const double lg2_1pz = M.lg2_1pz;
    SpecConst sc;
    sc.init_fast(a.params[m].p, lg_tab);
    const double cos_obs = M.cos_obs, sin_obs = M.sin_obs;
    const double* gth = a.geo_th + (size_t)m * 3 * VAG_MAX_THETA;
    const double* gph = a.geo_ph + (size_t)m * 2 * VAG_MAX_PHI;
    const int* rep_of = a.g_rep_of + (size_t)m * VAG_MAX_THETA;
    // row geometry of the next 64 rows, one row per lane (see vag_flux_series_kernel)
    double g_a = 0, g_b = 0, g_c = 0;
    int g_rep = 0;
    auto load_row_geometry = [&](int base) {
        const int pr = min(base + tid, p1 - 1);
        const int j = pr / n_phi_eff, i = pr - j * n_phi_eff;
        g_rep = rep_of[j];
        g_a = gth[VAG_MAX_THETA + j] * gph[i] * sin_obs + gth[j] * cos_obs;
        g_b = (1 - g_a) / C_C * one_plus_z;
        g_c = gth[2 * VAG_MAX_THETA + j] + gph[VAG_MAX_PHI + i];
    };
    auto lane_value = [&](double v, int l) {
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
    };

    const bool have = tid < a.n;
    const double t = have ? a.lg2_t_obs[tid] : 0;
    const int my_band = have ? a.band_idx[tid] : 0;
    double acc = 0;
    int kprev = -1;  // this point's interval in the row before
    int staged_rep = -1;
#ifdef VAG_SERIES_STAMPS  // developer aid: cycles of one wavefront per phase
    long long c_eat = 0, c_brk = 0, c_items = 0, c_interp = 0, c_rest = 0, c_mark = __builtin_readcyclecounter();
    int n_two = 0, n_one = 0;
#define VAG_PAIR_MARK(acc_) do { const long long now_ = __builtin_readcyclecounter(); acc_ += now_ - c_mark; c_mark = now_; } while (0)
#else
#define VAG_PAIR_MARK(acc_) do { } while (0)
#endif

    // one or two rows (pair, pair + 1) that share the staged block
    auto rows = [&](auto nr_tag, int pair, int gl) {
        constexpr int NR = decltype(nr_tag)::value;
        double cos_v[NR], t_coeff[NR], lg2_dOmega[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            cos_v[r] = lane_value(g_a, gl + r);
            t_coeff[r] = lane_value(g_b, gl + r);
            lg2_dOmega[r] = lane_value(g_c, gl + r);
        }
        VAG_PAIR_MARK(c_rest);
        // ---- EAT logs of every lattice node (calc_eat_non_spreading, observer.cpp:143-205)
        for (int k = tid; k < K; k += SERIES_THREADS) {
            const LdsTab c2 = lds_tab(s_par + k * VAG_NPAR);
            const vdouble2 Gu = c2[VP_GAMMA / 2], rt = c2[VP_R / 2];
            double arg_d[NR], arg_t[NR];
            bool odd = false;  // an argument the table form does not cover (zero, subnormal, negative, inf, NaN)
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                arg_d[r] = Gu.x - Gu.y * cos_v[r];
                arg_t[r] = rt.y * one_plus_z + t_coeff[r] * rt.x;
                odd = odd || (unsigned)((__double2hiint(arg_d[r]) >> 20) - 1) >= 2046u || (unsigned)((__double2hiint(arg_t[r]) >> 20) - 1) >= 2046u;
            }
            double ld[NR], lt[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                ld[r] = -log2_tab_nb(arg_d[r], lg_tab);
                lt[r] = log2_tab_nb(arg_t[r], lg_tab);
            }
            if (odd) {
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    ld[r] = -log2_tab(arg_d[r], lg_tab);
                    lt[r] = log2_tab(arg_t[r], lg_tab);
                }
            }
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                s_dop[r * KS + k] = ld[r];
                s_t[r * KS + k] = lt[r];
            }
        }
        wave_sync();
        VAG_PAIR_MARK(c_eat);
        // ---- bracket of this lane's point in both rows
        bool in[NR];
        int kq[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const double* st = s_t + r * KS;
            in[r] = have && t >= st[0] && t <= st[K - 1];
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) kq[r] = in[r] ? series_bracket_near(s_t + r * KS, K, t, kprev) : 0;
        int kmin = K, kmax = -1;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const unsigned long long mask = __ballot(in[r]);
            if (mask != 0) {  // the points ascend in time, so their intervals ascend with the lane
                kmin = min(kmin, __builtin_amdgcn_readlane(kq[r], __ffsll((long long)mask) - 1));
                kmax = max(kmax, __builtin_amdgcn_readlane(kq[r], 63 - __clzll((long long)mask)));
            }
            if (in[r]) kprev = kq[r];
        }
        VAG_PAIR_MARK(c_brk);
        if (kmax < 0) return;  // wave-uniform: no point inside either row
        // ---- boundary spectra per (band, node of the points' window) for both rows
        const int nn = kmax + 2 - kmin;
        const int total = nn * NB;
        const float inv_nn = 1.0f / (float)nn;
        for (int idx = tid; idx < total; idx += SERIES_THREADS) {
            const int b = (int)(((float)idx + 0.5f) * inv_nn);
            const int kk = kmin + idx - b * nn;
            const SpecRegs regs = load_spec_regs(lds_tab(s_par) + __mul24(kk, VAG_NPAR / 2));
            const double nu_b = s_band[b] + lg2_1pz;
            double x[NR], v[NR], dop[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                dop[r] = s_dop[r * KS + kk];
                x[r] = nu_b - dop[r];
            }
            log2_I_nu_rows<NR>(regs, sc, x, sp_tab, v);
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const double geom = (lg2_dOmega[r] + regs[VP_LG2_R2]) + 3.0 * dop[r];  // log2(dOmega r^2 D^3)
                s_Bw[(r * NB + b) * KS + kk] = v[r] + geom;
            }
        }
        wave_sync();
        VAG_PAIR_MARK(c_items);
        // ---- log-log interpolation at the point (observer.h:405-433), rows in order
        double e[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const double* st = s_t + r * KS;
            const double* Bb = s_Bw + (r * NB + my_band) * KS;
            const int k = kq[r];
            const double blo = Bb[k], bhi = Bb[k + 1], t_lo = st[k], t_hi = st[k + 1];
            const double sl = (bhi - blo) * (1.0 / (t_hi - t_lo));
            const double val = exp2_fast(blo + (t - t_lo) * sl);
            e[r] = (in[r] && isfinite(sl)) ? val : 0.0;
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) acc += e[r];
        VAG_PAIR_MARK(c_interp);
    };

    int pair = p0;
    while (pair < p1) {
        const int gl = (pair - p0) & 63;
        if (gl == 0) load_row_geometry(pair);
        const int rep = __builtin_amdgcn_readlane(g_rep, gl);
        if (pair > p0 && (pair - p0) % a.chunk == 0) {  // close the chunk before this row
            double* dst = chunk_partial + (size_t)(pair / a.chunk - 1) * a.n;
            if (have) dst[tid] = acc;
            acc = 0;
        }
        wave_sync();
        if (rep != staged_rep) {
            const double* src = a.cellpar + (a.lay.cell_off[m] + (long long)rep * K) * VAG_NPAR;
            const float inv_K = 1.0f / (float)K;
            for (int q0 = tid; q0 < VAG_NPAR * K; q0 += 4 * SERIES_THREADS) {
                double v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int q = q0 + u * SERIES_THREADS;
                    v[u] = q < VAG_NPAR * K ? src[q] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int q = q0 + u * SERIES_THREADS;
                    if (q < VAG_NPAR * K) {
                        const int par = (int)(((float)q + 0.5f) * inv_K), k = q - par * K;
                        s_par[k * VAG_NPAR + par] = v[u];
                    }
                }
            }
            staged_rep = rep;
            wave_sync();
        }
        // the next row joins when it exists, shares the photon block and the chunk, and its geometry sits in this lane block
        const bool two = pair + 1 < p1 && gl < 63 && (pair + 1 - p0) % a.chunk != 0 && __builtin_amdgcn_readlane(g_rep, (gl + 1) & 63) == rep;
        if (two) {
            rows(std::integral_constant<int, 2>{}, pair, gl);
            pair += 2;
        } else {
            rows(std::integral_constant<int, 1>{}, pair, gl);
            pair += 1;
        }
#ifdef VAG_SERIES_STAMPS
        n_two += two, n_one += !two;
#endif
    }
#ifdef VAG_SERIES_STAMPS
    VAG_PAIR_MARK(c_rest);
    if (m == 0 && vb == 0 && tid == 0)
        printf("pair wave 0: rows %d (%d pairs + %d single) K %d  cycles: eat %lld  bracket %lld  items %lld  interp %lld  rest %lld\n", p1 - p0,
               n_two, n_one, K, c_eat, c_brk, c_items, c_interp, c_rest);
#endif
    {
        double* dst = chunk_partial + (size_t)((p1 - 1) / a.chunk) * a.n;
        if (have) dst[tid] = acc;
    }
}

}  // namespace vag
