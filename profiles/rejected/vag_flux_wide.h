// REJECTED on measurement (round 3): bitwise the same fluxes as vag_flux_grid_kernel<false, FLUX_SYN>, but 27.0-27.7 ms against 21.7 ms
// per 512 C2 models.  With one workgroup per CU all sixteen wavefronts cross the same barrier every row, so the four wavefronts of a
// SIMD are in the same phase (all fighting for the VALU in the boundary spectra, then all waiting on LDS in the interpolation / EAT /
// bracket chains) -- staggering the phase order over the wavefronts of a SIMD recovered 0.65 ms only; two INDEPENDENT 512-lane
// workgroups drift against each other and overlap these phases for free.  Kept for the record; not part of the product.
//
// vag_flux_wide.h -- the grid flux kernel for large (nu, t) grids of plain synchrotron models (the C2 shape of the bench): ONE
// 1024-lane workgroup per CU instead of two of 512.
//
// Same algorithm, same arithmetic and the same summation order as vag_flux_grid_kernel<false, FLUX_SYN> (vag_kernels.h: fused
// Observer::observe + Observer::specific_flux, src/core/observer.cpp:143-205,439-454 and src/core/observer.h:355-445), so the
// fluxes are the same bits.  What changes is the schedule (profiles/r03_flux_phase_budget_before.txt):
//   * the two 512-lane workgroups of a CU each staged their own photon row and their own softplus / log2 tables; one workgroup of
//     sixteen wavefronts shares them, and the 39 KB that frees hold a SECOND boundary block, bracket buffer and Doppler / geometry
//     buffer, so a row needs ONE barrier instead of two: in one interval the workgroup interpolates row r (B), evaluates the
//     boundary spectra and the brackets of row r + 1 (A1) and the EAT logarithms of row r + 2 (A0);
//   * the latency-bound side jobs sit on eight "side" wavefronts (0-3: EAT logarithms, one lattice node per lane; 4-7: bracket
//     lookup, one requested time per lane) whose lanes own ONE (nu, t) accumulator slot each, the other eight own THREE: every
//     wavefront then carries about the same work per row, where the 512-lane kernel made four wavefronts wait ~1.8 k cycles at
//     the second barrier of every row for the four that held the lattice.
// Served shapes (the host checks them, vag_capi.hip: run_flux_grid): no SSC / spreading / pieces, lattice <= 256 nodes, <= 256
// requested times, <= 2048 slots; everything else stays with vag_flux_grid_kernel.
#pragma once
#include "vag_kernels.h"

namespace vag {

constexpr int WIDE_THREADS = 1024;
constexpr int WIDE_SIDE = 512;       // lanes of the side wavefronts: one slot each
constexpr int WIDE_MAX_K = 256;      // lattice nodes: one per lane of wavefronts 0-3
constexpr int WIDE_MAX_NT = 256;     // requested times: one per lane of wavefronts 4-7
constexpr int WIDE_MAX_SLOTS = 2048; // 512 x 1 + 512 x 3

__host__ __device__ inline size_t flux_wide_lds_bytes(int ks, int nt, int nnu) {
    const size_t d = (size_t)SP_LDS_DOUBLES + (size_t)VAG_NPAR * ks + 6 * (size_t)ks + 2 * (size_t)ks * nnu + 3 * (size_t)nt + nnu +
                     (size_t)nt * nnu;
    return sizeof(double) * d + sizeof(int) * (2 * (size_t)nt + 16);
}

__global__ void __launch_bounds__(WIDE_THREADS, 4)
vag_flux_grid_wide_kernel(FluxArgs a) {
    constexpr int THREADS = WIDE_THREADS;
    const int m = blockIdx.y;
    const VagGridMeta* Mp = a.meta + m;
    if (Mp->status != 0) return;
    const int n_phi_eff = Mp->n_phi_eff;
    const int n_pairs = Mp->n_theta * n_phi_eff;
    const int p0 = blockIdx.x * a.pairs_per_block;
    if (p0 >= n_pairs) return;
    const int p1 = min(n_pairs, p0 + a.pairs_per_block);
    const int tid = threadIdx.x, wave = tid >> 6;
    const int K = Mp->n_t, KS = a.k_stride;  // K <= KS <= WIDE_MAX_K
    const int nt = a.nt, nnu = a.nnu;
    const int slots = nt * nnu;

    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* s_sp = lds;                       // softplus + log2 tables
    double* s_par = s_sp + SP_LDS_DOUBLES;    // [KS][VAG_NPAR] staged photon / shock block of the representative row
    double* s_t = s_par + VAG_NPAR * KS;      // [2][KS]   \  EAT quantities of the rows r, r + 1 (mod 2)
    double* s_dop = s_t + 2 * KS;             // [2][KS]    |
    double* s_geom = s_dop + 2 * KS;          // [2][KS]   /
    double* s_B = s_geom + 2 * KS;            // [2][nnu][KS] boundary log2-luminosities
    double* s_tobs = s_B + 2 * (size_t)KS * nnu;
    double* s_nu = s_tobs + nt;
    double* s_w = s_nu + nnu;                 // [2][nt] position of each requested time inside its interval
    double* s_acc = s_w + 2 * nt;             // [nnu * nt] partial grid
    int* s_kidx = (int*)(s_acc + slots);      // [2][nt]
    int* s_win = s_kidx + 2 * nt;             // [2][4][2] window counts of wavefronts 0-3 per row buffer

    const vag_model_params* Pp = a.params + m;
    const double one_plus_z = 1 + Pp->z;
    const double opz_over_c = one_plus_z / C_C;
    {
        const double lg2_1pz = Mp->lg2_1pz;
        for (int i = tid; i < nt; i += THREADS) s_tobs[i] = a.lg2_t_obs[i];
        for (int l = tid; l < nnu; l += THREADS) s_nu[l] = a.lg2_nu_obs[l] + lg2_1pz;
        for (int i = tid; i < SP_LDS_DOUBLES; i += THREADS) s_sp[i] = a.sp_table[i];
    }
    SpecConst sc;
    sc.init(Pp->p);
    const double cos_obs = Mp->cos_obs, sin_obs = Mp->sin_obs;
    const int* rep_of = a.g_rep_of + (size_t)m * VAG_MAX_THETA;
    const LdsTab sp_tab = lds_tab(s_sp), lg_tab = lds_tab(s_sp + SP_TABLE_DOUBLES);
    const double* gth = a.geo_th + (size_t)m * 3 * VAG_MAX_THETA;
    const double* gph = a.geo_ph + (size_t)m * 2 * VAG_MAX_PHI;
    for (int s = tid; s < slots; s += THREADS) s_acc[s] = 0;
    for (int i = tid; i < 2 * nt; i += THREADS) s_kidx[i] = 0;
    // slot ownership (fixed per lane: no atomics, fixed sum order).  slot = l * nt + idx, packed as idx | l * KS << 16, sign bit =
    // no such slot.  Side lanes: slot tid.  Plain lanes (tid >= 512): tid, tid + 512, tid + 1024.
    const float inv_nt = __builtin_amdgcn_rcpf((float)nt);
    auto slot_desc = [&](int slot) -> int {
        const int l = (int)(((float)slot + 0.5f) * inv_nt);
        return slot < slots ? ((slot - __mul24(l, nt)) | (__mul24(l, KS) << 16)) : (int)0x80000000;
    };
    int desc[3];
    desc[0] = slot_desc(tid);
    desc[1] = wave >= 8 ? slot_desc(tid + 512) : (int)0x80000000;
    desc[2] = wave >= 8 ? slot_desc(tid + 1024) : (int)0x80000000;
    int breach = 0;
    (void)breach;

    // ---- the three jobs of an interval ----
    // A0: EAT logarithms of row (j, i) into buffer `buf` -- wavefronts 0-3, node k = tid (eat_row's expressions, rounding for rounding)
    struct EatIn {
        double cos_v, t_coeff, lg2_dOmega;
    };
    auto eat_geometry = [&](int j, int i) {
        double g_sin, g_cph, g_cos, g_dth, g_dph;
        sload5(gth + VAG_MAX_THETA + j, gph + i, gth + j, gth + 2 * VAG_MAX_THETA + j, gph + VAG_MAX_PHI + i, g_sin, g_cph, g_cos, g_dth, g_dph);
        EatIn e;
        e.cos_v = g_sin * g_cph * sin_obs + g_cos * cos_obs;
        e.t_coeff = (1 - e.cos_v) * opz_over_c;
        e.lg2_dOmega = g_dth + g_dph;
        return e;
    };
    // B of the row in buffer `buf` for the lane's slots [0, UE), optionally with the EAT chain of another row next to it
    auto interp = [&](int buf, auto n_slots, auto with_eat, const EatIn& e, int ebuf) {
        constexpr int UE = decltype(n_slots)::value;
        constexpr bool EAT = decltype(with_eat)::value;
        const double* sB = s_B + (size_t)buf * KS * nnu;
        const double* sw = s_w + buf * nt;
        const int* sk = s_kidx + buf * nt;
        [[maybe_unused]] vdouble2 e_Gu, e_rt;
        [[maybe_unused]] double e_r2 = 0;
        [[maybe_unused]] const int ek = min(tid, K - 1);
        if constexpr (EAT) {
            const double* c = s_par + ek * VAG_NPAR;
            const LdsTab c2 = lds_tab(c);
            e_Gu = c2[VP_GAMMA / 2], e_rt = c2[VP_R / 2];
            e_r2 = c[VP_LG2_R2];
        }
        int dv[UE], kq[UE], iq[UE];
        double lo[UE], hi[UE], wq[UE], aq[UE];
#pragma unroll
        for (int u = 0; u < UE; ++u) {
            dv[u] = desc[u];
            asm volatile("" : "+v"(dv[u]));  // no address arithmetic hoisted out of the row loop: the A1 loop has no registers to spare
            iq[u] = dv[u] & 0xffff;
            kq[u] = sk[iq[u]];
        }
#pragma unroll
        for (int u = 0; u < UE; ++u) {
            const int kk = ((dv[u] >> 16) & 0x7fff) + kq[u];
            lo[u] = sB[kk], hi[u] = sB[kk + 1];
            wq[u] = sw[iq[u]];
            aq[u] = s_acc[min(tid + u * 512, slots - 1)];
        }
        [[maybe_unused]] bool sp_a = false, sp_b = false;
        [[maybe_unused]] double e_dop = 0, e_lt = 0;
        if constexpr (EAT) {
            e_dop = -log2_tab_core(fma(-e_Gu.y, e.cos_v, e_Gu.x), lg_tab, sp_a);
            e_lt = log2_tab_core(fma(e.t_coeff, e_rt.x, e_rt.y * one_plus_z), lg_tab, sp_b);
        }
#pragma unroll
        for (int u = 0; u < UE; ++u) {
            const double x = fma(hi[u] - lo[u], wq[u], lo[u]);  // finite exactly for the terms that count (vag_kernels.h)
            aq[u] += exp2_fast(isfinite(x) ? x : -2000.0);
        }
#pragma unroll
        for (int u = 0; u < UE; ++u)
            if (dv[u] >= 0) s_acc[tid + u * 512] = aq[u];
        if constexpr (EAT) {
            if (sp_a) e_dop = -log2(fma(-e_Gu.y, e.cos_v, e_Gu.x));
            if (sp_b) e_lt = log2(fma(e.t_coeff, e_rt.x, e_rt.y * one_plus_z));
            const double w_lo = s_tobs[0], w_hi = s_tobs[nt - 1];
            WinCount wc;
            wc.add(tid < K, e_lt, w_lo, w_hi);
            if ((tid & 63) == 0) s_win[ebuf * 8 + wave * 2] = wc.n_lt, s_win[ebuf * 8 + wave * 2 + 1] = wc.n_le;
            if (tid < K) {
                s_dop[ebuf * KS + ek] = e_dop;
                s_t[ebuf * KS + ek] = e_lt;
                s_geom[ebuf * KS + ek] = (e.lg2_dOmega + e_r2) + 3.0 * e_dop;
            }
        }
    };
    // bracket lookup of the row in buffer `buf` (hints: the previous row's intervals in the other buffer) -- wavefronts 4-7
    auto bracket = [&](int buf) {
        const double* s_tc = s_t + buf * KS;
        const int idx = tid - 256;
        if (idx < 0 || idx >= nt) return;
        const double row_t0 = s_tc[0], row_tN = s_tc[K - 1];
        const double tq = s_tobs[idx];
        int kk = 0;
        double w = NAN;  // a time outside the row's lattice: no finite exponent, no contribution
        if (tq >= row_t0 && tq < row_tN) {
            int lo = s_kidx[(buf ^ 1) * nt + idx], hi;
            lo = min(max(lo, 1), K - 3);
            const double ta = s_tc[lo - 1], tb = s_tc[lo], tc = s_tc[lo + 1], td = s_tc[lo + 2];
            if (K >= 4 && ta <= tq && tq < td) {
                lo = tq < tb ? lo - 1 : (tq < tc ? lo : lo + 1);
            } else {
                lo = K >= 4 ? lo : 0;
                if (s_tc[lo] <= tq) {
                    int step = 1;
                    hi = lo + 1;
                    while (hi < K - 1 && s_tc[hi] <= tq) {
                        lo = hi;
                        step <<= 1;
                        hi = min(lo + step, K - 1);
                    }
                } else {
                    int step = 1;
                    hi = lo;
                    lo = hi - 1;
                    while (s_tc[lo] > tq) {
                        hi = lo;
                        step <<= 1;
                        lo = max(hi - step, 0);
                    }
                }
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (s_tc[mid] <= tq)
                        lo = mid;
                    else
                        hi = mid;
                }
            }
            kk = lo;
            const double t_lo = s_tc[lo];
            w = (tq - t_lo) * (1.0 / (s_tc[lo + 1] - t_lo));
        }
        s_kidx[buf * nt + idx] = kk;
        s_w[buf * nt + idx] = w;
    };
    // A1: boundary values of the row in buffer `buf`; returns whether the row lies inside the observation window (block-uniform)
    auto boundary = [&](int buf) -> bool {
        const double* s_tc = s_t + buf * KS;
        const double row_t0 = s_tc[0], row_tN = s_tc[K - 1];
        const double w_lo = s_tobs[0], w_hi = s_tobs[nt - 1];
        const bool in_window = !(row_tN < w_lo || row_t0 > w_hi);
        if (!in_window) return false;
        int n_lt = 0, n_le = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) n_lt += s_win[buf * 8 + 2 * w], n_le += s_win[buf * 8 + 2 * w + 1];
        n_lt = __builtin_amdgcn_readfirstlane(n_lt);
        n_le = __builtin_amdgcn_readfirstlane(n_le);
        const int k_lo = n_lt > 0 ? n_lt - 1 : 0;
        const int k_hi = min(max(n_le, k_lo + 1), K - 1);
        const int nk = k_hi - k_lo + 1;
        const int npair_nu = (nnu + 1) >> 1;
        const int total = nk * npair_nu;
        const float inv_nk = __builtin_amdgcn_rcpf((float)nk);
        const int dq_l = THREADS / nk, dq_k = THREADS - dq_l * nk;
        int lg = (int)(((float)tid + 0.5f) * inv_nk);
        int kk = tid - __mul24(lg, nk);
        double* sB = s_B + (size_t)buf * KS * nnu;
        const double* sd = s_dop + buf * KS;
        const double* sg = s_geom + buf * KS;
        int bofs = __mul24(lg, 2 * KS);
        const int top = (nnu - 1) * KS;
        for (int q = tid; q < total; q += THREADS) {
            const int k = k_lo + kk;
            const int l0 = lg * 2, l1 = min(l0 + 1, nnu - 1);
            const double dop = sd[k], geom = sg[k];
            const SpecRegs regs = load_spec_regs(lds_tab(s_par) + __mul24(k, VAG_NPAR / 2));
            const double b0 = log2_I_nu_fast(regs, 1, sc, s_nu[l0] - dop, sp_tab);
            const double b1 = log2_I_nu_fast(regs, 1, sc, s_nu[l1] - dop, sp_tab);
            sB[bofs + k] = b0 + geom;
            sB[min(bofs + KS, top) + k] = b1 + geom;
            kk += dq_k;
            lg += dq_l;
            bofs += dq_l * 2 * KS;
            if (kk >= nk) {
                kk -= nk;
                ++lg;
                bofs += 2 * KS;
            }
        }
        return true;
    };

    using N1 = std::integral_constant<int, 1>;
    using N3 = std::integral_constant<int, 3>;
    __syncthreads();
    int row = p0;
    int je = p0 / n_phi_eff, ie = p0 - je * n_phi_eff;  // (theta, phi) of the next row to take its EAT logarithms
    while (row < p1) {
        // a run of rows that share one representative photon row: stage it, then pipeline the run
        const int rep = sload_i32(rep_of + je);
        {
            const double* src = a.cellpar + (a.cell_off[m] + (long long)rep * K) * VAG_NPAR;
#pragma unroll 1
            for (int q = tid; q < VAG_NPAR * K; q += THREADS) {
                const int par = (int)(((float)q + 0.5f) / (float)K);
                s_par[(q - par * K) * VAG_NPAR + par] = src[(size_t)par * K + (q - par * K)];
            }
        }
        __syncthreads();
        int r_e = row, r_a = row, r_b = row;  // next row for A0 / A1 / B
        bool more = true, win_b = false;  // rows left in the run; the row B works on lies inside the observation window
        for (;;) {
            const bool doB = r_b < r_a, doA = r_a < r_e, doE = more;
            if (!doB && !doA && !doE) break;
            EatIn e{};
            if (doE && wave < 4) e = eat_geometry(je, ie);
            // Half of the wavefronts of every SIMD take the boundary spectra first and the latency-bound jobs (interpolation, EAT
            // logarithms, bracket lookup) afterwards, the other half the other way round: with all sixteen in the same phase the
            // SIMDs alternate between four wavefronts fighting for the VALU and four waiting on LDS (27.7 ms against the
            // two-workgroup kernel's 21.7 per 512 C2 models).  Wavefront w sits on SIMD w mod 4.
            const bool spectra_first = ((wave >> 2) & 1) != 0;
            bool win_new = false;
            if (doA && spectra_first) win_new = boundary(r_a & 1);
            if (wave < 4) {
                if (doE && doB && win_b)
                    interp(r_b & 1, N1{}, std::true_type{}, e, r_e & 1);
                else {
                    if (doB && win_b) interp(r_b & 1, N1{}, std::false_type{}, e, 0);
                    if (doE) {  // the logarithms alone (first two intervals of a run, rows outside the window)
                        const double w_lo = s_tobs[0], w_hi = s_tobs[nt - 1];
                        eat_row(s_par, KS, K, tid, 256, e.cos_v, e.t_coeff, one_plus_z, e.lg2_dOmega, s_t + (r_e & 1) * KS, s_dop + (r_e & 1) * KS,
                                s_geom + (r_e & 1) * KS, lg_tab, nullptr);
                        WinCount wc;
                        wc.add(tid < K, tid < K ? s_t[(r_e & 1) * KS + tid] : 0.0, w_lo, w_hi);
                        if ((tid & 63) == 0) s_win[(r_e & 1) * 8 + wave * 2] = wc.n_lt, s_win[(r_e & 1) * 8 + wave * 2 + 1] = wc.n_le;
                    }
                }
            } else if (wave < 8) {
                if (doA) bracket(r_a & 1);
                if (doB && win_b) interp(r_b & 1, N1{}, std::false_type{}, e, 0);
            } else {
                if (doB && win_b) interp(r_b & 1, N3{}, std::false_type{}, e, 0);
            }
            if (doA && !spectra_first) win_new = boundary(r_a & 1);
            __syncthreads();
            if (doB) ++r_b;
            if (doA) ++r_a;
            win_b = doA && win_new;  // the row A1 just served is the one B takes in the next interval
            if (doE) {
                ++r_e;
                if (++ie == n_phi_eff) ie = 0, ++je;
                more = r_e < p1 && sload_i32(rep_of + (r_e < p1 ? je : 0)) == rep;
            }
        }
        row = r_e;
    }
    __syncthreads();
    double* my_partial = a.partial + ((size_t)m * a.max_blocks + blockIdx.x) * slots;
    for (int s = tid; s < slots; s += THREADS) my_partial[s] = s_acc[s];
}

}  // namespace vag
