// vag_fit_rows.h -- the likelihood's flux pass with ONE (theta, phi) ROW PER LANE.
//
// vag_flux_series_kernel (vag_kernels.h) gives a row to a wavefront: four short phases (EAT logs of the lattice nodes, bracket
// search per point, boundary spectra per (band, node of the points' window), interpolation per point) that each wait on the one
// before and each fill part of the wavefront.  On a fit's shape -- a few bands, <= 64 data points, lattices of 14-46 nodes, ~770
// rows per walker -- that is 347 VALU instructions per row at 54 % VALU busy with the LDS of the staged rows holding occupancy
// at three wavefronts per SIMD (profiles/r02_pmc_walker_kernels.txt).
//
// Here a wavefront takes 64 consecutive rows of a model, one per lane, and all lanes walk the lattice nodes k together:
//   * the node's observer time and Doppler factor per lane (calc_eat_non_spreading, observer.cpp:143-205), one node ahead;
//   * the data points are sorted in time, so each lane keeps a cursor into them: the points inside (t[k-1], t[k]] are the next few;
//   * the node's boundary spectra for the fit's bands are evaluated where a lane has points next to it (the wavefront skips the
//     node when no lane does), SmoothPowerLawSyn::compute_log2_I_nu through the same evaluator as the other flux kernels;
//   * each of the lane's points in the interval is interpolated log-log (observer.h:405-433) and added to the wavefront's
//     per-point accumulator in LDS (ds_add_f64: lanes of one instruction are applied in lane order, so the sum is reproducible).
// No staged rows, no per-row LDS arrays, no phase barriers: every lane is busy on every step, and the only LDS besides the
// evaluator's tables is a few copies of 64 accumulators per wavefront.  A block's 64 rows and their order depend on the model alone,
// and a block writes one partial sum per data point and lattice segment, so a walker's ln L does not depend on what else is in the
// batch -- nor on which wavefront of the PERSISTENT launch served the block (round 4: workgroups that take items until none is left).
//
// Reference: Observer::specific_flux_series (src/core/observer.h:447-538).
#pragma once
#include "vag_ic_kernels.h"
#include "vag_kernels.h"

namespace vag {

constexpr int FITROWS_WAVES = 4;   // wavefronts per workgroup (they only share the tables and the data points)
constexpr int FITROWS_BANDS = 8;   // distinct frequencies handled (instantiations for <= 4 and <= 8; more: vag_flux_series_kernel)
constexpr int FITROWS_MAX_POINTS = 512;  // data points of a pass (SERIES_THREADS * SERIES_MAX_SLOTS)
constexpr int FITROWS_ROWS = 64;   // rows per block = lanes of a wavefront
constexpr int FITROWS_SEGS = 4;    // lattice segments per block: each (block, segment) is one partial sum of the model's tree
#ifndef VAG_FITROWS_STRIPES
#define VAG_FITROWS_STRIPES 4
#endif
constexpr int FITROWS_STRIPES = VAG_FITROWS_STRIPES;  // copies of a wavefront's per-point sums for short data sets: lane L adds to copy
                                                      // L mod 4 (rows near each other hit the same point in the same instruction)
__host__ __device__ inline int fit_rows_npad(int n) { return (n + SERIES_THREADS - 1) / SERIES_THREADS * SERIES_THREADS; }
__host__ __device__ inline int fit_rows_stripes(int n) { return n <= 128 ? FITROWS_STRIPES : 1; }

// How the LDS applies a wave64 ds_add_f64 (profiles/micro/lds_atomic.hip, MI355X): sixteen lanes at a time; lanes of such a group
// that hit one address take turns at ~3 cycles each (8.4 cycles per instruction without collisions, 44 with four lanes per address
// in every group, 191 with all sixteen), lanes that hit different addresses of one bank (32 banks of 4 B: doubles 16 apart) at ~2.
// Neighbouring rows reach the same requested time in the same instruction, so
//   * the copies ("stripes") of a wavefront's sums lie 16 / stripes doubles apart modulo 16 (the same slot of two copies in different
//     banks: with a stride that is a multiple of 16 -- 64 points, 4 x 100 slots -- the copies bought nothing), and
//   * lane L takes row 4 (L mod 16) + L / 16 of the wavefront's 64: the sixteen lanes the LDS serves together are rows four apart,
//     and rows next to each other are in different groups.
#ifndef VAG_ROWS_BANK_STRIDE
#define VAG_ROWS_BANK_STRIDE 1
#endif
#ifndef VAG_ROWS_PERMUTE
#define VAG_ROWS_PERMUTE 1
#endif
__host__ __device__ inline int rows_acc_stride(int n, int stripes) {
#if VAG_ROWS_BANK_STRIDE
    const int want = (16 / stripes) & 15;
    return n + ((want - n) & 15);
#else
    return n;
#endif
}
__host__ __device__ inline int rows_lane_row(int lane) {
#if VAG_ROWS_PERMUTE
    return ((lane & 15) << 2) | (lane >> 4);
#else
    return lane;
#endif
}

// bytes of LDS of one workgroup: tables, the points' times and bands, per wavefront the bands' frequencies and [stripes][stride] sums
__host__ __device__ inline size_t fit_rows_lds_bytes(int n) {
    const int np = fit_rows_npad(n), st = fit_rows_stripes(n);
    return sizeof(double) * (SP_LDS_DOUBLES + np + FITROWS_WAVES * SERIES_MAX_BANDS + (size_t)FITROWS_WAVES * st * rows_acc_stride(np, st)) + sizeof(int) * np;
}


// A frequency far beyond the synchrotron spectrum's exponential cut-off (SmoothPowerLawSyn::compute_log2_I_nu subtracts
// log2(e) nu' / nu_M, smooth-power-law-syn.cpp:159-167): at nu' > 2^10.5 nu_M that term alone is below -2089 while everything else of
// log2 I' + the geometry stays within a few hundred (a double's smallest denormal is 2^-1074), so 2^(...) is an exact zero in the
// reference's sum whatever the interval's other end holds (nu_M in the comoving frame is the burn-off limit, ~2.4e22 Hz / (1 + Y): it does not jump between lattice nodes).  When
// EVERY lane of the wavefront that needs the node is that far out, the evaluation is skipped and the value is -inf -- the interval's
// slope is then not finite and the interval adds nothing (observer.h:422-426), i.e. the same exact zero.  This is the TeV band of an
// SSC request (BASELINE configs[2] / [4]: 2.4e26 Hz), a quarter of the synchrotron pass's evaluations there.
// Returns true (and sets `value`) when the calling lanes may skip the band.  Used by the row-per-lane GRID kernel (vag_grid_rows.h).
constexpr double FLUX_DEAD_LG2 = 10.5;
VAG_DEV bool band_is_dead(double lg2_nu_comoving, double lg2_nu_max, double& value) {
#ifdef VAG_NO_DEAD_BANDS  // developer builds: evaluate everything
    return false;
#else
    if (__ballot(lg2_nu_comoving - lg2_nu_max < FLUX_DEAD_LG2) != 0) return false;  // (a NaN counts as dead: non-finite either way)
    value = -INFINITY;
    return true;
#endif
}

#ifndef VAG_ROWS_MIN_WG
#define VAG_ROWS_MIN_WG 3  // workgroups per CU the row-per-lane kernels are compiled for (developer builds: 4 = 128 VGPRs)
#endif
// LDS of a workgroup as the items see it (the tables and the data points are the launch's, the rest is the wavefront's own)
struct FitRowsLds {
    const double* s_sp;   // softplus + log2 tables
    const double* s_tp;   // [NP] log2 of the data points' times, ascending; +inf beyond n
    const int* s_band_of; // [NP] band of each point
    double* s_band;       // this wavefront's [SERIES_MAX_BANDS] log2 nu (1 + z) of the fit's bands for the model at hand
    double* s_acc;        // this wavefront's per-point sums [stripe][NS]
};

// One work item of the fit: segment(s) wseg (of W) of the lattice walk of block vb (64 rows) of model m, by one wavefront.
// COUNT: the item also tallies its work in the path's units -- a spectrum evaluation per (row, node whose boundary values
// are formed, band), an interpolation per (row, data point inside the row's lattice) -- into a.tally; untimed passes only.
template <int MODE, int NBMAX, bool SPREAD, bool COUNT = false>
VAG_DEV void fit_rows_item(const SeriesArgs& a, const FitRowsLds& L, int m, int vb, int wseg, int W, int lane) {
    const VagGridMeta* Mp = a.meta + m;
    const int n_pairs = Mp->n_theta * Mp->n_phi_eff;
    const int n = a.n, NB = a.n_bands, NP = fit_rows_npad(n), stripes = fit_rows_stripes(n), NS = rows_acc_stride(NP, stripes);
    const double* s_sp = L.s_sp;
    const double* s_tp = L.s_tp;
    const int* s_band_of = L.s_band_of;
    double* s_band = L.s_band;
    double* s_acc = L.s_acc;
    double* my_acc = s_acc + (lane % stripes) * NS;  // the copy this lane adds to
    wave_sync();  // (the previous item's reads of s_band are done)
    if (lane < NB) s_band[lane] = a.lg2_nu_obs[a.band_first[lane]] + Mp->lg2_1pz;
    wave_sync();
    // The lattice of a block is cut into FITROWS_SEGS segments, each with its own partial sum: the intervals between nodes are
    // independent, a segment starts from nothing but its first node.  W wavefronts share the block and take FITROWS_SEGS / W
    // consecutive segments each; one that walks several in a row flushes its accumulators at the cuts, and its running state
    // at a cut (cursor, the cut node's boundary values) is bit for bit what a wavefront starting there computes.  So the
    // partial sums do not depend on W: the host picks W = 4 for small batches (the longest sequential chain of the pass is a
    // quarter as long) and W = 1 for large ones (one prologue per block instead of four).
    const int p0 = vb * FITROWS_ROWS;
    const LdsTab sp_tab = lds_tab(s_sp), lg_tab = lds_tab(s_sp + SP_TABLE_DOUBLES);
    const int K = Mp->n_t, n_phi_eff = Mp->n_phi_eff;
    const double one_plus_z = 1 + a.params[m].z;
    SpecConst sc;
    sc.init_fast(a.params[m].p, lg_tab);
    int seg = wseg * (FITROWS_SEGS / W);                 // current segment
    const int seg_end = seg + FITROWS_SEGS / W;           // one past this wavefront's last segment
    auto cut = [&](int s) { return (K - 1) * s / FITROWS_SEGS; };  // segment s covers the steps (cut(s), cut(s + 1)]
    double* my_partial = a.partial + ((size_t)m * a.max_chunks + (size_t)vb * FITROWS_SEGS) * n;
    auto flush = [&](int s) {  // close segment s: its partial sum leaves, the accumulators start over
        wave_sync();
        for (int q = lane; q < NP; q += SERIES_THREADS) {
            double sum = s_acc[q];
            for (int c = 1; c < stripes; ++c) sum += s_acc[c * NS + q];  // fixed order
            if (q < n) my_partial[(size_t)s * n + q] = sum;
            for (int c = 0; c < stripes; ++c) s_acc[c * NS + q] = 0;
        }
        wave_sync();
    };
    const int k_first = cut(seg), k_last = cut(seg_end);
    if (k_last <= k_first) {  // fewer intervals than segments: nothing to walk here
        for (; seg < seg_end; ++seg) flush(seg);
        return;
    }

    // this lane's row
    const bool valid = p0 + rows_lane_row(lane) < n_pairs;
    const int pair = valid ? p0 + rows_lane_row(lane) : n_pairs - 1;
    const int j = pair / n_phi_eff, i = pair - j * n_phi_eff;
    const double* gth = a.geo_th + (size_t)m * 3 * Mp->th_stride;
    const double* gph = a.geo_ph + (size_t)m * 2 * Mp->ph_stride;
    const int rep = a.g_rep_of[(size_t)m * Mp->th_stride + j] + i * Mp->rep_phi_stride;  // ((phi, theta) pair rows: non-axisymmetric spreading jet)
    const double cos_phi = gph[i], lg2_dphi = gph[Mp->ph_stride + i];
    const double cos_v = gth[Mp->th_stride + j] * cos_phi * Mp->sin_obs + gth[j] * Mp->cos_obs;  // (non-spreading rows)
    const double t_coeff = (1 - cos_v) / C_C * one_plus_z;
    const double lg2_dOmega = gth[2 * Mp->th_stride + j] + lg2_dphi;
    const long long cell0 = a.lay.cell_off[m] + (long long)rep * K;
    const double* row = a.cellpar + cell0 * VAG_NPAR;  // [VAG_NPAR][K]
    const double* geo = SPREAD ? a.cellgeo + cell0 * 3 : nullptr;  // [3][K]
    const double sin_obs = Mp->sin_obs, cos_obs = Mp->cos_obs;
    int breach = 0;

    // EAT quantities of node k: log2 observer time, log2 Doppler factor, 2 log2 r
    // EAT quantities of a node: log2 observer time, log2 Doppler factor, log2 solid angle (explicit fma: a wavefront that starts
    // at a node and one that arrives there must form these sums the same way)
    auto eat = [&](double G, double u, double r, double teng, double g_cos, double g_sin, double g_ldc, double& lt, double& dop,
                   double& ldo) {
        if constexpr (SPREAD) {
            const double cv = fma(g_sin * cos_phi, sin_obs, g_cos * cos_obs);
            dop = -log2_tab(fma(-u, cv, G), lg_tab);
            lt = log2_tab(fma((1 - cv) * r, 1.0 / C_C, teng) * one_plus_z, lg_tab);
            ldo = g_ldc + lg2_dphi;
        } else {
            dop = -log2_tab(fma(-u, cos_v, G), lg_tab);
            lt = log2_tab(fma(t_coeff, r, teng * one_plus_z), lg_tab);
            ldo = lg2_dOmega;
        }
    };
    auto node = [&](int k, double& lt, double& dop, double& lr2, double& ldo) {
        lr2 = row[VP_LG2_R2 * K + k];
        const double gc = SPREAD ? geo[k] : 0, gs = SPREAD ? geo[K + k] : 0, gl = SPREAD ? geo[2 * K + k] : 0;
        eat(row[VP_GAMMA * K + k], row[VP_U * K + k], row[VP_R * K + k], row[VP_TENG * K + k], gc, gs, gl, lt, dop, ldo);
    };
    // boundary values B[b] = log2 I'(nu_b (1+z) / D_k) + log2(dOmega r^2 D^3) of node k for the fit's bands
    auto boundary = [&](int k, double dop, double lr2, double ldo, double (&B)[NBMAX]) {
        const double geom = ((SPREAD ? ldo : lg2_dOmega) + lr2) + 3.0 * dop;  // (non-spreading: the per-node copies stay dead)
        if constexpr (MODE == FLUX_SSC) {  // SSC tables (vag_ic_kernels.h)
            const double* hdr = a.ichdr + (size_t)(cell0 + k) * FLUX_IC_HDR;
            const double h0 = hdr[0], h1 = hdr[1], h2 = hdr[2], h3 = hdr[3], h4 = hdr[4];
            const double* tab = a.icpool + (unsigned long long)hdr[5];
#pragma unroll
            for (int b = 0; b < NBMAX; ++b)
                if (b < NB) B[b] = ic_table_eval_hdr(tab, h0, h1, h2, h3, h4, s_band[b] - dop, &breach) + geom;
        } else {
            SpecRegs regs;
#pragma unroll
            for (int w = 0; w < 13; ++w) regs.v[w] = row[w * K + k];
            regs.v[13] = lr2;
            if constexpr (MODE == FLUX_SYN_IC) {  // synchrotron with the IC correction above the cooling break
                const double* cq = a.cellq + cell0 * FLUX_NQ + k;
                IcQ q;
                q.head(cq, K);
                bool any = false;
#pragma unroll
                for (int b = 0; b < NBMAX; ++b) any = any || (b < NB && q.applies(s_band[b] - dop));
                if (any) q.rest(cq, K);
#pragma unroll
                for (int b = 0; b < NBMAX; ++b)
                    if (b < NB) B[b] = log2_I_nu_ic_core(regs, 1, q.applies(s_band[b] - dop), q, sc, s_band[b] - dop, sp_tab) + geom;
            } else {
#pragma unroll
                for (int b = 0; b < NBMAX; ++b)  // (no band_is_dead here: its test costs the walker kernel 2.6 % and a fit's bands hold data)
                    if (b < NB) B[b] = log2_I_nu_fast(regs, 1, sc, s_band[b] - dop, sp_tab) + geom;
            }
        }
    };

#ifdef VAG_SERIES_STAMPS  // developer aid: cycles of one wavefront per part of a step
    long long c_pro = 0, c_node = 0, c_scan = 0, c_bnd = 0, c_int = 0, c_mark = __builtin_readcyclecounter();
    int n_bnd = 0;
#define VAG_FR_MARK(acc_) do { __builtin_amdgcn_s_waitcnt(0); const long long now_ = __builtin_readcyclecounter(); acc_ += now_ - c_mark; c_mark = now_; } while (0)
#else
#define VAG_FR_MARK(acc_) do { } while (0)
#endif
    double lt_a, dop_a, lr2_a, ldo_a, lt_b, dop_b, lr2_b, ldo_b;
    node(k_first, lt_a, dop_a, lr2_a, ldo_a);
    node(k_first + 1, lt_b, dop_b, lr2_b, ldo_b);
    // the EAT members of the node after next are requested a whole step before they are used
    double nG = 1, nu_ = 0, nr = 0, nteng = 1, nlr2 = 0, ngc = 0, ngs = 0, ngl = 0;
    auto request = [&](int k) {
        const int kk = k < K ? k : K - 1;
        nG = row[VP_GAMMA * K + kk], nu_ = row[VP_U * K + kk], nr = row[VP_R * K + kk], nteng = row[VP_TENG * K + kk];
        nlr2 = row[VP_LG2_R2 * K + kk];
        if constexpr (SPREAD) ngc = geo[kk], ngs = geo[K + kk], ngl = geo[2 * K + kk];
    };
    request(k_first + 2);
    // cursor into the sorted points: the first one at or beyond node 0 (a point equal to node 0 belongs to interval 0), or the
    // first one beyond node k_first when the wavefront starts inside the lattice: bisection over the 64 slots (+inf beyond n)
    int p = n;
    if (valid) {
        int lo = -1, hi = NP;  // s_tp[lo] fails, s_tp[hi] passes
        while (hi - lo > 1) {  // NP + 1 candidates
            const int mid = (lo + hi) >> 1;
            const double tm = s_tp[mid];
            if (k_first == 0 ? tm >= lt_a : tm > lt_a)
                hi = mid;
            else
                lo = mid;
        }
        p = min(hi, n);
    }
    double Bprev[NBMAX], Bcur[NBMAX];
#pragma unroll
    for (int b = 0; b < NBMAX; ++b) Bprev[b] = Bcur[b] = 0;
    long long n_ev = 0, n_in = 0;  // COUNT
    {
        const bool need0 = p < n && s_tp[p] <= lt_b;  // the first interval holds a point: node k_first is one of its ends
        if (__ballot(need0) != 0 && need0) boundary(k_first, dop_a, lr2_a, ldo_a, Bprev);
        if (COUNT && need0) n_ev += NB;
    }
    while (cut(seg + 1) <= k_first) flush(seg++);  // leading segments without an interval
    VAG_FR_MARK(c_pro);
    for (int k = k_first + 1; k <= k_last; ++k) {
        // node k + 1, one step ahead: its time tells whether node k closes the interval before a point (beyond this wavefront's
        // last node the next wavefront evaluates it as its own first)
        double lt_c = -INFINITY, dop_c = 0, lr2_c = 0, ldo_c = 0;
        if (k + 1 <= k_last) {
            lr2_c = nlr2;
            eat(nG, nu_, nr, nteng, ngc, ngs, ngl, lt_c, dop_c, ldo_c);
        }
        request(k + 2);
        VAG_FR_MARK(c_node);
        // the lane's points inside (t[k-1], t[k]]: the next four at once (the times ascend, so the count is the number of
        // leading ones that fit); more than four is rare
        double tn[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) tn[q] = p + q < n ? s_tp[min(p + q, NP - 1)] : INFINITY;
        int cnt = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) cnt += tn[q] <= lt_b ? 1 : 0;
        int pe = p + cnt;
        double t_next = cnt == 0 ? tn[0] : cnt == 1 ? tn[1] : cnt == 2 ? tn[2] : cnt == 3 ? tn[3] : tn[4];
        if (cnt == 4) {
            while (pe < n && s_tp[pe] <= lt_b) ++pe;
            t_next = pe < n ? s_tp[pe] : INFINITY;
        }
        const bool need = pe > p || t_next <= lt_c;
        VAG_FR_MARK(c_scan);
        if (COUNT) n_ev += need ? NB : 0, n_in += pe - p;
        if (__ballot(need) != 0) {
            if (need) boundary(k, dop_b, lr2_b, ldo_b, Bcur);
#ifdef VAG_SERIES_STAMPS
            ++n_bnd;
#endif
            VAG_FR_MARK(c_bnd);
            if (__ballot(pe > p) != 0) {
                const double inv_dt = 1.0 / (lt_b - lt_a);
                // two points per turn: their interpolation chains (selects, slope, 2^x) are independent and interleave
                for (int q = p; q < pe; q += 2) {
                    const int q1 = min(q + 1, pe - 1);
                    const int b0 = s_band_of[q], b1 = s_band_of[q1];
                    const double t0 = s_tp[q], t1 = s_tp[q1];
                    double lo0 = Bprev[0], hi0 = Bcur[0], lo1 = Bprev[0], hi1 = Bcur[0];
#pragma unroll
                    for (int bb = 1; bb < NBMAX; ++bb) {
                        lo0 = b0 == bb ? Bprev[bb] : lo0;
                        hi0 = b0 == bb ? Bcur[bb] : hi0;
                        lo1 = b1 == bb ? Bprev[bb] : lo1;
                        hi1 = b1 == bb ? Bcur[bb] : hi1;
                    }
                    const double sl0 = (hi0 - lo0) * inv_dt, sl1 = (hi1 - lo1) * inv_dt;
                    const double v0 = exp2_fast(lo0 + (t0 - lt_a) * sl0), v1 = exp2_fast(lo1 + (t1 - lt_a) * sl1);
                    if (isfinite(sl0)) lds_add_f64(my_acc + q, v0);
                    if (q + 1 < pe && isfinite(sl1)) lds_add_f64(my_acc + q + 1, v1);
                }
            }
#pragma unroll
            for (int b = 0; b < NBMAX; ++b) Bprev[b] = Bcur[b];
            VAG_FR_MARK(c_int);
        }
        p = pe;
        lt_a = lt_b, lt_b = lt_c, dop_b = dop_c, lr2_b = lr2_c, ldo_b = ldo_c;
        while (seg < seg_end && cut(seg + 1) <= k) flush(seg++);  // step k closes segment(s)
    }
    if constexpr (MODE == FLUX_SSC) {
        if (breach) atomicOr(a.ic_status + m, ic_breach_status(breach));
    }
    if constexpr (COUNT) {
        if (valid && a.tally) {
            atomicAdd(a.tally, (unsigned long long)n_ev);
            atomicAdd(a.tally + 1, (unsigned long long)n_in);
        }
    }
#ifdef VAG_SERIES_STAMPS
    if (m == 0 && vb == 0 && lane == 0 && wseg == 0)
        printf("fit rows wave 0: K %d  cycles: prologue %lld  nodes %lld  scan %lld  boundary %lld (%d steps)  interp %lld\n", K, c_pro, c_node,
               c_scan, c_bnd, n_bnd, c_int);
#endif
}

// Persistent workgroups: the launch fills the GPU once (or holds as many wavefronts as there are items, if fewer) and every wavefront
// takes items -- (model, block of 64 rows, lattice segment) in the batch's order, most expensive models first -- until none is left.  A grid of one workgroup per four blocks had a workgroup's four wavefront slots wait for a successor every ~100 us of work
// (2.1-2.4 of 3 wavefronts per SIMD resident, `profiles/debug/uniform_pmc.sh`), and launched max-blocks-of-any-model workgroups for every
// model (two thirds of them empty on a ragged walker batch).  Which wavefront serves an item does not enter its partial sum.
// a.grid_nt carries W = wavefronts per block of 64 rows (1, 2 or 4), the launch's choice.  NBMAX = 4 or 8 bounds the bands held in
// registers per node.
// SPREAD: a spreading jet's polar angle evolves along the lattice, so the viewing cosine and the solid angle are per node
// (calc_t_obs + calc_solid_angle, observer.cpp:51-141; a.cellgeo holds cos theta, sin theta, log2|dcos| per cell).
template <int MODE, int NBMAX, bool SPREAD = false, bool COUNT = false>
__global__ void __launch_bounds__(SERIES_THREADS * FITROWS_WAVES, VAG_ROWS_MIN_WG)  // 168 VGPRs: three wavefronts per SIMD (170 would leave two)
vag_flux_fit_rows_kernel(SeriesArgs a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* s_sp = lds;
    const int n = a.n, NP = fit_rows_npad(n), stripes = fit_rows_stripes(n), NS = rows_acc_stride(NP, stripes);
    double* s_tp = s_sp + SP_LDS_DOUBLES;
    double* s_band = s_tp + NP;  // [FITROWS_WAVES][SERIES_MAX_BANDS]
    double* s_acc = s_band + FITROWS_WAVES * SERIES_MAX_BANDS + (size_t)wave * stripes * NS;
    int* s_band_of = (int*)(s_band + FITROWS_WAVES * SERIES_MAX_BANDS + (size_t)FITROWS_WAVES * stripes * NS);
    for (int i = threadIdx.x; i < SP_LDS_DOUBLES; i += blockDim.x) s_sp[i] = a.sp_table[i];
    for (int i = threadIdx.x; i < NP; i += blockDim.x) {
        s_tp[i] = i < n ? a.lg2_t_obs[i] : INFINITY;
        s_band_of[i] = i < n ? a.band_idx[i] : 0;
    }
    for (int i = lane; i < stripes * NS; i += SERIES_THREADS) s_acc[i] = 0;
    __syncthreads();  // the only workgroup-wide barrier
    int m_lo = 0;  // a wavefront's items ascend
    // the first item of a wavefront is its own number, the later ones come from the counter (one device-wide atomic per item: ~7 ns
    // each on one address, so the 3072 simultaneous first fetches of a launch were 25 us of a 128-walker call); a launch with no more
    // items than wavefronts never touches it
    int item = (int)blockIdx.x * FITROWS_WAVES + wave;
    for (;; ) {
        // (r05, as vag_flux_grid_rows_kernel<0>: a turn re-reads the arguments through an opaque kernel-argument pointer and makes the
        // lane / wavefront numbers opaque, so that the loop keeps the item number alive and not forty scalars)
#if !defined(VAG_HOST_DEBUG) && !defined(VAG_FIT_NO_REREAD)
        const SeriesArgs A = load_series_args();
        int lane_i = threadIdx.x & 63, wave_i = threadIdx.x >> 6;
        asm volatile("" : "+v"(lane_i), "+v"(wave_i));
        wave_i = __builtin_amdgcn_readfirstlane(wave_i);
#else
        const SeriesArgs& A = a;
        const int lane_i = lane, wave_i = wave;
#endif
        const FitRowsLds L{s_sp, s_tp, s_band_of, s_band + wave_i * SERIES_MAX_BANDS,
                           s_band + FITROWS_WAVES * SERIES_MAX_BANDS + (size_t)wave_i * stripes * NS};
        const int W = A.grid_nt, nb = A.nb;
        const int* __restrict__ blk_off = A.lay.row_off + nb + 1;  // [nb + 1] first block of every model (vag_grid_kernel's plan scan)
        const int total_items = blk_off[nb] * W;
        const int n_waves = (int)gridDim.x * FITROWS_WAVES;
        if (item >= total_items) break;
        const int blk = item / W, wseg = item - blk * W;
        int lo = m_lo, hi = nb;  // blk_off[lo] <= blk < blk_off[hi]: the model is the last one that starts at or before the block
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (blk_off[mid] <= blk)
                lo = mid;
            else
                hi = mid;
        }
        m_lo = lo;
        fit_rows_item<MODE, NBMAX, SPREAD, COUNT>(A, L, lo, blk - blk_off[lo], wseg, W, lane_i);
        if (total_items <= n_waves) break;
        if (lane_i == 0) item = n_waves + __hip_atomic_fetch_add(A.work, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        item = __builtin_amdgcn_readfirstlane(item);
    }
    // (the counter is put back to zero by the reduction kernel that follows every launch of this one)
}

}  // namespace vag
