// vag_grid_rows.h -- small (nu, t) grids of large batches with ONE (theta, phi) ROW PER LANE (the grid-request counterpart of
// vag_fit_rows.h).
//
// vag_flux_grid_kernel gives a row to a 256-lane workgroup: at four frequencies and a hundred times a row fills 22-78 % of the
// lanes per phase and pays two workgroup barriers, and the row skeleton (EAT logs, bracket search, barriers) has come to outweigh
// the spectra (~10 of 17-23 ms per pass on the C5 shape).  Here a wavefront takes 64 consecutive rows, one per
// lane, and all lanes walk the lattice nodes together: node times one node ahead, a cursor into the ascending requested times,
// the node's boundary values for the request's frequencies where a lane has times next to it, log-log interpolation of the
// lane's times in the interval (Observer::specific_flux, observer.h:355-445: t_row[k] <= t < t_row[k+1]), added to the
// wavefront's (nu, t) sums in LDS with ds_add_f64.  No staged rows, no barriers after the tables are loaded.
//
// Used when the batch has blocks of 64 rows enough to fill the GPU (a lone model would walk its lattice as one sequential chain)
// and the grid has at most 512 (nu, t) slots in at most four frequencies; everything else stays with vag_flux_grid_kernel.
#pragma once
#include "vag_fit_rows.h"
#include "vag_ic_kernels.h"

namespace vag {

#ifndef VAG_GRIDROWS_WAVES
#define VAG_GRIDROWS_WAVES 4
#endif
constexpr int GRIDROWS_WAVES = VAG_GRIDROWS_WAVES;
constexpr int GRIDROWS_BANDS = 4;     // frequencies
#ifndef VAG_GRIDROWS_MAX_NT
#define VAG_GRIDROWS_MAX_NT 128
#endif
constexpr int GRIDROWS_MAX_NT = VAG_GRIDROWS_MAX_NT;  // requested times
constexpr int GRIDROWS_MAX_SLOTS = 512;

// (Build option VAG_ROWS_RING=1, measured slower and off by default -- DESIGN.md 4h.)  With it the interpolation work of a wavefront
// goes through a RING of items in LDS: a lane pushes {the four exponents of one requested
// time inside its current lattice interval, the time's index}, and whenever 64 items have gathered ALL 64 lanes pop one each and do
// the exp2 + accumulate.  Walking the lattice, the lanes of a wavefront meet their requested times at different nodes (0, 1 or 2 per
// interval, 0.9 on average on the C5 shape), so the interpolation done in place ran at the trip count of the busiest lane with
// ~45 % of the lanes active: 22 of the SSC pass's 31 ms and 23 of the synchrotron pass's 52 ms per 1024 C5 members
// (profiles/r04_rows_ablation.txt).  A push is ~15 instructions under that mask; the expensive part now always runs on full lanes.
#ifndef VAG_ROWS_RING
#define VAG_ROWS_RING 0  // measured SLOWER (C5: 71 / 39 ms per pass against 52 / 31): see DESIGN.md 4h; kept as a build option
#endif
constexpr int GRIDROWS_RING = 128;  // items: a push pass adds at most 64 to at most 63 left over
constexpr int GRIDROWS_RING_BYTES = VAG_ROWS_RING ? GRIDROWS_RING * (2 * 16 + 4) : 0;  // {x0, x1}, {x2, x3}, time index

// copies of a wavefront's sums (vag_fit_rows.h: why, and how they are laid out): four for short requests, else three where three
// workgroups per CU still fit the 160 KB with them (C5's 4 x 100 slots: 24.3 against 25.2 ms per tabulated-SSC pass with two), else two
__host__ __device__ inline size_t grid_rows_lds_bytes_with(int slots, int stripes) {
    return sizeof(double) * (SP_LDS_DOUBLES + GRIDROWS_MAX_NT + SERIES_MAX_BANDS + (size_t)GRIDROWS_WAVES * stripes * rows_acc_stride(slots, stripes)) +
           (size_t)GRIDROWS_WAVES * GRIDROWS_RING_BYTES;
}
__host__ __device__ inline int grid_rows_stripes(int slots) {
#ifdef VAG_GRIDROWS_STRIPES_BIG
    return slots <= 128 ? 4 : VAG_GRIDROWS_STRIPES_BIG;
#else
    return slots <= 128 ? 4 : (3 * grid_rows_lds_bytes_with(slots, 3) <= 160 * 1024 ? 3 : 2);
#endif
}
__host__ __device__ inline size_t grid_rows_lds_bytes(int slots) { return grid_rows_lds_bytes_with(slots, grid_rows_stripes(slots)); }

// LDS of a workgroup as a block of rows sees it
struct GridRowsLds {
    const double* s_sp;    // softplus + log2 tables
    const double* s_tobs;  // [GRIDROWS_MAX_NT] log2 requested times, ascending; +inf beyond nt
    double* s_acc;         // this wavefront's sums [stripe][SS]
    char* ring_base;       // this wavefront's ring (VAG_ROWS_RING builds)
};

// One block of 64 rows (block vb of model m), by one wavefront.
template <int MODE>
VAG_DEV void grid_rows_item(const SeriesArgs& a, const GridRowsLds& L, int m, int vb, int lane) {
    const VagGridMeta* Mp = a.meta + m;
    const int n_pairs = Mp->n_theta * Mp->n_phi_eff;
    const int nt = a.grid_nt, NB = a.n_bands, slots = a.n;
    const int stripes = grid_rows_stripes(slots), SS = rows_acc_stride(slots, stripes);  // (the copies' stride: vag_fit_rows.h)
    const double* s_sp = L.s_sp;
    const double* s_tobs = L.s_tobs;
    double* s_acc = L.s_acc;
    double* my_acc = s_acc + (lane % stripes) * SS;
    [[maybe_unused]] vdouble2* ring_x01 = reinterpret_cast<vdouble2*>(L.ring_base);
    [[maybe_unused]] vdouble2* ring_x23 = ring_x01 + GRIDROWS_RING;
    [[maybe_unused]] int* ring_q = reinterpret_cast<int*>(ring_x23 + GRIDROWS_RING);
    [[maybe_unused]] int ring_tail = 0, ring_count = 0;  // wavefront-uniform
    wave_sync();  // (a previous block's sums have been read)
    for (int i = lane; i < stripes * SS; i += SERIES_THREADS) s_acc[i] = 0;
    wave_sync();
    const int p0 = vb * FITROWS_ROWS;
    const LdsTab sp_tab = lds_tab(s_sp), lg_tab = lds_tab(s_sp + SP_TABLE_DOUBLES);
    // log2 nu (1 + z) of the request's frequencies, wavefront-uniform: scalar loads into scalar registers (read back from LDS they
    // cost the node loop four round trips with a wait each)
    double nu_b[GRIDROWS_BANDS];
#pragma unroll
    for (int b = 0; b < GRIDROWS_BANDS; ++b) nu_b[b] = a.lg2_nu_obs[b < NB ? b : 0] + Mp->lg2_1pz;
    const int K = Mp->n_t, n_phi_eff = Mp->n_phi_eff;
    const double one_plus_z = 1 + a.params[m].z;
    SpecConst sc;
    sc.init_fast(a.params[m].p, lg_tab);
    int breach = 0;

    // this lane's row
    const bool valid = p0 + rows_lane_row(lane) < n_pairs;
    const int pair = valid ? p0 + rows_lane_row(lane) : n_pairs - 1;
    const int j = pair / n_phi_eff, i = pair - j * n_phi_eff;
    const double* gth = a.geo_th + (size_t)m * 3 * Mp->th_stride;
    const double* gph = a.geo_ph + (size_t)m * 2 * Mp->ph_stride;
    const int rep = a.g_rep_of[(size_t)m * Mp->th_stride + j] + i * Mp->rep_phi_stride;
    const double cos_v = gth[Mp->th_stride + j] * gph[i] * Mp->sin_obs + gth[j] * Mp->cos_obs;
    const double t_coeff = (1 - cos_v) / C_C * one_plus_z;
    const double lg2_dOmega = gth[2 * Mp->th_stride + j] + gph[Mp->ph_stride + i];
    const long long cell0 = a.lay.cell_off[m] + (long long)rep * K;
    const double* row = a.cellpar + cell0 * VAG_NPAR;  // [VAG_NPAR][K]

    auto eat = [&](double G, double u, double r, double teng, double& lt, double& dop) {
        dop = -log2_tab(fma(-u, cos_v, G), lg_tab);
        lt = log2_tab(fma(t_coeff, r, teng * one_plus_z), lg_tab);
    };
    auto node = [&](int k, double& lt, double& dop, double& lr2) {
        lr2 = row[VP_LG2_R2 * K + k];
        eat(row[VP_GAMMA * K + k], row[VP_U * K + k], row[VP_R * K + k], row[VP_TENG * K + k], lt, dop);
    };
    // boundary values B[l] = log2 I'(nu_l (1+z) / D_k) + log2(dOmega r^2 D^3) of node k
    auto boundary = [&](int k, double dop, double lr2, double (&B)[GRIDROWS_BANDS]) {
        const double geom = (lg2_dOmega + lr2) + 3.0 * dop;
        if constexpr (MODE == FLUX_SSC) {  // (only node 0: the loop below keeps the later nodes' look-ups in flight a node ahead)
            const double* hdr = a.ichdr + (size_t)(cell0 + k) * FLUX_IC_HDR;
            const double h0 = hdr[0], h1 = hdr[1], h2 = hdr[2], h3 = hdr[3], h4 = hdr[4];
            const double* tab = a.icpool + (unsigned long long)hdr[5];
#pragma unroll
            for (int b = 0; b < GRIDROWS_BANDS; ++b)
                if (b < NB) B[b] = ic_table_eval_hdr(tab, h0, h1, h2, h3, h4, nu_b[b] - dop, &breach) + geom;
        } else {
            SpecRegs regs;
#pragma unroll
#ifdef VAG_ROWS_ABLATE_LOADS  // timing experiment only: the node's constants without their memory round trip
            for (int w = 0; w < 13; ++w) regs.v[w] = row[w * K] + 1e-300 * k;
#else
            for (int w = 0; w < 13; ++w) regs.v[w] = row[w * K + k];
#endif
            regs.v[13] = lr2;
            if constexpr (MODE == FLUX_SYN_IC) {
                const double* cq = a.cellq + cell0 * FLUX_NQ + k;
                IcQ q;
                q.head(cq, K);
                bool any = false;
#pragma unroll
                for (int b = 0; b < GRIDROWS_BANDS; ++b) any = any || (b < NB && q.applies(nu_b[b] - dop));
                if (any) q.rest(cq, K);
#pragma unroll
                for (int b = 0; b < GRIDROWS_BANDS; ++b)
                    if (b < NB && !band_is_dead(nu_b[b] - dop, regs[VP_LG2_NUMAX], B[b]))
                        B[b] = log2_I_nu_ic_core(regs, 1, q.applies(nu_b[b] - dop), q, sc, nu_b[b] - dop, sp_tab) + geom;
            } else {
#pragma unroll
                for (int b = 0; b < GRIDROWS_BANDS; ++b)
                    if (b < NB && !band_is_dead(nu_b[b] - dop, regs[VP_LG2_NUMAX], B[b]))
                        B[b] = log2_I_nu_fast(regs, 1, sc, nu_b[b] - dop, sp_tab) + geom;
            }
        }
    };

    // n_take (wavefront-uniform) items leave the ring, one per lane: exp2 and the sums (observer.h:405-433).  An item whose
    // interval has no finite slope carries a non-finite exponent and adds nothing (observer.h:422-426).
    auto ring_pop = [&](int n_take) {
        wave_sync();  // the pushes of this wavefront are in LDS (same wavefront: program order; the fence is for the compiler)
        const int idx = (ring_tail + lane) & (GRIDROWS_RING - 1);
        const vdouble2 a01 = ring_x01[idx], a23 = ring_x23[idx];
        const int q = ring_q[idx];
        if (lane < n_take) {
            const double xs[GRIDROWS_BANDS] = {a01.x, a01.y, a23.x, a23.y};
#pragma unroll
            for (int b = 0; b < GRIDROWS_BANDS; ++b)
#ifdef VAG_ROWS_ABLATE_ATOMICS  // timing experiment only: the sums kept in a register
                if (b < NB && isfinite(xs[b])) breach += exp2_fast(xs[b]) > 1e300 ? 1 : 0;
#else
                if (b < NB && isfinite(xs[b])) lds_add_f64(my_acc + b * nt + q, exp2_fast(xs[b]));
#endif
        }
        ring_tail = (ring_tail + n_take) & (GRIDROWS_RING - 1);
        ring_count -= n_take;
        wave_sync();  // the slots may be written again
    };
    double lt_a, dop_a, lr2_a, lt_b, dop_b, lr2_b;
    node(0, lt_a, dop_a, lr2_a);
    node(1, lt_b, dop_b, lr2_b);
    double nG = 1, nu_ = 0, nr = 0, nteng = 1, nlr2 = 0;  // the node after next, requested a step before it is used
    auto request = [&](int k) {
        const int kk = k < K ? k : K - 1;
        nG = row[VP_GAMMA * K + kk], nu_ = row[VP_U * K + kk], nr = row[VP_R * K + kk], nteng = row[VP_TENG * K + kk];
        nlr2 = row[VP_LG2_R2 * K + kk];
    };
    request(2);
    // FLUX_SSC: a tabulated spectrum is a header read and a gather whose address depends on the node's Doppler factor -- two
    // dependent trips to L2 per node, which three or four wavefronts per SIMD cannot hide (r03: 54 % VALU busy).  Both are taken
    // off the chain: node k + 2's header is requested with its shock state, node k + 1's gathers are issued as soon as its
    // Doppler factor exists (one iteration before the values are used), for every lane whether or not it will need the node.
    [[maybe_unused]] double hq_n = 0, hq_first = 0, hq_last = 0, hq_tmin = 0, hq_tmax = 0, hq_off = 0;  // header of the node the next issue serves
    // a pending look-up is its position inside the table interval (NaN where the reference returns -inf: no table, or beyond
    // the last node -- either makes the interval's slope non-finite, which is all the sum asks) and the gathered pair
    [[maybe_unused]] double pfrac[GRIDROWS_BANDS];
    [[maybe_unused]] vdouble2_a8 pI[GRIDROWS_BANDS];
    [[maybe_unused]] const double* hdr_next = nullptr;  // header of node min(k + 2, K - 1) while iteration k runs: walked, never multiplied out
    [[maybe_unused]] auto request_hdr = [&](const double* hdr) {
        hq_n = hdr[0], hq_first = hdr[1], hq_last = hdr[2], hq_tmin = hdr[3], hq_tmax = hdr[4], hq_off = hdr[5];
    };
    [[maybe_unused]] int p_breach = 0;  // band-contract breach of the pending look-ups: counts only if the node is then used
    [[maybe_unused]] auto issue = [&](double dop) {  // a node's look-ups from the header at hand
        const double* tab = a.icpool + (unsigned long long)hq_off;
        p_breach = 0;
#pragma unroll
        for (int b = 0; b < GRIDROWS_BANDS; ++b)
            if (b < NB) {
                const IcTabQuery q = ic_table_query(hq_n, hq_first, hq_last, hq_tmin, hq_tmax, nu_b[b] - dop, &p_breach);
                pI[b] = ic_table_gather(tab, q.idx);
                pfrac[b] = q.none ? NAN : q.frac;
            }
    };
    if constexpr (MODE == FLUX_SSC) {
        hdr_next = a.ichdr + (size_t)(cell0 + 1) * FLUX_IC_HDR;  // (K >= 2: a lattice has at least two nodes)
        request_hdr(hdr_next);
        issue(dop_b);
        hdr_next += 2 < K ? FLUX_IC_HDR : 0;
        request_hdr(hdr_next);
    }
    // cursor into the ascending requested times: the first one at or beyond node 0 (bisection over the 128 slots, +inf beyond nt)
    int p = nt;
    if (valid) {
        int lo = -1, hi = GRIDROWS_MAX_NT;  // s_tobs[lo] < lt_a <= s_tobs[hi]
#pragma unroll
        for (int it = 0; it < 8; ++it) {  // 129 candidates
            const int mid = (lo + hi + 1) >> 1;
            const bool pass = mid >= GRIDROWS_MAX_NT || s_tobs[min(mid, GRIDROWS_MAX_NT - 1)] >= lt_a;
            if (pass)
                hi = mid;
            else
                lo = mid;
        }
        p = min(hi, nt);
    }
    double Bprev[GRIDROWS_BANDS] = {0, 0, 0, 0}, Bcur[GRIDROWS_BANDS] = {0, 0, 0, 0};
    {
        const bool need0 = p < nt && s_tobs[p] < lt_b;  // the first interval holds a requested time
        if (__ballot(need0) != 0 && need0) boundary(0, dop_a, lr2_a, Bprev);
    }
    for (int k = 1; k < K; ++k) {
        double lt_c = -INFINITY, dop_c = 0, lr2_c = 0;
        if (k + 1 < K) {
            lr2_c = nlr2;
            eat(nG, nu_, nr, nteng, lt_c, dop_c);
        }
        request(k + 2);
        // the lane's requested times inside [t[k-1], t[k]): the next four at once, more is rare
        double tn[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) tn[q] = p + q < nt ? s_tobs[min(p + q, GRIDROWS_MAX_NT - 1)] : INFINITY;
        int cnt = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) cnt += tn[q] < lt_b ? 1 : 0;
        int pe = p + cnt;
        double t_next = cnt == 0 ? tn[0] : cnt == 1 ? tn[1] : cnt == 2 ? tn[2] : cnt == 3 ? tn[3] : tn[4];
        if (cnt == 4) {
            while (pe < nt && s_tobs[pe] < lt_b) ++pe;
            t_next = pe < nt ? s_tobs[pe] : INFINITY;
        }
        const bool need = pe > p || t_next < lt_c;
        if constexpr (MODE == FLUX_SSC) {
            // node k's tabulated values from the gathers issued an iteration ago, then node k + 1's look-ups take their place
            // (past the last node: a repeat of it, never used) and node k + 2's header is requested
            const double geom = (lg2_dOmega + lr2_b) + 3.0 * dop_b;
#pragma unroll
            for (int b = 0; b < GRIDROWS_BANDS; ++b)
                if (b < NB) Bcur[b] = need ? fma(pfrac[b], pI[b].y - pI[b].x, pI[b].x) + geom : Bcur[b];
            breach |= need ? p_breach : 0;
#ifndef VAG_ROWS_ABLATE_EVAL  // timing experiment only: no table look-ups
            issue(dop_c);
            hdr_next += k + 2 < K ? FLUX_IC_HDR : 0;
            request_hdr(hdr_next);
#endif
        }
        if (__ballot(need) != 0) {
            if constexpr (MODE != FLUX_SSC) {
                if (need) boundary(k, dop_b, lr2_b, Bcur);
            }
#ifdef VAG_ROWS_ABLATE_INTERP  // timing experiment only: no interpolation / accumulation at all
            if (false) {
#else
            if (__ballot(pe > p) != 0) {
#endif
                const double inv_dt = 1.0 / (lt_b - lt_a);
                double d[GRIDROWS_BANDS];
#pragma unroll
                for (int b = 0; b < GRIDROWS_BANDS; ++b) d[b] = Bcur[b] - Bprev[b];  // slope finite <=> d finite (observer.h:422-426)
#if VAG_ROWS_RING
                // push the lane's times inside the interval, one per pass (the passes are short: the trip count of the busiest
                // lane no longer multiplies the exp2 work); 64 gathered items are worked off at once
                for (int q = p; __ballot(q < pe) != 0; ++q) {
                    const bool has = q < pe;
                    const double w = (s_tobs[min(q, GRIDROWS_MAX_NT - 1)] - lt_a) * inv_dt;  // position inside the interval, shared by the frequencies
                    double x[GRIDROWS_BANDS];
#pragma unroll
                    for (int b = 0; b < GRIDROWS_BANDS; ++b) x[b] = b < NB ? fma(d[b], w, Bprev[b]) : NAN;
                    const unsigned long long mask = __ballot(has);
                    const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                    if (has) {
                        const int idx = (ring_tail + ring_count + pos) & (GRIDROWS_RING - 1);
                        ring_x01[idx] = vdouble2{x[0], x[1]};
                        ring_x23[idx] = vdouble2{x[2], x[3]};
                        ring_q[idx] = q;
                    }
                    ring_count += __popcll(mask);
                    if (ring_count >= 64) ring_pop(64);
                }
#else
                // (Measured and rejected, r04: the four frequencies' exponentials as one straight-line block so that their Horner
                // chains interleave -- 168 VGPRs with 12-20 B of scratch, C5 53.7 / 30.4 ms per pass against 51.7 / 31.0.)
                for (int q = p; q < pe; ++q) {
                    const double w = (s_tobs[q] - lt_a) * inv_dt;  // position inside the interval, shared by the frequencies
#pragma unroll
                    for (int b = 0; b < GRIDROWS_BANDS; ++b)
#ifdef VAG_ROWS_ABLATE_ATOMICS  // timing experiment only: the sums kept in a register
                        if (b < NB && isfinite(d[b])) Bcur[b] += 1e-300 * exp2_fast(fma(d[b], w, Bprev[b]));
#elif defined(VAG_ROWS_ABLATE_EXP2)  // timing experiment only: the accumulate without the exponential
                        if (b < NB && isfinite(d[b])) lds_add_f64(my_acc + b * nt + q, fma(d[b], w, Bprev[b]));
#else
                        if (b < NB && isfinite(d[b])) lds_add_f64(my_acc + b * nt + q, exp2_fast(fma(d[b], w, Bprev[b])));
#endif
                }
#endif
            }
#pragma unroll
            for (int b = 0; b < GRIDROWS_BANDS; ++b) Bprev[b] = Bcur[b];
        }
        p = pe;
        lt_a = lt_b, lt_b = lt_c, dop_b = dop_c, lr2_b = lr2_c;
    }
#if VAG_ROWS_RING
    if (ring_count > 0) ring_pop(ring_count);
#endif
    if constexpr (MODE == FLUX_SSC) {
        if (breach) atomicOr(a.ic_status + m, ic_breach_status(breach));
    }
    wave_sync();
    double* dst = a.partial + ((size_t)m * a.max_chunks + vb) * slots;
    for (int s = lane; s < slots; s += SERIES_THREADS) {
        double sum = s_acc[s];
        for (int c = 1; c < stripes; ++c) sum += s_acc[c * SS + s];
        dst[s] = sum;
    }
}

// a.n = nt * nnu slots ([l][idx], nu outer), a.grid_nt = nt, a.n_bands = nnu; partial sums [nb][max_chunks][slots], one per block of
// 64 rows.  MODE as in vag_flux_grid_kernel (FLUX_SYN / FLUX_SYN_IC / FLUX_SSC).
// FLUX_SYN is a PERSISTENT launch like vag_flux_fit_rows_kernel (the launch fills the GPU once, every wavefront takes blocks -- its own
// number first, then from the counter -- until none is left): plain-synchrotron batches are the ragged ones with short wavefront lives
// (4096 off-axis top hats: 2.3-2.4 of 3 wavefronts per SIMD resident with a workgroup per four blocks), and this instantiation has the
// registers for the loop (144 of 168 VGPRs).  The IC-corrected pass sits at 167 of 168 and pays 120-170 B of scratch for it (61 against
// 47 ms per 1024 C5 members, measured), the tabulated-SSC pass gains nothing: both keep one workgroup per four blocks.
template <int MODE>
// 168 VGPRs: three wavefronts per SIMD; the SSC pass (a table look-up per band, no spectrum constants) fits 128 with 12 B of
// scratch and gains 9 % from the fourth wavefront, the others would spill 100-200 B per lane and lose 70 %
#ifndef VAG_ROWS_SSC_WG
#define VAG_ROWS_SSC_WG 3  // three workgroups per CU (168 VGPRs, no scratch): the look-ups in flight a node ahead need the registers, 4 spills 92-160 B
#endif
__global__ void __launch_bounds__(SERIES_THREADS * GRIDROWS_WAVES, MODE == FLUX_SSC ? VAG_ROWS_SSC_WG : VAG_ROWS_MIN_WG)
vag_flux_grid_rows_kernel(SeriesArgs a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* s_sp = lds;
    if constexpr (MODE != FLUX_SYN) {
        const VagGridMeta* Mp = a.meta + blockIdx.y;
        const int n_pairs = Mp->status == 0 ? Mp->n_theta * Mp->n_phi_eff : 0;
        if ((long long)blockIdx.x * GRIDROWS_WAVES * FITROWS_ROWS >= n_pairs) return;
    }
    const int nt = a.grid_nt, slots = a.n;
    const int stripes = grid_rows_stripes(slots), SS = rows_acc_stride(slots, stripes);
    double* s_tobs = s_sp + SP_LDS_DOUBLES;           // [GRIDROWS_MAX_NT] log2 requested times, ascending; +inf beyond nt
    double* s_nu = s_tobs + GRIDROWS_MAX_NT;          // [SERIES_MAX_BANDS] (the frequencies live in scalar registers)
    for (int i = threadIdx.x; i < SP_LDS_DOUBLES; i += blockDim.x) s_sp[i] = a.sp_table[i];
    for (int i = threadIdx.x; i < GRIDROWS_MAX_NT; i += blockDim.x) s_tobs[i] = i < nt ? a.lg2_t_obs[i] : INFINITY;
    __syncthreads();  // the only workgroup-wide barrier
    if constexpr (MODE != FLUX_SYN) {
        const int m = blockIdx.y, vb = blockIdx.x * GRIDROWS_WAVES + wave;
        const VagGridMeta* Mp = a.meta + m;
        if (vb * FITROWS_ROWS >= Mp->n_theta * Mp->n_phi_eff) return;
        // this wavefront's sums, and its ring behind every wavefront's sums (16-byte aligned: the doubles before it are an even count)
        const GridRowsLds L{s_sp, s_tobs, s_nu + SERIES_MAX_BANDS + (size_t)wave * stripes * SS,
                            reinterpret_cast<char*>(s_nu + SERIES_MAX_BANDS + (size_t)GRIDROWS_WAVES * stripes * SS) + (size_t)wave * GRIDROWS_RING_BYTES};
        grid_rows_item<MODE>(a, L, m, vb, lane);
    } else {
        // The loop keeps nothing alive but the block number: a turn re-reads the arguments and makes the lane / wavefront numbers opaque,
        // so that what a block derives from them is formed inside the turn instead of being held in registers across the loop.
        int item = (int)blockIdx.x * GRIDROWS_WAVES + wave;
        for (;;) {
#ifndef VAG_HOST_DEBUG
            const SeriesArgs A = load_series_args();
            int lane_i = threadIdx.x & 63, wave_i = threadIdx.x >> 6;
            asm volatile("" : "+v"(lane_i), "+v"(wave_i));
            wave_i = __builtin_amdgcn_readfirstlane(wave_i);
#else
            const SeriesArgs& A = a;
            const int lane_i = lane, wave_i = wave;
#endif
            const int nb = A.nb;
            const int* __restrict__ blk_off = A.lay.row_off + nb + 1;  // [nb + 1] first block of every model (vag_grid_kernel's plan scan)
            const int total_items = blk_off[nb];
            if (item >= total_items) break;
            int lo = 0, hi = nb;  // blk_off[lo] <= item < blk_off[hi]
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (blk_off[mid] <= item)
                    lo = mid;
                else
                    hi = mid;
            }
            const GridRowsLds L{s_sp, s_tobs, s_nu + SERIES_MAX_BANDS + (size_t)wave_i * stripes * SS,
                                reinterpret_cast<char*>(s_nu + SERIES_MAX_BANDS + (size_t)GRIDROWS_WAVES * stripes * SS) + (size_t)wave_i * GRIDROWS_RING_BYTES};
            grid_rows_item<MODE>(A, L, lo, item - blk_off[lo], lane_i);
            const int n_waves = (int)gridDim.x * GRIDROWS_WAVES;
            if (total_items <= n_waves) break;
            if (lane_i == 0) item = n_waves + __hip_atomic_fetch_add(A.work, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            item = __builtin_amdgcn_readfirstlane(item);
        }
        // (the counter is put back to zero by the reduction kernel that follows every launch of this one)
    }
}

}  // namespace vag
