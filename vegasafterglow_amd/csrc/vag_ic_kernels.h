// vag_ic_kernels.h -- kernels of the SSC / inverse-Compton tier.
#pragma once
#include <cstddef>
#include <type_traits>

#include "vag_ic.h"
#include "vag_kernels.h"

namespace vag {

// Rows of the per-cell electron/photon detail arrays (SoA over cells); 0..10 are also what Model.details exposes
// (pybind/pymodel.h:490-535).
enum {
    VD_GAMMA_M = 0, VD_GAMMA_C, VD_GAMMA_A, VD_GAMMA_MAX, VD_N_E, VD_COLUMN_DEN, VD_NU_M, VD_NU_C, VD_NU_A, VD_NU_MAX,
    VD_I_NU_MAX, VD_YC, VD_REGIME, VAG_NDET
};

// Extra per-cell parameters of the IC-corrected synchrotron spectrum: [row][VAG_NQ][n_t]
enum {
    VQ_LG2_NUC = 0, VQ_L1PYC, VQ_HASIC, VQ_LG2_KB, VQ_NSEG,
    VQ_S0, VQ_L0, VQ_C0, VQ_S1, VQ_L1, VQ_C1, VQ_S2, VQ_L2, VQ_C2,
    VAG_NQ
};

// SSC table of one cell: header + log2 I on the phase-locked output lattice (inverse-compton.h:354-369,595-606)
constexpr int IC_MAX_OUT = 192;  // three output nodes per lane; an unclamped table needs at most ~(IC_MAX_NU - 1) + (IC_MAX_G - 1) + 3
// SSC tables in HBM (r04): a header per cell -- {n (0: no table, -1: cell never queried), first node, last node, log2 of the theoretical
// minimum / maximum, offset of the table in the pool [doubles], 2 spare} -- and ONE pool that holds the tables back to back, each as
// long as its own output lattice (~70 nodes where the fixed layout reserved 192, none for the cells no row queries: 9 GB -> ~3 GB on
// the 512-model configs[2] batch).  The offsets are handed out by vag_ic_plan_kernel (one atomic reservation per workgroup).
constexpr int IC_HDR = 8;
enum { ICH_N = 0, ICH_FIRST, ICH_LAST, ICH_TMIN, ICH_TMAX, ICH_OFF };
constexpr int IC_PLAN = 24;  // per-cell record between vag_ic_plan_kernel and vag_ic_photon_kernel (ICP_* words): everything the wavefront needs of its cell but the 46 spectrum constants
#ifndef VAG_IC_MAX_NU
#define VAG_IC_MAX_NU 128  // (developer builds: 96 makes a cell's LDS 7.75 KB, i.e. five wavefronts per SIMD)
#endif
constexpr int IC_MAX_NU = VAG_IC_MAX_NU, IC_MAX_G = 64, IC_MAX_LAT = (IC_MAX_G - 1) + (IC_MAX_NU - 1) + 1;
static_assert(VAG_NQ == FLUX_NQ && IC_HDR == FLUX_IC_HDR, "keep vag_kernels.h forward constants in sync");
constexpr double IC_Q = 3.321928094887362 / 8;  // lattice_quantum
constexpr double IC_X0 = 0.47140452079103166;

// ------------------------------------------------------------------------------------------------
// IC cooling (IC_cooling, inverse-compton.h:729-768; update_gamma_c_Thomson/_KN and update_gamma_M,
// inverse-compton.cpp:192-251) in two kernels.  Cells of a row are tied together by ONE number: each cell's gamma_c fixed point
// starts from the previous cell's cooled gamma_c (and stops on a relative step, so the start value decides the iterate it stops
// at).  vag_ic_cooling_kernel walks that chain, one lane per representative row -- a pure latency chain, so it carries nothing
// else: per cell it leaves gamma_c, the gamma_c its last Y(gamma) was built for and the Thomson Y of that iteration.  Everything
// that follows from them per cell -- Y(gamma) itself, gamma_M's fixed point, Y_c, gamma_a, the regime, the stored segments --
// is done one lane per CELL at the head of vag_photons_ic_kernel.
// ------------------------------------------------------------------------------------------------
// gamma_M's fixed point of one cell (update_gamma_M, inverse-compton.cpp:236-251)
VAG_DEV double ic_gamma_M(double B, double gM, const IcY& Ys) {
    if (B == 0) return INFINITY;
    double gM_new = gamma_M_of(B, Ys.gamma_spectrum(gM));
    for (int guard = 0; fabs((gM - gM_new) / gM_new) > 1e-3 && guard < 10000; ++guard) {
        gM = gM_new;
        gM_new = gamma_M_of(B, Ys.gamma_spectrum(gM));
    }
    return gM;
}

__global__ void __launch_bounds__(64)
vag_ic_cooling_kernel(const vag_model_params* __restrict__ params, int nb, const VagGridMeta* __restrict__ meta,
                      Layout lay, int n_rows, const double* __restrict__ shock, long long n_cells,
                      double* __restrict__ det,
                      const int* __restrict__ inj_idx /* optional: reverse shock's injection cutoff per row */) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows || row >= lay.row_off[nb]) return;
    const int m = find_model(lay.row_off, nb, row);
    const VagGridMeta M = meta[m];
    if (M.status != 0) return;
    const vag_model_params P = params[m];
    const bool kn = (P.flags & VAG_FLAG_KN) != 0;
    const double e_over_B = P.eps_e / P.eps_B;
    const int nt = M.n_t;
    const long long c0 = lay.cell_off[m] + (long long)(row - lay.row_off[m]) * nt;
    double gamma_c_last = det[VD_GAMMA_C * n_cells + c0];
    const int k_inj = inj_idx ? inj_idx[row] : nt;
    double inj_gc = 0, inj_gm = 1, inj_gM = 0;  // cooled electrons of the crossing cell k_inj - 1
    // the cell's inputs one cell ahead of the chain
    double n_t_com = shock[VS_TCOMV * n_cells + c0], n_B = shock[VS_B * n_cells + c0];
    double n_gm = det[VD_GAMMA_M * n_cells + c0], n_gc = gamma_c_last;
    for (int k = 0; k < nt; ++k) {
        const long long c = c0 + k;
        const double t_com = n_t_com, B = n_B, gm = n_gm;
        double gc = n_gc;
        if (k + 1 < nt) {
            n_t_com = shock[VS_TCOMV * n_cells + c + 1], n_B = shock[VS_B * n_cells + c + 1];
            n_gm = det[VD_GAMMA_M * n_cells + c + 1], n_gc = det[VD_GAMMA_C * n_cells + c + 1];
        }
        double gc_used, Y_T;  // what the cell's Y(gamma) is built from
        IcY Ys;
        Ys.init_base(gm, P.p, B);
        const double lg2_gm = Ys.lg2_gamma_m;
        if (kn) {
            double gc_new = gamma_c_last;  // (the constructor's own update with this start value is overwritten by the first pass)
            int iter = 0;
            do {
                gc = gc_new;
                const double lg2_gc = log2_fast(gc);
                Y_T = thomson_Y_lg(e_over_B, P.p, gc < gm, lg2_gc - lg2_gm);
                Ys.update_cooling_breaks_lg(gc, lg2_gc, Y_T);
                gc_new = gamma_c_of(t_com, B, Ys.gamma_spectrum_lg(lg2_gc));
                iter++;
            } while (fabs((gc_new - gc) / gc) > 1e-3 && iter < 100);
            gc_used = gc;
            gc = gc_new;
        } else {
            Y_T = thomson_Y_lg(e_over_B, P.p, gc < gm, log2_fast(gc) - lg2_gm);
            double gc_new = gamma_c_last;
            for (int guard = 0; fabs((gc_new - gc) / gc) > 1e-3 && guard < 10000; ++guard) {
                gc = gc_new;
                Y_T = thomson_Y_lg(e_over_B, P.p, gc < gm, log2_fast(gc) - lg2_gm);
                gc_new = gamma_c_of(t_com, B, Y_T);
            }
            gc = gc_new;
            gc_used = gc;
        }
        if (k >= k_inj) {  // cool_relic_electrons inside IC_cooling, inverse-compton.h:752
            gc = cool_after_crossing(inj_gc, inj_gm, gm);
            det[VD_GAMMA_MAX * n_cells + c] = cool_after_crossing(inj_gM, inj_gm, gm);
        }
        if (k == k_inj - 1) {  // the crossing cell's own gamma_M is what the relic cells scale
            if (!kn) Ys.init(gm, gc_used, P.p, B, Y_T, false);
            inj_gc = gc;
            inj_gm = gm;
            inj_gM = ic_gamma_M(B, det[VD_GAMMA_MAX * n_cells + c], Ys);
        }
        det[VD_GAMMA_C * n_cells + c] = gc;
        det[VD_GAMMA_A * n_cells + c] = gc_used;  // } until vag_photons_ic_kernel puts gamma_a and Y_c there
        det[VD_YC * n_cells + c] = Y_T;           // }
        gamma_c_last = gc;
    }
}

// ------------------------------------------------------------------------------------------------
// One lane per cell: the rest of IC_cooling from what the chain left (see above), then the photons from the cooled electrons
// (generate_syn_photons + build) and the constants of the IC correction of the thin branch (inverse_compton_correction,
// inverse-compton.h:781-792).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
vag_photons_ic_kernel(const vag_model_params* __restrict__ params, int nb, const VagGridMeta* __restrict__ meta,
                      Layout lay, const double* __restrict__ shock, long long n_cells, double* __restrict__ det,
                      double* __restrict__ icy, double* __restrict__ cellpar, double* __restrict__ cellq,
                      const int* __restrict__ inj_idx /* optional: reverse shock's injection cutoff per row */) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cells || c >= lay.cell_off[nb]) return;  // n_cells is the arrays' stride (>= the batch's cell count)
    const int m = cell_model(lay.cell_off, nb, c);
    const VagGridMeta M = meta[m];
    if (M.status != 0) return;
    const int nt = M.n_t;
    const long long local = c - lay.cell_off[m];
    const int r = (int)(local / nt), k = (int)(local % nt);
    const vag_model_params P = params[m];
    const double B = shock[VS_B * n_cells + c];
    const double gm = det[VD_GAMMA_M * n_cells + c], gc = det[VD_GAMMA_C * n_cells + c], cd = det[VD_COLUMN_DEN * n_cells + c];
    IcY Ys;
    Ys.init(gm, det[VD_GAMMA_A * n_cells + c], P.p, B, det[VD_YC * n_cells + c], (P.flags & VAG_FLAG_KN) != 0);
    const int k_inj = inj_idx ? inj_idx[lay.row_off[m] + r] : nt;
    double gM = det[VD_GAMMA_MAX * n_cells + c];  // relic cells: already the crossing cell's, scaled by the chain
    if (k < k_inj) gM = ic_gamma_M(B, gM, Ys);
    const double Y_c = Ys.gamma_spectrum(gc);
    const double ga = syn_gamma_a_ic(B, syn_I_peak(B, cd), gm, gc, P.p, Ys, Y_c);
    det[VD_GAMMA_MAX * n_cells + c] = gM;
    det[VD_GAMMA_A * n_cells + c] = ga;
    det[VD_YC * n_cells + c] = Y_c;
    det[VD_REGIME * n_cells + c] = (double)determine_regime(ga, gc, gm);
    icy_store(Ys, icy, n_cells, c);
    CellOut o;
    syn_photons_build(o, gm, gc, ga, gM, cd, det[VD_N_E * n_cells + c], B, P.p, shock[VS_GAMMA * n_cells + c],
                      shock[VS_R * n_cells + c], shock[VS_TENG * n_cells + c]);
    double* dst = cellpar + (lay.cell_off[m] + (long long)r * nt) * VAG_NPAR + k;
#pragma unroll
    for (int q = 0; q < VAG_NPAR; ++q) dst[(long long)q * nt] = o.par[q];
    det[VD_NU_M * n_cells + c] = o.nu_m;
    det[VD_NU_C * n_cells + c] = o.nu_c;
    det[VD_NU_A * n_cells + c] = o.nu_a;
    det[VD_NU_MAX * n_cells + c] = o.nu_M;
    det[VD_I_NU_MAX * n_cells + c] = o.I_nu_max;
    const double Y_T = Ys.Y_T;
    double* q = cellq + (lay.cell_off[m] + (long long)r * nt) * VAG_NQ + k;
    q[(long long)VQ_LG2_NUC * nt] = log2(o.nu_c);
    q[(long long)VQ_L1PYC * nt] = log2(1. + Y_c);
    q[(long long)VQ_HASIC * nt] = (Y_c > 0 || Y_T > 0) ? 1.0 : 0.0;
    q[(long long)VQ_LG2_KB * nt] = log2(4 * C_PI * C_ME * C_C / (3 * C_E)) - log2(B);
    q[(long long)VQ_NSEG * nt] = (double)Ys.seg.size;
#pragma unroll
    for (int s = 0; s < 3; ++s) {  // as icy_store lays them out
        q[(long long)(VQ_S0 + 3 * s) * nt] = s < Ys.seg.size ? Ys.seg.slope[s] : 0;
        q[(long long)(VQ_S0 + 3 * s + 1) * nt] = s < Ys.seg.size ? Ys.seg.lg2_lower[s] : INFINITY;
        q[(long long)(VQ_S0 + 3 * s + 2) * nt] = s < Ys.seg.size ? Ys.seg.lg2_const[s] : 0;
    }
}

// IC-corrected synchrotron spectrum (compute_log2_spectrum, smooth-power-law-syn.cpp:80-92).  `c`/`st` address the
// 18-parameter block, `qv`/`qst` the IC extras of the same cell.  log2((1+Y_c)/(1+Y(nu))) = log2(1+Y_c) - sp(log2 Y).
// The cell's IC constants are requested in two batches -- (nu_c, has-IC) decide whether the correction applies; the other twelve
// are loaded together, not one per comparison, when it does -- so that an evaluation above the cooling break pays two memory
// round trips, not five (the constants come from L2 in the flux kernels).
struct IcQ {
    double lg2_nuc, has;
    double l1pyc, lg2_kb, nseg, c0, s0, l1, c1, s1, l2, c2, s2;
    template <class P2>
    VAG_DEV void head(const P2 qv, int qst) {
        lg2_nuc = qv[VQ_LG2_NUC * qst];
        has = qv[VQ_HASIC * qst];
    }
    template <class P2>
    VAG_DEV void rest(const P2 qv, int qst) {
        l1pyc = qv[VQ_L1PYC * qst], lg2_kb = qv[VQ_LG2_KB * qst], nseg = qv[VQ_NSEG * qst];
        c0 = qv[VQ_C0 * qst], s0 = qv[VQ_S0 * qst];
        l1 = qv[VQ_L1 * qst], c1 = qv[VQ_C1 * qst], s1 = qv[VQ_S1 * qst];
        l2 = qv[VQ_L2 * qst], c2 = qv[VQ_C2 * qst], s2 = qv[VQ_S2 * qst];
    }
    VAG_DEV bool applies(double lg2_nu) const { return lg2_nu > lg2_nuc && has != 0.0; }
};

// log2((1+Y_c)/(1+Y(nu))) = log2(1+Y_c) - sp(log2 Y) for an evaluation above the cooling break (q.rest loaded)
template <class Tab>
VAG_DEV double ic_thin_correction(const IcQ& q, double lg2_nu, Tab sp) {
    const double lg = 0.5 * (lg2_nu + q.lg2_kb);  // log2 gamma of the electrons radiating at nu
    const int n = (int)q.nseg;
    double z = q.c0 + q.s0 * lg;
    if (n > 2 && lg >= q.l2)
        z = q.c2 + q.s2 * lg;
    else if (n > 1 && lg >= q.l1)
        z = q.c1 + q.s1 * lg;
    // log2(1 + 2^z) without the reference's +-20 softplus shortcut (this term is an exact log2 there)
    const double a = fabs(z);
    const double g = a > 20.0 ? exp2_sat(-a) * LOG2E : (sp_fast(-a, sp));
    return q.l1pyc - (0.5 * (z + a) + g);
}

// IC-corrected synchrotron spectrum (compute_log2_spectrum, smooth-power-law-syn.cpp:80-92) given the thin-branch correction.
// STRAIGHT: the +-20 softplus shortcuts as selects (sp_fast_sel) -- same values; for a caller whose table sits in global memory, so
// that the three independent table reads of an evaluation are in flight together instead of one per branch.
// HAVE_NU: the caller holds nu = 2^lg2_nu already (a lattice node of vag_ic_photon_kernel) and the cut-off term takes it as it is.
template <bool STRAIGHT = false, bool HAVE_NU = false, class P1, class Tab>
VAG_DEV double log2_I_nu_ic_core(const P1 c, int st, bool corrected, const IcQ& q, const SpecConst& sc, double lg2_nu, Tab sp,
                                 double nu_val = 0.0) {
    auto sp_ = [&](double z) { return STRAIGHT ? sp_fast_sel(z, sp) : sp_fast(z, sp); };
    const double l_lo = c[VP_LG2_LO * st], l_hi = c[VP_LG2_HI * st];
    double thin = (lg2_nu - l_lo) * (1.0 / 3.0) - sp_(c[VP_DLO * st] * (lg2_nu - l_lo)) * c[VP_INV_SLO * st] -
                  sp_(c[VP_DHI * st] * (lg2_nu - l_hi)) * c[VP_INV_SHI * st];
    if (corrected) thin += ic_thin_correction(q, lg2_nu, sp);
    const double lx = lg2_nu - c[VP_LG2_NUM * st];
    double thick = 2.5 * lx;
    if (STRAIGHT) {
        const bool far = lx > sc.log2_x_far;
        const double s = -sc.smooth_thick * exp2_fast(2. / 3 * (far ? 0.0 : lx));
        const double add = sp_(-0.5 * lx + s);
        thick += far ? 0.0 : add;
    } else if (!(lx > sc.log2_x_far)) {
        const double s = -sc.smooth_thick * exp2_fast(2. / 3 * lx);
        thick += sp_(-0.5 * lx + s);
    }
    const double lb = thick + c[VP_TNORM * st];
    const double smooth_one = thin - sp_(c[VP_SAB * st] * (thin - lb)) * c[VP_INV_SAB * st];
    const double spec = c[VP_LG2_I * st] + (c[VP_INV_SLO * st] + smooth_one);
    if (HAVE_NU) return (lg2_nu - c[VP_LG2_NUMAX * st] < -20) ? spec : spec - c[VP_INV_NUMAX * st] * nu_val;
    if (lg2_nu - c[VP_LG2_NUMAX * st] < -20) return spec;
    return spec - c[VP_INV_NUMAX * st] * exp2_fast(lg2_nu);
}

// `c`/`st` address the 18-parameter block, `qv`/`qst` the IC extras of the same cell.
template <class P1, class P2, class Tab>
VAG_DEV double log2_I_nu_ic(const P1 c, int st, const P2 qv, int qst, const SpecConst& sc, double lg2_nu, Tab sp) {
    IcQ q;
    q.head(qv, qst);
    const bool corrected = q.applies(lg2_nu);
    if (corrected) q.rest(qv, qst);
    return log2_I_nu_ic_core(c, st, corrected, q, sc, lg2_nu, sp);
}
template <class P1, class P2, class Tab>
VAG_DEV double log2_I_nu_ic_straight(const P1 c, int st, const P2 qv, int qst, const SpecConst& sc, double lg2_nu, Tab sp) {
    IcQ q;
    q.head(qv, qst);
    q.rest(qv, qst);
    return log2_I_nu_ic_core<true>(c, st, q.applies(lg2_nu), q, sc, lg2_nu, sp);
}

// two frequencies on one cell (a work item of the grid flux kernel): the constants are loaded once for both
template <class P1, class P2, class Tab>
VAG_DEV void log2_I_nu_ic_pair(const P1 c, int st, const P2 qv, int qst, const SpecConst& sc, double x0, double x1, Tab sp, double& b0,
                               double& b1) {
    IcQ q;
    q.head(qv, qst);
    const bool corr0 = q.applies(x0), corr1 = q.applies(x1);
    if (corr0 || corr1) q.rest(qv, qst);
    b0 = log2_I_nu_ic_core(c, st, corr0, q, sc, x0, sp);
    b1 = log2_I_nu_ic_core(c, st, corr1, q, sc, x1, sp);
}

// ------------------------------------------------------------------------------------------------
// Per-k comoving band the observer will sample (single_shock_emission, pybind/pymodel.h:896-909): extrema of the
// Doppler factor over all (phi, theta) rows.  D^-1 = Gamma - u cos_v is monotone in cos_v, so the extrema over phi
// come from the extrema of cos_v per theta row.  One wavefront per model.
// ------------------------------------------------------------------------------------------------
// HUGE: a batch laid out with the grid kernel's third layout (more theta nodes than the LDS arrays below hold): the rows' extreme viewing
// cosines go through `cv_scratch` [nb][2][th_stride] in HBM instead.
template <bool HUGE>
__global__ void __launch_bounds__(64)
vag_ic_band_kernel(const vag_model_params* __restrict__ params, const VagGridMeta* __restrict__ meta,
                   const double* __restrict__ geo_th, const double* __restrict__ geo_ph, const int* __restrict__ g_rep_of,
                   const long long* __restrict__ cell_off, const double* __restrict__ cellpar,
                   const double* __restrict__ lg2_nu_obs, int nnu, double* __restrict__ band /* [nb][2][band_stride] */,
                   const double* __restrict__ cellgeo /* spreading jets: [rows][3][n_t], else nullptr */,
                   const int* __restrict__ unclamp /* [nb]: the model's tables span the full theoretical range */,
                   double debug_narrow /* 1, or a test's factor on the upper band edge (forces a band breach) */,
                   int band_stride /* >= the longest lattice of the batch */,
                   const double* __restrict__ tminmax /* [2] extrema of the requested observer times [s] */,
                   unsigned char* __restrict__ need /* [cells], cleared by the caller, or nullptr: every cell gets a table */,
                   double debug_need_shrink /* 1, or a test's factor on the window's upper end (forces a query of a skipped cell) */,
                   double* __restrict__ cv_scratch) {
    const int m = blockIdx.x, lane = threadIdx.x;
    const VagGridMeta M = meta[m];
    if (M.status != 0) return;
    if (unclamp[m]) {  // ICPhoton::compute_log2_I_nu's self-healing path (inverse-compton.h:626-635): nu_eval = [0, inf)
        for (int k = lane; k < M.n_t; k += 64) {
            band[((size_t)m * 2 + 0) * band_stride + k] = 0.0;
            band[((size_t)m * 2 + 1) * band_stride + k] = INFINITY;
        }
        if (need)  // (a rebuilt model keeps every table: the rebuild is the rare path)
            for (long long q = cell_off[m] + lane; q < cell_off[m + 1]; q += 64) need[q] = 1;
        return;
    }
    // A spreading jet keeps every table: its polar angle evolves along the lattice, so a row's observer times need not ascend with k
    // (a row swinging towards the line of sight arrives EARLIER at a later node), while the flux kernels place the observation window
    // by counting nodes as the reference does (observed_window, observer.h:324-338) -- the range test below would skip cells they
    // then query (found by the random sweep of spreading SSC jets, profiles/r04_sweep_final.txt; the loud status bit 4 caught it).
    if (need && cellgeo) {
        for (long long q = cell_off[m] + lane; q < cell_off[m + 1]; q += 64) need[q] = 1;
        need = nullptr;
    }
    __shared__ double s_cvmin_lds[HUGE ? 1 : VAG_MAX_THETA], s_cvmax_lds[HUGE ? 1 : VAG_MAX_THETA];
    double* s_cvmin = s_cvmin_lds;
    double* s_cvmax = s_cvmax_lds;
    if constexpr (HUGE) {
        s_cvmin = cv_scratch + (size_t)m * 2 * M.th_stride;
        s_cvmax = s_cvmin + M.th_stride;
    }
    const vag_model_params P = params[m];
    const double cos_obs = cos(P.theta_obs), sin_obs = sin(P.theta_obs);
    const double* gth = geo_th + (size_t)m * 3 * M.th_stride;
    const double* gph = geo_ph + (size_t)m * 2 * M.ph_stride;
    for (int j = lane; j < M.n_theta; j += 64) {
        double lo = INFINITY, hi = -INFINITY;
        for (int i = 0; i < M.n_phi_eff; ++i) {
            const double cv = gth[M.th_stride + j] * gph[i] * sin_obs + gth[j] * cos_obs;
            lo = fmin(lo, cv);
            hi = fmax(hi, cv);
        }
        s_cvmin[j] = lo;
        s_cvmax[j] = hi;
    }
    __syncthreads();
    double nu_lo = INFINITY, nu_hi = -INFINITY;
    for (int l = 0; l < nnu; ++l) {
        nu_lo = fmin(nu_lo, lg2_nu_obs[l]);
        nu_hi = fmax(nu_hi, lg2_nu_obs[l]);
    }
    const double lg2_1pz = log2(1 + P.z);
    const int nt = M.n_t;
    const int* rep_of = g_rep_of + (size_t)m * M.th_stride;
    double cphi_max = -INFINITY, cphi_min = INFINITY;
    for (int ii = 0; ii < M.n_phi_eff; ++ii) {
        cphi_max = fmax(cphi_max, gph[ii]);
        cphi_min = fmin(cphi_min, gph[ii]);
    }
    // Which cells does the flux integration query at all (non-spreading jets)?  The reference builds a cell's spectrum on its first query
    // (ICPhoton::compute_log2_I_nu, inverse-compton.h:614-620): a boundary value at node k is asked for only when the interval before
    // or after it holds a requested time for some (theta, phi) row of the cell (observer.h:355-445).  The test here is the range form
    // of that -- for some row, t_obs(k - 1) <= t_max and t_obs(k + 1) >= t_min of the request, with the rows' extreme viewing
    // cosines standing for the phi rows and a 1e-9 margin for the roundings of the flux kernels' log2 times -- a superset of the
    // queried cells (C3: 0.794 of the cells against 0.791 queried exactly; profiles/debug/ssc_needed_cells.py).  A cell that is
    // skipped and queried nevertheless answers loudly (header n = -1, status bit 4).
    const double one_plus_z = 1 + P.z;
    const double t_req_min = need ? tminmax[0] * U_SEC * (1 - 1e-9) : 0, t_req_max = need ? tminmax[1] * U_SEC * (1 + 1e-9) * debug_need_shrink : 0;
    auto t_obs = [&](const double* par, int kk, double cv) {  // calc_eat_non_spreading / calc_t_obs, observer.cpp:51-205
        return (par[(long long)VP_TENG * nt + kk] + (1 - cv) * par[(long long)VP_R * nt + kk] / C_C) * one_plus_z;
    };
    for (int k = lane; k < nt; k += 64) {
        double dmin_k = INFINITY, dmax_k = -INFINITY;
        const int k_prev = k > 0 ? k - 1 : 0, k_next = k + 1 < nt ? k + 1 : nt - 1;
        if (M.rep_phi_stride) {  // (phi, theta) pair rows (non-axisymmetric spreading jet): every pair's own cell at node k
            for (int j = 0; j < M.n_theta; ++j)
                for (int i = 0; i < M.n_phi_eff; ++i) {
                    const long long rr = rep_of[j] + (long long)i * M.rep_phi_stride;
                    const double* par = cellpar + (cell_off[m] + rr * nt) * VAG_NPAR;
                    const double* geo = cellgeo + (cell_off[m] + rr * nt) * 3;
                    const double G = par[(long long)VP_GAMMA * nt + k], u = par[(long long)VP_U * nt + k];
                    const double lg = -log2(G - u * (geo[nt + k] * gph[i] * sin_obs + geo[k] * cos_obs));
                    dmax_k = fmax(dmax_k, lg);
                    dmin_k = fmin(dmin_k, lg);
                }
        } else
        for (int j = 0; j < M.n_theta; ++j) {
            const double* par = cellpar + (cell_off[m] + (long long)rep_of[j] * nt) * VAG_NPAR;
            const double G = par[(long long)VP_GAMMA * nt + k], u = par[(long long)VP_U * nt + k];
            double cvmax = s_cvmax[j], cvmin = s_cvmin[j];
            if (cellgeo) {  // theta evolves: the extrema over phi of cos_v = sin th cos phi sin_obs + cos th cos_obs per cell
                const double* geo = cellgeo + (cell_off[m] + (long long)rep_of[j] * nt) * 3;
                const double ct = geo[k], st = geo[nt + k];
                cvmax = st * cphi_max * sin_obs + ct * cos_obs;
                cvmin = st * cphi_min * sin_obs + ct * cos_obs;
            }
            dmax_k = fmax(dmax_k, -log2(G - u * cvmax));
            dmin_k = fmin(dmin_k, -log2(G - u * cvmin));
            // non-spreading rows: the earliest-arriving row at node k - 1 is the one with the largest viewing cosine, the latest at node
            // k + 1 the one with the smallest (several theta rows may share the cell: every writer stores the same 1)
            // The test is written as "not excluded": a row whose state went non-finite (the outermost rows of a Gaussian jet with a
            // magnetar in a dense wind: NaN observer times from some node on) fails every comparison, and the flux kernels -- which
            // count the nodes before the window like the reference -- do visit those cells.
            if (need && !(t_obs(par, k_prev, cvmax) > t_req_max || t_obs(par, k_next, cvmin) < t_req_min))
                need[cell_off[m] + (long long)rep_of[j] * nt + k] = 1;
        }
        band[((size_t)m * 2 + 0) * band_stride + k] = exp2((nu_lo + lg2_1pz) - dmax_k);  // nu_eval_min_k
        band[((size_t)m * 2 + 1) * band_stride + k] = exp2((nu_hi + lg2_1pz) - dmin_k) * debug_narrow;  // nu_eval_max_k
    }
}

// A flux pass saw a query outside a model's clamped band but inside its theoretical range (status bit 2): the reference drops the
// clamp of that cell and rebuilds it (inverse-compton.h:626-635); here the model's tables are rebuilt unclamped -- the lattice is
// phase-locked, so the nodes both share carry the same values -- and the pass is repeated.
__global__ void vag_ic_unclamp_kernel(int* __restrict__ status, int* __restrict__ unclamp, int nb) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= nb) return;
    if (status[m] & 2) unclamp[m] = 1;
    status[m] &= ~2;
}

// ------------------------------------------------------------------------------------------------
// SSC spectrum of one cell (ICPhoton::generate_spectrum, inverse-compton.h:270-607): one wavefront per
// representative cell, after vag_ic_plan_kernel has laid out every cell's lattices one lane per cell.  Seed and
// electron lattices live in LDS; the walk over electron energies keeps a lane's seed node(s) in registers and adds
// each (energy, bin) term to two diagonal histograms in LDS, from which ONE suffix sum per cell gives the table on the
// phase-locked output lattice (see the comment at the loop).
// ------------------------------------------------------------------------------------------------
// LDS of one wavefront (= one cell).  The setup arrays are dead once every lane holds its seed nodes in registers, so
// the histograms and the KN-correction lattice reuse their memory: 11 KB per wavefront, i.e. 13-14 resident wavefronts
// per CU for a kernel that needs them to keep the VALU busy.
constexpr int IC_MAX_DIAG = 256;  // >= (IC_MAX_NU - 2) + 2 (IC_MAX_G - 1) + 1 = 253
struct IcShared {
    double nu[IC_MAX_NU];  // live throughout
    union {
        struct {  // setup only
            double lg2nu[IC_MAX_NU], dnu[IC_MAX_NU], fv_th[IC_MAX_NU], lg2fv[IC_MAX_NU], lg2r[IC_MAX_NU], inv_lg2r[IC_MAX_NU],
                ratio_th[IC_MAX_NU];
            double gam[IC_MAX_G], dNe[IC_MAX_G], ex[IC_MAX_NU];
        };
        struct {  // accumulation loop
            double D[IC_MAX_DIAG], E[IC_MAX_DIAG];  // diagonal histograms of dNe ex and dNe term (d = seed bin + 2 electron index)
            vdouble2 lat[IC_MAX_LAT];  // per lattice node: KN correction, its log2
            double dNe_i[IC_MAX_G];    // per electron energy, for the lanes of the tail pass (each at its own energy)
            int split_i[IC_MAX_G];
        };
    };
};

VAG_DEV double power_law_bin_integral(double f_lo, double f_hi, double nu_lo, double nu_hi, double lg2f_lo, double lg2f_hi,
                                      double lg2r, double inv_lg2r, double trap) {
    if (!(f_lo > 0) || !(f_hi > 0)) return trap;
    const double s1 = 1 + (lg2f_hi - lg2f_lo) * inv_lg2r;
    if (fabs(s1) > 1e-3) return (f_hi * nu_hi - f_lo * nu_lo) * rcp_ode(s1);  // 1e-3 < |s1| < inf: one Newton step (2e-15) is enough for a term of a sum
    return f_lo * nu_lo * lg2r * 0.6931471805599453;
}

// in place over IC_MAX_DIAG = 256 words: v[d] <- sum_{d' >= d} v[d'].  Four words per lane + shuffles.
VAG_DEV void suffix_scan4(double* __restrict__ v, int lane) {
    const int j = 4 * lane;
    const double a = v[j], b = v[j + 1], c = v[j + 2], d = v[j + 3];
    double S = (a + b) + (c + d);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double t = __shfl_down(S, off, 64);
        if (lane + off < 64) S += t;
    }
    double S_next = __shfl_down(S, 1, 64);
    if (lane == 63) S_next = 0;
    const double c3 = S_next + d, c2 = c3 + c, c1 = c2 + b;
    v[j + 3] = c3;
    v[j + 2] = c2;
    v[j + 1] = c1;
    v[j] = c1 + a;
}

VAG_DEV double read_lane(double v, int src) {  // src uniform across the wave
    const int l = __builtin_amdgcn_readfirstlane(src);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

#ifndef VAG_IC_WAVES
#define VAG_IC_WAVES 4
#endif
#ifndef VAG_IC_PERSISTENT
#define VAG_IC_PERSISTENT 0  // (developer builds: 1 = persistent wavefronts with the next records / constants prefetched, see the kernel's end)
#endif
#ifndef VAG_IC_ONE_NODE_MAX
#define VAG_IC_ONE_NODE_MAX 64  // seed lattices up to this size take one node per lane (developer builds: 0 = always two)
#endif
// Lattice plan of every cell, one LANE per cell (compute_grid_params + the sizes of initialize_grids, inverse-compton.h:297-369).
// These few hundred instructions are the same for all 64 lanes of the wavefront that builds the cell's spectrum, so they are
// done here at 1/64 of the cost and handed over in the cell's plan row (ICP_* words); the cell's header is final here, including
// the place of its table in the pool: the lanes of a workgroup add up their table lengths and ONE atomic per workgroup reserves the
// block (the order of the blocks in the pool follows the scheduler; the values in them do not depend on it).  Cells that get no
// table (failed model, degenerate or over-capacity lattice: n = 0 and the theoretical range; a cell no row queries: n = -1) are
// finished here.
enum { ICP_CELL = 0 /* the cell this record belongs to: records are stored compacted, runnable cells only */, ICP_NU_SIZE, ICP_G_SIZE, ICP_N_LO, ICP_LG2_NU0, ICP_LG2_G0, ICP_LG2_GM, ICP_INV_GM, ICP_INV_GMAX,
       ICP_SMOOTH_THICK, ICP_LOG2_X_FAR /* SpecConst of the model's p: a division and a library log2 per wavefront otherwise */,
       ICP_PHASE,
       // r05: what the spectrum kernel used to fetch through its model (params, meta, cell offsets: two further dependent round trips
       // per cell) and from the detail arrays -- the record is now the ONLY thing a wavefront needs before it can request the cell's
       // spectrum constants, so that a persistent wavefront can have both in flight one and two cells ahead
       ICP_ROW0 /* first cell of the cell's representative row */, ICP_NT, ICP_K, ICP_GAMMA_M, ICP_GAMMA_C, ICP_COLUMN_DEN, ICP_YC,
       ICP_REGIME, ICP_P, ICP_KN, ICP_N_IC, ICP_OFF /* = the header's n and offset */, ICP_N };
static_assert(ICP_N <= IC_PLAN, "plan row");
// ------------------------------------------------------------------------------------------------
// Cells whose lattices do not fit vag_ic_photon_kernel's LDS layout (more than IC_MAX_NU seed nodes, IC_MAX_G electron energies or
// IC_MAX_OUT output nodes: the reference sizes its arrays per cell and has no such limit).  The plan kernel marks their records (bit 1
// of ICP_KN), reserves the working arrays behind the cell's table in the pool and lists the records; one wavefront per listed record
// then runs ICPhoton::generate_spectrum in the reference's own form -- the scattering CDF per electron energy (build_cdf_thomson /
// build_cdf_KN, inverse-compton.h:415-481) and the walk over the output nodes (accumulate_IC, :483-527) -- with every array in HBM
// and the wavefront's lanes striding over the seed bins / output nodes.  Rare by construction (no cell of BASELINE's configs or of the
// prior boxes takes it), so it is written for plainness: __syncthreads() between its phases, no LDS beyond the cell's constants.
// It is also an independent restatement on the device of what the fast kernel computes through its diagonal histograms: the test
// that sends EVERY cell through it (VAG_DEBUG_IC_FAST_NU_MAX=0) compares the two.
// ------------------------------------------------------------------------------------------------
constexpr int IC_SLOW_MAX_NU = 2048, IC_SLOW_MAX_G = 2048, IC_SLOW_MAX_OUT = 4096;
constexpr int IC_SLOW_NU_ARRAYS = 15;
// doubles of working memory behind the table of a cell of the slow path
VAG_DEV int ic_slow_scratch(int nu_size, int g_size, int n_ic) {
    return IC_SLOW_NU_ARRAYS * nu_size + 2 * g_size + 2 * (g_size + nu_size - 1) + n_ic;
}
// output lattice node q of a table: phase + IC_Q (idx0 + 2 q), idx0 + 2 q an integer far below 2^53 formed in double -- exactly the
// value the reference converts from its integer (log2_nu_IC, inverse-compton.h:595-606)
VAG_DEV double ic_out_node(double phase, double idx0, int q) { return phase + IC_Q * (idx0 + 2.0 * (double)q); }
__global__ void __launch_bounds__(256)
vag_ic_plan_kernel(const vag_model_params* __restrict__ params, int nb, const VagGridMeta* __restrict__ meta, Layout lay,
                   long long n_cells, const double* __restrict__ det, const double* __restrict__ band,
                   double* __restrict__ ichdr /* [cells][IC_HDR] */, double* __restrict__ icplan /* [runnable cells][IC_PLAN], compacted */,
                   unsigned long long* __restrict__ pool_used /* [0] doubles handed out so far (starts at 2: slot 0 serves the empty tables),
                                                                 [1] records written so far, [2] doubles handed to cells of the slow path,
                                                                 [3] records of the slow path (all three start at 0) */,
                   int* __restrict__ ic_status,
                   unsigned long long* __restrict__ work /* optional [2]: (electron energy, seed frequency) terms / lattice nodes */,
                   int band_stride, const unsigned char* __restrict__ need /* [cells] or nullptr (vag_ic_band_kernel) */,
                   int fast_nu_max /* IC_MAX_NU, or a test's smaller limit: longer seed lattices take the slow path */,
                   unsigned long long slow_reserve /* doubles the slow path's cells may take from the pool altogether */,
                   int* __restrict__ slow_list /* [slow_cap] records of the slow path */, int slow_cap) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = c < n_cells && c < lay.cell_off[nb];  // n_cells is the arrays' stride (>= the batch's cell count)
    double* hdr = ichdr + (size_t)(live ? c : 0) * IC_HDR;
    double plan[IC_PLAN];  // the cell's record: written out below, at the place the workgroup reserves for it among the runnable cells
    bool slow_cell = false;
    int model = 0;
    // the cell's plan; returns the doubles it takes from the pool (0: no table)
    auto plan_cell = [&]() -> int {
        const int m = cell_model(lay.cell_off, nb, c);
        hdr[ICH_N] = 0;
        hdr[ICH_OFF] = 0;
        const int nt = meta[m].n_t;
        if (meta[m].status != 0) return 0;
        if (need && !need[c]) {  // no (theta, phi) row queries this cell: no table, and a query would be an engine fault (status bit 4)
            hdr[ICH_N] = -1;
            return 0;
        }
        const int k = (int)((c - lay.cell_off[m]) % nt);
        const double gamma_m = det[VD_GAMMA_M * n_cells + c], gamma_c = det[VD_GAMMA_C * n_cells + c];
        const double gamma_M = det[VD_GAMMA_MAX * n_cells + c];
        const double nu_m = det[VD_NU_M * n_cells + c], nu_a = det[VD_NU_A * n_cells + c], nu_M = det[VD_NU_MAX * n_cells + c];
        const double nu_eval_min = band[((size_t)m * 2 + 0) * band_stride + k];
        const double nu_eval_max = band[((size_t)m * 2 + 1) * band_stride + k];
        // compute_grid_params, inverse-compton.h:297-338
        const double tail_factor = dmax(-log(1e-2), 5.0);
        const double gamma_min = dmin(gamma_m, gamma_c) / 30;
        const double gamma_max = dmax(gamma_M * tail_factor, gamma_min);
        const double nu_min = dmin(nu_a, nu_m) / 10;
        const double nu_max = dmax(nu_M * tail_factor, nu_min);
        double nu_IC_min = 4 * IC_X0 * nu_min * gamma_min * gamma_min;
        const double nu_ic_base = 4 * IC_X0 * nu_M * gamma_M * gamma_M;
        const double nu_ic_cut = dmax(nu_ic_base * tail_factor * tail_factor, nu_ic_base * tail_factor);
        double nu_IC_max = nu_ic_cut * 2.0;
        const double theory_max = log2(nu_IC_max), theory_min = log2(nu_IC_min);
        nu_IC_min = dmax(nu_IC_min, dmin(nu_eval_min / 4.0, nu_IC_max / 16.0));
        nu_IC_max = dmin(nu_IC_max, dmax(nu_eval_max * 4.0, nu_IC_min * 16.0));
        hdr[ICH_TMIN] = theory_min;
        hdr[ICH_TMAX] = theory_max;
        auto posfin = [](double x) { return isfinite(x) && x > 0; };
        if (!(posfin(gamma_min) && posfin(gamma_max) && posfin(nu_min) && posfin(nu_max) && posfin(nu_IC_min) && posfin(nu_IC_max)))
            return 0;
        // initialize_grids, inverse-compton.h:340-369
        const double step = 2 * IC_Q;
        const double lg2_nu0 = log2(nu_min), lg2_g0 = log2(gamma_min);
        int nu_size = (int)ceil((log2(nu_max) - lg2_nu0) / step) + 1;
        int g_size = (int)ceil((log2(gamma_max) - lg2_g0) / step) + 1;
        if (nu_size < 2) nu_size = 2;
        if (g_size < 2) g_size = 2;
        const double phase = lg2_nu0 + 2 * lg2_g0 + log2(4 * IC_X0);
        const long n_lo = (long)floor((log2(nu_IC_min) - phase) / step);
        const long n_hi = (long)ceil((log2(nu_IC_max) - phase) / step);
        const long span = n_hi - n_lo;
        const int n_ic = (int)(span > 1 ? span : 1) + 1;
        // lattices beyond the fast kernel's LDS layout: vag_ic_photon_slow_kernel, whose working arrays follow the table in the pool
        slow_cell = nu_size > fast_nu_max || g_size > IC_MAX_G || n_ic > IC_MAX_OUT;
        int len = n_ic;
        if (slow_cell) {
            len += ic_slow_scratch(nu_size, g_size, n_ic);
            bool fits = nu_size <= IC_SLOW_MAX_NU && g_size <= IC_SLOW_MAX_G && n_ic <= IC_SLOW_MAX_OUT;
            if (fits) fits = atomicAdd(pool_used + 2, (unsigned long long)len) + (unsigned long long)len <= slow_reserve;
            if (!fits) {
                atomicOr(ic_status + m, 1);  // capacity: reported loudly by the host
                return 0;
            }
        }
        if (work) {  // instrumentation: the unit of vag_ic_photon_kernel's work model (bench.py)
            atomicAdd(work, (unsigned long long)g_size * (unsigned long long)nu_size);
            atomicAdd(work + 1, (unsigned long long)(g_size + nu_size + n_ic));
        }
        hdr[ICH_N] = (double)n_ic;
        hdr[ICH_FIRST] = ic_out_node(phase, (double)(n_lo * 2), 0);         // first and last node of the output lattice: what the flux
        hdr[ICH_LAST] = ic_out_node(phase, (double)(n_lo * 2), n_ic - 1);   // passes' evaluator needs of it (ic_table_eval_hdr)
        plan[ICP_PHASE] = phase;
        plan[ICP_CELL] = (double)c;  // < 2^53: exact
        plan[ICP_ROW0] = (double)(c - k);  // < 2^53: exact
        plan[ICP_NT] = (double)nt;
        plan[ICP_K] = (double)k;
        plan[ICP_GAMMA_M] = gamma_m;
        plan[ICP_GAMMA_C] = gamma_c;
        plan[ICP_COLUMN_DEN] = det[VD_COLUMN_DEN * n_cells + c];
        plan[ICP_YC] = det[VD_YC * n_cells + c];
        plan[ICP_REGIME] = det[VD_REGIME * n_cells + c];
        plan[ICP_P] = params[m].p;
        plan[ICP_KN] = ((params[m].flags & VAG_FLAG_KN) ? 1.0 : 0.0) + (slow_cell ? 2.0 : 0.0);  // bit 0: Klein-Nishina, bit 1: slow path
        plan[ICP_N_IC] = (double)n_ic;
        plan[ICP_NU_SIZE] = (double)nu_size;
        plan[ICP_G_SIZE] = (double)g_size;
        plan[ICP_N_LO] = (double)n_lo;
        plan[ICP_LG2_NU0] = lg2_nu0;
        plan[ICP_LG2_G0] = lg2_g0;
        plan[ICP_LG2_GM] = log2(gamma_m);  // uniform factors of the electron distribution (sample_distributions)
        plan[ICP_INV_GM] = 1 / gamma_m;
        plan[ICP_INV_GMAX] = 1 / gamma_M;
        SpecConst sc;
        sc.init(params[m].p);
        plan[ICP_SMOOTH_THICK] = sc.smooth_thick;
        plan[ICP_LOG2_X_FAR] = sc.log2_x_far;
        model = m;
        return len;
    };
    const int len = live ? plan_cell() : 0;
    // place of the table in the pool and of the record among the runnable cells: exclusive sums over the workgroup's lanes + the
    // workgroup's two reservations
    __shared__ unsigned long long s_base, s_base_run;
    __shared__ int s_wave[4], s_wave_run[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int incl = len, incl_run = len > 0 ? 1 : 0;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64), tr = __shfl_up(incl_run, off, 64);
        if (lane >= off) incl += t, incl_run += tr;
    }
    if (lane == 63) s_wave[w] = incl, s_wave_run[w] = incl_run;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        const int total_run = s_wave_run[0] + s_wave_run[1] + s_wave_run[2] + s_wave_run[3];
        s_base = total > 0 ? atomicAdd(pool_used, (unsigned long long)total) : 0ull;
        s_base_run = total_run > 0 ? atomicAdd(pool_used + 1, (unsigned long long)total_run) : 0ull;
    }
    __syncthreads();
    if (len > 0) {
        int before = incl - len, before_run = incl_run - 1;
        for (int q = 0; q < w; ++q) before += s_wave[q], before_run += s_wave_run[q];
        hdr[ICH_OFF] = (double)(s_base + (unsigned long long)before);  // < 2^53: exact
        plan[ICP_OFF] = hdr[ICH_OFF];
        const unsigned long long rec = s_base_run + (unsigned long long)before_run;
        double* dst = icplan + (size_t)rec * IC_PLAN;
#pragma unroll
        for (int q = 0; q < ICP_N; ++q) dst[q] = plan[q];
        if (slow_cell) {
            const unsigned long long at = atomicAdd(pool_used + 3, 1ull);
            if (at < (unsigned long long)slow_cap)
                slow_list[at] = (int)rec;
            else
                atomicOr(ic_status + model, 1);  // (a batch with more cells on the slow path than its list holds: capacity)
        }
    }
}

// r05: PERSISTENT wavefronts.  One wavefront per cell spent a fifth of its life (7 k of 36 k cycles, -DVAG_IC_STAMPS) in its prologue:
// plan row -> model -> params / meta / cell offsets -> the 46 spectrum constants, three dependent round trips to HBM that four resident
// wavefronts per SIMD cannot hide (VALU 78 % busy with every instruction counter well below its pipe's limit).  Now the launch is as
// many wavefronts as the device holds (4 per SIMD) and each takes the cells c = w, w + G, w + 2 G, ...; the plan record of cell c + 2 G and
// the constants of cell c + G (whose record arrived an iteration ago) are requested before cell c is worked on, so a cell starts with its
// constants in a register and its record in L2 (read again through the scalar cache: the record is wave-uniform).
// The spectrum kernel's workgroup is ONE wavefront: its lanes run in lockstep and the LDS serves a wavefront's instructions in order, so
// what __syncthreads() has to provide between its phases is only that the compiler keeps the memory accesses on their side of the line --
// not the s_waitcnt vmcnt(0) it also emits, which in the persistent loop would make every phase wait for the next cells' prefetches and
// every cell for its own table stores to land in HBM.
// (wave_sync() of vag_kernels.h: a wavefront-scope fence and a compiler barrier, no instruction)
struct IcPhotonArgs {
    long long n_run;    // records (= runnable cells) vag_ic_plan_kernel wrote ...
    const unsigned long long* n_run_dev;  // ... or, when the host did not wait for that count (likelihood calls), where it stands in HBM
    long long n_cells;  // stride of the SoA arrays (>= the batch's cell count)
    const double *icy, *cellpar, *cellq, *sp_table, *kn_lut, *icplan;
    double* icpool;
};
// The kernel's arguments read again from the kernel-argument segment (scalar loads through a pointer the optimiser cannot see through,
// as load_series_args of vag_grid_rows.h): the persistent loop keeps its cell number and two prefetched registers alive, not nine pointers.
typedef const IcPhotonArgs __attribute__((address_space(4))) KernargIcPhotonArgs;
VAG_DEV IcPhotonArgs load_ic_photon_args() {
    KernargIcPhotonArgs* p = (KernargIcPhotonArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    IcPhotonArgs a;
    __builtin_memcpy(&a, p, sizeof(IcPhotonArgs));
    return a;
}
__global__ void __launch_bounds__(64, VAG_IC_WAVES)
vag_ic_photon_kernel(IcPhotonArgs args_unused_directly) {
    const int lane = threadIdx.x;
    __shared__ IcShared sh;
    constexpr int CST_Q = VAG_NPAR, CST_Y = VAG_NPAR + VAG_NQ, CST_N = VAG_NPAR + VAG_NQ + VAG_NICY;
    static_assert(CST_N <= 64, "one constant per lane");
    static_assert(IC_PLAN <= 64, "one record word per lane");
    // `prefetch` requests the next cells' record / constants; a cell calls it once, where its own register demand is lowest
    auto cell = [&](const long long c, const double my_cst, auto prefetch) {
    // (the lane number made opaque per cell: what a cell derives from it -- LDS addresses, masks -- is formed inside the turn instead
    // of being hoisted out of the persistent loop and held in registers across it)
    int lane = threadIdx.x;
    asm volatile("" : "+v"(lane));
#ifdef VAG_IC_STAMPS  // developer aid: cycles of a wavefront per section
    long long c_t[10];
    int c_n = 0;
    c_t[c_n++] = __builtin_readcyclecounter();
#define VAG_IC_MARK() do { __builtin_amdgcn_s_waitcnt(0); c_t[c_n++] = __builtin_readcyclecounter(); } while (0)
#else
#define VAG_IC_MARK() do { } while (0)
#endif
    const IcPhotonArgs A = load_ic_photon_args();
    const double* __restrict__ sp_table = A.sp_table;
    const double* __restrict__ kn_lut = A.kn_lut;
    double* __restrict__ icpool = A.icpool;
    // wave-uniform and never written while this kernel runs: read through the constant address space, i.e. by scalar loads (a pointer
    // that arrives through the opaque argument block carries no such promise and would be read by vector loads + v_readfirstlane)
    typedef const double __attribute__((address_space(4))) ConstDouble;
    ConstDouble* plan = (ConstDouble*)(unsigned long long)(A.icplan + (size_t)c * IC_PLAN);
#if VAG_IC_PERSISTENT
    if ((int)plan[ICP_KN] & 2) {  // a cell of the slow path (vag_ic_photon_slow_kernel)
        prefetch();
        return;
    }
#endif
    double* tab = icpool + (unsigned long long)plan[ICP_OFF];  // this cell's table in the pool
    int nu_size = (int)plan[ICP_NU_SIZE], g_size = (int)plan[ICP_G_SIZE];
    const long n_lo = (long)plan[ICP_N_LO];
    const double lg2_nu0 = plan[ICP_LG2_NU0], lg2_g0 = plan[ICP_LG2_G0];
    const double lg2_gm = plan[ICP_LG2_GM], inv_gm = plan[ICP_INV_GM], inv_gM = plan[ICP_INV_GMAX];
    const int n_ic = (int)plan[ICP_N_IC];
    const double phase = plan[ICP_PHASE];
    const long idx0 = n_lo * 2;
    const double step = 2 * IC_Q;
    struct { double p; } P{plan[ICP_P]};
    const bool KN = ((int)plan[ICP_KN] & 1) != 0;
    const double gamma_m = plan[ICP_GAMMA_M], gamma_c = plan[ICP_GAMMA_C];
    const double column_den = plan[ICP_COLUMN_DEN];
    const double Y_c = plan[ICP_YC];
    const int regime = (int)plan[ICP_REGIME];
    // Every constant of the cell the two sampling loops below use arrives as ONE vector register (lane l holds constant l, requested
    // while the previous cell was worked on) and is handed round through LDS: left to the evaluators they are fetched where they are
    // used -- behind lane-divergent branches, i.e. as vector loads, one memory round trip after the other inside the loops (a dozen
    // per pass; the loops then took 40 % of this kernel's time); as scalar loads up front they overflow the scalar registers.
    wave_sync();
    for (int j = lane; j < nu_size; j += 64) {
        sh.lg2nu[j] = lg2_nu0 + step * (double)j;
        sh.nu[j] = exp2_sat(sh.lg2nu[j]);
    }
    for (int i = lane; i < g_size; i += 64) sh.gam[i] = exp2_sat(lg2_g0 + step * (double)i);
    sh.ex[lane] = my_cst;  // `ex` is not written before the Thomson CDF
    wave_sync();
    double cp[VAG_NPAR];
    for (int q : {VP_LG2_LO, VP_LG2_HI, VP_DLO, VP_INV_SLO, VP_DHI, VP_INV_SHI, VP_LG2_NUM, VP_TNORM, VP_SAB, VP_INV_SAB, VP_LG2_I,
                  VP_LG2_NUMAX, VP_INV_NUMAX})
        cp[q] = sh.ex[q];
    IcQ icq;
    icq.head(sh.ex + CST_Q, 1);
    icq.rest(sh.ex + CST_Q, 1);
    const bool y_any = sh.ex[CST_Y + VY_NSEG] != 0;
    const double yS0 = sh.ex[CST_Y + VY_S0], yC0 = sh.ex[CST_Y + VY_C0];
    const double yL1 = sh.ex[CST_Y + VY_L1], yS1 = sh.ex[CST_Y + VY_S1], yC1 = sh.ex[CST_Y + VY_C1];
    const double yL2 = sh.ex[CST_Y + VY_L2], yS2 = sh.ex[CST_Y + VY_S2], yC2 = sh.ex[CST_Y + VY_C2];
    VAG_IC_MARK();  // 1: prologue
#ifdef VAG_IC_ABLATE
    if (VAG_IC_ABLATE >= 4) { prefetch(); return; }  // prologue only: loads, lattice parameters, lattice nodes
#endif
    // sample_distributions, inverse-compton.h:371-399
    SpecConst sc;
    sc.smooth_thick = plan[ICP_SMOOTH_THICK], sc.log2_x_far = plan[ICP_LOG2_X_FAR];  // sc.init(P.p), done by the plan kernel
    // SynElectrons::compute_column_den (synchrotron.cpp:261-309) / gamma^2 * dgamma at the lattice energies: log2(gamma) is the
    // node's own exponent, the cell-uniform factors come from the plan, the two exponentials of a branch are one exp2
    const bool slow = regime == 1 || regime == 2 || regime == 5, fast = regime == 3 || regime == 4 || regime == 6;
#ifdef VAG_IC_STAMPS
    VAG_IC_MARK();  // 2a: spectrum constants
#endif
    for (int i = lane; i < g_size; i += 64) {
        constexpr double LOG2E_ = 1.4426950408889634;
        const double gi = sh.gam[i], rg = rcp_fast(gi), lg = lg2_g0 + step * (double)i;
        const double dgi = 0.5 * ((i + 1 < g_size ? sh.gam[i + 1] : gi) - (i > 0 ? sh.gam[i - 1] : gi));
        double spec = 0;
        if (slow)
            spec = (P.p - 1) * inv_gm * exp2_sat((-gi * inv_gM - gamma_m * rg) * LOG2E_ - P.p * (lg - lg2_gm)) * gamma_c *
                   rcp_fast(gi + gamma_c);
        else if (fast)
            spec = exp2_sat((-gi * inv_gM - gamma_c * rg) * LOG2E_) * gamma_c * rg * rg *
                   rcp_fast(1.0 + exp2_sat(dmin((P.p - 1) * (lg - lg2_gm), 1000.0)));
        double den = column_den * spec;
        if (gi > gamma_c) {  // icy_lg2_Y on the constants at hand: the last segment whose lower edge lg reaches (unused ones: +inf)
            double z = fma(yS0, lg, yC0);
            z = lg >= yL1 ? fma(yS1, lg, yC1) : z;
            z = lg >= yL2 ? fma(yS2, lg, yC2) : z;
            den = den * (1 + Y_c) * rcp_fast(1 + (y_any ? exp2_sat(z) : 0.0));
        }
        sh.dNe[i] = den * rg * rg * dgi;
    }
#ifdef VAG_IC_STAMPS
    VAG_IC_MARK();  // 2b: electron distribution
#endif
    for (int j = lane; j < nu_size; j += 64) {  // f = I_seed / nu^2 and its logarithm, from the logarithm
        const double x = sh.lg2nu[j];
        const double lf = log2_I_nu_ic_core<true, true>(cp, 1, icq.applies(x), icq, sc, x, sp_table, sh.nu[j]) - 2 * x;
        const double f = exp2_sat(lf);
        sh.fv_th[j] = f;
        sh.lg2fv[j] = f > 0 ? lf : -INFINITY;
    }
    wave_sync();
    VAG_IC_MARK();  // 2: sampled distributions
#ifdef VAG_IC_ABLATE
    if (VAG_IC_ABLATE >= 3) { prefetch(); return; }  // ... + the sampled electron and seed distributions
#endif
    const int nu_last = nu_size - 1;
    for (int j = lane; j < nu_last; j += 64) {
        sh.dnu[j] = sh.nu[j + 1] - sh.nu[j];
        sh.lg2r[j] = sh.lg2nu[j + 1] - sh.lg2nu[j];
        sh.inv_lg2r[j] = sh.lg2r[j] != 0 ? 1 / sh.lg2r[j] : 0;
    }
    wave_sync();
    // build_cdf_thomson, inverse-compton.h:415-430
    for (int j = lane; j < nu_last; j += 64) {
        const double trap = 0.5 * (sh.fv_th[j] + sh.fv_th[j + 1]) * sh.dnu[j];
        const double exact = power_law_bin_integral(sh.fv_th[j], sh.fv_th[j + 1], sh.nu[j], sh.nu[j + 1], sh.lg2fv[j],
                                                    sh.lg2fv[j + 1], sh.lg2r[j], sh.inv_lg2r[j], trap);
        sh.ex[j] = exact;
        sh.ratio_th[j] = trap > 0 ? exact / trap : 1;
    }
    wave_sync();
    VAG_IC_MARK();  // 3: Thomson CDF
#ifdef VAG_IC_ABLATE
    if (VAG_IC_ABLATE >= 2) { prefetch(); return; }
#endif
    // accumulate over electron energies (accumulate_IC, inverse-compton.h:483-527; build_cdf_KN, :432-481).  For electron
    // energy i the reference forms the scattering CDF over the seed bins, c_j(i) = sum_{m >= j} ex_m(i), and output node kk
    // takes dNe_i (c_{j+1}(i) + term_j(i)) at j = n_lo - 2 i + kk (c_0(i) alone for j < 0, nothing past the last bin).  All
    // lattice offsets are even, so the pairs (i, m) that reach node kk through the CDF are exactly those on the diagonals
    // d = m + 2 i > n_lo + kk, and the pair behind term_j sits on d = n_lo + kk:
    //     I[kk] = sum_{d > n_lo + kk} D[d] + E[n_lo + kk],   D[d] = sum_{m + 2 i = d} dNe_i ex_m(i),  E[d] = same with term_m(i).
    // So the loop only forms ex and term of the lane's bins and adds them to two diagonal histograms in LDS (ds_add_f64, the
    // lanes of one instruction hit different words); ONE suffix sum over D per cell replaces a 64-lane scan, the continuity
    // shift below the KN split, the exchange row and the three-slot gather per electron energy (below the split the shifted
    // Thomson CDF is the same as ex_m = the Thomson bin integral).  Lane l owns bin l through all energies; a seed lattice
    // with more than 64 bins (one cell in five) hands its bins 64.. to a second, PACKED pass: W = 8 ... 64 lanes per tail bin
    // set, 64 / W energy chunks side by side, so that eight tail bins cost g / 8 trips instead of doubling every trip.
    double I_acc[3] = {0, 0, 0};
    const double lg2nu_first = sh.lg2nu[0];
    // lane L keeps what electron energy i = L needs (read back with v_readlane, no LDS): gamma, dNe and the KN split index
    const double my_dNe = lane < g_size ? sh.dNe[lane] : 0.0;
    const double my_gam = lane < g_size ? sh.gam[lane] : 1.0;
    const int n_lo_i = (int)n_lo;
    auto energies = [&](auto kn_tag) {
        constexpr bool WITH_KN = decltype(kn_tag)::value;  // Thomson cells: every bin keeps its two constants, no lattice
        // Per bin [j, j+1], everything of ex / term that does not depend on the electron energy is folded into constants, so
        // that an energy costs the lane one lattice read and ~10 FP64 instructions:
        //   above the split     f nu = (f_th nu) corr,  s1 = 1 + (lf_N - lf) / lg2r = s_th + (lg2corr_N - lg2corr) / lg2r
        //                       ex = term = (A_N corr_N - A corr) / s1           (|s1| > 1e-3;  A corr lg2r ln 2 otherwise)
        //   seeds not both > 0  ex = term = the trapezoid (f + f_N) dnu / 2 = A_N corr_N + A corr with A := f_th dnu / 2 instead
        //   bin below the split ex = the Thomson integral, term = trap ratio_th: constants -- except the bin right below it, whose
        //                       upper edge carries the corrected f: term = K0 + K1 corr_N
        struct Bin {
            int j;
            bool bin, pos;
            double A, AN, s_th, ilr, Lr, exth, termth, K0, K1;
        };
        auto bin_constants = [&](int j) {
            Bin b;
            b.j = j;
            b.bin = j < nu_last;
            const int jj = b.bin ? j : 0;
            const double nu_a = sh.nu[jj], nuN = sh.nu[jj + 1], fth = sh.fv_th[jj], fthN = sh.fv_th[jj + 1];
            const double lth = sh.lg2fv[jj], lthN = sh.lg2fv[jj + 1];
            const double hd = 0.5 * sh.dnu[jj];  // half the bin width: the trapezoid's factor
            const double rth = sh.ratio_th[jj];
            b.ilr = sh.inv_lg2r[jj];
            b.exth = sh.ex[jj];  // Thomson bin integral (build_cdf_thomson)
            b.pos = fth > 0 && fthN > 0;
            b.A = b.pos ? fth * nu_a : fth * hd;
            b.AN = b.pos ? fthN * nuN : fthN * hd;
            b.s_th = b.pos ? 1 + (lthN - lth) * b.ilr : 1.0;
            b.Lr = sh.lg2r[jj] * 0.6931471805599453;
            b.K0 = fth * hd * rth;
            b.K1 = fthN * hd * rth;
            b.termth = (fth + fthN) * hd * rth;
            return b;
        };
#ifndef VAG_IC_MAIN_BINS
#define VAG_IC_MAIN_BINS 64  // bins of the main pass (test builds: 24 sends most bins of an ordinary cell through the tail pass)
#endif
        constexpr int MAIN = VAG_IC_MAIN_BINS;
        static_assert(MAIN <= 64, "one main bin per lane");
#if VAG_IC_MAIN_BINS < 64
        if (nu_last - MAIN > 64) __builtin_trap();  // test builds: the tail pass holds at most 64 bins
#endif
        Bin bm = bin_constants(lane);
        if (MAIN < 64 && lane >= MAIN) bm.bin = false;
        // the tail pass: W lanes side by side for the bins MAIN .. nu_last - 1, 64 / W chunks of T consecutive energies
        const int n_tail = nu_last > MAIN ? nu_last - MAIN : 0;
        int lgW = 3;
        while ((1 << lgW) < n_tail && lgW < 6) ++lgW;
        const int W = 1 << lgW, T = (g_size * W + 63) >> 6;
        const int i_tail0 = (lane >> lgW) * T;
        Bin bt = bm;
        if (n_tail > 0) bt = bin_constants(MAIN + (lane & (W - 1)));
        int my_split = nu_size;  // Thomson: no bin lies at or above the split
        // KN correction per node of the shared gamma-nu lattice, inverse-compton.h:566-574.  Both lattices step by two quanta, so only
        // the even nodes of the reference's lattice are ever read: node q here is its node 2 q.  And only the nodes from the Klein-
        // Nishina split upwards: energy i reads the nodes i + j, i + j + 1 of the bins j >= split(i) - 1 (see `pair`), i.e. nothing
        // below min_i (i + split(i)).  split(i) is the first seed node with gamma_i nu_j >= the split constant, clamped to the seed
        // lattice -- a function of i + j up to the rounding of the stored nodes (~1e-14 in log2) -- so i + split(i) >= split(0) - 1
        // for every i: the fill starts at q_min = split(0) - 2 (one node to spare; the nodes below it hold NaN, never read).  That is
        // a third of the lattice on the configs[2] cells and what makes the fill ONE pass of the wavefront instead of two (r05).
        int q_min = 0;
        double kx0 = 0;  // h nu gamma / (m_e c^2) at lattice node q_min + lane: the product of two stored nodes, no exp2
        const int n_lat = (g_size - 1) + (nu_size - 1) + 1;
        if constexpr (WITH_KN) {
            if (lane < g_size) {
                const double nu_split = 1e-4 * (C_ME * C_C2 / C_H) * rcp_fast(my_gam);  // (<= 1 ulp; the nodes themselves are exp2_sat values)
                // first node with nu >= nu_split (the reference scans from 0): lattice guess, then settle on the stored nodes
                const double lg2_split0 = log2(1e-4 * (C_ME * C_C2 / C_H)) - lg2_g0;
                int js = (int)ceil((lg2_split0 - step * (double)lane - lg2_nu0) * (1.0 / (2 * IC_Q)));  // (a guess: the loops below settle it)
                js = js < 0 ? 0 : (js > nu_last ? nu_last : js);
                while (js > 0 && sh.nu[js - 1] >= nu_split) --js;
                while (js < nu_last && sh.nu[js] < nu_split) ++js;
                my_split = js;
            }
            q_min = __builtin_amdgcn_readfirstlane(my_split) - 2;  // lane 0 = energy 0 (g_size >= 2)
            q_min = q_min < 0 ? 0 : q_min;
#ifdef VAG_IC_FULL_LATTICE
            q_min = 0;  // developer builds: every node, as before r05
#endif
            const int q = q_min + lane, jq = q < nu_last ? q : nu_last;
            int iq = q - jq;
            iq = iq < g_size ? iq : g_size - 1;  // lanes past the lattice: any stored node
            kx0 = sh.gam[iq] * sh.nu[jq] * (C_H / (C_ME * C_C2));
        }
        wave_sync();  // every setup array has been read: from here on their memory holds D / E / lat / dNe_i / split_i
        {   // D and E, contiguous: 512 doubles = four 16-byte stores per lane
            static_assert(offsetof(IcShared, E) == offsetof(IcShared, D) + sizeof(double) * IC_MAX_DIAG && IC_MAX_DIAG == 256, "D | E");
            vdouble2* z = reinterpret_cast<vdouble2*>(sh.D);
#pragma unroll
            for (int q = 0; q < 4; ++q) z[lane + 64 * q] = vdouble2{0.0, 0.0};
        }
        if constexpr (WITH_KN) {
            const double lg2_base = lg2_g0 + lg2nu_first;  // log2 of the first electron node times the first seed node
            for (int q0 = q_min; q0 < n_lat; q0 += 64) {
                const int q = q0 + lane;
                const double lg2_x = (lg2_base + step * (double)q) + log2(C_H / (C_ME * C_C2));
                const double x = q0 == q_min ? kx0 : exp2_sat(lg2_x);  // (a second pass: lattices with > 64 used nodes, rare)
                double cq, lq;
                compton_correction_pair_node(lg2_x, x, kn_lut, cq, lq);
                if (q < n_lat) sh.lat[q] = vdouble2{cq, lq};
            }
            for (int q = lane; q < q_min; q += 64) sh.lat[q] = vdouble2{__builtin_nan(""), __builtin_nan("")};
        }
        if (n_tail > 0) sh.dNe_i[lane] = my_dNe, sh.split_i[lane] = my_split;
        wave_sync();
        // HERE: behind the cell's last global read (the KN table words above) and ahead of the walk's ~18 k cycles of LDS and VALU
        // work.  Memory returns in order, so any later read of THIS cell would wait for the next cells' words as well; and the
        // sampling loops' ~70 registers of cell constants are dead.
        prefetch();
        VAG_IC_MARK();  // 4: KN lattice, split indices
        int g_run = g_size;
#ifdef VAG_IC_ABLATE
        if (VAG_IC_ABLATE >= 1) g_run = 0;
#endif
        // ex and term of one (energy, bin) pair, added to the diagonal histograms.  `asm volatile("")` inside a branch: keep it
        // a branch under the exec mask (scalar instructions) -- if-converted, the three-way choice costs ten v_cndmask per pair
        // (Measured and rejected, r04: the second histogram kept as E - D -- above the split and below it the edge term repeats the
        // bin integral, so only the bin right below the split feeds it and every other pair needs ONE ds_add_f64: 20.33 against
        // 19.87 ms per launch on the configs[2] batch; the branch around the rare second add costs more than the add.)
        auto pair = [&](int i, double dNe, int j_split, const Bin& b, const vdouble2& nd0, const vdouble2& nd1) {
            double ve = b.exth, vt = b.termth;
            if (WITH_KN && b.bin && b.j >= j_split - 1) {  // at or right below the split: the bin sees the lattice
                asm volatile("");
                const double cN = nd1.x;
                if (b.j >= j_split) {
                    asm volatile("");
                    const double u = b.A * nd0.x;
                    if (__builtin_expect(b.pos, 1)) {
                        asm volatile("");
                        const double s1 = fma(nd1.y - nd0.y, b.ilr, b.s_th);
                        if (__builtin_expect(fabs(s1) > 1e-3, 1)) {  // 1e-3 < |s1| < inf: one Newton step (2e-15) is enough for a term of a sum
                            asm volatile("");
                            ve = fma(b.AN, cN, -u) * rcp_ode(s1);
                        } else {
                            asm volatile("");
                            ve = u * b.Lr;
                        }
                    } else
                        ve = fma(b.AN, cN, u);
                    vt = ve;  // trap * (exact / trap); exact == trap == 0 when the bin is empty
                } else
                    vt = fma(b.K1, cN, b.K0);
            }
            if (b.bin) {
                // relaxed workgroup atomics on LDS words: ds_add_f64 without a return value, free to overlap the next pair's reads
                __hip_atomic_fetch_add(&sh.D[2 * i + b.j], dNe * ve, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(&sh.E[2 * i + b.j], dNe * vt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        };
        // Main pass.  Energy i needs the lattice nodes i + j, i + j + 1 of the lane's bin -- the nodes of energy i - 1 moved up by
        // one.  So ONE node (16 bytes: correction and its log2) is read per energy, a step ahead of its use, into a ring of three
        // slots; the loop is unrolled over the ring so that the slots are fixed registers.  (This loop is bound by the LDS pipe as
        // much as by the VALU -- profiles/micro/valu_throughput.hip: a ds_add_f64 or a two-word read holds it for 8 cycles -- so
        // the words are not re-read.)  Every lane reads, used or not: an LDS instruction costs the same under any mask; garbage
        // where the cell has no lattice or the index runs past it (still inside this wavefront's LDS), and those lanes never use it.
#ifndef VAG_IC_RING
#define VAG_IC_RING 3  // slots of the node ring: a node is read VAG_IC_RING - 2 energies ahead of its first use
#endif
        constexpr int R = VAG_IC_RING, D = R - 2;
        static_assert(R >= 3 && 64 + D + 64 < IC_MAX_LAT, "ring");
        vdouble2 ring[R];
        if constexpr (WITH_KN) {
#pragma unroll
            for (int q = 0; q <= D; ++q) ring[q] = sh.lat[lane + q];
        }
        const unsigned long long live_i = __ballot(my_dNe > 0);  // energies with electrons: a scalar bit test per energy
        for (int i0 = 0; i0 < g_run; i0 += R) {
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const int i = i0 + u;
                if constexpr (WITH_KN) ring[(u + 1 + D) % R] = sh.lat[i + 1 + D + lane];  // the upper node of energy i + D
                if (i < g_run) {
                    if ((live_i >> i) & 1) {  // uniform
                        const double dNe = read_lane(my_dNe, i);
                        pair(i, dNe, WITH_KN ? __builtin_amdgcn_readlane(my_split, __builtin_amdgcn_readfirstlane(i)) : 0, bm, ring[u % R],
                             ring[(u + 1) % R]);
                    }
                }
            }
        }
        // Tail pass: lane (chunk c, bin MAIN + w) walks the energies c T .. c T + T - 1
        if (n_tail > 0 && g_run > 0) {
            for (int t = 0; t < T; ++t) {
                const int i = i_tail0 + t;
                const bool live = i < g_size;
                const int ii = live ? i : 0;
                const double dNe = live ? sh.dNe_i[ii] : 0.0;
                vdouble2 nd0 = {0, 0}, nd1 = {0, 0};
                if constexpr (WITH_KN) nd0 = sh.lat[ii + bt.j], nd1 = sh.lat[ii + bt.j + 1];
                if (dNe > 0) pair(ii, dNe, WITH_KN ? sh.split_i[ii] : 0, bt, nd0, nd1);
            }
        }
    };
    if (KN)
        energies(std::true_type{});
    else
        energies(std::false_type{});
    wave_sync();
    suffix_scan4(sh.D, lane);  // D[d] <- sum_{d' >= d} D[d']
    wave_sync();
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
        const int kk = lane + 64 * s3;
        if (kk < n_ic) {
            const int d0 = n_lo_i + kk;
            const double above = d0 + 1 <= 0 ? sh.D[0] : (d0 + 1 < IC_MAX_DIAG ? sh.D[d0 + 1] : 0.0);
            const double at = d0 >= 0 && d0 < IC_MAX_DIAG ? sh.E[d0] : 0.0;
            I_acc[s3] = above + at;
        }
    }
    VAG_IC_MARK();  // 5: energy loop
    // log2 table on the output lattice, inverse-compton.h:595-606
    const double lg2_scale = log2(0.25 * C_SIGMAT);
    // vmcnt(0) stated before the table stores (the prefetched words arrived during the walk: no wait in practice): behind this line only
    // stores are in flight, so the next cell's first use of its prefetched constants needs no wait -- left to the compiler, the loop's
    // back edge gets a vmcnt(0) that makes every cell wait for its own stores to reach HBM
#if VAG_IC_PERSISTENT
    __builtin_amdgcn_s_waitcnt(0x0F70);
#endif
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int kk = lane + 64 * s;
        if (kk < n_ic) tab[kk] = log2_fast(I_acc[s]) + (phase + IC_Q * (double)(idx0 + 2L * kk)) + lg2_scale;
    }
#ifdef VAG_IC_STAMPS
    c_t[c_n++] = __builtin_readcyclecounter();  // (no wait for the stores)
    if (lane == 0 && (c % 70001) == 0)
        printf("ic record %lld: g %d nu %d out %d  cycles: prologue %lld  constants %lld  electrons %lld  seeds %lld  thomson cdf %lld  kn lattice %lld  energies %lld  output %lld  total %lld\n",
               c, g_size, nu_size, n_ic, c_t[1] - c_t[0], c_t[2] - c_t[1], c_t[3] - c_t[2], c_t[4] - c_t[3], c_t[5] - c_t[4], c_t[6] - c_t[5],
               c_t[7] - c_t[6], c_t[8] - c_t[7], c_t[8] - c_t[0]);
#endif
    };  // cell()

    long long n_tot;
    {
        const IcPhotonArgs A = load_ic_photon_args();
        n_tot = A.n_run_dev ? (long long)*A.n_run_dev : A.n_run;
    }
    // lane l < CST_N: constant l of a cell (18 of the synchrotron block, 14 IC extras, 14 words of Y(gamma)), placed by its record's words
    auto constants = [&](long long cell_c, long long row0, int nt, int k) -> double {
        const IcPhotonArgs A = load_ic_photon_args();
        const double* src = A.cellpar + row0 * VAG_NPAR + k;  // lanes past the list re-read the first word
        if (lane < CST_Q)
            src += (long long)lane * nt;
        else if (lane < CST_Y)
            src = A.cellq + row0 * VAG_NQ + k + (long long)(lane - CST_Q) * nt;
        else if (lane < CST_N)
            src = A.icy + cell_c + (long long)(lane - CST_Y) * A.n_cells;
        return *src;
    };
#if !VAG_IC_PERSISTENT
    // One wavefront per record.  Its prologue is two dependent round trips -- the record (scalar loads), then the constants -- where the
    // per-cell plan of r04 needed three (plan -> model -> params / meta / offsets -> constants).
    const long long c = blockIdx.x;
    if (c >= n_tot) return;
    typedef const double __attribute__((address_space(4))) ConstDouble;
    ConstDouble* rec = (ConstDouble*)(unsigned long long)(load_ic_photon_args().icplan + (size_t)c * IC_PLAN);
    if ((int)rec[ICP_KN] & 2) return;  // a cell of the slow path (vag_ic_photon_slow_kernel)
    cell(c, constants((long long)rec[ICP_CELL], (long long)rec[ICP_ROW0], (int)rec[ICP_NT], (int)rec[ICP_K]), []() {});
#else
    // Persistent wavefronts (developer build, measured and rejected in r05 -- DESIGN 4i): the launch is as many wavefronts as the device
    // holds, each takes the records w, w + G, ...; the record of c + 2 G and the constants of c + G are requested before the walk of
    // record c, so that a cell starts with its constants in a register and its record in L2.
    const long long G = gridDim.x;
    long long c = blockIdx.x;
    // lane l < IC_PLAN: word l of a record
    auto record = [&](long long cc) -> double {
        const IcPhotonArgs A = load_ic_photon_args();
        return cc < n_tot ? A.icplan[(size_t)cc * IC_PLAN + (lane < IC_PLAN ? lane : 0)] : 0.0;
    };
    auto constants_of = [&](long long cc, double rec) -> double {
        if (cc >= n_tot) return 0.0;  // uniform: past the last record
        return constants((long long)read_lane(rec, ICP_CELL), (long long)read_lane(rec, ICP_ROW0), (int)read_lane(rec, ICP_NT),
                         (int)read_lane(rec, ICP_K));
    };
    double rec1 = record(c);
    double cst0 = constants_of(c, rec1);
    rec1 = record(c + G);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // (as inside the loop: no read in flight on either way into a cell)
    for (; c < n_tot; c += G) {
        double rec2, cst1;
        cell(c, cst0, [&]() {
            // vmcnt(0) stated HERE, where it costs nothing (the cell's reads have been consumed; what may still be in flight is the previous
            // cell's table stores, ~20 k cycles old): the compiler's wait bookkeeping then knows that only the two requests below are
            // outstanding, and does not put a conservative vmcnt(0) -- i.e. a wait for them -- in front of the walk's first LDS reads
            // (registers that earlier global reads of this cell had as their destination).
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt / lgkmcnt untouched
            cst1 = constants_of(c + G, rec1);  // (first: rec1's last use, so that rec2 may take its register without a wait)
            rec2 = record(c + 2 * G);
        });
        cst0 = cst1;
        rec1 = rec2;
        wave_sync();  // the cell's histograms have been read: the next cell may overwrite them
    }
#endif
}

__global__ void __launch_bounds__(64)
vag_ic_photon_slow_kernel(IcPhotonArgs A, const int* __restrict__ slow_list, const unsigned long long* __restrict__ n_slow_dev,
                          int list_cap) {
    const int lane = threadIdx.x;
    constexpr int CST_Q = VAG_NPAR, CST_Y = VAG_NPAR + VAG_NQ, CST_N = VAG_NPAR + VAG_NQ + VAG_NICY;
    __shared__ double s_c[64];
    const long long n_slow = min((long long)*n_slow_dev, (long long)list_cap);
    // suffix sums over the bins [j_lo, j_hi): cdf[j] = sum_{j <= m < j_hi} ex[m], 64 bins per trip from the top
    auto suffix_sums = [&](const double* ex, double* cdf, int j_lo, int j_hi) {
        double carry = 0;
        for (int top = j_hi; top > j_lo; top -= 64) {
            const int j = top - 1 - lane;  // lane 0: the highest bin of the trip
            double v = j >= j_lo ? ex[j] : 0.0;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const double t = __shfl_up(v, off, 64);
                if (lane >= off) v += t;
            }
            v += carry;
            if (j >= j_lo) cdf[j] = v;
            carry = __shfl(v, 63, 64);
        }
    };
    for (long long s = blockIdx.x; s < n_slow; s += gridDim.x) {
        const double* plan = A.icplan + (size_t)slow_list[s] * IC_PLAN;
        const int nu_size = (int)plan[ICP_NU_SIZE], g_size = (int)plan[ICP_G_SIZE], n_ic = (int)plan[ICP_N_IC];
        const long n_lo = (long)plan[ICP_N_LO];
        const double lg2_nu0 = plan[ICP_LG2_NU0], lg2_g0 = plan[ICP_LG2_G0];
        const double lg2_gm = plan[ICP_LG2_GM], inv_gm = plan[ICP_INV_GM], inv_gM = plan[ICP_INV_GMAX];
        const double phase = plan[ICP_PHASE], p = plan[ICP_P];
        const bool KN = ((int)plan[ICP_KN] & 1) != 0;
        const double gamma_m = plan[ICP_GAMMA_M], gamma_c = plan[ICP_GAMMA_C], column_den = plan[ICP_COLUMN_DEN], Y_c = plan[ICP_YC];
        const int regime = (int)plan[ICP_REGIME];
        const long long cell_c = (long long)plan[ICP_CELL], row0 = (long long)plan[ICP_ROW0];
        const int nt = (int)plan[ICP_NT], k = (int)plan[ICP_K];
        const double step = 2 * IC_Q;
        const int nu_last = nu_size - 1, n_lat = (g_size - 1) + (nu_size - 1) + 1;
        double* tab = A.icpool + (unsigned long long)plan[ICP_OFF];
        double* w = tab + n_ic;  // the working arrays
        double *nu = w, *lg2nu = nu + nu_size, *dnu = lg2nu + nu_size, *lg2r = dnu + nu_size, *inv_lg2r = lg2r + nu_size;
        double *fv_th = inv_lg2r + nu_size, *lg2fv = fv_th + nu_size, *ex_th = lg2fv + nu_size, *ratio_th = ex_th + nu_size;
        double *cdf_th = ratio_th + nu_size, *fv_buf = cdf_th + nu_size, *lg2f_buf = fv_buf + nu_size, *ex_buf = lg2f_buf + nu_size;
        double *ratio_buf = ex_buf + nu_size, *cdf_buf = ratio_buf + nu_size;
        double *gam = cdf_buf + nu_size, *dNe = gam + g_size, *corr = dNe + g_size, *lcorr = corr + n_lat, *I_buf = lcorr + n_lat;
        __syncthreads();  // (the previous cell's readers of s_c)
        {   // lane l < CST_N: constant l of the cell (as vag_ic_photon_kernel)
            const double* src = A.cellpar + row0 * VAG_NPAR + k;
            if (lane < CST_Q)
                src += (long long)lane * nt;
            else if (lane < CST_Y)
                src = A.cellq + row0 * VAG_NQ + k + (long long)(lane - CST_Q) * nt;
            else if (lane < CST_N)
                src = A.icy + cell_c + (long long)(lane - CST_Y) * A.n_cells;
            s_c[lane] = *src;
        }
        for (int j = lane; j < nu_size; j += 64) {
            const double x = lg2_nu0 + step * (double)j;
            lg2nu[j] = x;
            nu[j] = exp2_sat(x);
        }
        for (int i = lane; i < g_size; i += 64) gam[i] = exp2_sat(lg2_g0 + step * (double)i);
        for (int q = lane; q < n_ic; q += 64) I_buf[q] = 0;
        __syncthreads();
        double cp[VAG_NPAR];
        for (int q = 0; q < VAG_NPAR; ++q) cp[q] = s_c[q];
        IcQ icq;
        icq.head(s_c + CST_Q, 1);
        icq.rest(s_c + CST_Q, 1);
        const bool y_any = s_c[CST_Y + VY_NSEG] != 0;
        const double yS0 = s_c[CST_Y + VY_S0], yC0 = s_c[CST_Y + VY_C0];
        const double yL1 = s_c[CST_Y + VY_L1], yS1 = s_c[CST_Y + VY_S1], yC1 = s_c[CST_Y + VY_C1];
        const double yL2 = s_c[CST_Y + VY_L2], yS2 = s_c[CST_Y + VY_S2], yC2 = s_c[CST_Y + VY_C2];
        // sample_distributions, inverse-compton.h:371-399 (the same expressions as vag_ic_photon_kernel: the two paths must hand the
        // accumulation identical samples)
        SpecConst sc;
        sc.smooth_thick = plan[ICP_SMOOTH_THICK], sc.log2_x_far = plan[ICP_LOG2_X_FAR];
        const bool slow_c = regime == 1 || regime == 2 || regime == 5, fast_c = regime == 3 || regime == 4 || regime == 6;
        for (int i = lane; i < g_size; i += 64) {
            constexpr double LOG2E_ = 1.4426950408889634;
            const double gi = gam[i], rg = rcp_fast(gi), lg = lg2_g0 + step * (double)i;
            const double dgi = 0.5 * ((i + 1 < g_size ? gam[i + 1] : gi) - (i > 0 ? gam[i - 1] : gi));
            double spec = 0;
            if (slow_c)
                spec = (p - 1) * inv_gm * exp2_sat((-gi * inv_gM - gamma_m * rg) * LOG2E_ - p * (lg - lg2_gm)) * gamma_c *
                       rcp_fast(gi + gamma_c);
            else if (fast_c)
                spec = exp2_sat((-gi * inv_gM - gamma_c * rg) * LOG2E_) * gamma_c * rg * rg *
                       rcp_fast(1.0 + exp2_sat(dmin((p - 1) * (lg - lg2_gm), 1000.0)));
            double den = column_den * spec;
            if (gi > gamma_c) {
                double z = fma(yS0, lg, yC0);
                z = lg >= yL1 ? fma(yS1, lg, yC1) : z;
                z = lg >= yL2 ? fma(yS2, lg, yC2) : z;
                den = den * (1 + Y_c) * rcp_fast(1 + (y_any ? exp2_sat(z) : 0.0));
            }
            dNe[i] = den * rg * rg * dgi;
        }
        for (int j = lane; j < nu_size; j += 64) {
            const double x = lg2nu[j];
            const double lf = log2_I_nu_ic_core<true, true>(cp, 1, icq.applies(x), icq, sc, x, A.sp_table, nu[j]) - 2 * x;
            const double f = exp2_sat(lf);
            fv_th[j] = f;
            lg2fv[j] = f > 0 ? lf : -INFINITY;
        }
        for (int j = lane; j < nu_last; j += 64) {
            dnu[j] = nu[j + 1] - nu[j];
            lg2r[j] = lg2nu[j + 1] - lg2nu[j];
            inv_lg2r[j] = lg2r[j] != 0 ? 1 / lg2r[j] : 0;
        }
        __syncthreads();
        // build_cdf_thomson, inverse-compton.h:415-430
        for (int j = lane; j < nu_last; j += 64) {
            const double trap = 0.5 * (fv_th[j] + fv_th[j + 1]) * dnu[j];
            const double exact = power_law_bin_integral(fv_th[j], fv_th[j + 1], nu[j], nu[j + 1], lg2fv[j], lg2fv[j + 1], lg2r[j],
                                                        inv_lg2r[j], trap);
            ex_th[j] = exact;
            ratio_th[j] = trap > 0 ? exact / trap : 1;
        }
        if (lane == 0) cdf_th[nu_last] = 0;
        if (KN) {  // the KN correction per node of the shared gamma-nu lattice, inverse-compton.h:566-574 (its even nodes: see the fast kernel)
            const double lg2_base = lg2_g0 + lg2_nu0;
            for (int q = lane; q < n_lat; q += 64) {
                double cq, lq;
                compton_correction_pair_lg2(lg2_base + step * (double)q, A.kn_lut, cq, lq);
                corr[q] = cq;
                lcorr[q] = lq;
            }
        }
        __syncthreads();
        suffix_sums(ex_th, cdf_th, 0, nu_last);
        __syncthreads();
        for (int i = 0; i < g_size; ++i) {  // compute_IC_spectrum's loop over the electron energies, :576-592 (all of it wave-uniform)
            const double dNe_i = dNe[i];
            if (!(dNe_i > 0)) continue;
            const double *fv = fv_th, *cdf = cdf_th, *ratio = ratio_th;
            if (KN) {  // build_cdf_KN, :432-481
                const double nu_split = 1e-4 * (C_ME * C_C2 / C_H) / gam[i];
                int below = 0;  // seed nodes short of the last one below the split = the index of the first one at or above it
                for (int j = lane; j < nu_last; j += 64) below += nu[j] < nu_split ? 1 : 0;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) below += __shfl_xor(below, off, 64);
                const int j_split = below;
                for (int j = j_split + lane; j <= nu_last; j += 64) {
                    fv_buf[j] = fv_th[j] * corr[i + j];
                    lg2f_buf[j] = lg2fv[j] + lcorr[i + j];
                }
                if (lane == 0) cdf_buf[nu_last] = 0;
                __syncthreads();
                for (int j = j_split + lane; j < nu_last; j += 64) {
                    const double trap = 0.5 * (fv_buf[j] + fv_buf[j + 1]) * dnu[j];
                    const double exact = power_law_bin_integral(fv_buf[j], fv_buf[j + 1], nu[j], nu[j + 1], lg2f_buf[j], lg2f_buf[j + 1],
                                                                lg2r[j], inv_lg2r[j], trap);
                    ex_buf[j] = exact;
                    ratio_buf[j] = trap > 0 ? exact / trap : 1;
                }
                __syncthreads();
                suffix_sums(ex_buf, cdf_buf, j_split, nu_last);
                __syncthreads();
                if (j_split > 0) {
                    const double delta = cdf_buf[j_split] - cdf_th[j_split];
                    __syncthreads();  // (every lane has read cdf_buf[j_split])
                    for (int j = lane; j < j_split; j += 64) {
                        fv_buf[j] = fv_th[j];
                        ratio_buf[j] = ratio_th[j];
                        cdf_buf[j] = cdf_th[j] + delta;
                    }
                }
                __syncthreads();
                fv = fv_buf, cdf = cdf_buf, ratio = ratio_buf;
            }
            // accumulate_IC, :483-527: output node kk sits at seed node j = n_lo - 2 i + kk (both lattices step by two quanta, so the
            // in-bin offset is always zero: frac = 0, rem = 1, f_seed = f_lo)
            const double cdf0 = cdf[0];
            if (cdf0 > 0) {
                for (int kk = lane; kk < n_ic; kk += 64) {
                    const long j = n_lo - 2L * i + kk;
                    if (j < 0)
                        I_buf[kk] += dNe_i * cdf0;
                    else if (j < nu_last)
                        I_buf[kk] += dNe_i * (cdf[j + 1] + 0.5 * (fv[j] + fv[j + 1]) * dnu[j] * ratio[j]);
                }
            }
            __syncthreads();  // the buffers are rewritten by the next energy
        }
        // log2 table on the output lattice, :595-606
        const double lg2_scale = log2(0.25 * C_SIGMAT);
        const long idx0 = n_lo * 2;
        for (int kk = lane; kk < n_ic; kk += 64) tab[kk] = log2_fast(I_buf[kk]) + (phase + IC_Q * (double)(idx0 + 2L * kk)) + lg2_scale;
    }
}

// ICPhoton::compute_log2_I_nu (inverse-compton.h:614-652) on a stored table: the cell's header {n, first node, last node, log2 of the
// theoretical minimum / maximum, offset} and its n values log2 I on the lattice first + 2 IC_Q q in the pool.  A query outside the clamped
// band but inside the theoretical range would make the reference rebuild the cell's spectrum; here it raises `*breach`.
//
// The reference scans forward to the largest node <= x (clamped to [0, n - 2]) and adds (x - node) * slope with the slope it stored
// per interval.  The lattice being uniform, that interval is floor((x - first) / step) and the interpolant Ilo + frac (Ihi - Ilo)
// with frac = (x - first) / step - interval: the same piecewise-linear function (continuous at the nodes, extrapolating with the end
// intervals' slopes), evaluated with ~2e-14 of rounding in log2 I instead of reproducing the scan's last bit -- 15 instructions and
// ONE 16-byte gather {Ilo, Ihi} where the node-exact form took 45 and two dependent 8-byte gathers (r04: this evaluator is the
// tabulated-SSC flux pass, a third of a configs[4] call).
// `breach` of the evaluators below -> bits of the model's status word: 2 = a query outside the clamped band but inside the
// theoretical range (the tables are rebuilt unclamped), 4 = a query of a cell that was given no table
VAG_DEV int ic_breach_status(int breach) { return ((breach & 1) << 1) | (breach & 4); }
typedef double vdouble2_a8 __attribute__((ext_vector_type(2), aligned(8)));
constexpr double IC_INV_STEP = 1.0 / (2 * IC_Q);
struct IcTabQuery {  // the look-up split at its memory access, for callers that issue the gather ahead of its use
    double frac;
    int idx;
    bool none;  // no table, or x beyond the last node: -inf (compute_log2_I_nu's early return)
};
VAG_DEV IcTabQuery ic_table_query(double h_n, double first, double last, double th_min, double th_max, double x, int* breach) {
    IcTabQuery q;
    const bool empty = h_n < 2.0, above = x > last;
    // (bitwise on purpose: four compares and three scalar mask operations, no branches around single compares)
    *breach |= (!empty & ((above & (x < th_max)) | ((x < first) & (x > th_min)))) ? 1 : 0;
    *breach |= h_n < 0 ? 4 : 0;  // a cell vag_ic_band_kernel declared outside every row's observation window: must not happen
    const double u = (x - first) * IC_INV_STEP;
    const double fl = __builtin_fmin(__builtin_fmax(floor(u), 0.0), __builtin_fmax(h_n - 2.0, 0.0));  // v_max_f64 / v_min_f64, no selects
    q.idx = (int)fl;
    q.frac = u - fl;
    q.none = empty || above;
    return q;
}
VAG_DEV vdouble2_a8 ic_table_gather(const double* __restrict__ tab /* the cell's table in the pool */, int idx) {
    return *reinterpret_cast<const vdouble2_a8*>(tab + idx);
}
VAG_DEV double ic_table_finish(const IcTabQuery& q, vdouble2_a8 I) { return q.none ? -INFINITY : fma(q.frac, I.y - I.x, I.x); }

VAG_DEV double ic_table_eval_hdr(const double* __restrict__ tab, double h_n, double first, double last, double th_min,
                                 double th_max, double x, int* breach) {
    const IcTabQuery q = ic_table_query(h_n, first, last, th_min, th_max, x, breach);
    return ic_table_finish(q, ic_table_gather(tab, q.idx));
}
VAG_DEV double ic_table_eval(const double* __restrict__ hdr, const double* __restrict__ pool, double x, int* breach) {
    return ic_table_eval_hdr(pool + (unsigned long long)hdr[ICH_OFF], hdr[ICH_N], hdr[ICH_FIRST], hdr[ICH_LAST], hdr[ICH_TMIN], hdr[ICH_TMAX],
                             x, breach);
}

}  // namespace vag
