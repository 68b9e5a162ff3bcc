// vag_kernels.h -- dynamics, per-cell radiation and equal-arrival-time flux kernels (gfx950).
#pragma once
#include "vag_device.h"
#include "vag_grid_kernel.h"
#include "vag_dyn_fast.h"

namespace vag {

// Compact storage: model m owns rows [row_off[m], row_off[m+1]) (its representative theta rows) and
// cells [cell_off[m], cell_off[m+1]) = rows x n_t.
struct Layout {
    const int* row_off;        // [nb+1]
    const long long* cell_off; // [nb+1]
};

VAG_DEV int find_model(const int* off, int nb, int idx) {  // largest m with off[m] <= idx
    int lo = 0, hi = nb;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (off[mid] <= idx)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

// Model of cell c (largest m with cell_off[m] <= c) for kernels whose lanes hold CONSECUTIVE cells: the first live lane's cell is looked up
// by bisection on the scalar unit (one chain of scalar-cache loads for the wavefront instead of one chain of vector loads per lane --
// 10-13 dependent trips were a fifth of a wavefront's life in vag_cells_kernel), then each lane steps forward from that model: 64
// consecutive cells span one or two models, rarely more.  Safe under any lane mask (the first live lane holds the smallest cell).
VAG_DEV int cell_model(const long long* __restrict__ cell_off, int nb, long long c) {
#ifndef VAG_HOST_DEBUG
    const long long c_first = ((long long)__builtin_amdgcn_readfirstlane((int)(c >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)c);
#else
    const long long c_first = c;
#endif
    int lo = 0, hi = nb;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (cell_off[mid] <= c_first)
            lo = mid;
        else
            hi = mid;
    }
    while (lo + 1 < nb && cell_off[lo + 1] <= c) ++lo;
    return lo;
}

// ------------------------------------------------------------------------------------------------
// Dynamics: one lane per representative (model, theta) row.  grid_solve_fwd_shock
// (src/dynamics/forward-shock.tpp:175-208) with the lattice generated on the fly, state saved through
// save_fwd_shock_state (forward-shock.tpp:151-173).  shock[VS_*] are SoA arrays over cells.
// ------------------------------------------------------------------------------------------------
template <bool SPREAD, bool INJECT>
__global__ void __launch_bounds__(64)
vag_dynamics_kernel(const vag_model_params* __restrict__ params, int nb, const VagGridMeta* __restrict__ meta,
                    const double* __restrict__ g_theta, const int* __restrict__ g_rep_start,
                    const double* __restrict__ g_tdec, Layout lay, int n_rows, double* __restrict__ shock,
                    long long n_cells, int* __restrict__ row_status, const double* __restrict__ sp_table, int rows_per_wave,
                    int* __restrict__ fail /* [4]: rows per non-zero status, zeroed by vag_grid_kernel */,
                    const double* __restrict__ g_phi /* (phi, theta) pair rows only */, const double* __restrict__ tminmax) {
    __shared__ __attribute__((aligned(16))) double s_lg[LOG_TAB_DOUBLES];  // log2_tab's table for the right-hand sides
    for (int i = threadIdx.x; i < LOG_TAB_DOUBLES; i += 64) s_lg[i] = sp_table[SP_TABLE_DOUBLES + i];
    __syncthreads();
    // a wavefront carries rows_per_wave rows (the host spreads a small batch over the chip: lanes of one wavefront pay for
    // each other's rejected steps and save loops)
    const int row = blockIdx.x * rows_per_wave + threadIdx.x;
    if ((int)threadIdx.x >= rows_per_wave || row >= n_rows || row >= lay.row_off[nb]) return;
    const int m = find_model(lay.row_off, nb, row);
    const VagGridMeta M = meta[m];
    if (M.status != 0) return;
    const int r = row - lay.row_off[m];
    // rows are representative theta rows, or -- Model(axisymmetric=False) with a spreading jet -- (phi i, theta j) pairs
    const int i_phi = M.rep_phi_stride ? r / M.rep_phi_stride : 0;
    const int j = M.rep_phi_stride ? r - i_phi * M.rep_phi_stride : g_rep_start[(size_t)m * M.th_stride + r];
    const vag_model_params P = params[m];
    Jet jet;
    jet_init(jet, P);
    FwdShock<SPREAD, INJECT> eq;
    eq.lg_tab = lds_tab(s_lg);
    medium_init(eq.med, P);
    const double theta0 = g_theta[(size_t)m * M.th_stride + j];
    const double t_dec = g_tdec[((size_t)m * 3 + 0) * M.th_stride + j];
    double t_start_row = g_tdec[((size_t)m * 3 + 1) * M.th_stride + j];
    double t_early_row = g_tdec[((size_t)m * 3 + 2) * M.th_stride + j];
    if (M.rep_phi_stride) {  // the pair's own lattice start: the viewing cosine of (phi_i, theta_j) (grid-refinement.h:462-469,619-625)
        double ts_raw;
        row_time_start(gamma_to_beta(jet_Gamma0(jet, theta0)), cos(theta0), sin(theta0), cos(g_phi[(size_t)m * M.ph_stride + i_phi]),
                       M.cos_obs, M.sin_obs, tminmax[0] * U_SEC, P.z, dmin(0.01 * t_dec, 1e-2 * U_SEC), t_start_row, t_early_row, ts_raw);
    }
    const int nt = M.n_t;
    double* o_teng = shock + VS_TENG * n_cells + lay.cell_off[m] + (long long)r * nt;
    double* o_tcomv = shock + VS_TCOMV * n_cells + lay.cell_off[m] + (long long)r * nt;
    double* o_r = shock + VS_R * n_cells + lay.cell_off[m] + (long long)r * nt;
    double* o_G = shock + VS_GAMMA * n_cells + lay.cell_off[m] + (long long)r * nt;
    double* o_Gth = shock + VS_GAMMA_TH * n_cells + lay.cell_off[m] + (long long)r * nt;
    double* o_B = shock + VS_B * n_cells + lay.cell_off[m] + (long long)r * nt;
    double* o_Np = shock + VS_NP * n_cells + lay.cell_off[m] + (long long)r * nt;
    double* o_th = shock + VS_THETA * n_cells + lay.cell_off[m] + (long long)r * nt;  // written for spreading jets only

    TimeLattice lat;
    lat.init(t_start_row, M.t_end, t_dec, M.t_num_tot);
    auto node = [&](int k) -> double { return M.has_early ? (k == 0 ? t_early_row : lat.node(k - 1)) : lat.node(k); };

    const double Gamma4 = jet_Gamma0(jet, theta0);
    eq.m_jet0 = jet_eps_k(jet, theta0) / Gamma4 / C_C2 / (1 + jet.sigma0);
    eq.gamma_m_coeff = (P.p - 2) / (P.p - 1) * P.eps_e * C_MP / C_ME / P.xi_e;
    eq.gamma_c_coeff = 6 * C_PI * C_ME * C_C / C_SIGMAT / (8 * C_PI * P.eps_B);
    eq.eps_e_eff = P.radiative_fireball ? P.eps_e : 0;
    eq.p = P.p;
    eq.eps_B = P.eps_B;
    eq.theta_s = 0;
    eq.dOmega0 = 1 - cos(theta0);
    eq.inj_L = (theta0 <= jet.theta_c) ? P.mag_L0 * (U_ERG / (4 * C_PI * U_SEC)) : 0.0;  // math::magnetar_injection, jet.h:518-527
    eq.inj_t0 = 1 / (P.mag_t0 * U_SEC);
    eq.inj_q = P.mag_q;
    constexpr int NS = 5 + (SPREAD ? 1 : 0) + (INJECT ? 1 : 0);
    if constexpr (SPREAD) {  // jet_spreading_edge over [theta.front(), theta.back()], grid-refinement.h:113-135
        const double th_min = g_theta[(size_t)m * M.th_stride], th_max = g_theta[(size_t)m * M.th_stride + M.n_theta - 1];
        const double step = (th_max - th_min) / 256;
        double theta_s = th_min, dp_min = 0;
        for (double th = th_min; th <= th_max; th += step) {
            const double lo = dmax(th - step, th_min), hi = dmin(th + step, th_max);
            const double dp = (jet_Gamma0(jet, hi) - jet_Gamma0(jet, lo)) / (hi - lo);
            if (dp < dp_min) {
                dp_min = dp;
                theta_s = th;
            }
        }
        eq.theta_s = (dp_min == 0) ? th_max : theta_s;
    }

    const double t_first = node(0);
    const double t_last = node(nt - 1);
    const double t0 = dmin(t_first, dmin(0.1 * U_SEC, 0.1 * t_dec));
    // set_init_state, forward-shock.tpp:120-149
    double s[NS];
    if constexpr (SPREAD) s[5] = theta0;
    if constexpr (INJECT) s[FwdShock<SPREAD, INJECT>::IDX_EPS] = jet_eps_k(jet, theta0);  // state.eps_jet, forward-shock.tpp:137-139
    const double beta4 = gamma_to_beta(Gamma4);
    s[3] = beta4 * C_C * t0 * Gamma4 * Gamma4 * (1 + beta4);
    s[4] = s[3] / sqrt((Gamma4 - 1) * (Gamma4 + 1)) / C_C;
    s[1] = medium_mass(eq.med, s[3]);
    s[0] = Gamma4;
    s[2] = enclosed_thermal_energy(eq.med, s[3], s[0], adiabatic_idx(s[0]), P.radiative_fireball ? P.eps_e : 0.0);

    if (s[0] <= GAMMA_CUT) {  // set_stopping_shock, shock-physics.h:388-397
        for (int k = 0; k < nt; ++k) {
            o_teng[k] = node(k);
            o_tcomv[k] = s[4];
            o_r[k] = s[3];
            o_G[k] = 1;
            o_Gth[k] = 1;
            o_B[k] = 0;
            o_Np[k] = 0;
            if constexpr (SPREAD) o_th[k] = theta0;
        }
        row_status[row] = 0;
        return;
    }
    // Shock ctor defaults for nodes never reached (shock.cpp:12-24)
    int k = 0;
    double t_k = t_first;
    // Flat attempt loop (Dopri5Flat, r05): a trip of the wavefront is one attempt of every unfinished row; the saves of the lattice nodes
    // an accepted step passed read the candidate before it is committed.  (Until r04 Dopri5::step repeated its attempt inside the call:
    // a wavefront repeated while ANY of its rows rejected.)
    Dopri5Flat<NS> st;
    st.init(s, t0, 0.01 * t0, P.rtol, eq);
    int status = 0, fails = 0, steps = 0;
#ifdef VAG_DYN_STAMPS  // developer aid: cycles of row 0 spent stepping / saving
    long long c_step = 0, c_save = 0, c_mark = __builtin_readcyclecounter();
    const long long c_begin = c_mark;
    int n_steps = 0;
#define VAG_DYN_MARK(acc) do { const long long now_ = __builtin_readcyclecounter(); acc += now_ - c_mark; c_mark = now_; } while (0)
#else
#define VAG_DYN_MARK(acc) do { } while (0)
#endif
    bool done = !(st.t <= t_last);
    while (__any(!done)) {
        if (done) continue;
        if (!st.attempt(eq)) {
            if (++fails >= 500) {  // max_step_checker.hpp:92
                status = 1;
                done = true;
            }
            continue;
        }
        fails = 0;
        VAG_DYN_MARK(c_step);
#ifdef VAG_DYN_STAMPS
        ++n_steps;
#endif
        const double tn = st.t + st.h;
        if (++steps > 100000) {
            status = 2;
            done = true;
        } else {
            while (k < nt && tn > t_k) {
                double q[NS];
                st.interp(t_k, q);
                if constexpr (SPREAD) o_th[k] = q[5];
                // save_fwd_shock_state
                const double comp = compression_fwd(q[0]);
                const double rho = medium_rho(eq.med, q[3]);
                const double Gth = (q[1] == 0) ? 1 : q[2] * rcp_fast(q[1] * C_C2) + 1;
                const double e_th = (Gth - 1) * (rho * comp) * C_C2;
                o_teng[k] = t_k;
                o_tcomv[k] = q[4];
                o_r[k] = q[3];
                o_G[k] = q[0];
                o_Gth[k] = Gth;
                o_B[k] = sqrt_fast(8 * C_PI * P.eps_B * e_th);
                o_Np[k] = q[1] / C_MP;
                ++k;
                if (k < nt) t_k = node(k);
            }
        }
        st.commit();
        if (!(st.t <= t_last)) done = true;
        VAG_DYN_MARK(c_save);
    }
#ifdef VAG_DYN_STAMPS
    if (row == 0)
        printf("dyn row 0: steps %d saves %d  cycles: stepping %lld saving %lld  (init before: see total) \n", n_steps, k, c_step, c_save);
    (void)c_begin;
#endif
    for (; k < nt; ++k) {  // unreached nodes keep the Shock constructor's defaults
        o_teng[k] = node(k);
        o_tcomv[k] = 0;
        o_r[k] = 0;
        o_G[k] = 1;
        o_Gth[k] = 1;
        o_B[k] = 0;
        o_Np[k] = 0;
        if constexpr (SPREAD) o_th[k] = 0;
    }
    row_status[row] = status;
    if (status > 0 && status < 4) atomicAdd(fail + status, 1);
}

// ------------------------------------------------------------------------------------------------
// Same solve for the common case (VagGridMeta::dyn_class == 0 for every model of the batch): vag_dyn_fast.h.  A workgroup is
// two wavefronts over the same rows_per_wave rows -- wavefront 0 integrates (flat attempt loop), wavefront 1 saves (dense
// output at the lattice nodes) -- coupled by an LDS ring.  Writes the interpolated state only: slot VS_GAMMA_TH holds U2_th
// and slot VS_NP holds m2 until vag_cells_kernel(raw_shock) finishes them (Gamma_th, B, N_p of save_fwd_shock_state).
// ------------------------------------------------------------------------------------------------
template <bool TALLY>
__global__ void __launch_bounds__(128)
vag_dynamics_fast_kernel(const vag_model_params* __restrict__ params, int nb, const VagGridMeta* __restrict__ meta,
                         const double* __restrict__ g_theta, const int* __restrict__ g_rep_start,
                         const double* __restrict__ g_tdec, Layout lay, int n_rows, double* __restrict__ shock,
                         long long n_cells, int* __restrict__ row_status, const double* __restrict__ sp_table, int rows_per_wave,
                         int* __restrict__ fail /* [16]: TALLY (vag_ctx_count_work) adds to the three 64-bit counters at fail + 8 */) {
    __shared__ __attribute__((aligned(16))) double s_lg[LOG_TAB_DOUBLES];
    __shared__ DynRing ring;
    __shared__ int s_status[64];
    const int lane = threadIdx.x & 63, role = threadIdx.x >> 6;  // role 0 integrates, role 1 saves
    for (int i = threadIdx.x; i < LOG_TAB_DOUBLES; i += 128) s_lg[i] = sp_table[SP_TABLE_DOUBLES + i];
    if (threadIdx.x == 0) ring.head = ring.tail = ring.fin = 0;
    if (role == 0) s_status[lane] = 0;
    const int row = blockIdx.x * rows_per_wave + lane;
    bool active = lane < rows_per_wave && row < n_rows && row < lay.row_off[nb];
    int m = 0;
    VagGridMeta M = {};
    if (active) {
        m = find_model(lay.row_off, nb, row);
        M = meta[m];
        active = M.status == 0;
    }
    const int nt = M.n_t;
    double s[5] = {2.0, 1.0, 1.0, 1.0, 1.0};
    double t0 = 1, t_last = 0, t_start_row = 1, t_early_row = 1, rtol = 1e-6;
    double *o_teng = nullptr, *o_tcomv = nullptr, *o_r = nullptr, *o_G = nullptr, *o_U = nullptr, *o_m2 = nullptr, *o_B = nullptr;
    TimeLattice lat;
    lat.init(1.0, 2.0, 1.0, 2);
    FsRhs<true> eq;
    eq.lg = lds_tab(s_lg);
    eq.m_jet0 = eq.gm_coeff = eq.inv_gc2 = eq.eps_e = eq.pm2 = eq.rho_ism = 1;
    eq.A = eq.r02 = 0;
    bool stopped = false;
    if (active) {
        const int r = row - lay.row_off[m];
        const int j = g_rep_start[(size_t)m * M.th_stride + r];
        const vag_model_params P = params[m];
        Jet jet;
        jet_init(jet, P);
        Medium med;
        medium_init(med, P);
        const double theta0 = g_theta[(size_t)m * M.th_stride + j];
        const double t_dec = g_tdec[((size_t)m * 3 + 0) * M.th_stride + j];
        t_start_row = g_tdec[((size_t)m * 3 + 1) * M.th_stride + j];
        t_early_row = g_tdec[((size_t)m * 3 + 2) * M.th_stride + j];
        const long long c0 = lay.cell_off[m] + (long long)r * nt;
        o_teng = shock + VS_TENG * n_cells + c0;
        o_tcomv = shock + VS_TCOMV * n_cells + c0;
        o_r = shock + VS_R * n_cells + c0;
        o_G = shock + VS_GAMMA * n_cells + c0;
        o_U = shock + VS_GAMMA_TH * n_cells + c0;
        o_B = shock + VS_B * n_cells + c0;
        o_m2 = shock + VS_NP * n_cells + c0;
        lat.init(t_start_row, M.t_end, t_dec, M.t_num_tot);
        const double Gamma4 = jet_Gamma0(jet, theta0);
        eq.m_jet0 = jet_eps_k(jet, theta0) / Gamma4 / C_C2 / (1 + jet.sigma0);
        eq.gm_coeff = (P.p - 2) / (P.p - 1) * P.eps_e * C_MP / C_ME / P.xi_e;
        eq.inv_gc2 = 2 / (6 * C_PI * C_ME * C_C / C_SIGMAT / (8 * C_PI * P.eps_B));
        eq.eps_e = P.radiative_fireball ? P.eps_e : 0;
        eq.pm2 = P.p > 2 ? P.p - 2 : 0.0;
        eq.rho_ism = med.rho_ism;
        eq.A = med.type == VAG_MEDIUM_ISM ? 0.0 : med.A;
        eq.r02 = med.type == VAG_MEDIUM_ISM ? 0.0 : med.r02;
        rtol = P.rtol;
        auto node0 = [&](int k) -> double { return M.has_early ? (k == 0 ? t_early_row : lat.node(k - 1)) : lat.node(k); };
        const double t_first = node0(0);
        t_last = node0(nt - 1);
        t0 = dmin(t_first, dmin(0.1 * U_SEC, 0.1 * t_dec));
        // set_init_state, forward-shock.tpp:120-149
        const double beta4 = gamma_to_beta(Gamma4);
        s[3] = beta4 * C_C * t0 * Gamma4 * Gamma4 * (1 + beta4);
        s[4] = s[3] / sqrt((Gamma4 - 1) * (Gamma4 + 1)) / C_C;
        s[0] = Gamma4;
        if (role == 0) {  // the integrator's start values (the saver only needs the lattice)
            s[1] = medium_mass(med, s[3]);
            s[2] = enclosed_thermal_energy(med, s[3], s[0], adiabatic_idx(s[0]), P.radiative_fireball ? P.eps_e : 0.0);
        }
        stopped = s[0] <= GAMMA_CUT;  // set_stopping_shock, shock-physics.h:388-397
    }
    auto node = [&](int k) -> double { return M.has_early ? (k == 0 ? t_early_row : lat.node(k - 1)) : lat.node(k); };
    __syncthreads();
#ifdef VAG_DYN_STAMPS
    if (blockIdx.x == 0 && lane == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);  // HW_REG_HW_ID
        printf("  role %d: hw_id 0x%08x wave %u simd %u pipe %u cu %u sh %u se %u\n", role, hw, hw & 15, (hw >> 4) & 3, (hw >> 6) & 3,
               (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7);
    }
#endif
    if (role == 0) {
        int status;
        unsigned long long* tally = reinterpret_cast<unsigned long long*>(fail + 8);
        const bool go = active && !stopped;
        if (__any(go && eq.A != 0)) {
            status = fs_integrator<FsRhs<true>, TALLY>(eq, s, t0, rtol, t_last, go, eq.lg, ring, lane, tally);
        } else {  // every row of this wavefront sits in a uniform medium
            FsRhs<false> ei;
            ei.m_jet0 = eq.m_jet0, ei.gm_coeff = eq.gm_coeff, ei.inv_gc2 = eq.inv_gc2, ei.eps_e = eq.eps_e, ei.pm2 = eq.pm2;
            ei.rho_ism = eq.rho_ism, ei.A = 0, ei.r02 = 0, ei.lg = eq.lg;
            status = fs_integrator<FsRhs<false>, TALLY>(ei, s, t0, rtol, t_last, go, eq.lg, ring, lane, tally);
        }
        s_status[lane] = status;
    } else {
        int k = 0;
        if (active && stopped) {  // raw form of the stopped shock: m2 = 0 finishes to Gamma_th = 1, B = N_p = 0
            for (; k < nt; ++k) {
                o_teng[k] = node(k);
                o_tcomv[k] = s[4];
                o_r[k] = s[3];
                o_G[k] = 1;
                o_U[k] = 0;
                o_B[k] = 0;
                o_m2[k] = 0;
            }
        }
        const int kk = fs_saver(ring, lane, active && !stopped, nt, node, o_teng, o_tcomv, o_r, o_G, o_U, o_m2);
        if (active && !stopped) {
            for (k = kk; k < nt; ++k) {  // unreached nodes keep the Shock constructor's defaults (shock.cpp:12-24)
                o_teng[k] = node(k);
                o_tcomv[k] = 0;
                o_r[k] = 0;
                o_G[k] = 1;
                o_U[k] = 0;
                o_B[k] = 0;
                o_m2[k] = 0;
            }
        }
    }
    __syncthreads();
    if (role == 1 && active) {
        const int status = stopped ? 0 : s_status[lane];
        row_status[row] = status;
        if (status > 0 && status < 4) atomicAdd(fail + status, 1);
    }
}

// ------------------------------------------------------------------------------------------------
// The same solve for batches that fill the GPU several times over: persistent single-wavefront workgroups whose lanes take rows from a
// device counter and save inline (vag_dyn_fast.h, "lane refill").  vag_dyn_prep_kernel prepares every row's start -- one lane per row,
// the expressions of the kernel above -- as a 160-byte record, writes the row's node times (an output of the stage) and finishes the
// rows that need no solve (a stopped shock); vag_dyn_scan_kernel / vag_dyn_file_kernel order the row queue by predicted length (counting
// sort over the preparation's per-chunk histograms); vag_dynamics_refill_kernel integrates and saves.
// ------------------------------------------------------------------------------------------------
// one row's start record, node times and -- for a stopped shock -- its whole output; returns the row's length class, or -1 for a row
// the solver has nothing to do for
VAG_DEV int dyn_prep_row(const vag_model_params* __restrict__ params, int nb, const VagGridMeta* __restrict__ meta,
                         const double* __restrict__ g_theta, const int* __restrict__ g_rep_start, const double* __restrict__ g_tdec,
                         Layout lay, int n_rows, double* __restrict__ shock, long long n_cells, int* __restrict__ row_status,
                         double* __restrict__ rowrec, int row) {
    if (row >= n_rows) return -1;
    double* rec = rowrec + (size_t)row * DYN_ROWREC;
    if (row >= lay.row_off[nb]) {
        rec[DR_NT] = __hiloint2double(2, 0);
        return -1;
    }
    const int m = find_model(lay.row_off, nb, row);
    const VagGridMeta M = meta[m];
    if (M.status != 0) {
        rec[DR_NT] = __hiloint2double(2, 0);
        return -1;
    }
    const int nt = M.n_t;
    const int r = row - lay.row_off[m];
    const int j = g_rep_start[(size_t)m * M.th_stride + r];
    const vag_model_params P = params[m];
    Jet jet;
    jet_init(jet, P);
    Medium med;
    medium_init(med, P);
    const double theta0 = g_theta[(size_t)m * M.th_stride + j];
    const double t_dec = g_tdec[((size_t)m * 3 + 0) * M.th_stride + j];
    const double t_start_row = g_tdec[((size_t)m * 3 + 1) * M.th_stride + j];
    const double t_early_row = g_tdec[((size_t)m * 3 + 2) * M.th_stride + j];
    const long long c0 = lay.cell_off[m] + (long long)r * nt;
    TimeLattice lat;
    lat.init(t_start_row, M.t_end, t_dec, M.t_num_tot);
    const double Gamma4 = jet_Gamma0(jet, theta0);
    rec[DR_EQ + 0] = jet_eps_k(jet, theta0) / Gamma4 / C_C2 / (1 + jet.sigma0);
    rec[DR_EQ + 1] = (P.p - 2) / (P.p - 1) * P.eps_e * C_MP / C_ME / P.xi_e;
    rec[DR_EQ + 2] = 2 / (6 * C_PI * C_ME * C_C / C_SIGMAT / (8 * C_PI * P.eps_B));
    rec[DR_EQ + 3] = P.radiative_fireball ? P.eps_e : 0;
    rec[DR_EQ + 4] = P.p > 2 ? P.p - 2 : 0.0;
    rec[DR_EQ + 5] = med.rho_ism;
    rec[DR_EQ + 6] = med.type == VAG_MEDIUM_ISM ? 0.0 : med.A;
    rec[DR_EQ + 7] = med.type == VAG_MEDIUM_ISM ? 0.0 : med.r02;
    auto node = [&](int k) -> double { return M.has_early ? (k == 0 ? t_early_row : lat.node(k - 1)) : lat.node(k); };
    const double t_first = node(0);
    const double t_last = node(nt - 1);
    const double t0 = dmin(t_first, dmin(0.1 * U_SEC, 0.1 * t_dec));
    // set_init_state, forward-shock.tpp:120-149
    double s[5];
    const double beta4 = gamma_to_beta(Gamma4);
    s[3] = beta4 * C_C * t0 * Gamma4 * Gamma4 * (1 + beta4);
    s[4] = s[3] / sqrt((Gamma4 - 1) * (Gamma4 + 1)) / C_C;
    s[0] = Gamma4;
    s[1] = medium_mass(med, s[3]);
    s[2] = enclosed_thermal_energy(med, s[3], s[0], adiabatic_idx(s[0]), P.radiative_fireball ? P.eps_e : 0.0);
    const bool stopped = s[0] <= GAMMA_CUT;  // set_stopping_shock, shock-physics.h:388-397
#pragma unroll
    for (int i = 0; i < 5; ++i) rec[DR_X + i] = s[i];
    rec[DR_T0] = t0;
    rec[DR_TLAST] = t_last;
    rec[DR_RTOL] = P.rtol;
    rec[DR_NT] = __hiloint2double(stopped ? 2 : 0, nt);
    rec[DR_C0] = __longlong_as_double(c0);
    double* o = shock + c0;
    for (int k = 0; k < nt; ++k) o[VS_TENG * n_cells + k] = node(k);
    if (stopped) {  // raw form of the stopped shock: m2 = 0 finishes to Gamma_th = 1, B = N_p = 0
        for (int k = 0; k < nt; ++k) {
            o[VS_TCOMV * n_cells + k] = s[4];
            o[VS_R * n_cells + k] = s[3];
            o[VS_GAMMA * n_cells + k] = 1;
            o[VS_GAMMA_TH * n_cells + k] = 0;
            o[VS_B * n_cells + k] = 0;
            o[VS_NP * n_cells + k] = 0;
        }
        row_status[row] = 0;
        return -1;
    }
    return dyn_row_bucket(Gamma4, fma(rec[DR_EQ + 6], 1.0 / fma(s[3], s[3], rec[DR_EQ + 7]), med.rho_ism), rec[DR_EQ + 0]);
}

__global__ void __launch_bounds__(DYN_CHUNK)
vag_dyn_prep_kernel(const vag_model_params* __restrict__ params, int nb, const VagGridMeta* __restrict__ meta,
                    const double* __restrict__ g_theta, const int* __restrict__ g_rep_start, const double* __restrict__ g_tdec,
                    Layout lay, int n_rows, double* __restrict__ shock, long long n_cells, int* __restrict__ row_status,
                    double* __restrict__ rowrec, signed char* __restrict__ cls /* [rows] length class of the row, -1: not in the queue */,
                    unsigned* __restrict__ hist /* [DYN_BUCKETS][chunks] rows of the class in the chunk */) {
    __shared__ unsigned s_hist[DYN_BUCKETS];
    if (threadIdx.x < DYN_BUCKETS) s_hist[threadIdx.x] = 0;
    __syncthreads();
    const int row = blockIdx.x * DYN_CHUNK + threadIdx.x;
    const int bucket = dyn_prep_row(params, nb, meta, g_theta, g_rep_start, g_tdec, lay, n_rows, shock, n_cells, row_status, rowrec, row);
    if (row < n_rows) cls[row] = (signed char)bucket;
    if (bucket >= 0) atomicAdd(&s_hist[bucket], 1u);
    __syncthreads();
    if (threadIdx.x < DYN_BUCKETS) hist[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = s_hist[threadIdx.x];
}

// exclusive offsets of the (class, chunk) groups in queue order -- longest class first, chunks ascending inside a class -- in place;
// queue[0] = rows in the queue, queue[1] = 0 (rows taken).  One workgroup: the histogram is 16 x (rows / 256) counts.
__global__ void __launch_bounds__(1024)
vag_dyn_scan_kernel(unsigned* __restrict__ hist, int n_chunks, unsigned* __restrict__ queue) {
    __shared__ unsigned s_part[1024];
    const int n = DYN_BUCKETS * n_chunks, tid = threadIdx.x;
    // element e of the scan = (class DYN_BUCKETS - 1 - e / n_chunks, chunk e % n_chunks); a thread owns a contiguous run of elements
    const int per = (n + 1023) / 1024, e0 = tid * per, e1 = min(n, e0 + per);
    auto at = [&](int e) -> unsigned& { return hist[(size_t)(DYN_BUCKETS - 1 - e / n_chunks) * n_chunks + e % n_chunks]; };
    unsigned sum = 0;
    for (int e = e0; e < e1; ++e) sum += at(e);
    s_part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele over the 1024 partial sums
        const unsigned v = tid >= off ? s_part[tid - off] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    unsigned run = s_part[tid] - sum;  // exclusive
    for (int e = e0; e < e1; ++e) {
        const unsigned c = at(e);
        at(e) = run;
        run += c;
    }
    if (tid == 1023) queue[0] = s_part[1023], queue[1] = 0;
}

__global__ void __launch_bounds__(DYN_CHUNK)
vag_dyn_file_kernel(const signed char* __restrict__ cls, const unsigned* __restrict__ offs /* [DYN_BUCKETS][chunks] */, int n_rows,
                    int* __restrict__ order) {
    __shared__ unsigned s_next[DYN_BUCKETS];
    if (threadIdx.x < DYN_BUCKETS) s_next[threadIdx.x] = offs[(size_t)threadIdx.x * gridDim.x + blockIdx.x];
    __syncthreads();
    const int row = blockIdx.x * DYN_CHUNK + threadIdx.x;
    const int b = row < n_rows ? cls[row] : -1;
    if (b >= 0) order[atomicAdd(&s_next[b], 1u)] = row;  // (the order inside a (class, chunk) group is the atomics': rows are independent)
}

#ifndef VAG_DYN_REFILL_WAVES
#define VAG_DYN_REFILL_WAVES 2  // wavefronts per SIMD the register allocation aims at
#endif
template <bool TALLY>
__global__ void __launch_bounds__(64, VAG_DYN_REFILL_WAVES)
vag_dynamics_refill_kernel(const double* __restrict__ rowrec, int n_rows, double* __restrict__ shock, long long n_cells,
                           int* __restrict__ row_status, const double* __restrict__ sp_table, int refill_min,
                           int* __restrict__ fail /* [16]: [1..3] failures, 64-bit tallies at fail + 8 */,
                           unsigned* __restrict__ queue, const int* __restrict__ order) {
    __shared__ __attribute__((aligned(16))) double s_lg[LOG_TAB_DOUBLES];
    const int lane = threadIdx.x;
    for (int i = lane; i < LOG_TAB_DOUBLES; i += 64) s_lg[i] = sp_table[SP_TABLE_DOUBLES + i];
    __syncthreads();
#ifdef VAG_DYN_PLACEMENT  // developer aid: where the dispatcher put the persistent wavefronts
    if (lane == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);   // HW_REG_HW_ID
        const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);  // HW_REG_XCC_ID
        printf("P %d %d xcc %u se %u sh %u cu %u simd %u wave %u\n", (int)blockIdx.x, 0, xcc & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15,
               (hw >> 4) & 3, hw & 15);
    }
#endif
    fs_solver_refill<TALLY>(rowrec, n_rows, queue, order, refill_min, lds_tab(s_lg), lane, shock, n_cells, row_status, fail,
                            reinterpret_cast<unsigned long long*>(fail + 8));
}

// ------------------------------------------------------------------------------------------------
// Spreading jets: per-cell polar geometry of the equal-arrival-time step (Observer::calc_t_obs +
// calc_solid_angle, src/core/observer.cpp:51-141): cos / sin of the evolved theta and log2 |cos th_hi - cos th_lo|,
// where the bin edges are midpoints to the neighbouring rows' theta interpolated at the same engine time.
// One lane per (row, k) cell; cellgeo is [row][3][n_t].
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
vag_spread_geo_kernel(int nb, const VagGridMeta* __restrict__ meta, Layout lay, const double* __restrict__ shock,
                      long long n_cells, double* __restrict__ cellgeo) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cells || c >= lay.cell_off[nb]) return;  // n_cells is the arrays' stride (>= the batch's cell count)
    const int m = cell_model(lay.cell_off, nb, c);
    const VagGridMeta M = meta[m];
    if (M.status != 0) return;
    const int nt = M.n_t;
    const long long local = c - lay.cell_off[m];
    const int row = (int)(local / nt), k = (int)(local % nt);  // structured symmetry: row index == theta index ...
    const int j = M.rep_phi_stride ? row % M.rep_phi_stride : row;  // ... within the row's phi slice ((phi, theta) pair rows)
    const int last = M.n_theta - 1;
    // the neighbours j - 1, j + 1 belong to the same phi slice (calc_solid_angle, observer.cpp:103-131): offset the slice's rows
    const double* teng = shock + VS_TENG * n_cells + lay.cell_off[m] + (long long)(row - j) * nt;
    const double* theta = shock + VS_THETA * n_cells + lay.cell_off[m] + (long long)(row - j) * nt;
    const double th = theta[(long long)j * nt + k];
    const double t_target = teng[(long long)j * nt + k];
    auto interp_theta = [&](int j_nb) {  // theta of row j_nb at engine time t_target (the reference walks a hint forward)
        const double* tn = teng + (long long)j_nb * nt;
        const double* thn = theta + (long long)j_nb * nt;
        // largest h with h == 0 or tn[h] < t_target, capped so that h + 1 exists
        int a = 0, b = nt - 1;
        while (b - a > 1) {
            const int mid = (a + b) >> 1;
            if (tn[mid] < t_target)
                a = mid;
            else
                b = mid;
        }
        int h = a;
        if (h + 1 < nt && tn[h + 1] < t_target) h = h + 1;
        if (h + 1 >= nt) return thn[nt - 1];
        const double w = (t_target - tn[h]) / (tn[h + 1] - tn[h]);
        return thn[h] + w * (thn[h + 1] - thn[h]);
    };
    const double th_lo = (j == 0) ? th : 0.5 * (th + interp_theta(j - 1));
    const double th_hi = (j == last) ? th : 0.5 * (th + interp_theta(j + 1));
    double* dst = cellgeo + (lay.cell_off[m] + (long long)row * nt) * 3 + k;
    dst[0] = cos(th);
    dst[(long long)nt] = sin(th);
    dst[2LL * nt] = log2(fabs(cos(th_hi) - cos(th_lo)));
}

// ------------------------------------------------------------------------------------------------
// Per-cell radiation: generate_syn_electrons + generate_syn_photons + SmoothPowerLawSyn::build,
// one lane per (row, k) cell.  Output: [row][VAG_NPAR][n_t] blocks in `cellpar`.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
vag_cells_kernel(const vag_model_params* __restrict__ params, int nb, const VagGridMeta* __restrict__ meta, Layout lay,
                 const double* shock, long long n_cells, double* __restrict__ cellpar,
                 double* __restrict__ cell_details /* optional [11][n_cells] */,
                 const int* __restrict__ inj_idx /* optional, per row: reverse shock's injection cutoff */,
                 double* raw_shock /* = shock when vag_dynamics_fast_kernel left (U2_th, m2) to be finished */,
                 bool write_back /* raw_shock: store the finished Gamma_th, B, N_p (read again by the IC cooling pass and by vag_details;
                                    a plain synchrotron call reads them nowhere else -- this kernel is bound by HBM, 28 doubles per cell with them) */) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cells || c >= lay.cell_off[nb]) return;  // n_cells is the arrays' stride (>= the batch's cell count)
    const int m = cell_model(lay.cell_off, nb, c);
    const VagGridMeta M = meta[m];
    if (M.status != 0) return;
    const int nt = M.n_t;
    const long long local = c - lay.cell_off[m];
    const int r = (int)(local / nt), k = (int)(local % nt);
    const vag_model_params P = params[m];
    double c_Gth, c_B, c_Np;
    if (!raw_shock) {
        c_Gth = shock[VS_GAMMA_TH * n_cells + c], c_B = shock[VS_B * n_cells + c], c_Np = shock[VS_NP * n_cells + c];
    } else {  // save_fwd_shock_state (forward-shock.tpp:151-173) on the interpolated state of this cell
        const double m2 = shock[VS_NP * n_cells + c], U = shock[VS_GAMMA_TH * n_cells + c];
        c_Gth = 1, c_B = 0, c_Np = 0;
        if (m2 != 0) {
            Medium med;
            medium_init(med, P);
            const double comp = compression_fwd(shock[VS_GAMMA * n_cells + c]);
            const double rho = medium_rho(med, shock[VS_R * n_cells + c]);
            c_Gth = U * rcp_fast(m2 * C_C2) + 1;
            const double e_th = (c_Gth - 1) * (rho * comp) * C_C2;
            c_B = sqrt_fast(8 * C_PI * P.eps_B * e_th);
            c_Np = m2 / C_MP;
        }
        if (write_back) {
            raw_shock[VS_GAMMA_TH * n_cells + c] = c_Gth;
            raw_shock[VS_B * n_cells + c] = c_B;
            raw_shock[VS_NP * n_cells + c] = c_Np;
        }
    }
    CellOut o;
    ElecBasic inj{0, 1, 0};
    bool relic = false;
    if (inj_idx) {  // Shock::is_relic, shock.h:56
        const int k_inj = inj_idx[lay.row_off[m] + r];
        if (k >= k_inj) {
            const long long ci = c - k + (k_inj - 1);
            inj = syn_elec_basic(shock[VS_TCOMV * n_cells + ci], shock[VS_GAMMA_TH * n_cells + ci], shock[VS_B * n_cells + ci],
                                 P.eps_e, P.p, P.xi_e);
            relic = true;
        }
    }
    syn_cell(o, shock[VS_TENG * n_cells + c], shock[VS_TCOMV * n_cells + c], shock[VS_R * n_cells + c],
             shock[VS_GAMMA * n_cells + c], c_Gth, c_B, c_Np, P.eps_e, P.p, P.xi_e, relic, inj);
    double* dst = cellpar + (lay.cell_off[m] + (long long)r * nt) * VAG_NPAR + k;
#pragma unroll
    for (int q = 0; q < VAG_NPAR; ++q) dst[(long long)q * nt] = o.par[q];
    if (cell_details) {
        const double v[11] = {o.gamma_m, o.gamma_c, o.gamma_a, o.gamma_M, o.N_e, o.column_den,
                              o.nu_m, o.nu_c, o.nu_a, o.nu_M, o.I_nu_max};
#pragma unroll
        for (int q = 0; q < 11; ++q) cell_details[q * n_cells + c] = v[q];
    }
}

// ------------------------------------------------------------------------------------------------
// Equal-arrival-time flux integration (Observer::observe + Observer::specific_flux,
// src/core/observer.cpp:143-205,439-454 and src/core/observer.h:355-445), fused:
//
// A workgroup owns a contiguous range of (theta j, phi i) "pairs" of one model (j-major).  The photon
// parameter rows of the representative row of j are staged in LDS once per j-group.  For each pair:
//   A0  lanes over k:        Doppler, observer time and geometry logs of the row  -> LDS
//   A1  lanes over (k, l):   boundary log2-luminosities B[k][l] inside the observation window -> LDS
//   B   lanes over (idx, l): log-log interpolation at the requested times, exp2, accumulate in registers
// Each lane owns fixed (idx, l) output slots, so the (phi, theta) sum needs no cross-lane reduction; a
// workgroup writes one partial grid, reduced deterministically by vag_reduce_kernel.
// ------------------------------------------------------------------------------------------------
constexpr int FLUX_THREADS = 512;
constexpr int FLUX_MAX_SLOTS = 8192;  // (l, idx) output slots per launch, accumulated in LDS  // (l, idx) slots per lane: nt * nnu <= FLUX_THREADS * FLUX_MAX_SLOTS

struct FluxArgs {
    const vag_model_params* params;
    const VagGridMeta* meta;
    const double* geo_th;  // [nb][3][VAG_MAX_THETA]
    const double* geo_ph;  // [nb][2][VAG_MAX_PHI]
    const int* g_rep_of;
    const long long* cell_off;
    const double* cellpar;
    const double* lg2_t_obs;  // [nt]   log2(t * unit::sec)
    const double* lg2_nu_obs; // [nnu]  log2(nu * unit::Hz)
    const double* sp_table;   // [SP_TABLE_DOUBLES] softplus interpolant (vag_device.h: sp_fast)
    double* partial;          // [nb][max_blocks][nnu*nt]
    double* partial2;         // FLUX_FUSED: the SSC component's partial grids, same shape
    int nt, nnu;
    int pairs_per_block;
    int max_blocks;  // blocks per model (gridDim.x)
    int k_stride;    // LDS row stride (>= max n_t in the batch)
    unsigned long long* work_count;  // optional [2]: exact spectrum evaluations / interpolations done (instrumentation)
    // SSC tier (vag_ic_kernels.h)
    const double* cellq;   // [rows][VAG_NQ][n_t] IC-correction constants of the synchrotron spectrum (MODE 1)
    const double* ichdr;   // [cells][IC_HDR] SSC table headers (MODE 2)
    const double* icpool;  // the tables, back to back (a header holds its table's offset)
    int* ic_status;        // [nb] bit 2: band-contract breach seen by the SSC flux pass
    const double* cellgeo; // [rows][3][n_t] per-cell cos(theta), sin(theta), log2|dcos| of a spreading jet (SPREAD kernels)
    const double* rowgeo;  // [nb][rowgeo_stride] row-geometry records written by vag_grid_kernel (read by the non-spreading kernels)
    int rowgeo_stride;
};

constexpr int ROWGEO_HDR = VAG_ROWGEO_HDR;  // row-geometry records of a model: written by vag_grid_kernel (vag_grid_kernel.h), layout there

// photon source of the flux kernels
constexpr int FLUX_SYN = 0;     // synchrotron, no inverse-Compton cooling
constexpr int FLUX_SYN_IC = 1;  // synchrotron with the IC correction above nu_c
constexpr int FLUX_SSC = 2;     // SSC tables
constexpr int FLUX_FUSED = 3;   // FLUX_SYN_IC and FLUX_SSC in one pass: EAT logs, bracket search and barriers paid once

template <class P1, class P2, class Tab>
VAG_DEV double log2_I_nu_ic(const P1 c, int st, const P2 qv, int qst, const SpecConst& sc, double lg2_nu, Tab sp);
template <class P1, class P2, class Tab>
VAG_DEV void log2_I_nu_ic_pair(const P1 c, int st, const P2 qv, int qst, const SpecConst& sc, double x0, double x1, Tab sp, double& b0,
                               double& b1);
VAG_DEV double ic_table_eval(const double* __restrict__ hdr, const double* __restrict__ pool, double x, int* breach);
VAG_DEV double ic_table_eval_hdr(const double* __restrict__ tab, double h_n, double first, double last, double th_min,
                                 double th_max, double x, int* breach);
VAG_DEV int ic_breach_status(int breach);
constexpr int FLUX_NQ = 14;          // == VAG_NQ (vag_ic_kernels.h)
constexpr int FLUX_IC_HDR = 8;       // == IC_HDR: doubles per cell header {n, first, last, theory min / max, pool offset, 2 spare}

// Observation window of a row from the partial counts its EAT step leaves in LDS (WinCount below): the row is sorted, so
// positions are counts (observed_window, observer.h:324-338):
//   n_lt = #{k : t[k] <  w_lo}  ->  k_lo = max(n_lt - 1, 0)                (last k with t[k+1] < w_lo)
//   n_le = #{k : t[k] <= w_hi}  ->  k_hi = clamp(n_le, k_lo + 1, K - 1)    (first node > w_hi)
// A wavefront that evaluates the EAT logs of some nodes counts them with two ballots and its lane 0 adds the pair to the row's
// two counters in LDS (`win`: integer LDS atomics, so the order does not matter); the counters of a buffer are cleared by the
// flux kernel once every wavefront has read them.
struct WinCount {
    int n_lt = 0, n_le = 0;
    VAG_DEV void add(bool valid, double t, double w_lo, double w_hi) {
#ifndef VAG_HOST_DEBUG
        n_lt += __popcll(__ballot(valid && t < w_lo));
        n_le += __popcll(__ballot(valid && t <= w_hi));
#endif
    }
    VAG_DEV void store(int* win, int tid) const {
#ifndef VAG_HOST_DEBUG
        if ((tid & 63) == 0 && (n_lt | n_le) != 0) {
            __hip_atomic_fetch_add(win, n_lt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(win + 1, n_le, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
#endif
    }
};

// Uniform values straight into scalar registers.  The compiler reads the per-row geometry with vector loads (it cannot prove that
// the kernel's own global stores do not alias it): 500-cycle round trips that every wavefront sat out at the start of each
// row's second interval.  s_load through the scalar cache instead (the records were written by an earlier kernel).
#ifndef VAG_HOST_DEBUG
VAG_DEV int sload_i32(const int* p) {
    int v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(v) : "s"(p) : "memory");
    return v;
}
typedef int vag_v8i __attribute__((ext_vector_type(8)));
typedef int vag_v4i __attribute__((ext_vector_type(4)));
struct RowGeo {
    double cos_th, sin_th, dth, cph, dph, cos_obs, sin_obs;
    int rep;
    // cos of the angle to the line of sight, sin th cos ph sin th_obs + cos th cos th_obs (observer.cpp:143-160), with its roundings
    // spelled out: left to the compiler's contraction the instantiations of the flux kernel round it differently, and a model's
    // fluxes must not depend on which of them served it
    VAG_DEV double cos_view() const { return fma(cos_th, cos_obs, (sin_th * cph) * sin_obs); }
};
// row (theta j, phi i) from the model's records: three scalar loads behind one base (offsets in scalar registers)
VAG_DEV RowGeo sload_rowgeo(const double* rg, int th_byte, int j, int i) {
    vag_v8i t;
    vag_v4i p, h;
    asm volatile("s_load_dwordx8 %0, %3, %4\n\ts_load_dwordx4 %1, %3, %5\n\ts_load_dwordx4 %2, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(t), "=&s"(p), "=&s"(h)
                 : "s"(rg), "s"(th_byte + 32 * j), "s"(ROWGEO_HDR * 8 + 16 * i)
                 : "memory");
    RowGeo g;
    g.cos_th = __hiloint2double(t[1], t[0]), g.sin_th = __hiloint2double(t[3], t[2]), g.dth = __hiloint2double(t[5], t[4]);
    g.rep = t[6];
    g.cph = __hiloint2double(p[1], p[0]), g.dph = __hiloint2double(p[3], p[2]);
    g.cos_obs = __hiloint2double(h[1], h[0]), g.sin_obs = __hiloint2double(h[3], h[2]);
    return g;
}
#else
struct RowGeo {
    double cos_th, sin_th, dth, cph, dph, cos_obs, sin_obs;
    int rep;
    // cos of the angle to the line of sight, sin th cos ph sin th_obs + cos th cos th_obs (observer.cpp:143-160), with its roundings
    // spelled out: left to the compiler's contraction the instantiations of the flux kernel round it differently, and a model's
    // fluxes must not depend on which of them served it
    VAG_DEV double cos_view() const { return fma(cos_th, cos_obs, (sin_th * cph) * sin_obs); }
};
VAG_DEV RowGeo sload_rowgeo(const double* rg, int th_byte, int j, int i) {
    const double* t = rg + th_byte / 8 + 4 * j;
    const double* p = rg + ROWGEO_HDR + 2 * i;
    RowGeo g;
    g.cos_th = t[0], g.sin_th = t[1], g.dth = t[2], g.rep = __double2loint(t[3]);
    g.cph = p[0], g.dph = p[1], g.cos_obs = rg[0], g.sin_obs = rg[1];
    return g;
}
VAG_DEV int sload_i32(const int* p) { return *p; }
#endif

// EAT quantities of one (theta j, phi i) row: Doppler, observer time and geometry logs
// (calc_eat_non_spreading + finalize_log_grids, observer.cpp:143-205,439-454) -> LDS.  s_par holds the staged row as
// [k][VAG_NPAR] blocks (144 B apart: conflict-free 16-byte LDS reads, one address per cell).  win != null: also the row's
// observation-window counts against [w_lo, w_hi] (WinCount; the caller has cleared the two counters).
VAG_DEV void eat_row(const double* __restrict__ s_par, int KS, int K, int tid, int nthreads, double cos_v,
                     double t_coeff, double one_plus_z, double lg2_dOmega, double* __restrict__ s_t,
                     double* __restrict__ s_dop, double* __restrict__ s_geom, LdsTab lg, int* win = nullptr, double w_lo = 0,
                     double w_hi = 0) {
    WinCount wc;
    for (int k = tid; k < K; k += nthreads) {
        const double* c = s_par + k * VAG_NPAR;
        const LdsTab c2 = lds_tab(c);
        const vdouble2 Gu = c2[VP_GAMMA / 2], rt = c2[VP_R / 2];
        const double G = Gu.x, u = Gu.y, r = rt.x;
#if defined(VAG_FLUX_ABLATE) && (VAG_FLUX_ABLATE & 8)
        s_dop[k] = G, s_t[k] = 3.0 + 0.1 * k + u, s_geom[k] = r;
        continue;
#endif
        const double lg2_dop = -log2_tab(fma(-u, cos_v, G), lg);  // explicit roundings: the flux kernels form these in two places
        const double time = fma(t_coeff, r, rt.y * one_plus_z);
        const double lt = log2_tab(time, lg);
        s_dop[k] = lg2_dop;
        s_t[k] = lt;
        s_geom[k] = (lg2_dOmega + c[VP_LG2_R2]) + 3.0 * lg2_dop;
        if (win) wc.add(true, lt, w_lo, w_hi);
    }
    if (win) wc.store(win, tid);
}

// Same for a spreading jet (calc_t_obs + calc_solid_angle, observer.cpp:51-141): theta evolves along k, so the viewing
// cosine and the solid angle are per cell; `geo` = this row's [3][K] block (cos theta, sin theta, log2|dcos|) in L2.
VAG_DEV void eat_row_spread(const double* __restrict__ s_par, int KS, int K, int tid, int nthreads,
                            const double* __restrict__ geo, double cos_phi, double sin_obs, double cos_obs, double lg2_dphi,
                            double one_plus_z, double* __restrict__ s_t, double* __restrict__ s_dop,
                            double* __restrict__ s_geom, LdsTab lg, int GS /* stride of geo's three rows: the row's whole lattice */,
                            int* win = nullptr, double w_lo = 0, double w_hi = 0) {
    WinCount wc;
    for (int k = tid; k < K; k += nthreads) {
        const double* c = s_par + k * VAG_NPAR;
        const LdsTab c2 = lds_tab(c);
        const vdouble2 Gu = c2[VP_GAMMA / 2], rt = c2[VP_R / 2];
        const double G = Gu.x, u = Gu.y, r = rt.x;
        const double cos_v = geo[GS + k] * cos_phi * sin_obs + geo[k] * cos_obs;
        const double lg2_dop = -log2_tab(G - u * cos_v, lg);
        const double time = (rt.y + (1 - cos_v) * r / C_C) * one_plus_z;
        const double lt = log2_tab(time, lg);
        s_dop[k] = lg2_dop;
        s_t[k] = lt;
        s_geom[k] = ((geo[2 * GS + k] + lg2_dphi) + c[VP_LG2_R2]) + 3.0 * lg2_dop;
        if (win) wc.add(true, lt, w_lo, w_hi);
    }
    if (win) wc.store(win, tid);
}

// Model.jet_E_iso / jet_Gamma0 / medium (pybind/pymodel.cpp:572-594): the engine's own profile functions on n abscissae.
// kind 0 -> E_iso(theta) [erg], 1 -> Gamma0(theta), 2 -> rho(r [cm]) [g/cm^3].
__global__ void __launch_bounds__(256)
vag_profile_kernel(const vag_model_params* __restrict__ params, int kind, const double* __restrict__ x, int n,
                   double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const vag_model_params P = params[0];
    if (kind == 2) {
        Medium med;
        medium_init(med, P);
        out[i] = medium_rho(med, x[i] * U_CM) / (U_G / U_CM3);
    } else {
        Jet jet;
        jet_init(jet, P);
        out[i] = kind == 0 ? jet_eps_k(jet, x[i]) / (U_ERG / (4 * C_PI)) : jet_Gamma0(jet, x[i]);
    }
}

// Model.details(): observer time [s] and Doppler factor of every (phi, theta, k) cell of model 0, the linear forms of what
// eat_row keeps as logs (ShockDetails.t_obs = obs.time / sec, .Doppler = exp2(lg2_doppler), pybind/pymodel.cpp:296-298).
// out_* are [n_phi_eff][n_theta][n_t]; one lane per cell.
__global__ void __launch_bounds__(256)
vag_eat_details_kernel(const vag_model_params* __restrict__ params, const VagGridMeta* __restrict__ meta,
                       const double* __restrict__ geo_th, const double* __restrict__ geo_ph, const int* __restrict__ rep_of,
                       const double* __restrict__ cellpar, const double* __restrict__ cellgeo /* spreading jets, else null */,
                       double* __restrict__ out_t, double* __restrict__ out_dop) {
    const VagGridMeta M = meta[0];
    if (M.status != 0) return;
    const int K = M.n_t, nth = M.n_theta;
    const long long total = (long long)M.n_phi_eff * nth * K;
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= total) return;
    const int k = (int)(q % K), j = (int)((q / K) % nth), i = (int)(q / ((long long)K * nth));
    const vag_model_params P = params[0];
    const double one_plus_z = 1 + P.z, cos_obs = cos(P.theta_obs), sin_obs = sin(P.theta_obs);
    const int rep = rep_of[j] + i * M.rep_phi_stride;  // ((phi, theta) pair rows of a non-axisymmetric spreading jet)
    const double* par = cellpar + (long long)rep * K * VAG_NPAR;
    const double G = par[VP_GAMMA * K + k], u = par[VP_U * K + k], r = par[VP_R * K + k], teng = par[VP_TENG * K + k];
    double cos_v, time;
    if (cellgeo) {
        const double* geo = cellgeo + (long long)rep * K * 3;
        cos_v = geo[K + k] * geo_ph[i] * sin_obs + geo[k] * cos_obs;
        time = (teng + (1 - cos_v) * r / C_C) * one_plus_z;
    } else {
        cos_v = geo_th[M.th_stride + j] * geo_ph[i] * sin_obs + geo_th[j] * cos_obs;
        time = teng * one_plus_z + (1 - cos_v) / C_C * one_plus_z * r;
    }
    out_t[q] = time / U_SEC;
    out_dop[q] = 1.0 / (G - u * cos_v);
}

// COUNT = true is the instrumentation variant (exact work tallies); timed runs use COUNT = false.
// MODE selects the photon source (FLUX_SYN / FLUX_SYN_IC / FLUX_SSC).
// 128 VGPRs (four workgroups of 256, two of 512 per CU) is what the occupancy of every measured shape hangs on: the C2 launch asks
// for 80 KB of LDS, two workgroups per CU, and a 136-VGPR scratch-free build of this kernel (launch bound 1) leaves ONE resident:
// 39.3 ms instead of 24.0 ms per 512 models.  The few spilled values (36 B per lane) sit outside the inner loops.
// PIECES: some lattice of the batch is longer than the staged row (see K_all below); the loop over pieces costs the C2 shape
// 3 % when it is compiled in, so the one-piece form is its own instantiation.
template <bool COUNT, int MODE, bool SPREAD = false, int THREADS = FLUX_THREADS, bool PIECES = false>
__global__ void __launch_bounds__(THREADS, 4)
vag_flux_grid_kernel(FluxArgs a) {
    const int m = blockIdx.y;
    const VagGridMeta* Mp = a.meta + m;
    if (Mp->status != 0) return;
    const int n_phi_eff = Mp->n_phi_eff;
    const int n_pairs = Mp->n_theta * n_phi_eff;
    const int p0 = blockIdx.x * a.pairs_per_block;
    if (p0 >= n_pairs) return;
    const int p1 = min(n_pairs, p0 + a.pairs_per_block);
    const int tid = threadIdx.x;
    // A lattice longer than the staged row (KS nodes) is taken in pieces [k0, k0 + K) that overlap by one node: every requested
    // time falls into exactly one piece's [first node, last node), the window / bracket / boundary-spectra steps are the same
    // per piece, and the accumulators stay in LDS across pieces.  K_all <= KS (every default-resolution model) is one piece.
    const int K_all = Mp->n_t, KS = a.k_stride;
    const int n_pieces = (!PIECES || K_all <= KS) ? 1 : (K_all - 1 + KS - 2) / (KS - 1);
    int K = min(K_all, KS), k0 = 0;
    const int nt = a.nt, nnu = a.nnu;
    const int slots = nt * nnu;

    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* s_sp = lds;                      // [SP_TABLE_DOUBLES] softplus table first: keeps it 16-byte aligned
    double* s_par = s_sp + SP_LDS_DOUBLES;   // [KS][VAG_NPAR] photon/shock parameter block of every cell of the staged row
    double* s_t = s_par + VAG_NPAR * KS;     // [2][KS] log2 observer time of the row's lattice nodes (double buffered)
    double* s_dop = s_t + 2 * KS;            // [KS] log2 Doppler factor
    double* s_geom = s_dop + KS;             // [KS] log2(dOmega r^2 D^3)
    double* s_B = s_geom + KS;               // [nnu][KS] boundary log2-luminosities (frequency-major)
    double* s_tobs = s_B + (size_t)KS * nnu; // [nt]
    double* s_nu = s_tobs + nt;              // [nnu]
    double* s_w = s_nu + nnu;                // [nt] fractional position of each requested time inside its interval
    double* s_acc = s_w + nt;                // [nnu*nt] this workgroup's partial grid (each lane owns fixed slots)
    double* s_q = s_acc + slots;             // end of the common part
    // The 14 IC-correction constants of a cell (FLUX_SYN_IC / FLUX_FUSED) are read from L2 where an evaluation lies above the
    // cooling break, not staged: 14 x KS doubles less LDS is one more resident workgroup per CU on the C5 and C3 shapes
    // (27.5 -> 23.9 ms and 11.8 -> 9.8 ms per synchrotron pass).
    constexpr bool HAS_Q = MODE == FLUX_SYN_IC || MODE == FLUX_FUSED;
    double* s_hdr = s_q;                                              // FLUX_FUSED: [KS][6] SSC table headers
    double* s_B2 = s_hdr + (MODE == FLUX_FUSED ? 6 * KS : 0);         // FLUX_FUSED: [nnu][KS] SSC boundary values
    double* s_acc2 = s_B2 + (MODE == FLUX_FUSED ? (size_t)KS * nnu : 0);  // FLUX_FUSED: [nnu*nt] SSC partial grid
    int* s_kidx = (int*)(s_acc2 + (MODE == FLUX_FUSED ? slots : 0));  // [nt]
    constexpr int NW = THREADS / 64;
    int* s_win = s_kidx + nt;                // [2][2] observation-window counts of the row in either s_t buffer (WinCount)
    int breach = 0;

    const vag_model_params* Pp = a.params + m;
    const double one_plus_z = 1 + Pp->z;
    const double opz_over_c = one_plus_z / C_C;  // (1 - cos) / c * (1 + z) of calc_eat_non_spreading with the division done once
    {
        const double lg2_1pz = Mp->lg2_1pz;
        for (int i = tid; i < nt; i += THREADS) s_tobs[i] = a.lg2_t_obs[i];
        for (int l = tid; l < nnu; l += THREADS) s_nu[l] = a.lg2_nu_obs[l] + lg2_1pz;
        for (int i = tid; i < SP_LDS_DOUBLES; i += THREADS) s_sp[i] = a.sp_table[i];  // softplus table + log2 table
    }
    SpecConst sc;
    sc.init(Pp->p);
    const double cos_obs = Mp->cos_obs, sin_obs = Mp->sin_obs;  // (SPREAD kernels; the others read them with the row's record)
    const int* rep_of = a.g_rep_of + (size_t)m * Mp->th_stride;
    const int rep_stride = Mp->rep_phi_stride;  // (phi, theta) pair rows of a non-axisymmetric spreading jet: row = rep_of[j] + i * stride
    // non-spreading kernels: the model's row-geometry records (written by vag_grid_kernel), one base address for everything a row needs
    const double* rg = a.rowgeo + (size_t)m * a.rowgeo_stride;
    const int rg_th = SPREAD ? 0 : sload_i32(reinterpret_cast<const int*>(rg + 2));  // byte offset of the theta records
    auto rep_at = [&](int j, int i) {
        if constexpr (SPREAD)
            return sload_i32(rep_of + j) + i * rep_stride;
        else
            return sload_i32(reinterpret_cast<const int*>(reinterpret_cast<const char*>(rg) + rg_th + 32 * j + 24));
    };
    const LdsTab sp_tab = lds_tab(s_sp), lg_tab = lds_tab(s_sp + SP_TABLE_DOUBLES);
    // A lane owns the slots tid + r * THREADS (slot = l * nt + idx): always the same lane per slot, so the LDS accumulator needs no
    // atomics and the sum order is fixed.  The interpolation phase takes them U at a time; the first U -- all of them for
    // nt * nnu <= U * THREADS -- keep their (idx | l * KS << 16) in a register for the whole kernel (sign bit: no such slot, the fields then read slot 0's), later ones
    // are worked out again per row.  l * KS < 32768: the boundary block [nnu][KS] has to fit LDS.
    constexpr int U = MODE == FLUX_FUSED ? 2 : 4;
    const float inv_nt = __builtin_amdgcn_rcpf((float)nt);
    auto slot_desc = [&](int slot) -> int {
        const int l = (int)(((float)slot + 0.5f) * inv_nt);  // slot / nt (exact: slot < 2^20)
        return slot < slots ? ((slot - __mul24(l, nt)) | (__mul24(l, KS) << 16)) : (int)0x80000000;
    };
    int desc0[U];
#pragma unroll
    for (int u = 0; u < U; ++u) desc0[u] = slot_desc(tid + u * THREADS);

    for (int s = tid; s < slots; s += THREADS) s_acc[s] = 0;
    for (int i = tid; i < nt; i += THREADS) s_kidx[i] = 0;  // no bracket hint yet
    if (tid < 4) s_win[tid] = 0;
    if constexpr (MODE == FLUX_FUSED)
        for (int s = tid; s < slots; s += THREADS) s_acc2[s] = 0;
    unsigned long long n_evals = 0, n_interps = 0;  // block-uniform tallies (COUNT variant only)
#ifdef VAG_FLUX_STAMPS  // developer aid: cycles every wavefront of one workgroup spends per phase (profiles/flux_stamps.py)
    long long c_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, c_mark = __builtin_readcyclecounter();
#define VAG_FLUX_MARK(i) do { const long long now_ = __builtin_readcyclecounter(); c_ph[i] += now_ - c_mark; c_mark = now_; } while (0)
#else
#define VAG_FLUX_MARK(i) do { } while (0)
#endif

    // Software pipeline over the (theta, phi) rows of this workgroup, two barriers per row:
    //   interval 1:  bracket lookup + interval reciprocals + A1 (boundary spectra) of row p
    //   interval 2:  B (interpolate/accumulate) of row p  ||  A0 (EAT logs) of row p+1 into the other s_t buffer
    // A0 is latency bound (two log2 per node on < half of the lanes) and hides behind B's exp2 work.
    // Measured and rejected (-DVAG_FLUX_ROTATE): the EAT logs of the next row on the OTHER half of the workgroup than the bracket
    // lookup (which runs on the lanes tid < nt), so that the two latency-bound side jobs of a row land on different wavefronts:
    // 23.97 vs 23.22 ms per 512 C2 models -- the waves that take the logs are then the last at the second barrier of every row.
#ifdef VAG_FLUX_ROTATE
    const int etid = (tid + THREADS / 2) & (THREADS - 1);
#else
    const int etid = tid;
#endif
    const double* gth = a.geo_th + (size_t)m * 3 * Mp->th_stride;
    const double* gph = a.geo_ph + (size_t)m * 2 * Mp->ph_stride;
    auto stage_and_eat = [&](int j, int i, int buf) {  // (theta, phi) indices of the row: walked with a carry, never divided out
        const double w_lo = s_tobs[0], w_hi = s_tobs[nt - 1];
        if constexpr (SPREAD) {
            const double* geo = a.cellgeo + (a.cell_off[m] + (long long)rep_at(j, i) * K_all) * 3 + k0;
            eat_row_spread(s_par, KS, K, etid, THREADS, geo, gph[i], sin_obs, cos_obs, gph[Mp->ph_stride + i], one_plus_z,
                           s_t + buf * KS, s_dop, s_geom, lg_tab, K_all, s_win + buf * 2, w_lo, w_hi);
        } else {
            const RowGeo g = sload_rowgeo(rg, rg_th, j, i);
            const double cos_v = g.cos_view();
            const double t_coeff = (1 - cos_v) * opz_over_c;
            const double lg2_dOmega = g.dth + g.dph;
            eat_row(s_par, KS, K, etid, THREADS, cos_v, t_coeff, one_plus_z, lg2_dOmega, s_t + buf * KS, s_dop, s_geom, lg_tab,
                    s_win + buf * 2, w_lo, w_hi);
        }
    };
    int staged_rep = -1;
    auto stage_row = [&](int j, int i) {  // block-uniform: (re)load the photon block when the representative row changes
        const int rep = rep_at(j, i);
        if (rep != staged_rep) {
            const double* src = a.cellpar + (a.cell_off[m] + (long long)rep * K_all) * VAG_NPAR + k0;
#pragma unroll 1
            for (int q = tid; q < VAG_NPAR * K; q += THREADS) {  // rare path: keep its register footprint small
                const int par = (int)(((float)q + 0.5f) / (float)K);
                s_par[(q - par * K) * VAG_NPAR + par] = src[(size_t)par * K_all + (q - par * K)];
            }
            if constexpr (MODE == FLUX_SSC) {
                // the SSC pass never evaluates the synchrotron block: its first six rows carry the table headers
                // (n, first node, last node, theory_min, theory_max, pool offset) instead, saving a dependent global round trip per evaluation
                __syncthreads();
                const double* hdr0 = a.ichdr + (size_t)(a.cell_off[m] + (long long)rep * K_all + k0) * FLUX_IC_HDR;
#pragma unroll 1
                for (int q = tid; q < 6 * K; q += THREADS) {
                    const int kk = q / 6, w = q - kk * 6;
                    s_par[kk * VAG_NPAR + w] = hdr0[(size_t)kk * FLUX_IC_HDR + w];
                }
            }
            if constexpr (MODE == FLUX_FUSED) {
                const double* hdr0 = a.ichdr + (size_t)(a.cell_off[m] + (long long)rep * K_all + k0) * FLUX_IC_HDR;
#pragma unroll 1
                for (int q = tid; q < 6 * K; q += THREADS) {
                    const int kk = q / 6, w = q - kk * 6;
                    s_hdr[kk * 6 + w] = hdr0[(size_t)kk * FLUX_IC_HDR + w];
                }
            }
            staged_rep = rep;
            return true;
        }
        return false;
    };
    for (int piece = 0; piece < n_pieces; ++piece) {
    k0 = piece * (KS - 1);
    K = min(KS, K_all - k0);
    if (piece > 0) {
        staged_rep = -1;
        for (int i = tid; i < nt; i += THREADS) s_kidx[i] = 0;
    }
    __syncthreads();
    int jn = p0 / n_phi_eff, in_ = p0 - jn * n_phi_eff;  // (theta, phi) of the NEXT row while the loop runs
    stage_row(jn, in_);
    __syncthreads();
    stage_and_eat(jn, in_, 0);
    __syncthreads();
    VAG_FLUX_MARK(7);
    for (int pair = p0; pair < p1; ++pair) {
        const int buf = (pair - p0) & 1;
        if (++in_ == n_phi_eff) in_ = 0, ++jn;
        const double* s_tc = s_t + buf * KS;  // log2 observer times of the current row
        // ---- bracket lookup: idx -> k with t_row[k] <= lg2 t_obs < t_row[k+1] (iterate_to, observer.h:309-313,
        //      405-433), interval reciprocals, and the observation window (observed_window, observer.h:324-338)
        const double row_t0 = s_tc[0], row_tN = s_tc[K - 1];
        for (int idx = tid; idx < nt; idx += THREADS) {
            const double tq = s_tobs[idx];
            // a requested time outside the row's lattice contributes nothing (observer.h:405-433): it keeps interval 0 and a NaN
            // position, which makes the interpolated exponent non-finite like a non-finite slope does
            int kk = 0;
            double w = NAN;
#if defined(VAG_FLUX_ABLATE) && (VAG_FLUX_ABLATE & 4)
            if (tq >= row_t0 && tq < row_tN) kk = min(idx, K - 2), w = 0.5;
            if (false) {
#else
            if (tq >= row_t0 && tq < row_tN) {
#endif
                // invariant: s_tc[lo] <= tq < s_tc[hi].  Neighbouring rows shift the lattice only slightly: the previous row's
                // interval (still in s_kidx) and its two neighbours are requested together -- four reads in flight, one round
                // trip in the common case -- and only a larger shift grows the bracket outwards and bisects it.
                int lo = s_kidx[idx], hi;
                lo = min(max(lo, 1), K - 3);  // first row of the workgroup / a time the previous row did not cover: node 1
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
                        while (s_tc[lo] > tq) {  // ends at the latest at node 0: s_tc[0] = row_t0 <= tq
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
                w = (tq - t_lo) * (1.0 / (s_tc[lo + 1] - t_lo));  // position inside the interval, shared by all nu
            }
            s_kidx[idx] = kk;
            s_w[idx] = w;
        }
        VAG_FLUX_MARK(0);
        const double w_lo = s_tobs[0], w_hi = s_tobs[nt - 1];
        const bool in_window = !(row_tN < w_lo || row_t0 > w_hi);  // block-uniform
        if (in_window) {
            // the window's node counts were left by the row's EAT step, one pair per wavefront (WinCount)
            const int lane = tid & 63;
            int n_lt = 0, n_le = 0;
#if defined(VAG_FLUX_ABLATE) && (VAG_FLUX_ABLATE & 16)
            n_lt = 1, n_le = K;
#else
            n_lt = __builtin_amdgcn_readfirstlane(s_win[buf * 2]);
            n_le = __builtin_amdgcn_readfirstlane(s_win[buf * 2 + 1]);
#endif
            const int k_lo = n_lt > 0 ? n_lt - 1 : 0;
            const int k_hi = min(max(n_le, k_lo + 1), K - 1);
            // ---- A1: boundary values B[l][k] = log2 I'(nu_l (1+z) / D_k) + geom_k for k in the window.
            //      work item = (cell k, pair of frequencies); lanes run over k fastest, so a wavefront shares its
            //      frequencies and the +-20 softplus shortcuts / optically-thick cut / nu_M cut-off branch coherently.
            const int nk = k_hi - k_lo + 1;
            const int npair_nu = (nnu + 1) >> 1;
            const int total = nk * npair_nu;
            const float inv_nk = __builtin_amdgcn_rcpf((float)nk);  // tid / nk below is exact with it: (tid + 0.5) / nk is >= 1e-3 away from an integer
            if constexpr (COUNT) {  // instrumentation pass: exact unit counts for the roofline
                int n_lt0 = 0, n_ltN = 0;
                for (int base = 0; base < nt; base += 64) {
                    const int ii = base + lane;
                    const double v = ii < nt ? s_tobs[ii] : INFINITY;
                    n_lt0 += __popcll(__ballot(v < row_t0));
                    n_ltN += __popcll(__ballot(v < row_tN));
                }
                n_evals += (unsigned long long)nk * nnu;
                n_interps += (unsigned long long)(n_ltN - n_lt0) * nnu;
            }
            // item q = lg * nk + kk walks in steps of THREADS: (lg, kk) advance by (dq_l, dq_k) with one carry
            const int dq_l = THREADS / nk, dq_k = THREADS - dq_l * nk;  // block-uniform (scalar unit)
            int lg = (int)(((float)tid + 0.5f) * inv_nk);               // tid / nk (exact: tid < 2^20)
            int kk = tid - __mul24(lg, nk);
            int bofs = __mul24(lg, 2 * KS);  // s_B offset of frequency row l0 = 2 lg
            const int top = (nnu - 1) * KS;
            for (int q = tid; q < total; q += THREADS) {
                const int k = k_lo + kk;
                const int l0 = lg * 2, l1 = min(l0 + 1, nnu - 1);
                const double dop = s_dop[k], geom = s_geom[k];
                const double* cp = s_par + __mul24(k, VAG_NPAR);
                double b0, b1;
#if defined(VAG_FLUX_ABLATE) && (VAG_FLUX_ABLATE & 1)  // developer aid: phase timing by omission
                b0 = cp[0], b1 = cp[1];
                if (false)
#endif
                if constexpr (MODE == FLUX_SYN) {
                    const SpecRegs regs = load_spec_regs(lds_tab(s_par) + __mul24(k, VAG_NPAR / 2));
#ifdef VAG_FLUX_DUAL_EVAL
                    log2_I_nu_fast2(regs, sc, s_nu[l0] - dop, s_nu[l1] - dop, sp_tab, b0, b1);
#else
                    b0 = log2_I_nu_fast(regs, 1, sc, s_nu[l0] - dop, sp_tab);
                    b1 = log2_I_nu_fast(regs, 1, sc, s_nu[l1] - dop, sp_tab);
#endif
                } else if constexpr (HAS_Q) {
                    const double* cq = a.cellq + (a.cell_off[m] + (long long)staged_rep * K_all) * FLUX_NQ + k0 + k;
                    log2_I_nu_ic_pair(cp, 1, cq, K_all, sc, s_nu[l0] - dop, s_nu[l1] - dop, sp_tab, b0, b1);
                    if constexpr (MODE == FLUX_FUSED) {
                        const double* hp = s_hdr + __mul24(k, 6);
                        const double h0 = hp[0], h1 = hp[1], h2 = hp[2], h3 = hp[3], h4 = hp[4];
                        const double* tab = a.icpool + (unsigned long long)hp[5];
                        const double c0 = ic_table_eval_hdr(tab, h0, h1, h2, h3, h4, s_nu[l0] - dop, &breach);
                        const double c1 = ic_table_eval_hdr(tab, h0, h1, h2, h3, h4, s_nu[l1] - dop, &breach);
                        s_B2[bofs + k] = c0 + geom;
                        s_B2[min(bofs + KS, top) + k] = c1 + geom;
                    }
                } else {
                    const double h0 = cp[0], h1 = cp[1], h2 = cp[2], h3 = cp[3], h4 = cp[4];
                    const double* tab = a.icpool + (unsigned long long)cp[5];
                    b0 = ic_table_eval_hdr(tab, h0, h1, h2, h3, h4, s_nu[l0] - dop, &breach);
                    b1 = ic_table_eval_hdr(tab, h0, h1, h2, h3, h4, s_nu[l1] - dop, &breach);
                }
                s_B[bofs + k] = b0 + geom;
                s_B[min(bofs + KS, top) + k] = b1 + geom;
                kk += dq_k;
                lg += dq_l;
                bofs += dq_l * 2 * KS;
                if (kk >= nk) {
                    kk -= nk;
                    ++lg;
                    bofs += 2 * KS;
                }
            }
        }
        VAG_FLUX_MARK(1);
        __syncthreads();
        VAG_FLUX_MARK(2);
        // ---- interval 2: B of this row and A0 of the next one (other s_t buffer; s_dop / s_geom are free once A1 is done).  A
        //      row that needs a different photon block is staged after B instead (block-uniform rare path).
        //      B: interpolate in log2 t, exponentiate, accumulate (observer.h:405-433), U slots of the lane at a time as
        //      straight-line code: the U chains (bracket index -> boundary pair -> exp2 -> accumulator) are independent, and a
        //      wavefront that holds lattice nodes of the next row carries that node's two EAT logarithms as a further chain in the
        //      same block, so the LDS round trips and the Horner chains interleave instead of queueing (alone, one slot's chain
        //      takes ~650 cycles for 27 instructions and a node's logs ~1500; profiles/r03_flux_phase_budget.txt).  A slot
        //      without a finite slope adds exp2(-2000) = 0: the sums keep their order and their bits.
#ifdef VAG_FLUX_PRIO  // measured and rejected: the short latency-bound interval ahead of the other workgroup's boundary spectra
        __builtin_amdgcn_s_setprio(VAG_FLUX_PRIO);  // (s_setprio 2 here, 0 before the barrier: 21.83 vs 21.65 ms per 512 C2 models)
#endif
        if (tid == THREADS - 1) s_win[buf * 2] = 0, s_win[buf * 2 + 1] = 0;  // read by everyone before the barrier; the row after next adds again
        const bool have_next = pair + 1 < p1;
        const bool same_rep = have_next && rep_at(have_next ? jn : 0, have_next ? in_ : 0) == staged_rep;
        // EAT logs inside the interpolation block: the first node of every lane (k = tid); further nodes of long lattices and the
        // rows without an interpolation phase go through stage_and_eat
#if defined(VAG_FLUX_NO_FUSE_EAT) || (defined(VAG_FLUX_ABLATE) && (VAG_FLUX_ABLATE & 8))
        const bool fuse_eat = false;
#else
        const bool fuse_eat = !SPREAD && same_rep && in_window;
#endif
        if (__builtin_expect(same_rep && !fuse_eat, 0)) stage_and_eat(jn, in_, buf ^ 1);  // rare: keep its scalars out of the hot path's registers
        VAG_FLUX_MARK(3);
        if (in_window) {
            auto interp_group = [&](const int (&dq)[U], int slot0, auto with_eat, auto u_begin, auto u_end) {
                constexpr int UB = decltype(u_begin)::value, UE = decltype(u_end)::value;
                // -- requests of the EAT chain
                [[maybe_unused]] vdouble2 e_Gu, e_rt;
                [[maybe_unused]] double e_r2 = 0, e_cos = 0, e_tc = 0, e_dom = 0;
                [[maybe_unused]] const int ek = min(tid, K - 1);
                if constexpr (decltype(with_eat)::value) {
                    const RowGeo g = sload_rowgeo(rg, rg_th, jn, in_);
                    e_cos = g.cos_view();
                    e_tc = (1 - e_cos) * opz_over_c;
                    e_dom = g.dth + g.dph;
                    const double* c = s_par + ek * VAG_NPAR;
                    const LdsTab c2 = lds_tab(c);
                    e_Gu = c2[VP_GAMMA / 2], e_rt = c2[VP_R / 2];
                    e_r2 = c[VP_LG2_R2];
                }
                // -- the U slots (the descriptors are made opaque per row: the compiler would otherwise keep a dozen LDS addresses
                //    derived from them in registers for the whole kernel, and the boundary-spectra loop has none to spare)
                int dv[U];
#pragma unroll
                for (int u = UB; u < UE; ++u) {
                    dv[u] = dq[u];
#ifndef VAG_HOST_DEBUG
                    asm volatile("" : "+v"(dv[u]));
#endif
                }
                int kq[U], iq[U];
                double lo[U], hi[U], wq[U], aq[U];
                [[maybe_unused]] double lo2[U], hi2[U], aq2[U];
#pragma unroll
                for (int u = UB; u < UE; ++u) {
                    iq[u] = dv[u] & 0xffff;
                    kq[u] = s_kidx[iq[u]];
                }
#pragma unroll
                for (int u = UB; u < UE; ++u) {
                    const int kk = ((dv[u] >> 16) & 0x7fff) + kq[u];
                    lo[u] = s_B[kk], hi[u] = s_B[kk + 1];
                    wq[u] = s_w[iq[u]];
                    aq[u] = s_acc[min(slot0 + u * THREADS, slots - 1)];
                    if constexpr (MODE == FLUX_FUSED) {
                        lo2[u] = s_B2[kk], hi2[u] = s_B2[kk + 1];
                        aq2[u] = s_acc2[min(slot0 + u * THREADS, slots - 1)];
                    }
                }
                [[maybe_unused]] bool sp_a = false, sp_b = false;
                [[maybe_unused]] double e_dop = 0, e_lt = 0;
                if constexpr (decltype(with_eat)::value) {
                    e_dop = -log2_tab_core(fma(-e_Gu.y, e_cos, e_Gu.x), lg_tab, sp_a);  // eat_row's expressions, rounding for rounding
                    e_lt = log2_tab_core(fma(e_tc, e_rt.x, e_rt.y * one_plus_z), lg_tab, sp_b);
                }
#pragma unroll
                for (int u = UB; u < UE; ++u) {
                    // the slope (hi - lo) / (t[k+1] - t[k]) is finite <=> hi - lo is (observer.h:422-426); with the position w in
                    // [0, 1) -- or NaN for a time outside the row -- the exponent is finite exactly for the terms that count
                    const double x = fma(hi[u] - lo[u], wq[u], lo[u]);
                    aq[u] += exp2_fast(isfinite(x) ? x : -2000.0);
                    if constexpr (MODE == FLUX_FUSED) {
                        const double x2 = fma(hi2[u] - lo2[u], wq[u], lo2[u]);
                        aq2[u] += exp2_fast(isfinite(x2) ? x2 : -2000.0);
                    }
                }
#pragma unroll
                for (int u = UB; u < UE; ++u)
                    if (dv[u] >= 0) {
                        s_acc[slot0 + u * THREADS] = aq[u];
                        if constexpr (MODE == FLUX_FUSED) s_acc2[slot0 + u * THREADS] = aq2[u];
                    }
                if constexpr (decltype(with_eat)::value) {
                    if (sp_a) e_dop = -log2(fma(-e_Gu.y, e_cos, e_Gu.x));  // never in practice: arguments the table does not serve
                    if (__builtin_expect(sp_b, 0)) e_lt = log2(fma(e_tc, e_rt.x, e_rt.y * one_plus_z));
                    WinCount wc;
                    wc.add(tid < K, e_lt, w_lo, w_hi);
                    wc.store(s_win + (buf ^ 1) * 2, tid);
                    if (tid < K) {
                        s_dop[ek] = e_dop;
                        s_t[(buf ^ 1) * KS + ek] = e_lt;
                        s_geom[ek] = (e_dom + e_r2) + 3.0 * e_dop;
                    }
                }
            };
#if defined(VAG_FLUX_ABLATE) && (VAG_FLUX_ABLATE & 2)
            if (false)
#endif
            {
                using I0 = std::integral_constant<int, 0>;
                using IH = std::integral_constant<int, U / 2>;
                using IU = std::integral_constant<int, U>;
                if (fuse_eat && (tid & ~63) < K) {  // the node's chain next to half of the slots (registers), then the other half
                    interp_group(desc0, tid, std::true_type{}, I0{}, IH{});
                    interp_group(desc0, tid, std::false_type{}, IH{}, IU{});
                } else {
                    interp_group(desc0, tid, std::false_type{}, I0{}, IU{});
                }
                for (int slot0 = tid + U * THREADS; slot0 - tid < slots; slot0 += U * THREADS) {  // block-uniform trip count
                    int dq[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) dq[u] = slot_desc(slot0 + u * THREADS);
                    interp_group(dq, slot0, std::false_type{}, I0{}, IU{});
                }
            }
            if (fuse_eat) {
                if (THREADS < 512 && K > THREADS) {  // nodes beyond the first per lane (256-lane workgroups on long lattices); their counts join the first's
                    const RowGeo g = sload_rowgeo(rg, rg_th, jn, in_);
                    const double cos_v = g.cos_view();
                    const double t_coeff = (1 - cos_v) * opz_over_c;
                    const double lg2_dOmega = g.dth + g.dph;
                    WinCount wc;
                    for (int k = tid + THREADS; k < K; k += THREADS) {
                        const double* c = s_par + k * VAG_NPAR;
                        const LdsTab c2 = lds_tab(c);
                        const vdouble2 Gu = c2[VP_GAMMA / 2], rt = c2[VP_R / 2];
                        const double lg2_dop = -log2_tab(fma(-Gu.y, cos_v, Gu.x), lg_tab);
                        const double lt = log2_tab(fma(t_coeff, rt.x, rt.y * one_plus_z), lg_tab);
                        s_dop[k] = lg2_dop;
                        s_t[(buf ^ 1) * KS + k] = lt;
                        s_geom[k] = (lg2_dOmega + c[VP_LG2_R2]) + 3.0 * lg2_dop;
                        wc.add(true, lt, w_lo, w_hi);
                    }
                    wc.store(s_win + (buf ^ 1) * 2, tid);
                }
            }
        }
        VAG_FLUX_MARK(4);
#ifdef VAG_FLUX_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        __syncthreads();
        VAG_FLUX_MARK(5);
        if (__builtin_expect(have_next && !same_rep, 0)) {  // once per theta row
            stage_row(jn, in_);
            __syncthreads();
            stage_and_eat(jn, in_, buf ^ 1);
            __syncthreads();
            VAG_FLUX_MARK(6);
        }
    }
    }  // pieces of the lattice
#ifdef VAG_FLUX_STAMPS
    if (blockIdx.y == 0 && blockIdx.x == 1 && (tid & 63) == 0)
        printf("flux wave %d rows %d cycles: bracket+window %lld  A1 %lld  barrier1 %lld  A0next %lld  B %lld  barrier2 %lld  restage %lld  prologue %lld\n",
               tid >> 6, p1 - p0, c_ph[0], c_ph[1], c_ph[2], c_ph[3], c_ph[4], c_ph[5], c_ph[6], c_ph[7]);
#endif
    if constexpr (COUNT) {
        if (tid == 0) {
            atomicAdd(a.work_count, n_evals);
            atomicAdd(a.work_count + 1, n_interps);
        }
    }
    if constexpr (MODE == FLUX_SSC || MODE == FLUX_FUSED) {
        if (breach) atomicOr(a.ic_status + m, ic_breach_status(breach));
    }
    __syncthreads();
    // partial grid of this workgroup, stored [l][idx] like the reference's F_nu (nu outer)
    double* my_partial = a.partial + ((size_t)m * a.max_blocks + blockIdx.x) * slots;
    for (int s = tid; s < slots; s += THREADS) my_partial[s] = s_acc[s];
    if constexpr (MODE == FLUX_FUSED) {
        double* my_partial2 = a.partial2 + ((size_t)m * a.max_blocks + blockIdx.x) * slots;
        for (int s = tid; s < slots; s += THREADS) my_partial2[s] = s_acc2[s];
    }
}

// Deterministic sum over a model's workgroup partials + normalisation
// F *= (1+z)/d_L^2 (observer.h:442), / unit::flux_den_cgs (pymodel.cpp:506-508); band mode applies the
// Boole weights of Observer::flux (observer.h:555-567) and / unit::flux_cgs (pymodel.cpp:403-405).
__global__ void __launch_bounds__(256)
vag_reduce_kernel(const vag_model_params* __restrict__ params, const VagGridMeta* __restrict__ meta,
                  const double* __restrict__ partial, int max_blocks, int pairs_per_block, int nt, int nnu,
                  const double* __restrict__ band_w /* NULL or [nnu] */, double* __restrict__ out,
                  int* __restrict__ work_counter = nullptr /* of a persistent launch before this one: back to zero */) {
    const int m = blockIdx.y;
    if (work_counter && blockIdx.x == 0 && m == 0 && threadIdx.x == 0) *work_counter = 0;
    const VagGridMeta M = meta[m];
    const vag_model_params P = params[m];
    const int slots = nt * nnu;
    const int nblk = (M.status == 0) ? (M.n_theta * M.n_phi_eff + pairs_per_block - 1) / pairs_per_block : 0;
    const double d_L = P.lumi_dist * U_CM;
    const double norm = (1 + P.z) / (d_L * d_L);
    const double* src = partial + (size_t)m * max_blocks * slots;
    if (!band_w) {
        for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < slots; s += gridDim.x * blockDim.x) {
            double v = 0;
            for (int b = 0; b < nblk; ++b) v += src[(size_t)b * slots + s];
            out[(size_t)m * slots + s] = (M.status == 0) ? (v * norm) / U_FLUX_DEN_CGS : NAN;
        }
    } else {
        for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < nt; idx += gridDim.x * blockDim.x) {
            double f = 0;
            for (int l = 0; l < nnu; ++l) {
                double v = 0;
                for (int b = 0; b < nblk; ++b) v += src[(size_t)b * slots + (size_t)l * nt + idx];
                f += (v * norm) * band_w[l];
            }
            out[(size_t)m * nt + idx] = (M.status == 0) ? f / U_FLUX_CGS : NAN;
        }
    }
}

// The same sum for plain (nu, t) grids with the partials spread over REDUCE_GROUPS lanes per output slot: a single model's request is
// cut into up to a few thousand workgroups (4 rows each, so that one model fills the chip), and one lane adding a thousand partials
// one after the other took 0.26 ms of a 0.75 ms configs[1] call and 1.0 ms of a 3.4 ms configs[4] call.  Lane group g of a slot adds
// the partials b = g, g + G, g + 2 G, ... in that order, the groups are then added in the order g = 0 .. G - 1: a fixed tree, a
// function of the model's own workgroup count only (a model's fluxes do not depend on the batch around it).
constexpr int REDUCE_GROUPS = 8, REDUCE_SLOTS = 64;
__global__ void __launch_bounds__(REDUCE_GROUPS * REDUCE_SLOTS)
vag_reduce_grid_kernel(const vag_model_params* __restrict__ params, const VagGridMeta* __restrict__ meta,
                       const double* __restrict__ partial, int max_blocks, int pairs_per_block, int slots, double* __restrict__ out,
                       int* __restrict__ work_counter = nullptr /* of a persistent launch before this one: back to zero */) {
    const int m = blockIdx.y;
    if (work_counter && blockIdx.x == 0 && m == 0 && threadIdx.x == 0) *work_counter = 0;
    const VagGridMeta M = meta[m];
    const int nblk = (M.status == 0) ? (M.n_theta * M.n_phi_eff + pairs_per_block - 1) / pairs_per_block : 0;
    const int sx = threadIdx.x % REDUCE_SLOTS, g = threadIdx.x / REDUCE_SLOTS;
    const int s = blockIdx.x * REDUCE_SLOTS + sx;
    __shared__ double s_part[REDUCE_GROUPS][REDUCE_SLOTS];
    const double* src = partial + (size_t)m * max_blocks * slots;
    double v = 0;
    if (s < slots)
        for (int b = g; b < nblk; b += REDUCE_GROUPS) v += src[(size_t)b * slots + s];
    s_part[g][sx] = v;
    __syncthreads();
    if (g == 0 && s < slots) {
        double sum = s_part[0][sx];
#pragma unroll
        for (int q = 1; q < REDUCE_GROUPS; ++q) sum += s_part[q][sx];
        const double d_L = params[m].lumi_dist * U_CM;
        const double norm = (1 + params[m].z) / (d_L * d_L);
        out[(size_t)m * slots + s] = (M.status == 0) ? (sum * norm) / U_FLUX_DEN_CGS : NAN;
    }
}

// ------------------------------------------------------------------------------------------------
// Series form (Observer::specific_flux_series, observer.h:447-538): n paired (t_s, nu_s) points.
// One workgroup (64 lanes) per (model, pair range); lanes over data points, two boundary evaluations per
// (row, point).  Sharing of boundary values between equal-frequency runs in the reference changes cost,
// not values.  Output partial[m][block][n].
// ------------------------------------------------------------------------------------------------
constexpr int SERIES_THREADS = 64;    // lanes that share one (theta, phi) row: ONE wavefront, so rows need no block barrier
constexpr int SERIES_WAVES = 4;       // most independent wavefronts per workgroup (the host picks 4, 2 or 1 by what stays resident); they only share the tables
constexpr int SERIES_MAX_SLOTS = 8;   // data points per lane: n <= 512
constexpr int SERIES_CHUNK = 8;       // (theta, phi) rows per partial sum: the unit of a model's summation tree, whatever the batch
constexpr int SERIES_MAX_BANDS = 8;   // distinct frequencies the shared-node path handles
// doubles of LDS one series wavefront owns (kept even: the cell blocks are read with 16-byte loads)
__host__ __device__ inline int series_region_doubles(int ks, bool ic, int n_bands) {
    (void)ic;  // the IC-correction constants are read from L2, not staged
    return ((VAG_NPAR + 3 + n_bands) * ks + 1) & ~1;
}

// LDS produced and consumed by the same wavefront: program order suffices in hardware, the fence keeps the compiler from
// moving accesses across it
VAG_DEV void wave_sync() {
#ifndef VAG_HOST_DEBUG
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
#endif
}

struct SeriesArgs {
    const vag_model_params* params;
    const VagGridMeta* meta;
    const double* geo_th;
    const double* geo_ph;
    const int* g_rep_of;
    Layout lay;
    const double* cellpar;
    const double* lg2_t_obs;  // [n]
    const double* lg2_nu_obs; // [n]
    int n;
    int pairs_per_block /* multiple of chunk */, max_blocks, max_chunks, k_stride;
    int chunk;  // rows per partial sum: SERIES_CHUNK for point series (batch-independent summation tree), pairs_per_block for grids
    double* partial; // [nb][max_chunks][n]
    const double* sp_table;
    const double* cellq;  // FLUX_SYN_IC: [cells][FLUX_NQ]
    const double* ichdr;  // FLUX_SSC: [cells][FLUX_IC_HDR] table headers
    const double* icpool; // FLUX_SSC: the tables, back to back
    int* ic_status;       // FLUX_SSC: per-model breach flag
    const double* cellgeo; // SPREAD: [rows][3][n_t] per-cell polar geometry
    // few distinct frequencies (a fit's bands), n <= 64: point s observes band band_idx[s], whose log2 nu is
    // lg2_nu_obs[band_first[b]].  n_bands = 0 turns the shared-node path off.
    int n_bands;
    const int* band_idx;
    const int* band_first;
    // A small (nu, t) GRID served by this kernel (grid_nt > 0): point s = l * grid_nt + idx observes frequency lg2_nu_obs[l] at
    // time lg2_t_obs[idx]; n = n_bands * grid_nt; band_idx / band_first are not read.
    int grid_nt;
    // persistent launches (vag_flux_fit_rows_kernel): the batch size and the work-item counter in device memory, zero between launches
    // (the reduction kernel behind the launch resets it); the items' model offsets are lay.row_off[nb + 1 ...] (the grid kernel's plan scan)
    int nb;
    int* work;
    // vag_ctx_count_work (the tallying instantiation of vag_flux_fit_rows_kernel only): [0] boundary-spectrum evaluations, [1] interpolations
    unsigned long long* tally;
};

// The kernel's arguments read again from the kernel-argument segment (constant address space: scalar loads), the pointer hidden from
// the optimiser first: a loop that calls this per turn re-reads the arguments per turn instead of keeping all of them in scalar
// registers across the loop (where they overflow into VGPR lanes, and those into scratch).
#ifndef VAG_HOST_DEBUG
typedef const SeriesArgs __attribute__((address_space(4))) KernargSeriesArgs;
VAG_DEV SeriesArgs load_series_args() {
    KernargSeriesArgs* p = (KernargSeriesArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    SeriesArgs a;
    __builtin_memcpy(&a, p, sizeof(SeriesArgs));
    return a;
}
#endif


// Series kernels: k with s_t[k] < t <= s_t[k+1] (t == s_t[0] -> 0), grown outwards from `hint` (the interval of the same
// point in the previous row) and then bisected.  Requires s_t[0] <= t <= s_t[K-1].
VAG_DEV int series_bracket(const double* __restrict__ s_t, int K, double t, int hint) {
    int lo = hint < 0 ? 0 : (hint > K - 2 ? K - 2 : hint), hi;
    if (s_t[lo] < t || lo == 0) {
        int step = 1;
        hi = lo + 1;
        while (hi < K - 1 && s_t[hi] < t) {
            lo = hi;
            step <<= 1;
            hi = min(lo + step, K - 1);
        }
    } else {
        int step = 1;
        hi = lo;
        lo = hi - 1;
        while (lo > 0 && !(s_t[lo] < t)) {
            hi = lo;
            step <<= 1;
            lo = max(hi - step, 0);
        }
    }
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (s_t[mid] < t)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

// NSLOT = data points per lane (n <= 64 * NSLOT): short series (a walker's 60 points) keep one point per lane in
// registers instead of eight, which is the difference between 3 and 5 resident wavefronts per SIMD.
// GRID: the kernel serves a small (nu, t) grid (SeriesArgs::grid_nt); a compile-time switch so that the fit instantiations do not
// carry its registers.
template <int MODE, bool SPREAD = false, int NSLOT = SERIES_MAX_SLOTS, bool GRID = false>
__global__ void __launch_bounds__(SERIES_THREADS * SERIES_WAVES)  // (a 128-VGPR cap spills and measured 12 % slower on the C4 fit)
vag_flux_series_kernel(SeriesArgs a) {
    const int m = blockIdx.y;
    const int wave = threadIdx.x >> 6, tid = threadIdx.x & 63;  // `tid`: lane inside this row's wavefront
    const int KS = a.k_stride;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* s_sp = lds;
    {  // a workgroup whose rows all lie beyond this model's grid (the launch is sized for the largest grid of the batch) leaves
       // before it stages anything
        const VagGridMeta* Mp = a.meta + m;
        if (Mp->status != 0 || (long long)blockIdx.x * (blockDim.x >> 6) * a.pairs_per_block >= (long long)Mp->n_theta * Mp->n_phi_eff)
            return;
    }
    for (int i = threadIdx.x; i < SP_LDS_DOUBLES; i += blockDim.x) s_sp[i] = a.sp_table[i];
    double* s_band = s_sp + SP_LDS_DOUBLES;  // [SERIES_MAX_BANDS] log2 nu of the fit's bands (shared-node path)
    if (threadIdx.x < a.n_bands) s_band[threadIdx.x] = a.lg2_nu_obs[GRID ? (int)threadIdx.x : a.band_first[threadIdx.x]];
    __syncthreads();  // the only workgroup-wide barrier: from here on every wavefront works alone on its own rows
    const int vb = blockIdx.x * (blockDim.x >> 6) + wave;  // virtual block = wavefront (the host launches 4, 2 or 1 per workgroup)
    if (vb >= a.max_blocks) return;
    const VagGridMeta M = a.meta[m];
    // partial sums are written per CHUNK of SERIES_CHUNK (theta, phi) rows, not per wavefront: the host picks the rows per
    // wavefront from the batch size, and a model's summation tree must not depend on what else is in the batch
    double* chunk_partial = a.partial + (size_t)m * a.max_chunks * a.n;
    if (M.status != 0) return;
    const int n_pairs = M.n_theta * M.n_phi_eff;
    const int p0 = vb * a.pairs_per_block;
    if (p0 >= n_pairs) return;
    const int p1 = min(n_pairs, p0 + a.pairs_per_block);
    // a lattice longer than the staged row is taken in overlapping pieces [k0, k0 + K), as in vag_flux_grid_kernel
    const int K_all = M.n_t;
    const int n_pieces = K_all <= KS ? 1 : (K_all - 1 + KS - 2) / (KS - 1);
    int K = min(K_all, KS), k0 = 0;
    double* s_par = s_band + SERIES_MAX_BANDS + (size_t)wave * series_region_doubles(KS, MODE == FLUX_SYN_IC, a.n_bands);
    double* s_t = s_par + VAG_NPAR * KS;
    double* s_dop = s_t + KS;
    double* s_geom = s_dop + KS;
    double* s_Bw = s_geom + KS;  // [n_bands][KS] boundary values of the shared-node path
    const LdsTab sp_tab = lds_tab(s_sp), lg_tab = lds_tab(s_sp + SP_TABLE_DOUBLES);
    int breach = 0;

    const double one_plus_z = 1 + a.params[m].z;
    const double lg2_1pz = M.lg2_1pz;  // observer constants come from the grid kernel: no library calls per wavefront
    SpecConst sc;
    sc.init_fast(a.params[m].p, lg_tab);
    const double cos_obs = M.cos_obs, sin_obs = M.sin_obs;
    const double* gth = a.geo_th + (size_t)m * 3 * M.th_stride;
    const double* gph = a.geo_ph + (size_t)m * 2 * M.ph_stride;
    const int* rep_of = a.g_rep_of + (size_t)m * M.th_stride;
    const int n_phi_eff = M.n_phi_eff;
    // Row geometry of the next 64 (theta, phi) rows, one row per lane, read back with v_readlane when the row's turn comes: the
    // five dependent global loads and the index division of a row leave its critical path (a wavefront works alone: nothing
    // else would hide them).  g_a / g_b / g_c = (cos of the viewing angle, time coefficient, log2 dOmega); a spreading jet
    // keeps (cos phi, -, log2 dphi) because its polar angle is per cell.
    double g_a = 0, g_b = 0, g_c = 0;
    int g_rep = 0;
    auto load_row_geometry = [&](int base) {
        const int pr = min(base + tid, p1 - 1);
        const int j = pr / n_phi_eff, i = pr - j * n_phi_eff;
        g_rep = rep_of[j] + i * M.rep_phi_stride;  // ((phi, theta) pair rows of a non-axisymmetric spreading jet)
        if constexpr (SPREAD) {
            g_a = gph[i];
            g_c = gph[M.ph_stride + i];
        } else {
            g_a = gth[M.th_stride + j] * gph[i] * sin_obs + gth[j] * cos_obs;
            g_b = (1 - g_a) / C_C * one_plus_z;
            g_c = gth[2 * M.th_stride + j] + gph[M.ph_stride + i];
        }
    };
    auto lane_value = [&](double v, int l) {
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
    };

    double acc[NSLOT], tq[NSLOT], nuq[NSLOT];
    int kprev[NSLOT];  // each point's interval in the previous row: the next row's search starts there
    int bandq[GRID ? NSLOT : 1];  // grid mode: the frequency index of each of this lane's points
#pragma unroll
    for (int q = 0; q < NSLOT; ++q) {
        acc[q] = 0;
        kprev[q] = -1;
        const int s = tid + q * SERIES_THREADS;
        if constexpr (GRID) {
            const int l = s < a.n ? s / a.grid_nt : 0;
            bandq[q] = l;
            tq[q] = s < a.n ? a.lg2_t_obs[s - l * a.grid_nt] : 0;
            nuq[q] = s < a.n ? a.lg2_nu_obs[l] + lg2_1pz : 0;
        } else {
            tq[q] = s < a.n ? a.lg2_t_obs[s] : 0;
            nuq[q] = s < a.n ? a.lg2_nu_obs[s] + lg2_1pz : 0;
        }
    }
    const int my_band = (NSLOT == 1 && a.n_bands > 0 && tid < a.n) ? a.band_idx[tid] : 0;
    int staged_rep = -1;
#ifdef VAG_SERIES_STAMPS  // developer aid: cycles of wavefront 0 of model 0 per phase
    long long c_stage = 0, c_eat = 0, c_pts = 0, c_brk = 0, c_items = 0, c_mark = __builtin_readcyclecounter();
#define VAG_SER_MARK(acc) do { const long long now_ = __builtin_readcyclecounter(); acc += now_ - c_mark; c_mark = now_; } while (0)
#else
#define VAG_SER_MARK(acc) do { } while (0)
#endif
    for (int piece = 0; piece < n_pieces; ++piece) {
    k0 = piece * (KS - 1);
    K = min(KS, K_all - k0);
    if (piece > 0) {
        staged_rep = -1;
#pragma unroll
        for (int q = 0; q < NSLOT; ++q) acc[q] = 0, kprev[q] = -1;
    }
    for (int pair = p0; pair < p1; ++pair) {
        const int gl = (pair - p0) & 63;
        if (gl == 0) load_row_geometry(pair);
        const int rep = __builtin_amdgcn_readlane(g_rep, gl);
        if (pair > p0 && (pair - p0) % a.chunk == 0) {  // p0 is a multiple of the chunk: close the chunk before this row
            double* dst = chunk_partial + (size_t)(pair / a.chunk - 1) * a.n;
#pragma unroll
            for (int q = 0; q < NSLOT; ++q) {
                const int s = tid + q * SERIES_THREADS;
                if (s < a.n) dst[s] = piece == 0 ? acc[q] : dst[s] + acc[q];
                acc[q] = 0;
            }
        }
        wave_sync();
        VAG_SER_MARK(c_pts);
        if (rep != staged_rep) {
            const double* src = a.cellpar + (a.lay.cell_off[m] + (long long)rep * K_all) * VAG_NPAR + k0;
            // four loads in flight per lane before the first LDS store: a wavefront stages alone, so the HBM / L2 latency of
            // its ~11 dependent round trips is otherwise fully exposed
            const float inv_K = 1.0f / (float)K;
            for (int q0 = tid; q0 < VAG_NPAR * K; q0 += 4 * SERIES_THREADS) {
                double v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int q = q0 + u * SERIES_THREADS;
                    const int par = (int)(((float)q + 0.5f) * inv_K);
                    v[u] = q < VAG_NPAR * K ? src[(size_t)par * K_all + (q - par * K)] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int q = q0 + u * SERIES_THREADS;
                    if (q < VAG_NPAR * K) {
                        const int par = (int)(((float)q + 0.5f) * inv_K), k = q - par * K;  // q / K, exact for q < 2^20
                        s_par[k * VAG_NPAR + par] = v[u];
                    }
                }
            }
            staged_rep = rep;
            wave_sync();
        }
        VAG_SER_MARK(c_stage);
        // IC-correction constants of the row's cells, read from L2 where an evaluation lies above the cooling break (FLUX_SYN_IC)
        const double* cq_row = MODE == FLUX_SYN_IC ? a.cellq + (a.lay.cell_off[m] + (long long)rep * K_all) * FLUX_NQ + k0 : nullptr;
        if constexpr (SPREAD) {
            const double* geo = a.cellgeo + (a.lay.cell_off[m] + (long long)rep * K_all) * 3 + k0;
            eat_row_spread(s_par, KS, K, tid, SERIES_THREADS, geo, lane_value(g_a, gl), sin_obs, cos_obs, lane_value(g_c, gl),
                           one_plus_z, s_t, s_dop, s_geom, lg_tab, K_all);
        } else {
            const double cos_v = lane_value(g_a, gl), t_coeff = lane_value(g_b, gl), lg2_dOmega = lane_value(g_c, gl);
            eat_row(s_par, KS, K, tid, SERIES_THREADS, cos_v, t_coeff, one_plus_z, lg2_dOmega, s_t, s_dop, s_geom, lg_tab);
        }
        wave_sync();
        VAG_SER_MARK(c_eat);
        const double row_t0 = s_t[0], row_tN = s_t[K - 1];
        if constexpr (GRID) {
            // A (nu, t) grid on the shared-node path: every frequency sees the same times, so the boundary spectra are evaluated
            // once per (frequency, lattice node of the row's observation window) -- Observer::specific_flux's own scheme
            // (observer.h:355-445) -- and each of the lane's points interpolates between two of them.  One wavefront per row,
            // accumulators in registers, no workgroup barrier: rows of a few hundred (nu, t) slots do not fill a workgroup.
            int kq[NSLOT];
            int kmin = 1 << 30, kmax = -1;
#pragma unroll
            for (int q = 0; q < NSLOT; ++q) {
                const int s = tid + q * SERIES_THREADS;
                kq[q] = -1;
                if (s < a.n && tq[q] >= row_t0 && tq[q] <= row_tN) {
                    kq[q] = kprev[q] = series_bracket(s_t, K, tq[q], kprev[q]);
                    kmin = min(kmin, kq[q]);
                    kmax = max(kmax, kq[q]);
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                kmin = min(kmin, __shfl_xor(kmin, off, 64));
                kmax = max(kmax, __shfl_xor(kmax, off, 64));
            }
            if (kmax >= 0) {  // wave-uniform
                const int nn = kmax + 2 - kmin;
                const int total = nn * a.n_bands;
                const float inv_nn = 1.0f / (float)nn;
                for (int idx = tid; idx < total; idx += SERIES_THREADS) {
                    const int b = (int)(((float)idx + 0.5f) * inv_nn);
                    const int kk = kmin + idx - b * nn;
                    const double x = (s_band[b] + lg2_1pz) - s_dop[kk];
                    double v;
                    if (MODE == FLUX_SYN) {
                        v = log2_I_nu_fast(load_spec_regs(lds_tab(s_par) + __mul24(kk, VAG_NPAR / 2)), 1, sc, x, sp_tab);  // seven 16-byte reads
                    } else if (MODE == FLUX_SYN_IC) {
                        v = log2_I_nu_ic(s_par + kk * VAG_NPAR, 1, cq_row + kk, K_all, sc, x, sp_tab);
                    } else {
                        const double* hdr = a.ichdr + (a.lay.cell_off[m] + (long long)rep * K_all + k0 + kk) * FLUX_IC_HDR;
                        v = ic_table_eval(hdr, a.icpool, x, &breach);
                    }
                    s_Bw[b * KS + kk] = v + s_geom[kk];
                }
                wave_sync();
#pragma unroll
                for (int q = 0; q < NSLOT; ++q) {
                    if (kq[q] >= 0) {
                        const int k = kq[q];
                        const double* Bb = s_Bw + bandq[q] * KS;
                        const double blo = Bb[k], bhi = Bb[k + 1];
                        const double sl = (bhi - blo) * (1.0 / (s_t[k + 1] - s_t[k]));
                        if (isfinite(sl)) acc[q] += exp2_fast(blo + (tq[q] - s_t[k]) * sl);
                    }
                }
            }
            continue;
        }
        if constexpr (NSLOT == 1) {
            if (a.n_bands > 0) {
                // Shared-node path (the reference's specific_flux_series shares boundary evaluations between points of one
                // band, observer.h:447-538): the sorted points of a fit fall into a short run of lattice intervals, so the
                // spectrum is evaluated once per (band, node of that run) and every point interpolates between two of them.
                // Same evaluator, same arguments, same interpolation arithmetic as the per-point path: identical results.
                const double t = tq[0];
                const bool in = tid < a.n && t >= row_t0 && t <= row_tN;
                int k = 0;
                if (in) k = kprev[0] = series_bracket(s_t, K, t, kprev[0]);
                const unsigned long long mask = __ballot(in);
                VAG_SER_MARK(c_brk);
                if (mask != 0) {  // wave-uniform
                    const int first = __ffsll((long long)mask) - 1, last = 63 - __clzll((long long)mask);
                    const int kmin = __builtin_amdgcn_readlane(k, first);  // t ascends with the point index, and so does k
                    const int nn = __builtin_amdgcn_readlane(k, last) + 2 - kmin;
                    const int total = nn * a.n_bands;
                    const float inv_nn = 1.0f / (float)nn;
                    for (int idx = tid; idx < total; idx += SERIES_THREADS) {
                        const int b = (int)(((float)idx + 0.5f) * inv_nn);
                        const int kk = kmin + idx - b * nn;
                        const double x = (s_band[b] + lg2_1pz) - s_dop[kk];
                        double v;
                        if (MODE == FLUX_SYN) {
                            v = log2_I_nu_fast(load_spec_regs(lds_tab(s_par) + __mul24(kk, VAG_NPAR / 2)), 1, sc, x, sp_tab);  // seven 16-byte reads
                        } else if (MODE == FLUX_SYN_IC) {
                            v = log2_I_nu_ic(s_par + kk * VAG_NPAR, 1, cq_row + kk, K_all, sc, x, sp_tab);
                        } else {
                            const double* hdr = a.ichdr + (a.lay.cell_off[m] + (long long)rep * K_all + k0 + kk) * FLUX_IC_HDR;
                            v = ic_table_eval(hdr, a.icpool, x, &breach);
                        }
                        s_Bw[b * KS + kk] = v + s_geom[kk];
                    }
                    wave_sync();
                    VAG_SER_MARK(c_items);
                    if (in) {
                        const double* Bb = s_Bw + my_band * KS;
                        const double blo = Bb[k], bhi = Bb[k + 1];
                        const double sl = (bhi - blo) * (1.0 / (s_t[k + 1] - s_t[k]));
                        if (isfinite(sl)) acc[0] += exp2_fast(blo + (t - s_t[k]) * sl);
                    }
                }
                continue;
            }
        }
#pragma unroll
        for (int q = 0; q < NSLOT; ++q) {
            const int s = tid + q * SERIES_THREADS;
            if (s < a.n) {
                const double t = tq[q];
                // iterate_through (observer.h:316-320): a point equal to a node belongs to the interval it closes
                if (t >= row_t0 && t <= row_tN) {
                    const int k = kprev[q] = series_bracket(s_t, K, t, kprev[q]);  // s_t[k] < t <= s_t[k+1]; t == row_t0 -> 0
                    double blo, bhi;
                    if (MODE == FLUX_SYN) {
                        blo = log2_I_nu_fast(load_spec_regs(lds_tab(s_par) + __mul24(k, VAG_NPAR / 2)), 1, sc, nuq[q] - s_dop[k], sp_tab);
                        bhi = log2_I_nu_fast(load_spec_regs(lds_tab(s_par) + __mul24(k + 1, VAG_NPAR / 2)), 1, sc, nuq[q] - s_dop[k + 1], sp_tab);
                    } else if (MODE == FLUX_SYN_IC) {
                        blo = log2_I_nu_ic(s_par + k * VAG_NPAR, 1, cq_row + k, K_all, sc, nuq[q] - s_dop[k], sp_tab);
                        bhi = log2_I_nu_ic(s_par + (k + 1) * VAG_NPAR, 1, cq_row + (k + 1), K_all, sc, nuq[q] - s_dop[k + 1],
                                           sp_tab);
                    } else {
                        const double* hdr = a.ichdr + (a.lay.cell_off[m] + (long long)rep * K_all + k0 + k) * FLUX_IC_HDR;
                        blo = ic_table_eval(hdr, a.icpool, nuq[q] - s_dop[k], &breach);
                        bhi = ic_table_eval(hdr + FLUX_IC_HDR, a.icpool, nuq[q] - s_dop[k + 1], &breach);
                    }
                    blo += s_geom[k];
                    bhi += s_geom[k + 1];
                    const double sl = (bhi - blo) * (1.0 / (s_t[k + 1] - s_t[k]));
                    if (isfinite(sl)) acc[q] += exp2_fast(blo + (t - s_t[k]) * sl);
                }
            }
        }
    }
    {
        double* dst = chunk_partial + (size_t)((p1 - 1) / a.chunk) * a.n;  // the last (possibly short) chunk of this wavefront
#pragma unroll
        for (int q = 0; q < NSLOT; ++q) {
            const int s = tid + q * SERIES_THREADS;
            if (s < a.n) dst[s] = piece == 0 ? acc[q] : dst[s] + acc[q];
        }
    }
    }  // pieces of the lattice
    VAG_SER_MARK(c_pts);
#ifdef VAG_SERIES_STAMPS
    if (m == 0 && vb == 0 && tid == 0)
        printf("series wave 0: rows %d K %d  cycles: staging %lld  eat %lld  bracket %lld  items %lld  interp+rest %lld\n", p1 - p0, K, c_stage, c_eat, c_brk, c_items, c_pts);
#endif
    if (MODE == FLUX_SSC && breach) atomicOr(a.ic_status + m, ic_breach_status(breach));
}

// chi^2 / log-likelihood of Fitter._evaluate (VegasAfterglow/fitting/fitter.py:497-533,
// fitting/samplers.py:61-70): one wavefront per walker.
__global__ void __launch_bounds__(64)
vag_loglike_kernel(const double* __restrict__ flux /* [nb][n] */, int n, const double* __restrict__ ln_flux,
                   const double* __restrict__ ln_err, const double* __restrict__ weight,
                   const int* __restrict__ valid /* [nb] */, double* __restrict__ out) {
    const int m = blockIdx.x;
    double chi2 = 0;
    for (int i = threadIdx.x; i < n; i += 64) {
        const double f = flux[(size_t)m * n + i];
        const double fm = (f != f) ? f : (f > 1e-300 ? f : 1e-300);
        const double q = (ln_flux[i] - log(fm)) / ln_err[i];
        chi2 += weight[i] * (q * q);
    }
    chi2 = wave_sum(chi2);
    if (threadIdx.x == 0) out[m] = (valid[m] && isfinite(chi2)) ? -0.5 * chi2 : -INFINITY;
}

}  // namespace vag
