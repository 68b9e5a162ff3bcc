// vag_rs_kernels.h -- reverse-shock tier kernels.
#pragma once
#include "vag_kernels.h"
#include "vag_rs.h"

namespace vag {

#ifndef VAG_PAIR_FLAT
#define VAG_PAIR_FLAT 1  // 0: the retry-inside-the-step loop of rounds 1-4 (developer builds, A/B)
#endif
#ifndef VAG_PAIR_K_LDS
#define VAG_PAIR_K_LDS 0  // 1: the flat loop's stage derivatives in LDS instead of registers (developer builds; r05: 368 B of scratch instead of 64, see DESIGN)
#endif

// Parameters of the reverse shock's own radiation passes: the same struct with rvs_rad's (eps_e, eps_B, p, xi_e) and
// its ssc / kn switches in the forward slots, so every post-dynamics kernel runs unchanged on the reverse shock
// (single_shock_emission is called twice with different Radiation objects, pybind/pymodel.h:955-958).
__global__ void vag_rvs_params_kernel(const vag_model_params* __restrict__ params, int nb, vag_model_params* __restrict__ out) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= nb) return;
    vag_model_params p = params[m];
    p.eps_e = p.rvs_eps_e;
    p.eps_B = p.rvs_eps_B;
    p.p = p.rvs_p;
    p.xi_e = p.rvs_xi_e;
    p.flags = ((p.flags & VAG_FLAG_RVS_SSC) ? VAG_FLAG_SSC : 0) | ((p.flags & VAG_FLAG_RVS_KN) ? VAG_FLAG_KN : 0);
    out[m] = p;
}

// ------------------------------------------------------------------------------------------------
// Coupled forward + reverse shock: one lane per representative (model, theta) row.  grid_solve_shock_pair
// (reverse-shock.tpp:470-590): adaptive DOPRI5 on the 11 evolving variables, crossing-end time bisected on the
// dense output, both shocks saved from the same state, early extrapolation of the reverse shock's thermal
// quantities.  shock_fwd / shock_rvs are [VS_*][cells]; inj_idx is per row.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
vag_dynamics_pair_kernel(const vag_model_params* __restrict__ params, int nb, const VagGridMeta* __restrict__ meta,
                         const double* __restrict__ g_theta, const int* __restrict__ g_rep_start,
                         const double* __restrict__ g_tdec, Layout lay, int n_rows, double* __restrict__ shock_fwd,
                         double* __restrict__ shock_rvs, long long n_cells, int* __restrict__ inj_idx,
                         int* __restrict__ row_status, int* __restrict__ fail,
                         const double* __restrict__ g_phi /* (phi, theta) pair rows only */, const double* __restrict__ tminmax,
                         int rows_per_wave /* rows a wavefront carries (pair_rows_per_wave): its lanes pay for each other's rejected steps */) {
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
    PairShock eq;
    medium_init(eq.med, P);
    const double theta0 = g_theta[(size_t)m * M.th_stride + j];
    const double t_dec = g_tdec[((size_t)m * 3 + 0) * M.th_stride + j];
    double t_start_row = g_tdec[((size_t)m * 3 + 1) * M.th_stride + j];
    double t_early_row = g_tdec[((size_t)m * 3 + 2) * M.th_stride + j];
    if (M.rep_phi_stride) {  // the pair's own lattice start (grid-refinement.h:462-469,619-625; the reverse-shock cut of :489-491)
        double ts_raw;
        row_time_start(gamma_to_beta(jet_Gamma0(jet, theta0)), cos(theta0), sin(theta0), cos(g_phi[(size_t)m * M.ph_stride + i_phi]),
                       M.cos_obs, M.sin_obs, tminmax[0] * U_SEC, P.z, dmin(dmin(0.01 * t_dec, 1e-2 * U_SEC), 0.01 * P.duration * U_SEC),
                       t_start_row, t_early_row, ts_raw);
    }
    const int nt = M.n_t;
    const long long c0 = lay.cell_off[m] + (long long)r * nt;
    double* F = shock_fwd + c0;
    double* R = shock_rvs + c0;
    auto put = [&](double* base, int k, double teng, double tcomv, double rr, double G, double Gth, double B, double Np) {
        base[VS_TENG * n_cells + k] = teng;
        base[VS_TCOMV * n_cells + k] = tcomv;
        base[VS_R * n_cells + k] = rr;
        base[VS_GAMMA * n_cells + k] = G;
        base[VS_GAMMA_TH * n_cells + k] = Gth;
        base[VS_B * n_cells + k] = B;
        base[VS_NP * n_cells + k] = Np;
        base[VS_THETA * n_cells + k] = (tcomv == 0 && rr == 0) ? 0.0 : theta0;  // the pair solver never spreads (diff.theta = 0)
    };

    const double T0 = P.duration * U_SEC;
    CrossLattice lat;
    lat.init(t_start_row, M.t_end, t_dec, T0, M.t_num_tot, M.t_num_base);
    auto node = [&](int k) -> double { return M.has_early ? (k == 0 ? t_early_row : lat.node(k - 1)) : lat.node(k); };

    eq.Gamma4 = jet_Gamma0(jet, theta0);
    eq.T0 = T0;
    eq.deps0_dt = jet_eps_k(jet, theta0) / T0;
    eq.dm0_dt = eq.deps0_dt / (eq.Gamma4 * C_C2) / (1 + jet.sigma0);
    eq.u4 = sqrt(eq.Gamma4 * eq.Gamma4 - 1) * C_C;
    eq.gamma_m_coeff = (P.p - 2) / (P.p - 1) * P.eps_e * C_MP / C_ME / P.xi_e;
    eq.gamma_c_coeff = 6 * C_PI * C_ME * C_C / C_SIGMAT / (8 * C_PI * P.eps_B);
    eq.eps_e_eff = P.radiative_fireball ? P.eps_e : 0;
    eq.p = P.p;
    eq.inj_L = (jet.magnetar && theta0 <= jet.theta_c) ? P.mag_L0 * (U_ERG / (4 * C_PI * U_SEC)) : 0.0;
    eq.inj_t0 = jet.magnetar ? 1 / (P.mag_t0 * U_SEC) : 0.0;
    eq.inj_q = P.mag_q;
    eq.cs4 = sound_speed(eq.Gamma4);
    eq.beta4 = gamma_to_beta(eq.Gamma4);
    const double eps_e_th = P.radiative_fireball ? P.eps_e : 0.0;

    int inj = nt;  // Shock::injection_idx default = t_size
    double V3_comv_x = 0, rho3_x = 0, B3_ordered_x = 0;  // FRShockEqn::save_cross_state
    auto save_both = [&](int k, double t_k, const double* q) {
        {   // save_fwd_shock_state (forward-shock.tpp:151-173): region 2
            const double comp = compression_fwd(q[RS_GAMMA]);
            const double rho = medium_rho(eq.med, q[RS_R]);
            const double Gth = q[RS_M2] == 0 ? 1.0 : q[RS_U2] * rcp_fast(q[RS_M2] * C_C2) + 1;  // Gamma_therm without the limiter
            const double e_th = (Gth - 1) * (rho * comp) * C_C2;
            put(F, k, t_k, q[RS_TCOMV], q[RS_R], q[RS_GAMMA], Gth, sqrt_fast(8 * C_PI * P.eps_B * e_th), q[RS_M2] / C_MP);
        }
        const double Gth3 = q[RS_M3] == 0 ? 1.0 : q[RS_U3] * rcp_fast(q[RS_M3] * C_C2) + 1;
        if (k <= inj) {  // save_rvs_shock_state (reverse-shock.tpp:393-426): still crossing
            const double sigma4 = eq.shell_sigma(q);
            const double comp34 = jump_4vel_f(rel_Gamma_f(eq.Gamma4, q[RS_GAMMA]), sigma4);
            const double rho4 = q[RS_M4] * rcp_fast(q[RS_R] * q[RS_R] * q[RS_X4]);
            const double Gth = Gth3 < GAMMA_CUT ? 1.0 : Gth3;  // the limiter of compute_Gamma_therm
            const double B4 = sigma4 > 0 ? sqrt_fast((4 * C_PI * C_C2) * sigma4 * rho4) : 0.0;
            const double e_th = (Gth - 1) * (rho4 * comp34) * C_C2;
            put(R, k, t_k, q[RS_TCOMV], q[RS_R], q[RS_GAMMA], Gth, sqrt_fast(8 * C_PI * P.rvs_eps_B * e_th) + B4 * comp34, q[RS_M3] / C_MP);
        } else {  // after the crossing: frozen shell expanding adiabatically
            const double V3_comv = q[RS_R] * q[RS_R] * q[RS_X3];
            const double comp = V3_comv_x * rcp_fast(V3_comv);
            const double e_th = (Gth3 - 1) * (rho3_x * comp) * C_C2;
            put(R, k, t_k, q[RS_TCOMV], q[RS_R], q[RS_GAMMA], Gth3, sqrt_fast(8 * C_PI * P.rvs_eps_B * e_th) + B3_ordered_x * comp,
                q[RS_M3] / C_MP);
        }
    };

    const double t_first = node(0), t_last = node(nt - 1);
    const double t0 = dmin(t_first, dmin(0.01 * U_SEC, 0.1 * t_dec));
    double s[RS_N];
    eq.init_state(s, t0, eps_e_th);
    if (s[RS_GAMMA] <= RS_GAMMA_LIMIT) {  // set_stopping_shock for both shocks, shock-physics.h:388-397
        for (int k = 0; k < nt; ++k) {
            const double tk = node(k);
            put(F, k, tk, s[RS_TCOMV], s[RS_R], 1, 1, 0, 0);
            put(R, k, tk, s[RS_TCOMV], s[RS_R], 1, 1, 0, 0);
        }
        inj_idx[row] = nt;
        row_status[row] = 0;
        return;
    }
    double rtol = P.rtol;
    if (eq.shell_sigma(s) > 0) rtol *= 0.1;  // defaults::solver::magnetized_rtol_factor
    int k = 0;
    double t_k = t_first;
    {   // nodes before the integration start take the analytic initial state at their own time
        double q[RS_N];
        while (k < nt && t_k < t0) {
            eq.init_state(q, t_k, eps_e_th);
            save_both(k, t_k, q);
            ++k;
            if (k < nt) t_k = node(k);
        }
    }
#if VAG_PAIR_FLAT
    // Flat attempt loop (see Dopri5Flat): a trip of the wavefront is ONE attempt of every unfinished row; a row that rejects simply
    // does not commit.  What follows an accepted step -- the crossing test, the bisection of the crossing time on the dense output, the
    // saves of the lattice nodes the step passed -- reads the candidate before it is committed: the same numbers the retry-loop form
    // read after its commit.
#if VAG_PAIR_K_LDS
    __shared__ double s_k[5 * RS_N * 64];  // 27.5 KB: five wavefronts per CU, more than the register file lets in anyway
    Dopri5Flat<RS_N, KLds<RS_N>> st;
    st.ks.base = s_k + threadIdx.x;
#else
    Dopri5Flat<RS_N> st;
#endif
    st.init(s, t0, 1e-9 * t0, rtol, eq);
    bool crossing = true, pending = false;
    double t_cross = 0;
    int status = 0, fails = 0, steps = 0;
    bool done = !(st.t <= t_last);
    while (__any(!done)) {
        if (!done) {
            if (!st.attempt(eq)) {
                if (++fails >= 500) {  // max_step_checker.hpp:92
                    status = 1;
                    done = true;
                }
            } else {
                fails = 0;
                const double tn = st.t + st.h;
                if (__builtin_expect(++steps > 100000, 0)) {
                    status = 2;
                    done = true;
                } else if (__builtin_expect(tn + st.dt == tn, 0)) {  // dt below one ulp of t: the reference gives up on this row
                    status = 3;
                    done = true;
                } else {
                    if (__builtin_expect(crossing && eq.crossing_complete(st.xn, tn), 0)) {  // locate_crossing_time, reverse-shock.tpp:484-497
                        double t_lo = st.t, t_hi = tn;
                        double q[RS_N];
                        for (int iter = 0; iter < 100 && (t_hi - t_lo) > 1e-12 * t_hi; ++iter) {
                            const double t_mid = 0.5 * (t_lo + t_hi);
                            st.interp(t_mid, q);
                            if (eq.crossing_complete(q, t_mid))
                                t_hi = t_mid;
                            else
                                t_lo = t_mid;
                        }
                        st.interp(t_hi, q);
                        t_cross = t_hi;
                        {   // save_cross_state, reverse-shock.tpp:297-310
                            V3_comv_x = q[RS_R] * q[RS_R] * q[RS_X3];
                            const double sigma4 = eq.shell_sigma(q);
                            const double comp34 = jump_4vel(rel_Gamma(eq.Gamma4, q[RS_GAMMA]), sigma4);
                            const double rho4 = q[RS_M4] / (q[RS_R] * q[RS_R] * q[RS_X4]);
                            rho3_x = rho4 * comp34;
                            B3_ordered_x = sqrt((4 * C_PI * C_C2) * sigma4 * rho4) * comp34;
                        }
                        crossing = false;
                        pending = true;
                    }
                    while (k < nt && tn > t_k) {
                        double q[RS_N];
                        st.interp(t_k, q);
                        if (pending && t_k >= t_cross) {
                            inj = k > 0 ? k : 1;
                            pending = false;
                        }
                        save_both(k, t_k, q);
                        ++k;
                        if (k < nt) t_k = node(k);
                    }
                }
                st.commit();  // (also the step that tripped the step cap or stalled, as Dopri5::step had)
                if (!(st.t <= t_last)) done = true;
            }
        }
    }
#else
    Dopri5<RS_N> st;
    st.init(s, t0, 1e-9 * t0, rtol, eq);
    bool crossing = true, pending = false;
    double t_cross = 0, t_step_start = t0;
    int status = 0;
#ifdef VAG_PAIR_STAMPS  // developer aid: where one row's cycles go
    long long c_step = 0, c_bis = 0, c_save = 0, c_mark = __builtin_readcyclecounter();
    const long long c_begin = c_mark;
    int n_steps = 0, n_waves_att = 0;
#define VAG_PAIR_MARK(acc) do { const long long now_ = __builtin_readcyclecounter(); acc += now_ - c_mark; c_mark = now_; } while (0)
#else
#define VAG_PAIR_MARK(acc) do { } while (0)
#endif
    for (int steps = 0; st.t <= t_last;) {
#ifdef VAG_PAIR_STAMPS
        ++n_waves_att;
#endif
        if (__builtin_expect(!st.step(eq), 0)) {
            status = 1;
            break;
        }
        if (__builtin_expect(++steps > 100000, 0)) {
            status = 2;
            break;
        }
        if (__builtin_expect(st.t + st.dt == st.t, 0)) {  // dt below one ulp of t: the reference gives up on this row
            status = 3;
            break;
        }
        VAG_PAIR_MARK(c_step);
#ifdef VAG_PAIR_STAMPS
        ++n_steps;
#endif
        if (__builtin_expect(crossing && eq.crossing_complete(st.x, st.t), 0)) {  // locate_crossing_time, reverse-shock.tpp:484-497
            double t_lo = t_step_start, t_hi = st.t;
            double q[RS_N];
            for (int iter = 0; iter < 100 && (t_hi - t_lo) > 1e-12 * t_hi; ++iter) {
                const double t_mid = 0.5 * (t_lo + t_hi);
                st.interp(t_mid, q);
                if (eq.crossing_complete(q, t_mid))
                    t_hi = t_mid;
                else
                    t_lo = t_mid;
            }
            st.interp(t_hi, q);
            t_cross = t_hi;
            {   // save_cross_state, reverse-shock.tpp:297-310
                V3_comv_x = q[RS_R] * q[RS_R] * q[RS_X3];
                const double sigma4 = eq.shell_sigma(q);
                const double comp34 = jump_4vel(rel_Gamma(eq.Gamma4, q[RS_GAMMA]), sigma4);
                const double rho4 = q[RS_M4] / (q[RS_R] * q[RS_R] * q[RS_X4]);
                rho3_x = rho4 * comp34;
                B3_ordered_x = sqrt((4 * C_PI * C_C2) * sigma4 * rho4) * comp34;
            }
            crossing = false;
            pending = true;
        }
        VAG_PAIR_MARK(c_bis);
        t_step_start = st.t;
        while (k < nt && st.t > t_k) {
            double q[RS_N];
            st.interp(t_k, q);
            if (pending && t_k >= t_cross) {
                inj = k > 0 ? k : 1;
                pending = false;
            }
            save_both(k, t_k, q);
            ++k;
            if (k < nt) t_k = node(k);
        }
        VAG_PAIR_MARK(c_save);
    }
#ifdef VAG_PAIR_STAMPS
    if (row % 997 == 0)
        printf("pair row %d: accepted %d attempts %d (own) loop trips %d  cycles: total %lld stepping %lld bisection %lld saving %lld\n", row, n_steps,
               st.n_att, n_waves_att, (long long)__builtin_readcyclecounter() - c_begin, c_step, c_bis, c_save);
#endif
#endif
    for (; k < nt; ++k) {  // unreached nodes keep the Shock constructor's defaults (shock.cpp:12-24)
        const double tk = node(k);
        put(F, k, tk, 0, 0, 1, 1, 0, 0);
        put(R, k, tk, 0, 0, 1, 1, 0, 0);
    }
    inj_idx[row] = inj;
    row_status[row] = status;
    if (status > 0 && status < 4) atomicAdd(fail + status, 1);
    // reverse_shock_early_extrap, reverse-shock.tpp:428-469 (this lane re-reads its own row)
    {
        const double* Gth = R + VS_GAMMA_TH * n_cells;
        int idx_cut = 0;
        for (; idx_cut < nt; ++idx_cut)
            if (Gth[idx_cut] > GAMMA_CUT) break;
        if (idx_cut == 0 || idx_cut >= nt - 2 || idx_cut >= inj) return;
        double* rB = R + VS_B * n_cells;
        double* rNp = R + VS_NP * n_cells;
        double* rG = R + VS_GAMMA_TH * n_cells;
        const double* rr = R + VS_R * n_cells;
        const double log2_r = log2(rr[idx_cut]);
        const double log2_Gth = log2(rG[idx_cut] - 1), log2_B = log2(rB[idx_cut]), log2_Np = log2(rNp[idx_cut]);
        const double dl = log2(rr[idx_cut + 2]) - log2_r;
        const double g_slope = (log2(rG[idx_cut + 2] - 1) - log2_Gth) / dl;
        const double B_slope = (log2(rB[idx_cut + 2]) - log2_B) / dl;
        const double N_slope = (log2(rNp[idx_cut + 2]) - log2_Np) / dl;
        for (int q = 0; q < idx_cut; q++) {
            const double dlog2_r = log2(rr[q]) - log2_r;
            rG[q] = 1 + exp2(log2_Gth + g_slope * dlog2_r);
            rB[q] = exp2(log2_B + B_slope * dlog2_r);
            rNp[q] = exp2(log2_Np + N_slope * dlog2_r);
        }
    }
}

}  // namespace vag
