// vag_rs.h -- coupled forward + reverse shock on the device (gfx950).
// Follows src/dynamics/reverse-shock.hpp:22-60, reverse-shock.tpp:11-590, shock-physics.h:40-245,401-437 and
// src/dynamics/shock.cpp:93-137 of the reference; the time lattice follows src/core/grid-refinement.cpp:166-197.
#pragma once
#include "vag_device.h"

namespace vag {

constexpr double SIGMA_CUT = 1e-6;        // defaults::cutoffs::sigma_cut
constexpr double RS_GAMMA_LIMIT = 1.03;   // reverse-shock.tpp:524
constexpr double RS_SEED_FRAC = 1e-8;     // reverse-shock.tpp:343

// state without theta (constant for non-spreading jets; its zero derivative never enters the error norm)
enum { RS_GAMMA = 0, RS_X4, RS_X3, RS_M2, RS_M3, RS_U2, RS_U3, RS_R, RS_TCOMV, RS_EPS4, RS_M4, RS_N };

// ---- per-row log time lattice with the pre-crossing part refined (logspace_with_cross_refinement +
//      make_time_grid, grid-refinement.cpp:166-197, grid-refinement.h:571-581) ----
struct CrossLattice {
    int n, n_pre, n_post, plain;
    double l0, lr, l3, step0, step1, step2;
    VAG_DEV void init(double ts, double t_end, double t_dec, double T0, int t_num, int base_t_num) {
        n = t_num;
        double t_refine = 10 * dmax(t_dec, T0);
        t_refine = dmin(dmax(t_refine, ts), t_end);
        l0 = log10(ts);
        l3 = log10(t_end);
        plain = (t_refine <= ts || t_refine >= t_end);
        step0 = (l3 - l0) / fmax(1.0, (double)(n - 1));
        n_pre = n_post = 0;
        lr = step1 = step2 = 0;
        if (plain) return;
        const double log_total = log10(t_end / ts);
        const double log_after = log10(t_end / t_refine);
        long np = (long)((double)base_t_num * log_after / log_total);
        if (np < 2) np = 2;
        if (np >= t_num) np = t_num / 2;
        n_post = (int)np;
        n_pre = t_num + 1 - n_post;
        lr = log10(t_refine);
        step1 = (lr - l0) / fmax(1.0, (double)(n_pre - 1));
        step2 = (l3 - lr) / fmax(1.0, (double)(n_post - 1));
    }
    VAG_DEV double node(int kk) const {
        double lg;
        if (plain) {
            lg = (n > 1 && kk == n - 1) ? l3 : l0 + step0 * (double)kk;
        } else if (kk < n_pre) {
            lg = (n_pre > 1 && kk == n_pre - 1) ? lr : l0 + step1 * (double)kk;
        } else {
            const int q = kk - n_pre + 1;  // node q of the second segment (its node 0 is t_refine, already emitted)
            lg = (n_post > 1 && q == n_post - 1) ? l3 : lr + step2 * (double)q;
        }
        return exp2_fast(lg * 3.321928094887362347870319429489390175865);  // 10^lg like TimeLattice::node
    }
};

VAG_DEV double smoothstep(double edge0, double edge1, double x) {
    double t = (x - edge0) / (edge1 - edge0);
    t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
    return t * t * (3.0 - 2.0 * t);
}

// compute_downstr_4vel, shock.cpp:93-137
VAG_DEV double downstr_4vel(double gamma_rel, double sigma) {
    const double ad = adiabatic_idx(gamma_rel);
    const double gm1 = gamma_rel - 1, adm2 = ad - 2, adm1 = ad - 1;
    if (sigma <= SIGMA_CUT) return sqrt(dmax(gm1 * adm1 * adm1 / (-ad * adm2 * gm1 + 2), 0.0));
    const double g2 = gamma_rel * gamma_rel, gp1 = gamma_rel + 1;
    const double term1 = -ad * adm2, term2 = g2 - 1;
    const double A = term1 * gm1 + 2;
    const double B = -gp1 * (-adm2 * (ad * g2 + 1) + ad * adm1 * gamma_rel) * sigma - gm1 * (term1 * (g2 - 2) + 2 * gamma_rel + 3);
    const double Cc = gp1 * (ad * (1 - ad / 4) * term2 + 1) * sigma * sigma +
                      term2 * (2 * gamma_rel + adm2 * (ad * gamma_rel - 1)) * sigma + gp1 * gm1 * gm1 * adm1 * adm1;
    const double D = -gm1 * gp1 * gp1 * adm2 * adm2 * sigma * sigma / 4;
    const double b = B / A, c = Cc / A, d = D / A;
    const double P = c - b * b / 3;
    const double Q = 2 * b * b * b / 27 - b * c / 3 + d;
    const double u = sqrt(dmax(-P, 0.0) / 3);
    const double denom = 2 * P * u;
    const double v = (denom != 0) ? dmin(dmax(3 * Q / denom, -1.0), 1.0) : 0.0;
    const double x_max = 2 * u * cos(acos(v) / 3) - b / 3;
    if (x_max <= 0) return 0;
    const double prod = -d / x_max;
    const double sum = (c - prod) / x_max;
    const double uds = (sum + sqrt(dmax(sum * sum - 4 * prod, 0.0))) / 2;
    return sqrt(dmax(uds, 0.0));
}

// compute_4vel_jump, shock-physics.h:58-66
VAG_DEV double jump_4vel(double gamma_rel, double sigma) {
    const double u_down = downstr_4vel(gamma_rel, sigma);
    const double u_up = sqrt((1 + u_down * u_down) * dmax((gamma_rel - 1) * (gamma_rel + 1), 0.0)) + u_down * gamma_rel;
    return (u_down == 0.) ? 4 * gamma_rel : u_up / u_down;
}

// compute_rel_Gamma, shock-physics.h:190-198
VAG_DEV double rel_Gamma(double g1, double g2) {
    const double u1u2 = sqrt(dmax((g1 - 1) * (g1 + 1) * (g2 - 1) * (g2 + 1), 0.0));
    const double d = g1 - g2;
    const double denom = g1 * g2 - 1 + u1u2;
    return denom <= 0 ? 1 : 1 + d * d / denom;
}

VAG_DEV double sound_speed(double G) {  // compute_sound_speed, shock-physics.h:77-80
    const double ad = adiabatic_idx(G);
    return sqrt(dmax(ad * (ad - 1) * (G - 1) / (1 + (G - 1) * ad), 0.0)) * C_C;
}

// ---- the same closed forms for the ODE right-hand side: hardware reciprocal / reciprocal square root + one Newton step (2e-15 /
//      4e-15, vag_dyn_fast.h) instead of the IEEE division (~25 instructions) and square-root (~30) sequences.  The right-hand side
//      of FRShockEqn holds 22 divisions and 8 square roots; the states are integrated to 1e-6.  profiles/r03_pair_stamps.txt: a
//      step attempt of vag_dynamics_pair_kernel took 50 k cycles with the library forms, two thirds of them these sequences. ----
VAG_DEV double rcp1(double x) {  // 1/x, x finite, non-zero, normal
    const double r = __builtin_amdgcn_rcp(x);
#ifdef VAG_PAIR_NEWTON2  // experiment: a second Newton step (<= 1 ulp) -- profiles/r03_rs_structured_diagnostic.txt
    const double r1 = fma(r, fma(-x, r, 1.0), r);
    return fma(r1, fma(-x, r1, 1.0), r1);
#else
    return fma(r, fma(-x, r, 1.0), r);
#endif
}
VAG_DEV double sqrt1(double x) {  // sqrt(x) with sqrt(0) = 0 kept; x >= 0
    const double y = __builtin_amdgcn_rsq(x);
    const double s0 = x * y;
#ifdef VAG_PAIR_NEWTON2
    const double h = 0.5 * y;
    const double s1 = fma(fma(-s0, s0, x), h, s0);
    const double s = fma(fma(-s1, s1, x), h, s1);
#else
    const double s = fma(fma(-s0, s0, x), 0.5 * y, s0);
#endif
    return x > 0 ? s : 0.0;
}
VAG_DEV double rel_Gamma_f(double g1, double g2) {
    const double u1u2 = sqrt1(dmax((g1 - 1) * (g1 + 1) * (g2 - 1) * (g2 + 1), 0.0));
    const double d = g1 - g2;
    const double denom = g1 * g2 - 1 + u1u2;
    return denom <= 0 ? 1 : 1 + d * d * rcp1(denom);
}
VAG_DEV double jump_4vel_f(double gamma_rel, double sigma) {  // sigma == 0: the hydrodynamic jump; else the exact cubic
    if (sigma > SIGMA_CUT) return jump_4vel(gamma_rel, sigma);
    const double ad = 4.0 / 3.0 + rcp1(3 * gamma_rel);
    const double gm1 = gamma_rel - 1, adm2 = ad - 2, adm1 = ad - 1;
    const double u_down = sqrt1(dmax(gm1 * adm1 * adm1 * rcp1(-ad * adm2 * gm1 + 2), 0.0));
    const double u_up = sqrt1((1 + u_down * u_down) * dmax(gm1 * (gamma_rel + 1), 0.0)) + u_down * gamma_rel;
    return (u_down == 0.) ? 4 * gamma_rel : u_up * rcp1(u_down);
}
VAG_DEV double sound_speed_f(double G) {
    const double ad = 4.0 / 3.0 + rcp1(3 * G);
    return sqrt1(dmax(ad * (ad - 1) * (G - 1) * rcp1(1 + (G - 1) * ad), 0.0)) * C_C;
}

VAG_DEV double downstr_B(double eps_B, double rho_up, double B_up, double Gamma_th, double comp) {  // shock-physics.h:354-360
    const double e_th = (Gamma_th - 1) * (rho_up * comp) * C_C2;
    return sqrt(8 * C_PI * eps_B * e_th) + B_up * comp;
}

VAG_DEV double Gamma_therm(double U_th, double mass, bool limiter) {  // shock-physics.h:290-301
    if (mass == 0) return 1;
    const double G = U_th / (mass * C_C2) + 1;
    return (limiter && G < GAMMA_CUT) ? 1 : G;
}

// enclosed_thermal_energy (generic Simpson form for every medium), shock-physics.h:427-437
VAG_DEV double enclosed_thermal_energy_generic(const Medium& med, double r, double Gamma, double ad, double eps_e) {
    const double cooling_exp = 3 * (ad - 1);
    const int N = 32;
    const double u_max = log(r), u_min = u_max - 18, h = (u_max - u_min) / N;
    auto f = [&](double u) {
        const double ri = exp(u);
        return medium_rho(med, ri) * ri * ri * ri * pow(ri / r, cooling_exp);
    };
    double sum = f(u_min) + f(u_max);
    for (int i = 1; i < N; i += 2) sum += 4 * f(u_min + i * h);
    for (int i = 2; i < N; i += 2) sum += 2 * f(u_min + i * h);
    return (1 - eps_e) * (Gamma - 1) * C_C2 * (sum * h / 3);
}

// FRShockEqn (reverse-shock.tpp:22-293) for the built-in jets: no extra energy / mass injection, sigma0 = 0.
struct PairShock {
    Medium med;
    double Gamma4, deps0_dt, dm0_dt, u4, T0;
    double gamma_m_coeff, gamma_c_coeff, eps_e_eff, p;  // RadiativeEfficiency(rad_fwd)
    double cs4;                                         // sound speed of the unshocked shell (constant)
    double beta4;
    double inj_L, inj_t0, inj_q;                        // magnetar injection (0 when off / outside theta_c), cf. FwdShock

    VAG_DEV double inject_weight(double t) const {  // smoothstep(1.5 T0, 0.5 T0, t) with the edge difference's reciprocal at hand
        double w = (t - 1.5 * T0) * (-1.0 / T0);
        w = w < 0.0 ? 0.0 : (w > 1.0 ? 1.0 : w);
        return w * w * (3.0 - 2.0 * w);
    }
    VAG_DEV double shell_sigma(const double* s) const {
        const double sigma = s[RS_EPS4] / (Gamma4 * s[RS_M4] * C_C2) - 1;
        return (sigma > SIGMA_CUT) ? sigma : 0;
    }
    VAG_DEV bool crossing_complete(const double* s, double t) const {  // reverse-shock.tpp:46-58
        if (s[RS_M3] < 0.999 * s[RS_M4]) return false;
        return !(inject_weight(t) > 1e-6);
    }
#ifdef VAG_PAIR_EXACT_MATH
    VAG_DEV void operator()(const double* raw, double* d, double t) const { rhs_exact(raw, d, t); }
#else
    // FRShockEqn::operator() (reverse-shock.tpp:60-293), same statements and guards as rhs_exact below, on rcp1 / sqrt1
    VAG_DEV void operator()(const double* raw, double* d, double t) const {
        const double Gamma = dmin(dmax(raw[RS_GAMMA], 1.0), Gamma4);
        const double m4 = raw[RS_M4];
        const double m3 = dmin(dmax(raw[RS_M3], 0.0), dmax(m4, 0.0));
        const double x3 = dmax(raw[RS_X3], 0.0), U3 = dmax(raw[RS_U3], 0.0);
        const double x4 = raw[RS_X4], m2 = raw[RS_M2], U2 = raw[RS_U2], r = raw[RS_R], t_comv = raw[RS_TCOMV];
        const double u3 = sqrt1((Gamma - 1) * (Gamma + 1));
        const double dr = u3 * (Gamma + u3) * C_C;
        const double dtc = Gamma + u3;
        d[RS_R] = dr;
        d[RS_TCOMV] = dtc;
        const double rho = med.type == VAG_MEDIUM_ISM ? med.rho_ism : (med.generic ? medium_rho(med, r) : fma(med.A, rcp1(fma(r, r, med.r02)), med.rho_ism));
        const double dm2 = r * r * rho * dr;
        d[RS_M2] = dm2;
        const double inject_w = inject_weight(t);
        const double deps_inj = (inj_L != 0) ? inj_L * exp2_sat(-inj_q * log2_fast(1 + t * inj_t0)) : 0.0;
        const double deps4 = ((inject_w > 1e-6) ? inject_w * deps0_dt : 0) + deps_inj;
        const double dm4 = (inject_w > 1e-6) ? inject_w * dm0_dt : 0;
        d[RS_EPS4] = deps4;
        d[RS_M4] = dm4;
        const double Gamma34 = rel_Gamma_f(Gamma4, Gamma);
        const double inv_m4 = rcp1(m4);  // m4 > 0 along a row (seeded by dm0_dt dt); the m4 <= 0 guards below keep the reference's
        double sigma = raw[RS_EPS4] * inv_m4 * (1.0 / (Gamma4 * C_C2)) - 1;
        sigma = (sigma > SIGMA_CUT) ? sigma : 0;
        const double comp_ratio = jump_4vel_f(Gamma34, sigma);
        const double f = (dm0_dt > 0 && dm4 > 0) ? dmin(dm4 * rcp1(dm0_dt), 1.0) : 0.0;
        const double cs34 = sound_speed_f(Gamma34);
        const double inv_G = rcp1(Gamma);
        {
            const double se = cs4 * dtc;
            d[RS_X4] = (f > 1e-6) ? f * u4 + (1 - f) * se : se;
        }
        double dx3;
        {
            const double se = cs34 * dtc;
            dx3 = se;
            if (!(m4 <= 0)) {
                const double remaining = dmax(m4 - m3, 0.0);
                const double crossing_w = f + (1.0 - f) * remaining * inv_m4;
                const double penetration = Gamma * comp_ratio * (1.0 / Gamma4) - 1;
                if (!(crossing_w < 1e-6) && !(penetration <= 0)) {
                    const double beta3 = u3 * inv_G;
                    const double dx3dt = (Gamma4 - Gamma) * (Gamma4 + Gamma) * (1 + beta3) * C_C *
                                         rcp1(Gamma4 * Gamma4 * (beta3 + beta4) * penetration);
                    double crossing = fabs(dx3dt * Gamma);
                    if (penetration < 1) {
                        const double va2 = sigma * rcp1(1 + sigma);
                        const double cs2 = cs34 * cs34 * (1.0 / (C_C * C_C));
                        crossing = dmin(crossing, sqrt1(va2 + cs2 * (1 - va2)) * C_C * dtc);
                    }
                    dx3 = crossing_w * crossing + (1.0 - crossing_w) * se;
                }
            }
            d[RS_X3] = dx3;
        }
        const double inv_x4 = rcp1(x4);
        double dm3 = 0.;
        if (!(m4 <= 0)) {
            const double remaining = dmax(m4 - m3, 0.0);
            if (!(remaining <= 0 && f < 1e-6)) {
                const double eff_mass = f * m4 + (1.0 - f) * remaining;
                const double dm3dt = (eff_mass * comp_ratio * inv_x4) * dx3;
                if (f > 1e-6) {
                    double cw = m3 * inv_m4;  // smoothstep(0, 1, m3 / m4)
                    cw = cw < 0.0 ? 0.0 : (cw > 1.0 ? 1.0 : cw);
                    const double cap_w = cw * cw * (3.0 - 2.0 * cw);
                    dm3 = (1.0 - cap_w) * dm3dt + cap_w * dmin(dm3dt, dm4);
                } else {
                    dm3 = dm3dt;
                }
            }
        }
        d[RS_M3] = dm3;
        const double ad2 = 4.0 / 3.0 + inv_G * (1.0 / 3.0), ad3 = 4.0 / 3.0 + rcp1(3 * Gamma34);
        const double inv_r = rcp1(r);
        double dU2, dU3;
        {
            const double e_th = (Gamma - 1) * 4 * Gamma * rho * C_C2;
            double eps_rad = 0;  // RadiativeEfficiency, shock-physics.h:247-288
            if (eps_e_eff != 0) {
                const double gamma_m = gamma_m_coeff * (Gamma - 1) + 1;
                const double gamma_bar = gamma_c_coeff * rcp1(e_th * t_comv);
                const double gamma_c = 0.5 * (gamma_bar + sqrt1(gamma_bar * gamma_bar + 4));
                const double ratio = gamma_m * rcp1(gamma_c);
                eps_rad = (ratio < 1 && p > 2) ? eps_e_eff * exp2_sat((p - 2) * log2_fast(ratio)) : eps_e_eff;
            }
            double dlnv = 2 * dr * inv_r;
            if (x4 > 0) dlnv += d[RS_X4] * inv_x4;
            dU2 = (1 - eps_rad) * (dm2 * (Gamma - 1) * C_C2) + (-(ad2 - 1) * dlnv * U2);
        }
        {
            double dlnv = 2 * dr * inv_r;
            if (x3 > 0) dlnv += dx3 * rcp1(x3);
            dU3 = dm3 * (Gamma34 - 1) * C_C2 + (-(ad3 - 1) * dlnv * U3);
        }
        d[RS_U2] = dU2;
        d[RS_U3] = dU3;
        {
            const double G2 = Gamma * Gamma, inv_G2 = inv_G * inv_G;
            const double Geff2 = (ad2 * G2 - ad2 + 1) * inv_G, Geff3 = (ad3 * G2 - ad3 + 1) * inv_G;
            const double dGeff2 = (ad2 * G2 + ad2 - 1) * inv_G2, dGeff3 = (ad3 * G2 + ad3 - 1) * inv_G2;
            const double a = (Gamma - 1) * C_C2 * dm2 + (Gamma - Gamma4) * C_C2 * dm3 + Geff2 * dU2 + Geff3 * dU3 - deps_inj;
            const double b = (m2 + m3) * C_C2 + dGeff2 * U2 + dGeff3 * U3;
            const double q = -a * rcp1(b);
            d[RS_GAMMA] = (b == 0 || isnan(q) || isinf(q)) ? 0 : q;
        }
    }
#endif
    VAG_DEV void rhs_exact(const double* raw, double* d, double t) const {
        const double Gamma = dmin(dmax(raw[RS_GAMMA], 1.0), Gamma4);
        const double m4 = raw[RS_M4];
        const double m3 = dmin(dmax(raw[RS_M3], 0.0), dmax(m4, 0.0));
        const double x3 = dmax(raw[RS_X3], 0.0), U3 = dmax(raw[RS_U3], 0.0);
        const double x4 = raw[RS_X4], m2 = raw[RS_M2], U2 = raw[RS_U2], r = raw[RS_R], t_comv = raw[RS_TCOMV];
        const double u3 = sqrt((Gamma - 1) * (Gamma + 1));
        const double dr = u3 * (Gamma + u3) * C_C;
        const double dtc = Gamma + u3;
        d[RS_R] = dr;
        d[RS_TCOMV] = dtc;
        const double rho = medium_rho(med, r);
        const double dm2 = r * r * rho * dr;
        d[RS_M2] = dm2;
        const double inject_w = smoothstep(T0 * 1.5, T0 * 0.5, t);
        // ejecta.deps_dt (magnetar) feeds region 4 and the energy balance (reverse-shock.tpp:84-86,228-230)
        const double deps_inj = (inj_L != 0) ? inj_L * exp2_sat(-inj_q * log2_fast(1 + t * inj_t0)) : 0.0;
        const double deps4 = ((inject_w > 1e-6) ? inject_w * deps0_dt : 0) + deps_inj;
        const double dm4 = (inject_w > 1e-6) ? inject_w * dm0_dt : 0;
        d[RS_EPS4] = deps4;
        d[RS_M4] = dm4;
        const double Gamma34 = rel_Gamma(Gamma4, Gamma);
        double sigma = raw[RS_EPS4] / (Gamma4 * m4 * C_C2) - 1;
        sigma = (sigma > SIGMA_CUT) ? sigma : 0;
        const double comp_ratio = jump_4vel(Gamma34, sigma);
        const double f = (dm0_dt > 0 && dm4 > 0) ? dmin(dm4 / dm0_dt, 1.0) : 0.0;
        const double cs34 = sound_speed(Gamma34);
        {
            const double se = cs4 * dtc;
            d[RS_X4] = (f > 1e-6) ? f * u4 + (1 - f) * se : se;
        }
        double dx3;
        {
            const double se = cs34 * dtc;
            dx3 = se;
            if (!(m4 <= 0)) {
                const double remaining = dmax(m4 - m3, 0.0);
                const double crossing_w = f + (1.0 - f) * remaining / m4;
                const double penetration = Gamma * comp_ratio / Gamma4 - 1;
                if (!(crossing_w < 1e-6) && !(penetration <= 0)) {
                    const double beta3 = gamma_to_beta(Gamma);
                    const double dx3dt = (Gamma4 - Gamma) * (Gamma4 + Gamma) * (1 + beta3) * C_C /
                                         (Gamma4 * Gamma4 * (beta3 + beta4) * penetration);
                    double crossing = fabs(dx3dt * Gamma);
                    if (penetration < 1) {
                        const double va2 = sigma / (1 + sigma);
                        const double cs2 = cs34 * cs34 / (C_C * C_C);
                        crossing = dmin(crossing, sqrt(va2 + cs2 * (1 - va2)) * C_C * dtc);
                    }
                    dx3 = crossing_w * crossing + (1.0 - crossing_w) * se;
                }
            }
            d[RS_X3] = dx3;
        }
        double dm3 = 0.;
        if (!(m4 <= 0)) {
            const double remaining = dmax(m4 - m3, 0.0);
            if (!(remaining <= 0 && f < 1e-6)) {
                const double eff_mass = f * m4 + (1.0 - f) * remaining;
                const double dm3dt = (eff_mass * comp_ratio / x4) * dx3;
                if (f > 1e-6) {
                    const double cap_w = smoothstep(0, 1.0, m3 / m4);
                    dm3 = (1.0 - cap_w) * dm3dt + cap_w * dmin(dm3dt, dm4);
                } else {
                    dm3 = dm3dt;
                }
            }
        }
        d[RS_M3] = dm3;
        const double ad2 = adiabatic_idx(Gamma), ad3 = adiabatic_idx(Gamma34);
        double dU2, dU3;
        {
            const double e_th = (Gamma - 1) * 4 * Gamma * rho * C_C2;
            double eps_rad = 0;  // RadiativeEfficiency, shock-physics.h:247-288
            if (eps_e_eff != 0) {
                const double gamma_m = gamma_m_coeff * (Gamma - 1) + 1;
                const double gamma_bar = gamma_c_coeff / (e_th * t_comv);
                const double gamma_c = 0.5 * (gamma_bar + sqrt(gamma_bar * gamma_bar + 4));
                const double ratio = gamma_m / gamma_c;
                eps_rad = (ratio < 1 && p > 2) ? eps_e_eff * exp2_sat((p - 2) * log2_fast(ratio)) : eps_e_eff;
            }
            double dlnv = 2 * dr / r;
            if (x4 > 0) dlnv += d[RS_X4] / x4;
            dU2 = (1 - eps_rad) * (dm2 * (Gamma - 1) * C_C2) + (-(ad2 - 1) * dlnv * U2);
        }
        {
            double dlnv = 2 * dr / r;
            if (x3 > 0) dlnv += dx3 / x3;
            dU3 = dm3 * (Gamma34 - 1) * C_C2 + (-(ad3 - 1) * dlnv * U3);
        }
        d[RS_U2] = dU2;
        d[RS_U3] = dU3;
        {
            const double G2 = Gamma * Gamma;
            const double Geff2 = (ad2 * G2 - ad2 + 1) / Gamma, Geff3 = (ad3 * G2 - ad3 + 1) / Gamma;
            const double dGeff2 = (ad2 * G2 + ad2 - 1) / G2, dGeff3 = (ad3 * G2 + ad3 - 1) / G2;
            const double a = (Gamma - 1) * C_C2 * dm2 + (Gamma - Gamma4) * C_C2 * dm3 + Geff2 * dU2 + Geff3 * dU3 - deps_inj;
            const double b = (m2 + m3) * C_C2 + dGeff2 * U2 + dGeff3 * U3;
            const double q = -a / b;
            d[RS_GAMMA] = (b == 0 || isnan(q) || isinf(q)) ? 0 : q;
        }
    }

    // FRShockEqn::set_init_state, reverse-shock.tpp:312-354 (+ compute_init_comv_shell_width :367-376)
    VAG_DEV void init_state(double* s, double t0, double eps_e_th) const {
        s[RS_R] = beta4 * C_C * t0 * Gamma4 * Gamma4 * (1 + beta4);
        s[RS_TCOMV] = s[RS_R] / sqrt((Gamma4 - 1) * (Gamma4 + 1)) / C_C;
        const double dt = dmin(t0, T0);
        s[RS_EPS4] = deps0_dt * dt;
        s[RS_M4] = dm0_dt * dt;
        s[RS_X4] = (t0 < T0) ? Gamma4 * t0 * beta4 * C_C : Gamma4 * T0 * beta4 * C_C + cs4 * (t0 - T0) * Gamma4;
        s[RS_M2] = enclosed_mass_generic(med, s[RS_R]);
        const double m_jet_total = dm0_dt * T0;
        s[RS_GAMMA] = (m_jet_total > 0 && s[RS_M2] > 0) ? Gamma4 / (1 + s[RS_M2] / m_jet_total) : Gamma4;
        s[RS_U2] = enclosed_thermal_energy_generic(med, s[RS_R], s[RS_GAMMA], adiabatic_idx(s[RS_GAMMA]), eps_e_th);
        const double Gamma34 = rel_Gamma(Gamma4, s[RS_GAMMA]);
        if (Gamma34 > 1 && s[RS_M4] > 0 && s[RS_X4] > 0) {
            const double comp_ratio = jump_4vel(Gamma34, shell_sigma(s));
            s[RS_X3] = s[RS_X4] * RS_SEED_FRAC;
            s[RS_M3] = s[RS_M4] * comp_ratio * s[RS_X3] / s[RS_X4];
            s[RS_U3] = (Gamma34 - 1) * s[RS_M3] * C_C2;
        } else {
            s[RS_M3] = 0;
            s[RS_U3] = 0;
            s[RS_X3] = 0;
        }
    }
};

}  // namespace vag
