// vag_ic.h -- inverse-Compton cooling and SSC photon spectra on the device.
// Reference: src/radiation/inverse-compton.h:270-776, src/radiation/inverse-compton.cpp:18-378.
#pragma once
#include "vag_device.h"
#include "vag_grid_kernel.h"

namespace vag {

constexpr double C_H = 6.63e-27 * U_ERG * U_SEC;  // con::h, src/util/macros.h:95

// ---- BrokenPowerLaw<5> restricted to the three segments build_segments() can create for regimes 0-2
//      (src/util/utilities.h:21-78, inverse-compton.cpp:112-131) ----
// The fixed points of vag_ic_cooling_kernel walk a row cell by cell on ONE lane: their instruction count is their latency.  So the
// pieces they iterate -- Y_T, the cooling breaks, the segments of Y(gamma), gamma_c -- run on the fast kernels of vag_device.h
// (log2_fast / exp2_sat: ~1 ulp, the library forms behind them for zeros, infinities and NaN) and on reciprocal / square-root
// estimates + Newton steps, with the library division and square root kept where an operand may be 0 or inf (B = 0 cells).
VAG_DEV double ic_pow(double a, double b) { return exp2_sat(b * log2_fast(a)); }  // fast_pow (fast-math.h:138-140)
VAG_DEV double ic_rcp(double x) { return (isfinite(x) && x != 0) ? rcp_fast(x) : 1.0 / x; }
VAG_DEV double ic_sqrt(double x) { return (isfinite(x) && x > 0) ? sqrt_fast(x) : sqrt(x); }

// Everything below is carried in log2: the segments are stored that way, and a cell's fixed point needs two logarithms per
// iteration (of gamma_c and of Y_T) instead of the seven its pieces would each take on their own.
struct Bpl {
    int size;
    double slope[3], lg2_lower[3], lg2_const[3];
    VAG_DEV void first_lg(double lg2_norm) {  // first_segment(norm, 1, 0)
        size = 1;
        slope[0] = 0;
        lg2_lower[0] = 0;
        lg2_const[0] = lg2_norm;
    }
    // (constant indices throughout: with a running index the three small arrays live in scratch memory, 168 B per lane of
    // vag_ic_cooling_kernel, and every evaluation of Y(gamma) in its fixed-point loops went through it)
    VAG_DEV void add_lg(double ll, double sl) {  // add_segment(lower, slope) given log2(lower)
        if (size == 1) {
            const double val = lg2_const[0] + slope[0] * ll;
            slope[1] = sl;
            lg2_lower[1] = ll;
            lg2_const[1] = val - sl * ll;
        } else {
            const double val = lg2_const[1] + slope[1] * ll;
            slope[2] = sl;
            lg2_lower[2] = ll;
            lg2_const[2] = val - sl * ll;
        }
        ++size;
    }
    VAG_DEV double eval_lg(double lx) const {
        if (size > 2 && lx >= lg2_lower[2]) return exp2_sat(lg2_const[2] + slope[2] * lx);
        if (size > 1 && lx >= lg2_lower[1]) return exp2_sat(lg2_const[1] + slope[1] * lx);
        return size > 0 ? exp2_sat(lg2_const[0] + slope[0] * lx) : 0.0;
    }
    VAG_DEV double eval(double x) const { return eval_lg(log2_fast(x)); }
};

// ---- InverseComptonY (inverse-compton.h:28-76, inverse-compton.cpp:18-187) ----
// (Its gamma0 -- update_gamma0, inverse-compton.cpp:51-93 -- is only read by code the reference has commented out, :101-114, and
// is not formed here.)
struct IcY {
    double gamma_m_hat, gamma_c_hat, gamma_self, Y_T;
    int regime;
    double gamma_m_, B_, p_, gamma_self3;
    double lg2_gamma_m, lg2_gamma_m_hat, lg2_gamma_self3;
    Bpl seg;

    VAG_DEV double gamma_hat(double g) const { return dmax(gamma_self3 * ic_rcp(g * g), 1.0); }
    VAG_DEV double gamma_spectrum(double g) const { return seg.eval(g); }
    VAG_DEV double gamma_spectrum_lg(double lg2_g) const { return seg.eval_lg(lg2_g); }
    VAG_DEV void build_segments(double lg2_gamma_c_hat) {
        seg.first_lg(log2_fast(Y_T));
        if (regime == 1) {
            seg.add_lg(lg2_gamma_c_hat, 0.5 * (p_ - 3.0));
            seg.add_lg(lg2_gamma_m_hat, -4.0 / 3.0);
        } else if (regime == 2) {
            seg.add_lg(lg2_gamma_m_hat, -0.5);
            seg.add_lg(lg2_gamma_c_hat, -4.0 / 3.0);
        }
    }
    VAG_DEV double lg2_gamma_hat(double lg2_g) const { return dmax(lg2_gamma_self3 - 2 * lg2_g, 0.0); }
    // update_cooling_breaks(gamma_c, Y_T) given log2(gamma_c)
    VAG_DEV void update_cooling_breaks_lg(double gamma_c, double lg2_gamma_c, double YT) {
        gamma_c_hat = gamma_hat(gamma_c);
        Y_T = YT;
        regime = (gamma_m_ < gamma_c) ? 1 : 2;
        build_segments(lg2_gamma_hat(lg2_gamma_c));
    }
    VAG_DEV void update_cooling_breaks(double gamma_c, double YT) { update_cooling_breaks_lg(gamma_c, log2_fast(gamma_c), YT); }
    // the part of the constructor that does not depend on gamma_c
    VAG_DEV void init_base(double gamma_m, double p, double B) {
        const double nu_m = syn_freq(gamma_m, B);
        gamma_m_hat = dmax((C_ME * C_C2 / C_H) * ic_rcp(nu_m), 1.0);
        lg2_gamma_m = log2_fast(gamma_m);
        lg2_gamma_m_hat = log2_fast(gamma_m_hat);
        const double lg2_self = (lg2_gamma_m_hat + 2 * lg2_gamma_m) * (1.0 / 3.0);  // (gamma_m_hat gamma_m^2)^(1/3)
        gamma_self = exp2_sat(lg2_self);
        gamma_self3 = gamma_self * gamma_self * gamma_self;
        lg2_gamma_self3 = 3 * lg2_self;
        B_ = B;
        gamma_m_ = gamma_m;
        p_ = p;
        gamma_c_hat = 1.0;
        Y_T = 0.0;
        regime = 0;
        seg.size = 0;
    }
    VAG_DEV void init(double gamma_m, double gamma_c, double p, double B, double YT, bool is_KN) {
        init_base(gamma_m, p, B);
        if (is_KN) {
            update_cooling_breaks(gamma_c, YT);
        } else {
            gamma_c_hat = gamma_hat(gamma_c);
            Y_T = YT;
            regime = 0;
            build_segments(0.0);
        }
    }
};

VAG_DEV double syn_gamma(double nu, double B) { return sqrt((4 * C_PI * C_ME * C_C / (3 * C_E)) * (nu / B)); }
VAG_DEV double icy_nu_spectrum(const IcY& y, double nu) { return y.gamma_spectrum(syn_gamma(nu, y.B_)); }

// compute_Thomson_Y (inverse-compton.cpp:192-198) given log2(gamma_c / gamma_m); `e_over_B` = eps_e / eps_B
VAG_DEV double thomson_Y_lg(double e_over_B, double p, bool below_gamma_m, double lg2_gc_over_gm) {
    const double eta_e = below_gamma_m ? 1 : exp2_sat((2 - p) * lg2_gc_over_gm);
    return 0.5 * (ic_sqrt(fma(4. * e_over_B, eta_e, 1.)) - 1.);
}
VAG_DEV double gamma_c_of(double t_comv, double B, double Y) {
    const double gamma_bar = (6 * C_PI * C_ME * C_C / C_SIGMAT) * ic_rcp(B * B * (1 + Y) * t_comv);
    return (gamma_bar + ic_sqrt(gamma_bar * gamma_bar + 4)) * 0.5;
}
VAG_DEV double gamma_M_of(double B, double Y) {
    if (B == 0) return INFINITY;
    return sqrt(6 * C_PI * C_E / C_SIGMAT / (B * (1 + Y)));
}

// compute_syn_gamma_a with IC terms (synchrotron.cpp:212-246)
VAG_DEV double syn_gamma_a_ic(double B, double I_peak, double gamma_m, double gamma_c, double p, const IcY& Ys, double Y_c) {
    const double gamma_peak = dmin(gamma_m, gamma_c);
    const double nu_peak = syn_freq(gamma_peak, B);
    const double kT = (gamma_peak - 1) * (C_ME * C_C2) / 3;
    double nu_a = fast_pow(I_peak * C_C2 / (cbrt(nu_peak) * 2 * kT), 0.6);
    if (nu_a > nu_peak) {
        if (gamma_c > gamma_m) {
            const double nu_m = syn_freq(gamma_m, B);
            nu_a = fast_pow(I_peak * C_C2 / (2 * kT) * fast_pow(nu_m, p / 2), 2 / (p + 4));
            const double nu_c = syn_freq(gamma_c, B);
            if (nu_a > nu_c) {
                nu_a = fast_pow(I_peak * C_C2 / (2 * kT) * sqrt(nu_c) * fast_pow(nu_m, p / 2), 2 / (p + 5));
                nu_a *= fast_pow((1 + Y_c) / (1 + icy_nu_spectrum(Ys, nu_a)), 2 / (p + 5));
            }
        } else {
            const double nu_c = syn_freq(gamma_c, B);
            nu_a = fast_pow(I_peak * C_C2 / (2 * kT) * sqrt(nu_c), 0.4);
            nu_a *= fast_pow((1 + Y_c) / (1 + icy_nu_spectrum(Ys, nu_a)), 0.4);
            const double nu_m = syn_freq(gamma_m, B);
            if (nu_a > nu_m) {
                nu_a = fast_pow(I_peak * C_C2 / (2 * kT) * sqrt(nu_c) * fast_pow(nu_m, p / 2), 2 / (p + 5));
                nu_a *= fast_pow((1 + Y_c) / (1 + icy_nu_spectrum(Ys, nu_a)), 2 / (p + 5));
            }
        }
    }
    return syn_gamma(nu_a, B) + 1;
}

VAG_DEV int determine_regime(double a, double c, double m) {
    if (a <= m && m <= c) return 1;
    if (m <= a && a <= c) return 2;
    if (a <= c && c <= m) return 3;
    if (c <= a && a <= m) return 4;
    if (m <= c && c <= a) return 5;
    if (c <= m && m <= a) return 6;
    return 0;
}

// Electron state of one cell kept between the radiation kernels of the SSC path (SynElectrons,
// src/radiation/synchrotron.h:20-43), SoA over cells.
enum { VE_GAMMA_M = 0, VE_GAMMA_C, VE_GAMMA_A, VE_GAMMA_M_MAX, VE_COLUMN_DEN, VE_YC, VE_REGIME, VAG_NELEC };
// InverseComptonY of one cell, SoA over cells
enum {
    VY_GAMMA_M_HAT = 0, VY_GAMMA_C_HAT, VY_YT, VY_B, VY_NSEG,
    VY_S0, VY_L0, VY_C0, VY_S1, VY_L1, VY_C1, VY_S2, VY_L2, VY_C2,
    VAG_NICY
};

VAG_DEV void icy_store(const IcY& y, double* base, long long n_cells, long long c) {
    base[VY_GAMMA_M_HAT * n_cells + c] = y.gamma_m_hat;
    base[VY_GAMMA_C_HAT * n_cells + c] = y.gamma_c_hat;
    base[VY_YT * n_cells + c] = y.Y_T;
    base[VY_B * n_cells + c] = y.B_;
    base[VY_NSEG * n_cells + c] = (double)y.seg.size;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        base[(VY_S0 + 3 * s) * n_cells + c] = s < y.seg.size ? y.seg.slope[s] : 0;
        base[(VY_L0 + 3 * s) * n_cells + c] = s < y.seg.size ? y.seg.lg2_lower[s] : INFINITY;
        base[(VY_C0 + 3 * s) * n_cells + c] = s < y.seg.size ? y.seg.lg2_const[s] : 0;
    }
}
// Y(gamma) from the stored segments: log2 Y = const_i + slope_i log2(gamma)
VAG_DEV double icy_lg2_Y(const double* base, long long n_cells, long long c, double lg2_gamma) {
    const int n = (int)base[VY_NSEG * n_cells + c];
    if (n == 0) return -INFINITY;
    for (int i = n - 1; i > 0; --i)
        if (lg2_gamma >= base[(VY_L0 + 3 * i) * n_cells + c])
            return base[(VY_C0 + 3 * i) * n_cells + c] + base[(VY_S0 + 3 * i) * n_cells + c] * lg2_gamma;
    return base[VY_C0 * n_cells + c] + base[VY_S0 * n_cells + c] * lg2_gamma;
}

// SynElectrons::compute_column_den (synchrotron.cpp:261-309)
VAG_DEV double electron_column_den(double gamma, double gamma_m, double gamma_c, double gamma_M, double p, int regime,
                                   double column_den, double Y_c, double Y_at_gamma) {
    // exp and fast_pow through the fast exp2 / log2 kernels (3e-16 / 1 ulp): this runs once per electron energy of every cell
    // on a lone wavefront, where the library versions were a third of the setup time of vag_ic_photon_kernel
    constexpr double LOG2E_ = 1.4426950408889634;
    double spec;
    if (regime == 1 || regime == 2 || regime == 5) {
        spec = (p - 1) / gamma_m * exp2_sat((-gamma / gamma_M - gamma_m / gamma) * LOG2E_) *
               exp2_sat(-p * log2_fast(gamma / gamma_m)) * gamma_c / (gamma + gamma_c);
    } else if (regime == 3 || regime == 4 || regime == 6) {
        spec = exp2_sat((-gamma / gamma_M - gamma_c / gamma) * LOG2E_) * gamma_c / (gamma * gamma) /
               (1.0 + exp2_sat((p - 1) * log2_fast(gamma / gamma_m)));
    } else {
        spec = 0;
    }
    if (gamma <= gamma_c) return column_den * spec;
    return column_den * spec * (1 + Y_c) / (1 + Y_at_gamma);
}

// Klein-Nishina cross-section ratio sigma/sigma_T (inverse-compton.cpp:257-283)
VAG_DEV double compton_ratio_from_x(double x) {
    if (x < 1e-2) return 1 - 2 * x;
    if (x > 1e2) return 3. / 8 * (log(2 * x) + 0.5) / x;
    const double l = log1p(2.0 * x);
    const double invx = 1.0 / x, invx2 = invx * invx;
    const double term1 = 1.0 + 2.0 * x, invt1 = 1.0 / term1, invt1_2 = invt1 * invt1;
    const double a = (1.0 + x) * invx2 * invx;
    const double b = 2.0 * x * (1.0 + x) * invt1 - l;
    const double c = 0.5 * l * invx;
    const double d = (1.0 + 3.0 * x) * invt1_2;
    return 0.75 * (a * b + c - d);
}
constexpr int KN_LUT_N = 128;
constexpr double KN_LG2_XMIN = -6.6438561897747247, KN_LG2_XMAX = 6.6438561897747247;
// lut: [2][KN_LUT_N] = ratio, log2 ratio (ComptonSigmaLUT, inverse-compton.cpp:285-343), built on the host
// compton_correction_pair (inverse-compton.cpp:356-385) at a node of a log-spaced lattice, given log2(nu): 1 - 2x below the
// table, the asymptotic form above it, linear in log2 x inside.  Inside the table's range the position comes from the
// logarithm directly (no exp2 -> log2 round trip per node), and the two tails are the closed forms written in log2(x):
// ln(2x) = (log2 x + 1) ln 2, 1/x = 2^-log2 x.  One wavefront usually holds nodes of all three ranges, so every branch counts.
VAG_DEV void compton_correction_pair_lg2(double lg2_nu, const double* __restrict__ lut, double& corr, double& lg2_corr) {
    const double lg2_x = lg2_nu + log2(C_H / (C_ME * C_C2));
    if (!(lg2_x > KN_LG2_XMIN)) {  // x <= 1e-2 (NaN lands here too)
        const double x = exp2_sat(lg2_x);
        if (!(x > 0)) {
            corr = 0;
            lg2_corr = -INFINITY;
            return;
        }
        corr = 1 - 2 * x;
        lg2_corr = -(2 * x + 2 * x * x) * 1.4426950408889634;
        return;
    }
    if (lg2_x >= KN_LG2_XMAX) {  // x >= 1e2: sigma/sigma_T = 3/8 (ln 2x + 1/2) / x
        const double a = 0.375 * ((lg2_x + 1.0) * LN2 + 0.5);
        corr = a * exp2_sat(-lg2_x);
        lg2_corr = log2_fast(a) - lg2_x;
        return;
    }
    constexpr double inv_step = 1.0 / ((KN_LG2_XMAX - KN_LG2_XMIN) / (double)(KN_LUT_N - 1));
    const double pos = dmin((lg2_x - KN_LG2_XMIN) * inv_step, (double)(KN_LUT_N - 1));
    const int idx = pos >= (double)(KN_LUT_N - 1) ? KN_LUT_N - 2 : (int)pos;
    const double frac = pos - (double)idx;
    corr = lut[idx] + (lut[idx + 1] - lut[idx]) * frac;
    lg2_corr = lut[KN_LUT_N + idx] + (lut[KN_LUT_N + idx + 1] - lut[KN_LUT_N + idx]) * frac;
}

// The same function for a wavefront that holds nodes of all three ranges (vag_ic_photon_kernel's lattice fill): straight-line, the
// table words requested first (two 16-byte gathers) and the two closed-form tails formed while they are in flight, then selected.
// `x` = 2^lg2_x comes from the caller (the product of an electron and a seed lattice node it already holds), so that neither tail
// pays an exp2; 1/x is a reciprocal estimate + two Newton steps.
typedef double vdouble2_u8 __attribute__((ext_vector_type(2), aligned(8)));
VAG_DEV void compton_correction_pair_node(double lg2_x, double x, const double* __restrict__ lut, double& corr, double& lg2_corr) {
    constexpr double inv_step = 1.0 / ((KN_LG2_XMAX - KN_LG2_XMIN) / (double)(KN_LUT_N - 1));
    const bool below = !(lg2_x > KN_LG2_XMIN), above = lg2_x >= KN_LG2_XMAX;  // NaN: below, and then x > 0 fails
    const double pos = __builtin_fmin(__builtin_fmax((lg2_x - KN_LG2_XMIN) * inv_step, 0.0), (double)(KN_LUT_N - 1));
    const int idx = min((int)pos, KN_LUT_N - 2);
    const double frac = pos - (double)idx;
    const vdouble2_u8 r = *reinterpret_cast<const vdouble2_u8*>(lut + idx);
    const vdouble2_u8 l = *reinterpret_cast<const vdouble2_u8*>(lut + KN_LUT_N + idx);
    // x >= 1e2: sigma / sigma_T = 3/8 (ln 2x + 1/2) / x   (a = 2 for the lanes of the other ranges: log2_fast's library path stays shut)
    const double a = above ? 0.375 * ((lg2_x + 1.0) * LN2 + 0.5) : 2.0;
    const double c_hi = a * rcp_fast(above ? x : 1.0), l_hi = log2_fast(a) - lg2_x;
    // x <= 1e-2: 1 - 2x and its logarithm's series
    const bool pos_x = x > 0;
    const double c_lo = pos_x ? 1 - 2 * x : 0.0, l_lo = pos_x ? -(2 * x + 2 * x * x) * 1.4426950408889634 : -INFINITY;
    const double c_in = r.x + (r.y - r.x) * frac, l_in = l.x + (l.y - l.x) * frac;
    corr = below ? c_lo : (above ? c_hi : c_in);
    lg2_corr = below ? l_lo : (above ? l_hi : l_in);
}

}  // namespace vag
