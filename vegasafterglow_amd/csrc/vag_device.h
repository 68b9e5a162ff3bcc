// vag_device.h -- gfx950 device-side physics of the forward-shock synchrotron path.
// All arithmetic FP64.  Each block cites the reference code it must agree with.
#pragma once
#include <hip/hip_runtime.h>

#include "vag_common.h"

#ifdef VAG_HOST_DEBUG  // developer aid: lets a host program step through the device physics
#define VAG_DEV __host__ __device__ inline
#else
#define VAG_DEV __device__ __forceinline__
#endif

namespace vag {

// ---- units and constants: src/util/macros.h:43-110 (same expression order => same roundings) ----
constexpr double U_LEN = 1.5e13;
constexpr double U_CM = 1 / U_LEN;
constexpr double U_SEC = 3e10 / U_LEN;
constexpr double U_CM2 = U_CM * U_CM;
constexpr double U_CM3 = U_CM * U_CM * U_CM;
constexpr double U_G = 1 / 2e33;
constexpr double U_GAUSS = 8.66e-11 / U_SEC;
constexpr double U_HZ = 1 / U_SEC;
constexpr double U_ERG = U_G * U_CM * U_CM / U_SEC / U_SEC;
constexpr double U_FLUX_CGS = U_ERG / U_CM2 / U_SEC;
constexpr double U_FLUX_DEN_CGS = U_ERG / U_CM2 / U_SEC / U_HZ;
constexpr double C_C = 1.0;
constexpr double C_C2 = C_C * C_C;
constexpr double C_MP = 1.67e-24 * U_G;
constexpr double C_ME = C_MP / 1836;
constexpr double C_E = 4.8e-10 / 4.472136e16 / 5.809475e19 / U_SEC;
constexpr double C_E2 = C_E * C_E;
constexpr double C_E3 = C_E2 * C_E;
constexpr double C_PI = 3.14159265358979323846;
constexpr double C_SIGMAT = 6.65e-25 * U_CM * U_CM;
constexpr double GAMMA_CUT = 1.0 + 1e-6;  // src/config/simulation-defaults.h:38-48
constexpr double LN2 = 0.693147180559945309417232121458176568;
constexpr double LOG2E = 1.442695040888963407359924681001892137;
constexpr double SQRT3 = 1.732050807568877293527446341505872367;

VAG_DEV double exp2_fast(double x);
VAG_DEV double exp2_sat(double x);
VAG_DEV double log2_fast(double x);
#ifdef VAG_HOST_DEBUG
struct vdouble2 {
    double x, y;
};
typedef const vdouble2* LdsTab;
#else
typedef double vdouble2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const vdouble2* LdsTab;  // softplus table in LDS: 32-bit address arithmetic
#endif
VAG_DEV LdsTab lds_tab(const double* p) { return (LdsTab)p; }
constexpr int LOG_TAB_N = 64;
constexpr int LOG_TAB_DOUBLES = 2 * LOG_TAB_N;
VAG_DEV double log2_tab(double x, LdsTab tab);  // 17-instruction log2 on a 64-entry LDS table, defined with the fast kernels below

VAG_DEV double dmin(double a, double b) { return b < a ? b : a; }
VAG_DEV double dmax(double a, double b) { return a < b ? b : a; }

// src/util/fast-math.h:179-202 (exact-libm build)
VAG_DEV double log2_softplus(double x) {
    if (x > 20.0) return x;
    if (x < -20.0) return 0.0;
    return log2(1.0 + exp2(x));
}
VAG_DEV double fast_pow(double a, double b) { return exp2(b * log2(a)); }

// 1/x and sqrt(x) for the ODE right-hand sides: hardware estimate + two Newton steps (<= 2 ulp) instead of the ~35- and
// ~25-instruction IEEE sequences.  One lane integrates one row, so a dynamics wavefront is VALU-issue bound and its
// instruction count IS its latency.  x must be finite, non-zero (rcp) / positive (sqrt) and normal.
VAG_DEV double rcp_fast(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.0), r);
    r = fma(r, fma(-x, r, 1.0), r);
    return r;
}
VAG_DEV double sqrt_fast(double x) {
    if (!(x > 0)) return sqrt(x);  // 0, negative, NaN: library semantics
    double y = __builtin_amdgcn_rsq(x);               // ~ 1/sqrt(x)
    y = y * fma(-0.5 * x * y, y, 1.5);                // Newton on 1/sqrt
    double s = x * y;                                 // ~ sqrt(x)
    s = fma(fma(-s, s, x), 0.5 * y, s);               // one correction with the residual in fma: 1.1e-16 measured
    return s;                                         // (profiles/micro/rcp_accuracy.hip; a second one changes nothing)
}
// 1/x to 2e-15 (one Newton step on the 4.6e-8 hardware estimate): enough inside an ODE right-hand side integrated to 1e-6
VAG_DEV double rcp_ode(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    return fma(r, fma(-x, r, 1.0), r);
}

// src/core/physics.h:36-61
VAG_DEV double gamma_to_beta(double g) { return sqrt((g - 1) * (g + 1)) / g; }
VAG_DEV double adiabatic_idx(double g) { return 4.0 / 3.0 + 1 / (3 * g); }
// src/core/grid-refinement.h:33-35
VAG_DEV double structure_weight(double G) { return G * sqrt(dmax((G - 1) * G, 0.0)); }

// ---- jet / medium in code units: src/environment/jet.h:84-259,421-441, medium.h:50-133,
//      unit handling of pybind/pymodel.cpp:47-210 ----
struct Jet {
    int type;
    double theta_c, eps_k, Gamma0, k_e, k_g, norm, inv_theta_c;
    double theta_w, E_iso_cgs, E_iso_w_cgs, Gm1, Gm1_w, T0;
    double sigma0;  // constant ejecta magnetisation (VAG_JET_MAGNETIZED_TOPHAT), 0 otherwise
    int magnetar;   // jet(..., magnetar=Magnetar(L0, t0, q)): generic-Ejecta profile forms + energy injection
    double mag_L0, mag_t0, mag_q;
};
struct Medium {
    int type;
    double rho_ism, A, r02;
    int generic;  // Wind with k_m != 2: the python-level Medium built from PyWind's CGS closed form (pymodel.cpp:167-185)
    double k_m, A_cgs, r0k_cgs, rho_ism_cgs;
};

VAG_DEV void jet_init(Jet& j, const vag_model_params& p) {
    j.type = p.jet_type;
    j.theta_c = p.theta_c;
    j.eps_k = (p.E_iso * U_ERG) / (4 * C_PI);
    j.Gamma0 = p.Gamma0;
    j.k_e = p.k_e;
    j.k_g = p.k_g;
    j.norm = -1 / (2 * p.theta_c * p.theta_c);
    j.inv_theta_c = 1 / p.theta_c;
    j.theta_w = p.theta_w;
    j.E_iso_cgs = p.E_iso;
    j.E_iso_w_cgs = p.E_iso_w;
    j.Gm1 = p.Gamma0 - 1;
    j.Gm1_w = p.Gamma0_w - 1;
    j.T0 = p.duration * U_SEC;
    j.sigma0 = (p.jet_type == VAG_JET_MAGNETIZED_TOPHAT) ? p.sigma0 : 0.0;
    j.magnetar = (p.flags & VAG_FLAG_MAGNETAR) != 0;
    j.mag_L0 = p.mag_L0;
    j.mag_t0 = p.mag_t0;
    j.mag_q = p.mag_q;
}
// Ejecta::deps_dt of math::magnetar_injection (jet.h:518-527) through convert_unit_jet (pymodel.cpp:196-199)
VAG_DEV double jet_deps_dt(const Jet& j, double theta, double t) {
    if (!j.magnetar || !(theta <= j.theta_c)) return 0.0;
    const double tt = 1 + (t / U_SEC) / j.mag_t0;
    return j.mag_L0 * exp2_sat(-j.mag_q * log2_fast(tt)) * (U_ERG / (4 * C_PI * U_SEC));
}
VAG_DEV double jet_eps_k(const Jet& j, double theta) {
    if (j.magnetar && j.type <= VAG_JET_POWERLAW) {  // math::tophat / gaussian / powerlaw in CGS, then convert_unit_jet
        double h;
        if (j.type == VAG_JET_TOPHAT)
            h = theta < j.theta_c ? j.E_iso_cgs : 0;
        else if (j.type == VAG_JET_GAUSSIAN)
            h = j.E_iso_cgs * exp(theta * theta / (-2 * j.theta_c * j.theta_c));
        else
            h = j.E_iso_cgs / (1 + fast_pow(theta / j.theta_c, j.k_e));
        return h * (U_ERG / (4 * C_PI));
    }
    switch (j.type) {
        case VAG_JET_TOPHAT: return theta < j.theta_c ? j.eps_k : 0;
        case VAG_JET_GAUSSIAN: return j.eps_k * exp(theta * theta * j.norm);
        case VAG_JET_POWERLAW: return j.eps_k / (1 + fast_pow(theta / j.theta_c, j.k_e));
        case VAG_JET_MAGNETIZED_TOPHAT: return (theta <= j.theta_c ? j.E_iso_cgs : 0.0) * (U_ERG / (4 * C_PI));
        case VAG_JET_STEP_POWERLAW:  // math::step_powerlaw, jet.h:418-426
            return (theta <= j.theta_c ? j.E_iso_cgs : j.E_iso_w_cgs * fast_pow(theta / j.theta_c, -j.k_e)) * (U_ERG / (4 * C_PI));
        case VAG_JET_POWERLAW_WING:  // math::powerlaw_wing, jet.h:403-411
            return (theta <= j.theta_c ? 0. : j.E_iso_w_cgs * fast_pow(theta / j.theta_c, -j.k_e)) * (U_ERG / (4 * C_PI));
        default: {
            const double h = theta <= j.theta_c ? j.E_iso_cgs : (theta <= j.theta_w ? j.E_iso_w_cgs : 0.);
            return h * (U_ERG / (4 * C_PI));
        }
    }
}
VAG_DEV double jet_Gamma0(const Jet& j, double theta) {
    if (j.magnetar && j.type <= VAG_JET_POWERLAW) {  // math::*_plus_one(theta_c, Gamma0 - 1, ...)
        double h;
        if (j.type == VAG_JET_TOPHAT)
            h = theta < j.theta_c ? j.Gm1 : 0;
        else if (j.type == VAG_JET_GAUSSIAN)
            h = j.Gm1 * exp(theta * theta / (-2 * j.theta_c * j.theta_c));
        else
            h = j.Gm1 / (1 + fast_pow(theta / j.theta_c, j.k_g));
        return h + 1;
    }
    switch (j.type) {
        case VAG_JET_TOPHAT: return theta < j.theta_c ? j.Gamma0 : 1;
        case VAG_JET_GAUSSIAN: return (j.Gamma0 - 1) * exp(theta * theta * j.norm) + 1;
        case VAG_JET_POWERLAW: return (j.Gamma0 - 1) / (1 + fast_pow(theta / j.theta_c, j.k_g)) + 1;
        case VAG_JET_MAGNETIZED_TOPHAT: return theta <= j.theta_c ? j.Gamma0 : 1.0;
        case VAG_JET_STEP_POWERLAW: return (theta <= j.theta_c ? j.Gm1 : j.Gm1_w * fast_pow(theta / j.theta_c, -j.k_g)) + 1;
        case VAG_JET_POWERLAW_WING: return (theta <= j.theta_c ? 0. : j.Gm1_w * fast_pow(theta / j.theta_c, -j.k_g)) + 1;
        default: {
            const double h = theta <= j.theta_c ? j.Gm1 : (theta <= j.theta_w ? j.Gm1_w : 0.);
            return h + 1;
        }
    }
}
VAG_DEV void medium_init(Medium& m, const vag_model_params& p) {
    m.type = p.medium_type;
    m.A = 0;
    m.r02 = 0;
    m.generic = 0;
    m.k_m = 2;
    m.A_cgs = m.r0k_cgs = m.rho_ism_cgs = 0;
    if (m.type == VAG_MEDIUM_ISM) {
        m.rho_ism = (p.n_ism / U_CM3) * C_MP;
    } else {
        const double n_ism = p.n_ism / U_CM3, n0 = p.n0 / U_CM3;
        m.A = p.A_star * 5e11 * U_G / U_CM;
        m.rho_ism = n_ism * C_MP;
        m.r02 = m.A / (n0 * 1.3 * C_MP);
        m.generic = (p.k_m != 2);
        if (m.generic) {
            const double mp_cgs = C_MP / U_G;
            m.k_m = p.k_m;
            m.A_cgs = p.A_star * 5e11 * pow(1e17, p.k_m - 2);
            m.rho_ism_cgs = p.n_ism * mp_cgs;
            m.r0k_cgs = m.A_cgs / (p.n0 * 1.3 * mp_cgs);
        }
    }
}
VAG_DEV double medium_rho(const Medium& m, double r) {
    if (m.type == VAG_MEDIUM_ISM) return m.rho_ism;
    if (m.generic) return (m.A_cgs / (m.r0k_cgs + pow(r / U_CM, m.k_m)) + m.rho_ism_cgs) * (U_G / U_CM3);
    return m.A / (m.r02 + r * r) + m.rho_ism;
}
// simpson_logspace of rho r^3 d(ln r): enclosed_mass, shock-physics.h:401-425
VAG_DEV double enclosed_mass_generic(const Medium& med, double r) {
    const int N = 32;
    const double u_max = log(r), u_min = u_max - 18, h = (u_max - u_min) / N;
    auto f = [&](double u) {
        const double ri = exp(u);
        return medium_rho(med, ri) * ri * ri * ri;
    };
    double sum = f(u_min) + f(u_max);
    for (int i = 1; i < N; i += 2) sum += 4 * f(u_min + i * h);
    for (int i = 2; i < N; i += 2) sum += 2 * f(u_min + i * h);
    return sum * h / 3;
}
VAG_DEV double medium_mass(const Medium& m, double r) {  // enclosed_mass_medium, shock-physics.h:439-446
    if (m.type != VAG_MEDIUM_ISM && m.generic) return enclosed_mass_generic(m, r);
    double mass = m.rho_ism * r * r * r / 3.0;
    if (m.type != VAG_MEDIUM_ISM && m.A != 0) {
        if (m.r02 > 0) {
            const double a = sqrt(m.r02);
            mass += m.A * (r - a * atan(r / a));
        } else {
            mass += m.A * r;
        }
    }
    return mass;
}

// Parameter validation on the device: same rules as vag_params_validate
// (pybind/pymodel.cpp:47-186, pybind/pymodel.h:205-260,613-649).
VAG_DEV bool params_valid(const vag_model_params& p) {
    auto fpos = [](double x) { return isfinite(x) && x > 0; };
    auto oi = [](double x, double lo, double hi) { return isfinite(x) && x > lo && x <= hi; };
    bool ok = p.jet_type >= 0 && p.jet_type <= VAG_JET_POWERLAW_WING && p.medium_type >= 0 &&
              p.medium_type <= VAG_MEDIUM_WIND;
    ok = ok && oi(p.theta_c, 0.0, C_PI / 2) && fpos(p.duration);
    if (p.jet_type != VAG_JET_POWERLAW_WING) ok = ok && fpos(p.E_iso) && isfinite(p.Gamma0) && p.Gamma0 > 1.0;
    if (p.jet_type == VAG_JET_STEP_POWERLAW || p.jet_type == VAG_JET_POWERLAW_WING)
        ok = ok && fpos(p.E_iso_w) && isfinite(p.Gamma0_w) && p.Gamma0_w > 1.0;
    if (p.jet_type == VAG_JET_POWERLAW || p.jet_type == VAG_JET_STEP_POWERLAW || p.jet_type == VAG_JET_POWERLAW_WING)
        ok = ok && fpos(p.k_e) && fpos(p.k_g);
    if (p.jet_type == VAG_JET_TWO_COMPONENT)
        ok = ok && oi(p.theta_w, 0.0, C_PI / 2) && p.theta_w > p.theta_c && fpos(p.E_iso_w) && isfinite(p.Gamma0_w) &&
             p.Gamma0_w > 1.0;
    ok = ok && isfinite(p.n_ism) && p.n_ism >= 0;
    if (p.medium_type == VAG_MEDIUM_WIND) ok = ok && fpos(p.A_star) && p.n0 > 0 && fpos(p.k_m);
    ok = ok && fpos(p.lumi_dist) && isfinite(p.z) && p.z >= 0 && isfinite(p.theta_obs) && p.theta_obs >= 0 &&
         p.theta_obs <= C_PI;
    ok = ok && oi(p.eps_e, 0.0, 1.0) && oi(p.eps_B, 0.0, 1.0) && oi(p.xi_e, 0.0, 1.0) && isfinite(p.p) && p.p > 1.0;
    ok = ok && isfinite(p.rtol) && p.rtol > 0 && p.rtol < 1 && fpos(p.phi_resol) && fpos(p.theta_resol) && fpos(p.t_resol);
    if (p.jet_type == VAG_JET_MAGNETIZED_TOPHAT) ok = ok && isfinite(p.sigma0) && p.sigma0 >= 0;
    if (p.flags & VAG_FLAG_MAGNETAR)
        ok = ok && fpos(p.mag_L0) && fpos(p.mag_t0) && fpos(p.mag_q) && p.jet_type != VAG_JET_POWERLAW_WING &&
             p.jet_type != VAG_JET_MAGNETIZED_TOPHAT;
    if (p.flags & VAG_FLAG_RVS)
        ok = ok && oi(p.rvs_eps_e, 0.0, 1.0) && oi(p.rvs_eps_B, 0.0, 1.0) && oi(p.rvs_xi_e, 0.0, 1.0) && isfinite(p.rvs_p) &&
             p.rvs_p > 1.0;
    return ok;
}

// ---- DOPRI5(4) with boost::odeint's controller and dense output, register resident ----
// external/boost/numeric/odeint/stepper/runge_kutta_dopri5.hpp:88-258,
// controlled_runge_kutta.hpp:56-156,752-782, dense_output_runge_kutta.hpp:324-361.
template <int N>
struct Dopri5 {
    double x[N], dx[N];          // current state / derivative (FSAL)
    double xo[N], dxo[N];        // previous state / derivative
    double k3[N], k4[N], k5[N], k6[N];
    double t, t_old, dt, eps;
#ifdef VAG_PAIR_STAMPS
    int n_att = 0;  // developer aid: step attempts made (accepted + rejected)
#endif

    template <class F>
    VAG_DEV void init(const double* x0, double t0, double dt0, double tol, F& f) {
#pragma unroll
        for (int i = 0; i < N; ++i) x[i] = x0[i];
        t = t0;
        dt = dt0;
        eps = tol;
        f(x, dx, t);
    }

    // one accepted step; returns false after 500 consecutive rejections
    template <class F>
    VAG_DEV bool step(F& f) {
        constexpr double a2 = 1.0 / 5, a3 = 3.0 / 10, a4 = 4.0 / 5, a5 = 8.0 / 9;
        constexpr double b21 = 1.0 / 5, b31 = 3.0 / 40, b32 = 9.0 / 40;
        constexpr double b41 = 44.0 / 45, b42 = -56.0 / 15, b43 = 32.0 / 9;
        constexpr double b51 = 19372.0 / 6561, b52 = -25360.0 / 2187, b53 = 64448.0 / 6561, b54 = -212.0 / 729;
        constexpr double b61 = 9017.0 / 3168, b62 = -355.0 / 33, b63 = 46732.0 / 5247, b64 = 49.0 / 176,
                         b65 = -5103.0 / 18656;
        constexpr double c1 = 35.0 / 384, c3 = 500.0 / 1113, c4 = 125.0 / 192, c5 = -2187.0 / 6784, c6 = 11.0 / 84;
        constexpr double dc1 = c1 - 5179.0 / 57600, dc3 = c3 - 7571.0 / 16695, dc4 = c4 - 393.0 / 640,
                         dc5 = c5 - -92097.0 / 339200, dc6 = c6 - 187.0 / 2100, dc7 = -1.0 / 40;
        t_old = t;
        for (int fails = 0; fails < 500; ++fails) {
            double xt[N], k2[N], xn[N], k7[N];
            const double h = dt;
#ifdef VAG_PAIR_STAMPS
            ++n_att;
#endif
#pragma unroll
            for (int i = 0; i < N; ++i) xt[i] = x[i] + (h * b21) * dx[i];
            f(xt, k2, t + h * a2);
#pragma unroll
            for (int i = 0; i < N; ++i) xt[i] = x[i] + (h * b31) * dx[i] + (h * b32) * k2[i];
            f(xt, k3, t + h * a3);
#pragma unroll
            for (int i = 0; i < N; ++i) xt[i] = x[i] + (h * b41) * dx[i] + (h * b42) * k2[i] + (h * b43) * k3[i];
            f(xt, k4, t + h * a4);
#pragma unroll
            for (int i = 0; i < N; ++i)
                xt[i] = x[i] + (h * b51) * dx[i] + (h * b52) * k2[i] + (h * b53) * k3[i] + (h * b54) * k4[i];
            f(xt, k5, t + h * a5);
#pragma unroll
            for (int i = 0; i < N; ++i)
                xt[i] = x[i] + (h * b61) * dx[i] + (h * b62) * k2[i] + (h * b63) * k3[i] + (h * b64) * k4[i] +
                        (h * b65) * k5[i];
            f(xt, k6, t + h);
#pragma unroll
            for (int i = 0; i < N; ++i)
                xn[i] = x[i] + (h * c1) * dx[i] + (h * c3) * k3[i] + (h * c4) * k4[i] + (h * c5) * k5[i] + (h * c6) * k6[i];
            f(xn, k7, t + h);
            double err = 0;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const double xe = (h * dc1) * dx[i] + (h * dc3) * k3[i] + (h * dc4) * k4[i] + (h * dc5) * k5[i] +
                                  (h * dc6) * k6[i] + (h * dc7) * k7[i];
                err = dmax(err, fabs(xe) * rcp_fast(eps + eps * (fabs(x[i]) + fabs(h) * fabs(dx[i]))));
            }
            if (err > 1.0) {
                dt = h * dmax(9.0 / 10.0 * exp2_sat(log2_fast(err) * (-1.0 / 3)), 1.0 / 5.0);
                continue;
            }
#pragma unroll
            for (int i = 0; i < N; ++i) {
                xo[i] = x[i];
                dxo[i] = dx[i];
                x[i] = xn[i];
                dx[i] = k7[i];
            }
            t = t + h;
            if (err < 0.5) {
                err = dmax(3.2e-4, err);  // 5^-5
                dt = h * (9.0 / 10.0 * exp2_fast(log2_fast(err) * (-1.0 / 5)));
            }
            return true;
        }
        return false;
    }

    // dense output at tq in (t_old, t]
    VAG_DEV void interp(double tq, double* out) const {
        constexpr double b1 = 35.0 / 384, b3 = 500.0 / 1113, b4 = 125.0 / 192, b5 = -2187.0 / 6784, b6 = 11.0 / 84;
        const double h = t - t_old;
        // boost's dense-output polynomials; the constant denominators are folded into reciprocals (<= 1 ulp apart from the
        // divisions, and seven IEEE division sequences shorter per saved node)
        const double th = (tq - t_old) * rcp_fast(h);
        const double X1 = 5.0 * (2558722523.0 - 31403016.0 * th) * (1.0 / 11282082432.0);
        const double X3 = 100.0 * (882725551.0 - 15701508.0 * th) * (1.0 / 32700410799.0);
        const double X4 = 25.0 * (443332067.0 - 31403016.0 * th) * (1.0 / 1880347072.0);
        const double X5 = 32805.0 * (23143187.0 - 3489224.0 * th) * (1.0 / 199316789632.0);
        const double X6 = 55.0 * (29972135.0 - 7076736.0 * th) * (1.0 / 822651844.0);
        const double X7 = 10.0 * (7414447.0 - 829305.0 * th) * (1.0 / 29380423.0);
        const double thm1 = th - 1.0, th2 = th * th;
        const double A = th2 * (3.0 - 2.0 * th);
        const double B = th2 * thm1;
        const double C = th2 * thm1 * thm1;
        const double D = th * thm1 * thm1;
        const double w1 = h * (A * b1 - C * X1 + D), w3 = h * (A * b3 + C * X3), w4 = h * (A * b4 - C * X4),
                     w5 = h * (A * b5 + C * X5), w6 = h * (A * b6 - C * X6), w7 = h * (B + C * X7);
#pragma unroll
        for (int i = 0; i < N; ++i)
            out[i] = xo[i] + w1 * dxo[i] + w3 * k3[i] + w4 * k4[i] + w5 * k5[i] + w6 * k6[i] + w7 * dx[i];
    }
};

// DOPRI5 for a FLAT attempt loop (r05; the form vag_dyn_fast.h gave the forward shock in r02): attempt() makes ONE step attempt from
// (x, dx, t) with the step dt and leaves the candidate (xn, k3 .. k7, h) beside the untouched state; the caller accepts or rejects it
// per lane and commit()s.  Dopri5::step (vag_device.h) repeats its attempt inside the call until it is accepted, so a wavefront of 64
// rows repeats while ANY row rejects -- with ~7 % of a row's attempts rejected nearly every trip of the old loop paid for two.  Same
// arithmetic per attempt, same controller (controlled_runge_kutta.hpp:752-782), same dense output (runge_kutta_dopri5.hpp:238-258).
// The stage derivatives k2 .. k6 of the last attempt live where KStore puts them: in registers (KRegs) or in LDS (KLds: 5 x 11 doubles per
// lane that the 256 + 254 registers of the retry-loop form shuffled between VGPRs and AGPRs -- two moves per double each way -- are
// one ds_write / ds_read each, requested in batches a stage ahead of their use).
template <int N>
struct KRegs {
    double k[5][N];
    VAG_DEV void put(int s, int i, double v) { k[s][i] = v; }
    VAG_DEV double get(int s, int i) const { return k[s][i]; }
};
template <int N>
struct KLds {  // [5][N][64] doubles of the wavefront's LDS, this lane's column
    volatile double* base;  // (volatile: the values must go through LDS, not be forwarded in registers)
    VAG_DEV void put(int s, int i, double v) { base[(s * N + i) * 64] = v; }
    VAG_DEV double get(int s, int i) const { return base[(s * N + i) * 64]; }
};
template <int N, class KStore = KRegs<N>>
struct Dopri5Flat {
    double x[N], dx[N];  // state / derivative at t (FSAL)
    double xn[N], k7[N];  // the last attempt: candidate state at t + h and its derivative
    KStore ks;            // k2 .. k6 of the last attempt (index 0 .. 4)
    double t, dt, h, eps;

    template <class F>
    VAG_DEV void init(const double* x0, double t0, double dt0, double tol, F& f) {
#pragma unroll
        for (int i = 0; i < N; ++i) x[i] = x0[i];
        t = t0;
        dt = dt0;
        eps = tol;
        f(x, dx, t);
    }
    // one attempt; true = accepted (dt is then the controller's proposal for the NEXT step), false = rejected (dt is the retry's step)
    template <class F>
    VAG_DEV bool attempt(F& f) {
        constexpr double a2 = 1.0 / 5, a3 = 3.0 / 10, a4 = 4.0 / 5, a5 = 8.0 / 9;
        constexpr double b21 = 1.0 / 5, b31 = 3.0 / 40, b32 = 9.0 / 40;
        constexpr double b41 = 44.0 / 45, b42 = -56.0 / 15, b43 = 32.0 / 9;
        constexpr double b51 = 19372.0 / 6561, b52 = -25360.0 / 2187, b53 = 64448.0 / 6561, b54 = -212.0 / 729;
        constexpr double b61 = 9017.0 / 3168, b62 = -355.0 / 33, b63 = 46732.0 / 5247, b64 = 49.0 / 176, b65 = -5103.0 / 18656;
        constexpr double c1 = 35.0 / 384, c3 = 500.0 / 1113, c4 = 125.0 / 192, c5 = -2187.0 / 6784, c6 = 11.0 / 84;
        constexpr double dc1 = c1 - 5179.0 / 57600, dc3 = c3 - 7571.0 / 16695, dc4 = c4 - 393.0 / 640,
                         dc5 = c5 - -92097.0 / 339200, dc6 = c6 - 187.0 / 2100, dc7 = -1.0 / 40;
        double xt[N], kr[N];
        h = dt;
#pragma unroll
        for (int i = 0; i < N; ++i) xt[i] = x[i] + (h * b21) * dx[i];
        f(xt, kr, t + h * a2);
#pragma unroll
        for (int i = 0; i < N; ++i) ks.put(0, i, kr[i]);
#pragma unroll
        for (int i = 0; i < N; ++i) xt[i] = x[i] + (h * b31) * dx[i] + (h * b32) * kr[i];
        f(xt, kr, t + h * a3);
#pragma unroll
        for (int i = 0; i < N; ++i) ks.put(1, i, kr[i]);
#pragma unroll
        for (int i = 0; i < N; ++i) xt[i] = x[i] + (h * b41) * dx[i] + (h * b42) * ks.get(0, i) + (h * b43) * kr[i];
        f(xt, kr, t + h * a4);
#pragma unroll
        for (int i = 0; i < N; ++i) ks.put(2, i, kr[i]);
#pragma unroll
        for (int i = 0; i < N; ++i) xt[i] = x[i] + (h * b51) * dx[i] + (h * b52) * ks.get(0, i) + (h * b53) * ks.get(1, i) + (h * b54) * kr[i];
        f(xt, kr, t + h * a5);
#pragma unroll
        for (int i = 0; i < N; ++i) ks.put(3, i, kr[i]);
#pragma unroll
        for (int i = 0; i < N; ++i)
            xt[i] = x[i] + (h * b61) * dx[i] + (h * b62) * ks.get(0, i) + (h * b63) * ks.get(1, i) + (h * b64) * ks.get(2, i) + (h * b65) * kr[i];
        f(xt, kr, t + h);
#pragma unroll
        for (int i = 0; i < N; ++i) ks.put(4, i, kr[i]);
        double k3[N], k4[N], k5[N];  // (read once for the candidate and the error estimate)
#pragma unroll
        for (int i = 0; i < N; ++i) k3[i] = ks.get(1, i), k4[i] = ks.get(2, i), k5[i] = ks.get(3, i);
        const double* k6 = kr;
        double xe[N];  // the error estimate up to its last term (the sum is formed left to right: same roundings as in one expression)
#pragma unroll
        for (int i = 0; i < N; ++i) {
            xn[i] = x[i] + (h * c1) * dx[i] + (h * c3) * k3[i] + (h * c4) * k4[i] + (h * c5) * k5[i] + (h * c6) * k6[i];
            xe[i] = (h * dc1) * dx[i] + (h * dc3) * k3[i] + (h * dc4) * k4[i] + (h * dc5) * k5[i] + (h * dc6) * k6[i];
        }
        f(xn, k7, t + h);
        double err = 0;
#pragma unroll
        for (int i = 0; i < N; ++i)
            err = dmax(err, fabs(xe[i] + (h * dc7) * k7[i]) * rcp_fast(eps + eps * (fabs(x[i]) + fabs(h) * fabs(dx[i]))));
        if (err > 1.0) {
            dt = h * dmax(9.0 / 10.0 * exp2_sat(log2_fast(err) * (-1.0 / 3)), 1.0 / 5.0);
            return false;
        }
        if (err < 0.5) {
            err = dmax(3.2e-4, err);  // 5^-5
            dt = h * (9.0 / 10.0 * exp2_fast(log2_fast(err) * (-1.0 / 5)));
        }
        return true;
    }
    VAG_DEV void commit() {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            x[i] = xn[i];
            dx[i] = k7[i];
        }
        t = t + h;
    }
    // dense output of the last attempt at tq in (t, t + h], BEFORE commit()
    VAG_DEV void interp(double tq, double* out) const {
        constexpr double b1 = 35.0 / 384, b3 = 500.0 / 1113, b4 = 125.0 / 192, b5 = -2187.0 / 6784, b6 = 11.0 / 84;
        const double hh = (t + h) - t;  // (the committed form measures t_new - t_old)
        const double th = (tq - t) * rcp_fast(hh);
        const double X1 = 5.0 * (2558722523.0 - 31403016.0 * th) * (1.0 / 11282082432.0);
        const double X3 = 100.0 * (882725551.0 - 15701508.0 * th) * (1.0 / 32700410799.0);
        const double X4 = 25.0 * (443332067.0 - 31403016.0 * th) * (1.0 / 1880347072.0);
        const double X5 = 32805.0 * (23143187.0 - 3489224.0 * th) * (1.0 / 199316789632.0);
        const double X6 = 55.0 * (29972135.0 - 7076736.0 * th) * (1.0 / 822651844.0);
        const double X7 = 10.0 * (7414447.0 - 829305.0 * th) * (1.0 / 29380423.0);
        const double thm1 = th - 1.0, th2 = th * th;
        const double A = th2 * (3.0 - 2.0 * th);
        const double B = th2 * thm1;
        const double C = th2 * thm1 * thm1;
        const double D = th * thm1 * thm1;
        const double w1 = hh * (A * b1 - C * X1 + D), w3 = hh * (A * b3 + C * X3), w4 = hh * (A * b4 - C * X4),
                     w5 = hh * (A * b5 + C * X5), w6 = hh * (A * b6 - C * X6), w7 = hh * (B + C * X7);
#pragma unroll
        for (int i = 0; i < N; ++i)
            out[i] = x[i] + w1 * dx[i] + w3 * ks.get(1, i) + w4 * ks.get(2, i) + w5 * ks.get(3, i) + w6 * ks.get(4, i) + w7 * k7[i];
    }
};


// ---- deceleration time estimate: src/core/grid-refinement.h:402-453 ----
VAG_DEV double estimate_t_dec(const Jet& jet, const Medium& med, double theta) {
    const double gamma = jet_Gamma0(jet, theta);
    const double beta = gamma_to_beta(gamma);
    const double m_jet = jet_eps_k(jet, theta) / (gamma * C_C2) / (1.0 + jet.sigma0);  // HasSigma<Ejecta>: x / 1.0 is exact
    const double target = m_jet / gamma;
    const double r_min = 1e-3;
    const double r_max = r_min * 1e40;
    const double kin = (1 - beta) / (beta * C_C);
    if (target <= 0) return r_min * kin;
    if (med.type == VAG_MEDIUM_ISM) {
        const double rho = med.rho_ism;
        if (rho > 0) {
            const double r3_dec = r_min * r_min * r_min + 3 * target / rho;
            const double r_dec = cbrt(dmax(r3_dec, 0.0));
            return dmin(r_dec, r_max) * kin;
        }
        return r_max * kin;
    }
    const int N = 256;
    const double u_min = log(1e-3);
    const double u_max = u_min + 40 * log(10.0);
    const double du = (u_max - u_min) / N;
    double mass = 0;
    double r_prev = exp(u_min);
    double f_prev = medium_rho(med, r_prev) * r_prev * r_prev;
    for (int i = 1; i <= N; ++i) {
        const double r_i = exp(u_min + i * du);
        const double f_i = medium_rho(med, r_i) * r_i * r_i;
        const double dr = r_i - r_prev;
        mass += 0.5 * (f_prev + f_i) * dr;
        if (mass >= target) {
            const double r_dec = r_prev + (target - (mass - 0.5 * (f_prev + f_i) * dr)) / f_i;
            return r_dec * kin;
        }
        f_prev = f_i;
        r_prev = r_i;
    }
    return exp(u_max) * kin;
}

// ---- per-row log time lattice with a 3x denser band around t_dec:
//      src/core/grid-refinement.h:533-581 (logspace_with_band_refinement, make_time_grid) ----
struct TimeLattice {
    int n, n1, n2, n3, plain;
    double l0, l1, l2, l3, step;
    double s1, s2, s3;  // per-segment log10 steps (the reference divides per node; one rounding apart)
    VAG_DEV void init(double ts, double t_end, double t_dec, int n_nodes) {
        n = n_nodes;
        double b_lo = dmax(t_dec / 3, ts);
        double b_hi = dmin(3 * t_dec, t_end);
        l0 = log10(ts);
        l3 = log10(t_end);
        plain = (!(b_hi > b_lo) || n < 8);
        s1 = s2 = s3 = 0;
        if (plain) {
            step = (l3 - l0) / fmax(1.0, (double)(n - 1));  // xt::linspace, xbuilder.hpp:460-471
            n1 = n2 = n3 = 0;
            l1 = l2 = 0;
            return;
        }
        l1 = log10(b_lo);
        l2 = log10(b_hi);
        const double w1 = l1 - l0, w2 = 3.0 * (l2 - l1), w3 = l3 - l2;
        const int segs = n - 1;
        n1 = (int)round((double)segs * w1 / (w1 + w2 + w3));
        n3 = (int)round((double)segs * w3 / (w1 + w2 + w3));
        if (segs - 2 < n1) n1 = segs - 2;
        if (segs - 1 - n1 - 1 < n3) n3 = segs - 1 - n1 - 1;
        n2 = segs - n1 - n3;
        step = 0;
        s1 = n1 > 0 ? (l1 - l0) / (double)n1 : 0;
        s2 = n2 > 0 ? (l2 - l1) / (double)n2 : 0;
        s3 = n3 > 0 ? (l3 - l2) / (double)n3 : 0;
    }
    // node kk in [0, n): 10^lg through the fast exp2 (3.3e-16), not the library pow
    VAG_DEV double node(int kk) const {
        double lg;
        if (plain) {
            lg = (n > 1 && kk == n - 1) ? l3 : l0 + step * (double)kk;
        } else if (kk < n1) {
            lg = l0 + s1 * (double)kk;
        } else if (kk < n1 + n2) {
            lg = l1 + s2 * (double)(kk - n1);
        } else {
            const int q = kk - n1 - n2;
            lg = (n3 > 0 && q < n3) ? l2 + s3 * (double)q : l3;
        }
        return exp2_fast(lg * 3.321928094887362347870319429489390175865);
    }
};

// ---- forward-shock blast wave: src/dynamics/forward-shock.tpp:10-173, shock-physics.h ----
template <bool SPREAD = false, bool INJECT = false>
struct FwdShock {
    Medium med;
    double m_jet0, gamma_m_coeff, gamma_c_coeff, eps_e_eff, p, eps_B;
    double theta_s, dOmega0;  // SPREAD: jet_spreading_edge (grid-refinement.h:113-135), 1 - cos(theta0)
    double inj_L, inj_t0, inj_q;  // INJECT: magnetar luminosity per solid angle in code units (0 outside theta_c), t0 [code], q
    LdsTab lg_tab;                // log2_tab's table, staged in LDS by the dynamics kernel
    static constexpr int IDX_EPS = 5 + (SPREAD ? 1 : 0);  // eps_jet follows theta in the state vector

    // state [Gamma, m2, U2_th, r, t_comv (, theta)]; theta is constant for non-spreading jets and its zero derivative
    // never contributes to the error norm, so it is only integrated when the jet spreads (forward-shock.tpp:36-40).
    VAG_DEV void operator()(const double* s, double* d, double t) const {
        const double Gamma = s[0], m2 = s[1], U = s[2], r = s[3], t_comv = s[4];
        double deps_jet = 0;
        if constexpr (INJECT) {  // ForwardState::eps_jet (forward-shock.hpp:36-44): d eps_jet / dt = deps_dt(t)
            deps_jet = inj_L * exp2_sat(-inj_q * log2_fast(1 + t * inj_t0));
            d[IDX_EPS] = deps_jet;
        }
        const double u2 = (Gamma - 1) * (Gamma + 1);
        const double u = sqrt_fast(u2);
        const double dr = u * (Gamma + u) * C_C;
        d[3] = dr;
        d[4] = Gamma + u;
        const double inv_G = rcp_ode(Gamma);
        double dth = 0, sin_th = 0, cos_th = 1;
        if constexpr (SPREAD) {
            const double theta = s[5];
            if (theta < 0.5 * C_PI)  // compute_dtheta_dt, shock-physics.h:141-145
                dth = dr * rcp_ode(2 * Gamma * r) * sqrt_fast((2 * u2 + 3) * rcp_ode(4 * u2 + 3)) * rcp_ode(1 + u * theta_s * 7);
            d[5] = dth;
            sin_th = sin(theta);
            cos_th = cos(theta);
        }
        const double rho = medium_rho(med, r);
        const double dm = r * r * rho * dr;
        d[1] = dm;
        const double e_th = (Gamma - 1) * 4 * Gamma * rho * C_C2;
        double eps_rad = 0;  // RadiativeEfficiency, shock-physics.h:247-288
        if (eps_e_eff != 0) {
            const double gamma_m = gamma_m_coeff * (Gamma - 1) + 1;
            const double gamma_bar = gamma_c_coeff * rcp_ode(e_th * t_comv);
            const double gamma_c = 0.5 * (gamma_bar + sqrt_fast(gamma_bar * gamma_bar + 4));
            const double ratio = gamma_m * rcp_ode(gamma_c);
            eps_rad = (ratio < 1 && p > 2) ? eps_e_eff * exp2_sat((p - 2) * log2_tab(ratio, lg_tab)) : eps_e_eff;
        }
        const double ad = 4.0 / 3.0 + inv_G / 3;  // adiabatic_idx
        const double Gamma2 = Gamma * Gamma;
        const double Gamma_eff = (ad * (Gamma2 - 1) + 1) * inv_G;
        const double dGamma_eff = (ad * (Gamma2 + 1) - 1) * (inv_G * inv_G);
        const double inv_r = rcp_ode(r);
        double dlnV = 3 * inv_r * dr;
        double dm_swept = dm, m_swept = m2, Ueff = U;
        if constexpr (SPREAD) {  // compute_dGamma_dt, forward-shock.tpp:77-84
            const double inv_dO = rcp_ode(dOmega0);
            const double f_spread = (1 - cos_th) * inv_dO;
            dm_swept = dm * f_spread + m2 * inv_dO * sin_th * dth;
            m_swept = m2 * f_spread;
            dlnV += sin_th * rcp_ode(1 - cos_th) * dth;
            Ueff = U * f_spread;
        }
        const double a1 = -(Gamma - 1) * (Gamma_eff + 1) * C_C2 * dm_swept + deps_jet;  // energy_inject, forward-shock.tpp:89-91
        const double a2 = (ad - 1) * Gamma_eff * Ueff * dlnV;
        const double b1 = (m_jet0 + m_swept) * C_C2;
        const double b2 = (dGamma_eff + Gamma_eff * (ad - 1) * inv_G) * Ueff;
        const double dG = (a1 + a2) * rcp_ode(b1 + b2);
        d[0] = dG;
        double dlnV2 = 3 * inv_r * dr - dG * inv_G;
        double dm_u = dm;
        if constexpr (SPREAD) {  // compute_dU_dt, forward-shock.tpp:109-115
            const double factor = sin_th * rcp_ode(1 - cos_th) * dth;
            dm_u = dm + m2 * factor;
            dlnV2 += factor;
            dlnV2 += factor * rcp_ode(ad - 1);
        }
        d[2] = (1 - eps_rad) * (Gamma - 1) * C_C2 * dm_u - (ad - 1) * dlnV2 * U;
    }
};

// enclosed_thermal_energy_medium, shock-physics.h:401-468
VAG_DEV double enclosed_thermal_energy(const Medium& med, double r, double Gamma, double ad, double eps_e) {
    const double cooling_exp = 3 * (ad - 1);
    if (med.type == VAG_MEDIUM_ISM) {
        const double pow_exp = 3 + cooling_exp;
        const double x0 = exp(-18.0);
        const double attenuation = 1 - pow(x0, pow_exp);
        const double integral = med.rho_ism * r * r * r * attenuation / pow_exp;
        return (1 - eps_e) * (Gamma - 1) * C_C2 * integral;
    }
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

// compute_compression(1, Gamma, 0): shock-physics.h:58-66,190-203,349-352; shock.cpp:93-100
VAG_DEV double compression_fwd(double Gd) {
    const double dd = 1 - Gd;
    const double denom = Gd - 1;
    const double g = denom <= 0 ? 1 : 1 + dd * dd * rcp_fast(denom);
    const double ad = 4.0 / 3.0 + rcp_fast(3 * g);
    const double gm1 = g - 1, adm2 = ad - 2, adm1 = ad - 1;
    const double u_down = sqrt_fast(dmax(gm1 * adm1 * adm1 * rcp_fast(-ad * adm2 * gm1 + 2), 0.0));
    const double u_up = sqrt_fast((1 + u_down * u_down) * dmax((g - 1) * (g + 1), 0.0)) + u_down * g;
    return (u_down == 0.) ? 4 * g : u_up * rcp_fast(u_down);
}

// ---- synchrotron electrons + photons for one cell: src/radiation/synchrotron.cpp:45-254,315-408,
//      smooth-power-law-syn.cpp:49-153.  Writes the VAG_NPAR block. ----
VAG_DEV double syn_freq(double gamma, double B) {
    if (B == 0 || !isfinite(gamma)) return 0;
    return 3 * C_E / (4 * C_PI * C_ME * C_C) * B * gamma * gamma;
}
VAG_DEV double syn_I_peak(double B, double column_den) {
    const double P = B * ((C_PI / 4) * 0.92 * SQRT3 * C_E3 / (C_ME * C_C2));
    return P * column_den / (4 * C_PI);
}
VAG_DEV double sigmoid2(double x) { return 1.0 / (1.0 + exp2(-x)); }
VAG_DEV double blend(double w, double a, double b) { return w * a + (1.0 - w) * b; }

struct CellOut {
    double par[VAG_NPAR];
    // for Model.details-style inspection
    double gamma_m, gamma_c, gamma_a, gamma_M, N_e, column_den, nu_m, nu_c, nu_a, nu_M, I_nu_max;
};

// generate_syn_photons for one cell (synchrotron.cpp:376-408) + SmoothPowerLawSyn::build (smooth-power-law-syn.cpp:94-153)
VAG_DEV void syn_photons_build(CellOut& o, double gamma_m, double gamma_c, double gamma_a, double gamma_M, double column_den,
                               double N_e, double B, double p, double Gamma, double r, double t_eng) {
    const double nu_M = syn_freq(gamma_M, B), nu_m = syn_freq(gamma_m, B), nu_c = syn_freq(gamma_c, B),
                 nu_a = syn_freq(gamma_a, B);
    const double I_nu_max = syn_I_peak(B, column_den);
    const double l_I = log2(I_nu_max), l_m = log2(nu_m), l_c = log2(nu_c), l_a = log2(nu_a), l_M = log2(nu_M);
    const double s_swap = 4.0, s_floor = 0.1;
    const double w_slow = sigmoid2(s_swap * (l_c - l_m));
    const double soft_offset = log2_softplus(-s_swap * fabs(l_c - l_m)) / s_swap;
    const double l_lo = dmin(l_m, l_c) - soft_offset;
    const double l_hi = dmax(l_m, l_c) + soft_offset;
    const double s_m_slow = dmax(1.84 - 0.40 * p, s_floor);
    const double s_c_slow = dmax(1.15 - 0.06 * p, s_floor);
    const double s_c_fast = 0.597;
    const double s_m_fast = dmax(3.34 - 0.82 * p, s_floor);
    const double smooth_lo = blend(w_slow, s_m_slow, s_c_fast);
    const double smooth_hi = blend(w_slow, s_c_slow, s_m_fast);
    const double alpha_mid = blend(w_slow, -0.5 * (p - 1.0), -0.5);
    const double diff_lo = smooth_lo * (1.0 / 3.0 - alpha_mid);
    const double diff_hi = smooth_hi * (alpha_mid + 0.5 * p);
    const double uu = sigmoid2(s_swap * (l_a - l_m));
    const double vv = sigmoid2(s_swap * (l_a - l_c));
    const double w_below = (1.0 - uu) * (1.0 - vv);
    const double w_above = uu * vv;
    const double s_a_mid = dmax(1.47 - 0.21 * p, s_floor);
    const double s_a_above = dmax(0.94 - 0.14 * p, s_floor);
    const double s_a_blend = w_below * 1.64 + w_above * s_a_above + (1.0 - w_below - w_above) * s_a_mid;
    // sharp thin/thick forms at nu_a (smooth-power-law-syn.cpp:49-78)
    double thin_a, thick_a;
    if (l_m < l_c) {
        if (l_a < l_m)
            thin_a = (l_a - l_m) / 3.0;
        else if (l_a < l_c)
            thin_a = 0.5 * (1.0 - p) * (l_a - l_m);
        else
            thin_a = 0.5 * (1.0 - p) * (l_c - l_m) - 0.5 * p * (l_a - l_c);
    } else {
        if (l_a < l_c)
            thin_a = (l_a - l_c) / 3.0;
        else if (l_a < l_m)
            thin_a = -0.5 * (l_a - l_c);
        else
            thin_a = -0.5 * (l_m - l_c) - 0.5 * p * (l_a - l_m);
    }
    thick_a = (l_a < l_m) ? 2. * (l_a - l_m) : 2.5 * (l_a - l_m);

    o.par[VP_LG2_I] = l_I;
    o.par[VP_LG2_NUM] = l_m;
    o.par[VP_LG2_NUMAX] = l_M;
    o.par[VP_INV_NUMAX] = LOG2E * (1.0 / nu_M);
    o.par[VP_TNORM] = thin_a - thick_a;
    o.par[VP_SAB] = s_a_blend;
    o.par[VP_INV_SAB] = 1.0 / s_a_blend;
    o.par[VP_LG2_LO] = l_lo;
    o.par[VP_LG2_HI] = l_hi;
    o.par[VP_DLO] = diff_lo;
    o.par[VP_DHI] = diff_hi;
    o.par[VP_INV_SLO] = 1.0 / smooth_lo;
    o.par[VP_INV_SHI] = 1.0 / smooth_hi;
    o.par[VP_GAMMA] = Gamma;
    o.par[VP_U] = sqrt((Gamma - 1) * (Gamma + 1));
    o.par[VP_R] = r;
    o.par[VP_LG2_R2] = 2.0 * log2(r);
    o.par[VP_TENG] = t_eng;
    o.gamma_m = gamma_m;
    o.gamma_c = gamma_c;
    o.gamma_a = gamma_a;
    o.gamma_M = gamma_M;
    o.N_e = N_e;
    o.column_den = column_den;
    o.nu_m = nu_m;
    o.nu_c = nu_c;
    o.nu_a = nu_a;
    o.nu_M = nu_M;
    o.I_nu_max = I_nu_max;
}


// gamma_M, gamma_m and the synchrotron-only gamma_c of one cell (synchrotron.cpp:45-110,334-347)
struct ElecBasic {
    double gamma_M, gamma_m, gamma_c;
};
VAG_DEV ElecBasic syn_elec_basic(double t_comv, double Gamma_th, double B, double eps_e, double p, double xi_e) {
    ElecBasic e;
    const double gamma_M = (B == 0) ? INFINITY : sqrt(6 * C_PI * C_E / C_SIGMAT / (B * (1 + 0.)));
    const double gamma_ave_m1 = eps_e * (Gamma_th - 1) * (C_MP / C_ME) / xi_e;
    double gm_m1;
    if (p > 2) {
        gm_m1 = (p - 2) / (p - 1) * gamma_ave_m1;
    } else if (p < 2) {
        gm_m1 = pow((2 - p) / (p - 1) * gamma_ave_m1 * pow(gamma_M, p - 2), 1 / (p - 1));
    } else {  // root_bisect, utilities.h:230-241
        auto eq = [&](double x) { return x * log(gamma_M) - (x + 1) * log(x) - gamma_ave_m1 - log(gamma_M); };
        double low = 0, high = gamma_M;
        for (int it = 0; it < 1000 && (high - low) > fabs((high + low) * 0.5) * 1e-6; ++it) {
            const double mid = 0.5 * (high + low);
            if (eq(mid) * eq(high) > 0)
                high = mid;
            else
                low = mid;
        }
        gm_m1 = 0.5 * (high + low);
    }
    const double gamma_bar = (6 * C_PI * C_ME * C_C / C_SIGMAT) / (B * B * (1 + 0.) * t_comv) * 1;
    e.gamma_M = gamma_M;
    e.gamma_m = gm_m1 + 1;
    e.gamma_c = (gamma_bar + sqrt(gamma_bar * gamma_bar + 4)) / 2;
    return e;
}

// cool_after_crossing, synchrotron.cpp:190-195
VAG_DEV double cool_after_crossing(double gamma_x, double gamma_m_x, double gamma_m) {
    return (gamma_x - 1) * ((gamma_m - 1) / (gamma_m_x - 1)) + 1;
}

// `inj` = electrons frozen at the crossing cell for a relic cell of a reverse shock (cool_relic_electrons,
// synchrotron.h:187-201), used when `relic` (by value: a pointer would put the caller's copy in scratch memory).
VAG_DEV void syn_cell(CellOut& o, double t_eng, double t_comv, double r, double Gamma, double Gamma_th, double B,
                      double N_p, double eps_e, double p, double xi_e, bool relic = false, ElecBasic inj = ElecBasic{0, 1, 0}) {
    // --- electrons (synchrotron.cpp:315-360) ---
    const ElecBasic eb = syn_elec_basic(t_comv, Gamma_th, B, eps_e, p, xi_e);
    double gamma_M = eb.gamma_M;
    const double gamma_m = eb.gamma_m;
    double f_syn = (gamma_m - 1) / gamma_m;
    if (p > 3) f_syn = fast_pow(f_syn, (p - 1) / 2);
    const double N_e = N_p * xi_e * f_syn;
    const double column_den = N_e / (r * r);
    const double I_peak = syn_I_peak(B, column_den);
    double gamma_c = eb.gamma_c;
    if (relic) {
        gamma_c = cool_after_crossing(inj.gamma_c, inj.gamma_m, gamma_m);
        gamma_M = cool_after_crossing(inj.gamma_M, inj.gamma_m, gamma_m);
    }
    // compute_syn_gamma_a with no IC (ratio exactly 1), synchrotron.cpp:212-246
    double gamma_a;
    {
        const double gamma_peak = dmin(gamma_m, gamma_c);
        const double nu_peak = syn_freq(gamma_peak, B);
        const double kT = (gamma_peak - 1) * (C_ME * C_C2) / 3;
        double nu_a = fast_pow(I_peak * C_C2 / (cbrt(nu_peak) * 2 * kT), 0.6);
        if (nu_a > nu_peak) {
            if (gamma_c > gamma_m) {
                const double nu_m = syn_freq(gamma_m, B);
                nu_a = fast_pow(I_peak * C_C2 / (2 * kT) * fast_pow(nu_m, p / 2), 2 / (p + 4));
                const double nu_c = syn_freq(gamma_c, B);
                if (nu_a > nu_c)
                    nu_a = fast_pow(I_peak * C_C2 / (2 * kT) * sqrt(nu_c) * fast_pow(nu_m, p / 2), 2 / (p + 5));
            } else {
                const double nu_c = syn_freq(gamma_c, B);
                nu_a = fast_pow(I_peak * C_C2 / (2 * kT) * sqrt(nu_c), 0.4);
                const double nu_m = syn_freq(gamma_m, B);
                if (nu_a > nu_m)
                    nu_a = fast_pow(I_peak * C_C2 / (2 * kT) * sqrt(nu_c) * fast_pow(nu_m, p / 2), 2 / (p + 5));
            }
        }
        gamma_a = sqrt((4 * C_PI * C_ME * C_C / (3 * C_E)) * (nu_a / B)) + 1;
    }
    syn_photons_build(o, gamma_m, gamma_c, gamma_a, gamma_M, column_den, N_e, B, p, Gamma, r, t_eng);
}

// Per-model constants of the optically thick branch (smooth-power-law-syn.cpp:102-107)
struct SpecConst {
    double smooth_thick, log2_x_far;
    VAG_DEV void init(double p) {
        smooth_thick = (3.44 * p - 1.41) / LN2;
        log2_x_far = 1.5 * log2(20.0 / smooth_thick);
    }
    // the same through the 64-entry log2 table of a flux workgroup: the value only places the shortcut beyond which the
    // optically thick softplus term is dropped, so the last bits of it do not reach the spectrum
    template <class Tab>
    VAG_DEV void init_fast(double p, Tab lg) {
        smooth_thick = (3.44 * p - 1.41) / LN2;
        log2_x_far = 1.5 * (4.321928094887363 - log2_tab(smooth_thick, lg));
    }
};

// SmoothPowerLawSyn::compute_log2_I_nu without IC (smooth-power-law-syn.cpp:15-46,80-92,159-167).
// `c` points at the cell's parameter column with stride `st` between parameters.
template <class PtrT>
VAG_DEV double log2_I_nu(const PtrT c, int st, const SpecConst& sc, double lg2_nu) {
    const double l_lo = c[VP_LG2_LO * st], l_hi = c[VP_LG2_HI * st];
    const double thin = (lg2_nu - l_lo) / 3.0 - log2_softplus(c[VP_DLO * st] * (lg2_nu - l_lo)) * c[VP_INV_SLO * st] -
                        log2_softplus(c[VP_DHI * st] * (lg2_nu - l_hi)) * c[VP_INV_SHI * st];
    const double lx = lg2_nu - c[VP_LG2_NUM * st];
    double thick = 2.5 * lx;
    if (!(lx > sc.log2_x_far)) {
        const double s = -sc.smooth_thick * exp2(2. / 3 * lx);
        thick += log2_softplus(-0.5 * lx + s);
    }
    const double lb = thick + c[VP_TNORM * st];
    const double smooth_one = thin - log2_softplus(c[VP_SAB * st] * (thin - lb)) * c[VP_INV_SAB * st];
    const double spec = c[VP_LG2_I * st] + (c[VP_INV_SLO * st] + smooth_one);
    if (lg2_nu - c[VP_LG2_NUMAX * st] < -20) return spec;
    return spec - c[VP_INV_NUMAX * st] * exp2(lg2_nu);
}

// ---- fast FP64 kernels for the hot evaluator (accuracy verified at context creation / in tests) ----
// On CDNA4 every wave64 VALU instruction -- FP64 FMA, 32-bit integer add, register move alike -- occupies its SIMD for
// four cycles, and 32-bit integer multiplies / 64-bit multiply-adds take four times that.  The kernels below are
// therefore written for instruction COUNT: Horner chains with the coefficients in SGPRs (no accumulator copies),
// magic-number rounding instead of float<->int conversions, 24-bit multiplies and 32-bit LDS addresses.
//
// g(a) = log2(1 + 2^-a) on [0, 20]: 201 node-centred intervals of width 1/10 (interval i covers |a*10 - i| <= 1/2),
// degree-5 Chebyshev-node interpolants (max abs error < 5e-14 in log2 units); table built on the host in extended
// precision (vag_capi.hip: build_softplus_table).  Three 16-byte reads + 5 FMAs per call.
constexpr int SP_PER_UNIT = 10;
constexpr int SP_INTERVALS = 20 * SP_PER_UNIT + 1;
constexpr int SP_NCOEF = 6;
constexpr int SP_TABLE_DOUBLES = SP_INTERVALS * SP_NCOEF;
constexpr double SP_MAGIC = 6755399441055744.0;  // 1.5 * 2^52: adding it rounds to an integer held in the low word

VAG_DEV const vdouble2* sp_row(const double* tab, int idx) { return reinterpret_cast<const vdouble2*>(tab) + idx * (SP_NCOEF / 2); }
#ifndef VAG_HOST_DEBUG
VAG_DEV LdsTab sp_row(LdsTab tab, int idx) { return tab + __mul24(idx, SP_NCOEF / 2); }
#endif

// The 13 spectral members of one staged cell in registers: seven 16-byte LDS reads, shared by every frequency the
// work item evaluates.  Indexed with the VP_* names like the strided blocks in HBM.
struct SpecRegs {
    double v[14];
    VAG_DEV double operator[](int i) const { return v[i]; }
};
VAG_DEV SpecRegs load_spec_regs(LdsTab cell) {
    SpecRegs r;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const vdouble2 t = cell[i];
        r.v[2 * i] = t.x;
        r.v[2 * i + 1] = t.y;
    }
    return r;
}

// log2_softplus (src/util/fast-math.h:179-185) = max(z,0) + g(|z|) with the reference's +-20 shortcuts
// max(z, 0) as ONE v_max_f64 (the compiler's form quiets a possible NaN first: a second v_max_f64 per call)
VAG_DEV double relu_f64(double z) {
#ifdef VAG_HOST_DEBUG
    return z > 0 ? z : 0.0;
#else
    double r;
    asm("v_max_f64 %0, %1, 0" : "=v"(r) : "v"(z));
    return r;
#endif
}
template <class Tab>
VAG_DEV double sp_fast(double z, Tab tab) {
    const double a = fabs(z);
#ifndef VAG_SP_BRANCHLESS
    if (a > 20.0) return relu_f64(z);
#endif
    const double t = fma(a, (double)SP_PER_UNIT, SP_MAGIC);  // nearest node: integer in the low mantissa word
    const int idx = (int)min((unsigned)__double2loint(t), (unsigned)(SP_INTERVALS - 1));  // the clamp only acts on NaN
    const double tau = fma(a, (double)SP_PER_UNIT, -(t - SP_MAGIC));  // in [-1/2, 1/2]
    const auto c2 = sp_row(tab, idx);
    const vdouble2 c01 = c2[0], c23 = c2[1], c45 = c2[2];
    double p = fma(c45.y, tau, c45.x);
    p = fma(p, tau, c23.y);
    p = fma(p, tau, c23.x);
    p = fma(p, tau, c01.y);
    p = fma(p, tau, c01.x);
#ifdef VAG_SP_BRANCHLESS
    p = a > 20.0 ? 0.0 : p;  // straight-line code: independent softplus terms of one evaluation interleave
#endif
    return fma(0.5, z + a, p);  // max(z, 0) = (z + |z|) / 2, exact
}

// 2^x: round-to-nearest split + degree-11 minimax polynomial in f on [-0.5, 0.5] (max rel err 2.1e-16) + ldexp.  Large |x|
// saturate through v_ldexp_f64 (0 / inf) exactly like exp2.  Horner: 11 dependent FMAs, each `v_fma_f64 p, p, f, s[coef]` -- the
// Estrin form (of the earlier degree-12 Taylor polynomial, kept behind VAG_EXP2_ESTRIN) needed 15 + 5 accumulator copies.
#ifdef VAG_HOST_DEBUG
VAG_DEV double fma3(double a, double b, double c) { return fma(a, b, c); }
#else
// a * b + c as ONE instruction when c is a loop-invariant constant: the compiler's own choice is the two-address
// v_fmac_f64 plus a register copy of the constant per Horner step.
VAG_DEV double fma3(double a, double b, double c) {
    double r;
#ifdef VAG_FMA3_VGPR_CONST
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
#else
    // the constant as the instruction's one scalar operand: with a "v" constraint the compiler parks every coefficient in a VGPR
    // pair for the whole kernel (22 VGPRs of the flux kernel's 128 for exp2_fast alone)
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
#endif
    return r;
}
#endif
VAG_DEV double exp2_fast(double x) {  // finite x only: +-inf would give inf - inf in the range reduction (see exp2_sat)
    const double n = rint(x);
    const double f = x - n;
#ifdef VAG_EXP2_ESTRIN
    const double f2 = f * f, f4 = f2 * f2, f8 = f4 * f4;
    const double q01 = fma(0.6931471805599453, f, 1.0), q23 = fma3(0.05550410866482158, f, 0.24022650695910072);
    const double q45 = fma3(0.0013333558146428443, f, 0.009618129107628477);
    const double q67 = fma3(1.5252733804059841e-05, f, 0.0001540353039338161);
    const double q89 = fma3(1.01780860092397e-07, f, 1.321548679014431e-06);
    const double qab = fma3(4.4455382718708116e-10, f, 7.054911620801123e-09);
    const double q03 = fma(q23, f2, q01), q47 = fma(q67, f2, q45), q8b = fma(qab, f2, q89);
    const double q07 = fma(q47, f4, q03);
    const double q8c = fma(2.5678435993488206e-11, f4, q8b);
    const double p = fma(q8c, f8, q07);
#else
    // minimax q of degree 10 for (2^f - 1) / f on [-1/2, 1/2] (profiles/micro/exp2_minimax.py: Remez in 60-digit arithmetic;
    // the rounded Horner form is within 2.1e-16 of 2^f, the degree-12 Taylor form it replaces was at 3.3e-16 with one step more)
    double p = fma(4.4566755710138823e-10, f, 7.072586181346367e-09);
    p = fma3(p, f, 1.0178051727399186e-07);
    p = fma3(p, f, 1.321544258661287e-06);
    p = fma3(p, f, 1.5252733853282354e-05);
    p = fma3(p, f, 0.00015403530441738514);
    p = fma3(p, f, 0.0013333558146396002);
    p = fma3(p, f, 0.009618129107606887);
    p = fma3(p, f, 0.05550410866482166);
    p = fma3(p, f, 0.24022650695910097);
    p = fma3(p, f, 0.6931471805599453);
    p = fma(p, f, 1.0);
#endif
    return ldexp(p, (int)n);
}

// exp2_fast for arguments that may be +-inf (log2 of 0 / overflowed ratios in the ODE right-hand sides and the IC
// corrections): saturates to 0 / inf like exp2; NaN passes through.
VAG_DEV double exp2_sat(double x) { return exp2_fast(dmin(dmax(x, -1100.0), 1100.0)); }

// log2(x) for positive, finite, normal x (the EAT step only sees such values; anything else is routed to the
// library log2).  Exponent/mantissa split around sqrt(2), s = f/(2+f), degree-14 even polynomial in s with the
// classic fdlibm/musl log() coefficients (Lg1..Lg7, |err| < 2^-58 on this range); result within ~1 ulp.
VAG_DEV double log2_fast(double x) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(x);
    const int eb = (int)(bits >> 52);
    if (eb == 0 || eb >= 2047) return log2(x);  // zero, subnormal, negative, inf, nan
    // mantissa in [1, 2); fold (sqrt2, 2) down so that m is in [sqrt(1/2), sqrt(2)]
    double m = __longlong_as_double((long long)((bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL));
    int e = eb - 1023;
    if (m > 1.4142135623730951) {
        m *= 0.5;
        e += 1;
    }
    const double f = m - 1.0;
    const double den = 2.0 + f, rden = rcp_fast(den);  // den in [1.7, 2.42]
    double s = f * rden;
    s = fma(fma(-den, s, f), rden, s);  // one residual step: f / den to <= 1 ulp without the IEEE division sequence
    const double z = s * s, w = z * z;
    const double t1 = w * (3.999999999940941908e-01 + w * (2.222219843214978396e-01 + w * 1.531383769920937332e-01));
    const double t2 = z * (6.666666666666735130e-01 +
                           w * (2.857142874366239149e-01 + w * (1.818357216161805012e-01 + w * 1.479819860511658591e-01)));
    const double R = t1 + t2;
    const double hfsq = 0.5 * f * f;
    const double ln_m = f - (hfsq - s * (hfsq + R));  // log(1 + f)
    return fma(ln_m, LOG2E, (double)e);
}

// log2 for the EAT step of the flux kernels (two per lattice node and (theta, phi) row): 64-entry table
// {1/c_i rounded, -log2(that rounded value)} over the mantissa, r = m / c_i - 1 with |r| <= 2^-7, degree-7 series of
// ln(1 + r) (truncation 2e-18).  17 instructions instead of the 40 of log2_fast; the table (1 KB) sits behind the
// softplus table in LDS.  Zero, subnormal, negative, inf, NaN go to the library log2.
constexpr int SP_LDS_DOUBLES = SP_TABLE_DOUBLES + LOG_TAB_DOUBLES;  // what a flux workgroup keeps in LDS
// the table form without its fall-back branch (straight-line callers interleave several of these): `special` is set for an argument
// the table does not serve (zero, subnormal, negative, inf, NaN) -- the caller then takes the library log2 for that value
VAG_DEV double log2_tab_core(double x, LdsTab tab, bool& special) {
    const int hi = __double2hiint(x), lo = __double2loint(x);
    const int eb = hi >> 20;
    special = (unsigned)(eb - 1) >= 2046u;
    const double m = __hiloint2double((hi & 0x000fffff) | 0x3ff00000, lo);  // mantissa in [1, 2)
    const vdouble2 t = tab[(hi >> 14) & (LOG_TAB_N - 1)];
    const double r = fma(m, t.x, -1.0);
    double q = fma(1.0 / 7, r, -1.0 / 6);
    q = fma(q, r, 0.2);
    q = fma(q, r, -0.25);
    q = fma(q, r, 1.0 / 3);
    q = fma(q, r, -0.5);
    const double ln_m = fma(q * r, r, r);
    return fma(ln_m, LOG2E, (double)(eb - 1023) + t.y);
}
VAG_DEV double log2_tab(double x, LdsTab tab) {
    const int eb = __double2hiint(x) >> 20;
    if ((unsigned)(eb - 1) >= 2046u) return log2(x);
    bool special;
    return log2_tab_core(x, tab, special);
}

#ifndef VAG_HOST_DEBUG
// ---- cross-lane helpers on the DPP path (no LDS round trip): gfx9 row_shr / row_bcast / wave_shr controls ----
template <int CTRL, int ROW_MASK>
VAG_DEV double dpp_zero(double v) {  // value of the DPP source lane; 0 where the source is invalid or the row is masked off
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
VAG_DEV double wave_prefix_sum(double x) {  // inclusive sum over lanes 0..lane (Kogge-Stone in rows of 16, then row totals)
    x += dpp_zero<0x111, 0xf>(x);  // row_shr:1
    x += dpp_zero<0x112, 0xf>(x);  // row_shr:2
    x += dpp_zero<0x114, 0xf>(x);  // row_shr:4
    x += dpp_zero<0x118, 0xf>(x);  // row_shr:8
    x += dpp_zero<0x142, 0xa>(x);  // row_bcast:15 -> rows 1, 3 add the total of the row before
    x += dpp_zero<0x143, 0xc>(x);  // row_bcast:31 -> rows 2, 3 add the total of rows 0-1
    return x;
}
VAG_DEV double from_lane_below(double v) { return dpp_zero<0x138, 0xf>(v); }  // wave_shr:1: lane - 1's value, 0 into lane 0
// *p += v on an LDS word without a return value (the hardware applies the lanes of one instruction in lane order)
VAG_DEV void lds_add_f64(double* p, double v) {
    asm volatile("ds_add_f64 %0, %1" ::"v"((unsigned)(size_t)(__attribute__((address_space(3))) double*)p), "v"(v) : "memory");
}
#endif

// compute_log2_I_nu (smooth-power-law-syn.cpp:15-46,80-92,159-167) on the fast kernels above.
template <class PtrT, class Tab>
VAG_DEV double log2_I_nu_fast(const PtrT& c, int st, const SpecConst& sc, double lg2_nu, Tab sp) {
    const double l_lo = c[VP_LG2_LO * st], l_hi = c[VP_LG2_HI * st];
    const double thin = (lg2_nu - l_lo) * (1.0 / 3.0) - sp_fast(c[VP_DLO * st] * (lg2_nu - l_lo), sp) * c[VP_INV_SLO * st] -
                        sp_fast(c[VP_DHI * st] * (lg2_nu - l_hi), sp) * c[VP_INV_SHI * st];
    const double lx = lg2_nu - c[VP_LG2_NUM * st];
    double thick = 2.5 * lx;
    if (!(lx > sc.log2_x_far)) {
        const double s = -sc.smooth_thick * exp2_fast(2. / 3 * lx);
        thick += sp_fast(-0.5 * lx + s, sp);
    }
    const double lb = thick + c[VP_TNORM * st];
    const double smooth_one = thin - sp_fast(c[VP_SAB * st] * (thin - lb), sp) * c[VP_INV_SAB * st];
    const double spec = c[VP_LG2_I * st] + (c[VP_INV_SLO * st] + smooth_one);
    if (lg2_nu - c[VP_LG2_NUMAX * st] < -20) return spec;
    return spec - c[VP_INV_NUMAX * st] * exp2_fast(lg2_nu);
}

// The same evaluator for the TWO frequencies of a boundary work item as one straight-line block: no lane-divergent branches (the
// +-20 softplus shortcuts, the far-thick cut and the nu_M cut-off become selects on the same values), so the two evaluations and
// the independent softplus / exp2 chains inside each interleave.  Same arithmetic per taken path, hence the same bits as
// log2_I_nu_fast.
template <class Tab>
VAG_DEV double sp_fast_sel(double z, Tab tab) {
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
    return a > 20.0 ? (z > 0 ? z : 0.0) : r;
}
template <class PtrT, class Tab>
VAG_DEV void log2_I_nu_fast2(const PtrT& c, const SpecConst& sc, double xa, double xb, Tab sp, double& ba, double& bb) {
    const double l_lo = c[VP_LG2_LO], l_hi = c[VP_LG2_HI];
    double out[2];
    const double xs[2] = {xa, xb};
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const double lg2_nu = xs[e];
        const double thin = (lg2_nu - l_lo) * (1.0 / 3.0) - sp_fast_sel(c[VP_DLO] * (lg2_nu - l_lo), sp) * c[VP_INV_SLO] -
                            sp_fast_sel(c[VP_DHI] * (lg2_nu - l_hi), sp) * c[VP_INV_SHI];
        const double lx = lg2_nu - c[VP_LG2_NUM];
        const bool far = lx > sc.log2_x_far;
        const double s = -sc.smooth_thick * exp2_fast(2. / 3 * (far ? 0.0 : lx));
        const double th = 2.5 * lx;
        const double thick = far ? th : th + sp_fast_sel(-0.5 * lx + s, sp);
        const double lb = thick + c[VP_TNORM];
        const double smooth_one = thin - sp_fast_sel(c[VP_SAB] * (thin - lb), sp) * c[VP_INV_SAB];
        const double spec = c[VP_LG2_I] + (c[VP_INV_SLO] + smooth_one);
        const bool below = lg2_nu - c[VP_LG2_NUMAX] < -20;
        const double cut = spec - c[VP_INV_NUMAX] * exp2_fast(below ? 0.0 : lg2_nu);
        out[e] = below ? spec : cut;
    }
    ba = out[0], bb = out[1];
}

}  // namespace vag
