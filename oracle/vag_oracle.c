/*
 * vag_oracle.c -- TEST INFRASTRUCTURE: the CPU parity checker (see vag_oracle.h).
 *
 * A scalar, single-threaded C11 restatement of VegasAfterglow's light-curve path.
 * Every function cites the reference file:line it follows (paths relative to the
 * VegasAfterglow tree).  Written from the algorithm, with flat arrays instead of xtensor
 * containers; arithmetic order follows the reference: bit-identical to a strict-FP build of
 * the reference's own sources (oracle/_ref/libvag_ref_strict.so) on the configurations
 * tests/test_oracle.py pins, within ~2e-6 of its -O3 -ffp-contract=fast build.  Scope: every
 * closed-form jet and medium of the reference's registry, lateral spreading, magnetar
 * injection, Model(axisymmetric=False) (with a spreading jet: one lattice and one solve per
 * (phi, theta) node, coord_t::phi_size), forward and reverse shock, synchrotron with
 * self-absorption, SSC with Thomson / Klein-Nishina cooling, grid / series / band / exposure
 * requests, Model.details and the fitter's log-likelihood.
 */
#include "vag_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * Units and constants: src/util/macros.h:43-110 (expression order kept so the rounded
 * values are bit-identical), cutoffs/defaults: src/config/simulation-defaults.h:38-127
 * ---------------------------------------------------------------------------------------- */
#define U_LEN 1.5e13
#define U_CM (1 / U_LEN)
#define U_SEC (3e10 / U_LEN)
#define U_CM2 (U_CM * U_CM)
#define U_CM3 (U_CM * U_CM * U_CM)
#define U_G (1 / 2e33)
#define U_GAUSS (8.66e-11 / U_SEC)
#define U_HZ (1 / U_SEC)
#define U_ERG (U_G * U_CM * U_CM / U_SEC / U_SEC)
#define U_FLUX_CGS (U_ERG / U_CM2 / U_SEC)
#define U_FLUX_DEN_CGS (U_ERG / U_CM2 / U_SEC / U_HZ)

#define C_C 1.0
#define C_C2 (C_C * C_C)
#define C_MP (1.67e-24 * U_G)
#define C_ME (C_MP / 1836)
#define C_E (4.8e-10 / 4.472136e16 / 5.809475e19 / U_SEC)
#define C_E2 (C_E * C_E)
#define C_E3 (C_E2 * C_E)
#define C_PI 3.14159265358979323846
#define C_SIGMAT (6.65e-25 * U_CM * U_CM)
#define C_GAMMA_CUT (1.0 + 1e-6)
#define C_SIGMA_CUT 1e-6

#define DEF_MIN_THETA_POINTS 36
#define DEF_THETA_MIN 1e-6
#define DEF_ODE_RTOL 1e-6
#define DEF_BINARY_SEARCH_EPS 1e-9
#define DEF_MAX_ODE_STEPS 100000
#define DEF_THETA_SAMPLES 200

#define LOG2_10 (2.302585092994045684017991454684364208 / 0.693147180559945309417232121458176568)
#define M_LN2_ 0.693147180559945309417232121458176568
#define M_LOG2E_ 1.442695040888963407359924681001892137
#define M_SQRT3_ 1.732050807568877293527446341505872367

static _Thread_local char g_err[256];

const char* vag_oracle_last_error(void) {
    return g_err;
}

static int fail(const char* msg) {
    snprintf(g_err, sizeof g_err, "%s", msg);
    return -1;
}

static double dmin(double a, double b) {
    return b < a ? b : a; /* std::min */
}
static double dmax(double a, double b) {
    return a < b ? b : a; /* std::max */
}

/* src/util/fast-math.h:40-202 with AFTERGLOW_FAST_MATH off: exact libm */
static double log2_softplus(double x) {
    if (x > 20.0) return x;
    if (x < -20.0) return 0.0;
    return log2(1.0 + exp2(x));
}
static double fast_pow(double a, double b) {
    return exp2(b * log2(a));
}
static double log2_broken_power_ratio(double log2_x, double log2_x_break, double s_delta_beta, double s) {
    return -log2_softplus(s_delta_beta * (log2_x - log2_x_break)) / s;
}

/* src/core/physics.h:36-61 */
static double gamma_to_beta(double gamma) {
    return sqrt((gamma - 1) * (gamma + 1)) / gamma;
}
static double adiabatic_idx(double gamma) {
    return 4.0 / 3.0 + 1 / (3 * gamma);
}

/* ------------------------------------------------------------------------------------------
 * Jet and medium in code units.  src/environment/jet.h:84-259,421-441,
 * pybind/pymodel.cpp:47-146,188-210; src/environment/medium.h:50-133
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int type;
    double theta_c, eps_k, Gamma0; /* eps_k = E_iso/(4 pi) in code units (named jets) */
    double k_e, k_g, norm;         /* Gaussian: norm = -1/(2 theta_c^2) */
    double theta_w, E_iso_cgs, E_iso_w_cgs, Gm1, Gm1_w; /* two-component (Ejecta built in CGS) */
    double T0;
    double sigma0; /* constant ejecta magnetisation (VAG_JET_MAGNETIZED_TOPHAT), 0 otherwise */
    int spreading; /* jet(..., spreading=True) */
    int magnetar;  /* jet(..., magnetar=Magnetar(L0, t0, q)): the jet is built on the generic Ejecta (pymodel.cpp:47-128) */
    double mag_L0, mag_t0, mag_q;
} jet_t;

typedef struct {
    int type;
    double rho_ism; /* ISM: n*mp; Wind: floor */
    double A, r02;  /* Wind */
    int generic;    /* Wind with k_m != 2: python-level Medium built from a CGS closed form (pymodel.cpp:167-185) */
    double k_m, A_cgs, r0k_cgs, rho_ism_cgs;
} medium_t;

static void jet_init(jet_t* j, const vag_model_params* p) {
    memset(j, 0, sizeof *j);
    j->type = p->jet_type;
    j->theta_c = p->theta_c;
    j->eps_k = (p->E_iso * U_ERG) / (4 * C_PI);
    j->Gamma0 = p->Gamma0;
    j->k_e = p->k_e;
    j->k_g = p->k_g;
    j->norm = -1 / (2 * p->theta_c * p->theta_c);
    j->theta_w = p->theta_w;
    j->E_iso_cgs = p->E_iso;
    j->E_iso_w_cgs = p->E_iso_w;
    j->Gm1 = p->Gamma0 - 1;
    j->Gm1_w = p->Gamma0_w - 1;
    j->T0 = p->duration * U_SEC;
    j->sigma0 = (p->jet_type == VAG_JET_MAGNETIZED_TOPHAT) ? p->sigma0 : 0.0;
    j->spreading = (p->flags & VAG_FLAG_SPREADING) != 0;
    j->magnetar = (p->flags & VAG_FLAG_MAGNETAR) != 0;
    j->mag_L0 = p->mag_L0;
    j->mag_t0 = p->mag_t0;
    j->mag_q = p->mag_q;
}

/* is the jet an `Ejecta` object in the reference (python-level profile forms, HasSigma / HasDedt / HasDmdt)? */
static int jet_is_ejecta(const jet_t* j) {
    return j->type >= VAG_JET_TWO_COMPONENT || j->magnetar;
}

/* Ejecta::deps_dt of math::magnetar_injection (jet.h:518-527) through convert_unit_jet (pymodel.cpp:196-199) */
static double jet_deps_dt(const jet_t* j, double theta, double t) {
    if (!j->magnetar) return 0.0;
    double v = 0.;
    if (theta <= j->theta_c) {
        const double tt = 1 + (t / U_SEC) / j->mag_t0;
        v = j->mag_L0 * fast_pow(tt, -j->mag_q);
    }
    return v * (U_ERG / (4 * C_PI * U_SEC));
}

static double jet_eps_k(const jet_t* j, double theta) {
    if (j->magnetar && j->type <= VAG_JET_POWERLAW) { /* math::tophat / gaussian / powerlaw in CGS, then convert_unit_jet */
        double h;
        if (j->type == VAG_JET_TOPHAT)
            h = theta < j->theta_c ? j->E_iso_cgs : 0;
        else if (j->type == VAG_JET_GAUSSIAN)
            h = j->E_iso_cgs * exp(theta * theta / (-2 * j->theta_c * j->theta_c));
        else
            h = j->E_iso_cgs / (1 + fast_pow(theta / j->theta_c, j->k_e));
        return h * (U_ERG / (4 * C_PI));
    }
    switch (j->type) {
        case VAG_JET_TOPHAT: return theta < j->theta_c ? j->eps_k : 0;
        case VAG_JET_GAUSSIAN: return j->eps_k * exp(theta * theta * j->norm);
        case VAG_JET_POWERLAW: return j->eps_k / (1 + fast_pow(theta / j->theta_c, j->k_e));
        case VAG_JET_MAGNETIZED_TOPHAT: /* Ejecta(E_iso = E if theta <= theta_c else 0) in CGS, then convert_unit_jet */
            return (theta <= j->theta_c ? j->E_iso_cgs : 0.0) * (U_ERG / (4 * C_PI));
        case VAG_JET_STEP_POWERLAW: /* math::step_powerlaw, jet.h:418-426 */
            return (theta <= j->theta_c ? j->E_iso_cgs : j->E_iso_w_cgs * fast_pow(theta / j->theta_c, -j->k_e)) *
                   (U_ERG / (4 * C_PI));
        case VAG_JET_POWERLAW_WING: /* math::powerlaw_wing, jet.h:403-411 */
            return (theta <= j->theta_c ? 0. : j->E_iso_w_cgs * fast_pow(theta / j->theta_c, -j->k_e)) * (U_ERG / (4 * C_PI));
        default: { /* math::two_component (jet.h:421-433) in CGS, then convert_unit_jet */
            double h = theta <= j->theta_c ? j->E_iso_cgs : (theta <= j->theta_w ? j->E_iso_w_cgs : 0.);
            return h * (U_ERG / (4 * C_PI));
        }
    }
}

static double jet_Gamma0(const jet_t* j, double theta) {
    if (j->magnetar && j->type <= VAG_JET_POWERLAW) { /* math::*_plus_one(theta_c, Gamma0 - 1, ...) */
        double h;
        if (j->type == VAG_JET_TOPHAT)
            h = theta < j->theta_c ? j->Gm1 : 0;
        else if (j->type == VAG_JET_GAUSSIAN)
            h = j->Gm1 * exp(theta * theta / (-2 * j->theta_c * j->theta_c));
        else
            h = j->Gm1 / (1 + fast_pow(theta / j->theta_c, j->k_g));
        return h + 1;
    }
    switch (j->type) {
        case VAG_JET_TOPHAT: return theta < j->theta_c ? j->Gamma0 : 1;
        case VAG_JET_GAUSSIAN: return (j->Gamma0 - 1) * exp(theta * theta * j->norm) + 1;
        case VAG_JET_POWERLAW: return (j->Gamma0 - 1) / (1 + fast_pow(theta / j->theta_c, j->k_g)) + 1;
        case VAG_JET_MAGNETIZED_TOPHAT: return theta <= j->theta_c ? j->Gamma0 : 1.0;
        case VAG_JET_STEP_POWERLAW: return (theta <= j->theta_c ? j->Gm1 : j->Gm1_w * fast_pow(theta / j->theta_c, -j->k_g)) + 1;
        case VAG_JET_POWERLAW_WING: return (theta <= j->theta_c ? 0. : j->Gm1_w * fast_pow(theta / j->theta_c, -j->k_g)) + 1;
        default: {
            double h = theta <= j->theta_c ? j->Gm1 : (theta <= j->theta_w ? j->Gm1_w : 0.);
            return h + 1;
        }
    }
}

static void medium_init(medium_t* m, const vag_model_params* p) {
    memset(m, 0, sizeof *m);
    m->type = p->medium_type;
    if (m->type == VAG_MEDIUM_ISM) {
        m->rho_ism = (p->n_ism / U_CM3) * C_MP;
    } else {
        const double n_ism = p->n_ism / U_CM3, n0 = p->n0 / U_CM3;
        m->A = p->A_star * 5e11 * U_G / U_CM;
        m->rho_ism = n_ism * C_MP;
        m->r02 = m->A / (n0 * 1.3 * C_MP);
        m->generic = (p->k_m != 2);
        if (m->generic) { /* CGS closure of PyWind, evaluated through convert_unit_medium (pymodel.cpp:212-224) */
            const double r0_cgs = 1e17;
            const double mp_cgs = C_MP / U_G;
            m->k_m = p->k_m;
            m->A_cgs = p->A_star * 5e11 * pow(r0_cgs, p->k_m - 2);
            m->rho_ism_cgs = p->n_ism * mp_cgs;
            m->r0k_cgs = m->A_cgs / (p->n0 * 1.3 * mp_cgs);
        }
    }
}

static double medium_rho(const medium_t* m, double r) {
    if (m->type == VAG_MEDIUM_ISM) return m->rho_ism;
    if (m->generic) return (m->A_cgs / (m->r0k_cgs + pow(r / U_CM, m->k_m)) + m->rho_ism_cgs) * (U_G / U_CM3);
    return m->A / (m->r02 + r * r) + m->rho_ism;
}

static double enclosed_mass_generic(const medium_t* med, double r);
static double medium_mass(const medium_t* m, double r) { /* enclosed_mass_medium, shock-physics.h:439-446 */
    if (m->type == VAG_MEDIUM_ISM) return m->rho_ism * r * r * r / 3.0;
    if (m->generic) return enclosed_mass_generic(m, r); /* python-level Medium has no mass(): Simpson in log radius */
    double mass = m->rho_ism * r * r * r / 3.0;
    if (m->A != 0) {
        if (m->r02 > 0) {
            const double a = sqrt(m->r02);
            mass += m->A * (r - a * atan(r / a));
        } else {
            mass += m->A * r;
        }
    }
    return mass;
}

/* ------------------------------------------------------------------------------------------
 * DOPRI5 with step-size control and dense output, boost::odeint semantics:
 * external/boost/numeric/odeint/stepper/runge_kutta_dopri5.hpp:88-258,
 * controlled_runge_kutta.hpp:56-156,752-782, dense_output_runge_kutta.hpp:324-361,
 * algebra/default_operations.hpp:431-447, integrate/max_step_checker.hpp:92
 * ---------------------------------------------------------------------------------------- */
#define ODE_MAXN 12
typedef void (*rhs_fn)(const double* x, double* dxdt, double t, void* ctx);

typedef struct {
    int n;
    double eps_abs, eps_rel;
    double x[2][ODE_MAXN], dx[2][ODE_MAXN];
    int cur;
    double k2[ODE_MAXN], k3[ODE_MAXN], k4[ODE_MAXN], k5[ODE_MAXN], k6[ODE_MAXN];
    double t, t_old, dt;
    int deriv_init;
} dopri5_t;

static void dopri5_init(dopri5_t* s, int n, double eps_abs, double eps_rel, const double* x0, double t0, double dt0) {
    s->n = n;
    s->eps_abs = eps_abs;
    s->eps_rel = eps_rel;
    s->cur = 0;
    for (int i = 0; i < n; ++i) s->x[0][i] = x0[i];
    s->t = t0;
    s->dt = dt0;
    s->deriv_init = 0;
}

/* one controlled trial; returns 1 on success */
static int dopri5_try_step(dopri5_t* s, rhs_fn f, void* ctx) {
    const double a2 = 1.0 / 5, a3 = 3.0 / 10, a4 = 4.0 / 5, a5 = 8.0 / 9;
    const double b21 = 1.0 / 5;
    const double b31 = 3.0 / 40, b32 = 9.0 / 40;
    const double b41 = 44.0 / 45, b42 = -56.0 / 15, b43 = 32.0 / 9;
    const double b51 = 19372.0 / 6561, b52 = -25360.0 / 2187, b53 = 64448.0 / 6561, b54 = -212.0 / 729;
    const double b61 = 9017.0 / 3168, b62 = -355.0 / 33, b63 = 46732.0 / 5247, b64 = 49.0 / 176,
                 b65 = -5103.0 / 18656;
    const double c1 = 35.0 / 384, c3 = 500.0 / 1113, c4 = 125.0 / 192, c5 = -2187.0 / 6784, c6 = 11.0 / 84;
    const double dc1 = c1 - 5179.0 / 57600, dc3 = c3 - 7571.0 / 16695, dc4 = c4 - 393.0 / 640,
                 dc5 = c5 - -92097.0 / 339200, dc6 = c6 - 187.0 / 2100, dc7 = -1.0 / 40;
    const int n = s->n;
    const double* in = s->x[s->cur];
    const double* k1 = s->dx[s->cur];
    double* out = s->x[1 - s->cur];
    double* k7 = s->dx[1 - s->cur];
    double xt[ODE_MAXN], xerr[ODE_MAXN];
    const double t = s->t, dt = s->dt;

    for (int i = 0; i < n; ++i) xt[i] = 1.0 * in[i] + (dt * b21) * k1[i];
    f(xt, s->k2, t + dt * a2, ctx);
    for (int i = 0; i < n; ++i) xt[i] = 1.0 * in[i] + (dt * b31) * k1[i] + (dt * b32) * s->k2[i];
    f(xt, s->k3, t + dt * a3, ctx);
    for (int i = 0; i < n; ++i) xt[i] = 1.0 * in[i] + (dt * b41) * k1[i] + (dt * b42) * s->k2[i] + (dt * b43) * s->k3[i];
    f(xt, s->k4, t + dt * a4, ctx);
    for (int i = 0; i < n; ++i)
        xt[i] = 1.0 * in[i] + (dt * b51) * k1[i] + (dt * b52) * s->k2[i] + (dt * b53) * s->k3[i] + (dt * b54) * s->k4[i];
    f(xt, s->k5, t + dt * a5, ctx);
    for (int i = 0; i < n; ++i)
        xt[i] = 1.0 * in[i] + (dt * b61) * k1[i] + (dt * b62) * s->k2[i] + (dt * b63) * s->k3[i] + (dt * b64) * s->k4[i] +
                (dt * b65) * s->k5[i];
    f(xt, s->k6, t + dt, ctx);
    for (int i = 0; i < n; ++i)
        out[i] = 1.0 * in[i] + (dt * c1) * k1[i] + (dt * c3) * s->k3[i] + (dt * c4) * s->k4[i] + (dt * c5) * s->k5[i] +
                 (dt * c6) * s->k6[i];
    f(out, k7, t + dt, ctx);
    for (int i = 0; i < n; ++i)
        xerr[i] = (dt * dc1) * k1[i] + (dt * dc3) * s->k3[i] + (dt * dc4) * s->k4[i] + (dt * dc5) * s->k5[i] +
                  (dt * dc6) * s->k6[i] + (dt * dc7) * k7[i];

    double err = 0;
    for (int i = 0; i < n; ++i) {
        const double e = fabs(xerr[i]) / (s->eps_abs + s->eps_rel * (1.0 * fabs(in[i]) + (1.0 * fabs(dt)) * fabs(k1[i])));
        err = dmax(err, e);
    }
    if (err > 1.0) {
        s->dt = dt * dmax(9.0 / 10.0 * pow(err, -1.0 / (4 - 1)), 1.0 / 5.0);
        return 0;
    }
    s->t = t + dt;
    if (err < 0.5) {
        err = dmax(pow(5.0, -5.0), err);
        s->dt = dt * (9.0 / 10.0 * pow(err, -1.0 / 5));
    }
    return 1;
}

/* dense-output do_step; returns 0 ok, -1 after 500 consecutive rejections */
static int dopri5_do_step(dopri5_t* s, rhs_fn f, void* ctx) {
    if (!s->deriv_init) {
        f(s->x[s->cur], s->dx[s->cur], s->t, ctx);
        s->deriv_init = 1;
    }
    s->t_old = s->t;
    int fails = 0;
    while (!dopri5_try_step(s, f, ctx)) {
        if (++fails >= 500) return -1;
    }
    s->cur = 1 - s->cur;
    return 0;
}

static void dopri5_calc_state(const dopri5_t* s, double t, double* x) {
    const double b1 = 35.0 / 384, b3 = 500.0 / 1113, b4 = 125.0 / 192, b5 = -2187.0 / 6784, b6 = 11.0 / 84;
    const double* x_old = s->x[1 - s->cur];
    const double* k1 = s->dx[1 - s->cur];
    const double* k7 = s->dx[s->cur];
    const double dt = s->t - s->t_old;
    const double theta = (t - s->t_old) / dt;
    const double X1 = 5.0 * (2558722523.0 - 31403016.0 * theta) / 11282082432.0;
    const double X3 = 100.0 * (882725551.0 - 15701508.0 * theta) / 32700410799.0;
    const double X4 = 25.0 * (443332067.0 - 31403016.0 * theta) / 1880347072.0;
    const double X5 = 32805.0 * (23143187.0 - 3489224.0 * theta) / 199316789632.0;
    const double X6 = 55.0 * (29972135.0 - 7076736.0 * theta) / 822651844.0;
    const double X7 = 10.0 * (7414447.0 - 829305.0 * theta) / 29380423.0;
    const double theta_m_1 = theta - 1.0;
    const double theta_sq = theta * theta;
    const double A = theta_sq * (3.0 - 2.0 * theta);
    const double B = theta_sq * theta_m_1;
    const double C = theta_sq * theta_m_1 * theta_m_1;
    const double D = theta * theta_m_1 * theta_m_1;
    const double b1_theta = A * b1 - C * X1 + D;
    const double b3_theta = A * b3 + C * X3;
    const double b4_theta = A * b4 - C * X4;
    const double b5_theta = A * b5 + C * X5;
    const double b6_theta = A * b6 - C * X6;
    const double b7_theta = B + C * X7;
    for (int i = 0; i < s->n; ++i)
        x[i] = 1.0 * x_old[i] + (dt * b1_theta) * k1[i] + (dt * b3_theta) * s->k3[i] + (dt * b4_theta) * s->k4[i] +
               (dt * b5_theta) * s->k5[i] + (dt * b6_theta) * s->k6[i] + (dt * b7_theta) * k7[i];
}

/* ------------------------------------------------------------------------------------------
 * Grid container: src/core/mesh.h:66-95 (Coord), axisymmetric => one phi slice of t
 * ---------------------------------------------------------------------------------------- */
enum { SYM_STRUCTURED = 0, SYM_PHI_SYMMETRIC = 1, SYM_PIECEWISE = 2, SYM_ISOTROPIC = 3 };

typedef struct {
    int n_phi, n_theta, n_t, n_reps;
    int phi_size; /* phi slices of t / Shock / electrons / photons that are kept: 1, or n_phi for Model(axisymmetric=False) with a
                   * SPREADING jet, whose lattices start per (phi, theta) node (grid-refinement.h:462-469,619-625).  A "row" below is
                   * (slice i, theta j) -> i * n_theta + j; with one slice it is the theta index. */
    double* phi;
    double* theta;
    double* t; /* [phi_size * n_theta][n_t] engine-frame lattice */
    int* reps; /* representative rows */
    int symmetry, phi_mirrored;
    int spreading; /* Coord::spreading, mesh.h:92 */
    int jet_3d;    /* Model(axisymmetric=False) with more than one phi node: Observer::jet_3d, observer.cpp:215. The named
                    * jets are phi-independent, so without spreading every phi slice of t / Shock equals slice 0 and only slice 0
                    * is kept (phi_size = 1). */
    double theta_view;
} coord_t;

static void coord_free(coord_t* c) {
    free(c->phi);
    free(c->theta);
    free(c->t);
    free(c->reps);
    memset(c, 0, sizeof *c);
}

/* xt::linspace / xt::logspace: external/xtensor/generators/xbuilder.hpp:231-280,460-484 */
static void linspace(double start, double stop, int n, double* out) {
    const double step = (stop - start) / fmax(1.0, (double)(n - 1));
    for (int i = 0; i < n; ++i) out[i] = (n > 1 && i == n - 1) ? stop : start + step * (double)i;
}
static void logspace10(double start, double stop, int n, double* out) {
    linspace(start, stop, n, out);
    for (int i = 0; i < n; ++i) out[i] = pow(10.0, out[i]);
}

/* src/core/grid-refinement.h:33-35 */
static double structure_weight(double Gamma) {
    return Gamma * sqrt(dmax((Gamma - 1) * Gamma, 0.0));
}

/* src/core/grid-refinement.h:41-86 */
static int find_jet_jumps(const jet_t* jet, double gamma_cut, double* jumps, int max_jumps) {
    const int n_scan = 512;
    const double eps = DEF_BINARY_SEARCH_EPS;
    const double theta_lo = DEF_THETA_MIN;
    const double theta_hi = C_PI / 2;
    const double dtheta = (theta_hi - theta_lo) / (n_scan - 1);
    if (jet_Gamma0(jet, theta_hi) >= gamma_cut) {
        jumps[0] = theta_hi;
        return 1;
    }
    int n = 0;
    double prev_th = theta_lo;
    double prev_G = jet_Gamma0(jet, theta_lo);
    for (int j = 1; j < n_scan; ++j) {
        const double cur_th = theta_lo + dtheta * (double)j;
        const double cur_G = jet_Gamma0(jet, cur_th);
        if (prev_G >= gamma_cut || cur_G >= gamma_cut) {
            const double dG = fabs(cur_G - prev_G);
            const double scale = dmax(prev_G - 1, cur_G - 1);
            if (scale > 0 && dG > 0.5 * scale) {
                double lo = prev_th, hi = cur_th;
                while (hi - lo > eps) {
                    const double mid = 0.5 * (lo + hi);
                    const double G_mid = jet_Gamma0(jet, mid);
                    if (fabs(G_mid - prev_G) < fabs(G_mid - cur_G)) {
                        lo = mid;
                    } else {
                        hi = mid;
                    }
                }
                if (n < max_jumps) jumps[n++] = prev_G > cur_G ? lo : hi;
            }
        }
        prev_th = cur_th;
        prev_G = cur_G;
    }
    return n;
}

/* src/core/grid-refinement.h:89-111 */
static void find_theta_range(const jet_t* jet, double gamma_cut, double* th_min, double* th_max) {
    const int n_scan = 512;
    const double theta_lo = DEF_THETA_MIN;
    const double theta_hi = C_PI / 2;
    double theta_max = theta_hi, theta_min = theta_lo;
    const double step = (theta_hi - theta_lo) / n_scan;
    for (double th = theta_hi; th >= theta_lo; th -= step) {
        if (jet_Gamma0(jet, th) >= gamma_cut) {
            theta_max = th;
            break;
        }
    }
    for (double th = theta_lo; th <= theta_hi; th += step) {
        if (jet_Gamma0(jet, th) >= gamma_cut) {
            theta_min = th;
            break;
        }
    }
    *th_min = theta_min;
    *th_max = theta_max;
}

/* src/core/grid-refinement.h:138-189.  Returns malloc'ed x_out[num]. */
static double* inverse_cdf_sampling(rhs_fn pdf, void* ctx, double min, double max, int num, int sample_num,
                                    int log_sample, int midpoint) {
    const double rtol = DEF_ODE_RTOL;
    double* x_i = malloc(sizeof(double) * sample_num);
    double* cdf_i = calloc(sample_num, sizeof(double));
    if (log_sample)
        logspace10(log10(min), log10(max), sample_num, x_i);
    else
        linspace(min, max, sample_num, x_i);

    dopri5_t st;
    const double x0 = 0;
    dopri5_init(&st, 1, rtol, rtol, &x0, min, (max - min) / 1e3);
    for (int k = 1, steps = 0; st.t <= max;) {
        if (dopri5_do_step(&st, pdf, ctx) != 0) break;
        if (++steps > DEF_MAX_ODE_STEPS) break;
        while (k < sample_num && st.t > x_i[k]) {
            dopri5_calc_state(&st, x_i[k], &cdf_i[k]);
            ++k;
        }
    }

    double* cdf_out = malloc(sizeof(double) * (num > 0 ? num : 1));
    const double front = cdf_i[0], back = cdf_i[sample_num - 1];
    if (midpoint) {
        for (int k = 0; k < num; ++k) cdf_out[k] = front + (back - front) * ((double)k + 0.5) / num;
    } else {
        linspace(front, back, num, cdf_out);
    }
    double* x_out = calloc(num > 0 ? num : 1, sizeof(double));
    for (int k = 0; k < num; ++k) {
        for (int j = 0; j < sample_num; ++j) {
            if (cdf_out[k] <= cdf_i[j]) {
                if (j == 0) {
                    x_out[k] = x_i[j];
                } else {
                    const double denom = cdf_i[j] - cdf_i[j - 1];
                    if (denom > 0) {
                        const double slope = (x_i[j] - x_i[j - 1]) / denom;
                        x_out[k] = x_i[j - 1] + slope * (cdf_out[k] - cdf_i[j - 1]);
                    } else {
                        x_out[k] = x_i[j - 1];
                    }
                }
                break;
            }
        }
    }
    free(x_i);
    free(cdf_i);
    free(cdf_out);
    return x_out;
}

/* src/core/grid-refinement.h:191-291 */
typedef struct {
    const jet_t* jet;
    double theta_v, core_weight, view_weight, Gamma_peak_sq, Gamma_v_sq, doppler_alpha, floor_weight;
} theta_pdf_ctx;

static void theta_pdf(const double* cdf, double* pdf, double theta, void* vctx) {
    (void)cdf;
    const theta_pdf_ctx* c = vctx;
    const double Gamma = jet_Gamma0(c->jet, theta);
    const double beta = gamma_to_beta(Gamma);
    const double doppler = (1 - beta) / (1 - beta * cos(theta - c->theta_v));
    const double structure = structure_weight(Gamma);
    const double dtheta = theta - c->theta_v;
    *pdf = c->core_weight * c->Gamma_peak_sq * theta / (1.0 + c->Gamma_peak_sq * theta * theta) +
           c->view_weight * c->Gamma_v_sq * fabs(dtheta) / (1.0 + c->Gamma_v_sq * dtheta * dtheta) +
           (1 + c->doppler_alpha * doppler) * structure + c->floor_weight;
}

static size_t beam_pts(double log_decades, double theta_resol, double beam_coeff, double offset) {
    return (size_t)(dmax(0.0, log_decades - offset) * theta_resol * beam_coeff);
}

static double* adaptive_theta_grid(const jet_t* jet, double theta_min, double theta_max, size_t base_pts, double theta_v,
                                   double theta_resol, int* n_out) {
    const double core_beam_coeff = 55.0, view_beam_coeff = 25.0, doppler_alpha0 = 12.0, floor_fraction = 0.25;
    const int scan_pts = 100;
    const double theta_extent = theta_max - theta_min;
    double peak_weight = 0, Gamma_peak = 1.0, struct_sum = 0, Gamma_v = 1.0;
    int last_bright = 0;
    for (int i = 0; i <= scan_pts; ++i) {
        const double theta = theta_min + theta_extent * i / scan_pts;
        const double Gamma = jet_Gamma0(jet, theta);
        const double w = structure_weight(Gamma);
        struct_sum += w;
        if (w > peak_weight) {
            peak_weight = w;
            Gamma_peak = Gamma;
            last_bright = i;
        } else if (w > 0.01 * peak_weight) {
            last_bright = i;
        }
        const double dth = theta - theta_v;
        Gamma_v = dmax(Gamma_v, Gamma / sqrt(1.0 + Gamma * Gamma * dth * dth));
    }
    const double floor_weight = floor_fraction * peak_weight;
    const double CDF_est = (struct_sum / scan_pts + floor_weight) * theta_extent;
    const double theta_bright = theta_min + theta_extent * last_bright / scan_pts;

    Gamma_peak = dmax(Gamma_peak, Gamma_v);
    const double doppler_alpha = doppler_alpha0 * sqrt(peak_weight / dmax(structure_weight(Gamma_v), 1.0));
    const double beam_offset = 1.0;
    const double Gamma_peak_sq = Gamma_peak * Gamma_peak;
    const double Gamma_v_sq = Gamma_v * Gamma_v;
    const size_t core_beam_pts =
        beam_pts(log10(dmax(1.0, Gamma_peak * (theta_bright - theta_min))), theta_resol, core_beam_coeff, beam_offset);
    const size_t view_beam_pts =
        (theta_v * Gamma_peak > 3.0)
            ? beam_pts(log10(dmax(1.0, Gamma_v * dmax(theta_v - theta_min, theta_max - theta_v))), theta_resol,
                       view_beam_coeff, 0.0)
            : 0;
    const size_t total_pts = base_pts + core_beam_pts + view_beam_pts;

    const double core_cdf =
        0.5 * log((1.0 + Gamma_peak_sq * theta_max * theta_max) / (1.0 + Gamma_peak_sq * theta_min * theta_min));
    const double core_weight =
        (core_beam_pts > 0 && core_cdf > 0) ? (double)core_beam_pts / base_pts * CDF_est / core_cdf : 0.0;
    const double theta_v_left = theta_v - theta_min;
    const double theta_v_right = theta_max - theta_v;
    const double view_cdf = 0.5 * (log(1.0 + Gamma_v_sq * theta_v_left * theta_v_left) +
                                   log(1.0 + Gamma_v_sq * theta_v_right * theta_v_right));
    const double view_weight =
        (view_beam_pts > 0 && view_cdf > 0) ? (double)view_beam_pts / base_pts * CDF_est / view_cdf : 0.0;

    theta_pdf_ctx ctx = {jet, theta_v, core_weight, view_weight, Gamma_peak_sq, Gamma_v_sq, doppler_alpha, floor_weight};
    *n_out = (int)total_pts;
    return inverse_cdf_sampling(theta_pdf, &ctx, theta_min, theta_max, (int)total_pts, DEF_THETA_SAMPLES, 1, 0);
}

static int cmp_double(const void* a, const void* b) {
    const double x = *(const double*)a, y = *(const double*)b;
    return (x > y) - (x < y);
}

/* src/core/grid-refinement.cpp:136-160 */
static int jump_refinement_grid(const double* jumps, int n_jumps, double theta_min, double theta_max, double avg_spacing,
                                double* points) {
    int n = 0;
    const double tight = avg_spacing / 8;
    for (int idx = 0; idx < n_jumps; ++idx) {
        const double jump_theta = jumps[idx];
        if (jump_theta >= C_PI / 2 - 0.01) continue;
        if (jump_theta - tight >= theta_min) points[n++] = jump_theta - tight;
        if (jump_theta + tight <= theta_max) points[n++] = jump_theta + tight;
        if (jump_theta >= theta_min && jump_theta <= theta_max) points[n++] = jump_theta;
    }
    qsort(points, n, sizeof(double), cmp_double);
    int m = 0;
    for (int i = 0; i < n; ++i)
        if (m == 0 || points[m - 1] != points[i]) points[m++] = points[i];
    return m;
}

/* src/core/grid-refinement.h:362-393 */
static int merge_grids(const double* a, int na, const double* b, int nb, double* out) {
    int n = 0, i = 0, j = 0;
#define ADD_UNIQUE(v)                          \
    do {                                       \
        const double v_ = (v);                 \
        if (n == 0 || out[n - 1] != v_) out[n++] = v_; \
    } while (0)
    while (i < na && j < nb) {
        if (a[i] <= b[j]) {
            ADD_UNIQUE(a[i++]);
            if (a[i - 1] == b[j]) j++;
        } else {
            ADD_UNIQUE(b[j++]);
        }
    }
    while (i < na) ADD_UNIQUE(a[i++]);
    while (j < nb) ADD_UNIQUE(b[j++]);
#undef ADD_UNIQUE
    return n;
}

/* src/core/grid-refinement.h:296-360 */
typedef struct {
    const jet_t* jet;
    const double* theta_grid;
    const double* dcos;
    int n_theta;
    double cos_tv, sin_tv, floor_weight;
} phi_pdf_ctx;

static double phi_weight(const phi_pdf_ctx* c, double phi) {
    const double cos_phi = cos(phi);
    double w = 0;
    for (int it = 0; it < c->n_theta; ++it) {
        const double theta = c->theta_grid[it];
        const double Gamma = jet_Gamma0(c->jet, theta);
        const double beta = gamma_to_beta(Gamma);
        const double cos_alpha = cos(theta) * c->cos_tv + sin(theta) * c->sin_tv * cos_phi;
        const double a = (1 - beta) / (1 - beta * cos_alpha);
        w += a * structure_weight(Gamma) * c->dcos[it];
    }
    return w;
}

static void phi_pdf(const double* cdf, double* pdf, double phi, void* vctx) {
    (void)cdf;
    const phi_pdf_ctx* c = vctx;
    *pdf = phi_weight(c, phi) + c->floor_weight;
}

static double* adaptive_phi_grid(const jet_t* jet, size_t phi_num, double theta_v, const double* theta_grid, int n_theta,
                                 int is_axisymmetric, double phi_max, double self_boost_cap, int* n_out) {
    if (theta_v == 0 && is_axisymmetric) {
        double* out = malloc(sizeof(double) * (phi_num > 0 ? phi_num : 1));
        linspace(0., 2 * C_PI, (int)phi_num, out);
        *n_out = (int)phi_num;
        return out;
    }
    const int half_range = phi_max < 2 * C_PI;
    double* dcos = malloc(sizeof(double) * n_theta);
    for (int it = 0; it < n_theta; ++it) {
        const double left = (it == 0) ? 0.0 : 0.5 * (theta_grid[it - 1] + theta_grid[it]);
        const double right = (it == n_theta - 1) ? theta_grid[it] : 0.5 * (theta_grid[it] + theta_grid[it + 1]);
        dcos[it] = fabs(cos(left) - cos(right));
    }
    phi_pdf_ctx ctx = {jet, theta_grid, dcos, n_theta, cos(theta_v), sin(theta_v), 0.0};
    const int scan_pts = 100;
    double peak_weight = 0, sum_weight = 0;
    for (int s = 0; s <= scan_pts; ++s) {
        const double phi = phi_max * (double)s / scan_pts;
        const double w = phi_weight(&ctx, phi);
        peak_weight = dmax(peak_weight, w);
        sum_weight += w;
    }
    const double floor_weight = 0.05 * peak_weight;
    if (self_boost_cap > 0 && peak_weight > 0) {
        const double mean_pdf = sum_weight / (scan_pts + 1) + floor_weight;
        const double concentration = (peak_weight + floor_weight) / mean_pdf;
        double boost = concentration / 5;
        boost = boost < 1.0 ? 1.0 : (self_boost_cap < boost ? self_boost_cap : boost); /* std::clamp */
        phi_num = (size_t)((double)phi_num * boost);
    }
    ctx.floor_weight = floor_weight;
    double* out = inverse_cdf_sampling(phi_pdf, &ctx, 0, phi_max, (int)phi_num, DEF_THETA_SAMPLES, 0, half_range);
    free(dcos);
    *n_out = (int)phi_num;
    return out;
}

/* src/core/grid-refinement.h:402-453 */
static double estimate_t_dec(const jet_t* jet, const medium_t* med, double theta) {
    const double gamma = jet_Gamma0(jet, theta);
    const double beta = gamma_to_beta(gamma);
    double m_jet = jet_eps_k(jet, theta) / (gamma * C_C2);
    if (jet_is_ejecta(jet)) m_jet /= (1.0 + jet->sigma0); /* HasSigma<Ejecta> */
    const double target = m_jet / gamma;
    const double r_min = 1e-3;
    const double r_max = r_min * pow(10.0, 40.0);
    if (target <= 0) return r_min * (1 - beta) / (beta * C_C);

    if (med->type == VAG_MEDIUM_ISM) {
        const double rho = medium_rho(med, r_min);
        if (rho > 0) {
            const double r3_dec = r_min * r_min * r_min + 3 * target / rho;
            const double r_dec = cbrt(dmax(r3_dec, 0.0));
            return dmin(r_dec, r_max) * (1 - beta) / (beta * C_C);
        }
        return r_max * (1 - beta) / (beta * C_C);
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
            return r_dec * (1 - beta) / (beta * C_C);
        }
        f_prev = f_i;
        r_prev = r_i;
    }
    return exp(u_max) * (1 - beta) / (beta * C_C);
}

/* src/core/grid-refinement.h:533-569: grid[n] */
static void logspace_with_band_refinement(double ts, double t_end, double b_lo, double b_hi, size_t n, double factor,
                                          double* grid) {
    b_lo = dmax(b_lo, ts);
    b_hi = dmin(b_hi, t_end);
    if (!(b_hi > b_lo) || n < 8) {
        logspace10(log10(ts), log10(t_end), (int)n, grid);
        return;
    }
    const double l0 = log10(ts), l1 = log10(b_lo), l2 = log10(b_hi), l3 = log10(t_end);
    const double w1 = l1 - l0, w2 = factor * (l2 - l1), w3 = l3 - l2;
    const size_t segs = n - 1;
    size_t n1 = (size_t)round((double)segs * w1 / (w1 + w2 + w3));
    size_t n3 = (size_t)round((double)segs * w3 / (w1 + w2 + w3));
    if (segs - 2 < n1) n1 = segs - 2;
    if (segs - 1 - n1 - 1 < n3) n3 = segs - 1 - n1 - 1;
    const size_t n2 = segs - n1 - n3;
    size_t idx = 0;
    for (size_t k = 0; k < n1; ++k) grid[idx++] = l0 + (l1 - l0) * (double)k / (double)n1;
    for (size_t k = 0; k < n2; ++k) grid[idx++] = l1 + (l2 - l1) * (double)k / (double)n2;
    for (size_t k = 0; k <= n3; ++k) grid[idx++] = (n3 > 0) ? l2 + (l3 - l2) * (double)k / (double)n3 : l3;
    for (size_t k = 0; k < n; ++k) grid[k] = pow(10.0, grid[k]);
}

/* Coord::detect_symmetry, src/core/mesh.h:121-187 (isotropic medium) */
static void detect_symmetry(coord_t* c, const jet_t* jet) {
    c->reps = malloc(sizeof(int) * c->phi_size * c->n_theta);
    c->n_reps = 0;
    c->spreading = jet->spreading;
    if (jet->spreading) { /* every row evolves on its own: Symmetry::structured */
        for (int j = 0; j < c->phi_size * c->n_theta; ++j) c->reps[c->n_reps++] = j;
        c->symmetry = SYM_STRUCTURED;
        return;
    }
    c->reps[c->n_reps++] = 0;
    for (int j = 1; j < c->n_theta; ++j) {
        const double ta = c->theta[j - 1], tb = c->theta[j];
        if (jet_eps_k(jet, ta) != jet_eps_k(jet, tb) || jet_Gamma0(jet, ta) != jet_Gamma0(jet, tb)) {
            c->reps[c->n_reps++] = j;
        }
    }
    if (c->n_reps == 1)
        c->symmetry = SYM_ISOTROPIC;
    else if (c->n_reps < c->n_theta)
        c->symmetry = SYM_PIECEWISE;
    else
        c->symmetry = SYM_PHI_SYMMETRIC;
}

/* logspace_with_cross_refinement, src/core/grid-refinement.cpp:166-197: result[t_num] */
static void logspace_with_cross_refinement(double t_start, double t_end, double t_refine, size_t t_num, size_t base_t_num,
                                           double* result) {
    t_refine = dmin(dmax(t_refine, t_start), t_end); /* std::clamp */
    if (t_refine <= t_start || t_refine >= t_end) {
        logspace10(log10(t_start), log10(t_end), (int)t_num, result);
        return;
    }
    const double log_total = log10(t_end / t_start);
    const double log_after = log10(t_end / t_refine);
    size_t n_post = (size_t)(base_t_num * log_after / log_total);
    if (n_post < 2) n_post = 2;
    if (n_post >= t_num) n_post = t_num / 2;
    const size_t n_pre = t_num + 1 - n_post;
    double* seg = malloc(sizeof(double) * ((n_pre > n_post ? n_pre : n_post) + 1));
    size_t idx = 0;
    for (size_t k = 0; k < t_num; ++k) result[k] = 0;
    logspace10(log10(t_start), log10(t_refine), (int)n_pre, seg);
    for (size_t k = 0; k < n_pre; ++k) result[idx++] = seg[k];
    logspace10(log10(t_refine), log10(t_end), (int)n_post, seg);
    for (size_t k = 1; k < n_post && idx < t_num; ++k) result[idx++] = seg[k];
    free(seg);
}

/* build_time_grid + scan_time_bounds + compute_time_grid_size + make_time_grid + store_time_grid,
 * src/core/grid-refinement.h:472-528,571-636 */
static void build_time_grid(coord_t* c, const jet_t* jet, const medium_t* med, double t_min, double t_max, double z,
                            double t_resol, int is_rvs, int is_axisymmetric) {
    const int nth = c->n_theta;
    const double t_end = 1.01 * t_max / (1 + z);
    const double cos_tv = cos(c->theta_view), sin_tv = sin(c->theta_view);
    double min_raw = t_end, min_guarded = t_end, min_cut = t_end, max_ref = 0;
    double* t_dec = malloc(sizeof(double) * nth);
    double* t_start_row = malloc(sizeof(double) * c->phi_size * nth); /* TimeScanResult::t_start / early_t, grid-refinement.h:462-469 */
    double* early_t_row = malloc(sizeof(double) * c->phi_size * nth);
    /* phi_size = |phi| for Model(axisymmetric=False): the global bounds then scan every phi node (:484-507); the per-row
     * caches keep the slices the lattices below read (slice 0, or every slice of a spreading jet) */
    const int phi_size = is_axisymmetric ? 1 : c->n_phi;
    for (int i = 0; i < phi_size; ++i)
        for (int j = 0; j < nth; ++j) {
            const double b = gamma_to_beta(jet_Gamma0(jet, c->theta[j]));
            const double cos_a = cos(c->theta[j]) * cos_tv + sin(c->theta[j]) * sin_tv * cos(c->phi[i]);
            const double ts = 0.99 * t_min * (1 - b) / (1 - cos_a * b) / (1 + z);
            const double td = (i == 0) ? estimate_t_dec(jet, med, c->theta[j]) : t_dec[j];
            t_dec[j] = td;
            double cut = dmin(0.01 * td, 1e-2 * U_SEC);
            if (is_rvs) {
                cut = dmin(cut, 0.01 * jet->T0);
                max_ref = dmax(max_ref, 10.0 * dmax(td, jet->T0));
            }
            if (i < c->phi_size) {
                t_start_row[i * nth + j] = dmax(ts, cut);
                early_t_row[i * nth + j] = 0.99 * dmin(ts, cut);
            }
            min_raw = dmin(min_raw, ts);
            min_guarded = dmin(min_guarded, dmax(ts, cut));
            min_cut = dmin(min_cut, cut);
        }
    const double min_t_early = min_raw, min_t_start = min_guarded;
    const int has_early = min_raw < min_cut;
    const size_t t_num_base = (size_t)(dmax(log10(t_end / min_t_start), 1.0) * t_resol);
    size_t t_num_rvs_extra = 0; /* compute_time_grid_size: rvs_refinement_ratio = 2 */
    if (is_rvs && max_ref > min_t_start) {
        const double log_pre_span = log10(dmin(max_ref, t_end) / min_t_start);
        t_num_rvs_extra = (size_t)((2.0 - 1.0) * log_pre_span * t_resol);
    }
    const size_t t_num_tot = t_num_base + t_num_rvs_extra;
    const size_t t_num = t_num_tot + (has_early ? 1 : 0);
    c->n_t = (int)t_num;
    c->t = calloc((size_t)c->phi_size * nth * t_num, sizeof(double));
    double* grid = malloc(sizeof(double) * (t_num_tot > 0 ? t_num_tot : 1));
    for (int r = 0; r < c->n_reps; ++r) {
        const int j_rep = c->reps[r];
        const int j_end = (r + 1 < c->n_reps) ? c->reps[r + 1] : nth;
        if (c->symmetry < SYM_PHI_SYMMETRIC) { /* structured: every row has its own start and early point (:619-625) */
            const int R = j_rep, j = R % nth; /* row (phi slice, theta j) */
            if (is_rvs) {
                const double t_cross_limit = dmax(t_dec[j], jet->T0);
                logspace_with_cross_refinement(t_start_row[R], t_end, 10 * t_cross_limit, t_num_tot, t_num_base, grid);
            } else {
                logspace_with_band_refinement(t_start_row[R], t_end, t_dec[j] / 3, 3 * t_dec[j], t_num_tot, 3.0, grid);
            }
            double* row = c->t + (size_t)R * t_num;
            if (has_early) {
                row[0] = early_t_row[R];
                for (size_t k = 0; k < t_num_tot; ++k) row[1 + k] = grid[k];
            } else {
                for (size_t k = 0; k < t_num_tot; ++k) row[k] = grid[k];
            }
            continue;
        }
        if (is_rvs) { /* make_time_grid, grid-refinement.h:571-581 */
            const double t_cross_limit = dmax(t_dec[j_rep], jet->T0);
            logspace_with_cross_refinement(min_t_start, t_end, 10 * t_cross_limit, t_num_tot, t_num_base, grid);
        } else {
            logspace_with_band_refinement(min_t_start, t_end, t_dec[j_rep] / 3, 3 * t_dec[j_rep], t_num_tot, 3.0, grid);
        }
        for (int j = j_rep; j < j_end; ++j) {
            double* row = c->t + (size_t)j * t_num;
            if (has_early) {
                row[0] = min_t_early;
                for (size_t k = 0; k < t_num_tot; ++k) row[1 + k] = grid[k];
            } else {
                for (size_t k = 0; k < t_num_tot; ++k) row[k] = grid[k];
            }
        }
    }
    free(grid);
    free(t_dec);
    free(t_start_row);
    free(early_t_row);
}

/* auto_grid, src/core/grid-refinement.h:639-706 */
static int auto_grid(coord_t* c, const jet_t* jet, const medium_t* med, double t_obs_min, double t_obs_max,
                     double theta_cut, double theta_view, double z, double phi_resol, double theta_resol, double t_resol,
                     int is_rvs, int is_axisymmetric) {
    memset(c, 0, sizeof *c);
    c->theta_view = theta_view;
    const size_t min_theta_num = DEF_MIN_THETA_POINTS;
    double jumps[64];
    const int n_jumps = find_jet_jumps(jet, C_GAMMA_CUT, jumps, 64);
    double inner_edge, outer_edge;
    find_theta_range(jet, C_GAMMA_CUT, &inner_edge, &outer_edge);
    for (int i = 0; i < n_jumps; ++i) outer_edge = dmax(outer_edge, jumps[i]);
    const double theta_min = dmax(DEF_THETA_MIN, inner_edge);
    const double theta_max = dmin(outer_edge, theta_cut);
    const size_t theta_num = min_theta_num + (size_t)((theta_max - theta_min) * 180 / C_PI * theta_resol);

    int n_base = 0;
    double* base_theta = adaptive_theta_grid(jet, theta_min, theta_max, theta_num, theta_view, theta_resol, &n_base);
    const double avg_spacing = (theta_max - theta_min) / n_base;
    double feature[3 * 64];
    const int n_feat = jump_refinement_grid(jumps, n_jumps, theta_min, theta_max, avg_spacing, feature);
    c->theta = malloc(sizeof(double) * (n_base + n_feat + 1));
    c->n_theta = merge_grids(base_theta, n_base, feature, n_feat, c->theta);
    free(base_theta);

    size_t phi_base = (size_t)(360 * phi_resol);
    if (phi_base < 1) phi_base = 1;
    const int mirror_phi = is_axisymmetric && theta_view != 0 && phi_base > 4;
    if (mirror_phi) {
        const size_t n_half = (phi_base + 1) / 2;
        c->phi = adaptive_phi_grid(jet, n_half, theta_view, c->theta, c->n_theta, is_axisymmetric, C_PI, 5.0, &c->n_phi);
        c->phi_mirrored = 1;
    } else {
        const double doppler_sharpness = jet_Gamma0(jet, theta_view) * sin(theta_view);
        const double phi_boost = sqrt(dmax(doppler_sharpness / (2 * C_PI), 1.0));
        size_t phi_num = (size_t)(phi_base * phi_boost);
        if (phi_num < 1) phi_num = 1;
        if (phi_num > phi_base * 5) phi_num = phi_base * 5;
        if (phi_num <= 2) {
            c->phi = malloc(sizeof(double) * phi_num);
            linspace(0., 2 * C_PI, (int)phi_num, c->phi);
            c->n_phi = (int)phi_num;
        } else {
            c->phi = adaptive_phi_grid(jet, phi_num, theta_view, c->theta, c->n_theta, is_axisymmetric, 2 * C_PI, 0, &c->n_phi);
        }
        if (phi_num >= 2) {
            const double shift = 0.5 * (c->phi[1] - c->phi[0]);
            for (int i = 0; i < c->n_phi; ++i) c->phi[i] += shift;
        }
    }
    c->jet_3d = !is_axisymmetric && c->n_phi > 1;
    c->phi_size = (!is_axisymmetric && jet->spreading) ? c->n_phi : 1;
    detect_symmetry(c, jet);
    build_time_grid(c, jet, med, t_obs_min, t_obs_max, z, t_resol, is_rvs, is_axisymmetric);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Forward-shock dynamics: src/dynamics/forward-shock.tpp:10-236, shock-physics.h:58-132,
 * 247-288,299-371,401-469; src/dynamics/shock.cpp:93-138
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const medium_t* med;
    double m_jet0;
    /* RadiativeEfficiency, shock-physics.h:247-288 */
    double gamma_m_coeff, gamma_c_coeff, eps_e_eff, p;
    double eps_e, eps_B;
    int radiative;
    const jet_t* jet;
    double theta0;
    double theta_s; /* jet_spreading_edge, grid-refinement.h:113-135 */
} fwd_eqn_t;

static double radiative_efficiency(const fwd_eqn_t* e, double t_comv, double Gamma_th, double e_th) {
    if (e->eps_e_eff == 0) return 0;
    const double gamma_m = e->gamma_m_coeff * (Gamma_th - 1) + 1;
    const double gamma_bar = e->gamma_c_coeff / (e_th * t_comv);
    const double gamma_c = 0.5 * (gamma_bar + sqrt(gamma_bar * gamma_bar + 4));
    const double ratio = gamma_m / gamma_c;
    if (ratio < 1 && e->p > 2) return e->eps_e_eff * fast_pow(ratio, e->p - 2);
    return e->eps_e_eff;
}

/* state: [Gamma, m2, U2_th, r, t_comv, theta]  (forward-shock.hpp:22-47) */
static void fwd_rhs(const double* s, double* d, double t, void* vctx) {
    const fwd_eqn_t* e = vctx;
    /* state[6] = eps_jet (ForwardState::energy_inject, forward-shock.hpp:36-44): only its derivative enters the
     * dynamics, but the variable takes part in the step-size control */
    const double deps_jet = jet_is_ejecta(e->jet) ? jet_deps_dt(e->jet, e->theta0, t) : 0.0;
    d[6] = deps_jet;
    const double Gamma = s[0], m2 = s[1], U2_th = s[2], r = s[3], t_comv = s[4];
    const double u2 = (Gamma - 1) * (Gamma + 1);
    const double u = sqrt(u2);
    const double dr = u * (Gamma + u) * C_C; /* compute_dr_dt(Gamma,u), shock-physics.h:130-132 */
    d[3] = dr;
    d[4] = Gamma + u;
    const int spreading = e->jet->spreading;
    if (spreading && s[5] < 0.5 * C_PI) { /* compute_dtheta_dt, shock-physics.h:141-145 */
        const double Q = 7;
        const double f = 1 / (1 + u * e->theta_s * Q);
        d[5] = dr / (2 * Gamma * r) * sqrt((2 * u2 + 3) / (4 * u2 + 3)) * f;
    } else {
        d[5] = 0;
    }
    double sin_theta = 0, cos_theta = 1;
    if (spreading) {
        sin_theta = sin(s[5]);
        cos_theta = cos(s[5]);
    }
    const double rho = medium_rho(e->med, r);
    d[1] = r * r * rho * dr;
    const double e_th = (Gamma - 1) * 4 * Gamma * rho * C_C2;
    const double eps_rad = radiative_efficiency(e, t_comv, Gamma, e_th);
    const double ad_idx = adiabatic_idx(Gamma);
    /* compute_dGamma_dt, forward-shock.tpp:62-101 */
    {
        double dm_dt_swept = d[1];
        double m_swept = m2;
        const double Gamma2 = Gamma * Gamma;
        const double Gamma_eff = (ad_idx * (Gamma2 - 1) + 1) / Gamma;
        const double dGamma_eff = (ad_idx * (Gamma2 + 1) - 1) / Gamma2;
        double dlnVdt = 3 / r * dr;
        const double m_jet = e->m_jet0;
        double U = U2_th;
        if (spreading) {
            const double dOmega0 = 1 - cos(e->theta0);
            const double f_spread = (1 - cos_theta) / dOmega0;
            dm_dt_swept = dm_dt_swept * f_spread + m_swept / dOmega0 * sin_theta * d[5];
            m_swept *= f_spread;
            dlnVdt += sin_theta / (1 - cos_theta) * d[5];
            U *= f_spread;
        }
        double a1 = -(Gamma - 1) * (Gamma_eff + 1) * C_C2 * dm_dt_swept;
        const double a2 = (ad_idx - 1) * Gamma_eff * U * dlnVdt;
        if (jet_is_ejecta(e->jet)) a1 += deps_jet; /* energy_inject, forward-shock.tpp:89-91 */
        const double b1 = (m_jet + m_swept) * C_C2;
        const double b2 = (dGamma_eff + Gamma_eff * (ad_idx - 1) / Gamma) * U;
        d[0] = (a1 + a2) / (b1 + b2);
    }
    /* compute_dU_dt, forward-shock.tpp:103-118 */
    {
        double dm_dt_swept = d[1];
        double dlnVdt = 3 / r * dr - d[0] / Gamma;
        if (spreading) {
            const double factor = sin_theta / (1 - cos_theta) * d[5];
            dm_dt_swept = dm_dt_swept + m2 * factor;
            dlnVdt += factor;
            dlnVdt += factor / (ad_idx - 1);
        }
        d[2] = (1 - eps_rad) * (Gamma - 1) * C_C2 * dm_dt_swept - (ad_idx - 1) * dlnVdt * U2_th;
    }
}

/* simpson_logspace / enclosed_thermal_energy, shock-physics.h:401-437 */
static double enclosed_thermal_energy_generic(const medium_t* med, double r, double Gamma, double ad_idx, double eps_e) {
    const double cooling_exp = 3 * (ad_idx - 1);
    const int N = 32;
    const double u_max = log(r);
    const double u_min = u_max - 18;
    const double h = (u_max - u_min) / N;
#define F_(u_) (medium_rho(med, exp(u_)) * exp(u_) * exp(u_) * exp(u_) * pow(exp(u_) / r, cooling_exp))
    double sum = F_(u_min) + F_(u_max);
    for (int i = 1; i < N; i += 2) sum += 4 * F_(u_min + i * h);
    for (int i = 2; i < N; i += 2) sum += 2 * F_(u_min + i * h);
#undef F_
    return (1 - eps_e) * (Gamma - 1) * C_C2 * (sum * h / 3);
}

/* enclosed_thermal_energy_medium, shock-physics.h:451-468 */
static double enclosed_thermal_energy_medium(const medium_t* med, double r, double Gamma, double ad_idx, double eps_e) {
    if (med->type == VAG_MEDIUM_ISM) {
        const double rho = medium_rho(med, r);
        const double cooling_exp = 3 * (ad_idx - 1);
        const double pow_exp = 3 + cooling_exp;
        const double x0 = exp(-18.0);
        const double attenuation = 1 - pow(x0, pow_exp);
        const double integral = rho * r * r * r * attenuation / pow_exp;
        return (1 - eps_e) * (Gamma - 1) * C_C2 * integral;
    }
    return enclosed_thermal_energy_generic(med, r, Gamma, ad_idx, eps_e);
}

/* set_init_state, forward-shock.tpp:120-149 */
static void fwd_set_init_state(const fwd_eqn_t* e, double* s, double t0) {
    const double Gamma4 = jet_Gamma0(e->jet, e->theta0);
    const double beta4 = gamma_to_beta(Gamma4);
    s[3] = beta4 * C_C * t0 * Gamma4 * Gamma4 * (1 + beta4);
    s[4] = s[3] / sqrt((Gamma4 - 1) * (Gamma4 + 1)) / C_C;
    s[5] = e->theta0;
    s[6] = jet_eps_k(e->jet, e->theta0); /* state.eps_jet, forward-shock.tpp:137-139 */
    s[1] = medium_mass(e->med, s[3]);
    s[0] = Gamma4;
    const double ad_idx = adiabatic_idx(s[0]);
    s[2] = enclosed_thermal_energy_medium(e->med, s[3], s[0], ad_idx, e->radiative ? e->eps_e : 0.0);
}

/* compute_downstr_4vel (sigma = 0 branch), src/dynamics/shock.cpp:93-100 */
static double compute_downstr_4vel0(double gamma_rel) {
    const double ad_idx = adiabatic_idx(gamma_rel);
    const double gamma_m_1 = gamma_rel - 1;
    const double ad_idx_m_2 = ad_idx - 2;
    const double ad_idx_m_1 = ad_idx - 1;
    return sqrt(dmax(gamma_m_1 * ad_idx_m_1 * ad_idx_m_1 / (-ad_idx * ad_idx_m_2 * gamma_m_1 + 2), 0.0));
}

/* compute_compression(1, Gamma, 0): shock-physics.h:58-66,190-203,349-352 */
static double compute_compression_fwd(double Gamma_downstr) {
    const double gamma1 = 1;
    const double u1u2 = sqrt(dmax((gamma1 - 1) * (gamma1 + 1) * (Gamma_downstr - 1) * (Gamma_downstr + 1), 0.0));
    const double dd = gamma1 - Gamma_downstr;
    const double denom = gamma1 * Gamma_downstr - 1 + u1u2;
    const double gamma_rel = denom <= 0 ? 1 : 1 + dd * dd / denom;
    const double u_down = compute_downstr_4vel0(gamma_rel);
    const double u_up = sqrt((1 + u_down * u_down) * dmax((gamma_rel - 1) * (gamma_rel + 1), 0.0)) + u_down * gamma_rel;
    double ratio_u = u_up / u_down;
    if (u_down == 0.) ratio_u = 4 * gamma_rel;
    return ratio_u;
}

typedef struct {
    int n_theta, n_t;
    double *t_comv, *r, *theta, *Gamma, *Gamma_th, *B, *N_p; /* [n_theta][n_t] */
    int* injection_idx;                                      /* [n_theta]; n_t = always injecting (shock.h:50-56) */
} shock_t;

static void shock_alloc(shock_t* s, int nth, int nt) {
    const size_t n = (size_t)nth * nt;
    s->n_theta = nth;
    s->n_t = nt;
    s->t_comv = calloc(n, sizeof(double));
    s->r = calloc(n, sizeof(double));
    s->theta = calloc(n, sizeof(double));
    s->Gamma = malloc(n * sizeof(double));
    s->Gamma_th = malloc(n * sizeof(double));
    s->B = calloc(n, sizeof(double));
    s->N_p = calloc(n, sizeof(double));
    for (size_t i = 0; i < n; ++i) s->Gamma[i] = s->Gamma_th[i] = 1; /* Shock ctor, shock.cpp:12-24 */
    s->injection_idx = malloc(sizeof(int) * nth);
    for (int j = 0; j < nth; ++j) s->injection_idx[j] = nt;
}
static void shock_free(shock_t* s) {
    free(s->t_comv);
    free(s->r);
    free(s->theta);
    free(s->Gamma);
    free(s->Gamma_th);
    free(s->B);
    free(s->N_p);
    free(s->injection_idx);
    memset(s, 0, sizeof *s);
}

/* save_fwd_shock_state, forward-shock.tpp:151-173 */
static void save_fwd_shock_state(shock_t* sh, size_t o, const fwd_eqn_t* e, const double* s) {
    const double comp_ratio = compute_compression_fwd(s[0]);
    const double rho = medium_rho(e->med, s[3]);
    const double U_th = s[2];
    const double Gamma_th = (s[1] == 0) ? 1 : U_th / (s[1] * C_C2) + 1;
    const double rho_downstr = rho * comp_ratio;
    const double e_th = (Gamma_th - 1) * rho_downstr * C_C2;
    const double B = sqrt(8 * C_PI * e->eps_B * e_th) + 0 * comp_ratio;
    sh->t_comv[o] = s[4];
    sh->r[o] = s[3];
    sh->theta[o] = s[5];
    sh->Gamma[o] = s[0];
    sh->Gamma_th[o] = Gamma_th;
    sh->B[o] = B;
    sh->N_p[o] = s[1] / C_MP;
}

/* grid_solve_fwd_shock, forward-shock.tpp:175-208 */
static int grid_solve_fwd_shock(int j, const double* t, int nt, shock_t* sh, const fwd_eqn_t* e, double rtol) {
    double state[7];
    const double t_dec = estimate_t_dec(e->jet, e->med, e->theta0);
    const double t0 = dmin(t[0], dmin(0.1 * U_SEC, 0.1 * t_dec));
    fwd_set_init_state(e, state, t0);
    const size_t base = (size_t)j * nt;
    if (state[0] <= C_GAMMA_CUT) { /* set_stopping_shock, shock-physics.h:388-397 */
        for (int k = 0; k < nt; ++k) {
            sh->t_comv[base + k] = state[4];
            sh->r[base + k] = state[3];
            sh->theta[base + k] = state[5];
            sh->Gamma[base + k] = 1;
            sh->Gamma_th[base + k] = 1;
            sh->B[base + k] = 0;
            sh->N_p[base + k] = 0;
        }
        return 0;
    }
    dopri5_t st;
    dopri5_init(&st, 7, rtol, rtol, state, t0, 0.01 * t0);
    for (int k = 0, steps = 0; st.t <= t[nt - 1];) {
        if (dopri5_do_step(&st, fwd_rhs, (void*)e) != 0) return fail("forward shock ODE: step size underflow");
        if (++steps > DEF_MAX_ODE_STEPS) {
            fprintf(stderr, "Warning: forward shock ODE exceeded %d steps at (j=%d), giving up\n", DEF_MAX_ODE_STEPS, j);
            return 0;
        }
        while (k < nt && st.t > t[k]) {
            dopri5_calc_state(&st, t[k], state);
            save_fwd_shock_state(sh, base + k, e, state);
            ++k;
        }
    }
    return 0;
}

/* generate_fwd_shock + Shock::broadcast_groups: forward-shock.tpp:210-236, shock.cpp:42-91 */
/* jet_spreading_edge, grid-refinement.h:113-135 */
static double jet_spreading_edge(const jet_t* jet, double theta_min, double theta_max) {
    const double step = (theta_max - theta_min) / 256;
    double theta_s = theta_min;
    double dp_min = 0;
    for (double theta = theta_min; theta <= theta_max; theta += step) {
        const double th_lo = dmax(theta - step, theta_min);
        const double th_hi = dmin(theta + step, theta_max);
        const double dp = (jet_Gamma0(jet, th_hi) - jet_Gamma0(jet, th_lo)) / (th_hi - th_lo);
        if (dp < dp_min) {
            dp_min = dp;
            theta_s = theta;
        }
    }
    if (dp_min == 0) theta_s = theta_max;
    return theta_s;
}

static int generate_fwd_shock(shock_t* sh, const coord_t* c, const medium_t* med, const jet_t* jet,
                              const vag_model_params* p) {
    const int n_rows = c->phi_size * c->n_theta;
    shock_alloc(sh, n_rows, c->n_t);
    const double theta_s = jet->spreading ? jet_spreading_edge(jet, c->theta[0], c->theta[c->n_theta - 1]) : 0;
    for (int r = 0; r < c->n_reps; ++r) {
        const int j = c->reps[r]; /* row */
        fwd_eqn_t e;
        e.med = med;
        e.jet = jet;
        e.theta0 = c->theta[j % c->n_theta];
        e.theta_s = theta_s;
        e.m_jet0 = jet_eps_k(jet, e.theta0) / jet_Gamma0(jet, e.theta0) / C_C2;
        if (jet_is_ejecta(jet)) e.m_jet0 /= 1 + jet->sigma0;
        e.radiative = p->radiative_fireball != 0;
        e.eps_e = p->eps_e;
        e.eps_B = p->eps_B;
        e.p = p->p;
        e.gamma_m_coeff = (p->p - 2) / (p->p - 1) * p->eps_e * C_MP / C_ME / p->xi_e;
        e.gamma_c_coeff = 6 * C_PI * C_ME * C_C / C_SIGMAT / (8 * C_PI * p->eps_B);
        e.eps_e_eff = e.radiative ? p->eps_e : 0;
        if (grid_solve_fwd_shock(j, c->t + (size_t)j * c->n_t, c->n_t, sh, &e, p->rtol) != 0) return -1;
    }
    /* broadcast representative rows to their groups; theta(j,k) = coord.theta(j) */
    const int nt = c->n_t;
    for (int r = 0; r < c->n_reps; ++r) {
        const int j0 = c->reps[r];
        const int j1 = (r + 1 < c->n_reps) ? c->reps[r + 1] : n_rows;
        for (int j = j0 + 1; j < j1; ++j) {
            for (int k = 0; k < nt; ++k) {
                const size_t o = (size_t)j * nt + k, s = (size_t)j0 * nt + k;
                sh->t_comv[o] = sh->t_comv[s];
                sh->r[o] = sh->r[s];
                sh->theta[o] = c->theta[j % c->n_theta];
                sh->Gamma[o] = sh->Gamma[s];
                sh->Gamma_th[o] = sh->Gamma_th[s];
                sh->B[o] = sh->B[s];
                sh->N_p[o] = sh->N_p[s];
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Coupled forward + reverse shock: src/dynamics/reverse-shock.hpp:22-60, reverse-shock.tpp:11-614,
 * shock-physics.h:40-245,401-437, src/dynamics/shock.cpp:93-137
 * state: [Gamma, x4, x3, m2, m3, U2_th, U3_th, r, t_comv, theta, eps4, m4]
 * ---------------------------------------------------------------------------------------- */
enum { RS_GAMMA = 0, RS_X4, RS_X3, RS_M2, RS_M3, RS_U2, RS_U3, RS_R, RS_TCOMV, RS_THETA, RS_EPS4, RS_M4, RS_N };
#define C_SIGMA_CUT 1e-6

typedef struct {
    const medium_t* med;
    const jet_t* jet;
    double theta0, Gamma4, deps0_dt, dm0_dt, u4, T0;
    fwd_eqn_t rad_fwd;          /* RadiativeEfficiency(rad_fwd) + eps_e / eps_B / radiative of the forward shock */
    double eps_B_rvs;
    double u_x, r_x, B3_ordered_x, V3_comv_x, rho3_x; /* save_cross_state */
} rvs_eqn_t;

static double smoothstep(double edge0, double edge1, double x) {
    double t = (x - edge0) / (edge1 - edge0);
    if (t < 0.0)
        t = 0.0;
    else if (t > 1.0)
        t = 1.0;
    return t * t * (3.0 - 2.0 * t);
}

/* compute_downstr_4vel, src/dynamics/shock.cpp:93-137 */
static double compute_downstr_4vel(double gamma_rel, double sigma) {
    const double ad_idx = adiabatic_idx(gamma_rel);
    const double gamma_m_1 = gamma_rel - 1;
    const double ad_idx_m_2 = ad_idx - 2;
    const double ad_idx_m_1 = ad_idx - 1;
    if (sigma <= C_SIGMA_CUT)
        return sqrt(dmax(gamma_m_1 * ad_idx_m_1 * ad_idx_m_1 / (-ad_idx * ad_idx_m_2 * gamma_m_1 + 2), 0.0));
    const double gamma_sq = gamma_rel * gamma_rel;
    const double gamma_p_1 = gamma_rel + 1;
    const double term1 = -ad_idx * ad_idx_m_2;
    const double term2 = gamma_sq - 1;
    const double A = term1 * gamma_m_1 + 2;
    const double B = -gamma_p_1 * (-ad_idx_m_2 * (ad_idx * gamma_sq + 1) + ad_idx * ad_idx_m_1 * gamma_rel) * sigma -
                     gamma_m_1 * (term1 * (gamma_sq - 2) + 2 * gamma_rel + 3);
    const double Cc = gamma_p_1 * (ad_idx * (1 - ad_idx / 4) * term2 + 1) * sigma * sigma +
                      term2 * (2 * gamma_rel + ad_idx_m_2 * (ad_idx * gamma_rel - 1)) * sigma +
                      gamma_p_1 * gamma_m_1 * gamma_m_1 * ad_idx_m_1 * ad_idx_m_1;
    const double D = -gamma_m_1 * gamma_p_1 * gamma_p_1 * ad_idx_m_2 * ad_idx_m_2 * sigma * sigma / 4;
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

/* compute_4vel_jump, shock-physics.h:58-66 */
static double compute_4vel_jump(double gamma_rel, double sigma_upstr) {
    const double u_down = compute_downstr_4vel(gamma_rel, sigma_upstr);
    const double u_up = sqrt((1 + u_down * u_down) * dmax((gamma_rel - 1) * (gamma_rel + 1), 0.0)) + u_down * gamma_rel;
    double ratio_u = u_up / u_down;
    if (u_down == 0.) ratio_u = 4 * gamma_rel;
    return ratio_u;
}

/* compute_rel_Gamma, shock-physics.h:190-198 */
static double compute_rel_Gamma(double gamma1, double gamma2) {
    const double u1u2 = sqrt(dmax((gamma1 - 1) * (gamma1 + 1) * (gamma2 - 1) * (gamma2 + 1), 0.0));
    const double d = gamma1 - gamma2;
    const double denom = gamma1 * gamma2 - 1 + u1u2;
    if (denom <= 0) return 1;
    return 1 + d * d / denom;
}

static double compute_compression(double Gamma_upstr, double Gamma_downstr, double sigma_upstr) {
    return compute_4vel_jump(compute_rel_Gamma(Gamma_upstr, Gamma_downstr), sigma_upstr);
}

/* compute_sound_speed, shock-physics.h:77-80 */
static double compute_sound_speed(double Gamma_rel) {
    const double ad_idx = adiabatic_idx(Gamma_rel);
    return sqrt(dmax(ad_idx * (ad_idx - 1) * (Gamma_rel - 1) / (1 + (Gamma_rel - 1) * ad_idx), 0.0)) * C_C;
}

static double compute_upstr_B(double rho_up, double sigma) {
    return sqrt((4 * C_PI * C_C2) * sigma * rho_up);
}

/* compute_downstr_B, shock-physics.h:354-360 */
static double compute_downstr_B(double eps_B, double rho_upstr, double B_upstr, double Gamma_th, double comp_ratio) {
    const double rho_downstr = rho_upstr * comp_ratio;
    const double e_th = (Gamma_th - 1) * rho_downstr * C_C2;
    return sqrt(8 * C_PI * eps_B * e_th) + B_upstr * comp_ratio;
}

/* compute_Gamma_therm, shock-physics.h:290-301 */
static double compute_Gamma_therm(double U_th, double mass, int limiter) {
    if (mass == 0) return 1;
    const double Gamma_th = U_th / (mass * C_C2) + 1;
    if (limiter && Gamma_th < C_GAMMA_CUT) return 1; /* con::gamma_therm_cut == 1 + 1e-6 */
    return Gamma_th;
}

static double rvs_shell_sigma(const rvs_eqn_t* e, const double* s) {
    const double sigma = s[RS_EPS4] / (e->Gamma4 * s[RS_M4] * C_C2) - 1;
    return (sigma > C_SIGMA_CUT) ? sigma : 0;
}

static double rvs_injection_efficiency(const rvs_eqn_t* e, const double* d) {
    if (e->dm0_dt > 0 && d[RS_M4] > 0) return dmin(d[RS_M4] / e->dm0_dt, 1.0);
    return 0.0;
}

/* FRShockEqn::crossing_complete, reverse-shock.tpp:46-58 (named jets: no mass injection) */
static int rvs_crossing_complete(const rvs_eqn_t* e, const double* s, double t) {
    if (s[RS_M3] < 0.999 * s[RS_M4]) return 0;
    if (smoothstep(e->T0 * 1.5, e->T0 * 0.5, t) > 1e-6) return 0;
    return 1;
}

/* FRShockEqn::operator(), reverse-shock.tpp:251-293 with the compute_* members of :60-249 */
static void rvs_rhs(const double* raw, double* d, double t, void* vctx) {
    const rvs_eqn_t* e = vctx;
    double s[RS_N];
    for (int i = 0; i < RS_N; ++i) s[i] = raw[i];
    s[RS_GAMMA] = dmin(dmax(s[RS_GAMMA], 1.0), e->Gamma4);
    s[RS_M3] = dmin(dmax(s[RS_M3], 0.0), dmax(s[RS_M4], 0.0));
    s[RS_X3] = dmax(s[RS_X3], 0.0);
    s[RS_U3] = dmax(s[RS_U3], 0.0);
    const double Gamma = s[RS_GAMMA], Gamma4 = e->Gamma4;
    const double u3 = sqrt((Gamma - 1) * (Gamma + 1));
    d[RS_R] = u3 * (Gamma + u3) * C_C;
    d[RS_TCOMV] = Gamma + u3;
    const double rho = medium_rho(e->med, s[RS_R]);
    d[RS_M2] = s[RS_R] * s[RS_R] * rho * d[RS_R];
    const double inject_w = smoothstep(e->T0 * 1.5, e->T0 * 0.5, t);
    d[RS_EPS4] = (inject_w > 1e-6) ? inject_w * e->deps0_dt : 0;
    d[RS_M4] = (inject_w > 1e-6) ? inject_w * e->dm0_dt : 0;
    if (jet_is_ejecta(e->jet)) { /* python-level Ejecta: deps_dt = magnetar injection or zero, dm_dt the zero function */
        d[RS_EPS4] += jet_deps_dt(e->jet, e->theta0, t);
        d[RS_M4] += 0.0;
    }
    const double Gamma34 = compute_rel_Gamma(Gamma4, Gamma);
    const double sigma = rvs_shell_sigma(e, s);
    const double comp_ratio = compute_4vel_jump(Gamma34, sigma);
    const double f = rvs_injection_efficiency(e, d);
    { /* compute_dx4_dt */
        const double sound_expansion = compute_sound_speed(Gamma4) * d[RS_TCOMV];
        d[RS_X4] = (f > 1e-6) ? f * e->u4 + (1 - f) * sound_expansion : sound_expansion;
    }
    { /* compute_dx3_dt */
        const double sound_expansion = compute_sound_speed(Gamma34) * d[RS_TCOMV];
        double dx3 = sound_expansion;
        if (!(s[RS_M4] <= 0)) {
            const double remaining = dmax(s[RS_M4] - s[RS_M3], 0.0);
            const double crossing_w = f + (1.0 - f) * remaining / s[RS_M4];
            const double penetration = Gamma * comp_ratio / Gamma4 - 1;
            if (!(crossing_w < 1e-6) && !(penetration <= 0)) {
                const double beta3 = gamma_to_beta(Gamma);
                const double beta4 = gamma_to_beta(Gamma4);
                const double dx3dt = (Gamma4 - Gamma) * (Gamma4 + Gamma) * (1 + beta3) * C_C /
                                     (Gamma4 * Gamma4 * (beta3 + beta4) * penetration);
                double crossing = fabs(dx3dt * Gamma);
                if (penetration < 1) {
                    const double cs = compute_sound_speed(Gamma34);
                    const double va2 = sigma / (1 + sigma);
                    const double cs2 = cs * cs / (C_C * C_C);
                    const double v_ms = sqrt(va2 + cs2 * (1 - va2)) * C_C;
                    crossing = dmin(crossing, v_ms * d[RS_TCOMV]);
                }
                dx3 = crossing_w * crossing + (1.0 - crossing_w) * sound_expansion;
            }
        }
        d[RS_X3] = dx3;
    }
    { /* compute_dm3_dt */
        double dm3 = 0.;
        if (!(s[RS_M4] <= 0)) {
            const double remaining = dmax(s[RS_M4] - s[RS_M3], 0.0);
            if (!(remaining <= 0 && f < 1e-6)) {
                const double eff_mass = f * s[RS_M4] + (1.0 - f) * remaining;
                const double column_den3 = eff_mass * comp_ratio / s[RS_X4];
                const double dm3dt = column_den3 * d[RS_X3];
                if (f > 1e-6) {
                    const double ratio = s[RS_M3] / s[RS_M4];
                    const double cap_w = smoothstep(0, 1.0, ratio);
                    const double capped_rate = dmin(dm3dt, d[RS_M4]);
                    dm3 = (1.0 - cap_w) * dm3dt + cap_w * capped_rate;
                } else {
                    dm3 = dm3dt;
                }
            }
        }
        d[RS_M3] = dm3;
    }
    { /* compute_dU2_dt */
        const double e_th = (Gamma - 1) * 4 * Gamma * rho * C_C2;
        const double eps_rad = radiative_efficiency(&e->rad_fwd, s[RS_TCOMV], Gamma, e_th);
        const double ad_idx = adiabatic_idx(Gamma);
        const double shock_heating = d[RS_M2] * (Gamma - 1) * C_C2;
        double dlnvdt = 2 * d[RS_R] / s[RS_R];
        if (s[RS_X4] > 0) dlnvdt += d[RS_X4] / s[RS_X4];
        const double adiabatic_cooling = -(ad_idx - 1) * dlnvdt * s[RS_U2];
        d[RS_U2] = (1 - eps_rad) * shock_heating + adiabatic_cooling;
    }
    { /* compute_dU3_dt */
        const double ad_idx = adiabatic_idx(Gamma34);
        double dlnvdt = 2 * d[RS_R] / s[RS_R];
        if (s[RS_X3] > 0) dlnvdt += d[RS_X3] / s[RS_X3];
        const double adiabatic_cooling = -(ad_idx - 1) * dlnvdt * s[RS_U3];
        const double shock_heating = d[RS_M3] * (Gamma34 - 1) * C_C2;
        d[RS_U3] = shock_heating + adiabatic_cooling;
    }
    { /* compute_dGamma_dt */
        const double ad_idx2 = adiabatic_idx(Gamma);
        const double ad_idx3 = adiabatic_idx(Gamma34);
        const double Gamma_eff2 = (ad_idx2 * Gamma * Gamma - ad_idx2 + 1) / Gamma;
        const double Gamma_eff3 = (ad_idx3 * Gamma * Gamma - ad_idx3 + 1) / Gamma;
        const double Gamma2 = Gamma * Gamma;
        const double dGamma_eff2_dGamma = (ad_idx2 * Gamma2 + ad_idx2 - 1) / Gamma2;
        const double dGamma_eff3_dGamma = (ad_idx3 * Gamma2 + ad_idx3 - 1) / Gamma2;
        double deps_dt = 0;
        if (jet_is_ejecta(e->jet)) deps_dt = jet_deps_dt(e->jet, s[RS_THETA], t);
        const double a = (Gamma - 1) * C_C2 * d[RS_M2] + (Gamma - Gamma4) * C_C2 * d[RS_M3] + Gamma_eff2 * d[RS_U2] +
                         Gamma_eff3 * d[RS_U3] - deps_dt;
        const double b = (s[RS_M2] + s[RS_M3]) * C_C2 + dGamma_eff2_dGamma * s[RS_U2] + dGamma_eff3_dGamma * s[RS_U3];
        if (b == 0 || isnan(-a / b) || isinf(-a / b))
            d[RS_GAMMA] = 0;
        else
            d[RS_GAMMA] = -a / b;
    }
    d[RS_THETA] = 0;
}

/* simpson_logspace / enclosed_mass, shock-physics.h:401-425 */
static double enclosed_mass_generic(const medium_t* med, double r) {
    const int N = 32;
    const double u_max = log(r);
    const double u_min = u_max - 18;
    const double h = (u_max - u_min) / N;
#define F_(u_) (medium_rho(med, exp(u_)) * exp(u_) * exp(u_) * exp(u_))
    double sum = F_(u_min) + F_(u_max);
    for (int i = 1; i < N; i += 2) sum += 4 * F_(u_min + i * h);
    for (int i = 2; i < N; i += 2) sum += 2 * F_(u_min + i * h);
#undef F_
    return sum * h / 3;
}

/* compute_init_comv_shell_width, reverse-shock.tpp:367-376 */
static double compute_init_comv_shell_width(double Gamma4, double t0, double T) {
    const double beta4 = gamma_to_beta(Gamma4);
    if (t0 < T) return Gamma4 * t0 * beta4 * C_C;
    const double cs = compute_sound_speed(Gamma4);
    return Gamma4 * T * beta4 * C_C + cs * (t0 - T) * Gamma4;
}

/* FRShockEqn::set_init_state, reverse-shock.tpp:312-354 */
static void rvs_set_init_state(const rvs_eqn_t* e, double* s, double t0) {
    const double Gamma4 = e->Gamma4;
    const double beta4 = gamma_to_beta(Gamma4);
    s[RS_R] = beta4 * C_C * t0 * Gamma4 * Gamma4 * (1 + beta4);
    s[RS_TCOMV] = s[RS_R] / sqrt((Gamma4 - 1) * (Gamma4 + 1)) / C_C;
    s[RS_THETA] = e->theta0;
    const double dt = dmin(t0, e->T0);
    s[RS_EPS4] = e->deps0_dt * dt;
    s[RS_M4] = e->dm0_dt * dt;
    s[RS_X4] = compute_init_comv_shell_width(Gamma4, t0, e->T0);
    s[RS_M2] = enclosed_mass_generic(e->med, s[RS_R]);
    const double m_jet_total = e->dm0_dt * e->T0;
    if (m_jet_total > 0 && s[RS_M2] > 0)
        s[RS_GAMMA] = Gamma4 / (1 + s[RS_M2] / m_jet_total);
    else
        s[RS_GAMMA] = Gamma4;
    const double ad_idx = adiabatic_idx(s[RS_GAMMA]);
    s[RS_U2] = enclosed_thermal_energy_generic(e->med, s[RS_R], s[RS_GAMMA], ad_idx,
                                               e->rad_fwd.radiative ? e->rad_fwd.eps_e : 0.0);
    const double Gamma34 = compute_rel_Gamma(Gamma4, s[RS_GAMMA]);
    if (Gamma34 > 1 && s[RS_M4] > 0 && s[RS_X4] > 0) {
        const double seed_frac = 1e-8;
        const double sigma = rvs_shell_sigma(e, s);
        const double comp_ratio = compute_4vel_jump(Gamma34, sigma);
        s[RS_X3] = s[RS_X4] * seed_frac;
        s[RS_M3] = s[RS_M4] * comp_ratio * s[RS_X3] / s[RS_X4];
        s[RS_U3] = (Gamma34 - 1) * s[RS_M3] * C_C2;
    } else {
        s[RS_M3] = 0;
        s[RS_U3] = 0;
        s[RS_X3] = 0;
    }
}

/* FRShockEqn::save_cross_state, reverse-shock.tpp:297-310 */
static void rvs_save_cross_state(rvs_eqn_t* e, const double* s) {
    e->r_x = s[RS_R];
    e->u_x = sqrt((s[RS_GAMMA] - 1) * (s[RS_GAMMA] + 1));
    e->V3_comv_x = e->r_x * e->r_x * s[RS_X3];
    const double sigma4 = rvs_shell_sigma(e, s);
    const double comp_ratio34 = compute_compression(e->Gamma4, s[RS_GAMMA], sigma4);
    const double rho4 = s[RS_M4] / (s[RS_R] * s[RS_R] * s[RS_X4]);
    e->rho3_x = rho4 * comp_ratio34;
    const double B4 = compute_upstr_B(rho4, sigma4);
    e->B3_ordered_x = B4 * comp_ratio34;
}

static void write_shock_state(shock_t* sh, size_t o, double t_comv, double r, double theta, double Gamma, double Gamma_th,
                              double B, double mass) {
    sh->t_comv[o] = t_comv;
    sh->r[o] = r;
    sh->theta[o] = theta;
    sh->Gamma[o] = Gamma;
    sh->Gamma_th[o] = Gamma_th;
    sh->B[o] = B;
    sh->N_p[o] = mass / C_MP;
}

/* save_fwd_shock_state (forward-shock.tpp:151-173) on the pair state: region 2 */
static void save_fwd_state_of_pair(shock_t* sh, size_t o, const rvs_eqn_t* e, const double* s) {
    const double comp_ratio = compute_compression(1, s[RS_GAMMA], 0);
    const double rho = medium_rho(e->med, s[RS_R]);
    const double Gamma_th = compute_Gamma_therm(s[RS_U2], s[RS_M2], 0);
    const double B = compute_downstr_B(e->rad_fwd.eps_B, rho, 0, Gamma_th, comp_ratio);
    write_shock_state(sh, o, s[RS_TCOMV], s[RS_R], s[RS_THETA], s[RS_GAMMA], Gamma_th, B, s[RS_M2]);
}

/* save_rvs_shock_state, reverse-shock.tpp:393-426 */
static void save_rvs_shock_state(shock_t* sh, int j, int k, const rvs_eqn_t* e, const double* s) {
    const size_t o = (size_t)j * sh->n_t + k;
    if (k <= sh->injection_idx[j]) {
        const double sigma4 = rvs_shell_sigma(e, s);
        const double comp_ratio34 = compute_compression(e->Gamma4, s[RS_GAMMA], sigma4);
        const double rho4 = s[RS_M4] / (s[RS_R] * s[RS_R] * s[RS_X4]);
        const double Gamma3_th = compute_Gamma_therm(s[RS_U3], s[RS_M3], 1);
        const double B4 = compute_upstr_B(rho4, sigma4);
        const double B3 = compute_downstr_B(e->eps_B_rvs, rho4, B4, Gamma3_th, comp_ratio34);
        write_shock_state(sh, o, s[RS_TCOMV], s[RS_R], s[RS_THETA], s[RS_GAMMA], Gamma3_th, B3, s[RS_M3]);
    } else {
        const double V3_comv = s[RS_R] * s[RS_R] * s[RS_X3];
        const double comp_ratio = e->V3_comv_x / V3_comv;
        const double Gamma3_th = compute_Gamma_therm(s[RS_U3], s[RS_M3], 0);
        const double B3 = compute_downstr_B(e->eps_B_rvs, e->rho3_x, e->B3_ordered_x, Gamma3_th, comp_ratio);
        write_shock_state(sh, o, s[RS_TCOMV], s[RS_R], s[RS_THETA], s[RS_GAMMA], Gamma3_th, B3, s[RS_M3]);
    }
}

/* reverse_shock_early_extrap, reverse-shock.tpp:428-469 */
static void reverse_shock_early_extrap(shock_t* sh, int j) {
    const int nt = sh->n_t;
    const size_t b = (size_t)j * nt;
    int idx_cut = 0;
    for (; idx_cut < nt; ++idx_cut)
        if (sh->Gamma_th[b + idx_cut] > C_GAMMA_CUT) break;
    const int offset = 2;
    if (idx_cut == 0 || idx_cut >= nt - offset || idx_cut >= sh->injection_idx[j]) return;
    const double log2_r = log2(sh->r[b + idx_cut]);
    const double log2_Gamma_th = log2(sh->Gamma_th[b + idx_cut] - 1);
    const double log2_B = log2(sh->B[b + idx_cut]);
    const double log2_N_p = log2(sh->N_p[b + idx_cut]);
    const double gamma_slope =
        (log2(sh->Gamma_th[b + idx_cut + offset] - 1) - log2_Gamma_th) / (log2(sh->r[b + idx_cut + offset]) - log2_r);
    const double B_slope = (log2(sh->B[b + idx_cut + offset]) - log2_B) / (log2(sh->r[b + idx_cut + offset]) - log2_r);
    const double N_p_slope =
        (log2(sh->N_p[b + idx_cut + offset]) - log2_N_p) / (log2(sh->r[b + idx_cut + offset]) - log2_r);
    for (int k = 0; k < idx_cut; k++) {
        const double dlog2_r = log2(sh->r[b + k]) - log2_r;
        sh->Gamma_th[b + k] = 1 + exp2(log2_Gamma_th + gamma_slope * dlog2_r);
        sh->B[b + k] = exp2(log2_B + B_slope * dlog2_r);
        sh->N_p[b + k] = exp2(log2_N_p + N_p_slope * dlog2_r);
    }
}

/* grid_solve_shock_pair + locate_crossing_time, reverse-shock.tpp:470-590 */
static int grid_solve_shock_pair(int j, const double* t, int nt, shock_t* fwd, shock_t* rvs, rvs_eqn_t* e, double rtol) {
    double state[RS_N];
    const double t_dec = estimate_t_dec(e->jet, e->med, e->theta0);
    const double t0 = dmin(t[0], dmin(0.01 * U_SEC, 0.1 * t_dec));
    rvs_set_init_state(e, state, t0);
    const size_t base = (size_t)j * nt;
    if (state[RS_GAMMA] <= 1.03) { /* RS_Gamma_limit; set_stopping_shock for both shocks */
        for (int k = 0; k < nt; ++k) {
            shock_t* both[2] = {fwd, rvs};
            for (int q = 0; q < 2; ++q) {
                both[q]->t_comv[base + k] = state[RS_TCOMV];
                both[q]->r[base + k] = state[RS_R];
                both[q]->theta[base + k] = state[RS_THETA];
                both[q]->Gamma[base + k] = 1;
                both[q]->Gamma_th[base + k] = 1;
                both[q]->B[base + k] = 0;
                both[q]->N_p[base + k] = 0;
            }
        }
        return 0;
    }
    if (rvs_shell_sigma(e, state) > 0) rtol *= 0.1; /* defaults::solver::magnetized_rtol_factor */
    dopri5_t st;
    dopri5_init(&st, RS_N, rtol, rtol, state, t0, 1e-9 * t0);
    int k = 0;
    for (; t[k] < t0; k++) {
        rvs_set_init_state(e, state, t[k]);
        save_fwd_state_of_pair(fwd, base + k, e, state);
        save_rvs_shock_state(rvs, j, k, e, state);
    }
    int reverse_shock_crossing = 1, injection_idx_pending = 0;
    double t_cross = 0, t_step_start = t0;
    for (int steps = 0; st.t <= t[nt - 1];) {
        if (dopri5_do_step(&st, rvs_rhs, e) != 0) return fail("reverse shock ODE: step size underflow");
        if (++steps > DEF_MAX_ODE_STEPS) {
            fprintf(stderr, "Warning: reverse shock ODE exceeded %d steps at (j=%d), giving up\n", DEF_MAX_ODE_STEPS, j);
            break;
        }
        if (st.t + st.dt == st.t) {
            fprintf(stderr, "Warning: reverse shock ODE stalled (dt below time ulp) at (j=%d), giving up\n", j);
            break;
        }
        if (reverse_shock_crossing && rvs_crossing_complete(e, st.x[st.cur], st.t)) {
            double t_lo = t_step_start, t_hi = st.t;
            for (int iter = 0; iter < 100 && (t_hi - t_lo) > 1e-12 * t_hi; ++iter) {
                const double t_mid = 0.5 * (t_lo + t_hi);
                dopri5_calc_state(&st, t_mid, state);
                if (rvs_crossing_complete(e, state, t_mid))
                    t_hi = t_mid;
                else
                    t_lo = t_mid;
            }
            dopri5_calc_state(&st, t_hi, state);
            t_cross = t_hi;
            rvs_save_cross_state(e, state);
            reverse_shock_crossing = 0;
            injection_idx_pending = 1;
        }
        t_step_start = st.t;
        while (k < nt && st.t > t[k]) {
            dopri5_calc_state(&st, t[k], state);
            if (injection_idx_pending && t[k] >= t_cross) {
                rvs->injection_idx[j] = k > 0 ? k : 1;
                injection_idx_pending = 0;
            }
            save_fwd_state_of_pair(fwd, base + k, e, state);
            save_rvs_shock_state(rvs, j, k, e, state);
            ++k;
        }
    }
    reverse_shock_early_extrap(rvs, j);
    return 0;
}

static void broadcast_shock_groups(shock_t* sh, const coord_t* c) { /* Shock::broadcast_groups, shock.cpp:42-91 */
    const int nt = c->n_t;
    for (int r = 0; r < c->n_reps; ++r) {
        const int j0 = c->reps[r];
        const int j1 = (r + 1 < c->n_reps) ? c->reps[r + 1] : c->phi_size * c->n_theta;
        for (int j = j0 + 1; j < j1; ++j) {
            sh->injection_idx[j] = sh->injection_idx[j0];
            for (int k = 0; k < nt; ++k) {
                const size_t o = (size_t)j * nt + k, s = (size_t)j0 * nt + k;
                sh->t_comv[o] = sh->t_comv[s];
                sh->r[o] = sh->r[s];
                sh->theta[o] = c->theta[j % c->n_theta];
                sh->Gamma[o] = sh->Gamma[s];
                sh->Gamma_th[o] = sh->Gamma_th[s];
                sh->B[o] = sh->B[s];
                sh->N_p[o] = sh->N_p[s];
            }
        }
    }
}

/* generate_shock_pair, reverse-shock.tpp:592-614 */
static int generate_shock_pair(shock_t* fwd, shock_t* rvs, const coord_t* c, const medium_t* med, const jet_t* jet,
                               const vag_model_params* p) {
    shock_alloc(fwd, c->phi_size * c->n_theta, c->n_t);
    shock_alloc(rvs, c->phi_size * c->n_theta, c->n_t);
    for (int r = 0; r < c->n_reps; ++r) {
        const int j = c->reps[r]; /* row */
        rvs_eqn_t e;
        memset(&e, 0, sizeof e);
        e.med = med;
        e.jet = jet;
        e.theta0 = c->theta[j % c->n_theta];
        e.T0 = jet->T0;
        e.Gamma4 = jet_Gamma0(jet, e.theta0);
        e.deps0_dt = jet_eps_k(jet, e.theta0) / jet->T0;
        e.dm0_dt = e.deps0_dt / (e.Gamma4 * C_C2);
        e.u4 = sqrt(e.Gamma4 * e.Gamma4 - 1) * C_C;
        if (jet_is_ejecta(jet)) e.dm0_dt /= 1 + jet->sigma0; /* HasSigma<Ejecta> */
        e.rad_fwd.med = med;
        e.rad_fwd.jet = jet;
        e.rad_fwd.radiative = p->radiative_fireball != 0;
        e.rad_fwd.eps_e = p->eps_e;
        e.rad_fwd.eps_B = p->eps_B;
        e.rad_fwd.p = p->p;
        e.rad_fwd.gamma_m_coeff = (p->p - 2) / (p->p - 1) * p->eps_e * C_MP / C_ME / p->xi_e;
        e.rad_fwd.gamma_c_coeff = 6 * C_PI * C_ME * C_C / C_SIGMAT / (8 * C_PI * p->eps_B);
        e.rad_fwd.eps_e_eff = e.rad_fwd.radiative ? p->eps_e : 0;
        e.eps_B_rvs = p->rvs_eps_B;
        if (grid_solve_shock_pair(j, c->t + (size_t)j * c->n_t, c->n_t, fwd, rvs, &e, p->rtol) != 0) return -1;
    }
    broadcast_shock_groups(fwd, c);
    broadcast_shock_groups(rvs, c);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Equal-arrival-time grids: Observer::observe (non-spreading, axisymmetric =>
 * geom_pre_logged_), src/core/observer.cpp:17-37,143-205,211-235,419-454
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int n_phi_eff, n_theta, n_t;
    int phi_size; /* phi slices of the cells (coord_t::phi_size): cell row of (phi i, theta j) = (phi_size > 1 ? i : 0) * n_theta + j */
    double *lg2_t, *lg2_doppler, *lg2_geom; /* [n_phi_eff][n_theta][n_t] */
    double one_plus_z, lumi_dist;
} eat_t;

static void eat_free(eat_t* e) {
    free(e->lg2_t);
    free(e->lg2_doppler);
    free(e->lg2_geom);
    memset(e, 0, sizeof *e);
}

static void observe(eat_t* o, const coord_t* c, const shock_t* sh, double lumi_dist, double z) {
    const int nth = c->n_theta, nt = c->n_t;
    const int eff_phi = (c->theta_view == 0 && !c->jet_3d) ? 1 : c->n_phi;
    o->n_phi_eff = eff_phi;
    o->phi_size = c->phi_size;
    o->n_theta = nth;
    o->n_t = nt;
    o->one_plus_z = 1 + z;
    o->lumi_dist = lumi_dist;
    const size_t n = (size_t)eff_phi * nth * nt;
    o->lg2_t = malloc(n * sizeof(double));
    o->lg2_doppler = malloc(n * sizeof(double));
    o->lg2_geom = malloc(n * sizeof(double));

    const double cos_obs = cos(c->theta_view), sin_obs = sin(c->theta_view);
    double* dphi = malloc(sizeof(double) * eff_phi);
    if (eff_phi == 1) {
        dphi[0] = 2 * C_PI;
    } else if (c->phi_mirrored) {
        const int last = eff_phi - 1;
        for (int i = 0; i < eff_phi; ++i) {
            const double left = (i > 0) ? 0.5 * (c->phi[i - 1] + c->phi[i]) : 0.0;
            const double right = (i < last) ? 0.5 * (c->phi[i] + c->phi[i + 1]) : C_PI;
            dphi[i] = 2 * (right - left);
        }
    } else {
        const int last = eff_phi - 1;
        for (int i = 0; i < eff_phi; ++i)
            dphi[i] = 0.5 * (c->phi[i + 1 < last ? i + 1 : last] - c->phi[i > 0 ? i - 1 : 0]);
    }
    if (c->spreading) { /* calc_t_obs + calc_solid_angle + finalize_log_grids (geometry not pre-logged), observer.cpp:51-141,439-454 */
        const int last = nth - 1;
        const int n_slice = c->phi_size; /* shock.theta.shape(0): the theta neighbours of a cell are those of its own phi slice */
        double* dcos = malloc(sizeof(double) * (size_t)n_slice * nth * nt);
        for (int is = 0; is < n_slice; ++is)
        for (int j = 0; j < nth; ++j) {
            const int j_p1 = (j == last) ? last : j + 1;
            const size_t R = (size_t)is * nth + j;
            int k_hint_lo = 0, k_hint_hi = 0;
            for (int k = 0; k < nt; ++k) {
                const double t_target = c->t[R * nt + k];
                double th_lo, th_hi;
#define INTERP_THETA_(j_nb, hint, out)                                                                          \
    do {                                                                                                        \
        const double* tn = c->t + ((size_t)is * nth + (j_nb)) * nt;                                             \
        const double* thn = sh->theta + ((size_t)is * nth + (j_nb)) * nt;                                       \
        while ((hint) + 1 < nt && tn[(hint) + 1] < t_target) (hint)++;                                          \
        if ((hint) + 1 >= nt) {                                                                                 \
            (out) = thn[nt - 1];                                                                                \
        } else {                                                                                                \
            const double w = (t_target - tn[hint]) / (tn[(hint) + 1] - tn[hint]);                               \
            (out) = thn[hint] + w * (thn[(hint) + 1] - thn[hint]);                                              \
        }                                                                                                       \
    } while (0)
                const double th_jk = sh->theta[R * nt + k];
                if (j == 0) {
                    th_lo = th_jk;
                } else {
                    double nb;
                    INTERP_THETA_(j - 1, k_hint_lo, nb);
                    th_lo = 0.5 * (th_jk + nb);
                }
                if (j == last) {
                    th_hi = th_jk;
                } else {
                    double nb;
                    INTERP_THETA_(j_p1, k_hint_hi, nb);
                    th_hi = 0.5 * (th_jk + nb);
                }
#undef INTERP_THETA_
                dcos[R * nt + k] = cos(th_hi) - cos(th_lo);
            }
        }
        for (int i = 0; i < eff_phi; ++i) {
            const double cos_phi = cos(c->phi[i] - 0.0);
            const int i_eff = n_slice > 1 ? i : 0; /* i * jet_3d on the slices that are kept */
            for (int j = 0; j < nth; ++j)
                for (int k = 0; k < nt; ++k) {
                    const size_t s = ((size_t)i_eff * nth + j) * nt + k;
                    const size_t q = ((size_t)i * nth + j) * nt + k;
                    const double gamma_ = sh->Gamma[s], r = sh->r[s];
                    const double cos_v = sin(sh->theta[s]) * cos_phi * sin_obs + cos(sh->theta[s]) * cos_obs;
                    const double dop_lin = gamma_ - sqrt((gamma_ - 1) * (gamma_ + 1)) * cos_v;
                    const double time = (c->t[s] + (1 - cos_v) * r / C_C) * o->one_plus_z;
                    const double dOmega = fabs(dcos[s] * dphi[i]);
                    o->lg2_doppler[q] = -log2(dop_lin);
                    o->lg2_t[q] = log2(time);
                    o->lg2_geom[q] = log2(dOmega * r * r) + 3.0 * o->lg2_doppler[q];
                }
        }
        free(dcos);
        free(dphi);
        return;
    }
    double* lg2_r2 = malloc(sizeof(double) * (size_t)nth * nt);
    for (size_t q = 0; q < (size_t)nth * nt; ++q) lg2_r2[q] = 2.0 * log2(sh->r[q]);

    const int last = nth - 1;
    for (int i = 0; i < eff_phi; ++i) {
        const double cos_phi = cos(c->phi[i] - 0.0);
        const double dphi_i = dphi[i];
        double cos_th_carry = 0;
        for (int j = 0; j < nth; ++j) {
            const double th_j = sh->theta[(size_t)j * nt];
            const double ct = cos(th_j), st = sin(th_j);
            const double cos_v = st * cos_phi * sin_obs + ct * cos_obs;
            const double t_coeff = (1 - cos_v) / C_C * o->one_plus_z;
            const double cos_th_lo = (j == 0) ? ct : cos_th_carry;
            double cos_th_hi;
            if (j == last) {
                cos_th_hi = ct;
            } else {
                const double th_hi = 0.5 * (th_j + sh->theta[(size_t)(j + 1) * nt]);
                cos_th_hi = cos(th_hi);
            }
            const double dOmega = fabs((cos_th_hi - cos_th_lo) * dphi_i);
            cos_th_carry = cos_th_hi;
            const double lg2_dOmega = c->jet_3d ? 0 : log2(dOmega); /* geom_pre_logged_ = !spreading && jet_3d == 0 */
            for (int k = 0; k < nt; ++k) {
                const size_t s = (size_t)j * nt + k;
                const size_t q = ((size_t)i * nth + j) * nt + k;
                const double gamma_ = sh->Gamma[s];
                const double r = sh->r[s];
                const double dop_lin = gamma_ - sqrt((gamma_ - 1) * (gamma_ + 1)) * cos_v;
                const double time = c->t[s] * o->one_plus_z + t_coeff * r;
                const double geom = c->jet_3d ? log2(dOmega * r * r) : lg2_dOmega + lg2_r2[s];
                /* finalize_log_grids */
                o->lg2_doppler[q] = -log2(dop_lin);
                o->lg2_t[q] = log2(time);
                o->lg2_geom[q] = geom + 3.0 * o->lg2_doppler[q];
            }
        }
    }
    free(lg2_r2);
    free(dphi);
}

/* ------------------------------------------------------------------------------------------
 * Synchrotron electrons and photons: src/radiation/synchrotron.cpp:45-254,315-408,
 * smooth-power-law-syn.cpp:15-167
 * ---------------------------------------------------------------------------------------- */
/* BrokenPowerLaw<5>, src/util/utilities.h:21-78 */
typedef struct {
    int size;
    double slope[5], log2_lower[5], log2_const[5];
} bpl_t;

static void bpl_first_segment(bpl_t* b, double norm, double lower, double slope) {
    b->size = 0;
    const double log2_lower = log2(lower);
    const double log2_val = log2(norm);
    const int s = b->size++;
    b->slope[s] = slope;
    b->log2_lower[s] = log2_lower;
    b->log2_const[s] = log2_val - slope * log2_lower;
}
static void bpl_add_segment(bpl_t* b, double lower, double slope) {
    const double log2_lower = log2(lower);
    const int prev = b->size - 1;
    const double log2_val = b->log2_const[prev] + b->slope[prev] * log2_lower;
    const int s = b->size++;
    b->slope[s] = slope;
    b->log2_lower[s] = log2_lower;
    b->log2_const[s] = log2_val - slope * log2_lower;
}
static double bpl_eval(const bpl_t* b, double x) {
    const double log2_x = log2(x);
    for (int i = b->size - 1; i > 0; --i)
        if (log2_x >= b->log2_lower[i]) return exp2(b->log2_const[i] + b->slope[i] * log2_x);
    return (b->size > 0) ? exp2(b->log2_const[0] + b->slope[0] * log2_x) : 0.0;
}

/* InverseComptonY, src/radiation/inverse-compton.h:28-76, inverse-compton.cpp:18-187 */
typedef struct {
    double gamma_m_hat, gamma_c_hat, gamma_self, gamma0, Y_T;
    int regime;
    double gamma_m_, B_, p_, gamma_self3;
    bpl_t seg;
} icy_t;

typedef struct {
    double gamma_m, gamma_c, gamma_a, gamma_M, N_e, column_den;
    int regime;
    double p, Y_c;
    icy_t Ys;
} electrons_t;

typedef struct {
    double Y_c;
    icy_t Ys;
    double nu_m, nu_c, nu_a, nu_M, I_nu_max, p;
    /* cached by build(), smooth-power-law-syn.cpp:94-153 */
    double log2_I_nu_max, log2_nu_m, log2_nu_c, log2_nu_a, log2_nu_M, inv_nu_M;
    double log2_norm, log2_thick_norm, smooth_thick, log2_x_far, s_a_blend;
    double log2_nu_lo, log2_nu_hi, smooth_lo, smooth_hi, diff_lo, diff_hi;
} photons_t;

#define C_H (6.63e-27 * U_ERG * U_SEC) /* con::h, src/util/macros.h:95 */

static double compute_syn_freq(double gamma, double B);
static double compute_syn_gamma(double nu, double B);

static void icy_default(icy_t* y) { /* InverseComptonY::InverseComptonY(), inverse-compton.cpp:37-43 */
    memset(y, 0, sizeof *y);
    y->gamma_m_hat = 1.0;
    y->gamma_c_hat = 1.0;
    y->gamma_self = 1;
    y->gamma0 = 1;
    y->Y_T = 0.0;
    y->regime = 0;
    y->gamma_m_ = 1;
    y->B_ = 0;
    y->p_ = 2.3;
    y->gamma_self3 = 1;
    y->seg.size = 0;
}
static double icy_gamma_hat(const icy_t* y, double gamma) {
    return dmax(y->gamma_self3 / (gamma * gamma), 1.0);
}
static double icy_gamma_spectrum(const icy_t* y, double gamma) {
    return bpl_eval(&y->seg, gamma);
}
static double icy_nu_spectrum(const icy_t* y, double nu) {
    return icy_gamma_spectrum(y, compute_syn_gamma(nu, y->B_));
}
/* build_segments, inverse-compton.cpp:112-175 (regimes 3-5 are unreachable: update_cooling_breaks only sets 1 or 2) */
static void icy_build_segments(icy_t* y) {
    switch (y->regime) {
        case 0:
            bpl_first_segment(&y->seg, y->Y_T, 1.0, 0.0);
            break;
        case 1:
            bpl_first_segment(&y->seg, y->Y_T, 1.0, 0.0);
            bpl_add_segment(&y->seg, y->gamma_c_hat, 0.5 * (y->p_ - 3.0));
            bpl_add_segment(&y->seg, y->gamma_m_hat, -4.0 / 3.0);
            break;
        default:
            bpl_first_segment(&y->seg, y->Y_T, 1.0, 0.0);
            bpl_add_segment(&y->seg, y->gamma_m_hat, -0.5);
            bpl_add_segment(&y->seg, y->gamma_c_hat, -4.0 / 3.0);
            break;
    }
}
/* update_gamma0, inverse-compton.cpp:48-88 */
static void icy_update_gamma0(icy_t* y, double gamma_c) {
    if (y->Y_T < 1) {
        y->gamma0 = 0.0;
        return;
    }
    if (y->gamma_m_ < gamma_c) {
        y->gamma0 = fast_pow(y->Y_T, 2.0 / (3.0 - y->p_)) * y->gamma_c_hat;
        if (y->gamma0 > y->gamma_m_hat)
            y->gamma0 = y->gamma_m_hat * fast_pow(y->Y_T, 3.0 / 4.0) * fast_pow(gamma_c / y->gamma_m_, 0.75 * (y->p_ - 3.0));
    } else {
        if (y->gamma_m_ < y->gamma_m_hat) {
            y->gamma0 = y->Y_T * y->Y_T * y->gamma_m_hat;
            if (y->gamma0 > y->gamma_c_hat) y->gamma0 = fast_pow(y->Y_T * gamma_c / y->gamma_m_, 3.0 / 4.0) * y->gamma_c_hat;
        } else {
            y->gamma0 = y->Y_T * y->Y_T * y->gamma_m_hat;
            if (y->gamma0 > y->gamma_self) y->gamma0 = sqrt(y->Y_T * y->gamma_m_ * y->gamma_m_hat);
        }
    }
}
/* update_cooling_breaks, inverse-compton.cpp:90-110 */
static void icy_update_cooling_breaks(icy_t* y, double gamma_c, double Y_T) {
    y->gamma_c_hat = icy_gamma_hat(y, gamma_c);
    y->Y_T = Y_T;
    icy_update_gamma0(y, gamma_c);
    y->regime = (y->gamma_m_ < gamma_c) ? 1 : 2;
    icy_build_segments(y);
}
/* InverseComptonY(gamma_m, gamma_c, p, B, Y_T, is_KN), inverse-compton.cpp:18-35 */
static void icy_init(icy_t* y, double gamma_m, double gamma_c, double p, double B, double Y_T, int is_KN) {
    icy_default(y);
    y->gamma_c_hat = 1;
    const double nu_m = compute_syn_freq(gamma_m, B);
    y->gamma_m_hat = dmax(C_ME * C_C2 / C_H / nu_m, 1.0);
    y->gamma_self = fast_pow(y->gamma_m_hat * gamma_m * gamma_m, 1.0 / 3.0);
    y->gamma_self3 = y->gamma_self * y->gamma_self * y->gamma_self;
    y->B_ = B;
    y->gamma_m_ = gamma_m;
    y->p_ = p;
    if (is_KN) {
        icy_update_cooling_breaks(y, gamma_c, Y_T);
    } else {
        y->gamma_c_hat = icy_gamma_hat(y, gamma_c);
        y->Y_T = Y_T;
        y->regime = 0;
        icy_build_segments(y);
    }
}

static int order3(double a, double b, double c) {
    return a <= b && b <= c;
}
static int determine_regime(double a, double c, double m) {
    if (order3(a, m, c)) return 1;
    if (order3(m, a, c)) return 2;
    if (order3(a, c, m)) return 3;
    if (order3(c, a, m)) return 4;
    if (order3(m, c, a)) return 5;
    if (order3(c, m, a)) return 6;
    return 0;
}

static double compute_syn_I_peak(double B, double column_den) {
    const double sin_angle_ave = C_PI / 4;
    const double Fx_max = 0.92;
    const double P = B * (sin_angle_ave * Fx_max * M_SQRT3_ * C_E3 / (C_ME * C_C2));
    return P * column_den / (4 * C_PI);
}

static double compute_syn_freq(double gamma, double B) {
    if (B == 0 || !isfinite(gamma)) return 0;
    return 3 * C_E / (4 * C_PI * C_ME * C_C) * B * gamma * gamma;
}

static double compute_syn_gamma(double nu, double B) {
    return sqrt((4 * C_PI * C_ME * C_C / (3 * C_E)) * (nu / B));
}

static double compute_syn_gamma_M(double B, double Y) {
    if (B == 0) return INFINITY;
    return sqrt(6 * C_PI * C_E / C_SIGMAT / (B * (1 + Y)));
}

/* root_bisect, src/util/utilities.h:230-241 specialised to the p == 2 equation */
static double gamma_m_p2_eq(double x, double gamma_M, double gamma_ave_minus_1) {
    return x * log(gamma_M) - (x + 1) * log(x) - gamma_ave_minus_1 - log(gamma_M);
}

static double compute_syn_gamma_m(double Gamma_th, double gamma_M, double eps_e, double p, double xi) {
    const double gamma_ave_minus_1 = eps_e * (Gamma_th - 1) * (C_MP / C_ME) / xi;
    double gamma_m_minus_1 = 1;
    if (p > 2) {
        gamma_m_minus_1 = (p - 2) / (p - 1) * gamma_ave_minus_1;
    } else if (p < 2) {
        gamma_m_minus_1 = pow((2 - p) / (p - 1) * gamma_ave_minus_1 * pow(gamma_M, p - 2), 1 / (p - 1));
    } else {
        double low = 0, high = gamma_M;
        const double eps = 1e-6;
        for (int iter = 0; iter < 1000 && (high - low) > fabs((high + low) * 0.5) * eps; ++iter) {
            const double mid = 0.5 * (high + low);
            if (gamma_m_p2_eq(mid, gamma_M, gamma_ave_minus_1) * gamma_m_p2_eq(high, gamma_M, gamma_ave_minus_1) > 0)
                high = mid;
            else
                low = mid;
        }
        gamma_m_minus_1 = 0.5 * (high + low);
    }
    return gamma_m_minus_1 + 1;
}

static double compute_gamma_c(double t_comv, double B, double Y) {
    const double gamma_bar = (6 * C_PI * C_ME * C_C / C_SIGMAT) / (B * B * (1 + Y) * t_comv) * 1;
    return (gamma_bar + sqrt(gamma_bar * gamma_bar + 4)) / 2;
}

/* compute_syn_gamma_a, synchrotron.cpp:212-246 (Ys default-constructed + Y_c = 0 in the first pass) */
static double compute_syn_gamma_a(double B, double I_syn_peak, double gamma_m, double gamma_c, double p, const icy_t* Ys,
                                  double Y_c) {
    const double gamma_peak = dmin(gamma_m, gamma_c);
    const double nu_peak = compute_syn_freq(gamma_peak, B);
    const double kT = (gamma_peak - 1) * (C_ME * C_C2) / 3;
    double nu_a = fast_pow(I_syn_peak * C_C2 / (cbrt(nu_peak) * 2 * kT), 0.6);
    if (nu_a > nu_peak) {
        if (gamma_c > gamma_m) {
            const double nu_m = compute_syn_freq(gamma_m, B);
            nu_a = fast_pow(I_syn_peak * C_C2 / (2 * kT) * fast_pow(nu_m, p / 2), 2 / (p + 4));
            const double nu_c = compute_syn_freq(gamma_c, B);
            if (nu_a > nu_c) {
                nu_a = fast_pow(I_syn_peak * C_C2 / (2 * kT) * sqrt(nu_c) * fast_pow(nu_m, p / 2), 2 / (p + 5));
                const double ic = (1 + Y_c) / (1 + icy_nu_spectrum(Ys, nu_a));
                nu_a *= fast_pow(ic, 2 / (p + 5));
            }
        } else {
            const double nu_c = compute_syn_freq(gamma_c, B);
            nu_a = fast_pow(I_syn_peak * C_C2 / (2 * kT) * sqrt(nu_c), 0.4);
            double ic = (1 + Y_c) / (1 + icy_nu_spectrum(Ys, nu_a));
            nu_a *= fast_pow(ic, 0.4);
            const double nu_m = compute_syn_freq(gamma_m, B);
            if (nu_a > nu_m) {
                nu_a = fast_pow(I_syn_peak * C_C2 / (2 * kT) * sqrt(nu_c) * fast_pow(nu_m, p / 2), 2 / (p + 5));
                ic = (1 + Y_c) / (1 + icy_nu_spectrum(Ys, nu_a));
                nu_a *= fast_pow(ic, 2 / (p + 5));
            }
        }
    }
    return compute_syn_gamma(nu_a, B) + 1;
}

static double cyclotron_correction(double gamma_m, double p) {
    double f = (gamma_m - 1) / gamma_m;
    if (p > 3) f = fast_pow(f, (p - 1) / 2);
    return f;
}

/* one cell of generate_syn_electrons, synchrotron.cpp:315-360 */
/* cool_after_crossing, synchrotron.cpp:190-195 */
static double cool_after_crossing(double gamma_x, double gamma_m_x, double gamma_m) {
    const double gamma_syn = gamma_x;
    const double f_ad = (gamma_m - 1) / (gamma_m_x - 1);
    return (gamma_syn - 1) * f_ad + 1;
}

/* one cell of generate_syn_electrons (synchrotron.cpp:315-361); `inj` = the electrons frozen at the crossing cell
 * (injection_idx - 1) when this cell is a relic one (cool_relic_electrons, synchrotron.h:187-201), else NULL */
static void syn_electrons_cell(electrons_t* e, double t_com, double B, double r, double Gamma_th, double N_p,
                               const vag_model_params* rad, const electrons_t* inj) {
    e->gamma_M = compute_syn_gamma_M(B, 0.);
    e->gamma_m = compute_syn_gamma_m(Gamma_th, e->gamma_M, rad->eps_e, rad->p, rad->xi_e);
    const double f_syn = cyclotron_correction(e->gamma_m, rad->p);
    e->N_e = N_p * rad->xi_e * f_syn;
    e->column_den = e->N_e / (r * r);
    const double I_nu_peak = compute_syn_I_peak(B, e->column_den);
    if (inj) {
        e->gamma_c = cool_after_crossing(inj->gamma_c, inj->gamma_m, e->gamma_m);
        e->gamma_M = cool_after_crossing(inj->gamma_M, inj->gamma_m, e->gamma_m);
    } else {
        e->gamma_c = compute_gamma_c(t_com, B, 0.);
    }
    icy_default(&e->Ys);
    e->Y_c = 0;
    e->gamma_a = compute_syn_gamma_a(B, I_nu_peak, e->gamma_m, e->gamma_c, rad->p, &e->Ys, 0.0);
    e->regime = determine_regime(e->gamma_a, e->gamma_c, e->gamma_m);
    e->p = rad->p;
}

static double sigmoid2(double x) {
    return 1.0 / (1.0 + exp2(-x));
}
static double blend(double w, double a, double b) {
    return w * a + (1.0 - w) * b;
}

/* smooth-power-law-syn.cpp:49-78 */
static double log2_optical_thick_sharp(const photons_t* ph, double log2_nu) {
    if (log2_nu < ph->log2_nu_m) return 2. * (log2_nu - ph->log2_nu_m);
    return 2.5 * (log2_nu - ph->log2_nu_m);
}
static double log2_optical_thin_sharp(const photons_t* ph, double log2_nu) {
    const double p = ph->p, lm = ph->log2_nu_m, lc = ph->log2_nu_c;
    if (lm < lc) {
        if (log2_nu < lm) return (log2_nu - lm) / 3.0;
        if (log2_nu < lc) return 0.5 * (1.0 - p) * (log2_nu - lm);
        return 0.5 * (1.0 - p) * (lc - lm) - 0.5 * p * (log2_nu - lc);
    }
    if (log2_nu < lc) return (log2_nu - lc) / 3.0;
    if (log2_nu < lm) return -0.5 * (log2_nu - lc);
    return -0.5 * (lm - lc) - 0.5 * p * (log2_nu - lm);
}

/* SmoothPowerLawSyn::build, smooth-power-law-syn.cpp:94-153 */
static void photons_build(photons_t* ph) {
    const double p = ph->p;
    ph->log2_I_nu_max = log2(ph->I_nu_max);
    ph->log2_nu_m = log2(ph->nu_m);
    ph->log2_nu_c = log2(ph->nu_c);
    ph->log2_nu_a = log2(ph->nu_a);
    ph->log2_nu_M = log2(ph->nu_M);
    ph->inv_nu_M = 1.0 / ph->nu_M;
    ph->smooth_thick = (3.44 * p - 1.41) / M_LN2_;
    ph->log2_x_far = 1.5 * log2(20.0 / ph->smooth_thick);
    const double s_swap = 4.0, s_floor = 0.1;
    const double w_slow = sigmoid2(s_swap * (ph->log2_nu_c - ph->log2_nu_m));
    const double soft_offset = log2_softplus(-s_swap * fabs(ph->log2_nu_c - ph->log2_nu_m)) / s_swap;
    ph->log2_nu_lo = dmin(ph->log2_nu_m, ph->log2_nu_c) - soft_offset;
    ph->log2_nu_hi = dmax(ph->log2_nu_m, ph->log2_nu_c) + soft_offset;
    const double s_m_slow = dmax(1.84 - 0.40 * p, s_floor);
    const double s_c_slow = dmax(1.15 - 0.06 * p, s_floor);
    const double s_c_fast = 0.597;
    const double s_m_fast = dmax(3.34 - 0.82 * p, s_floor);
    ph->smooth_lo = blend(w_slow, s_m_slow, s_c_fast);
    ph->smooth_hi = blend(w_slow, s_c_slow, s_m_fast);
    const double alpha_mid = blend(w_slow, -0.5 * (p - 1.0), -0.5);
    ph->diff_lo = ph->smooth_lo * (1.0 / 3.0 - alpha_mid);
    ph->diff_hi = ph->smooth_hi * (alpha_mid + 0.5 * p);
    const double u = sigmoid2(s_swap * (ph->log2_nu_a - ph->log2_nu_m));
    const double v = sigmoid2(s_swap * (ph->log2_nu_a - ph->log2_nu_c));
    const double w_below = (1.0 - u) * (1.0 - v);
    const double w_above = u * v;
    const double s_a_below = 1.64;
    const double s_a_mid = dmax(1.47 - 0.21 * p, s_floor);
    const double s_a_above = dmax(0.94 - 0.14 * p, s_floor);
    ph->s_a_blend = w_below * s_a_below + w_above * s_a_above + (1.0 - w_below - w_above) * s_a_mid;
    ph->log2_norm = 1.0 / ph->smooth_lo;
    ph->log2_thick_norm = log2_optical_thin_sharp(ph, ph->log2_nu_a) - log2_optical_thick_sharp(ph, ph->log2_nu_a);
}

/* one cell of generate_syn_photons, synchrotron.cpp:376-408 */
static void syn_photons_cell(photons_t* ph, const electrons_t* e, double B, double p) {
    ph->p = p;
    ph->Ys = e->Ys;
    ph->Y_c = e->Y_c;
    ph->nu_M = compute_syn_freq(e->gamma_M, B);
    ph->nu_m = compute_syn_freq(e->gamma_m, B);
    ph->nu_c = compute_syn_freq(e->gamma_c, B);
    ph->nu_a = compute_syn_freq(e->gamma_a, B);
    ph->I_nu_max = compute_syn_I_peak(B, e->column_den);
    photons_build(ph);
}

/* SmoothPowerLawSyn::compute_log2_I_nu with Y == 0, smooth-power-law-syn.cpp:15-46,80-92,159-167 */
static double compute_log2_I_nu(const photons_t* ph, double log2_nu) {
    const double thin = (log2_nu - ph->log2_nu_lo) / 3.0 +
                        log2_broken_power_ratio(log2_nu, ph->log2_nu_lo, ph->diff_lo, ph->smooth_lo) +
                        log2_broken_power_ratio(log2_nu, ph->log2_nu_hi, ph->diff_hi, ph->smooth_hi);
    double thin_ic = thin;
    /* IC steepening above nu_c on the thin branch only (smooth-power-law-syn.cpp:83-90, inverse-compton.h:781-792) */
    if (log2_nu > ph->log2_nu_c && (ph->Y_c > 0 || ph->Ys.Y_T > 0)) {
        const double nu = exp2(log2_nu);
        thin_ic += log2((1. + ph->Y_c) / (1 + icy_nu_spectrum(&ph->Ys, nu)));
    }
    double thick;
    {
        const double log2_x = log2_nu - ph->log2_nu_m;
        if (log2_x > ph->log2_x_far) {
            thick = 2.5 * log2_x;
        } else {
            const double s = -ph->smooth_thick * exp2(2. / 3 * log2_x);
            thick = 2.5 * log2_x + log2_softplus(-0.5 * log2_x + s);
        }
    }
    const double log2_b = thick + ph->log2_thick_norm;
    /* log2_smooth_one(a, b, s) = a - softplus(s (a - b)) / s */
    const double smooth_one = thin_ic - log2_softplus(ph->s_a_blend * (thin_ic - log2_b)) / ph->s_a_blend;
    const double spec = ph->log2_I_nu_max + (ph->log2_norm + smooth_one);
    if (log2_nu - ph->log2_nu_M < -20) return spec;
    return spec - M_LOG2E_ * ph->inv_nu_M * exp2(log2_nu);
}

/* generate_syn_electrons + generate_syn_photons (+ apply_ic_cooling, pybind/pymodel.h:567-577) + broadcast_symmetry
 * (utilities.h:293-320) */
static void ic_cooling_row(electrons_t* el, const shock_t* sh, size_t row0, int nt, const vag_model_params* rad, int kn);

static void generate_syn(electrons_t* el, photons_t* ph, const shock_t* sh, const coord_t* c, const vag_model_params* p) {
    const int nt = c->n_t;
    const int ssc = (p->flags & VAG_FLAG_SSC) != 0, kn = (p->flags & VAG_FLAG_KN) != 0;
    for (int r = 0; r < c->n_reps; ++r) {
        const int j0 = c->reps[r];
        const int j1 = (r + 1 < c->n_reps) ? c->reps[r + 1] : c->phi_size * c->n_theta;
        for (int k = 0; k < nt; ++k) {
            const size_t o = (size_t)j0 * nt + k;
            const int k_inj = sh->injection_idx[j0];
            syn_electrons_cell(&el[o], sh->t_comv[o], sh->B[o], sh->r[o], sh->Gamma_th[o], sh->N_p[o], p,
                               k >= k_inj ? &el[(size_t)j0 * nt + k_inj - 1] : NULL);
            syn_photons_cell(&ph[o], &el[o], sh->B[o], p->p);
        }
        if (ssc) { /* Thomson_cooling / KN_cooling: electrons updated in place, photons regenerated */
            ic_cooling_row(el, sh, (size_t)j0 * nt, nt, p, kn);
            for (int k = 0; k < nt; ++k) {
                const size_t o = (size_t)j0 * nt + k;
                syn_photons_cell(&ph[o], &el[o], sh->B[o], el[o].p);
            }
        }
        for (int j = j0 + 1; j < j1; ++j)
            for (int k = 0; k < nt; ++k) {
                el[(size_t)j * nt + k] = el[(size_t)j0 * nt + k];
                ph[(size_t)j * nt + k] = ph[(size_t)j0 * nt + k];
            }
    }
}

/* ------------------------------------------------------------------------------------------
 * Inverse-Compton cooling and SSC photons: src/radiation/inverse-compton.h:270-776,
 * src/radiation/inverse-compton.cpp:192-378
 * ---------------------------------------------------------------------------------------- */
static double compute_Thomson_Y(const vag_model_params* rad, double gamma_m, double gamma_c) {
    const double eta_e = (gamma_c < gamma_m) ? 1 : fast_pow(gamma_c / gamma_m, 2 - rad->p); /* eta_rad_Thomson */
    const double b = eta_e * rad->eps_e / rad->eps_B;
    return 0.5 * (sqrt(1. + 4. * b) - 1.);
}

/* update_gamma_c_Thomson, inverse-compton.cpp:192-204 */
static void update_gamma_c_Thomson(double* gamma_c, icy_t* Ys, const vag_model_params* rad, double B, double t_com,
                                   double gamma_m, double gamma_c_last) {
    double Y_T = compute_Thomson_Y(rad, gamma_m, *gamma_c);
    double gamma_c_new = gamma_c_last;
    while (fabs((gamma_c_new - *gamma_c) / *gamma_c) > 1e-3) {
        *gamma_c = gamma_c_new;
        Y_T = compute_Thomson_Y(rad, gamma_m, *gamma_c);
        gamma_c_new = compute_gamma_c(t_com, B, Y_T);
    }
    *gamma_c = gamma_c_new;
    icy_init(Ys, gamma_m, *gamma_c, rad->p, B, Y_T, 0);
}

/* update_gamma_c_KN, inverse-compton.cpp:206-235 */
static void update_gamma_c_KN(double* gamma_c, icy_t* Ys, const vag_model_params* rad, double B, double t_com,
                              double gamma_m, double gamma_c_last) {
    double gamma_c_new = gamma_c_last;
    double Y_T = compute_Thomson_Y(rad, gamma_m, gamma_c_new);
    icy_init(Ys, gamma_m, gamma_c_new, rad->p, B, Y_T, 1);
    const int max_iter = 100;
    int iter = 0;
    do {
        *gamma_c = gamma_c_new;
        Y_T = compute_Thomson_Y(rad, gamma_m, *gamma_c);
        icy_update_cooling_breaks(Ys, *gamma_c, Y_T);
        const double Y_c = icy_gamma_spectrum(Ys, *gamma_c);
        gamma_c_new = compute_gamma_c(t_com, B, Y_c);
        iter++;
    } while (fabs((gamma_c_new - *gamma_c) / *gamma_c) > 1e-3 && iter < max_iter);
    *gamma_c = gamma_c_new;
}

/* update_gamma_M, inverse-compton.cpp:237-251 */
static void update_gamma_M(double* gamma_M, const icy_t* Ys, double B) {
    if (B == 0) {
        *gamma_M = INFINITY;
        return;
    }
    double Y_M = icy_gamma_spectrum(Ys, *gamma_M);
    double gamma_M_new = compute_syn_gamma_M(B, Y_M);
    while (fabs((*gamma_M - gamma_M_new) / gamma_M_new) > 1e-3) {
        *gamma_M = gamma_M_new;
        Y_M = icy_gamma_spectrum(Ys, *gamma_M);
        gamma_M_new = compute_syn_gamma_M(B, Y_M);
    }
}

/* IC_cooling for one representative row (sequential in k: gamma_c_last), inverse-compton.h:729-768 */
static void ic_cooling_row(electrons_t* el, const shock_t* sh, size_t row0, int nt, const vag_model_params* rad, int kn) {
    for (int k = 0; k < nt; ++k) {
        const size_t o = row0 + k;
        electrons_t* e = &el[o];
        const double t_com = sh->t_comv[o], B = sh->B[o];
        const double gamma_c_last = el[row0 + (k > 0 ? k - 1 : 0)].gamma_c;
        if (kn)
            update_gamma_c_KN(&e->gamma_c, &e->Ys, rad, B, t_com, e->gamma_m, gamma_c_last);
        else
            update_gamma_c_Thomson(&e->gamma_c, &e->Ys, rad, B, t_com, e->gamma_m, gamma_c_last);
        update_gamma_M(&e->gamma_M, &e->Ys, B);
        {   /* cool_relic_electrons, inverse-compton.h:752 */
            const int k_inj = sh->injection_idx[row0 / nt];
            if (k >= k_inj) {
                const electrons_t* inj = &el[row0 + k_inj - 1];
                e->gamma_c = cool_after_crossing(inj->gamma_c, inj->gamma_m, e->gamma_m);
                e->gamma_M = cool_after_crossing(inj->gamma_M, inj->gamma_m, e->gamma_m);
            }
        }
        const double I_nu_peak = compute_syn_I_peak(B, e->column_den);
        e->Y_c = icy_gamma_spectrum(&e->Ys, e->gamma_c);
        e->gamma_a = compute_syn_gamma_a(B, I_nu_peak, e->gamma_m, e->gamma_c, e->p, &e->Ys, e->Y_c);
        e->regime = determine_regime(e->gamma_a, e->gamma_c, e->gamma_m);
    }
}

/* SynElectrons::compute_spectrum / compute_column_den, synchrotron.cpp:261-309 */
static double electrons_column_den(const electrons_t* e, double gamma) {
    double spec;
    switch (e->regime) {
        case 1:
        case 2:
        case 5:
            spec = (e->p - 1) / e->gamma_m * exp(-gamma / e->gamma_M - e->gamma_m / gamma) *
                   fast_pow(gamma / e->gamma_m, -e->p) * e->gamma_c / (gamma + e->gamma_c);
            break;
        case 3:
        case 4:
        case 6:
            spec = exp(-gamma / e->gamma_M - e->gamma_c / gamma) * e->gamma_c / (gamma * gamma) /
                   (1.0 + fast_pow(gamma / e->gamma_m, e->p - 1));
            break;
        default:
            spec = 0;
    }
    if (gamma <= e->gamma_c) return e->column_den * spec;
    return e->column_den * spec * (1 + e->Y_c) / (1 + icy_gamma_spectrum(&e->Ys, gamma));
}

/* Klein-Nishina cross-section ratio and its LUT, inverse-compton.cpp:257-378 */
static double compton_ratio_from_x(double x) {
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
#define KN_LUT_N 128
static double kn_ratio[KN_LUT_N], kn_lg2_ratio[KN_LUT_N], kn_inv_step;
static int kn_ready = 0;
static const double KN_LG2_XMIN = -6.6438561897747247, KN_LG2_XMAX = 6.6438561897747247;
static void kn_lut_init(void) {
    const double step = (KN_LG2_XMAX - KN_LG2_XMIN) / (double)(KN_LUT_N - 1);
    kn_inv_step = 1.0 / step;
    for (int i = 0; i < KN_LUT_N; ++i) {
        const double lg2_x = KN_LG2_XMIN + step * (double)i;
        kn_ratio[i] = compton_ratio_from_x(exp2(lg2_x));
        kn_lg2_ratio[i] = log2(kn_ratio[i]);
    }
    kn_ready = 1;
}
static void compton_correction_pair(double nu, double* corr, double* lg2_corr) {
    const double x = C_H / (C_ME * C_C2) * nu;
    if (!(x > 0)) {
        *corr = 0;
        *lg2_corr = -INFINITY;
        return;
    }
    if (x <= 1e-2) {
        *corr = 1 - 2 * x;
        *lg2_corr = -(2 * x + 2 * x * x) * 1.4426950408889634;
        return;
    }
    if (x >= 1e2) {
        *corr = compton_ratio_from_x(x);
        *lg2_corr = log2(*corr);
        return;
    }
    if (!kn_ready) kn_lut_init();
    const double pos = (log2(x) - KN_LG2_XMIN) * kn_inv_step;
    if (pos <= 0) {
        *corr = kn_ratio[0];
        *lg2_corr = kn_lg2_ratio[0];
        return;
    }
    if (pos >= (double)(KN_LUT_N - 1)) {
        *corr = kn_ratio[KN_LUT_N - 1];
        *lg2_corr = kn_lg2_ratio[KN_LUT_N - 1];
        return;
    }
    const size_t idx = (size_t)pos;
    const double frac = pos - (double)idx;
    *corr = kn_ratio[idx] + (kn_ratio[idx + 1] - kn_ratio[idx]) * frac;
    *lg2_corr = kn_lg2_ratio[idx] + (kn_lg2_ratio[idx + 1] - kn_lg2_ratio[idx]) * frac;
}

/* ICPhoton, inverse-compton.h:86-607 */
typedef struct {
    electrons_t electrons;
    photons_t photons;
    int KN, generated, n_ic;
    double nu_eval_min, nu_eval_max;
    double log2_nu_theory_max, log2_nu_theory_min;
    long ic_idx0;
    double *log2_nu_IC, *log2_I_nu_IC, *interp_slope;
} icphoton_t;

#define IC_Q (3.321928094887362 / 8) /* lattice_quantum */
#define IC_GAMMA_MULT 2
#define IC_NU_MULT 2
#define IC_IC_MULT 2
#define IC_X0 0.47140452079103166

static double compute_log2_I_nu(const photons_t* ph, double log2_nu);

static void icphoton_free(icphoton_t* ic) {
    free(ic->log2_nu_IC);
    free(ic->log2_I_nu_IC);
    free(ic->interp_slope);
    ic->log2_nu_IC = ic->log2_I_nu_IC = ic->interp_slope = NULL;
    ic->n_ic = 0;
}

static void icphoton_copy(icphoton_t* dst, const icphoton_t* src) { /* broadcast_symmetry deep copy */
    *dst = *src;
    if (src->n_ic > 0) {
        dst->log2_nu_IC = malloc(sizeof(double) * src->n_ic);
        dst->log2_I_nu_IC = malloc(sizeof(double) * src->n_ic);
        dst->interp_slope = malloc(sizeof(double) * src->n_ic);
        memcpy(dst->log2_nu_IC, src->log2_nu_IC, sizeof(double) * src->n_ic);
        memcpy(dst->log2_I_nu_IC, src->log2_I_nu_IC, sizeof(double) * src->n_ic);
        memcpy(dst->interp_slope, src->interp_slope, sizeof(double) * (src->n_ic - 1));
    }
}

static double power_law_bin_integral(double f_lo, double f_hi, double nu_lo, double nu_hi, double lg2f_lo, double lg2f_hi,
                                     double lg2r, double inv_lg2r, double trap) {
    if (!(f_lo > 0) || !(f_hi > 0)) return trap;
    const double s1 = 1 + (lg2f_hi - lg2f_lo) * inv_lg2r;
    if (fabs(s1) > 1e-3) return (f_hi * nu_hi - f_lo * nu_lo) / s1;
    return f_lo * nu_lo * lg2r * 0.6931471805599453;
}

static int build_lattice(double lg2_lo, double lg2_hi, double step, double** grid, double** lg2_grid) {
    size_t n = (size_t)ceil((lg2_hi - lg2_lo) / step) + 1;
    if (n < 2) n = 2;
    *grid = malloc(sizeof(double) * n);
    *lg2_grid = malloc(sizeof(double) * n);
    for (size_t i = 0; i < n; ++i) {
        (*lg2_grid)[i] = lg2_lo + step * (double)i;
        (*grid)[i] = exp2((*lg2_grid)[i]);
    }
    return (int)n;
}

/* Instrumentation of generate_spectrum (round 6): with `on` set, every table build also counts the unit of work of the device's
 * spectrum kernel -- (electron energy i, seed bin m) terms dN_e(i) x (bin integral of m at energy i) -- and how many of them are NOT
 * NEEDED: the bin lies above every output node of the clamped table, or the term is below 2^-bits of the final value of the SMALLEST
 * output node it enters (a bin enters every node below it; the output falls with frequency).  `energies_dead` counts the (cell,
 * energy) walks in which every bin is such a term: what a wavefront could skip as a whole.  Not part of any checked result. */
static struct {
    int on;
    double bits;
    long long cells, terms, terms_negligible, energies, energies_dead;
} g_ic_tally = {0, 60.0, 0, 0, 0, 0, 0};

/* generate_spectrum: compute_grid_params + initialize_grids + sample_distributions + compute_IC_spectrum */
static void icphoton_generate(icphoton_t* ic) {
    icphoton_free(ic);
    const electrons_t* el = &ic->electrons;
    const photons_t* ph = &ic->photons;
    /* compute_grid_params, inverse-compton.h:297-338 */
    const double tail_factor = dmax(-log(1e-2), 5.0);
    const double gamma_min = dmin(el->gamma_m, el->gamma_c) / 30;
    const double gamma_max = dmax(el->gamma_M * tail_factor, gamma_min);
    const double nu_min = dmin(ph->nu_a, ph->nu_m) / 10;
    const double nu_max = dmax(ph->nu_M * tail_factor, nu_min);
    double nu_IC_min = 4 * IC_X0 * nu_min * gamma_min * gamma_min;
    const double nu_ic_base = 4 * IC_X0 * ph->nu_M * el->gamma_M * el->gamma_M;
    const double nu_ic_single_cut = dmax(nu_ic_base * tail_factor * tail_factor, nu_ic_base * tail_factor);
    double nu_IC_max = nu_ic_single_cut * 2.0;
    ic->log2_nu_theory_max = log2(nu_IC_max);
    ic->log2_nu_theory_min = log2(nu_IC_min);
    nu_IC_min = dmax(nu_IC_min, dmin(ic->nu_eval_min / 4.0, nu_IC_max / 16.0));
    nu_IC_max = dmin(nu_IC_max, dmax(ic->nu_eval_max * 4.0, nu_IC_min * 16.0));
#define POSFIN_(x) (isfinite(x) && (x) > 0)
    if (!(POSFIN_(gamma_min) && POSFIN_(gamma_max) && POSFIN_(nu_min) && POSFIN_(nu_max) && POSFIN_(nu_IC_min) &&
          POSFIN_(nu_IC_max))) {
        ic->generated = 1;
        return;
    }
#undef POSFIN_
    /* initialize_grids, inverse-compton.h:340-369 */
    double *nu_seed, *lg2_nu_seed, *gamma, *lg2_gamma;
    const int nu_size = build_lattice(log2(nu_min), log2(nu_max), IC_NU_MULT * IC_Q, &nu_seed, &lg2_nu_seed);
    double* dnu_seed = malloc(sizeof(double) * nu_size);
    for (int j = 0; j + 1 < nu_size; ++j) dnu_seed[j] = nu_seed[j + 1] - nu_seed[j];
    const int g_size = build_lattice(log2(gamma_min), log2(gamma_max), IC_GAMMA_MULT * IC_Q, &gamma, &lg2_gamma);
    const double q = IC_Q;
    const double phase = lg2_nu_seed[0] + 2 * lg2_gamma[0] + log2(4 * IC_X0);
    const long n_lo = (long)floor((log2(nu_IC_min) - phase) / (q * IC_IC_MULT));
    const long n_hi = (long)ceil((log2(nu_IC_max) - phase) / (q * IC_IC_MULT));
    const long span = n_hi - n_lo;
    const int n_ic = (int)(span > 1 ? span : 1) + 1;
    ic->ic_idx0 = n_lo * IC_IC_MULT;
    ic->n_ic = n_ic;
    ic->log2_nu_IC = malloc(sizeof(double) * n_ic);
    ic->log2_I_nu_IC = malloc(sizeof(double) * n_ic);
    ic->interp_slope = calloc(n_ic, sizeof(double));
    for (int k = 0; k < n_ic; ++k) ic->log2_nu_IC[k] = phase + q * (double)(ic->ic_idx0 + (long)(IC_IC_MULT * k));
    /* sample_distributions, inverse-compton.h:371-399 */
    double* dN_e_boost = malloc(sizeof(double) * g_size);
    for (int i = 0; i < g_size; ++i) {
        const double gi = gamma[i];
        const double dgi = 0.5 * ((i + 1 < g_size ? gamma[i + 1] : gamma[i]) - (i > 0 ? gamma[i - 1] : gamma[i]));
        dN_e_boost[i] = electrons_column_den(el, gi) / (gi * gi) * dgi;
    }
    double* I_nu_seed = malloc(sizeof(double) * nu_size);
    for (int j = 0; j < nu_size; ++j) I_nu_seed[j] = exp2(compute_log2_I_nu(ph, lg2_nu_seed[j]));
    /* compute_IC_spectrum, inverse-compton.h:529-607 */
    const int nu_last = nu_size - 1;
    double* I_buf = calloc(n_ic, sizeof(double));
    double* cdf_buf = calloc(nu_size, sizeof(double));
    double* fv_buf = calloc(nu_size, sizeof(double));
    double* ratio_buf = calloc(nu_size, sizeof(double));
    double* fv_th = malloc(sizeof(double) * nu_size);
    double* lg2fv_th = malloc(sizeof(double) * nu_size);
    double* lg2r = calloc(nu_size, sizeof(double));
    double* inv_lg2r = calloc(nu_size, sizeof(double));
    double* cdf_th = calloc(nu_size, sizeof(double));
    double* ratio_th = calloc(nu_size, sizeof(double));
    for (int j = 0; j < nu_size; ++j) {
        const double f = I_nu_seed[j] / (nu_seed[j] * nu_seed[j]);
        fv_th[j] = f;
        lg2fv_th[j] = (f > 0) ? log2(f) : -INFINITY;
        if (j + 1 < nu_size) {
            lg2r[j] = lg2_nu_seed[j + 1] - lg2_nu_seed[j];
            inv_lg2r[j] = (lg2r[j] != 0) ? 1 / lg2r[j] : 0;
        }
    }
    /* build_cdf_thomson, inverse-compton.h:415-430 */
#define BUILD_CDF_TH_(cdf, ratio)                                                                                      \
    do {                                                                                                               \
        (cdf)[nu_last] = 0;                                                                                            \
        for (int j = nu_last - 1; j >= 0; --j) {                                                                       \
            const double trap = 0.5 * (fv_th[j] + fv_th[j + 1]) * dnu_seed[j];                                         \
            const double exact = power_law_bin_integral(fv_th[j], fv_th[j + 1], nu_seed[j], nu_seed[j + 1], lg2fv_th[j], \
                                                        lg2fv_th[j + 1], lg2r[j], inv_lg2r[j], trap);                  \
            (cdf)[j] = (cdf)[j + 1] + exact;                                                                           \
            (ratio)[j] = (trap > 0) ? exact / trap : 1;                                                                \
        }                                                                                                              \
    } while (0)
    /* accumulate_IC, inverse-compton.h:483-527 */
    const double expq[2] = {exp2(IC_Q * 0.0), exp2(IC_Q * 1.0)};
#define ACCUMULATE_IC_(dNe, n_off_, fv)                                                                   \
    do {                                                                                                  \
        const long ns_top = ((long)nu_size - 1) * (long)IC_NU_MULT;                                       \
        if (cdf_buf[0] <= 0) break;                                                                       \
        const double plateau = (dNe)*cdf_buf[0];                                                          \
        int kk = 0;                                                                                       \
        long nn = (n_off_);                                                                               \
        for (; kk < n_ic && nn < 0; ++kk, nn += IC_IC_MULT) I_buf[kk] += plateau;                         \
        for (; kk < n_ic && nn < ns_top; ++kk, nn += IC_IC_MULT) {                                        \
            const size_t j = (size_t)nn / IC_NU_MULT;                                                     \
            const size_t fi = (size_t)nn % IC_NU_MULT;                                                    \
            const double nu_lo = nu_seed[j], dnu = dnu_seed[j];                                           \
            const double f_lo = (fv)[j], f_hi = (fv)[j + 1];                                              \
            const double frac = (nu_lo * expq[fi] - nu_lo) / dnu;                                         \
            const double rem = 1.0 - frac;                                                                \
            const double f_seed = f_lo * rem + f_hi * frac;                                               \
            I_buf[kk] += (dNe) * (cdf_buf[j + 1] + 0.5 * (f_seed + f_hi) * rem * dnu * ratio_buf[j]);     \
        }                                                                                                 \
    } while (0)
    if (ic->KN) {
        BUILD_CDF_TH_(cdf_th, ratio_th);
        const size_t n_lat = IC_GAMMA_MULT * ((size_t)g_size - 1) + IC_NU_MULT * ((size_t)nu_size - 1) + 1;
        double* corr_lat = malloc(sizeof(double) * n_lat);
        double* lg2corr_lat = malloc(sizeof(double) * n_lat);
        const double lg2_base = log2(gamma[0]) + lg2_nu_seed[0];
        for (size_t k = 0; k < n_lat; ++k)
            compton_correction_pair(exp2(lg2_base + IC_Q * (double)k), &corr_lat[k], &lg2corr_lat[k]);
        for (int i = 0; i < g_size; ++i) {
            if (dN_e_boost[i] <= 0) continue;
            /* build_cdf_KN, inverse-compton.h:432-481 */
            const size_t i_gamma = IC_GAMMA_MULT * (size_t)i;
            const double nu_split = 1e-4 * (C_ME * C_C2 / C_H) / gamma[i];
            int j_split = 0;
            while (j_split < nu_last && nu_seed[j_split] < nu_split) ++j_split;
            double corr = corr_lat[i_gamma + IC_NU_MULT * (size_t)nu_last];
            double lg2_corr = lg2corr_lat[i_gamma + IC_NU_MULT * (size_t)nu_last];
            fv_buf[nu_last] = fv_th[nu_last] * corr;
            double lg2f_hi = lg2fv_th[nu_last] + lg2_corr;
            cdf_buf[nu_last] = 0;
            for (int j = nu_last - 1; j >= j_split; --j) {
                corr = corr_lat[i_gamma + IC_NU_MULT * (size_t)j];
                lg2_corr = lg2corr_lat[i_gamma + IC_NU_MULT * (size_t)j];
                fv_buf[j] = fv_th[j] * corr;
                const double lg2f_lo = lg2fv_th[j] + lg2_corr;
                const double trap = 0.5 * (fv_buf[j] + fv_buf[j + 1]) * dnu_seed[j];
                const double exact = power_law_bin_integral(fv_buf[j], fv_buf[j + 1], nu_seed[j], nu_seed[j + 1], lg2f_lo,
                                                            lg2f_hi, lg2r[j], inv_lg2r[j], trap);
                cdf_buf[j] = cdf_buf[j + 1] + exact;
                ratio_buf[j] = (trap > 0) ? exact / trap : 1;
                lg2f_hi = lg2f_lo;
            }
            if (j_split > 0) {
                const double delta = cdf_buf[j_split] - cdf_th[j_split];
                for (int j = j_split - 1; j >= 0; --j) {
                    fv_buf[j] = fv_th[j];
                    ratio_buf[j] = ratio_th[j];
                    cdf_buf[j] = cdf_th[j] + delta;
                }
            }
            ACCUMULATE_IC_(dN_e_boost[i], ic->ic_idx0 - 2 * (long)IC_GAMMA_MULT * i, fv_buf);
        }
        free(corr_lat);
        free(lg2corr_lat);
    } else {
        BUILD_CDF_TH_(cdf_buf, ratio_buf);
        for (int i = 0; i < g_size; ++i) {
            if (dN_e_boost[i] <= 0) continue;
            ACCUMULATE_IC_(dN_e_boost[i], ic->ic_idx0 - 2 * (long)IC_GAMMA_MULT * i, fv_th);
        }
    }
    if (g_ic_tally.on) { /* second walk against the finished I_buf */
        const double thr = exp2(-g_ic_tally.bits);
        g_ic_tally.cells += 1;
        if (!ic->KN) BUILD_CDF_TH_(cdf_buf, ratio_buf);
        for (int i = 0; i < g_size; ++i) {
            if (dN_e_boost[i] <= 0) continue;
            if (ic->KN) { /* the energy's own CDF: the KN-corrected bins above the split, the Thomson ones below (build_cdf_KN) */
                const size_t n_lat_unused = 0;
                (void)n_lat_unused;
                const double lg2_base = log2(gamma[0]) + lg2_nu_seed[0];
                const size_t i_gamma = IC_GAMMA_MULT * (size_t)i;
                const double nu_split = 1e-4 * (C_ME * C_C2 / C_H) / gamma[i];
                int j_split = 0;
                while (j_split < nu_last && nu_seed[j_split] < nu_split) ++j_split;
                double corr, lg2_corr;
                compton_correction_pair(exp2(lg2_base + IC_Q * (double)(i_gamma + IC_NU_MULT * (size_t)nu_last)), &corr, &lg2_corr);
                fv_buf[nu_last] = fv_th[nu_last] * corr;
                double lg2f_hi = lg2fv_th[nu_last] + lg2_corr;
                cdf_buf[nu_last] = 0;
                for (int j = nu_last - 1; j >= j_split; --j) {
                    compton_correction_pair(exp2(lg2_base + IC_Q * (double)(i_gamma + IC_NU_MULT * (size_t)j)), &corr, &lg2_corr);
                    fv_buf[j] = fv_th[j] * corr;
                    const double lg2f_lo = lg2fv_th[j] + lg2_corr;
                    const double trap = 0.5 * (fv_buf[j] + fv_buf[j + 1]) * dnu_seed[j];
                    cdf_buf[j] = cdf_buf[j + 1] + power_law_bin_integral(fv_buf[j], fv_buf[j + 1], nu_seed[j], nu_seed[j + 1], lg2f_lo, lg2f_hi,
                                                                         lg2r[j], inv_lg2r[j], trap);
                    lg2f_hi = lg2f_lo;
                }
                if (j_split > 0) {
                    const double delta = cdf_buf[j_split] - cdf_th[j_split];
                    for (int j = j_split - 1; j >= 0; --j) cdf_buf[j] = cdf_th[j] + delta;
                }
            }
            const long n_off = ic->ic_idx0 - 2 * (long)IC_GAMMA_MULT * i;
            int all_dead = 1;
            for (int m = 0; m < nu_last; ++m) {
                const double term = dN_e_boost[i] * (cdf_buf[m] - cdf_buf[m + 1]);
                /* the last output node whose lattice position nn = n_off + IC_IC_MULT kk falls into bin m or below it */
                long kk_max = ((long)m * IC_NU_MULT + (IC_NU_MULT - 1) - n_off) / (long)IC_IC_MULT;
                if ((long)m * IC_NU_MULT + (IC_NU_MULT - 1) - n_off < 0) kk_max = -1;
                if (kk_max > n_ic - 1) kk_max = n_ic - 1;
                const int dead = kk_max < 0 || !(term >= thr * I_buf[kk_max]);
                g_ic_tally.terms += 1;
                g_ic_tally.terms_negligible += dead;
                all_dead &= dead;
            }
            g_ic_tally.energies += 1;
            g_ic_tally.energies_dead += all_dead;
        }
    }
#undef BUILD_CDF_TH_
#undef ACCUMULATE_IC_
    const double log2_scale = log2(0.25 * C_SIGMAT);
    for (int i = 0; i < n_ic; ++i) {
        ic->log2_I_nu_IC[i] = log2(I_buf[i]) + ic->log2_nu_IC[i] + log2_scale;
        if (i > 0) {
            const double dl = ic->log2_nu_IC[i] - ic->log2_nu_IC[i - 1];
            ic->interp_slope[i - 1] = (dl != 0) ? (ic->log2_I_nu_IC[i] - ic->log2_I_nu_IC[i - 1]) / dl : 0;
        }
    }
    free(nu_seed); free(lg2_nu_seed); free(dnu_seed); free(gamma); free(lg2_gamma); free(dN_e_boost); free(I_nu_seed);
    free(I_buf); free(cdf_buf); free(fv_buf); free(ratio_buf); free(fv_th); free(lg2fv_th); free(lg2r); free(inv_lg2r);
    free(cdf_th); free(ratio_th);
    ic->generated = 1;
}

/* ICPhoton::compute_log2_I_nu, inverse-compton.h:614-652 (band-contract rebuild included) */
static double icphoton_log2_I_nu(icphoton_t* ic, double log2_nu) {
    if (!ic->generated) icphoton_generate(ic);
    int n = ic->n_ic;
    if (n >= 2 && ((log2_nu > ic->log2_nu_IC[n - 1] && log2_nu < ic->log2_nu_theory_max) ||
                   (log2_nu < ic->log2_nu_IC[0] && log2_nu > ic->log2_nu_theory_min))) {
        ic->nu_eval_min = 0;
        ic->nu_eval_max = INFINITY;
        icphoton_generate(ic);
        n = ic->n_ic;
    }
    if (n < 2 || log2_nu > ic->log2_nu_IC[n - 1]) return -INFINITY;
    int idx = 0;
    while (idx + 2 < n && ic->log2_nu_IC[idx + 1] <= log2_nu) ++idx;
    return ic->log2_I_nu_IC[idx] + (log2_nu - ic->log2_nu_IC[idx]) * ic->interp_slope[idx];
}

/* ------------------------------------------------------------------------------------------
 * Flux integration: src/core/observer.h:309-338 (iterate_to / observed_window),
 * 355-445 (specific_flux), 447-538 (specific_flux_series), 555-567 (flux);
 * Boole weights src/core/quadrature.h:153-196
 * ---------------------------------------------------------------------------------------- */
static void iterate_to(double value, const double* arr, int n, int* it) {
    while (*it < n && arr[*it] < value) (*it)++;
}
static void iterate_through(double value, const double* arr, int n, int* it) {
    while (*it < n && arr[*it] <= value) (*it)++;
}
static int observed_window(const double* t_row, int t_grid, double w_lo, double w_hi, int* k_lo, int* k_hi) {
    if (t_row[t_grid - 1] < w_lo || t_row[0] > w_hi) return 0;
    *k_lo = 0;
    while (*k_lo + 1 < t_grid && t_row[*k_lo + 1] < w_lo) (*k_lo)++;
    *k_hi = *k_lo + 1;
    while (*k_hi + 1 < t_grid && t_row[*k_hi] <= w_hi) (*k_hi)++;
    return 1;
}

/* photon-grid evaluator: log2 I_nu'(cell (j,k), log2 nu') -- synchrotron cells or SSC tables */
typedef double (*cell_eval_fn)(void* grid, int j, int k, int t_grid, double log2_nu);
static double eval_syn_cell(void* grid, int j, int k, int t_grid, double log2_nu) {
    return compute_log2_I_nu((const photons_t*)grid + (size_t)j * t_grid + k, log2_nu);
}
static double eval_ic_cell(void* grid, int j, int k, int t_grid, double log2_nu) {
    return icphoton_log2_I_nu((icphoton_t*)grid + (size_t)j * t_grid + k, log2_nu);
}

/* Instrumentation of specific_flux (round 6): with a reference F of the same request (code units, as specific_flux returns
 * it) set, the pass counts the boundary evaluations B[l][k] of every row's window (the flux kernels' unit of work: n_evals) and those
 * among them that are NOT NEEDED -- every interpolated term they enter lies below 2^-bits of the final flux of its (nu, t) bin, or they
 * serve no requested time at all -- and the same for the interpolated terms themselves.  Not part of any checked result. */
static struct {
    const double* F_ref;
    double bits;
    long long n_evals, n_evals_negligible, n_interps, n_interps_negligible;
} g_flux_tally = {NULL, 60.0, 0, 0, 0, 0};

/* F[nnu][nt] in code units */
static void specific_flux(const eat_t* o, cell_eval_fn eval, void* grid, const double* t_obs, int nt_obs, const double* nu_obs,
                          int nnu, double* F) {
    const int t_grid = o->n_t;
    double* lg2_t_obs = malloc(sizeof(double) * nt_obs);
    double* lg2_nu_src = malloc(sizeof(double) * nnu);
    for (int i = 0; i < nt_obs; ++i) lg2_t_obs[i] = log2(t_obs[i]);
    for (int l = 0; l < nnu; ++l) lg2_nu_src[l] = log2(nu_obs[l]) + log2(o->one_plus_z);
    for (size_t q = 0; q < (size_t)nnu * nt_obs; ++q) F[q] = 0;
    double* boundary = calloc((size_t)t_grid * nnu, sizeof(double));
    double* slope = calloc(nnu, sizeof(double));
    double* lo = calloc(nnu, sizeof(double));
    double* col = calloc((size_t)nnu * nt_obs, sizeof(double));
    unsigned char* needed = g_flux_tally.F_ref ? calloc((size_t)t_grid * nnu, 1) : NULL;
    const double tally_norm = o->one_plus_z / (o->lumi_dist * o->lumi_dist);

    for (int i = 0; i < o->n_phi_eff; ++i) {
        for (int j = 0; j < o->n_theta; ++j) {
            const size_t row = ((size_t)i * o->n_theta + j) * t_grid;
            const int cell_row = (o->phi_size > 1 ? i : 0) * o->n_theta + j; /* photons: eff_i = i * jet_3d, observer.h:365 */
            const double* t_row = o->lg2_t + row;
            const double* dop_row = o->lg2_doppler + row;
            const double* geom_row = o->lg2_geom + row;
            int k_lo, k_hi;
            if (!observed_window(t_row, t_grid, lg2_t_obs[0], lg2_t_obs[nt_obs - 1], &k_lo, &k_hi)) continue;
            for (int k = k_lo; k <= k_hi; ++k)
                for (int l = 0; l < nnu; ++l)
                    boundary[(size_t)k * nnu + l] = eval(grid, cell_row, k, t_grid, lg2_nu_src[l] - dop_row[k]) + geom_row[k];

            int t_idx = 0;
            iterate_to(t_row[0], lg2_t_obs, nt_obs, &t_idx);
            const int col_first = t_idx;
            for (int k = k_lo; k < k_hi && t_idx < nt_obs; ++k) {
                if (t_row[k + 1] < lg2_t_obs[t_idx]) continue;
                const int idx_start = t_idx;
                iterate_to(t_row[k + 1], lg2_t_obs, nt_obs, &t_idx);
                const double t_lo_val = t_row[k];
                const double inv_t_ratio = 1.0 / (t_row[k + 1] - t_lo_val);
                for (int l = 0; l < nnu; ++l) {
                    const double s = (boundary[(size_t)(k + 1) * nnu + l] - boundary[(size_t)k * nnu + l]) * inv_t_ratio;
                    const int ok = isfinite(s);
                    slope[l] = ok ? s : 0.0;
                    lo[l] = ok ? boundary[(size_t)k * nnu + l] : -INFINITY;
                }
                for (int idx = idx_start; idx < t_idx; ++idx) {
                    const double dlg2_t = lg2_t_obs[idx] - t_lo_val;
                    for (int l = 0; l < nnu; ++l) col[(size_t)l * nt_obs + idx] = lo[l] + dlg2_t * slope[l];
                }
                if (needed)
                    for (int idx = idx_start; idx < t_idx; ++idx)
                        for (int l = 0; l < nnu; ++l) {
                            const double thr = log2(g_flux_tally.F_ref[(size_t)l * nt_obs + idx] / tally_norm) - g_flux_tally.bits;
                            const int counts = col[(size_t)l * nt_obs + idx] >= thr;
                            g_flux_tally.n_interps += 1;
                            g_flux_tally.n_interps_negligible += !counts;
                            if (counts) needed[(size_t)k * nnu + l] = needed[(size_t)(k + 1) * nnu + l] = 1;
                        }
            }
            if (needed)
                for (int k = k_lo; k <= k_hi; ++k)
                    for (int l = 0; l < nnu; ++l) {
                        g_flux_tally.n_evals += 1;
                        g_flux_tally.n_evals_negligible += !needed[(size_t)k * nnu + l];
                        needed[(size_t)k * nnu + l] = 0;
                    }
            for (int l = 0; l < nnu; ++l)
                for (int idx = col_first; idx < t_idx; ++idx)
                    F[(size_t)l * nt_obs + idx] += exp2(col[(size_t)l * nt_obs + idx]);
        }
    }
    const double norm = o->one_plus_z / (o->lumi_dist * o->lumi_dist);
    for (size_t q = 0; q < (size_t)nnu * nt_obs; ++q) F[q] *= norm;
    free(needed);
    free(lg2_t_obs);
    free(lg2_nu_src);
    free(boundary);
    free(slope);
    free(lo);
    free(col);
}

/* F[n] in code units */
static void specific_flux_series(const eat_t* o, cell_eval_fn eval, void* grid, const double* t_obs, const double* nu_obs,
                                 int n, double* F) {
    const int t_grid = o->n_t;
    double* lg2_t_obs = malloc(sizeof(double) * n);
    double* lg2_nu_src = malloc(sizeof(double) * n);
    double* col = malloc(sizeof(double) * n);
    for (int i = 0; i < n; ++i) {
        lg2_t_obs[i] = log2(t_obs[i]);
        lg2_nu_src[i] = log2(nu_obs[i]) + log2(o->one_plus_z);
        F[i] = 0;
        col[i] = -INFINITY;
    }
    for (int i = 0; i < o->n_phi_eff; ++i) {
        for (int j = 0; j < o->n_theta; ++j) {
            const size_t row = ((size_t)i * o->n_theta + j) * t_grid;
            const int cell_row = (o->phi_size > 1 ? i : 0) * o->n_theta + j;
            const double* t_row = o->lg2_t + row;
            const double* dop_row = o->lg2_doppler + row;
            const double* geom_row = o->lg2_geom + row;
            int idx = 0;
            iterate_to(t_row[0], lg2_t_obs, n, &idx);
            const int col_first = idx;
            int carry_k = t_grid;
            double carry_nu = 0, carry_val = 0;
            for (int k = 0; idx < n && k < t_grid - 1; ++k) {
                if (t_row[k + 1] < lg2_t_obs[idx]) continue;
                const int block_first = idx;
                iterate_through(t_row[k + 1], lg2_t_obs, n, &idx);
                const double inv_dt = 1.0 / (t_row[k + 1] - t_row[k]);
                double prev_nu = NAN, prev_lo = 0, prev_hi = 0;
                for (int s = block_first; s < idx; ++s) {
                    const double lg2_nu = lg2_nu_src[s];
                    double lo, hi;
                    if (lg2_nu == prev_nu) {
                        lo = prev_lo;
                        hi = prev_hi;
                    } else {
                        const int carried = (s == block_first && carry_k == k && carry_nu == lg2_nu);
                        lo = carried ? carry_val : eval(grid, cell_row, k, t_grid, lg2_nu - dop_row[k]) + geom_row[k];
                        hi = eval(grid, cell_row, k + 1, t_grid, lg2_nu - dop_row[k + 1]) + geom_row[k + 1];
                        prev_nu = lg2_nu;
                        prev_lo = lo;
                        prev_hi = hi;
                    }
                    const double sl = (hi - lo) * inv_dt;
                    col[s] = isfinite(sl) ? lo + (lg2_t_obs[s] - t_row[k]) * sl : -INFINITY;
                    carry_k = k + 1;
                    carry_nu = lg2_nu;
                    carry_val = hi;
                }
            }
            for (int s = col_first; s < idx; ++s) F[s] += exp2(col[s]);
        }
    }
    const double norm = o->one_plus_z / (o->lumi_dist * o->lumi_dist);
    for (int i = 0; i < n; ++i) F[i] *= norm;
    free(lg2_t_obs);
    free(lg2_nu_src);
    free(col);
}

/* compute_boole_weights, src/core/quadrature.h:153-196 */
static void compute_boole_weights(const double* grid, int n, double* w) {
    for (int i = 0; i < n; ++i) w[i] = 0;
    if (n < 2) return;
    const double h = log(grid[1] / grid[0]);
    const double cb = 2.0 * h / 45.0;
    int j = 0;
    for (; j + 4 < n; j += 4) {
        w[j] += cb * 7;
        w[j + 1] += cb * 32;
        w[j + 2] += cb * 12;
        w[j + 3] += cb * 32;
        w[j + 4] += cb * 7;
    }
    const int remaining = n - 1 - j;
    if (remaining == 3) {
        const double c38 = 3.0 * h / 8.0;
        w[j] += c38;
        w[j + 1] += c38 * 3;
        w[j + 2] += c38 * 3;
        w[j + 3] += c38;
    } else if (remaining == 2) {
        const double c13 = h / 3.0;
        w[j] += c13;
        w[j + 1] += c13 * 4;
        w[j + 2] += c13;
    } else if (remaining == 1) {
        w[j] += 0.5 * h;
        w[j + 1] += 0.5 * h;
    }
    for (int i = 0; i < n; ++i) w[i] *= grid[i];
}

/* ------------------------------------------------------------------------------------------
 * Validation: pybind/error_handling.h:31-69 macros applied as in pybind/pymodel.cpp:47-186,
 * pybind/pymodel.h:205-260,613-649
 * ---------------------------------------------------------------------------------------- */
static int finite_pos(double x) {
    return isfinite(x) && x > 0;
}
static int range_oi(double x, double lo, double hi) { /* (lo, hi] */
    return isfinite(x) && x > lo && x <= hi;
}

int vag_oracle_params_validate(const vag_model_params* p) {
    if (p->jet_type < 0 || p->jet_type > VAG_JET_POWERLAW_WING) return fail("unknown jet_type");
    if (p->jet_type == VAG_JET_MAGNETIZED_TOPHAT && !(isfinite(p->sigma0) && p->sigma0 >= 0))
        return fail("sigma0 must be finite and non-negative");
    if (p->medium_type < 0 || p->medium_type > VAG_MEDIUM_WIND) return fail("unknown medium_type");
    if (!range_oi(p->theta_c, 0.0, C_PI / 2)) return fail("theta_c must be in (0, pi/2]");
    if (p->jet_type != VAG_JET_POWERLAW_WING) { /* PowerLawWing has no core (pymodel.cpp:90-110) */
        if (!finite_pos(p->E_iso)) return fail("E_iso must be positive and finite");
        if (!(isfinite(p->Gamma0) && p->Gamma0 > 1.0)) return fail("Gamma0 must be > 1");
    }
    if (!finite_pos(p->duration)) return fail("duration must be positive and finite");
    if (p->jet_type == VAG_JET_STEP_POWERLAW || p->jet_type == VAG_JET_POWERLAW_WING) {
        if (!finite_pos(p->E_iso_w)) return fail("E_iso_w must be positive and finite");
        if (!(isfinite(p->Gamma0_w) && p->Gamma0_w > 1.0)) return fail("Gamma0_w must be > 1");
    }
    if (p->jet_type == VAG_JET_POWERLAW || p->jet_type == VAG_JET_STEP_POWERLAW || p->jet_type == VAG_JET_POWERLAW_WING) {
        if (!finite_pos(p->k_e)) return fail("k_e must be positive and finite");
        if (!finite_pos(p->k_g)) return fail("k_g must be positive and finite");
    }
    if (p->jet_type == VAG_JET_TWO_COMPONENT) {
        if (!range_oi(p->theta_w, 0.0, C_PI / 2)) return fail("theta_w must be in (0, pi/2]");
        if (!(p->theta_w > p->theta_c)) return fail("theta_w (wing angle) must be greater than theta_c (core angle)");
        if (!finite_pos(p->E_iso_w)) return fail("E_iso_w must be positive and finite");
        if (!(isfinite(p->Gamma0_w) && p->Gamma0_w > 1.0)) return fail("Gamma0_w must be > 1");
    }
    if (p->medium_type == VAG_MEDIUM_ISM) {
        if (!(isfinite(p->n_ism) && p->n_ism >= 0)) return fail("n_ism must be non-negative and finite");
    } else {
        if (!finite_pos(p->A_star)) return fail("A_star must be positive and finite");
        if (!(isfinite(p->n_ism) && p->n_ism >= 0)) return fail("n_ism must be non-negative and finite");
        if (!(p->n0 > 0)) return fail("n0 must be > 0 (or +inf for no floor)");
        if (!finite_pos(p->k_m)) return fail("k_m must be positive and finite");
    }
    if (!finite_pos(p->lumi_dist)) return fail("lumi_dist must be positive and finite");
    if (!(isfinite(p->z) && p->z >= 0)) return fail("z must be non-negative and finite");
    if (!(isfinite(p->theta_obs) && p->theta_obs >= 0 && p->theta_obs <= C_PI)) return fail("theta_obs must be in [0, pi]");
    if (!range_oi(p->eps_e, 0.0, 1.0)) return fail("eps_e must be in (0, 1]");
    if (!range_oi(p->eps_B, 0.0, 1.0)) return fail("eps_B must be in (0, 1]");
    if (!range_oi(p->xi_e, 0.0, 1.0)) return fail("xi_e must be in (0, 1]");
    if (!(isfinite(p->p) && p->p > 1.0)) return fail("p must be > 1");
    if (p->flags & VAG_FLAG_RVS) { /* rvs_rad is a Radiation too: same checks (pymodel.h:241-260) */
        if (!range_oi(p->rvs_eps_e, 0.0, 1.0)) return fail("rvs eps_e must be in (0, 1]");
        if (!range_oi(p->rvs_eps_B, 0.0, 1.0)) return fail("rvs eps_B must be in (0, 1]");
        if (!range_oi(p->rvs_xi_e, 0.0, 1.0)) return fail("rvs xi_e must be in (0, 1]");
        if (!(isfinite(p->rvs_p) && p->rvs_p > 1.0)) return fail("rvs p must be > 1");
        if (!finite_pos(p->duration)) return fail("duration must be positive and finite");
    }
    if (p->flags & VAG_FLAG_MAGNETAR) { /* PyMagnetar ctor, pymodel.h:45-49; PowerLawWing takes no magnetar */
        if (!finite_pos(p->mag_L0) || !finite_pos(p->mag_t0) || !finite_pos(p->mag_q)) return fail("magnetar L0, t0, q must be positive and finite");
        if (p->jet_type == VAG_JET_POWERLAW_WING || p->jet_type == VAG_JET_MAGNETIZED_TOPHAT)
            return fail("this jet type takes no magnetar");
    }
    if (p->flags & ~(VAG_FLAG_SSC | VAG_FLAG_KN | VAG_FLAG_RVS | VAG_FLAG_RVS_SSC | VAG_FLAG_RVS_KN | VAG_FLAG_SPREADING |
                     VAG_FLAG_MAGNETAR | VAG_FLAG_NON_AXISYMMETRIC))
        return fail("unknown bits set in flags");
    if (!(isfinite(p->rtol) && p->rtol > 0 && p->rtol < 1)) return fail("rtol must be in (0, 1)");
    if (!finite_pos(p->phi_resol) || !finite_pos(p->theta_resol) || !finite_pos(p->t_resol))
        return fail("resolutions must be positive and finite");
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Orchestration: PyModel::compute_emission / single_shock_emission, pybind/pymodel.h:873-961
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    jet_t jet;
    medium_t med;
    coord_t coord;
    shock_t shock;
    eat_t eat;
    electrons_t* el;
    photons_t* ph;
    /* reverse shock (Model(rvs_rad=...), pymodel.h:940-958): shares coord and the EAT grids */
    int has_rvs;
    shock_t rvs_shock;
    electrons_t* rvs_el;
    photons_t* rvs_ph;
} pipeline_t;

/* one emitting region: forward (0) or reverse (1) shock */
typedef struct {
    electrons_t* el;
    photons_t* ph;
    int ssc, kn;
} emitter_t;

static int pipeline_emitters(const pipeline_t* pl, const vag_model_params* p, emitter_t em[2]) {
    em[0].el = pl->el;
    em[0].ph = pl->ph;
    em[0].ssc = (p->flags & VAG_FLAG_SSC) != 0;
    em[0].kn = (p->flags & VAG_FLAG_KN) != 0;
    if (!pl->has_rvs) return 1;
    em[1].el = pl->rvs_el;
    em[1].ph = pl->rvs_ph;
    em[1].ssc = (p->flags & VAG_FLAG_RVS_SSC) != 0;
    em[1].kn = (p->flags & VAG_FLAG_RVS_KN) != 0;
    return 2;
}

static void pipeline_free(pipeline_t* pl) {
    coord_free(&pl->coord);
    shock_free(&pl->shock);
    if (pl->has_rvs) shock_free(&pl->rvs_shock);
    eat_free(&pl->eat);
    free(pl->el);
    free(pl->ph);
    free(pl->rvs_el);
    free(pl->rvs_ph);
    pl->el = NULL;
    pl->ph = NULL;
    pl->rvs_el = NULL;
    pl->rvs_ph = NULL;
}

/* t_obs_min/max in code units */
static int run_pipeline(pipeline_t* pl, const vag_model_params* p, double t_obs_min, double t_obs_max) {
    memset(pl, 0, sizeof *pl);
    if (vag_oracle_params_validate(p) != 0) return -1;
    jet_init(&pl->jet, p);
    medium_init(&pl->med, p);
    pl->has_rvs = (p->flags & VAG_FLAG_RVS) != 0;
    auto_grid(&pl->coord, &pl->jet, &pl->med, t_obs_min, t_obs_max, C_PI / 2, p->theta_obs, p->z, p->phi_resol,
              p->theta_resol, p->t_resol, pl->has_rvs, !(p->flags & VAG_FLAG_NON_AXISYMMETRIC));
    const int rc = pl->has_rvs ? generate_shock_pair(&pl->shock, &pl->rvs_shock, &pl->coord, &pl->med, &pl->jet, p)
                               : generate_fwd_shock(&pl->shock, &pl->coord, &pl->med, &pl->jet, p);
    if (rc != 0) {
        pipeline_free(pl);
        return -1;
    }
    observe(&pl->eat, &pl->coord, &pl->shock, p->lumi_dist * U_CM, p->z);
    const size_t n = (size_t)pl->coord.phi_size * pl->coord.n_theta * pl->coord.n_t;
    pl->el = calloc(n, sizeof(electrons_t));
    pl->ph = calloc(n, sizeof(photons_t));
    generate_syn(pl->el, pl->ph, &pl->shock, &pl->coord, p);
    if (pl->has_rvs) { /* same passes with rvs_rad on the reverse shock's arrays */
        vag_model_params rp = *p;
        rp.eps_e = p->rvs_eps_e;
        rp.eps_B = p->rvs_eps_B;
        rp.p = p->rvs_p;
        rp.xi_e = p->rvs_xi_e;
        rp.flags = ((p->flags & VAG_FLAG_RVS_SSC) ? VAG_FLAG_SSC : 0) | ((p->flags & VAG_FLAG_RVS_KN) ? VAG_FLAG_KN : 0);
        pl->rvs_el = calloc(n, sizeof(electrons_t));
        pl->rvs_ph = calloc(n, sizeof(photons_t));
        generate_syn(pl->rvs_el, pl->rvs_ph, &pl->rvs_shock, &pl->coord, &rp);
    }
    return 0;
}

/* generate_IC_photons with the per-k observation-band clamp of single_shock_emission
 * (pybind/pymodel.h:896-914, inverse-compton.h:656-690): one ICPhoton per (theta, k), deep-copied across each
 * symmetry group like broadcast_symmetry does. */
static icphoton_t* make_ic_photons(pipeline_t* pl, const emitter_t* em, const double* nu_obs, int nnu) {
    const coord_t* c = &pl->coord;
    const eat_t* o = &pl->eat;
    const int nth = c->n_theta, nt = c->n_t;
    const int kn = em->kn;
    double nu_lo = nu_obs[0], nu_hi = nu_obs[0];
    for (int l = 1; l < nnu; ++l) {
        if (nu_obs[l] < nu_lo) nu_lo = nu_obs[l];
        if (nu_obs[l] > nu_hi) nu_hi = nu_obs[l];
    }
    const double lg2_1pz = log2(o->one_plus_z);
    const double lg2_nu_lo = log2(nu_lo) + lg2_1pz, lg2_nu_hi = log2(nu_hi) + lg2_1pz;
    const int n_rows = c->phi_size * nth;
    icphoton_t* ic = calloc((size_t)n_rows * nt, sizeof(icphoton_t));
    for (int k = 0; k < nt; ++k) {
        double dmin_k = INFINITY, dmax_k = -INFINITY;
        for (int i = 0; i < o->n_phi_eff; ++i)
            for (int j = 0; j < nth; ++j) {
                const double d = o->lg2_doppler[((size_t)i * nth + j) * nt + k];
                if (d < dmin_k) dmin_k = d;
                if (d > dmax_k) dmax_k = d;
            }
        const double nu_eval_min = exp2(lg2_nu_lo - dmax_k), nu_eval_max = exp2(lg2_nu_hi - dmin_k);
        for (int r = 0; r < c->n_reps; ++r) {
            const int j0 = c->reps[r];
            const int j1 = (r + 1 < c->n_reps) ? c->reps[r + 1] : n_rows;
            icphoton_t* q = &ic[(size_t)j0 * nt + k];
            q->electrons = em->el[(size_t)j0 * nt + k];
            q->photons = em->ph[(size_t)j0 * nt + k];
            q->KN = kn;
            q->nu_eval_min = nu_eval_min;
            q->nu_eval_max = nu_eval_max;
            q->log2_nu_theory_max = INFINITY;
            q->log2_nu_theory_min = -INFINITY;
            icphoton_generate(q); /* precompute = true: symmetry != structured */
            for (int j = j0 + 1; j < j1; ++j) icphoton_copy(&ic[(size_t)j * nt + k], q);
        }
    }
    return ic;
}
static void free_ic_photons(icphoton_t* ic, size_t n) {
    for (size_t q = 0; q < n; ++q) icphoton_free(&ic[q]);
    free(ic);
}

static int check_times(const double* t, int nt) {
    if (nt <= 0) return fail("time array must be non-empty");
    for (int i = 1; i < nt; ++i)
        if (t[i] < t[i - 1]) return fail("time array must be in ascending order");
    return 0;
}

static void minmax(const double* a, int n, double* lo, double* hi) {
    *lo = a[0];
    *hi = a[0];
    for (int i = 1; i < n; ++i) {
        if (a[i] < *lo) *lo = a[i];
        if (a[i] > *hi) *hi = a[i];
    }
}

/* fwd.sync -> out_sync, fwd.ssc -> out_ssc (may be NULL), each [nnu][nt] */
/* All four FluxDict components of the grid (pybind/pybind.cpp:472-483): out4 = {fwd.sync, fwd.ssc, rvs.sync, rvs.ssc},
 * each [nnu][nt]; NULL entries are skipped, disabled components are zeros. */
int vag_oracle_flux_density_grid_components4(const vag_model_params* p, const double* t, int nt, const double* nu, int nnu,
                                             double* const* out4) {
    if (check_times(t, nt) != 0) return -1;
    if (nnu <= 0) return fail("frequency array must be non-empty");
    double* t_obs = malloc(sizeof(double) * nt);
    double* nu_obs = malloc(sizeof(double) * nnu);
    for (int i = 0; i < nt; ++i) t_obs[i] = t[i] * U_SEC;
    for (int l = 0; l < nnu; ++l) nu_obs[l] = nu[l] * U_HZ;
    double lo, hi;
    minmax(t_obs, nt, &lo, &hi);
    pipeline_t pl;
    int rc = run_pipeline(&pl, p, lo, hi);
    if (rc == 0) {
        const size_t nout = (size_t)nnu * nt;
        for (int c = 0; c < 4; ++c)
            if (out4[c])
                for (size_t q = 0; q < nout; ++q) out4[c][q] = 0;
        emitter_t em[2];
        const int n_em = pipeline_emitters(&pl, p, em);
        for (int e = 0; e < n_em; ++e) {
            double* sync = out4[2 * e];
            double* ssc = out4[2 * e + 1];
            if (sync) {
                specific_flux(&pl.eat, eval_syn_cell, em[e].ph, t_obs, nt, nu_obs, nnu, sync);
                for (size_t q = 0; q < nout; ++q) sync[q] = sync[q] / U_FLUX_DEN_CGS;
            }
            if (ssc && em[e].ssc) {
                const size_t ncell = (size_t)pl.coord.phi_size * pl.coord.n_theta * pl.coord.n_t;
                icphoton_t* ic = make_ic_photons(&pl, &em[e], nu_obs, nnu);
                specific_flux(&pl.eat, eval_ic_cell, ic, t_obs, nt, nu_obs, nnu, ssc);
                for (size_t q = 0; q < nout; ++q) ssc[q] = ssc[q] / U_FLUX_DEN_CGS;
                free_ic_photons(ic, ncell);
            }
        }
        pipeline_free(&pl);
    }
    free(t_obs);
    free(nu_obs);
    return rc;
}

int vag_oracle_flux_density_grid_components(const vag_model_params* p, const double* t, int nt, const double* nu, int nnu,
                                            double* out, double* out_ssc) {
    double* out4[4] = {out, out_ssc, NULL, NULL};
    return vag_oracle_flux_density_grid_components4(p, t, nt, nu, nnu, out4);
}

/* Instrumentation (g_flux_tally above): counts[4] = {boundary evaluations, of which not needed at 2^-bits, interpolated terms, of which
 * below 2^-bits of their bin's final flux} for a forward-shock synchrotron grid request.  Not thread-safe; used by
 * profiles/debug/negligible_work.py only. */
int vag_oracle_flux_tally(const vag_model_params* p, const double* t, int nt, const double* nu, int nnu, double bits, long long* counts) {
    if (p->flags & (VAG_FLAG_SSC | VAG_FLAG_RVS)) return VAG_E_INVALID;
    const size_t n = (size_t)nnu * nt;
    double* ref = malloc(sizeof(double) * n);
    double* again = malloc(sizeof(double) * n);
    double* out4[4] = {ref, NULL, NULL, NULL};
    int rc = vag_oracle_flux_density_grid_components4(p, t, nt, nu, nnu, out4);
    if (rc == 0) {
        for (size_t q = 0; q < n; ++q) ref[q] *= U_FLUX_DEN_CGS; /* back to what specific_flux returns */
        g_flux_tally.F_ref = ref;
        g_flux_tally.bits = bits;
        g_flux_tally.n_evals = g_flux_tally.n_evals_negligible = g_flux_tally.n_interps = g_flux_tally.n_interps_negligible = 0;
        out4[0] = again;
        rc = vag_oracle_flux_density_grid_components4(p, t, nt, nu, nnu, out4);
        g_flux_tally.F_ref = NULL;
        counts[0] = g_flux_tally.n_evals, counts[1] = g_flux_tally.n_evals_negligible;
        counts[2] = g_flux_tally.n_interps, counts[3] = g_flux_tally.n_interps_negligible;
    }
    free(ref);
    free(again);
    return rc;
}

/* Instrumentation (g_ic_tally above): counts[5] = {cells with a table, (energy, seed bin) terms, of which not needed at 2^-bits, (cell,
 * energy) walks, of which every term is not needed} over all SSC table builds of one grid request.  Used by
 * profiles/debug/negligible_work.py only. */
int vag_oracle_ssc_tally(const vag_model_params* p, const double* t, int nt, const double* nu, int nnu, double bits, long long* counts) {
    const size_t n = (size_t)nnu * nt;
    double* buf = malloc(sizeof(double) * n * 4);
    double* out4[4] = {buf, buf + n, (p->flags & VAG_FLAG_RVS) ? buf + 2 * n : NULL,
                       ((p->flags & VAG_FLAG_RVS) && (p->flags & VAG_FLAG_RVS_SSC)) ? buf + 3 * n : NULL};
    g_ic_tally.on = 1;
    g_ic_tally.bits = bits;
    g_ic_tally.cells = g_ic_tally.terms = g_ic_tally.terms_negligible = g_ic_tally.energies = g_ic_tally.energies_dead = 0;
    const int rc = vag_oracle_flux_density_grid_components4(p, t, nt, nu, nnu, out4);
    g_ic_tally.on = 0;
    counts[0] = g_ic_tally.cells, counts[1] = g_ic_tally.terms, counts[2] = g_ic_tally.terms_negligible;
    counts[3] = g_ic_tally.energies, counts[4] = g_ic_tally.energies_dead;
    free(buf);
    return rc;
}

/* Model.flux_density_grid: total = fwd.sync + fwd.ssc (PyFlux::calc_total, pymodel.cpp:350-364) */
int vag_oracle_flux_density_grid(const vag_model_params* p, const double* t, int nt, const double* nu, int nnu,
                                 double* out) {
    const size_t n = (size_t)(nnu > 0 ? nnu : 1) * (nt > 0 ? nt : 1);
    const int want[4] = {1, (p->flags & VAG_FLAG_SSC) != 0, (p->flags & VAG_FLAG_RVS) != 0,
                         (p->flags & VAG_FLAG_RVS) && (p->flags & VAG_FLAG_RVS_SSC)};
    double* comp[4] = {out, NULL, NULL, NULL};
    for (int c = 1; c < 4; ++c)
        if (want[c]) comp[c] = malloc(sizeof(double) * n);
    const int rc = vag_oracle_flux_density_grid_components4(p, t, nt, nu, nnu, comp);
    for (int c = 1; c < 4; ++c) {
        if (rc == 0 && comp[c])
            for (size_t q = 0; q < (size_t)nnu * nt; ++q) out[q] += comp[c][q];
        free(comp[c]);
    }
    return rc;
}

int vag_oracle_flux_density(const vag_model_params* p, const double* t, const double* nu, int n, double* out) {
    if (check_times(t, n) != 0) return -1;
    double* t_obs = malloc(sizeof(double) * n);
    double* nu_obs = malloc(sizeof(double) * n);
    for (int i = 0; i < n; ++i) {
        t_obs[i] = t[i] * U_SEC;
        nu_obs[i] = nu[i] * U_HZ;
    }
    double lo, hi;
    minmax(t_obs, n, &lo, &hi);
    pipeline_t pl;
    int rc = run_pipeline(&pl, p, lo, hi);
    if (rc == 0) {
        emitter_t em[2];
        const int n_em = pipeline_emitters(&pl, p, em);
        double* tmp = malloc(sizeof(double) * n);
        for (int e = 0; e < n_em; ++e) {
            specific_flux_series(&pl.eat, eval_syn_cell, em[e].ph, t_obs, nu_obs, n, e == 0 ? out : tmp);
            if (e == 0)
                for (int i = 0; i < n; ++i) out[i] = out[i] / U_FLUX_DEN_CGS;
            else
                for (int i = 0; i < n; ++i) out[i] += tmp[i] / U_FLUX_DEN_CGS;
            if (em[e].ssc) {
                const size_t ncell = (size_t)pl.coord.phi_size * pl.coord.n_theta * pl.coord.n_t;
                icphoton_t* ic = make_ic_photons(&pl, &em[e], nu_obs, n);
                specific_flux_series(&pl.eat, eval_ic_cell, ic, t_obs, nu_obs, n, tmp);
                for (int i = 0; i < n; ++i) out[i] += tmp[i] / U_FLUX_DEN_CGS;
                free_ic_photons(ic, ncell);
            }
        }
        free(tmp);
        pipeline_free(&pl);
    }
    free(t_obs);
    free(nu_obs);
    return rc;
}

/* PyModel::flux_density_exposures: generate_exposure_sampling + series flux + average_exposure_flux,
 * pybind/pymodel.cpp:412-496.  out[n]. */
typedef struct {
    double t;
    int src;
} expo_pt;
static int cmp_expo(const void* a, const void* b) {
    const expo_pt* x = a;
    const expo_pt* y = b;
    if (x->t < y->t) return -1;
    if (x->t > y->t) return 1;
    return (x->src > y->src) - (x->src < y->src); /* stable tie-break: original sample order */
}
int vag_oracle_flux_density_exposures(const vag_model_params* p, const double* t, const double* nu,
                                      const double* expo_time, int n, int num_points, double* out) {
    if (n <= 0) return fail("time array must be non-empty");
    if (num_points < 2) return fail("num_points must be at least 2 to sample within each exposure time");
    for (int i = 0; i < n; ++i)
        if (!(isfinite(expo_time[i]) && expo_time[i] > 0)) return fail("expo_time must be finite and > 0");
    const int total = n * num_points;
    expo_pt* pts = malloc(sizeof(expo_pt) * total);
    double* ts = malloc(sizeof(double) * total);
    double* nus = malloc(sizeof(double) * total);
    double* F = malloc(sizeof(double) * total);
    for (int i = 0, j = 0; i < n; ++i) {
        const double dt = expo_time[i] / (double)(num_points - 1);
        for (int k = 0; k < num_points; ++k, ++j) {
            pts[j].t = t[i] + k * dt;
            pts[j].src = j;
        }
    }
    qsort(pts, total, sizeof(expo_pt), cmp_expo);
    for (int j = 0; j < total; ++j) {
        ts[j] = pts[j].t;
        nus[j] = nu[pts[j].src / num_points];
    }
    int rc = vag_oracle_flux_density(p, ts, nus, total, F);
    if (rc == 0) {
        for (int i = 0; i < n; ++i) out[i] = 0;
        for (int j = 0; j < total; ++j) out[pts[j].src / num_points] += F[j];
        for (int i = 0; i < n; ++i) out[i] /= (double)num_points;
    }
    free(pts);
    free(ts);
    free(nus);
    free(F);
    return rc;
}

int vag_oracle_flux(const vag_model_params* p, const double* t, int nt, double nu_min, double nu_max, int num_nu,
                    double* out) {
    if (check_times(t, nt) != 0) return -1;
    if (!(nu_min > 0)) return fail("nu_min must be positive");
    if (!(nu_max > nu_min)) return fail("nu_max must be greater than nu_min");
    if (num_nu < 2) return fail("num_nu must be at least 2");
    double* t_obs = malloc(sizeof(double) * nt);
    double* nu_obs = malloc(sizeof(double) * num_nu);
    double* w = malloc(sizeof(double) * num_nu);
    double* F = malloc(sizeof(double) * (size_t)num_nu * nt);
    for (int i = 0; i < nt; ++i) t_obs[i] = t[i] * U_SEC;
    logspace10(log10(nu_min * U_HZ), log10(nu_max * U_HZ), num_nu, nu_obs);
    double lo, hi;
    minmax(t_obs, nt, &lo, &hi);
    pipeline_t pl;
    int rc = run_pipeline(&pl, p, lo, hi);
    if (rc == 0) {
        compute_boole_weights(nu_obs, num_nu, w);
        emitter_t em[2];
        const int n_em = pipeline_emitters(&pl, p, em);
        double* band = malloc(sizeof(double) * nt);
        for (int j = 0; j < nt; ++j) out[j] = 0;
        for (int e = 0; e < n_em; ++e) {
            for (int pass = 0; pass < 2; ++pass) { /* Observer::flux per component, then PyFlux::calc_total */
                icphoton_t* ic = NULL;
                if (pass == 1) {
                    if (!em[e].ssc) continue;
                    ic = make_ic_photons(&pl, &em[e], nu_obs, num_nu);
                    specific_flux(&pl.eat, eval_ic_cell, ic, t_obs, nt, nu_obs, num_nu, F);
                } else {
                    specific_flux(&pl.eat, eval_syn_cell, em[e].ph, t_obs, nt, nu_obs, num_nu, F);
                }
                for (int j = 0; j < nt; ++j) band[j] = 0;
                for (int i = 0; i < num_nu; ++i)
                    for (int j = 0; j < nt; ++j) band[j] += F[(size_t)i * nt + j] * w[i];
                if (e == 0 && pass == 0)
                    for (int j = 0; j < nt; ++j) out[j] = band[j] / U_FLUX_CGS;
                else
                    for (int j = 0; j < nt; ++j) out[j] += band[j] / U_FLUX_CGS;
                if (ic) free_ic_photons(ic, (size_t)pl.coord.phi_size * pl.coord.n_theta * pl.coord.n_t);
            }
        }
        free(band);
        pipeline_free(&pl);
    }
    free(t_obs);
    free(nu_obs);
    free(w);
    free(F);
    return rc;
}

static int details_impl(const vag_model_params* p, double t_min, double t_max, vag_details_shape* shape,
                        const vag_details_out* out, double** extra, int n_extra, int* n_phi_eff, const double* probe_lg2_nu,
                        int n_probe, int want_rvs) {
    pipeline_t pl;
    if (run_pipeline(&pl, p, t_min * U_SEC, t_max * U_SEC) != 0) return -1;
    if (want_rvs && !pl.has_rvs) {
        pipeline_free(&pl);
        return fail("model has no reverse shock");
    }
    const shock_t* shock = want_rvs ? &pl.rvs_shock : &pl.shock;
    const electrons_t* els = want_rvs ? pl.rvs_el : pl.el;
    const photons_t* phs = want_rvs ? pl.rvs_ph : pl.ph;
    const coord_t* c = &pl.coord;
    const int nth = c->n_theta, nt = c->n_t;
    shape->n_phi = c->n_phi;
    shape->n_theta = nth;
    shape->n_t = nt;
    shape->n_reps = c->n_reps / c->phi_size; /* Coord::theta_reps: representatives per phi slice */
    shape->symmetry = c->symmetry;
    shape->phi_mirrored = c->phi_mirrored;
    if (n_phi_eff) *n_phi_eff = pl.eat.n_phi_eff;
    if (out) {
        const size_t n = (size_t)nth * nt;
        if (out->phi) memcpy(out->phi, c->phi, sizeof(double) * c->n_phi);
        if (out->theta) memcpy(out->theta, c->theta, sizeof(double) * nth);
#define COPY_(dst, src, scale) \
    if (dst)                   \
        for (size_t q = 0; q < n; ++q) (dst)[q] = (src)[q] * (scale);
        COPY_(out->t_src, c->t, 1 / U_SEC)
        COPY_(out->Gamma, shock->Gamma, 1)
        COPY_(out->r, shock->r, 1 / U_CM)
        COPY_(out->t_comv, shock->t_comv, 1 / U_SEC)
        COPY_(out->B, shock->B, 1 / U_GAUSS)
        COPY_(out->N_p, shock->N_p, 1)
        COPY_(out->Gamma_th, shock->Gamma_th, 1)
#undef COPY_
#define EX_(i) ((extra && (i) < n_extra) ? extra[i] : NULL)
        for (size_t q = 0; q < n; ++q) {
            const electrons_t* e = &els[q];
            const photons_t* ph = &phs[q];
            if (EX_(0)) EX_(0)[q] = e->gamma_m;
            if (EX_(1)) EX_(1)[q] = e->gamma_c;
            if (EX_(2)) EX_(2)[q] = e->gamma_a;
            if (EX_(3)) EX_(3)[q] = e->gamma_M;
            if (EX_(4)) EX_(4)[q] = e->N_e;
            if (EX_(5)) EX_(5)[q] = e->column_den;
            if (EX_(6)) EX_(6)[q] = ph->nu_m;
            if (EX_(7)) EX_(7)[q] = ph->nu_c;
            if (EX_(8)) EX_(8)[q] = ph->nu_a;
            if (EX_(9)) EX_(9)[q] = ph->nu_M;
            if (EX_(10)) EX_(10)[q] = ph->I_nu_max;
            if (EX_(14))
                for (int s = 0; s < n_probe; ++s) EX_(14)[q * n_probe + s] = compute_log2_I_nu(ph, probe_lg2_nu[s]);
            if (EX_(15)) EX_(15)[q] = (double)shock->injection_idx[q / nt];
        }
        const size_t ne = (size_t)pl.eat.n_phi_eff * n;
        if (EX_(11)) memcpy(EX_(11), pl.eat.lg2_t, sizeof(double) * ne);
        if (EX_(12)) memcpy(EX_(12), pl.eat.lg2_doppler, sizeof(double) * ne);
        if (EX_(13)) memcpy(EX_(13), pl.eat.lg2_geom, sizeof(double) * ne);
#undef EX_
    }
    pipeline_free(&pl);
    return 0;
}

/* Model.jet_E_iso / jet_Gamma0 / medium (pybind/pymodel.cpp:572-594): kind 0 -> E_iso(theta) [erg], 1 -> Gamma0(theta),
 * 2 -> rho(r [cm]) [g/cm^3] */
int vag_oracle_profile(const vag_model_params* p, int kind, const double* x, int n, double* out) {
    if (vag_oracle_params_validate(p) != 0) return -1;
    jet_t jet;
    medium_t med;
    jet_init(&jet, p);
    medium_init(&med, p);
    for (int i = 0; i < n; ++i) {
        if (kind == 0)
            out[i] = jet_eps_k(&jet, x[i]) / (U_ERG / (4 * C_PI));
        else if (kind == 1)
            out[i] = jet_Gamma0(&jet, x[i]);
        else if (kind == 2)
            out[i] = medium_rho(&med, x[i] * U_CM) / (U_G / U_CM3);
        else
            return fail("profile kind must be 0, 1 or 2");
    }
    return 0;
}

int vag_oracle_details(const vag_model_params* p, double t_min, double t_max, vag_details_shape* shape,
                       const vag_details_out* out, double** extra, int n_extra, int* n_phi_eff,
                       const double* probe_lg2_nu, int n_probe) {
    return details_impl(p, t_min, t_max, shape, out, extra, n_extra, n_phi_eff, probe_lg2_nu, n_probe, 0);
}

int vag_oracle_details_rvs(const vag_model_params* p, double t_min, double t_max, vag_details_shape* shape,
                           const vag_details_out* out, double** extra, int n_extra, int* n_phi_eff,
                           const double* probe_lg2_nu, int n_probe) {
    return details_impl(p, t_min, t_max, shape, out, extra, n_extra, n_phi_eff, probe_lg2_nu, n_probe, 1);
}

/* Fitter._evaluate / _chi2_sum / eval_one: VegasAfterglow/fitting/fitter.py:497-533,
 * fitting/samplers.py:61-70, transformer fitting/utils.py:110-135 */
int vag_oracle_loglike_batch(const vag_fit_spec* spec, const double* theta, int nb, int ndim, double* out) {
    if (ndim != spec->ndim || ndim > 16) return fail("ndim mismatch");
    const int n = spec->n_data;
    int nmax = n > 0 ? n : 1;
    for (int g = 0; g < spec->n_bands; ++g)
        if (spec->bands[g].n > nmax) nmax = spec->bands[g].n;
    double* F = malloc(sizeof(double) * nmax);
    for (int b = 0; b < nb; ++b) {
        vag_model_params p = spec->base;
        double* fields = &p.theta_c;
        double a_v = spec->a_v_fixed;
        for (int d = 0; d < ndim; ++d) {
            const double v = theta[(size_t)b * ndim + d];
            const double val = spec->is_log[d] ? pow(10.0, v) : v;
            if (spec->slot[d] == VAG_P_A_V)
                a_v = val;
            else
                fields[spec->slot[d]] = val;
        }
        double chi2 = 0;
        int bad = 0;
        if (n > 0) { /* point data, fitter.py:510-522 */
            if (vag_oracle_flux_density(&p, spec->t, spec->nu, n, F) != 0) bad = 1;
            for (int i = 0; i < n && !bad; ++i) {
                double f = F[i];
                if (spec->ext_kernel && a_v != 0.0) f = f * exp(-a_v * spec->ext_kernel[i]);
                const double fm = (f != f) ? f : (f > 1e-300 ? f : 1e-300); /* np.maximum propagates NaN */
                const double diff = spec->ln_flux[i] - log(fm);
                const double q = diff / spec->ln_err[i];
                chi2 += spec->weight[i] * (q * q);
            }
        }
        for (int g = 0; g < spec->n_bands && !bad; ++g) { /* band-integrated groups, fitter.py:524-531 */
            const vag_band_obs* bd = &spec->bands[g];
            if (vag_oracle_flux(&p, bd->t, bd->n, bd->nu_min, bd->nu_max, bd->num_points, F) != 0) {
                bad = 1;
                break;
            }
            double c2 = 0;
            for (int i = 0; i < bd->n; ++i) {
                const double f = F[i];
                const double fm = (f != f) ? f : (f > 1e-300 ? f : 1e-300);
                const double q = (bd->ln_flux[i] - log(fm)) / bd->ln_err[i];
                c2 += bd->weight[i] * (q * q);
            }
            chi2 += c2;
        }
        out[b] = (!bad && isfinite(chi2)) ? -0.5 * chi2 : -INFINITY;
    }
    free(F);
    return 0;
}
