// vag_dyn_fast.h -- forward-shock dynamics for the common case (no lateral spreading, no energy injection, ISM or the
// analytic Wind medium): grid_solve_fwd_shock (src/dynamics/forward-shock.tpp:175-208) with boost's adaptive DOPRI5
// (controlled_runge_kutta.hpp:56-156, dense_output_runge_kutta.hpp:324-361), one lane per representative (model, theta) row.
//
// Why a second kernel next to vag_dynamics_kernel: a row is ~105 sequential steps x 6 right-hand sides, and a lone
// wavefront pays 4.2 (independent) to 6.3-7 (dependent) cycles per FP64 instruction whatever its lane count, 16.5 per
// v_rcp/v_rsq_f64 and ~25 per compare-and-select on doubles (profiles/micro/issue_rate.hip), so the latency of ONE row --
// which is the latency of a single light curve and of a small walker block -- is the instruction count of a step attempt.
//   * The right-hand side is straight-line code: no library fall-backs, no IEEE division sequences, clamps through
//     v_max/v_min_f64 instead of selects, one Newton step on the reciprocal / square-root estimates (2e-15 / 4e-15:
//     the states are integrated to 1e-6), merged reciprocals.  A step attempt is ONE basic block of ~1000 instructions
//     (the general kernel: 4200 with 75 branches).
//   * ONE flat loop of step attempts: a rejected step of one lane does not make its 63 neighbours repeat theirs
//     (the general kernel nests the retry loop inside the step, so a wavefront of 46 rows retried almost every step).
//   * The dense-output saves leave the sequential chain: a workgroup is an INTEGRATOR wavefront and a SAVER wavefront on
//     two SIMDs of one CU.  The integrator pushes every accepted step (t, t + h, x, k1, k3..k7 of each lane) into an LDS
//     ring; the saver walks each lane's time lattice, evaluates boost's dense output at the nodes the step passed and
//     writes them, concurrently with the next attempts.  Only the interpolated state (Gamma, m2, U, r, t_comv) is saved; the
//     derived shock quantities (compression ratio, Gamma_th, B, N_p: save_fwd_shock_state, forward-shock.tpp:151-173) are
//     finished by vag_cells_kernel, one lane per cell.
// The controller's arithmetic (error norm, 0.9 safety, 5x / 0.2x limits, FSAL) is the one of Dopri5<N>::step.
#pragma once
#include "vag_device.h"

namespace vag {

VAG_DEV double vmax(double a, double b) { return __builtin_fmax(a, b); }
VAG_DEV double vmin(double a, double b) { return __builtin_fmin(a, b); }
// sqrt(x) to 4e-15 for strictly positive normal x: v_rsq_f64 (5e-8) and one coupled correction, s0 + (y/2)(x - s0^2)
// (profiles/micro/rsq_accuracy.hip)
VAG_DEV double sqrt_ode(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double s0 = x * y;
    return fma(fma(-s0, s0, x), 0.5 * y, s0);
}
// exp2_fast (vag_device.h) with the Horner coefficients as scalar operands: this kernel has no VGPRs to park them in, and a
// v_mov_b64 per coefficient and call costs as much as the FMA it feeds.  Finite arguments only.
VAG_DEV double fma_sc(double a, double b, double c) {  // a * b + c, c in SGPRs
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
}
VAG_DEV double exp2_ode(double x) {
    const double n = rint(x);
    const double f = x - n;
    double p = fma_sc(f, 2.5678435993488206e-11, 4.4455382718708116e-10);
    p = fma_sc(p, f, 7.054911620801123e-09);
    p = fma_sc(p, f, 1.01780860092397e-07);
    p = fma_sc(p, f, 1.321548679014431e-06);
    p = fma_sc(p, f, 1.5252733804059841e-05);
    p = fma_sc(p, f, 0.0001540353039338161);
    p = fma_sc(p, f, 0.0013333558146428443);
    p = fma_sc(p, f, 0.009618129107628477);
    p = fma_sc(p, f, 0.05550410866482158);
    p = fma_sc(p, f, 0.24022650695910072);
    p = fma_sc(p, f, 0.6931471805599453);
    p = fma(p, f, 1.0);
    return ldexp(p, (int)n);
}
// log2_tab (vag_device.h) for positive normal arguments, no fall-back branch
VAG_DEV double log2_tab_nb(double x, LdsTab tab) {
    const int hi = __double2hiint(x), lo = __double2loint(x);
    const int eb = hi >> 20;
    const double m = __hiloint2double((hi & 0x000fffff) | 0x3ff00000, lo);
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

// ForwardShockEqn::operator() (forward-shock.tpp:10-70) for theta = const, no injection; medium rho = A / (r02 + r^2) + rho_ism
// (Wind, medium.h:91-133) or rho_ism alone.
template <bool WIND>
struct FsRhs {
    double m_jet0, gm_coeff, inv_gc2 /* 2 / gamma_c_coeff */, eps_e, pm2 /* p - 2, or 0 when p <= 2 */, rho_ism, A, r02;
    LdsTab lg;
    VAG_DEV void operator()(const double* s, double* d) const {
        const double G = s[0], m2 = s[1], U = s[2], r = s[3], tc = s[4];
        const double Gm1 = G - 1;
        const double u2 = Gm1 * (G + 1);
        const double u = sqrt_ode(vmax(u2, 1e-300));  // Gamma == 1 exactly would give u = 1e-150 instead of 0
        const double Gu = G + u;
        const double dr = u * Gu;
        d[3] = dr;
        d[4] = Gu;
        const double inv_Gr = rcp_ode(G * r);  // one reciprocal for 1/Gamma and 1/r
        const double inv_G = inv_Gr * r, inv_r = inv_Gr * G;
        double rh = rho_ism;
        if constexpr (WIND) rh = fma(A, rcp_ode(fma(r, r, r02)), rho_ism);
        const double dm = r * r * rh * dr;
        d[1] = dm;
        const double e_th = Gm1 * 4 * G * rh;
        // RadiativeEfficiency (shock-physics.h:247-288): gamma_m / gamma_c with gamma_c = (gamma_bar + sqrt(gamma_bar^2 + 4)) / 2,
        // gamma_bar = gc_coeff / (e_th t_comv) = 1 / q:  ratio = gamma_m 2q / (1 + sqrt(1 + 4 q^2)) -- one reciprocal instead of
        // three, no cancellation at either end
        const double gamma_m = fma(gm_coeff, Gm1, 1.0);
        const double q2 = (e_th * tc) * inv_gc2;  // 2 q
        const double ratio = gamma_m * q2 * rcp_ode(1.0 + sqrt_ode(fma(q2, q2, 1.0)));
        // eps_e (gamma_m / gamma_c)^(p-2) in slow cooling, eps_e otherwise: the clamp to [1e-300, 1] makes the power 1 in fast
        // cooling and ~0 before anything is swept (gamma_c = inf); pm2 is 0 when p <= 2
        const double eps_rad = eps_e * exp2_ode(pm2 * log2_tab_nb(vmin(vmax(ratio, 1e-300), 1.0), lg));
        const double ad = fma(inv_G, 1.0 / 3.0, 4.0 / 3.0);
        const double adm1 = ad - 1;
        const double G2 = G * G;
        const double Geff = fma(ad, G2 - 1, 1.0) * inv_G;
        const double dGeff = fma(ad, G2 + 1, -1.0) * (inv_G * inv_G);
        const double dlnV = 3 * inv_r * dr;
        const double a1 = -Gm1 * (Geff + 1) * dm;
        const double a2 = adm1 * Geff * U * dlnV;
        const double b1 = m_jet0 + m2;
        const double b2 = (dGeff + Geff * adm1 * inv_G) * U;
        const double dG = (a1 + a2) * rcp_ode(b1 + b2);
        d[0] = dG;
        const double dlnV2 = fma(-dG, inv_G, dlnV);
        d[2] = (1 - eps_rad) * Gm1 * dm - adm1 * dlnV2 * U;
    }
};

// ---- integrator -> saver ring in LDS ----
constexpr int DYN_NSLOT = 2;                 // accepted-step records in flight (39 KB of LDS: four workgroups per CU)
constexpr int DYN_REC = 2 + 5 * 7;           // t, t + h, then x, k1, k3, k4, k5, k6, k7 (5 variables each)
constexpr int DYN_PAIRS = (DYN_REC + 1) / 2;
struct DynRing {
    vdouble2 rec[DYN_NSLOT][DYN_PAIRS][64];  // lane-fastest 16-byte pairs: conflict-free ds_write_b128 / ds_read_b128
    int flag[DYN_NSLOT][64];                 // 1: this lane accepted a step in this slot
    int head, tail, fin;                     // slots published / consumed; integrator finished
};
VAG_DEV int lds_load_acquire(const int* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
VAG_DEV void lds_store_release(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
// Publishing store without the wait for the data writes before it: one wavefront's LDS operations are performed in issue order by
// the CU's LDS unit, so a reader that sees the new counter also sees the records written before it.  The compiler barrier keeps the
// issue order; there is no s_waitcnt lgkmcnt(0) in the integrator's chain.
VAG_DEV void lds_store_ordered(int* p, int v) {
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
}

// The flat attempt loop of one wavefront (lane = row).  Returns the lane's solver status (0 ok, 1 step underflow,
// 2 step cap).
// TALLY (vag_ctx_count_work, untimed passes): the right-hand sides of the rows and the loop's lane utilisation.  Everything the tally adds is
// compiled out of the product instantiation: an extra counter and reference parameter cost the latency-bound loop 0.14 ms per
// 1024-walker step when they were tried unconditionally (r05).
// tally[0] += right-hand sides, tally[1] += live lanes summed over the wavefront's attempts, tally[2] += attempts x 64 (the lane slots they
// occupied): tally[1] / tally[2] is the lane utilisation of the attempt loop.
template <class Eq, bool TALLY = false>
VAG_DEV int fs_integrator(const Eq& eq, double* x, double t0, double eps, double t_last, bool active, LdsTab lg, DynRing& ring,
                          int lane, unsigned long long* tally = nullptr) {
    constexpr int N = 5;
    constexpr double a21 = 1.0 / 5, b31 = 3.0 / 40, b32 = 9.0 / 40;
    constexpr double b41 = 44.0 / 45, b42 = -56.0 / 15, b43 = 32.0 / 9;
    constexpr double b51 = 19372.0 / 6561, b52 = -25360.0 / 2187, b53 = 64448.0 / 6561, b54 = -212.0 / 729;
    constexpr double b61 = 9017.0 / 3168, b62 = -355.0 / 33, b63 = 46732.0 / 5247, b64 = 49.0 / 176, b65 = -5103.0 / 18656;
    constexpr double c1 = 35.0 / 384, c3 = 500.0 / 1113, c4 = 125.0 / 192, c5 = -2187.0 / 6784, c6 = 11.0 / 84;
    constexpr double dc1 = c1 - 5179.0 / 57600, dc3 = c3 - 7571.0 / 16695, dc4 = c4 - 393.0 / 640,
                     dc5 = c5 - -92097.0 / 339200, dc6 = c6 - 187.0 / 2100, dc7 = -1.0 / 40;
    double dx[N];
    double t = t0, dt = 0.01 * t0;
    int fails = 0, steps = 0, status = 0, head = 0, tail_seen = 0;
    [[maybe_unused]] int n_rej = 0;
    [[maybe_unused]] unsigned long long live_sum = 0, wave_attempts = 0;
    bool done = !active;
    if (active) {
        eq(x, dx);
        done = !(t <= t_last);
    }
#ifdef VAG_DYN_STAMPS  // developer aid: cycles of the attempt loop, attempts made, cycles spent waiting for the saver
    long long c_begin = __builtin_readcyclecounter(), c_wait = 0, c_body = 0;
    int n_attempts = 0, n_spins = 0;
#endif
    while (__any(!done)) {
#ifdef VAG_DYN_STAMPS
        ++n_attempts;
#if VAG_DYN_STAMPS > 1  // (every s_memtime costs ~100 cycles itself: the detailed stamps distort the total)
        const long long c_top = __builtin_readcyclecounter();
#endif
#endif
        bool accepted = false, commit = false;
        double xn[N], k3[N], k4[N], k5[N], k6[N], k7[N];
        double tn = t;
        if constexpr (TALLY) {
            live_sum += (unsigned long long)__popcll(__ballot(!done));
            wave_attempts += 1;
        }
        if (!done) {
            const double h = dt;
            double xt[N], k2[N];
#pragma unroll
            for (int i = 0; i < N; ++i) xt[i] = x[i] + (h * a21) * dx[i];
            eq(xt, k2);
#pragma unroll
            for (int i = 0; i < N; ++i) xt[i] = x[i] + (h * b31) * dx[i] + (h * b32) * k2[i];
            eq(xt, k3);
#pragma unroll
            for (int i = 0; i < N; ++i) xt[i] = x[i] + (h * b41) * dx[i] + (h * b42) * k2[i] + (h * b43) * k3[i];
            eq(xt, k4);
#pragma unroll
            for (int i = 0; i < N; ++i)
                xt[i] = x[i] + (h * b51) * dx[i] + (h * b52) * k2[i] + (h * b53) * k3[i] + (h * b54) * k4[i];
            eq(xt, k5);
#pragma unroll
            for (int i = 0; i < N; ++i)
                xt[i] = x[i] + (h * b61) * dx[i] + (h * b62) * k2[i] + (h * b63) * k3[i] + (h * b64) * k4[i] + (h * b65) * k5[i];
            eq(xt, k6);
#pragma unroll
            for (int i = 0; i < N; ++i)
                xn[i] = x[i] + (h * c1) * dx[i] + (h * c3) * k3[i] + (h * c4) * k4[i] + (h * c5) * k5[i] + (h * c6) * k6[i];
            eq(xn, k7);
            double err = 0;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const double xe = (h * dc1) * dx[i] + (h * dc3) * k3[i] + (h * dc4) * k4[i] + (h * dc5) * k5[i] +
                                  (h * dc6) * k6[i] + (h * dc7) * k7[i];
                err = vmax(err, fabs(xe) * rcp_ode(fma(eps, fma(fabs(h), fabs(dx[i]), fabs(x[i])), eps)));
            }
            // controlled_runge_kutta.hpp:752-782: reject above 1 (shrink by max(0.9 err^-1/3, 0.2)); grow by
            // 0.9 max(err, 5^-5)^-1/5 when err < 0.5.  One power serves both branches.
            const bool reject = err > 1.0;
            const double lg_e = log2_tab_nb(vmin(vmax(err, 3.2e-4), 1e300), lg);
            const double fac = 0.9 * exp2_ode(lg_e * (reject ? -1.0 / 3 : -1.0 / 5));
            dt = h * (reject ? vmax(fac, 0.2) : (err < 0.5 ? fac : 1.0));
            if (reject) {
                if constexpr (TALLY) ++n_rej;
                if (++fails >= 500) {
                    status = 1;
                    done = true;
                }
            } else {
                fails = 0;
                commit = true;
                tn = t + h;
                if (++steps > 100000) {
                    status = 2;
                    done = true;
                } else {
                    accepted = true;
                    done = !(tn <= t_last);
                }
            }
        }
#if defined(VAG_DYN_STAMPS) && VAG_DYN_STAMPS > 1
        c_body += __builtin_readcyclecounter() - c_top;
#endif
#if defined(VAG_DYN_ABLATE) && (VAG_DYN_ABLATE & 4)
        if (false) {
#else
        if (__any(accepted)) {  // publish the step
#endif  // the saver interpolates it at the lattice nodes it passed
            const int slot = head & (DYN_NSLOT - 1);
            {   // ring full?  (the saver is normally far ahead)
#if defined(VAG_DYN_STAMPS) && VAG_DYN_STAMPS > 1
                const long long w0 = __builtin_readcyclecounter();
#endif
                while (head - tail_seen >= DYN_NSLOT) {  // re-read the saver's position only when the cached one says "full"
                    tail_seen = lds_load_acquire(&ring.tail);
                    if (head - tail_seen >= DYN_NSLOT) {
                        __builtin_amdgcn_s_sleep(1);
#ifdef VAG_DYN_STAMPS
                        ++n_spins;
#endif
                    }
                }
#if defined(VAG_DYN_STAMPS) && VAG_DYN_STAMPS > 1
                c_wait += __builtin_readcyclecounter() - w0;
#endif
            }
            ring.flag[slot][lane] = accepted ? 1 : 0;
#if defined(VAG_DYN_ABLATE) && (VAG_DYN_ABLATE & 1)
            if (false) {
#else
            if (accepted) {
#endif
                double v[2 * DYN_PAIRS];
                v[0] = t, v[1] = tn, v[2 * DYN_PAIRS - 1] = 0;
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    v[2 + i] = x[i];
                    v[7 + i] = dx[i];
                    v[12 + i] = k3[i];
                    v[17 + i] = k4[i];
                    v[22 + i] = k5[i];
                    v[27 + i] = k6[i];
                    v[32 + i] = k7[i];
                }
#pragma unroll
                for (int p = 0; p < DYN_PAIRS; ++p) {
                    vdouble2 w;
                    w.x = v[2 * p], w.y = v[2 * p + 1];
                    ring.rec[slot][p][lane] = w;
                }
            }
            ++head;
            lds_store_ordered(&ring.head, head);
        }
        if (commit) {  // also the step that tripped the step cap, as Dopri5::step would have
#pragma unroll
            for (int i = 0; i < N; ++i) {
                x[i] = xn[i];
                dx[i] = k7[i];
            }
            t = tn;
        }
    }
    lds_store_release(&ring.fin, 1);
    if constexpr (TALLY) {
        if (active) atomicAdd(tally, (unsigned long long)(1 + 6 * (steps + n_rej)));  // FSAL: six new evaluations per attempt, one at the start
        if (lane == 0) {
            atomicAdd(tally + 1, live_sum);
            atomicAdd(tally + 2, 64ull * wave_attempts);
        }
    }
#ifdef VAG_DYN_STAMPS
    if (blockIdx.x == 0 && lane == 0)
        printf("fast dyn wave 0: attempts %d, lane 0 steps %d; cycles total %lld, attempt bodies %lld, waiting for the saver %lld (%d spins)\n",
               n_attempts, steps, (long long)__builtin_readcyclecounter() - c_begin, c_body, c_wait, n_spins);
#endif
    return status;
}

// The saver wavefront: dense output (runge_kutta_dopri5.hpp:238-258) of every published step at the lattice nodes it
// passed; returns the number of nodes written for this lane's row.
template <class Node>
VAG_DEV int fs_saver(DynRing& ring, int lane, bool active, int nt, Node& node, double* __restrict__ o_teng,
                     double* __restrict__ o_tcomv, double* __restrict__ o_r, double* __restrict__ o_G,
                     double* __restrict__ o_U, double* __restrict__ o_m2) {
    constexpr int N = 5;
    constexpr double B1 = 35.0 / 384, B3 = 500.0 / 1113, B4 = 125.0 / 192, B5 = -2187.0 / 6784, B6 = 11.0 / 84;
    int k = 0, its = 0;
    double t_k = active ? node(0) : 0.0;
#ifdef VAG_DYN_STAMPS
    long long s_busy = 0, s_max = 0;
    int s_polls = 0;
#endif
    for (;;) {
        int head = lds_load_acquire(&ring.head);
#ifdef VAG_DYN_STAMPS
        ++s_polls;
        const long long s_t0 = __builtin_readcyclecounter();
#endif
        if (head <= its) {
            if (lds_load_acquire(&ring.fin)) {
                head = lds_load_acquire(&ring.head);  // everything published before fin
                if (head <= its) break;
            } else {
                __builtin_amdgcn_s_sleep(1);
                continue;
            }
        }
        const int slot = its & (DYN_NSLOT - 1);
#if defined(VAG_DYN_ABLATE) && (VAG_DYN_ABLATE & 2)
        if (false) {
#else
        if (active && ring.flag[slot][lane]) {
#endif
            const vdouble2 tt = ring.rec[slot][0][lane];
            const double t = tt.x, tn = tt.y;
            if (k < nt && tn > t_k) {
                double v[2 * DYN_PAIRS];
#pragma unroll
                for (int p = 1; p < DYN_PAIRS; ++p) {
                    const vdouble2 w = ring.rec[slot][p][lane];
                    v[2 * p] = w.x, v[2 * p + 1] = w.y;
                }
                double x[N], dx[N], k3[N], k4[N], k5[N], k6[N], k7[N];
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    x[i] = v[2 + i];
                    dx[i] = v[7 + i];
                    k3[i] = v[12 + i];
                    k4[i] = v[17 + i];
                    k5[i] = v[22 + i];
                    k6[i] = v[27 + i];
                    k7[i] = v[32 + i];
                }
                const double hh = tn - t;
                const double inv_hh = rcp_fast(hh);
                do {
                    const double th = (t_k - t) * inv_hh;
                    const double X1 = 5.0 * (2558722523.0 - 31403016.0 * th) * (1.0 / 11282082432.0);
                    const double X3 = 100.0 * (882725551.0 - 15701508.0 * th) * (1.0 / 32700410799.0);
                    const double X4 = 25.0 * (443332067.0 - 31403016.0 * th) * (1.0 / 1880347072.0);
                    const double X5 = 32805.0 * (23143187.0 - 3489224.0 * th) * (1.0 / 199316789632.0);
                    const double X6 = 55.0 * (29972135.0 - 7076736.0 * th) * (1.0 / 822651844.0);
                    const double X7 = 10.0 * (7414447.0 - 829305.0 * th) * (1.0 / 29380423.0);
                    const double thm1 = th - 1.0, th2 = th * th;
                    const double A_ = th2 * (3.0 - 2.0 * th);
                    const double B_ = th2 * thm1;
                    const double C_ = th2 * thm1 * thm1;
                    const double D_ = th * thm1 * thm1;
                    const double w1 = hh * (A_ * B1 - C_ * X1 + D_), w3 = hh * (A_ * B3 + C_ * X3), w4 = hh * (A_ * B4 - C_ * X4),
                                 w5 = hh * (A_ * B5 + C_ * X5), w6 = hh * (A_ * B6 - C_ * X6), w7 = hh * (B_ + C_ * X7);
                    double q[N];
#pragma unroll
                    for (int i = 0; i < N; ++i)
                        q[i] = x[i] + w1 * dx[i] + w3 * k3[i] + w4 * k4[i] + w5 * k5[i] + w6 * k6[i] + w7 * k7[i];
                    o_teng[k] = t_k;
                    o_G[k] = q[0];
                    o_m2[k] = q[1];
                    o_U[k] = q[2];
                    o_r[k] = q[3];
                    o_tcomv[k] = q[4];
                    ++k;
                    if (k < nt) t_k = node(k);
                } while (k < nt && tn > t_k);
            }
        }
        ++its;
        lds_store_release(&ring.tail, its);
#ifdef VAG_DYN_STAMPS
        {
            const long long d = __builtin_readcyclecounter() - s_t0;
            s_busy += d;
            s_max = d > s_max ? d : s_max;
        }
#endif
    }
#ifdef VAG_DYN_STAMPS
    if (blockIdx.x == 0 && lane == 0)
        printf("  saver: slots %d, polls %d, busy cycles %lld (max per slot %lld), lane 0 nodes %d\n", its, s_polls, s_busy, s_max, k);
#endif
    return k;
}


// ------------------------------------------------------------------------------------------------
// Throughput form of the same solve (round 6): LANE REFILL, saves inline.  The kernel above is built for the latency of ONE row: a
// wavefront of 64 rows runs until its slowest row is done (rows take 60 ... 130 attempts: a quarter of the lane-attempts of a large
// batch are finished lanes waiting, vag_plan.ode_lane_attempts / ode_lane_slots), and the saver wavefront with its 39 KB ring keeps a CU
// at four integrators -- ONE per SIMD, and a wavefront issues at most one instruction per four cycles whatever its kind, so the FP64 pipe
// of a full GPU idles 45 % of the time (profiles/r06_pmc_dyn.txt: 0.55 VALU busy at 1.7 wavefronts per SIMD).  For batches that fill
// the GPU several times over the workgroups are PERSISTENT single wavefronts: a lane whose row is finished takes the next row from a
// device counter and goes on in place (rows are independent: same bits), and the dense output is evaluated inline by the lanes whose
// step passed a lattice node -- no saver, no ring, no LDS beyond the logarithm table, so two integrators share every SIMD and fill each
// other's issue slots.  What a lane needs to start a row is prepared beforehand, one lane per row, by vag_dyn_prep_kernel (the
// start-up of a row is ~3 k instructions of branchy code -- library logarithms of the lattice, the enclosed mass and energy -- which a
// refill inside the attempt loop would make the whole wavefront wait for): a 160-byte record per row and the row's node times, which
// are an output anyway (VS_TENG).  ("persistent-lane work queue"; forward-shock.tpp:194-207 is the loop over rows.)
// ------------------------------------------------------------------------------------------------
constexpr int DYN_ROWREC = 20;  // doubles per row record
// The row queue is ordered: longest rows first.  A row's step count is predictable from its start record -- on 7111 rows of an
// 8192-walker configs[3] step (18 ... 134 steps, standard deviation 27) a linear form in ln Gamma0, ln(Gamma0 - 1), ln rho and ln m_jet
// leaves a residual of 1.1 steps (profiles/r06_row_attempts.txt) -- so vag_dyn_prep_kernel files every row under its predicted length
// (DYN_BUCKETS lists of 8 steps each) and the wavefronts empty the lists from the longest down: a wavefront's 64 rows are of one
// length (few lanes wait for a straggler), and what is still running when the lists are empty are the SHORT rows (LPT scheduling).
// The prediction only orders work: a poor one costs time, never a bit.
// The queue is ONE array in that order, made by a counting sort that keeps neighbours together (vag_dyn_prep_kernel: histogram per chunk of
// 256 rows in LDS; vag_dyn_scan_kernel: offsets, longest class first, chunks in order inside a class; vag_dyn_file_kernel: scatter), so
// the 64 rows a wavefront holds come from a window of a few thousand rows of the batch.  Two simpler forms lost: sixteen lists filled
// by atomics (arrival order: a wavefront's lanes then saved into 64 unrelated places of seven 135 MB arrays, the kernel ran at 11 % of
// the VALU issue slots, and 433 k insertions on sixteen counters cost another 0.8 ms: 6.0 against 1.86 ms per 8192 walkers), and lists
// per chunk walked by the wavefronts (2.95 ms with 4096-row chunks, 18.7 with 256-row ones: finding the next non-empty list is a chain
// of dependent loads).
constexpr int DYN_BUCKETS = 16;
constexpr int DYN_CHUNK = 256;  // rows per chunk = the preparation kernel's workgroup
VAG_DEV int dyn_row_bucket(double Gamma0, double rho, double m_jet) {
    const double pred = 101.1 + 7.43 * log(Gamma0) + 3.04 * log(fmax(Gamma0 - 1, 1e-12)) + 1.81 * (log(rho) - log(m_jet));
    const int b = (int)(pred * 0.125);
    return pred == pred ? (b < 0 ? 0 : (b > DYN_BUCKETS - 1 ? DYN_BUCKETS - 1 : b)) : DYN_BUCKETS - 1;
}
enum {
    DR_X = 0,       // Gamma, m2, U, r, t_comv at t0 (set_init_state)
    DR_T0 = 5,
    DR_TLAST = 6,
    DR_RTOL = 7,
    DR_EQ = 8,      // m_jet0, gm_coeff, inv_gc2, eps_e, pm2, rho_ism, A, r02
    DR_NT = 16,     // (n_t of the row, state: 0 integrate, 2 nothing to do -- not evaluated, or a stopped shock the preparation wrote)
    DR_C0 = 17,     // first cell of the row in the shock arrays (the bits of a long long)
};

template <bool TALLY>
VAG_DEV void fs_solver_refill(const double* __restrict__ rowrec, int n_rows,
                              unsigned* __restrict__ queue /* [0] rows in the sorted queue, [1] rows taken */,
                              const int* __restrict__ order /* the queue: rows, longest predicted first */, int refill_min, LdsTab lg, int lane,
                              double* __restrict__ shock, long long n_cells, int* __restrict__ row_status, int* __restrict__ fail,
                              unsigned long long* __restrict__ tally) {
    constexpr int N = 5;
    constexpr double a21 = 1.0 / 5, b31 = 3.0 / 40, b32 = 9.0 / 40;
    constexpr double b41 = 44.0 / 45, b42 = -56.0 / 15, b43 = 32.0 / 9;
    constexpr double b51 = 19372.0 / 6561, b52 = -25360.0 / 2187, b53 = 64448.0 / 6561, b54 = -212.0 / 729;
    constexpr double b61 = 9017.0 / 3168, b62 = -355.0 / 33, b63 = 46732.0 / 5247, b64 = 49.0 / 176, b65 = -5103.0 / 18656;
    constexpr double c1 = 35.0 / 384, c3 = 500.0 / 1113, c4 = 125.0 / 192, c5 = -2187.0 / 6784, c6 = 11.0 / 84;
    constexpr double dc1 = c1 - 5179.0 / 57600, dc3 = c3 - 7571.0 / 16695, dc4 = c4 - 393.0 / 640,
                     dc5 = c5 - -92097.0 / 339200, dc6 = c6 - 187.0 / 2100, dc7 = -1.0 / 40;
    constexpr double B1 = 35.0 / 384, B3 = 500.0 / 1113, B4 = 125.0 / 192, B5 = -2187.0 / 6784, B6 = 11.0 / 84;
    // the Wind form serves every row: with A = 0 its density term is fma(0, finite, rho_ism) = rho_ism, the uniform-medium bits
    FsRhs<true> eq;
    eq.lg = lg;
    eq.m_jet0 = eq.gm_coeff = eq.inv_gc2 = eq.eps_e = eq.pm2 = eq.rho_ism = 1;
    eq.A = eq.r02 = 0;
    double x[N] = {2.0, 1.0, 1.0, 1.0, 1.0}, dx[N] = {0, 0, 0, 0, 0};
    double t = 1, dt = 0.01, t_last = 0, eps = 1e-6;
    int fails = 0, steps = 0, status = 0, row = -1;
    // the row's lattice: next node k of nt, its time and the one after it (requested a node ahead: a load is a microsecond)
    int k = 0, nt = 0;
    double t_k = 0, t_k1 = 0;
    double* o = shock;  // &shock[VS_TENG][c0]
    [[maybe_unused]] int n_rej = 0;
    [[maybe_unused]] unsigned long long live_sum = 0, wave_attempts = 0, rhs_sum = 0;
    bool done = true, drained = false;
    auto retire = [&]() {  // unreached nodes keep the Shock constructor's defaults (shock.cpp:12-24); their times are written already
        for (; k < nt; ++k) {
            o[VS_TCOMV * n_cells + k] = 0;
            o[VS_R * n_cells + k] = 0;
            o[VS_GAMMA * n_cells + k] = 1;
            o[VS_GAMMA_TH * n_cells + k] = 0;
            o[VS_B * n_cells + k] = 0;
            o[VS_NP * n_cells + k] = 0;
        }
        row_status[row] = status;
        if (status > 0 && status < 4) atomicAdd(fail + status, 1);
        if constexpr (TALLY) rhs_sum += (unsigned long long)(1 + 6 * (steps + n_rej));
#ifdef VAG_DYN_ROWSTATS  // developer aid: what predicts a row's attempt count? (profiles/debug/row_attempts.py)
        if (row % 61 == 0) {
            const double* r = rowrec + (size_t)row * DYN_ROWREC;
            printf("R %d steps %d G0 %.6e t0 %.6e tlast %.6e mjet %.6e rho %.6e A %.6e nt %d\n", row, steps, r[DR_X], r[DR_T0], r[DR_TLAST], r[DR_EQ], r[DR_EQ + 5],
                   r[DR_EQ + 6], nt);
        }
#endif
        row = -1;
    };
    for (;;) {
        const unsigned long long mdone = __ballot(done);
        const int n_done = __popcll(mdone);
        if (!drained && n_done >= refill_min) {
            if (done && row >= 0) retire();
            const int first = __ffsll((long long)mdone) - 1;
            unsigned base = 0, filed = 0;
            if (lane == first) {
                base = atomicAdd(queue + 1, (unsigned)n_done);
                filed = queue[0];
            }
            base = (unsigned)__shfl((int)base, first);
            filed = (unsigned)__shfl((int)filed, first);
            if (base + (unsigned)n_done >= filed) drained = true;
            const unsigned slot = base + (unsigned)__popcll(mdone & ((1ull << lane) - 1ull));
            const long long mine = done && slot < filed ? order[slot] : -1;
            if (mine >= 0 && mine < n_rows) {
                const double* r = rowrec + (size_t)mine * DYN_ROWREC;
                const int state = __double2hiint(r[DR_NT]);
                if (state == 0) {
                    row = (int)mine;
#pragma unroll
                    for (int i = 0; i < N; ++i) x[i] = r[DR_X + i];
                    t = r[DR_T0], t_last = r[DR_TLAST], eps = r[DR_RTOL];
                    eq.m_jet0 = r[DR_EQ + 0], eq.gm_coeff = r[DR_EQ + 1], eq.inv_gc2 = r[DR_EQ + 2], eq.eps_e = r[DR_EQ + 3];
                    eq.pm2 = r[DR_EQ + 4], eq.rho_ism = r[DR_EQ + 5], eq.A = r[DR_EQ + 6], eq.r02 = r[DR_EQ + 7];
                    nt = __double2loint(r[DR_NT]);
                    o = shock + __double_as_longlong(r[DR_C0]);
                    k = 0;
                    t_k = o[0];
                    t_k1 = nt > 1 ? o[1] : 0.0;
                    dt = 0.01 * t;
                    fails = steps = status = 0;
                    if constexpr (TALLY) n_rej = 0;
                    eq(x, dx);
                    done = !(t <= t_last);
                }
            }
            continue;
        }
        if (n_done == 64) break;  // nothing live and nothing left to take
        if constexpr (TALLY) {
            live_sum += (unsigned long long)(64 - n_done);
            wave_attempts += 1;
        }
        // ---- one step attempt of every live lane: the arithmetic of fs_integrator, expression for expression ----
        bool accepted = false, commit = false;
        double xn[N], k3[N], k4[N], k5[N], k6[N], k7[N];
        double tn = t;
        if (!done) {
            const double h = dt;
            double xt[N], k2[N];
#pragma unroll
            for (int i = 0; i < N; ++i) xt[i] = x[i] + (h * a21) * dx[i];
            eq(xt, k2);
#pragma unroll
            for (int i = 0; i < N; ++i) xt[i] = x[i] + (h * b31) * dx[i] + (h * b32) * k2[i];
            eq(xt, k3);
#pragma unroll
            for (int i = 0; i < N; ++i) xt[i] = x[i] + (h * b41) * dx[i] + (h * b42) * k2[i] + (h * b43) * k3[i];
            eq(xt, k4);
#pragma unroll
            for (int i = 0; i < N; ++i)
                xt[i] = x[i] + (h * b51) * dx[i] + (h * b52) * k2[i] + (h * b53) * k3[i] + (h * b54) * k4[i];
            eq(xt, k5);
#pragma unroll
            for (int i = 0; i < N; ++i)
                xt[i] = x[i] + (h * b61) * dx[i] + (h * b62) * k2[i] + (h * b63) * k3[i] + (h * b64) * k4[i] + (h * b65) * k5[i];
            eq(xt, k6);
#pragma unroll
            for (int i = 0; i < N; ++i)
                xn[i] = x[i] + (h * c1) * dx[i] + (h * c3) * k3[i] + (h * c4) * k4[i] + (h * c5) * k5[i] + (h * c6) * k6[i];
            eq(xn, k7);
            double err = 0;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const double xe = (h * dc1) * dx[i] + (h * dc3) * k3[i] + (h * dc4) * k4[i] + (h * dc5) * k5[i] +
                                  (h * dc6) * k6[i] + (h * dc7) * k7[i];
                err = vmax(err, fabs(xe) * rcp_ode(fma(eps, fma(fabs(h), fabs(dx[i]), fabs(x[i])), eps)));
            }
            const bool reject = err > 1.0;
            const double lg_e = log2_tab_nb(vmin(vmax(err, 3.2e-4), 1e300), lg);
            const double fac = 0.9 * exp2_ode(lg_e * (reject ? -1.0 / 3 : -1.0 / 5));
            dt = h * (reject ? vmax(fac, 0.2) : (err < 0.5 ? fac : 1.0));
            if (reject) {
                if constexpr (TALLY) ++n_rej;
                if (++fails >= 500) {
                    status = 1;
                    done = true;
                }
            } else {
                fails = 0;
                commit = true;
                tn = t + h;
                if (++steps > 100000) {
                    status = 2;
                    done = true;
                } else {
                    accepted = true;
                    done = !(tn <= t_last);
                }
            }
        }
        // ---- dense output (runge_kutta_dopri5.hpp:238-258) at the lattice nodes the accepted step passed: fs_saver's expressions ----
        const bool saves = accepted && k < nt && tn > t_k;
        if (__any(saves)) {
            if (saves) {
                const double hh = tn - t;
                const double inv_hh = rcp_fast(hh);
                do {
                    const double th = (t_k - t) * inv_hh;
                    const double X1 = 5.0 * (2558722523.0 - 31403016.0 * th) * (1.0 / 11282082432.0);
                    const double X3 = 100.0 * (882725551.0 - 15701508.0 * th) * (1.0 / 32700410799.0);
                    const double X4 = 25.0 * (443332067.0 - 31403016.0 * th) * (1.0 / 1880347072.0);
                    const double X5 = 32805.0 * (23143187.0 - 3489224.0 * th) * (1.0 / 199316789632.0);
                    const double X6 = 55.0 * (29972135.0 - 7076736.0 * th) * (1.0 / 822651844.0);
                    const double X7 = 10.0 * (7414447.0 - 829305.0 * th) * (1.0 / 29380423.0);
                    const double thm1 = th - 1.0, th2 = th * th;
                    const double A_ = th2 * (3.0 - 2.0 * th);
                    const double B_ = th2 * thm1;
                    const double C_ = th2 * thm1 * thm1;
                    const double D_ = th * thm1 * thm1;
                    const double w1 = hh * (A_ * B1 - C_ * X1 + D_), w3 = hh * (A_ * B3 + C_ * X3), w4 = hh * (A_ * B4 - C_ * X4),
                                 w5 = hh * (A_ * B5 + C_ * X5), w6 = hh * (A_ * B6 - C_ * X6), w7 = hh * (B_ + C_ * X7);
                    double q[N];
#pragma unroll
                    for (int i = 0; i < N; ++i)
                        q[i] = x[i] + w1 * dx[i] + w3 * k3[i] + w4 * k4[i] + w5 * k5[i] + w6 * k6[i] + w7 * k7[i];
                    o[VS_GAMMA * n_cells + k] = q[0];
                    o[VS_NP * n_cells + k] = q[1];
                    o[VS_GAMMA_TH * n_cells + k] = q[2];
                    o[VS_R * n_cells + k] = q[3];
                    o[VS_TCOMV * n_cells + k] = q[4];
                    ++k;
                    t_k = t_k1;
                    if (k + 1 < nt) t_k1 = o[k + 1];
                } while (k < nt && tn > t_k);
            }
        }
        if (commit) {
#pragma unroll
            for (int i = 0; i < N; ++i) {
                x[i] = xn[i];
                dx[i] = k7[i];
            }
            t = tn;
        }
    }
    if (row >= 0) retire();  // rows that finished after the queue ran dry
    if constexpr (TALLY) {
        if (rhs_sum) atomicAdd(tally, rhs_sum);
        if (lane == 0) {
            atomicAdd(tally + 1, live_sum);
            atomicAdd(tally + 2, 64ull * wave_attempts);
        }
    }
}

}  // namespace vag
