// vag_common.h -- layouts shared by the host C-ABI code and the gfx950 kernels.
#pragma once
#include <stdint.h>

#include "../../include/vegasafterglow_amd.h"

// Static per-model capacities of the adaptive grid (src/core/grid-refinement.h:639-706 decides
// the actual sizes at run time; these bound them).
#define VAG_MAX_THETA 1280  // theta nodes per model in the LARGE LDS layout of the grid kernel (VagGridMeta::th_stride is the HBM stride of a batch)
#define VAG_MAX_PHI 2560    // phi nodes per model in that layout
#define VAG_HUGE_THETA 16384  // the third layout (round 6): the grid kernel's scratch arrays in HBM instead of LDS, for models whose angular grids
#define VAG_HUGE_PHI 32768    //   outgrow the large one (the reference sizes its grids freely, grid-refinement.h:639-706); node indices stay below 2^15
#define VAG_GRID_THETA 256  // what the grid kernel's small (default) LDS layout holds; a batch that needs more is laid out again
#define VAG_GRID_PHI 208    //   with the large layout (vag_grid_kernel<true>).  256 / 208 is 20 420 B of LDS: eight models per CU, as many
                            //   as the kernel's 206 VGPRs allow (320 / 640 was 27 KB: five; the configs' ensembles reach 177 / 191)
#define VAG_ROWGEO_HDR 4    // doubles ahead of a model's row-geometry records (vag_grid_kernel.h writes them, the flux grid kernel reads them)
#define VAG_MAX_TIME (1 << 20)  // time-lattice nodes per row: no kernel holds a row at once (the flux kernels stage at most 512 nodes at a time and take
                                // longer lattices in pieces), so this is a sanity bound on memory, not a layout limit (8192 until round 5)
#define VAG_MAX_NU 64       // frequencies per grid call
#define VAG_MAX_JUMPS 16

// symmetry levels, src/core/mesh.h:55-60
#define VAG_SYM_STRUCTURED 0
#define VAG_SYM_PHI_SYMMETRIC 1
#define VAG_SYM_PIECEWISE 2
#define VAG_SYM_ISOTROPIC 3

// Per-model result of the grid kernel (Coord of src/core/mesh.h:66-95 minus the big arrays).
struct VagGridMeta {
    int32_t status;       // 0 ok, VAG_E_* otherwise
    int32_t n_phi;        // |phi|
    int32_t n_theta;      // |theta|
    int32_t n_t;          // time nodes per row (incl. early point)
    int32_t n_reps;       // representative theta rows actually solved
    int32_t symmetry;     // VAG_SYM_*
    int32_t phi_mirrored; // phi in [0, pi], weights doubled
    int32_t n_phi_eff;    // Observer::eff_phi_grid (observer.cpp:218-222)
    int32_t t_num_tot;    // lattice nodes without the early point
    int32_t has_early;    // extra early node at index 0 (grid-refinement.h:583-591)
    int32_t flags;        // VAG_FLAG_* of the model (the host checks they are uniform over a batch)
    int32_t t_num_base;   // lattice nodes a forward-only run would use (post-crossing density of a reverse-shock run)
    int32_t dyn_class;    // 0: vag_dynamics_fast_kernel applies (ISM / analytic Wind, no spreading, no injection, no reverse shock)
    int32_t rep_phi_stride;  // 0, or n_theta for Model(axisymmetric=False) with a spreading jet: the ODE rows are (phi, theta) PAIRS,
                             // row of (theta j, phi i) = rep_of[j] + i * rep_phi_stride (one lattice and one solve per pair, grid-refinement.h:619-625)
    int32_t th_stride, ph_stride;  // stride of the per-model angular arrays in HBM (theta / rep_of / tdec / geo_th; phi / geo_ph): what the
                                   // layout the batch was laid out with holds (VAG_GRID_* by default, VAG_MAX_* for the large one)
    double t_early;  // engine frame, code units
    double t_start;  // min_t_start
    double t_end;    // 1.01 t_max / (1+z)
    // observer constants every flux wavefront needs (computed once here instead of once per wavefront)
    double cos_obs, sin_obs;  // of theta_obs
    double lg2_1pz;           // log2(1 + z)
};

// Batch plan computed on the device from the grid results (vag_plan_kernel): the compact layout's totals, what decides
// the kernels of the later stages, and whether the buffers the host sized in advance were large enough.
struct VagDevPlan {
    int32_t rows, max_k, max_pairs, n_ok, n_invalid, n_capacity;
    int32_t flags_first;  // VAG_FLAG_* of the first valid model (-1: none)
    int32_t flags_mixed;  // some valid model carries other flags
    int32_t dyn_class;    // OR of VagGridMeta::dyn_class
    int32_t overflow;     // a total exceeds the capacity the host launched with: every model was invalidated, nothing was written
    int64_t cells, pairs, eat;
    int32_t seq;          // host copy only: the call this summary belongs to (written last, after a system-scope fence)
    int32_t pad_;
};

// Per-cell parameter block: [row][VAG_NPAR][n_t] in HBM, [k][VAG_NPAR] (144 B per cell) once staged in LDS.
// 0..12: cached SmoothPowerLawSyn members read by compute_log2_I_nu (src/radiation/smooth-power-law-syn.h:20-47);
// 13..17: what the EAT step needs (src/core/observer.cpp:143-205).  Members used together sit in 16-byte aligned
// pairs, so the staged block is read with seven conflict-free ds_read_b128.
enum {
    VP_LG2_LO = 0,  // log2_nu_lo_
    VP_LG2_HI,      // log2_nu_hi_
    VP_DLO,         // diff_lo_
    VP_INV_SLO,     // 1 / smooth_lo_  (== log2_norm_, smooth-power-law-syn.cpp:151)
    VP_DHI,         // diff_hi_
    VP_INV_SHI,     // 1 / smooth_hi_
    VP_LG2_NUM,     // log2_nu_m
    VP_TNORM,       // log2_thick_norm_
    VP_SAB,         // s_a_blend_
    VP_INV_SAB,     // 1 / s_a_blend_
    VP_LG2_I,       // log2_I_nu_max
    VP_LG2_NUMAX,   // log2_nu_M
    VP_INV_NUMAX,   // log2(e) / nu_M
    VP_LG2_R2,      // 2 log2 r
    VP_GAMMA,       // bulk Lorentz factor
    VP_U,           // sqrt((Gamma-1)(Gamma+1))
    VP_R,           // radius
    VP_TENG,        // engine-frame lattice time
    VAG_NPAR
};
static_assert(VAG_NPAR % 2 == 0 && VP_GAMMA % 2 == 0 && VP_R % 2 == 0, "16-byte pairs");

// Shock-state arrays written by the dynamics kernel (Shock, src/dynamics/shock.h:26-88), SoA over cells.
enum { VS_TENG = 0, VS_TCOMV, VS_R, VS_GAMMA, VS_GAMMA_TH, VS_B, VS_NP, VS_THETA /* spreading jets only */, VAG_NSHOCK };
