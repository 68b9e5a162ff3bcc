/*
 * vag_oracle.h -- TEST INFRASTRUCTURE (the parity checker), not product code.
 *
 * Scalar C11 restatement of VegasAfterglow's forward-shock synchrotron light-curve path
 * (SURVEY.md section 8a rows 1-14).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  Pinned against the reference's own golden vectors
 * (tests/golden/*.npz) and against oracle/_ref/libvag_ref.so (the real C++ sources
 * compiled in place) by tests/test_oracle_*.py.
 */
#ifndef VAG_ORACLE_H
#define VAG_ORACLE_H

#include "../include/vegasafterglow_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

const char* vag_oracle_last_error(void);

/* Model.flux_density_grid: out[nnu][nt]  (pybind/pymodel.cpp:498-514) */
int vag_oracle_flux_density_grid(const vag_model_params* p, const double* t, int nt, const double* nu, int nnu,
                                 double* out);
/* fwd.sync and fwd.ssc components of the grid (out_ssc may be NULL), each [nnu][nt] */
int vag_oracle_flux_density_grid_components(const vag_model_params* p, const double* t, int nt, const double* nu, int nnu,
                                            double* out_sync, double* out_ssc);
/* all four components {fwd.sync, fwd.ssc, rvs.sync, rvs.ssc}, each [nnu][nt]; NULL entries are skipped */
int vag_oracle_flux_density_grid_components4(const vag_model_params* p, const double* t, int nt, const double* nu, int nnu,
                                             double* const* out4);
/* Model.flux_density: out[n]  (pybind/pymodel.cpp:373-389) */
int vag_oracle_flux_density(const vag_model_params* p, const double* t, const double* nu, int n, double* out);
/* Model.flux: out[nt]  (pybind/pymodel.cpp:391-410) */
int vag_oracle_flux(const vag_model_params* p, const double* t, int nt, double nu_min, double nu_max, int num_nu,
                    double* out);
/* Model.flux_density_exposures: out[n]  (pybind/pymodel.cpp:412-496) */
int vag_oracle_flux_density_exposures(const vag_model_params* p, const double* t, const double* nu,
                                      const double* expo_time, int n, int num_points, double* out);
/* Model.details-like intermediates; same protocol as oracle/ref_driver.cpp:vag_ref_details. */
int vag_oracle_details(const vag_model_params* p, double t_min, double t_max, vag_details_shape* shape,
                       const vag_details_out* out, double** extra, int n_extra, int* n_phi_eff,
                       const double* probe_lg2_nu, int n_probe);
/* same for the reverse shock of a Model(rvs_rad=...); extra[15] = injection_idx per cell */
int vag_oracle_details_rvs(const vag_model_params* p, double t_min, double t_max, vag_details_shape* shape,
                           const vag_details_out* out, double** extra, int n_extra, int* n_phi_eff,
                           const double* probe_lg2_nu, int n_probe);
/* Model.jet_E_iso / jet_Gamma0 / medium (pybind/pymodel.cpp:572-594): kind 0 E_iso(theta) [erg], 1 Gamma0(theta), 2 rho(r [cm]) [g/cm^3] */
int vag_oracle_profile(const vag_model_params* p, int kind, const double* x, int n, double* out);
/* Fitter log-likelihood for nb walkers (fitter.py:497-533, samplers.py:61-70): out[nb]. */
int vag_oracle_loglike_batch(const vag_fit_spec* spec, const double* theta, int nb, int ndim, double* out);
/* Same validation rules as the product's vag_params_validate (pybind/pymodel.cpp:47-186). */
int vag_oracle_params_validate(const vag_model_params* p);

#ifdef __cplusplus
}
#endif
#endif
