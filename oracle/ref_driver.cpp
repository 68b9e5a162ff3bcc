// TEST INFRASTRUCTURE ONLY -- never linked into the product.
//
// C-ABI shim over the *real* VegasAfterglow C++ sources, compiled in place from
// /root/reference by oracle/Makefile into oracle/_ref/libvag_ref.so (git-ignored).
// No reference source is copied here: this file only calls the reference's public
// C++ API in the order pybind/pymodel.h:922-961 (PyModel::compute_emission) and
// pybind/pymodel.h:873-920 (single_shock_emission) do for a forward-shock,
// synchrotron-only model, and restates the unit conversions of the pybind factories
// (pybind/pymodel.cpp:47-224,368-410,498-514).
//
// It pins the C restatement in oracle/vag_oracle.c (tests/test_oracle_vs_ref.py) and is the
// "reference" CPU baseline of bench.py when present.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <variant>

#include "../include/vegasafterglow_amd.h"
#include "afterglow.h"

// pybind/shock_dispatch.h:17-28 restated (that header lives beside pybind-only code)
std::pair<Coord, Shock> solve_fwd_shock_like(JetVariant const& jet, MediumVariant const& medium, Array const& t_obs,
                                             Real theta_w, const vag_model_params& p, RadParams const& rad) {
    return std::visit(
        [&](auto const& j, auto const& med) {
            auto coord = auto_grid(j, med, t_obs, theta_w, p.theta_obs, p.z, false, p.phi_resol, p.theta_resol,
                                   p.t_resol, !(p.flags & VAG_FLAG_NON_AXISYMMETRIC));
            auto shock = generate_fwd_shock(coord, med, j, rad, p.rtol);
            return std::pair{std::move(coord), std::move(shock)};
        },
        jet, medium);
}

// pybind/shock_dispatch.h:29-43 restated
std::tuple<Coord, Shock, Shock> solve_shock_pair_like(JetVariant const& jet, MediumVariant const& medium, Array const& t_obs,
                                                      Real theta_w, const vag_model_params& p, RadParams const& fwd_rad,
                                                      RadParams const& rvs_rad) {
    return std::visit(
        [&](auto const& j, auto const& med) {
            auto coord = auto_grid(j, med, t_obs, theta_w, p.theta_obs, p.z, true, p.phi_resol, p.theta_resol, p.t_resol,
                                   !(p.flags & VAG_FLAG_NON_AXISYMMETRIC));
            auto [fwd, rvs] = generate_shock_pair(coord, med, j, fwd_rad, rvs_rad, p.rtol);
            return std::tuple{std::move(coord), std::move(fwd), std::move(rvs)};
        },
        jet, medium);
}

namespace {

thread_local std::string g_err;

JetVariant make_jet(const vag_model_params& p) {
    // pybind/pymodel.cpp:47-146
    const bool spreading = (p.flags & VAG_FLAG_SPREADING) != 0;
    const bool magnetar = (p.flags & VAG_FLAG_MAGNETAR) != 0;
    // convert_unit_jet for an Ejecta (pymodel.cpp:188-210)
    auto convert = [](Ejecta jet) {
        const auto eps_k_cgs = jet.eps_k;
        jet.eps_k = [=](Real phi, Real theta) { return eps_k_cgs(phi, theta) * (unit::erg / (4 * con::pi)); };
        const auto deps_dt_cgs = jet.deps_dt;
        jet.deps_dt = [=](Real phi, Real theta, Real t) {
            return deps_dt_cgs(phi, theta, t / unit::sec) * (unit::erg / (4 * con::pi * unit::sec));
        };
        const auto dm_dt_cgs = jet.dm_dt;
        jet.dm_dt = [=](Real phi, Real theta, Real t) {
            return dm_dt_cgs(phi, theta, t / unit::sec) * (unit::g / (4 * con::pi * unit::sec));
        };
        jet.T0 *= unit::sec;
        return jet;
    };
    if (magnetar) {  // the named factories switch to the generic Ejecta when a magnetar is attached (pymodel.cpp:47-128)
        Ejecta jet;
        switch (p.jet_type) {
            case VAG_JET_TOPHAT:
                jet.eps_k = math::tophat(p.theta_c, p.E_iso);
                jet.Gamma0 = math::tophat_plus_one(p.theta_c, p.Gamma0 - 1);
                break;
            case VAG_JET_GAUSSIAN:
                jet.eps_k = math::gaussian(p.theta_c, p.E_iso);
                jet.Gamma0 = math::gaussian_plus_one(p.theta_c, p.Gamma0 - 1);
                break;
            case VAG_JET_POWERLAW:
                jet.eps_k = math::powerlaw(p.theta_c, p.E_iso, p.k_e);
                jet.Gamma0 = math::powerlaw_plus_one(p.theta_c, p.Gamma0 - 1, p.k_g);
                break;
            case VAG_JET_TWO_COMPONENT:
                jet.eps_k = math::two_component(p.theta_c, p.theta_w, p.E_iso, p.E_iso_w);
                jet.Gamma0 = math::two_component_plus_one(p.theta_c, p.theta_w, p.Gamma0 - 1, p.Gamma0_w - 1);
                break;
            case VAG_JET_STEP_POWERLAW:
                jet.eps_k = math::step_powerlaw(p.theta_c, p.E_iso, p.E_iso_w, p.k_e);
                jet.Gamma0 = math::step_powerlaw_plus_one(p.theta_c, p.Gamma0 - 1, p.Gamma0_w - 1, p.k_g);
                break;
            default:
                throw std::invalid_argument("this jet type takes no magnetar");
        }
        jet.spreading = spreading;
        jet.T0 = p.duration;
        jet.deps_dt = math::magnetar_injection(p.mag_t0, p.mag_q, p.mag_L0, p.theta_c);
        return convert(jet);
    }
    switch (p.jet_type) {
        case VAG_JET_TOPHAT:
            return TophatJet(p.theta_c, p.E_iso * unit::erg, p.Gamma0, spreading, p.duration * unit::sec);
        case VAG_JET_GAUSSIAN:
            return GaussianJet(p.theta_c, p.E_iso * unit::erg, p.Gamma0, spreading, p.duration * unit::sec);
        case VAG_JET_POWERLAW:
            return PowerLawJet(p.theta_c, p.E_iso * unit::erg, p.Gamma0, p.k_e, p.k_g, spreading, p.duration * unit::sec);
        case VAG_JET_TWO_COMPONENT: {
            // PyTwoComponentJet (pymodel.cpp:130-146) + convert_unit_jet (pymodel.cpp:188-210)
            Ejecta jet;
            jet.eps_k = math::two_component(p.theta_c, p.theta_w, p.E_iso, p.E_iso_w);
            jet.Gamma0 = math::two_component_plus_one(p.theta_c, p.theta_w, p.Gamma0 - 1, p.Gamma0_w - 1);
            jet.spreading = spreading;
            jet.T0 = p.duration;
            const auto eps_k_cgs = jet.eps_k;
            jet.eps_k = [=](Real phi, Real theta) { return eps_k_cgs(phi, theta) * (unit::erg / (4 * con::pi)); };
            const auto deps_dt_cgs = jet.deps_dt;
            jet.deps_dt = [=](Real phi, Real theta, Real t) {
                return deps_dt_cgs(phi, theta, t / unit::sec) * (unit::erg / (4 * con::pi * unit::sec));
            };
            const auto dm_dt_cgs = jet.dm_dt;
            jet.dm_dt = [=](Real phi, Real theta, Real t) {
                return dm_dt_cgs(phi, theta, t / unit::sec) * (unit::g / (4 * con::pi * unit::sec));
            };
            jet.T0 *= unit::sec;
            return jet;
        }
        case VAG_JET_STEP_POWERLAW:
        case VAG_JET_POWERLAW_WING: {
            // PyStepPowerLawJet / PyPowerLawWing (pymodel.cpp:90-125) + convert_unit_jet (pymodel.cpp:188-210)
            Ejecta jet;
            if (p.jet_type == VAG_JET_STEP_POWERLAW) {
                jet.eps_k = math::step_powerlaw(p.theta_c, p.E_iso, p.E_iso_w, p.k_e);
                jet.Gamma0 = math::step_powerlaw_plus_one(p.theta_c, p.Gamma0 - 1, p.Gamma0_w - 1, p.k_g);
            } else {
                jet.eps_k = math::powerlaw_wing(p.theta_c, p.E_iso_w, p.k_e);
                jet.Gamma0 = math::powerlaw_wing_plus_one(p.theta_c, p.Gamma0_w - 1, p.k_g);
            }
            jet.spreading = spreading;
            jet.T0 = p.duration;
            const auto eps_k_cgs = jet.eps_k;
            jet.eps_k = [=](Real phi, Real theta) { return eps_k_cgs(phi, theta) * (unit::erg / (4 * con::pi)); };
            const auto deps_dt_cgs = jet.deps_dt;
            jet.deps_dt = [=](Real phi, Real theta, Real t) {
                return deps_dt_cgs(phi, theta, t / unit::sec) * (unit::erg / (4 * con::pi * unit::sec));
            };
            const auto dm_dt_cgs = jet.dm_dt;
            jet.dm_dt = [=](Real phi, Real theta, Real t) {
                return dm_dt_cgs(phi, theta, t / unit::sec) * (unit::g / (4 * con::pi * unit::sec));
            };
            jet.T0 *= unit::sec;
            return jet;
        }
        case VAG_JET_MAGNETIZED_TOPHAT: {
            // tests/python/golden/regenerate.py:141-149 (_magnetized_tophat) through the Ejecta factory of
            // pybind/pybind.cpp:224-272 and convert_unit_jet (pymodel.cpp:188-210)
            const Real theta_c = p.theta_c, E_iso = p.E_iso, Gamma0 = p.Gamma0, sigma0 = p.sigma0;
            Ejecta jet(BinaryFunc([=](Real, Real theta) { return theta <= theta_c ? E_iso : 0.0; }),
                       BinaryFunc([=](Real, Real theta) { return theta <= theta_c ? Gamma0 : 1.0; }),
                       BinaryFunc([=](Real, Real) { return sigma0; }), TernaryFunc(func::zero_3d), TernaryFunc(func::zero_3d),
                       spreading, p.duration);
            const auto eps_k_cgs = jet.eps_k;
            jet.eps_k = [=](Real phi, Real theta) { return eps_k_cgs(phi, theta) * (unit::erg / (4 * con::pi)); };
            const auto deps_dt_cgs = jet.deps_dt;
            jet.deps_dt = [=](Real phi, Real theta, Real t) {
                return deps_dt_cgs(phi, theta, t / unit::sec) * (unit::erg / (4 * con::pi * unit::sec));
            };
            const auto dm_dt_cgs = jet.dm_dt;
            jet.dm_dt = [=](Real phi, Real theta, Real t) {
                return dm_dt_cgs(phi, theta, t / unit::sec) * (unit::g / (4 * con::pi * unit::sec));
            };
            jet.T0 *= unit::sec;
            return jet;
        }
    }
    throw std::invalid_argument("unknown jet_type");
}

MediumVariant make_medium(const vag_model_params& p) {
    // pybind/pymodel.cpp:148-186 (k_m == 2 branch)
    if (p.medium_type == VAG_MEDIUM_ISM) {
        return ISM(p.n_ism / unit::cm3);
    }
    if (p.medium_type == VAG_MEDIUM_WIND) {
        if (p.k_m == 2) return Wind(p.A_star, p.n_ism / unit::cm3, p.n0 / unit::cm3);
        // general k_m: PyWind's CGS closure (pymodel.cpp:167-185) wrapped by convert_unit_medium (pymodel.cpp:212-224)
        const Real k_m = p.k_m, n_ism = p.n_ism, n0 = p.n0;
        constexpr Real r0_cgs = 1e17;
        const Real mp_cgs = con::mp / unit::g;
        const Real A_cgs = p.A_star * 5e11 * std::pow(r0_cgs, k_m - 2);
        const Real rho_ism_cgs = n_ism * mp_cgs;
        const Real r0k_cgs = A_cgs / (n0 * 1.3 * mp_cgs);
        Medium medium;
        medium.rho = [=](Real, Real, Real r) noexcept {
            return (A_cgs / (r0k_cgs + std::pow(r / unit::cm, k_m)) + rho_ism_cgs) * (unit::g / unit::cm3);
        };
        medium.isotropic = true;
        return medium;
    }
    throw std::invalid_argument("unknown medium_type");
}

struct Pipeline {
    Coord coord;
    Shock shock;
    Observer obs;
    SynElectronGrid elec;
    SynPhotonGrid phot;
    bool ssc{false}, kn{false};
    // reverse shock (Model(rvs_rad=...)): same EAT grids, own electrons / photons (pymodel.h:940-958)
    bool rvs{false}, rvs_ssc{false}, rvs_kn{false};
    Shock rvs_shock;
    SynElectronGrid rvs_elec;
    SynPhotonGrid rvs_phot;
};

void run_pipeline(const vag_model_params& p, Array const& t_obs, Pipeline& out) {
    RadParams rad{p.eps_e, p.eps_B, p.p, p.xi_e};
    rad.radiative = p.radiative_fireball != 0;
    JetVariant jet = make_jet(p);
    MediumVariant med = make_medium(p);
    const Real theta_w = con::pi / 2; // pymodel.h:866
    const Real lumi_dist = p.lumi_dist * unit::cm;
    out.rvs = (p.flags & VAG_FLAG_RVS) != 0;
    if (out.rvs) {
        RadParams rrad{p.rvs_eps_e, p.rvs_eps_B, p.rvs_p, p.rvs_xi_e};
        rrad.radiative = rad.radiative;
        auto [coord, shock, rshock] = solve_shock_pair_like(jet, med, t_obs, theta_w, p, rad, rrad);
        out.coord = std::move(coord);
        out.shock = std::move(shock);
        out.rvs_shock = std::move(rshock);
    } else {
        auto [coord, shock] = solve_fwd_shock_like(jet, med, t_obs, theta_w, p, rad);
        out.coord = std::move(coord);
        out.shock = std::move(shock);
    }
    out.obs.observe(out.coord, out.shock, lumi_dist, p.z);
    out.elec = generate_syn_electrons(out.shock, out.coord);
    out.phot = generate_syn_photons(out.shock, out.elec, out.coord);
    out.ssc = (p.flags & VAG_FLAG_SSC) != 0;
    out.kn = (p.flags & VAG_FLAG_KN) != 0;
    if (out.ssc) {  // apply_ic_cooling, pybind/pymodel.h:567-577
        if (out.kn)
            KN_cooling(out.elec, out.phot, out.shock, out.coord);
        else
            Thomson_cooling(out.elec, out.phot, out.shock, out.coord);
    }
    if (out.rvs) {
        out.rvs_elec = generate_syn_electrons(out.rvs_shock, out.coord);
        out.rvs_phot = generate_syn_photons(out.rvs_shock, out.rvs_elec, out.coord);
        out.rvs_ssc = (p.flags & VAG_FLAG_RVS_SSC) != 0;
        out.rvs_kn = (p.flags & VAG_FLAG_RVS_KN) != 0;
        if (out.rvs_ssc) {
            if (out.rvs_kn)
                KN_cooling(out.rvs_elec, out.rvs_phot, out.rvs_shock, out.coord);
            else
                Thomson_cooling(out.rvs_elec, out.rvs_phot, out.rvs_shock, out.coord);
        }
    }
}

// SSC photons with the per-k observation band clamp of single_shock_emission, pybind/pymodel.h:896-914
auto make_ic_photons(Pipeline& pl, Array const& nu_obs, bool rvs = false) {
    const Real lg2_1pz = fast_log2(pl.obs.one_plus_z);
    const Real lg2_nu_lo = fast_log2(xt::amin(nu_obs)()) + lg2_1pz;
    const Real lg2_nu_hi = fast_log2(xt::amax(nu_obs)()) + lg2_1pz;
    const Array lg2_dop_min_k = xt::amin(pl.obs.lg2_doppler, {0, 1});
    const Array lg2_dop_max_k = xt::amax(pl.obs.lg2_doppler, {0, 1});
    const Array nu_eval_min_k = xt::exp2(lg2_nu_lo - lg2_dop_max_k);
    const Array nu_eval_max_k = xt::exp2(lg2_nu_hi - lg2_dop_min_k);
    if (rvs) return generate_IC_photons(pl.rvs_elec, pl.rvs_phot, pl.rvs_kn, pl.coord, nu_eval_min_k, nu_eval_max_k);
    return generate_IC_photons(pl.elec, pl.phot, pl.kn, pl.coord, nu_eval_min_k, nu_eval_max_k);
}

} // namespace

#define VAG_REF_API extern "C" __attribute__((visibility("default")))

VAG_REF_API const char* vag_ref_last_error(void) {
    return g_err.c_str();
}

// Model.flux_density_grid: out[nnu][nt]
VAG_REF_API int vag_ref_flux_density_grid(const vag_model_params* p, const double* t, int nt, const double* nu, int nnu,
                                          double* out) {
    try {
        Array t_obs = Array::from_shape({size_t(nt)});
        Array nu_obs = Array::from_shape({size_t(nnu)});
        for (int i = 0; i < nt; ++i) t_obs(i) = t[i] * unit::sec;
        for (int i = 0; i < nnu; ++i) nu_obs(i) = nu[i] * unit::Hz;
        Pipeline pl;
        run_pipeline(*p, t_obs, pl);
        // each component is converted to CGS first, then summed (flux_func + PyFlux::calc_total, pymodel.cpp:350-364,506-510)
        MeshGrid F = pl.obs.specific_flux(t_obs, nu_obs, pl.phot) / unit::flux_den_cgs;
        if (pl.ssc) {
            auto ic = make_ic_photons(pl, nu_obs);
            const MeshGrid G = pl.obs.specific_flux(t_obs, nu_obs, ic) / unit::flux_den_cgs;
            F += G;
        }
        if (pl.rvs) {
            const MeshGrid R = pl.obs.specific_flux(t_obs, nu_obs, pl.rvs_phot) / unit::flux_den_cgs;
            F += R;
            if (pl.rvs_ssc) {
                auto ic = make_ic_photons(pl, nu_obs, true);
                const MeshGrid G = pl.obs.specific_flux(t_obs, nu_obs, ic) / unit::flux_den_cgs;
                F += G;
            }
        }
        for (int l = 0; l < nnu; ++l)
            for (int i = 0; i < nt; ++i) out[size_t(l) * nt + i] = F(l, i);
        return 0;
    } catch (std::exception const& e) {
        g_err = e.what();
        return -1;
    }
}

// Model.flux_density_grid components: fwd.sync and fwd.ssc, each [nnu][nt] (ssc zeros when disabled)
VAG_REF_API int vag_ref_flux_density_grid_components(const vag_model_params* p, const double* t, int nt, const double* nu,
                                                     int nnu, double* out_sync, double* out_ssc) {
    try {
        Array t_obs = Array::from_shape({size_t(nt)});
        Array nu_obs = Array::from_shape({size_t(nnu)});
        for (int i = 0; i < nt; ++i) t_obs(i) = t[i] * unit::sec;
        for (int i = 0; i < nnu; ++i) nu_obs(i) = nu[i] * unit::Hz;
        Pipeline pl;
        run_pipeline(*p, t_obs, pl);
        MeshGrid F = pl.obs.specific_flux(t_obs, nu_obs, pl.phot);
        for (int l = 0; l < nnu; ++l)
            for (int i = 0; i < nt; ++i) out_sync[size_t(l) * nt + i] = F(l, i) / unit::flux_den_cgs;
        for (size_t q = 0; q < size_t(nnu) * nt; ++q) out_ssc[q] = 0;
        if (pl.ssc) {
            auto ic = make_ic_photons(pl, nu_obs);
            MeshGrid G = pl.obs.specific_flux(t_obs, nu_obs, ic);
            for (int l = 0; l < nnu; ++l)
                for (int i = 0; i < nt; ++i) out_ssc[size_t(l) * nt + i] = G(l, i) / unit::flux_den_cgs;
        }
        return 0;
    } catch (std::exception const& e) {
        g_err = e.what();
        return -1;
    }
}

// All four FluxDict components of the grid: out4 = {fwd.sync, fwd.ssc, rvs.sync, rvs.ssc}, each [nnu][nt] (zeros when off)
VAG_REF_API int vag_ref_flux_density_grid_components4(const vag_model_params* p, const double* t, int nt, const double* nu,
                                                      int nnu, double* const* out4) {
    try {
        Array t_obs = Array::from_shape({size_t(nt)});
        Array nu_obs = Array::from_shape({size_t(nnu)});
        for (int i = 0; i < nt; ++i) t_obs(i) = t[i] * unit::sec;
        for (int i = 0; i < nnu; ++i) nu_obs(i) = nu[i] * unit::Hz;
        Pipeline pl;
        run_pipeline(*p, t_obs, pl);
        auto store = [&](double* dst, MeshGrid const& F) {
            for (int l = 0; l < nnu; ++l)
                for (int i = 0; i < nt; ++i) dst[size_t(l) * nt + i] = F(l, i) / unit::flux_den_cgs;
        };
        for (int c = 0; c < 4; ++c)
            for (size_t q = 0; q < size_t(nnu) * nt; ++q) out4[c][q] = 0;
        store(out4[0], pl.obs.specific_flux(t_obs, nu_obs, pl.phot));
        if (pl.ssc) {
            auto ic = make_ic_photons(pl, nu_obs);
            store(out4[1], pl.obs.specific_flux(t_obs, nu_obs, ic));
        }
        if (pl.rvs) {
            store(out4[2], pl.obs.specific_flux(t_obs, nu_obs, pl.rvs_phot));
            if (pl.rvs_ssc) {
                auto ic = make_ic_photons(pl, nu_obs, true);
                store(out4[3], pl.obs.specific_flux(t_obs, nu_obs, ic));
            }
        }
        return 0;
    } catch (std::exception const& e) {
        g_err = e.what();
        return -1;
    }
}

// Model.flux_density (series): out[n]
VAG_REF_API int vag_ref_flux_density(const vag_model_params* p, const double* t, const double* nu, int n, double* out) {
    try {
        Array t_obs = Array::from_shape({size_t(n)});
        Array nu_obs = Array::from_shape({size_t(n)});
        for (int i = 0; i < n; ++i) {
            t_obs(i) = t[i] * unit::sec;
            nu_obs(i) = nu[i] * unit::Hz;
        }
        Pipeline pl;
        run_pipeline(*p, t_obs, pl);
        Array F = pl.obs.specific_flux_series(t_obs, nu_obs, pl.phot) / unit::flux_den_cgs;
        if (pl.ssc) {
            auto ic = make_ic_photons(pl, nu_obs);
            const Array G = pl.obs.specific_flux_series(t_obs, nu_obs, ic) / unit::flux_den_cgs;
            F += G;
        }
        if (pl.rvs) {
            const Array R = pl.obs.specific_flux_series(t_obs, nu_obs, pl.rvs_phot) / unit::flux_den_cgs;
            F += R;
            if (pl.rvs_ssc) {
                auto ic = make_ic_photons(pl, nu_obs, true);
                const Array G = pl.obs.specific_flux_series(t_obs, nu_obs, ic) / unit::flux_den_cgs;
                F += G;
            }
        }
        for (int i = 0; i < n; ++i) out[i] = F(i);
        return 0;
    } catch (std::exception const& e) {
        g_err = e.what();
        return -1;
    }
}

// Model.flux (band): out[nt]
VAG_REF_API int vag_ref_flux(const vag_model_params* p, const double* t, int nt, double nu_min, double nu_max,
                             int num_nu, double* out) {
    try {
        Array t_obs = Array::from_shape({size_t(nt)});
        for (int i = 0; i < nt; ++i) t_obs(i) = t[i] * unit::sec;
        const Array nu_obs = xt::logspace(std::log10(nu_min * unit::Hz), std::log10(nu_max * unit::Hz), size_t(num_nu));
        Pipeline pl;
        run_pipeline(*p, t_obs, pl);
        // each enabled component is integrated and converted on its own, then summed (pymodel.cpp:350-364,391-410)
        Array F = pl.obs.flux(t_obs, nu_obs, pl.phot) / unit::flux_cgs;
        if (pl.ssc) {
            auto ic = make_ic_photons(pl, nu_obs);
            const Array G = pl.obs.flux(t_obs, nu_obs, ic) / unit::flux_cgs;
            F += G;
        }
        if (pl.rvs) {
            const Array R = pl.obs.flux(t_obs, nu_obs, pl.rvs_phot) / unit::flux_cgs;
            F += R;
            if (pl.rvs_ssc) {
                auto ic = make_ic_photons(pl, nu_obs, true);
                const Array G = pl.obs.flux(t_obs, nu_obs, ic) / unit::flux_cgs;
                F += G;
            }
        }
        for (int i = 0; i < nt; ++i) out[i] = F(i);
        return 0;
    } catch (std::exception const& e) {
        g_err = e.what();
        return -1;
    }
}

// Model.details-like intermediates.  Two-call protocol: shape first (out == NULL), then arrays.
// Extra arrays beyond vag_details_out are returned through `extra` (all [n_theta][n_t] unless noted):
//   extra[0]=gamma_m extra[1]=gamma_c extra[2]=gamma_a extra[3]=gamma_M extra[4]=N_e extra[5]=column_den
//   extra[6]=nu_m extra[7]=nu_c extra[8]=nu_a extra[9]=nu_M extra[10]=I_nu_max  (code units)
//   extra[11]=lg2_t extra[12]=lg2_doppler extra[13]=lg2_geom : [n_phi_eff][n_theta][n_t]
//   extra[14]=log2 I_nu at probe log2-frequencies: [n_theta][n_t][n_probe]
static int details_impl(const vag_model_params* p, double t_min, double t_max, vag_details_shape* shape,
                        const vag_details_out* out, double** extra, int n_extra, int* n_phi_eff, const double* probe_lg2_nu,
                        int n_probe, bool want_rvs) {
    try {
        Array t_obs = Array::from_shape({size_t(2)});
        t_obs(0) = t_min * unit::sec;
        t_obs(1) = t_max * unit::sec;
        Pipeline pl;
        run_pipeline(*p, t_obs, pl);
        if (want_rvs && !pl.rvs) throw std::invalid_argument("model has no reverse shock");
        Shock const& shock = want_rvs ? pl.rvs_shock : pl.shock;
        SynElectronGrid const& elec = want_rvs ? pl.rvs_elec : pl.elec;
        SynPhotonGrid const& phot = want_rvs ? pl.rvs_phot : pl.phot;
        auto const& c = pl.coord;
        const size_t nphi = c.phi.size(), nth = c.theta.size(), nt = c.t.shape()[2];
        shape->n_phi = int(nphi);
        shape->n_theta = int(nth);
        shape->n_t = int(nt);
        shape->n_reps = int(c.theta_reps.size());
        shape->symmetry = int(c.symmetry);
        shape->phi_mirrored = c.phi_mirrored ? 1 : 0;
        const size_t nphi_eff = pl.obs.lg2_t.shape()[0];
        if (n_phi_eff) *n_phi_eff = int(nphi_eff);
        if (!out) return 0;
        auto copy2 = [&](double* dst, auto const& src, double scale) {
            if (!dst) return;
            for (size_t j = 0; j < nth; ++j)
                for (size_t k = 0; k < nt; ++k) dst[j * nt + k] = src(0, j, k) * scale;
        };
        if (out->phi)
            for (size_t i = 0; i < nphi; ++i) out->phi[i] = c.phi(i);
        if (out->theta)
            for (size_t j = 0; j < nth; ++j) out->theta[j] = c.theta(j);
        copy2(out->t_src, c.t, 1 / unit::sec);
        copy2(out->Gamma, shock.Gamma, 1);
        copy2(out->r, shock.r, 1 / unit::cm);
        copy2(out->t_comv, shock.t_comv, 1 / unit::sec);
        copy2(out->B, shock.B, 1 / unit::Gauss);
        copy2(out->N_p, shock.N_p, 1);
        copy2(out->Gamma_th, shock.Gamma_th, 1);
        auto ex = [&](int idx) -> double* { return (extra && idx < n_extra) ? extra[idx] : nullptr; };
        for (size_t j = 0; j < nth; ++j)
            for (size_t k = 0; k < nt; ++k) {
                auto const& e = elec(0, j, k);
                auto const& ph = phot(0, j, k);
                const size_t o = j * nt + k;
                if (ex(0)) ex(0)[o] = e.gamma_m;
                if (ex(1)) ex(1)[o] = e.gamma_c;
                if (ex(2)) ex(2)[o] = e.gamma_a;
                if (ex(3)) ex(3)[o] = e.gamma_M;
                if (ex(4)) ex(4)[o] = e.N_e;
                if (ex(5)) ex(5)[o] = e.column_den;
                if (ex(6)) ex(6)[o] = ph.nu_m;
                if (ex(7)) ex(7)[o] = ph.nu_c;
                if (ex(8)) ex(8)[o] = ph.nu_a;
                if (ex(9)) ex(9)[o] = ph.nu_M;
                if (ex(10)) ex(10)[o] = ph.I_nu_max;
                if (ex(14))
                    for (int q = 0; q < n_probe; ++q) ex(14)[o * n_probe + q] = ph.compute_log2_I_nu(probe_lg2_nu[q]);
                if (ex(15)) ex(15)[o] = double(shock.injection_idx(0, j));
            }
        for (size_t i = 0; i < nphi_eff; ++i)
            for (size_t j = 0; j < nth; ++j)
                for (size_t k = 0; k < nt; ++k) {
                    const size_t o = (i * nth + j) * nt + k;
                    if (ex(11)) ex(11)[o] = pl.obs.lg2_t(i, j, k);
                    if (ex(12)) ex(12)[o] = pl.obs.lg2_doppler(i, j, k);
                    if (ex(13)) ex(13)[o] = pl.obs.lg2_geom_factor(i, j, k);
                }
        return 0;
    } catch (std::exception const& e) {
        g_err = e.what();
        return -1;
    }
}

VAG_REF_API int vag_ref_details(const vag_model_params* p, double t_min, double t_max, vag_details_shape* shape,
                                const vag_details_out* out, double** extra, int n_extra, int* n_phi_eff,
                                const double* probe_lg2_nu, int n_probe) {
    return details_impl(p, t_min, t_max, shape, out, extra, n_extra, n_phi_eff, probe_lg2_nu, n_probe, false);
}

// Same protocol for the reverse shock's arrays; extra[15] = injection_idx per cell [n_theta][n_t].
VAG_REF_API int vag_ref_details_rvs(const vag_model_params* p, double t_min, double t_max, vag_details_shape* shape,
                                    const vag_details_out* out, double** extra, int n_extra, int* n_phi_eff,
                                    const double* probe_lg2_nu, int n_probe) {
    return details_impl(p, t_min, t_max, shape, out, extra, n_extra, n_phi_eff, probe_lg2_nu, n_probe, true);
}
