// ref_harness.cc - drives the REFERENCE's own physics headers (included from where they lie under
// /root/reference at build time; nothing is copied) behind the oracle's state container, so that
// tests can run reference and restatement on identical inputs.  Built only in the build container
// by oracle/Makefile into oracle/_ref/libelmref.so (git-ignored; travels to the GPU box as a binary).
//
// TEST INFRASTRUCTURE ONLY - see elm_oracle.h.
//
// Covered: every hot-path header that compiles without netcdf-c:
//   canopy_hydrology.h  surface_radiation.h  canopy_temperature.h (+qsat.h, surface_resistance.h)
//   bareground_fluxes.h (+friction_velocity.h)  snow_snicar.h  soil_moist_stress.h  atm_physics.h
//   soil_thermal_properties.h  pentadiagonal_solver.h  phase_change.h  surface_fluxes.h  conserved_quantity_evaluators.h
//   init_topography.h  init_snow_state.h  soil_texture_hydraulic_model.h  init_soil_state.h
// Not in this file: canopy_fluxes.h, photosynthesis.h, surface_albedo.h (pft_data.h -> read_input.hh -> read_netcdf.hh ->
//   netcdf.h) - ref_harness_canopy.cc runs them (it skips the readers through read_input.hh's own include guard).
// The loops below follow the argument wiring of driver/kokkos/*_kokkos.cc (cited per function).

#include "array.hh"
#include "elm_constants.h"
#include "land_data.h"

#include "atm_physics.h"
#include "bareground_fluxes.h"
#include "canopy_hydrology.h"
#include "canopy_temperature.h"
#include "friction_velocity.h"
#include "qsat.h"
#include "snow_snicar.h"
#include "soil_moist_stress.h"
#include "surface_radiation.h"
// soil / snow temperature: the headers that build without Kokkos (soil_temperature.h, soil_temp_rhs.h and
// soil_temp_lhs.h include invoke_kernel.hh, whose serial branch does not compile)
#include "pentadiagonal_solver.h"
#include "phase_change.h"
#include "soil_thermal_properties.h"
#include "conserved_quantity_evaluators.h"
#include "surface_fluxes.h"
#include "init_timestep.h"
#include "atm_physics.h"
#include "phenology_physics.h"
// cold-start initialisation (initialize_elm_kokkos.cc:373-428)
#include "init_snow_state.h"
#include "init_soil_state.h"
#include "init_topography.h"
#include "soil_texture_hydraulic_model.h"
// the host scalars of kokkos_init_timestep: the reference's two small source files, compiled into this library as they lie
#include "day_length.h"
#include "incident_shortwave.h"
#include "day_length.cc"
#include "incident_shortwave.cc"

#include "elm_oracle.h"

#include <cstring>
#include <exception>
#ifdef _OPENMP
#include <omp.h>
#endif

using AD1 = ELM::Array<double, 1>;
using AI1 = ELM::Array<int, 1>;
using AD2 = ELM::Array<double, 2>;
using AD3 = ELM::Array<double, 3>;

static ELM::LandType land_of(const elmo_state* S)
{
  ELM::LandType L;
  L.ltype = S->land.ltype;
  L.ctype = S->land.ctype;
  L.vtype = S->land.vtype;
  L.urbpoi = S->land.urbpoi != 0;
  L.lakpoi = S->land.lakpoi != 0;
  return L;
}

#define V(f, n) AD1(n, S->f + (size_t)c * (n))
static const uint32_t REF_THREW = 1u << 31;

extern "C" {

// The five wrapper loops below run under "#pragma omp parallel for schedule(static)": what
// Kokkos::parallel_for(RangePolicy<OpenMP>(0, ncols)) does with the same lambdas (src/utils/invoke_kernel.hh:24-27).
// Columns are independent, so results do not depend on the thread count.
int elmref_max_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void elmref_set_threads(int n)
{
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#endif
}

// canopy_hydrology_kokkos.cc:98-112
void elmref_frac_wet(elmo_state* S)
{
  const ELM::LandType L = land_of(S);
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < S->ncols; c++) {
    ELM::canopy_hydrology::fraction_wet(L, S->frac_veg_nosno[c], S->dewmx, S->elai[c], S->esai[c], S->h2ocan[c],
                                        S->fwet[c], S->fdry[c]);
  }
}

// canopy_hydrology_kokkos.cc:7-95
void elmref_canopy_hydrology(elmo_state* S, double dt)
{
  const ELM::LandType L = land_of(S);
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < S->ncols; c++) {
    double qflx_irrig = 0.0;
    double qflx_candrip = 0.0, qflx_through_snow = 0.0, qflx_through_rain = 0.0, fracsnow = 0.0, fracrain = 0.0;
    ELM::canopy_hydrology::interception(L, S->frac_veg_nosno[c], S->forc_rain[c], S->forc_snow[c], S->dewmx,
                                        S->elai[c], S->esai[c], dt, S->h2ocan[c], qflx_candrip, qflx_through_snow,
                                        qflx_through_rain, fracsnow, fracrain);
    ELM::canopy_hydrology::ground_flux(L, S->do_capsnow[c], S->frac_veg_nosno[c], S->forc_rain[c], S->forc_snow[c],
                                       qflx_irrig, qflx_candrip, qflx_through_snow, qflx_through_rain, fracsnow,
                                       fracrain, S->qflx_snwcp_liq[c], S->qflx_snwcp_ice[c], S->qflx_snow_grnd[c],
                                       S->qflx_rain_grnd[c]);
    ELM::canopy_hydrology::snow_init(L, dt, S->do_capsnow[c], S->oldfflag, S->forc_tbot[c], S->t_grnd[c],
                                     S->qflx_snow_grnd[c], S->qflx_snow_melt[c], S->n_melt[c], S->snow_depth[c],
                                     S->h2osno[c], S->int_snow[c], V(swe_old, 5), V(h2osoi_liq, 20), V(h2osoi_ice, 20),
                                     V(t_soisno, 20), V(frac_iceold, 20), S->snl[c], V(dz, 20), V(zsoi, 20),
                                     V(zisoi, 21), V(snw_rds, 5), S->frac_sno_eff[c], S->frac_sno[c]);
    ELM::canopy_hydrology::fraction_h2osfc(L, S->micro_sigma[c], S->h2osno[c], S->h2osfc[c], V(h2osoi_liq, 20),
                                           S->frac_sno[c], S->frac_sno_eff[c], S->frac_h2osfc[c]);
  }
}

// surface_radiation_kokkos.cc:7-97 (library is built with -DNDEBUG: the layer-sum assert of
// surface_radiation_impl.hh:173 would otherwise abort the test process)
void elmref_surface_radiation(elmo_state* S)
{
  const ELM::LandType L = land_of(S);
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < S->ncols; c++) {
    double trd_[2] = {0.0, 0.0}, tri_[2] = {0.0, 0.0};
    AD1 trd(2, trd_), tri(2, tri_);
    ELM::surface_radiation::canopy_sunshade_fractions(L, S->nrad[c], S->elai[c], V(tlai_z, 1), V(fsun_z, 1),
                                                      V(forc_solad, 2), V(forc_solai, 2), V(fabd_sun_z, 1),
                                                      V(fabd_sha_z, 1), V(fabi_sun_z, 1), V(fabi_sha_z, 1),
                                                      V(parsun_z, 1), V(parsha_z, 1), V(laisun_z, 1), V(laisha_z, 1),
                                                      S->laisun[c], S->laisha[c]);
    ELM::surface_radiation::initialize_flux(L, S->sabg_soil[c], S->sabg_snow[c], S->sabg[c], S->sabv[c], S->fsa[c],
                                            V(sabg_lyr, 6));
    ELM::surface_radiation::total_absorbed_radiation(L, S->snl[c], V(ftdd, 2), V(ftid, 2), V(ftii, 2), V(forc_solad, 2),
                                                     V(forc_solai, 2), V(fabd, 2), V(fabi, 2), V(albsod, 2),
                                                     V(albsoi, 2), V(albsnd, 2), V(albsni, 2), V(albgrd, 2),
                                                     V(albgri, 2), S->sabv[c], S->fsa[c], S->sabg[c], S->sabg_soil[c],
                                                     S->sabg_snow[c], trd, tri);
    ELM::surface_radiation::layer_absorbed_radiation(L, S->snl[c], S->sabg[c], S->sabg_snow[c], S->snow_depth[c],
                                                     V(flx_absdv, 6), V(flx_absdn, 6), V(flx_absiv, 6), V(flx_absin, 6),
                                                     trd, tri, V(sabg_lyr, 6));
    ELM::surface_radiation::reflected_radiation(L, V(albd, 2), V(albi, 2), V(forc_solad, 2), V(forc_solai, 2),
                                                S->fsr[c]);
  }
}

// canopy_temperature_kokkos.cc:6-131
void elmref_canopy_temperature(elmo_state* S)
{
  const ELM::LandType L = land_of(S);
  AD1 displar(ELMO_MXPFT, S->displar), z0mr(ELMO_MXPFT, S->z0mr);
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < S->ncols; c++) {
    double qred = 0.0, hr = 0.0, soilalpha = 0.0;
    bool veg_active = S->veg_active[c] != 0;
    ELM::canopy_temperature::old_ground_temp(L, S->t_h2osfc[c], V(t_soisno, 20), S->t_h2osfc_bef[c], V(tssbef, 20));
    ELM::canopy_temperature::ground_temp(L, S->snl[c], S->frac_sno_eff[c], S->frac_h2osfc[c], S->t_h2osfc[c],
                                         V(t_soisno, 20), S->t_grnd[c]);
    ELM::canopy_temperature::calc_soilalpha(L, S->frac_sno[c], S->frac_h2osfc[c], V(h2osoi_liq, 20), V(h2osoi_ice, 20),
                                            V(dz, 20), V(t_soisno, 20), V(watsat, 15), V(sucsat, 15), V(bsw, 15),
                                            V(watdry, 15), V(watopt, 15), qred, hr, soilalpha);
    ELM::canopy_temperature::calc_soilbeta(L, S->frac_sno[c], S->frac_h2osfc[c], V(watsat, 15), V(watfc, 15),
                                           V(h2osoi_liq, 20), V(h2osoi_ice, 20), V(dz, 20), S->soilbeta[c]);
    ELM::canopy_temperature::humidities(L, S->snl[c], S->forc_qbot[c], S->forc_pbot[c], S->t_h2osfc[c], S->t_grnd[c],
                                        S->frac_sno[c], S->frac_sno_eff[c], S->frac_h2osfc[c], qred, hr,
                                        V(t_soisno, 20), S->qg_snow[c], S->qg_soil[c], S->qg[c], S->qg_h2osfc[c],
                                        S->dqgdT[c]);
    ELM::canopy_temperature::ground_properties(L, S->snl[c], S->frac_sno[c], S->forc_thbot[c], S->forc_qbot[c],
                                               S->elai[c], S->esai[c], S->htop[c], displar, z0mr, V(h2osoi_liq, 20),
                                               V(h2osoi_ice, 20), S->emg[c], S->emv[c], S->htvp[c], S->z0mg[c],
                                               S->z0hg[c], S->z0qg[c], S->z0mv[c], S->z0hv[c], S->z0qv[c], S->thv[c],
                                               S->z0m[c], S->displa[c]);
    ELM::canopy_temperature::forcing_height(L, veg_active, S->frac_veg_nosno[c], S->z0m[c], S->z0mg[c],
                                            S->forc_tbot[c], S->displa[c], S->forc_hgt_u_patch[c],
                                            S->forc_hgt_t_patch[c], S->forc_hgt_q_patch[c], S->thm[c]);
    ELM::canopy_temperature::init_energy_fluxes(L, S->eflx_sh_tot[c], S->eflx_lh_tot[c], S->eflx_sh_veg[c],
                                                S->qflx_evap_tot[c], S->qflx_evap_veg[c], S->qflx_tran_veg[c]);
  }
}

// bareground_fluxes_kokkos.cc:7-123
void elmref_bareground_fluxes(elmo_state* S)
{
  const ELM::LandType L = land_of(S);
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < S->ncols; c++) {
    double zldis = 0.0, displa = 0.0, dth = 0.0, dqh = 0.0, obu = 0.0, ur = 0.0, um = 0.0, temp1 = 0.0, temp2 = 0.0,
           temp12m = 0.0, temp22m = 0.0, ustar = 0.0;
    double forc_rho = ELM::atm_forcing_physics::derive_forc_rho(S->forc_pbot[c], S->forc_qbot[c], S->forc_tbot[c]);
    ELM::bareground_fluxes::initialize_flux(L, S->frac_veg_nosno[c], S->forc_u[c], S->forc_v[c], S->forc_qbot[c],
                                            S->forc_thbot[c], S->forc_hgt_u_patch[c], S->thm[c], S->thv[c],
                                            S->t_grnd[c], S->qg[c], S->z0mg[c], S->dlrad[c], S->ulrad[c], zldis, displa,
                                            dth, dqh, obu, ur, um);
    ELM::bareground_fluxes::stability_iteration(L, S->frac_veg_nosno[c], S->forc_hgt_t_patch[c],
                                                S->forc_hgt_u_patch[c], S->forc_hgt_q_patch[c], S->z0mg[c], zldis,
                                                displa, dth, dqh, ur, S->forc_qbot[c], S->forc_thbot[c], S->thv[c],
                                                S->z0hg[c], S->z0qg[c], obu, um, temp1, temp2, temp12m, temp22m, ustar);
    ELM::bareground_fluxes::compute_flux(L, S->frac_veg_nosno[c], S->snl[c], forc_rho, S->soilbeta[c], S->dqgdT[c],
                                         S->htvp[c], S->t_h2osfc[c], S->qg_snow[c], S->qg_soil[c], S->qg_h2osfc[c],
                                         V(t_soisno, 20), S->forc_pbot[c], dth, dqh, temp1, temp2, temp12m, temp22m,
                                         ustar, S->forc_qbot[c], S->thm[c], S->cgrnds[c], S->cgrndl[c], S->cgrnd[c],
                                         S->eflx_sh_grnd[c], S->eflx_sh_tot[c], S->eflx_sh_snow[c], S->eflx_sh_soil[c],
                                         S->eflx_sh_h2osfc[c], S->qflx_evap_soi[c], S->qflx_evap_tot[c],
                                         S->qflx_ev_snow[c], S->qflx_ev_soil[c], S->qflx_ev_h2osfc[c], S->t_ref2m[c],
                                         S->q_ref2m[c], S->rh_ref2m[c]);
  }
}

// The SNICAR half of albedo_kokkos.cc:96-301: both passes (direct, diffuse) of
// init_timestep -> snow_aerosol_mie_params -> snow_radiative_transfer_solver -> snow_albedo_radiation_factor
// with the wrapper's zero-filled scratch.  Inputs: coszen, h2osno, snl, h2osoi_liq/ice, snw_rds, cnc_*,
// albsoi (taken from the state as it stands).  Outputs: S.albsnd, S.albsni and the caller's
// flx_absd_snw / flx_absi_snw [ncols][6][2] (wrapper-local in the reference).
void elmref_snicar(elmo_state* S, double* flx_absd_snw_out, double* flx_absi_snw_out)
{
  const bool urbpoi = S->land.urbpoi != 0;
  elmo_snicar* T = &S->snicar;
#define T1(n) AD1(5, T->n)
#define TM(n) AD2(5, ELMO_MIE_N, T->n)
#define TB(n) AD2(10, 5, T->n)
  AD3 bcenh(8, 10, 5, T->bcenh);
  for (int64_t c = 0; c < S->ncols; c++) {
    int snw_rds_lcl_[5] = {0};
    double h2osoi_ice_lcl_[5] = {0}, h2osoi_liq_lcl_[5] = {0}, albout_lcl_[5] = {0}, flx_slrd_lcl_[5] = {0},
           flx_slri_lcl_[5] = {0};
    double flx_abs_lcl_[30] = {0}, mss_[40] = {0}, g_star_[25] = {0}, omega_star_[25] = {0}, tau_star_[25] = {0};
    double* flx_absd = flx_absd_snw_out + (size_t)c * 12;
    double* flx_absi = flx_absi_snw_out + (size_t)c * 12;
    std::memset(flx_absd, 0, 12 * sizeof(double));
    std::memset(flx_absi, 0, 12 * sizeof(double));
    // surface_albedo::init_timestep's aerosol wiring (surface_albedo_impl.hh:141-150) is not called in this file (the whole
    // albedo wrapper by the reference's functions: ref_harness_canopy.cc); it is a plain copy, reproduced here as input preparation
    for (int i = 0; i < 5; i++) {
      mss_[i * 8 + 0] = S->cnc_bcphi[c * 5 + i];
      mss_[i * 8 + 1] = S->cnc_bcpho[c * 5 + i];
      mss_[i * 8 + 4] = S->cnc_dst1[c * 5 + i];
      mss_[i * 8 + 5] = S->cnc_dst2[c * 5 + i];
      mss_[i * 8 + 6] = S->cnc_dst3[c * 5 + i];
      mss_[i * 8 + 7] = S->cnc_dst4[c * 5 + i];
    }
    AI1 snw_rds_lcl(5, snw_rds_lcl_);
    AD1 h2osoi_ice_lcl(5, h2osoi_ice_lcl_), h2osoi_liq_lcl(5, h2osoi_liq_lcl_), albout_lcl(5, albout_lcl_),
        flx_slrd_lcl(5, flx_slrd_lcl_), flx_slri_lcl(5, flx_slri_lcl_);
    AD2 flx_abs_lcl(6, 5, flx_abs_lcl_), mss(5, 8, mss_), g_star(5, 5, g_star_), omega_star(5, 5, omega_star_),
        tau_star(5, 5, tau_star_);
    int snl_top = 0, snl_btm = 0, flg_nosnl = 0;
    double mu_not = 0.0;
    try {
      for (int flg_slr_in = 1; flg_slr_in <= 2; flg_slr_in++) {
        AD2 flx_abs(6, 2, flg_slr_in == 1 ? flx_absd : flx_absi);
        AD1 albout(2, (flg_slr_in == 1 ? S->albsnd : S->albsni) + (size_t)c * 2);
        ELM::snow_snicar::init_timestep(urbpoi, flg_slr_in, S->coszen[c], S->h2osno[c], S->snl[c], V(h2osoi_liq, 20),
                                        V(h2osoi_ice, 20), V(snw_rds, 5), snl_top, snl_btm, flx_abs_lcl, flx_abs,
                                        flg_nosnl, h2osoi_ice_lcl, h2osoi_liq_lcl, snw_rds_lcl, mu_not, flx_slrd_lcl,
                                        flx_slri_lcl);
        ELM::snow_snicar::snow_aerosol_mie_params(
            urbpoi, flg_slr_in, snl_top, snl_btm, S->coszen[c], S->h2osno[c], snw_rds_lcl, h2osoi_ice_lcl,
            h2osoi_liq_lcl, T1(ss_alb_oc1), T1(asm_prm_oc1), T1(ext_cff_mss_oc1), T1(ss_alb_oc2), T1(asm_prm_oc2),
            T1(ext_cff_mss_oc2), T1(ss_alb_dst1), T1(asm_prm_dst1), T1(ext_cff_mss_dst1), T1(ss_alb_dst2),
            T1(asm_prm_dst2), T1(ext_cff_mss_dst2), T1(ss_alb_dst3), T1(asm_prm_dst3), T1(ext_cff_mss_dst3),
            T1(ss_alb_dst4), T1(asm_prm_dst4), T1(ext_cff_mss_dst4), TM(ss_alb_snw_drc), TM(asm_prm_snw_drc),
            TM(ext_cff_mss_snw_drc), TM(ss_alb_snw_dfs), TM(asm_prm_snw_dfs), TM(ext_cff_mss_snw_dfs), TB(ss_alb_bc1),
            TB(asm_prm_bc1), TB(ext_cff_mss_bc1), TB(ss_alb_bc2), TB(asm_prm_bc2), TB(ext_cff_mss_bc2), bcenh, mss,
            g_star, omega_star, tau_star);
        ELM::snow_snicar::snow_radiative_transfer_solver(urbpoi, flg_slr_in, flg_nosnl, snl_top, snl_btm, S->coszen[c],
                                                         S->h2osno[c], mu_not, flx_slrd_lcl, flx_slri_lcl,
                                                         V(albsoi, 2), g_star, omega_star, tau_star, albout_lcl,
                                                         flx_abs_lcl);
        ELM::snow_snicar::snow_albedo_radiation_factor(urbpoi, flg_slr_in, snl_top, S->coszen[c], mu_not, S->h2osno[c],
                                                       snw_rds_lcl, V(albsoi, 2), albout_lcl, flx_abs_lcl, albout,
                                                       flx_abs);
      }
    } catch (const std::exception&) {
      S->err_flags[c] |= REF_THREW;
    }
  }
#undef T1
#undef TM
#undef TB
}

// The soil-moisture-stress block of canopy_fluxes::initialize_flux (canopy_fluxes_impl.hh:131-139):
// btran starts at btran0 = 0; writes S.eff_porosity, S.rootr, S.btran for every column.
void elmref_soil_moist_stress(elmo_state* S)
{
  for (int64_t c = 0; c < S->ncols; c++) {
    const elmo_pft_psn* psn = &S->pft_psn[S->vtype[c]];
    S->btran[c] = 0.0;
    ELM::soil_moist_stress::calc_effective_soilporosity(V(watsat, 15), V(h2osoi_ice, 20), V(dz, 20),
                                                        V(eff_porosity, 15));
    double h2osoi_liqvol[20];
    ELM::soil_moist_stress::calc_volumetric_h2oliq(V(eff_porosity, 15), V(h2osoi_liq, 20), V(dz, 20), h2osoi_liqvol);
    ELM::soil_moist_stress::calc_root_moist_stress(h2osoi_liqvol, V(rootfr, 15), V(t_soisno, 20), psn->tc_stress,
                                                   V(sucsat, 15), V(watsat, 15), V(bsw, 15), psn->smpso, psn->smpsc,
                                                   V(eff_porosity, 15), S->altmax_indx[c], S->altmax_lastyear_indx[c],
                                                   V(rootr, 15), S->btran[c]);
  }
}

// element-wise probes of the scalar helpers used inside canopy_fluxes
void elmref_qsat(int64_t n, const double* T, const double* p, double* es, double* esdT, double* qs, double* qsdT)
{
  for (int64_t i = 0; i < n; i++) ELM::qsat(T[i], p[i], es[i], esdT[i], qs[i], qsdT[i]);
}

void elmref_forc_derived(int64_t n, const double* pbot, const double* qbot, const double* tbot, double* rho,
                         double* po2, double* pco2)
{
  for (int64_t i = 0; i < n; i++) {
    rho[i] = ELM::atm_forcing_physics::derive_forc_rho(pbot[i], qbot[i], tbot[i]);
    po2[i] = ELM::atm_forcing_physics::derive_forc_po2(pbot[i]);
    pco2[i] = ELM::atm_forcing_physics::derive_forc_pco2(pbot[i]);
  }
}

// out[i*7 + k]: um, obu (monin_obukhov_length), ustar, temp1, temp2, temp12m, temp22m - the call sequence at the
// head of each canopy_fluxes / bareground_fluxes stability iteration
void elmref_friction(int64_t n, const double* ur, const double* thv, const double* dthv, const double* zldis,
                     const double* z0m, const double* z0h, const double* z0q, const double* hgt_u,
                     const double* hgt_t, const double* hgt_q, const double* displa, double* out)
{
  for (int64_t i = 0; i < n; i++) {
    double um, obu, ustar, temp1, temp2, temp12m, temp22m;
    ELM::friction_velocity::monin_obukhov_length(ur[i], thv[i], dthv[i], zldis[i], z0m[i], um, obu);
    ELM::friction_velocity::friction_velocity_wind(hgt_u[i], displa[i], um, obu, z0m[i], ustar);
    ELM::friction_velocity::friction_velocity_temp(hgt_t[i], displa[i], obu, z0h[i], temp1);
    ELM::friction_velocity::friction_velocity_humidity(hgt_q[i], hgt_t[i], displa[i], obu, z0h[i], z0q[i], temp1,
                                                       temp2);
    ELM::friction_velocity::friction_velocity_temp2m(obu, z0h[i], temp12m);
    ELM::friction_velocity::friction_velocity_humidity2m(obu, z0h[i], z0q[i], temp12m, temp22m);
    double* o = out + i * 7;
    o[0] = um;
    o[1] = obu;
    o[2] = ustar;
    o[3] = temp1;
    o[4] = temp2;
    o[5] = temp12m;
    o[6] = temp22m;
  }
}


// soil_temperature_kokkos.cc:92-104: the soil_thermal_props lambda (dummy ltype 1, :77-79).
// thk/tk/cv [ncols][20], scal [ncols][3] = {tk_h2osfc, c_h2osfc, dz_h2osfc}
void elmref_soil_thermal(elmo_state* S, double* thk_out, double* tk_out, double* cv_out, double* scal_out)
{
  const int n = (int)S->ncols;
  AD2 h2osoi_liq(n, 20, S->h2osoi_liq), h2osoi_ice(n, 20, S->h2osoi_ice), t_soisno(n, 20, S->t_soisno), dz(n, 20, S->dz);
  AD2 watsat(n, 15, S->watsat), tkmg(n, 15, S->tkmg), tkdry(n, 15, S->tkdry), csol(n, 20, S->csol);
  AD2 zsoi(n, 20, S->zsoi), zisoi(n, 21, S->zisoi);
  AD2 thk(n, 20, thk_out), tk(n, 20, tk_out), cv(n, 20, cv_out);
  for (int c = 0; c < n; c++) {
    const int ltype = 1;
    ELM::soil_thermal::calc_soil_tk(c, ltype, h2osoi_liq, h2osoi_ice, t_soisno, dz, watsat, tkmg, tkdry, thk);
    ELM::soil_thermal::calc_snow_tk(c, S->snl[c], S->frac_sno[c], h2osoi_liq, h2osoi_ice, dz, thk);
    ELM::soil_thermal::calc_face_tk(c, S->snl[c], thk, zsoi, zisoi, tk);
    ELM::soil_thermal::calc_soil_heat_capacity(c, ltype, S->snl[c], S->h2osno[c], watsat, h2osoi_ice, h2osoi_liq, dz, csol,
                                               cv);
    ELM::soil_thermal::calc_snow_heat_capacity(c, S->snl[c], S->frac_sno[c], h2osoi_ice, h2osoi_liq, cv);
    scal_out[c * 3 + 0] = ELM::soil_thermal::calc_h2osfc_tk(c, S->h2osfc[c], thk, zsoi);
    scal_out[c * 3 + 1] = ELM::soil_thermal::calc_h2osfc_heat_capacity(S->snl[c], S->h2osfc[c], S->frac_h2osfc[c]);
    scal_out[c * 3 + 2] = ELM::soil_thermal::calc_h2osfc_height(S->snl[c], S->h2osfc[c], S->frac_h2osfc[c]);
  }
}

// soil_temperature_kokkos.cc:215-225: solver::PDMA on given systems; lhs [n][21][5], rhs [n][21] (solution on return)
void elmref_pdma(int64_t n_, int* snl, double* lhs, double* rhs)
{
  const int n = (int)n_;
  AI1 snl_(n, snl);
  AD3 LHS(n, 21, 5, lhs);
  AD2 A(n, 20, 0.0), B(n, 19, 0.0), Z(n, 21, 0.0);
  AD2 RHS(n, 21, rhs);
  for (int c = 0; c < n; c++) ELM::solver::PDMA(c, snl_, LHS, A, B, Z, RHS);
}

// soil_temperature_kokkos.cc:245-266: the phase_change lambda, with the wrapper-local dhsdT / c_h2osfc given
void elmref_phase_change(elmo_state* S, double dt, const double* dhsdT, const double* c_h2osfc)
{
  for (int64_t c = 0; c < S->ncols; c++) {
    const int ltype = 1;
    double* fact = S->fact + (size_t)c * 20;
    ELM::soil_temp::phase_change_h2osfc(S->snl[c], dt, S->frac_sno[c], S->frac_h2osfc[c], dhsdT[c], c_h2osfc[c], fact[4],
                                        S->t_h2osfc[c], S->h2osfc[c], S->xmf_h2osfc[c], S->qflx_h2osfc_ice[c],
                                        S->eflx_h2osfc_snow[c], S->h2osno[c], S->int_snow[c], S->snow_depth[c],
                                        S->h2osoi_ice[(size_t)c * 20 + 4], S->t_soisno[(size_t)c * 20 + 4]);
    ELM::soil_temp::phase_change_soisno(S->snl[c], ltype, dt, dhsdT[c], S->frac_h2osfc[c], S->frac_sno_eff[c], V(fact, 20),
                                        V(watsat, 15), V(sucsat, 15), V(bsw, 15), V(dz, 20), S->h2osno[c], S->snow_depth[c],
                                        S->xmf[c], S->qflx_snofrz[c], S->qflx_snow_melt[c], S->qflx_snomelt[c],
                                        S->eflx_snomelt[c], AI1(20, S->imelt + (size_t)c * 20), V(qflx_snofrz_lyr, 5),
                                        V(h2osoi_ice, 20), V(h2osoi_liq, 20), V(t_soisno, 20));
  }
}

// surface_fluxes_kokkos.cc:5-107
void elmref_surface_fluxes(elmo_state* S, double dt)
{
  const bool urbpoi = S->land.urbpoi != 0;
  for (int64_t c = 0; c < S->ncols; c++) {
    const int soitop = 5;
    const int snotop = soitop - S->snl[c];
    double* tssbef = S->tssbef + (size_t)c * 20;
    ELM::surface_fluxes::initial_flux_calc(urbpoi, S->snl[c], S->frac_sno_eff[c], S->frac_h2osfc[c], S->t_h2osfc_bef[c],
                                           tssbef[snotop], tssbef[soitop], S->t_grnd[c], S->cgrnds[c], S->cgrndl[c],
                                           S->eflx_sh_grnd[c], S->qflx_evap_soi[c], S->qflx_ev_snow[c], S->qflx_ev_soil[c],
                                           S->qflx_ev_h2osfc[c]);
    ELM::surface_fluxes::update_surface_fluxes(
        urbpoi, S->do_capsnow[c], S->snl[c], dt, S->t_grnd[c], S->htvp[c], S->frac_sno_eff[c], S->frac_h2osfc[c],
        S->t_h2osfc_bef[c], S->sabg_soil[c], S->sabg_snow[c], S->dlrad[c], S->frac_veg_nosno[c], S->emg[c], S->forc_lwrad[c],
        tssbef[snotop], tssbef[soitop], S->h2osoi_ice[(size_t)c * 20 + snotop], S->h2osoi_liq[(size_t)c * 20 + soitop],
        S->eflx_sh_veg[c], S->qflx_evap_veg[c], S->qflx_evap_soi[c], S->eflx_sh_grnd[c], S->qflx_ev_snow[c],
        S->qflx_ev_soil[c], S->qflx_ev_h2osfc[c], S->eflx_soil_grnd[c], S->eflx_sh_tot[c], S->qflx_evap_tot[c],
        S->eflx_lh_tot[c], S->qflx_evap_grnd[c], S->qflx_sub_snow[c], S->qflx_dew_snow[c], S->qflx_dew_grnd[c],
        S->qflx_snwcp_liq[c], S->qflx_snwcp_ice[c]);
    ELM::surface_fluxes::lwrad_outgoing(urbpoi, S->snl[c], S->frac_veg_nosno[c], S->forc_lwrad[c], S->frac_sno_eff[c],
                                        tssbef[snotop], tssbef[soitop], S->frac_h2osfc[c], S->t_h2osfc_bef[c], S->t_grnd[c],
                                        S->ulrad[c], S->emg[c], S->eflx_lwrad_out[c], S->eflx_lwrad_net[c]);
    S->soil_e_balance[c] = ELM::surface_fluxes::soil_energy_balance(
        S->land.ctype, S->snl[c], S->eflx_soil_grnd[c], S->xmf[c], S->xmf_h2osfc[c], S->frac_h2osfc[c], S->t_h2osfc[c],
        S->t_h2osfc_bef[c], dt, S->eflx_h2osfc_snow[c], S->frac_sno_eff[c], V(t_soisno, 20), V(tssbef, 20), V(fact, 20));
  }
}

// conserved_quantity_kokkos.cc:8-81; diag [ncols][8] = dtend_column_h2o, errh2o, errh2osno, dwb, errsol, errlon, errseb, netrad
void elmref_evaluate_conservation(elmo_state* S, double dt, double* diag)
{
  const double hydrology_source_sink = 0.0;
  for (int64_t c = 0; c < S->ncols; c++) {
    double* d = diag + (size_t)c * 8;
    d[0] = ELM::conservation_eval::column_water_mass(S->h2ocan[c], S->h2osno[c], S->h2osfc[c], V(h2osoi_ice, 20),
                                                     V(h2osoi_liq, 20));
    d[3] = ELM::conservation_eval::dh2o_dt(S->dtbegin_column_h2o[c], d[0], dt);
    d[1] = ELM::conservation_eval::column_water_balance_error(S->dtbegin_column_h2o[c], d[0], hydrology_source_sink,
                                                              S->forc_rain[c], S->forc_snow[c], S->qflx_evap_tot[c],
                                                              S->qflx_snwcp_ice[c], dt);
    d[2] = ELM::conservation_eval::snow_water_balance_error(
        S->snl[c], S->qflx_dew_snow[c], S->qflx_dew_grnd[c], S->qflx_sub_snow[c], S->qflx_evap_grnd[c], S->qflx_snow_melt[c],
        S->qflx_snwcp_ice[c], S->qflx_snwcp_liq[c], S->qflx_sl_top_soil[c], S->frac_sno_eff[c], S->qflx_rain_grnd[c],
        S->qflx_snow_grnd[c], S->qflx_h2osfc_ice[c], S->h2osno[c], S->h2osno_old[c], dt, S->do_capsnow[c] != 0);
    d[4] = ELM::conservation_eval::solar_shortwave_balance_error(S->fsa[c], S->fsr[c], V(forc_solad, 2), V(forc_solai, 2));
    d[5] = ELM::conservation_eval::solar_longwave_balance_error(S->eflx_lwrad_out[c], S->eflx_lwrad_net[c], S->forc_lwrad[c]);
    d[6] = ELM::conservation_eval::surface_energy_balance_error(S->sabv[c], S->sabg_chk[c], S->forc_lwrad[c],
                                                                S->eflx_lwrad_out[c], S->eflx_sh_tot[c], S->eflx_lh_tot[c],
                                                                S->eflx_soil_grnd[c]);
    d[7] = ELM::conservation_eval::net_radiation(S->fsa[c], S->eflx_lwrad_net[c]);
  }
}

// init_timestep_kokkos.cc:55-75: the per-column kernel of kokkos_init_timestep
void elmref_init_timestep(elmo_state* S)
{
  for (int64_t c = 0; c < S->ncols; c++) {
    S->h2osno_old[c] = S->h2osno[c];
    S->dtbegin_column_h2o[c] = ELM::conservation_eval::column_water_mass(S->h2ocan[c], S->h2osno[c], S->h2osfc[c],
                                                                          V(h2osoi_ice, 20), V(h2osoi_liq, 20));
    ELM::init_timestep(S->land.lakpoi != 0, S->veg_active[c] != 0, S->frac_veg_nosno_alb[c], S->snl[c], S->h2osno[c],
                       V(h2osoi_ice, 20), V(h2osoi_liq, 20), S->do_capsnow[c], S->frac_veg_nosno[c], V(frac_iceold, 20));
  }
}
// get_forcing (atm_forcing_kokkos.cc:47-75): the reference's own functors over the two bracketing records of each stream
// ([record][cell], as AtmDataManager::data), each applied to every cell in the wrapper's order
void elmref_get_forcing(elmo_state* S, const double* wt1, const double* wt2, int qbot_is_rh)
{
  using namespace ELM::atm_forcing_physics;
  const int n = (int)S->ncols;
  auto stream = [n](const double* f) {  // [col][2] of the oracle state -> [2][col]
    AD2 a(2, n);
    for (int c = 0; c < n; c++) {
      a(0, c) = f[c * 2];
      a(1, c) = f[c * 2 + 1];
    }
    return a;
  };
  AD2 tb = stream(S->atm_tbot), pb = stream(S->atm_pbot), qb = stream(S->atm_qbot), fl = stream(S->atm_flds),
      fs = stream(S->atm_fsds), pr = stream(S->atm_prec), wd = stream(S->atm_wind);
  AD1 forc_tbot(n, S->forc_tbot), forc_thbot(n, S->forc_thbot), forc_pbot(n, S->forc_pbot), forc_qbot(n, S->forc_qbot),
      forc_lwrad(n, S->forc_lwrad), coszen(n, S->coszen), forc_rain(n, S->forc_rain), forc_snow(n, S->forc_snow),
      forc_u(n, S->forc_u), forc_v(n, S->forc_v), forc_hgt(n, S->forc_hgt), hu(n, S->forc_hgt_u_patch),
      ht(n, S->forc_hgt_t_patch), hq(n, S->forc_hgt_q_patch);
  AD2 solai(n, 2, S->forc_solai), solad(n, 2, S->forc_solad);
  const int t_idx = 0;
  ProcessTBOT<AD1, AD2> f0(t_idx, wt1[0], wt2[0], tb, forc_tbot, forc_thbot);
  ProcessPBOT<AD1, AD2> f1(t_idx, wt1[1], wt2[1], pb, forc_pbot);
  ProcessQBOT<AD1, AD2, ELM::AtmForcType::QBOT> f2q(t_idx, wt1[2], wt2[2], qb, forc_tbot, forc_pbot, forc_qbot);
  ProcessQBOT<AD1, AD2, ELM::AtmForcType::RH> f2r(t_idx, wt1[2], wt2[2], qb, forc_tbot, forc_pbot, forc_qbot);
  ProcessFLDS<AD1, AD2> f3(t_idx, wt1[3], wt2[3], fl, forc_pbot, forc_qbot, forc_tbot, forc_lwrad);
  ProcessFSDS<AD1, AD2> f4(t_idx, fs, coszen, solai, solad);
  ProcessPREC<AD1, AD2> f5(t_idx, pr, forc_tbot, forc_rain, forc_snow);
  ProcessWIND<AD1, AD2> f6(t_idx, wt1[6], wt2[6], wd, forc_u, forc_v);
  ProcessZBOT<AD1> f7(forc_hgt, hu, ht, hq);
  for (int c = 0; c < n; c++) f0(c);
  for (int c = 0; c < n; c++) f1(c);
  for (int c = 0; c < n; c++) qbot_is_rh ? f2r(c) : f2q(c);
  for (int c = 0; c < n; c++) f3(c);
  for (int c = 0; c < n; c++) f4(c);
  for (int c = 0; c < n; c++) f5(c);
  for (int c = 0; c < n; c++) f6(c);
  for (int c = 0; c < n; c++) f7(c);
}

// ComputePhenology (phenology_physics_impl.hh:22-69) over the two bracketing months ([month][cell])
void elmref_phenology(elmo_state* S, double wt1, double wt2)
{
  const int n = (int)S->ncols;
  auto months = [n](const double* f) {
    AD2 a(2, n);
    for (int c = 0; c < n; c++) {
      a(0, c) = f[c * 2];
      a(1, c) = f[c * 2 + 1];
    }
    return a;
  };
  AD2 mlai = months(S->mlai), msai = months(S->msai), mhtop = months(S->mhtop), mhbot = months(S->mhbot);
  AD1 snow_depth(n, S->snow_depth), frac_sno(n, S->frac_sno), elai(n, S->elai), esai(n, S->esai), htop(n, S->htop),
      hbot(n, S->hbot), tlai(n, S->tlai), tsai(n, S->tsai);
  AI1 vtype(n, S->vtype), fvna(n, S->frac_veg_nosno_alb);
  ELM::phenology::ComputePhenology<AI1, AD1, AD2> f(mlai, msai, mhtop, mhbot, snow_depth, frac_sno, vtype, wt1, wt2, 0, elai,
                                                    esai, htop, hbot, tlai, tsai, fvna);
  for (int c = 0; c < n; c++) f(c);
}
// the "init functions" lambda of ELM::initialize_kokkos_elm (initialize_elm_kokkos.cc:373-428), with the reference's own
// functions in its order
void elmref_initialize_state(elmo_state* S)
{
  const ELM::LandType L = land_of(S);
  for (int64_t c = 0; c < S->ncols; c++) {
    S->topo_slope[c] = ELM::init_topo_slope(S->topo_slope[c]);
    S->n_melt[c] = ELM::init_melt_factor(L.ltype, S->topo_std[c]);
    S->micro_sigma[c] = ELM::init_micro_sigma(S->topo_slope[c]);
    ELM::init_snow_layers(S->snow_depth[c], L.lakpoi, S->snl[c], V(dz, 20), V(zsoi, 20), V(zisoi, 21));
    ELM::init_soil_hydraulics(S->organic_max, V(pct_sand, 15), V(pct_clay, 15), V(organic, 15), V(zsoi, 20), V(watsat, 15),
                              V(bsw, 15), V(sucsat, 15), V(watdry, 15), V(watopt, 15), V(watfc, 15), V(tkmg, 15), V(tkdry, 15),
                              V(csol, 20));
    const int vt = S->vtype[c];
    ELM::init_vegrootfr(vt, S->roota_par[vt], S->rootb_par[vt], V(zisoi, 21), V(rootfr, 15));
    ELM::init_soil_temp(L, S->snl[c], V(t_soisno, 20), S->t_grnd[c]);
    ELM::init_snow_state(L.urbpoi, S->snl[c], S->h2osno[c], S->int_snow[c], S->snow_depth[c], S->h2osfc[c], S->h2ocan[c],
                         S->frac_h2osfc[c], S->fwet[c], S->fdry[c], S->frac_sno[c], V(snw_rds, 5));
    ELM::init_soilh2o_state(L, S->snl[c], V(watsat, 15), V(t_soisno, 20), V(dz, 20), V(h2osoi_vol, 15), V(h2osoi_liq, 20),
                            V(h2osoi_ice, 20));
  }
}

// incident_shortwave.cc / day_length.cc on arrays of arguments (init_timestep_kokkos.cc:26-34)
void elmref_solar(int64_t n, const double* lat, const double* lon, const double* dt, const double* jday, double* cosz,
                  double* dayl, double* max_dayl)
{
  for (int64_t i = 0; i < n; i++) {
    cosz[i] = ELM::incident_shortwave::average_cosz(lat[i], lon[i], dt[i], jday[i]);
    dayl[i] = ELM::daylength(lat[i], ELM::incident_shortwave::declination_angle_sin(static_cast<int>(jday[i])));
    max_dayl[i] = ELM::max_daylength(lat[i]);
  }
}

} // extern "C"
