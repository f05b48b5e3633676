/*
 * elmo_driver.c - oracle state container and restatement of the reference's L3 dispatch wrappers
 * (driver/kokkos/*_kokkos.cc).  TEST INFRASTRUCTURE - see elm_oracle.h.
 *
 * Each elmo_<wrapper>() does what the reference wrapper's lambda does for column idx: same L2 call
 * order, same argument wiring (including the quirks: S.forc_tbot passed as forc_t, S.zsoi/S.zisoi as
 * z/zi, wrapper-local fabd_sun/fabd_sha, qflx_irrig hard-wired 0), with the wrapper's zero-filled
 * ViewD1(ncols) temporaries as zero-initialised locals.
 */
#include "elm_oracle.h"
#include "elmo_const.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------------ */
/* registry                                                                                         */
/* ------------------------------------------------------------------------------------------------ */
#define KIND_D 0
#define KIND_I 1
#define KIND_B 2

typedef struct {
  const char *name;
  int kind;
  int nlev;
  size_t offset;
} field_desc;

#include <stddef.h>
static const field_desc g_fields[] = {
#define ELMO_DESC(name, kind, nlev) {#name, KIND_##kind, nlev, offsetof(elmo_state, name)},
    ELMO_FIELDS(ELMO_DESC)
#undef ELMO_DESC
};
enum { NFIELDS = sizeof(g_fields) / sizeof(g_fields[0]) };

static size_t kind_size(int kind) { return kind == KIND_D ? sizeof(double) : (kind == KIND_I ? sizeof(int) : 1); }

elmo_state *elmo_create(int64_t ncols)
{
  elmo_state *S = (elmo_state *)calloc(1, sizeof(elmo_state));
  if (!S) return NULL;
  S->ncols = ncols;
  /* LandType() defaults (land_data.h:38) and ELMState defaults (elm_state.h:223-224) */
  S->land.ltype = 1;
  S->land.ctype = 0;
  S->land.vtype = 2;
  S->land.urbpoi = 0;
  S->land.lakpoi = 0;
  S->dewmx = 0.1;
  S->oldfflag = 1;
  for (int i = 0; i < NFIELDS; i++) {
    void *p = calloc((size_t)(ncols > 0 ? ncols : 1) * g_fields[i].nlev, kind_size(g_fields[i].kind));
    if (!p) {
      elmo_destroy(S);
      return NULL;
    }
    *(void **)((char *)S + g_fields[i].offset) = p;
  }
  S->err_flags = (uint32_t *)calloc((size_t)(ncols > 0 ? ncols : 1), sizeof(uint32_t));
  return S;
}

void elmo_destroy(elmo_state *S)
{
  if (!S) return;
  for (int i = 0; i < NFIELDS; i++) free(*(void **)((char *)S + g_fields[i].offset));
  free(S->err_flags);
  free(S);
}

int elmo_num_fields(void) { return NFIELDS; }
const char *elmo_field_name(int i) { return (i >= 0 && i < NFIELDS) ? g_fields[i].name : NULL; }

void *elmo_field_ptr(elmo_state *S, const char *name, int *nlev, int *kind)
{
  if (strcmp(name, "err_flags") == 0) {
    if (nlev) *nlev = 1;
    if (kind) *kind = KIND_I;
    return S->err_flags;
  }
  for (int i = 0; i < NFIELDS; i++) {
    if (strcmp(name, g_fields[i].name) == 0) {
      if (nlev) *nlev = g_fields[i].nlev;
      if (kind) *kind = g_fields[i].kind;
      return *(void **)((char *)S + g_fields[i].offset);
    }
  }
  return NULL;
}

elmo_snicar *elmo_snicar_ptr(elmo_state *S) { return &S->snicar; }
double *elmo_snowage_ptr(elmo_state *S) { return &S->snowage[0][0]; }
elmo_pft_psn *elmo_pft_psn_ptr(elmo_state *S) { return S->pft_psn; }
elmo_pft_alb *elmo_pft_alb_ptr(elmo_state *S) { return S->pft_alb; }
double *elmo_z0mr_ptr(elmo_state *S) { return S->z0mr; }
double *elmo_displar_ptr(elmo_state *S) { return S->displar; }
double *elmo_albsat_ptr(elmo_state *S) { return &S->albsat[0][0]; }
double *elmo_albdry_ptr(elmo_state *S) { return &S->albdry[0][0]; }

void elmo_set_scalars(elmo_state *S, int ltype, int ctype, int vtype, int urbpoi, int lakpoi, double dewmx,
                      int oldfflag, double dayl, double max_dayl)
{
  S->land.ltype = ltype;
  S->land.ctype = ctype;
  S->land.vtype = vtype;
  S->land.urbpoi = urbpoi;
  S->land.lakpoi = lakpoi;
  S->dewmx = dewmx;
  S->oldfflag = oldfflag;
  S->dayl = dayl;
  S->max_dayl = max_dayl;
}

void elmo_set_threads(int n)
{
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int elmo_get_max_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------------------------------------ */
/* wrappers                                                                                         */
/* ------------------------------------------------------------------------------------------------ */
#define LV(f, n) (S->f + (size_t)c * (n))

/* canopy_hydrology_kokkos.cc:98-112 */
void elmo_frac_wet(elmo_state *S)
{
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < S->ncols; c++) {
    elmo_ch_fraction_wet(&S->land, S->frac_veg_nosno[c], S->dewmx, S->elai[c], S->esai[c], S->h2ocan[c], &S->fwet[c],
                         &S->fdry[c]);
  }
}

/* canopy_hydrology_kokkos.cc:7-95 */
void elmo_canopy_hydrology(elmo_state *S, double dt)
{
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < S->ncols; c++) {
    double qflx_irrig = 0.0; /* hardwired (:24) */
    double qflx_candrip = 0.0, qflx_through_snow = 0.0, qflx_through_rain = 0.0, fracsnow = 0.0, fracrain = 0.0;
    elmo_ch_interception(&S->land, S->frac_veg_nosno[c], S->forc_rain[c], S->forc_snow[c], S->dewmx, S->elai[c],
                         S->esai[c], dt, &S->h2ocan[c], &qflx_candrip, &qflx_through_snow, &qflx_through_rain,
                         &fracsnow, &fracrain);
    elmo_ch_ground_flux(&S->land, S->do_capsnow[c], S->frac_veg_nosno[c], S->forc_rain[c], S->forc_snow[c], qflx_irrig,
                        qflx_candrip, qflx_through_snow, qflx_through_rain, fracsnow, fracrain, &S->qflx_snwcp_liq[c],
                        &S->qflx_snwcp_ice[c], &S->qflx_snow_grnd[c], &S->qflx_rain_grnd[c]);
    elmo_ch_snow_init(&S->land, dt, S->do_capsnow[c], S->oldfflag, S->forc_tbot[c], S->t_grnd[c], S->qflx_snow_grnd[c],
                      S->qflx_snow_melt[c], S->n_melt[c], &S->snow_depth[c], &S->h2osno[c], &S->int_snow[c],
                      LV(swe_old, 5), LV(h2osoi_liq, 20), LV(h2osoi_ice, 20), LV(t_soisno, 20), LV(frac_iceold, 20),
                      &S->snl[c], LV(dz, 20), LV(zsoi, 20), LV(zisoi, 21), LV(snw_rds, 5), &S->frac_sno_eff[c],
                      &S->frac_sno[c]);
    elmo_ch_fraction_h2osfc(&S->land, S->micro_sigma[c], S->h2osno[c], &S->h2osfc[c], LV(h2osoi_liq, 20),
                            &S->frac_sno[c], &S->frac_sno_eff[c], &S->frac_h2osfc[c]);
  }
}

/* surface_radiation_kokkos.cc:7-97 */
void elmo_surface_radiation(elmo_state *S)
{
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < S->ncols; c++) {
    double trd[2] = {0.0, 0.0}, tri[2] = {0.0, 0.0};
    elmo_sr_canopy_sunshade_fractions(&S->land, S->nrad[c], S->elai[c], LV(tlai_z, 1), LV(fsun_z, 1),
                                      LV(forc_solad, 2), LV(forc_solai, 2), LV(fabd_sun_z, 1), LV(fabd_sha_z, 1),
                                      LV(fabi_sun_z, 1), LV(fabi_sha_z, 1), LV(parsun_z, 1), LV(parsha_z, 1),
                                      LV(laisun_z, 1), LV(laisha_z, 1), &S->laisun[c], &S->laisha[c]);
    elmo_sr_initialize_flux(&S->land, &S->sabg_soil[c], &S->sabg_snow[c], &S->sabg[c], &S->sabv[c], &S->fsa[c],
                            LV(sabg_lyr, 6));
    elmo_sr_total_absorbed_radiation(&S->land, S->snl[c], LV(ftdd, 2), LV(ftid, 2), LV(ftii, 2), LV(forc_solad, 2),
                                     LV(forc_solai, 2), LV(fabd, 2), LV(fabi, 2), LV(albsod, 2), LV(albsoi, 2),
                                     LV(albsnd, 2), LV(albsni, 2), LV(albgrd, 2), LV(albgri, 2), &S->sabv[c],
                                     &S->fsa[c], &S->sabg[c], &S->sabg_soil[c], &S->sabg_snow[c], trd, tri);
    S->err_flags[c] |= elmo_sr_layer_absorbed_radiation(&S->land, S->snl[c], S->sabg[c], S->sabg_snow[c],
                                                         S->snow_depth[c], LV(flx_absdv, 6), LV(flx_absdn, 6),
                                                         LV(flx_absiv, 6), LV(flx_absin, 6), trd, tri, LV(sabg_lyr, 6));
    elmo_sr_reflected_radiation(&S->land, LV(albd, 2), LV(albi, 2), LV(forc_solad, 2), LV(forc_solai, 2), &S->fsr[c]);
  }
}

/* canopy_temperature_kokkos.cc:6-131 */
void elmo_canopy_temperature(elmo_state *S)
{
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < S->ncols; c++) {
    double qred = 0.0, hr = 0.0, soilalpha = 0.0;
    elmo_ct_old_ground_temp(&S->land, S->t_h2osfc[c], LV(t_soisno, 20), &S->t_h2osfc_bef[c], LV(tssbef, 20));
    elmo_ct_ground_temp(&S->land, S->snl[c], S->frac_sno_eff[c], S->frac_h2osfc[c], S->t_h2osfc[c], LV(t_soisno, 20),
                        &S->t_grnd[c]);
    elmo_ct_calc_soilalpha(&S->land, S->frac_sno[c], S->frac_h2osfc[c], LV(h2osoi_liq, 20), LV(h2osoi_ice, 20),
                           LV(dz, 20), LV(t_soisno, 20), LV(watsat, 15), LV(sucsat, 15), LV(bsw, 15), LV(watdry, 15),
                           LV(watopt, 15), &qred, &hr, &soilalpha);
    elmo_ct_calc_soilbeta(&S->land, S->frac_sno[c], S->frac_h2osfc[c], LV(watsat, 15), LV(watfc, 15),
                          LV(h2osoi_liq, 20), LV(h2osoi_ice, 20), LV(dz, 20), &S->soilbeta[c]);
    elmo_ct_humidities(&S->land, S->snl[c], S->forc_qbot[c], S->forc_pbot[c], S->t_h2osfc[c], S->t_grnd[c],
                       S->frac_sno[c], S->frac_sno_eff[c], S->frac_h2osfc[c], qred, hr, LV(t_soisno, 20),
                       &S->qg_snow[c], &S->qg_soil[c], &S->qg[c], &S->qg_h2osfc[c], &S->dqgdT[c]);
    elmo_ct_ground_properties(&S->land, S->snl[c], S->frac_sno[c], S->forc_thbot[c], S->forc_qbot[c], S->elai[c],
                              S->esai[c], S->htop[c], S->displar, S->z0mr, LV(h2osoi_liq, 20), LV(h2osoi_ice, 20),
                              &S->emg[c], &S->emv[c], &S->htvp[c], &S->z0mg[c], &S->z0hg[c], &S->z0qg[c], &S->z0mv[c],
                              &S->z0hv[c], &S->z0qv[c], &S->thv[c], &S->z0m[c], &S->displa[c]);
    elmo_ct_forcing_height(&S->land, S->veg_active[c], S->frac_veg_nosno[c], S->z0m[c], S->z0mg[c], S->forc_tbot[c],
                           S->displa[c], &S->forc_hgt_u_patch[c], &S->forc_hgt_t_patch[c], &S->forc_hgt_q_patch[c],
                           &S->thm[c]);
    elmo_ct_init_energy_fluxes(&S->land, &S->eflx_sh_tot[c], &S->eflx_lh_tot[c], &S->eflx_sh_veg[c],
                               &S->qflx_evap_tot[c], &S->qflx_evap_veg[c], &S->qflx_tran_veg[c]);
  }
}

/* bareground_fluxes_kokkos.cc:7-123 */
void elmo_bareground_fluxes(elmo_state *S) { elmo_bareground_fluxes_given(S, NULL); }

/* rho_in (optional): forc_rho per column from the fixture, as test/test_BGFlux.cc:226 passes it */
void elmo_bareground_fluxes_given(elmo_state *S, const double *rho_in)
{
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < S->ncols; c++) {
    double zldis = 0.0, displa = 0.0, dth = 0.0, dqh = 0.0, obu = 0.0, ur = 0.0, um = 0.0, temp1 = 0.0, temp2 = 0.0,
           temp12m = 0.0, temp22m = 0.0, ustar = 0.0;
    double forc_rho = rho_in ? rho_in[c] : elmo_derive_forc_rho(S->forc_pbot[c], S->forc_qbot[c], S->forc_tbot[c]);
    elmo_bg_initialize_flux(&S->land, S->frac_veg_nosno[c], S->forc_u[c], S->forc_v[c], S->forc_qbot[c],
                            S->forc_thbot[c], S->forc_hgt_u_patch[c], S->thm[c], S->thv[c], S->t_grnd[c], S->qg[c],
                            S->z0mg[c], &S->dlrad[c], &S->ulrad[c], &zldis, &displa, &dth, &dqh, &obu, &ur, &um);
    elmo_bg_stability_iteration(&S->land, S->frac_veg_nosno[c], S->forc_hgt_t_patch[c], S->forc_hgt_u_patch[c],
                                S->forc_hgt_q_patch[c], S->z0mg[c], zldis, displa, dth, dqh, ur, S->forc_qbot[c],
                                S->forc_thbot[c], S->thv[c], &S->z0hg[c], &S->z0qg[c], &obu, &um, &temp1, &temp2,
                                &temp12m, &temp22m, &ustar);
    elmo_bg_compute_flux(&S->land, S->frac_veg_nosno[c], S->snl[c], forc_rho, S->soilbeta[c], S->dqgdT[c], S->htvp[c],
                         S->t_h2osfc[c], S->qg_snow[c], S->qg_soil[c], S->qg_h2osfc[c], LV(t_soisno, 20),
                         S->forc_pbot[c], dth, dqh, temp1, temp2, temp12m, temp22m, ustar, S->forc_qbot[c], S->thm[c],
                         &S->cgrnds[c], &S->cgrndl[c], &S->cgrnd[c], &S->eflx_sh_grnd[c], &S->eflx_sh_tot[c],
                         &S->eflx_sh_snow[c], &S->eflx_sh_soil[c], &S->eflx_sh_h2osfc[c], &S->qflx_evap_soi[c],
                         &S->qflx_evap_tot[c], &S->qflx_ev_snow[c], &S->qflx_ev_soil[c], &S->qflx_ev_h2osfc[c],
                         &S->t_ref2m[c], &S->q_ref2m[c], &S->rh_ref2m[c]);
  }
}

/* canopy_fluxes_kokkos.cc:6-265.  psn_pft(idx) of the reference is the PFT parameter struct of column
 * idx; here it is looked up in the shared table by vtype[idx]. */
void elmo_canopy_fluxes(elmo_state *S, double dt) { elmo_canopy_fluxes_given(S, dt, NULL, NULL, NULL, NULL); }

/* Same three L2 calls, but - when the arrays are non-NULL - with forc_rho / forc_po2 / forc_pco2 handed in
 * per column the way test/test_CanFlux.cc:421-449 feeds them from the fixture instead of deriving them.
 * niter (optional) receives the leaf-temperature iteration count of each column. */
void elmo_canopy_fluxes_given(elmo_state *S, double dt, const double *rho_in, const double *po2_in,
                              const double *pco2_in, int *niter)
{
#pragma omp parallel for schedule(dynamic, 64)
  for (int64_t c = 0; c < S->ncols; c++) {
    elmo_cf_scratch w;
    memset(&w, 0, sizeof(w));
    const elmo_pft_psn *psn = &S->pft_psn[S->vtype[c]];
    double forc_po2 = po2_in ? po2_in[c] : elmo_derive_forc_po2(S->forc_pbot[c]);
    double forc_pco2 = pco2_in ? pco2_in[c] : elmo_derive_forc_pco2(S->forc_pbot[c]);
    double forc_rho = rho_in ? rho_in[c] : elmo_derive_forc_rho(S->forc_pbot[c], S->forc_qbot[c], S->forc_tbot[c]);
    unsigned err = 0;
    err |= elmo_cf_initialize_flux(&S->land, S->snl[c], S->frac_veg_nosno[c], S->frac_sno[c], S->forc_hgt_u_patch[c],
                                   S->thm[c], S->thv[c], S->max_dayl, S->dayl, S->altmax_indx[c],
                                   S->altmax_lastyear_indx[c], LV(t_soisno, 20), LV(h2osoi_ice, 20), LV(h2osoi_liq, 20),
                                   LV(dz, 20), LV(rootfr, 15), psn->tc_stress, LV(sucsat, 15), LV(watsat, 15),
                                   LV(bsw, 15), psn->smpso, psn->smpsc, S->elai[c], S->esai[c], S->emv[c], S->emg[c],
                                   S->qg[c], S->t_grnd[c], S->forc_tbot[c], S->forc_pbot[c], S->forc_lwrad[c],
                                   S->forc_u[c], S->forc_v[c], S->forc_qbot[c], S->forc_thbot[c], S->z0mg[c],
                                   &S->btran[c], &S->displa[c], &S->z0mv[c], &S->z0hv[c], &S->z0qv[c], LV(rootr, 15),
                                   LV(eff_porosity, 15), &w, &S->t_veg[c]);
    err |= elmo_cf_stability_iteration(
        &S->land, dt, S->snl[c], S->frac_veg_nosno[c], S->frac_sno[c], S->forc_hgt_u_patch[c], S->forc_hgt_t_patch[c],
        S->forc_hgt_q_patch[c], S->fwet[c], S->fdry[c], S->laisun[c], S->laisha[c], forc_rho, S->snow_depth[c],
        S->soilbeta[c], S->frac_h2osfc[c], S->t_h2osfc[c], S->sabv[c], S->h2ocan[c], S->htop[c], LV(t_soisno, 20),
        S->displa[c], S->elai[c], S->esai[c], S->t_grnd[c], S->forc_pbot[c], S->forc_qbot[c], S->forc_thbot[c],
        S->z0mg[c], S->z0mv[c], S->z0hv[c], S->z0qv[c], S->thm[c], S->thv[c], S->qg[c], psn, S->nrad[c], S->t10[c],
        LV(tlai_z, 1), S->vcmaxcintsha[c], S->vcmaxcintsun[c], LV(parsha_z, 1), LV(parsun_z, 1), LV(laisha_z, 1),
        LV(laisun_z, 1), forc_pco2, forc_po2, &S->btran[c], &S->qflx_tran_veg[c], &S->qflx_evap_veg[c],
        &S->eflx_sh_veg[c], &w, &S->t_veg[c], niter ? &niter[c] : NULL);
    elmo_cf_compute_flux(&S->land, dt, S->snl[c], S->frac_veg_nosno[c], S->frac_sno[c], LV(t_soisno, 20),
                         S->frac_h2osfc[c], S->t_h2osfc[c], S->sabv[c], S->qg_snow[c], S->qg_soil[c], S->qg_h2osfc[c],
                         S->dqgdT[c], S->htvp[c], &w, S->t_veg[c], S->t_grnd[c], S->forc_pbot[c], S->qflx_tran_veg[c],
                         S->qflx_evap_veg[c], S->eflx_sh_veg[c], S->forc_qbot[c], forc_rho, S->thm[c], S->emv[c],
                         S->emg[c], S->forc_lwrad[c], &S->h2ocan[c], &S->eflx_sh_grnd[c], &S->eflx_sh_snow[c],
                         &S->eflx_sh_soil[c], &S->eflx_sh_h2osfc[c], &S->qflx_evap_soi[c], &S->qflx_ev_snow[c],
                         &S->qflx_ev_soil[c], &S->qflx_ev_h2osfc[c], &S->dlrad[c], &S->ulrad[c], &S->cgrnds[c],
                         &S->cgrndl[c], &S->cgrnd[c], &S->t_ref2m[c], &S->q_ref2m[c], &S->rh_ref2m[c]);
    S->err_flags[c] |= err;
  }
}

/* albedo_kokkos.cc:10-376 */
void elmo_albedo_snicar(elmo_state *S) { elmo_albedo_snicar_ex(S, NULL, NULL); }

/* fabd_sun_out / fabd_sha_out (optional, [ncols][2]): the wrapper-local fabd_sun / fabd_sha, which the
 * reference computes but never stores in the state; test/test_SurfAlb.cc compares them. */
void elmo_albedo_snicar_ex(elmo_state *S, double *fabd_sun_out, double *fabd_sha_out)
{
#pragma omp parallel for schedule(dynamic, 64)
  for (int64_t c = 0; c < S->ncols; c++) {
    /* wrapper-local, zero-filled Views (:19-38) */
    int snw_rds_lcl[5] = {0};
    double h2osoi_ice_lcl[5] = {0}, h2osoi_liq_lcl[5] = {0}, albout_lcl[5] = {0}, flx_slrd_lcl[5] = {0},
           flx_slri_lcl[5] = {0}, tsai_z[1] = {0}, fabd_sun[2] = {0}, fabd_sha[2] = {0};
    double flx_abs_lcl[6 * 5] = {0}, mss_cnc_aer_in_fdb[5 * 8] = {0}, g_star[5 * 5] = {0}, omega_star[5 * 5] = {0},
           tau_star[5 * 5] = {0}, flx_absd_snw[6 * 2] = {0}, flx_absi_snw[6 * 2] = {0};
    int snl_top = 0, snl_btm = 0, flg_nosnl = 0;
    double mu_not = 0.0;
    unsigned err = 0;
    const elmo_pft_alb *alb_pft = &S->pft_alb[S->vtype[c]];
    const int urbpoi = S->land.urbpoi;

    elmo_sa_init_timestep(urbpoi, S->elai[c], LV(cnc_bcphi, 5), LV(cnc_bcpho, 5), LV(cnc_dst1, 5), LV(cnc_dst2, 5),
                          LV(cnc_dst3, 5), LV(cnc_dst4, 5), &S->vcmaxcintsun[c], &S->vcmaxcintsha[c], LV(albsod, 2),
                          LV(albsoi, 2), LV(albgrd, 2), LV(albgri, 2), LV(albd, 2), LV(albi, 2), LV(fabd, 2), fabd_sun,
                          fabd_sha, LV(fabi, 2), LV(fabi_sun, 2), LV(fabi_sha, 2), LV(ftdd, 2), LV(ftid, 2),
                          LV(ftii, 2), LV(flx_absdv, 6), LV(flx_absdn, 6), LV(flx_absiv, 6), LV(flx_absin, 6),
                          mss_cnc_aer_in_fdb);
    elmo_sa_soil_albedo(&S->land, S->snl[c], S->t_grnd[c], S->coszen[c], LV(h2osoi_vol, 15), S->albsat[S->isoicol[c]],
                        S->albdry[S->isoicol[c]], LV(albsod, 2), LV(albsoi, 2));
    for (int flg_slr_in = 1; flg_slr_in <= 2; flg_slr_in++) {
      double *flx_abs = (flg_slr_in == 1) ? flx_absd_snw : flx_absi_snw;
      double *albout = (flg_slr_in == 1) ? LV(albsnd, 2) : LV(albsni, 2);
      err |= elmo_sn_init_timestep(urbpoi, flg_slr_in, S->coszen[c], S->h2osno[c], S->snl[c], LV(h2osoi_liq, 20),
                                   LV(h2osoi_ice, 20), LV(snw_rds, 5), &snl_top, &snl_btm, flx_abs_lcl, flx_abs,
                                   &flg_nosnl, h2osoi_ice_lcl, h2osoi_liq_lcl, snw_rds_lcl, &mu_not, flx_slrd_lcl,
                                   flx_slri_lcl);
      elmo_sn_snow_aerosol_mie_params(urbpoi, flg_slr_in, snl_top, snl_btm, S->coszen[c], S->h2osno[c], snw_rds_lcl,
                                      h2osoi_ice_lcl, h2osoi_liq_lcl, &S->snicar, mss_cnc_aer_in_fdb, g_star,
                                      omega_star, tau_star);
      err |= elmo_sn_snow_radiative_transfer_solver(urbpoi, flg_slr_in, flg_nosnl, snl_top, snl_btm, S->coszen[c],
                                                    S->h2osno[c], mu_not, flx_slrd_lcl, flx_slri_lcl, LV(albsoi, 2),
                                                    g_star, omega_star, tau_star, albout_lcl, flx_abs_lcl);
      elmo_sn_snow_albedo_radiation_factor(urbpoi, flg_slr_in, snl_top, S->coszen[c], mu_not, S->h2osno[c],
                                           snw_rds_lcl, LV(albsoi, 2), albout_lcl, flx_abs_lcl, albout, flx_abs);
    }
    elmo_sa_ground_albedo(urbpoi, S->coszen[c], S->frac_sno[c], LV(albsod, 2), LV(albsoi, 2), LV(albsnd, 2),
                          LV(albsni, 2), LV(albgrd, 2), LV(albgri, 2));
    elmo_sa_flux_absorption_factor(&S->land, S->coszen[c], S->frac_sno[c], LV(albsod, 2), LV(albsoi, 2), LV(albsnd, 2),
                                   LV(albsni, 2), flx_absd_snw, flx_absi_snw, LV(flx_absdv, 6), LV(flx_absdn, 6),
                                   LV(flx_absiv, 6), LV(flx_absin, 6));
    err |= elmo_sa_canopy_layer_lai(urbpoi, S->elai[c], S->esai[c], S->tlai[c], S->tsai[c], &S->nrad[c], LV(tlai_z, 1),
                                    tsai_z, LV(fsun_z, 1), LV(fabd_sun_z, 1), LV(fabd_sha_z, 1), LV(fabi_sun_z, 1),
                                    LV(fabi_sha_z, 1));
    elmo_sa_two_stream_solver(&S->land, S->nrad[c], S->coszen[c], S->t_veg[c], S->fwet[c], S->elai[c], S->esai[c],
                              LV(tlai_z, 1), tsai_z, LV(albgrd, 2), LV(albgri, 2), alb_pft, &S->vcmaxcintsun[c],
                              &S->vcmaxcintsha[c], LV(albd, 2), LV(ftid, 2), LV(ftdd, 2), LV(fabd, 2), fabd_sun,
                              fabd_sha, LV(albi, 2), LV(ftii, 2), LV(fabi, 2), LV(fabi_sun, 2), LV(fabi_sha, 2),
                              LV(fsun_z, 1), LV(fabd_sun_z, 1), LV(fabd_sha_z, 1), LV(fabi_sun_z, 1),
                              LV(fabi_sha_z, 1));
    if (fabd_sun_out) {
      fabd_sun_out[c * 2] = fabd_sun[0];
      fabd_sun_out[c * 2 + 1] = fabd_sun[1];
    }
    if (fabd_sha_out) {
      fabd_sha_out[c * 2] = fabd_sha[0];
      fabd_sha_out[c * 2 + 1] = fabd_sha[1];
    }
    S->err_flags[c] |= err;
  }
}

/* elm_kokkos_interface.cc:289-307 */
/* soil_temperature_kokkos.cc:6-278.  The wrapper's "dummy ltype" (:77-79) is 1 (istsoil) for every column. */
void elmo_soil_temperature_ex(elmo_state *S, double dt, double *lhs_out, double *rhs_out, double *sol_out,
                              double *cv_out, double *hs_out)
{
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < S->ncols; c++) {
    const int ltype = 1;
    const int snl = S->snl[c];
    const int nlevsno = ELMO_NLEVSNO;
    double thk[20], tk[20], cv[20], fn[20];
    /* soil_thermal_props (:92-104) */
    elmo_st_calc_soil_tk(ltype, LV(h2osoi_liq, 20), LV(h2osoi_ice, 20), LV(t_soisno, 20), LV(dz, 20), LV(watsat, 15),
                         LV(tkmg, 15), LV(tkdry, 15), thk);
    elmo_st_calc_snow_tk(snl, S->frac_sno[c], LV(h2osoi_liq, 20), LV(h2osoi_ice, 20), LV(dz, 20), thk);
    elmo_st_calc_face_tk(snl, thk, LV(zsoi, 20), LV(zisoi, 21), tk);
    elmo_st_calc_soil_heat_capacity(ltype, snl, S->h2osno[c], LV(watsat, 15), LV(h2osoi_ice, 20), LV(h2osoi_liq, 20),
                                    LV(dz, 20), LV(csol, 20), cv);
    elmo_st_calc_snow_heat_capacity(snl, S->frac_sno[c], LV(h2osoi_ice, 20), LV(h2osoi_liq, 20), cv);
    const double tk_h2osfc = elmo_st_calc_h2osfc_tk(S->h2osfc[c], thk, LV(zsoi, 20));
    const double c_h2osfc = elmo_st_calc_h2osfc_heat_capacity(snl, S->h2osfc[c], S->frac_h2osfc[c]);
    const double dz_h2osfc = elmo_st_calc_h2osfc_height(snl, S->h2osfc[c], S->frac_h2osfc[c]);
    /* surface_heat_fluxes (:121-142) */
    const int soitop = nlevsno;
    const int snotop = nlevsno - snl;
    S->sabg_chk[c] = elmo_st_check_absorbed_solar(S->frac_sno_eff[c], S->sabg_snow[c], S->sabg_soil[c]);
    const double hs_soil =
        elmo_st_calc_surface_heat_flux(S->frac_veg_nosno[c], S->dlrad[c], S->emg[c], S->forc_lwrad[c], S->htvp[c],
                                       S->sabg_soil[c], LV(t_soisno, 20)[soitop], S->eflx_sh_soil[c], S->qflx_ev_soil[c]);
    const double hs_h2osfc =
        elmo_st_calc_surface_heat_flux(S->frac_veg_nosno[c], S->dlrad[c], S->emg[c], S->forc_lwrad[c], S->htvp[c],
                                       S->sabg_soil[c], S->t_h2osfc[c], S->eflx_sh_h2osfc[c], S->qflx_ev_h2osfc[c]);
    const double hs_top_snow =
        elmo_st_calc_surface_heat_flux(S->frac_veg_nosno[c], S->dlrad[c], S->emg[c], S->forc_lwrad[c], S->htvp[c],
                                       LV(sabg_lyr, 6)[snotop], LV(t_soisno, 20)[snotop], S->eflx_sh_snow[c],
                                       S->qflx_ev_snow[c]);
    const double dhsdT = elmo_st_calc_dhsdT(S->cgrnd[c], S->emg[c], S->t_grnd[c]);
    /* diffusive_heat_flux (:153-170) */
    elmo_st_calc_diffusive_heat_flux(snl, tk, LV(t_soisno, 20), LV(zsoi, 20), fn);
    elmo_st_calc_heat_flux_matrix_factor(snl, dt, cv, LV(dz, 20), LV(zsoi, 20), LV(zisoi, 21), LV(fact, 20));
    /* set_RHS / set_LHS (:181-185) */
    double rhs[21], lhs[21 * 5];
    elmo_st_set_rhs(dt, snl, hs_top_snow, dhsdT, hs_soil, S->frac_sno_eff[c], LV(t_soisno, 20), LV(fact, 20), fn,
                    LV(sabg_lyr, 6), LV(zsoi, 20), tk_h2osfc, S->t_h2osfc[c], dz_h2osfc, c_h2osfc, hs_h2osfc, rhs);
    elmo_st_set_lhs(dt, snl, dz_h2osfc, c_h2osfc, tk_h2osfc, S->frac_h2osfc[c], S->frac_sno_eff[c], dhsdT, LV(zsoi, 20),
                    LV(fact, 20), tk, lhs);
    if (lhs_out) memcpy(lhs_out + (size_t)c * 105, lhs, sizeof lhs);
    if (rhs_out) memcpy(rhs_out + (size_t)c * 21, rhs, sizeof rhs);
    if (cv_out) memcpy(cv_out + (size_t)c * 20, cv, sizeof cv);
    if (hs_out) {
      hs_out[(size_t)c * 4 + 0] = hs_soil;
      hs_out[(size_t)c * 4 + 1] = hs_h2osfc;
      hs_out[(size_t)c * 4 + 2] = hs_top_snow;
      hs_out[(size_t)c * 4 + 3] = dhsdT;
    }
    /* solve (:215-225): A, B, Z are freshly zero-allocated Views */
    double A[20] = {0}, B[19] = {0}, Z[21] = {0};
    elmo_st_pdma(snl, lhs, A, B, Z, rhs);
    if (sol_out) memcpy(sol_out + (size_t)c * 21, rhs, sizeof rhs);
    /* update_temperature (:232-237) */
    elmo_st_update_temperature(snl, S->frac_h2osfc[c], rhs, &S->t_h2osfc[c], LV(t_soisno, 20));
    /* phase change (:245-266) */
    elmo_st_phase_change_h2osfc(snl, dt, S->frac_sno[c], S->frac_h2osfc[c], dhsdT, c_h2osfc, LV(fact, 20)[nlevsno - 1],
                                &S->t_h2osfc[c], &S->h2osfc[c], &S->xmf_h2osfc[c], &S->qflx_h2osfc_ice[c],
                                &S->eflx_h2osfc_snow[c], &S->h2osno[c], &S->int_snow[c], &S->snow_depth[c],
                                &LV(h2osoi_ice, 20)[nlevsno - 1], &LV(t_soisno, 20)[nlevsno - 1]);
    elmo_st_phase_change_soisno(snl, ltype, dt, dhsdT, S->frac_h2osfc[c], S->frac_sno_eff[c], LV(fact, 20),
                                LV(watsat, 15), LV(sucsat, 15), LV(bsw, 15), LV(dz, 20), &S->h2osno[c],
                                &S->snow_depth[c], &S->xmf[c], &S->qflx_snofrz[c], &S->qflx_snow_melt[c],
                                &S->qflx_snomelt[c], &S->eflx_snomelt[c], LV(imelt, 20), LV(qflx_snofrz_lyr, 5),
                                LV(h2osoi_ice, 20), LV(h2osoi_liq, 20), LV(t_soisno, 20));
    /* update_t_grnd (:275-280) */
    elmo_st_update_t_grnd(snl, S->frac_h2osfc[c], S->frac_sno_eff[c], S->t_h2osfc[c], LV(t_soisno, 20), &S->t_grnd[c]);
  }
}

void elmo_soil_temperature(elmo_state *S, double dt) { elmo_soil_temperature_ex(S, dt, NULL, NULL, NULL, NULL, NULL); }

/* snow_hydrology_kokkos.cc:23-188: snow_water; compute_aerosol_deposition; aerosol_phase_change, transpiration,
 * snow_compaction, combine_layers, divide_layers, prune_snow_layers; update_aerosol_mass_and_concen; snow_aging - five
 * parallel_for launches in the reference, one pass per column here (a column only reads what the same column wrote) */
/* One call of the wrapper per value of `stage` (0..9, in the wrapper's order), over all columns: what
   elmo_snow_hydrology does, cut where the reference's own functions begin and end, so that tests can run the reference's
   function of one stage (oracle/ref_harness_snow.cc) on exactly the inputs the restatement's stage saw.
     0 snow_water  1 aerosol deposition  2 aerosol_phase_change  3 transpiration  4 snow_compaction  5 combine_layers
     6 divide_layers  7 prune_snow_layers  8 aerosol mass and concentration  9 snow_aging */
static void snow_hydrology_stage_col(elmo_state *S, double dt, int64_t c, int stage, uint32_t *err)
{
  switch (stage) {
    case 0:
      elmo_snow_water(S->do_capsnow[c], S->snl[c], dt, S->frac_sno_eff[c], S->h2osno[c], S->qflx_sub_snow[c],
                      S->qflx_evap_grnd[c], S->qflx_dew_snow[c], S->qflx_dew_grnd[c], S->qflx_rain_grnd[c], S->qflx_snomelt[c],
                      &S->qflx_snow_melt[c], &S->qflx_top_soil[c], &S->int_snow[c], &S->frac_sno[c], &S->mflx_neg_snow[c],
                      LV(h2osoi_liq, 20), LV(h2osoi_ice, 20), LV(mss_bcphi, 5), LV(mss_bcpho, 5), LV(mss_dst1, 5),
                      LV(mss_dst2, 5), LV(mss_dst3, 5), LV(mss_dst4, 5), LV(dz, 20), err);
      break;
    case 1: {
      const double aer[11] = {S->aer_bcphi[c],  S->aer_bcpho[c],  S->aer_bcdep[c],  S->aer_dst1_1[c], S->aer_dst1_2[c], S->aer_dst2_1[c],
                              S->aer_dst2_2[c], S->aer_dst3_1[c], S->aer_dst3_2[c], S->aer_dst4_1[c], S->aer_dst4_2[c]};
      elmo_aerosol_deposition(dt, S->snl[c], aer, LV(mss_bcphi, 5), LV(mss_bcpho, 5), LV(mss_dst1, 5), LV(mss_dst2, 5),
                              LV(mss_dst3, 5), LV(mss_dst4, 5));
      break;
    }
    case 2:
      elmo_aerosol_phase_change(S->snl[c], dt, S->qflx_sub_snow[c], LV(h2osoi_liq, 20), LV(h2osoi_ice, 20), LV(mss_bcphi, 5),
                                LV(mss_bcpho, 5));
      break;
    case 3: elmo_transpiration(S->veg_active[c], S->qflx_tran_veg[c], LV(rootr, 15), LV(qflx_rootsoi, 15)); break;
    case 4:
      elmo_snow_compaction(S->snl[c], S->land.ltype, dt, S->int_snow[c], S->n_melt[c], S->frac_sno[c], LV(imelt, 20),
                           LV(swe_old, 5), LV(h2osoi_liq, 20), LV(h2osoi_ice, 20), LV(t_soisno, 20), LV(frac_iceold, 20),
                           LV(dz, 20));
      break;
    case 5:
      elmo_combine_layers(S->land.urbpoi, S->land.ltype, dt, &S->snl[c], &S->h2osno[c], &S->snow_depth[c], &S->frac_sno_eff[c],
                          &S->frac_sno[c], &S->int_snow[c], &S->qflx_sl_top_soil[c], &S->qflx_snow2topsoi[c],
                          &S->mflx_snowlyr_col[c], LV(t_soisno, 20), LV(h2osoi_ice, 20), LV(h2osoi_liq, 20), LV(snw_rds, 5),
                          LV(mss_bcphi, 5), LV(mss_bcpho, 5), LV(mss_dst1, 5), LV(mss_dst2, 5), LV(mss_dst3, 5),
                          LV(mss_dst4, 5), LV(dz, 20), LV(zsoi, 20), LV(zisoi, 21), err);
      break;
    case 6:
      elmo_divide_layers(S->frac_sno[c], &S->snl[c], LV(h2osoi_ice, 20), LV(h2osoi_liq, 20), LV(t_soisno, 20), LV(snw_rds, 5),
                         LV(mss_bcphi, 5), LV(mss_bcpho, 5), LV(mss_dst1, 5), LV(mss_dst2, 5), LV(mss_dst3, 5), LV(mss_dst4, 5),
                         LV(dz, 20), LV(zsoi, 20), LV(zisoi, 21), err);
      break;
    case 7:
      elmo_prune_snow_layers(S->snl[c], LV(h2osoi_ice, 20), LV(h2osoi_liq, 20), LV(t_soisno, 20), LV(dz, 20), LV(zsoi, 20),
                             LV(zisoi, 21));
      break;
    case 8: {
      double *const mss[6] = {LV(mss_bcphi, 5), LV(mss_bcpho, 5), LV(mss_dst1, 5), LV(mss_dst2, 5), LV(mss_dst3, 5), LV(mss_dst4, 5)};
      double *const cnc[6] = {LV(cnc_bcphi, 5), LV(cnc_bcpho, 5), LV(cnc_dst1, 5), LV(cnc_dst2, 5), LV(cnc_dst3, 5), LV(cnc_dst4, 5)};
      elmo_aerosol_mass_and_concen(dt, S->snl[c], S->do_capsnow[c], S->qflx_snwcp_ice[c], LV(h2osoi_ice, 20), LV(h2osoi_liq, 20),
                                   mss, cnc);
      break;
    }
    default:
      elmo_snow_aging(S->do_capsnow[c], S->snl[c], S->frac_sno[c], dt, S->qflx_snwcp_ice[c], S->qflx_snow_grnd[c], S->h2osno[c],
                      LV(dz, 20), LV(h2osoi_liq, 20), LV(h2osoi_ice, 20), LV(t_soisno, 20), LV(qflx_snofrz_lyr, 5),
                      S->snowage[0], S->snowage[1], S->snowage[2], LV(snw_rds, 5), err);
  }
}

void elmo_snow_hydrology_stage(elmo_state *S, double dt, int stage)
{
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < S->ncols; c++) {
    uint32_t err = 0;
    snow_hydrology_stage_col(S, dt, c, stage, &err);
    S->err_flags[c] |= err;
  }
}

void elmo_snow_hydrology(elmo_state *S, double dt)
{
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < S->ncols; c++) {
    uint32_t err = 0;
    for (int stage = 0; stage < ELMO_SNOW_HYDROLOGY_STAGES; stage++) snow_hydrology_stage_col(S, dt, c, stage, &err);
    S->err_flags[c] |= err;
  }
}

/* surface_fluxes_kokkos.cc:5-107 */
void elmo_surface_fluxes(elmo_state *S, double dt)
{
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < S->ncols; c++) {
    const int soitop = ELMO_NLEVSNO;
    const int snotop = soitop - S->snl[c];
    const int urbpoi = S->land.urbpoi;
    elmo_sf_initial_flux_calc(urbpoi, S->snl[c], S->frac_sno_eff[c], S->frac_h2osfc[c], S->t_h2osfc_bef[c],
                              LV(tssbef, 20)[snotop], LV(tssbef, 20)[soitop], S->t_grnd[c], S->cgrnds[c], S->cgrndl[c],
                              &S->eflx_sh_grnd[c], &S->qflx_evap_soi[c], &S->qflx_ev_snow[c], &S->qflx_ev_soil[c],
                              &S->qflx_ev_h2osfc[c]);
    elmo_sf_update_surface_fluxes(urbpoi, S->do_capsnow[c], S->snl[c], dt, S->t_grnd[c], S->htvp[c], S->frac_sno_eff[c],
                                  S->frac_h2osfc[c], S->t_h2osfc_bef[c], S->sabg_soil[c], S->sabg_snow[c], S->dlrad[c],
                                  S->frac_veg_nosno[c], S->emg[c], S->forc_lwrad[c], LV(tssbef, 20)[snotop],
                                  LV(tssbef, 20)[soitop], LV(h2osoi_ice, 20)[snotop], LV(h2osoi_liq, 20)[soitop],
                                  S->eflx_sh_veg[c], S->qflx_evap_veg[c], &S->qflx_evap_soi[c], &S->eflx_sh_grnd[c],
                                  &S->qflx_ev_snow[c], &S->qflx_ev_soil[c], &S->qflx_ev_h2osfc[c], &S->eflx_soil_grnd[c],
                                  &S->eflx_sh_tot[c], &S->qflx_evap_tot[c], &S->eflx_lh_tot[c], &S->qflx_evap_grnd[c],
                                  &S->qflx_sub_snow[c], &S->qflx_dew_snow[c], &S->qflx_dew_grnd[c], &S->qflx_snwcp_liq[c],
                                  &S->qflx_snwcp_ice[c]);
    elmo_sf_lwrad_outgoing(urbpoi, S->snl[c], S->frac_veg_nosno[c], S->forc_lwrad[c], S->frac_sno_eff[c],
                           LV(tssbef, 20)[snotop], LV(tssbef, 20)[soitop], S->frac_h2osfc[c], S->t_h2osfc_bef[c],
                           S->t_grnd[c], S->ulrad[c], S->emg[c], &S->eflx_lwrad_out[c], &S->eflx_lwrad_net[c]);
    S->soil_e_balance[c] = elmo_sf_soil_energy_balance(S->land.ctype, S->snl[c], S->eflx_soil_grnd[c], S->xmf[c],
                                                       S->xmf_h2osfc[c], S->frac_h2osfc[c], S->t_h2osfc[c],
                                                       S->t_h2osfc_bef[c], dt, S->eflx_h2osfc_snow[c], S->frac_sno_eff[c],
                                                       LV(t_soisno, 20), LV(tssbef, 20), LV(fact, 20));
  }
}

/* init_timestep_kokkos.cc:55-75 with ELM::init_timestep (src/physics/init_timestep_impl.hh:7-42) */
void elmo_init_timestep(elmo_state *S)
{
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < S->ncols; c++) {
    S->h2osno_old[c] = S->h2osno[c];
    S->dtbegin_column_h2o[c] =
        elmo_ce_column_water_mass(S->h2ocan[c], S->h2osno[c], S->h2osfc[c], LV(h2osoi_ice, 20), LV(h2osoi_liq, 20));
    const double H2OSNO_MAX = 1000.0; /* elm_constants.h */
    S->do_capsnow[c] = (S->h2osno[c] > H2OSNO_MAX) ? 1 : 0;
    S->frac_veg_nosno[c] = S->veg_active[c] ? S->frac_veg_nosno_alb[c] : 0;
    if (!S->land.lakpoi) {
      for (int i = 0; i < ELMO_NLEVSNO; i++) {
        if (i >= ELMO_NLEVSNO - S->snl[c]) {
          LV(frac_iceold, 20)[i] = LV(h2osoi_ice, 20)[i] / (LV(h2osoi_liq, 20)[i] + LV(h2osoi_ice, 20)[i]);
        }
      }
    }
  }
}

void elmo_set_init_params(elmo_state *S, double organic_max, const double *roota_par, const double *rootb_par)
{
  S->organic_max = organic_max;
  for (int p = 0; p < ELMO_MXPFT; p++) {
    S->roota_par[p] = roota_par[p];
    S->rootb_par[p] = rootb_par[p];
  }
}

/* the "init functions" lambda of ELM::initialize_kokkos_elm (initialize_elm_kokkos.cc:373-428), in its order.  (Its first
 * line, S.psn_pft(idx) = pft_data.get_pft_psn(S.vtype(idx)), is the PFT-table lookup every wrapper here does by vtype.) */
void elmo_initialize_state(elmo_state *S)
{
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < S->ncols; c++) {
    S->topo_slope[c] = elmo_init_topo_slope(S->topo_slope[c]);
    S->n_melt[c] = elmo_init_melt_factor(S->land.ltype, S->topo_std[c]);
    S->micro_sigma[c] = elmo_init_micro_sigma(S->topo_slope[c]);
    elmo_init_snow_layers(S->snow_depth[c], S->land.lakpoi, &S->snl[c], LV(dz, 20), LV(zsoi, 20), LV(zisoi, 21));
    elmo_init_soil_hydraulics(S->organic_max, LV(pct_sand, 15), LV(pct_clay, 15), LV(organic, 15), LV(zsoi, 20), LV(watsat, 15),
                              LV(bsw, 15), LV(sucsat, 15), LV(watdry, 15), LV(watopt, 15), LV(watfc, 15), LV(tkmg, 15),
                              LV(tkdry, 15), LV(csol, 20));
    const int vt = S->vtype[c];
    elmo_init_vegrootfr(vt, S->roota_par[vt], S->rootb_par[vt], LV(zisoi, 21), LV(rootfr, 15));
    elmo_init_soil_temp(&S->land, S->snl[c], LV(t_soisno, 20), &S->t_grnd[c]);
    elmo_init_snow_state(S->land.urbpoi, S->snl[c], &S->h2osno[c], &S->int_snow[c], &S->snow_depth[c], &S->h2osfc[c],
                         &S->h2ocan[c], &S->frac_h2osfc[c], &S->fwet[c], &S->fdry[c], &S->frac_sno[c], LV(snw_rds, 5));
    elmo_init_soilh2o_state(&S->land, S->snl[c], LV(watsat, 15), LV(t_soisno, 20), LV(dz, 20), LV(h2osoi_vol, 15),
                            LV(h2osoi_liq, 20), LV(h2osoi_ice, 20));
  }
}

/* conserved_quantity_kokkos.cc:8-81 */
void elmo_evaluate_conservation(elmo_state *S, double dt, double *diag)
{
  const double hydrology_source_sink = 0.0; /* hardwired (:22) */
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < S->ncols; c++) {
    double *d = diag + (size_t)c * 8;
    d[0] = elmo_ce_column_water_mass(S->h2ocan[c], S->h2osno[c], S->h2osfc[c], LV(h2osoi_ice, 20), LV(h2osoi_liq, 20));
    d[3] = elmo_ce_dh2o_dt(S->dtbegin_column_h2o[c], d[0], dt);
    d[1] = elmo_ce_column_water_balance_error(S->dtbegin_column_h2o[c], d[0], hydrology_source_sink, S->forc_rain[c],
                                              S->forc_snow[c], S->qflx_evap_tot[c], S->qflx_snwcp_ice[c], dt);
    d[2] = elmo_ce_snow_water_balance_error(S->snl[c], S->qflx_dew_snow[c], S->qflx_dew_grnd[c], S->qflx_sub_snow[c],
                                            S->qflx_evap_grnd[c], S->qflx_snow_melt[c], S->qflx_snwcp_ice[c],
                                            S->qflx_snwcp_liq[c], S->qflx_sl_top_soil[c], S->frac_sno_eff[c],
                                            S->qflx_rain_grnd[c], S->qflx_snow_grnd[c], S->qflx_h2osfc_ice[c],
                                            S->h2osno[c], S->h2osno_old[c], dt, S->do_capsnow[c]);
    d[4] = elmo_ce_solar_shortwave_balance_error(S->fsa[c], S->fsr[c], LV(forc_solad, 2), LV(forc_solai, 2));
    d[5] = elmo_ce_solar_longwave_balance_error(S->eflx_lwrad_out[c], S->eflx_lwrad_net[c], S->forc_lwrad[c]);
    d[6] = elmo_ce_surface_energy_balance_error(S->sabv[c], S->sabg_chk[c], S->forc_lwrad[c], S->eflx_lwrad_out[c],
                                                S->eflx_sh_tot[c], S->eflx_lh_tot[c], S->eflx_soil_grnd[c]);
    d[7] = elmo_ce_net_radiation(S->fsa[c], S->eflx_lwrad_net[c]);
  }
}

/* counterparts of the ref_harness.cc probes elmref_soil_thermal / elmref_pdma / elmref_phase_change */
void elmo_soil_thermal(elmo_state *S, double *thk_out, double *tk_out, double *cv_out, double *scal_out)
{
  for (int64_t c = 0; c < S->ncols; c++) {
    const int ltype = 1;
    double *thk = thk_out + (size_t)c * 20, *tk = tk_out + (size_t)c * 20, *cv = cv_out + (size_t)c * 20;
    elmo_st_calc_soil_tk(ltype, LV(h2osoi_liq, 20), LV(h2osoi_ice, 20), LV(t_soisno, 20), LV(dz, 20), LV(watsat, 15),
                         LV(tkmg, 15), LV(tkdry, 15), thk);
    elmo_st_calc_snow_tk(S->snl[c], S->frac_sno[c], LV(h2osoi_liq, 20), LV(h2osoi_ice, 20), LV(dz, 20), thk);
    elmo_st_calc_face_tk(S->snl[c], thk, LV(zsoi, 20), LV(zisoi, 21), tk);
    elmo_st_calc_soil_heat_capacity(ltype, S->snl[c], S->h2osno[c], LV(watsat, 15), LV(h2osoi_ice, 20),
                                    LV(h2osoi_liq, 20), LV(dz, 20), LV(csol, 20), cv);
    elmo_st_calc_snow_heat_capacity(S->snl[c], S->frac_sno[c], LV(h2osoi_ice, 20), LV(h2osoi_liq, 20), cv);
    scal_out[c * 3 + 0] = elmo_st_calc_h2osfc_tk(S->h2osfc[c], thk, LV(zsoi, 20));
    scal_out[c * 3 + 1] = elmo_st_calc_h2osfc_heat_capacity(S->snl[c], S->h2osfc[c], S->frac_h2osfc[c]);
    scal_out[c * 3 + 2] = elmo_st_calc_h2osfc_height(S->snl[c], S->h2osfc[c], S->frac_h2osfc[c]);
  }
}

void elmo_pdma(int64_t n, const int *snl, const double *lhs, double *rhs)
{
  for (int64_t c = 0; c < n; c++) {
    double A[20] = {0}, B[19] = {0}, Z[21] = {0};
    elmo_st_pdma(snl[c], lhs + (size_t)c * 105, A, B, Z, rhs + (size_t)c * 21);
  }
}

void elmo_phase_change(elmo_state *S, double dt, const double *dhsdT, const double *c_h2osfc)
{
  for (int64_t c = 0; c < S->ncols; c++) {
    const int ltype = 1;
    elmo_st_phase_change_h2osfc(S->snl[c], dt, S->frac_sno[c], S->frac_h2osfc[c], dhsdT[c], c_h2osfc[c],
                                LV(fact, 20)[4], &S->t_h2osfc[c], &S->h2osfc[c], &S->xmf_h2osfc[c],
                                &S->qflx_h2osfc_ice[c], &S->eflx_h2osfc_snow[c], &S->h2osno[c], &S->int_snow[c],
                                &S->snow_depth[c], &LV(h2osoi_ice, 20)[4], &LV(t_soisno, 20)[4]);
    elmo_st_phase_change_soisno(S->snl[c], ltype, dt, dhsdT[c], S->frac_h2osfc[c], S->frac_sno_eff[c], LV(fact, 20),
                                LV(watsat, 15), LV(sucsat, 15), LV(bsw, 15), LV(dz, 20), &S->h2osno[c],
                                &S->snow_depth[c], &S->xmf[c], &S->qflx_snofrz[c], &S->qflx_snow_melt[c],
                                &S->qflx_snomelt[c], &S->eflx_snomelt[c], LV(imelt, 20), LV(qflx_snofrz_lyr, 5),
                                LV(h2osoi_ice, 20), LV(h2osoi_liq, 20), LV(t_soisno, 20));
  }
}

void elmo_timestep7(elmo_state *S, double dt)
{
  elmo_frac_wet(S);
  elmo_albedo_snicar(S);
  elmo_canopy_hydrology(S, dt);
  elmo_surface_radiation(S);
  elmo_canopy_temperature(S);
  elmo_bareground_fluxes(S);
  elmo_canopy_fluxes(S, dt);
}

/* photosynthesis() alone, one call per element - the unit stability_iteration calls twice per trip (test infrastructure:
 * the same interface as elmref_photosynthesis of ref_harness_canopy.cc, which runs the reference's own function).
 * in[n][15]: tlai_z, par_z, lai_z, forc_pbot, t_veg, t10, esat_tv, eair, oair, cair, rb, btran, dayl_factor, thm, vcmaxcint;
 * out[n][2]: ci_z (in/out: untouched at night), rs; err[n]: the ELMO_ERR_PSN_* bits the call raised. */
void elmo_photosynthesis_batch(int64_t n, const elmo_pft_psn *table, const int *vtype, const int *nrad, const double *in,
                               double *out, unsigned *err)
{
  for (int64_t i = 0; i < n; i++) {
    const double *x = in + i * 15;
    double tlai_z[1] = {x[0]}, par_z[1] = {x[1]}, lai_z[1] = {x[2]};
    double ci_z[1] = {out[i * 2]};
    double rs = out[i * 2 + 1];
    err[i] = elmo_psn_photosynthesis(&table[vtype[i]], nrad[i], x[3], x[4], x[5], x[6], x[7], x[8], x[9], x[10], x[11], x[12],
                                     x[13], tlai_z, x[14], par_z, lai_z, ci_z, &rs);
    out[i * 2] = ci_z[0];
    out[i * 2 + 1] = rs;
  }
}
