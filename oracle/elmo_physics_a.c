/*
 * elmo_physics_a.c - oracle restatement, part A: qsat, atmosphere-derived scalars, canopy hydrology,
 * surface radiation, canopy temperature, friction velocity, bare-ground fluxes.
 * TEST INFRASTRUCTURE - see elm_oracle.h.  References are paths under /root/reference.
 */
#include "elm_oracle.h"
#include "elmo_const.h"

#include <math.h>

/* ------------------------------------------------------------------------------------------------
 * src/physics/qsat_impl.hh:7-78  ELM::qsat
 * ---------------------------------------------------------------------------------------------- */
void elmo_qsat(double T, double p, double *es, double *esdT, double *qs, double *qsdT)
{
  static const double a[9] = {6.11213476,      0.444007856,     0.143064234e-01, 0.264461437e-03, 0.305903558e-05,
                              0.196237241e-07, 0.892344772e-10, -0.373208410e-12, 0.209339997e-15};
  static const double b[9] = {0.444017302,     0.286064092e-01, 0.794683137e-03,  0.121211669e-04, 0.103354611e-06,
                              0.404125005e-09, -0.788037859e-12, -0.114596802e-13, 0.381294516e-16};
  static const double c[9] = {6.11123516,      0.503109514,     0.188369801e-01, 0.420547422e-03, 0.614396778e-05,
                              0.602780717e-07, 0.387940929e-09, 0.149436277e-11, 0.262655803e-14};
  static const double d[9] = {0.503277922,     0.377289173e-01, 0.126801703e-02, 0.249468427e-04, 0.313703411e-06,
                              0.257180651e-08, 0.133268878e-10, 0.394116744e-13, 0.498070196e-16};
  double td = T - TFRZ;
  if (td > 100.0) td = 100.0;
  if (td < -75.0) td = -75.0;
  double e, edT;
  if (td >= 0.0) {
    e = a[0] + td * (a[1] + td * (a[2] + td * (a[3] + td * (a[4] + td * (a[5] + td * (a[6] + td * (a[7] + td * a[8])))))));
    edT = b[0] + td * (b[1] + td * (b[2] + td * (b[3] + td * (b[4] + td * (b[5] + td * (b[6] + td * (b[7] + td * b[8])))))));
  } else {
    e = c[0] + td * (c[1] + td * (c[2] + td * (c[3] + td * (c[4] + td * (c[5] + td * (c[6] + td * (c[7] + td * c[8])))))));
    edT = d[0] + td * (d[1] + td * (d[2] + td * (d[3] + td * (d[4] + td * (d[5] + td * (d[6] + td * (d[7] + td * d[8])))))));
  }
  e = e * 100.0;
  edT = edT * 100.0;
  double vp = 1.0 / (p - 0.378 * e);
  double vp1 = 0.622 * vp;
  double vp2 = vp1 * vp;
  *es = e;
  *esdT = edT;
  *qs = e * vp1;
  *qsdT = edT * vp2 * p;
}

/* src/physics/atm_physics_impl.hh:246-272 */
double elmo_derive_forc_vp(double forc_qbot, double forc_pbot) { return forc_qbot * forc_pbot / (0.622 + 0.378 * forc_qbot); }
double elmo_derive_forc_rho(double forc_pbot, double forc_qbot, double forc_tbot)
{
  return (forc_pbot - 0.378 * elmo_derive_forc_vp(forc_qbot, forc_pbot)) / (RAIR * forc_tbot);
}
double elmo_derive_forc_po2(double forc_pbot) { return O2_MOLAR_CONST * forc_pbot; }
double elmo_derive_forc_pco2(double forc_pbot) { return CO2_PPMV * 1.0e-6 * forc_pbot; }

/* ------------------------------------------------------------------------------------------------
 * src/physics/canopy_hydrology_impl.hh
 * ---------------------------------------------------------------------------------------------- */

/* :8-67 interception */
void elmo_ch_interception(const elmo_land *L, int frac_veg_nosno, double forc_rain, double forc_snow, double dewmx,
                          double elai, double esai, double dtime, double *h2ocan, double *qflx_candrip,
                          double *qflx_through_snow, double *qflx_through_rain, double *fracsnow, double *fracrain)
{
  if (L->lakpoi) return;
  if (L->ltype == istsoil || L->ltype == istwet || L->urbpoi || L->ltype == istcrop) {
    *qflx_candrip = 0.0;
    *qflx_through_snow = 0.0;
    *qflx_through_rain = 0.0;
    *fracsnow = 0.0;
    *fracrain = 0.0;
    if (L->ctype != icol_sunwall && L->ctype != icol_shadewall) {
      if (frac_veg_nosno == 1 && (forc_rain + forc_snow) > 0.0) {
        *fracsnow = forc_snow / (forc_snow + forc_rain);
        *fracrain = forc_rain / (forc_snow + forc_rain);
        double h2ocanmx = dewmx * (elai + esai);
        double fpi = 0.25 * (1.0 - exp(-0.5 * (elai + esai)));
        *qflx_through_snow = forc_snow * (1.0 - fpi);
        *qflx_through_rain = forc_rain * (1.0 - fpi);
        double qflx_prec_intr = (forc_snow + forc_rain) * fpi;
        *h2ocan = dmax(0.0, (*h2ocan + dtime * qflx_prec_intr));
        *qflx_candrip = 0.0;
        double xrun = (*h2ocan - h2ocanmx) / dtime;
        if (xrun > 0.0) {
          *qflx_candrip = xrun;
          *h2ocan = h2ocanmx;
        }
      }
    }
  } else if (L->ltype == istice || L->ltype == istice_mec) {
    *h2ocan = 0.0;
    *qflx_candrip = 0.0;
    *qflx_through_snow = 0.0;
    *qflx_through_rain = 0.0;
    *fracsnow = 0.0;
    *fracrain = 0.0;
  }
}

/* :83-120 ground_flux */
void elmo_ch_ground_flux(const elmo_land *L, int do_capsnow, int frac_veg_nosno, double forc_rain, double forc_snow,
                         double qflx_irrig, double qflx_candrip, double qflx_through_snow, double qflx_through_rain,
                         double fracsnow, double fracrain, double *qflx_snwcp_liq, double *qflx_snwcp_ice,
                         double *qflx_snow_grnd, double *qflx_rain_grnd)
{
  if (L->lakpoi) return;
  double snow, rain;
  if (L->ctype != icol_sunwall && L->ctype != icol_shadewall) {
    if (frac_veg_nosno == 0) {
      snow = forc_snow;
      rain = forc_rain;
    } else {
      snow = qflx_through_snow + (qflx_candrip * fracsnow);
      rain = qflx_through_rain + (qflx_candrip * fracrain);
    }
  } else {
    snow = 0.0;
    rain = 0.0;
  }
  rain = rain + qflx_irrig;
  if (do_capsnow) {
    *qflx_snwcp_liq = rain;
    *qflx_snwcp_ice = snow;
    *qflx_snow_grnd = 0.0;
    *qflx_rain_grnd = 0.0;
  } else {
    *qflx_snwcp_liq = 0.0;
    *qflx_snwcp_ice = 0.0;
    *qflx_snow_grnd = snow;
    *qflx_rain_grnd = rain;
  }
}

/* :123-143 fraction_wet (note the literal exponent 0.666666666666) */
void elmo_ch_fraction_wet(const elmo_land *L, int frac_veg_nosno, double dewmx, double elai, double esai,
                          double h2ocan, double *fwet, double *fdry)
{
  if (L->lakpoi) return;
  if (frac_veg_nosno == 1) {
    if (h2ocan > 0.0) {
      double vegt = frac_veg_nosno * (elai + esai);
      double dewmxi = 1.0 / dewmx;
      double f = pow(((dewmxi / vegt) * h2ocan), 0.666666666666);
      *fwet = dmin(f, 1.0);
    } else {
      *fwet = 0.0;
    }
    *fdry = (1.0 - *fwet) * elai / (elai + esai);
  } else {
    *fwet = 0.0;
    *fdry = 0.0;
  }
}

/* :146-308 snow_init */
void elmo_ch_snow_init(const elmo_land *L, double dtime, int do_capsnow, int oldfflag, double forc_t, double t_grnd,
                       double qflx_snow_grnd, double qflx_snow_melt, double n_melt, double *snow_depth,
                       double *h2osno, double *int_snow, double *swe_old, double *h2osoi_liq, double *h2osoi_ice,
                       double *t_soisno, double *frac_iceold, int *snl, double *dz, double *z, double *zi,
                       double *snw_rds, double *frac_sno_eff, double *frac_sno)
{
  const double accum_factor = 0.1;
  const int nlevsno = ELMO_NLEVSNO;
  if (L->lakpoi) return;
  double dz_snowf, newsnow, bifall = 0.0, temp_intsnow;
  const double temp_snow_depth = *snow_depth;
  for (int j = 0; j < nlevsno - *snl; j++) swe_old[j] = 0.0;
  for (int j = nlevsno - *snl; j < nlevsno; j++) swe_old[j] = h2osoi_liq[j] + h2osoi_ice[j];

  if (do_capsnow) {
    dz_snowf = 0.0;
    newsnow = qflx_snow_grnd * dtime;
    *frac_sno = 1.0;
    *int_snow = 5.e2;
  } else {
    if (forc_t > TFRZ + 2.0) {
      bifall = 50.0 + 1.7 * pow(17.0, 1.5);
    } else if (forc_t > TFRZ - 15.0) {
      bifall = 50.0 + 1.7 * pow((forc_t - TFRZ + 15.0), 1.5);
    } else {
      bifall = 50.0;
    }
    newsnow = qflx_snow_grnd * dtime;
    *int_snow = dmax(*int_snow, *h2osno);
    const double snowmelt = qflx_snow_melt * dtime;

    if (*h2osno > 0.0) {
      if (snowmelt > 0.0) {
        double smr = dmin(1.0, (*h2osno / *int_snow));
        *frac_sno = 1.0 - pow((acos(dmin(1.0, (2.0 * smr - 1.0))) / ELM_PI), n_melt);
      }
      if (newsnow > 0.0) {
        double fsno_new = 1.0 - (1.0 - tanh(accum_factor * newsnow)) * (1.0 - *frac_sno);
        *frac_sno = fsno_new;
        temp_intsnow =
            (*h2osno + newsnow) / (0.5 * (cos(ELM_PI * pow((1.0 - dmax(*frac_sno, 1.e-6)), (1.0 / n_melt))) + 1.0));
        *int_snow = dmin(1.e8, temp_intsnow);
      }
      /* subgridflag == 1 (elm_constants.h:12) */
      if (!L->urbpoi) {
        if (*frac_sno > 0.0) {
          *snow_depth = *snow_depth + newsnow / (bifall * *frac_sno);
        } else {
          *snow_depth = 0.0;
        }
      } else {
        *snow_depth = *snow_depth + newsnow / bifall;
      }
      if (oldfflag == 1) {
        if (*snow_depth > 0.0) {
          *frac_sno =
              tanh(*snow_depth / (2.5 * ZLND * pow((dmin(800.0, ((*h2osno + newsnow) / *snow_depth / 100.0))), 1.0)));
        }
        if (*h2osno < 1.0) {
          *frac_sno = dmin(*frac_sno, *h2osno);
        }
      }
    } else {
      if (newsnow > 0.0) {
        double z_avg = newsnow / bifall;
        *frac_sno = tanh(accum_factor * newsnow);
        *int_snow = 0.0;
        temp_intsnow =
            (*h2osno + newsnow) / (0.5 * (cos(ELM_PI * pow((1.0 - dmax(*frac_sno, 1.e-6)), (1.0 / n_melt))) + 1.0));
        *int_snow = dmin(1.e8, temp_intsnow);
        if (!L->urbpoi) {
          *snow_depth = z_avg / *frac_sno;
        } else {
          *snow_depth = newsnow / bifall;
        }
        if (oldfflag == 1) {
          if (*snow_depth > 0.0) {
            *frac_sno =
                tanh(*snow_depth / (2.5 * ZLND * pow((dmin(800.0, ((*h2osno + newsnow) / *snow_depth / 100.0))), 1.0)));
          }
        }
      } else {
        *snow_depth = 0.0;
        *frac_sno = 0.0;
      }
    }
    *h2osno = *h2osno + newsnow;
    *int_snow = *int_snow + newsnow;
    dz_snowf = (*snow_depth - temp_snow_depth);
  }
  /* frac_sno_eff (subgridflag == 1) */
  if (L->ltype == istsoil || L->ltype == istcrop) {
    *frac_sno_eff = *frac_sno;
  } else {
    *frac_sno_eff = 1.0;
  }
  if (L->ltype == istwet && t_grnd > TFRZ) {
    *h2osno = 0.0;
    *snow_depth = 0.0;
  }
  int newnode = 0;
  if (*snl == 0 && qflx_snow_grnd > 0.0 && (*frac_sno * *snow_depth) >= 0.01) {
    newnode = 1;
    *snl = 1;
    dz[nlevsno - 1] = *snow_depth;
    z[nlevsno - 1] = -0.5 * dz[nlevsno - 1];
    zi[nlevsno - 1] = -dz[nlevsno - 1];
    t_soisno[nlevsno - 1] = dmin(TFRZ, forc_t);
    h2osoi_ice[nlevsno - 1] = *h2osno;
    h2osoi_liq[nlevsno - 1] = 0.0;
    frac_iceold[nlevsno - 1] = 1.0;
    snw_rds[nlevsno - 1] = SNW_RDS_MIN;
  }
  if (*snl > 0 && newnode == 0) {
    h2osoi_ice[nlevsno - *snl] = h2osoi_ice[nlevsno - *snl] + newsnow;
    dz[nlevsno - *snl] = dz[nlevsno - *snl] + dz_snowf;
  }
}

/* :312-357 fraction_h2osfc */
void elmo_ch_fraction_h2osfc(const elmo_land *L, double micro_sigma, double h2osno, double *h2osfc,
                             double *h2osoi_liq, double *frac_sno, double *frac_sno_eff, double *frac_h2osfc)
{
  const double min_h2osfc = 1.e-8;
  if (L->lakpoi) return;
  if (L->ltype == istsoil || L->ltype == istcrop) {
    if (*h2osfc > min_h2osfc) {
      double d = 0.0;
      double sigma = 1.0e3 * micro_sigma;
      for (int l = 0; l < 10; l++) {
        double fd = 0.5 * d * (1.0 + erf(d / (sigma * sqrt(2.0)))) +
                    sigma / sqrt(2.0 * ELM_PI) * exp(-pow(d, 2) / (2.0 * pow(sigma, 2))) - *h2osfc;
        double dfdd = 0.5 * (1.0 + erf(d / (sigma * sqrt(2.0))));
        d = d - fd / dfdd;
      }
      *frac_h2osfc = 0.5 * (1.0 + erf(d / (sigma * sqrt(2.0))));
    } else {
      *frac_h2osfc = 0.0;
      h2osoi_liq[ELMO_NLEVSNO] = h2osoi_liq[ELMO_NLEVSNO] + *h2osfc;
      *h2osfc = 0.0;
    }
    if (*frac_sno > (1.0 - *frac_h2osfc) && h2osno > 0.0) {
      if (*frac_h2osfc > 0.01) {
        *frac_h2osfc = dmax((1.0 - *frac_sno), 0.01);
        *frac_sno = 1.0 - *frac_h2osfc;
      } else {
        *frac_sno = 1.0 - *frac_h2osfc;
      }
      *frac_sno_eff = *frac_sno;
    }
  } else {
    *frac_h2osfc = 0.0;
  }
}

/* ------------------------------------------------------------------------------------------------
 * src/physics/surface_radiation_impl.hh
 * ---------------------------------------------------------------------------------------------- */

/* :202-238 canopy_sunshade_fractions */
void elmo_sr_canopy_sunshade_fractions(const elmo_land *L, int nrad, double elai, const double *tlai_z,
                                       const double *fsun_z, const double *forc_solad, const double *forc_solai,
                                       const double *fabd_sun_z, const double *fabd_sha_z, const double *fabi_sun_z,
                                       const double *fabi_sha_z, double *parsun_z, double *parsha_z,
                                       double *laisun_z, double *laisha_z, double *laisun, double *laisha)
{
  (void)elai;
  if (L->urbpoi) return;
  const int ipar = 0;
  for (int iv = 0; iv < nrad; iv++) {
    parsun_z[iv] = 0.0;
    parsha_z[iv] = 0.0;
    laisun_z[iv] = 0.0;
    laisha_z[iv] = 0.0;
  }
  *laisun = 0.0;
  *laisha = 0.0;
  for (int iv = 0; iv < nrad; iv++) {
    laisun_z[iv] = tlai_z[iv] * fsun_z[iv];
    laisha_z[iv] = tlai_z[iv] * (1.0 - fsun_z[iv]);
    *laisun += laisun_z[iv];
    *laisha += laisha_z[iv];
  }
  for (int iv = 0; iv < nrad; iv++) {
    parsun_z[iv] = forc_solad[ipar] * fabd_sun_z[iv] + forc_solai[ipar] * fabi_sun_z[iv];
    parsha_z[iv] = forc_solad[ipar] * fabd_sha_z[iv] + forc_solai[ipar] * fabi_sha_z[iv];
  }
}

/* :9-27 initialize_flux */
void elmo_sr_initialize_flux(const elmo_land *L, double *sabg_soil, double *sabg_snow, double *sabg, double *sabv,
                             double *fsa, double *sabg_lyr)
{
  if (L->urbpoi) return;
  *sabg_soil = 0.0;
  *sabg_snow = 0.0;
  *sabg = 0.0;
  *sabv = 0.0;
  *fsa = 0.0;
  for (int j = 0; j < ELMO_NLEVSNO + 1; j++) sabg_lyr[j] = 0.0;
}

/* :30-74 total_absorbed_radiation (the snl==0 reset sits inside the band loop) */
void elmo_sr_total_absorbed_radiation(const elmo_land *L, int snl, const double *ftdd, const double *ftid,
                                      const double *ftii, const double *forc_solad, const double *forc_solai,
                                      const double *fabd, const double *fabi, const double *albsod,
                                      const double *albsoi, const double *albsnd, const double *albsni,
                                      const double *albgrd, const double *albgri, double *sabv, double *fsa,
                                      double *sabg, double *sabg_soil, double *sabg_snow, double *trd, double *tri)
{
  if (L->urbpoi) return;
  for (int ib = 0; ib < ELMO_NUMRAD; ib++) {
    double cad = forc_solad[ib] * fabd[ib];
    double cai = forc_solai[ib] * fabi[ib];
    *sabv += cad + cai;
    *fsa += cad + cai;
    trd[ib] = forc_solad[ib] * ftdd[ib];
    tri[ib] = forc_solad[ib] * ftid[ib] + forc_solai[ib] * ftii[ib];
    double absrad = trd[ib] * (1.0 - albsod[ib]) + tri[ib] * (1.0 - albsoi[ib]);
    *sabg_soil += absrad;
    absrad = trd[ib] * (1.0 - albsnd[ib]) + tri[ib] * (1.0 - albsni[ib]);
    *sabg_snow += absrad;
    absrad = trd[ib] * (1.0 - albgrd[ib]) + tri[ib] * (1.0 - albgri[ib]);
    *sabg += absrad;
    *fsa += absrad;
    if (snl == 0) {
      *sabg_snow = *sabg;
      *sabg_soil = *sabg;
    }
    /* subgridflag() == 1: second reset is compiled out */
  }
}

/* :77-176 layer_absorbed_radiation; returns ELMO_ERR_SURFRAD_LAYER_SUM where the reference asserts */
unsigned elmo_sr_layer_absorbed_radiation(const elmo_land *L, int snl, double sabg, double sabg_snow,
                                          double snow_depth, const double *flx_absdv, const double *flx_absdn,
                                          const double *flx_absiv, const double *flx_absin, const double *trd,
                                          const double *tri, double *sabg_lyr)
{
  const int nlevsno = ELMO_NLEVSNO;
  (void)snow_depth;
  if (L->urbpoi) return 0;
  double err_sum = 0.0;
  double sabg_snl_sum = 0.0;
  if (snl == 0) {
    for (int i = 0; i <= nlevsno; i++) sabg_lyr[i] = 0.0;
    sabg_lyr[nlevsno] = sabg;
    sabg_snl_sum = sabg_lyr[nlevsno];
  } else {
    for (int i = 0; i < nlevsno + 1; i++) {
      sabg_lyr[i] = flx_absdv[i] * trd[0] + flx_absdn[i] * trd[1] + flx_absiv[i] * tri[0] + flx_absin[i] * tri[1];
      if (i >= nlevsno - snl) sabg_snl_sum += sabg_lyr[i];
    }
    if (fabs(sabg_snl_sum - sabg_snow) > 0.00001) {
      if (snl == 0) {
        for (int j = 0; j < nlevsno; j++) sabg_lyr[j] = 0.0;
        sabg_lyr[nlevsno] = sabg;
      } else if (snl == 1) {
        for (int j = 0; j < nlevsno - 1; j++) sabg_lyr[j] = 0.0;
        sabg_lyr[nlevsno - 1] = sabg_snow * 0.6;
        sabg_lyr[nlevsno] = sabg_snow * 0.4;
      } else {
        for (int j = 0; j <= nlevsno; j++) sabg_lyr[j] = 0.0;
        sabg_lyr[nlevsno - snl] = sabg_snow * 0.75;
        sabg_lyr[nlevsno - snl + 1] = sabg_snow * 0.25;
      }
    }
    /* subgridflag() == 1: shallow-snow branch compiled out */
  }
  for (int j = 0; j <= nlevsno; j++) err_sum += sabg_lyr[j];
  return (fabs(err_sum - sabg_snow) > 0.00001) ? ELMO_ERR_SURFRAD_LAYER_SUM : 0u;
}

/* :179-199 reflected_radiation */
void elmo_sr_reflected_radiation(const elmo_land *L, const double *albd, const double *albi,
                                 const double *forc_solad, const double *forc_solai, double *fsr)
{
  if (!L->urbpoi) {
    double rvis = albd[0] * forc_solad[0] + albi[0] * forc_solai[0];
    double rnir = albd[1] * forc_solad[1] + albi[1] * forc_solai[1];
    *fsr = rvis + rnir;
  } else {
    double fsr_vis_d = albd[0] * forc_solad[0];
    double fsr_nir_d = albd[1] * forc_solad[1];
    double fsr_vis_i = albi[0] * forc_solai[0];
    double fsr_nir_i = albi[1] * forc_solai[1];
    *fsr = fsr_vis_d + fsr_nir_d + fsr_vis_i + fsr_nir_i;
  }
}

/* ------------------------------------------------------------------------------------------------
 * src/physics/canopy_temperature_impl.hh (+ surface_resistance_impl.hh:9-46)
 * ---------------------------------------------------------------------------------------------- */

/* :9-29 old_ground_temp */
void elmo_ct_old_ground_temp(const elmo_land *L, double t_h2osfc, const double *t_soisno, double *t_h2osfc_bef,
                             double *tssbef)
{
  if (L->lakpoi) return;
  for (int i = 0; i < ELMO_NLEVTOT; i++) {
    if ((L->ctype == icol_sunwall || L->ctype == icol_shadewall || L->ctype == icol_roof) && i > 5 /*nlevurb*/) {
      tssbef[i] = SPVAL;
    } else {
      tssbef[i] = t_soisno[i];
    }
    *t_h2osfc_bef = t_h2osfc;
  }
}

/* :32-48 ground_temp */
void elmo_ct_ground_temp(const elmo_land *L, int snl, double frac_sno_eff, double frac_h2osfc, double t_h2osfc,
                         const double *t_soisno, double *t_grnd)
{
  const int nlevsno = ELMO_NLEVSNO;
  if (L->lakpoi) return;
  if (snl > 0) {
    *t_grnd = frac_sno_eff * t_soisno[nlevsno - snl] + (1.0 - frac_sno_eff - frac_h2osfc) * t_soisno[nlevsno] +
              frac_h2osfc * t_h2osfc;
  } else {
    *t_grnd = (1.0 - frac_h2osfc) * t_soisno[nlevsno] + frac_h2osfc * t_h2osfc;
  }
}

/* :51-130 calc_soilalpha */
void elmo_ct_calc_soilalpha(const elmo_land *L, double frac_sno, double frac_h2osfc, const double *h2osoi_liq,
                            const double *h2osoi_ice, const double *dz, const double *t_soisno, const double *watsat,
                            const double *sucsat, const double *bsw, const double *watdry, const double *watopt,
                            double *qred, double *hr, double *soilalpha)
{
  const int nlevsno = ELMO_NLEVSNO;
  const double smpmin = -1.e8;
  (void)watdry;
  (void)watopt;
  *qred = 1.0;
  if (L->lakpoi) return;
  if (L->ltype != istwet && L->ltype != istice && L->ltype != istice_mec) {
    if (L->ltype == istsoil || L->ltype == istcrop) {
      double wx = (h2osoi_liq[nlevsno] / DENH2O + h2osoi_ice[nlevsno] / DENICE) / dz[nlevsno];
      double fac = dmin(1.0, wx / watsat[0]);
      fac = dmax(fac, 0.01);
      double psit = -sucsat[0] * pow(fac, (-bsw[0]));
      psit = dmax(smpmin, psit);
      *hr = exp(psit / ROVERG / t_soisno[nlevsno]);
      *qred = (1.0 - frac_sno - frac_h2osfc) * *hr + frac_sno + frac_h2osfc;
      *soilalpha = *qred;
    } else if (L->ctype == icol_sunwall || L->ctype == icol_shadewall) {
      *qred = 0.0;
    } else if (L->ctype == icol_roof || L->ctype == icol_road_imperv) {
      *qred = 1.0;
    }
  } else {
    *soilalpha = SPVAL;
  }
}

/* :133-140 calc_soilbeta -> surface_resistance_impl.hh:9-46 calc_soilevap_stress.
 * The reference compares Land.ltype against icol_* constants (71..75): with ltype in 1..9 those
 * branches are dead and soilbeta is left unchanged for such land units. */
void elmo_ct_calc_soilbeta(const elmo_land *L, double frac_sno, double frac_h2osfc, const double *watsat,
                           const double *watfc, const double *h2osoi_liq, const double *h2osoi_ice, const double *dz,
                           double *soilbeta)
{
  const int nlevsno = ELMO_NLEVSNO;
  if (L->lakpoi) return;
  if (L->ltype != istwet && L->ltype != istice && L->ltype != istice_mec) {
    if (L->ltype == istsoil || L->ltype == istcrop) {
      double wx = (h2osoi_liq[nlevsno] / DENH2O + h2osoi_ice[nlevsno] / DENICE) / dz[nlevsno];
      double fac = dmin(1.0, wx / watsat[0]);
      fac = dmax(fac, 0.01);
      (void)fac;
      if (wx < watfc[0]) {
        double fac_fc = dmin(1.0, wx / watfc[0]);
        fac_fc = dmax(fac_fc, 0.01);
        *soilbeta = (1.0 - frac_sno - frac_h2osfc) * 0.25 * pow(1.0 - cos(ELM_PI * fac_fc), 2.0) + frac_sno + frac_h2osfc;
      } else {
        *soilbeta = 1.0;
      }
    } else if (L->ltype == icol_road_perv) {
      *soilbeta = 0.0;
    } else if (L->ltype == icol_sunwall || L->ltype == icol_shadewall) {
      *soilbeta = 0.0;
    } else if (L->ltype == icol_roof || L->ltype == icol_road_imperv) {
      *soilbeta = 0.0;
    }
  } else {
    *soilbeta = 1.0;
  }
}

/* :143-202 humidities (conditions "qsatg > forc_q && forc_q > qsatg" are always false, kept as written) */
void elmo_ct_humidities(const elmo_land *L, int snl, double forc_q, double forc_pbot, double t_h2osfc, double t_grnd,
                        double frac_sno, double frac_sno_eff, double frac_h2osfc, double qred, double hr,
                        const double *t_soisno, double *qg_snow, double *qg_soil, double *qg, double *qg_h2osfc,
                        double *dqgdT)
{
  const int nlevsno = ELMO_NLEVSNO;
  if (L->lakpoi) return;
  double eg, qsatg, degdT, qsatgdT;
  if (L->ltype == istsoil || L->ltype == istcrop) {
    elmo_qsat(t_soisno[nlevsno - snl], forc_pbot, &eg, &degdT, &qsatg, &qsatgdT);
    if (qsatg > forc_q && forc_q > qsatg) {
      qsatg = forc_q;
      qsatgdT = 0.0;
    }
    *qg_snow = qsatg;
    *dqgdT = frac_sno * qsatgdT;
    elmo_qsat(t_soisno[nlevsno], forc_pbot, &eg, &degdT, &qsatg, &qsatgdT);
    if (qsatg > forc_q && forc_q > hr * qsatg) {
      qsatg = forc_q;
      qsatgdT = 0.0;
    }
    *qg_soil = hr * qsatg;
    *dqgdT = *dqgdT + (1.0 - frac_sno - frac_h2osfc) * hr * qsatgdT;
    if (snl == 0) {
      *qg_snow = *qg_soil;
      *dqgdT = (1.0 - frac_h2osfc) * hr * *dqgdT;
    }
    elmo_qsat(t_h2osfc, forc_pbot, &eg, &degdT, &qsatg, &qsatgdT);
    if (qsatg > forc_q && forc_q > qsatg) {
      qsatg = forc_q;
      qsatgdT = 0.0;
    }
    *qg_h2osfc = qsatg;
    *dqgdT = *dqgdT + frac_h2osfc * qsatgdT;
    *qg = frac_sno_eff * *qg_snow + (1.0 - frac_sno_eff - frac_h2osfc) * *qg_soil + frac_h2osfc * *qg_h2osfc;
  } else {
    elmo_qsat(t_grnd, forc_pbot, &eg, &degdT, &qsatg, &qsatgdT);
    *qg = qred * qsatg;
    *dqgdT = qred * qsatgdT;
    if (qsatg > forc_q && forc_q > qred * qsatg) {
      *qg = forc_q;
      *dqgdT = 0.0;
    }
    *qg_snow = *qg;
    *qg_soil = *qg;
    *qg_h2osfc = *qg;
  }
}

/* :205-257 ground_properties (z0mr/displar indexed by Land.vtype, as in the reference) */
void elmo_ct_ground_properties(const elmo_land *L, int snl, double frac_sno, double forc_th, double forc_q,
                               double elai, double esai, double htop, const double *displar, const double *z0mr,
                               const double *h2osoi_liq, const double *h2osoi_ice, double *emg, double *emv,
                               double *htvp, double *z0mg, double *z0hg, double *z0qg, double *z0mv, double *z0hv,
                               double *z0qv, double *thv, double *z0m, double *displa)
{
  const int nlevsno = ELMO_NLEVSNO;
  if (L->lakpoi) return;
  if (!L->urbpoi) {
    if (L->ltype == istice || L->ltype == istice_mec) {
      *emg = 0.97;
    } else {
      *emg = (1.0 - frac_sno) * 0.96 + frac_sno * 0.97;
    }
  }
  const double avmuir = 1.0;
  *emv = 1.0 - exp(-(elai + esai) / avmuir);
  *htvp = HVAP;
  if (h2osoi_liq[nlevsno - snl] <= 00 && h2osoi_ice[nlevsno - snl] > 0.0) {
    *htvp = HSUB;
  }
  if (frac_sno > 0.0) {
    *z0mg = ZSNO;
  } else {
    *z0mg = ZLND;
  }
  *z0hg = *z0mg;
  *z0qg = *z0mg;
  *z0m = z0mr[L->vtype] * htop;
  *displa = displar[L->vtype] * htop;
  *z0mv = *z0m;
  *z0hv = *z0mv;
  *z0qv = *z0mv;
  *thv = forc_th * (1.0 + 0.61 * forc_q);
}

/* :260-296 forcing_height */
void elmo_ct_forcing_height(const elmo_land *L, int veg_active, int frac_veg_nosno, double z0m, double z0mg,
                            double forc_t, double displa, double *forc_hgt_u_patch, double *forc_hgt_t_patch,
                            double *forc_hgt_q_patch, double *thm)
{
  if (veg_active) {
    if (L->ltype == istsoil || L->ltype == istcrop) {
      if (frac_veg_nosno == 0) {
        *forc_hgt_u_patch += z0mg + displa;
        *forc_hgt_t_patch += z0mg + displa;
        *forc_hgt_q_patch += z0mg + displa;
      } else {
        *forc_hgt_u_patch += z0m + displa;
        *forc_hgt_t_patch += z0m + displa;
        *forc_hgt_q_patch += z0m + displa;
      }
    } else if (L->ltype == istwet || L->ltype == istice || L->ltype == istice_mec) {
      *forc_hgt_u_patch += z0mg;
      *forc_hgt_t_patch += z0mg;
      *forc_hgt_q_patch += z0mg;
    } else if (L->urbpoi) {
      const double z_0_town = 0.0, z_d_town = 0.0;
      *forc_hgt_u_patch += z_0_town + z_d_town;
      *forc_hgt_t_patch += z_0_town + z_d_town;
      *forc_hgt_q_patch += z_0_town + z_d_town;
    }
  }
  *thm = forc_t + 0.0098 * *forc_hgt_t_patch;
}

/* :299-327 init_energy_fluxes */
void elmo_ct_init_energy_fluxes(const elmo_land *L, double *eflx_sh_tot, double *eflx_lh_tot, double *eflx_sh_veg,
                                double *qflx_evap_tot, double *qflx_evap_veg, double *qflx_tran_veg)
{
  (void)L;
  *eflx_sh_tot = 0.0;
  *eflx_lh_tot = 0.0;
  *eflx_sh_veg = 0.0;
  *qflx_evap_tot = 0.0;
  *qflx_evap_veg = 0.0;
  *qflx_tran_veg = 0.0;
}

/* ------------------------------------------------------------------------------------------------
 * src/physics/friction_velocity_impl.hh
 * ---------------------------------------------------------------------------------------------- */

/* :17-24 */
static double stability_func1(double zeta)
{
  const double chik2 = sqrt(1.0 - 16.0 * zeta);
  double chik = sqrt(chik2);
  return 2.0 * log((1.0 + chik) * 0.5) + log((1.0 + chik2) * 0.5) - 2.0 * atan(chik) + ELM_PI * 0.5;
}

/* :27-33 */
static double stability_func2(double zeta)
{
  const double chik2 = sqrt(1.0 - 16.0 * zeta);
  return 2.0 * log((1.0 + chik2) * 0.5);
}

/* :36-61 monin_obukhov_length */
void elmo_fv_monin_obukhov_length(double ur, double thv, double dthv, double zldis, double z0m, double *um,
                                  double *obu)
{
  const double wc = 0.5;
  if (dthv >= 0.0) {
    *um = dmax(ur, 0.1);
  } else {
    *um = sqrt(ur * ur + wc * wc);
  }
  const double rib = GRAV * zldis * dthv / (thv * *um * *um);
  double zeta;
  if (rib >= 0.0) {
    zeta = rib * log(zldis / z0m) / (1.0 - 5.0 * dmin(rib, 0.19));
    zeta = dmin(2.0, dmax(zeta, 0.01));
  } else {
    zeta = rib * log(zldis / z0m);
    zeta = dmax(-100.0, dmin(zeta, -0.01));
  }
  *obu = zldis / zeta;
}

/* :64-83 friction_velocity_wind */
void elmo_fv_wind(double forc_hgt_u_patch, double displa, double um, double obu, double z0m, double *ustar)
{
  const double zetam = 1.574;
  const double zldis = forc_hgt_u_patch - displa;
  const double zeta = zldis / obu;
  if (zeta < (-zetam)) {
    *ustar = VKC * um /
             (log(-zetam * obu / z0m) - stability_func1(-zetam) + stability_func1(z0m / obu) +
              1.14 * (pow((-zeta), 0.333) - pow(zetam, 0.333)));
  } else if (zeta < 0.0) {
    *ustar = VKC * um / (log(zldis / z0m) - stability_func1(zeta) + stability_func1(z0m / obu));
  } else if (zeta <= 1.0) {
    *ustar = VKC * um / (log(zldis / z0m) + 5.0 * zeta - 5.0 * z0m / obu);
  } else {
    *ustar = VKC * um / (log(obu / z0m) + 5.0 - 5.0 * z0m / obu + (5.0 * log(zeta) + zeta - 1.0));
  }
}

/* shared profile expression of :86-172; zetat = 0.465.  Each caller below spells out the argument
 * wiring of its reference function; the arithmetic expression (operand order included) is the same
 * in all four reference functions except for the "5.0 * (z0h / obu)" grouping in temp2m's last branch. */
static double profile_t(double zldis, double obu, double z0, int paren_last)
{
  const double zetat = 0.465;
  const double zeta = zldis / obu;
  if (zeta < -zetat) {
    return VKC / (log(-zetat * obu / z0) - stability_func2(-zetat) + stability_func2(z0 / obu) +
                  0.8 * (pow(zetat, -0.333) - pow((-zeta), -0.333)));
  } else if (zeta < 0.0) {
    return VKC / (log(zldis / z0) - stability_func2(zeta) + stability_func2(z0 / obu));
  } else if (zeta <= 1.0) {
    return VKC / (log(zldis / z0) + 5.0 * zeta - 5.0 * z0 / obu);
  } else {
    if (paren_last) return VKC / (log(obu / z0) + 5.0 - 5.0 * (z0 / obu) + (5.0 * log(zeta) + zeta - 1.0));
    return VKC / (log(obu / z0) + 5.0 - 5.0 * z0 / obu + (5.0 * log(zeta) + zeta - 1.0));
  }
}

/* :86-104 friction_velocity_temp */
void elmo_fv_temp(double forc_hgt_t_patch, double displa, double obu, double z0h, double *temp1)
{
  *temp1 = profile_t(forc_hgt_t_patch - displa, obu, z0h, 0);
}

/* :107-131 friction_velocity_humidity */
void elmo_fv_humidity(double forc_hgt_q_patch, double forc_hgt_t_patch, double displa, double obu, double z0h,
                      double z0q, double temp1, double *temp2)
{
  if (forc_hgt_q_patch == forc_hgt_t_patch && z0q == z0h) {
    *temp2 = temp1;
  } else {
    *temp2 = profile_t(forc_hgt_q_patch - displa, obu, z0q, 0);
  }
}

/* :134-150 friction_velocity_temp2m */
void elmo_fv_temp2m(double obu, double z0h, double *temp12m) { *temp12m = profile_t(2.0 + z0h, obu, z0h, 1); }

/* :153-172 friction_velocity_humidity2m */
void elmo_fv_humidity2m(double obu, double z0h, double z0q, double temp12m, double *temp22m)
{
  if (z0q == z0h) {
    *temp22m = temp12m;
  } else {
    *temp22m = profile_t(2.0 + z0q, obu, z0q, 0);
  }
}

/* ------------------------------------------------------------------------------------------------
 * src/physics/bareground_fluxes_impl.hh
 * ---------------------------------------------------------------------------------------------- */

/* :7-27 initialize_flux */
void elmo_bg_initialize_flux(const elmo_land *L, int frac_veg_nosno, double forc_u, double forc_v, double forc_q,
                             double forc_th, double forc_hgt_u_patch, double thm, double thv, double t_grnd,
                             double qg, double z0mg, double *dlrad, double *ulrad, double *zldis, double *displa,
                             double *dth, double *dqh, double *obu, double *ur, double *um)
{
  if (!L->lakpoi && !L->urbpoi && frac_veg_nosno == 0) {
    *ur = dmax(1.0, sqrt(forc_u * forc_u + forc_v * forc_v));
    *dth = thm - t_grnd;
    *dqh = forc_q - qg;
    *zldis = forc_hgt_u_patch;
    double dthv = *dth * (1.0 + 0.61 * forc_q) + 0.61 * forc_th * *dqh;
    *displa = 0.0;
    *dlrad = 0.0;
    *ulrad = 0.0;
    elmo_fv_monin_obukhov_length(*ur, thv, dthv, *zldis, z0mg, um, obu);
  }
}

/* :30-79 stability_iteration (3 fixed iterations) */
void elmo_bg_stability_iteration(const elmo_land *L, int frac_veg_nosno, double forc_hgt_t_patch,
                                 double forc_hgt_u_patch, double forc_hgt_q_patch, double z0mg, double zldis,
                                 double displa, double dth, double dqh, double ur, double forc_q, double forc_th,
                                 double thv, double *z0hg, double *z0qg, double *obu, double *um, double *temp1,
                                 double *temp2, double *temp12m, double *temp22m, double *ustar)
{
  const int niters = 3;
  const double beta = 1.0;
  const double zii = 1000.0;
  for (int i = 0; i < niters; i++) {
    if (!L->lakpoi && !L->urbpoi && frac_veg_nosno == 0) {
      elmo_fv_wind(forc_hgt_u_patch, displa, *um, *obu, z0mg, ustar);
      elmo_fv_temp(forc_hgt_t_patch, displa, *obu, *z0hg, temp1);
      elmo_fv_humidity(forc_hgt_q_patch, forc_hgt_t_patch, displa, *obu, *z0hg, *z0qg, *temp1, temp2);
      elmo_fv_temp2m(*obu, *z0hg, temp12m);
      elmo_fv_humidity2m(*obu, *z0hg, *z0qg, *temp12m, temp22m);

      double tstar = *temp1 * dth;
      double qstar = *temp2 * dqh;
      double thvstar = tstar * (1.0 + 0.61 * forc_q) + 0.61 * forc_th * qstar;
      *z0hg = z0mg / exp(0.13 * pow((*ustar * z0mg / 1.5e-5), 0.45));
      *z0qg = *z0hg;
      double zeta = zldis * VKC * GRAV * thvstar / (pow(*ustar, 2.0) * thv);
      if (zeta >= 0.0) {
        zeta = dmin(2.0, dmax(zeta, 0.01));
        *um = dmax(ur, 0.1);
      } else {
        zeta = dmax(-100.0, dmin(zeta, -0.01));
        double wc = beta * pow((-GRAV * *ustar * thvstar * zii / thv), 0.333);
        *um = sqrt(ur * ur + wc * wc);
      }
      *obu = zldis / zeta;
    }
  }
}

/* :82-161 compute_flux */
void elmo_bg_compute_flux(const elmo_land *L, int frac_veg_nosno, int snl, double forc_rho, double soilbeta,
                          double dqgdT, double htvp, double t_h2osfc, double qg_snow, double qg_soil,
                          double qg_h2osfc, const double *t_soisno, double forc_pbot, double dth, double dqh,
                          double temp1, double temp2, double temp12m, double temp22m, double ustar, double forc_q,
                          double thm, double *cgrnds, double *cgrndl, double *cgrnd, double *eflx_sh_grnd,
                          double *eflx_sh_tot, double *eflx_sh_snow, double *eflx_sh_soil, double *eflx_sh_h2osfc,
                          double *qflx_evap_soi, double *qflx_evap_tot, double *qflx_ev_snow, double *qflx_ev_soil,
                          double *qflx_ev_h2osfc, double *t_ref2m, double *q_ref2m, double *rh_ref2m)
{
  const int nlevsno = ELMO_NLEVSNO;
  if (!L->lakpoi) {
    *cgrnd = 0.0;
    *cgrnds = 0.0;
    *cgrndl = 0.0;
  }
  if (!L->lakpoi && !L->urbpoi && frac_veg_nosno == 0) {
    double rah, raw, raih, raiw;
    double e_ref2m, de2mdT, qsat_ref2m, dqsat2mdT;
    rah = 1.0 / (temp1 * ustar);
    raw = 1.0 / (temp2 * ustar);
    raih = forc_rho * CPAIR / rah;
    if (dqh > 0.0) {
      raiw = forc_rho / raw;
    } else {
      raiw = soilbeta * forc_rho / raw;
    }
    *cgrnds = raih;
    *cgrndl = raiw * dqgdT;
    *cgrnd = *cgrnds + htvp * *cgrndl;
    *eflx_sh_grnd = -raih * dth;
    *eflx_sh_tot = *eflx_sh_grnd;
    *eflx_sh_snow = -raih * (thm - t_soisno[nlevsno - snl]);
    *eflx_sh_soil = -raih * (thm - t_soisno[nlevsno]);
    *eflx_sh_h2osfc = -raih * (thm - t_h2osfc);
    *qflx_evap_soi = -raiw * dqh;
    *qflx_evap_tot = *qflx_evap_soi;
    *qflx_ev_snow = -raiw * (forc_q - qg_snow);
    *qflx_ev_soil = -raiw * (forc_q - qg_soil);
    *qflx_ev_h2osfc = -raiw * (forc_q - qg_h2osfc);
    *t_ref2m = thm + temp1 * dth * (1.0 / temp12m - 1.0 / temp1);
    *q_ref2m = forc_q + temp2 * dqh * (1.0 / temp22m - 1.0 / temp2);
    elmo_qsat(*t_ref2m, forc_pbot, &e_ref2m, &de2mdT, &qsat_ref2m, &dqsat2mdT);
    *rh_ref2m = dmin(100.0, (*q_ref2m / qsat_ref2m * 100.0));
  }
}
