/*
 * elmo_physics_c.c - oracle restatement, part C: surface albedo (two-stream canopy) and SNICAR-AD snow optics.
 * TEST INFRASTRUCTURE - see elm_oracle.h.  References are paths under /root/reference.
 */
#include "elm_oracle.h"
#include "elmo_const.h"

#include <math.h>

/* src/physics/surface_albedo.h:56-57 */
#define SA_MPE 1.e-06
#define SA_EXTKN 0.30

/* src/physics/snow_snicar.h:27-35 */
#define SN_MIN_SNW 1.0e-30
#define SN_RDS_MAX_TBL 1500
#define SN_RDS_MIN_TBL 30

/* ------------------------------------------------------------------------------------------------
 * src/physics/surface_albedo_impl.hh
 * ---------------------------------------------------------------------------------------------- */

/* :68-86 vegsol / novegsol */
static int sa_vegsol(const elmo_land *L, double coszen, double elai, double esai)
{
  return (!L->urbpoi && coszen > 0.0 && (L->ltype == istsoil || L->ltype == istcrop) && (elai + esai) > 0.0);
}
static int sa_novegsol(const elmo_land *L, double coszen, double elai, double esai)
{
  if (!L->urbpoi && coszen > 0.0) {
    if (!((L->ltype == istsoil || L->ltype == istcrop) && (elai + esai) > 0.0)) return 1;
  }
  return 0;
}

/* :90-151 init_timestep (nlevcan == 1) */
void elmo_sa_init_timestep(int urbpoi, double elai, const double *mss_cnc_bcphi, const double *mss_cnc_bcpho,
                           const double *mss_cnc_dst1, const double *mss_cnc_dst2, const double *mss_cnc_dst3,
                           const double *mss_cnc_dst4, double *vcmaxcintsun, double *vcmaxcintsha, double *albsod,
                           double *albsoi, double *albgrd, double *albgri, double *albd, double *albi, double *fabd,
                           double *fabd_sun, double *fabd_sha, double *fabi, double *fabi_sun, double *fabi_sha,
                           double *ftdd, double *ftid, double *ftii, double *flx_absdv, double *flx_absdn,
                           double *flx_absiv, double *flx_absin, double *mss_cnc_aer_in_fdb)
{
  if (!urbpoi) {
    for (int ib = 0; ib < ELMO_NUMRAD; ++ib) {
      albsod[ib] = 0.0;
      albsoi[ib] = 0.0;
      albgrd[ib] = 0.0;
      albgri[ib] = 0.0;
      albd[ib] = 1.0;
      albi[ib] = 1.0;
      fabd[ib] = 0.0;
      fabd_sun[ib] = 0.0;
      fabd_sha[ib] = 0.0;
      fabi[ib] = 0.0;
      fabi_sun[ib] = 0.0;
      fabi_sha[ib] = 0.0;
      ftdd[ib] = 0.0;
      ftid[ib] = 0.0;
      ftii[ib] = 0.0;
    }
    for (int i = 0; i <= ELMO_NLEVSNO; ++i) {
      flx_absdv[i] = 0.0;
      flx_absdn[i] = 0.0;
      flx_absiv[i] = 0.0;
      flx_absin[i] = 0.0;
    }
    *vcmaxcintsun = 0.0;
    *vcmaxcintsha = (1.0 - exp(-SA_EXTKN * elai)) / SA_EXTKN;
    if (elai > 0.0) {
      *vcmaxcintsha /= elai;
    } else {
      *vcmaxcintsha = 0.0;
    }
  }
  for (int i = 0; i < ELMO_NLEVSNO; ++i) {
    double *row = mss_cnc_aer_in_fdb + i * ELMO_SNO_NBR_AER;
    row[0] = mss_cnc_bcphi[i];
    row[1] = mss_cnc_bcpho[i];
    row[2] = 0.0;
    row[3] = 0.0;
    row[4] = mss_cnc_dst1[i];
    row[5] = mss_cnc_dst2[i];
    row[6] = mss_cnc_dst3[i];
    row[7] = mss_cnc_dst4[i];
  }
}

/* :690-754 soil_albedo */
void elmo_sa_soil_albedo(const elmo_land *L, int snl, double t_grnd, double coszen, const double *h2osoi_vol,
                         const double *albsat, const double *albdry, double *albsod, double *albsoi)
{
  static const double albice[2] = {0.8, 0.55};
  static const double alblak[2] = {0.60, 0.40};
  static const double alblakwi[2] = {0.10, 0.10};
  const double calb = 95.6;
  const int lakepuddling = 0;
  if (L->urbpoi) return;
  if (coszen > 0.0) {
    for (int ib = 0; ib < ELMO_NUMRAD; ib++) {
      if (L->ltype == istsoil || L->ltype == istcrop) {
        double inc = dmax(0.11 - 0.40 * h2osoi_vol[0], 0.0);
        albsod[ib] = dmin(albsat[ib] + inc, albdry[ib]);
        albsoi[ib] = albsod[ib];
      } else if (L->ltype == istice || L->ltype == istice_mec) {
        albsod[ib] = albice[ib];
        albsoi[ib] = albsod[ib];
      } else {
        if (L->ltype == istdlak && !lakepuddling && snl == 0) {
          double sicefr = 1.0 - exp(-calb * (TFRZ - t_grnd) / TFRZ);
          albsod[ib] = sicefr * alblak[ib] + (1.0 - sicefr) * dmax(alblakwi[ib], 0.05 / (dmax(0.001, coszen) + 0.15));
          albsoi[ib] = sicefr * alblak[ib] + (1.0 - sicefr) * dmax(alblakwi[ib], 0.10);
        } else {
          albsod[ib] = alblak[ib];
          albsoi[ib] = albsod[ib];
        }
      }
    }
  }
}

/* :155-167 ground_albedo */
void elmo_sa_ground_albedo(int urbpoi, double coszen, double frac_sno, const double *albsod, const double *albsoi,
                           const double *albsnd, const double *albsni, double *albgrd, double *albgri)
{
  if (!urbpoi && coszen > 0.0) {
    for (int ib = 0; ib < ELMO_NUMRAD; ++ib) {
      albgrd[ib] = albsod[ib] * (1.0 - frac_sno) + albsnd[ib] * frac_sno;
      albgri[ib] = albsoi[ib] * (1.0 - frac_sno) + albsni[ib] * frac_sno;
    }
  }
}

/* :171-211 flux_absorption_factor (subgridflag == 1) */
void elmo_sa_flux_absorption_factor(const elmo_land *L, double coszen, double frac_sno, const double *albsod,
                                    const double *albsoi, const double *albsnd, const double *albsni,
                                    const double *flx_absd_snw, const double *flx_absi_snw, double *flx_absdv,
                                    double *flx_absdn, double *flx_absiv, double *flx_absin)
{
  if (!L->urbpoi && coszen > 0.0) {
    for (int i = 0; i <= ELMO_NLEVSNO; ++i) {
      for (int ib = 0; ib < ELMO_NUMRAD; ++ib) {
        const double fd = flx_absd_snw[i * 2 + ib];
        const double fi = flx_absi_snw[i * 2 + ib];
        if (L->ltype == istdlak) {
          if (ib == 0) {
            flx_absdv[i] = fd * frac_sno + ((1.0 - frac_sno) * (1.0 - albsod[ib]) * (fd / (1.0 - albsnd[ib])));
            flx_absiv[i] = fi * frac_sno + ((1.0 - frac_sno) * (1.0 - albsoi[ib]) * (fi / (1.0 - albsni[ib])));
          } else if (ib == 1) {
            flx_absdn[i] = fd * frac_sno + ((1.0 - frac_sno) * (1.0 - albsod[ib]) * (fd / (1.0 - albsnd[ib])));
            flx_absin[i] = fi * frac_sno + ((1.0 - frac_sno) * (1.0 - albsoi[ib]) * (fi / (1.0 - albsni[ib])));
          }
        } else {
          if (ib == 0) {
            flx_absdv[i] = fd * (1.0 - albsnd[ib]);
            flx_absiv[i] = fi * (1.0 - albsni[ib]);
          } else if (ib == 1) {
            flx_absdn[i] = fd * (1.0 - albsnd[ib]);
            flx_absin[i] = fi * (1.0 - albsni[ib]);
          }
        }
      }
    }
  }
}

/* :215-319 canopy_layer_lai (nlevcan == 1: single big-leaf layer) */
unsigned elmo_sa_canopy_layer_lai(int urbpoi, double elai, double esai, double tlai, double tsai, int *nrad,
                                  double *tlai_z, double *tsai_z, double *fsun_z, double *fabd_sun_z,
                                  double *fabd_sha_z, double *fabi_sun_z, double *fabi_sha_z)
{
  unsigned err = 0;
  (void)tlai;
  (void)tsai;
  if (!urbpoi) {
    *nrad = 1;
    tlai_z[0] = elai;
    tsai_z[0] = esai;
    double laisum = 0.0;
    double saisum = 0.0;
    for (int iv = 0; iv < *nrad; ++iv) {
      laisum += tlai_z[iv];
      saisum += tsai_z[iv];
    }
    if (fabs(laisum - elai) > SA_MPE || fabs(saisum - esai) > SA_MPE) {
      err |= ELMO_ERR_ALB_CANOPY_LAYERS;
    }
    for (int iv = 0; iv < *nrad; ++iv) {
      fabd_sun_z[iv] = 0.0;
      fabd_sha_z[iv] = 0.0;
      fabi_sun_z[iv] = 0.0;
      fabi_sha_z[iv] = 0.0;
      fsun_z[iv] = 0.0;
    }
  }
  return err;
}

/* :323-687 two_stream_solver (nlevcan == 1 branch) */
void elmo_sa_two_stream_solver(const elmo_land *L, int nrad, double coszen, double t_veg, double fwet, double elai,
                               double esai, const double *tlai_z, const double *tsai_z, const double *albgrd,
                               const double *albgri, const elmo_pft_alb *alb_pft, double *vcmaxcintsun,
                               double *vcmaxcintsha, double *albd, double *ftid, double *ftdd, double *fabd,
                               double *fabd_sun, double *fabd_sha, double *albi, double *ftii, double *fabi,
                               double *fabi_sun, double *fabi_sha, double *fsun_z, double *fabd_sun_z,
                               double *fabd_sha_z, double *fabi_sun_z, double *fabi_sha_z)
{
  static const double omegas[2] = {0.8, 0.4};
  const double betads = 0.5;
  const double betais = 0.5;
  (void)nrad;
  (void)tlai_z;
  (void)tsai_z;
  if (sa_vegsol(L, coszen, elai, esai)) {
    double omega[2], rho[2], tau[2];
    const double wl = elai / dmax(elai + esai, SA_MPE);
    const double ws = esai / dmax(elai + esai, SA_MPE);
    const double cosz = dmax(0.001, coszen);
    double chil = dmin(dmax(alb_pft->xl, -0.4), 0.6);
    if (fabs(chil) <= 0.01) {
      chil = 0.01;
    }
    const double phi1 = 0.5 - 0.633 * chil - 0.330 * chil * chil;
    const double phi2 = 0.877 * (1.0 - 2.0 * phi1);
    const double gdir = phi1 + phi2 * cosz;
    const double twostext = gdir / cosz;
    const double avmu = (1.0 - phi1 / phi2 * log((phi1 + phi2) / phi1)) / phi2;
    const double temp0 = gdir + phi2 * cosz;
    const double temp1 = phi1 * cosz;
    const double temp2 = (1.0 - temp1 / temp0 * log((temp1 + temp0) / temp1));

    for (int ib = 0; ib < ELMO_NUMRAD; ib++) {
      rho[ib] = dmax(alb_pft->rhol[ib] * wl + alb_pft->rhos[ib] * ws, SA_MPE);
      tau[ib] = dmax(alb_pft->taul[ib] * wl + alb_pft->taus[ib] * ws, SA_MPE);
      const double omegal = rho[ib] + tau[ib];
      const double asu = 0.5 * omegal * gdir / temp0 * temp2;
      const double betadl = (1.0 + avmu * twostext) / (omegal * avmu * twostext) * asu;
      const double betail = 0.5 * ((rho[ib] + tau[ib]) + (rho[ib] - tau[ib]) * pow(((1.0 + chil) / 2.0), 2.0)) / omegal;
      double tmp0, tmp1, tmp2;
      if (t_veg > TFRZ) {
        tmp0 = omegal;
        tmp1 = betadl;
        tmp2 = betail;
      } else {
        tmp0 = (1.0 - fwet) * omegal + fwet * omegas[ib];
        tmp1 = ((1.0 - fwet) * omegal * betadl + fwet * omegas[ib] * betads) / tmp0;
        tmp2 = ((1.0 - fwet) * omegal * betail + fwet * omegas[ib] * betais) / tmp0;
      }
      omega[ib] = tmp0;
      const double betad = tmp1;
      const double betai = tmp2;

      const double b = 1.0 - omega[ib] + omega[ib] * betai;
      const double c1 = omega[ib] * betai;
      tmp0 = avmu * twostext;
      const double d = tmp0 * omega[ib] * betad;
      const double f = tmp0 * omega[ib] * (1.0 - betad);
      tmp1 = b * b - c1 * c1;
      const double h = sqrt(tmp1) / avmu;
      const double sigma = tmp0 * tmp0 - tmp1;
      const double p1 = b + avmu * h;
      const double p2 = b - avmu * h;
      const double p3 = b + tmp0;
      const double p4 = b - tmp0;

      double t1 = dmin(h * (elai + esai), 40.0);
      double s1 = exp(-t1);
      t1 = dmin(twostext * (elai + esai), 40.0);
      double s2 = exp(-t1);

      /* direct beam */
      double u1 = b - c1 / albgrd[ib];
      double u2 = b - c1 * albgrd[ib];
      double u3 = f + c1 * albgrd[ib];
      tmp2 = u1 - avmu * h;
      double tmp3 = u1 + avmu * h;
      double d1 = p1 * tmp2 / s1 - p2 * tmp3 * s1;
      double tmp4 = u2 + avmu * h;
      double tmp5 = u2 - avmu * h;
      double d2 = tmp4 / s1 - tmp5 * s1;
      double h1 = -d * p4 - c1 * f;
      double tmp6 = d - h1 * p3 / sigma;
      double tmp7 = (d - c1 - h1 / sigma * (u1 + tmp0)) * s2;
      double h2 = (tmp6 * tmp2 / s1 - p2 * tmp7) / d1;
      double h3 = -(tmp6 * tmp3 * s1 - p1 * tmp7) / d1;
      double h4 = -f * p3 - c1 * d;
      double tmp8 = h4 / sigma;
      double tmp9 = (u3 - tmp8 * (u2 - tmp0)) * s2;
      double h5 = -(tmp8 * tmp4 / s1 + tmp9) / d2;
      double h6 = (tmp8 * tmp5 * s1 + tmp9) / d2;

      albd[ib] = h1 / sigma + h2 + h3;
      ftid[ib] = h4 * s2 / sigma + h5 * s1 + h6 / s1;
      ftdd[ib] = s2;
      fabd[ib] = 1.0 - albd[ib] - (1.0 - albgrd[ib]) * ftdd[ib] - (1.0 - albgri[ib]) * ftid[ib];

      double a1 = h1 / sigma * (1.0 - s2 * s2) / (2.0 * twostext) + h2 * (1.0 - s2 * s1) / (twostext + h) +
                  h3 * (1.0 - s2 / s1) / (twostext - h);
      double a2 = h4 / sigma * (1.0 - s2 * s2) / (2.0 * twostext) + h5 * (1.0 - s2 * s1) / (twostext + h) +
                  h6 * (1.0 - s2 / s1) / (twostext - h);
      fabd_sun[ib] = (1.0 - omega[ib]) * (1.0 - s2 + 1.0 / avmu * (a1 + a2));
      fabd_sha[ib] = fabd[ib] - fabd_sun[ib];

      /* diffuse */
      u1 = b - c1 / albgri[ib];
      u2 = b - c1 * albgri[ib];
      tmp2 = u1 - avmu * h;
      tmp3 = u1 + avmu * h;
      d1 = p1 * tmp2 / s1 - p2 * tmp3 * s1;
      tmp4 = u2 + avmu * h;
      tmp5 = u2 - avmu * h;
      d2 = tmp4 / s1 - tmp5 * s1;
      double h7 = (c1 * tmp2) / (d1 * s1);
      double h8 = (-c1 * tmp3 * s1) / d1;
      double h9 = tmp4 / (d2 * s1);
      double h10 = (-tmp5 * s1) / d2;

      albi[ib] = h7 + h8;
      ftii[ib] = h9 * s1 + h10 / s1;
      fabi[ib] = 1.0 - albi[ib] - (1.0 - albgri[ib]) * ftii[ib];

      a1 = h7 * (1.0 - s2 * s1) / (twostext + h) + h8 * (1.0 - s2 / s1) / (twostext - h);
      a2 = h9 * (1.0 - s2 * s1) / (twostext + h) + h10 * (1.0 - s2 / s1) / (twostext - h);
      fabi_sun[ib] = (1.0 - omega[ib]) / avmu * (a1 + a2);
      fabi_sha[ib] = fabi[ib] - fabi_sun[ib];

      if (ib == 0) {
        /* nlevcan == 1: sun/shade big leaf */
        fsun_z[0] = (1.0 - s2) / t1;
        double laisum = elai + esai;
        fabd_sun_z[0] = fabd_sun[ib] / (fsun_z[0] * laisum);
        fabi_sun_z[0] = fabi_sun[ib] / (fsun_z[0] * laisum);
        fabd_sha_z[0] = fabd_sha[ib] / ((1.0 - fsun_z[0]) * laisum);
        fabi_sha_z[0] = fabi_sha[ib] / ((1.0 - fsun_z[0]) * laisum);
        double extkb = twostext;
        *vcmaxcintsun = (1.0 - exp(-(SA_EXTKN + extkb) * elai)) / (SA_EXTKN + extkb);
        *vcmaxcintsha = (1.0 - exp(-SA_EXTKN * elai)) / SA_EXTKN - *vcmaxcintsun;
        if (elai > 0.0) {
          *vcmaxcintsun = *vcmaxcintsun / (fsun_z[0] * elai);
          *vcmaxcintsha = *vcmaxcintsha / ((1.0 - fsun_z[0]) * elai);
        } else {
          *vcmaxcintsun = 0.0;
          *vcmaxcintsha = 0.0;
        }
      }
    }
  } else if (sa_novegsol(L, coszen, elai, esai)) {
    for (int ib = 0; ib < ELMO_NUMRAD; ++ib) {
      fabd[ib] = 0.0;
      fabd_sun[ib] = 0.0;
      fabd_sha[ib] = 0.0;
      fabi[ib] = 0.0;
      fabi_sun[ib] = 0.0;
      fabi_sha[ib] = 0.0;
      ftdd[ib] = 1.0;
      ftid[ib] = 0.0;
      ftii[ib] = 1.0;
      albd[ib] = albgrd[ib];
      albi[ib] = albgri[ib];
    }
  }
}

/* ------------------------------------------------------------------------------------------------
 * src/physics/snow_snicar_impl.hh
 * ---------------------------------------------------------------------------------------------- */

/* :9-103 init_timestep.  The reference's zeroing loops advance `i` instead of `ib` (:22-31), so only
 * flx_abs(0..1,0) and flx_abs_lcl(0..4,0) are cleared; restated literally. */
unsigned elmo_sn_init_timestep(int urbpoi, int flg_slr_in, double coszen, double h2osno, int snl,
                               const double *h2osoi_liq, const double *h2osoi_ice, const double *snw_rds,
                               int *snl_top, int *snl_btm, double *flx_abs_lcl, double *flx_abs, int *flg_nosnl,
                               double *h2osoi_ice_lcl, double *h2osoi_liq_lcl, int *snw_rds_lcl, double *mu_not,
                               double *flx_slrd_lcl, double *flx_slri_lcl)
{
  const int nlevsno = ELMO_NLEVSNO;
  unsigned err = 0;
  if (urbpoi) return 0;
  for (int i = 0; i <= nlevsno; ++i) {
    for (int ib = 0; i < ELMO_NUMRAD; ++i) {
      flx_abs[i * ELMO_NUMRAD + ib] = 0.0;
    }
  }
  for (int i = 0; i <= nlevsno; ++i) {
    for (int ib = 0; i < ELMO_NUMRAD_SNW; ++i) {
      flx_abs_lcl[i * ELMO_NUMRAD_SNW + ib] = 0.0;
    }
  }
  if ((coszen > 0.0) && (h2osno > SN_MIN_SNW)) {
    int snl_lcl;
    if (snl == 0) {
      *flg_nosnl = 1;
      snl_lcl = 1;
      h2osoi_ice_lcl[nlevsno - 1] = h2osno;
      h2osoi_liq_lcl[nlevsno - 1] = 0.0;
      snw_rds_lcl[nlevsno - 1] = (int)round(SNW_RDS_MIN);
    } else {
      *flg_nosnl = 0;
      snl_lcl = snl;
      for (int i = 0; i < nlevsno; ++i) {
        h2osoi_liq_lcl[i] = h2osoi_liq[i];
        h2osoi_ice_lcl[i] = h2osoi_ice[i];
        snw_rds_lcl[i] = (int)round(snw_rds[i]);
      }
    }
    *snl_btm = nlevsno - 1;
    *snl_top = nlevsno - snl_lcl;
    for (int i = *snl_top; i <= *snl_btm; ++i) {
      if ((snw_rds_lcl[i] < SN_RDS_MIN_TBL) || (snw_rds_lcl[i] > SN_RDS_MAX_TBL)) {
        err |= ELMO_ERR_SNICAR_RDS;
        /* the reference throws; keep the table index in range so a flagged column cannot read out of bounds */
        snw_rds_lcl[i] = snw_rds_lcl[i] < SN_RDS_MIN_TBL ? SN_RDS_MIN_TBL : SN_RDS_MAX_TBL;
      }
    }
    *mu_not = dmax(coszen, 0.01);
    if (flg_slr_in == 1) {
      for (int b = 0; b < ELMO_NUMRAD_SNW; ++b) {
        flx_slrd_lcl[b] = 1.0 / (*mu_not * ELM_PI);
        flx_slri_lcl[b] = 0.0;
      }
    } else if (flg_slr_in == 2) {
      for (int b = 0; b < ELMO_NUMRAD_SNW; ++b) {
        flx_slrd_lcl[b] = 0.0;
        flx_slri_lcl[b] = 1.0;
      }
    } else {
      err |= ELMO_ERR_SNICAR_FLAG;
    }
  }
  return err;
}

/* :107-310 snow_aerosol_mie_params */
void elmo_sn_snow_aerosol_mie_params(int urbpoi, int flg_slr_in, int snl_top, int snl_btm, double coszen,
                                     double h2osno, const int *snw_rds_lcl, const double *h2osoi_ice_lcl,
                                     const double *h2osoi_liq_lcl, const elmo_snicar *T,
                                     const double *mss_cnc_aer_in, double *g_star, double *omega_star,
                                     double *tau_star)
{
  const int nlevsno = ELMO_NLEVSNO;
  const int naer = ELMO_SNO_NBR_AER;
  const double rds_bcint_lcl = 100.0;
  const double rds_bcext_lcl = 100.0;
  if (urbpoi) return;
  if (!((coszen > 0.0) && (h2osno > SN_MIN_SNW))) return;

  double mss_cnc_aer_lcl[ELMO_NLEVSNO][ELMO_SNO_NBR_AER];
  for (int i = 0; i < nlevsno; ++i)
    for (int j = 0; j < naer; ++j) mss_cnc_aer_lcl[i][j] = mss_cnc_aer_in[i * naer + j];

  for (int bnd_idx = 0; bnd_idx < ELMO_NUMRAD_SNW; ++bnd_idx) {
    if ((bnd_idx == 4) || (bnd_idx == 3)) {
      for (int i = 0; i < nlevsno; ++i)
        for (int j = 0; j < naer; ++j) mss_cnc_aer_lcl[i][j] = 0.0;
    }
    double ss_alb_snw_lcl[ELMO_NLEVSNO], asm_prm_snw_lcl[ELMO_NLEVSNO], ext_cff_mss_snw_lcl[ELMO_NLEVSNO];
    if (flg_slr_in == 1) {
      for (int i = snl_top; i <= snl_btm; ++i) {
        const int rds_idx = snw_rds_lcl[i] - SN_RDS_MIN_TBL;
        ss_alb_snw_lcl[i] = T->ss_alb_snw_drc[bnd_idx * ELMO_MIE_N + rds_idx];
        asm_prm_snw_lcl[i] = T->asm_prm_snw_drc[bnd_idx * ELMO_MIE_N + rds_idx];
        ext_cff_mss_snw_lcl[i] = T->ext_cff_mss_snw_drc[bnd_idx * ELMO_MIE_N + rds_idx];
      }
    } else if (flg_slr_in == 2) {
      for (int i = snl_top; i <= snl_btm; ++i) {
        const int rds_idx = snw_rds_lcl[i] - SN_RDS_MIN_TBL;
        ss_alb_snw_lcl[i] = T->ss_alb_snw_dfs[bnd_idx * ELMO_MIE_N + rds_idx];
        asm_prm_snw_lcl[i] = T->asm_prm_snw_dfs[bnd_idx * ELMO_MIE_N + rds_idx];
        ext_cff_mss_snw_lcl[i] = T->ext_cff_mss_snw_dfs[bnd_idx * ELMO_MIE_N + rds_idx];
      }
    }
    double ss_alb_aer_lcl[ELMO_SNO_NBR_AER], asm_prm_aer_lcl[ELMO_SNO_NBR_AER], ext_cff_mss_aer_lcl[ELMO_SNO_NBR_AER];
    ss_alb_aer_lcl[2] = T->ss_alb_oc1[bnd_idx];
    asm_prm_aer_lcl[2] = T->asm_prm_oc1[bnd_idx];
    ext_cff_mss_aer_lcl[2] = T->ext_cff_mss_oc1[bnd_idx];
    ss_alb_aer_lcl[3] = T->ss_alb_oc2[bnd_idx];
    asm_prm_aer_lcl[3] = T->asm_prm_oc2[bnd_idx];
    ext_cff_mss_aer_lcl[3] = T->ext_cff_mss_oc2[bnd_idx];
    ss_alb_aer_lcl[4] = T->ss_alb_dst1[bnd_idx];
    asm_prm_aer_lcl[4] = T->asm_prm_dst1[bnd_idx];
    ext_cff_mss_aer_lcl[4] = T->ext_cff_mss_dst1[bnd_idx];
    ss_alb_aer_lcl[5] = T->ss_alb_dst2[bnd_idx];
    asm_prm_aer_lcl[5] = T->asm_prm_dst2[bnd_idx];
    ext_cff_mss_aer_lcl[5] = T->ext_cff_mss_dst2[bnd_idx];
    ss_alb_aer_lcl[6] = T->ss_alb_dst3[bnd_idx];
    asm_prm_aer_lcl[6] = T->asm_prm_dst3[bnd_idx];
    ext_cff_mss_aer_lcl[6] = T->ext_cff_mss_dst3[bnd_idx];
    ss_alb_aer_lcl[7] = T->ss_alb_dst4[bnd_idx];
    asm_prm_aer_lcl[7] = T->asm_prm_dst4[bnd_idx];
    ext_cff_mss_aer_lcl[7] = T->ext_cff_mss_dst4[bnd_idx];

    double tau[ELMO_NLEVSNO], omega[ELMO_NLEVSNO], g[ELMO_NLEVSNO];
    for (int i = snl_top; i <= snl_btm; ++i) {
      int idx_bcint_icerds;
      if (snw_rds_lcl[i] < 125) {
        double tmp1 = snw_rds_lcl[i] / 50; /* integer division, as in the reference (:250) */
        idx_bcint_icerds = (int)round(tmp1) - 1;
      } else if (snw_rds_lcl[i] < 175) {
        idx_bcint_icerds = 1;
      } else {
        double tmp1 = (snw_rds_lcl[i] / 250) + 2; /* integer division (:255) */
        idx_bcint_icerds = (int)round(tmp1) - 1;
      }
      int idx_bcint_nclrds = (int)round(rds_bcint_lcl / 50) - 1;
      int idx_bcext_nclrds = (int)round(rds_bcext_lcl / 50) - 1;
      if (idx_bcint_icerds < 0) idx_bcint_icerds = 0;
      if (idx_bcint_icerds > 7) idx_bcint_icerds = 7;
      if (idx_bcint_nclrds < 0) idx_bcint_nclrds = 0;
      if (idx_bcint_nclrds > 9) idx_bcint_nclrds = 9;
      if (idx_bcext_nclrds < 0) idx_bcext_nclrds = 0;
      if (idx_bcext_nclrds > 9) idx_bcext_nclrds = 9;

      double enh_fct = T->bcenh[(idx_bcint_icerds * 10 + idx_bcint_nclrds) * 5 + bnd_idx];
      ss_alb_aer_lcl[0] = T->ss_alb_bc1[idx_bcint_nclrds * 5 + bnd_idx];
      asm_prm_aer_lcl[0] = T->asm_prm_bc1[idx_bcint_nclrds * 5 + bnd_idx];
      ext_cff_mss_aer_lcl[0] = T->ext_cff_mss_bc1[idx_bcint_nclrds * 5 + bnd_idx] * enh_fct;
      ss_alb_aer_lcl[1] = T->ss_alb_bc2[idx_bcext_nclrds * 5 + bnd_idx];
      asm_prm_aer_lcl[1] = T->asm_prm_bc2[idx_bcext_nclrds * 5 + bnd_idx];
      ext_cff_mss_aer_lcl[1] = T->ext_cff_mss_bc2[idx_bcext_nclrds * 5 + bnd_idx];

      double L_snw = h2osoi_ice_lcl[i] + h2osoi_liq_lcl[i];
      double tau_snw = L_snw * ext_cff_mss_snw_lcl[i];
      double tau_aer[ELMO_SNO_NBR_AER];
      for (int j = 0; j < naer; ++j) {
        double L_aer = L_snw * mss_cnc_aer_lcl[i][j];
        tau_aer[j] = L_aer * ext_cff_mss_aer_lcl[j];
      }
      double tau_sum = 0.0, omega_sum = 0.0, g_sum = 0.0;
      for (int j = 0; j < naer; ++j) {
        tau_sum += tau_aer[j];
        omega_sum += (tau_aer[j] * ss_alb_aer_lcl[j]);
        g_sum += (tau_aer[j] * ss_alb_aer_lcl[j] * asm_prm_aer_lcl[j]);
      }
      tau[i] = tau_sum + tau_snw;
      omega[i] = (1.0 / tau[i]) * (omega_sum + (ss_alb_snw_lcl[i] * tau_snw));
      g[i] = (1.0 / (tau[i] * omega[i])) * (g_sum + (asm_prm_snw_lcl[i] * ss_alb_snw_lcl[i] * tau_snw));
    }
    /* DELTA == 1 */
    for (int i = snl_top; i <= snl_btm; ++i) {
      g_star[bnd_idx * nlevsno + i] = g[i] / (1.0 + g[i]);
      omega_star[bnd_idx * nlevsno + i] = ((1.0 - pow(g[i], 2.0)) * omega[i]) / (1.0 - (omega[i] * pow(g[i], 2.0)));
      tau_star[bnd_idx * nlevsno + i] = (1.0 - (omega[i] * pow(g[i], 2.0))) * tau[i];
    }
  }
}

/* :313-670 snow_radiative_transfer_solver */
unsigned elmo_sn_snow_radiative_transfer_solver(int urbpoi, int flg_slr_in, int flg_nosnl, int snl_top, int snl_btm,
                                                double coszen, double h2osno, double mu_not,
                                                const double *flx_slrd_lcl, const double *flx_slri_lcl,
                                                const double *albsoi, const double *g_star,
                                                const double *omega_star, const double *tau_star, double *albout_lcl,
                                                double *flx_abs_lcl)
{
  static const double difgauspt[8] = {0.9894009, 0.9445750, 0.8656312, 0.7554044,
                                      0.6178762, 0.4580168, 0.2816036, 0.0950125};
  static const double difgauswt[8] = {0.0271525, 0.0622535, 0.0951585, 0.1246290,
                                      0.1495960, 0.1691565, 0.1826034, 0.1894506};
  const int nlevsno = ELMO_NLEVSNO;
  const int ngmax = 8;
  const double puny = 1.0e-11;
  const double argmax = 10.0;
  const double exp_min = exp(-argmax);
  const double c0 = 0.0, c1 = 1.0, c3 = 3.0, c4 = 4.0, cp5 = 0.5, cp75 = 0.75, c1p5 = 1.5, trmin = 0.001;
  unsigned err = 0;
  if (urbpoi) return 0;
  if (!((coszen > 0.0) && (h2osno > SN_MIN_SNW))) return 0;

  double trndir[ELMO_NLEVSNO + 1], trntdr[ELMO_NLEVSNO + 1], trndif[ELMO_NLEVSNO + 1], rupdir[ELMO_NLEVSNO + 1],
      rupdif[ELMO_NLEVSNO + 1], rdndif[ELMO_NLEVSNO + 1], dfdir[ELMO_NLEVSNO + 1], dfdif[ELMO_NLEVSNO + 1],
      dftmp[ELMO_NLEVSNO + 1];
  double rdir[ELMO_NLEVSNO], rdif_a[ELMO_NLEVSNO], rdif_b[ELMO_NLEVSNO], tdir[ELMO_NLEVSNO], tdif_a[ELMO_NLEVSNO],
      tdif_b[ELMO_NLEVSNO], trnlay[ELMO_NLEVSNO], F_abs[ELMO_NLEVSNO];
  const int snl_btm_itf = nlevsno;

  for (int bnd_idx = 0; bnd_idx < ELMO_NUMRAD_SNW; ++bnd_idx) {
    for (int i = snl_top; i <= snl_btm_itf; ++i) {
      trndir[i] = c0;
      trntdr[i] = c0;
      trndif[i] = c0;
      rupdir[i] = c0;
      rupdif[i] = c0;
      rdndif[i] = c0;
    }
    trndir[snl_top] = c1;
    trntdr[snl_top] = c1;
    trndif[snl_top] = c1;
    rdndif[snl_top] = c0;

    for (int i = snl_top; i <= snl_btm; ++i) {
      rdir[i] = c0;
      rdif_a[i] = c0;
      rdif_b[i] = c0;
      tdir[i] = c0;
      tdif_a[i] = c0;
      tdif_b[i] = c0;
      trnlay[i] = c0;
      if (trntdr[i] > trmin) {
        const double ts = tau_star[bnd_idx * nlevsno + i];
        const double ws = omega_star[bnd_idx * nlevsno + i];
        const double gs = g_star[bnd_idx * nlevsno + i];
        const double lm = sqrt(c3 * (c1 - ws) * (c1 - ws * gs));
        const double ue = c1p5 * (c1 - ws * gs) / lm;
        const double extins = dmax(exp_min, exp(-lm * ts));
        const double ne = ((ue + c1) * (ue + c1) / extins) - ((ue - c1) * (ue - c1) * extins);
        rdif_a[i] = (pow(ue, 2.0) - c1) * (c1 / extins - extins) / ne;
        tdif_a[i] = c4 * ue / ne;
        trnlay[i] = dmax(exp_min, exp(-ts / mu_not));
        double alp = cp75 * ws * mu_not * ((c1 + gs * (c1 - ws)) / (c1 - lm * lm * mu_not * mu_not));
        double gam = cp5 * ws * ((c1 + c3 * gs * (c1 - ws) * mu_not * mu_not) / (c1 - lm * lm * mu_not * mu_not));
        double apg = alp + gam;
        double amg = alp - gam;
        rdir[i] = apg * rdif_a[i] + amg * (tdif_a[i] * trnlay[i] - c1);
        tdir[i] = apg * tdif_a[i] + (amg * rdif_a[i] - apg + c1) * trnlay[i];
        const double R1 = rdif_a[i];
        const double T1 = tdif_a[i];
        double swt = c0, smr = c0, smt = c0;
        for (int ng = 0; ng < ngmax; ++ng) {
          const double mu = difgauspt[ng];
          const double gwt = difgauswt[ng];
          swt = swt + mu * gwt;
          const double trn = dmax(exp_min, exp(-ts / mu));
          alp = cp75 * ws * mu * ((c1 + gs * (c1 - ws)) / (c1 - lm * lm * mu * mu));
          gam = cp5 * ws * ((c1 + c3 * gs * (c1 - ws) * mu * mu) / (c1 - lm * lm * mu * mu));
          apg = alp + gam;
          amg = alp - gam;
          const double rdr = apg * R1 + amg * T1 * trn - amg;
          const double tdr = apg * T1 + amg * R1 * trn - apg * trn + trn;
          smr = smr + mu * rdr * gwt;
          smt = smt + mu * tdr * gwt;
        }
        rdif_a[i] = smr / swt;
        tdif_a[i] = smt / swt;
        rdif_b[i] = rdif_a[i];
        tdif_b[i] = tdif_a[i];
      }
      trndir[i + 1] = trndir[i] * trnlay[i];
      const double refkm1 = c1 / (c1 - rdndif[i] * rdif_a[i]);
      const double tdrrdir = trndir[i] * rdir[i];
      const double tdndif = trntdr[i] - trndir[i];
      trntdr[i + 1] = trndir[i] * tdir[i] + (tdndif + tdrrdir * rdndif[i]) * refkm1 * tdif_a[i];
      rdndif[i + 1] = rdif_b[i] + (tdif_b[i] * rdndif[i] * refkm1 * tdif_a[i]);
      trndif[i + 1] = trndif[i] * refkm1 * tdif_a[i];
    }

    rupdir[snl_btm_itf] = albsoi[1];
    rupdif[snl_btm_itf] = albsoi[1];
    if (bnd_idx == 0) {
      rupdir[snl_btm_itf] = albsoi[0];
      rupdif[snl_btm_itf] = albsoi[0];
    }
    for (int i = snl_btm; i >= snl_top; --i) {
      const double refkp1 = c1 / (c1 - rdif_b[i] * rupdif[i + 1]);
      rupdir[i] = rdir[i] + (trnlay[i] * rupdir[i + 1] + (tdir[i] - trnlay[i]) * rupdif[i + 1]) * refkp1 * tdif_b[i];
      rupdif[i] = rdif_a[i] + tdif_a[i] * rupdif[i + 1] * refkp1 * tdif_b[i];
    }

    double refk;
    for (int i = snl_top; i <= snl_btm_itf; ++i) {
      refk = c1 / (c1 - rdndif[i] * rupdif[i]);
      dfdir[i] = trndir[i] + (trntdr[i] - trndir[i]) * (c1 - rupdif[i]) * refk -
                 trndir[i] * rupdir[i] * (c1 - rdndif[i]) * refk;
      if (dfdir[i] < puny) dfdir[i] = c0;
      dfdif[i] = trndif[i] * (c1 - rupdif[i]) * refk;
      if (dfdif[i] < puny) dfdif[i] = c0;
    }

    double albedo, F_sfc_pls;
    if (flg_slr_in == 1) {
      albedo = rupdir[snl_top];
      for (int i = snl_top; i <= snl_btm_itf; ++i) dftmp[i] = dfdir[i];
      refk = c1 / (c1 - rdndif[snl_top] * rupdif[snl_top]);
      F_sfc_pls = (trndir[snl_top] * rupdir[snl_top] + (trntdr[snl_top] - trndir[snl_top]) * rupdif[snl_top]) * refk;
    } else {
      albedo = rupdif[snl_top];
      for (int i = snl_top; i <= snl_btm_itf; ++i) dftmp[i] = dfdif[i];
      refk = c1 / (c1 - rdndif[snl_top] * rupdif[snl_top]);
      F_sfc_pls = trndif[snl_top] * rupdif[snl_top] * refk;
    }

    for (int i = snl_top; i <= snl_btm; ++i) {
      F_abs[i] = dftmp[i] - dftmp[i + 1];
      flx_abs_lcl[i * ELMO_NUMRAD_SNW + bnd_idx] = F_abs[i];
      if (flx_abs_lcl[i * ELMO_NUMRAD_SNW + bnd_idx] < -0.00001) err |= ELMO_ERR_SNICAR_NEG_ABS;
    }
    const double F_btm_net = dftmp[snl_btm_itf];
    flx_abs_lcl[nlevsno * ELMO_NUMRAD_SNW + bnd_idx] = F_btm_net;
    if (flg_nosnl == 1) {
      flx_abs_lcl[(nlevsno - 1) * ELMO_NUMRAD_SNW + bnd_idx] = F_abs[nlevsno - 1];
      flx_abs_lcl[nlevsno * ELMO_NUMRAD_SNW + bnd_idx] = F_btm_net;
    }
    for (int i = snl_top; i <= nlevsno; ++i) {
      if (flx_abs_lcl[i * ELMO_NUMRAD_SNW + bnd_idx] < 0.0) flx_abs_lcl[i * ELMO_NUMRAD_SNW + bnd_idx] = 0.0;
    }
    double F_abs_sum = 0.0;
    for (int i = snl_top; i <= snl_btm; ++i) F_abs_sum = F_abs_sum + F_abs[i];
    const double energy_sum =
        (mu_not * ELM_PI * flx_slrd_lcl[bnd_idx]) + flx_slri_lcl[bnd_idx] - (F_abs_sum + F_btm_net + F_sfc_pls);
    if (fabs(energy_sum) > 0.00001) err |= ELMO_ERR_SNICAR_ENERGY;
    albout_lcl[bnd_idx] = albedo;
    if (albout_lcl[bnd_idx] > 1.0) err |= ELMO_ERR_SNICAR_ALBEDO;
  }
  return err;
}

/* :673-771 snow_albedo_radiation_factor */
void elmo_sn_snow_albedo_radiation_factor(int urbpoi, int flg_slr_in, int snl_top, double coszen, double mu_not,
                                          double h2osno, const int *snw_rds_lcl, const double *albsoi,
                                          const double *albout_lcl, const double *flx_abs_lcl, double *albout,
                                          double *flx_abs)
{
  const int nlevsno = ELMO_NLEVSNO;
  const double sza_a0 = 0.085730, sza_a1 = -0.630883, sza_a2 = 1.303723;
  const double sza_b0 = 1.467291, sza_b1 = -3.338043, sza_b2 = 6.807489;
  const int nir_bnd_bgn = 1, nir_bnd_end = 4;
  const double mu_75 = 0.2588;
  if (urbpoi) return;
  if ((coszen > 0.0) && (h2osno > SN_MIN_SNW)) {
    double flx_wgt[ELMO_NUMRAD_SNW] = {0.0, 0.0, 0.0, 0.0, 0.0};
    if (flg_slr_in == 1) {
      flx_wgt[0] = 1.0;
      flx_wgt[1] = 0.49352158521175;
      flx_wgt[2] = 0.18099494230665;
      flx_wgt[3] = 0.12094898498813;
      flx_wgt[4] = 0.20453448749347;
    } else if (flg_slr_in == 2) {
      flx_wgt[0] = 1.0;
      flx_wgt[1] = 0.58581507618433;
      flx_wgt[2] = 0.20156903770812;
      flx_wgt[3] = 0.10917889346386;
      flx_wgt[4] = 0.10343699264369;
    }
    albout[0] = albout_lcl[0];
    double flx_sum = 0.0;
    double flx_wgt_sum = 0.0;
    for (int b = nir_bnd_bgn; b <= nir_bnd_end; ++b) {
      flx_sum += flx_wgt[b] * albout_lcl[b];
      flx_wgt_sum += flx_wgt[b];
    }
    albout[1] = flx_sum / flx_wgt_sum;
    for (int i = 0; i <= nlevsno; ++i) flx_abs[i * 2 + 0] = flx_abs_lcl[i * ELMO_NUMRAD_SNW + 0];
    for (int i = snl_top; i <= nlevsno; ++i) {
      flx_sum = 0.0;
      for (int b = nir_bnd_bgn; b <= nir_bnd_end; ++b) flx_sum += flx_wgt[b] * flx_abs_lcl[i * ELMO_NUMRAD_SNW + b];
      flx_abs[i * 2 + 1] = flx_sum / flx_wgt_sum;
    }
    if ((mu_not < mu_75) && (flg_slr_in == 1)) {
      const double sza_c1 = sza_a0 + sza_a1 * mu_not + sza_a2 * pow(mu_not, 2.0);
      const double sza_c0 = sza_b0 + sza_b1 * mu_not + sza_b2 * pow(mu_not, 2.0);
      const double sza_factor = sza_c1 * (log10(snw_rds_lcl[snl_top] * 1.0) - 6.0) + sza_c0;
      const double flx_sza_adjust = albout[1] * (sza_factor - 1.0) * flx_wgt_sum;
      albout[1] *= sza_factor;
      flx_abs[snl_top * 2 + 1] -= flx_sza_adjust;
    }
  } else if ((coszen > 0.0) && (h2osno < SN_MIN_SNW) && (h2osno > 0.0)) {
    albout[0] = albsoi[0];
    albout[1] = albsoi[1];
  } else {
    albout[0] = 0.0;
    albout[1] = 0.0;
  }
}
