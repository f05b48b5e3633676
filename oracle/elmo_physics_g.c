/* elmo_physics_g.c - snow hydrology, aerosol mass bookkeeping and the transpiration sink.
 * TEST INFRASTRUCTURE ONLY (see elm_oracle.h).
 *
 * Restates src/physics/snow_hydrology_impl.hh (snow_aging :50, snow_water :273, aerosol_phase_change :502,
 * snow_compaction :553, combine_layers :658, divide_layers :902, combine :1297, prune_snow_layers :1327),
 * src/physics/aerosol_physics_impl.hh (:10-107) and src/physics/transpiration_impl.hh (:15-28), one function per
 * reference function.
 *
 * Pinning.  The reference has no fixture for this path.  snow_hydrology.h reaches netcdf.h only through the file readers
 * (snow_hydrology.h:5 -> snicar_data.h:6 -> read_input.hh -> read_netcdf.hh), which no function here calls; with
 * read_input.hh skipped through its own include guard the header builds as it lies (oracle/ref_harness_snow.cc says exactly
 * how: one macro, one declaration of a reference name, no stand-in for netcdf or for any reference code), so snow_water,
 * aerosol_phase_change, transpiration, snow_compaction, combine_layers (+ combine), divide_layers, prune_snow_layers and
 * snow_aging (with the reference's own SnwRdsTable) are run against these restatements BIT FOR BIT, one wrapper stage at a
 * time on identical inputs, over chained model steps in which packs are built, split, merged, pruned and aged
 * (tests/test_oracle_vs_ref.py::test_snow_hydrology_stages_bitwise_vs_reference; columns that take one of the reference's two
 * out-of-bounds reads, (A) and (B) below, are left out - its result there is undefined).  PARITY UNPINNED remain the two
 * whole-array aerosol functions compute_aerosol_deposition / update_aerosol_mass_and_concen: their only body is a lambda
 * handed to the Kokkos dispatch (aerosol_physics_impl.hh:59, :106), which cannot be instantiated without Kokkos; those are
 * checked structurally (tests/test_snow_hydrology_oracle.py).
 *
 * Where the reference's result is not defined, the choice made here (and in the HIP kernels, include/elmk.h):
 *   (A) snow_water reads vol_ice[i+i] (:388, meant i+1) from a five-element stack array.  i = 0, 1, 2 are in bounds and
 *       are reproduced literally (the wrong layer for i = 0 and 2).  i = 3 reads vol_ice[6], out of bounds: vol_ice[i+1]
 *       is used there and ELMO_WARN_SNOW_WATER_OOB is raised.
 *   (B) combine_layers' shift loop (:871-885) runs one element too far: `k > nlevsno - snl - 1` copies element top - 1 into
 *       top.  With five layers that is element -1, out of bounds: 0.0 is used and ELMO_WARN_SNOW_COMBINE_OOB is raised
 *       (the element ends up above the pack, where prune_snow_layers / the aerosol update / snow_aging reset it anyway).
 *   (C) static_cast<int>(std::round(x)) of snow_aging's table indices (:121-123) is undefined for non-finite or huge x
 *       (zero layer thickness); x86-64's cvttsd2si gives INT_MIN there, which is what round_to_int returns.
 * Reference quirks kept as they are: snow_water works on level nlevsno - snl even when snl == 0 (the top soil layer);
 * depleted ice is reset to 0.9 kg/m2 (:303, :311); snow_aging clamps snw_rds to SNW_RDS_MIN from both sides (:217-223), so
 * every aged layer ends at SNW_RDS_MIN unless the arithmetic gave NaN; divide_layers' last bounds check tests rds[3]
 * instead of rds[4] (:1252); combine_layers' first loop keeps its bounds from before any removal (:675).
 */
#include <limits.h>
#include <math.h>

#include "elm_oracle.h"
#include "elmo_const.h"

#define NSNO ELMO_NLEVSNO
#define CPICE 2.11727e3 /* elm_constants.h:40 */
#define CPWAT 4.188e3   /* elm_constants.h:41 */

/* static_cast<int>(x) as x86-64 evaluates it (cvttsd2si): INT_MIN for NaN and for values outside the int range */
static int round_to_int(double x)
{
  const double r = round(x);
  if (!(r > -2147483649.0 && r < 2147483648.0)) return INT_MIN;
  return (int)r;
}

/* snow_hydrology_impl.hh:50-244 */
void elmo_snow_aging(int do_capsnow, int snl, double frac_sno, double dtime, double qflx_snwcp_ice, double qflx_snow_grnd,
                     double h2osno, const double *dz, const double *h2osoi_liq, const double *h2osoi_ice,
                     const double *t_soisno, const double *qflx_snofrz_lyr, const double *snowage_tau,
                     const double *snowage_kappa, const double *snowage_drdt0, double *snw_rds, uint32_t *err)
{
  const double snw_rds_refrz = 1000.0;
  const double C2_liq_Brun89 = 4.22e-13;
  if (snl > 0) {
    const int snl_btm = NSNO - 1;
    const int snl_top = NSNO - snl;
    for (int i = 0; i < snl_top; ++i) snw_rds[i] = 0.0;
    for (int i = snl_top; i <= snl_btm; ++i) {
      const double h2osno_lyr = h2osoi_liq[i] + h2osoi_ice[i];
      double t_snotop, t_snobtm;
      if (i == snl_top) {
        t_snotop = t_soisno[snl_top];
        t_snobtm = (t_soisno[i + 1] * dz[i] + t_soisno[i] * dz[i + 1]) / (dz[i] + dz[i + 1]);
      } else {
        t_snotop = (t_soisno[i - 1] * dz[i] + t_soisno[i] * dz[i - 1]) / (dz[i] + dz[i - 1]);
        t_snobtm = (t_soisno[i + 1] * dz[i] + t_soisno[i] * dz[i + 1]) / (dz[i] + dz[i + 1]);
      }
      const double cdz = frac_sno * dz[i];
      const double dTdz = fabs((t_snotop - t_snobtm) / cdz);
      double rhos = (h2osoi_liq[i] + h2osoi_ice[i]) / cdz;
      rhos = dmax(50.0, rhos);
      int T_idx = round_to_int((t_soisno[i] - 223) / 5);
      int Tgrd_idx = round_to_int(dTdz / 10);
      int rhos_idx = round_to_int((rhos - 50) / 50);
      if (T_idx < 0) T_idx = 0;
      if (T_idx > 10) T_idx = 10;
      if (Tgrd_idx < 0) Tgrd_idx = 0;
      if (Tgrd_idx > 30) Tgrd_idx = 30;
      if (rhos_idx < 0) rhos_idx = 0;
      if (rhos_idx > 7) rhos_idx = 7;
      const int k = (T_idx * 31 + Tgrd_idx) * 8 + rhos_idx; /* ArrayD3(idx_T_max + 1, idx_Tgrd_max + 1, idx_rhos_max + 1), row-major */
      const double bst_tau = snowage_tau[k], bst_kappa = snowage_kappa[k], bst_drdt0 = snowage_drdt0[k];
      double dr_fresh = snw_rds[i] - SNW_RDS_MIN;
      if (fabs(dr_fresh) < 1.0e-8) {
        dr_fresh = 0.0;
      } else if (dr_fresh < 0.0) {
        *err |= ELMO_ERR_SNOW_AGE_DRFRESH; /* throw at :152 */
      }
      double dr = (bst_drdt0 * pow(bst_tau / (dr_fresh + bst_tau), 1.0 / bst_kappa)) * (dtime / 3600.0);
      const double frc_liq = dmin(0.1, (h2osoi_liq[i] / (h2osoi_liq[i] + h2osoi_ice[i])));
      const double dr_wet = 1.0e18 * (dtime * (C2_liq_Brun89 * pow(frc_liq, 3.0)) / (4.0 * ELM_PI * pow(snw_rds[i], 2.0)));
      dr += dr_wet;
      double newsnow;
      if (do_capsnow) {
        newsnow = dmax(0.0, (qflx_snwcp_ice * dtime));
      } else {
        newsnow = dmax(0.0, (qflx_snow_grnd * dtime));
      }
      const double refrzsnow = dmax(0.0, (qflx_snofrz_lyr[i] * dtime));
      double frc_refrz = refrzsnow / h2osno_lyr;
      double frc_newsnow;
      if (i == snl_top) {
        frc_newsnow = newsnow / h2osno_lyr;
      } else {
        frc_newsnow = 0.0;
      }
      double frc_oldsnow;
      if ((frc_refrz + frc_newsnow) > 1.0) {
        frc_refrz = frc_refrz / (frc_refrz + frc_newsnow);
        frc_newsnow = 1.0 - frc_refrz;
        frc_oldsnow = 0.0;
      } else {
        frc_oldsnow = 1.0 - frc_refrz - frc_newsnow;
      }
      snw_rds[i] = (snw_rds[i] + dr) * frc_oldsnow + SNW_RDS_MIN * frc_newsnow + snw_rds_refrz * frc_refrz;
      if (snw_rds[i] < SNW_RDS_MIN) snw_rds[i] = SNW_RDS_MIN;
      if (snw_rds[i] > SNW_RDS_MIN) snw_rds[i] = SNW_RDS_MIN; /* (:221-223: the upper bound is SNW_RDS_MIN as well) */
    }
  }
  if (snl == 0) {
    if (h2osno > 0.0) snw_rds[NSNO - 1] = SNW_RDS_MIN;
  }
}

/* snow_hydrology_impl.hh:273-490 */
void elmo_snow_water(int do_capsnow, int snl, double dtime, double frac_sno_eff, double h2osno, double qflx_sub_snow,
                     double qflx_evap_grnd, double qflx_dew_snow, double qflx_dew_grnd, double qflx_rain_grnd,
                     double qflx_snomelt, double *qflx_snow_melt, double *qflx_top_soil, double *int_snow, double *frac_sno,
                     double *mflx_neg_snow, double *h2osoi_liq, double *h2osoi_ice, double *mss_bcphi, double *mss_bcpho,
                     double *mss_dst1, double *mss_dst2, double *mss_dst3, double *mss_dst4, double *dz, uint32_t *err)
{
  *mflx_neg_snow = 0.0;
  const int top = NSNO - snl;
  if (do_capsnow) {
    const double wgdif = h2osoi_ice[top] - frac_sno_eff * qflx_sub_snow * dtime;
    h2osoi_ice[top] = wgdif;
    if (wgdif < 0.0) {
      h2osoi_ice[top] = 0.9;
      h2osoi_liq[top] = h2osoi_liq[top] + wgdif;
    }
    h2osoi_liq[top] = h2osoi_liq[top] - frac_sno_eff * qflx_evap_grnd * dtime;
  } else {
    const double wgdif = h2osoi_ice[top] + frac_sno_eff * (qflx_dew_snow - qflx_sub_snow) * dtime;
    h2osoi_ice[top] = wgdif;
    if (wgdif < 0.0) {
      h2osoi_ice[top] = 0.9;
      h2osoi_liq[top] = h2osoi_liq[top] + wgdif;
    }
    h2osoi_liq[top] = h2osoi_liq[top] + frac_sno_eff * (qflx_rain_grnd + qflx_dew_grnd - qflx_evap_grnd) * dtime;
  }
  if (h2osoi_liq[top] < 0.0) {
    for (int i = top; i <= NSNO; ++i) {
      const double wgdif = h2osoi_liq[i];
      if (wgdif >= 0.0) break;
      h2osoi_liq[i] = 0.0;
      *mflx_neg_snow = wgdif / dtime;
    }
  }

  double vol_ice[NSNO], vol_liq[NSNO], eff_porosity[NSNO];
  for (int i = 0; i < NSNO; ++i) vol_ice[i] = vol_liq[i] = eff_porosity[i] = 0.0; /* (never read below `top`) */
  for (int i = top; i < NSNO; ++i) {
    vol_ice[i] = dmin(1.0, h2osoi_ice[i] / (dz[i] * frac_sno_eff * DENICE));
    eff_porosity[i] = 1.0 - vol_ice[i];
    vol_liq[i] = dmin(eff_porosity[i], h2osoi_liq[i] / (dz[i] * frac_sno_eff * DENH2O));
  }

  double qin = 0.0, qin_bc_phi = 0.0, qin_bc_pho = 0.0, qin_dst1 = 0.0, qin_dst2 = 0.0, qin_dst3 = 0.0, qin_dst4 = 0.0;
  double qout = 0.0;
  const double scvng_fct_mlt_bcphi = 0.20, scvng_fct_mlt_bcpho = 0.03, scvng_fct_mlt_dst1 = 0.02, scvng_fct_mlt_dst2 = 0.02,
               scvng_fct_mlt_dst3 = 0.01, scvng_fct_mlt_dst4 = 0.01;
  const double wimp = 0.05, ssi = 0.033;
  for (int i = top; i < NSNO; ++i) {
    h2osoi_liq[i] = h2osoi_liq[i] + qin;
    mss_bcphi[i] = mss_bcphi[i] + qin_bc_phi;
    mss_bcpho[i] = mss_bcpho[i] + qin_bc_pho;
    mss_dst1[i] = mss_dst1[i] + qin_dst1;
    mss_dst2[i] = mss_dst2[i] + qin_dst2;
    mss_dst3[i] = mss_dst3[i] + qin_dst3;
    mss_dst4[i] = mss_dst4[i] + qin_dst4;
    if (i < NSNO - 1) {
      if (eff_porosity[i] < wimp || eff_porosity[i + 1] < wimp) {
        qout = 0.0;
      } else {
        qout = dmax(0.0, (vol_liq[i] - ssi * eff_porosity[i]) * dz[i] * frac_sno_eff);
        /* :388 reads vol_ice[i+i]; choice (A) of the file header for i = 3 */
        double vi;
        if (i + i < NSNO) {
          vi = vol_ice[i + i];
        } else {
          vi = vol_ice[i + 1];
          *err |= ELMO_WARN_SNOW_WATER_OOB;
        }
        qout = dmin(qout, (1.0 - vi - vol_liq[i + 1]) * dz[i + 1] * frac_sno_eff);
      }
    } else {
      qout = dmax(0.0, (vol_liq[i] - ssi * eff_porosity[i]) * dz[i] * frac_sno_eff);
    }
    qout *= 1000.0;
    h2osoi_liq[i] -= qout;
    qin = qout;
    double mss_liqice = h2osoi_liq[i] + h2osoi_ice[i];
    if (mss_liqice < 1.0e-30) mss_liqice = 1.0e-30;
#define SCAVENGE(mss, fct, qin_x)                       \
  {                                                     \
    double qo = qout * (fct) * ((mss)[i] / mss_liqice); \
    if (qo > (mss)[i]) qo = (mss)[i];                   \
    (mss)[i] = (mss)[i] - qo;                           \
    qin_x = qo;                                         \
  }
    SCAVENGE(mss_bcphi, scvng_fct_mlt_bcphi, qin_bc_phi)
    SCAVENGE(mss_bcpho, scvng_fct_mlt_bcpho, qin_bc_pho)
    SCAVENGE(mss_dst1, scvng_fct_mlt_dst1, qin_dst1)
    SCAVENGE(mss_dst2, scvng_fct_mlt_dst2, qin_dst2)
    SCAVENGE(mss_dst3, scvng_fct_mlt_dst3, qin_dst3)
    SCAVENGE(mss_dst4, scvng_fct_mlt_dst4, qin_dst4)
#undef SCAVENGE
  }
  for (int i = top; i < NSNO; ++i) dz[i] = dmax(dz[i], h2osoi_liq[i] / DENH2O + h2osoi_ice[i] / DENICE);
  if (snl > 0) {
    *qflx_snow_melt += qout / dtime;
    *qflx_top_soil = (qout / dtime) + (1.0 - frac_sno_eff) * qflx_rain_grnd;
    *int_snow += frac_sno_eff * (qflx_dew_snow + qflx_dew_grnd + qflx_rain_grnd) * dtime;
  } else {
    *qflx_snow_melt = qflx_snomelt;
    *qflx_top_soil = qflx_rain_grnd + qflx_snomelt;
    if (h2osno <= 0.0) *int_snow = 0.0;
    if (h2osno <= 0.0) *frac_sno = 0.0;
  }
}

/* snow_hydrology_impl.hh:502-548 */
void elmo_aerosol_phase_change(int snl, double dtime, double qflx_sub_snow, const double *h2osoi_liq, const double *h2osoi_ice,
                               double *mss_bcphi, double *mss_bcpho)
{
  const int top = NSNO - snl;
  const double subsnow = dmax(0.0, (qflx_sub_snow * dtime));
  double frc_sub;
  if ((h2osoi_liq[top] + h2osoi_ice[top]) > 0.0) {
    frc_sub = subsnow / (h2osoi_liq[top] + h2osoi_ice[top]);
  } else {
    frc_sub = 0.0;
  }
  for (int i = top; i < NSNO; ++i) {
    if (i != top) frc_sub = 0.0;
    double frc_transfer = frc_sub;
    if (frc_transfer > 1.0) frc_transfer = 1.0;
    const double dm_int = mss_bcphi[i] * frc_transfer;
    mss_bcphi[i] -= dm_int;
    mss_bcpho[i] += dm_int;
  }
}

/* transpiration_impl.hh:15-28 (nlevsoi = 10, elm_constants.h:90) */
void elmo_transpiration(int veg_active, double qflx_tran_veg, const double *rootr, double *qflx_rootsoi)
{
  if (veg_active) {
    for (int i = 0; i < 10; ++i) qflx_rootsoi[i] = rootr[i] * qflx_tran_veg;
  }
}

/* snow_hydrology_impl.hh:553-645 */
void elmo_snow_compaction(int snl, int ltype, double dtime, double int_snow, double n_melt, double frac_sno, const int *imelt,
                          const double *swe_old, const double *h2osoi_liq, const double *h2osoi_ice, const double *t_soisno,
                          const double *frac_iceold, double *dz)
{
  const double c2 = 23.e-3, c3 = 2.777e-6, c4 = 0.04, c5 = 2.0, dm = 100.0, eta0 = 9.0e+5;
  const int top = NSNO - snl;
  double burden = 0.0;
  for (int i = top; i < NSNO; ++i) {
    double wx = h2osoi_ice[i] + h2osoi_liq[i];
    double vd = 1.0 - (h2osoi_ice[i] / DENICE + h2osoi_liq[i] / DENH2O) / dz[i];
    wx = (h2osoi_ice[i] + h2osoi_liq[i]);
    vd = 1.0 - (h2osoi_ice[i] / DENICE + h2osoi_liq[i] / DENH2O) / (frac_sno * dz[i]);
    if (vd > 0.001 && h2osoi_ice[i] > 0.1) {
      const double bi = h2osoi_ice[i] / (frac_sno * dz[i]);
      const double fi = h2osoi_ice[i] / wx;
      const double td = TFRZ - t_soisno[i];
      const double dexpf = exp(-c4 * td);
      double ddz1 = -c3 * dexpf;
      if (bi > dm) ddz1 *= exp(-46.0e-3 * (bi - dm));
      if (h2osoi_liq[i] > 0.01 * dz[i] * frac_sno) ddz1 *= c5;
      const double ddz2 = -(burden + wx / 2.0) * exp(-0.08 * td - c2 * bi) / eta0;
      double ddz3;
      if (imelt[i] == 1) {
        if (ltype == istsoil || ltype == istcrop) { /* subgridflag() == 1 */
          ddz3 = dmax(0.0, dmin(1.0, (swe_old[i] - wx) / wx));
          double wsum = 0.0;
          if ((swe_old[i] - wx) > 0.0) {
            if (i == top) {
              for (int j = top; j < NSNO; ++j) wsum += h2osoi_liq[j] + h2osoi_ice[j];
            }
            const double fsno_melt = 1.0 - pow(acos(2.0 * dmin(1.0, wsum / int_snow) - 1.0) / ELM_PI, n_melt);
            ddz3 -= dmax(0.0, (fsno_melt - frac_sno) / frac_sno);
          }
          ddz3 = -1.0 / dtime * ddz3;
        } else {
          ddz3 = -1.0 / dtime * dmax(0.0, (frac_iceold[i] - fi) / frac_iceold[i]);
        }
      } else {
        ddz3 = 0.0;
      }
      const double pdzdtc = ddz1 + ddz2 + ddz3;
      dz[i] = dmax(dz[i] * (1.0 + pdzdtc * dtime), (h2osoi_ice[i] / DENICE + h2osoi_liq[i] / DENH2O) / frac_sno);
    }
    burden += wx;
  }
}

/* snow_hydrology_impl.hh:1297-1321 */
static void combine(double dz2, double wliq2, double wice2, double t2, double *dz, double *wliq, double *wice, double *t)
{
  const double h = (CPICE * *wice + CPWAT * *wliq) * (*t - TFRZ) + HFUS * *wliq;
  const double h2 = (CPICE * wice2 + CPWAT * wliq2) * (t2 - TFRZ) + HFUS * wliq2;
  *wice += wice2;
  *wliq += wliq2;
  const double tc = TFRZ + (h + h2 - HFUS * *wliq) / (CPICE * *wice + CPWAT * *wliq);
  *dz += dz2;
  *t = tc;
}

/* element k - 1 of a level array in the shift loops: choice (B) of the file header for k - 1 == -1 */
static double below(const double *a, int km1, uint32_t *err)
{
  if (km1 < 0) {
    *err |= ELMO_WARN_SNOW_COMBINE_OOB;
    return 0.0;
  }
  return a[km1];
}

/* snow_hydrology_impl.hh:658-898 */
void elmo_combine_layers(int urbpoi, int ltype, double dtime, int *snl_io, double *h2osno, double *snow_depth,
                         double *frac_sno_eff, double *frac_sno, double *int_snow, double *qflx_sl_top_soil,
                         double *qflx_snow2topsoi, double *mflx_snowlyr_col, double *t_soisno, double *h2osoi_ice,
                         double *h2osoi_liq, double *snw_rds, double *mss_bcphi, double *mss_bcpho, double *mss_dst1,
                         double *mss_dst2, double *mss_dst3, double *mss_dst4, double *dz, double *z, double *zi, uint32_t *err)
{
  static const double dzmin[5] = {0.010, 0.015, 0.025, 0.055, 0.115};
  int snl = *snl_io;
  *qflx_sl_top_soil = 0.0;
  *qflx_snow2topsoi = 0.0;
  *mflx_snowlyr_col = 0.0;

  int top_old = NSNO - snl;
  for (int i = top_old; i < NSNO; ++i) {
    if (h2osoi_ice[i] <= .01) {
      if (ltype == istsoil || urbpoi || ltype == istcrop) {
        h2osoi_liq[i + 1] += h2osoi_liq[i];
        h2osoi_ice[i + 1] += h2osoi_ice[i];
        if (i == NSNO - 1) {
          *qflx_sl_top_soil = (h2osoi_liq[i] + h2osoi_ice[i]) / dtime;
          *mflx_snowlyr_col += *qflx_sl_top_soil;
        }
        if (i != NSNO - 1) {
          dz[i + 1] += dz[i];
          mss_bcphi[i + 1] += mss_bcphi[i];
          mss_bcpho[i + 1] += mss_bcpho[i];
          mss_dst1[i + 1] += mss_dst1[i];
          mss_dst2[i + 1] += mss_dst2[i];
          mss_dst3[i + 1] += mss_dst3[i];
          mss_dst4[i + 1] += mss_dst4[i];
        }
      } else if (ltype != istsoil && !urbpoi && ltype != istcrop && i != NSNO - 1) {
        h2osoi_liq[i + 1] += h2osoi_liq[i];
        h2osoi_ice[i + 1] += h2osoi_ice[i];
        dz[i + 1] += dz[i];
        mss_bcphi[i + 1] += mss_bcphi[i];
        mss_bcpho[i + 1] += mss_bcpho[i];
        mss_dst1[i + 1] += mss_dst1[i];
        mss_dst2[i + 1] += mss_dst2[i];
        mss_dst3[i + 1] += mss_dst3[i];
        mss_dst4[i + 1] += mss_dst4[i];
      }
      const int top = NSNO - snl;
      if (i > top && snl > 1) {
        for (int ii = i; ii > top; --ii) {
          if (ltype != istsoil && ltype != istcrop && !urbpoi && ii == NSNO - 1) {
            *qflx_sl_top_soil = (h2osoi_liq[ii] + h2osoi_ice[ii]) / dtime;
          }
          t_soisno[ii] = t_soisno[ii - 1];
          h2osoi_liq[ii] = h2osoi_liq[ii - 1];
          h2osoi_ice[ii] = h2osoi_ice[ii - 1];
          mss_bcphi[ii] = mss_bcphi[ii - 1];
          mss_bcpho[ii] = mss_bcpho[ii - 1];
          mss_dst1[ii] = mss_dst1[ii - 1];
          mss_dst2[ii] = mss_dst2[ii - 1];
          mss_dst3[ii] = mss_dst3[ii - 1];
          mss_dst4[ii] = mss_dst4[ii - 1];
          snw_rds[ii] = snw_rds[ii - 1];
          dz[ii] = dz[ii - 1];
        }
      }
      snl -= 1;
    }
  }

  *h2osno = 0.0;
  *snow_depth = 0.0;
  double zwice = 0.0, zwliq = 0.0;
  top_old = NSNO - snl;
  for (int i = top_old; i < NSNO; ++i) {
    *h2osno += h2osoi_ice[i] + h2osoi_liq[i];
    *snow_depth += dz[i];
    zwice += h2osoi_ice[i];
    zwliq += h2osoi_liq[i];
  }

  if (*snow_depth > 0.0 &&
      ((*frac_sno_eff * *snow_depth < 0.01) || (*h2osno / (*frac_sno_eff * *snow_depth) < 50.0))) {
    snl = 0;
    *h2osno = zwice;
    for (int i = 0; i < NSNO; ++i) {
      mss_bcphi[i] = 0.0;
      mss_bcpho[i] = 0.0;
      mss_dst1[i] = 0.0;
      mss_dst2[i] = 0.0;
      mss_dst3[i] = 0.0;
      mss_dst4[i] = 0.0;
    }
    if (*h2osno <= 0.0) *snow_depth = 0.0;
    if (ltype == istsoil || urbpoi || ltype == istcrop) {
      h2osoi_liq[NSNO - 1] = 0.0;
      h2osoi_liq[NSNO] += zwliq;
      *qflx_snow2topsoi = zwliq / dtime;
      *mflx_snowlyr_col += zwliq / dtime;
    }
    if (ltype == istwet || ltype == istice || ltype == istice_mec) h2osoi_liq[NSNO - 1] = 0.0;
  }

  if (*h2osno <= 0.0) {
    *snow_depth = 0.0;
    *frac_sno = 0.0;
    *frac_sno_eff = 0.0;
    *int_snow = 0.0;
  }

  if (snl > 1) {
    int mssi = 0;
    top_old = NSNO - snl;
    for (int i = top_old; i < NSNO; ++i) {
      if ((*frac_sno_eff * dz[i] < dzmin[mssi]) || ((h2osoi_ice[i] + h2osoi_liq[i]) / (*frac_sno_eff * dz[i]) < 50.0)) {
        int neibor;
        if (i == NSNO - snl) {
          neibor = i + 1;
        } else if (i == NSNO - 1) {
          neibor = i - 1;
        } else {
          neibor = i + 1;
          if ((dz[i - 1] + dz[i]) < (dz[i + 1] + dz[i])) neibor = i - 1;
        }
        int j, l;
        if (neibor > i) {
          j = neibor;
          l = i;
        } else {
          j = i;
          l = neibor;
        }
        mss_bcphi[j] += mss_bcphi[l];
        mss_bcpho[j] += mss_bcpho[l];
        mss_dst1[j] += mss_dst1[l];
        mss_dst2[j] += mss_dst2[l];
        mss_dst3[j] += mss_dst3[l];
        mss_dst4[j] += mss_dst4[l];
        snw_rds[j] = (snw_rds[j] * (h2osoi_liq[j] + h2osoi_ice[j]) + snw_rds[l] * (h2osoi_liq[l] + h2osoi_ice[l])) /
                     (h2osoi_liq[j] + h2osoi_ice[j] + h2osoi_liq[l] + h2osoi_ice[l]);
        combine(dz[l], h2osoi_liq[l], h2osoi_ice[l], t_soisno[l], &dz[j], &h2osoi_liq[j], &h2osoi_ice[j], &t_soisno[j]);
        if (j - 1 > NSNO - snl) {
          for (int k = j - 1; k > NSNO - snl - 1; --k) {
            t_soisno[k] = below(t_soisno, k - 1, err);
            h2osoi_ice[k] = below(h2osoi_ice, k - 1, err);
            h2osoi_liq[k] = below(h2osoi_liq, k - 1, err);
            mss_bcphi[k] = below(mss_bcphi, k - 1, err);
            mss_bcpho[k] = below(mss_bcpho, k - 1, err);
            mss_dst1[k] = below(mss_dst1, k - 1, err);
            mss_dst2[k] = below(mss_dst2, k - 1, err);
            mss_dst3[k] = below(mss_dst3, k - 1, err);
            mss_dst4[k] = below(mss_dst4, k - 1, err);
            snw_rds[k] = below(snw_rds, k - 1, err);
            dz[k] = below(dz, k - 1, err);
          }
        }
        snl -= 1;
        if (snl <= 1) break;
      } else {
        mssi += 1;
      }
    }
  }

  for (int i = NSNO - 1; i >= NSNO - snl; --i) {
    z[i] = zi[i + 1] - 0.5 * dz[i];
    zi[i] = zi[i + 1] - dz[i];
  }
  *snl_io = snl;
}

/* one step of divide_layers: the part of layer k beyond `keep` metres moves into layer k + 1 (:987-1041 for k = 0, and the
 * three copies of it below) */
typedef struct {
  double dzsno[NSNO], swice[NSNO], swliq[NSNO], tsno[NSNO], mbc_phi[NSNO], mbc_pho[NSNO], mdst1[NSNO], mdst2[NSNO],
      mdst3[NSNO], mdst4[NSNO], rds[NSNO];
} snow_stack;

static void move_excess(snow_stack *s, int k, double keep, int check, uint32_t *err)
{
  const double drr = s->dzsno[k] - keep;
  double propor = drr / s->dzsno[k];
  double zwice = propor * s->swice[k];
  double zwliq = propor * s->swliq[k];
  const double zmbc_phi = propor * s->mbc_phi[k];
  const double zmbc_pho = propor * s->mbc_pho[k];
  const double zmdst1 = propor * s->mdst1[k];
  const double zmdst2 = propor * s->mdst2[k];
  const double zmdst3 = propor * s->mdst3[k];
  const double zmdst4 = propor * s->mdst4[k];
  propor = keep / s->dzsno[k];
  s->swice[k] *= propor;
  s->swliq[k] *= propor;
  s->mbc_phi[k] *= propor;
  s->mbc_pho[k] *= propor;
  s->mdst1[k] *= propor;
  s->mdst2[k] *= propor;
  s->mdst3[k] *= propor;
  s->mdst4[k] *= propor;
  s->dzsno[k] = keep;
  s->mbc_phi[k + 1] += zmbc_phi;
  s->mbc_pho[k + 1] += zmbc_pho;
  s->mdst1[k + 1] += zmdst1;
  s->mdst2[k + 1] += zmdst2;
  s->mdst3[k + 1] += zmdst3;
  s->mdst4[k + 1] += zmdst4;
  s->rds[k + 1] = (s->rds[k + 1] * (s->swliq[k + 1] + s->swice[k + 1]) + s->rds[k] * (zwliq + zwice)) /
                  (s->swliq[k + 1] + s->swice[k + 1] + zwliq + zwice);
  if (s->rds[check] < 30 || s->rds[check] > 1500) *err |= ELMO_ERR_SNOW_DIVIDE_RDS; /* snw_rds_min_tbl / max_tbl */
  combine(drr, zwliq, zwice, s->tsno[k], &s->dzsno[k + 1], &s->swliq[k + 1], &s->swice[k + 1], &s->tsno[k + 1]);
}

/* the new layer k + 1 gets half of layer k (:1043-1076 for k = 1 and its copies); tsno_test: the element whose temperature
 * the freezing-point test reads (the new layer's, except at :1139 where the reference tests tsno[2] for the new layer 3) */
static void split_layer(snow_stack *s, int k, int tsno_test)
{
  const double dtdz = (s->tsno[k - 1] - s->tsno[k]) / ((s->dzsno[k - 1] + s->dzsno[k]) / 2.0);
  s->dzsno[k] /= 2.0;
  s->swice[k] /= 2.0;
  s->swliq[k] /= 2.0;
  s->dzsno[k + 1] = s->dzsno[k];
  s->swice[k + 1] = s->swice[k];
  s->swliq[k + 1] = s->swliq[k];
  s->tsno[k + 1] = s->tsno[k] - dtdz * s->dzsno[k] / 2.0;
  if (s->tsno[tsno_test] >= TFRZ) {
    s->tsno[k + 1] = s->tsno[k];
  } else {
    s->tsno[k] += dtdz * s->dzsno[k] / 2.0;
  }
  s->mbc_phi[k] /= 2.0;
  s->mbc_phi[k + 1] = s->mbc_phi[k];
  s->mbc_pho[k] /= 2.0;
  s->mbc_pho[k + 1] = s->mbc_pho[k];
  s->mdst1[k] /= 2.0;
  s->mdst1[k + 1] = s->mdst1[k];
  s->mdst2[k] /= 2.0;
  s->mdst2[k + 1] = s->mdst2[k];
  s->mdst3[k] /= 2.0;
  s->mdst3[k + 1] = s->mdst3[k];
  s->mdst4[k] /= 2.0;
  s->mdst4[k + 1] = s->mdst4[k];
  s->rds[k + 1] = s->rds[k];
}

/* snow_hydrology_impl.hh:902-1288 */
void elmo_divide_layers(double frac_sno, int *snl_io, double *h2osoi_ice, double *h2osoi_liq, double *t_soisno, double *snw_rds,
                        double *mss_bcphi, double *mss_bcpho, double *mss_dst1, double *mss_dst2, double *mss_dst3,
                        double *mss_dst4, double *dz, double *z, double *zi, uint32_t *err)
{
  snow_stack s;
  const int snl = *snl_io;
  for (int i = 0; i < NSNO; ++i) {
    s.dzsno[i] = s.swice[i] = s.swliq[i] = s.tsno[i] = s.mbc_phi[i] = s.mbc_pho[i] = 0.0;
    s.mdst1[i] = s.mdst2[i] = s.mdst3[i] = s.mdst4[i] = s.rds[i] = 0.0; /* (elements >= snl are written before they are read) */
  }
  int msno = snl;
  int top = NSNO - snl;
  for (int i = 0; i < snl; ++i) {
    s.dzsno[i] = frac_sno * dz[i + top];
    s.swice[i] = h2osoi_ice[i + top];
    s.swliq[i] = h2osoi_liq[i + top];
    s.tsno[i] = t_soisno[i + top];
    s.mbc_phi[i] = mss_bcphi[i + top];
    s.mbc_pho[i] = mss_bcpho[i + top];
    s.mdst1[i] = mss_dst1[i + top];
    s.mdst2[i] = mss_dst2[i + top];
    s.mdst3[i] = mss_dst3[i + top];
    s.mdst4[i] = mss_dst4[i + top];
    s.rds[i] = snw_rds[i + top];
  }
  if (msno == 1) {
    if (s.dzsno[0] > 0.03) { /* :956-980: one layer becomes two equal ones (no temperature gradient term) */
      msno = 2;
      s.dzsno[0] /= 2.0;
      s.swice[0] /= 2.0;
      s.swliq[0] /= 2.0;
      s.dzsno[1] = s.dzsno[0];
      s.swice[1] = s.swice[0];
      s.swliq[1] = s.swliq[0];
      s.tsno[1] = s.tsno[0];
      s.mbc_phi[0] /= 2.0;
      s.mbc_phi[1] = s.mbc_phi[0];
      s.mbc_pho[0] /= 2.0;
      s.mbc_pho[1] = s.mbc_pho[0];
      s.mdst1[0] /= 2.0;
      s.mdst1[1] = s.mdst1[0];
      s.mdst2[0] /= 2.0;
      s.mdst2[1] = s.mdst2[0];
      s.mdst3[0] /= 2.0;
      s.mdst3[1] = s.mdst3[0];
      s.mdst4[0] /= 2.0;
      s.mdst4[1] = s.mdst4[0];
      s.rds[1] = s.rds[0];
    }
  }
  if (msno > 1) {
    if (s.dzsno[0] > 0.02) {
      move_excess(&s, 0, 0.02, 1, err);
      if (msno <= 2 && s.dzsno[1] > 0.07) {
        msno = 3;
        split_layer(&s, 1, 2);
      }
    }
  }
  if (msno > 2) {
    if (s.dzsno[1] > 0.05) {
      move_excess(&s, 1, 0.05, 2, err);
      if (msno <= 3 && s.dzsno[2] > 0.18) {
        msno = 4;
        split_layer(&s, 2, 2); /* (:1139 tests tsno[2], not the new layer's tsno[3]) */
      }
    }
  }
  if (msno > 3) {
    if (s.dzsno[2] > 0.11) {
      move_excess(&s, 2, 0.11, 3, err);
      if (msno <= 4 && s.dzsno[3] > 0.41) {
        msno = 5;
        split_layer(&s, 3, 4);
      }
    }
  }
  if (msno > 4) {
    if (s.dzsno[3] > 0.23) move_excess(&s, 3, 0.23, 3, err); /* (:1252 checks rds[3], not rds[4]) */
  }
  top = NSNO - msno;
  for (int i = top; i < NSNO; ++i) {
    dz[i] = s.dzsno[i - top] / frac_sno;
    h2osoi_ice[i] = s.swice[i - top];
    h2osoi_liq[i] = s.swliq[i - top];
    t_soisno[i] = s.tsno[i - top];
    mss_bcphi[i] = s.mbc_phi[i - top];
    mss_bcpho[i] = s.mbc_pho[i - top];
    mss_dst1[i] = s.mdst1[i - top];
    mss_dst2[i] = s.mdst2[i - top];
    mss_dst3[i] = s.mdst3[i - top];
    mss_dst4[i] = s.mdst4[i - top];
    snw_rds[i] = s.rds[i - top];
  }
  for (int i = NSNO - 1; i >= top; --i) {
    z[i] = zi[i + 1] - 0.5 * dz[i];
    zi[i] = zi[i + 1] - dz[i];
  }
  *snl_io = msno;
}

/* snow_hydrology_impl.hh:1327-1349 */
void elmo_prune_snow_layers(int snl, double *h2osoi_ice, double *h2osoi_liq, double *t_soisno, double *dz, double *z, double *zi)
{
  const int top = NSNO - snl;
  for (int i = 0; i < top; ++i) {
    h2osoi_ice[i] = 0.0;
    h2osoi_liq[i] = 0.0;
    t_soisno[i] = 0.0;
    dz[i] = 0.0;
    z[i] = 0.0;
    zi[i] = 0.0;
  }
}

/* aerosol_physics_impl.hh:36-64: one column of compute_aerosol_deposition; aer[11] = bcphi, bcpho, bcdep, dst1_1, dst1_2,
 * dst2_1, dst2_2, dst3_1, dst3_2, dst4_1, dst4_2 (AerosolFileInput, aerosol_data.h:11-22) */
void elmo_aerosol_deposition(double dtime, int snl, const double *aer, double *mss_bcphi, double *mss_bcpho, double *mss_dst1,
                             double *mss_dst2, double *mss_dst3, double *mss_dst4)
{
  if (snl > 0) {
    const int j = NSNO - snl;
    mss_bcphi[j] += (aer[0] * dtime);
    mss_bcpho[j] += ((aer[1] + aer[2]) * dtime);
    mss_dst1[j] += ((aer[3] + aer[4]) * dtime);
    mss_dst2[j] += ((aer[5] + aer[6]) * dtime);
    mss_dst3[j] += ((aer[7] + aer[8]) * dtime);
    mss_dst4[j] += ((aer[9] + aer[10]) * dtime);
  }
}

/* aerosol_physics_impl.hh:10-31, :67-106: one column of update_aerosol_mass_and_concen; mss / cnc: six arrays of five levels */
void elmo_aerosol_mass_and_concen(double dtime, int snl, int do_capsnow, double qflx_snwcp_ice, const double *h2osoi_ice,
                                  const double *h2osoi_liq, double *const mss[6], double *const cnc[6])
{
  const int snotop = NSNO - snl;
  for (int sl = 0; sl < NSNO; sl++) {
    const double snowmass = (sl < snotop) ? 1.e-12 : h2osoi_ice[sl] + h2osoi_liq[sl];
    const double snowcap_scl_fct =
        (sl == snotop && do_capsnow) ? (snowmass / (snowmass + qflx_snwcp_ice * dtime)) : (sl < snotop) ? 0.0 : 1.0;
    for (int a = 0; a < 6; a++) mss[a][sl] *= snowcap_scl_fct;
    const double snwmss_inv = 1.0 / snowmass;
    for (int a = 0; a < 6; a++) cnc[a][sl] = mss[a][sl] * snwmss_inv;
  }
}
