/* elmo_physics_h.c - cold-start initialisation of a column: what ELM::initialize_kokkos_elm runs once per column after the
 * input files are read (driver/kokkos/initialize_elm_kokkos.cc:373-428, the "init functions" lambda).  It is the producer
 * of the state the hot path consumes: hydraulic / thermal soil parameters, root fractions, the initial snow mesh, soil
 * temperature and water.  TEST INFRASTRUCTURE ONLY (see elm_oracle.h).
 *
 * Restates, one function per reference function:
 *   src/physics/init_topography_impl.hh          init_topo_slope :7, init_melt_factor :14, init_micro_sigma :31
 *   src/physics/init_snow_state_impl.hh          init_snow_state :11-63, init_snow_layers :67-151
 *   src/physics/soil_texture_hydraulic_model_impl.hh  pedotransfer :7, soil_hydraulic_params :19-94, init_soil_hydraulics :98-123
 *   src/physics/init_soil_state_impl.hh          init_soil_temp :11-54, init_soilh2o_state :65-176, init_vegrootfr :180-213
 * All four headers compile here without netcdf, so this file is pinned bit for bit against the reference itself
 * (oracle/ref_harness.cc: elmref_initialize_state; tests/test_oracle_vs_ref.py).
 *
 * Reference behaviour kept as it is: init_snow_state zeroes snow_depth and h2osno AFTER init_snow_layers has built the
 * layer mesh from snow_depth (so a column can start with snl > 0 and no snow mass; the snow-cover-fraction branch behind
 * `snow_depth > 0` is dead); init_soil_hydraulics writes csol(0..14) by SOIL index although csol has 20 levels and
 * soil_temperature reads it by LEVEL index (soil_thermal_properties_impl.hh:185); init_soilh2o_state's last loop overwrites
 * the liquid / ice split of every layer of every land type (the reference's own TODO).  GCC folds pow(x, 2.0) to x*x and
 * pow(x, 1.0) to x at -O2; written out here. */
#include <math.h>

#include "elm_oracle.h"
#include "elmo_const.h"

#define NSNO ELMO_NLEVSNO
#define NGRND 15 /* nlevgrnd, elm_constants.h:89 */
#define NSOI 10  /* nlevsoi :90 */
#define NBED 15  /* nlevbed :91 */
#define NURB 5   /* nlevurb :87 */
#define BDSNO 250.0
#define SECSPDAY 86400.0

/* init_topography_impl.hh:7-11 */
double elmo_init_topo_slope(double raw_topo_slope) { return dmax(raw_topo_slope, 0.2); }

/* :14-28 */
double elmo_init_melt_factor(int ltype, double topo_std)
{
  double n_melt;
  if (ltype == istice_mec) {
    n_melt = 10.0;
  } else {
    n_melt = 200.0 / dmax(10.0, topo_std);
  }
  return n_melt;
}

/* :31-38 */
double elmo_init_micro_sigma(double topo_slope)
{
  const double slopebeta = 3.0;
  const double slopemax = 0.4;
  const double slope0 = pow(slopemax, (-1.0 / slopebeta));
  return pow((topo_slope + slope0), -slopebeta);
}

/* init_snow_state_impl.hh:67-151; dz, z: 20 levels, zi: 21 */
void elmo_init_snow_layers(double snow_depth, int lakpoi, int *snl_io, double *dz, double *z, double *zi)
{
  int snl = *snl_io;
  for (int i = 0; i < NSNO; i++) {
    dz[i] = SPVAL;
    z[i] = SPVAL;
    zi[i] = SPVAL;
  }
  if (!lakpoi) {
    if (snow_depth < 0.01) {
      snl = 0;
      for (int i = 0; i < NSNO; i++) {
        dz[i] = 0.0;
        z[i] = 0.0;
        zi[i] = 0.0;
      }
      zi[NSNO] = 0.0;
    } else {
      if ((snow_depth >= 0.01) && (snow_depth <= 0.03)) {
        snl = 1;
        dz[4] = snow_depth;
      } else if ((snow_depth > 0.03) && (snow_depth <= 0.04)) {
        snl = 2;
        dz[3] = snow_depth / 2.0;
        dz[4] = dz[3];
      } else if ((snow_depth > 0.04) && (snow_depth <= 0.07)) {
        snl = 2;
        dz[3] = 0.02;
        dz[4] = snow_depth - dz[3];
      } else if ((snow_depth > 0.07) && (snow_depth <= 0.12)) {
        snl = 3;
        dz[2] = 0.02;
        dz[3] = (snow_depth - 0.02) / 2.0;
        dz[4] = dz[3];
      } else if ((snow_depth > 0.12) && (snow_depth <= 0.18)) {
        snl = 3;
        dz[2] = 0.02;
        dz[3] = 0.05;
        dz[4] = snow_depth - dz[2] - dz[3];
      } else if ((snow_depth > 0.18) && (snow_depth <= 0.29)) {
        snl = 4;
        dz[1] = 0.02;
        dz[2] = 0.05;
        dz[3] = (snow_depth - dz[1] - dz[2]) / 2.0;
        dz[4] = dz[3];
      } else if ((snow_depth > 0.29) && (snow_depth <= 0.41)) {
        snl = 4;
        dz[1] = 0.02;
        dz[2] = 0.05;
        dz[3] = 0.11;
        dz[4] = snow_depth - dz[1] - dz[2] - dz[3];
      } else if ((snow_depth > 0.41) && (snow_depth <= 0.64)) {
        snl = 5;
        dz[0] = 0.02;
        dz[1] = 0.05;
        dz[2] = 0.11;
        dz[3] = (snow_depth - dz[0] - dz[1] - dz[2]) / 2.0;
        dz[4] = dz[3];
      } else if (snow_depth > 0.64) {
        snl = 5;
        dz[0] = 0.02;
        dz[1] = 0.05;
        dz[2] = 0.11;
        dz[3] = 0.23;
        dz[4] = snow_depth - dz[0] - dz[1] - dz[2] - dz[3];
      }
    }
    for (int j = NSNO - 1; j >= NSNO - snl; j--) {
      z[j] = zi[j + 1] - 0.5 * dz[j];
      zi[j] = zi[j + 1] - dz[j];
    }
  } else {
    snl = 0;
    for (int i = 0; i < NSNO; i++) {
      dz[i] = 0.0;
      z[i] = 0.0;
      zi[i] = 0.0;
    }
    zi[NSNO] = 0.0;
  }
  *snl_io = snl;
}

/* soil_texture_hydraulic_model_impl.hh:7-16 */
static void pedotransfer(double pct_sand, double pct_clay, double *watsat, double *bsw, double *sucsat, double *xksat)
{
  *watsat = 0.489 - 0.00126 * pct_sand;
  *bsw = 2.91 + 0.159 * pct_clay;
  *sucsat = 10.0 * pow(10.0, (1.88 - 0.0131 * pct_sand));
  *xksat = 0.0070556 * pow(10.0, (-0.884 + 0.0153 * pct_sand));
}

/* :19-94 */
void elmo_soil_hydraulic_params(double pct_sand, double pct_clay, double zsoi, double om_frac, double *watsat, double *bsw,
                                double *sucsat, double *watdry, double *watopt, double *watfc, double *tkmg, double *tkdry,
                                double *csol)
{
  const double zsapric = 0.5, pcalpha = 0.5, pcbeta = 0.139, om_tkd = 0.05, om_tkm = 0.25, om_csol = 2.5;
  double xksat;
  pedotransfer(pct_sand, pct_clay, watsat, bsw, sucsat, &xksat);
  const double om_watsat = dmax(0.93 - 0.1 * (zsoi / zsapric), 0.83);
  const double om_b = dmin(2.7 + 9.3 * (zsoi / zsapric), 12.0);
  const double om_sucsat = dmin(10.3 - 0.2 * (zsoi / zsapric), 10.1);
  const double om_hksat = dmax(0.28 - 0.2799 * (zsoi / zsapric), 0.0001);

  const double bulk_den = (1.0 - *watsat) * 2.7e3;
  const double tkm = (1.0 - om_frac) * (8.8 * pct_sand + 2.92 * pct_clay) / (pct_sand + pct_clay) + om_tkm * om_frac;
  *watsat = (1.0 - om_frac) * *watsat + om_watsat * om_frac;
  *bsw = (1.0 - om_frac) * (2.91 + 0.159 * pct_clay) + om_frac * om_b;
  *sucsat = (1.0 - om_frac) * *sucsat + om_sucsat * om_frac;

  double perc_frac;
  if (om_frac > pcalpha) {
    const double perc_norm = pow((1.0 - pcalpha), -pcbeta);
    perc_frac = perc_norm * pow((om_frac - pcalpha), pcbeta);
  } else {
    perc_frac = 0.0;
  }
  const double uncon_frac = (1.0 - om_frac) + (1.0 - perc_frac) * om_frac;
  double uncon_hksat;
  if (om_frac < 1.0) {
    uncon_hksat = uncon_frac / ((1.0 - om_frac) / xksat + ((1.0 - perc_frac) * om_frac) / om_hksat);
  } else {
    uncon_hksat = 0.0;
  }
  const double hksat = uncon_frac * uncon_hksat + (perc_frac * om_frac) * om_hksat;

  *tkmg = pow(tkm, (1.0 - *watsat));
  *tkdry = ((0.135 * bulk_den + 64.7) / (2.7e3 - 0.947 * bulk_den)) * (1.0 - om_frac) + om_tkd * om_frac;
  *csol = ((1.0 - om_frac) * (2.128 * pct_sand + 2.385 * pct_clay) / (pct_sand + pct_clay) + om_csol * om_frac) * 1.0e6;
  *watdry = *watsat * pow((316230.0 / *sucsat), (-1.0 / *bsw));
  *watopt = *watsat * pow((158490.0 / *sucsat), (-1.0 / *bsw));
  *watfc = *watsat * pow((0.1 / (hksat * SECSPDAY)), (1.0 / (2.0 * *bsw + 3.0)));
}

/* :98-123; pct_sand, pct_clay, organic: nlevgrnd values; zsoi: 20 levels; the outputs by soil index */
void elmo_init_soil_hydraulics(double organic_max, const double *pct_sand, const double *pct_clay, const double *organic,
                               const double *zsoi, double *watsat, double *bsw, double *sucsat, double *watdry,
                               double *watopt, double *watfc, double *tkmg, double *tkdry, double *csol)
{
  const double csol_bedrock = 2.0e6;
  for (int i = 0; i < NSOI; ++i) {
    const double q = organic[i] / organic_max;
    const double om_frac = q * q; /* pow(q, 2.0) at -O2 */
    elmo_soil_hydraulic_params(pct_sand[i], pct_clay[i], zsoi[i + NSNO], om_frac, &watsat[i], &bsw[i], &sucsat[i], &watdry[i],
                               &watopt[i], &watfc[i], &tkmg[i], &tkdry[i], &csol[i]);
  }
  for (int i = NSOI; i < NGRND; ++i) {
    elmo_soil_hydraulic_params(pct_sand[NSOI - 1], pct_clay[NSOI - 1], zsoi[i + NSNO], 0.0, &watsat[i], &bsw[i], &sucsat[i],
                               &watdry[i], &watopt[i], &watfc[i], &tkmg[i], &tkdry[i], &csol[i]);
    csol[i] = csol_bedrock;
  }
}

/* init_soil_state_impl.hh:180-213; zi: 21 interface depths, rootfr: nlevgrnd */
void elmo_init_vegrootfr(int vtype, double roota_par, double rootb_par, const double *zi, double *rootfr)
{
  for (int i = NSOI; i < NGRND; ++i) rootfr[i] = 0.0;
  if (vtype != 0 /* PFT::noveg() */) {
    double totrootfr = 0.0; /* (summed and never used in the reference) */
    for (int i = 0; i < NSOI - 1; i++) {
      rootfr[i] = 0.5 * (exp(-roota_par * zi[i + NSNO]) + exp(-rootb_par * zi[i + NSNO]) - exp(-roota_par * zi[i + 1 + NSNO]) -
                         exp(-rootb_par * zi[i + 1 + NSNO]));
      if (i < NBED) totrootfr += rootfr[i];
    }
    (void)totrootfr;
    rootfr[NSOI - 1] = 0.5 * (exp(-roota_par * zi[NSOI - 1 + NSNO]) + exp(-rootb_par * zi[NSOI - 1 + NSNO]));
  } else {
    for (int i = 0; i < NSOI; i++) rootfr[i] = 0.0;
  }
  for (int i = NSOI; i < NGRND; i++) rootfr[i] = 0.0;
}

/* :11-54 */
void elmo_init_soil_temp(const elmo_land *L, int snl, double *t_soisno, double *t_grnd)
{
  if (snl > 0) {
    for (int i = NSNO - snl; i < NSNO; ++i) t_soisno[i] = 250.0;
  }
  if (!L->lakpoi) {
    if (L->ltype == istice || L->ltype == istice_mec) {
      for (int i = NSNO; i < NGRND + NSNO; ++i) t_soisno[i] = 250.0;
    } else if (L->ltype == istwet) {
      for (int i = NSNO; i < NGRND + NSNO; ++i) t_soisno[i] = 277.0;
    } else if (L->urbpoi) {
      if (L->ctype == icol_road_perv || L->ctype == icol_road_imperv) {
        for (int i = NSNO; i < NGRND + NSNO; ++i) t_soisno[i] = 274.0;
      } else if (L->ctype == icol_sunwall || L->ctype == icol_shadewall || L->ctype == icol_roof) {
        for (int i = NSNO; i < NURB + NSNO; ++i) t_soisno[i] = 292.0;
      }
    } else {
      for (int i = NSNO; i < NGRND + NSNO; ++i) t_soisno[i] = 274.0;
    }
    *t_grnd = t_soisno[NSNO - snl];
  }
}

/* init_snow_state_impl.hh:11-63 */
void elmo_init_snow_state(int urbpoi, int snl, double *h2osno, double *int_snow, double *snow_depth, double *h2osfc,
                          double *h2ocan, double *frac_h2osfc, double *fwet, double *fdry, double *frac_sno, double *snw_rds)
{
  *h2osno = 0.0;
  *int_snow = 0.0;
  *snow_depth = 0.0;
  *h2osfc = 0.0;
  *h2ocan = 0.0;
  *frac_h2osfc = 0.0;
  *fwet = 0.0;
  *fdry = 0.0;
  if (urbpoi) {
    *frac_sno = dmin(*snow_depth / 0.05, 1.0);
  } else {
    *frac_sno = 0.0;
    if (*snow_depth > 0.0) { /* (never: snow_depth was set to zero three lines up) */
      const double snowbd = dmin(400.0, *h2osno / *snow_depth);
      const double fmelt = snowbd / 100.0; /* pow(x, 1.0) */
      *frac_sno = tanh(*snow_depth / (2.5 * ZLND * fmelt));
    }
  }
  if (snl > 0) {
    for (int i = 0; i < NSNO - snl; ++i) snw_rds[i] = 0.0;
    for (int i = NSNO - snl; i < NSNO; ++i) snw_rds[i] = SNW_RDS_MIN;
  } else if (*h2osno > 0.0) {
    snw_rds[NSNO - 1] = SNW_RDS_MIN;
    for (int i = 0; i < NSNO - 1; ++i) snw_rds[i] = 0.0;
  } else {
    for (int i = 0; i < NSNO; ++i) snw_rds[i] = 0.0;
  }
}

/* init_soil_state_impl.hh:65-176 */
void elmo_init_soilh2o_state(const elmo_land *L, int snl, const double *watsat, const double *t_soisno, const double *dz,
                             double *h2osoi_vol, double *h2osoi_liq, double *h2osoi_ice)
{
  for (int i = 0; i < NGRND; ++i) h2osoi_vol[i] = SPVAL;
  for (int i = 0; i < NGRND + NSNO; ++i) h2osoi_liq[i] = SPVAL;
  for (int i = 0; i < NGRND + NSNO; ++i) h2osoi_ice[i] = SPVAL;
  int nlevs = NGRND;
  if (!L->lakpoi) {
    if (L->ltype == istsoil || L->ltype == istcrop) {
      for (int i = 0; i < NGRND; ++i) {
        if (i >= NBED) {
          h2osoi_vol[i] = 0.0;
        } else {
          h2osoi_vol[i] = 0.15;
        }
      }
    } else if (L->urbpoi) {
      if (L->ctype == icol_road_perv) {
        for (int i = 0; i < NGRND; ++i) {
          if (i < NBED) {
            h2osoi_vol[i] = 0.3;
          } else {
            h2osoi_vol[i] = 0.0;
          }
        }
      } else if (L->ctype == icol_road_imperv) {
        for (int i = 0; i < NGRND; ++i) h2osoi_vol[i] = 0.0;
      } else {
        nlevs = NURB;
        for (int i = 0; i < NURB; ++i) h2osoi_vol[i] = 0.0;
      }
    } else if (L->ltype == istwet) {
      for (int i = 0; i < NGRND; ++i) {
        if (i >= NBED) {
          h2osoi_vol[i] = 0.0;
        } else {
          h2osoi_vol[i] = 1.0;
        }
      }
    } else if (L->ltype == istice || L->ltype == istice_mec) {
      for (int i = 0; i < NGRND; ++i) h2osoi_vol[i] = 1.0;
    }
    for (int i = 0; i < nlevs; ++i) {
      const int o = i + NSNO;
      h2osoi_vol[i] = dmin(h2osoi_vol[i], watsat[i]);
      if (t_soisno[o] <= TFRZ) {
        h2osoi_ice[o] = dz[o] * DENICE * h2osoi_vol[i];
        h2osoi_liq[o] = 0.0;
      } else {
        h2osoi_ice[o] = 0.0;
        h2osoi_liq[o] = dz[o] * DENH2O * h2osoi_vol[i];
      }
    }
    for (int i = 0; i < NSNO; ++i) {
      if (i >= NSNO - snl) {
        h2osoi_ice[i] = dz[i] * 250.0;
        h2osoi_liq[i] = 0.0;
      }
    }
  } else {
    for (int i = 0; i < NSNO; ++i) {
      if (i >= NSNO - snl) {
        h2osoi_ice[i] = dz[i] * BDSNO;
        h2osoi_liq[i] = 0.0;
      }
    }
    for (int i = 0; i < NGRND; ++i) {
      const int o = i + NSNO;
      if (i < NSOI) {
        h2osoi_vol[i] = watsat[i];
        h2osoi_liq[o] = SPVAL;
        h2osoi_ice[o] = SPVAL;
      } else {
        h2osoi_vol[i] = 0.0;
      }
    }
  }
  for (int i = 0; i < NGRND; ++i) {
    const int o = i + NSNO;
    if (t_soisno[o] <= TFRZ) {
      h2osoi_ice[o] = dz[o] * DENICE * h2osoi_vol[i];
      h2osoi_liq[o] = 0.0;
    } else {
      h2osoi_ice[o] = 0.0;
      h2osoi_liq[o] = dz[o] * DENH2O * h2osoi_vol[i];
    }
  }
}
