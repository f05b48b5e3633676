/* elmo_physics_d.c - soil / snow column temperature: thermal properties, the 21-row pentadiagonal system of
 * snow + standing-surface-water + soil temperature, its solve, and phase change.
 * TEST INFRASTRUCTURE ONLY (see elm_oracle.h).
 *
 * Restates, one function per reference function, on ONE column (level arrays contiguous, [lev]):
 *   src/physics/soil_thermal_properties_impl.hh   src/physics/soil_temperature_impl.hh
 *   src/physics/soil_temp_rhs_impl.hh             src/physics/soil_temp_lhs_impl.hh
 *   src/physics/pentadiagonal_solver_impl.hh      src/physics/phase_change_impl.hh
 *
 * Pinning: the reference has no fixture for this path.  soil_thermal_properties.h, pentadiagonal_solver.h and
 * phase_change.h compile here as shipped and are run against these restatements bit for bit (tests/test_oracle_vs_ref.py).
 * soil_temperature.h, soil_temp_rhs.h and soil_temp_lhs.h include invoke_kernel.hh, whose serial branch names a function
 * that is only defined under ENABLE_KOKKOS; with ONE declaration of that name in front of the includes (the reference's own
 * entity, no body, never instantiated: oracle/ref_harness_soil.cc) they compile, and the WHOLE wrapper - surface heat
 * fluxes, diffusive fluxes, matrix factor, soil_temp::detail::get_rhs_* / get_matrix_* / assemble_*, solve, temperature
 * update, phase change, ground temperature - is run by the reference's own per-column functions against
 * elmo_soil_temperature: right-hand side, banded matrix, heat fluxes and every state field BIT FOR BIT, also chained and with
 * ponded water / melting packs (test_soil_temperature_whole_wrapper_bitwise).  The structural tests
 * (tests/test_soil_temperature_oracle.py) stay as a second, reference-free check.
 */
#include <math.h>

#include "elm_oracle.h"
#include "elmo_const.h"

#define NSNO ELMO_NLEVSNO
#define NGRND ELMO_NLEVGRND
#define NTOT ELMO_NLEVTOT
#define NLEVBED 15 /* elm_constants.h:91 */

/* soil_thermal_properties.h:15-18, soil_thermal_properties_impl.hh:99 */
#define TKICE 2.290
#define TKWAT 0.57
#define TKBDRK 3.0
#define THIN_SFCLAYER 1.0e-6
#define TKAIR 0.023
/* elm_constants.h:40-41 */
#define CPICE 2.11727e3
#define CPWAT 4.188e3
/* soil_temperature.h:169, soil_temperature_impl.hh:104 */
#define CNFAC 0.5
#define CAPR 0.34

/* ---------------- soil_thermal_properties_impl.hh ---------------- */

/* :20-93 */
void elmo_st_calc_soil_tk(int ltype, const double *h2osoi_liq, const double *h2osoi_ice, const double *t_soisno,
                          const double *dz, const double *watsat, const double *tkmg, const double *tkdry, double *thk)
{
  for (int i = NSNO; i < NGRND + NSNO; ++i) {
    if (ltype != istwet && ltype != istice && ltype != istice_mec) {
      double satw = (h2osoi_liq[i] / DENH2O + h2osoi_ice[i] / DENICE) / (dz[i] * watsat[i - NSNO]);
      satw = dmin(1.0, satw);
      if (satw > 1.0e-6) {
        double dke;
        if (t_soisno[i] >= TFRZ) {
          dke = dmax(0.0, log10(satw) + 1.0);
        } else {
          dke = satw;
        }
        const double fl = (h2osoi_liq[i] / (DENH2O * dz[i])) /
                          (h2osoi_liq[i] / (DENH2O * dz[i]) + h2osoi_ice[i] / (DENICE * dz[i]));
        const double dksat =
            tkmg[i - NSNO] * pow(TKWAT, fl * watsat[i - NSNO]) * pow(TKICE, (1.0 - fl) * watsat[i - NSNO]);
        thk[i] = dke * dksat + (1.0 - dke) * tkdry[i - NSNO];
      } else {
        thk[i] = tkdry[i - NSNO];
      }
      if (i >= NSNO + NLEVBED) thk[i] = TKBDRK;
    } else if (ltype == istice || ltype == istice_mec) {
      thk[i] = TKWAT;
      if (t_soisno[i] < TFRZ) thk[i] = TKICE;
    } else if (ltype == istwet) {
      if (i >= NSNO + NLEVBED) {
        thk[i] = TKBDRK;
      } else {
        thk[i] = TKWAT;
        if (t_soisno[i] < TFRZ) thk[i] = TKICE;
      }
    }
  }
}

/* :96-125 */
void elmo_st_calc_snow_tk(int snl, double frac_sno, const double *h2osoi_liq, const double *h2osoi_ice,
                          const double *dz, double *thk)
{
  const int top = NSNO - snl;
  for (int i = 0; i < top; ++i) thk[i] = 0.0;
  for (int i = top; i < NSNO; ++i) {
    const double bw = (h2osoi_ice[i] + h2osoi_liq[i]) / (frac_sno * dz[i]);
    thk[i] = TKAIR + (7.75e-5 * bw + 1.105e-6 * bw * bw) * (TKICE - TKAIR);
  }
}

/* :132-159 */
void elmo_st_calc_face_tk(int snl, const double *thk, const double *z, const double *zi, double *tk)
{
  const int top = NSNO - snl;
  for (int i = 0; i < top; ++i) tk[i] = 0.0;
  const int bot = NGRND + NSNO - 1;
  for (int i = top; i < bot; ++i) {
    tk[i] = thk[i] * thk[i + 1] * (z[i + 1] - z[i]) / (thk[i] * (z[i + 1] - zi[i + 1]) + thk[i + 1] * (zi[i + 1] - z[i]));
  }
  tk[bot] = 0.0;
}

/* :163-202 (csol is indexed with the snow+soil index, watsat with the soil index) */
void elmo_st_calc_soil_heat_capacity(int ltype, int snl, double h2osno, const double *watsat, const double *h2osoi_ice,
                                     const double *h2osoi_liq, const double *dz, const double *csol, double *cv)
{
  for (int i = NSNO; i < NGRND + NSNO; ++i) {
    if (ltype != istwet && ltype != istice && ltype != istice_mec) {
      cv[i] = csol[i] * (1.0 - watsat[i - NSNO]) * dz[i] + (h2osoi_ice[i] * CPICE + h2osoi_liq[i] * CPWAT);
    } else if (ltype == istwet) {
      cv[i] = (h2osoi_ice[i] * CPICE + h2osoi_liq[i] * CPWAT);
      if (i >= NSNO + NLEVBED) cv[i] = csol[i] * dz[i];
    } else if (ltype == istice || ltype == istice_mec) {
      cv[i] = (h2osoi_ice[i] * CPICE + h2osoi_liq[i] * CPWAT);
    }
    if (i == NSNO && snl == 0 && h2osno > 0.0) cv[i] += CPICE * h2osno;
  }
}

/* :206-234 */
void elmo_st_calc_snow_heat_capacity(int snl, double frac_sno, const double *h2osoi_ice, const double *h2osoi_liq,
                                     double *cv)
{
  const int top = NSNO - snl;
  for (int i = 0; i < top; ++i) cv[i] = 0.0;
  for (int i = top; i < NSNO; ++i) {
    if (frac_sno > 0.0) {
      cv[i] = dmax(THIN_SFCLAYER, (CPWAT * h2osoi_liq[i] + CPICE * h2osoi_ice[i]) / frac_sno);
    } else {
      cv[i] = THIN_SFCLAYER;
    }
  }
}

/* :238-251 */
double elmo_st_calc_h2osfc_tk(double h2osfc, const double *thk, const double *z)
{
  const double zh2osfc = 1.0e-3 * (0.5 * h2osfc);
  return TKWAT * thk[NSNO] * (z[NSNO] + zh2osfc) / (TKWAT * z[NSNO] + thk[NSNO] * zh2osfc);
}

/* :255-266 */
double elmo_st_calc_h2osfc_heat_capacity(int snl, double h2osfc, double frac_h2osfc)
{
  (void)snl;
  if ((h2osfc > THIN_SFCLAYER) && (frac_h2osfc > THIN_SFCLAYER)) {
    return dmax(THIN_SFCLAYER, CPWAT * h2osfc / frac_h2osfc);
  } else {
    return THIN_SFCLAYER;
  }
}

/* :269-279 */
double elmo_st_calc_h2osfc_height(int snl, double h2osfc, double frac_h2osfc)
{
  (void)snl;
  if ((h2osfc > THIN_SFCLAYER) && (frac_h2osfc > THIN_SFCLAYER)) {
    return dmax(THIN_SFCLAYER, 1.0e-3 * h2osfc / frac_h2osfc);
  } else {
    return THIN_SFCLAYER;
  }
}

/* ---------------- soil_temperature_impl.hh ---------------- */

/* :79-82, :86-89 */
static double calc_lwrad_emit(double emg, double temp) { return emg * STEBOL * pow(temp, 4.0); }
static double calc_dlwrad_emit(double emg, double t_grnd) { return 4.0 * emg * STEBOL * pow(t_grnd, 3.0); }

/* :13-26 */
double elmo_st_calc_surface_heat_flux(int frac_veg_nosno, double dlrad, double emg, double forc_lwrad, double htvp,
                                      double solar_abg, double temp, double eflx_sh, double qflx_ev)
{
  return solar_abg + dlrad + (1.0 - frac_veg_nosno) * emg * forc_lwrad - calc_lwrad_emit(emg, temp) -
         (eflx_sh + qflx_ev * htvp);
}

/* :28-32 */
double elmo_st_calc_dhsdT(double cgrnd, double emg, double t_grnd) { return -cgrnd - calc_dlwrad_emit(emg, t_grnd); }

/* :34-38 */
double elmo_st_check_absorbed_solar(double frac_sno_eff, double sabg_snow, double sabg_soil)
{
  return frac_sno_eff * sabg_snow + (1.0 - frac_sno_eff) * sabg_soil;
}

/* :45-75 */
void elmo_st_calc_diffusive_heat_flux(int snl, const double *tk, const double *t_soisno, const double *z, double *fn)
{
  const int top = NSNO - snl;
  for (int i = 0; i < top; ++i) fn[i] = 0.0;
  for (int i = top; i < NGRND + NSNO - 1; ++i) fn[i] = tk[i] * (t_soisno[i + 1] - t_soisno[i]) / (z[i + 1] - z[i]);
  fn[NGRND + NSNO - 1] = 0.0;
}

/* :93-121 */
void elmo_st_calc_heat_flux_matrix_factor(int snl, double dtime, const double *cv, const double *dz, const double *z,
                                          const double *zi, double *fact)
{
  const int top = NSNO - snl;
  for (int i = 0; i < top; ++i) fact[i] = 0.0;
  fact[top] = dtime / cv[top] * dz[top] / (0.5 * (z[top] - zi[top] + CAPR * (z[top + 1] - zi[top])));
  for (int i = top + 1; i < NGRND + NSNO; ++i) fact[i] = dtime / cv[i];
}

/* :154-177 */
void elmo_st_update_temperature(int snl, double frac_h2osfc, const double *tvector, double *t_h2osfc, double *t_soisno)
{
  const int top = NSNO - snl;
  for (int i = top; i < NSNO; ++i) t_soisno[i] = tvector[i];
  for (int i = NSNO; i < NSNO + NGRND; ++i) t_soisno[i] = tvector[i + 1];
  *t_h2osfc = (frac_h2osfc != 0.0) ? tvector[NSNO] : t_soisno[NSNO];
}

/* :179-205 */
void elmo_st_update_t_grnd(int snl, double frac_h2osfc, double frac_sno_eff, double t_h2osfc, const double *t_soisno,
                           double *t_grnd)
{
  if (snl > 0) {
    const int top = NSNO - snl;
    if (frac_h2osfc != 0.0) {
      *t_grnd = frac_sno_eff * t_soisno[top] + (1.0 - frac_sno_eff - frac_h2osfc) * t_soisno[NSNO] + frac_h2osfc * t_h2osfc;
    } else {
      *t_grnd = frac_sno_eff * t_soisno[top] + (1.0 - frac_sno_eff) * t_soisno[NSNO];
    }
  } else {
    if (frac_h2osfc != 0.0) {
      *t_grnd = (1.0 - frac_h2osfc) * t_soisno[NSNO] + frac_h2osfc * t_h2osfc;
    } else {
      *t_grnd = t_soisno[NSNO];
    }
  }
}

/* ---------------- soil_temp_rhs_impl.hh: set_RHS (:31-70) = the four detail functions on one column ---------- */
void elmo_st_set_rhs(double dtime, int snl, double hs_top_snow, double dhsdT, double hs_soil, double frac_sno_eff,
                     const double *t_soisno, const double *fact, const double *fn, const double *sabg_lyr,
                     const double *z, double tk_h2osfc, double t_h2osfc, double dz_h2osfc, double c_h2osfc,
                     double hs_h2osfc, double *rhs_vec /*[21]*/)
{
  double rt_snow[NSNO], rt_ssw, rt_soil[NGRND];
  /* get_rhs_snow :77-107 */
  {
    const int top = NSNO - snl;
    for (int i = 0; i < top; ++i) rt_snow[i] = 0.0;
    /* the reference writes rt_snow(c, top) also when snl == 0 (top == nlevsno, one past the snow block of that
       column, i.e. the first snow entry of the next column or past the end of the View): no effect on any result */
    if (top < NSNO) {
      rt_snow[top] = t_soisno[top] + fact[top] * (hs_top_snow - dhsdT * t_soisno[top] + CNFAC * fn[top]);
    }
    for (int i = top + 1; i < NSNO; ++i) {
      rt_snow[i] = t_soisno[i] + CNFAC * fact[i] * (fn[i] - fn[i - 1]) + fact[i] * sabg_lyr[i];
    }
  }
  /* get_rhs_ssw :111-133 */
  {
    const double fn_h2osfc = tk_h2osfc * (t_soisno[NSNO] - t_h2osfc) / (0.5 * dz_h2osfc + z[NSNO]);
    rt_ssw = t_h2osfc + (dtime / c_h2osfc) * (hs_h2osfc - dhsdT * t_h2osfc + CNFAC * fn_h2osfc);
  }
  /* get_rhs_soil :135-177 */
  {
    if (snl == 0) {
      rt_soil[0] = t_soisno[NSNO] + fact[NSNO] * (hs_top_snow - dhsdT * t_soisno[NSNO] + CNFAC * fn[NSNO]);
    } else {
      rt_soil[0] = t_soisno[NSNO] + fact[NSNO] * ((1.0 - frac_sno_eff) * (hs_soil - dhsdT * t_soisno[NSNO]) +
                                                  CNFAC * (fn[NSNO] - frac_sno_eff * fn[NSNO - 1]));
      rt_soil[0] += frac_sno_eff * fact[NSNO] * sabg_lyr[NSNO];
    }
    const int bot = NGRND + NSNO - 1;
    for (int j = NSNO + 1; j < bot; ++j) rt_soil[j - NSNO] = t_soisno[j] + CNFAC * fact[j] * (fn[j] - fn[j - 1]);
    rt_soil[NGRND - 1] = t_soisno[bot] - CNFAC * fact[bot] * fn[bot - 1] + fact[bot] * fn[bot];
  }
  /* assemble_rhs :180-204 */
  for (int i = 0; i < NSNO; ++i) rhs_vec[i] = rt_snow[i];
  rhs_vec[NSNO] = rt_ssw;
  for (int i = 0; i < NGRND; ++i) rhs_vec[i + NSNO + 1] = rt_soil[i];
}

/* ---------------- soil_temp_lhs_impl.hh: set_LHS (:112-158) = the eight detail functions on one column -------- */
void elmo_st_set_lhs(double dtime, int snl, double dz_h2osfc, double c_h2osfc, double tk_h2osfc, double frac_h2osfc,
                     double frac_sno_eff, double dhsdT, const double *z, const double *fact, const double *tk,
                     double *lhs /*[21][5]*/)
{
  double b_snow[NSNO][5], b_soil[NGRND][5], b_ssw[5], b_snow_soil[5], b_ssw_soil[5], b_soil_snow[5], b_soil_ssw[5];
  /* get_matrix_snow :165-203 */
  for (int i = 0; i < NSNO; ++i)
    for (int j = 0; j < 5; ++j) b_snow[i][j] = 0.0;
  if (snl > 0) {
    const int top = NSNO - snl;
    double dzp = z[top + 1] - z[top];
    b_snow[top][3] = 0.0;
    b_snow[top][2] = 1.0 + (1.0 - CNFAC) * fact[top] * tk[top] / dzp - fact[top] * dhsdT;
    if (snl > 1) b_snow[top][1] = -(1.0 - CNFAC) * fact[top] * tk[top] / dzp;
    for (int i = top + 1; i < NSNO; ++i) {
      const double dzm = z[i] - z[i - 1];
      dzp = z[i + 1] - z[i];
      b_snow[i][3] = -(1.0 - CNFAC) * fact[i] * tk[i - 1] / dzm;
      b_snow[i][2] = 1.0 + (1.0 - CNFAC) * fact[i] * (tk[i] / dzp + tk[i - 1] / dzm);
      if (i != NSNO - 1) b_snow[i][1] = -(1.0 - CNFAC) * fact[i] * tk[i] / dzp;
    }
  }
  /* get_matrix_snow_soil :206-227 */
  for (int j = 0; j < 5; ++j) b_snow_soil[j] = 0.0;
  if (snl > 0) b_snow_soil[0] = -(1.0 - CNFAC) * fact[NSNO - 1] * tk[NSNO - 1] / (z[NSNO] - z[NSNO - 1]);
  /* get_matrix_soil :230-292 */
  for (int i = 0; i < NGRND; ++i)
    for (int j = 0; j < 5; ++j) b_soil[i][j] = 0.0;
  if (snl == 0) {
    const double dzp = z[NSNO + 1] - z[NSNO];
    b_soil[0][2] = 1.0 + (1.0 - CNFAC) * fact[NSNO] * tk[NSNO] / dzp - fact[NSNO] * dhsdT;
    b_soil[0][1] = -(1.0 - CNFAC) * fact[NSNO] * tk[NSNO] / dzp;
  } else {
    const double dzm = z[NSNO] - z[NSNO - 1];
    const double dzp = z[NSNO + 1] - z[NSNO];
    b_soil[0][2] = 1.0 + (1.0 - CNFAC) * fact[NSNO] * (tk[NSNO] / dzp + frac_sno_eff * tk[NSNO - 1] / dzm) -
                   (1.0 - frac_sno_eff) * fact[NSNO] * dhsdT;
    b_soil[0][1] = -(1.0 - CNFAC) * fact[NSNO] * tk[NSNO] / dzp;
  }
  for (int i = 1; i < NGRND - 1; ++i) {
    const int offset = i + NSNO;
    const double dzm = z[offset] - z[offset - 1];
    const double dzp = z[offset + 1] - z[offset];
    b_soil[i][3] = -(1.0 - CNFAC) * fact[offset] * tk[offset - 1] / dzm;
    b_soil[i][2] = 1.0 + (1.0 - CNFAC) * fact[offset] * (tk[offset] / dzp + tk[offset - 1] / dzm);
    b_soil[i][1] = -(1.0 - CNFAC) * fact[offset] * tk[offset] / dzp;
  }
  {
    const int bot = NGRND + NSNO - 1;
    double dzm = z[bot] - z[bot - 1];
    b_soil[NGRND - 1][3] = -(1.0 - CNFAC) * fact[bot] * tk[bot - 1] / dzm;
    b_soil[NGRND - 1][2] = 1.0 + (1.0 - CNFAC) * fact[bot] * tk[bot - 1] / dzm;
    b_soil[NGRND - 1][1] = 0.0;
    if (frac_h2osfc != 0.0) {
      dzm = 0.5 * dz_h2osfc + z[NSNO];
      b_soil[0][2] += frac_h2osfc * ((1.0 - CNFAC) * fact[NSNO] * tk_h2osfc / dzm + fact[NSNO] * dhsdT);
    }
  }
  /* get_matrix_soil_snow :295-319 */
  for (int j = 0; j < 5; ++j) b_soil_snow[j] = 0.0;
  if (snl == 0) {
    b_soil_snow[4] = 0.0;
  } else {
    const double dzm = (z[NSNO] - z[NSNO - 1]);
    b_soil_snow[4] = -frac_sno_eff * (1.0 - CNFAC) * fact[NSNO] * tk[NSNO - 1] / dzm;
  }
  /* get_matrix_ssw :322-342 */
  for (int j = 0; j < 5; ++j) b_ssw[j] = 0.0;
  b_ssw[2] = 1.0 + (1.0 - CNFAC) * (dtime / c_h2osfc) * tk_h2osfc / (0.5 * dz_h2osfc + z[NSNO]) -
             (dtime / c_h2osfc) * dhsdT;
  /* get_matrix_ssw_soil :345-364 */
  for (int j = 0; j < 5; ++j) b_ssw_soil[j] = 0.0;
  b_ssw_soil[1] = -(1.0 - CNFAC) * (dtime / c_h2osfc) * tk_h2osfc / (0.5 * dz_h2osfc + z[NSNO]);
  /* get_matrix_soil_ssw :367-390 */
  for (int j = 0; j < 5; ++j) b_soil_ssw[j] = 0.0;
  if (frac_h2osfc != 0.0) {
    b_soil_ssw[3] = -frac_h2osfc * (1.0 - CNFAC) * fact[NSNO] * tk_h2osfc / (0.5 * dz_h2osfc + z[NSNO]);
  }
  /* assemble_lhs :393-481 */
  for (int i = 0; i < NTOT + 1; ++i)
    for (int j = 0; j < 5; ++j) lhs[i * 5 + j] = 0.0;
  for (int bnd = 1; bnd <= 2; ++bnd) lhs[0 * 5 + bnd] = b_snow[0][bnd];
  for (int lev = 1; lev <= 3; ++lev)
    for (int bnd = 1; bnd <= 3; ++bnd) lhs[lev * 5 + bnd] = b_snow[lev][bnd];
  for (int bnd = 2; bnd <= 3; ++bnd) lhs[(NSNO - 1) * 5 + bnd] = b_snow[NSNO - 1][bnd];
  lhs[(NSNO - 1) * 5 + 0] = b_snow_soil[0];
  lhs[NSNO * 5 + 2] = b_ssw[2];
  lhs[NSNO * 5 + 1] = b_ssw_soil[1];
  for (int bnd = 1; bnd <= 2; ++bnd) lhs[(NSNO + 1) * 5 + bnd] = b_soil[0][bnd];
  lhs[(NSNO + 1) * 5 + 3] = b_soil_ssw[3];
  lhs[(NSNO + 1) * 5 + 4] = b_soil_snow[4];
  for (int lev = NSNO + 2; lev < NGRND + NSNO; ++lev)
    for (int bnd = 1; bnd <= 3; ++bnd) lhs[lev * 5 + bnd] = b_soil[lev - 6][bnd];
  for (int bnd = 2; bnd <= 3; ++bnd) lhs[(NSNO + NGRND) * 5 + bnd] = b_soil[NGRND - 1][bnd];
}

/* ---------------- pentadiagonal_solver_impl.hh:16-76 (A, B, Z enter zero-filled, as freshly allocated Views) --- */
void elmo_st_pdma(int snl, const double *LHS /*[21][5]*/, double *A /*[20]*/, double *B /*[19]*/, double *Z /*[21]*/,
                  double *RHS /*[21]*/)
{
#define L(i, b) LHS[(i)*5 + (b)]
  const int N = NTOT + 1;
  const int top = NSNO - snl;
  double U1 = 1.0 / L(top, 2);
  A[top] = L(top, 1) * U1;
  B[top] = L(top, 0) * U1;
  Z[top] = RHS[top] * U1;
  double Y1 = L(top + 1, 3);
  U1 = 1.0 / (L(top + 1, 2) - A[top] * Y1);
  A[top + 1] = (L(top + 1, 1) - B[top] * Y1) * U1;
  B[top + 1] = L(top + 1, 0) * U1;
  Z[top + 1] = (RHS[top + 1] - Z[top] * Y1) * U1;
  for (int i = top + 2; i < N - 2; ++i) {
    Y1 = L(i, 3) - A[i - 2] * L(i, 4);
    U1 = 1.0 / (L(i, 2) - B[i - 2] * L(i, 4) - A[i - 1] * Y1);
    A[i] = (L(i, 1) - B[i - 1] * Y1) * U1;
    B[i] = L(i, 0) * U1;
    Z[i] = (RHS[i] - Z[i - 2] * L(i, 4) - Z[i - 1] * Y1) * U1;
  }
  Y1 = L(N - 2, 3) - A[N - 4] * L(N - 2, 4);
  U1 = 1.0 / (L(N - 2, 2) - B[N - 4] * L(N - 2, 4) - A[N - 3] * Y1);
  A[N - 2] = (L(N - 2, 1) - B[N - 3] * Y1) * U1;
  const double Y2 = L(N - 1, 3) - A[N - 3] * L(N - 1, 4);
  const double U2 = 1.0 / (L(N - 1, 2) - B[N - 3] * L(N - 1, 4) - A[N - 2] * Y2);
  Z[N - 2] = (RHS[N - 2] - Z[N - 3] * L(N - 2, 4) - Z[N - 3] * Y1) * U1;
  Z[N - 1] = (RHS[N - 1] - Z[N - 2] * L(N - 1, 4) - Z[N - 2] * Y2) * U2;
  RHS[N - 1] = Z[N - 1];
  RHS[N - 2] = Z[N - 2] - A[N - 2] * RHS[N - 1];
  for (int i = N - 3; i >= 0; --i) RHS[i] = Z[i] - A[i] * RHS[i + 1] - B[i] * RHS[i + 2];
#undef L
}

/* ---------------- phase_change_impl.hh ---------------- */

/* :11-151 */
void elmo_st_phase_change_h2osfc(int snl, double dtime, double frac_sno, double frac_h2osfc, double dhsdT,
                                 double c_h2osfc, double fact_sl1, double *t_h2osfc, double *h2osfc,
                                 double *xmf_h2osfc, double *qflx_h2osfc_to_ice, double *eflx_h2osfc_to_snow,
                                 double *h2osno, double *int_snow, double *snow_depth, double *h2osoi_ice_sl1,
                                 double *t_soisno_sl1)
{
  *qflx_h2osfc_to_ice = 0.0;
  *eflx_h2osfc_to_snow = 0.0;
  *xmf_h2osfc = 0.0;
  if (frac_h2osfc > 0.0 && *t_h2osfc <= TFRZ) {
    const double tinc = TFRZ - *t_h2osfc;
    *t_h2osfc = TFRZ;
    const double hm = frac_h2osfc * (dhsdT * tinc - tinc * c_h2osfc / dtime);
    const double xm = hm * dtime / HFUS;
    const double temp1 = *h2osfc + xm;
    const double z_avg = frac_sno * *snow_depth;
    double rho_avg;
    if (z_avg > 0.0) {
      rho_avg = dmin(800.0, *h2osno / z_avg);
    } else {
      rho_avg = 200.0;
    }
    if (temp1 >= 0.0) {
      *h2osno -= xm;
      *int_snow -= xm;
      if (snl > 0) *h2osoi_ice_sl1 -= xm;
      *h2osfc += xm;
      *xmf_h2osfc = hm;
      *qflx_h2osfc_to_ice = -xm / dtime;
      if (frac_sno > 0 && snl > 0) {
        *snow_depth = *h2osno / (rho_avg * frac_sno);
      } else {
        *snow_depth = *h2osno / DENICE;
      }
      if (snl == 0) {
        *t_soisno_sl1 = *t_h2osfc;
        *eflx_h2osfc_to_snow = 0.0;
      } else {
        double c1, c2;
        if (snl == 1) {
          c1 = frac_sno * (dtime / fact_sl1 - dhsdT * dtime);
        } else {
          c1 = frac_sno / fact_sl1 * dtime;
        }
        if (frac_h2osfc != 0.0) {
          c2 = (-CPWAT * xm - frac_h2osfc * dhsdT * dtime);
        } else {
          c2 = 0.0;
        }
        *t_soisno_sl1 = (c1 * *t_soisno_sl1 + c2 * *t_h2osfc) / (c1 + c2);
        *eflx_h2osfc_to_snow = (*t_h2osfc - *t_soisno_sl1) * c2 / dtime;
      }
    } else {
      rho_avg = (*h2osno * rho_avg + *h2osfc * DENICE) / (*h2osno + *h2osfc);
      *h2osno += *h2osfc;
      *int_snow += *h2osfc;
      *qflx_h2osfc_to_ice = *h2osfc / dtime;
      if (snl > 0) *h2osoi_ice_sl1 = *h2osoi_ice_sl1 + *h2osfc;
      *t_h2osfc = *t_h2osfc - temp1 * HFUS / (dtime * dhsdT - c_h2osfc);
      *xmf_h2osfc = hm - frac_h2osfc * temp1 * HFUS / dtime;
      double c1, c2;
      if (snl == 0) {
        *t_soisno_sl1 = *t_h2osfc;
      } else if (snl == 1) {
        c1 = frac_sno * (dtime / fact_sl1 - dhsdT * dtime);
        if (frac_h2osfc != 0.0) {
          c2 = frac_h2osfc * (c_h2osfc - dtime * dhsdT);
        } else {
          c2 = 0.0;
        }
        *t_soisno_sl1 = (c1 * *t_soisno_sl1 + c2 * *t_h2osfc) / (c1 + c2);
        *t_h2osfc = *t_soisno_sl1;
      } else {
        c1 = frac_sno / fact_sl1 * dtime;
        if (frac_h2osfc != 0.0) {
          c2 = frac_h2osfc * (c_h2osfc - dtime * dhsdT);
        } else {
          c2 = 0.0;
        }
        *t_soisno_sl1 = (c1 * *t_soisno_sl1 + c2 * *t_h2osfc) / (c1 + c2);
        *t_h2osfc = *t_soisno_sl1;
      }
      *h2osfc = 0.0;
      if (frac_sno > 0.0 && snl > 0) {
        *snow_depth = *h2osno / (rho_avg * frac_sno);
      } else {
        *snow_depth = *h2osno / DENICE;
      }
    }
  }
}

/* :182-418 */
void elmo_st_phase_change_soisno(int snl, int ltype, double dtime, double dhsdT, double frac_h2osfc,
                                 double frac_sno_eff, const double *fact, const double *watsat, const double *sucsat,
                                 const double *bsw, const double *dz, double *h2osno, double *snow_depth, double *xmf,
                                 double *qflx_snofrz, double *qflx_snow_melt, double *qflx_snomelt,
                                 double *eflx_snomelt, int *imelt, double *qflx_snofrz_lyr, double *h2osoi_ice,
                                 double *h2osoi_liq, double *t_soisno)
{
  *xmf = 0.0;
  *qflx_snofrz = 0.0;
  *qflx_snow_melt = 0.0;
  *qflx_snomelt = 0.0;
  for (int i = 0; i < NSNO; ++i) qflx_snofrz_lyr[i] = 0.0;
  const int top = NSNO - snl;
  for (int i = top; i < NSNO + NGRND; ++i) imelt[i] = 0;
  double tinc[NTOT];
  double supercool[NGRND];
  for (int i = 0; i < NTOT; ++i) tinc[i] = 0.0; /* (uninitialised in the reference; only read where assigned) */

  for (int i = top; i < NSNO; ++i) {
    if (h2osoi_ice[i] > 0.0 && t_soisno[i] > TFRZ) {
      imelt[i] = 1;
      tinc[i] = TFRZ - t_soisno[i];
      t_soisno[i] = TFRZ;
    }
    if (h2osoi_liq[i] > 0.0 && t_soisno[i] < TFRZ) {
      imelt[i] = 2;
      tinc[i] = TFRZ - t_soisno[i];
      t_soisno[i] = TFRZ;
    }
  }
  for (int i = NSNO; i < NSNO + NGRND; ++i) {
    if (h2osoi_ice[i] > 0.0 && t_soisno[i] > TFRZ) {
      imelt[i] = 1;
      tinc[i] = TFRZ - t_soisno[i];
      t_soisno[i] = TFRZ;
    }
    supercool[i - NSNO] = 0.0;
    if (ltype == istsoil || ltype == istcrop || ltype == icol_road_perv) {
      if (t_soisno[i] < TFRZ) {
        const double smp = HFUS * (TFRZ - t_soisno[i]) / (GRAV * t_soisno[i]) * 1000.0;
        supercool[i - NSNO] = watsat[i - NSNO] * pow(smp / sucsat[i - NSNO], -1.0 / bsw[i - NSNO]);
        supercool[i - NSNO] *= dz[i] * 1000.0;
      }
    }
    if (h2osoi_liq[i] > supercool[i - NSNO] && t_soisno[i] < TFRZ) {
      imelt[i] = 2;
      tinc[i] = TFRZ - t_soisno[i];
      t_soisno[i] = TFRZ;
    }
    if (snl == 0 && *h2osno > 0.0 && i == NSNO) {
      if (t_soisno[i] > TFRZ) {
        imelt[i] = 1;
        tinc[i] = TFRZ - t_soisno[i];
        t_soisno[i] = TFRZ;
      }
    }
  }

  for (int i = top; i < NSNO + NGRND; ++i) {
    double hm = 0.0;
    if (imelt[i] > 0) {
      if (i == top) {
        if (i < NSNO) {
          hm = frac_sno_eff * (dhsdT * tinc[i] - tinc[i] / fact[i]);
        } else {
          const double temp_hm = dhsdT * tinc[i] - tinc[i] / fact[i];
          hm = (frac_h2osfc != 0.0) ? temp_hm - frac_h2osfc * (dhsdT * tinc[i]) : temp_hm;
        }
      } else if (i == NSNO) {
        hm = (1.0 - frac_sno_eff - frac_h2osfc) * dhsdT * tinc[i] - tinc[i] / fact[i];
      } else {
        if (i < NSNO) {
          hm = -frac_sno_eff * (tinc[i] / fact[i]);
        } else {
          hm = -tinc[i] / fact[i];
        }
      }
    }
    if (imelt[i] == 1 && hm < 0.0) {
      hm = 0.0;
      imelt[i] = 0;
    }
    if (imelt[i] == 2 && hm > 0.0) {
      hm = 0.0;
      imelt[i] = 0;
    }
    if (imelt[i] > 0 && fabs(hm) > 0.0) {
      double xm = hm * dtime / HFUS;
      if (i == NSNO) {
        if (snl == 0 && *h2osno > 0.0 && xm > 0.0) {
          const double temp1 = *h2osno;
          *h2osno = dmax(0.0, temp1 - xm);
          const double propor = *h2osno / temp1;
          *snow_depth *= propor;
          const double heatr = hm - HFUS * (temp1 - *h2osno) / dtime;
          if (heatr > 0.0) {
            xm = heatr * dtime / HFUS;
            hm = heatr;
          } else {
            xm = 0.0;
            hm = 0.0;
          }
          *qflx_snomelt = dmax(0.0, temp1 - *h2osno) / dtime;
          *xmf = HFUS * *qflx_snomelt;
          *qflx_snow_melt = *qflx_snomelt;
        }
      }
      double heatr = 0.0;
      const double wmass0 = h2osoi_ice[i] + h2osoi_liq[i];
      const double wice0 = h2osoi_ice[i];
      if (xm > 0.0) {
        h2osoi_ice[i] = dmax(0.0, wice0 - xm);
        heatr = hm - HFUS * (wice0 - h2osoi_ice[i]) / dtime;
      } else if (xm < 0.0) {
        if (i < NSNO) {
          h2osoi_ice[i] = dmin(wmass0, wice0 - xm);
        } else {
          if (wmass0 < supercool[i - NSNO]) {
            h2osoi_ice[i] = 0.0;
          } else {
            h2osoi_ice[i] = dmin(wmass0 - supercool[i - NSNO], wice0 - xm);
          }
        }
        heatr = hm - HFUS * (wice0 - h2osoi_ice[i]) / dtime;
      }
      h2osoi_liq[i] = dmax(0.0, wmass0 - h2osoi_ice[i]);
      if (fabs(heatr) > 0.0) {
        if (i == top) {
          if (snl == 0) {
            t_soisno[i] += fact[i] * heatr / (1.0 - (1.0 - frac_h2osfc) * fact[i] * dhsdT);
          } else {
            t_soisno[i] += (fact[i] / frac_sno_eff) * heatr / (1.0 - fact[i] * dhsdT);
          }
        } else if (i == NSNO) {
          t_soisno[i] += fact[i] * heatr / (1.0 - (1.0 - frac_sno_eff - frac_h2osfc) * fact[i] * dhsdT);
        } else {
          if (i >= NSNO) {
            t_soisno[i] += fact[i] * heatr;
          } else {
            if (frac_sno_eff > 0.0) t_soisno[i] += (fact[i] / frac_sno_eff) * heatr;
          }
        }
        if (i < NSNO) {
          if (h2osoi_liq[i] * h2osoi_ice[i] > 0.0) t_soisno[i] = TFRZ;
        }
      }
      *xmf += HFUS * (wice0 - h2osoi_ice[i]) / dtime;
      if (imelt[i] == 1 && i < NSNO) *qflx_snomelt += dmax(0.0, (wice0 - h2osoi_ice[i])) / dtime;
      if (imelt[i] == 2 && i < NSNO) qflx_snofrz_lyr[i] = dmax(0.0, (h2osoi_ice[i] - wice0)) / dtime;
    }
  }
  *eflx_snomelt = *qflx_snomelt * HFUS;
  for (int i = 0; i < NSNO; ++i) {
    if (imelt[i] == 2 && i < NSNO) *qflx_snofrz += qflx_snofrz_lyr[i];
  }
}
