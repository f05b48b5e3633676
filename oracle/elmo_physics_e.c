/* elmo_physics_e.c - surface fluxes after the temperature solve, and the conservation diagnostics.
 * TEST INFRASTRUCTURE ONLY (see elm_oracle.h).
 *
 * Restates src/physics/surface_fluxes_impl.hh and src/physics/conserved_quantity_evaluators_impl.hh, one function
 * per reference function.  Both headers compile here, so every function is pinned bit for bit against the reference
 * itself (tests/test_oracle_vs_ref.py); the reference has no fixture for them.
 * Reference quirks kept: pow(t_h2osfc_bef, 40) in the ground heat flux (:177, an obvious 4.0 typo),
 * pow(emg * sb * t_grnd0, 3.0) * (4.0 * tinc) (:182), (t_h2osfc / dtime) in the soil energy balance (:273), the
 * shadowed snow_sources / snow_sinks (conserved_quantity_evaluators_impl.hh:52-57).
 */
#include <math.h>

#include "elm_oracle.h"
#include "elmo_const.h"

#define NSNO ELMO_NLEVSNO
#define NGRND ELMO_NLEVGRND
#define NLEVURB 5 /* elm_constants.h */

/* surface_fluxes_impl.hh:8-19 */
static double prev_tgrnd(int snl, double frac_sno_eff, double frac_h2osfc, double t_h2osfc_bef, double tssbef_snotop,
                         double tssbef_soitop)
{
  if (snl > 0) {
    return frac_sno_eff * tssbef_snotop + (1.0 - frac_sno_eff - frac_h2osfc) * tssbef_soitop + frac_h2osfc * t_h2osfc_bef;
  } else {
    return (1.0 - frac_h2osfc) * tssbef_soitop + frac_h2osfc * t_h2osfc_bef;
  }
}

/* :31-45 */
static double evap_ratio(double h2osoi_ice_snotop, double h2osoi_liq_snotop, double dtime, double qflx_evap_soi)
{
  double egsmax = (h2osoi_ice_snotop + h2osoi_liq_snotop) / dtime;
  if (egsmax < 0.0) egsmax = 0.0;
  if (qflx_evap_soi > egsmax) {
    return egsmax / qflx_evap_soi;
  } else {
    return 1.0;
  }
}

/* :73-95 */
void elmo_sf_initial_flux_calc(int urbpoi, int snl, double frac_sno_eff, double frac_h2osfc, double t_h2osfc_bef,
                               double tssbef_snotop, double tssbef_soitop, double t_grnd, double cgrnds, double cgrndl,
                               double *eflx_sh_grnd, double *qflx_evap_soi, double *qflx_ev_snow, double *qflx_ev_soil,
                               double *qflx_ev_h2osfc)
{
  const double t_grnd0 = prev_tgrnd(snl, frac_sno_eff, frac_h2osfc, t_h2osfc_bef, tssbef_snotop, tssbef_soitop);
  const double tinc = t_grnd - t_grnd0;
  *eflx_sh_grnd += tinc * cgrnds;
  *qflx_evap_soi += tinc * cgrndl;
  if (!urbpoi) {
    *qflx_ev_snow += tinc * cgrndl;
    *qflx_ev_soil += tinc * cgrndl;
    *qflx_ev_h2osfc += tinc * cgrndl;
  } else {
    *qflx_ev_snow = *qflx_evap_soi;
    *qflx_ev_soil = 0.0;
    *qflx_ev_h2osfc = 0.0;
  }
}

/* :147-238 */
void elmo_sf_update_surface_fluxes(int urbpoi, int do_capsnow, int snl, double dtime, double t_grnd, double htvp,
                                   double frac_sno_eff, double frac_h2osfc, double t_h2osfc_bef, double sabg_soil,
                                   double sabg_snow, double dlrad, double frac_veg_nosno, double emg, double forc_lwrad,
                                   double tssbef_snotop, double tssbef_soitop, double h2osoi_ice_snotop,
                                   double h2osoi_liq_snotop, double eflx_sh_veg, double qflx_evap_veg,
                                   double *qflx_evap_soi, double *eflx_sh_grnd, double *qflx_ev_snow,
                                   double *qflx_ev_soil, double *qflx_ev_h2osfc, double *eflx_soil_grnd,
                                   double *eflx_sh_tot, double *qflx_evap_tot, double *eflx_lh_tot, double *qflx_evap_grnd,
                                   double *qflx_sub_snow, double *qflx_dew_snow, double *qflx_dew_grnd,
                                   double *qflx_snwcp_liq, double *qflx_snwcp_ice)
{
  const double egirat = evap_ratio(h2osoi_ice_snotop, h2osoi_liq_snotop, dtime, *qflx_evap_soi);
  if (egirat < 1.0) {
    const double save_qflx_evap_soi = *qflx_evap_soi;
    *qflx_evap_soi *= egirat;
    *eflx_sh_grnd += (save_qflx_evap_soi - *qflx_evap_soi) * htvp;
    *qflx_ev_snow *= egirat;
    *qflx_ev_soil *= egirat;
    *qflx_ev_h2osfc *= egirat;
  }
  if (!urbpoi) {
    const double lw_grnd = (frac_sno_eff * pow(tssbef_snotop, 4.0) +
                            (1.0 - frac_sno_eff - frac_h2osfc) * pow(tssbef_soitop, 4.0) + frac_h2osfc * pow(t_h2osfc_bef, 40));
    const double t_grnd0 = prev_tgrnd(snl, frac_sno_eff, frac_h2osfc, t_h2osfc_bef, tssbef_snotop, tssbef_soitop);
    const double tinc = t_grnd - t_grnd0;
    *eflx_soil_grnd = ((1.0 - frac_sno_eff) * sabg_soil + frac_sno_eff * sabg_snow) + dlrad +
                      (1.0 - frac_veg_nosno) * emg * forc_lwrad - emg * STEBOL * lw_grnd -
                      pow(emg * STEBOL * t_grnd0, 3.0) * (4.0 * tinc) - (*eflx_sh_grnd + *qflx_evap_soi * htvp);
  }
  *eflx_sh_tot = eflx_sh_veg + *eflx_sh_grnd;
  *qflx_evap_tot = qflx_evap_veg + *qflx_evap_soi;
  *eflx_lh_tot = HVAP * qflx_evap_veg + htvp * *qflx_evap_soi;
  *qflx_evap_grnd = 0.0;
  *qflx_sub_snow = 0.0;
  *qflx_dew_snow = 0.0;
  *qflx_dew_grnd = 0.0;
  if (*qflx_ev_snow >= 0.0) {
    if ((h2osoi_liq_snotop + h2osoi_ice_snotop) > 0.0) {
      *qflx_evap_grnd = dmax(*qflx_ev_snow * (h2osoi_liq_snotop / (h2osoi_liq_snotop + h2osoi_ice_snotop)), 0.0);
    } else {
      *qflx_evap_grnd = 0.0;
    }
    *qflx_sub_snow = *qflx_ev_snow - *qflx_evap_grnd;
  } else {
    if (t_grnd < TFRZ) {
      *qflx_dew_snow = fabs(*qflx_ev_snow);
    } else {
      *qflx_dew_grnd = fabs(*qflx_ev_snow);
    }
  }
  if (snl > 0 && do_capsnow) {
    *qflx_snwcp_liq = *qflx_snwcp_liq + frac_sno_eff * *qflx_dew_grnd;
    *qflx_snwcp_ice = *qflx_snwcp_ice + frac_sno_eff * *qflx_dew_snow;
  }
}

/* :247-265 */
void elmo_sf_lwrad_outgoing(int urbpoi, int snl, int frac_veg_nosno, double forc_lwrad, double frac_sno_eff,
                            double tssbef_snotop, double tssbef_soitop, double frac_h2osfc, double t_h2osfc_bef,
                            double t_grnd, double ulrad, double emg, double *eflx_lwrad_out, double *eflx_lwrad_net)
{
  if (!urbpoi) {
    const double lw_grnd = (frac_sno_eff * pow(tssbef_snotop, 4.0) +
                            (1.0 - frac_sno_eff - frac_h2osfc) * pow(tssbef_soitop, 4.0) + frac_h2osfc * pow(t_h2osfc_bef, 4.0));
    const double t_grnd0 = prev_tgrnd(snl, frac_sno_eff, frac_h2osfc, t_h2osfc_bef, tssbef_snotop, tssbef_soitop);
    const double tinc = t_grnd - t_grnd0;
    *eflx_lwrad_out = ulrad + (1 - frac_veg_nosno) * (1.0 - emg) * forc_lwrad +
                      (1 - frac_veg_nosno) * emg * STEBOL * lw_grnd + 4.0 * emg * STEBOL * pow(t_grnd0, 3.0) * tinc;
    *eflx_lwrad_net = *eflx_lwrad_out - forc_lwrad;
  }
}

/* :268-294 */
double elmo_sf_soil_energy_balance(int ctype, int snl, double eflx_soil_grnd, double xmf, double xmf_h2osfc,
                                   double frac_h2osfc, double t_h2osfc, double t_h2osfc_bef, double dtime,
                                   double eflx_h2osfc_to_snow, double frac_sno_eff, const double *t_soisno,
                                   const double *tssbef, const double *fact)
{
  const double eflx_building_heat = 0.0;
  double errsoi = eflx_soil_grnd - xmf - xmf_h2osfc - frac_h2osfc * (t_h2osfc - t_h2osfc_bef) * (t_h2osfc / dtime);
  errsoi += eflx_h2osfc_to_snow;
  if (ctype == icol_sunwall || ctype == icol_shadewall || ctype == icol_roof) errsoi += eflx_building_heat;
  for (int j = 0; j < NGRND + NSNO; ++j) {
    if ((ctype != icol_sunwall && ctype != icol_shadewall && ctype != icol_roof) || (j < NLEVURB)) {
      if (j >= NSNO - snl && j < NSNO) errsoi -= frac_sno_eff * (t_soisno[j] - tssbef[j]) / fact[j];
      if (j >= NSNO) errsoi -= (t_soisno[j] - tssbef[j]) / fact[j];
    }
  }
  return errsoi;
}

/* ---------------- conserved_quantity_evaluators_impl.hh ---------------- */

/* :7-15 */
double elmo_ce_column_water_mass(double h2ocan, double h2osno, double h2osfc, const double *h2osoi_ice,
                                 const double *h2osoi_liq)
{
  double water = h2ocan + h2osno + h2osfc;
  for (int i = 0; i < NGRND + NSNO; ++i) water += h2osoi_ice[i] + h2osoi_liq[i];
  return water;
}

/* :19-22 */
double elmo_ce_dh2o_dt(double begwb, double endwb, double dtime) { return (endwb - begwb) / dtime; }

/* :26-33 */
double elmo_ce_column_water_balance_error(double begwb, double endwb, double hydrology_source_sink, double forc_rain,
                                          double forc_snow, double qflx_evap_tot, double qflx_snwcp_ice, double dtime)
{
  return (endwb - begwb) - (forc_rain + forc_snow - hydrology_source_sink - qflx_evap_tot - qflx_snwcp_ice) * dtime;
}

/* :37-70 */
double elmo_ce_snow_water_balance_error(int snl, double qflx_dew_snow, double qflx_dew_grnd, double qflx_sub_snow,
                                        double qflx_evap_grnd, double qflx_snow_melt, double qflx_snwcp_ice,
                                        double qflx_snwcp_liq, double qflx_sl_top_soil, double frac_sno_eff,
                                        double qflx_rain_grnd, double qflx_snow_grnd, double qflx_h2osfc_ice,
                                        double h2osno, double h2osno_old, double dtime, int do_capsnow)
{
  if (snl > 0) {
    double snow_sources, snow_sinks; /* the inner (shadowing) pair of the reference; its first values are overwritten */
    if (do_capsnow) {
      snow_sources = frac_sno_eff * (qflx_dew_snow + qflx_dew_grnd) + qflx_h2osfc_ice + qflx_snow_grnd + qflx_rain_grnd;
      snow_sinks = frac_sno_eff * (qflx_sub_snow + qflx_evap_grnd) + qflx_snwcp_ice + qflx_snwcp_liq + qflx_snow_melt +
                   qflx_sl_top_soil;
    } else {
      const double qflx_snow_h2osfc = 0.0;
      snow_sources = (qflx_snow_grnd - qflx_snow_h2osfc) + frac_sno_eff * (qflx_rain_grnd + qflx_dew_snow + qflx_dew_grnd) +
                     qflx_h2osfc_ice;
      snow_sinks = frac_sno_eff * (qflx_sub_snow + qflx_evap_grnd) + qflx_snow_melt + qflx_sl_top_soil;
    }
    return (h2osno - h2osno_old) - (snow_sources - snow_sinks) * dtime;
  } else {
    return 0.0;
  }
}

/* :73-83, :86-95, :98-106, :109-113 */
double elmo_ce_solar_shortwave_balance_error(double fsa, double fsr, const double *forc_solad, const double *forc_solai)
{
  return fsa + fsr - (forc_solad[0] + forc_solad[1] + forc_solai[0] + forc_solai[1]);
}
double elmo_ce_solar_longwave_balance_error(double eflx_lwrad_out, double eflx_lwrad_net, double forc_lwrad)
{
  return eflx_lwrad_out - eflx_lwrad_net - forc_lwrad;
}
double elmo_ce_surface_energy_balance_error(double sabv, double sabg_chk, double forc_lwrad, double eflx_lwrad_out,
                                            double eflx_sh_tot, double eflx_lh_tot, double eflx_soil_grnd)
{
  return sabv + sabg_chk + forc_lwrad - eflx_lwrad_out - eflx_sh_tot - eflx_lh_tot - eflx_soil_grnd;
}
double elmo_ce_net_radiation(double fsa, double eflx_lwrad_net) { return fsa - eflx_lwrad_net; }
