// k_soil_temperature.hip - kokkos_soil_temperature (driver/kokkos/soil_temperature_kokkos.cc:6-278): the only vertical
// solve of the reference, called right after the seven water+energy wrappers (elm_kokkos_interface.cc:310).
//
//   soil_thermal::calc_soil_tk :20, calc_snow_tk :96, calc_face_tk :132, calc_soil_heat_capacity :163,
//   calc_snow_heat_capacity :206, calc_h2osfc_tk :238, _heat_capacity :255, _height :269
//                                                       (src/physics/soil_thermal_properties_impl.hh)
//   soil_temp::calc_surface_heat_flux :13, calc_dhsdT :28, check_absorbed_solar :34, calc_diffusive_heat_flux :45,
//   calc_heat_flux_matrix_factor :93, update_temperature :154, update_t_grnd :179
//                                                       (src/physics/soil_temperature_impl.hh)
//   soil_temp::set_RHS (soil_temp_rhs_impl.hh:31-204), set_LHS (soil_temp_lhs_impl.hh:112-481)
//   solver::PDMA (pentadiagonal_solver_impl.hh:16-76)
//   soil_temp::phase_change_h2osfc :11, phase_change_soisno :182   (src/physics/phase_change_impl.hh)
//
// The reference runs nine parallel_for launches and materialises tk, cv, fn, rhs[21], lhs[21][5], A, B, Z and ten
// more scratch Views in memory.  Here one thread owns one column: the 21-row, 5-band system (snow layers, standing
// surface water, soil layers) is assembled, solved and applied in registers, every array statically indexed.
// The number of active snow layers differs from lane to lane, so loops that the reference starts at the top active
// layer run over all rows with a predicate, and rows above the snow pack enter the forward sweep as identity rows:
// with their A, B, Z equal to zero the general recurrence reproduces the reference's special first and second rows
// exactly (x - 0*y == x), so the solve is bit-identical to the reference's whatever the layer count.
#include "elmk_dev.h"
#include "elmk_kernels.h"

namespace elmk {

#define LV(f, lev) S->f[(int64_t)(lev) * ld + c]

constexpr int NLEVBED = 15;               // elm_constants.h:91
constexpr double ST_TKICE = 2.290;        // soil_thermal_properties.h:15-18
constexpr double ST_TKWAT = 0.57;
constexpr double ST_TKBDRK = 3.0;
constexpr double ST_THIN_SFCLAYER = 1.0e-6;
constexpr double ST_TKAIR = 0.023;        // soil_thermal_properties_impl.hh:106
constexpr double ST_CPICE = 2.11727e3;    // elm_constants.h:40-41
constexpr double ST_CPWAT = 4.188e3;
constexpr double ST_CNFAC = 0.5;          // soil_temperature.h:169
constexpr double ST_CAPR = 0.34;          // soil_temperature_impl.hh:104
constexpr int NROW = NLEVTOT + 1;         // snow + standing surface water + soil

// soil_temperature_impl.hh:13-26
__device__ __forceinline__ double st_surface_heat_flux(int frac_veg_nosno, double dlrad, double emg, double forc_lwrad,
                                                       double htvp, double solar_abg, double temp, double eflx_sh,
                                                       double qflx_ev)
{
  return solar_abg + dlrad + (1.0 - frac_veg_nosno) * emg * forc_lwrad - emg * STEBOL * pow(temp, 4.0) -
         (eflx_sh + qflx_ev * htvp);
}

__global__ __launch_bounds__(256) void k_soil_temperature(const DevState* __restrict__ S, double dtime)
{
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t ld = S->ld;
  if (c >= S->ncols) return;
  const int ltype = 1;  // the wrapper's "dummy ltype" (soil_temperature_kokkos.cc:77-79)
  const int snl = S->snl[c];
  const int top = NLEVSNO - snl;
  const double frac_sno = S->frac_sno[c], frac_sno_eff = S->frac_sno_eff[c], frac_h2osfc = S->frac_h2osfc[c];
  const double h2osfc0 = S->h2osfc[c], h2osno0 = S->h2osno[c];

  double t[NLEVTOT], z[NLEVTOT], dzl[NLEVTOT], liq[NLEVTOT], ice[NLEVTOT];
#pragma unroll
  for (int i = 0; i < NLEVTOT; i++) {
    t[i] = LV(t_soisno, i);
    z[i] = LV(zsoi, i);
    dzl[i] = LV(dz, i);
    liq[i] = LV(h2osoi_liq, i);
    ice[i] = LV(h2osoi_ice, i);
  }

  // ---- soil_thermal_props (:92-104): layer conductivity thk, interface conductivity tk, heat capacity cv
  double thk[NLEVTOT], cv[NLEVTOT];
#pragma unroll
  for (int i = NLEVSNO; i < NLEVTOT; i++) {  // calc_soil_tk, calc_soil_heat_capacity (ltype == 1: soil branch)
    const double watsat = LV(watsat, i - NLEVSNO);
    double satw = (liq[i] / DENH2O + ice[i] / DENICE) / (dzl[i] * watsat);
    satw = dmin(1.0, satw);
    const double tkdry = LV(tkdry, i - NLEVSNO);
    if (satw > 1.0e-6) {
      double dke;
      if (t[i] >= TFRZ) {
        dke = dmax(0.0, log10(satw) + 1.0);
      } else {
        dke = satw;
      }
      const double fl = (liq[i] / (DENH2O * dzl[i])) / (liq[i] / (DENH2O * dzl[i]) + ice[i] / (DENICE * dzl[i]));
      const double dksat = LV(tkmg, i - NLEVSNO) * pow(ST_TKWAT, fl * watsat) * pow(ST_TKICE, (1.0 - fl) * watsat);
      thk[i] = dke * dksat + (1.0 - dke) * tkdry;
    } else {
      thk[i] = tkdry;
    }
    if (i >= NLEVSNO + NLEVBED) thk[i] = ST_TKBDRK;
    cv[i] = LV(csol, i) * (1.0 - watsat) * dzl[i] + (ice[i] * ST_CPICE + liq[i] * ST_CPWAT);
    if (i == NLEVSNO && snl == 0 && h2osno0 > 0.0) cv[i] += ST_CPICE * h2osno0;
  }
#pragma unroll
  for (int i = 0; i < NLEVSNO; i++) {  // calc_snow_tk, calc_snow_heat_capacity
    if (i < top) {
      thk[i] = 0.0;
      cv[i] = 0.0;
    } else {
      const double bw = (ice[i] + liq[i]) / (frac_sno * dzl[i]);
      thk[i] = ST_TKAIR + (7.75e-5 * bw + 1.105e-6 * bw * bw) * (ST_TKICE - ST_TKAIR);
      if (frac_sno > 0.0) {
        cv[i] = dmax(ST_THIN_SFCLAYER, (ST_CPWAT * liq[i] + ST_CPICE * ice[i]) / frac_sno);
      } else {
        cv[i] = ST_THIN_SFCLAYER;
      }
    }
  }
  (void)ltype;
  double tk[NLEVTOT];  // calc_face_tk: tk[i] is the interface between cells i and i+1
#pragma unroll
  for (int i = 0; i < NLEVTOT - 1; i++) {
    if (i < top) {
      tk[i] = 0.0;
    } else {
      const double zi1 = LV(zisoi, i + 1);
      tk[i] = thk[i] * thk[i + 1] * (z[i + 1] - z[i]) / (thk[i] * (z[i + 1] - zi1) + thk[i + 1] * (zi1 - z[i]));
    }
  }
  tk[NLEVTOT - 1] = 0.0;
  double tk_h2osfc, c_h2osfc, dz_h2osfc;
  {
    const double zh2osfc = 1.0e-3 * (0.5 * h2osfc0);
    tk_h2osfc = ST_TKWAT * thk[NLEVSNO] * (z[NLEVSNO] + zh2osfc) / (ST_TKWAT * z[NLEVSNO] + thk[NLEVSNO] * zh2osfc);
    if ((h2osfc0 > ST_THIN_SFCLAYER) && (frac_h2osfc > ST_THIN_SFCLAYER)) {
      c_h2osfc = dmax(ST_THIN_SFCLAYER, ST_CPWAT * h2osfc0 / frac_h2osfc);
      dz_h2osfc = dmax(ST_THIN_SFCLAYER, 1.0e-3 * h2osfc0 / frac_h2osfc);
    } else {
      c_h2osfc = ST_THIN_SFCLAYER;
      dz_h2osfc = ST_THIN_SFCLAYER;
    }
  }

  // ---- surface_heat_fluxes (:121-142)
  double t_h2osfc = S->t_h2osfc[c];
  double hs_soil, hs_h2osfc, hs_top_snow, dhsdT;
  {
    const int fvn = S->frac_veg_nosno[c];
    const double dlrad = S->dlrad[c], emg = S->emg[c], forc_lwrad = S->forc_lwrad[c], htvp = S->htvp[c];
    const double sabg_soil = S->sabg_soil[c];
    S->sabg_chk[c] = frac_sno_eff * S->sabg_snow[c] + (1.0 - frac_sno_eff) * sabg_soil;
    hs_soil = st_surface_heat_flux(fvn, dlrad, emg, forc_lwrad, htvp, sabg_soil, t[NLEVSNO], S->eflx_sh_soil[c],
                                   S->qflx_ev_soil[c]);
    hs_h2osfc = st_surface_heat_flux(fvn, dlrad, emg, forc_lwrad, htvp, sabg_soil, t_h2osfc, S->eflx_sh_h2osfc[c],
                                     S->qflx_ev_h2osfc[c]);
    double t_top = t[NLEVSNO];  // snotop == nlevsno when there is no snow
#pragma unroll
    for (int i = 0; i < NLEVSNO; i++)
      if (i == top) t_top = t[i];
    hs_top_snow = st_surface_heat_flux(fvn, dlrad, emg, forc_lwrad, htvp, LV(sabg_lyr, top), t_top, S->eflx_sh_snow[c],
                                       S->qflx_ev_snow[c]);
    dhsdT = -S->cgrnd[c] - 4.0 * emg * STEBOL * pow(S->t_grnd[c], 3.0);
  }

  // ---- diffusive_heat_flux (:153-170): fn, fact
  double fn[NLEVTOT], fact[NLEVTOT];
#pragma unroll
  for (int i = 0; i < NLEVTOT - 1; i++) {
    fn[i] = (i < top) ? 0.0 : tk[i] * (t[i + 1] - t[i]) / (z[i + 1] - z[i]);
  }
  fn[NLEVTOT - 1] = 0.0;
#pragma unroll
  for (int i = 0; i < NLEVTOT; i++) {
    if (i < top) {
      fact[i] = 0.0;
    } else if (i == top) {  // top active layer (i <= nlevsno, so z[i + 1] exists)
      const double zit = LV(zisoi, i);
      fact[i] = dtime / cv[i] * dzl[i] / (0.5 * (z[i] - zit + ST_CAPR * (z[i + 1] - zit)));
    } else {
      fact[i] = dtime / cv[i];
    }
    LV(fact, i) = fact[i];
  }

  // ---- set_RHS / set_LHS: rows 0..4 snow, row 5 standing surface water, rows 6..20 soil; band 0 = 2nd superdiagonal,
  //      1 = 1st superdiagonal, 2 = diagonal, 3 = 1st subdiagonal, 4 = 2nd subdiagonal.  Rows above the snow pack
  //      become identity rows for the sweep (the reference never visits them).
  double L0[NROW], L1[NROW], L2[NROW], L3[NROW], L4[NROW], R[NROW];
#pragma unroll
  for (int i = 0; i < NROW; i++) {
    L0[i] = L1[i] = L2[i] = L3[i] = L4[i] = 0.0;
    R[i] = 0.0;
  }
  const double onemcn = 1.0 - ST_CNFAC;
#pragma unroll
  for (int i = 0; i < NLEVSNO; i++) {  // get_rhs_snow (:77-107), get_matrix_snow (:165-203), snow_soil (:206-227)
    if (i == top) {
      R[i] = t[i] + fact[i] * (hs_top_snow - dhsdT * t[i] + ST_CNFAC * fn[i]);
      const double dzp = z[i + 1] - z[i];
      L2[i] = 1.0 + onemcn * fact[i] * tk[i] / dzp - fact[i] * dhsdT;
      if (snl > 1) L1[i] = -onemcn * fact[i] * tk[i] / dzp;
    } else if (i > top) {
      R[i] = t[i] + ST_CNFAC * fact[i] * (fn[i] - fn[i - 1]) + fact[i] * LV(sabg_lyr, i);
      const double dzm = z[i] - z[i - 1];
      const double dzp = z[i + 1] - z[i];
      L3[i] = -onemcn * fact[i] * tk[i - 1] / dzm;
      L2[i] = 1.0 + onemcn * fact[i] * (tk[i] / dzp + tk[i - 1] / dzm);
      if (i != NLEVSNO - 1) L1[i] = -onemcn * fact[i] * tk[i] / dzp;
    }
  }
  if (snl > 0) {
    L0[NLEVSNO - 1] = -onemcn * fact[NLEVSNO - 1] * tk[NLEVSNO - 1] / (z[NLEVSNO] - z[NLEVSNO - 1]);
  }
  {  // standing surface water row: get_rhs_ssw (:111-133), get_matrix_ssw (:322-342), ssw_soil (:345-364)
    const double dzw = 0.5 * dz_h2osfc + z[NLEVSNO];
    const double fn_h2osfc = tk_h2osfc * (t[NLEVSNO] - t_h2osfc) / dzw;
    R[NLEVSNO] = t_h2osfc + (dtime / c_h2osfc) * (hs_h2osfc - dhsdT * t_h2osfc + ST_CNFAC * fn_h2osfc);
    L2[NLEVSNO] = 1.0 + onemcn * (dtime / c_h2osfc) * tk_h2osfc / dzw - (dtime / c_h2osfc) * dhsdT;
    L1[NLEVSNO] = -onemcn * (dtime / c_h2osfc) * tk_h2osfc / dzw;
  }
  {  // soil rows: get_rhs_soil (:135-177), get_matrix_soil (:230-292), soil_snow (:295-319), soil_ssw (:367-390)
    constexpr int s0 = NLEVSNO;   // level index of the top soil layer
    constexpr int r0 = NLEVSNO + 1;  // its matrix row
    const double dzp0 = z[s0 + 1] - z[s0];
    if (snl == 0) {
      R[r0] = t[s0] + fact[s0] * (hs_top_snow - dhsdT * t[s0] + ST_CNFAC * fn[s0]);
      L2[r0] = 1.0 + onemcn * fact[s0] * tk[s0] / dzp0 - fact[s0] * dhsdT;
      L1[r0] = -onemcn * fact[s0] * tk[s0] / dzp0;
    } else {  // the snow / soil interface layer
      const double dzm = z[s0] - z[s0 - 1];
      R[r0] = t[s0] + fact[s0] * ((1.0 - frac_sno_eff) * (hs_soil - dhsdT * t[s0]) +
                                  ST_CNFAC * (fn[s0] - frac_sno_eff * fn[s0 - 1]));
      R[r0] += frac_sno_eff * fact[s0] * LV(sabg_lyr, s0);
      L2[r0] = 1.0 + onemcn * fact[s0] * (tk[s0] / dzp0 + frac_sno_eff * tk[s0 - 1] / dzm) -
               (1.0 - frac_sno_eff) * fact[s0] * dhsdT;
      L1[r0] = -onemcn * fact[s0] * tk[s0] / dzp0;
      L4[r0] = -frac_sno_eff * onemcn * fact[s0] * tk[s0 - 1] / dzm;
    }
#pragma unroll
    for (int j = s0 + 1; j < NLEVTOT - 1; j++) {
      const double dzm = z[j] - z[j - 1];
      const double dzp = z[j + 1] - z[j];
      R[j + 1] = t[j] + ST_CNFAC * fact[j] * (fn[j] - fn[j - 1]);
      L3[j + 1] = -onemcn * fact[j] * tk[j - 1] / dzm;
      L2[j + 1] = 1.0 + onemcn * fact[j] * (tk[j] / dzp + tk[j - 1] / dzm);
      L1[j + 1] = -onemcn * fact[j] * tk[j] / dzp;
    }
    constexpr int bot = NLEVTOT - 1;
    {
      const double dzm = z[bot] - z[bot - 1];
      R[bot + 1] = t[bot] - ST_CNFAC * fact[bot] * fn[bot - 1] + fact[bot] * fn[bot];
      L3[bot + 1] = -onemcn * fact[bot] * tk[bot - 1] / dzm;
      L2[bot + 1] = 1.0 + onemcn * fact[bot] * tk[bot - 1] / dzm;
    }
    if (frac_h2osfc != 0.0) {  // diagonal correction and coupling for standing surface water
      const double dzm = 0.5 * dz_h2osfc + z[s0];
      L2[r0] += frac_h2osfc * (onemcn * fact[s0] * tk_h2osfc / dzm + fact[s0] * dhsdT);
      L3[r0] = -frac_h2osfc * onemcn * fact[s0] * tk_h2osfc / (0.5 * dz_h2osfc + z[s0]);
    }
  }
#pragma unroll
  for (int i = 0; i < NLEVSNO; i++) {
    if (i < top) {  // identity row
      L0[i] = L1[i] = L3[i] = L4[i] = 0.0;
      L2[i] = 1.0;
      R[i] = 0.0;
    }
  }

  // ---- solver::PDMA (pentadiagonal_solver_impl.hh:16-76)
  double A[NROW], B[NROW], Z[NROW];
  {
    double Am2 = 0.0, Am1 = 0.0, Bm2 = 0.0, Bm1 = 0.0, Zm2 = 0.0, Zm1 = 0.0;
    constexpr int N = NROW;
#pragma unroll
    for (int i = 0; i < N - 2; i++) {
      const double Y1 = L3[i] - Am2 * L4[i];
      const double U1 = 1.0 / (L2[i] - Bm2 * L4[i] - Am1 * Y1);
      A[i] = (L1[i] - Bm1 * Y1) * U1;
      B[i] = L0[i] * U1;
      Z[i] = (R[i] - Zm2 * L4[i] - Zm1 * Y1) * U1;
      Am2 = Am1;
      Am1 = A[i];
      Bm2 = Bm1;
      Bm1 = B[i];
      Zm2 = Zm1;
      Zm1 = Z[i];
    }
    // second row from the bottom and the bottom row, in the reference's own (slightly different) form (:55-66)
    const double Y1 = L3[N - 2] - A[N - 4] * L4[N - 2];
    const double U1 = 1.0 / (L2[N - 2] - B[N - 4] * L4[N - 2] - A[N - 3] * Y1);
    A[N - 2] = (L1[N - 2] - B[N - 3] * Y1) * U1;
    const double Y2 = L3[N - 1] - A[N - 3] * L4[N - 1];
    const double U2 = 1.0 / (L2[N - 1] - B[N - 3] * L4[N - 1] - A[N - 2] * Y2);
    Z[N - 2] = (R[N - 2] - Z[N - 3] * L4[N - 2] - Z[N - 3] * Y1) * U1;
    Z[N - 1] = (R[N - 1] - Z[N - 2] * L4[N - 1] - Z[N - 2] * Y2) * U2;
    R[N - 1] = Z[N - 1];
    R[N - 2] = Z[N - 2] - A[N - 2] * R[N - 1];
#pragma unroll
    for (int i = N - 3; i >= 0; --i) R[i] = Z[i] - A[i] * R[i + 1] - B[i] * R[i + 2];
  }

  // ---- update_temperature (soil_temperature_impl.hh:154-177)
#pragma unroll
  for (int i = 0; i < NLEVSNO; i++)
    if (i >= top) t[i] = R[i];
#pragma unroll
  for (int i = NLEVSNO; i < NLEVTOT; i++) t[i] = R[i + 1];
  t_h2osfc = (frac_h2osfc != 0.0) ? R[NLEVSNO] : t[NLEVSNO];

  // ---- phase_change_h2osfc (phase_change_impl.hh:11-151); *_sl1 = the snow layer next to the ground (level 4)
  double h2osfc = h2osfc0, h2osno = h2osno0, int_snow = S->int_snow[c], snow_depth = S->snow_depth[c];
  {
    double qflx_h2osfc_to_ice = 0.0, eflx_h2osfc_to_snow = 0.0, xmf_h2osfc = 0.0;
    constexpr int sl1 = NLEVSNO - 1;
    const double fact_sl1 = fact[sl1];
    if (frac_h2osfc > 0.0 && t_h2osfc <= TFRZ) {
      const double tinc = TFRZ - t_h2osfc;
      t_h2osfc = TFRZ;
      const double hm = frac_h2osfc * (dhsdT * tinc - tinc * c_h2osfc / dtime);
      const double xm = hm * dtime / HFUS;
      const double temp1 = h2osfc + xm;
      const double z_avg = frac_sno * snow_depth;
      double rho_avg;
      if (z_avg > 0.0) {
        rho_avg = dmin(800.0, h2osno / z_avg);
      } else {
        rho_avg = 200.0;
      }
      if (temp1 >= 0.0) {
        h2osno -= xm;
        int_snow -= xm;
        if (snl > 0) ice[sl1] -= xm;
        h2osfc += xm;
        xmf_h2osfc = hm;
        qflx_h2osfc_to_ice = -xm / dtime;
        if (frac_sno > 0 && snl > 0) {
          snow_depth = h2osno / (rho_avg * frac_sno);
        } else {
          snow_depth = h2osno / DENICE;
        }
        if (snl == 0) {
          t[sl1] = t_h2osfc;
          eflx_h2osfc_to_snow = 0.0;
        } else {
          double c1, c2;
          if (snl == 1) {
            c1 = frac_sno * (dtime / fact_sl1 - dhsdT * dtime);
          } else {
            c1 = frac_sno / fact_sl1 * dtime;
          }
          if (frac_h2osfc != 0.0) {
            c2 = (-ST_CPWAT * xm - frac_h2osfc * dhsdT * dtime);
          } else {
            c2 = 0.0;
          }
          t[sl1] = (c1 * t[sl1] + c2 * t_h2osfc) / (c1 + c2);
          eflx_h2osfc_to_snow = (t_h2osfc - t[sl1]) * c2 / dtime;
        }
      } else {
        rho_avg = (h2osno * rho_avg + h2osfc * DENICE) / (h2osno + h2osfc);
        h2osno += h2osfc;
        int_snow += h2osfc;
        qflx_h2osfc_to_ice = h2osfc / dtime;
        if (snl > 0) ice[sl1] = ice[sl1] + h2osfc;
        t_h2osfc = t_h2osfc - temp1 * HFUS / (dtime * dhsdT - c_h2osfc);
        xmf_h2osfc = hm - frac_h2osfc * temp1 * HFUS / dtime;
        double c1, c2;
        if (snl == 0) {
          t[sl1] = t_h2osfc;
        } else if (snl == 1) {
          c1 = frac_sno * (dtime / fact_sl1 - dhsdT * dtime);
          if (frac_h2osfc != 0.0) {
            c2 = frac_h2osfc * (c_h2osfc - dtime * dhsdT);
          } else {
            c2 = 0.0;
          }
          t[sl1] = (c1 * t[sl1] + c2 * t_h2osfc) / (c1 + c2);
          t_h2osfc = t[sl1];
        } else {
          c1 = frac_sno / fact_sl1 * dtime;
          if (frac_h2osfc != 0.0) {
            c2 = frac_h2osfc * (c_h2osfc - dtime * dhsdT);
          } else {
            c2 = 0.0;
          }
          t[sl1] = (c1 * t[sl1] + c2 * t_h2osfc) / (c1 + c2);
          t_h2osfc = t[sl1];
        }
        h2osfc = 0.0;
        if (frac_sno > 0.0 && snl > 0) {
          snow_depth = h2osno / (rho_avg * frac_sno);
        } else {
          snow_depth = h2osno / DENICE;
        }
      }
    }
    S->xmf_h2osfc[c] = xmf_h2osfc;
    S->qflx_h2osfc_ice[c] = qflx_h2osfc_to_ice;
    S->eflx_h2osfc_snow[c] = eflx_h2osfc_to_snow;
  }

  // ---- phase_change_soisno (phase_change_impl.hh:182-418)
  {
    double xmf = 0.0, qflx_snofrz = 0.0, qflx_snow_melt = 0.0, qflx_snomelt = 0.0;
    double frz_lyr[NLEVSNO];
    int imelt[NLEVTOT];
    double tinc[NLEVTOT];
#pragma unroll
    for (int i = 0; i < NLEVSNO; i++) frz_lyr[i] = 0.0;
#pragma unroll
    for (int i = 0; i < NLEVTOT; i++) {
      imelt[i] = 0;  // (levels above the pack keep their stale flag in the state; their freezing rate is 0 either way)
      tinc[i] = 0.0;
    }
#pragma unroll
    for (int i = 0; i < NLEVSNO; i++) {  // snow layers (:222-238)
      if (i >= top) {
        if (ice[i] > 0.0 && t[i] > TFRZ) {
          imelt[i] = 1;
          tinc[i] = TFRZ - t[i];
          t[i] = TFRZ;
        }
        if (liq[i] > 0.0 && t[i] < TFRZ) {
          imelt[i] = 2;
          tinc[i] = TFRZ - t[i];
          t[i] = TFRZ;
        }
      }
    }
    double supercool[NLEVGRND];
#pragma unroll
    for (int i = NLEVSNO; i < NLEVTOT; i++) {  // soil layers (:241-273)
      if (ice[i] > 0.0 && t[i] > TFRZ) {
        imelt[i] = 1;
        tinc[i] = TFRZ - t[i];
        t[i] = TFRZ;
      }
      supercool[i - NLEVSNO] = 0.0;
      // ltype == istsoil for every column (the wrapper's dummy ltype): Zhao (1997) / Koren (1999) supercooled water
      if (t[i] < TFRZ) {
        const double smp = HFUS * (TFRZ - t[i]) / (GRAV * t[i]) * 1000.0;
        supercool[i - NLEVSNO] = LV(watsat, i - NLEVSNO) * pow(smp / LV(sucsat, i - NLEVSNO), -1.0 / LV(bsw, i - NLEVSNO));
        supercool[i - NLEVSNO] *= dzl[i] * 1000.0;
      }
      if (liq[i] > supercool[i - NLEVSNO] && t[i] < TFRZ) {
        imelt[i] = 2;
        tinc[i] = TFRZ - t[i];
        t[i] = TFRZ;
      }
      if (snl == 0 && h2osno > 0.0 && i == NLEVSNO) {
        if (t[i] > TFRZ) {
          imelt[i] = 1;
          tinc[i] = TFRZ - t[i];
          t[i] = TFRZ;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NLEVTOT; i++) {  // all active layers (:277-409)
      if (i < top) continue;
      double hm = 0.0;
      if (imelt[i] > 0) {
        if (i == top) {
          if (i < NLEVSNO) {
            hm = frac_sno_eff * (dhsdT * tinc[i] - tinc[i] / fact[i]);
          } else {
            const double temp_hm = dhsdT * tinc[i] - tinc[i] / fact[i];
            hm = (frac_h2osfc != 0.0) ? temp_hm - frac_h2osfc * (dhsdT * tinc[i]) : temp_hm;
          }
        } else if (i == NLEVSNO) {
          hm = (1.0 - frac_sno_eff - frac_h2osfc) * dhsdT * tinc[i] - tinc[i] / fact[i];
        } else {
          if (i < NLEVSNO) {
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
        if (i == NLEVSNO) {
          if (snl == 0 && h2osno > 0.0 && xm > 0.0) {
            const double temp1 = h2osno;
            h2osno = dmax(0.0, temp1 - xm);
            const double propor = h2osno / temp1;
            snow_depth *= propor;
            const double heatr = hm - HFUS * (temp1 - h2osno) / dtime;
            if (heatr > 0.0) {
              xm = heatr * dtime / HFUS;
              hm = heatr;
            } else {
              xm = 0.0;
              hm = 0.0;
            }
            qflx_snomelt = dmax(0.0, temp1 - h2osno) / dtime;
            xmf = HFUS * qflx_snomelt;
            qflx_snow_melt = qflx_snomelt;
          }
        }
        double heatr = 0.0;
        const double wmass0 = ice[i] + liq[i];
        const double wice0 = ice[i];
        if (xm > 0.0) {
          ice[i] = dmax(0.0, wice0 - xm);
          heatr = hm - HFUS * (wice0 - ice[i]) / dtime;
        } else if (xm < 0.0) {
          if (i < NLEVSNO) {
            ice[i] = dmin(wmass0, wice0 - xm);
          } else {
            const double sc = supercool[i >= NLEVSNO ? i - NLEVSNO : 0];
            if (wmass0 < sc) {
              ice[i] = 0.0;
            } else {
              ice[i] = dmin(wmass0 - sc, wice0 - xm);
            }
          }
          heatr = hm - HFUS * (wice0 - ice[i]) / dtime;
        }
        liq[i] = dmax(0.0, wmass0 - ice[i]);
        if (fabs(heatr) > 0.0) {
          if (i == top) {
            if (snl == 0) {
              t[i] += fact[i] * heatr / (1.0 - (1.0 - frac_h2osfc) * fact[i] * dhsdT);
            } else {
              t[i] += (fact[i] / frac_sno_eff) * heatr / (1.0 - fact[i] * dhsdT);
            }
          } else if (i == NLEVSNO) {
            t[i] += fact[i] * heatr / (1.0 - (1.0 - frac_sno_eff - frac_h2osfc) * fact[i] * dhsdT);
          } else {
            if (i >= NLEVSNO) {
              t[i] += fact[i] * heatr;
            } else {
              if (frac_sno_eff > 0.0) t[i] += (fact[i] / frac_sno_eff) * heatr;
            }
          }
          if (i < NLEVSNO) {
            if (liq[i] * ice[i] > 0.0) t[i] = TFRZ;
          }
        }
        xmf += HFUS * (wice0 - ice[i]) / dtime;
        if (imelt[i] == 1 && i < NLEVSNO) qflx_snomelt += dmax(0.0, (wice0 - ice[i])) / dtime;
        if (imelt[i] == 2 && i < NLEVSNO) frz_lyr[i < NLEVSNO ? i : 0] = dmax(0.0, (ice[i] - wice0)) / dtime;
      }
    }
#pragma unroll
    for (int i = 0; i < NLEVSNO; i++) {
      if (imelt[i] == 2) qflx_snofrz += frz_lyr[i];
      LV(qflx_snofrz_lyr, i) = frz_lyr[i];
    }
#pragma unroll
    for (int i = 0; i < NLEVTOT; i++) {
      if (i >= top) LV(imelt, i) = imelt[i];
    }
    S->xmf[c] = xmf;
    S->qflx_snofrz[c] = qflx_snofrz;
    S->qflx_snow_melt[c] = qflx_snow_melt;
    S->qflx_snomelt[c] = qflx_snomelt;
    S->eflx_snomelt[c] = qflx_snomelt * HFUS;
  }

  // ---- state writes and update_t_grnd (soil_temperature_impl.hh:179-205)
#pragma unroll
  for (int i = 0; i < NLEVTOT; i++) {
    // levels above the snow pack are never written by the reference, except level 4 by phase_change_h2osfc
    if (i >= top || i == NLEVSNO - 1) {
      LV(t_soisno, i) = t[i];
      LV(h2osoi_ice, i) = ice[i];
    }
    if (i >= top) LV(h2osoi_liq, i) = liq[i];
  }
  S->t_h2osfc[c] = t_h2osfc;
  S->h2osfc[c] = h2osfc;
  S->h2osno[c] = h2osno;
  S->int_snow[c] = int_snow;
  S->snow_depth[c] = snow_depth;
  {
    double t_top = t[NLEVSNO];
#pragma unroll
    for (int i = 0; i < NLEVSNO; i++)
      if (i == top) t_top = t[i];
    double t_grnd;
    if (snl > 0) {
      if (frac_h2osfc != 0.0) {
        t_grnd = frac_sno_eff * t_top + (1.0 - frac_sno_eff - frac_h2osfc) * t[NLEVSNO] + frac_h2osfc * t_h2osfc;
      } else {
        t_grnd = frac_sno_eff * t_top + (1.0 - frac_sno_eff) * t[NLEVSNO];
      }
    } else {
      if (frac_h2osfc != 0.0) {
        t_grnd = (1.0 - frac_h2osfc) * t[NLEVSNO] + frac_h2osfc * t_h2osfc;
      } else {
        t_grnd = t[NLEVSNO];
      }
    }
    S->t_grnd[c] = t_grnd;
  }
}

void launch_soil_temperature(const DevState* S, int64_t n, double dt, hipStream_t st)
{
  if (n <= 0) return;
  hipLaunchKernelGGL(k_soil_temperature, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, S, dt);
}

}  // namespace elmk
