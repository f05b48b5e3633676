// ref_harness_soil.cc - kokkos_soil_temperature (driver/kokkos/soil_temperature_kokkos.cc:6-278) run by the REFERENCE's own
// per-column functions, included from where they lie under /root/reference at build time (nothing is copied), behind the
// oracle's state container.  Part of oracle/_ref/libelmref.so (oracle/Makefile, build container only).
//
// TEST INFRASTRUCTURE ONLY - see elm_oracle.h.
//
// Why this is a separate translation unit: soil_temperature.h, soil_temp_rhs.h and soil_temp_lhs.h include invoke_kernel.hh,
// whose serial branch (the one compiled without Kokkos) names ELM::impl::apply_parallel_for_tuple_impl at :70 but only
// defines it under ENABLE_KOKKOS (:33-37), so the header does not compile as shipped.  One declaration of that name - the
// reference's own entity, no body, never instantiated, no stand-in for Kokkos or for any reference code - in front of the
// includes is all it takes; every instruction executed below is the reference's.  The column-loop functions set_RHS /
// set_LHS (soil_temp_rhs_impl.hh:31-70, soil_temp_lhs_impl.hh:104-150) only dispatch through Kokkos (three-argument
// apply_parallel_for), so the per-column functions they wrap (soil_temp::detail::get_rhs_* / get_matrix_* / assemble_*) are
// called here in their order, with the wrapper's temporaries allocated as the wrapper allocates them.
#include <cstddef>
#include <utility>

namespace ELM::impl {
template <typename F, typename T, std::size_t... I>
constexpr decltype(auto) apply_parallel_for_tuple_impl(F&&, T&&, std::index_sequence<I...>);
}

#include "array.hh"
#include "elm_constants.h"
#include "land_data.h"

#include "pentadiagonal_solver.h"
#include "phase_change.h"
#include "soil_temp_lhs.h"
#include "soil_temp_rhs.h"
#include "soil_temperature.h"
#include "soil_thermal_properties.h"

#include "elm_oracle.h"

using AD1 = ELM::Array<double, 1>;
using AI1 = ELM::Array<int, 1>;
using AD2 = ELM::Array<double, 2>;
using AD3 = ELM::Array<double, 3>;

// The whole wrapper.  lhs_out [ncols][21][5], rhs_out [ncols][21] (right-hand side BEFORE the solve), hs_out [ncols][4] =
// {hs_soil, hs_h2osfc, hs_top_snow, dhsdT}: any may be NULL.
extern "C" void elmref_soil_temperature(elmo_state* S, double dtime, double* lhs_out, double* rhs_out, double* hs_out)
{
  namespace st = ELM::soil_temp;
  const int n = (int)S->ncols;
  const int nlevsno = 5, nlevgrnd = 15, nband = 5;
  AI1 snl(n, S->snl), imelt_unused(1, 0);
  AD1 frac_sno_eff(n, S->frac_sno_eff), frac_sno(n, S->frac_sno), frac_h2osfc(n, S->frac_h2osfc), t_h2osfc(n, S->t_h2osfc),
      t_grnd(n, S->t_grnd);
  AD2 h2osoi_liq(n, 20, S->h2osoi_liq), h2osoi_ice(n, 20, S->h2osoi_ice), t_soisno(n, 20, S->t_soisno), dz(n, 20, S->dz);
  AD2 watsat(n, 15, S->watsat), tkmg(n, 15, S->tkmg), tkdry(n, 15, S->tkdry), csol(n, 20, S->csol);
  AD2 zsoi(n, 20, S->zsoi), zisoi(n, 21, S->zisoi), fact(n, 20, S->fact), sabg_lyr(n, 6, S->sabg_lyr);

  // :77-79 dummy ltype
  const int ltype = 1;

  // :86-105 soil thermal properties
  AD2 tk(n, nlevgrnd + nlevsno, 0.0), cv(n, nlevgrnd + nlevsno, 0.0);
  AD1 tk_h2osfc(n, 0.0), c_h2osfc(n, 0.0), dz_h2osfc(n, 0.0);
  {
    AD2 thk(n, nlevgrnd + nlevsno, 0.0);
    for (int c = 0; c < n; c++) {
      ELM::soil_thermal::calc_soil_tk(c, ltype, h2osoi_liq, h2osoi_ice, t_soisno, dz, watsat, tkmg, tkdry, thk);
      ELM::soil_thermal::calc_snow_tk(c, snl(c), frac_sno(c), h2osoi_liq, h2osoi_ice, dz, thk);
      ELM::soil_thermal::calc_face_tk(c, snl(c), thk, zsoi, zisoi, tk);
      ELM::soil_thermal::calc_soil_heat_capacity(c, ltype, snl(c), S->h2osno[c], watsat, h2osoi_ice, h2osoi_liq, dz, csol, cv);
      ELM::soil_thermal::calc_snow_heat_capacity(c, snl(c), frac_sno(c), h2osoi_ice, h2osoi_liq, cv);
      tk_h2osfc(c) = ELM::soil_thermal::calc_h2osfc_tk(c, S->h2osfc[c], thk, zsoi);
      c_h2osfc(c) = ELM::soil_thermal::calc_h2osfc_heat_capacity(snl(c), S->h2osfc[c], frac_h2osfc(c));
      dz_h2osfc(c) = ELM::soil_thermal::calc_h2osfc_height(snl(c), S->h2osfc[c], frac_h2osfc(c));
    }
  }

  // :114-144 surface heat fluxes
  AD1 hs_soil(n, 0.0), hs_h2osfc(n, 0.0), hs_top_snow(n, 0.0), dhsdT(n, 0.0);
  const int soitop = nlevsno;
  for (int c = 0; c < n; c++) {
    const int snotop = nlevsno - snl(c);
    S->sabg_chk[c] = st::check_absorbed_solar(frac_sno_eff(c), S->sabg_snow[c], S->sabg_soil[c]);
    hs_soil(c) = st::calc_surface_heat_flux(S->frac_veg_nosno[c], S->dlrad[c], S->emg[c], S->forc_lwrad[c], S->htvp[c],
                                            S->sabg_soil[c], t_soisno(c, soitop), S->eflx_sh_soil[c], S->qflx_ev_soil[c]);
    hs_h2osfc(c) = st::calc_surface_heat_flux(S->frac_veg_nosno[c], S->dlrad[c], S->emg[c], S->forc_lwrad[c], S->htvp[c],
                                              S->sabg_soil[c], t_h2osfc(c), S->eflx_sh_h2osfc[c], S->qflx_ev_h2osfc[c]);
    hs_top_snow(c) = st::calc_surface_heat_flux(S->frac_veg_nosno[c], S->dlrad[c], S->emg[c], S->forc_lwrad[c], S->htvp[c],
                                                sabg_lyr(c, snotop), t_soisno(c, snotop), S->eflx_sh_snow[c], S->qflx_ev_snow[c]);
    dhsdT(c) = st::calc_dhsdT(S->cgrnd[c], S->emg[c], t_grnd(c));
  }

  // :153-172 diffusive heat flux and matrix factor (per-column slices, as the Kokkos::subview calls)
  AD2 fn(n, nlevgrnd + nlevsno, 0.0);
#define ROW(a, w) AD1(w, &a(c, 0))
  for (int c = 0; c < n; c++) {
    st::calc_diffusive_heat_flux(snl(c), ROW(tk, 20), ROW(t_soisno, 20), ROW(zsoi, 20), ROW(fn, 20));
    st::calc_heat_flux_matrix_factor(snl(c), dtime, ROW(cv, 20), ROW(dz, 20), ROW(zsoi, 20), ROW(zisoi, 21), ROW(fact, 20));
  }

  // :181-186 right-hand side and matrix: the bodies of set_RHS / set_LHS (soil_temp_rhs_impl.hh:52-67, soil_temp_lhs_impl.hh:123-147)
  AD2 rhs_vector(n, nlevgrnd + nlevsno + 1, 0.0);
  AD3 lhs_matrix(n, nlevgrnd + nlevsno + 1, nband, 0.0);
  {
    AD1 fn_h2osfc(n, 0.0), rt_ssw(n, 0.0);
    AD2 rt_snow(n, nlevsno, 0.0), rt_soil(n, nlevgrnd, 0.0);
    for (int c = 0; c < n; c++) {
      st::detail::get_rhs_snow(c, snl, hs_top_snow, dhsdT, t_soisno, fact, fn, sabg_lyr, rt_snow);
      st::detail::get_rhs_ssw(c, dtime, tk_h2osfc, t_h2osfc, dz_h2osfc, c_h2osfc, hs_h2osfc, dhsdT, t_soisno, zsoi, fn_h2osfc,
                              rt_ssw);
      st::detail::get_rhs_soil(c, snl, hs_soil, hs_top_snow, frac_sno_eff, dhsdT, t_soisno, fact, fn, sabg_lyr, rt_soil);
      st::detail::assemble_rhs(c, rt_snow, rt_ssw, rt_soil, rhs_vector);
    }
  }
  {
    AD3 bmatrix_snow(n, nlevsno, nband, 0.0), bmatrix_soil(n, nlevgrnd, nband, 0.0);
    AD2 bmatrix_ssw(n, nband, 0.0), bmatrix_snow_soil(n, nband, 0.0), bmatrix_ssw_soil(n, nband, 0.0),
        bmatrix_soil_snow(n, nband, 0.0), bmatrix_soil_ssw(n, nband, 0.0);
    for (int c = 0; c < n; c++) {
      st::detail::get_matrix_snow(c, snl, dhsdT, zsoi, fact, tk, bmatrix_snow);
      st::detail::get_matrix_snow_soil(c, snl, zsoi, fact, tk, bmatrix_snow_soil);
      st::detail::get_matrix_soil(c, snl, dhsdT, frac_sno_eff, frac_h2osfc, dz_h2osfc, tk_h2osfc, zsoi, fact, tk, bmatrix_soil);
      st::detail::get_matrix_soil_snow(c, snl, frac_sno_eff, zsoi, fact, tk, bmatrix_soil_snow);
      st::detail::get_matrix_ssw(c, dtime, dz_h2osfc, c_h2osfc, tk_h2osfc, dhsdT, zsoi, bmatrix_ssw);
      st::detail::get_matrix_ssw_soil(c, dtime, dz_h2osfc, c_h2osfc, tk_h2osfc, zsoi, bmatrix_ssw_soil);
      st::detail::get_matrix_soil_ssw(c, dtime, frac_h2osfc, dz_h2osfc, tk_h2osfc, fact, zsoi, bmatrix_soil_ssw);
      st::detail::assemble_lhs(c, bmatrix_snow_soil, bmatrix_ssw_soil, bmatrix_soil_snow, bmatrix_soil_ssw, bmatrix_ssw,
                               bmatrix_snow, bmatrix_soil, lhs_matrix);
    }
  }
  for (int c = 0; c < n; c++) {
    if (rhs_out)
      for (int i = 0; i < 21; i++) rhs_out[(size_t)c * 21 + i] = rhs_vector(c, i);
    if (lhs_out)
      for (int i = 0; i < 21; i++)
        for (int j = 0; j < 5; j++) lhs_out[((size_t)c * 21 + i) * 5 + j] = lhs_matrix(c, i, j);
    if (hs_out) {
      hs_out[(size_t)c * 4 + 0] = hs_soil(c);
      hs_out[(size_t)c * 4 + 1] = hs_h2osfc(c);
      hs_out[(size_t)c * 4 + 2] = hs_top_snow(c);
      hs_out[(size_t)c * 4 + 3] = dhsdT(c);
    }
  }

  // :210-225 solve
  {
    const int N = nlevgrnd + nlevsno + 1;
    AD2 A(n, N - 1, 0.0), B(n, N - 2, 0.0), Z(n, N, 0.0);
    for (int c = 0; c < n; c++) ELM::solver::PDMA(c, snl, lhs_matrix, A, B, Z, rhs_vector);
  }
  // :232-237 new temperatures
  for (int c = 0; c < n; c++) st::update_temperature(c, snl, frac_h2osfc, rhs_vector, t_h2osfc, t_soisno);
  // :245-266 phase change
  for (int c = 0; c < n; c++) {
    st::phase_change_h2osfc(snl(c), dtime, frac_sno(c), frac_h2osfc(c), dhsdT(c), c_h2osfc(c), fact(c, nlevsno - 1), t_h2osfc(c),
                            S->h2osfc[c], S->xmf_h2osfc[c], S->qflx_h2osfc_ice[c], S->eflx_h2osfc_snow[c], S->h2osno[c],
                            S->int_snow[c], S->snow_depth[c], h2osoi_ice(c, nlevsno - 1), t_soisno(c, nlevsno - 1));
    st::phase_change_soisno(snl(c), ltype, dtime, dhsdT(c), frac_h2osfc(c), frac_sno_eff(c), ROW(fact, 20),
                            AD1(15, S->watsat + (size_t)c * 15), AD1(15, S->sucsat + (size_t)c * 15),
                            AD1(15, S->bsw + (size_t)c * 15), ROW(dz, 20), S->h2osno[c], S->snow_depth[c], S->xmf[c],
                            S->qflx_snofrz[c], S->qflx_snow_melt[c], S->qflx_snomelt[c], S->eflx_snomelt[c],
                            AI1(20, S->imelt + (size_t)c * 20), AD1(5, S->qflx_snofrz_lyr + (size_t)c * 5), ROW(h2osoi_ice, 20),
                            ROW(h2osoi_liq, 20), ROW(t_soisno, 20));
  }
  // :273-276 ground temperature
  for (int c = 0; c < n; c++) st::update_t_grnd(c, snl, frac_h2osfc, frac_sno_eff, t_h2osfc, t_soisno, t_grnd);
#undef ROW
}
