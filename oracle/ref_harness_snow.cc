// ref_harness_snow.cc - the REFERENCE's own snow-hydrology functions (src/physics/snow_hydrology.h, transpiration.h),
// included from where they lie under /root/reference at build time (nothing is copied), run one wrapper stage at a time
// behind the oracle's state container.  Built into oracle/_ref/libelmref_snow.so (oracle/Makefile, build container only).
//
// TEST INFRASTRUCTURE ONLY - see elm_oracle.h.
//
// How it builds without netcdf.  snow_hydrology.h:5 includes snicar_data.h for the class template SnwRdsTable (the three
// snow-aging tables, snicar_data.h:74-84); snicar_data.h:6 also includes read_input.hh (the file readers) -> read_netcdf.hh ->
// netcdf.h, which the image lacks.  No function below reads a file.  read_input.hh carries a classic include guard
// (read_input.hh:1, ELM_UTILS_READ_INPUT_HH_): with that macro defined the reference's own header skips itself, and nothing
// is put in its place - no netcdf.h, no reader, no body of any kind.  snicar_data_impl.hh:173 then needs one NAME that
// read_input.hh would have declared, ELM::IO::read_netcdf (read_input.hh:237-245), inside a function template this file never
// instantiates; it is declared below as the reference declares it, without a body.  Every instruction executed below is the
// reference's, SnwRdsTable included.  (Round 3 first did this with forward declarations of SnwRdsTable and of
// ELM::snow::combine in front of snow_hydrology_impl.hh, which left snow_aging out; including the header itself needs neither.)
//
// The stage numbers are those of elmo_snow_hydrology_stage (elm_oracle.h): the reference runs 0 snow_water,
// 2 aerosol_phase_change, 3 transpiration, 4 snow_compaction, 5 combine_layers, 6 divide_layers, 7 prune_snow_layers and
// 9 snow_aging with the argument wiring of driver/kokkos/snow_hydrology_kokkos.cc:32-186.  Stages 1 and 8
// (compute_aerosol_deposition, update_aerosol_mass_and_concen, aerosol_physics_impl.hh:33-110) have no reference run: they are
// whole-array functions whose only body is a lambda handed to the three-argument apply_parallel_for, which exists under Kokkos
// alone (invoke_kernel.hh:41-47 takes two) - they cannot be instantiated here.
//
// skip[c] != 0: column c is left untouched.  The tests set it where the restatement reports that the reference reads
// outside an array (ELMO_WARN_SNOW_WATER_OOB: vol_ice[i+i] with i = 3, snow_hydrology_impl.hh:388; ELMO_WARN_SNOW_COMBINE_OOB:
// element -1 in the five-layer shift, :871-885): there the reference's result is whatever lies next to the array.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstddef>
#include <cstring>
#include <stdexcept>
#include <string>

#include "array.hh"
#include "compile_options.hh"
#include "elm_constants.h"
#include "land_data.h"
#include "mpi_types.hh"

#define ELM_UTILS_READ_INPUT_HH_ /* read_input.hh:1-2 - the reference's own include guard (see the header of this file) */
namespace ELM::IO {
template <typename T, size_t D>
void read_netcdf(const Comm_type& comm, const std::string& filename, const std::string& varname, const std::array<GO, D>& start,
                 const std::array<GO, D>& count, T* arr);
}  // namespace ELM::IO

#include "snow_hydrology.h"
#include "transpiration.h"

// aerosol_physics.h -> aerosol_physics_impl.hh -> invoke_kernel.hh, whose serial branch names
// ELM::impl::apply_parallel_for_tuple_impl (:70) and only declares it under ENABLE_KOKKOS (:33-37): the same single declaration of
// the reference's own entity that ref_harness_soil.cc carries - no body, never instantiated.  With it the header compiles, and
// its two scalar helpers ELM::aero_impl::get_snow_mass / get_snowcap_scl_fct (aerosol_physics_impl.hh:10-31) - the only
// arithmetic of update_aerosol_mass_and_concen besides six multiplications by their results - are the reference's own.
namespace ELM::impl {
template <typename F, typename T, std::size_t... I>
constexpr decltype(auto) apply_parallel_for_tuple_impl(F&&, T&&, std::index_sequence<I...>);
}
#include "aerosol_physics.h"

#include "elm_oracle.h"

using AD1 = ELM::Array<double, 1>;
using AI1 = ELM::Array<int, 1>;
using AD3 = ELM::Array<double, 3>;
#define V(f, n) AD1(n, S->f + (size_t)c * (n))

extern "C" int elmref_snow_hydrology_stage(elmo_state* S, double dt, int stage, const unsigned char* skip)
{
  if (!(stage == 0 || (stage >= 2 && stage <= 7) || stage == 9)) return -1;
  int threw = 0;
  // S.snw_rds_table (elm_state.h): the reference's own table type, allocated by its own constructor
  // (snicar_data_impl.hh:42-47: [idx_T_max + 1][idx_Tgrd_max + 1][idx_rhos_max + 1] = 11 x 31 x 8) and filled from the state
  ELM::SnwRdsTable<AD3> table;
  if (stage == 9) {
    std::memcpy(table.snowage_tau.data(), S->snowage[0], sizeof(double) * ELMO_SNOWAGE_N);
    std::memcpy(table.snowage_kappa.data(), S->snowage[1], sizeof(double) * ELMO_SNOWAGE_N);
    std::memcpy(table.snowage_drdt0.data(), S->snowage[2], sizeof(double) * ELMO_SNOWAGE_N);
  }
  for (int64_t c = 0; c < S->ncols; c++) {
    if (skip && skip[c]) continue;
    try {
      switch (stage) {
        case 0:  // snow_hydrology_kokkos.cc:32-60
          ELM::snow::snow_water(S->do_capsnow[c], S->snl[c], dt, S->frac_sno_eff[c], S->h2osno[c], S->qflx_sub_snow[c],
                                S->qflx_evap_grnd[c], S->qflx_dew_snow[c], S->qflx_dew_grnd[c], S->qflx_rain_grnd[c],
                                S->qflx_snomelt[c], S->qflx_snow_melt[c], S->qflx_top_soil[c], S->int_snow[c], S->frac_sno[c],
                                S->mflx_neg_snow[c], V(h2osoi_liq, 20), V(h2osoi_ice, 20), V(mss_bcphi, 5), V(mss_bcpho, 5),
                                V(mss_dst1, 5), V(mss_dst2, 5), V(mss_dst3, 5), V(mss_dst4, 5), V(dz, 20));
          break;
        case 2:  // :75-82
          ELM::snow::aerosol_phase_change(S->snl[c], dt, S->qflx_sub_snow[c], V(h2osoi_liq, 20), V(h2osoi_ice, 20), V(mss_bcphi, 5),
                                          V(mss_bcpho, 5));
          break;
        case 3:  // :85-87
          ELM::trans::transpiration(S->veg_active[c] != 0, S->qflx_tran_veg[c], V(rootr, 15), V(qflx_rootsoi, 15));
          break;
        case 4:  // :89-101
          ELM::snow::snow_compaction(S->snl[c], S->land.ltype, dt, S->int_snow[c], S->n_melt[c], S->frac_sno[c],
                                     AI1(20, S->imelt + (size_t)c * 20), V(swe_old, 5), V(h2osoi_liq, 20), V(h2osoi_ice, 20),
                                     V(t_soisno, 20), V(frac_iceold, 20), V(dz, 20));
          break;
        case 5:  // :104-129
          ELM::snow::combine_layers(S->land.urbpoi != 0, S->land.ltype, dt, S->snl[c], S->h2osno[c], S->snow_depth[c],
                                    S->frac_sno_eff[c], S->frac_sno[c], S->int_snow[c], S->qflx_sl_top_soil[c],
                                    S->qflx_snow2topsoi[c], S->mflx_snowlyr_col[c], V(t_soisno, 20), V(h2osoi_ice, 20),
                                    V(h2osoi_liq, 20), V(snw_rds, 5), V(mss_bcphi, 5), V(mss_bcpho, 5), V(mss_dst1, 5),
                                    V(mss_dst2, 5), V(mss_dst3, 5), V(mss_dst4, 5), V(dz, 20), V(zsoi, 20), V(zisoi, 21));
          break;
        case 6:  // :132-147
          ELM::snow::divide_layers(S->frac_sno[c], S->snl[c], V(h2osoi_ice, 20), V(h2osoi_liq, 20), V(t_soisno, 20), V(snw_rds, 5),
                                   V(mss_bcphi, 5), V(mss_bcpho, 5), V(mss_dst1, 5), V(mss_dst2, 5), V(mss_dst3, 5),
                                   V(mss_dst4, 5), V(dz, 20), V(zsoi, 20), V(zisoi, 21));
          break;
        case 7:  // :150-157
          ELM::snow::prune_snow_layers(S->snl[c], V(h2osoi_ice, 20), V(h2osoi_liq, 20), V(t_soisno, 20), V(dz, 20), V(zsoi, 20),
                                       V(zisoi, 21));
          break;
        default:  // :170-186
          ELM::snow::snow_aging(S->do_capsnow[c], S->snl[c], S->frac_sno[c], dt, S->qflx_snwcp_ice[c], S->qflx_snow_grnd[c],
                                S->h2osno[c], V(dz, 20), V(h2osoi_liq, 20), V(h2osoi_ice, 20), V(t_soisno, 20),
                                V(qflx_snofrz_lyr, 5), table, V(snw_rds, 5));
      }
    } catch (const std::exception&) {
      S->err_flags[c] |= 1u << 31;  // the reference threw (divide_layers' radius checks, :1032 ...)
      threw++;
    }
  }
  return threw;
}

// ELM::aero_impl::get_snow_mass and get_snowcap_scl_fct (aerosol_physics_impl.hh:10-31) over n (layer, column) pairs:
// snowmass_out[i], scl_out[i] for snow_idx[i], snotop[i], do_capsnow[i], h2osoi_ice[i], h2osoi_liq[i], qflx_snwcp_ice[i].
// The pin of elmo_aerosol_mass_and_concen (oracle/elmo_physics_g.c): with these two values per layer the rest of
// update_aerosol_mass_and_concen (:84-103) is `mss *= scl; cnc = mss * (1.0 / snowmass)` on six species.
extern "C" void elmref_aerosol_helpers(int64_t n, const int* snow_idx, const int* snotop, const int* do_capsnow, const double* h2osoi_ice,
                                       const double* h2osoi_liq, const double* qflx_snwcp_ice, double dtime, double* snowmass_out,
                                       double* scl_out)
{
  for (int64_t i = 0; i < n; i++) {
    const double m = ELM::aero_impl::get_snow_mass(snow_idx[i], snotop[i], h2osoi_ice[i], h2osoi_liq[i]);
    snowmass_out[i] = m;
    scl_out[i] = ELM::aero_impl::get_snowcap_scl_fct(snow_idx[i], snotop[i], do_capsnow[i], m, qflx_snwcp_ice[i], dtime);
  }
}
