// ref_harness_snow.cc - the REFERENCE's own snow-hydrology functions (src/physics/snow_hydrology_impl.hh, transpiration.h),
// included from where they lie under /root/reference at build time (nothing is copied), run one wrapper stage at a time
// behind the oracle's state container.  Part of oracle/_ref/libelmref.so (oracle/Makefile, build container only).
//
// TEST INFRASTRUCTURE ONLY - see elm_oracle.h.
//
// Why this is a separate translation unit: snow_hydrology.h itself cannot be included here (snow_hydrology.h:5 ->
// snicar_data.h:6 -> read_input.hh -> read_netcdf.hh -> netcdf.h, which the image lacks), but the file that holds the
// function BODIES, snow_hydrology_impl.hh, needs only what is included below plus two NAMES that snow_hydrology.h would
// have declared before it:
//   * the class template name SnwRdsTable (snicar_data.h:75) - it appears in the parameter list of snow_aging, which this
//     harness never instantiates (snow_aging therefore stays "parity unpinned");
//   * the prototype of ELM::snow::combine (snow_hydrology.h:134-141), which divide_layers calls before the body at
//     snow_hydrology_impl.hh:1305 is seen.
// Both are declarations of the reference's own entities - no body, no stand-in for netcdf or for any reference code - so
// every instruction executed below is the reference's.
//
// The stage numbers are those of elmo_snow_hydrology_stage (elm_oracle.h): the reference runs 0 snow_water,
// 2 aerosol_phase_change, 3 transpiration, 4 snow_compaction, 5 combine_layers, 6 divide_layers, 7 prune_snow_layers with
// the argument wiring of driver/kokkos/snow_hydrology_kokkos.cc:32-160.  Stages 1 and 8 (compute_aerosol_deposition,
// update_aerosol_mass_and_concen: whole-array functions that only dispatch through Kokkos, aerosol_physics_impl.hh:59,:106)
// and 9 (snow_aging) have no reference run here.
//
// skip[c] != 0: column c is left untouched.  The tests set it where the restatement reports that the reference reads
// outside an array (ELMO_WARN_SNOW_WATER_OOB: vol_ice[i+i] with i = 3, snow_hydrology_impl.hh:388; ELMO_WARN_SNOW_COMBINE_OOB:
// element -1 in the five-layer shift, :871-885): there the reference's result is whatever lies next to the array.
#include <algorithm>
#include <cmath>
#include <stdexcept>

#include "array.hh"
#include "compile_options.hh"
#include "elm_constants.h"
#include "land_data.h"
#include "snow_snicar.h"
#include "transpiration.h"

namespace ELM {
template <class ArrayD3>
struct SnwRdsTable;
}
namespace ELM::snow {
void combine(const double&, const double&, const double&, const double&, double&, double&, double&, double&);
}
#include "snow_hydrology_impl.hh"

#include "elm_oracle.h"

using AD1 = ELM::Array<double, 1>;
using AI1 = ELM::Array<int, 1>;
#define V(f, n) AD1(n, S->f + (size_t)c * (n))

extern "C" int elmref_snow_hydrology_stage(elmo_state* S, double dt, int stage, const unsigned char* skip)
{
  if (!(stage == 0 || (stage >= 2 && stage <= 7))) return -1;
  int threw = 0;
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
        default:  // :150-157
          ELM::snow::prune_snow_layers(S->snl[c], V(h2osoi_ice, 20), V(h2osoi_liq, 20), V(t_soisno, 20), V(dz, 20), V(zsoi, 20),
                                       V(zisoi, 21));
      }
    } catch (const std::exception&) {
      S->err_flags[c] |= 1u << 31;  // the reference threw (divide_layers' radius checks, :1032 ...)
      threw++;
    }
  }
  return threw;
}
