// elmk_albedo_col.h - stage 1 of kokkos_albedo_snicar for one column (surface_albedo_impl.hh: canopy_layer_lai :215,
// soil_albedo :690) and its SNICAR queue class: shared by k_alb_classify (k_albedo_snicar.hip) and the fused step's
// k_fz_prep (k_canopy_fluxes.hip), which does this classification in the same pass as frac_wet.
#pragma once
#include "elmk_dev.h"

namespace elmk {

#ifndef LV
#define LV(f, lev) S->f[(int64_t)(lev) * ld + c]
#endif

constexpr double SN_MIN_SNW = 1.0e-30;  // snow_snicar.h:27
constexpr double SA_MPE = 1.e-06;       // surface_albedo.h:56
constexpr double SA_EXTKN = 0.30;       // surface_albedo.h:57

// per-column body of stage 1; returns the number of snow layers if the column must go through SNICAR, else 0
__device__ __forceinline__ int alb_main_column(const DevState* __restrict__ S, const int64_t c, const int64_t ld, const Land& L)
{
  const double coszen = S->coszen[c];
  const double elai = S->elai[c];

  // ---- canopy_layer_lai (:215-319), nlevcan == 1: one big-leaf layer
  S->nrad[c] = 1;
  S->tlai_z[c] = elai;
  // (laisum/saisum of a single layer equal elai/esai exactly: the reference's consistency throw cannot fire)

  if (!(coszen > 0.0)) return -1;  // night column: stage 3 writes the init_timestep defaults

  // =========================== sunlit column ===========================
  const double h2osno = S->h2osno[c];
  const int snl = S->snl[c];

  // ---- soil_albedo (:690-754)
  double albsod[2], albsoi[2];
  {
    const double albice[2] = {0.8, 0.55};
    const double alblak[2] = {0.60, 0.40};
    const double alblakwi[2] = {0.10, 0.10};
    if (L.ltype == istsoil || L.ltype == istcrop) {
      const int col = S->isoicol[c];
      const double inc = dmax(0.11 - 0.40 * LV(h2osoi_vol, 0), 0.0);
#pragma unroll
      for (int ib = 0; ib < 2; ib++) {
        albsod[ib] = dmin(S->albsat[col][ib] + inc, S->albdry[col][ib]);
        albsoi[ib] = albsod[ib];
      }
    } else if (L.ltype == istice || L.ltype == istice_mec) {
#pragma unroll
      for (int ib = 0; ib < 2; ib++) {
        albsod[ib] = albice[ib];
        albsoi[ib] = albsod[ib];
      }
    } else if (L.ltype == istdlak && snl == 0) {
      const double t_grnd = S->t_grnd[c];
      const double sicefr = 1.0 - elmk_exp(-95.6 * (TFRZ - t_grnd) / TFRZ);
#pragma unroll
      for (int ib = 0; ib < 2; ib++) {
        albsod[ib] = sicefr * alblak[ib] + (1.0 - sicefr) * dmax(alblakwi[ib], 0.05 / (dmax(0.001, coszen) + 0.15));
        albsoi[ib] = sicefr * alblak[ib] + (1.0 - sicefr) * dmax(alblakwi[ib], 0.10);
      }
    } else {
#pragma unroll
      for (int ib = 0; ib < 2; ib++) {
        albsod[ib] = alblak[ib];
        albsoi[ib] = albsod[ib];
      }
    }
  }


  // sunlit: leave the soil albedos for stage 2 and queue the column by its snow-layer count
  LV(albsod, 0) = albsod[0];
  LV(albsod, 1) = albsod[1];
  LV(albsoi, 0) = albsoi[0];
  LV(albsoi, 1) = albsoi[1];
  if (h2osno > SN_MIN_SNW) return snl == 0 ? 1 : snl;  // snl == 0: one fictitious fresh-snow layer (flg_nosnl, :42-48)
  return 0;
}

}  // namespace elmk
