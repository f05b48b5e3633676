// elmk_stream.h - the per-column bodies of the streaming wrappers between albedo and the leaf-temperature iteration:
//
//   canopy_hydrology_col   kokkos_canopy_hydrology   driver/kokkos/canopy_hydrology_kokkos.cc:7-95
//   surface_radiation_col  kokkos_surface_radiation  driver/kokkos/surface_radiation_kokkos.cc:7-97
//   canopy_temperature_col kokkos_canopy_temperature driver/kokkos/canopy_temperature_kokkos.cc:6-131
//
// Each body is ONE piece of source used twice: by the wrapper's own kernel (k_water_energy.hip: every input read from
// the state in HBM, every output written to it) and by the fused streaming stage of elmk_timestep7_fused (k_fz_stream,
// k_canopy_fluxes.hip), where the columns' values that a later body reads are handed over in registers (ColFwd) instead
// of being re-read.  FUSED only changes WHERE an input comes from - `FW(field, memory expression)` - never an
// expression, so both forms produce the same bits.  Outputs are always written to the state.
#pragma once
#include "elmk_dev.h"
#include "elmk_albedo_fin.h"

namespace elmk {

#define LV(f, lev) S->f[(int64_t)(lev) * ld + c]

// what the bodies of one column hand from one to the next in the fused stage
struct ColFwd {
  // canopy_hydrology ->
  int fvn, snl;
  double frac_sno, frac_sno_eff, frac_h2osfc, h2ocan, snow_depth, forc_tbot;
  // surface_radiation ->
  int nrad;
  double laisun, laisha, parsun_z, parsha_z, laisun_z, laisha_z, sabv;
  // canopy_temperature -> (t_soisno itself is NOT handed on: 20 values would cost 40 registers for the whole pass; the
  // second reader finds the column's lines in L1 / L2, they do not come from HBM twice)
  double t_h2osfc, t_grnd, qg, thm, thv, emv, emg, z0mg, z0m, displa, hgt_u, hgt_t, hgt_q, soilbeta, elai, esai, htop, forc_q,
      forc_pbot, forc_th;
};
#define FW(field, mem) (FUSED ? w.field : (mem))

// =====================================================================================================
// kokkos_frac_wet (driver/kokkos/canopy_hydrology_kokkos.cc:98-112): canopy_hydrology::fraction_wet,
// src/physics/canopy_hydrology_impl.hh:123-143
// =====================================================================================================
__device__ __forceinline__ void frac_wet_col(const DevState* __restrict__ S, const int64_t c, const Land& L)
{
  if (L.lakpoi) return;
  const int fvn = S->frac_veg_nosno[c];
  double fwet, fdry;
  if (fvn == 1) {
    const double elai = S->elai[c], esai = S->esai[c], h2ocan = S->h2ocan[c];
    if (h2ocan > 0.0) {
      const double vegt = fvn * (elai + esai);
      const double dewmxi = 1.0 / S->dewmx;
      fwet = elmk_pow(((dewmxi / vegt) * h2ocan), 0.666666666666);
      fwet = dmin(fwet, 1.0);
    } else {
      fwet = 0.0;
    }
    fdry = (1.0 - fwet) * elai / (elai + esai);
  } else {
    fwet = 0.0;
    fdry = 0.0;
  }
  S->fwet[c] = fwet;
  S->fdry[c] = fdry;
}

// =====================================================================================================
// interception :8, ground_flux :83, snow_init :146, fraction_h2osfc :312 of canopy_hydrology_impl.hh
// =====================================================================================================
// fraction_h2osfc's Newton solve (:325-333: ten fixed iterations, three erf and one exp each) only runs where water is ponded
// - a few columns in a hundred - so as part of the per-column body it kept a whole wave busy for the few lanes that needed it
// (VALU lane utilisation of k_canopy_hydrology on the branch-mix tier: 15 %, profiles/r03_lane_utilisation_before.txt).
// The body is therefore cut in two around it and the solves of a workgroup are done together: every thread hands its
// (sigma, h2osfc) to a list in LDS, the first threads of the workgroup solve one entry each with all their lanes busy, and
// every column takes its frac_h2osfc back.  Which lane evaluates the ten iterations changes nothing about their result.
struct HydroMid {
  bool live;  // false: the wrapper does nothing more on this column (lake land units)
  int snl;
  double frac_sno, frac_sno_eff, h2osno, snow_depth, int_snow, h2osfc;
};
__device__ __forceinline__ double pond_fraction(const double sigma, const double h2osfc)
{
  double d = 0.0;
#pragma unroll 1
  for (int l = 0; l < 10; l++) {
    const double fd = 0.5 * d * (1.0 + elmk_erf(d / (sigma * sqrt(2.0)))) +
                      sigma / sqrt(2.0 * ELM_PI) * elmk_exp(-elmk_sq(d) / (2.0 * elmk_sq(sigma))) - h2osfc;
    const double dfdd = 0.5 * (1.0 + elmk_erf(d / (sigma * sqrt(2.0))));
    d = d - fd / dfdd;
  }
  return 0.5 * (1.0 + elmk_erf(d / (sigma * sqrt(2.0))));
}
// All threads of the (256-thread) workgroup call this; returns the thread's own frac_h2osfc if it asked for one.
#ifndef ELMK_POND_POOL
#define ELMK_POND_POOL 1  // 0: every column solves in its own lane (development A/B)
#endif
__device__ __forceinline__ double block_pond_fraction(const bool need, const double sigma, const double h2osfc)
{
#if !ELMK_POND_POOL
  return need ? pond_fraction(sigma, h2osfc) : 0.0;
#endif
  __shared__ double s_sigma[256], s_h2osfc[256], s_frac[256];
  __shared__ uint32_t s_owner[256];
  __shared__ uint32_t s_n;
  if (threadIdx.x == 0) s_n = 0u;
  __syncthreads();
  const unsigned long long m = __ballot(need);
  if (m != 0ull) {
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0u;
    if (lane == leader) base = atomicAdd(&s_n, (uint32_t)__popcll(m));
    base = __shfl(base, leader, 64);
    if (need) {
      const uint32_t slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      s_sigma[slot] = sigma;
      s_h2osfc[slot] = h2osfc;
      s_owner[slot] = threadIdx.x;
    }
  }
  __syncthreads();
  if (threadIdx.x < s_n) s_frac[s_owner[threadIdx.x]] = pond_fraction(s_sigma[threadIdx.x], s_h2osfc[threadIdx.x]);
  __syncthreads();
  return need ? s_frac[threadIdx.x] : 0.0;
}

template <bool FUSED>
__device__ __forceinline__ void canopy_hydrology_head(const DevState* __restrict__ S, const int64_t c, const int64_t ld, const Land& L,
                                                      const double dtime, ColFwd& w, HydroMid& mid, bool& pond, double& pond_sigma)
{
  mid.live = false;
  pond = false;
  pond_sigma = 0.0;
  const int fvn = S->frac_veg_nosno[c];
  int snl = S->snl[c];
  const double forc_t = S->forc_tbot[c];
  double h2ocan = S->h2ocan[c];
  double snow_depth = S->snow_depth[c], frac_sno = S->frac_sno[c];
  w.fvn = fvn;
  w.forc_tbot = forc_t;
  if (L.lakpoi) {  // the wrapper does nothing on lake land units: later bodies see the state as it is
    if (FUSED) {
      w.snl = snl;
      w.frac_sno = frac_sno;
      w.frac_sno_eff = S->frac_sno_eff[c];
      w.frac_h2osfc = S->frac_h2osfc[c];
      w.h2ocan = h2ocan;
      w.snow_depth = snow_depth;
    }
    return;
  }
  const int do_capsnow = S->do_capsnow[c];
  const double forc_rain = S->forc_rain[c], forc_snow = S->forc_snow[c];
  const double h2ocan_in = h2ocan;

  // ---- interception (:8-67); the five wrapper temporaries start at 0 (zero-filled Views, :11-15)
  double qflx_candrip = 0.0, qflx_through_snow = 0.0, qflx_through_rain = 0.0, fracsnow = 0.0, fracrain = 0.0;
  if (L.ltype == istsoil || L.ltype == istwet || L.urbpoi || L.ltype == istcrop) {
    if (L.ctype != icol_sunwall && L.ctype != icol_shadewall) {
      if (fvn == 1 && (forc_rain + forc_snow) > 0.0) {
        const double elai = S->elai[c], esai = S->esai[c];
        fracsnow = forc_snow / (forc_snow + forc_rain);
        fracrain = forc_rain / (forc_snow + forc_rain);
        const double h2ocanmx = S->dewmx * (elai + esai);
        const double fpi = 0.25 * (1.0 - elmk_exp(-0.5 * (elai + esai)));
        qflx_through_snow = forc_snow * (1.0 - fpi);
        qflx_through_rain = forc_rain * (1.0 - fpi);
        const double qflx_prec_intr = (forc_snow + forc_rain) * fpi;
        h2ocan = dmax(0.0, (h2ocan + dtime * qflx_prec_intr));
        qflx_candrip = 0.0;
        const double xrun = (h2ocan - h2ocanmx) / dtime;
        if (xrun > 0.0) {
          qflx_candrip = xrun;
          h2ocan = h2ocanmx;
        }
      }
    }
  } else if (L.ltype == istice || L.ltype == istice_mec) {
    h2ocan = 0.0;
  }
  if (h2ocan != h2ocan_in || (h2ocan_in != h2ocan_in)) S->h2ocan[c] = h2ocan;
  w.h2ocan = h2ocan;

  // ---- ground_flux (:83-120); qflx_irrig is hard-wired 0 by the wrapper (:24)
  double prec_snow, prec_rain;
  if ((L.ctype != icol_sunwall) && (L.ctype != icol_shadewall)) {
    if (fvn == 0) {
      prec_snow = forc_snow;
      prec_rain = forc_rain;
    } else {
      prec_snow = qflx_through_snow + (qflx_candrip * fracsnow);
      prec_rain = qflx_through_rain + (qflx_candrip * fracrain);
    }
  } else {
    prec_snow = 0.0;
    prec_rain = 0.0;
  }
  prec_rain = prec_rain + 0.0;
  double qflx_snow_grnd, qflx_rain_grnd;
  if (do_capsnow) {
    S->qflx_snwcp_liq[c] = prec_rain;
    S->qflx_snwcp_ice[c] = prec_snow;
    qflx_snow_grnd = 0.0;
    qflx_rain_grnd = 0.0;
  } else {
    S->qflx_snwcp_liq[c] = 0.0;
    S->qflx_snwcp_ice[c] = 0.0;
    qflx_snow_grnd = prec_snow;
    qflx_rain_grnd = prec_rain;
  }
  S->qflx_snow_grnd[c] = qflx_snow_grnd;
  S->qflx_rain_grnd[c] = qflx_rain_grnd;

  // ---- snow_init (:146-308); wrapper passes S.forc_tbot as forc_t and S.zsoi / S.zisoi as z / zi (:64-83)
  const double accum_factor = 0.1;
  double h2osno = S->h2osno[c], int_snow = S->int_snow[c];
  const double n_melt = S->n_melt[c];
  const double temp_snow_depth = snow_depth;
  double top_ice = 0.0;  // h2osoi_ice at the top snow layer, needed again below
#pragma unroll
  for (int j = 0; j < NLEVSNO; j++) {
    double swe = 0.0;
    if (j >= NLEVSNO - snl) {
      const double ice = LV(h2osoi_ice, j);
      swe = LV(h2osoi_liq, j) + ice;
      if (j == NLEVSNO - snl) top_ice = ice;
    }
    LV(swe_old, j) = swe;
  }
  double dz_snowf, newsnow;
  if (do_capsnow) {
    dz_snowf = 0.0;
    newsnow = qflx_snow_grnd * dtime;
    frac_sno = 1.0;
    int_snow = 5.e2;
  } else {
    double bifall;
    if (forc_t > TFRZ + 2.0) {
      bifall = 50.0 + 1.7 * elmk_pow(17.0, 1.5);
    } else if (forc_t > TFRZ - 15.0) {
      bifall = 50.0 + 1.7 * elmk_pow((forc_t - TFRZ + 15.0), 1.5);
    } else {
      bifall = 50.0;
    }
    newsnow = qflx_snow_grnd * dtime;
    int_snow = dmax(int_snow, h2osno);
    const double snowmelt = S->qflx_snow_melt[c] * dtime;
    if (h2osno > 0.0) {
      if (snowmelt > 0.0) {
        const double smr = dmin(1.0, (h2osno / int_snow));
        frac_sno = 1.0 - elmk_pow((elmk_acos(dmin(1.0, (2.0 * smr - 1.0))) / ELM_PI), n_melt);
      }
      if (newsnow > 0.0) {
        const double fsno_new = 1.0 - (1.0 - elmk_tanh(accum_factor * newsnow)) * (1.0 - frac_sno);
        frac_sno = fsno_new;
        const double temp_intsnow =
            (h2osno + newsnow) / (0.5 * (elmk_cos(ELM_PI * elmk_pow((1.0 - dmax(frac_sno, 1.e-6)), (1.0 / n_melt))) + 1.0));
        int_snow = dmin(1.e8, temp_intsnow);
      }
      if (!L.urbpoi) {  // subgridflag() == 1
        if (frac_sno > 0.0) {
          snow_depth = snow_depth + newsnow / (bifall * frac_sno);
        } else {
          snow_depth = 0.0;
        }
      } else {
        snow_depth = snow_depth + newsnow / bifall;
      }
      if (S->oldfflag == 1) {
        if (snow_depth > 0.0) {
          frac_sno = elmk_tanh(snow_depth / (2.5 * ZLND * elmk_pow1((dmin(800.0, ((h2osno + newsnow) / snow_depth / 100.0))))));
        }
        if (h2osno < 1.0) {
          frac_sno = dmin(frac_sno, h2osno);
        }
      }
    } else {
      if (newsnow > 0.0) {
        const double z_avg = newsnow / bifall;
        frac_sno = elmk_tanh(accum_factor * newsnow);
        int_snow = 0.0;
        const double temp_intsnow =
            (h2osno + newsnow) / (0.5 * (elmk_cos(ELM_PI * elmk_pow((1.0 - dmax(frac_sno, 1.e-6)), (1.0 / n_melt))) + 1.0));
        int_snow = dmin(1.e8, temp_intsnow);
        if (!L.urbpoi) {
          snow_depth = z_avg / frac_sno;
        } else {
          snow_depth = newsnow / bifall;
        }
        if (S->oldfflag == 1) {
          if (snow_depth > 0.0) {
            frac_sno =
                elmk_tanh(snow_depth / (2.5 * ZLND * elmk_pow1((dmin(800.0, ((h2osno + newsnow) / snow_depth / 100.0))))));
          }
        }
      } else {
        snow_depth = 0.0;
        frac_sno = 0.0;
      }
    }
    h2osno = h2osno + newsnow;
    int_snow = int_snow + newsnow;
    dz_snowf = (snow_depth - temp_snow_depth);
  }
  double frac_sno_eff;
  if (L.ltype == istsoil || L.ltype == istcrop) {
    frac_sno_eff = frac_sno;  // subgridflag() == 1
  } else {
    frac_sno_eff = 1.0;
  }
  if (L.ltype == istwet && S->t_grnd[c] > TFRZ) {
    h2osno = 0.0;
    snow_depth = 0.0;
  }
  int newnode = 0;
  if (snl == 0 && qflx_snow_grnd > 0.0 && (frac_sno * snow_depth) >= 0.01) {
    newnode = 1;
    snl = 1;
    const int k = NLEVSNO - 1;
    const double dzk = snow_depth;
    LV(dz, k) = dzk;
    LV(zsoi, k) = -0.5 * dzk;
    LV(zisoi, k) = -dzk;
    LV(t_soisno, k) = dmin(TFRZ, forc_t);
    LV(h2osoi_ice, k) = h2osno;
    LV(h2osoi_liq, k) = 0.0;
    LV(frac_iceold, k) = 1.0;
    LV(snw_rds, k) = SNW_RDS_MIN;
    S->snl[c] = snl;
  }
  if (snl > 0 && newnode == 0) {
    const int k = NLEVSNO - snl;
    LV(h2osoi_ice, k) = top_ice + newsnow;
    LV(dz, k) = LV(dz, k) + dz_snowf;
  }

  // ---- fraction_h2osfc (:312-357): the solve itself happens between head and tail (block_pond_fraction)
  const double h2osfc = S->h2osfc[c];
  if ((L.ltype == istsoil || L.ltype == istcrop) && h2osfc > 1.e-8) {
    pond = true;
    pond_sigma = 1.0e3 * S->micro_sigma[c];
  }
  mid.live = true;
  mid.snl = snl;
  mid.frac_sno = frac_sno;
  mid.frac_sno_eff = frac_sno_eff;
  mid.h2osno = h2osno;
  mid.snow_depth = snow_depth;
  mid.int_snow = int_snow;
  mid.h2osfc = h2osfc;
}

// pond_frac: block_pond_fraction's answer for this column (read only where canopy_hydrology_head asked for it)
template <bool FUSED>
__device__ __forceinline__ void canopy_hydrology_tail(const DevState* __restrict__ S, const int64_t c, const int64_t ld, const Land& L,
                                                      ColFwd& w, const HydroMid& mid, const double pond_frac)
{
  if (!mid.live) return;
  const int snl = mid.snl;
  double frac_sno = mid.frac_sno, frac_sno_eff = mid.frac_sno_eff;
  const double h2osno = mid.h2osno, snow_depth = mid.snow_depth, int_snow = mid.int_snow;
  double h2osfc = mid.h2osfc;
  double frac_h2osfc;
  if (L.ltype == istsoil || L.ltype == istcrop) {
    if (h2osfc > 1.e-8) {
      frac_h2osfc = pond_frac;
    } else {
      frac_h2osfc = 0.0;
      LV(h2osoi_liq, NLEVSNO) = LV(h2osoi_liq, NLEVSNO) + h2osfc;
      h2osfc = 0.0;
      S->h2osfc[c] = h2osfc;
    }
    if (frac_sno > (1.0 - frac_h2osfc) && h2osno > 0.0) {
      if (frac_h2osfc > 0.01) {
        frac_h2osfc = dmax((1.0 - frac_sno), 0.01);
        frac_sno = 1.0 - frac_h2osfc;
      } else {
        frac_sno = 1.0 - frac_h2osfc;
      }
      frac_sno_eff = frac_sno;
    }
  } else {
    frac_h2osfc = 0.0;
  }
  S->snow_depth[c] = snow_depth;
  S->h2osno[c] = h2osno;
  S->int_snow[c] = int_snow;
  S->frac_sno[c] = frac_sno;
  S->frac_sno_eff[c] = frac_sno_eff;
  S->frac_h2osfc[c] = frac_h2osfc;
  w.snl = snl;
  w.frac_sno = frac_sno;
  w.frac_sno_eff = frac_sno_eff;
  w.frac_h2osfc = frac_h2osfc;
  w.snow_depth = snow_depth;
}

// the whole wrapper for one column; EVERY thread of the workgroup calls it (inside: the thread has a column)
template <bool FUSED>
__device__ __forceinline__ void canopy_hydrology_col(const DevState* __restrict__ S, const int64_t c, const int64_t ld, const Land& L,
                                                     const double dtime, ColFwd& w, const bool inside)
{
  HydroMid mid;
  mid.live = false;
  bool pond = false;
  double sigma = 0.0;
  if (inside) canopy_hydrology_head<FUSED>(S, c, ld, L, dtime, w, mid, pond, sigma);
  const double pond_frac = block_pond_fraction(pond, sigma, mid.live ? mid.h2osfc : 0.0);
  if (inside) canopy_hydrology_tail<FUSED>(S, c, ld, L, w, mid, pond_frac);
}

// =====================================================================================================
// canopy_sunshade_fractions :202, initialize_flux :9, total_absorbed_radiation :30, layer_absorbed_radiation :77,
// reflected_radiation :179 of surface_radiation_impl.hh
// =====================================================================================================
// ALBFWD (the fused step on a non-urban land unit): the albedo stage's outputs come from registers - a, and flx[i] = {flx_absdv,
// flx_absdn, flx_absiv, flx_absin}(i) - instead of from the state it has just written them to (AF / AFX below)
#define AF(field, ib) (ALBFWD ? a.field[ib] : (double)LV(field, ib))
template <bool FUSED, bool ALBFWD = false>
__device__ __forceinline__ void surface_radiation_col(const DevState* __restrict__ S, const int64_t c, const int64_t ld, const Land& L,
                                                      ColFwd& w, const AlbFwd& a, const double (&flx)[6][4])
{
  const int snl = FW(snl, S->snl[c]);
  double solad[2], solai[2];
#pragma unroll
  for (int ib = 0; ib < 2; ib++) {
    solad[ib] = LV(forc_solad, ib);
    solai[ib] = LV(forc_solai, ib);
  }
  w.nrad = 0;
  w.laisun = w.laisha = w.parsun_z = w.parsha_z = w.laisun_z = w.laisha_z = w.sabv = 0.0;

  if (!L.urbpoi) {
    // ---- canopy_sunshade_fractions: nlevcan == 1, so nrad is 0 or 1
    const int nrad = S->nrad[c];
    double laisun = 0.0, laisha = 0.0;
    w.nrad = nrad;
    if (nrad > 0) {
      const double tlai_z = S->tlai_z[c], fsun_z = ALBFWD ? a.fsun_z : (double)S->fsun_z[c];
      const double laisun_z = tlai_z * fsun_z;
      const double laisha_z = tlai_z * (1.0 - fsun_z);
      laisun += laisun_z;
      laisha += laisha_z;
      const double parsun_z = solad[0] * (ALBFWD ? a.fabd_sun_z : (double)S->fabd_sun_z[c]) + solai[0] * (ALBFWD ? a.fabi_sun_z : (double)S->fabi_sun_z[c]);
      const double parsha_z = solad[0] * (ALBFWD ? a.fabd_sha_z : (double)S->fabd_sha_z[c]) + solai[0] * (ALBFWD ? a.fabi_sha_z : (double)S->fabi_sha_z[c]);
      S->laisun_z[c] = laisun_z;
      S->laisha_z[c] = laisha_z;
      S->parsun_z[c] = parsun_z;
      S->parsha_z[c] = parsha_z;
      w.laisun_z = laisun_z;
      w.laisha_z = laisha_z;
      w.parsun_z = parsun_z;
      w.parsha_z = parsha_z;
    } else if (FUSED) {  // canopy_fluxes reads these as they stand in the state
      w.laisun_z = S->laisun_z[c];
      w.laisha_z = S->laisha_z[c];
      w.parsun_z = S->parsun_z[c];
      w.parsha_z = S->parsha_z[c];
    }
    S->laisun[c] = laisun;
    S->laisha[c] = laisha;
    w.laisun = laisun;
    w.laisha = laisha;

    // ---- initialize_flux + total_absorbed_radiation (the snl == 0 reset sits inside the band loop, :62-65)
    double sabg_soil = 0.0, sabg_snow = 0.0, sabg = 0.0, sabv = 0.0, fsa = 0.0;
    double trd[2], tri[2];
#pragma unroll
    for (int ib = 0; ib < 2; ib++) {
      const double cad = solad[ib] * AF(fabd, ib);
      const double cai = solai[ib] * AF(fabi, ib);
      sabv += cad + cai;
      fsa += cad + cai;
      trd[ib] = solad[ib] * AF(ftdd, ib);
      tri[ib] = solad[ib] * AF(ftid, ib) + solai[ib] * AF(ftii, ib);
      double absrad = trd[ib] * (1.0 - AF(albsod, ib)) + tri[ib] * (1.0 - AF(albsoi, ib));
      sabg_soil += absrad;
      absrad = trd[ib] * (1.0 - AF(albsnd, ib)) + tri[ib] * (1.0 - AF(albsni, ib));
      sabg_snow += absrad;
      absrad = trd[ib] * (1.0 - AF(albgrd, ib)) + tri[ib] * (1.0 - AF(albgri, ib));
      sabg += absrad;
      fsa += absrad;
      if (snl == 0) {
        sabg_snow = sabg;
        sabg_soil = sabg;
      }
    }
    S->sabg_soil[c] = sabg_soil;
    S->sabg_snow[c] = sabg_snow;
    S->sabg[c] = sabg;
    S->sabv[c] = sabv;
    S->fsa[c] = fsa;
    w.sabv = sabv;

    // ---- layer_absorbed_radiation (:77-176)
    double lyr[NLEVSNO + 1];
    if (snl == 0) {
#pragma unroll
      for (int i = 0; i < NLEVSNO; i++) lyr[i] = 0.0;
      lyr[NLEVSNO] = sabg;
    } else {
      double sabg_snl_sum = 0.0;
#pragma unroll
      for (int i = 0; i < NLEVSNO + 1; i++) {
        lyr[i] = (ALBFWD ? flx[i][0] : (double)LV(flx_absdv, i)) * trd[0] + (ALBFWD ? flx[i][1] : (double)LV(flx_absdn, i)) * trd[1] +
                 (ALBFWD ? flx[i][2] : (double)LV(flx_absiv, i)) * tri[0] + (ALBFWD ? flx[i][3] : (double)LV(flx_absin, i)) * tri[1];
        if (i >= NLEVSNO - snl) sabg_snl_sum += lyr[i];
      }
      if (fabs(sabg_snl_sum - sabg_snow) > 0.00001) {
        if (snl == 1) {
#pragma unroll
          for (int j = 0; j < NLEVSNO - 1; j++) lyr[j] = 0.0;
          lyr[NLEVSNO - 1] = sabg_snow * 0.6;
          lyr[NLEVSNO] = sabg_snow * 0.4;
        } else {
#pragma unroll
          for (int j = 0; j <= NLEVSNO; j++) {
            double v = 0.0;
            if (j == NLEVSNO - snl) v = sabg_snow * 0.75;
            if (j == NLEVSNO - snl + 1) v = sabg_snow * 0.25;
            lyr[j] = v;
          }
        }
      }
    }
    double err_sum = 0.0;
#pragma unroll
    for (int j = 0; j <= NLEVSNO; j++) {
      err_sum += lyr[j];
      LV(sabg_lyr, j) = lyr[j];
    }
    if (fabs(err_sum - sabg_snow) > 0.00001) atomicOr(ELMK_GENERIC(&S->err_flags[c]), ELMK_ERR_SURFRAD_LAYER_SUM);
  } else if (FUSED) {  // urban: the wrapper leaves these alone; canopy_fluxes reads them as they stand
    w.nrad = S->nrad[c];
    w.laisun = S->laisun[c];
    w.laisha = S->laisha[c];
    w.laisun_z = S->laisun_z[c];
    w.laisha_z = S->laisha_z[c];
    w.parsun_z = S->parsun_z[c];
    w.parsha_z = S->parsha_z[c];
    w.sabv = S->sabv[c];
  }

  // ---- reflected_radiation (:179-199)
  double fsr;
  if (!L.urbpoi) {
    const double rvis = AF(albd, 0) * solad[0] + AF(albi, 0) * solai[0];
    const double rnir = AF(albd, 1) * solad[1] + AF(albi, 1) * solai[1];
    fsr = rvis + rnir;
  } else {
    const double fsr_vis_d = LV(albd, 0) * solad[0];
    const double fsr_nir_d = LV(albd, 1) * solad[1];
    const double fsr_vis_i = LV(albi, 0) * solai[0];
    const double fsr_nir_i = LV(albi, 1) * solai[1];
    fsr = fsr_vis_d + fsr_nir_d + fsr_vis_i + fsr_nir_i;
  }
  S->fsr[c] = fsr;
}
#undef AF

// =====================================================================================================
// old_ground_temp :9, ground_temp :32, calc_soilalpha :51, calc_soilbeta :133 (-> surface_resistance_impl.hh:9),
// humidities :143, ground_properties :205, forcing_height :260, init_energy_fluxes :299 of canopy_temperature_impl.hh
// =====================================================================================================
// SOIL_COPIED: the fused step's early kernel (k_fz_pre, k_canopy_fluxes.hip) has already saved the soil levels of t_soisno
// into tssbef - canopy_hydrology, which runs between the two, only ever writes snow levels of t_soisno
template <bool FUSED, bool SOIL_COPIED = false>
__device__ __forceinline__ void canopy_temperature_col(const DevState* __restrict__ S, const int64_t c, const int64_t ld, const Land& L,
                                                       ColFwd& w)
{
  const int snl = FW(snl, S->snl[c]);
  const int top = NLEVSNO - snl;
  const double t_h2osfc = S->t_h2osfc[c];
  w.t_h2osfc = t_h2osfc;

  // ---- old_ground_temp: stream t_soisno -> tssbef, keeping the two levels used below
  double t_top = 0.0, t_soi0 = 0.0;
  const bool wall = (L.ctype == icol_sunwall || L.ctype == icol_shadewall || L.ctype == icol_roof);
#pragma unroll
  for (int i = 0; i < (SOIL_COPIED ? NLEVSNO + 1 : NLEVTOT); i++) {
    const double t = LV(t_soisno, i);
    if (i <= NLEVSNO) {
      if (i == top) t_top = t;
      if (i == NLEVSNO) t_soi0 = t;
    }
    if (!L.lakpoi && !(SOIL_COPIED && i >= NLEVSNO)) LV(tssbef, i) = (wall && i > 5) ? SPVAL : t;
  }
  if (!L.lakpoi) S->t_h2osfc_bef[c] = t_h2osfc;

  const double frac_sno = FW(frac_sno, S->frac_sno[c]), frac_sno_eff = FW(frac_sno_eff, S->frac_sno_eff[c]),
               frac_h2osfc = FW(frac_h2osfc, S->frac_h2osfc[c]);
  const double forc_q = S->forc_qbot[c], forc_pbot = S->forc_pbot[c];
  w.forc_q = forc_q;
  w.forc_pbot = forc_pbot;

  // ---- ground_temp (:32-48)
  double t_grnd = S->t_grnd[c];
  if (!L.lakpoi) {
    if (snl > 0) {
      t_grnd = frac_sno_eff * t_top + (1.0 - frac_sno_eff - frac_h2osfc) * t_soi0 + frac_h2osfc * t_h2osfc;
    } else {
      t_grnd = (1.0 - frac_h2osfc) * t_soi0 + frac_h2osfc * t_h2osfc;
    }
    S->t_grnd[c] = t_grnd;
  }
  w.t_grnd = t_grnd;

  // top-soil water shared by soilalpha / soilbeta, top-layer water for htvp
  const double liq_soi0 = LV(h2osoi_liq, NLEVSNO), ice_soi0 = LV(h2osoi_ice, NLEVSNO);
  double liq_top = liq_soi0, ice_top = ice_soi0;
  if (snl > 0) {
    liq_top = LV(h2osoi_liq, top);
    ice_top = LV(h2osoi_ice, top);
  }

  // ---- calc_soilalpha (:51-130); qred / hr are zero-filled wrapper temporaries
  double qred = 1.0, hr = 0.0;
  if (!L.lakpoi) {
    if (L.ltype != istwet && L.ltype != istice && L.ltype != istice_mec) {
      if (L.ltype == istsoil || L.ltype == istcrop) {
        const double wx = (liq_soi0 / DENH2O + ice_soi0 / DENICE) / LV(dz, NLEVSNO);
        double fac = dmin(1.0, wx / LV(watsat, 0));
        fac = dmax(fac, 0.01);
        double psit = -LV(sucsat, 0) * elmk_pow(fac, (-LV(bsw, 0)));
        psit = dmax(-1.e8, psit);
        hr = elmk_exp(psit / ROVERG / t_soi0);
        qred = (1.0 - frac_sno - frac_h2osfc) * hr + frac_sno + frac_h2osfc;
      } else if (L.ctype == icol_sunwall || L.ctype == icol_shadewall) {
        qred = 0.0;
      } else if (L.ctype == icol_roof || L.ctype == icol_road_imperv) {
        qred = 1.0;
      }
    }
  }

  // ---- calc_soilbeta -> calc_soilevap_stress (surface_resistance_impl.hh:9-46; compares ltype with icol_*,
  //      so for non-soil/crop, non-wet/ice land units soilbeta keeps its previous value)
  bool soilbeta_set = false;
  double soilbeta = 0.0;
  if (!L.lakpoi) {
    if (L.ltype != istwet && L.ltype != istice && L.ltype != istice_mec) {
      if (L.ltype == istsoil || L.ltype == istcrop) {
        const double wx = (liq_soi0 / DENH2O + ice_soi0 / DENICE) / LV(dz, NLEVSNO);
        const double watfc0 = LV(watfc, 0);
        if (wx < watfc0) {
          double fac_fc = dmin(1.0, wx / watfc0);
          fac_fc = dmax(fac_fc, 0.01);
          soilbeta = (1.0 - frac_sno - frac_h2osfc) * 0.25 * elmk_sq(1.0 - elmk_cos(ELM_PI * fac_fc)) + frac_sno + frac_h2osfc;
        } else {
          soilbeta = 1.0;
        }
        soilbeta_set = true;
      } else if (L.ltype == icol_road_perv || L.ltype == icol_sunwall || L.ltype == icol_shadewall ||
                 L.ltype == icol_roof || L.ltype == icol_road_imperv) {
        soilbeta = 0.0;
        soilbeta_set = true;
      }
    } else {
      soilbeta = 1.0;
      soilbeta_set = true;
    }
  }
  if (soilbeta_set) {
    S->soilbeta[c] = soilbeta;
  } else if (FUSED) {
    soilbeta = S->soilbeta[c];
  }
  w.soilbeta = soilbeta;

  // ---- humidities (:143-202)
  double qg = 0.0;
  if (!L.lakpoi) {
    double eg, degdT, qsatg, qsatgdT;
    double qg_snow, qg_soil, qg_h2osfc, dqgdT;
    if (L.ltype == istsoil || L.ltype == istcrop) {
      qsat(t_top, forc_pbot, eg, degdT, qsatg, qsatgdT);
      if (qsatg > forc_q && forc_q > qsatg) {  // never true; kept as in the reference (:159)
        qsatg = forc_q;
        qsatgdT = 0.0;
      }
      qg_snow = qsatg;
      dqgdT = frac_sno * qsatgdT;
      qsat(t_soi0, forc_pbot, eg, degdT, qsatg, qsatgdT);
      if (qsatg > forc_q && forc_q > hr * qsatg) {
        qsatg = forc_q;
        qsatgdT = 0.0;
      }
      qg_soil = hr * qsatg;
      dqgdT = dqgdT + (1.0 - frac_sno - frac_h2osfc) * hr * qsatgdT;
      if (snl == 0) {
        qg_snow = qg_soil;
        dqgdT = (1.0 - frac_h2osfc) * hr * dqgdT;
      }
      qsat(t_h2osfc, forc_pbot, eg, degdT, qsatg, qsatgdT);
      if (qsatg > forc_q && forc_q > qsatg) {
        qsatg = forc_q;
        qsatgdT = 0.0;
      }
      qg_h2osfc = qsatg;
      dqgdT = dqgdT + frac_h2osfc * qsatgdT;
      qg = frac_sno_eff * qg_snow + (1.0 - frac_sno_eff - frac_h2osfc) * qg_soil + frac_h2osfc * qg_h2osfc;
    } else {
      qsat(t_grnd, forc_pbot, eg, degdT, qsatg, qsatgdT);
      qg = qred * qsatg;
      dqgdT = qred * qsatgdT;
      if (qsatg > forc_q && forc_q > qred * qsatg) {
        qg = forc_q;
        dqgdT = 0.0;
      }
      qg_snow = qg;
      qg_soil = qg;
      qg_h2osfc = qg;
    }
    S->qg_snow[c] = qg_snow;
    S->qg_soil[c] = qg_soil;
    S->qg[c] = qg;
    S->qg_h2osfc[c] = qg_h2osfc;
    S->dqgdT[c] = dqgdT;
  } else if (FUSED) {
    qg = S->qg[c];
  }
  w.qg = qg;

  // ---- ground_properties (:205-257); z0mr / displar indexed by Land.vtype as in the reference
  double z0mg = 0.0, z0m = 0.0, displa = 0.0;
  const double elai = S->elai[c], esai = S->esai[c], htop = S->htop[c];
  const double forc_th = S->forc_thbot[c];
  w.elai = elai;
  w.esai = esai;
  w.htop = htop;
  w.forc_th = forc_th;
  if (!L.lakpoi) {
    if (!L.urbpoi) {
      double emg;
      if (L.ltype == istice || L.ltype == istice_mec) {
        emg = 0.97;
      } else {
        emg = (1.0 - frac_sno) * 0.96 + frac_sno * 0.97;
      }
      S->emg[c] = emg;
      w.emg = emg;
    } else if (FUSED) {
      w.emg = S->emg[c];
    }
    const double avmuir = 1.0;
    const double emv = 1.0 - elmk_exp(-(elai + esai) / avmuir);
    S->emv[c] = emv;
    w.emv = emv;
    double htvp = HVAP;
    if (liq_top <= 00 && ice_top > 0.0) htvp = HSUB;
    S->htvp[c] = htvp;
    z0mg = (frac_sno > 0.0) ? ZSNO : ZLND;
    S->z0mg[c] = z0mg;
    S->z0hg[c] = z0mg;
    S->z0qg[c] = z0mg;
    z0m = S->z0mr[L.vtype] * htop;
    displa = S->displar[L.vtype] * htop;
    S->z0m[c] = z0m;
    S->displa[c] = displa;
    S->z0mv[c] = z0m;
    S->z0hv[c] = z0m;
    S->z0qv[c] = z0m;
    const double thv = forc_th * (1.0 + 0.61 * forc_q);
    S->thv[c] = thv;
    w.thv = thv;
  } else {
    z0mg = S->z0mg[c];
    z0m = S->z0m[c];
    displa = S->displa[c];
    if (FUSED) {
      w.emg = S->emg[c];
      w.emv = S->emv[c];
      w.thv = S->thv[c];
    }
  }
  w.z0mg = z0mg;
  w.z0m = z0m;
  w.displa = displa;

  // ---- forcing_height (:260-296): += on the patch heights (driver resets them to forc_hgt each step)
  double hgt_t = S->forc_hgt_t_patch[c];
  double hgt_u = 0.0, hgt_q = 0.0;
  bool touched = false;
  if (S->veg_active[c]) {
    double add = 0.0;
    touched = true;
    if (L.ltype == istsoil || L.ltype == istcrop) {
      add = (FW(fvn, S->frac_veg_nosno[c]) == 0) ? (z0mg + displa) : (z0m + displa);
    } else if (L.ltype == istwet || L.ltype == istice || L.ltype == istice_mec) {
      add = z0mg;
    } else if (L.urbpoi) {
      add = 0.0 + 0.0;
    } else {
      touched = false;
    }
    if (touched) {
      hgt_u = S->forc_hgt_u_patch[c] + add;
      S->forc_hgt_u_patch[c] = hgt_u;
      hgt_t = hgt_t + add;
      S->forc_hgt_t_patch[c] = hgt_t;
      hgt_q = S->forc_hgt_q_patch[c] + add;
      S->forc_hgt_q_patch[c] = hgt_q;
    }
  }
  if (FUSED && !touched) {
    hgt_u = S->forc_hgt_u_patch[c];
    hgt_q = S->forc_hgt_q_patch[c];
  }
  w.hgt_u = hgt_u;
  w.hgt_t = hgt_t;
  w.hgt_q = hgt_q;
  const double thm = S->forc_tbot[c] + 0.0098 * hgt_t;
  S->thm[c] = thm;
  w.thm = thm;

  // ---- init_energy_fluxes (:299-327)
  S->eflx_sh_tot[c] = 0.0;
  S->eflx_lh_tot[c] = 0.0;
  S->eflx_sh_veg[c] = 0.0;
  S->qflx_evap_tot[c] = 0.0;
  S->qflx_evap_veg[c] = 0.0;
  S->qflx_tran_veg[c] = 0.0;
}

}  // namespace elmk
