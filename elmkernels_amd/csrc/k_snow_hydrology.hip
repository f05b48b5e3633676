// k_snow_hydrology.hip - kokkos_snow_hydrology (driver/kokkos/snow_hydrology_kokkos.cc:23-188), the call between
// soil_temperature and surface_fluxes in ELMInterface::advance (elm_kokkos_interface.cc:313):
//
//   snow::snow_water             src/physics/snow_hydrology_impl.hh:273-490   liquid percolation and aerosol flushing
//   compute_aerosol_deposition   src/physics/aerosol_physics_impl.hh:36-64
//   snow::aerosol_phase_change   snow_hydrology_impl.hh:502-548
//   trans::transpiration         src/physics/transpiration_impl.hh:15-28
//   snow::snow_compaction        snow_hydrology_impl.hh:553-645
//   snow::combine_layers         :658-898  (combine :1297-1321)
//   snow::divide_layers          :902-1288
//   snow::prune_snow_layers      :1327-1349
//   update_aerosol_mass_and_concen  aerosol_physics_impl.hh:67-106
//   snow::snow_aging             snow_hydrology_impl.hh:50-244
//
// The reference runs these as five parallel_for launches over columns; a column only ever reads what the same column
// wrote, so here one thread takes its column through all of them in one launch.  The work is layer re-meshing: short
// data-dependent loops over at most five snow levels whose bounds differ from lane to lane, i.e. dynamically indexed
// level arrays and long chains of dependent accesses to them.  With the arrays addressed in place in global memory
// (the first version) every link of those chains was a trip to L2: PMC showed the waves waiting 85 % of their cycles and
// 3.9 KB of HBM traffic per column (intermediate values written out between passes).  Now THE PACK LIVES IN LDS for the
// duration of the kernel: the 70 rows a column can touch (levels 0..5 of liquid, ice, temperature, thickness and
// interface depth - level 5 is the top soil layer, which snow_water and combine_layers reach -, levels 0..4 of node
// depth, grain radius and the six aerosol masses) are loaded as coalesced rows into [row][lane] slices (conflict-free,
// dynamic indexing costs nothing there), the passes below run on LDS as the reference's control flow statement for
// statement, and every row is written back once at the end (an unchanged value goes back as the same bits).
// 70 rows x 64 lanes x 8 B = 35 KB per one-wave workgroup, four workgroups per CU.
//
// Where the reference's result is not defined (include/elmk.h, ELMK_WARN_SNOW_*): snow_water's vol_ice[i+i] for i = 3 is
// out of bounds (vol_ice[i+1] is used); combine_layers' shift loop reads element -1 of the level arrays when the pack has
// five layers (0.0 is used); snow_aging's float -> int conversion of a non-finite index (x86-64's INT_MIN is used).
// Parity: bit for bit against the oracle's restatement, which is itself pinned bit for bit by the reference's own
// snow_hydrology.h for every function (snow_aging and its tables included) but the two aerosol bookkeeping functions (those
// two: parity unpinned, checked structurally - oracle/elmo_physics_g.c).
#include "elmk_dev.h"
#include "elmk_kernels.h"

namespace elmk {

namespace {

constexpr double CPICE = 2.11727e3;  // elm_constants.h:40
constexpr double CPWAT = 4.188e3;    // elm_constants.h:41
constexpr int NAER = 6;              // bcphi, bcpho, dst1..dst4

// the level arrays of one column
constexpr int SN_WG = 64;  // one wave per workgroup: no barrier, each lane only touches its own LDS slices
typedef __attribute__((address_space(3))) double* lds_f64;
struct SnowCol {
  lds_f64 liq, ice, t, dz, z, zi, rds;
  lds_f64 m[NAER];
};
#define AT(p, i) (p)[(i) * SN_WG]
// rows of the LDS pack: six levels for the arrays that are reached at the top soil layer, five for the others
constexpr int SN_ROW_LIQ = 0, SN_ROW_ICE = 6, SN_ROW_T = 12, SN_ROW_DZ = 18, SN_ROW_ZI = 24, SN_ROW_Z = 30, SN_ROW_RDS = 35,
              SN_ROW_M = 40, SN_ROWS = 70;

// static_cast<int>(std::round(x)) as x86-64 evaluates it (cvttsd2si): INT_MIN for NaN and out-of-range values
__device__ __forceinline__ int round_to_int(double x)
{
  const double r = __builtin_round(x);
  if (!(r > -2147483649.0 && r < 2147483648.0)) return (int)0x80000000;
  return (int)r;
}

// snow_hydrology_impl.hh:1297-1321
__device__ __forceinline__ void snow_combine(double dz2, double wliq2, double wice2, double t2, double& dz, double& wliq,
                                             double& wice, double& t)
{
  const double h = (CPICE * wice + CPWAT * wliq) * (t - TFRZ) + HFUS * wliq;
  const double h2 = (CPICE * wice2 + CPWAT * wliq2) * (t2 - TFRZ) + HFUS * wliq2;
  wice += wice2;
  wliq += wliq2;
  const double tc = TFRZ + (h + h2 - HFUS * wliq) / (CPICE * wice + CPWAT * wliq);
  dz += dz2;
  t = tc;
}

// ---- snow_water (:273-490) ---------------------------------------------------------------------------
__device__ __forceinline__ void snow_water(const SnowCol& K, const int do_capsnow, const int snl, const double dtime,
                                           const double frac_sno_eff, const double h2osno, const double qflx_sub_snow,
                                           const double qflx_evap_grnd, const double qflx_dew_snow, const double qflx_dew_grnd,
                                           const double qflx_rain_grnd, const double qflx_snomelt, double& qflx_snow_melt,
                                           double& qflx_top_soil, double& int_snow, double& frac_sno, double& mflx_neg_snow,
                                           uint32_t& err)
{
  mflx_neg_snow = 0.0;
  const int top = NLEVSNO - snl;
  {
    double ice_top = AT(K.ice, top), liq_top = AT(K.liq, top);
    double wgdif;
    if (do_capsnow) {
      wgdif = ice_top - frac_sno_eff * qflx_sub_snow * dtime;
    } else {
      wgdif = ice_top + frac_sno_eff * (qflx_dew_snow - qflx_sub_snow) * dtime;
    }
    ice_top = wgdif;
    if (wgdif < 0.0) {
      ice_top = 0.9;
      liq_top = liq_top + wgdif;
    }
    if (do_capsnow) {
      liq_top = liq_top - frac_sno_eff * qflx_evap_grnd * dtime;
    } else {
      liq_top = liq_top + frac_sno_eff * (qflx_rain_grnd + qflx_dew_grnd - qflx_evap_grnd) * dtime;
    }
    AT(K.ice, top) = ice_top;
    AT(K.liq, top) = liq_top;
    if (liq_top < 0.0) {  // reduce deeper layers' liquid water sequentially (the top soil level included)
      for (int i = top; i <= NLEVSNO; ++i) {
        const double w = AT(K.liq, i);
        if (w >= 0.0) break;
        AT(K.liq, i) = 0.0;
        mflx_neg_snow = w / dtime;
      }
    }
  }

  // percolation: one pass down the pack, carrying the layer below's porosity terms one step ahead
  double qin = 0.0, qout = 0.0;
  double qin_a[NAER] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  const double scav[NAER] = {0.20, 0.03, 0.02, 0.02, 0.01, 0.01};  // scavenging factors of the six species (:365-370)
  const double wimp = 0.05, ssi = 0.033;
  double vol_ice[NLEVSNO], vol_liq[NLEVSNO], eff_por[NLEVSNO];
#pragma unroll
  for (int i = 0; i < NLEVSNO; ++i) {
    vol_ice[i] = vol_liq[i] = eff_por[i] = 0.0;
    if (i >= top) {
      const double dzi = AT(K.dz, i);
      vol_ice[i] = dmin(1.0, AT(K.ice, i) / (dzi * frac_sno_eff * DENICE));
      eff_por[i] = 1.0 - vol_ice[i];
      vol_liq[i] = dmin(eff_por[i], AT(K.liq, i) / (dzi * frac_sno_eff * DENH2O));
    }
  }
#pragma unroll
  for (int i = 0; i < NLEVSNO; ++i) {
    if (i < top) continue;
    double liq = AT(K.liq, i) + qin;
    if (i < NLEVSNO - 1) {
      if (eff_por[i] < wimp || eff_por[i + 1] < wimp) {
        qout = 0.0;
      } else {
        qout = dmax(0.0, (vol_liq[i] - ssi * eff_por[i]) * AT(K.dz, i) * frac_sno_eff);
        // the reference reads vol_ice[i+i] (:388): in bounds for i <= 2 (and then literal), out of bounds for i = 3
        double vi;
        if (i + i < NLEVSNO) {
          vi = vol_ice[(i + i < NLEVSNO) ? i + i : 0];
        } else {
          vi = vol_ice[i + 1];
          err |= ELMK_WARN_SNOW_WATER_OOB;
        }
        qout = dmin(qout, (1.0 - vi - vol_liq[i + 1]) * AT(K.dz, i + 1) * frac_sno_eff);
      }
    } else {
      qout = dmax(0.0, (vol_liq[i] - ssi * eff_por[i]) * AT(K.dz, i) * frac_sno_eff);
    }
    qout *= 1000.0;
    liq -= qout;
    AT(K.liq, i) = liq;
    qin = qout;
    double mss_liqice = liq + AT(K.ice, i);
    if (mss_liqice < 1.0e-30) mss_liqice = 1.0e-30;
#pragma unroll
    for (int a = 0; a < NAER; a++) {
      const double m = AT(K.m[a], i) + qin_a[a];
      double qo = qout * scav[a] * (m / mss_liqice);
      if (qo > m) qo = m;
      AT(K.m[a], i) = m - qo;
      qin_a[a] = qo;
    }
  }
  for (int i = top; i < NLEVSNO; ++i) AT(K.dz, i) = dmax(AT(K.dz, i), AT(K.liq, i) / DENH2O + AT(K.ice, i) / DENICE);
  if (snl > 0) {
    qflx_snow_melt += qout / dtime;
    qflx_top_soil = (qout / dtime) + (1.0 - frac_sno_eff) * qflx_rain_grnd;
    int_snow += frac_sno_eff * (qflx_dew_snow + qflx_dew_grnd + qflx_rain_grnd) * dtime;
  } else {
    qflx_snow_melt = qflx_snomelt;
    qflx_top_soil = qflx_rain_grnd + qflx_snomelt;
    if (h2osno <= 0.0) int_snow = 0.0;
    if (h2osno <= 0.0) frac_sno = 0.0;
  }
}

// ---- snow_compaction (:553-645) -----------------------------------------------------------------------
// imelt_r / melt_r: the per-layer inputs the loop reads from the state - imelt and, by land unit, swe_old (soil, crop) or
// frac_iceold - loaded by the caller beside the pack (a load issued inside the loop is a stall of the one wave a SIMD holds);
// the loop is unrolled over the five levels so that they stay in registers
__device__ __forceinline__ void snow_compaction(const SnowCol& K, const int snl, const int ltype, const double dtime,
                                                const double int_snow, const double n_melt, const double frac_sno,
                                                const int (&imelt_r)[NLEVSNO], const double (&melt_r)[NLEVSNO])
{
  const double c2 = 23.e-3, c3 = 2.777e-6, c4 = 0.04, c5 = 2.0, dm = 100.0, eta0 = 9.0e+5;
  const int top = NLEVSNO - snl;
  double burden = 0.0;
#pragma unroll
  for (int i = 0; i < NLEVSNO; ++i) {
    if (i < top) continue;
    const double ice = AT(K.ice, i), liq = AT(K.liq, i), dzi = AT(K.dz, i);
    const double wx = (ice + liq);
    const double vd = 1.0 - (ice / DENICE + liq / DENH2O) / (frac_sno * dzi);
    if (vd > 0.001 && ice > 0.1) {
      const double bi = ice / (frac_sno * dzi);
      const double fi = ice / wx;
      const double td = TFRZ - AT(K.t, i);
      const double dexpf = elmk_exp(-c4 * td);
      double ddz1 = -c3 * dexpf;
      if (bi > dm) ddz1 *= elmk_exp(-46.0e-3 * (bi - dm));
      if (liq > 0.01 * dzi * frac_sno) ddz1 *= c5;
      const double ddz2 = -(burden + wx / 2.0) * elmk_exp(-0.08 * td - c2 * bi) / eta0;
      double ddz3;
      if (imelt_r[i] == 1) {
        if (ltype == istsoil || ltype == istcrop) {  // subgridflag() == 1
          const double swe = melt_r[i];
          ddz3 = dmax(0.0, dmin(1.0, (swe - wx) / wx));
          double wsum = 0.0;
          if ((swe - wx) > 0.0) {
            if (i == top) {
              for (int j = top; j < NLEVSNO; ++j) wsum += AT(K.liq, j) + AT(K.ice, j);
            }
            const double fsno_melt = 1.0 - elmk_pow(elmk_acos(2.0 * dmin(1.0, wsum / int_snow) - 1.0) / ELM_PI, n_melt);
            ddz3 -= dmax(0.0, (fsno_melt - frac_sno) / frac_sno);
          }
          ddz3 = -1.0 / dtime * ddz3;
        } else {
          const double fio = melt_r[i];
          ddz3 = -1.0 / dtime * dmax(0.0, (fio - fi) / fio);
        }
      } else {
        ddz3 = 0.0;
      }
      const double pdzdtc = ddz1 + ddz2 + ddz3;
      AT(K.dz, i) = dmax(dzi * (1.0 + pdzdtc * dtime), (ice / DENICE + liq / DENH2O) / frac_sno);
    }
    burden += wx;
  }
}

// copy element `from` of every level array of the pack into element `to` (the shift loops of combine_layers);
// from == -1 does not exist (file header): 0.0
__device__ __forceinline__ void shift_level(const SnowCol& K, const int to, const int from, uint32_t& err)
{
  if (from < 0) err |= ELMK_WARN_SNOW_COMBINE_OOB;
  const bool ok = from >= 0;
  const int f = ok ? from : 0;
  const double t = AT(K.t, f), liq = AT(K.liq, f), ice = AT(K.ice, f), rds = AT(K.rds, f), dz = AT(K.dz, f);
  AT(K.t, to) = ok ? t : 0.0;
  AT(K.liq, to) = ok ? liq : 0.0;
  AT(K.ice, to) = ok ? ice : 0.0;
#pragma unroll
  for (int a = 0; a < NAER; a++) {
    const double m = AT(K.m[a], f);
    AT(K.m[a], to) = ok ? m : 0.0;
  }
  AT(K.rds, to) = ok ? rds : 0.0;
  AT(K.dz, to) = ok ? dz : 0.0;
}

// ---- combine_layers (:658-898) --------------------------------------------------------------------------
__device__ __forceinline__ void combine_layers(const SnowCol& K, const bool urbpoi, const int ltype, const double dtime, int& snl,
                                               double& h2osno, double& snow_depth, double& frac_sno_eff, double& frac_sno,
                                               double& int_snow, double& qflx_sl_top_soil, double& qflx_snow2topsoi,
                                               double& mflx_snowlyr_col, uint32_t& err)
{
  const double dzmin[5] = {0.010, 0.015, 0.025, 0.055, 0.115};
  qflx_sl_top_soil = 0.0;
  qflx_snow2topsoi = 0.0;
  mflx_snowlyr_col = 0.0;
  const bool soil_like = (ltype == istsoil || urbpoi || ltype == istcrop);

  int top_old = NLEVSNO - snl;
  for (int i = top_old; i < NLEVSNO; ++i) {
    if (AT(K.ice, i) <= .01) {  // 0.01 to avoid runaway ice build-up: the layer goes into the one below
      const double liq_i = AT(K.liq, i), ice_i = AT(K.ice, i);
      if (soil_like || i != NLEVSNO - 1) {
        AT(K.liq, i + 1) += liq_i;
        AT(K.ice, i + 1) += ice_i;
      }
      if (soil_like && i == NLEVSNO - 1) {
        qflx_sl_top_soil = (liq_i + ice_i) / dtime;
        mflx_snowlyr_col += qflx_sl_top_soil;
      }
      if (i != NLEVSNO - 1) {
        AT(K.dz, i + 1) += AT(K.dz, i);
#pragma unroll
        for (int a = 0; a < NAER; a++) AT(K.m[a], i + 1) += AT(K.m[a], i);
      }
      // shift all elements above this down one
      const int top = NLEVSNO - snl;
      if (i > top && snl > 1) {
        for (int ii = i; ii > top; --ii) {
          if (!soil_like && ii == NLEVSNO - 1) qflx_sl_top_soil = (AT(K.liq, ii) + AT(K.ice, ii)) / dtime;
          shift_level(K, ii, ii - 1, err);
        }
      }
      snl -= 1;
    }
  }

  h2osno = 0.0;
  snow_depth = 0.0;
  double zwice = 0.0, zwliq = 0.0;
  top_old = NLEVSNO - snl;
  for (int i = top_old; i < NLEVSNO; ++i) {
    const double ice = AT(K.ice, i), liq = AT(K.liq, i);
    h2osno += ice + liq;
    snow_depth += AT(K.dz, i);
    zwice += ice;
    zwliq += liq;
  }

  // all snow gone: the liquid water ponds on the soil surface
  if (snow_depth > 0.0 && ((frac_sno_eff * snow_depth < 0.01) || (h2osno / (frac_sno_eff * snow_depth) < 50.0))) {
    snl = 0;
    h2osno = zwice;
#pragma unroll
    for (int i = 0; i < NLEVSNO; ++i) {
#pragma unroll
      for (int a = 0; a < NAER; a++) AT(K.m[a], i) = 0.0;
    }
    if (h2osno <= 0.0) snow_depth = 0.0;
    if (soil_like) {
      AT(K.liq, NLEVSNO - 1) = 0.0;
      AT(K.liq, NLEVSNO) += zwliq;
      qflx_snow2topsoi = zwliq / dtime;
      mflx_snowlyr_col += zwliq / dtime;
    }
    if (ltype == istwet || ltype == istice || ltype == istice_mec) AT(K.liq, NLEVSNO - 1) = 0.0;
  }
  if (h2osno <= 0.0) {
    snow_depth = 0.0;
    frac_sno = 0.0;
    frac_sno_eff = 0.0;
    int_snow = 0.0;
  }

  // two or more layers: thin or light layers are combined with a neighbour
  if (snl > 1) {
    int mssi = 0;
    top_old = NLEVSNO - snl;
    for (int i = top_old; i < NLEVSNO; ++i) {
      const double dzi = AT(K.dz, i);
      if ((frac_sno_eff * dzi < dzmin[mssi]) || ((AT(K.ice, i) + AT(K.liq, i)) / (frac_sno_eff * dzi) < 50.0)) {
        int neibor;
        if (i == NLEVSNO - snl) {
          neibor = i + 1;
        } else if (i == NLEVSNO - 1) {
          neibor = i - 1;
        } else {
          neibor = i + 1;
          if ((AT(K.dz, i - 1) + dzi) < (AT(K.dz, i + 1) + dzi)) neibor = i - 1;
        }
        // nodes l and j are combined and stored as node j
        const int j = (neibor > i) ? neibor : i;
        const int l = (neibor > i) ? i : neibor;
#pragma unroll
        for (int a = 0; a < NAER; a++) AT(K.m[a], j) += AT(K.m[a], l);
        double liq_j = AT(K.liq, j), ice_j = AT(K.ice, j), t_j = AT(K.t, j), dz_j = AT(K.dz, j);
        const double liq_l = AT(K.liq, l), ice_l = AT(K.ice, l);
        AT(K.rds, j) = (AT(K.rds, j) * (liq_j + ice_j) + AT(K.rds, l) * (liq_l + ice_l)) / (liq_j + ice_j + liq_l + ice_l);
        snow_combine(AT(K.dz, l), liq_l, ice_l, AT(K.t, l), dz_j, liq_j, ice_j, t_j);
        AT(K.dz, j) = dz_j;
        AT(K.liq, j) = liq_j;
        AT(K.ice, j) = ice_j;
        AT(K.t, j) = t_j;
        // shift all elements above this down one (the reference's bound runs one element past the top of the pack)
        if (j - 1 > NLEVSNO - snl) {
          for (int k = j - 1; k > NLEVSNO - snl - 1; --k) shift_level(K, k, k - 1, err);
        }
        snl -= 1;
        if (snl <= 1) break;
      } else {
        mssi += 1;
      }
    }
  }

  // node depths and layer interfaces
  for (int i = NLEVSNO - 1; i >= NLEVSNO - snl; --i) {
    const double zi1 = AT(K.zi, i + 1), dzi = AT(K.dz, i);
    AT(K.z, i) = zi1 - 0.5 * dzi;
    AT(K.zi, i) = zi1 - dzi;
  }
}

// ---- divide_layers (:902-1288): the pack as a stack of at most five elements, top first ----------------
struct SnowStack {
  double dz[NLEVSNO], ice[NLEVSNO], liq[NLEVSNO], t[NLEVSNO], rds[NLEVSNO];
  double m[NAER][NLEVSNO];
};
// the part of element k beyond `keep` metres goes into element k + 1; `chk`: the element whose radius the reference
// checks against the Mie table's range afterwards (k + 1, except in the last copy, :1252, which checks k)
template <int k, int chk>
__device__ __forceinline__ void stack_move_excess(SnowStack& s, const double keep, uint32_t& err)
{
  const double drr = s.dz[k] - keep;
  double propor = drr / s.dz[k];
  double zwice = propor * s.ice[k];
  double zwliq = propor * s.liq[k];
  double zm[NAER];
#pragma unroll
  for (int a = 0; a < NAER; a++) zm[a] = propor * s.m[a][k];
  propor = keep / s.dz[k];
  s.ice[k] *= propor;
  s.liq[k] *= propor;
#pragma unroll
  for (int a = 0; a < NAER; a++) s.m[a][k] *= propor;
  s.dz[k] = keep;
#pragma unroll
  for (int a = 0; a < NAER; a++) s.m[a][k + 1] += zm[a];
  s.rds[k + 1] = (s.rds[k + 1] * (s.liq[k + 1] + s.ice[k + 1]) + s.rds[k] * (zwliq + zwice)) /
                 (s.liq[k + 1] + s.ice[k + 1] + zwliq + zwice);
  if (s.rds[chk] < 30 || s.rds[chk] > 1500) err |= ELMK_ERR_SNOW_DIVIDE_RDS;  // snw_rds_min_tbl, snw_rds_max_tbl
  snow_combine(drr, zwliq, zwice, s.t[k], s.dz[k + 1], s.liq[k + 1], s.ice[k + 1], s.t[k + 1]);
}
// element k is halved into k and a new element k + 1; the new element is kept below freezing (`tst`: the element whose
// temperature that test reads - the new one, except at :1139 where the reference reads element 2 for the new element 3)
template <int k, int tst>
__device__ __forceinline__ void stack_split(SnowStack& s)
{
  const double dtdz = (s.t[k - 1] - s.t[k]) / ((s.dz[k - 1] + s.dz[k]) / 2.0);
  s.dz[k] /= 2.0;
  s.ice[k] /= 2.0;
  s.liq[k] /= 2.0;
  s.dz[k + 1] = s.dz[k];
  s.ice[k + 1] = s.ice[k];
  s.liq[k + 1] = s.liq[k];
  s.t[k + 1] = s.t[k] - dtdz * s.dz[k] / 2.0;
  if (s.t[tst] >= TFRZ) {
    s.t[k + 1] = s.t[k];
  } else {
    s.t[k] += dtdz * s.dz[k] / 2.0;
  }
#pragma unroll
  for (int a = 0; a < NAER; a++) {
    s.m[a][k] /= 2.0;
    s.m[a][k + 1] = s.m[a][k];
  }
  s.rds[k + 1] = s.rds[k];
}

__device__ __forceinline__ void divide_layers(const SnowCol& K, const double frac_sno, int& snl, uint32_t& err)
{
  if (snl == 0) return;  // nothing to divide, nothing written (msno = 0: every loop of the reference is empty)
  SnowStack s;
  int msno = snl;
  const int top0 = NLEVSNO - snl;
#pragma unroll
  for (int i = 0; i < NLEVSNO; ++i) {
    const bool in = i < snl;
    const int lev = in ? i + top0 : NLEVSNO - 1;
    s.dz[i] = in ? frac_sno * AT(K.dz, lev) : 0.0;
    s.ice[i] = in ? AT(K.ice, lev) : 0.0;
    s.liq[i] = in ? AT(K.liq, lev) : 0.0;
    s.t[i] = in ? AT(K.t, lev) : 0.0;
    s.rds[i] = in ? AT(K.rds, lev) : 0.0;
#pragma unroll
    for (int a = 0; a < NAER; a++) s.m[a][i] = in ? AT(K.m[a], lev) : 0.0;
  }
  if (msno == 1) {
    if (s.dz[0] > 0.03) {  // one layer becomes two equal ones (:956-980)
      msno = 2;
      s.dz[0] /= 2.0;
      s.ice[0] /= 2.0;
      s.liq[0] /= 2.0;
      s.dz[1] = s.dz[0];
      s.ice[1] = s.ice[0];
      s.liq[1] = s.liq[0];
      s.t[1] = s.t[0];
#pragma unroll
      for (int a = 0; a < NAER; a++) {
        s.m[a][0] /= 2.0;
        s.m[a][1] = s.m[a][0];
      }
      s.rds[1] = s.rds[0];
    }
  }
  if (msno > 1) {
    if (s.dz[0] > 0.02) {
      stack_move_excess<0, 1>(s, 0.02, err);
      if (msno <= 2 && s.dz[1] > 0.07) {
        msno = 3;
        stack_split<1, 2>(s);
      }
    }
  }
  if (msno > 2) {
    if (s.dz[1] > 0.05) {
      stack_move_excess<1, 2>(s, 0.05, err);
      if (msno <= 3 && s.dz[2] > 0.18) {
        msno = 4;
        stack_split<2, 2>(s);
      }
    }
  }
  if (msno > 3) {
    if (s.dz[2] > 0.11) {
      stack_move_excess<2, 3>(s, 0.11, err);
      if (msno <= 4 && s.dz[3] > 0.41) {
        msno = 5;
        stack_split<3, 4>(s);
      }
    }
  }
  if (msno > 4) {
    if (s.dz[3] > 0.23) stack_move_excess<3, 3>(s, 0.23, err);
  }
  snl = msno;
  const int top = NLEVSNO - snl;
#pragma unroll
  for (int i = 0; i < NLEVSNO; ++i) {
    if (i < msno) {
      const int lev = i + top;
      AT(K.dz, lev) = s.dz[i] / frac_sno;
      AT(K.ice, lev) = s.ice[i];
      AT(K.liq, lev) = s.liq[i];
      AT(K.t, lev) = s.t[i];
#pragma unroll
      for (int a = 0; a < NAER; a++) AT(K.m[a], lev) = s.m[a][i];
      AT(K.rds, lev) = s.rds[i];
    }
  }
  for (int i = NLEVSNO - 1; i >= top; --i) {
    const double zi1 = AT(K.zi, i + 1), dzi = AT(K.dz, i);
    AT(K.z, i) = zi1 - 0.5 * dzi;
    AT(K.zi, i) = zi1 - dzi;
  }
}

// ---- snow_aging (:50-244) ---------------------------------------------------------------------------------
// Two passes over the five levels, both unrolled (a level above the pack is skipped): the first forms the three table indices
// of every layer and issues all the table reads together - fifteen gathers in flight instead of three at a time behind
// each layer's arithmetic, with one wave per SIMD to hide them - the second does each layer's arithmetic as before.  Layers
// do not depend on each other here (the pass writes snw_rds only and reads its own element of it), so every layer sees
// exactly the operands it saw in the layer-by-layer form.  snofrz_r: qflx_snofrz_lyr of the five levels, read by the caller.
__device__ __forceinline__ void snow_aging(const DevState* __restrict__ S, const SnowCol& K, const int do_capsnow, const int snl,
                                           const double frac_sno, const double dtime, const double qflx_snwcp_ice,
                                           const double qflx_snow_grnd, const double h2osno, const double (&snofrz_r)[NLEVSNO],
                                           uint32_t& err)
{
  const double snw_rds_refrz = 1000.0;
  const double C2_liq_Brun89 = 4.22e-13;
  if (snl > 0) {
    const int snl_top = NLEVSNO - snl;
#pragma unroll
    for (int i = 0; i < NLEVSNO; ++i)
      if (i < snl_top) AT(K.rds, i) = 0.0;
    const gptr<const double> tab = S->snowage;
    double bst_tau[NLEVSNO], bst_kappa[NLEVSNO], bst_drdt0[NLEVSNO];
#pragma unroll
    for (int i = 0; i < NLEVSNO; ++i) {
      int k = 0;  // (a level above the pack reads element 0 and does not use it: the reads stay branch-free)
      if (i >= snl_top) {
        const double liq = AT(K.liq, i), ice = AT(K.ice, i), dzi = AT(K.dz, i), ti = AT(K.t, i);
        double t_snotop, t_snobtm;
        {
          const double t_dn = AT(K.t, i + 1), dz_dn = AT(K.dz, i + 1);
          if (i == snl_top) {
            t_snotop = ti;
            t_snobtm = (t_dn * dzi + ti * dz_dn) / (dzi + dz_dn);
          } else {
            const double t_up = AT(K.t, i > 0 ? i - 1 : 0), dz_up = AT(K.dz, i > 0 ? i - 1 : 0);
            t_snotop = (t_up * dzi + ti * dz_up) / (dzi + dz_up);
            t_snobtm = (t_dn * dzi + ti * dz_dn) / (dzi + dz_dn);
          }
        }
        const double cdz = frac_sno * dzi;
        const double dTdz = fabs((t_snotop - t_snobtm) / cdz);
        double rhos = (liq + ice) / cdz;
        rhos = dmax(50.0, rhos);
        int T_idx = round_to_int((ti - 223) / 5);
        int Tgrd_idx = round_to_int(dTdz / 10);
        int rhos_idx = round_to_int((rhos - 50) / 50);
        if (T_idx < 0) T_idx = 0;
        if (T_idx > 10) T_idx = 10;
        if (Tgrd_idx < 0) Tgrd_idx = 0;
        if (Tgrd_idx > 30) Tgrd_idx = 30;
        if (rhos_idx < 0) rhos_idx = 0;
        if (rhos_idx > 7) rhos_idx = 7;
        k = (T_idx * 31 + Tgrd_idx) * 8 + rhos_idx;
      }
      bst_tau[i] = tab[k];
      bst_kappa[i] = tab[ELMK_SNOWAGE_N + k];
      bst_drdt0[i] = tab[2 * ELMK_SNOWAGE_N + k];
    }
#pragma unroll
    for (int i = 0; i < NLEVSNO; ++i) {
      if (i < snl_top) continue;
      const double liq = AT(K.liq, i), ice = AT(K.ice, i);
      const double h2osno_lyr = liq + ice;
      const double rds = AT(K.rds, i);
      double dr_fresh = rds - SNW_RDS_MIN;
      if (fabs(dr_fresh) < 1.0e-8) {
        dr_fresh = 0.0;
      } else if (dr_fresh < 0.0) {
        err |= ELMK_ERR_SNOW_AGE_DRFRESH;
      }
      double dr = (bst_drdt0[i] * elmk_pow(bst_tau[i] / (dr_fresh + bst_tau[i]), 1.0 / bst_kappa[i])) * (dtime / 3600.0);
      const double frc_liq = dmin(0.1, (liq / (liq + ice)));
      const double dr_wet = 1.0e18 * (dtime * (C2_liq_Brun89 * elmk_pow(frc_liq, 3.0)) / (4.0 * ELM_PI * elmk_sq(rds)));
      dr += dr_wet;
      double newsnow;
      if (do_capsnow) {
        newsnow = dmax(0.0, (qflx_snwcp_ice * dtime));
      } else {
        newsnow = dmax(0.0, (qflx_snow_grnd * dtime));
      }
      const double refrzsnow = dmax(0.0, (snofrz_r[i] * dtime));
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
      double r = (rds + dr) * frc_oldsnow + SNW_RDS_MIN * frc_newsnow + snw_rds_refrz * frc_refrz;
      if (r < SNW_RDS_MIN) r = SNW_RDS_MIN;
      if (r > SNW_RDS_MIN) r = SNW_RDS_MIN;  // (:221-223: the reference's upper bound is SNW_RDS_MIN as well)
      AT(K.rds, i) = r;
    }
  }
  if (snl == 0) {
    if (h2osno > 0.0) AT(K.rds, NLEVSNO - 1) = SNW_RDS_MIN;
  }
}

}  // namespace

__global__ __launch_bounds__(SN_WG) void k_snow_hydrology(const DevState* __restrict__ S, const double dtime)
{
  __shared__ double s_pack[SN_ROWS][SN_WG];
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= S->ncols) return;
  const int64_t ld = S->ld;
  const Land L = S->land;
  const int lane = (int)threadIdx.x;
  int snl = S->snl[c];  // (first: the per-layer reads further down are only issued for a wave that holds a snow pack)
  // the pack: global rows -> LDS (all loads independent: one batch in flight)
  const dfield g6[5] = {S->h2osoi_liq + c, S->h2osoi_ice + c, S->t_soisno + c, S->dz + c, S->zisoi + c};
  const dfield g5[2 + NAER] = {S->zsoi + c, S->snw_rds + c, S->mss_bcphi + c, S->mss_bcpho + c,
                                     S->mss_dst1 + c, S->mss_dst2 + c, S->mss_dst3 + c, S->mss_dst4 + c};
#pragma unroll
  for (int f = 0; f < 5; f++)
#pragma unroll
    for (int i = 0; i < 6; i++) s_pack[f * 6 + i][lane] = g6[f][(int64_t)i * ld];
#pragma unroll
  for (int f = 0; f < 2 + NAER; f++)
#pragma unroll
    for (int i = 0; i < 5; i++) s_pack[SN_ROW_Z + f * 5 + i][lane] = g5[f][(int64_t)i * ld];
  SnowCol K;
  K.liq = (lds_f64)&s_pack[SN_ROW_LIQ][lane];
  K.ice = (lds_f64)&s_pack[SN_ROW_ICE][lane];
  K.t = (lds_f64)&s_pack[SN_ROW_T][lane];
  K.dz = (lds_f64)&s_pack[SN_ROW_DZ][lane];
  K.zi = (lds_f64)&s_pack[SN_ROW_ZI][lane];
  K.z = (lds_f64)&s_pack[SN_ROW_Z][lane];
  K.rds = (lds_f64)&s_pack[SN_ROW_RDS][lane];
#pragma unroll
  for (int a = 0; a < NAER; a++) K.m[a] = (lds_f64)&s_pack[SN_ROW_M + a * 5][lane];
  uint32_t err = 0;

  // every per-column scalar of the wrapper, read here beside the pack (the compiler cannot move a load above the stores
  // of an earlier pass, and with one wave per SIMD a load issued where it is used is a stall)
  const int do_capsnow = S->do_capsnow[c];
  double frac_sno_eff = S->frac_sno_eff[c], frac_sno = S->frac_sno[c], h2osno = S->h2osno[c], int_snow = S->int_snow[c];
  const double qflx_sub_snow = S->qflx_sub_snow[c];
  double qflx_snow_melt = S->qflx_snow_melt[c], qflx_top_soil = S->qflx_top_soil[c], mflx_neg_snow;
  const double qflx_evap_grnd = S->qflx_evap_grnd[c], qflx_dew_snow = S->qflx_dew_snow[c], qflx_dew_grnd = S->qflx_dew_grnd[c],
               qflx_rain_grnd = S->qflx_rain_grnd[c], qflx_snomelt = S->qflx_snomelt[c];
  const double dep0 = S->aer_bcphi[c], dep1a = S->aer_bcpho[c], dep1b = S->aer_bcdep[c], dep2a = S->aer_dst1_1[c],
               dep2b = S->aer_dst1_2[c], dep3a = S->aer_dst2_1[c], dep3b = S->aer_dst2_2[c], dep4a = S->aer_dst3_1[c],
               dep4b = S->aer_dst3_2[c], dep5a = S->aer_dst4_1[c], dep5b = S->aer_dst4_2[c];
  const bool veg_active = S->veg_active[c] != 0;
  const double qflx_tran_veg = S->qflx_tran_veg[c];
  double rootr[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) rootr[i] = S->rootr[(int64_t)i * ld + c];
  const double n_melt = S->n_melt[c];
  double snow_depth = S->snow_depth[c];
  // the per-layer inputs of snow_compaction (imelt; swe_old on soil / crop land units, frac_iceold elsewhere) and of snow_aging
  // (qflx_snofrz_lyr): none of them is written here
  int imelt_r[NLEVSNO];
  double melt_r[NLEVSNO], snofrz_r[NLEVSNO];
#pragma unroll
  for (int i = 0; i < NLEVSNO; ++i) {
    imelt_r[i] = 0;
    melt_r[i] = 0.0;
    snofrz_r[i] = 0.0;
  }
  if (__ballot(snl > 0) != 0ull) {  // (wave-uniform; a column without layers reads none of them)
    const bool soilcrop = (L.ltype == istsoil || L.ltype == istcrop);
#pragma unroll
    for (int i = 0; i < NLEVSNO; ++i) {
      imelt_r[i] = S->imelt[(int64_t)i * ld + c];
      melt_r[i] = soilcrop ? (double)S->swe_old[(int64_t)i * ld + c] : (double)S->frac_iceold[(int64_t)i * ld + c];
      snofrz_r[i] = S->qflx_snofrz_lyr[(int64_t)i * ld + c];
    }
  }
  const double qflx_snwcp_ice = S->qflx_snwcp_ice[c], qflx_snow_grnd = S->qflx_snow_grnd[c];

  snow_water(K, do_capsnow, snl, dtime, frac_sno_eff, h2osno, qflx_sub_snow, qflx_evap_grnd, qflx_dew_snow,
             qflx_dew_grnd, qflx_rain_grnd, qflx_snomelt, qflx_snow_melt, qflx_top_soil, int_snow, frac_sno,
             mflx_neg_snow, err);
  S->qflx_snow_melt[c] = qflx_snow_melt;
  S->qflx_top_soil[c] = qflx_top_soil;
  S->mflx_neg_snow[c] = mflx_neg_snow;

  // compute_aerosol_deposition (aerosol_physics_impl.hh:36-64): the top snow layer receives the deposition of the step
  if (snl > 0) {
    const int j = NLEVSNO - snl;
    AT(K.m[0], j) += (dep0 * dtime);
    AT(K.m[1], j) += ((dep1a + dep1b) * dtime);
    AT(K.m[2], j) += ((dep2a + dep2b) * dtime);
    AT(K.m[3], j) += ((dep3a + dep3b) * dtime);
    AT(K.m[4], j) += ((dep4a + dep4b) * dtime);
    AT(K.m[5], j) += ((dep5a + dep5b) * dtime);
  }

  // aerosol_phase_change (:502-548): sublimation moves within-ice black carbon to the external state, top layer only
  {
    const int top = NLEVSNO - snl;
    const double subsnow = dmax(0.0, (qflx_sub_snow * dtime));
    const double w = AT(K.liq, top) + AT(K.ice, top);
    double frc_sub;
    if (w > 0.0) {
      frc_sub = subsnow / w;
    } else {
      frc_sub = 0.0;
    }
    for (int i = top; i < NLEVSNO; ++i) {
      if (i != top) frc_sub = 0.0;
      double frc_transfer = frc_sub;
      if (frc_transfer > 1.0) frc_transfer = 1.0;
      const double m = AT(K.m[0], i);
      const double dm_int = m * frc_transfer;
      AT(K.m[0], i) = m - dm_int;
      AT(K.m[1], i) += dm_int;
    }
  }

  // transpiration (transpiration_impl.hh:15-28; nlevsoi = 10)
  if (veg_active) {
#pragma unroll
    for (int i = 0; i < 10; ++i) S->qflx_rootsoi[(int64_t)i * ld + c] = rootr[i] * qflx_tran_veg;
  }

  snow_compaction(K, snl, L.ltype, dtime, int_snow, n_melt, frac_sno, imelt_r, melt_r);

  double qflx_sl_top_soil, qflx_snow2topsoi, mflx_snowlyr_col;
  combine_layers(K, L.urbpoi != 0, L.ltype, dtime, snl, h2osno, snow_depth, frac_sno_eff, frac_sno, int_snow, qflx_sl_top_soil,
                 qflx_snow2topsoi, mflx_snowlyr_col, err);
  divide_layers(K, frac_sno, snl, err);

  // prune_snow_layers (:1327-1349)
  {
    const int top = NLEVSNO - snl;
#pragma unroll
    for (int i = 0; i < NLEVSNO; ++i) {
      if (i < top) {
        AT(K.ice, i) = 0.0;
        AT(K.liq, i) = 0.0;
        AT(K.t, i) = 0.0;
        AT(K.dz, i) = 0.0;
        AT(K.z, i) = 0.0;
        AT(K.zi, i) = 0.0;
      }
    }
  }
  S->snl[c] = snl;
  S->h2osno[c] = h2osno;
  S->snow_depth[c] = snow_depth;
  S->frac_sno_eff[c] = frac_sno_eff;
  S->frac_sno[c] = frac_sno;
  S->int_snow[c] = int_snow;
  S->qflx_sl_top_soil[c] = qflx_sl_top_soil;
  S->qflx_snow2topsoi[c] = qflx_snow2topsoi;
  S->mflx_snowlyr_col[c] = mflx_snowlyr_col;

  // update_aerosol_mass_and_concen (aerosol_physics_impl.hh:10-31, :67-106)
  {
    const int snotop = NLEVSNO - snl;
    const dfield cnc[NAER] = {S->cnc_bcphi + c, S->cnc_bcpho + c, S->cnc_dst1 + c, S->cnc_dst2 + c, S->cnc_dst3 + c, S->cnc_dst4 + c};
#pragma unroll
    for (int sl = 0; sl < NLEVSNO; sl++) {
      const double snowmass = (sl < snotop) ? 1.e-12 : AT(K.ice, sl) + AT(K.liq, sl);
      const double scl = (sl == snotop && do_capsnow) ? (snowmass / (snowmass + qflx_snwcp_ice * dtime)) : (sl < snotop) ? 0.0 : 1.0;
      const double snwmss_inv = 1.0 / snowmass;
#pragma unroll
      for (int a = 0; a < NAER; a++) {
        const double m = AT(K.m[a], sl) * scl;
        AT(K.m[a], sl) = m;
        cnc[a][(int64_t)sl * ld] = m * snwmss_inv;
      }
    }
  }

  snow_aging(S, K, do_capsnow, snl, frac_sno, dtime, qflx_snwcp_ice, qflx_snow_grnd, h2osno, snofrz_r, err);
  if (err) S->err_flags[c] |= err;
  // the pack: LDS -> global rows.  Of level 5 only liquid and ice can have changed; temperature, thickness and interface
  // depth of the top soil layer are read-only here.
#pragma unroll
  for (int f = 0; f < 5; f++)
#pragma unroll
    for (int i = 0; i < (f < 2 ? 6 : 5); i++) g6[f][(int64_t)i * ld] = s_pack[f * 6 + i][lane];
#pragma unroll
  for (int f = 0; f < 2 + NAER; f++)
#pragma unroll
    for (int i = 0; i < 5; i++) g5[f][(int64_t)i * ld] = s_pack[SN_ROW_Z + f * 5 + i][lane];
}

void launch_snow_hydrology(const DevState* S, int64_t n, double dt, hipStream_t st)
{
  if (n > 0) hipLaunchKernelGGL(k_snow_hydrology, dim3((unsigned)((n + SN_WG - 1) / SN_WG)), dim3(SN_WG), 0, st, S, dt);
}

}  // namespace elmk
