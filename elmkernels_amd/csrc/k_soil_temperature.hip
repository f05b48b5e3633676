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
// more scratch Views in memory.  Here ONE launch does the wrapper, one thread per column going down it once: the raw
// inputs of a level arrive as one row of coalesced loads two levels ahead of their use, its thermal properties (the
// transcendental work: two pow and a log10 per soil level), its matrix factor, the diffusive fluxes across its faces
// and its row of the 21-row, 5-band system (snow layers, standing surface water, soil layers) are formed from a
// sliding window of three levels and pushed straight into the forward sweep, whose A and Z stay in LDS; the back
// substitution leaves the new temperatures in the same LDS rows, and phase change (its three loops fused level by
// level) is the only pass that writes temperature, ice and liquid back.  Per column the kernel reads each input once
// (ice, liquid and the matrix factor a second time for phase change: no on-chip room for them) and writes each output
// once - 3.5 KB measured against the 2.9 KB tally of DESIGN.md section 9.
// All level loops are rolled and unrolled by two - the level index is the same for every lane, so its branches are
// uniform and every access is a coalesced row of the SoA state; the two alternating register sets of the unrolled loops
// are the software pipeline (st_load_level).  A workgroup is eight waves sharing one copy of the math tables:
// 2 x 19 rows x 512 lanes x 8 B + 7 KB = 159 KB of LDS, one workgroup per CU, two waves per SIMD with no spill.
// The number of active snow layers differs from lane to lane, so loops that the reference starts at the top active
// layer run over all rows with a predicate, and rows above the snow pack enter the forward sweep as identity rows:
// with their A, B, Z equal to zero the general recurrence reproduces the reference's special first and second rows
// exactly (x - 0*y == x), so the solve is bit-identical to the reference's whatever the layer count.
#define ELMK_MATH_LDS 1  // exp / log / pow tables of elmk_math.h in LDS: every kernel below that evaluates them calls elmk_math_lds_init first
#include "elmk_dev.h"
#include "elmk_kernels.h"

namespace elmk {

#define LV(f, lev) S->f[(int64_t)(lev) * ld + c]
// a level value this kernel reads exactly once (ST_NT_ONCE: with the nontemporal hint; the values phase change reads a second
// time - liquid, ice, thickness, porosity - must stay cached: with the hint on every load the solve is 16-30 % slower)
#ifndef ST_NT_ONCE
#define ST_NT_ONCE 0
#endif
#if ST_NT_ONCE && !defined(ELMK_STATE_F32)
#define LVN(f, lev) __builtin_nontemporal_load(&S->f[(int64_t)(lev) * ld + c])
#else
#define LVN(f, lev) LV(f, lev)
#endif

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
  return solar_abg + dlrad + (1.0 - frac_veg_nosno) * emg * forc_lwrad - emg * STEBOL * elmk_pow(temp, 4.0) -
         (eflx_sh + qflx_ev * htvp);
}

// what the sweep reads of one level, issued as one row of loads three levels ahead of its use (the level index is the same
// for every lane, so the soil-only fields sit behind a uniform branch; levels above the snow pack are never read)
struct StLevIn {
  double liq, ice, dz, t, z, zi, sabg, watsat, tkdry, tkmg, csol;
};

__device__ __forceinline__ StLevIn st_load_level(const DevState* __restrict__ S, const int64_t c, const int64_t ld, const int i,
                                                 const int top)
{
  // No branch around any load: the compiler can only count the loads in flight (s_waitcnt vmcnt) along straight-line code.
  // Levels past the bottom re-read the bottom level, lanes whose snow pack starts below level i read the top soil level
  // (always there, and theirs to read a few trips later anyway); none of those values is used.
  const int lv = (i < NLEVTOT) ? i : NLEVTOT - 1;
  const int li = (lv >= top) ? lv : NLEVSNO;
  const int js = (li >= NLEVSNO) ? li - NLEVSNO : 0;
  const int lsab = (li <= NLEVSNO) ? li : NLEVSNO;
  const int lcs = (li >= NLEVSNO) ? li : NLEVSNO;
  StLevIn L;
  L.liq = LV(h2osoi_liq, li);
  L.ice = LV(h2osoi_ice, li);
  L.dz = LV(dz, li);
  L.t = LVN(t_soisno, li);
  L.z = LVN(zsoi, li);
  L.zi = LVN(zisoi, li);
  L.sabg = LVN(sabg_lyr, lsab);
  L.watsat = LV(watsat, js);
  L.tkdry = LVN(tkdry, js);
  L.tkmg = LVN(tkmg, js);
  L.csol = LVN(csol, lcs);
  return L;
}

// thermal conductivity and heat capacity of level i (soil_thermal_properties_impl.hh: calc_soil_tk :20 and
// calc_soil_heat_capacity :163 with the wrapper's ltype == 1, calc_snow_tk :96, calc_snow_heat_capacity :206).
// i is the same for every lane (the level loops are rolled), so the branches on it are uniform.
__device__ __forceinline__ void st_level_props(const StLevIn& L, const int i, const int top, const int snl, const double frac_sno,
                                               const double h2osno, double& thk, double& cv)
{
  const double liq = L.liq, ice = L.ice, dzl = L.dz;
  if (i >= NLEVSNO) {
    const double watsat = L.watsat;
    double satw = (liq / DENH2O + ice / DENICE) / (dzl * watsat);
    satw = dmin(1.0, satw);
    const double tkdry = L.tkdry;
    if (satw > 1.0e-6) {
      double dke;
      if (L.t >= TFRZ) {
        dke = dmax(0.0, elmk_log10(satw) + 1.0);
      } else {
        dke = satw;
      }
      const double fl = (liq / (DENH2O * dzl)) / (liq / (DENH2O * dzl) + ice / (DENICE * dzl));
      const double dksat = L.tkmg * elmk_pow_literal_base(ST_TKWAT, fl * watsat) * elmk_pow_literal_base(ST_TKICE, (1.0 - fl) * watsat);
      thk = dke * dksat + (1.0 - dke) * tkdry;
    } else {
      thk = tkdry;
    }
    if (i >= NLEVSNO + NLEVBED) thk = ST_TKBDRK;
    cv = L.csol * (1.0 - watsat) * dzl + (ice * ST_CPICE + liq * ST_CPWAT);
    if (i == NLEVSNO && snl == 0 && h2osno > 0.0) cv += ST_CPICE * h2osno;
  } else {
    if (i < top) {
      thk = 0.0;
      cv = 0.0;
    } else {
      const double bw = (ice + liq) / (frac_sno * dzl);
      thk = ST_TKAIR + (7.75e-5 * bw + 1.105e-6 * bw * bw) * (ST_TKICE - ST_TKAIR);
      if (frac_sno > 0.0) {
        cv = dmax(ST_THIN_SFCLAYER, (ST_CPWAT * liq + ST_CPICE * ice) / frac_sno);
      } else {
        cv = ST_THIN_SFCLAYER;
      }
    }
  }
}

// state of the forward sweep of solver::PDMA (pentadiagonal_solver_impl.hh:16-76).  A and Z of rows 0..18 stay in LDS,
// [row][lane] (the row loops are rolled, a register array would need dynamic indexing; [row][lane] is conflict-free).
// The back substitution overwrites Z(r) with the solution of row r, so the new temperatures never leave the CU
// before phase change has had them.  B = l0 * U1 is zero in every row but one - only the snow layer next to the
// ground has a second superdiagonal entry (l0, get_matrix_snow_soil) - so that one value stays in a register and the
// back substitution multiplies by a literal zero elsewhere (the same arithmetic as the reference's 0 * U1 product for
// every finite solution).  The recurrence itself only needs the last two rows, kept here.
#ifndef ST_WG_N
#define ST_WG_N 512  // (64 / 128 / 192 measured in round 4: profiles/r04_soil_workgroup_ab.txt)
#endif
constexpr int ST_WG = ST_WG_N;
constexpr int ST_LDS_ROWS = NROW - 2;
typedef double StLds[ST_LDS_ROWS][ST_WG];
struct StSweep {
  double B4;  // B of row NLEVSNO - 1
  double Am2, Am1, Bm2, Bm1, Zm2, Zm1;
  double Y1, U1, r19, l4_19, A19;  // kept from the second row from the bottom for the reference's form of the last two
};

// one row of the forward sweep.  Rows above the snow pack arrive as identity rows; with their A, B, Z equal to zero
// the general recurrence gives the reference's special first and second rows exactly.
__device__ __forceinline__ void st_push_row(StSweep& w, StLds& sA, StLds& sZ, const int lane, const int r, const double l0,
                                            const double l1, const double l2, const double l3, const double l4, const double rr)
{
  constexpr int N = NROW;
  if (r < N - 2) {
    const double Y1 = l3 - w.Am2 * l4;
    const double U1 = 1.0 / (l2 - w.Bm2 * l4 - w.Am1 * Y1);
    const double a = (l1 - w.Bm1 * Y1) * U1;
    const double b = l0 * U1;
    const double z = (rr - w.Zm2 * l4 - w.Zm1 * Y1) * U1;
    sA[r][lane] = a;
    sZ[r][lane] = z;
    if (r == NLEVSNO - 1) w.B4 = b;
    w.Am2 = w.Am1;
    w.Am1 = a;
    w.Bm2 = w.Bm1;
    w.Bm1 = b;
    w.Zm2 = w.Zm1;
    w.Zm1 = z;
  } else if (r == N - 2) {  // (:55-58); Am1 = A(N-3), Am2 = A(N-4) here, likewise B
    w.Y1 = l3 - w.Am2 * l4;
    w.U1 = 1.0 / (l2 - w.Bm2 * l4 - w.Am1 * w.Y1);
    w.A19 = (l1 - w.Bm1 * w.Y1) * w.U1;
    w.r19 = rr;
    w.l4_19 = l4;
  } else {  // bottom row (:61-66); Z(N-2) in the reference's own form, with Z(N-3) in both products
    const double Y2 = l3 - w.Am1 * l4;
    const double U2 = 1.0 / (l2 - w.Bm1 * l4 - w.A19 * Y2);
    const double z19 = (w.r19 - w.Zm1 * w.l4_19 - w.Zm1 * w.Y1) * w.U1;
    const double z20 = (rr - z19 * l4 - z19 * Y2) * U2;
    w.Zm2 = z19;  // handed to the back substitution
    w.Zm1 = z20;
  }
}

// what phase_change_soisno reads of one level beside its temperature (which is in LDS); the inputs of the supercooled-water
// limit are only read for soil levels below freezing
struct StPcIn {
  double ice, liq, fact, watsat, sucsat, bsw, dz;
};

__device__ __forceinline__ StPcIn st_load_pc(const DevState* __restrict__ S, const int64_t c, const int64_t ld, const int i,
                                             const int top, const double t)
{
  // branch-free like st_load_level; lanes that do not need the supercooled-water inputs all read element 0 (one line)
  const int lv = (i < NLEVTOT) ? i : NLEVTOT - 1;
  const int li = (lv >= top) ? lv : NLEVSNO;
  const bool need = (lv >= NLEVSNO) && (t < TFRZ);
  const int64_t js = need ? (int64_t)(li - NLEVSNO) * ld + c : 0;
  const int64_t jd = need ? (int64_t)li * ld + c : 0;
  StPcIn P;
  P.ice = LV(h2osoi_ice, li);
  P.liq = LV(h2osoi_liq, li);
  P.fact = LV(fact, li);
  P.watsat = S->watsat[js];
  P.sucsat = S->sucsat[js];
  P.bsw = S->bsw[js];
  P.dz = S->dz[jd];
  return P;
}

__global__ __launch_bounds__(ST_WG, 2) void k_soil_temperature(const DevState* __restrict__ S, double dtime)
{
  __shared__ StLds sA, sZ;
  elmk_math_lds_init<false>();  // pow and log10 in the thermal properties, the surface heat fluxes and the supercooled-water limit
  const int lane = (int)threadIdx.x;
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t ld = S->ld;
  if (c >= S->ncols) return;
  // (the wrapper's "dummy ltype" is 1 - istsoil - for every column, soil_temperature_kokkos.cc:77-79)
  const int snl = S->snl[c];
  const int top = NLEVSNO - snl;
  // the first three levels' inputs: in flight while the column scalars arrive and the surface fluxes are formed
  const StLevIn in0 = st_load_level(S, c, ld, 0, top);
  StLevIn inX = st_load_level(S, c, ld, 1, top);
  StLevIn inY = st_load_level(S, c, ld, 2, top);
  const double frac_sno = S->frac_sno[c], frac_sno_eff = S->frac_sno_eff[c], frac_h2osfc = S->frac_h2osfc[c];
  const double h2osfc0 = S->h2osfc[c], h2osno0 = S->h2osno[c];
  const double onemcn = 1.0 - ST_CNFAC;

  // ---- surface_heat_fluxes (:121-142)
  double t_h2osfc = S->t_h2osfc[c];
  double hs_soil, hs_h2osfc, hs_top_snow, dhsdT;
  {
    const int fvn = S->frac_veg_nosno[c];
    const double dlrad = S->dlrad[c], emg = S->emg[c], forc_lwrad = S->forc_lwrad[c], htvp = S->htvp[c];
    const double sabg_soil = S->sabg_soil[c];
    const double t_soi0 = LV(t_soisno, NLEVSNO);
    S->sabg_chk[c] = frac_sno_eff * S->sabg_snow[c] + (1.0 - frac_sno_eff) * sabg_soil;
    hs_soil = st_surface_heat_flux(fvn, dlrad, emg, forc_lwrad, htvp, sabg_soil, t_soi0, S->eflx_sh_soil[c],
                                   S->qflx_ev_soil[c]);
    hs_h2osfc = st_surface_heat_flux(fvn, dlrad, emg, forc_lwrad, htvp, sabg_soil, t_h2osfc, S->eflx_sh_h2osfc[c],
                                     S->qflx_ev_h2osfc[c]);
    hs_top_snow = st_surface_heat_flux(fvn, dlrad, emg, forc_lwrad, htvp, LV(sabg_lyr, top), LV(t_soisno, top),
                                       S->eflx_sh_snow[c], S->qflx_ev_snow[c]);
    dhsdT = -S->cgrnd[c] - 4.0 * emg * STEBOL * elmk_pow(S->t_grnd[c], 3.0);
  }
  // heat capacity and height of standing surface water (soil_thermal_properties_impl.hh:255-279)
  double c_h2osfc, dz_h2osfc;
  if ((h2osfc0 > ST_THIN_SFCLAYER) && (frac_h2osfc > ST_THIN_SFCLAYER)) {
    c_h2osfc = dmax(ST_THIN_SFCLAYER, ST_CPWAT * h2osfc0 / frac_h2osfc);
    dz_h2osfc = dmax(ST_THIN_SFCLAYER, 1.0e-3 * h2osfc0 / frac_h2osfc);
  } else {
    c_h2osfc = ST_THIN_SFCLAYER;
    dz_h2osfc = ST_THIN_SFCLAYER;
  }

  // ---- one pass down the column: thermal properties, matrix factor, diffusive fluxes, the row of the system
  //      (rows 0..4 snow, row 5 standing surface water, rows 6..20 soil; band 0 = 2nd superdiagonal, 1 = 1st
  //      superdiagonal, 2 = diagonal, 3 = 1st subdiagonal, 4 = 2nd subdiagonal) and the forward sweep.  Only the
  //      sweep's recurrence stays in registers; level quantities live in a sliding window (previous, current, next),
  //      and the raw inputs of levels i + 2 .. i + 4 are the loads in flight.
  StSweep w;
  w.B4 = 0.0;
  w.Am2 = w.Am1 = w.Bm2 = w.Bm1 = w.Zm2 = w.Zm1 = 0.0;
  w.Y1 = w.U1 = w.r19 = w.l4_19 = w.A19 = 0.0;
  double fact_sl1 = 0.0;  // matrix factor of the snow layer next to the ground (for phase_change_h2osfc)
  double thk_cur, cv_cur;
  st_level_props(in0, 0, top, snl, frac_sno, h2osno0, thk_cur, cv_cur);
  double z_cur = in0.z, t_cur = in0.t, zi_cur = in0.zi, dz_cur = in0.dz, sabg_cur = in0.sabg;
  double z_prev = 0.0, tk_prev = 0.0, fn_prev = 0.0;
  // One level of the pass.  nx holds the inputs of level i + 1; as soon as they are consumed the same registers take the
  // loads of level i + 3.  Two such sets alternate (the loop below is unrolled by two), so a set is never copied - a copy
  // would have to wait for the loads it copies - and the rows of levels i + 2 and i + 3 are in flight during level i.
  auto sweep_level = [&](const int i, StLevIn& nx) __attribute__((always_inline)) {
    double thk_nxt = 0.0, cv_nxt = 0.0;
    if (i + 1 < NLEVTOT) st_level_props(nx, i + 1, top, snl, frac_sno, h2osno0, thk_nxt, cv_nxt);
    const double z_nxt = nx.z, t_nxt = nx.t, zi_nxt = nx.zi, dz_nxt = nx.dz, sabg_nxt = nx.sabg;
    nx = st_load_level(S, c, ld, i + 3, top);
    // calc_face_tk (:132), calc_diffusive_heat_flux (soil_temperature_impl.hh:45): interface i | i+1
    double tk_i = 0.0, fn_i = 0.0;
    if (i < NLEVTOT - 1 && i >= top) {
      const double zi1 = zi_nxt;
      tk_i = thk_cur * thk_nxt * (z_nxt - z_cur) / (thk_cur * (z_nxt - zi1) + thk_nxt * (zi1 - z_cur));
      fn_i = tk_i * (t_nxt - t_cur) / (z_nxt - z_cur);
    }
    // calc_heat_flux_matrix_factor (:93)
    double fact_i;
    if (i < top) {
      fact_i = 0.0;
    } else if (i == top) {
      const double zit = zi_cur;
      fact_i = dtime / cv_cur * dz_cur / (0.5 * (z_cur - zit + ST_CAPR * (z_nxt - zit)));
    } else {
      fact_i = dtime / cv_cur;
    }
    LV(fact, i) = fact_i;
    if (i == NLEVSNO - 1) fact_sl1 = fact_i;
    double l0 = 0.0, l1 = 0.0, l2 = 0.0, l3 = 0.0, l4 = 0.0, rr = 0.0;
    if (i < NLEVSNO) {  // snow rows: get_rhs_snow, get_matrix_snow, get_matrix_snow_soil
      if (i < top) {
        l2 = 1.0;  // identity row
      } else if (i == top) {
        const double dzp = z_nxt - z_cur;
        rr = t_cur + fact_i * (hs_top_snow - dhsdT * t_cur + ST_CNFAC * fn_i);
        l2 = 1.0 + onemcn * fact_i * tk_i / dzp - fact_i * dhsdT;
        if (snl > 1) l1 = -onemcn * fact_i * tk_i / dzp;
      } else {
        const double dzm = z_cur - z_prev;
        const double dzp = z_nxt - z_cur;
        rr = t_cur + ST_CNFAC * fact_i * (fn_i - fn_prev) + fact_i * sabg_cur;
        l3 = -onemcn * fact_i * tk_prev / dzm;
        l2 = 1.0 + onemcn * fact_i * (tk_i / dzp + tk_prev / dzm);
        if (i != NLEVSNO - 1) l1 = -onemcn * fact_i * tk_i / dzp;
      }
      if (i == NLEVSNO - 1 && snl > 0) l0 = -onemcn * fact_i * tk_i / (z_nxt - z_cur);
      st_push_row(w, sA, sZ, lane, i, l0, l1, l2, l3, l4, rr);
    } else {
      double tk_h2osfc = 0.0;
      if (i == NLEVSNO) {  // calc_h2osfc_tk (:238); the standing-surface-water row: get_rhs_ssw, get_matrix_ssw(_soil)
        const double zh2osfc = 1.0e-3 * (0.5 * h2osfc0);
        tk_h2osfc = ST_TKWAT * thk_cur * (z_cur + zh2osfc) / (ST_TKWAT * z_cur + thk_cur * zh2osfc);
        const double dzw = 0.5 * dz_h2osfc + z_cur;
        const double fn_h2osfc = tk_h2osfc * (t_cur - t_h2osfc) / dzw;
        const double rw = t_h2osfc + (dtime / c_h2osfc) * (hs_h2osfc - dhsdT * t_h2osfc + ST_CNFAC * fn_h2osfc);
        const double w2 = 1.0 + onemcn * (dtime / c_h2osfc) * tk_h2osfc / dzw - (dtime / c_h2osfc) * dhsdT;
        const double w1 = -onemcn * (dtime / c_h2osfc) * tk_h2osfc / dzw;
        st_push_row(w, sA, sZ, lane, NLEVSNO, 0.0, w1, w2, 0.0, 0.0, rw);
      }
      // soil rows: get_rhs_soil, get_matrix_soil, get_matrix_soil_snow, get_matrix_soil_ssw
      if (i == NLEVSNO) {
        const double dzp = z_nxt - z_cur;
        if (snl == 0) {
          rr = t_cur + fact_i * (hs_top_snow - dhsdT * t_cur + ST_CNFAC * fn_i);
          l2 = 1.0 + onemcn * fact_i * tk_i / dzp - fact_i * dhsdT;
          l1 = -onemcn * fact_i * tk_i / dzp;
        } else {
          const double dzm = z_cur - z_prev;
          rr = t_cur + fact_i * ((1.0 - frac_sno_eff) * (hs_soil - dhsdT * t_cur) + ST_CNFAC * (fn_i - frac_sno_eff * fn_prev));
          rr += frac_sno_eff * fact_i * sabg_cur;
          l2 = 1.0 + onemcn * fact_i * (tk_i / dzp + frac_sno_eff * tk_prev / dzm) - (1.0 - frac_sno_eff) * fact_i * dhsdT;
          l1 = -onemcn * fact_i * tk_i / dzp;
          l4 = -frac_sno_eff * onemcn * fact_i * tk_prev / dzm;
        }
        if (frac_h2osfc != 0.0) {
          const double dzm = 0.5 * dz_h2osfc + z_cur;
          l2 += frac_h2osfc * (onemcn * fact_i * tk_h2osfc / dzm + fact_i * dhsdT);
          l3 = -frac_h2osfc * onemcn * fact_i * tk_h2osfc / (0.5 * dz_h2osfc + z_cur);
        }
      } else if (i < NLEVTOT - 1) {
        const double dzm = z_cur - z_prev;
        const double dzp = z_nxt - z_cur;
        rr = t_cur + ST_CNFAC * fact_i * (fn_i - fn_prev);
        l3 = -onemcn * fact_i * tk_prev / dzm;
        l2 = 1.0 + onemcn * fact_i * (tk_i / dzp + tk_prev / dzm);
        l1 = -onemcn * fact_i * tk_i / dzp;
      } else {
        const double dzm = z_cur - z_prev;
        rr = t_cur - ST_CNFAC * fact_i * fn_prev + fact_i * fn_i;
        l3 = -onemcn * fact_i * tk_prev / dzm;
        l2 = 1.0 + onemcn * fact_i * tk_prev / dzm;
      }
      st_push_row(w, sA, sZ, lane, i + 1, l0, l1, l2, l3, l4, rr);
    }
    z_prev = z_cur;
    z_cur = z_nxt;
    t_cur = t_nxt;
    zi_cur = zi_nxt;
    dz_cur = dz_nxt;
    sabg_cur = sabg_nxt;
    tk_prev = tk_i;
    fn_prev = fn_i;
    thk_cur = thk_nxt;
    cv_cur = cv_nxt;
  };
#pragma unroll 1
  for (int i = 0; i < NLEVTOT; i += 2) {
    sweep_level(i, inX);
    sweep_level(i + 1, inY);
  }

  // ---- back substitution (:69-74) and update_temperature (soil_temperature_impl.hh:154-177): row r holds level r for
  //      the snow layers, standing surface water for r == 5, level r - 1 below.  The solution of row r replaces Z(r) in
  //      LDS (rows 19 and 20 - levels 18, 19 - stay in registers); the phase-change pass takes the temperatures from there
  //      and is the one that writes them to the state.
  double t_soi0, t_ssw;
  const double x20 = w.Zm1;                    // R(N-1) = Z(N-1)
  const double x19 = w.Zm2 - w.A19 * x20;      // R(N-2) = Z(N-2) - A(N-2) R(N-1)
  {
    double r2 = x20, r1 = x19;
    t_soi0 = 0.0;
    t_ssw = 0.0;
#pragma unroll 1
    for (int r = NROW - 3; r >= 0; --r) {
      const double x = sZ[r][lane] - sA[r][lane] * r1 - ((r == NLEVSNO - 1) ? w.B4 : 0.0) * r2;
      r2 = r1;
      r1 = x;
      sZ[r][lane] = x;
      if (r == NLEVSNO + 1) t_soi0 = x;
      if (r == NLEVSNO) t_ssw = x;
    }
  }
  t_h2osfc = (frac_h2osfc != 0.0) ? t_ssw : t_soi0;
  // the snow layer next to the ground (new value if active, else unchanged)
  double t_sl1 = (NLEVSNO - 1 >= top) ? sZ[NLEVSNO - 1][lane] : LV(t_soisno, NLEVSNO - 1);
  double ice_sl1 = LV(h2osoi_ice, NLEVSNO - 1);
  // temperature of level i after the solve (levels >= top)
  // (one LDS read from a clamped row, then a value select: a select between an LDS row and a register would go through
  //  a generic pointer)
  auto st_tnew = [&](const int i) __attribute__((always_inline)) {
    const int r = i < NLEVSNO ? i : (i < NLEVTOT - 2 ? i + 1 : ST_LDS_ROWS - 1);
    const double v = sZ[r][lane];
    return i == NLEVTOT - 2 ? x19 : (i == NLEVTOT - 1 ? x20 : v);
  };
#define ST_TNEW(i) st_tnew(i)

  // the first two levels of the phase-change pass: in flight across phase_change_h2osfc
  StPcIn pX = st_load_pc(S, c, ld, 0, top, 0.0);
  StPcIn pY = st_load_pc(S, c, ld, 1, top, 0.0);

  // ---- phase_change_h2osfc (phase_change_impl.hh:11-151)
  double h2osfc = h2osfc0, h2osno = h2osno0, int_snow = S->int_snow[c], snow_depth = S->snow_depth[c];
  {
    double qflx_h2osfc_to_ice = 0.0, eflx_h2osfc_to_snow = 0.0, xmf_h2osfc = 0.0;
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
        if (snl > 0) ice_sl1 -= xm;
        h2osfc += xm;
        xmf_h2osfc = hm;
        qflx_h2osfc_to_ice = -xm / dtime;
        if (frac_sno > 0 && snl > 0) {
          snow_depth = h2osno / (rho_avg * frac_sno);
        } else {
          snow_depth = h2osno / DENICE;
        }
        if (snl == 0) {
          t_sl1 = t_h2osfc;
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
          t_sl1 = (c1 * t_sl1 + c2 * t_h2osfc) / (c1 + c2);
          eflx_h2osfc_to_snow = (t_h2osfc - t_sl1) * c2 / dtime;
        }
      } else {
        rho_avg = (h2osno * rho_avg + h2osfc * DENICE) / (h2osno + h2osfc);
        h2osno += h2osfc;
        int_snow += h2osfc;
        qflx_h2osfc_to_ice = h2osfc / dtime;
        if (snl > 0) ice_sl1 = ice_sl1 + h2osfc;
        t_h2osfc = t_h2osfc - temp1 * HFUS / (dtime * dhsdT - c_h2osfc);
        xmf_h2osfc = hm - frac_h2osfc * temp1 * HFUS / dtime;
        double c1, c2;
        if (snl == 0) {
          t_sl1 = t_h2osfc;
        } else if (snl == 1) {
          c1 = frac_sno * (dtime / fact_sl1 - dhsdT * dtime);
          if (frac_h2osfc != 0.0) {
            c2 = frac_h2osfc * (c_h2osfc - dtime * dhsdT);
          } else {
            c2 = 0.0;
          }
          t_sl1 = (c1 * t_sl1 + c2 * t_h2osfc) / (c1 + c2);
          t_h2osfc = t_sl1;
        } else {
          c1 = frac_sno / fact_sl1 * dtime;
          if (frac_h2osfc != 0.0) {
            c2 = frac_h2osfc * (c_h2osfc - dtime * dhsdT);
          } else {
            c2 = 0.0;
          }
          t_sl1 = (c1 * t_sl1 + c2 * t_h2osfc) / (c1 + c2);
          t_h2osfc = t_sl1;
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

  // ---- phase_change_soisno (phase_change_impl.hh:182-418).  The reference identifies melting / freezing layers in
  //      two loops and then processes all layers; every step of a layer only reads that layer (and column scalars
  //      that level 5 alone modifies after its own identification), so the three loops are fused level by level.
  double xmf = 0.0, qflx_snofrz = 0.0, qflx_snow_melt = 0.0, qflx_snomelt = 0.0;
  double t_top_new = 0.0;  // temperature of the top active layer after phase change (for update_t_grnd)
  // level 4 as phase_change_h2osfc left it (it writes that level even when there is no snow layer)
  LV(t_soisno, NLEVSNO - 1) = t_sl1;
  LV(h2osoi_ice, NLEVSNO - 1) = ice_sl1;
  // one level; px holds its inputs and is reloaded with those of level i + 2 (two alternating sets, as in the sweep)
  auto pc_level = [&](const int i, StPcIn& px) __attribute__((always_inline)) {
    const StPcIn P = px;
    {
      const int ip = i + 2;
      px = st_load_pc(S, c, ld, ip, top, (ip >= NLEVSNO && ip < NLEVTOT) ? ST_TNEW(ip) : 0.0);
    }
    if (i < top) {
      if (i < NLEVSNO) LV(qflx_snofrz_lyr, i) = 0.0;
      return;
    }
    double t = ST_TNEW(i);
    double ice = P.ice, liq = P.liq;
    if (i == NLEVSNO - 1) {  // as phase_change_h2osfc left it
      t = t_sl1;
      ice = ice_sl1;
    }
    const double fact_i = P.fact;
    int imelt = 0;
    double tinc = 0.0, supercool = 0.0;
    if (i < NLEVSNO) {  // snow (:222-238)
      if (ice > 0.0 && t > TFRZ) {
        imelt = 1;
        tinc = TFRZ - t;
        t = TFRZ;
      }
      if (liq > 0.0 && t < TFRZ) {
        imelt = 2;
        tinc = TFRZ - t;
        t = TFRZ;
      }
    } else {  // soil (:241-273); ltype == istsoil: Zhao (1997) / Koren (1999) supercooled water
      if (ice > 0.0 && t > TFRZ) {
        imelt = 1;
        tinc = TFRZ - t;
        t = TFRZ;
      }
      if (t < TFRZ) {
        const double smp = HFUS * (TFRZ - t) / (GRAV * t) * 1000.0;
        supercool = P.watsat * elmk_pow(smp / P.sucsat, -1.0 / P.bsw);
        supercool *= P.dz * 1000.0;
      }
      if (liq > supercool && t < TFRZ) {
        imelt = 2;
        tinc = TFRZ - t;
        t = TFRZ;
      }
      if (snl == 0 && h2osno > 0.0 && i == NLEVSNO) {
        if (t > TFRZ) {
          imelt = 1;
          tinc = TFRZ - t;
          t = TFRZ;
        }
      }
    }
    // energy surplus / deficit and the rate of melting / freezing (:277-409)
    double hm = 0.0;
    if (imelt > 0) {
      if (i == top) {
        if (i < NLEVSNO) {
          hm = frac_sno_eff * (dhsdT * tinc - tinc / fact_i);
        } else {
          const double temp_hm = dhsdT * tinc - tinc / fact_i;
          hm = (frac_h2osfc != 0.0) ? temp_hm - frac_h2osfc * (dhsdT * tinc) : temp_hm;
        }
      } else if (i == NLEVSNO) {
        hm = (1.0 - frac_sno_eff - frac_h2osfc) * dhsdT * tinc - tinc / fact_i;
      } else {
        if (i < NLEVSNO) {
          hm = -frac_sno_eff * (tinc / fact_i);
        } else {
          hm = -tinc / fact_i;
        }
      }
    }
    if (imelt == 1 && hm < 0.0) {
      hm = 0.0;
      imelt = 0;
    }
    if (imelt == 2 && hm > 0.0) {
      hm = 0.0;
      imelt = 0;
    }
    double frz = 0.0;
    const bool moved = imelt > 0 && fabs(hm) > 0.0;
    if (moved) {
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
      const double wmass0 = ice + liq;
      const double wice0 = ice;
      if (xm > 0.0) {
        ice = dmax(0.0, wice0 - xm);
        heatr = hm - HFUS * (wice0 - ice) / dtime;
      } else if (xm < 0.0) {
        if (i < NLEVSNO) {
          ice = dmin(wmass0, wice0 - xm);
        } else {
          if (wmass0 < supercool) {
            ice = 0.0;
          } else {
            ice = dmin(wmass0 - supercool, wice0 - xm);
          }
        }
        heatr = hm - HFUS * (wice0 - ice) / dtime;
      }
      liq = dmax(0.0, wmass0 - ice);
      if (fabs(heatr) > 0.0) {
        if (i == top) {
          if (snl == 0) {
            t += fact_i * heatr / (1.0 - (1.0 - frac_h2osfc) * fact_i * dhsdT);
          } else {
            t += (fact_i / frac_sno_eff) * heatr / (1.0 - fact_i * dhsdT);
          }
        } else if (i == NLEVSNO) {
          t += fact_i * heatr / (1.0 - (1.0 - frac_sno_eff - frac_h2osfc) * fact_i * dhsdT);
        } else {
          if (i >= NLEVSNO) {
            t += fact_i * heatr;
          } else {
            if (frac_sno_eff > 0.0) t += (fact_i / frac_sno_eff) * heatr;
          }
        }
        if (i < NLEVSNO) {
          if (liq * ice > 0.0) t = TFRZ;
        }
      }
      xmf += HFUS * (wice0 - ice) / dtime;
      if (imelt == 1 && i < NLEVSNO) qflx_snomelt += dmax(0.0, (wice0 - ice)) / dtime;
      if (imelt == 2 && i < NLEVSNO) frz = dmax(0.0, (ice - wice0)) / dtime;
    }
    if (i < NLEVSNO) {
      LV(qflx_snofrz_lyr, i) = frz;
      if (imelt == 2) qflx_snofrz += frz;
    }
    LV(imelt, i) = imelt;
    LV(t_soisno, i) = t;
    if (moved) {  // (a level without phase change keeps the ice and liquid it was read with: nothing to write)
      LV(h2osoi_ice, i) = ice;
      LV(h2osoi_liq, i) = liq;
    }
    if (i == top) t_top_new = t;
    if (i == NLEVSNO) t_soi0 = t;
  };
#pragma unroll 1
  for (int i = 0; i < NLEVTOT; i += 2) {
    pc_level(i, pX);
    pc_level(i + 1, pY);
  }
  S->xmf[c] = xmf;
  S->qflx_snofrz[c] = qflx_snofrz;
  S->qflx_snow_melt[c] = qflx_snow_melt;
  S->qflx_snomelt[c] = qflx_snomelt;
  S->eflx_snomelt[c] = qflx_snomelt * HFUS;
  S->t_h2osfc[c] = t_h2osfc;
  S->h2osfc[c] = h2osfc;
  S->h2osno[c] = h2osno;
  S->int_snow[c] = int_snow;
  S->snow_depth[c] = snow_depth;

  // ---- update_t_grnd (soil_temperature_impl.hh:179-205)
  {
    double t_grnd;
    if (snl > 0) {
      if (frac_h2osfc != 0.0) {
        t_grnd = frac_sno_eff * t_top_new + (1.0 - frac_sno_eff - frac_h2osfc) * t_soi0 + frac_h2osfc * t_h2osfc;
      } else {
        t_grnd = frac_sno_eff * t_top_new + (1.0 - frac_sno_eff) * t_soi0;
      }
    } else {
      if (frac_h2osfc != 0.0) {
        t_grnd = (1.0 - frac_h2osfc) * t_soi0 + frac_h2osfc * t_h2osfc;
      } else {
        t_grnd = t_soi0;
      }
    }
    S->t_grnd[c] = t_grnd;
  }
}

void launch_soil_temperature(const DevState* S, int64_t n, double dt, hipStream_t st)
{
  if (n <= 0) return;
  hipLaunchKernelGGL(k_soil_temperature, dim3((unsigned)((n + ST_WG - 1) / ST_WG)), dim3(ST_WG), 0, st, S, dt);
}

}  // namespace elmk
