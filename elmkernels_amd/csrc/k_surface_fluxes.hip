// k_surface_fluxes.hip - the short streaming wrappers around the budgets: the per-column part of kokkos_init_timestep
// at the start of a step, and the two that close the budgets after the temperature solve:
//   kokkos_surface_fluxes        driver/kokkos/surface_fluxes_kokkos.cc:5-107
//     surface_fluxes::initial_flux_calc :73, update_surface_fluxes :147, lwrad_outgoing :247, soil_energy_balance :268
//                                                            (src/physics/surface_fluxes_impl.hh)
//   kokkos_evaluate_conservation driver/kokkos/conserved_quantity_kokkos.cc:8-81
//     conservation_eval::column_water_mass :7 ... net_radiation :109   (conserved_quantity_evaluators_impl.hh)
// Both are streaming kernels, one thread per column.  The reference keeps the eight conservation diagnostics in
// wrapper-local Views and prints column 0; here they are written to a scratch array and reduced on the device to
// (min, max, sum) per diagnostic - the three numbers a multi-GPU run all-reduces (MIN, MAX, SUM), as the reference's
// min_max_sum utility does over MPI (src/utils/min_max_sum.hh:57-66).
// Reference quirks kept: pow(t_h2osfc_bef, 40) (:177), pow(emg * sb * t_grnd0, 3.0) * (4.0 * tinc) (:182),
// (t_h2osfc / dtime) in the soil energy balance (:273).
#include "elmk_dev.h"
#include "elmk_kernels.h"

namespace elmk {

#define LV(f, lev) S->f[(int64_t)(lev) * ld + c]

constexpr int NLEVURB = 5;  // elm_constants.h

// surface_fluxes_impl.hh:8-19
__device__ __forceinline__ double sf_prev_tgrnd(int snl, double frac_sno_eff, double frac_h2osfc, double t_h2osfc_bef,
                                                double tssbef_snotop, double tssbef_soitop)
{
  if (snl > 0) {
    return frac_sno_eff * tssbef_snotop + (1.0 - frac_sno_eff - frac_h2osfc) * tssbef_soitop + frac_h2osfc * t_h2osfc_bef;
  } else {
    return (1.0 - frac_h2osfc) * tssbef_soitop + frac_h2osfc * t_h2osfc_bef;
  }
}

__global__ __launch_bounds__(256) void k_surface_fluxes(const DevState* __restrict__ S, double dtime)
{
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t ld = S->ld;
  if (c >= S->ncols) return;
  const Land L = S->land;
  const bool urbpoi = L.urbpoi != 0;
  const int snl = S->snl[c];
  const int snotop = NLEVSNO - snl;
  const double frac_sno_eff = S->frac_sno_eff[c], frac_h2osfc = S->frac_h2osfc[c], t_h2osfc_bef = S->t_h2osfc_bef[c];
  const double tssbef_snotop = LV(tssbef, snotop), tssbef_soitop = LV(tssbef, NLEVSNO);
  const double t_grnd = S->t_grnd[c], htvp = S->htvp[c], emg = S->emg[c], forc_lwrad = S->forc_lwrad[c];
  const int fvn = S->frac_veg_nosno[c];
  double eflx_sh_grnd = S->eflx_sh_grnd[c], qflx_evap_soi = S->qflx_evap_soi[c], qflx_ev_snow = S->qflx_ev_snow[c];
  double qflx_ev_soil = S->qflx_ev_soil[c], qflx_ev_h2osfc = S->qflx_ev_h2osfc[c];

  // ---- initial_flux_calc (:73-95)
  const double t_grnd0 = sf_prev_tgrnd(snl, frac_sno_eff, frac_h2osfc, t_h2osfc_bef, tssbef_snotop, tssbef_soitop);
  const double tinc = t_grnd - t_grnd0;
  {
    const double cgrnds = S->cgrnds[c], cgrndl = S->cgrndl[c];
    eflx_sh_grnd += tinc * cgrnds;
    qflx_evap_soi += tinc * cgrndl;
    if (!urbpoi) {
      qflx_ev_snow += tinc * cgrndl;
      qflx_ev_soil += tinc * cgrndl;
      qflx_ev_h2osfc += tinc * cgrndl;
    } else {
      qflx_ev_snow = qflx_evap_soi;
      qflx_ev_soil = 0.0;
      qflx_ev_h2osfc = 0.0;
    }
  }

  // ---- update_surface_fluxes (:147-238)
  const double h2osoi_ice_snotop = LV(h2osoi_ice, snotop);
  const double h2osoi_liq_snotop = LV(h2osoi_liq, NLEVSNO);  // the wrapper passes h2osoi_liq(idx, soitop) (:58)
  {
    double egsmax = (h2osoi_ice_snotop + h2osoi_liq_snotop) / dtime;  // evap_ratio (:31-45)
    if (egsmax < 0.0) egsmax = 0.0;
    const double egirat = (qflx_evap_soi > egsmax) ? egsmax / qflx_evap_soi : 1.0;
    if (egirat < 1.0) {
      const double save_qflx_evap_soi = qflx_evap_soi;
      qflx_evap_soi *= egirat;
      eflx_sh_grnd += (save_qflx_evap_soi - qflx_evap_soi) * htvp;
      qflx_ev_snow *= egirat;
      qflx_ev_soil *= egirat;
      qflx_ev_h2osfc *= egirat;
    }
  }
  double eflx_soil_grnd = S->eflx_soil_grnd[c];
  if (!urbpoi) {
    const double lw_grnd = (frac_sno_eff * elmk_pow(tssbef_snotop, 4.0) + (1.0 - frac_sno_eff - frac_h2osfc) * elmk_pow(tssbef_soitop, 4.0) +
                            frac_h2osfc * elmk_pow(t_h2osfc_bef, 40));
    eflx_soil_grnd = ((1.0 - frac_sno_eff) * S->sabg_soil[c] + frac_sno_eff * S->sabg_snow[c]) + S->dlrad[c] +
                     (1.0 - (double)fvn) * emg * forc_lwrad - emg * STEBOL * lw_grnd -
                     elmk_pow(emg * STEBOL * t_grnd0, 3.0) * (4.0 * tinc) - (eflx_sh_grnd + qflx_evap_soi * htvp);
  }
  const double eflx_sh_veg = S->eflx_sh_veg[c], qflx_evap_veg = S->qflx_evap_veg[c];
  S->eflx_sh_tot[c] = eflx_sh_veg + eflx_sh_grnd;
  S->qflx_evap_tot[c] = qflx_evap_veg + qflx_evap_soi;
  S->eflx_lh_tot[c] = HVAP * qflx_evap_veg + htvp * qflx_evap_soi;
  double qflx_evap_grnd = 0.0, qflx_sub_snow = 0.0, qflx_dew_snow = 0.0, qflx_dew_grnd = 0.0;
  if (qflx_ev_snow >= 0.0) {
    if ((h2osoi_liq_snotop + h2osoi_ice_snotop) > 0.0) {
      qflx_evap_grnd = dmax(qflx_ev_snow * (h2osoi_liq_snotop / (h2osoi_liq_snotop + h2osoi_ice_snotop)), 0.0);
    } else {
      qflx_evap_grnd = 0.0;
    }
    qflx_sub_snow = qflx_ev_snow - qflx_evap_grnd;
  } else {
    if (t_grnd < TFRZ) {
      qflx_dew_snow = fabs(qflx_ev_snow);
    } else {
      qflx_dew_grnd = fabs(qflx_ev_snow);
    }
  }
  if (snl > 0 && S->do_capsnow[c]) {
    S->qflx_snwcp_liq[c] = S->qflx_snwcp_liq[c] + frac_sno_eff * qflx_dew_grnd;
    S->qflx_snwcp_ice[c] = S->qflx_snwcp_ice[c] + frac_sno_eff * qflx_dew_snow;
  }
  S->eflx_sh_grnd[c] = eflx_sh_grnd;
  S->qflx_evap_soi[c] = qflx_evap_soi;
  S->qflx_ev_snow[c] = qflx_ev_snow;
  S->qflx_ev_soil[c] = qflx_ev_soil;
  S->qflx_ev_h2osfc[c] = qflx_ev_h2osfc;
  S->eflx_soil_grnd[c] = eflx_soil_grnd;
  S->qflx_evap_grnd[c] = qflx_evap_grnd;
  S->qflx_sub_snow[c] = qflx_sub_snow;
  S->qflx_dew_snow[c] = qflx_dew_snow;
  S->qflx_dew_grnd[c] = qflx_dew_grnd;

  // ---- lwrad_outgoing (:247-265)
  if (!urbpoi) {
    const double lw_grnd = (frac_sno_eff * elmk_pow(tssbef_snotop, 4.0) + (1.0 - frac_sno_eff - frac_h2osfc) * elmk_pow(tssbef_soitop, 4.0) +
                            frac_h2osfc * elmk_pow(t_h2osfc_bef, 4.0));
    const double out = S->ulrad[c] + (1 - fvn) * (1.0 - emg) * forc_lwrad + (1 - fvn) * emg * STEBOL * lw_grnd +
                       4.0 * emg * STEBOL * elmk_pow(t_grnd0, 3.0) * tinc;
    S->eflx_lwrad_out[c] = out;
    S->eflx_lwrad_net[c] = out - forc_lwrad;
  }

  // ---- soil_energy_balance (:268-294)
  {
    const double t_h2osfc = S->t_h2osfc[c];
    double errsoi = eflx_soil_grnd - S->xmf[c] - S->xmf_h2osfc[c] - frac_h2osfc * (t_h2osfc - t_h2osfc_bef) * (t_h2osfc / dtime);
    errsoi += S->eflx_h2osfc_snow[c];
    const bool wall = (L.ctype == icol_sunwall || L.ctype == icol_shadewall || L.ctype == icol_roof);
    if (wall) errsoi += 0.0;  // eflx_building_heat
#pragma unroll 1
    for (int j = 0; j < NLEVTOT; ++j) {
      if (!wall || (j < NLEVURB)) {
        if (j >= NLEVSNO - snl && j < NLEVSNO) errsoi -= frac_sno_eff * (LV(t_soisno, j) - LV(tssbef, j)) / LV(fact, j);
        if (j >= NLEVSNO) errsoi -= (LV(t_soisno, j) - LV(tssbef, j)) / LV(fact, j);
      }
    }
    S->soil_e_balance[c] = errsoi;
  }
}

// ---- conserved_quantity_kokkos.cc:8-81: the eight diagnostics per column -> scratch diag[k][column]
constexpr int NDIAG = 8;
__global__ __launch_bounds__(256) void k_conservation(const DevState* __restrict__ S, double dtime)
{
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t ld = S->ld;
  if (c >= S->ncols) return;
  const double hydrology_source_sink = 0.0;  // hardwired (:22)
  const gptr<double> d = S->cons_diag + c;
  const double h2osno = S->h2osno[c];
  double water = S->h2ocan[c] + h2osno + S->h2osfc[c];  // column_water_mass (:7-15)
#pragma unroll 1
  for (int i = 0; i < NLEVTOT; ++i) water += LV(h2osoi_ice, i) + LV(h2osoi_liq, i);
  const double begwb = S->dtbegin_column_h2o[c];
  const double qflx_snwcp_ice = S->qflx_snwcp_ice[c];
  d[0] = water;
  d[(int64_t)3 * ld] = (water - begwb) / dtime;  // dh2o_dt
  d[ld] = (water - begwb) - (S->forc_rain[c] + S->forc_snow[c] - hydrology_source_sink - S->qflx_evap_tot[c] - qflx_snwcp_ice) * dtime;
  {  // snow_water_balance_error (:37-70)
    double err = 0.0;
    if (S->snl[c] > 0) {
      const double qflx_dew_snow = S->qflx_dew_snow[c], qflx_dew_grnd = S->qflx_dew_grnd[c], qflx_sub_snow = S->qflx_sub_snow[c];
      const double qflx_evap_grnd = S->qflx_evap_grnd[c], qflx_snow_melt = S->qflx_snow_melt[c];
      const double qflx_sl_top_soil = S->qflx_sl_top_soil[c], frac_sno_eff = S->frac_sno_eff[c];
      const double qflx_rain_grnd = S->qflx_rain_grnd[c], qflx_snow_grnd = S->qflx_snow_grnd[c];
      const double qflx_h2osfc_ice = S->qflx_h2osfc_ice[c];
      double snow_sources, snow_sinks;
      if (S->do_capsnow[c]) {
        snow_sources = frac_sno_eff * (qflx_dew_snow + qflx_dew_grnd) + qflx_h2osfc_ice + qflx_snow_grnd + qflx_rain_grnd;
        snow_sinks = frac_sno_eff * (qflx_sub_snow + qflx_evap_grnd) + qflx_snwcp_ice + S->qflx_snwcp_liq[c] + qflx_snow_melt +
                     qflx_sl_top_soil;
      } else {
        const double qflx_snow_h2osfc = 0.0;
        snow_sources = (qflx_snow_grnd - qflx_snow_h2osfc) + frac_sno_eff * (qflx_rain_grnd + qflx_dew_snow + qflx_dew_grnd) +
                       qflx_h2osfc_ice;
        snow_sinks = frac_sno_eff * (qflx_sub_snow + qflx_evap_grnd) + qflx_snow_melt + qflx_sl_top_soil;
      }
      err = (h2osno - S->h2osno_old[c]) - (snow_sources - snow_sinks) * dtime;
    }
    d[(int64_t)2 * ld] = err;
  }
  const double fsa = S->fsa[c], forc_lwrad = S->forc_lwrad[c];
  const double eflx_lwrad_out = S->eflx_lwrad_out[c], eflx_lwrad_net = S->eflx_lwrad_net[c];
  d[(int64_t)4 * ld] = fsa + S->fsr[c] - (LV(forc_solad, 0) + LV(forc_solad, 1) + LV(forc_solai, 0) + LV(forc_solai, 1));
  d[(int64_t)5 * ld] = eflx_lwrad_out - eflx_lwrad_net - forc_lwrad;
  d[(int64_t)6 * ld] = S->sabv[c] + S->sabg_chk[c] + forc_lwrad - eflx_lwrad_out - S->eflx_sh_tot[c] - S->eflx_lh_tot[c] -
                       S->eflx_soil_grnd[c];
  d[(int64_t)7 * ld] = fsa - eflx_lwrad_net;
}

// (min, max, sum) of each diagnostic: stage 1 one partial triple per workgroup, stage 2 one workgroup per diagnostic
__global__ __launch_bounds__(256) void k_cons_reduce1(const double* __restrict__ diag, int64_t ld, int64_t n, double* __restrict__ part)
{
  __shared__ double s_min[256], s_max[256], s_sum[256];
  const int k = blockIdx.y;
  const double* __restrict__ x = diag + (int64_t)k * ld;
  double mn = INFINITY, mx = -INFINITY, sm = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double v = x[i];
    mn = fmin(mn, v);
    mx = fmax(mx, v);
    sm += v;
  }
  s_min[threadIdx.x] = mn;
  s_max[threadIdx.x] = mx;
  s_sum[threadIdx.x] = sm;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      s_min[threadIdx.x] = fmin(s_min[threadIdx.x], s_min[threadIdx.x + s]);
      s_max[threadIdx.x] = fmax(s_max[threadIdx.x], s_max[threadIdx.x + s]);
      s_sum[threadIdx.x] += s_sum[threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double* o = part + ((int64_t)k * gridDim.x + blockIdx.x) * 3;
    o[0] = s_min[0];
    o[1] = s_max[0];
    o[2] = s_sum[0];
  }
}

__global__ __launch_bounds__(256) void k_cons_reduce2(const double* __restrict__ part, int nblk, double* __restrict__ out)
{
  __shared__ double s_min[256], s_max[256], s_sum[256];
  const int k = blockIdx.x;
  double mn = INFINITY, mx = -INFINITY, sm = 0.0;
  for (int i = threadIdx.x; i < nblk; i += blockDim.x) {
    const double* p = part + ((int64_t)k * nblk + i) * 3;
    mn = fmin(mn, p[0]);
    mx = fmax(mx, p[1]);
    sm += p[2];
  }
  s_min[threadIdx.x] = mn;
  s_max[threadIdx.x] = mx;
  s_sum[threadIdx.x] = sm;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      s_min[threadIdx.x] = fmin(s_min[threadIdx.x], s_min[threadIdx.x + s]);
      s_max[threadIdx.x] = fmax(s_max[threadIdx.x], s_max[threadIdx.x + s]);
      s_sum[threadIdx.x] += s_sum[threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[k * 3 + 0] = s_min[0];
    out[k * 3 + 1] = s_max[0];
    out[k * 3 + 2] = s_sum[0];
  }
}

// ---- the per-column kernel of kokkos_init_timestep (init_timestep_kokkos.cc:55-75): h2osno_old, the column water
//      mass the conservation check starts from, ELM::init_timestep (src/physics/init_timestep_impl.hh:7-42)
__global__ __launch_bounds__(256) void k_init_timestep(const DevState* __restrict__ S)
{
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t ld = S->ld;
  if (c >= S->ncols) return;
  const double h2osno = S->h2osno[c];
  S->h2osno_old[c] = h2osno;
  double water = S->h2ocan[c] + h2osno + S->h2osfc[c];
#pragma unroll 1
  for (int i = 0; i < NLEVTOT; ++i) water += LV(h2osoi_ice, i) + LV(h2osoi_liq, i);
  S->dtbegin_column_h2o[c] = water;
  S->do_capsnow[c] = (h2osno > 1000.0) ? 1 : 0;  // H2OSNO_MAX (elm_constants.h)
  S->frac_veg_nosno[c] = S->veg_active[c] ? S->frac_veg_nosno_alb[c] : 0;
  if (!S->land.lakpoi) {
    const int snl = S->snl[c];
#pragma unroll
    for (int i = 0; i < NLEVSNO; i++) {
      if (i >= NLEVSNO - snl) {
        const double ice = LV(h2osoi_ice, i);
        LV(frac_iceold, i) = ice / (LV(h2osoi_liq, i) + ice);
      }
    }
  }
}

void launch_init_timestep(const DevState* S, int64_t n, hipStream_t st)
{
  if (n <= 0) return;
  hipLaunchKernelGGL(k_init_timestep, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, S);
}

void launch_surface_fluxes(const DevState* S, int64_t n, double dt, hipStream_t st)
{
  if (n <= 0) return;
  hipLaunchKernelGGL(k_surface_fluxes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, S, dt);
}

// diag: 8 x ld doubles; part: 8 x ELMK_CONS_NPART x 3 doubles; out: 8 x 3 doubles (all device memory)
void launch_conservation(const DevState* S, int64_t n, int64_t ld, double dt, const double* diag, double* part, double* out,
                         hipStream_t st)
{
  if (n <= 0) return;
  hipLaunchKernelGGL(k_conservation, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, S, dt);
  hipLaunchKernelGGL(k_cons_reduce1, dim3(ELMK_CONS_NPART, NDIAG), dim3(256), 0, st, diag, ld, n, part);
  hipLaunchKernelGGL(k_cons_reduce2, dim3(NDIAG), dim3(256), 0, st, part, ELMK_CONS_NPART, out);
}

}  // namespace elmk
