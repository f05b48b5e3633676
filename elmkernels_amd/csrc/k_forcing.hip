// k_forcing.hip - the per-column functors kokkos_init_timestep runs ahead of its own kernel:
//
//   get_forcing (driver/kokkos/atm_forcing_kokkos.cc:47-75): eight parallel_for launches in the reference, one per
//   forcing stream - ComputeAtmForcing_TBOT, _PBOT, _QBOT|RH, _FLDS, _FSDS, _PREC, _WIND, _ZBOT
//   (src/physics/atm_physics_impl.hh:27-245).  A column only reads what the earlier functors wrote for the SAME column
//   (tbot -> qbot, lwrad, rain/snow; pbot -> qbot, lwrad), so they are one streaming kernel here, in the wrapper's order.
//   The raw streams are the two records of AtmDataManager::data(ntimes, ncells) that bracket the model time, held as the
//   two-level state fields atm_* (level 0 = record t_idx, level 1 = t_idx + 1): already time-major, i.e. SoA.
//
//   ComputePhenology (src/physics/phenology_physics_impl.hh:22-69, run by update_phenology,
//   driver/kokkos/phenology_kokkos.cc:59-62) over the two bracketing months mlai .. mhbot.
//
// The time logic that picks t_idx and the weights (AtmDataManager::forc_t_idx_check_bounds, forcing_time_weights,
// atm_data_impl.hh:147-199) works on dates on the host and stays with the caller; the readers are file I/O.
// Algorithmic bytes per column: get_forcing 7 x 16 + 8 (coszen) read, 17 x 8 written = 256; phenology 4 x 16 + 20 read,
// 6 x 8 + 4 written = 136.
#include "elmk_dev.h"
#include "elmk_kernels.h"

namespace elmk {

#define LV(f, lev) S->f[(int64_t)(lev) * ld + c]

// atm_physics_impl.hh:205-245
__device__ __forceinline__ double interp_forcing(double wt1, double wt2, double forc1, double forc2) { return forc1 * wt1 + forc2 * wt2; }
__device__ __forceinline__ double tdc(double t) { return dmin(50.0, dmax(-50.0, (t - TFRZ))); }
__device__ __forceinline__ double esatw(double t)
{
  const double a0 = 6.107799961, a1 = 4.436518521e-01, a2 = 1.428945805e-02, a3 = 2.650648471e-04, a4 = 3.031240396e-06,
               a5 = 2.034080948e-08, a6 = 6.136820929e-11;
  return 100.0 * (a0 + t * (a1 + t * (a2 + t * (a3 + t * (a4 + t * (a5 + t * a6))))));
}
__device__ __forceinline__ double esati(double t)
{
  const double b0 = 6.109177956, b1 = 5.034698970e-01, b2 = 1.886013408e-02, b3 = 4.176223716e-04, b4 = 5.824720280e-06,
               b5 = 4.838803174e-08, b6 = 1.838826904e-10;
  return 100.0 * (b0 + t * (b1 + t * (b2 + t * (b3 + t * (b4 + t * (b5 + t * b6))))));
}

struct ForcingWeights {
  double wt1[8], wt2[8];  // TBOT, PBOT, QBOT|RH, FLDS, FSDS, PREC, WIND, ZBOT (the last three and FSDS unused)
  int qbot_is_rh;
};

__global__ __launch_bounds__(256) void k_get_forcing(const DevState* __restrict__ S, const ForcingWeights W)
{
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t ld = S->ld;
  if (c >= S->ncols) return;
  // ProcessTBOT :38-42
  const double tbot = dmin(interp_forcing(W.wt1[0], W.wt2[0], LV(atm_tbot, 0), LV(atm_tbot, 1)), 323.0);
  S->forc_tbot[c] = tbot;
  S->forc_thbot[c] = tbot;
  // ProcessPBOT :55-58
  const double pbot = dmax(interp_forcing(W.wt1[1], W.wt2[1], LV(atm_pbot, 0), LV(atm_pbot, 1)), 4.0e4);
  S->forc_pbot[c] = pbot;
  // ProcessQBOT :73-81
  double qbot = dmax(interp_forcing(W.wt1[2], W.wt2[2], LV(atm_qbot, 0), LV(atm_qbot, 1)), 1.0e-9);
  if (W.qbot_is_rh) {
    const double e = (tbot > TFRZ) ? esatw(tdc(tbot)) : esati(tdc(tbot));
    const double qsat = 0.622 * e / (pbot - 0.378 * e);
    qbot *= qsat / 100.0;
  }
  S->forc_qbot[c] = qbot;
  // ProcessFLDS :97-107
  const double flds = interp_forcing(W.wt1[3], W.wt2[3], LV(atm_flds, 0), LV(atm_flds, 1));
  double lwrad = flds;
  if (flds <= 50.0 || flds >= 600.0) {
    const double e = pbot * qbot / (0.622 + 0.378 * qbot);
    const double ea = 0.70 + 5.95e-5 * 0.01 * e * elmk_exp(1500.0 / tbot);
    lwrad = ea * STEBOL * elmk_pow(tbot, 4.0);
  }
  S->forc_lwrad[c] = lwrad;
  // ProcessFSDS :122-142 (record t_idx only); pow(x, 2.0) is x * x in the reference's optimised builds (elmk_math.h)
  {
    const double swndr = dmax(LV(atm_fsds, 0) * S->coszen[c] * 0.5, 0.0);
    const double swndf = swndr, swvdr = swndr, swvdf = swndr;
    const double ratio_rvrf_vis =
        dmin(0.99, dmax(0.17639 + 0.00380 * swvdr - 9.0039e-06 * elmk_sq(swvdr) + 8.1351e-09 * elmk_pow(swvdr, 3.0), 0.01));
    const double ratio_rvrf_nir =
        dmin(0.99, dmax(0.29548 + 0.00504 * swndr - 1.4957e-05 * elmk_sq(swndr) + 1.4881e-08 * elmk_pow(swndr, 3.0), 0.01));
    LV(forc_solad, 0) = ratio_rvrf_vis * swvdr;
    LV(forc_solad, 1) = ratio_rvrf_nir * swndr;
    LV(forc_solai, 0) = (1.0 - ratio_rvrf_vis) * swvdf;
    LV(forc_solai, 1) = (1.0 - ratio_rvrf_nir) * swndf;
  }
  // ProcessPREC :157-163 (record t_idx only)
  {
    const double frac1 = (tbot - TFRZ) * 0.5;
    const double frac2 = dmin(1.0, dmax(0.0, frac1));
    const double prec = dmax(LV(atm_prec, 0), 0.0);
    S->forc_rain[c] = frac2 * prec;
    S->forc_snow[c] = (1.0 - frac2) * prec;
  }
  // ProcessWIND :177-181
  S->forc_u[c] = interp_forcing(W.wt1[6], W.wt2[6], LV(atm_wind, 0), LV(atm_wind, 1));
  S->forc_v[c] = 0.0;
  // ProcessZBOT :195-203 (hardwired 30 m)
  S->forc_hgt[c] = 30.0;
  S->forc_hgt_u_patch[c] = 30.0;
  S->forc_hgt_t_patch[c] = 30.0;
  S->forc_hgt_q_patch[c] = 30.0;
}

// phenology_physics_impl.hh:22-69
__global__ __launch_bounds__(256) void k_phenology(const DevState* __restrict__ S, double wt1, double wt2)
{
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t ld = S->ld;
  if (c >= S->ncols) return;
  constexpr int noveg = 0, nbrdlf_dcd_brl_shrub = 11;  // elm_constants.h:56,67
  const int vtype = S->vtype[c];
  double tlai = 0.0, tsai = 0.0, htop = 0.0, hbot = 0.0;
  if (vtype != noveg) {
    tlai = wt1 * LV(mlai, 0) + wt2 * LV(mlai, 1);
    tsai = wt1 * LV(msai, 0) + wt2 * LV(msai, 1);
    htop = wt1 * LV(mhtop, 0) + wt2 * LV(mhtop, 1);
    hbot = wt1 * LV(mhbot, 0) + wt2 * LV(mhbot, 1);
  }
  S->tlai[c] = tlai;
  S->tsai[c] = tsai;
  S->htop[c] = htop;
  S->hbot[c] = hbot;
  const double snow_depth = S->snow_depth[c], frac_sno = S->frac_sno[c];
  double fb;
  if (vtype > noveg && vtype <= nbrdlf_dcd_brl_shrub) {
    const double ol = dmin(dmax(snow_depth - hbot, 0.0), htop - hbot);
    fb = 1.0 - ol / dmax(1.e-06, htop - hbot);
  } else {
    fb = 1.0 - dmax(dmin(snow_depth, 0.2), 0.0) / 0.2;  // 0.2 m buries grasses
  }
  double elai = dmax(tlai * (1.0 - frac_sno) + tlai * fb * frac_sno, 0.0);
  double esai = dmax(tsai * (1.0 - frac_sno) + tsai * fb * frac_sno, 0.0);
  if (elai < 0.05) elai = 0.0;
  if (esai < 0.05) esai = 0.0;
  S->elai[c] = elai;
  S->esai[c] = esai;
  S->frac_veg_nosno_alb[c] = ((elai + esai) >= 0.05) ? 1 : 0;
}

void launch_get_forcing(const DevState* S, int64_t n, const double* wt1, const double* wt2, int qbot_is_rh, hipStream_t st)
{
  if (n <= 0) return;
  ForcingWeights W;
  for (int i = 0; i < 8; i++) {
    W.wt1[i] = wt1[i];
    W.wt2[i] = wt2[i];
  }
  W.qbot_is_rh = qbot_is_rh;
  hipLaunchKernelGGL(k_get_forcing, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, S, W);
}

void launch_phenology(const DevState* S, int64_t n, double wt1, double wt2, hipStream_t st)
{
  if (n <= 0) return;
  hipLaunchKernelGGL(k_phenology, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, S, wt1, wt2);
}

}  // namespace elmk
