// k_water_energy.hip - the five streaming kernels of the timestep:
//   k_frac_wet, k_canopy_hydrology, k_surface_radiation, k_canopy_temperature, k_bareground_fluxes
//
// One thread per column, 256-thread workgroups, SoA [lev][column] state: every load/store below is a
// coalesced 512-byte wave request.  All five are HBM-bound (DESIGN.md gives the algorithmic bytes per
// column); level arrays are touched only at the levels the physics reads, temporaries the reference
// materialises as ViewD1(ncols) per call live in registers.
//
// Arithmetic follows the reference expression by expression (operand order kept, no FMA contraction)
// so fp64 results stay within rounding of the CPU path; file:line of each restated routine is given.
#include "elmk_dev.h"
#include "elmk_kernels.h"
#include "elmk_stream.h"

namespace elmk {

#define COL_GUARD()                                                   \
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   \
  if (c >= S->ncols) return;                                          \
  const int64_t ld = S->ld;                                           \
  (void)ld

// =====================================================================================================
// kokkos_frac_wet (driver/kokkos/canopy_hydrology_kokkos.cc:98-112)
//   canopy_hydrology::fraction_wet  src/physics/canopy_hydrology_impl.hh:123-143
// =====================================================================================================
__global__ __launch_bounds__(256) void k_frac_wet(const DevState* __restrict__ S)
{
  COL_GUARD();
  frac_wet_col(S, c, S->land);
}

// =====================================================================================================
// kokkos_canopy_hydrology (driver/kokkos/canopy_hydrology_kokkos.cc:7-95), kokkos_surface_radiation
// (surface_radiation_kokkos.cc:7-97), kokkos_canopy_temperature (canopy_temperature_kokkos.cc:6-131): the per-column
// bodies are in elmk_stream.h (shared with the fused streaming stage, k_fz_stream in k_canopy_fluxes.hip)
// =====================================================================================================
__global__ __launch_bounds__(256) void k_canopy_hydrology(const DevState* __restrict__ S, double dtime)
{
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t ld = S->ld;
  ColFwd w;
  canopy_hydrology_col<false>(S, c, ld, S->land, dtime, w, c < S->ncols);  // (every thread: the pond solves are pooled per workgroup)
}

__global__ __launch_bounds__(256) void k_surface_radiation(const DevState* __restrict__ S)
{
  COL_GUARD();
  ColFwd w;
  const AlbFwd no_a{};  // (the wrapper's own kernel reads the albedo stage's outputs from the state)
  const double no_flx[6][4] = {};
  surface_radiation_col<false>(S, c, ld, S->land, w, no_a, no_flx);
}

__global__ __launch_bounds__(256) void k_canopy_temperature(const DevState* __restrict__ S)
{
  COL_GUARD();
  ColFwd w;
  canopy_temperature_col<false>(S, c, ld, S->land, w);
}

// =====================================================================================================
// kokkos_bareground_fluxes (driver/kokkos/bareground_fluxes_kokkos.cc:7-123)
//   initialize_flux :7, stability_iteration :30, compute_flux :82 of bareground_fluxes_impl.hh
//   active only where frac_veg_nosno == 0; everywhere else just cgrnd/cgrnds/cgrndl = 0
// =====================================================================================================
// stage 1 (every column, streaming): compute_flux's unconditional cgrnd reset (:97-102) and the queue of bare columns
// (1024-thread workgroups: one global atomic per workgroup on the bare-ground list's counter, and same-address atomics retire
//  one after the other at ~14 ns each - with 256-thread workgroups they were most of this kernel's time on a mixed tile,
//  profiles/r03_classify_atomics_ab.txt)
#ifndef BG_MAIN_THREADS
#define BG_MAIN_THREADS 1024
#endif
__global__ __launch_bounds__(BG_MAIN_THREADS) void k_bg_main(const DevState* __restrict__ S)
{
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const Land L = S->land;
  if (L.lakpoi) return;
  const bool inside = c < S->ncols;
  if (inside) {
    S->cgrnd[c] = 0.0;
    S->cgrnds[c] = 0.0;
    S->cgrndl[c] = 0.0;
  }
  const bool bare = inside && !L.urbpoi && S->frac_veg_nosno[c] == 0;
  block_classify_append<1>(S->lists, S->ld, S->counters, LIST_BG, bare ? 0 : -1, (int32_t)c);
}


// stage 2 (bare columns only, from the queue): the Monin-Obukhov iteration and the fluxes
__global__ __launch_bounds__(256) void k_bg_flux(const DevState* __restrict__ S, const int given)
{
  const int64_t ld = S->ld;
  const uint32_t count = ELMK_LIST_COUNT(S, LIST_BG);
  const gptr<const int32_t> list = S->lists + (int64_t)LIST_BG * ld;
  for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < count; q += gridDim.x * blockDim.x) {
  const int64_t c = list[q];

  const double forc_pbot = S->forc_pbot[c], forc_q = S->forc_qbot[c], forc_th = S->forc_thbot[c];
  double forc_rho = derive_forc_rho(forc_pbot, forc_q, S->forc_tbot[c]);
  if (given & 1) forc_rho = S->cf_given[c];  // elmk_bareground_fluxes_given
  const double thm = S->thm[c], thv = S->thv[c], t_grnd = S->t_grnd[c], qg = S->qg[c], z0mg = S->z0mg[c];
  const double hgt_u = S->forc_hgt_u_patch[c], hgt_t = S->forc_hgt_t_patch[c], hgt_q = S->forc_hgt_q_patch[c];
  const double forc_u = S->forc_u[c], forc_v = S->forc_v[c];

  // ---- initialize_flux
  const double ur = dmax(1.0, sqrt(forc_u * forc_u + forc_v * forc_v));
  const double dth = thm - t_grnd;
  const double dqh = forc_q - qg;
  const double zldis = hgt_u;
  const double dthv = dth * (1.0 + 0.61 * forc_q) + 0.61 * forc_th * dqh;
  const double displa = 0.0;
  S->dlrad[c] = 0.0;
  S->ulrad[c] = 0.0;
  double um, obu;
  monin_obukhov_length(ur, thv, dthv, zldis, z0mg, um, obu);

  // ---- stability_iteration: 3 fixed iterations
  const FvConst FV = fv_const();
  double z0hg = S->z0hg[c], z0qg = S->z0qg[c];
  double ustar = 0.0, temp1 = 0.0, temp2 = 0.0, temp12m = 0.0, temp22m = 0.0;
#pragma unroll 1
  for (int i = 0; i < 3; i++) {
    friction_profiles<false>(hgt_u, hgt_t, hgt_q, displa, um, obu, z0mg, z0hg, z0qg, FV, ustar, temp1, temp2, temp12m, temp22m);
    const double tstar = temp1 * dth;
    const double qstar = temp2 * dqh;
    const double thvstar = tstar * (1.0 + 0.61 * forc_q) + 0.61 * forc_th * qstar;
    z0hg = z0mg / elmk_exp(0.13 * elmk_pow((ustar * z0mg / 1.5e-5), 0.45));
    z0qg = z0hg;
    double zeta = zldis * VKC * GRAV * thvstar / (elmk_sq(ustar) * thv);
    if (zeta >= 0.0) {
      zeta = dmin(2.0, dmax(zeta, 0.01));
      um = dmax(ur, 0.1);
    } else {
      zeta = dmax(-100.0, dmin(zeta, -0.01));
      const double wc = 1.0 * elmk_pow((-GRAV * ustar * thvstar * 1000.0 / thv), 0.333);
      um = sqrt(ur * ur + wc * wc);
    }
    obu = zldis / zeta;
  }
  S->z0hg[c] = z0hg;
  S->z0qg[c] = z0qg;

  // ---- compute_flux
  const int snl = S->snl[c];
  const double t_soi0 = LV(t_soisno, NLEVSNO);
  const double t_top = (snl > 0) ? LV(t_soisno, NLEVSNO - snl) : t_soi0;
  const double t_h2osfc = S->t_h2osfc[c];
  const double rah = 1.0 / (temp1 * ustar);
  const double raw = 1.0 / (temp2 * ustar);
  const double raih = forc_rho * CPAIR / rah;
  double raiw;
  if (dqh > 0.0) {
    raiw = forc_rho / raw;
  } else {
    raiw = S->soilbeta[c] * forc_rho / raw;
  }
  const double cgrnds = raih;
  const double cgrndl = raiw * S->dqgdT[c];
  // (given & 4: the fused step.  There canopy_fluxes' compute_flux follows in the same call and resets cgrnd* on every column
  //  (canopy_fluxes_impl.hh:474-479) - k_fz_stream has already stored those zeros - and this kernel may run beside
  //  k_cf_finish on a side stream: it must not store to a field that kernel stores to.)
  if (!(given & 4)) {
    S->cgrnds[c] = cgrnds;
    S->cgrndl[c] = cgrndl;
    S->cgrnd[c] = cgrnds + S->htvp[c] * cgrndl;
  }
  const double eflx_sh_grnd = -raih * dth;
  S->eflx_sh_grnd[c] = eflx_sh_grnd;
  S->eflx_sh_tot[c] = eflx_sh_grnd;
  S->eflx_sh_snow[c] = -raih * (thm - t_top);
  S->eflx_sh_soil[c] = -raih * (thm - t_soi0);
  S->eflx_sh_h2osfc[c] = -raih * (thm - t_h2osfc);
  const double qflx_evap_soi = -raiw * dqh;
  S->qflx_evap_soi[c] = qflx_evap_soi;
  S->qflx_evap_tot[c] = qflx_evap_soi;
  S->qflx_ev_snow[c] = -raiw * (forc_q - S->qg_snow[c]);
  S->qflx_ev_soil[c] = -raiw * (forc_q - S->qg_soil[c]);
  S->qflx_ev_h2osfc[c] = -raiw * (forc_q - S->qg_h2osfc[c]);
  const double t_ref2m = thm + temp1 * dth * (1.0 / temp12m - 1.0 / temp1);
  const double q_ref2m = forc_q + temp2 * dqh * (1.0 / temp22m - 1.0 / temp2);
  double e_ref2m, de2mdT, qsat_ref2m, dqsat2mdT;
  qsat(t_ref2m, forc_pbot, e_ref2m, de2mdT, qsat_ref2m, dqsat2mdT);
  S->t_ref2m[c] = t_ref2m;
  S->q_ref2m[c] = q_ref2m;
  S->rh_ref2m[c] = dmin(100.0, (q_ref2m / qsat_ref2m * 100.0));
  }
  // The list is left empty for the next call by the last workgroup that had entries to work on (every launch costs ~4.5 us:
  // a reset kernel in front of k_bg_main was a quarter of this wrapper on a vegetated region).  The workgroups that find no
  // entry - all of them when there is no bare column - touch nothing: the count they read is already what a reset would
  // write.  The list's unused queue-head word counts the workgroups that are done.
  const uint32_t nwork = (count + 255u) / 256u;
  const uint32_t participants = nwork < gridDim.x ? nwork : gridDim.x;
  // (no fence: the only thing the reset is ordered against is the workgroups' read of the count at their start, which the
  //  use of its value has long completed; a release fence here writes back the XCD's L2 once per workgroup - +75 us measured)
  __syncthreads();  // every wave of this workgroup has read the count before thread 0 may let the last workgroup reset it
  if (blockIdx.x < participants && threadIdx.x == 0) {
    if (atomicAdd(ELMK_GENERIC(&ELMK_LIST_HEAD(S, LIST_BG)), 1u) == participants - 1u) {
      ELMK_LIST_COUNT(S, LIST_BG) = 0u;
      ELMK_LIST_HEAD(S, LIST_BG) = 0u;
    }
  }
}

// ---- host launchers ---------------------------------------------------------------------------------
static inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + 255) / 256)); }

void launch_frac_wet(const DevState* S, int64_t n, hipStream_t st)
{
  if (n > 0) hipLaunchKernelGGL(k_frac_wet, grid_for(n), dim3(256), 0, st, S);
}
void launch_canopy_hydrology(const DevState* S, int64_t n, double dt, hipStream_t st)
{
  if (n > 0) hipLaunchKernelGGL(k_canopy_hydrology, grid_for(n), dim3(256), 0, st, S, dt);
}
void launch_surface_radiation(const DevState* S, int64_t n, hipStream_t st)
{
  if (n > 0) hipLaunchKernelGGL(k_surface_radiation, grid_for(n), dim3(256), 0, st, S);
}
void launch_canopy_temperature(const DevState* S, int64_t n, hipStream_t st)
{
  if (n > 0) hipLaunchKernelGGL(k_canopy_temperature, grid_for(n), dim3(256), 0, st, S);
}
// the list stage alone (fused step: k_fz_prep has reset the list and k_fz_stream has filled it)
void launch_bareground_list(const DevState* S, int64_t n, hipStream_t st)
{
  if (n <= 0) return;
  const unsigned full = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(k_bg_flux, dim3(full < 2048u ? full : 2048u), dim3(256), 0, st, S, 4);
}

void launch_bareground_fluxes(const DevState* S, int64_t n, hipStream_t st, int given)
{
  if (n <= 0) return;
  const unsigned full = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(k_bg_main, dim3((unsigned)((n + BG_MAIN_THREADS - 1) / BG_MAIN_THREADS)), dim3(BG_MAIN_THREADS), 0, st, S);
  hipLaunchKernelGGL(k_bg_flux, dim3(full < 2048u ? full : 2048u), dim3(256), 0, st, S, given);
}

}  // namespace elmk
