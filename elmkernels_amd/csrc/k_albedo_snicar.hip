// k_albedo_snicar.hip - kokkos_albedo_snicar (driver/kokkos/albedo_kokkos.cc:10-376)
//
//   surface_albedo::init_timestep :90, soil_albedo :690, ground_albedo :155, flux_absorption_factor :171,
//   canopy_layer_lai :215, two_stream_solver :323           (src/physics/surface_albedo_impl.hh)
//   snow_snicar::init_timestep :9, snow_aerosol_mie_params :107, snow_radiative_transfer_solver :313,
//   snow_albedo_radiation_factor :673, run twice (direct, diffuse)   (src/physics/snow_snicar_impl.hh)
//
// Three stages.
//   k_alb_classify (one thread per column, coalesced): canopy_layer_lai, soil albedo of the sunlit columns, and the
//     sunlit snow-covered columns queued by their number of snow layers NL.
//   k_alb_snicar<NL> (queue-driven): the ten independent (pass, band) solves of a column - direct / diffuse x five
//     spectral bands, each the Mie/aerosol mixing and the Delta-Eddington adding-doubling solve of NL layers with an
//     8-point Gauss quadrature - run on ten lanes, six columns per wave; layer loops are compile-time (template NL),
//     so every per-layer array is statically indexed and lives in registers, and a wave never carries lanes with
//     different layer counts.  The band results are combined across the lanes in the reference's summation order
//     (snow_snicar_impl.hh:724-741) and the two SnowOut of the column go to a scratch array.  One lane holds one
//     band's state (165-280 VGPRs, two waves per SIMD) instead of a whole column's (250-450, one wave per SIMD).
//     The five NL queues are independent launches and run beside each other on side streams.
//   k_alb_final (one thread per column, coalesced): the init_timestep defaults of the night columns; for sunlit
//     columns ground_albedo, flux_absorption_factor and the canopy two-stream solution.  Every output is written once.
// The ~210 doubles of per-column scratch that the wrapper allocates as zero-filled Views on every call
// (albedo_kokkos.cc:19-38) shrink to 28 (the SNICAR products).  Lookup tables (Mie [3][5][1471] x2, BC, aerosol) sit
// in one 355 KB device buffer that stays L2-resident; the gather index is round(snw_rds)-30.
#define ELMK_MATH_LDS 1  // exp / log / pow tables of elmk_math.h in LDS: every kernel below that evaluates them calls elmk_math_lds_init first
#include "elmk_dev.h"
#include "elmk_kernels.h"
#include "elmk_albedo_col.h"
#include "elmk_snicar.h"

namespace elmk {


__device__ __forceinline__ void alb_finish(const DevState* __restrict__ S, const int64_t c, const int64_t ld, const Land& L,
                                           const bool day, const double coszen, const double elai, const double esai,
                                           const double frac_sno, const double (&albsod)[2], const double (&albsoi)[2],
                                           const SnowOut& sd, const SnowOut& si, double vcmaxcintsun, double vcmaxcintsha);

// ground_albedo (:155-167), flux_absorption_factor (:171-211, subgridflag == 1) and two_stream_solver (:323-687,
// nlevcan == 1) for one sunlit column, given soil albedos and the SNICAR products; for a column without sun (day ==
// false) the values surface_albedo::init_timestep leaves (:90-151) and snow_albedo_radiation_factor's "no sun" branch
// (snow_snicar_impl.hh:758-765).  Every output is STORED BY ALL LANES TOGETHER: a wave that holds sunlit and dark
// columns would otherwise write every 128-byte line twice, half of it each time (measured: 904 instead of 490 bytes
// per column written by this kernel on the fixture-tiled state).
__device__ __forceinline__ void alb_finish(const DevState* __restrict__ S, const int64_t c, const int64_t ld, const Land& L,
                                           const bool day, const double coszen, const double elai, const double esai,
                                           const double frac_sno, const double (&albsod)[2], const double (&albsoi)[2],
                                           const SnowOut& sd, const SnowOut& si, double vcmaxcintsun, double vcmaxcintsha)
{
  // ---- ground_albedo (:155-167) and flux_absorption_factor (:171-211, subgridflag == 1)
  double albgrd[2] = {0.0, 0.0}, albgri[2] = {0.0, 0.0};
#pragma unroll
  for (int ib = 0; ib < 2; ib++) {
    double o_sod = 0.0, o_soi = 0.0, o_snd = 0.0, o_sni = 0.0;
    if (day) {
      albgrd[ib] = albsod[ib] * (1.0 - frac_sno) + sd.alb[ib] * frac_sno;
      albgri[ib] = albsoi[ib] * (1.0 - frac_sno) + si.alb[ib] * frac_sno;
      o_sod = albsod[ib];
      o_soi = albsoi[ib];
      o_snd = sd.alb[ib];
      o_sni = si.alb[ib];
    }
    LV(albsod, ib) = o_sod;
    LV(albsoi, ib) = o_soi;
    LV(albsnd, ib) = o_snd;
    LV(albsni, ib) = o_sni;
    LV(albgrd, ib) = albgrd[ib];
    LV(albgri, ib) = albgri[ib];
  }
#pragma unroll
  for (int i = 0; i < 6; i++) {
    double dv = 0.0, dn = 0.0, iv = 0.0, in = 0.0;
    if (day) {
      if (L.ltype == istdlak) {
        dv = sd.fabs_[i][0] * frac_sno + ((1.0 - frac_sno) * (1.0 - albsod[0]) * (sd.fabs_[i][0] / (1.0 - sd.alb[0])));
        iv = si.fabs_[i][0] * frac_sno + ((1.0 - frac_sno) * (1.0 - albsoi[0]) * (si.fabs_[i][0] / (1.0 - si.alb[0])));
        dn = sd.fabs_[i][1] * frac_sno + ((1.0 - frac_sno) * (1.0 - albsod[1]) * (sd.fabs_[i][1] / (1.0 - sd.alb[1])));
        in = si.fabs_[i][1] * frac_sno + ((1.0 - frac_sno) * (1.0 - albsoi[1]) * (si.fabs_[i][1] / (1.0 - si.alb[1])));
      } else {
        dv = sd.fabs_[i][0] * (1.0 - sd.alb[0]);
        iv = si.fabs_[i][0] * (1.0 - si.alb[0]);
        dn = sd.fabs_[i][1] * (1.0 - sd.alb[1]);
        in = si.fabs_[i][1] * (1.0 - si.alb[1]);
      }
    }
    LV(flx_absdv, i) = dv;
    LV(flx_absdn, i) = dn;
    LV(flx_absiv, i) = iv;
    LV(flx_absin, i) = in;
  }

  // ---- two_stream_solver (:323-687), nlevcan == 1
  double albd[2], albi[2], ftdd[2], ftid[2], ftii[2], fabd[2], fabi[2], fabi_sun[2], fabi_sha[2];
  double fsun_z = 0.0, fabd_sun_z = 0.0, fabd_sha_z = 0.0, fabi_sun_z = 0.0, fabi_sha_z = 0.0;
  const bool soilcrop = (L.ltype == istsoil || L.ltype == istcrop);
  if (!day) {  // init_timestep's values (:117-135)
#pragma unroll
    for (int ib = 0; ib < 2; ++ib) {
      fabd[ib] = 0.0;
      fabi[ib] = 0.0;
      fabi_sun[ib] = 0.0;
      fabi_sha[ib] = 0.0;
      ftdd[ib] = 0.0;
      ftid[ib] = 0.0;
      ftii[ib] = 0.0;
      albd[ib] = 1.0;
      albi[ib] = 1.0;
    }
  } else if (soilcrop && (elai + esai) > 0.0) {  // vegsol
    const double* __restrict__ A = S->pft_alb[S->vtype[c]];  // rhol[2] rhos[2] taul[2] taus[2] xl
    const double t_veg = S->t_veg[c], fwet = S->fwet[c];
    const double omegas[2] = {0.8, 0.4};
    const double betads = 0.5, betais = 0.5;
    const double wl = elai / dmax(elai + esai, SA_MPE);
    const double ws = esai / dmax(elai + esai, SA_MPE);
    const double cosz = dmax(0.001, coszen);
    double chil = dmin(dmax(A[8], -0.4), 0.6);
    if (fabs(chil) <= 0.01) chil = 0.01;
    const double phi1 = 0.5 - 0.633 * chil - 0.330 * chil * chil;
    const double phi2 = 0.877 * (1.0 - 2.0 * phi1);
    const double gdir = phi1 + phi2 * cosz;
    const double twostext = gdir / cosz;
    const double avmu = (1.0 - phi1 / phi2 * elmk_log((phi1 + phi2) / phi1)) / phi2;
    const double temp0 = gdir + phi2 * cosz;
    const double temp1 = phi1 * cosz;
    const double temp2 = (1.0 - temp1 / temp0 * elmk_log((temp1 + temp0) / temp1));
#pragma unroll
    for (int ib = 0; ib < 2; ib++) {
      const double rho = dmax(A[0 + ib] * wl + A[2 + ib] * ws, SA_MPE);
      const double tau = dmax(A[4 + ib] * wl + A[6 + ib] * ws, SA_MPE);
      const double omegal = rho + tau;
      const double asu = 0.5 * omegal * gdir / temp0 * temp2;
      const double betadl = (1.0 + avmu * twostext) / (omegal * avmu * twostext) * asu;
      const double betail = 0.5 * ((rho + tau) + (rho - tau) * elmk_sq(((1.0 + chil) / 2.0))) / omegal;
      double tmp0, tmp1, tmp2;
      if (t_veg > TFRZ) {
        tmp0 = omegal;
        tmp1 = betadl;
        tmp2 = betail;
      } else {
        tmp0 = (1.0 - fwet) * omegal + fwet * omegas[ib];
        tmp1 = ((1.0 - fwet) * omegal * betadl + fwet * omegas[ib] * betads) / tmp0;
        tmp2 = ((1.0 - fwet) * omegal * betail + fwet * omegas[ib] * betais) / tmp0;
      }
      const double omega = tmp0;
      const double betad = tmp1;
      const double betai = tmp2;
      const double b = 1.0 - omega + omega * betai;
      const double c1 = omega * betai;
      tmp0 = avmu * twostext;
      const double d = tmp0 * omega * betad;
      const double f = tmp0 * omega * (1.0 - betad);
      tmp1 = b * b - c1 * c1;
      const double h = sqrt(tmp1) / avmu;
      const double sigma = tmp0 * tmp0 - tmp1;
      const double p1 = b + avmu * h;
      const double p2 = b - avmu * h;
      const double p3 = b + tmp0;
      const double p4 = b - tmp0;
      double t1 = dmin(h * (elai + esai), 40.0);
      const double s1 = elmk_exp(-t1);
      t1 = dmin(twostext * (elai + esai), 40.0);
      const double s2 = elmk_exp(-t1);
      // direct beam
      double u1 = b - c1 / albgrd[ib];
      double u2 = b - c1 * albgrd[ib];
      const double u3 = f + c1 * albgrd[ib];
      tmp2 = u1 - avmu * h;
      double tmp3 = u1 + avmu * h;
      double d1 = p1 * tmp2 / s1 - p2 * tmp3 * s1;
      double tmp4 = u2 + avmu * h;
      double tmp5 = u2 - avmu * h;
      double d2 = tmp4 / s1 - tmp5 * s1;
      const double h1 = -d * p4 - c1 * f;
      const double tmp6 = d - h1 * p3 / sigma;
      const double tmp7 = (d - c1 - h1 / sigma * (u1 + tmp0)) * s2;
      const double h2 = (tmp6 * tmp2 / s1 - p2 * tmp7) / d1;
      const double h3 = -(tmp6 * tmp3 * s1 - p1 * tmp7) / d1;
      const double h4 = -f * p3 - c1 * d;
      const double tmp8 = h4 / sigma;
      const double tmp9 = (u3 - tmp8 * (u2 - tmp0)) * s2;
      const double h5 = -(tmp8 * tmp4 / s1 + tmp9) / d2;
      const double h6 = (tmp8 * tmp5 * s1 + tmp9) / d2;
      albd[ib] = h1 / sigma + h2 + h3;
      ftid[ib] = h4 * s2 / sigma + h5 * s1 + h6 / s1;
      ftdd[ib] = s2;
      fabd[ib] = 1.0 - albd[ib] - (1.0 - albgrd[ib]) * ftdd[ib] - (1.0 - albgri[ib]) * ftid[ib];
      double a1 = h1 / sigma * (1.0 - s2 * s2) / (2.0 * twostext) + h2 * (1.0 - s2 * s1) / (twostext + h) +
                  h3 * (1.0 - s2 / s1) / (twostext - h);
      double a2 = h4 / sigma * (1.0 - s2 * s2) / (2.0 * twostext) + h5 * (1.0 - s2 * s1) / (twostext + h) +
                  h6 * (1.0 - s2 / s1) / (twostext - h);
      const double fabd_sun = (1.0 - omega) * (1.0 - s2 + 1.0 / avmu * (a1 + a2));
      const double fabd_sha = fabd[ib] - fabd_sun;  // wrapper-local in the reference (albedo_kokkos.cc:27-28)
      // diffuse
      u1 = b - c1 / albgri[ib];
      u2 = b - c1 * albgri[ib];
      tmp2 = u1 - avmu * h;
      tmp3 = u1 + avmu * h;
      d1 = p1 * tmp2 / s1 - p2 * tmp3 * s1;
      tmp4 = u2 + avmu * h;
      tmp5 = u2 - avmu * h;
      d2 = tmp4 / s1 - tmp5 * s1;
      const double h7 = (c1 * tmp2) / (d1 * s1);
      const double h8 = (-c1 * tmp3 * s1) / d1;
      const double h9 = tmp4 / (d2 * s1);
      const double h10 = (-tmp5 * s1) / d2;
      albi[ib] = h7 + h8;
      ftii[ib] = h9 * s1 + h10 / s1;
      fabi[ib] = 1.0 - albi[ib] - (1.0 - albgri[ib]) * ftii[ib];
      a1 = h7 * (1.0 - s2 * s1) / (twostext + h) + h8 * (1.0 - s2 / s1) / (twostext - h);
      a2 = h9 * (1.0 - s2 * s1) / (twostext + h) + h10 * (1.0 - s2 / s1) / (twostext - h);
      fabi_sun[ib] = (1.0 - omega) / avmu * (a1 + a2);
      fabi_sha[ib] = fabi[ib] - fabi_sun[ib];
      if (ib == 0) {
        fsun_z = (1.0 - s2) / t1;
        const double laisum = elai + esai;
        fabd_sun_z = fabd_sun / (fsun_z * laisum);
        fabi_sun_z = fabi_sun[ib] / (fsun_z * laisum);
        fabd_sha_z = fabd_sha / ((1.0 - fsun_z) * laisum);
        fabi_sha_z = fabi_sha[ib] / ((1.0 - fsun_z) * laisum);
        const double extkb = twostext;
        vcmaxcintsun = (1.0 - elmk_exp(-(SA_EXTKN + extkb) * elai)) / (SA_EXTKN + extkb);
        vcmaxcintsha = (1.0 - elmk_exp(-SA_EXTKN * elai)) / SA_EXTKN - vcmaxcintsun;
        if (elai > 0.0) {
          vcmaxcintsun = vcmaxcintsun / (fsun_z * elai);
          vcmaxcintsha = vcmaxcintsha / ((1.0 - fsun_z) * elai);
        } else {
          vcmaxcintsun = 0.0;
          vcmaxcintsha = 0.0;
        }
      }
    }
  } else {  // novegsol (:672-686)
#pragma unroll
    for (int ib = 0; ib < 2; ++ib) {
      fabd[ib] = 0.0;
      fabi[ib] = 0.0;
      fabi_sun[ib] = 0.0;
      fabi_sha[ib] = 0.0;
      ftdd[ib] = 1.0;
      ftid[ib] = 0.0;
      ftii[ib] = 1.0;
      albd[ib] = albgrd[ib];
      albi[ib] = albgri[ib];
    }
  }
#pragma unroll
  for (int ib = 0; ib < 2; ib++) {
    LV(albd, ib) = albd[ib];
    LV(albi, ib) = albi[ib];
    LV(ftdd, ib) = ftdd[ib];
    LV(ftid, ib) = ftid[ib];
    LV(ftii, ib) = ftii[ib];
    LV(fabd, ib) = fabd[ib];
    LV(fabi, ib) = fabi[ib];
    LV(fabi_sun, ib) = fabi_sun[ib];
    LV(fabi_sha, ib) = fabi_sha[ib];
  }
  S->vcmaxcintsun[c] = vcmaxcintsun;
  S->vcmaxcintsha[c] = vcmaxcintsha;
  S->fsun_z[c] = fsun_z;
  S->fabd_sun_z[c] = fabd_sun_z;
  S->fabd_sha_z[c] = fabd_sha_z;
  S->fabi_sun_z[c] = fabi_sun_z;
  S->fabi_sha_z[c] = fabi_sha_z;
}

// =====================================================================================================
// stage 1 (coalesced, every column): canopy_layer_lai, soil albedo of the sunlit columns, and the classification of
// the sunlit snow-covered columns by their number of snow layers (lists LIST_ALB_1..5)
// =====================================================================================================
#ifndef ALB_CLASSIFY_THREADS
#define ALB_CLASSIFY_THREADS 1024  // one atomic per workgroup and non-empty queue: with 256-thread workgroups the 3 907 atomics on
                                   // ONE counter (the fixture-tiled tier fills a single queue) were the kernel's whole time (57 us)
#endif
__global__ __launch_bounds__(ALB_CLASSIFY_THREADS) void k_alb_classify(const DevState* __restrict__ S)
{
  const Land L = S->land;
  // (stage 1 evaluates exp on deep-lake columns only - the ice fraction of soil_albedo: no table copy for the other land units)
  if (L.ltype == istdlak) elmk_math_lds_init<false>();
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t ld = S->ld;
  if (L.urbpoi) return;  // every routine of this wrapper is a no-op on urban points
  const bool inside = c < S->ncols;
  int nl = -1;  // >= 1: sunlit column with snow, goes through SNICAR with nl layers
  if (inside) nl = alb_main_column(S, c, ld, L);
  block_classify_append<5>(S->lists, ld, S->counters, LIST_ALB_1, nl >= 1 ? nl - 1 : -1, (int32_t)c);
}

// =====================================================================================================
// stage 2: SNICAR for the sunlit snow-covered columns with exactly NL (possibly fictitious) snow layers.
// Ten lanes per column - (direct, diffuse) x 5 bands - six columns per wave; lanes 60..63 idle.
// The two SnowOut (one per pass) go to the scratch array alb_snow, by column.
// =====================================================================================================
// (alb_snow holds 28 doubles per column: pass x {alb[2], fabs_[6][2]})
template <int NL>
__global__ __launch_bounds__(256, 2) void k_alb_snicar(const DevState* __restrict__ S)
{
  snicar_workgroup<NL>(S, blockIdx.x, gridDim.x);
}
// the queue of NL >= 2 layers
template <int NL>
static void launch_snicar_deep(const DevState* S, const unsigned capped, const int64_t n, hipStream_t st)
{
  hipLaunchKernelGGL(k_alb_snicar<NL>, dim3(capped), dim3(256), 0, st, S);
}
// The four queues of 5..2 layers as ONE launch: the first `per_queue` workgroups take the five-layer queue, the next the
// four-layer queue, ...; every launch costs ~4.5 us whether its queue holds columns or not (four empty queues on a snow-free
// region were 18 us of the wrapper), and a queue's tail overlaps the next queue's start.  The price: one register allocation
// for all (the five-layer body's), which takes the two-layer queue from three waves per SIMD to two.
#ifndef ALB_DEEP_MERGED
#define ALB_DEEP_MERGED 1
#endif
#ifndef ALB_DEEP_PER_QUEUE
#define ALB_DEEP_PER_QUEUE 1024
#endif
__global__ __launch_bounds__(256, 2) void k_alb_snicar_deep(const DevState* __restrict__ S, const unsigned per_queue)
{
  const unsigned q = blockIdx.x / per_queue, b = blockIdx.x - q * per_queue;
  switch (q) {
    case 0: snicar_workgroup<5>(S, b, per_queue); break;
    case 1: snicar_workgroup<4>(S, b, per_queue); break;
    case 2: snicar_workgroup<3>(S, b, per_queue); break;
    default: snicar_workgroup<2>(S, b, per_queue);
  }
}
static void launch_snicar_deep_all(const DevState* S, const unsigned capped, const int64_t n, hipStream_t st)
{
  if (ALB_DEEP_MERGED) {
    // (workgroups walk their queue with a stride, so fewer of them do the same work; an empty queue costs what its workgroups
    //  cost to dispatch - 16 384 of them 14 us, 4 096 the 4.7 us of any launch)
    const unsigned per_queue = capped < (unsigned)ALB_DEEP_PER_QUEUE ? capped : (unsigned)ALB_DEEP_PER_QUEUE;
    hipLaunchKernelGGL(k_alb_snicar_deep, dim3(4u * per_queue), dim3(256), 0, st, S, per_queue);
  } else {
    launch_snicar_deep<5>(S, capped, n, st);
    launch_snicar_deep<4>(S, capped, n, st);
    launch_snicar_deep<3>(S, capped, n, st);
    launch_snicar_deep<2>(S, capped, n, st);
  }
}

// =====================================================================================================
// stage 3 (coalesced, every column): night defaults (init_timestep), or for a sunlit column ground_albedo,
// flux_absorption_factor and the canopy two-stream solution from the soil albedos and the SNICAR products
// =====================================================================================================
__global__ __launch_bounds__(256) void k_alb_final(const DevState* __restrict__ S)
{
  elmk_math_lds_init<false>();
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t ld = S->ld;
  const Land L = S->land;
  // The layer-count queues have been drained by now: leave them empty for the next classification (they are empty at
  // context creation and after every call, so neither k_alb_classify nor the fused step's k_fz_prep resets them).
  if (blockIdx.x == 0 && threadIdx.x < 6) {
    ELMK_LIST_COUNT(S, LIST_ALB_0 + threadIdx.x) = 0u;
    ELMK_LIST_HEAD(S, LIST_ALB_0 + threadIdx.x) = 0u;
  }
  if (L.urbpoi || c >= S->ncols) return;
  const double coszen = S->coszen[c];
  const double elai = S->elai[c];
  // init_timestep values of the leaf-to-canopy scaling coefficients (overwritten by two_stream where vegetated)
  double vcmaxcintsun = 0.0;
  double vcmaxcintsha = (1.0 - elmk_exp(-SA_EXTKN * elai)) / SA_EXTKN;
  if (elai > 0.0) {
    vcmaxcintsha /= elai;
  } else {
    vcmaxcintsha = 0.0;
  }
  const bool day = coszen > 0.0;  // nothing after init_timestep runs without sun except snow_albedo_radiation_factor's
                                  // "no sun" branch (:758-765): alb_finish stores those values for the dark lanes
  double albsod[2] = {0.0, 0.0}, albsoi[2] = {0.0, 0.0};
  double esai = 0.0, frac_sno = 0.0;
  SnowOut sd, si;
#pragma unroll
  for (int i = 0; i < 6; i++) {
    sd.fabs_[i][0] = sd.fabs_[i][1] = 0.0;
    si.fabs_[i][0] = si.fabs_[i][1] = 0.0;
  }
  sd.alb[0] = sd.alb[1] = si.alb[0] = si.alb[1] = 0.0;
  if (day) {
    esai = S->esai[c];
    frac_sno = S->frac_sno[c];
    const double h2osno = S->h2osno[c];
    albsod[0] = LV(albsod, 0);  // written by stage 1
    albsod[1] = LV(albsod, 1);
    albsoi[0] = LV(albsoi, 0);
    albsoi[1] = LV(albsoi, 1);
    if (h2osno > SN_MIN_SNW) {
      const gptr<const double> o = S->alb_snow + c;
      sd.alb[0] = o[0];
      sd.alb[1] = o[ld];
      si.alb[0] = o[(int64_t)14 * ld];
      si.alb[1] = o[(int64_t)15 * ld];
#pragma unroll
      for (int i = 0; i < 6; i++) {
        sd.fabs_[i][0] = o[(int64_t)(2 + 2 * i) * ld];
        sd.fabs_[i][1] = o[(int64_t)(3 + 2 * i) * ld];
        si.fabs_[i][0] = o[(int64_t)(16 + 2 * i) * ld];
        si.fabs_[i][1] = o[(int64_t)(17 + 2 * i) * ld];
      }
    } else if (h2osno < SN_MIN_SNW && h2osno > 0.0) {
      // no snow radiative transfer: snow_albedo_radiation_factor's remaining branches (snow_snicar_impl.hh:758-765)
      sd.alb[0] = si.alb[0] = albsoi[0];
      sd.alb[1] = si.alb[1] = albsoi[1];
    }
  }
  alb_finish(S, c, ld, L, day, coszen, elai, esai, frac_sno, albsod, albsoi, sd, si, vcmaxcintsun, vcmaxcintsha);
}

void launch_albedo_snicar(const DevState* S, int64_t n, hipStream_t st, const SideStreams* side, bool classify)
{
  if (n <= 0) return;
  const dim3 block(256);
  const unsigned full = (unsigned)((n + 255) / 256);
  // stage 2 is grid-stride over a device-side count: 24 columns per workgroup
  const unsigned want = (unsigned)((n + 23) / 24);
  const unsigned capped = want < 4096u ? want : 4096u;
  if (classify)
    hipLaunchKernelGGL(k_alb_classify, dim3((unsigned)((n + ALB_CLASSIFY_THREADS - 1) / ALB_CLASSIFY_THREADS)), dim3(ALB_CLASSIFY_THREADS), 0, st, S);
  // The five layer-count queues are independent.  (One persistent launch draining all five lists through a chunk counter
  // was measured 30 % slower: every wave then pays the deepest list's register footprint, and the five unrolled bodies
  // compete for the instruction cache.)  With many columns every non-empty queue fills the GPU by itself and an empty one
  // costs a few microseconds, so the launches simply follow each other; the fork and join through side streams cost
  // ~35 us of dependency latency per call and only pay when the queues are too short to fill the machine.
  if (n >= 262144) {
    launch_snicar_deep_all(S, capped, n, st);
    hipLaunchKernelGGL(k_alb_snicar<1>, dim3(capped), block, 0, st, S);
  } else {
    (void)hipEventRecord(side->fork, st);
    for (int i = 0; i < 4; i++) (void)hipStreamWaitEvent(side->s[i], side->fork, 0);
    launch_snicar_deep<5>(S, capped, n, st);  // longest work on the caller's stream
    launch_snicar_deep<4>(S, capped, n, side->s[0]);
    launch_snicar_deep<3>(S, capped, n, side->s[1]);
    launch_snicar_deep<2>(S, capped, n, side->s[2]);
    hipLaunchKernelGGL(k_alb_snicar<1>, dim3(capped), block, 0, side->s[3], S);
    for (int i = 0; i < 4; i++) {
      (void)hipEventRecord(side->join[i], side->s[i]);
      (void)hipStreamWaitEvent(st, side->join[i], 0);
    }
  }
  hipLaunchKernelGGL(k_alb_final, dim3(full), block, 0, st, S);
}

// The same stage in two parts around a kernel of the caller's that does the single-layer SNICAR queue itself (the fused step's
// k_fz_snicar_pre): part 0 = the queues of 5..2 layers, part 1 = k_alb_final.  *snicar_grid: the grid k_alb_snicar<1> would get.
void launch_albedo_snicar_part(const DevState* S, int64_t n, hipStream_t st, int part, unsigned* snicar_grid)
{
  if (n <= 0) return;
  const dim3 block(256);
  const unsigned want = (unsigned)((n + 23) / 24);
  const unsigned capped = want < 4096u ? want : 4096u;
  if (snicar_grid) *snicar_grid = capped;
  if (part == 0) {
    launch_snicar_deep_all(S, capped, n, st);
  } else {
    hipLaunchKernelGGL(k_alb_final, dim3((unsigned)((n + 255) / 256)), block, 0, st, S);
  }
}

}  // namespace elmk
