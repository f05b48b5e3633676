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
#include "elmk_albedo_fin.h"

namespace elmk {


// (stage 3's per-column pieces - ground_albedo, flux_absorption_factor, two_stream_solver - live in elmk_albedo_fin.h: the fused
//  step's k_fz_stream runs the same source around surface_radiation's body)

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
  const AlbIn x = alb_final_inputs(S, c, ld, S->frac_sno[c], S->h2osno[c]);
  AlbFwd a;
  double flx[6][4];
  alb_ground(S, c, ld, x.day, x.frac_sno, x.albsod, x.albsoi, x.sd_alb, x.si_alb, a);
  alb_flux_abs_all(S, c, ld, L, x, flx);
  alb_two_stream(S, c, ld, L, x.day, x.coszen, x.elai, x.esai, x.vcmaxcintsun, x.vcmaxcintsha, a);
}

void launch_albedo_snicar(const DevState* S, int64_t n, hipStream_t st, const SideStreams* side, bool classify, bool final)
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
  if (final) hipLaunchKernelGGL(k_alb_final, dim3(full), block, 0, st, S);
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
