// Micro-benchmark (development, GPU box): how much HBM bandwidth does a streaming kernel get at LOW occupancy, as a function of
// the loads a wave keeps in flight?  The soil-column solve runs two waves per SIMD (236 VGPRs, 159 KB of LDS per 512-thread
// workgroup) and moves its bytes at 5.0 TB/s; the many-row copy of chunk_layout.hip at full occupancy reaches 5.9 TB/s at the same
// 10 M columns.  Here: one thread per column, 512-thread workgroups, K rows read and K rows written of a [row][column] arena, G loads
// issued back to back before the first use, dynamic LDS sized so that WPS waves per SIMD are resident.
//   hipcc --offload-arch=gfx950 -O3 occupancy_bw.hip -o occupancy_bw && ./occupancy_bw
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

extern __shared__ double dyn_lds[];

template <int K, int G, int T>
__global__ __launch_bounds__(T) void k(double* __restrict__ a, int64_t n)
{
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (threadIdx.x == 0) dyn_lds[0] = 0.0;  // (the allocation must be referenced)
  if (c >= n) return;
  double acc = 0.0;
#pragma unroll 1
  for (int g = 0; g < K; g += G) {
    double v[G];
#pragma unroll
    for (int j = 0; j < G; j++) v[j] = a[c + (int64_t)(g + j) * n];
#pragma unroll
    for (int j = 0; j < G; j++) acc += v[j];
#pragma unroll
    for (int j = 0; j < G; j++) a[c + (int64_t)(K + g + j) * n] = v[j] + acc * 1e-300;
  }
}

template <int K, int G, int T>
void run(double* a, int64_t n, int wps)
{
  // waves per SIMD = workgroups per CU * (T / 64) / 4; LDS per workgroup chosen so that exactly that many workgroups fit in 160 KB
  const int wg_per_cu = wps * 4 / (T / 64);
  const size_t lds = wg_per_cu > 0 ? (size_t)(160 * 1024 / wg_per_cu) - 512 : 0;
  (void)hipFuncSetAttribute((const void*)k<K, G, T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const unsigned grid = (unsigned)((n + T - 1) / T);
  hipLaunchKernelGGL((k<K, G, T>), dim3(grid), dim3(T), lds, 0, a, n);
  (void)hipEventRecord(e0);
  const int it = 3;
  for (int i = 0; i < it; i++) hipLaunchKernelGGL((k<K, G, T>), dim3(grid), dim3(T), lds, 0, a, n);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("threads %4d  waves/SIMD %d  K %3d+%3d rows  %2d loads in flight  %7.1f GB/s  (%.3f ms)  %s\n", T, wps, K, K, G,
         2.0 * K * 8.0 * n * it / (ms * 1e-3) / 1e9, ms / it, hipGetErrorString(hipGetLastError()));
  fflush(stdout);
}

int main()
{
  const int64_t n = 10 * (1 << 20);
  const int K = 128;
  double* a;
  if (hipMalloc(&a, (size_t)2 * K * n * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  (void)hipMemset(a, 0, (size_t)2 * K * n * 8);
  for (int wps : {1, 2, 4, 8}) {
    run<K, 4, 512>(a, n, wps);
    run<K, 8, 512>(a, n, wps);
    run<K, 16, 512>(a, n, wps);
    run<K, 32, 512>(a, n, wps);
    run<K, 64, 512>(a, n, wps);
  }
  for (int wps : {1, 2, 4}) {
    run<K, 16, 256>(a, n, wps);
    run<K, 32, 256>(a, n, wps);
  }
  return 0;
}
