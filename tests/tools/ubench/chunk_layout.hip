// Micro-benchmark (development, GPU box): what would a CHUNKED column-state layout buy the streaming kernels?
// One thread per column reads K rows and writes K other rows of ONE arena (8 bytes per lane each), like a physics body that
// streams a column state: the arena is [chunk of CH columns][2K rows][CH]; CH = n is the SoA layout elmk uses today (a
// column's rows lie n*8 bytes apart: 8 MB at 1 M columns, 80 MB at 10 M), CH = 64 the wave tile of profiles/r03_stream_layout_ubench.txt.
// Loads are issued in groups of G with a dependent use between groups.  NT = nontemporal loads and stores.
//   hipcc --offload-arch=gfx950 -O3 chunk_layout.hip -o chunk_layout && ./chunk_layout
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

template <int K, int G, bool NT>
__global__ __launch_bounds__(256) void k(double* __restrict__ a, int64_t n, int64_t ch, int64_t chunk_stride)
{
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  const int64_t q = c / ch;  // (ch is a power of two or n in every call below; the compiler sees a 64-bit divide, once)
  const int64_t base = q * chunk_stride + (c - q * ch);
  double acc = 0.0;
#pragma unroll
  for (int g = 0; g < K; g += G) {
    double v[G];
#pragma unroll
    for (int j = 0; j < G; j++) v[j] = NT ? __builtin_nontemporal_load(&a[base + (int64_t)(g + j) * ch]) : a[base + (int64_t)(g + j) * ch];
#pragma unroll
    for (int j = 0; j < G; j++) acc += v[j];
#pragma unroll
    for (int j = 0; j < G; j++) {
      if (NT) __builtin_nontemporal_store(v[j] + acc * 1e-300, &a[base + (int64_t)(K + g + j) * ch]);
      else a[base + (int64_t)(K + g + j) * ch] = v[j] + acc * 1e-300;
    }
    if (G < K) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

template <int K, int G, bool NT>
void run(double* a, int64_t n, int64_t ch)
{
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int64_t nch = (n + ch - 1) / ch;
  const int64_t chunk_stride = 2 * K * ch;
  (void)nch;
  const unsigned grid = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL((k<K, G, NT>), dim3(grid), dim3(256), 0, 0, a, n, ch, chunk_stride);
  (void)hipEventRecord(e0);
  const int it = 5;
  for (int i = 0; i < it; i++) hipLaunchKernelGGL((k<K, G, NT>), dim3(grid), dim3(256), 0, 0, a, n, ch, chunk_stride);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("n %9lld  K %3d+%3d rows  group %3d  %s  CH %9lld  %7.1f GB/s  (%.3f ms)\n", (long long)n, K, K, G, NT ? "nt" : "  ", (long long)ch,
         2.0 * K * 8.0 * n * it / (ms * 1e-3) / 1e9, ms / it);
  fflush(stdout);
}

// odd = false: chunk sizes that are powers of two; odd = true: + 64 columns each (a row stride of exactly 32 KB could alias HBM channels)
template <int K, int G>
void sweep(double* a, int64_t n, bool odd)
{
  const int64_t o = odd ? 64 : 0;
  const int64_t chs[] = {n, 65536 + o, 16384 + o, 4096 + o, 1024 + o, 256 + o, odd ? 4096 : 64};
  for (int64_t ch : chs) run<K, G, false>(a, n, ch);
  run<K, G, true>(a, n, n);
  run<K, G, true>(a, n, 4096 + o);
  if (odd) run<K, G, true>(a, n, 16384 + o);
}

int main(int argc, char** argv)
{
  const int64_t nmax = 10 * (1 << 20);
  const int KMAX = 128;
  double* a;
  if (hipMalloc(&a, (size_t)2 * KMAX * nmax * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  (void)hipMemset(a, 0, (size_t)2 * KMAX * nmax * 8);
  if (argc > 1) {  // ./chunk_layout odd: the product's column counts, chunk sizes that are not powers of two
    for (int64_t n : {(int64_t)1000000, (int64_t)10000000}) {
      sweep<128, 8>(a, n, true);
      sweep<128, 32>(a, n, true);
    }
    return 0;
  }
  for (int64_t n : {(int64_t)1 << 20, nmax}) {
    sweep<64, 8>(a, n, false);
    sweep<128, 8>(a, n, false);
    sweep<128, 32>(a, n, false);
  }
  return 0;
}
