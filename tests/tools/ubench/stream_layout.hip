// Micro-benchmark (development, GPU box): does the HBM layout of a many-field column state bound the streaming kernels?
// One thread per column reads K fields and writes K fields (8 bytes per lane each), columns = 1 M, in two layouts:
//   soa    field-major  [field][column]           - what elmk uses: K read streams + K write streams, 8 MB apart
//   tiled  [tile of 64 columns][field][64]        - a wave's K loads fall into one contiguous K*512-byte block
// and with the loads issued (a) all up front, (b) in groups of G with a dependent use between groups (what a physics body
// with control flow between its loads looks like).
//   hipcc --offload-arch=gfx950 -O3 stream_layout.hip -o stream_layout && ./stream_layout
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int K, int G, bool TILED>
__global__ __launch_bounds__(256) void k(const double* __restrict__ in, double* __restrict__ out, int64_t n, int64_t ld)
{
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  const int64_t base = TILED ? (c >> 6) * (int64_t)(K * 64) + (c & 63) : c;
  const int64_t fs = TILED ? 64 : ld;
  double acc = 0.0;
#pragma unroll
  for (int g = 0; g < K; g += G) {
    double v[G];
#pragma unroll
    for (int j = 0; j < G; j++) v[j] = in[base + (int64_t)(g + j) * fs];
    // a dependent use: the next group's addresses wait for nothing, but the stores of this group need acc
#pragma unroll
    for (int j = 0; j < G; j++) acc += v[j];
#pragma unroll
    for (int j = 0; j < G; j++) out[base + (int64_t)(g + j) * fs] = v[j] + acc * 1e-300;
    if (G < K) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // a body that consumes its inputs before it reads on
  }
}

template <int K, int G, bool TILED>
void run(const double* in, double* out, int64_t n, int64_t ld)
{
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const unsigned grid = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL((k<K, G, TILED>), dim3(grid), dim3(256), 0, 0, in, out, n, ld);
  (void)hipEventRecord(e0);
  const int it = 10;
  for (int i = 0; i < it; i++) hipLaunchKernelGGL((k<K, G, TILED>), dim3(grid), dim3(256), 0, 0, in, out, n, ld);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("K %3d fields  group %3d  %-5s  %7.1f GB/s  (%.3f ms)\n", K, G, TILED ? "tiled" : "soa", 2.0 * K * 8.0 * n * it / (ms * 1e-3) / 1e9, ms / it);
}

int main()
{
  const int64_t n = 1 << 20, ld = n;
  const int KMAX = 128;
  double *in, *out;
  (void)hipMalloc(&in, (size_t)KMAX * ld * 8);
  (void)hipMalloc(&out, (size_t)KMAX * ld * 8);
  (void)hipMemset(in, 0, (size_t)KMAX * ld * 8);
  (void)hipMemset(out, 0, (size_t)KMAX * ld * 8);
  run<16, 16, false>(in, out, n, ld); run<16, 16, true>(in, out, n, ld);
  run<64, 64, false>(in, out, n, ld); run<64, 64, true>(in, out, n, ld);
  run<64, 16, false>(in, out, n, ld); run<64, 16, true>(in, out, n, ld);
  run<64, 8, false>(in, out, n, ld);  run<64, 8, true>(in, out, n, ld);
  run<64, 4, false>(in, out, n, ld);  run<64, 4, true>(in, out, n, ld);
  run<128, 8, false>(in, out, n, ld); run<128, 8, true>(in, out, n, ld);
  run<128, 32, false>(in, out, n, ld); run<128, 32, true>(in, out, n, ld);
  return 0;
}
