// Micro-benchmark (development, GPU box): cycles per wave64 fp64 instruction for dependent chains vs independent
// streams, at one and two waves per SIMD.  hipcc --offload-arch=gfx950 -O3 fp64_issue.hip -o fp64_issue && ./fp64_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int ILP, int OP>
__global__ __launch_bounds__(256) void k(double* out, double a, double b, int iters)
{
  double x[ILP];
  for (int j = 0; j < ILP; j++) x[j] = a + threadIdx.x * 1e-9 + j;
  const uint64_t t0 = clock64();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
#pragma unroll
      for (int j = 0; j < ILP; j++) {
        if (OP == 0) x[j] = __builtin_fma(x[j], a, b);
        if (OP == 1) x[j] = x[j] * a;
        if (OP == 2) x[j] = x[j] + b;
        if (OP == 3) x[j] = b / x[j];
        if (OP == 4) x[j] = sqrt(x[j]);
      }
    }
  }
  const uint64_t t1 = clock64();
  double s = 0;
  for (int j = 0; j < ILP; j++) s += x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) ((uint64_t*)out)[1 << 20] = t1 - t0;
}

template <int ILP, int OP>
void run(const char* name, double* d, int wg_per_cu)
{
  const int iters = 2000;
  uint64_t cyc = 0;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grid = 256 * wg_per_cu;  // 256 CUs; a workgroup is 4 waves = one per SIMD
  hipLaunchKernelGGL((k<ILP, OP>), dim3(grid), dim3(256), 0, 0, d, 1.0000001, 1e-9, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<ILP, OP>), dim3(grid), dim3(256), 0, 0, d, 1.0000001, 1e-9, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(&cyc, (uint64_t*)d + (1 << 20), 8, hipMemcpyDeviceToHost);
  const double n = (double)iters * 16 * ILP;
  printf("%-6s ILP %d  waves/SIMD %d: %.2f shader-clock ticks per op per wave, %.3f ms -> %.2f ns per op per SIMD-slot\n", name, ILP,
         wg_per_cu, (double)cyc / n, ms, ms * 1e6 / (n * wg_per_cu));
}

int main()
{
  double* d;
  hipMalloc(&d, ((1 << 20) + 8) * 8);
  for (int w = 1; w <= 2; w++) {
    run<1, 0>("fma", d, w); run<2, 0>("fma", d, w); run<4, 0>("fma", d, w);
    run<1, 1>("mul", d, w); run<4, 1>("mul", d, w);
    run<1, 2>("add", d, w); run<4, 2>("add", d, w);
    run<1, 3>("div", d, w); run<2, 3>("div", d, w);
    run<1, 4>("sqrt", d, w); run<2, 4>("sqrt", d, w);
  }
  return 0;
}
