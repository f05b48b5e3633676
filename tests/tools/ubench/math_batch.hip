// Micro-benchmark (development, GPU box): ns per elmk_math call per wave at one wave per SIMD - one dependent chain of scalar
// calls against the batched forms (elmk_exp_n / log_n / pow_n / atan_n: N main paths in one basic block), LDS tables.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I elmkernels_amd/csrc -I include tests/tools/ubench/math_batch.hip -o /tmp/mb && /tmp/mb
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ELMK_MATH_LDS 1
#include "elmk_math.h"

template <int N, int OP>
__global__ __launch_bounds__(256) void k(double* out, double a, int iters)
{
  elmk_math_lds_init<true>();
  double x[N];
  for (int j = 0; j < N; j++) x[j] = a + threadIdx.x * 1e-3 + j * 0.01;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      double v[N];
#pragma unroll
      for (int j = 0; j < N; j++) {
        if (OP == 0) v[j] = x[j] * 1e-3;
        if (OP == 1) v[j] = x[j] + 1.5;
        if (OP == 2) v[j] = x[j];
        if (OP == 3) v[j] = x[j];
        if (OP == 4) v[j] = x[j] * 1e-3 + 1.0;  // log near 1
      }
      if (OP == 0) elmk_exp_n<N>(v);
      if (OP == 1 || OP == 4) elmk_log_n<N>(v);
      if (OP == 2) {
        double e[N];
#pragma unroll
        for (int j = 0; j < N; j++) e[j] = 0.333;
        elmk_pow_n<N>(v, e);
      }
      if (OP == 3) elmk_atan_n<N>(v);
#pragma unroll
      for (int j = 0; j < N; j++) x[j] = v[j] + (OP == 0 ? 0.5 : (OP == 1 || OP == 4 ? 2.0 : 1.0));
    }
  }
  double s = 0;
  for (int j = 0; j < N; j++) s += x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int N, int OP>
void run(const char* name, double* d)
{
  const int iters = 1000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<N, OP>), dim3(256), dim3(256), 0, 0, d, 1.2, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<N, OP>), dim3(256), dim3(256), 0, 0, d, 1.2, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-10s batch %d: %.1f ns per call per wave (1 wave per SIMD)\n", name, N, ms * 1e6 / ((double)iters * 4 * N));
}

int main()
{
  double* d;
  hipMalloc(&d, (1 << 20) * 8);
  run<1, 0>("exp", d); run<2, 0>("exp", d); run<3, 0>("exp", d); run<4, 0>("exp", d);
  run<1, 1>("log", d); run<2, 1>("log", d); run<3, 1>("log", d); run<4, 1>("log", d);
  run<1, 4>("log~1", d); run<2, 4>("log~1", d); run<4, 4>("log~1", d);
  run<1, 2>("pow", d); run<2, 2>("pow", d); run<4, 2>("pow", d);
  run<1, 3>("atan", d); run<2, 3>("atan", d); run<4, 3>("atan", d);
  return 0;
}
