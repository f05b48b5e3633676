// Micro-benchmark (development, GPU box): ns per elmk_math call per wave at one wave per SIMD, for one dependent chain
// of calls vs four independent chains (what the compiler interleaves across the functions' special-case branches), LDS tables.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I elmkernels_amd/csrc math_issue.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ELMK_MATH_LDS 1
#include "elmk_math.h"

template <int ILP, int OP>
__global__ __launch_bounds__(256) void k(double* out, double a, int iters)
{
  elmk_math_lds_init<true>();
  double x[ILP];
  for (int j = 0; j < ILP; j++) x[j] = a + threadIdx.x * 1e-3 + j * 0.01;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
      for (int j = 0; j < ILP; j++) {
        if (OP == 0) x[j] = elmk_exp(x[j] * 1e-3) + 0.5;
        if (OP == 1) x[j] = elmk_log(x[j] + 1.5) + 2.0;
        if (OP == 2) x[j] = elmk_pow(x[j], 0.333) + 1.0;
        if (OP == 3) x[j] = elmk_atan(x[j]) + 1.0;
        if (OP == 4) x[j] = exp(x[j] * 1e-3) + 0.5;
        if (OP == 5) x[j] = log(x[j] + 1.5) + 2.0;
        if (OP == 6) x[j] = pow(x[j], 0.333) + 1.0;
        if (OP == 9) x[j] = elmk_erf(x[j] * 0.3) + 1.0;
        if (OP == 10) x[j] = erf(x[j] * 0.3) + 1.0;
        if (OP == 11) x[j] = elmk_tanh(x[j] * 0.3) + 1.0;
        if (OP == 12) x[j] = tanh(x[j] * 0.3) + 1.0;
        if (OP == 13) x[j] = elmk_cos(x[j]) + 1.5;
        if (OP == 14) x[j] = cos(x[j]) + 1.5;
        if (OP == 7) x[j] = elmk_log(x[j] * 1e-3 + 1.0) + 2.0;  // arguments near 1
        if (OP == 8) x[j] = log(x[j] * 1e-3 + 1.0) + 2.0;
      }
    }
  }
  double s = 0;
  for (int j = 0; j < ILP; j++) s += x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ILP, int OP>
void run(const char* name, double* d)
{
  const int iters = 1000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<ILP, OP>), dim3(256), dim3(256), 0, 0, d, 1.2, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<ILP, OP>), dim3(256), dim3(256), 0, 0, d, 1.2, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-10s ILP %d: %.1f ns per call per wave (1 wave per SIMD)\n", name, ILP, ms * 1e6 / ((double)iters * 4 * ILP));
}

int main()
{
  double* d;
  hipMalloc(&d, (1 << 20) * 8);
  run<1, 0>("elmk_exp", d); run<4, 0>("elmk_exp", d); run<1, 4>("ocml exp", d); run<4, 4>("ocml exp", d);
  run<1, 1>("elmk_log", d); run<4, 1>("elmk_log", d); run<1, 5>("ocml log", d); run<4, 5>("ocml log", d);
  run<1, 7>("elmk_log~1", d); run<4, 7>("elmk_log~1", d); run<1, 8>("ocml log~1", d);
  run<1, 2>("elmk_pow", d); run<4, 2>("elmk_pow", d); run<1, 6>("ocml pow", d); run<4, 6>("ocml pow", d);
  run<1, 3>("elmk_atan", d); run<4, 3>("elmk_atan", d);
  run<1, 9>("elmk_erf", d); run<1, 10>("ocml erf", d); run<1, 11>("elmk_tanh", d); run<1, 12>("ocml tanh", d);
  run<1, 13>("elmk_cos", d); run<1, 14>("ocml cos", d);
  return 0;
}
