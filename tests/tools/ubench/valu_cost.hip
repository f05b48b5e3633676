// Micro-benchmark (development, GPU box): issue cost of the VALU instructions k_cf_iterate is made of, as SIMD time per
// wave64 instruction with 1, 2 and 4 waves per SIMD (8 independent streams per wave, so dependencies do not bound it).
//   hipcc --offload-arch=gfx950 -O3 valu_cost.hip -o valu_cost && ./valu_cost
// The instruction-count model of DESIGN.md prices a trip with these numbers.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

enum { FMA64, ADD64, MUL64, RCP64, SQRT64, RSQ64, DIVSCALE, DIVFMAS, DIVFIXUP, LDEXP64, MAX64, CMP64, CVT_I2D, MOV32, MOV64, CNDMASK, ADDU32, AND32, FMA32, LSHLADD64, NOPS };
static const char* NAMES[] = {"v_fma_f64", "v_add_f64", "v_mul_f64", "v_rcp_f64", "v_sqrt_f64", "v_rsq_f64", "v_div_scale_f64", "v_div_fmas_f64",
                              "v_div_fixup_f64", "v_ldexp_f64", "v_max_f64", "v_cmp_lt_f64", "v_cvt_f64_i32", "v_mov_b32", "v_mov_b64",
                              "v_cndmask_b32", "v_add_u32", "v_and_b32", "v_fma_f32", "v_lshl_add_u64"};

template <int OP>
__device__ __forceinline__ void op(double& x, uint32_t& u, double a, double b)
{
  if (OP == FMA64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
  if (OP == ADD64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(b));
  if (OP == MUL64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(a));
  if (OP == RCP64) asm volatile("v_rcp_f64 %0, %0" : "+v"(x));
  if (OP == SQRT64) asm volatile("v_sqrt_f64 %0, %0" : "+v"(x));
  if (OP == RSQ64) asm volatile("v_rsq_f64 %0, %0" : "+v"(x));
  if (OP == DIVSCALE) asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(x) : "v"(a) : "vcc");
  if (OP == DIVFMAS) asm volatile("v_div_fmas_f64 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b) : "vcc");
  if (OP == DIVFIXUP) asm volatile("v_div_fixup_f64 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
  if (OP == LDEXP64) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(x) : "v"(u));
  if (OP == MAX64) asm volatile("v_max_f64 %0, %0, %1" : "+v"(x) : "v"(a));
  if (OP == CMP64) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(x), "v"(a) : "vcc");
  if (OP == CVT_I2D) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(x) : "v"(u));
  if (OP == MOV32) asm volatile("v_mov_b32 %0, %1" : "=v"(u) : "v"(u));
  if (OP == MOV64) asm volatile("v_mov_b64 %0, %1" : "=v"(x) : "v"(a));
  if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u) : "v"(u) : "vcc");
  if (OP == ADDU32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u) : "v"(u));
  if (OP == AND32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u) : "v"(u));
  if (OP == FMA32) {
    float f = __uint_as_float(u);
    asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f));
    u = __float_as_uint(f);
  }
  if (OP == LSHLADD64) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(x) : "v"(a));
}

template <int OP>
__global__ __launch_bounds__(256) void k(double* out, double a, double b, int iters)
{
  double x[8];
  uint32_t u[8];
  for (int j = 0; j < 8; j++) {
    x[j] = a + threadIdx.x * 1e-9 + j;
    u[j] = threadIdx.x + j;
  }
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
#pragma unroll
      for (int j = 0; j < 8; j++) op<OP>(x[j], u[j], a, b);
    }
  }
  double s = 0;
  for (int j = 0; j < 8; j++) s += x[j] + u[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(double* d, double ghz)
{
  const int iters = 1000;
  printf("%-16s", NAMES[OP]);
  for (int w = 1; w <= 4; w *= 2) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * w;  // 256 CUs; a 256-thread workgroup is one wave per SIMD
    hipLaunchKernelGGL((k<OP>), dim3(grid), dim3(256), 0, 0, d, 1.0000001, 1e-9, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP>), dim3(grid), dim3(256), 0, 0, d, 1.0000001, 1e-9, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters * 128 * w;  // wave instructions per SIMD
    printf("  %dw/SIMD %6.3f ns (%5.2f clk @%.2f GHz)", w, ms * 1e6 / n, ms * 1e6 / n * ghz, ghz);
  }
  printf("\n");
}

template <int OP>
void run_all(double* d, double ghz)
{
  run<OP>(d, ghz);
  if constexpr (OP + 1 < NOPS) run_all<OP + 1>(d, ghz);
}

int main()
{
  double* d;
  hipMalloc(&d, (size_t)(1 << 22) * 8);
  int khz = 0;
  hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
  printf("SIMD time per wave64 instruction (8 independent streams per wave); device clock attribute %.2f GHz\n", khz * 1e-6);
  run_all<0>(d, khz * 1e-6);
  return 0;
}
