// elmk_math.h - exp / log / pow / log10 that return the bits of the host libm the reference runs on.
//
// Why this exists.  The reference's physics calls <cmath>; on the platform it is built and tested on (x86-64 Linux, glibc)
// exp, log and pow are Szabolcs Nagy's table-driven routines (glibc >= 2.28: sysdeps/ieee754/dbl-64/e_exp.c, e_log.c,
// e_pow.c; the same code is published as ARM optimized-routines math/exp.c, log.c, pow.c).  They are accurate to about
// 0.51 ulp, i.e. NOT correctly rounded, and the device libm (ocml) rounds a few per cent of arguments the other way.  The
// leaf-temperature iteration of canopy_fluxes contains tolerance-terminated root finds, so one differing last bit can
// change an inner trip count and move an output by 1e-8 relative - far outside the north star's 1e-12.  Restating the
// host algorithm on the device removes the cause instead of widening the tolerance: same tables
// (elmk_math_tables.h), same operation sequence, same fused multiply-adds.
//
// The operation sequence - in particular WHICH a*b+c are fused - is the one of the x86-64 FMA build of glibc 2.35
// (__exp_fma / __log_fma / __pow_fma, selected at run time on every CPU with FMA + AVX2; read from the disassembly of the
// image's libm.so.6, Ubuntu GLIBC 2.35-0ubuntu3.11).  Every fused operation below is an explicit fma(); the file must be
// compiled with -ffp-contract=off so that nothing else is fused.  log10 is glibc's e_log10.c (a scaling around log()),
// which has no FMA build: plain multiplies and adds.
//
// The same header compiles for the host with gcc (tests/test_math_host.py compares it with the live libm on 10^8
// arguments, bit for bit) and for the device with hipcc (tests/test_gpu_parity.py::test_device_math_bits).
// No errno, no floating-point exception flags; results (including inf / nan / subnormal) are identical.
//
// Provenance and licence (see NOTICE at the repository root): the ALGORITHMS below are those of the GNU C Library 2.35 -
//   exp / log / pow: Szabolcs Nagy's routines (glibc e_exp.c, e_log.c, e_pow.c; also Arm optimized-routines, MIT);
//   atan / cos / acos: IBM Accurate Mathematical Library (glibc s_atan.c, s_sin.c, e_asin.c; LGPL-2.1-or-later);
//   tanh / expm1 / erf / log10: fdlibm (Copyright (C) 1993 by Sun Microsystems, Inc. - permission to use, copy, modify,
//   and distribute this software is freely granted, provided that this notice is preserved) -
// restated here from the published algorithms and the disassembly of the image's libm.so.6; each section names its source.
// Bit-identity is defined against THAT libm build (glibc 2.35, x86-64, FMA variants).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define ELMK_MFN __device__ __forceinline__
#define ELMK_MFN_NOINLINE __device__ __noinline__
#define ELMK_MATH_TABLE static __device__ const uint64_t
#else
#define ELMK_MFN static inline
#define ELMK_MFN_NOINLINE static
#define ELMK_MATH_TABLE static const uint64_t
#endif
#include "elmk_math_tables.h"

// Where the functions read their tables.  Default: the read-only arrays above (device global memory, cached in L1 / L2).
// A translation unit whose kernels evaluate these functions in their inner loops defines ELMK_MATH_LDS before including
// this header: the tables are then workgroup-local copies in LDS (a divergent 64-lane gather costs ~100 cycles there
// instead of a trip through the vector memory pipeline, which a kernel running one or two waves per SIMD cannot hide),
// and EVERY kernel of that unit that calls one of the functions must execute elmk_math_lds_init<>() first, with all
// threads of the workgroup, before any early return.
#if defined(__HIPCC__) && defined(ELMK_MATH_LDS)
static __shared__ uint64_t elmk_lds_exp_tab[256];
static __shared__ uint64_t elmk_lds_log_tab[256];
static __shared__ uint64_t elmk_lds_powlog_tab[384];
static __shared__ uint64_t elmk_lds_atan_tab[1687];
#define ELMK_T_EXP elmk_lds_exp_tab
#define ELMK_T_LOG elmk_lds_log_tab
#define ELMK_T_POWLOG elmk_lds_powlog_tab
#define ELMK_T_ATAN elmk_lds_atan_tab
// exp + log + pow tables (7 KB); WITH_ATAN adds the 13.5 KB atan table
template <bool WITH_ATAN>
__device__ __forceinline__ void elmk_math_lds_init()
{
  for (int i = threadIdx.x; i < 256; i += blockDim.x) {
    elmk_lds_exp_tab[i] = elmk_exp_tab[i];
    elmk_lds_log_tab[i] = elmk_log_tab[i];
  }
  for (int i = threadIdx.x; i < 384; i += blockDim.x) elmk_lds_powlog_tab[i] = elmk_powlog_tab[i];
  if (WITH_ATAN)
    for (int i = threadIdx.x; i < 1687; i += blockDim.x) elmk_lds_atan_tab[i] = elmk_atan_tab[i];
  __syncthreads();
}
#else
#define ELMK_T_EXP elmk_exp_tab
#define ELMK_T_LOG elmk_log_tab
#define ELMK_T_POWLOG elmk_powlog_tab
#define ELMK_T_ATAN elmk_atan_tab
#endif

ELMK_MFN uint64_t elmk_asu64(double x)
{
  union { double f; uint64_t i; } u;
  u.f = x;
  return u.i;
}
ELMK_MFN double elmk_asf64(uint64_t i)
{
  union { double f; uint64_t i; } u;
  u.i = i;
  return u.f;
}
#define ELMK_FMA(a, b, c) __builtin_fma((a), (b), (c))
#define ELMK_INF elmk_asf64(0x7ff0000000000000ull)
#define ELMK_NAN elmk_asf64(0x7ff8000000000000ull)

// ---- 2^(k/128)·(1 + tmp): shared tail of exp and pow (e_exp.c / e_pow.c "specialcase" included) ----------------------
// tmp = tail + r + r^2 (C2 + r C3) + r^4 (C4 + r C5), fused exactly as the FMA build does
ELMK_MFN double elmk_exp_poly(double r, double tail)
{
  const double C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3, C4 = 0x1.55555cf172b91p-5, C5 = 0x1.1111167a4d017p-7;
  const double p23 = ELMK_FMA(r, C3, C2);
  const double tr = tail + r;
  const double r2 = r * r;
  const double p45 = ELMK_FMA(r, C5, C4);
  const double lo = ELMK_FMA(p23, r2, tr);
  const double r4 = r2 * r2;
  return ELMK_FMA(r4, p45, lo);
}

// result would over/underflow the normal range: e_exp.c specialcase() / e_pow.c specialcase() (the latter handles a signed
// scale; with sbits >= 0 the two are the same function)
ELMK_MFN double elmk_exp_special(double tmp, uint64_t sbits, uint64_t ki)
{
  if ((ki & 0x80000000ull) == 0) {  // k > 0: the exponent of scale might have overflowed by <= 460
    sbits -= 1009ull << 52;
    const double scale = elmk_asf64(sbits);
    return 0x1p1009 * ELMK_FMA(scale, tmp, scale);
  }
  sbits += 1022ull << 52;  // k < 0: care in the subnormal range
  const double scale = elmk_asf64(sbits);
  const double st = scale * tmp;
  double y = scale + st;
  if (__builtin_fabs(y) < 1.0) {
    const double one = (y < 0.0) ? -1.0 : 1.0;
    double lo = scale - y + st;
    const double hi = one + y;
    lo = one - hi + y + lo;
    y = (hi + lo) - one;
    if (y == 0.0) y = elmk_asf64(sbits & 0x8000000000000000ull);
  }
  return 0x1p-1022 * y;
}

// exp(x + xtail) with the sign of the result flipped when sign_bias != 0 (e_pow.c exp_inline); exp(x) = (x, 0, 0) except
// for the two differences noted at elmk_exp
ELMK_MFN double elmk_exp_core(double x, double xtail, uint32_t sign_bias, int with_tail)
{
  const double InvLn2N = 0x1.71547652b82fep0 * 128, Shift = 0x1.8p52;
  const double NegLn2hiN = -0x1.62e42fefa0000p-8, NegLn2loN = -0x1.cf79abc9e3b3ap-47;
  uint32_t abstop = (uint32_t)(elmk_asu64(x) >> 52) & 0x7ff;
  if (abstop - 0x3c9u >= 0x3fu) {  // |x| < 2^-54 or |x| >= 512
    if (abstop - 0x3c9u >= 0x80000000u) {
      const double one = 1.0 + x;
      return sign_bias ? -one : one;
    }
    if (abstop >= 0x409u) {  // |x| >= 1024 (inf and nan are handled by the callers)
      if (elmk_asu64(x) >> 63) return sign_bias ? -0.0 : 0.0;  // __math_uflow
      return sign_bias ? -ELMK_INF : ELMK_INF;                 // __math_oflow
    }
    abstop = 0;  // large x: special-cased below
  }
  double kd = ELMK_FMA(x, InvLn2N, Shift);
  const uint64_t ki = elmk_asu64(kd);
  kd -= Shift;
  double r = ELMK_FMA(kd, NegLn2hiN, x);
  r = ELMK_FMA(kd, NegLn2loN, r);
  if (with_tail) r = xtail + r;
  const uint32_t idx = 2u * (uint32_t)(ki & 127u);
  const uint64_t top = (ki + sign_bias) << 45;
  const double tail = elmk_asf64(ELMK_T_EXP[idx]);
  const uint64_t sbits = ELMK_T_EXP[idx + 1] + top;
  const double tmp = elmk_exp_poly(r, tail);
  if (abstop == 0) return elmk_exp_special(tmp, sbits, ki);
  const double scale = elmk_asf64(sbits);
  return ELMK_FMA(scale, tmp, scale);
}

// ---- exp: glibc 2.35 sysdeps/ieee754/dbl-64/e_exp.c (__exp_fma) -------------------------------------------------------
// every case, in the source's control flow
ELMK_MFN double elmk_exp_general(double x)
{
  const uint64_t ix = elmk_asu64(x);
  const uint32_t abstop = (uint32_t)(ix >> 52) & 0x7ff;
  if (abstop >= 0x409u) {  // |x| >= 1024, inf, nan: the cases exp_inline leaves to its caller
    if (ix == 0xfff0000000000000ull) return 0.0;
    if (abstop >= 0x7ffu) return 1.0 + x;
    return (ix >> 63) ? 0.0 : ELMK_INF;
  }
  return elmk_exp_core(x, 0.0, 0, 0);
}
// The function the kernels call.  Same bits; different shape: the main path (2^-54 <= |x| < 512) is evaluated
// unconditionally as ONE straight-line block and everything else is a single, rarely taken branch afterwards.  The
// source's nest of range checks costs a wave that runs alone on its SIMD four exec-mask save / restore sequences per
// call and keeps the scheduler from overlapping the table read with the argument reduction: 197 -> ~90 ns per call in
// a dependent chain (tests/tools/ubench/math_issue.hip).
// main path as a branch-free block: the value for 2^-54 <= |x| < 512, `rare` set for every other argument
ELMK_MFN double elmk_exp_main(double x, int* rare)
{
  const double InvLn2N = 0x1.71547652b82fep0 * 128, Shift = 0x1.8p52;
  const double NegLn2hiN = -0x1.62e42fefa0000p-8, NegLn2loN = -0x1.cf79abc9e3b3ap-47;
  const uint32_t abstop = (uint32_t)(elmk_asu64(x) >> 52) & 0x7ff;
  double kd = ELMK_FMA(x, InvLn2N, Shift);
  const uint64_t ki = elmk_asu64(kd);
  kd -= Shift;
  double r = ELMK_FMA(kd, NegLn2hiN, x);
  r = ELMK_FMA(kd, NegLn2loN, r);
  const uint32_t idx = 2u * (uint32_t)(ki & 127u);
  const double tail = elmk_asf64(ELMK_T_EXP[idx]);
  const uint64_t sbits = ELMK_T_EXP[idx + 1] + (ki << 45);
  const double tmp = elmk_exp_poly(r, tail);
  const double scale = elmk_asf64(sbits);
  *rare = (abstop - 0x3c9u >= 0x3fu);
  return ELMK_FMA(scale, tmp, scale);
}
ELMK_MFN double elmk_exp(double x)
{
  int rare;
  double y = elmk_exp_main(x, &rare);
  if (__builtin_expect(rare, 0)) y = elmk_exp_general(x);
  return y;
}

// ---- log: glibc 2.35 sysdeps/ieee754/dbl-64/e_log.c (__log_fma) -------------------------------------------------------
// log for 1 - 2^-4 <= x < 1 + 0x1.09p-4 (e_log.c's "close to 1.0" polynomial), straight-line; x == 1 gives +0 as the
// source's explicit test does
ELMK_MFN double elmk_log_near1(double x)
{
  const double B0 = -0x1p-1, B1 = 0x1.5555555555577p-2, B2 = -0x1.ffffffffffdcbp-3, B3 = 0x1.999999995dd0cp-3,
               B4 = -0x1.55555556745a7p-3, B5 = 0x1.24924a344de3p-3, B6 = -0x1.fffffa4423d65p-4,
               B7 = 0x1.c7184282ad6cap-4, B8 = -0x1.999eb43b068ffp-4, B9 = 0x1.78182f7afd085p-4,
               B10 = -0x1.5521375d145cdp-4;
  const double r = x - 1.0;
  double p1 = ELMK_FMA(r, B2, B1);
  double p4 = ELMK_FMA(r, B5, B4);
  const double r2 = r * r;
  double p7 = ELMK_FMA(r, B8, B7);
  p1 = ELMK_FMA(r2, B3, p1);
  p4 = ELMK_FMA(r2, B6, p4);
  const double r3 = r * r2;
  p7 = ELMK_FMA(r2, B9, p7);
  p7 = ELMK_FMA(r3, B10, p7);
  p4 = ELMK_FMA(p7, r3, p4);
  p1 = ELMK_FMA(p4, r3, p1);
  const double t = ELMK_FMA(r, 0x1p27, r);
  const double rhi = ELMK_FMA(-0x1p27, r, t);
  const double rhi2 = rhi * rhi;
  const double rlo = r - rhi;
  const double hi = ELMK_FMA(rhi2, B0, r);
  const double t8 = r - hi;
  const double s = r + rhi;
  double lo = ELMK_FMA(rhi2, B0, t8);
  const double u = B0 * rlo;
  lo = ELMK_FMA(u, s, lo);
  const double y = ELMK_FMA(p1, r3, lo);
  return hi + y;
}

ELMK_MFN double elmk_log_general(double x)
{
  uint64_t ix = elmk_asu64(x);
  const uint32_t top = (uint32_t)(ix >> 48);
  if (ix - 0x3fee000000000000ull < 0x3090000000000ull) {  // 1 - 2^-4 <= x < 1 + 0x1.09p-4
    if (ix == 0x3ff0000000000000ull) return 0.0;
    return elmk_log_near1(x);
  }
  if (top - 0x0010u >= 0x7ff0u - 0x0010u) {  // x < 2^-1022, inf or nan
    if (ix * 2 == 0) return -ELMK_INF;
    if (ix == 0x7ff0000000000000ull) return x;
    if ((top & 0x8000u) || (top & 0x7ff0u) == 0x7ff0u) return (x != x) ? x + x : ELMK_NAN;  // __math_invalid
    ix = elmk_asu64(x * 0x1p52);  // subnormal: normalise
    ix -= 52ull << 52;
  }
  const double Ln2hi = 0x1.62e42fefa3800p-1, Ln2lo = 0x1.ef35793c76730p-45;
  const double A0 = -0x1.0000000000001p-1, A1 = 0x1.555555551305bp-2, A2 = -0x1.fffffffeb459p-3, A3 = 0x1.999b324f10111p-3,
               A4 = -0x1.55575e506c89fp-3;
  const uint64_t tmp = ix - 0x3fe6000000000000ull;
  const uint32_t i = (uint32_t)(tmp >> 45) & 127u;
  const int k = (int32_t)(uint32_t)(tmp >> 32) >> 20;  // (int64_t)tmp >> 52, from the high word
  const uint64_t iz = ix - (tmp & 0xfffull << 52);
  const double invc = elmk_asf64(ELMK_T_LOG[2 * i]), logc = elmk_asf64(ELMK_T_LOG[2 * i + 1]);
  const double z = elmk_asf64(iz);
  const double kd = (double)k;
  const double r = ELMK_FMA(z, invc, -1.0);
  const double w = ELMK_FMA(kd, Ln2hi, logc);
  const double p12 = ELMK_FMA(r, A2, A1);
  const double hi = w + r;
  const double r2 = r * r;
  double lo = w - hi;
  lo = lo + r;
  lo = ELMK_FMA(kd, Ln2lo, lo);
  const double r3 = r * r2;
  double p34 = ELMK_FMA(r, A4, A3);
  lo = ELMK_FMA(r2, A0, lo);
  p34 = ELMK_FMA(p34, r2, p12);
  const double y = ELMK_FMA(r3, p34, lo);
  return y + hi;
}

// The function the kernels call: same bits; the table path (positive normal x outside [1 - 2^-4, 1 + 0x1.09p-4)) is one
// straight-line block with a single rarely taken branch behind it for zero / negative / subnormal / inf / nan; the
// near-1 polynomial stays a branch of its own (evaluating both and selecting was measured: 191 -> 236 ns per call).
// table path as a branch-free block: the value for positive normal x outside [1 - 2^-4, 1 + 0x1.09p-4); `near1` set inside
// that interval (its own polynomial, elmk_log_near1), `rare` set for zero / negative / subnormal / inf / nan
ELMK_MFN double elmk_log_main(double x, int* near1, int* rare)
{
  const uint64_t ix = elmk_asu64(x);
  const double Ln2hi = 0x1.62e42fefa3800p-1, Ln2lo = 0x1.ef35793c76730p-45;
  const double A0 = -0x1.0000000000001p-1, A1 = 0x1.555555551305bp-2, A2 = -0x1.fffffffeb459p-3, A3 = 0x1.999b324f10111p-3,
               A4 = -0x1.55575e506c89fp-3;
  const uint32_t top = (uint32_t)(ix >> 48);
  const uint64_t tmp = ix - 0x3fe6000000000000ull;
  const uint32_t i = (uint32_t)(tmp >> 45) & 127u;
  const int k = (int32_t)(uint32_t)(tmp >> 32) >> 20;  // (int64_t)tmp >> 52, from the high word
  const uint64_t iz = ix - (tmp & 0xfffull << 52);
  const double invc = elmk_asf64(ELMK_T_LOG[2 * i]), logc = elmk_asf64(ELMK_T_LOG[2 * i + 1]);
  const double z = elmk_asf64(iz);
  const double kd = (double)k;
  const double r = ELMK_FMA(z, invc, -1.0);
  const double w = ELMK_FMA(kd, Ln2hi, logc);
  const double p12 = ELMK_FMA(r, A2, A1);
  const double hi = w + r;
  const double r2 = r * r;
  double lo = w - hi;
  lo = lo + r;
  lo = ELMK_FMA(kd, Ln2lo, lo);
  const double r3 = r * r2;
  double p34 = ELMK_FMA(r, A4, A3);
  lo = ELMK_FMA(r2, A0, lo);
  p34 = ELMK_FMA(p34, r2, p12);
  *near1 = (ix - 0x3fee000000000000ull < 0x3090000000000ull);
  *rare = (top - 0x0010u >= 0x7ff0u - 0x0010u);
  return ELMK_FMA(r3, p34, lo) + hi;
}
ELMK_MFN double elmk_log(double x)
{
  const uint64_t ix = elmk_asu64(x);
  if (ix - 0x3fee000000000000ull < 0x3090000000000ull) return elmk_log_general(x);  // near 1: its own polynomial
  int near1, rare;
  double y = elmk_log_main(x, &near1, &rare);
  if (__builtin_expect(rare, 0)) y = elmk_log_general(x);
  return y;
}

// ---- log10: glibc 2.35 sysdeps/ieee754/dbl-64/e_log10.c (fdlibm scaling around __ieee754_log; no FMA build) ------------
ELMK_MFN double elmk_log10(double x)
{
  const double two54 = 1.80143985094819840000e+16, ivln10 = 4.34294481903251816668e-01,
               log10_2hi = 3.01029995663611771306e-01, log10_2lo = 3.69423907715893078616e-13;
  uint64_t ix = elmk_asu64(x);
  int32_t hx = (int32_t)(ix >> 32);
  const uint32_t lx = (uint32_t)ix;
  int32_t k = 0;
  if (hx < 0x00100000) {  // x < 2^-1022
    if (((hx & 0x7fffffff) | lx) == 0) return -ELMK_INF;
    if (hx < 0) return (x != x) ? x + x : ELMK_NAN;
    k -= 54;
    x *= two54;
    ix = elmk_asu64(x);
    hx = (int32_t)(ix >> 32);
  }
  if (hx >= 0x7ff00000) return x + x;
  k += (hx >> 20) - 1023;
  const int32_t i = (int32_t)(((uint32_t)k & 0x80000000u) >> 31);
  hx = (hx & 0x000fffff) | ((0x3ff - i) << 20);
  const double y = (double)(k + i);
  x = elmk_asf64(((uint64_t)(uint32_t)hx << 32) | (elmk_asu64(x) & 0xffffffffull));
  const double z = y * log10_2lo + ivln10 * elmk_log(x);
  return z + y * log10_2hi;
}

// ---- pow: glibc 2.35 sysdeps/ieee754/dbl-64/e_pow.c (__pow_fma) -------------------------------------------------------
// 0: y is not an integer, 1: odd integer, 2: even integer
ELMK_MFN int elmk_checkint(uint64_t iy)
{
  const int e = (int)(iy >> 52) & 0x7ff;
  if (e < 0x3ff) return 0;
  if (e > 0x3ff + 52) return 2;
  if (iy & ((1ull << (0x3ff + 52 - e)) - 1)) return 0;
  if (iy & (1ull << (0x3ff + 52 - e))) return 1;
  return 2;
}
ELMK_MFN int elmk_zeroinfnan(uint64_t i) { return 2 * i - 1 >= 2 * 0x7ff0000000000000ull - 1; }

ELMK_MFN double elmk_pow_general(double x, double y)
{
  uint32_t sign_bias = 0;
  uint64_t ix = elmk_asu64(x);
  const uint64_t iy = elmk_asu64(y);
  uint32_t topx = (uint32_t)(ix >> 52);
  const uint32_t topy = (uint32_t)(iy >> 52);
  if (topx - 0x001u >= 0x7ffu - 0x001u || (topy & 0x7ffu) - 0x3beu >= 0x43eu - 0x3beu) {
    // x < 2^-1022, inf or nan; or |y| < 2^-65, |y| >= 2^63 or nan
    if (elmk_zeroinfnan(iy)) {
      if (2 * iy == 0) return 1.0;
      if (ix == 0x3ff0000000000000ull) return 1.0;
      if (2 * ix > 2 * 0x7ff0000000000000ull || 2 * iy > 2 * 0x7ff0000000000000ull) return x + y;
      if (2 * ix == 2 * 0x3ff0000000000000ull) return 1.0;
      if ((2 * ix < 2 * 0x3ff0000000000000ull) == !(iy >> 63)) return 0.0;  // |x|<1 && y==inf or |x|>1 && y==-inf
      return y * y;
    }
    if (elmk_zeroinfnan(ix)) {
      double x2 = x * x;
      if ((ix >> 63) && elmk_checkint(iy) == 1) x2 = -x2;
      return (iy >> 63) ? 1.0 / x2 : x2;
    }
    if (ix >> 63) {  // finite x < 0
      const int yint = elmk_checkint(iy);
      if (yint == 0) return ELMK_NAN;  // __math_invalid
      if (yint == 1) sign_bias = 0x800u << 7;
      ix &= 0x7fffffffffffffffull;
      topx &= 0x7ffu;
    }
    if ((topy & 0x7ffu) - 0x3beu >= 0x43eu - 0x3beu) {
      if (ix == 0x3ff0000000000000ull) return 1.0;
      if ((topy & 0x7ffu) < 0x3beu) return ix > 0x3ff0000000000000ull ? 1.0 + y : 1.0 - y;  // |y| < 2^-65
      return ((ix > 0x3ff0000000000000ull) == (topy < 0x800u)) ? ELMK_INF : 0.0;
    }
    if (topx == 0) {  // subnormal x: normalise
      ix = elmk_asu64(x * 0x1p52);
      ix &= 0x7fffffffffffffffull;
      ix -= 52ull << 52;
    }
  }
  // log_inline: log(x) = hi + lo to ~68 bits
  const double Ln2hi = 0x1.62e42fefa3800p-1, Ln2lo = 0x1.ef35793c76730p-45;
  const double A0 = -0x1p-1, A1 = -0x1.555555555556p-1, A2 = 0x1.0000000000006p-1, A3 = 0x1.999999959554ep-1,
               A4 = -0x1.555555529a47ap-1, A5 = -0x1.2495b9b4845e9p0, A6 = 0x1.0002b8b263fc3p0;
  const uint64_t tmp = ix - 0x3fe6955500000000ull;
  const uint32_t i = (uint32_t)(tmp >> 45) & 127u;
  const int k = (int32_t)(uint32_t)(tmp >> 32) >> 20;  // (int64_t)tmp >> 52, from the high word
  const uint64_t iz = ix - (tmp & 0xfffull << 52);
  const double z = elmk_asf64(iz);
  const double kd = (double)k;
  const double invc = elmk_asf64(ELMK_T_POWLOG[3 * i]), logc = elmk_asf64(ELMK_T_POWLOG[3 * i + 1]),
               logctail = elmk_asf64(ELMK_T_POWLOG[3 * i + 2]);
  const double r = ELMK_FMA(z, invc, -1.0);
  const double t1 = ELMK_FMA(kd, Ln2hi, logc);
  const double t2 = t1 + r;
  const double lo1 = ELMK_FMA(kd, Ln2lo, logctail);
  const double lo2 = t1 - t2 + r;
  const double ar = A0 * r;
  const double ar2 = r * ar;
  const double ar3 = r * ar2;
  const double hi = t2 + ar2;
  const double lo3 = ELMK_FMA(ar, r, -ar2);
  const double lo4 = t2 - hi + ar2;
  const double p12 = ELMK_FMA(r, A2, A1);
  const double p34 = ELMK_FMA(r, A4, A3);
  double p = ELMK_FMA(r, A6, A5);
  p = ELMK_FMA(p, ar2, p34);
  p = ELMK_FMA(ar2, p, p12);
  double lo = lo1 + lo2;
  lo = lo + lo3;
  lo = lo + lo4;
  lo = ELMK_FMA(p, ar3, lo);
  const double lhi = hi + lo;
  const double llo = hi - lhi + lo;
  // y·log(x) = ehi + elo
  const double ehi = y * lhi;
  double elo = ELMK_FMA(lhi, y, -ehi);
  elo = ELMK_FMA(y, llo, elo);
  return elmk_exp_core(ehi, elo, sign_bias, 1);
}

// The function the kernels call: same bits; for positive normal x, 2^-65 <= |y| < 2^63 and 2^-54 <= |y log x| < 512 -
// everything the physics does - log_inline and exp_inline run as one straight-line block; any other argument takes the
// single branch to the general form afterwards.
// main path as a branch-free block; `rare` set for every argument pair outside it
// (a macro so that log_inline's table can be named at compile time: LDS or global, see elmk_pow_literal_base)
#define ELMK_POW_MAIN_BODY(powlog_tab) \
  const uint64_t ix = elmk_asu64(x); \
  const uint32_t topx = (uint32_t)(ix >> 52); \
  const uint32_t topy = (uint32_t)(elmk_asu64(y) >> 52); \
  const int rare_arg = (topx - 0x001u >= 0x7ffu - 0x001u) | ((topy & 0x7ffu) - 0x3beu >= 0x43eu - 0x3beu); \
  const double Ln2hi = 0x1.62e42fefa3800p-1, Ln2lo = 0x1.ef35793c76730p-45; \
  const double A0 = -0x1p-1, A1 = -0x1.555555555556p-1, A2 = 0x1.0000000000006p-1, A3 = 0x1.999999959554ep-1, \
               A4 = -0x1.555555529a47ap-1, A5 = -0x1.2495b9b4845e9p0, A6 = 0x1.0002b8b263fc3p0; \
  const uint64_t tmp = ix - 0x3fe6955500000000ull; \
  const uint32_t i = (uint32_t)(tmp >> 45) & 127u; \
  const int k = (int32_t)(uint32_t)(tmp >> 32) >> 20; \
  const uint64_t iz = ix - (tmp & 0xfffull << 52); \
  const double z = elmk_asf64(iz); \
  const double kd = (double)k; \
  const double invc = elmk_asf64(powlog_tab[3 * i]), logc = elmk_asf64(powlog_tab[3 * i + 1]), \
               logctail = elmk_asf64(powlog_tab[3 * i + 2]); \
  const double r = ELMK_FMA(z, invc, -1.0); \
  const double t1 = ELMK_FMA(kd, Ln2hi, logc); \
  const double t2 = t1 + r; \
  const double lo1 = ELMK_FMA(kd, Ln2lo, logctail); \
  const double lo2 = t1 - t2 + r; \
  const double ar = A0 * r; \
  const double ar2 = r * ar; \
  const double ar3 = r * ar2; \
  const double hi = t2 + ar2; \
  const double lo3 = ELMK_FMA(ar, r, -ar2); \
  const double lo4 = t2 - hi + ar2; \
  const double p12 = ELMK_FMA(r, A2, A1); \
  const double p34 = ELMK_FMA(r, A4, A3); \
  double p = ELMK_FMA(r, A6, A5); \
  p = ELMK_FMA(p, ar2, p34); \
  p = ELMK_FMA(ar2, p, p12); \
  double lo = lo1 + lo2; \
  lo = lo + lo3; \
  lo = lo + lo4; \
  lo = ELMK_FMA(p, ar3, lo); \
  const double lhi = hi + lo; \
  const double llo = hi - lhi + lo; \
  const double ehi = y * lhi; \
  double elo = ELMK_FMA(lhi, y, -ehi); \
  elo = ELMK_FMA(y, llo, elo); \
  /* exp_inline, main path */ \
  const double InvLn2N = 0x1.71547652b82fep0 * 128, Shift = 0x1.8p52; \
  const double NegLn2hiN = -0x1.62e42fefa0000p-8, NegLn2loN = -0x1.cf79abc9e3b3ap-47; \
  const uint32_t abstop = (uint32_t)(elmk_asu64(ehi) >> 52) & 0x7ff; \
  double ekd = ELMK_FMA(ehi, InvLn2N, Shift); \
  const uint64_t ki = elmk_asu64(ekd); \
  ekd -= Shift; \
  double er = ELMK_FMA(ekd, NegLn2hiN, ehi); \
  er = ELMK_FMA(ekd, NegLn2loN, er); \
  er = elo + er; \
  const uint32_t idx = 2u * (uint32_t)(ki & 127u); \
  const double tail = elmk_asf64(ELMK_T_EXP[idx]); \
  const uint64_t sbits = ELMK_T_EXP[idx + 1] + (ki << 45); \
  const double etmp = elmk_exp_poly(er, tail); \
  const double scale = elmk_asf64(sbits); \
  *rare = rare_arg | (abstop - 0x3c9u >= 0x3fu); \
  return ELMK_FMA(scale, etmp, scale);
ELMK_MFN double elmk_pow_main(double x, double y, int* rare)
{
  ELMK_POW_MAIN_BODY(ELMK_T_POWLOG)
}
// the same block reading log_inline's table from the read-only array in global memory whatever ELMK_MATH_LDS says (see
// elmk_pow_literal_base)
ELMK_MFN double elmk_pow_main_gtab(double x, double y, int* rare)
{
  ELMK_POW_MAIN_BODY(elmk_powlog_tab)
}
ELMK_MFN double elmk_pow(double x, double y)
{
  int rare;
  double res = elmk_pow_main(x, y, &rare);
  if (__builtin_expect(rare, 0)) res = elmk_pow_general(x, y);
  return res;
}
// pow with a base that is a literal in the source: same bits.  log_inline reads the read-only table in global memory whatever
// ELMK_MATH_LDS says, so that the compiler evaluates the whole logarithm (table row, polynomial, hi + lo split) at compile time
// and only y log x and exp_inline are left to run.
ELMK_MFN double elmk_pow_literal_base(const double x, double y)
{
  int rare;
  double res = elmk_pow_main_gtab(x, y, &rare);
  if (__builtin_expect(rare, 0)) res = elmk_pow_general(x, y);
  return res;
}

// ---- atan: glibc 2.35 sysdeps/ieee754/dbl-64/s_atan.c (__atan_fma; IBM Accurate Mathematical Library, the version with
// the multi-precision fall-backs removed) --------------------------------------------------------------------------------
ELMK_MFN double elmk_atan_signed(double y, double x)  // |y| with the sign of x
{
  return elmk_asf64((elmk_asu64(y) & 0x7fffffffffffffffull) | (elmk_asu64(x) & 0x8000000000000000ull));
}
ELMK_MFN double elmk_atan_general(double x)
{
  const double d3 = -0x1.5555555555555p-2, d5 = 0x1.99999999997fdp-3, d7 = -0x1.24924923f7603p-3, d9 = 0x1.c71c6e5129a3bp-4,
               d11 = -0x1.7458022b13c25p-4, d13 = 0x1.375f08b31cbcep-4;
  const double HPI = 0x1.921fb54442d18p+0, HPI1 = 0x1.1a62633145c07p-54;
  const uint64_t ix = elmk_asu64(x);
  if ((ix & 0x7ff0000000000000ull) == 0x7ff0000000000000ull && (ix & 0x000fffffffffffffull) != 0) return x + x;  // nan
  const double u = (x < 0.0) ? -x : x;
  if (u < 1.0) {
    if (u < 0x1p-4) {
      if (u < 0x1.bb67ap-27) return x;
      const double v = x * x;
      double p = d13;
      p = ELMK_FMA(v, p, d11);
      p = ELMK_FMA(v, p, d9);
      p = ELMK_FMA(v, p, d7);
      p = ELMK_FMA(v, p, d5);
      p = ELMK_FMA(v, p, d3);
      return ELMK_FMA(x * v, p, x);
    }
    const int i = (int)(ELMK_FMA(u, 256.0, 0x1p52) - 0x1p52) - 16;
    const uint64_t* c = ELMK_T_ATAN + 7 * i;
    const double z = u - elmk_asf64(c[0]);
    double p = elmk_asf64(c[6]);
    p = ELMK_FMA(z, p, elmk_asf64(c[5]));
    p = ELMK_FMA(z, p, elmk_asf64(c[4]));
    p = ELMK_FMA(z, p, elmk_asf64(c[3]));
    p = ELMK_FMA(z, p, elmk_asf64(c[2]));
    return elmk_atan_signed(ELMK_FMA(p, z, elmk_asf64(c[1])), x);
  }
  if (u < 16.0) {
    const double w = 1.0 / u;
    const double t1 = u * w;
    const double a = 1.0 - t1;
    const double t2 = ELMK_FMA(u, w, -t1);
    const int i = (int)(ELMK_FMA(w, 256.0, 0x1p52) - 0x1p52) - 16;
    const uint64_t* c = ELMK_T_ATAN + 7 * i;
    const double t3 = a - t2;
    const double zz = w - elmk_asf64(c[0]);
    const double z = ELMK_FMA(t3, w, zz);
    double p = elmk_asf64(c[6]);
    p = ELMK_FMA(z, p, elmk_asf64(c[5]));
    p = ELMK_FMA(z, p, elmk_asf64(c[4]));
    p = ELMK_FMA(z, p, elmk_asf64(c[3]));
    p = ELMK_FMA(z, p, elmk_asf64(c[2]));
    const double yy = ELMK_FMA(-z, p, HPI1);
    const double t = HPI - elmk_asf64(c[1]);
    return elmk_atan_signed(t + yy, x);
  }
  if (u < 0x1.49ff2p+52) {
    const double w = 1.0 / u;
    const double v = w * w;
    const double t1 = u * w;
    double p = d13;
    p = ELMK_FMA(v, p, d11);
    p = ELMK_FMA(v, p, d9);
    p = ELMK_FMA(v, p, d7);
    p = ELMK_FMA(v, p, d5);
    p = ELMK_FMA(v, p, d3);
    const double wv = w * v;
    const double t2 = ELMK_FMA(u, w, -t1);
    double a = 1.0 - t1;
    const double yy = wv * p;
    a = a - t2;
    const double t3 = HPI - w;
    const double ww = a * w;
    const double cor = (HPI - t3) - w;
    double r = cor + HPI1;
    r = r - ww;
    r = r - yy;
    return elmk_atan_signed(r + t3, x);
  }
  return (x > 0.0) ? HPI : -HPI;
}

// ---- pow(x, 2.0) and pow(x, 1.0) as the reference's COMPILER evaluates them ---------------------------------------------
// GCC replaces pow(x, 2.0) / std::pow(x, 2) by x * x at -O1 and above and pow(x, 1.0) by x at every level, without
// -ffast-math (checked with the image's gcc 11.4: `mulsd %xmm0, %xmm0`, no call).  The oracle and oracle/_ref (the
// reference's own headers) are -O2 builds, so that is what "the reference's result" is at these call sites; glibc's
// pow(x, 2.0) itself rounds the other way for a small fraction of arguments (it is a 0.52 ulp routine), which is what the
// reference's default Debug (-O0) build would return.
ELMK_MFN double elmk_sq(double x) { return x * x; }
ELMK_MFN double elmk_pow1(double x) { return x; }

// The function the kernels call: same bits; the 1 <= |x| < 16 range - all the physics asks for (stability function of the
// unstable surface layer, chi = (1 - 16 zeta)^(1/4) with -100 <= zeta < 0) - as one straight-line block, any other argument
// through the single branch to the general form.
// main path (1 <= |x| < 16) as a branch-free block; `rare` set for every other argument
ELMK_MFN double elmk_atan_main(double x, int* rare)
{
  const double HPI = 0x1.921fb54442d18p+0, HPI1 = 0x1.1a62633145c07p-54;
  const double u = __builtin_fabs(x);
  const double w = 1.0 / u;
  const double t1 = u * w;
  const double a = 1.0 - t1;
  const double t2 = ELMK_FMA(u, w, -t1);
  int i = (int)(ELMK_FMA(w, 256.0, 0x1p52) - 0x1p52) - 16;
  i = (i < 0) ? 0 : ((i > 240) ? 240 : i);  // only ever out of range for arguments that take the general form below
  const uint64_t* c = ELMK_T_ATAN + 7 * i;
  const double t3 = a - t2;
  const double zz = w - elmk_asf64(c[0]);
  const double z = ELMK_FMA(t3, w, zz);
  double p = elmk_asf64(c[6]);
  p = ELMK_FMA(z, p, elmk_asf64(c[5]));
  p = ELMK_FMA(z, p, elmk_asf64(c[4]));
  p = ELMK_FMA(z, p, elmk_asf64(c[3]));
  p = ELMK_FMA(z, p, elmk_asf64(c[2]));
  const double yy = ELMK_FMA(-z, p, HPI1);
  const double t = HPI - elmk_asf64(c[1]);
  *rare = !(u >= 1.0 && u < 16.0);
  return elmk_atan_signed(t + yy, x);
}
ELMK_MFN double elmk_atan(double x)
{
  int rare;
  double y = elmk_atan_main(x, &rare);
  if (__builtin_expect(rare, 0)) y = elmk_atan_general(x);
  return y;
}

#if defined(__cplusplus)
// ---- batched forms -----------------------------------------------------------------------------------------------------
// N independent arguments evaluated in ONE basic block (main paths of all of them, then a single rarely taken branch for
// whichever of them needs the general form).  A kernel that runs one wave per SIMD executes a dependent fp64 chain at half
// the issue rate of the pipe (8 cycles of latency, 4 of issue); N >= 2 chains side by side fill it.  Same functions, same
// bits as N calls of the scalar forms.
template <int N>
ELMK_MFN void elmk_exp_n(double (&v)[N])
{
  double y[N];
  int rare[N], any = 0;
#pragma unroll
  for (int i = 0; i < N; i++) y[i] = elmk_exp_main(v[i], &rare[i]);
#pragma unroll
  for (int i = 0; i < N; i++) any |= rare[i];
  if (__builtin_expect(any, 0)) {
#pragma unroll
    for (int i = 0; i < N; i++)
      if (rare[i]) y[i] = elmk_exp_general(v[i]);
  }
#pragma unroll
  for (int i = 0; i < N; i++) v[i] = y[i];
}
template <int N>
ELMK_MFN void elmk_log_n(double (&v)[N])
{
  double y[N];
  int near1[N], rare[N], any_near = 0, any_rare = 0;
#pragma unroll
  for (int i = 0; i < N; i++) y[i] = elmk_log_main(v[i], &near1[i], &rare[i]);
#pragma unroll
  for (int i = 0; i < N; i++) {
    any_near |= near1[i];
    any_rare |= rare[i];
  }
  if (any_near) {  // common for the stability functions of a near-neutral surface layer: all N side by side as well
    double n1[N];
#pragma unroll
    for (int i = 0; i < N; i++) n1[i] = elmk_log_near1(v[i]);
#pragma unroll
    for (int i = 0; i < N; i++)
      if (near1[i]) y[i] = (v[i] == 1.0) ? 0.0 : n1[i];
  }
  if (__builtin_expect(any_rare, 0)) {
#pragma unroll
    for (int i = 0; i < N; i++)
      if (rare[i] && !near1[i]) y[i] = elmk_log_general(v[i]);
  }
#pragma unroll
  for (int i = 0; i < N; i++) v[i] = y[i];
}
// v[i] = pow(v[i], e[i])
template <int N>
ELMK_MFN void elmk_pow_n(double (&v)[N], const double (&e)[N])
{
  double y[N];
  int rare[N], any = 0;
#pragma unroll
  for (int i = 0; i < N; i++) y[i] = elmk_pow_main(v[i], e[i], &rare[i]);
#pragma unroll
  for (int i = 0; i < N; i++) any |= rare[i];
  if (__builtin_expect(any, 0)) {
#pragma unroll
    for (int i = 0; i < N; i++)
      if (rare[i]) y[i] = elmk_pow_general(v[i], e[i]);
  }
#pragma unroll
  for (int i = 0; i < N; i++) v[i] = y[i];
}
template <int N>
ELMK_MFN void elmk_atan_n(double (&v)[N])
{
  double y[N];
  int rare[N], any = 0;
#pragma unroll
  for (int i = 0; i < N; i++) y[i] = elmk_atan_main(v[i], &rare[i]);
#pragma unroll
  for (int i = 0; i < N; i++) any |= rare[i];
  if (__builtin_expect(any, 0)) {
#pragma unroll
    for (int i = 0; i < N; i++)
      if (rare[i]) y[i] = elmk_atan_general(v[i]);
  }
#pragma unroll
  for (int i = 0; i < N; i++) v[i] = y[i];
}
#endif  // __cplusplus


// ---- expm1, tanh: glibc 2.35 sysdeps/ieee754/dbl-64/s_expm1.c, s_tanh.c (fdlibm; these have no FMA build on x86-64:
// plain multiplies and adds in the source's order) ------------------------------------------------------------------------
ELMK_MFN double elmk_with_high_word(double y, uint32_t hi)
{
  return elmk_asf64(((uint64_t)hi << 32) | (elmk_asu64(y) & 0xffffffffull));
}
ELMK_MFN double elmk_expm1(double x)
{
  const double one = 1.0, huge = 1.0e+300, tiny = 1.0e-300, o_threshold = 7.09782712893383973096e+02,
               ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10, invln2 = 1.44269504088896338700e+00;
  const double Q1 = -3.33333333333331316428e-02, Q2 = 1.58730158725481460165e-03, Q3 = -7.93650757867487942473e-05,
               Q4 = 4.00821782732936239552e-06, Q5 = -2.01099218183624371326e-07;
  double y, hi, lo, c = 0.0, t, e, hxs, hfx, r1, h2, h4, R1, R2, R3;
  int32_t k;
  const uint32_t hx0 = (uint32_t)(elmk_asu64(x) >> 32);
  const uint32_t xsb = hx0 & 0x80000000u;
  const uint32_t hx = hx0 & 0x7fffffffu;
  if (hx >= 0x4043687Au) {    // |x| >= 56 ln2
    if (hx >= 0x40862E42u) {  // |x| >= 709.78
      if (hx >= 0x7ff00000u) {
        if (((hx & 0xfffffu) | (uint32_t)elmk_asu64(x)) != 0) return x + x;  // nan
        return (xsb == 0) ? x : -1.0;                                       // exp(+-inf) - 1
      }
      if (x > o_threshold) return huge * huge;  // overflow
    }
    if (xsb != 0) return tiny - one;  // x < -56 ln2: -1 (the reference adds an inexact flag)
  }
  if (hx > 0x3fd62e42u) {    // |x| > 0.5 ln2
    if (hx < 0x3FF0A2B2u) {  // and |x| < 1.5 ln2
      if (xsb == 0) {
        hi = x - ln2_hi;
        lo = ln2_lo;
        k = 1;
      } else {
        hi = x + ln2_hi;
        lo = -ln2_lo;
        k = -1;
      }
    } else {
      k = (int32_t)(invln2 * x + ((xsb == 0) ? 0.5 : -0.5));
      t = (double)k;
      hi = x - t * ln2_hi;  // t * ln2_hi is exact here
      lo = t * ln2_lo;
    }
    x = hi - lo;
    c = (hi - x) - lo;
  } else if (hx < 0x3c900000u) {  // |x| < 2^-54
    t = huge + x;
    return x - (t - (huge + x));
  } else {
    k = 0;
  }
  // x is now in the primary range
  hfx = 0.5 * x;
  hxs = x * hfx;
  R1 = one + hxs * Q1;
  h2 = hxs * hxs;
  R2 = Q2 + hxs * Q3;
  h4 = h2 * h2;
  R3 = Q4 + hxs * Q5;
  r1 = R1 + h2 * R2 + h4 * R3;
  t = 3.0 - r1 * hfx;
  e = hxs * ((r1 - t) / (6.0 - x * t));
  if (k == 0) return x - (x * e - hxs);  // c is 0
  e = (x * (e - c) - c);
  e -= hxs;
  if (k == -1) return 0.5 * (x - e) - 0.5;
  if (k == 1) {
    if (x < -0.25) return -2.0 * (e - (x + 0.5));
    return one + 2.0 * (x - e);
  }
  if (k <= -2 || k > 56) {  // suffice to return exp(x) - 1
    y = one - (e - x);
    y = elmk_with_high_word(y, (uint32_t)(elmk_asu64(y) >> 32) + ((uint32_t)k << 20));
    return y - one;
  }
  if (k < 20) {
    t = elmk_with_high_word(one, 0x3ff00000u - (0x200000u >> k));  // 1 - 2^-k
    y = t - (e - x);
    y = elmk_with_high_word(y, (uint32_t)(elmk_asu64(y) >> 32) + ((uint32_t)k << 20));
  } else {
    t = elmk_with_high_word(one, (uint32_t)(0x3ff - k) << 20);  // 2^-k
    y = x - (e + t);
    y += one;
    y = elmk_with_high_word(y, (uint32_t)(elmk_asu64(y) >> 32) + ((uint32_t)k << 20));
  }
  return y;
}

ELMK_MFN double elmk_tanh(double x)
{
  const double one = 1.0, two = 2.0, tiny = 1.0e-300;
  double t, z;
  const int32_t jx = (int32_t)(elmk_asu64(x) >> 32);
  const uint32_t lx = (uint32_t)elmk_asu64(x);
  const int32_t ix = jx & 0x7fffffff;
  if (ix >= 0x7ff00000) {  // inf or nan
    if (jx >= 0) return one / x + one;
    return one / x - one;
  }
  if (ix < 0x40360000) {  // |x| < 22
    if (((uint32_t)ix | lx) == 0) return x;
    if (ix < 0x3c800000) return x * (one + x);  // |x| < 2^-55
    if (ix >= 0x3ff00000) {                     // |x| >= 1
      t = elmk_expm1(two * __builtin_fabs(x));
      z = one - two / (t + two);
    } else {
      t = elmk_expm1(-two * __builtin_fabs(x));
      z = -t / (t + two);
    }
  } else {
    z = one - tiny;  // |x| >= 22: +-1
  }
  return (jx >= 0) ? z : -z;
}

// ---- cos: glibc 2.35 sysdeps/ieee754/dbl-64/s_sin.c (__cos_fma; IBM Accurate Mathematical Library) --------------------
// Exact for |x| < 105414350 (0x419921FB in the high word) - the ranges s_sin.c handles with do_cos / do_sin /
// reduce_sincos.  Beyond that the reference reduces with __branred (Payne-Hanek); no call site of the physics can get
// there (the arguments are pi * [0, 1]), and this function returns NaN there instead of restating it.
#define ELMK_SC(k, j) elmk_asf64(elmk_sincos_tab[4 * (k) + (j)])
ELMK_MFN double elmk_sc_do_cos(double x, double dx)
{
  const double big = 0x1.8p+45, sn3 = -0x1.5555555555515p-3, sn5 = 0x1.11110e829872fp-7, cs2 = 0.5, cs4 = -0x1.5555555555535p-5,
               cs6 = 0x1.6c16bedd9e239p-10;
  if (x < 0) dx = -dx;
  const double ax = __builtin_fabs(x);
  const double u = big + ax;
  x = (ax - (u - big)) + dx;
  const int k = (int)(uint32_t)elmk_asu64(u);
  const double xx = x * x;
  const double ps = ELMK_FMA(xx, sn5, sn3);
  const double s = ELMK_FMA(x * xx, ps, x);
  double pc = ELMK_FMA(xx, cs6, cs4);
  pc = ELMK_FMA(xx, pc, cs2);
  const double c = xx * pc;
  const double sn = ELMK_SC(k, 0), ssn = ELMK_SC(k, 1), cs = ELMK_SC(k, 2), ccs = ELMK_SC(k, 3);
  double cor = ELMK_FMA(-s, ssn, ccs);
  cor = ELMK_FMA(-c, cs, cor);
  cor = ELMK_FMA(-s, sn, cor);
  return cs + cor;
}
ELMK_MFN double elmk_sc_do_sin(double x, double dx)
{
  const double big = 0x1.8p+45, sn3 = -0x1.5555555555515p-3, sn5 = 0x1.11110e829872fp-7, cs2 = 0.5, cs4 = -0x1.5555555555535p-5,
               cs6 = 0x1.6c16bedd9e239p-10;
  const double s1 = -0x1.5555555555555p-3, s2 = 0x1.1111111110ecep-7, s3 = -0x1.a01a019db08b8p-13, s4 = 0x1.71de27b9a7ed9p-19,
               s5 = -0x1.addffc2fcdf59p-26;
  const double xold = x;
  if (__builtin_fabs(x) < 0.126) {  // TAYLOR_SIN(x * x, x, dx)
    const double xx = x * x;
    double p = ELMK_FMA(xx, s5, s4);
    p = ELMK_FMA(xx, p, s3);
    p = ELMK_FMA(xx, p, s2);
    p = ELMK_FMA(xx, p, s1);
    const double q = ELMK_FMA(p, x, -(0.5 * dx));
    const double t = ELMK_FMA(xx, q, dx);
    return x + t;
  }
  if (x <= 0) dx = -dx;
  const double ax = __builtin_fabs(x);
  const double u = big + ax;
  x = ax - (u - big);
  const int k = (int)(uint32_t)elmk_asu64(u);
  const double xx = x * x;
  const double ps = ELMK_FMA(xx, sn5, sn3);
  const double sd = ELMK_FMA(x * xx, ps, dx);
  double pc = ELMK_FMA(xx, cs6, cs4);
  pc = ELMK_FMA(xx, pc, cs2);
  const double s = x + sd;
  const double c = ELMK_FMA(x, dx, xx * pc);
  const double sn = ELMK_SC(k, 0), ssn = ELMK_SC(k, 1), cs = ELMK_SC(k, 2), ccs = ELMK_SC(k, 3);
  double cor = ELMK_FMA(s, ccs, ssn);
  cor = ELMK_FMA(-c, sn, cor);
  cor = ELMK_FMA(s, cs, cor);
  return elmk_atan_signed(sn + cor, xold);  // copysign
}
ELMK_MFN double elmk_cos(double x)
{
  const double hp0 = 0x1.921fb54442d18p+0, hp1 = 0x1.1a62633145c07p-54, hpinv = 0x1.45f306dc9c883p-1, toint = 0x1.8p+52,
               mp1 = 0x1.921fb58p+0, mp2 = -0x1.dde973cp-27, pp3 = -0x1.cb3b398p-55, pp4 = -0x1.d747f23e32ed7p-83;
  const uint32_t k = (uint32_t)(elmk_asu64(x) >> 32) & 0x7fffffffu;
  if (k < 0x3e400000u) return 1.0;                      // |x| < 2^-27
  if (k < 0x3feb6000u) return elmk_sc_do_cos(x, 0.0);  // |x| < 0.855469
  if (k < 0x400368fdu) {                                // |x| < 2.426265
    const double y = hp0 - __builtin_fabs(x);
    const double a = y + hp1;
    const double da = (y - a) + hp1;
    return elmk_sc_do_sin(a, da);
  }
  if (k < 0x419921FBu) {  // |x| < 105414350: reduce_sincos, then do_sincos(a, da, n + 1)
    const double t = ELMK_FMA(x, hpinv, toint);
    const double xn = t - toint;
    const int n = ((int)(uint32_t)elmk_asu64(t) & 3) + 1;
    double y = ELMK_FMA(-xn, mp1, x);
    y = ELMK_FMA(-xn, mp2, y);
    const double t2 = ELMK_FMA(-xn, pp3, y);
    double db = ELMK_FMA(-pp3, xn, y - t2);
    const double b = ELMK_FMA(-xn, pp4, t2);
    db = db + ELMK_FMA(-xn, pp4, t2 - b);
    const double r = (n & 1) ? elmk_sc_do_cos(b, db) : elmk_sc_do_sin(b, db);
    return (n & 2) ? -r : r;
  }
  if (k < 0x7ff00000u) return ELMK_NAN;  // outside the restated range (see above)
  return x / x;                          // inf, nan
}

// ---- erf: glibc 2.35 sysdeps/ieee754/dbl-64/s_erf.c (fdlibm rational approximations in glibc's pairwise-grouped form;
// no FMA build; the two exp() calls of the 1.25 <= |x| < 6 range are the new exp above) -----------------------------------
ELMK_MFN double elmk_erf(double x)
{
  const double one = 1.0, tiny = 1e-300, erx = 0x1.b0ac16p-1, efx = 0x1.06eba8214db69p-3, efx16 = 0x1.06eba8214db69p+1;
  const int32_t hx = (int32_t)(elmk_asu64(x) >> 32);
  const int32_t ix = hx & 0x7fffffff;
  if (ix >= 0x7ff00000) {  // erf(nan) = nan, erf(+-inf) = +-1
    const int i = (int)(((uint32_t)hx >> 31) << 1);
    return (double)(1 - i) + one / x;
  }
  if (ix < 0x3feb0000) {    // |x| < 0.84375
    if (ix < 0x3e300000) {  // |x| < 2^-28
      if (ix < 0x00800000) return 0.0625 * (16.0 * x + efx16 * x);  // avoid underflow
      return x + efx * x;
    }
    const double z = x * x;
    const double r1 = 0x1.06eba8214db68p-3 + z * -0x1.4cd7d691cb913p-2;
    const double z2 = z * z;
    const double r2 = z * -0x1.7a291236668e4p-8 - 0x1.d2a51dbd7194fp-6;
    const double z4 = z2 * z2;
    const double s1 = one + z * 0x1.97779cddadc09p-2;
    const double s2 = 0x1.0a54c5536cebap-4 + z * 0x1.4d022c4d36b0fp-8;
    const double s3 = 0x1.15dc9221c1a1p-13 + z * -0x1.09c4342a2612p-18;
    const double r = r1 + z2 * r2 + z4 * -0x1.8ead6120016acp-16;
    const double s = s1 + z2 * s2 + z4 * s3;
    const double y = r / s;
    return x + x * y;
  }
  if (ix < 0x3ff40000) {  // 0.84375 <= |x| < 1.25
    const double s = __builtin_fabs(x) - one;
    const double P1 = s * 0x1.a8d00ad92b34dp-2 - 0x1.359b8bef77538p-9;
    const double s2 = s * s;
    const double Q1 = one + s * 0x1.b3e6618eee323p-4;
    const double s4 = s2 * s2;
    const double P2 = s * 0x1.45fca805120e4p-2 - 0x1.7d240fbb8c3f1p-2;
    const double s6 = s4 * s2;
    const double Q2 = 0x1.14af092eb6f33p-1 + s * 0x1.2635cd99fe9a7p-4;
    const double P3 = s * 0x1.22a36599795ebp-5 - 0x1.c63983d3e28ecp-4;
    const double Q3 = 0x1.02660e763351fp-3 + s * 0x1.bedc26b51dd1cp-7;
    const double P4 = -0x1.1bf380a96073fp-9;
    const double Q4 = 0x1.88b545735151dp-7;
    const double P = P1 + s2 * P2 + s4 * P3 + s6 * P4;
    const double Q = Q1 + s2 * Q2 + s4 * Q3 + s6 * Q4;
    if (hx >= 0) return erx + P / Q;
    return -erx - P / Q;
  }
  if (ix >= 0x40180000) {  // 6 <= |x| < inf
    if (hx >= 0) return one - tiny;
    return tiny - one;
  }
  const double ax = __builtin_fabs(x);
  const double s = one / (ax * ax);
  double R, S;
  if (ix < 0x4006DB6E) {  // |x| < 1 / 0.35
    const double R1 = s * -0x1.63416e4ba736p-1 - 0x1.43412600d6435p-7;
    const double s2 = s * s;
    const double S1 = one + s * 0x1.3a6b9bd707687p+4;
    const double s4 = s2 * s2;
    const double R2 = s * -0x1.f300ae4cba38dp+5 - 0x1.51e0441b0e726p+3;
    const double s6 = s4 * s2;
    const double S2 = 0x1.1350c526ae721p+7 + s * 0x1.b290dd58a1a71p+8;
    const double s8 = s4 * s4;
    const double R3 = s * -0x1.7135cebccabb2p+7 - 0x1.44cb184282266p+7;
    const double S3 = 0x1.42b1921ec2868p+9 + s * 0x1.ad02157700314p+8;
    const double R4 = s * -0x1.3a0efc69ac25cp+3 - 0x1.4526557e4d2f2p+6;
    const double S4 = 0x1.b28a3ee48ae2cp+6 + s * 0x1.a47ef8e484a93p+2;
    R = R1 + s2 * R2 + s4 * R3 + s6 * R4;
    S = S1 + s2 * S2 + s4 * S3 + s6 * S4 + s8 * -0x1.eeff2ee749a62p-5;
  } else {  // |x| >= 1 / 0.35
    const double R1 = s * -0x1.993ba70c285dep-1 - 0x1.4341239e86f4ap-7;
    const double s2 = s * s;
    const double S1 = one + s * 0x1.e568b261d519p+4;
    const double s4 = s2 * s2;
    const double R2 = s * -0x1.4145d43c5ed98p+7 - 0x1.1c209555f995ap+4;
    const double s6 = s4 * s2;
    const double S2 = 0x1.45cae221b9f0ap+8 + s * 0x1.802eb189d5118p+10;
    const double R3 = s * -0x1.004616a2e5992p+10 - 0x1.3ec881375f228p+9;
    const double S3 = 0x1.8ffb7688c246ap+11 + s * 0x1.3f219cedf3be6p+11;
    const double S4 = 0x1.da874e79fe763p+8 + s * -0x1.670e242712d62p+4;
    R = R1 + s2 * R2 + s4 * R3 + s6 * -0x1.e384e9bdc383fp+8;
    S = S1 + s2 * S2 + s4 * S3 + s6 * S4;
  }
  const double z = elmk_asf64(elmk_asu64(ax) & 0xffffffff00000000ull);
  const double r = elmk_exp(-z * z - 0.5625) * elmk_exp((z - ax) * (z + ax) + R / S);
  if (hx >= 0) return one - r / ax;
  return r / ax - one;
}

// ---- acos: glibc 2.35 sysdeps/ieee754/dbl-64/e_asin.c (__ieee754_acos, FMA build; IBM Accurate Mathematical Library, the
// version without the multi-precision fall-backs) -----------------------------------------------------------------------
#define ELMK_AS(i) elmk_asf64(elmk_asncs_tab[i])
// the table ranges share one shape: xx = |x| - x_i (with x's sign folded in), a polynomial of degree deg in xx whose last
// coefficient is at n + deg + 1, then asin(x_i) in two pieces
ELMK_MFN double elmk_acos_row(double x, int positive, int n, int deg)
{
  const double hp0 = 0x1.921fb54442d18p+0, hp1 = 0x1.1a62633145c07p-54;
  const double xx = (positive ? x : -x) - ELMK_AS(n);
  double p = ELMK_AS(n + deg);
  for (int j = deg - 1; j >= 2; --j) p = ELMK_FMA(xx, p, ELMK_AS(n + j));
  p = ELMK_FMA(xx * xx, p, ELMK_AS(n + deg + 1));
  const double t = ELMK_FMA(xx, ELMK_AS(n + 1), p);
  const double y = ELMK_AS(n + deg + 2);
  if (positive) return (hp1 - t) + (hp0 - y);
  return (t + hp1) + (y + hp0);
}
ELMK_MFN double elmk_acos(double x)
{
  const double hp0 = 0x1.921fb54442d18p+0, hp1 = 0x1.1a62633145c07p-54;
  const double f1 = 0x1.55555555554f9p-3, f2 = 0x1.333333336127dp-4, f3 = 0x1.6db6dae42c0e4p-5, f4 = 0x1.f1c7e04f4ad99p-6,
               f5 = 0x1.6e442c822d419p-6, f6 = 0x1.292d80f453c72p-6;
  const int32_t m = (int32_t)(elmk_asu64(x) >> 32);
  const uint32_t lo = (uint32_t)elmk_asu64(x);
  const int32_t k = m & 0x7fffffff;
  if (k < 0x3c880000) return hp0;  // |x| < 2^-55
  if (k < 0x3fc00000) {            // |x| < 0.125
    const double x2 = x * x;
    double p = ELMK_FMA(x2, f6, f5);
    p = ELMK_FMA(x2, p, f4);
    p = ELMK_FMA(x2, p, f3);
    p = ELMK_FMA(x2, p, f2);
    p = ELMK_FMA(x2, p, f1);
    const double r = hp0 - x;
    double cor = hp0 - r;
    const double x3 = x * x2;
    cor = cor - x;
    cor = cor + hp1;
    cor = ELMK_FMA(-p, x3, cor);
    return r + cor;
  }
  if (k < 0x3fe00000) {  // |x| < 0.5
    const int n = (k < 0x3fd00000) ? 11 * ((k & 0x000f8000) >> 15) : 11 * ((k & 0x000fc000) >> 14) + 352;
    return elmk_acos_row(x, m > 0, n, 6);
  }
  if (k < 0x3fe80000) return elmk_acos_row(x, m > 0, 1056 + 12 * ((k >> 13) & 0x7f), 7);  // |x| < 0.75
  if (k < 0x3fed8000) return elmk_acos_row(x, m > 0, 992 + 13 * ((k >> 13) & 0x7f), 8);   // |x| < 0.921875
  if (k < 0x3fee8000) return elmk_acos_row(x, m > 0, 884 + 14 * ((k >> 13) & 0x7f), 9);   // |x| < 0.953125
  if (k < 0x3fef0000) return elmk_acos_row(x, m > 0, 768 + 15 * ((k >> 13) & 0x7f), 10);  // |x| < 0.96875
  if (k < 0x3ff00000) {  // |x| < 1: acos(|x|) = 2 asin(sqrt((1 - |x|) / 2)), the root by table seed + one Newton step
    const double rt0 = 0x1.fffffffecc1ddp-1, rt1 = 0x1.fffffff757304p-2, rt2 = 0x1.800496769c91ap-2, rt3 = 0x1.4006318d1dab9p-2;
    const double z = 0.5 * ((m > 0) ? (1.0 - x) : (1.0 + x));
    const uint32_t hz = (uint32_t)(elmk_asu64(z) >> 32);
    double t = elmk_asf64(elmk_inroot_tab[(hz >> 14) & 0x7f]) * elmk_asf64((uint64_t)(0x3ff + 511 - (int)(hz >> 21)) << 52);
    const double r = ELMK_FMA(-(t * t), z, 1.0);
    double q = ELMK_FMA(r, rt3, rt2);
    q = ELMK_FMA(r, q, rt1);
    q = ELMK_FMA(r, q, rt0);
    t = q * t;
    const double c = z * t;
    const double h = ELMK_FMA(-c, 0.5 * t, 1.5);
    const double w = ELMK_FMA(c, 0x1p27, c);
    const double y = ELMK_FMA(-0x1p27, c, w);
    const double ty = ELMK_FMA(h, c, y);
    const double cc = ELMK_FMA(-y, y, z) / ty;
    double p = ELMK_FMA(z, f6, f5);
    p = ELMK_FMA(z, p, f4);
    p = ELMK_FMA(z, p, f3);
    p = ELMK_FMA(z, p, f2);
    p = ELMK_FMA(z, p, f1);
    p = p * z;
    const double pc = p * (y + cc);
    if (m >= 0) {
      double res = cc + pc;
      res = res + y;
      return res + res;
    }
    const double a = hp1 - cc;
    const double b = hp0 - y;
    double res = a - pc;
    res = res + b;
    return res + res;
  }
  if (k == 0x3ff00000 && lo == 0) return (m > 0) ? 0.0 : 0x1.921fb54442d18p+1;  // acos(1) = 0, acos(-1) = pi
  if (k > 0x7ff00000 || (k == 0x7ff00000 && lo != 0)) return x + x;               // nan
  return ELMK_NAN;                                                                // |x| > 1
}
