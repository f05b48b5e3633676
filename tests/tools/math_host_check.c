/* Host check of elmkernels_amd/csrc/elmk_math.h against the live libm, bit for bit (any two NaNs count as equal).
 * gcc -O2 -mfma -ffp-contract=off -fopenmp tests/tools/math_host_check.c -lm ; ./a.out <n per class> <seed>
 * Prints one line per function: "<fn> n=<evaluated> mismatches=<count>" and the first few offending arguments. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../elmkernels_amd/csrc/elmk_math.h"

static inline uint64_t mix(uint64_t z)
{
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
static inline double u01(uint64_t r) { return (double)(r >> 11) * 0x1p-53; }
static inline int same(double a, double b)
{
  if (a != a && b != b) return 1;
  return elmk_asu64(a) == elmk_asu64(b);
}
static const double YS[] = {3.0, 4.0, 0.333, -0.333, 0.666666666666, 1.5, 2.0, 0.5, -1.0, 40.0, 0.25, 1.0 / 3.0, -0.5, 2.5, 7.0, -2.0};

int main(int argc, char** argv)
{
  const long n = argc > 1 ? atol(argv[1]) : 1000000;
  const uint64_t seed = argc > 2 ? strtoull(argv[2], 0, 0) : 1;
  long bad[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tot[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const char* nm[10] = {"exp", "log", "log10", "pow", "atan", "expm1", "tanh", "cos", "erf", "acos"};
  for (int cls = 0; cls < 6; cls++) {
    long b0 = 0, b1 = 0, b2 = 0, b3 = 0, b4 = 0, b5 = 0, b6 = 0, b7 = 0, b8 = 0, b9 = 0;
#pragma omp parallel for reduction(+ : b0, b1, b2, b3, b4, b5, b6, b7, b8, b9) schedule(static)
    for (long j = 0; j < n; j++) {
      const uint64_t r1 = mix(seed * 1000003ull + (uint64_t)cls * 0x100000000ull + 2 * j), r2 = mix(r1 + 12345), r3 = mix(r2 + 777);
      double xe, xl, px, py;
      switch (cls) {
        case 0: xe = (u01(r1) - 0.5) * 1500.0; xl = exp((u01(r1) - 0.5) * 1400.0); px = xl; py = (u01(r2) - 0.5) * 100.0; break;
        case 1: xe = (u01(r1) - 0.5) * 2.0; xl = 0.9 + 0.2 * u01(r1); px = 2.0 * u01(r1); py = (u01(r2) - 0.5) * 10.0; break;
        case 2: xe = elmk_asf64(r1); xl = elmk_asf64(r1 & 0x7fffffffffffffffull); px = elmk_asf64(r1); py = elmk_asf64(r2); break;
        case 3: xe = (u01(r1) - 0.5) * 100.0; xl = exp((u01(r1) - 0.5) * 46.0); px = 1000.0 * u01(r1); py = YS[r2 & 15]; break;
        case 4: xe = -745.2 + 40.0 * u01(r1); xl = elmk_asf64(r1 & 0x000fffffffffffffull); /* subnormal */
                px = -(double)(r1 % 50) * ((r3 & 1) ? 1.0 : 0.37); py = (double)((long)(r2 % 21) - 10) * ((r3 & 2) ? 1.0 : 0.5); break;
        default: xe = 700.0 + 12.0 * u01(r1); xl = 1.0 + (u01(r1) - 0.5) * 0x1p-20; px = exp((u01(r1) - 0.5) * 2.0); py = (u01(r2) - 0.5) * 3000.0; break;
      }
      b0 += !same(exp(xe), elmk_exp(xe));
      b1 += !same(log(xl), elmk_log(xl));
      b2 += !same(log10(xl), elmk_log10(xl));
      b3 += !same(pow(px, py), elmk_pow(px, py));
      b4 += !same(atan(xe), elmk_atan(xe)) + !same(atan(px), elmk_atan(px));
      b5 += !same(expm1(xe), elmk_expm1(xe)) + !same(expm1(py), elmk_expm1(py));
      b6 += !same(tanh(xe), elmk_tanh(xe)) + !same(tanh(py), elmk_tanh(py));
      { const double a1 = 2.0 * u01(r3) - 1.0, a2 = (r2 & 1 ? 1.0 : -1.0) * (1.0 - 0.04 * u01(r1) * u01(r2));
        b9 += !same(acos(a1), elmk_acos(a1)) + !same(acos(a2), elmk_acos(a2)) + !same(acos(xe), elmk_acos(xe)); }
      b8 += !same(erf(xe), elmk_erf(xe)) + !same(erf(py), elmk_erf(py)) + !same(erf(xl - 1.0), elmk_erf(xl - 1.0));
      { const double c1 = 3.14159265358979323846 * u01(r3), c2 = (cls == 2 && !(fabs(xe) < 1.0e8)) ? c1 : xe, c3 = px * 1.0e5 * (u01(r2) - 0.5);
        b7 += !same(cos(c1), elmk_cos(c1)) + !same(cos(c2), elmk_cos(c2)) + !same(cos(c3), elmk_cos(c3) ) * (fabs(c3) < 1.0e8); }
    }
    bad[0] += b0; bad[1] += b1; bad[2] += b2; bad[3] += b3; bad[4] += b4; bad[5] += b5; bad[6] += b6; bad[7] += b7; bad[8] += b8; bad[9] += b9;
    for (int f = 0; f < 4; f++) tot[f] += n;
    tot[4] += 2 * n; tot[5] += 2 * n; tot[6] += 2 * n; tot[7] += 3 * n; tot[8] += 3 * n; tot[9] += 3 * n;
    if (b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7 + b8 + b9) {  /* show a few */
      int shown = 0;
      for (long j = 0; j < n && shown < 4; j++) {
        const uint64_t r1 = mix(seed * 1000003ull + (uint64_t)cls * 0x100000000ull + 2 * j), r2 = mix(r1 + 12345), r3 = mix(r2 + 777);
        double xe, xl, px, py;
        switch (cls) {
          case 0: xe = (u01(r1) - 0.5) * 1500.0; xl = exp((u01(r1) - 0.5) * 1400.0); px = xl; py = (u01(r2) - 0.5) * 100.0; break;
          case 1: xe = (u01(r1) - 0.5) * 2.0; xl = 0.9 + 0.2 * u01(r1); px = 2.0 * u01(r1); py = (u01(r2) - 0.5) * 10.0; break;
          case 2: xe = elmk_asf64(r1); xl = elmk_asf64(r1 & 0x7fffffffffffffffull); px = elmk_asf64(r1); py = elmk_asf64(r2); break;
          case 3: xe = (u01(r1) - 0.5) * 100.0; xl = exp((u01(r1) - 0.5) * 46.0); px = 1000.0 * u01(r1); py = YS[r2 & 15]; break;
          case 4: xe = -745.2 + 40.0 * u01(r1); xl = elmk_asf64(r1 & 0x000fffffffffffffull);
                  px = -(double)(r1 % 50) * ((r3 & 1) ? 1.0 : 0.37); py = (double)((long)(r2 % 21) - 10) * ((r3 & 2) ? 1.0 : 0.5); break;
          default: xe = 700.0 + 12.0 * u01(r1); xl = 1.0 + (u01(r1) - 0.5) * 0x1p-20; px = exp((u01(r1) - 0.5) * 2.0); py = (u01(r2) - 0.5) * 3000.0; break;
        }
        if (!same(exp(xe), elmk_exp(xe))) { printf("  cls %d exp(%a) = %a, got %a\n", cls, xe, exp(xe), elmk_exp(xe)); shown++; }
        if (!same(log(xl), elmk_log(xl))) { printf("  cls %d log(%a) = %a, got %a\n", cls, xl, log(xl), elmk_log(xl)); shown++; }
        if (!same(log10(xl), elmk_log10(xl))) { printf("  cls %d log10(%a) = %a, got %a\n", cls, xl, log10(xl), elmk_log10(xl)); shown++; }
        if (!same(atan(xe), elmk_atan(xe))) { printf("  cls %d atan(%a) = %a, got %a\n", cls, xe, atan(xe), elmk_atan(xe)); shown++; }
        if (!same(atan(px), elmk_atan(px))) { printf("  cls %d atan(%a) = %a, got %a\n", cls, px, atan(px), elmk_atan(px)); shown++; }
        { const double c1 = 3.14159265358979323846 * u01(r3), c2 = (cls == 2 && !(fabs(xe) < 1.0e8)) ? c1 : xe, c3 = px * 1.0e5 * (u01(r2) - 0.5);
          if (!same(cos(c1), elmk_cos(c1))) { printf("  cls %d cos(%a) = %a, got %a\n", cls, c1, cos(c1), elmk_cos(c1)); shown++; }
          if (!same(cos(c2), elmk_cos(c2))) { printf("  cls %d cos(%a) = %a, got %a\n", cls, c2, cos(c2), elmk_cos(c2)); shown++; }
          if (fabs(c3) < 1.0e8 && !same(cos(c3), elmk_cos(c3))) { printf("  cls %d cos(%a) = %a, got %a\n", cls, c3, cos(c3), elmk_cos(c3)); shown++; } }
        if (!same(erf(xe), elmk_erf(xe))) { printf("  cls %d erf(%a) = %a, got %a\n", cls, xe, erf(xe), elmk_erf(xe)); shown++; }
        if (!same(erf(py), elmk_erf(py))) { printf("  cls %d erf(%a) = %a, got %a\n", cls, py, erf(py), elmk_erf(py)); shown++; }
        { const double a1 = 2.0 * u01(r3) - 1.0, a2 = (r2 & 1 ? 1.0 : -1.0) * (1.0 - 0.04 * u01(r1) * u01(r2));
          if (!same(acos(a1), elmk_acos(a1))) { printf("  cls %d acos(%a) = %a, got %a\n", cls, a1, acos(a1), elmk_acos(a1)); shown++; }
          if (!same(acos(a2), elmk_acos(a2))) { printf("  cls %d acos(%a) = %a, got %a\n", cls, a2, acos(a2), elmk_acos(a2)); shown++; }
          if (!same(acos(xe), elmk_acos(xe))) { printf("  cls %d acos(%a) = %a, got %a\n", cls, xe, acos(xe), elmk_acos(xe)); shown++; } }
        if (!same(expm1(xe), elmk_expm1(xe))) { printf("  cls %d expm1(%a) = %a, got %a\n", cls, xe, expm1(xe), elmk_expm1(xe)); shown++; }
        if (!same(expm1(py), elmk_expm1(py))) { printf("  cls %d expm1(%a) = %a, got %a\n", cls, py, expm1(py), elmk_expm1(py)); shown++; }
        if (!same(tanh(xe), elmk_tanh(xe))) { printf("  cls %d tanh(%a) = %a, got %a\n", cls, xe, tanh(xe), elmk_tanh(xe)); shown++; }
        if (!same(tanh(py), elmk_tanh(py))) { printf("  cls %d tanh(%a) = %a, got %a\n", cls, py, tanh(py), elmk_tanh(py)); shown++; }
        if (!same(pow(px, py), elmk_pow(px, py))) { printf("  cls %d pow(%a, %a) = %a, got %a\n", cls, px, py, pow(px, py), elmk_pow(px, py)); shown++; }
      }
    }
  }
  for (int f = 0; f < 10; f++) printf("%s n=%ld mismatches=%ld\n", nm[f], tot[f], bad[f]);
  long allbad = 0;
  for (int f = 0; f < 10; f++) allbad += bad[f];
  return allbad != 0;
}
