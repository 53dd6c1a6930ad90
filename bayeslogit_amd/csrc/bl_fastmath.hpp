// bl_fastmath.hpp -- fp64 log and exp for the samplers' inner loops.
// ocml's log(double) is ~98 VALU instructions on gfx950 and exp ~42 (measured from the ISA);
// the attempt body of the PG(1,z) sampler (bl_pg1_sm.hpp) calls both for every proposal, so they are the
// largest line items of the draw kernels.  These versions are the classic argument-reduction
// + short polynomial forms (log: fdlibm e_log.c's s = f/(2+f) series with its Lg1..Lg7
// coefficients and hi/lo split of ln 2; exp: k ln2 reduction + degree-13 Taylor/Horner on
// |r| <= ln2/2), < 1.5 ulp on the ranges the samplers use (checked against libm in
// tests/test_host_harness.py): ~38 and ~24 VALU instructions.
// Domain: bl_log(x) for finite normal x > 0; bl_exp(x) for any finite x (underflows to 0,
// overflows to +inf).  Portable (host + device).
#pragma once
#include "bl_portable.hpp"
#include <string.h>

namespace bl {

// a/b to ~1 ulp without the IEEE division sequence (v_div_scale/fmas/fixup): hardware
// reciprocal estimate, two Newton steps, one correction.  b normal, no overflow handling.
BL_HD double bl_div(double a, double b)
{
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(b);
#else
  double r = 1.0 / b;
#endif
  r = fma(fma(-b, r, 1.0), r, r);
  r = fma(fma(-b, r, 1.0), r, r);
  const double q = a * r;
  return fma(fma(-b, q, a), r, q);
}

// sqrt(x) to ~1 ulp for normal x in [1e-300, 1e300] without the IEEE sequence's scaling and fix-up:
// hardware reciprocal-square-root estimate, two coupled Newton steps (Goldschmidt), one correction.
BL_HD double bl_sqrt(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
  const double y = __builtin_amdgcn_rsq(x);
#else
  const double y = 1.0 / sqrt(x);
#endif
  double g = x * y, h = 0.5 * y;
  double r = fma(-g, h, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  r = fma(-g, h, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  return fma(fma(-g, g, x), h, g);
}

BL_HD double bl_log(double x)
{
  constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  constexpr double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                   Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                   Lg7 = 1.479819860511658591e-01;
  uint64_t bits;
  memcpy(&bits, &x, 8);
  int k = (int)(bits >> 52) - 1023;
  uint64_t mant = bits & 0x000FFFFFFFFFFFFFull;
  // m in [sqrt(1/2), sqrt(2)): if the mantissa is above sqrt(2) halve it and bump the exponent
  const uint64_t over = mant > 0x6A09E667F3BCDull ? 1ull : 0ull;
  k += (int)over;
  const uint64_t mb = mant | ((1023ull - over) << 52);
  double m;
  memcpy(&m, &mb, 8);
  const double f = m - 1.0;
  const double s = bl_div(f, 2.0 + f);
  const double z = s * s;
  const double w = z * z;
  const double t1 = w * fma_vvs(w, fma_vvs(w, Lg6, Lg4), Lg2);
  const double t2 = z * fma_vvs(w, fma_vvs(w, fma_vvs(w, Lg7, Lg5), Lg3), Lg1);
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double dk = (double)k;
  return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

// exp(x) for -708 <= x <= 709.78 (no range checks)
BL_HD double bl_exp_core(double x)
{
  constexpr double inv_ln2 = 1.44269504088896338700e+00;
  constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  // k = round(x / ln 2) by the 1.5 * 2^52 trick: the integer lands in the low mantissa bits
  const double shifted = x * inv_ln2 + 0x1.8p52;
  const double kd = shifted - 0x1.8p52;
  uint64_t sb;
  memcpy(&sb, &shifted, 8);
  const int k = (int)(uint32_t)sb;
  const double r = (x - kd * ln2_hi) - kd * ln2_lo;
  // exp(r), |r| <= 0.3466: Taylor to r^13 (truncation 4e-18)
  double p = fma_vvs(r, 1.0 / 6227020800.0, 1.0 / 479001600.0);
  p = fma_vvs(p, r, 1.0 / 39916800.0);
  p = fma_vvs(p, r, 1.0 / 3628800.0);
  p = fma_vvs(p, r, 1.0 / 362880.0);
  p = fma_vvs(p, r, 1.0 / 40320.0);
  p = fma_vvs(p, r, 1.0 / 5040.0);
  p = fma_vvs(p, r, 1.0 / 720.0);
  p = fma_vvs(p, r, 1.0 / 120.0);
  p = fma_vvs(p, r, 1.0 / 24.0);
  p = fma_vvs(p, r, 1.0 / 6.0);
  p = p * r + 0.5;
  p = p * r + 1.0;
  p = p * r + 1.0;
  // p in [0.70, 1.42]: scale by 2^k through the exponent field (result stays normal: x >= -708)
  uint64_t pb;
  memcpy(&pb, &p, 8);
  pb += (uint64_t)(uint32_t)k << 52;   // wraps mod 2^64: correct for negative k
  double out;
  memcpy(&out, &pb, 8);
  return out;
}

BL_HD double bl_exp(double x)
{
  if (x < -708.0) return 0.0;   // below: results would be subnormal; the samplers treat them as 0
  if (x > 709.78) return __builtin_huge_val();
  return bl_exp_core(x);
}

// bl_exp as straight-line code (selects instead of the two early returns; same values): for callers that
// interleave it with matrix instructions inside one basic block
BL_HD double bl_exp_straight(double x)
{
  const bool lo = x < -708.0, hi = x > 709.78;
  const double r = bl_exp_core((lo || hi) ? 0.0 : x);
  return lo ? 0.0 : (hi ? __builtin_huge_val() : r);
}

}  // namespace bl
