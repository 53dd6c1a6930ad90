// bl_qnorm.hpp -- inverse of the standard normal CDF.  Portable (host + device).
#pragma once
#include "bl_fastmath.hpp"
#include "bl_portable.hpp"

namespace bl {

// Phi^{-1}(p), lower tail: Wichura (1988) AS 241 PPND16 (published algorithm).
// FAST: the divide, the logarithm and the square root in bl_fastmath's short forms (< 1.5 ulp; p in [2^-53, 1 - 2^-53]): the
// attempt bodies call this once per left-piece attempt, and with libm's sequences the two branches a wavefront nearly always
// has both of (85 % / 15 % of the lanes) were 220 vector instructions against 100.
// Horner with every step ONE v_fma_f64 whose addend sits in a scalar register pair (bl_portable.hpp: the compiler's own choice
// for a constant addend is v_mov_b64 + v_fmac_f64 -- the three pairs of degree-7 polynomials below were 90 of the 128 register
// moves of a class-2 attempt of the PG(1,z) sampler).  Same rounding as the fused multiply-add the device build contracts to.
#define BL_H7(r, c7, c6, c5, c4, c3, c2, c1, c0)                                                                              \
  fma_vvs(fma_vvs(fma_vvs(fma_vvs(fma_vvs(fma_vvs(fma_vvs((r), (c7), (c6)), (r), (c5)), (r), (c4)), (r), (c3)), (r), (c2)), \
                  (r), (c1)), (r), (c0))
template <bool FAST>
BL_HD double qnorm_t(double p)
{
  const double q = p - 0.5;
  double r, val;
  if (fabs(q) <= 0.425) {
    r = 0.180625 - q * q;
    const double num = q * BL_H7(r, 2509.0809287301226727, 33430.575583588128105, 67265.770927008700853, 45921.953931549871457,
                                 13731.693765509461125, 1971.5909503065514427, 133.14166789178437745, 3.387132872796366608);
    const double den = BL_H7(r, 5226.495278852545925, 28729.085735721942674, 39307.89580009271061, 21213.794301586595867,
                             5394.1960214247511077, 687.1870074920579083, 42.313330701600911252, 1.0);
    return FAST ? bl_div(num, den) : num / den;
  }
  r = q < 0 ? p : 1.0 - p;
  r = FAST ? bl_sqrt(-bl_log(r)) : sqrt(-log(r));
  double num, den;
  if (r <= 5.0) {
    r -= 1.6;
    num = BL_H7(r, 7.7454501427834140764e-4, 0.0227238449892691845833, 0.24178072517745061177, 1.27045825245236838258,
                3.64784832476320460504, 5.7694972214606914055, 4.6303378461565452959, 1.42343711074968357734);
    den = BL_H7(r, 1.05075007164441684324e-9, 5.475938084995344946e-4, 0.0151986665636164571966, 0.14810397642748007459,
                0.68976733498510000455, 1.6763848301838038494, 2.05319162663775882187, 1.0);
  } else {
    r -= 5.0;
    num = BL_H7(r, 2.01033439929228813265e-7, 2.71155556874348757815e-5, 0.0012426609473880784386, 0.026532189526576123093,
                0.29656057182850489123, 1.7848265399172913358, 5.4637849111641143699, 6.6579046435011037772);
    den = BL_H7(r, 2.04426310338993978564e-15, 1.4215117583164458887e-7, 1.8463183175100546818e-5, 7.868691311456132591e-4,
                0.0148753612908506148525, 0.13692988092273580531, 0.59983220655588793769, 1.0);
  }
  val = FAST ? bl_div(num, den) : num / den;
  return q < 0.0 ? -val : val;
}
#undef BL_H7
BL_HD double qnorm(double p) { return qnorm_t<false>(p); }


}  // namespace bl
