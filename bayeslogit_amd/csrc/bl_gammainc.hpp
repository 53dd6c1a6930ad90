// bl_gammainc.hpp -- the unnormalised upper incomplete gamma function the samplers' mixture
// weights need: RNG::Gamma(n) (1 - RNG::p_gamma_rate(...)) at Code/C/PolyaGammaSP.cpp:220-222 and
// 1 - RNG::p_gamma_rate at Code/C/PolyaGammaAlt.cpp:70-75 (both from the absent RNG library).
// Portable (host + device).
//
//   Gamma(a, x) = exp(-x) x^a / (b0 + a1/(b1 + a2/(b2 + ...))),   b_i = x + 2i + 1 - a,  a_i = -i (i - a)
//
// (Legendre's continued fraction; converges for every x > 0, in 10-30 terms where the samplers call
// it: x >= 1.18 a).  Written as the forward recurrence of the convergents' numerators and denominators
// -- four FMAs a term, no division -- with one division every eight terms for the convergence test,
// instead of the modified-Lentz form's two divisions a term.  Returning the fraction alone lets the
// callers cancel exp(-x) x^a against their own exponents analytically (no lgamma, no overflow).
#pragma once
#include "bl_fastmath.hpp"

namespace bl {

// true if any lane of the wavefront has `need` set (host build: the one caller)
BL_HD bool wave_any(bool need)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return __ballot(need) != 0ull;
#else
  return need;
#endif
}

// h(a, x) with Gamma(a, x) = exp(-x) x^a h(a, x).  x > 0.  The loop is wave-uniform: every lane
// runs until the slowest lane of its wavefront has converged (relative change < 1e-15 over eight terms).
BL_HD double upper_gamma_cf(double a, double x, int& status)
{
  const double b0 = x + 1.0 - a;
  double Am = 1.0, Bm = 0.0, A = b0, B = 1.0;   // convergents f_i = A_i / B_i of b0 + K(a_i / b_i)
  double h = bl_div(1.0, b0);
  double fi = 0.0, b = b0;
  bool done = !(x > 0.0) || !(a == a);          // also false for NaN: those lanes stop at once
  int k = 0;
  while (wave_any(!done)) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      fi += 1.0;
      b += 2.0;
      const double an = -fi * (fi - a);
      const double An = fma(b, A, an * Am), Bn = fma(b, B, an * Bm);
      Am = A; Bm = B; A = An; B = Bn;
    }
    if (fabs(A) > 1e100) { A *= 1e-100; B *= 1e-100; Am *= 1e-100; Bm *= 1e-100; }
    const double hn = bl_div(B, A);
    if (!done) {                                 // a lane's value is frozen once it has converged: the result
      done = fabs(hn - h) <= 1e-15 * fabs(hn);   // does not depend on the other lanes of its wavefront
      h = hn;
    }
    if (++k > 1000) {                            // 8000 terms
      if (!done) status |= 1;
      break;
    }
  }
  return h;
}

}  // namespace bl
