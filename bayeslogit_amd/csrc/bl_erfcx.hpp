// bl_erfcx.hpp -- scaled complementary error function erfcx(x) = exp(x^2) erfc(x), x >= 0.
// erfcx(x) (1 + 2x) is expanded in Chebyshev polynomials of t = (x - 3)/(x + 3) in [-1, 1)
// (28 terms; coefficients computed with 40-digit arithmetic by the snippet quoted in
// DESIGN.md; relative error < 4e-16 on [0, 1e5], checked in tests).  Evaluated by
// Clenshaw's recurrence: no exp, no log.  Host- and device-callable so that the same
// code is unit-tested on the CPU (tests/host_harness) and runs in the kernels.
#pragma once
#include "bl_fastmath.hpp"
#include "bl_portable.hpp"

namespace bl {

BL_HD double erfcx_pos(double x)
{
  constexpr double c[28] = {
    1.1775625741965600463, 0.0053539045396156767694, -0.093775503422842090522,
    0.054366525557443220495, -0.018976596707845207386, 0.0044534260614627115695,
    -0.00063355317105531469398, 0.000017661719523171591851, 0.000012841532864056670734,
    -2.0072285659059914438e-6, -1.8075227904148371024e-7, 7.5498283438250928278e-8,
    1.8856685238386027904e-9, -2.6995190497983639975e-9, -1.4600791199391048198e-11,
    1.0361607445969490144e-10, 1.475715976623469818e-12, -4.2773400643969881618e-12,
    -2.0792249285998015243e-13, 1.813450200323934102e-13, 1.9734255749969863969e-14,
    -7.287943734362214387e-15, -1.5169879302699402579e-15, 2.3514090455290405133e-16,
    1.0006151438958986636e-16, -2.0526637756950822749e-18, -5.6034549938209255408e-18,
    -6.1914971215037826687e-19
  };
  const double t = bl_div(x - 3.0, x + 3.0);      // (the IEEE division sequence is ~30 instructions on gfx950; two of them were
                                                  // most of this function)
  const double t2 = 2.0 * t;
  double b1 = 0.0, b2 = 0.0;
#pragma unroll
  for (int k = 27; k >= 1; --k) {
    const double b0 = t2 * b1 - b2 + c[k];
    b2 = b1;
    b1 = b0;
  }
  return bl_div(t * b1 - b2 + c[0], 1.0 + 2.0 * x);
}

// exp(t pi^2/8 - 1/(2t)) at the Devroye truncation point t = 0.64 (see pg1_mass)
constexpr double kMassC = 1.0083530457090713831;

}  // namespace bl
