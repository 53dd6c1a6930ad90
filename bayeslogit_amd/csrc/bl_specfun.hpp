// bl_specfun.hpp -- device special functions the reference takes from the absent
// RNG library: RNG::p_norm (Phi / log Phi), RNG::p_gamma_rate (regularised lower
// incomplete gamma), RNG::p_igauss (inverse-Gaussian CDF).  gfx950 only.
// Call sites in the reference: PolyaGamma.cpp:61,74-75; PolyaGammaAlt.cpp:56,66,73;
// PolyaGammaSP.cpp:218,222.
#pragma once
#include <hip/hip_runtime.h>

namespace bl {

constexpr double kLogSqrt2Pi = 0.918938533204672741780329736406;
constexpr double kSqrtHalf = 0.70710678118654752440084436210485;

// log Phi(x): erfc where it does not underflow, Mills-ratio series in the far tail.
__device__ inline double log_pnorm(double x)
{
  if (x >= 0.0) return log1p(-0.5 * erfc(x * kSqrtHalf));
  if (x > -37.0) return log(0.5 * erfc(-x * kSqrtHalf));
  const double x2 = x * x, r = 1.0 / x2;
  const double s = 1.0 - r * (1.0 - 3.0 * r * (1.0 - 5.0 * r * (1.0 - 7.0 * r * (1.0 - 9.0 * r))));
  return -0.5 * x2 - log(-x) - kLogSqrt2Pi + log(s);
}

__device__ __forceinline__ double pnorm(double x) { return 0.5 * erfc(-x * kSqrtHalf); }

// P(a, x): series below a+1, modified-Lentz continued fraction for Q above.
// lg = lgamma(a), passed in by callers that need it themselves.
__device__ inline double reg_lower_gamma(double a, double x, double lg)
{
  if (!(x > 0.0)) return 0.0;
  if (isinf(x)) return 1.0;
  if (x < a + 1.0) {
    double ap = a, del = 1.0 / a, sum = del;
    for (int n = 0; n < 2000; ++n) {
      ap += 1.0;
      del *= x / ap;
      sum += del;
      if (fabs(del) < fabs(sum) * 1e-17) break;
    }
    return sum * exp(-x + a * log(x) - lg);
  }
  const double tiny = 1e-300;
  double b = x + 1.0 - a, c = 1.0 / tiny, d = 1.0 / b, h = d;
  for (int i = 1; i < 2000; ++i) {
    const double an = -(double)i * ((double)i - a);
    b += 2.0;
    d = an * d + b;
    if (fabs(d) < tiny) d = tiny;
    c = b + an / c;
    if (fabs(c) < tiny) c = tiny;
    d = 1.0 / d;
    const double del = d * c;
    h *= del;
    if (fabs(del - 1.0) < 1e-16) break;
  }
  const double q = exp(-x + a * log(x) - lg) * h;
  return 1.0 - q;
}

__device__ __forceinline__ double p_gamma_rate(double x, double shape, double rate)
{
  return reg_lower_gamma(shape, rate * x, lgamma(shape));
}
__device__ __forceinline__ double p_gamma_rate(double x, double shape, double rate, double lgam_shape)
{
  return reg_lower_gamma(shape, rate * x, lgam_shape);
}

// Inverse-Gaussian CDF, second term in log space.
__device__ inline double p_igauss(double x, double mu, double lambda)
{
  const double Z = 1.0 / mu;
  const double s = sqrt(lambda / x);
  const double b = s * (x * Z - 1.0);
  const double a = -s * (x * Z + 1.0);
  return exp(log_pnorm(b)) + exp(2.0 * lambda * Z + log_pnorm(a));
}

}  // namespace bl
