/* specfun.c -- oracle (test infrastructure, see bl_oracle.h): the special
 * functions the reference takes from the absent RNG library
 * (RNG::p_norm, RNG::p_gamma_rate, RNG::p_igauss).
 * Checked against scipy known answers in tests/test_oracle_specfun.py.
 */
#include "bl_oracle.h"
#include <math.h>
#include <float.h>

#define BL_LOG_SQRT_2PI 0.918938533204672741780329736406

/* RNG::p_norm(x, log): Phi(x) or log Phi(x) = R's pnorm(x, log.p=).
 * Call sites PolyaGamma.cpp:61,74-75; PolyaGammaAlt.cpp:56. */
double bl_p_norm(double x, int use_log)
{
  if (!use_log) return 0.5 * erfc(-x * M_SQRT1_2);
  if (x >= 0.0) return log1p(-0.5 * erfc(x * M_SQRT1_2));
  if (x > -37.0) return log(0.5 * erfc(-x * M_SQRT1_2));
  /* asymptotic Mills-ratio series, |x| >= 37: relative error < 1e-15 */
  double x2 = x * x, r = 1.0 / x2;
  double s = 1.0 - r * (1.0 - 3.0 * r * (1.0 - 5.0 * r * (1.0 - 7.0 * r * (1.0 - 9.0 * r))));
  return -0.5 * x2 - log(-x) - BL_LOG_SQRT_2PI + log(s);
}

/* regularised lower incomplete gamma P(a, x): series for x < a+1, modified
 * Lentz continued fraction for Q otherwise (Numerical Recipes 6.2 structure,
 * published algorithm). */
static double reg_lower_gamma(double a, double x)
{
  if (!(x > 0.0)) return 0.0;
  if (isinf(x)) return 1.0;
  double lg = lgamma(a);
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
    double an = -(double)i * ((double)i - a);
    b += 2.0;
    d = an * d + b; if (fabs(d) < tiny) d = tiny;
    c = b + an / c; if (fabs(c) < tiny) c = tiny;
    d = 1.0 / d;
    double del = d * c;
    h *= del;
    if (fabs(del - 1.0) < 1e-16) break;
  }
  double q = exp(-x + a * log(x) - lg) * h;
  return 1.0 - q;
}

/* RNG::p_gamma_rate(x, shape, rate) = pgamma(x, shape, rate=rate).
 * Call sites PolyaGammaAlt.cpp:66,73; PolyaGammaSP.cpp:222. */
double bl_p_gamma_rate(double x, double shape, double rate)
{
  return reg_lower_gamma(shape, rate * x);
}

/* RNG::p_igauss(x, mu, lambda): inverse-Gaussian CDF, Code/R/PG.R:15-23,
 * second term kept in log space as Code/R/SPSample.R:523-532 does. */
double bl_p_igauss(double x, double mu, double lambda)
{
  double Z = 1.0 / mu;
  double s = sqrt(lambda / x);
  double b = s * (x * Z - 1.0);
  double a = -s * (x * Z + 1.0);
  return exp(bl_p_norm(b, 1)) + exp(2.0 * lambda * Z + bl_p_norm(a, 1));
}
