/* pg_alt.c -- oracle (test infrastructure, see bl_oracle.h).
 * Restates Code/C/PolyaGammaAlt.{h,cpp}: the alternating-series sampler for
 * real shape h in [1,4], chained for larger h.
 */
#include "bl_oracle.h"
#include <math.h>

#define PG_PI   3.141592653589793238462643383279502884197
#define PISQ    (PG_PI * PG_PI)      /* PolyaGammaAlt.h:9 */
#define HALFPI  (0.5 * PG_PI)        /* PolyaGammaAlt.h:10 */

/* free rtinvchi2(h, trunc, r) -- PolyaGammaAlt.cpp:6-22 */
static double alt_rtinvchi2(double h, double trunc, bl_rng *r)
{
  double h2 = h * h;
  double R = trunc / h2;
  double E1 = bl_expon_rate(r, 1.0);
  double E2 = bl_expon_rate(r, 1.0);
  while ((E1 * E1) > (2 * E2 / R)) {
    E1 = bl_expon_rate(r, 1.0);
    E2 = bl_expon_rate(r, 1.0);
  }
  double X = 1 + E1 * R;
  X = R / (X * X);
  X = h2 * X;
  return X;
}

/* PolyaGammaAlt::a_coef(n, x, h) -- PolyaGammaAlt.cpp:26-35 */
double bl_alt_a_coef(int n, double x, double h)
{
  double d_n = 2.0 * (double)n + h;
  double log_out = h * log(2.0) - lgamma(h) + lgamma(n + h)
                 - lgamma(n + 1) + log(d_n)
                 - 0.5 * log(2.0 * PG_PI * x * x * x) - 0.5 * d_n * d_n / x;
  return exp(log_out);
}

/* PolyaGammaAlt::a_coef_recursive -- PolyaGammaAlt.cpp:37-49 */
static double a_coef_recursive(double n, double x, double h, double coef_h, double *gnh_over_gn1_gh)
{
  double d_n = 2.0 * (double)n + h;
  if (n != 0)
    *gnh_over_gn1_gh *= (n + h - 1) / n;
  else
    *gnh_over_gn1_gh = 1.0;
  double coef = coef_h * *gnh_over_gn1_gh;
  double log_kernel = -0.5 * (log(x * x * x) + d_n * d_n / x) + log(d_n);
  return coef * exp(log_kernel);
}

/* PolyaGammaAlt::pigauss(x, z, lambda) -- PolyaGammaAlt.cpp:51-58 */
static double alt_pigauss(double x, double z, double lambda)
{
  double b = sqrt(lambda / x) * (x * z - 1);
  double a = sqrt(lambda / x) * (x * z + 1) * -1.0;
  return bl_p_norm(b, 0) + exp(2 * lambda * z) * bl_p_norm(a, 0);
}

/* PolyaGammaAlt::w_left -- PolyaGammaAlt.cpp:60-68 */
double bl_alt_w_left(double trunc, double h, double z)
{
  double out;
  if (z != 0)
    out = exp(h * (log(2.0) - z)) * alt_pigauss(trunc, z / h, h * h);
  else
    out = exp(h * log(2.0)) * (1.0 - bl_p_gamma_rate(1 / trunc, 0.5, 0.5 * h * h));
  return out;
}

/* PolyaGammaAlt::w_right -- PolyaGammaAlt.cpp:70-75 */
double bl_alt_w_right(double trunc, double h, double z)
{
  double lambda_z = PISQ * 0.125 + 0.5 * z * z;
  return exp(h * log(HALFPI / lambda_z)) * (1.0 - bl_p_gamma_rate(trunc, h, lambda_z));
}

/* PolyaGammaAlt::rtigauss(h, z, trunc, r) -- PolyaGammaAlt.cpp:77-97 */
static double alt_rtigauss(double h, double z, double trunc, bl_rng *r)
{
  z = fabs(z);
  double mu = h / z;
  double X = trunc + 1.0;
  if (mu > trunc) {
    double alpha = 0.0;
    while (bl_unif(r) > alpha) {
      X = alt_rtinvchi2(h, trunc, r);
      alpha = exp(-0.5 * z * z * X);
    }
  } else {
    while (X > trunc)
      X = bl_igauss(r, mu, h * h);
  }
  return X;
}

/* PolyaGammaAlt::g_tilde(x, h, trunc) -- PolyaGammaAlt.cpp:99-108 */
double bl_alt_g_tilde(double x, double h, double trunc)
{
  double out;
  if (x > trunc)
    out = exp(h * log(0.5 * PG_PI) + (h - 1) * log(x) - PISQ * 0.125 * x - lgamma(h));
  else
    out = h * exp(h * log(2.0) - 0.5 * log(2.0 * PG_PI * x * x * x) - 0.5 * h * h / x);
  return out;
}

/* PolyaGammaAlt::draw_abridged(h, z, r, max_inner) -- PolyaGammaAlt.cpp:114-203 */
double bl_alt_draw_abridged(double h, double z, bl_rng *r, int max_inner)
{
  if (h < 1 || h > 4) return 0;                       /* :116-119 */
  z = fabs(z) * 0.5;
  int idx = (int)floor((h - 1.0) * 100.0);
  double trunc = bl_trunc_schedule[idx];
  double rate_z = 0.125 * PG_PI * PG_PI + 0.5 * z * z;
  double weight_left = bl_alt_w_left(trunc, h, z);
  double weight_right = bl_alt_w_right(trunc, h, z);
  double prob_right = weight_right / (weight_right + weight_left);
  double coef1_h = exp(h * log(2.0) - 0.5 * log(2.0 * PG_PI));
  double gnh_over_gn1_gh = 1.0;
  int num_trials = 0;
  while (num_trials < 10000) {
    num_trials++;
    double X, Y;
    double uu = bl_unif(r);
    if (uu < prob_right)
      X = bl_ltgamma(r, h, rate_z, trunc);
    else
      X = alt_rtigauss(h, z, trunc, r);
    double S = a_coef_recursive(0.0, X, h, coef1_h, &gnh_over_gn1_gh);
    double a_n = S;
    double gt = bl_alt_g_tilde(X, h, trunc);
    Y = bl_unif(r) * gt;
    int decreasing = 0;
    int n = 0;
    int go = 1;
    while (go && n < max_inner) {
      ++n;
      double prev = a_n;
      a_n = a_coef_recursive((double)n, X, h, coef1_h, &gnh_over_gn1_gh);
      decreasing = a_n <= prev;
      if (n % 2 == 1) {
        S = S - a_n;
        if (Y <= S && decreasing) return 0.25 * X;
      } else {
        S = S + a_n;
        if (Y > S && decreasing) go = 0;
      }
    }
  }
  return -1.0;                                        /* :202 */
}

/* PolyaGammaAlt::draw(h, z, r) -- PolyaGammaAlt.cpp:205-225 (forwards the
 * default max_inner = 200 to draw_abridged, :218-222). */
double bl_alt_draw(double h, double z, bl_rng *r)
{
  if (h < 1) return 0;
  double n = floor((h - 1.0) / 4.0);
  double remain = h - 4.0 * n;
  double x = 0.0;
  for (int i = 0; i < (int)n; i++)
    x += bl_alt_draw_abridged(4.0, z, r, 200);
  if (remain > 4.0)
    x += bl_alt_draw_abridged(0.5 * remain, z, r, 200) + bl_alt_draw_abridged(0.5 * remain, z, r, 200);
  else
    x += bl_alt_draw_abridged(remain, z, r, 200);
  return x;
}
