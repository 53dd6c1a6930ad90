/* pg_devroye.c -- oracle (test infrastructure, see bl_oracle.h).
 * Restates Code/C/PolyaGamma.{h,cpp}: the Devroye-style J*(1,z) sampler, the
 * truncated sum of gammas, and the closed-form moments.
 */
#include "bl_oracle.h"
#include <math.h>

/* PolyaGamma.h:34-38 */
#define PG_PI     3.141592653589793238462643383279502884197
#define PG_TRUNC  0.64
#define PG_TRUNC_RECIP (1.0 / PG_TRUNC)

/* PolyaGamma::a(n, x) -- PolyaGamma.cpp:41-55 */
double bl_pg_a(int n, double x)
{
  double K = (n + 0.5) * PG_PI;
  double y = 0.0;
  if (x > PG_TRUNC) {
    y = K * exp(-0.5 * K * K * x);
  } else if (x > 0) {
    double expnt = -1.5 * (log(0.5 * PG_PI) + log(x)) + log(K) - 2.0 * (n + 0.5) * (n + 0.5) / x;
    y = exp(expnt);
  }
  return y;
}

/* PolyaGamma::mass_texpon(Z) -- PolyaGamma.cpp:65-80 */
double bl_pg_mass_texpon(double Z)
{
  double t = PG_TRUNC;
  double fz = 0.125 * PG_PI * PG_PI + 0.5 * Z * Z;
  double b = sqrt(1.0 / t) * (t * Z - 1);
  double a = sqrt(1.0 / t) * (t * Z + 1) * -1.0;
  double x0 = log(fz) + fz * t;
  double xb = x0 - Z + bl_p_norm(b, 1);
  double xa = x0 + Z + bl_p_norm(a, 1);
  double qdivp = 4 / PG_PI * (exp(xb) + exp(xa));
  return 1.0 / (1.0 + qdivp);
}

/* PolyaGamma::rtigauss(Z, r) -- PolyaGamma.cpp:82-115 */
double bl_pg_rtigauss(double Z, bl_rng *r)
{
  Z = fabs(Z);
  double t = PG_TRUNC;
  double X = t + 1.0;
  if (PG_TRUNC_RECIP > Z) {            /* mu > t */
    double alpha = 0.0;
    while (bl_unif(r) > alpha) {
      double E1 = bl_expon_rate(r, 1.0);
      double E2 = bl_expon_rate(r, 1.0);
      while (E1 * E1 > 2 * E2 / t) {
        E1 = bl_expon_rate(r, 1.0);
        E2 = bl_expon_rate(r, 1.0);
      }
      X = 1 + E1 * t;
      X = t / (X * X);
      alpha = exp(-0.5 * Z * Z * X);
    }
  } else {
    double mu = 1.0 / Z;
    while (X > t) {
      double Y = bl_norm(r, 0.0, 1.0);
      Y *= Y;
      double half_mu = 0.5 * mu;
      double mu_Y = mu * Y;
      X = mu + half_mu * mu_Y - half_mu * sqrt(4 * mu_Y + mu_Y * mu_Y);
      if (bl_unif(r) > mu / (mu + X))
        X = mu * mu / X;
    }
  }
  return X;
}

/* PolyaGamma::draw_like_devroye(Z, r) -- PolyaGamma.cpp:151-202.
 * mass_texpon(Z) is re-evaluated per proposal there (:170); it is a pure
 * function of Z so evaluating it once gives the identical value. */
double bl_pg_draw_like_devroye(double Z, bl_rng *r)
{
  Z = fabs(Z) * 0.5;
  double fz = 0.125 * PG_PI * PG_PI + 0.5 * Z * Z;
  double mass = bl_pg_mass_texpon(Z);
  double X = 0.0, S = 1.0, Y = 0.0;
  for (;;) {
    if (bl_unif(r) < mass)
      X = PG_TRUNC + bl_expon_rate(r, 1) / fz;
    else
      X = bl_pg_rtigauss(Z, r);
    S = bl_pg_a(0, X);
    Y = bl_unif(r) * S;
    int n = 0;
    int go = 1;
    while (go) {
      ++n;
      if (n % 2 == 1) {
        S = S - bl_pg_a(n, X);
        if (Y <= S) return 0.25 * X;
      } else {
        S = S + bl_pg_a(n, X);
        if (Y > S) go = 0;
      }
    }
  }
}

/* PolyaGamma::draw(int n, z, r) -- PolyaGamma.cpp:126-140 (NTHROW build:
 * n < 1 is clamped to 1, Makevars:11). */
double bl_pg_draw_devroye(int n, double z, bl_rng *r)
{
  if (n < 1) n = 1;
  double sum = 0.0;
  for (int i = 0; i < n; ++i)
    sum += bl_pg_draw_like_devroye(z, r);
  return sum;
}

/* PolyaGamma::draw_sum_of_gammas(n, z, r) -- PolyaGamma.cpp:142-149, with
 * bvec[k] = 4 pi^2 (k+1/2)^2 from set_trunc, :19-39 (trunc < 1 -> 1). */
double bl_pg_draw_sum_of_gammas(double b, double z, int trunc, bl_rng *r)
{
  if (trunc < 1) trunc = 1;
  double x = 0;
  double kappa = z * z;
  for (int k = 0; k < trunc; ++k) {
    double d = ((double)k + 0.5);
    double bk = 4 * PG_PI * PG_PI * d * d;
    x += bl_gamma_scale(r, b, 1.0) / (bk + kappa);
  }
  return 2.0 * x;
}

/* PolyaGamma::jj_m1 / jj_m2 / pg_m1 / pg_m2 -- PolyaGamma.cpp:208-239 */
double bl_jj_m1(double b, double z)
{
  z = fabs(z);
  double m1;
  if (z > 1e-12)
    m1 = b * tanh(z) / z;
  else
    m1 = b * (1 - (1.0 / 3) * pow(z, 2) + (2.0 / 15) * pow(z, 4) - (17.0 / 315) * pow(z, 6));
  return m1;
}

double bl_jj_m2(double b, double z)
{
  z = fabs(z);
  double m2;
  if (z > 1e-12)
    m2 = (b + 1) * b * pow(tanh(z) / z, 2) + b * ((tanh(z) - z) / pow(z, 3));
  else
    m2 = (b + 1) * b * pow(1 - (1.0 / 3) * pow(z, 2) + (2.0 / 15) * pow(z, 4) - (17.0 / 315) * pow(z, 6), 2)
       + b * ((-1.0 / 3) + (2.0 / 15) * pow(z, 2) - (17.0 / 315) * pow(z, 4));
  return m2;
}

double bl_pg_m1(double b, double z) { return bl_jj_m1(b, 0.5 * z) * 0.25; }
double bl_pg_m2(double b, double z) { return bl_jj_m2(b, 0.5 * z) * 0.0625; }
