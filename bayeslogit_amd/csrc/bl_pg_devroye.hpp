// bl_pg_devroye.hpp -- device PG(1, z) draw by Devroye's J*(1, z/2) method, the
// integer-shape sum, the truncated sum of gammas and the closed-form moments.
// Behaviour follows Code/C/PolyaGamma.cpp (cited per function); the code is
// organised for one-draw-per-lane execution on 64-wide wavefronts.  gfx950 only.
#pragma once
#include "bl_rng.hpp"
#include "bl_specfun.hpp"

namespace bl {

constexpr double kTrunc = 0.64;                 // PolyaGamma.h:37
constexpr double kTruncRecip = 1.0 / 0.64;      // PolyaGamma.h:38
constexpr double kLogHalfPi = 0.45158270528945486472619522989488;  // log(pi/2)

enum : int { ST_OK = 0, ST_ITER_CAP = 1, ST_BAD_SHAPE = 2, ST_ALT_FALLTHROUGH = 4 };

// Series coefficient a_n(x), PolyaGamma.cpp:41-55.  logx = log(x) is passed in so
// the proposal's single log is shared by every term of the alternating series.
__device__ __forceinline__ double pg_a(int n, double x, double logx)
{
  const double nh = n + 0.5;
  const double K = nh * kPi;
  if (x > kTrunc) return K * exp(-0.5 * K * K * x);
  if (x > 0.0) return exp(-1.5 * (kLogHalfPi + logx) + log(K) - 2.0 * nh * nh / x);
  return 0.0;
}

// Mass of the exponential (right) piece of the proposal, PolyaGamma.cpp:65-80.
// Pure function of Z: evaluated once per draw instead of once per proposal.
__device__ inline double pg_mass_texpon(double Z, double fz)
{
  const double t = kTrunc;
  const double rt = sqrt(1.0 / t);
  const double b = rt * (t * Z - 1.0);
  const double a = rt * (t * Z + 1.0) * -1.0;
  const double x0 = log(fz) + fz * t;
  const double xb = x0 - Z + log_pnorm(b);
  const double xa = x0 + Z + log_pnorm(a);
  const double qdivp = 4.0 / kPi * (exp(xb) + exp(xa));
  return 1.0 / (1.0 + qdivp);
}

// IG(1/Z, 1) truncated to (0, 0.64], PolyaGamma.cpp:82-115.
__device__ inline double pg_rtigauss(double Z, Stream& r, int& status)
{
  const double t = kTrunc;
  double X = t + 1.0;
  if (kTruncRecip > Z) {
    double alpha = 0.0;
    int it = 0;
    while (r.unif() > alpha) {
      double E1 = r.expon(1.0);
      double E2 = r.expon(1.0);
      while (E1 * E1 > 2.0 * E2 / t) {
        E1 = r.expon(1.0);
        E2 = r.expon(1.0);
        if (++it > 1000000) { status |= ST_ITER_CAP; break; }
      }
      X = 1.0 + E1 * t;
      X = t / (X * X);
      alpha = exp(-0.5 * Z * Z * X);
      if (++it > 1000000) { status |= ST_ITER_CAP; break; }
    }
  } else {
    const double mu = 1.0 / Z;
    int it = 0;
    while (X > t) {
      double Y = r.norm(0.0, 1.0);
      Y *= Y;
      const double half_mu = 0.5 * mu;
      const double mu_Y = mu * Y;
      X = mu + half_mu * mu_Y - half_mu * sqrt(4.0 * mu_Y + mu_Y * mu_Y);
      if (r.unif() > mu / (mu + X)) X = mu * mu / X;
      if (++it > 1000000) { status |= ST_ITER_CAP; X = t; break; }
    }
  }
  return X;
}

// One PG(1, z) draw, PolyaGamma.cpp:151-202.  The reference's accept/reject and
// series loops are uncapped; a lane that never exits would hang its wavefront,
// so both are capped here and the cap is reported through `status`.
__device__ inline double pg_draw_like_devroye(double z, Stream& r, int& status)
{
  const double Z = fabs(z) * 0.5;
  const double fz = 0.125 * kPi * kPi + 0.5 * Z * Z;
  const double mass = pg_mass_texpon(Z, fz);
  double X = 0.0;
  for (int trial = 0; trial < 100000; ++trial) {
    if (r.unif() < mass)
      X = kTrunc + r.expon(1.0) / fz;
    else
      X = pg_rtigauss(Z, r, status);
    const double logx = log(X);
    double S = pg_a(0, X, logx);
    const double Y = r.unif() * S;
    int n = 0;
    bool go = true;
    while (go) {
      ++n;
      if (n & 1) {
        S = S - pg_a(n, X, logx);
        if (Y <= S) return 0.25 * X;
      } else {
        S = S + pg_a(n, X, logx);
        if (Y > S) go = false;
      }
      if (n > 100000) { status |= ST_ITER_CAP; return 0.25 * X; }
    }
  }
  status |= ST_ITER_CAP;
  return 0.25 * X;
}

// PolyaGamma::draw(int n, z, r), PolyaGamma.cpp:126-140 (NTHROW build clamps n<1 to 1).
__device__ inline double pg_draw_devroye(int n, double z, Stream& r, int& status)
{
  if (n < 1) { n = 1; status |= ST_BAD_SHAPE; }
  double sum = 0.0;
  for (int i = 0; i < n; ++i) sum += pg_draw_like_devroye(z, r, status);
  return sum;
}

// PolyaGamma::draw_sum_of_gammas, PolyaGamma.cpp:142-149 with bvec of :19-39.
__device__ inline double pg_draw_sum_of_gammas(double b, double z, int trunc, Stream& r)
{
  if (trunc < 1) trunc = 1;
  double x = 0.0;
  const double kappa = z * z;
  for (int k = 0; k < trunc; ++k) {
    const double d = (double)k + 0.5;
    const double bk = 4.0 * kPi * kPi * d * d;
    x += gamma_scale(r, b, 1.0) / (bk + kappa);
  }
  return 2.0 * x;
}

// PolyaGamma::jj_m1 / jj_m2 / pg_m1 / pg_m2, PolyaGamma.cpp:208-239.
__device__ __host__ inline double jj_m1(double b, double z)
{
  z = fabs(z);
  if (z > 1e-12) return b * tanh(z) / z;
  return b * (1 - (1.0 / 3) * pow(z, 2) + (2.0 / 15) * pow(z, 4) - (17.0 / 315) * pow(z, 6));
}
__device__ __host__ inline double jj_m2(double b, double z)
{
  z = fabs(z);
  if (z > 1e-12) return (b + 1) * b * pow(tanh(z) / z, 2) + b * ((tanh(z) - z) / pow(z, 3));
  return (b + 1) * b * pow(1 - (1.0 / 3) * pow(z, 2) + (2.0 / 15) * pow(z, 4) - (17.0 / 315) * pow(z, 6), 2) +
         b * ((-1.0 / 3) + (2.0 / 15) * pow(z, 2) - (17.0 / 315) * pow(z, 4));
}
__device__ __host__ inline double pg_m1(double b, double z) { return jj_m1(b, 0.5 * z) * 0.25; }
__device__ __host__ inline double pg_m2(double b, double z) { return jj_m2(b, 0.5 * z) * 0.0625; }

}  // namespace bl
