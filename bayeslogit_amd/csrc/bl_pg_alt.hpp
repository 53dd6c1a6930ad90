// bl_pg_alt.hpp -- device alternating-series PG(h, z) sampler for real h in [1,4],
// chained for larger h.  Behaviour follows Code/C/PolyaGammaAlt.cpp (cited per
// function).  gfx950 only.
#pragma once
#include "bl_pg_devroye.hpp"
#include "bl_tables.hpp"

namespace bl {

constexpr double kLog2 = 0.69314718055994530941723212145818;
constexpr double kPiSq = kPi * kPi;
constexpr double kHalfPi = 0.5 * kPi;

// free rtinvchi2(h, trunc, r), PolyaGammaAlt.cpp:6-22
__device__ inline double alt_rtinvchi2(double h, double trunc, Stream& r)
{
  const double h2 = h * h;
  return rtinvchi2(r, h2, trunc);   // same exponential-pair draw with scale = h^2
}

// a_n(x, h) by the running ratio Gamma(n+h)/(Gamma(n+1)Gamma(h)), PolyaGammaAlt.cpp:37-49.
// lx3 = log(x^3) is hoisted: it does not depend on n.
__device__ __forceinline__ double alt_a_coef_rec(double n, double x, double lx3, double h, double coef_h,
                                                 double& ratio)
{
  const double d_n = 2.0 * n + h;
  if (n != 0.0)
    ratio *= (n + h - 1.0) / n;
  else
    ratio = 1.0;
  const double coef = coef_h * ratio;
  const double log_kernel = -0.5 * (lx3 + d_n * d_n / x) + log(d_n);
  return coef * exp(log_kernel);
}

// pigauss(x, z, lambda), PolyaGammaAlt.cpp:51-58
__device__ inline double alt_pigauss(double x, double z, double lambda)
{
  const double s = sqrt(lambda / x);
  const double b = s * (x * z - 1.0);
  const double a = s * (x * z + 1.0) * -1.0;
  return pnorm(b) + exp(2.0 * lambda * z) * pnorm(a);
}

// w_left, PolyaGammaAlt.cpp:60-68
__device__ inline double alt_w_left(double trunc, double h, double z)
{
  if (z != 0.0) return exp(h * (kLog2 - z)) * alt_pigauss(trunc, z / h, h * h);
  return exp(h * kLog2) * (1.0 - p_gamma_rate(1.0 / trunc, 0.5, 0.5 * h * h));
}

// w_right, PolyaGammaAlt.cpp:70-75
__device__ inline double alt_w_right(double trunc, double h, double z, double lgam_h)
{
  const double lambda_z = kPiSq * 0.125 + 0.5 * z * z;
  return exp(h * log(kHalfPi / lambda_z)) * (1.0 - p_gamma_rate(trunc, h, lambda_z, lgam_h));
}

// rtigauss(h, z, trunc, r), PolyaGammaAlt.cpp:77-97
__device__ inline double alt_rtigauss(double h, double z, double trunc, Stream& r, int& status)
{
  z = fabs(z);
  const double mu = h / z;
  double X = trunc + 1.0;
  if (mu > trunc) {
    double alpha = 0.0;
    int it = 0;
    while (r.unif() > alpha) {
      X = alt_rtinvchi2(h, trunc, r);
      alpha = exp(-0.5 * z * z * X);
      if (++it > 1000000) { status |= ST_ITER_CAP; break; }
    }
  } else {
    int it = 0;
    while (X > trunc) {
      X = igauss(r, mu, h * h);
      if (++it > 1000000) { status |= ST_ITER_CAP; X = trunc; break; }
    }
  }
  return X;
}

// g_tilde(x, h, trunc), PolyaGammaAlt.cpp:99-108
__device__ inline double alt_g_tilde(double x, double logx, double h, double trunc, double lgam_h)
{
  if (x > trunc) return exp(h * log(0.5 * kPi) + (h - 1.0) * logx - kPiSq * 0.125 * x - lgam_h);
  return h * exp(h * kLog2 - 0.5 * log(2.0 * kPi * x * x * x) - 0.5 * h * h / x);
}

// The part of draw_abridged that depends on (h, z) only, PolyaGammaAlt.cpp:117-140: it draws nothing,
// so a draw that calls draw_abridged several times with the same (h, z) (PolyaGammaAlt.cpp:216-222)
// evaluates it once.
struct AltSetup {
  double h, z, trunc, rate_z, prob_right, coef1_h, lgam_h;
  bool ok;
};

__device__ inline AltSetup alt_setup(double h, double z, int& status)
{
  AltSetup s;
  s.ok = !(h < 1.0 || h > 4.0);
  if (!s.ok) {
    status |= ST_BAD_SHAPE;
    return s;
  }
  s.h = h;
  s.z = fabs(z) * 0.5;
  const int idx = (int)floor((h - 1.0) * 100.0);
  s.trunc = kTruncSchedule[idx];
  s.rate_z = 0.125 * kPi * kPi + 0.5 * s.z * s.z;
  s.lgam_h = lgamma(h);
  const double weight_left = alt_w_left(s.trunc, h, s.z);
  const double weight_right = alt_w_right(s.trunc, h, s.z, s.lgam_h);
  s.prob_right = weight_right / (weight_right + weight_left);
  s.coef1_h = exp(h * kLog2 - 0.5 * log(2.0 * kPi));
  return s;
}

// draw_abridged(h, z, r, max_inner), PolyaGammaAlt.cpp:114-203: the trial loop (:142-201)
__device__ inline double alt_draw_abridged(const AltSetup& s, Stream& r, int max_inner, int& status)
{
  if (!s.ok) return 0.0;
  const double h = s.h, z = s.z, trunc = s.trunc;
  double ratio = 1.0;

  for (int trial = 0; trial < 10000; ++trial) {
    double X;
    const double uu = r.unif();
    if (uu < s.prob_right)
      X = ltgamma(r, h, s.rate_z, trunc);
    else
      X = alt_rtigauss(h, z, trunc, r, status);
    const double logx = log(X);
    const double lx3 = log(X * X * X);
    double S = alt_a_coef_rec(0.0, X, lx3, h, s.coef1_h, ratio);
    double a_n = S;
    const double gt = alt_g_tilde(X, logx, h, trunc, s.lgam_h);
    const double Y = r.unif() * gt;
    int n = 0;
    bool go = true;
    while (go && n < max_inner) {
      ++n;
      const double prev = a_n;
      a_n = alt_a_coef_rec((double)n, X, lx3, h, s.coef1_h, ratio);
      const bool decreasing = a_n <= prev;
      if (n & 1) {
        S = S - a_n;
        if (Y <= S && decreasing) return 0.25 * X;
      } else {
        S = S + a_n;
        if (Y > S && decreasing) go = false;
      }
    }
  }
  status |= ST_ALT_FALLTHROUGH;
  return -1.0;   // PolyaGammaAlt.cpp:202
}

// draw(h, z, r), PolyaGammaAlt.cpp:205-225
__device__ inline double alt_draw(double h, double z, Stream& r, int& status)
{
  if (h < 1.0) { status |= ST_BAD_SHAPE; return 0.0; }
  const double n = floor((h - 1.0) / 4.0);
  const double remain = h - 4.0 * n;
  double x = 0.0;
  if ((int)n > 0) {
    const AltSetup s4 = alt_setup(4.0, z, status);
    for (int i = 0; i < (int)n; i++) x += alt_draw_abridged(s4, r, 200, status);
  }
  if (remain > 4.0) {
    const AltSetup sh = alt_setup(0.5 * remain, z, status);
    const double a = alt_draw_abridged(sh, r, 200, status);
    const double b = alt_draw_abridged(sh, r, 200, status);
    x += a + b;
  } else {
    const AltSetup sr = alt_setup(remain, z, status);
    x += alt_draw_abridged(sr, r, 200, status);
  }
  return x;
}

}  // namespace bl
