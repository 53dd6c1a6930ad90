// bl_pg_sp.hpp -- device saddle-point-approximation PG(n, z) sampler for large
// shape, with the y = tan(sqrt v)/sqrt v inversion it needs.  Behaviour follows
// Code/C/InvertY.cpp and Code/C/PolyaGammaSP.cpp (cited per function), including
// the integer-division literals (1/3), (2/15), (17/315) which evaluate to 0 in
// the compiled reference (SURVEY.md hazard H5): they are written as 0.0 here so
// results match what the reference computes.  gfx950 only.
#pragma once
#include "bl_pg_devroye.hpp"
#include "bl_tables.hpp"

namespace bl {

constexpr double kIyTol = 1e-8;          // InvertY.hpp:8 (global tol used by y_eval/ydy_eval)
constexpr double kH5Third = 0.0;         // (1/3)
constexpr double kH5TwoFifteenths = 0.0; // (2/15)
constexpr double kH5_17_315 = 0.0;       // (17/315)

// y_eval, InvertY.cpp:10-21
__device__ inline double iy_y_eval(double v)
{
  const double r = sqrt(fabs(v));
  if (v > kIyTol) return tan(r) / r;
  if (v < -kIyTol) return tanh(r) / r;
  return 1.0 + kH5Third * v + kH5TwoFifteenths * v * v + kH5_17_315 * v * v * v;
}

// v_eval(y, tol=1e-9, max_iter=1000), InvertY.cpp:57-99 (with ydy_eval/fdf_eval :23-48)
__device__ inline double iy_v_eval(double y)
{
  if (y < kYGrid[0]) return -1.0 / (y * y);
  if (y > kYGrid[80]) {
    const double v = atan(0.5 * y * kPi);
    return v * v;
  }
  if (y == 1.0) return 0.0;
  const double id = (log(y) / log(2.0) + 4.0) / 0.1;
  const int idlow = (int)id;
  const double vl = kVGrid[idlow];
  const double vh = kVGrid[idlow + 1];
  int iter = 0;
  double diff = 1e-9 + 1.0;
  // The reference starts Newton at the bracket's left end (InvertY.cpp:79); starting from the secant
  // through the bracket's ends instead saves about two tan/tanh evaluations and converges to the same
  // root (the stopping rule |dv| <= 1e-9 leaves an error of ~1e-18 either way).
  const double yl = kYGrid[idlow], yh = kYGrid[idlow + 1];
  double vnew = vl + (y - yl) * (vh - vl) / (yh - yl);
  vnew = vnew > vh ? vh : vnew;
  vnew = vnew < vl ? vl : vnew;
  double vold = vnew;
  while (diff > 1e-9 && iter < 1000) {
    iter++;
    vold = vnew;
    const double yv = iy_y_eval(vold);
    double dy;
    if (fabs(vold) >= kIyTol)
      dy = 0.5 * (yv * yv + (1.0 - yv) / vold);
    else
      dy = 0.5 * (yv * yv - kH5Third - kH5TwoFifteenths * vold);
    const double f0 = yv - y;
    vnew = vold - f0 / dy;
    vnew = vnew > vh ? vh : vnew;
    vnew = vnew < vl ? vl : vnew;
    diff = fabs(vnew - vold);
  }
  return vnew;
}

// PolyaGammaSP::rtigauss(mu, lambda, trunc, r), PolyaGammaSP.cpp:57-76
__device__ inline double sp_rtigauss(double mu, double lambda, double trunc, Stream& r, int& status)
{
  double X = trunc + 1.0;
  if (trunc < mu) {
    double alpha = 0.0;
    int it = 0;
    while (r.unif() > alpha) {
      X = rtinvchi2(r, lambda, trunc);
      alpha = exp(-0.5 * lambda / (mu * mu) * X);
      if (++it > 1000000) { status |= ST_ITER_CAP; break; }
    }
  } else {
    int it = 0;
    while (X > trunc) {
      X = igauss(r, mu, lambda);
      if (++it > 1000000) { status |= ST_ITER_CAP; X = trunc; break; }
    }
  }
  return X;
}

// y_func, PolyaGammaSP.cpp:78-90 (tol 1e-6)
__device__ inline double sp_y_func(double v)
{
  const double r = sqrt(fabs(v));
  if (v > 1e-6) return tan(r) / r;
  if (v < -1e-6) return tanh(r) / r;
  return 1.0 + kH5Third * v + kH5TwoFifteenths * v * v + kH5_17_315 * v * v * v;
}

// cos_rt, PolyaGammaSP.cpp:92-101
__device__ __forceinline__ double sp_cos_rt(double v)
{
  const double r = sqrt(fabs(v));
  return v >= 0.0 ? cos(r) : cosh(r);
}

// tangent_to_eta (phi_func + delta_func), PolyaGammaSP.cpp:103-146
// logcoshz = log(cosh(|z|)) is passed in (the same value serves both tangents and sp_approx).
__device__ inline void sp_tangent_to_eta(double x, double z, double mid, double logcoshz, double& slope, double& icept)
{
  const double v = iy_v_eval(x);
  const double t = 0.5 * v + 0.5 * z * z;
  const double phi_val = logcoshz - log(sp_cos_rt(v)) - t * x;
  const double phi_der = -1.0 * t;
  double delta_val, delta_der;
  if (x >= mid) {
    delta_val = log(x) - log(mid);
    delta_der = 1.0 / x;
  } else {
    delta_val = 0.5 * (1.0 - 1.0 / x) - 0.5 * (1.0 - 1.0 / mid);
    delta_der = 0.5 / (x * x);
  }
  const double eta_val = phi_val - delta_val;
  const double eta_der = phi_der - delta_der;
  slope = eta_der;
  icept = eta_val - eta_der * x;
}

// sp_approx(x, n, z), PolyaGammaSP.cpp:148-167.  log cosh z is hoisted by the caller.
__device__ inline double sp_approx(double x, double n, double z, double logcoshz, double lcn)
{
  const double v = iy_v_eval(x);
  const double t = 0.5 * v + 0.5 * (z * z);
  const double phi = logcoshz - log(sp_cos_rt(v)) - t * x;
  double K2;
  if (fabs(v) >= 1e-6)
    K2 = x * x + (1.0 - x) / v;
  else
    K2 = x * x - kH5Third - kH5TwoFifteenths * v;
  const double log_spa = lcn - 0.5 * log(K2) + n * phi;
  return exp(log_spa);
}

// draw(d, n, z, r, maxiter), PolyaGammaSP.cpp:169-264.  Returns the iteration count.
__device__ inline int sp_draw(double& d, double n, double z, Stream& r, int maxiter, int& status)
{
  if (n < 1.0) status |= ST_BAD_SHAPE;
  z = 0.5 * fabs(z);
  const double xl = sp_y_func(-1.0 * z * z);
  const double md = xl * 1.1;
  const double xr = xl * 1.2;
  const double vmd = iy_v_eval(md);
  double K2md;
  if (fabs(vmd) >= 1e-6)
    K2md = md * md + (1.0 - md) / vmd;
  else
    K2md = md * md - kH5Third - kH5TwoFifteenths * vmd;
  const double m2 = md * md;
  const double al = m2 * md / K2md;
  const double ar = m2 / K2md;

  const double logcoshz = log(cosh(z));
  double ls, li, rs, ri;
  sp_tangent_to_eta(xl, z, md, logcoshz, ls, li);
  sp_tangent_to_eta(xr, z, md, logcoshz, rs, ri);
  const double rl = -1.0 * ls;
  const double rr = -1.0 * rs;
  const double il = li;
  const double ir = ri;

  const double lcn = 0.5 * log(0.5 * n / kPi);
  const double rt2rl = sqrt(2.0 * rl);
  const double logmd = log(md);
  const double half_log_al = 0.5 * log(al);
  const double half_log_ar = 0.5 * log(ar);

  const double wl = exp(half_log_al - n * rt2rl + n * il + 0.5 * n * 1.0 / md) * p_igauss(md, 1.0 / rt2rl, n);
  // Gamma(n) enters through its logarithm, shared with the incomplete-gamma evaluation (one lgamma
  // instead of tgamma + lgamma; the weight changes by < 1e-13 relative)
  const double lgn = lgamma(n);
  const double wr = exp(half_log_ar + lcn - n * log(n * rr) + n * ir - n * logmd + lgn) *
                    (1.0 - p_gamma_rate(md, n, n * rr, lgn));
  const double pl = wl / (wl + wr);

  bool go = true;
  int iter = 0;
  double X = 2.0;
  while (go && iter < maxiter) {
    iter++;
    double F;
    if (r.unif() < pl) {
      X = sp_rtigauss(1.0 / rt2rl, n, md, r, status);
      const double phi_ev = n * (il - rl * X) + 0.5 * n * ((1.0 - 1.0 / X) - (1.0 - 1.0 / md));
      F = exp(half_log_al + lcn - 1.5 * log(X) + phi_ev);
    } else {
      X = ltgamma(r, n, n * rr, md);
      const double phi_ev = n * (ir - rr * X) + n * (log(X) - logmd);
      F = exp(half_log_ar + lcn + phi_ev) / X;
    }
    const double spa = sp_approx(X, n, z, logcoshz, lcn);
    if (F * r.unif() < spa) go = false;
  }
  d = n * 0.25 * X;
  return iter;
}

}  // namespace bl
