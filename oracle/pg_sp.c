/* pg_sp.c -- oracle (test infrastructure, see bl_oracle.h).
 * Restates Code/C/InvertY.{hpp,cpp} and Code/C/PolyaGammaSP.{h,cpp}: the
 * saddle-point-approximation sampler for large shape.
 *
 * Hazard H5: the reference writes (1/3), (2/15), (17/315) with
 * integer operands, which are 0 in C++.  They are restated as the literal 0.0
 * they evaluate to, so the oracle computes what the compiled reference computes.
 */
#include "bl_oracle.h"
#include <math.h>

#define PG_PI 3.141592653589793238462643383279502884197
#define IY_TOL 1e-8            /* InvertY.hpp:8, the global `tol` */
#define H5_ONE_THIRD      0.0  /* (1/3)    InvertY.cpp:19,33; PolyaGammaSP.cpp:88,163,188 */
#define H5_TWO_FIFTEENTHS 0.0  /* (2/15)   */
#define H5_17_315         0.0  /* (17/315) */

/* y_eval(v) -- InvertY.cpp:10-21 */
double bl_y_eval(double v)
{
  double y;
  double r = sqrt(fabs(v));
  if (v > IY_TOL)
    y = tan(r) / r;
  else if (v < -1 * IY_TOL)
    y = tanh(r) / r;
  else
    y = 1 + H5_ONE_THIRD * v + H5_TWO_FIFTEENTHS * v * v + H5_17_315 * v * v * v;
  return y;
}

/* ydy_eval(v, &y, &dy) -- InvertY.cpp:23-35 */
void bl_ydy_eval(double v, double *yp, double *dyp)
{
  double y = bl_y_eval(v);
  *yp = y;
  if (fabs(v) >= IY_TOL)
    *dyp = 0.5 * (y * y + (1 - y) / v);
  else
    *dyp = 0.5 * (y * y - H5_ONE_THIRD - H5_TWO_FIFTEENTHS * v);
}

/* fdf_eval(v, &y, &f, &df): f = y(v) - y -- InvertY.cpp:43-48 */
void bl_fdf_eval(double v, double y, double *fp, double *dfp)
{
  bl_ydy_eval(v, fp, dfp);
  *fp -= y;
}

/* v_eval(y, tol=1e-9, max_iter=1000) -- InvertY.cpp:57-99 */
double bl_v_eval(double y)
{
  const double tol = 1e-9;
  const int max_iter = 1000;
  double ylower = bl_ygrid[0];
  double yupper = bl_ygrid[80];
  if (y < ylower) {
    return -1. / (y * y);
  } else if (y > yupper) {
    double v = atan(0.5 * y * PG_PI);
    return v * v;
  } else if (y == 1) return 0.0;

  double id = (log(y) / log(2.0) + 4.0) / 0.1;
  int idlow = (int)id;
  int idhigh = (int)id + 1;
  double vl = bl_vgrid[idlow];
  double vh = bl_vgrid[idhigh];
  int iter = 0;
  double diff = tol + 1.0;
  double vnew = vl, vold = vl;
  double f0, f1;
  while (diff > tol && iter < max_iter) {
    iter++;
    vold = vnew;
    bl_fdf_eval(vold, y, &f0, &f1);
    vnew = vold - f0 / f1;
    vnew = vnew > vh ? vh : vnew;
    vnew = vnew < vl ? vl : vnew;
    diff = fabs(vnew - vold);
  }
  return vnew;
}

/* PolyaGammaSP::rtigauss(mu, lambda, trunc, r) -- PolyaGammaSP.cpp:57-76 */
static double sp_rtigauss(double mu, double lambda, double trunc, bl_rng *r)
{
  double X = trunc + 1.0;
  if (trunc < mu) {
    double alpha = 0.0;
    while (bl_unif(r) > alpha) {
      X = bl_rtinvchi2(r, lambda, trunc);
      alpha = exp(-0.5 * lambda / (mu * mu) * X);
    }
  } else {
    while (X > trunc)
      X = bl_igauss(r, mu, lambda);
  }
  return X;
}

/* PolyaGammaSP::y_func(v) -- PolyaGammaSP.cpp:78-90 (tol 1e-6) */
double bl_sp_y_func(double v)
{
  double tol = 1e-6;
  double y;
  double r = sqrt(fabs(v));
  if (v > tol)
    y = tan(r) / r;
  else if (v < -1 * tol)
    y = tanh(r) / r;
  else
    y = 1 + H5_ONE_THIRD * v + H5_TWO_FIFTEENTHS * v * v + H5_17_315 * v * v * v;
  return y;
}

/* PolyaGammaSP::cos_rt(v) -- PolyaGammaSP.cpp:92-101 */
static double cos_rt(double v)
{
  double r = sqrt(fabs(v));
  return v >= 0 ? cos(r) : cosh(r);
}

/* delta_func, phi_func, tangent_to_eta -- PolyaGammaSP.cpp:103-146 */
void bl_sp_tangent_to_eta(double x, double z, double mid, double *slope, double *icept)
{
  /* phi_func :115-126 */
  double v = bl_v_eval(x);
  double u = 0.5 * v;
  double t = u + 0.5 * z * z;
  double phi_val = log(cosh(fabs(z))) - log(cos_rt(v)) - t * x;
  double phi_der = -1.0 * t;
  /* delta_func :103-113 */
  double delta_val, delta_der;
  if (x >= mid) {
    delta_val = log(x) - log(mid);
    delta_der = 1.0 / x;
  } else {
    delta_val = 0.5 * (1 - 1.0 / x) - 0.5 * (1 - 1.0 / mid);
    delta_der = 0.5 / (x * x);
  }
  double eta_val = phi_val - delta_val;
  double eta_der = phi_der - delta_der;
  *slope = eta_der;
  *icept = eta_val - eta_der * x;
}

/* PolyaGammaSP::sp_approx(x, n, z) -- PolyaGammaSP.cpp:148-167 */
double bl_sp_approx(double x, double n, double z)
{
  double v = bl_v_eval(x);
  double u = 0.5 * v;
  double z2 = z * z;
  double t = u + 0.5 * z2;
  double phi = log(cosh(z)) - log(cos_rt(v)) - t * x;
  double K2;
  if (fabs(v) >= 1e-6)
    K2 = x * x + (1 - x) / v;
  else
    K2 = x * x - H5_ONE_THIRD - H5_TWO_FIFTEENTHS * v;
  double log_spa = 0.5 * log(0.5 * n / PG_PI) - 0.5 * log(K2) + n * phi;
  return exp(log_spa);
}

/* PolyaGammaSP::draw(d, n, z, r, maxiter) -- PolyaGammaSP.cpp:169-264 */
int bl_sp_draw(double *d, double n, double z, bl_rng *r, int maxiter)
{
  z = 0.5 * fabs(z);
  double xl = bl_sp_y_func(-1 * z * z);
  double md = xl * 1.1;
  double xr = xl * 1.2;

  double vmd = bl_v_eval(md);
  double K2md;
  if (fabs(vmd) >= 1e-6)
    K2md = md * md + (1 - md) / vmd;
  else
    K2md = md * md - H5_ONE_THIRD - H5_TWO_FIFTEENTHS * vmd;
  double m2 = md * md;
  double al = m2 * md / K2md;
  double ar = m2 / K2md;

  double ls, li, rs, ri;
  bl_sp_tangent_to_eta(xl, z, md, &ls, &li);
  bl_sp_tangent_to_eta(xr, z, md, &rs, &ri);
  double rl = -1. * ls;
  double rr = -1. * rs;
  double il = li;
  double ir = ri;

  double lcn = 0.5 * log(0.5 * n / PG_PI);
  double rt2rl = sqrt(2 * rl);

  double wl = exp(0.5 * log(al) - n * rt2rl + n * il + 0.5 * n * 1. / md)
            * bl_p_igauss(md, 1. / rt2rl, n);
  double wr = exp(0.5 * log(ar) + lcn - n * log(n * rr) + n * ir - n * log(md))
            * tgamma(n) * (1.0 - bl_p_gamma_rate(md, n, n * rr));
  double wt = wl + wr;
  double pl = wl / wt;

  int go = 1;
  int iter = 0;
  double X = 2.0;
  double F = 0.0;
  while (go && iter < maxiter) {
    iter++;
    double phi_ev;
    if (bl_unif(r) < pl) {
      X = sp_rtigauss(1. / rt2rl, n, md, r);
      phi_ev = n * (il - rl * X) + 0.5 * n * ((1. - 1. / X) - (1. - 1. / md));
      F = exp(0.5 * log(al) + lcn - 1.5 * log(X) + phi_ev);
    } else {
      X = bl_ltgamma(r, n, n * rr, md);
      phi_ev = n * (ir - rr * X) + n * (log(X) - log(md));
      F = exp(0.5 * log(ar) + lcn + phi_ev) / X;
    }
    double spa = bl_sp_approx(X, n, z);
    if (F * bl_unif(r) < spa) go = 0;
  }
  *d = n * 0.25 * X;
  return iter;
}
