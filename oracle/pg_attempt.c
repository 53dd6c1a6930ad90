/* pg_attempt.c -- oracle (test infrastructure, see bl_oracle.h).
 *
 * The alternating-series sampler (Code/C/PolyaGammaAlt.cpp:114-225) and the saddle-point
 * sampler (Code/C/PolyaGammaSP.cpp:169-264) in the ATTEMPT FORM the HIP kernels execute, next
 * to their literal restatements in pg_alt.c / pg_sp.c.  Same mixture proposals, same
 * acceptance events with the same probabilities; what changes is how the observation's
 * uniforms are spent: one attempt = one Philox block = two uniforms (u1, u2).
 *
 *   u1  picks the piece on a fresh proposal and is recycled (conditional on {u1 < p}, u1/p is
 *       uniform and independent of the event) into the proposal variate: E = -log(w) for the
 *       exponential-driven pieces, N(0,1) = qnorm(w) for the inverse-Gaussian candidate;
 *   u2  decides the accept/reject events of the attempt through nested thresholds:
 *         left-truncated gamma (Dagpunar; Code/R/Ch.R:83-114): u2 <= rho(x), else a new
 *             exponential (state RIGHT: the piece is kept, as the reference's inner loop does);
 *         inverse-chi-square pair + exp(-z^2 X/2) (PolyaGammaAlt.cpp:6-22, :82-88): the pair
 *             test holds with probability exp(-R E1^2/2), so one comparison u2 <= A decides
 *             both, else a new pair (state LEFT);
 *         inverse-Gaussian candidate (RNG::igauss): reciprocal flip on u2 vs mu/(mu + x0),
 *             retry while X > trunc (state LEFT);
 *       and the remainder of u2 (u2/rho, u2/A, or what the flip left) is the uniform of the
 *       final test: the alternating series (Alt) or F U < sp_approx (SP).
 *
 * tests/test_oracle_attempt.py pins each attempt form to its literal form's distribution
 * (two-sample KS, moments) and the HIP parity tests compare with the attempt form draw for
 * draw.  Where the attempt form evaluates a quantity by a different formula than the literal
 * code (weights without cancellation, v(x) from a fitted table instead of Newton), the two
 * formulas are compared value for value in the same test file.
 */
#include "bl_oracle.h"
#include <math.h>

#define PG_PI 3.141592653589793238462643383279502884197
#define WMIN 0x1.0p-53
#define WMAX (1.0 - 0x1.0p-53)

static double clamp01(double w)
{
  if (w < WMIN) w = WMIN;
  if (w > WMAX) w = WMAX;
  return w;
}

/* Gamma(a, x) / (exp(-x) x^a): modified Lentz continued fraction (published algorithm,
 * Numerical Recipes 6.2), converges for every x > 0, fast for x > a + 1. */
double bl_upper_gamma_cf(double a, double x)
{
  const double tiny = 1e-300;
  double b = x + 1.0 - a, c = 1.0 / tiny, d = 1.0 / b, h = d;
  for (int i = 1; i < 5000; ++i) {
    double an = -(double)i * ((double)i - a);
    b += 2.0;
    d = an * d + b; if (fabs(d) < tiny) d = tiny;
    c = b + an / c; if (fabs(c) < tiny) c = tiny;
    d = 1.0 / d;
    double del = d * c;
    h *= del;
    if (fabs(del - 1.0) < 1e-16) break;
  }
  return h;
}

/* Dagpunar's constants for Gamma(a, 1) left-truncated at b (Code/R/Ch.R:83-114) */
static void dagpunar(double a, double b, double *ic0, double *omc, double *log_m)
{
  if (a == 1.0) { *ic0 = 1.0; *omc = 0.0; *log_m = 0.0; return; }
  double d1 = b - a, d3 = a - 1.0;
  double c0 = 0.5 * (d1 + sqrt(d1 * d1 + 4.0 * b)) / b;
  *ic0 = 1.0 / c0;
  *omc = 1.0 - c0;
  *log_m = d3 * (log(d3 / *omc) - 1.0);
}

/* ============================================================== alternating series */

/* (h, z)-only part of draw_abridged, PolyaGammaAlt.cpp:117-140 */
void bl_alt_par_of(bl_alt_par *p, double h, double z)
{
  p->h = h;
  p->Z = fabs(z) * 0.5;                                         /* :122 */
  int idx = (int)floor((h - 1.0) * 100.0);                      /* :124 */
  p->t = bl_trunc_schedule[idx];
  p->fz = 0.125 * PG_PI * PG_PI + 0.5 * p->Z * p->Z;            /* :128 */
  double wl = bl_alt_w_left(p->t, h, p->Z);                     /* :129 */
  /* w_right (:70-75) = (pi/2 / fz)^h Q(h, fz t) with Q from the continued fraction / closed form
   * instead of 1 - P (no cancellation) */
  double x = p->fz * p->t;
  double q = exp(-x + h * log(x) - lgamma(h)) * bl_upper_gamma_cf(h, x);
  double wr = exp(h * log(0.5 * PG_PI / p->fz)) * q;            /* :130 */
  p->p = wr / (wr + wl);                                        /* :131 */
  p->ip = 1.0 / p->p;
  p->iq = 1.0 / (1.0 - p->p);
  p->small = !(p->Z * p->t >= h);                               /* mu = h/Z > t, :81 (Z = 0: mu = inf) */
  p->R = p->t / (h * h);                                        /* rtinvchi2 :9 */
  p->b = p->t * p->fz;                                          /* ltgamma: b = trunc * rate */
  dagpunar(h, p->b, &p->ic0, &p->omc, &p->log_m);
  /* log of a_0/g_tilde right of t without its x-dependent part: (4/pi)^h Gamma(h+1) / sqrt(2 pi) */
  p->cR = h * log(4.0 / PG_PI) + lgamma(h + 1.0) - 0.5 * log(2.0 * PG_PI);
}

/* the alternating series test of PolyaGammaAlt.cpp:156-196 on a_n/g_tilde:
 * s0 = a_0(X)/g_tilde(X), v = the uniform.  a_n/a_{n-1} = ((n+h-1)/n) ((2n+h)/(2n+h-2)) exp(-2(2n+h-1)/X).
 * Returns 1 accept, 0 reject (also when max_inner terms decide nothing: the reference starts a new trial). */
static int alt_series(double X, double h, double s0, double v, int max_inner)
{
  double iX = 1.0 / X;
  double e = exp(-2.0 * (h + 1.0) * iX);
  double q2 = exp(-4.0 * iX);
  double a_prev = s0, S = s0;
  for (int n = 1; n <= max_inner; ++n) {
    double dn = (double)n;
    double rn = ((dn + h - 1.0) * (2.0 * dn + h)) / (dn * (2.0 * dn + h - 2.0)) * e;
    double a = a_prev * rn;
    int decreasing = a <= a_prev;
    if (n & 1) {
      S -= a;
      if (v <= S && decreasing) return 1;
    } else {
      S += a;
      if (v > S && decreasing) return 0;
    }
    a_prev = a;
    e *= q2;
  }
  return 0;
}

/* one attempt.  *state: 0 fresh, 1 inside the left piece, 2 inside the right piece.
 * Returns 1 when an abridged draw completed (value 0.25 * *X). */
int bl_alt_attempt(int *state, double *X, const bl_alt_par *p, double u1, double u2)
{
  const double h = p->h, t = p->t;
  int right;
  double w;
  if (*state == 0) {
    right = u1 < p->p;                                          /* :149 */
    w = right ? u1 * p->ip : (u1 - p->p) * p->iq;
  } else {
    right = *state == 2;
    w = u1;
  }
  w = clamp01(w);
  double Xc, v, s0;
  if (right) {                                                  /* r.ltgamma(h, rate_z, trunc), :150 */
    double E = -log(w);
    if (h == 1.0) {
      Xc = t + E / p->fz;
      v = u2;
    } else {
      double x = p->b + E * p->ic0;
      double rho = exp((h - 1.0) * log(x) - x * p->omc - p->log_m);
      if (u2 > rho) { *state = 2; return 0; }
      v = clamp01(u2 / rho);
      Xc = x / p->fz;                                           /* trunc * (x / b) */
    }
    s0 = exp(p->cR - (h + 0.5) * log(Xc) + 0.125 * PG_PI * PG_PI * Xc - 0.5 * h * h / Xc);
  } else if (p->small) {                                        /* rtigauss, mu > t: :81-89 with :6-22 */
    double E1 = -log(w);
    double d = 1.0 + E1 * p->R;
    Xc = t / (d * d);
    double A = exp(-0.5 * (p->R * E1 * E1 + p->Z * p->Z * Xc));
    if (u2 > A) { *state = 1; return 0; }
    v = clamp01(u2 / A);
    s0 = 1.0;
  } else {                                                      /* rtigauss, mu <= t: :91-94 */
    double mu = h / p->Z, lam = h * h;
    double nu = bl_qnorm(w);
    double y = nu * nu;
    double muy = mu * y;
    double x0 = mu + 0.5 * mu * muy / lam - 0.5 * mu / lam * sqrt(4.0 * mu * lam * y + muy * muy);
    double pk = mu / (mu + x0);
    int flip = u2 > pk;
    Xc = flip ? mu * mu / x0 : x0;
    if (Xc > t) { *state = 1; return 0; }
    v = clamp01(flip ? (u2 - pk) / (1.0 - pk) : u2 / pk);
    s0 = 1.0;
  }
  *state = 0;
  *X = Xc;
  return alt_series(Xc, h, s0, v, 200);
}

/* draw_abridged in the attempt form; *nblk += blocks consumed */
static double alt_abridged_attempt(const bl_alt_par *p, bl_rng *r)
{
  int state = 0, trials = 0;
  double X = 0.0;
  for (;;) {
    if (state == 0 && ++trials > 10000) return -1.0;            /* :142, :202 */
    double u1 = bl_unif(r), u2 = bl_unif(r);
    if (bl_alt_attempt(&state, &X, p, u1, u2)) return 0.25 * X;
  }
}

/* PolyaGammaAlt::draw(h, z, r), :205-225, on the attempt form.  The floor((h-1)/4) draws at shape 4 read the
 * observation's stream from block 0, the remainder's draw(s) from block 2^31: the HIP path draws the two
 * groups as separate tasks (possibly on different lanes) and adds the two sums. */
double bl_alt_draw_attempt(double h, double z, bl_rng *r)
{
  if (h < 1) return 0;
  double n = floor((h - 1.0) / 4.0);
  double remain = h - 4.0 * n;
  double x = 0.0;
  bl_alt_par p;
  if ((int)n > 0) {
    bl_alt_par_of(&p, 4.0, z);
    for (int i = 0; i < (int)n; i++) x += alt_abridged_attempt(&p, r);
  }
  r->ctr[3] = 0x80000000u;
  r->pos = 2;
  if (remain > 4.0) {
    bl_alt_par_of(&p, 0.5 * remain, z);
    double a = alt_abridged_attempt(&p, r);
    double b = alt_abridged_attempt(&p, r);
    x += a + b;
  } else {
    bl_alt_par_of(&p, remain, z);
    x += alt_abridged_attempt(&p, r);
  }
  return x;
}

/* ==================================================================== saddle point */

/* v(x), -log cos_rt(v(x)), log K2(x) for x in [2^-4, 2^4] from the fitted table (oracle/vtab.c,
 * scripts/gen_vtab.py); outside that range the reference's asymptotic forms (InvertY.cpp:62-68)
 * and the literal cos_rt / K2 (PolyaGammaSP.cpp:92-101, :159-163). */
void bl_sp_vlk(double x, double logx, double *v, double *L, double *lK2)
{
  double s = logx * 1.4426950408889634074;
  if (s >= -4.0 && s <= 4.0) {
    double u = 2.0 * (s + 4.0);
    int k = (int)u;
    if (k > 15) k = 15;
    double tau = 2.0 * (u - (double)k) - 1.0;
    double acc[3];
    for (int f = 0; f < 3; ++f) {
      const double *c = bl_vtab[f][k];
      double a = c[10];
      for (int j = 9; j >= 0; --j) a = a * tau + c[j];
      acc[f] = a;
    }
    *v = s * acc[0];
    *L = s * acc[1];
    *lK2 = acc[2];
    if (fabs(*v) < 1e-6) *lK2 = 2.0 * logx;          /* K2 = x^2 - (1/3) - (2/15) v with integer literals, :163 */
    return;
  }
  double vv;
  if (s < -4.0) {
    vv = -1.0 / (x * x);
  } else {
    vv = atan(0.5 * x * PG_PI);
    vv = vv * vv;
  }
  double r = sqrt(fabs(vv));
  *v = vv;
  *L = -log(vv >= 0 ? cos(r) : cosh(r));
  *lK2 = log(x * x + (1.0 - x) / vv);
}

/* (n, z)-only part of PolyaGammaSP::draw, :171-229 */
void bl_sp_par_of(bl_sp_par *p, double n, double z)
{
  double Z = 0.5 * fabs(z);                                     /* :172 */
  double Z2 = Z * Z;
  double e2 = exp(-2.0 * Z);
  double xl = Z2 > 1e-6 ? (1.0 - e2) / ((1.0 + e2) * Z) : 1.0;  /* y_func(-z^2), :78-90 */
  double lcZ = Z + log1p(e2) - log(2.0);                        /* log cosh Z */
  double md = xl * 1.1, xr = xl * 1.2;                          /* :175-176 */
  double logxl = log(xl);
  double logmd = logxl + log(1.1), logxr = logxl + log(1.2);
  p->n = n; p->Z2 = Z2; p->md = md; p->logmd = logmd; p->lcZ = lcZ;
  double vmd, Lmd, lK2md, vr, Lr, lK2r;
  bl_sp_vlk(md, logmd, &vmd, &Lmd, &lK2md);                     /* :182-188 */
  bl_sp_vlk(xr, logxr, &vr, &Lr, &lK2r);
  p->lhal = 0.5 * (3.0 * logmd - lK2md);                        /* 0.5 log(md^3 / K2md), :190 */
  p->lhar = 0.5 * (2.0 * logmd - lK2md);                        /* :191 */
  /* tangent to eta at xl (:197): v(xl) = -Z^2 exactly (xl = y_func(-Z^2)), so t = 0 and phi(xl) = 0;
   * when y_func returned 1 (Z^2 <= 1e-6) v_eval(1) = 0 */
  double vl = Z2 > 1e-6 ? -Z2 : 0.0;
  double tl = 0.5 * vl + 0.5 * Z2;
  double phil = Z2 > 1e-6 ? 0.0 : lcZ - tl * xl;
  p->rl = tl + 0.5 / (xl * xl);                                 /* -(phi' - delta'), delta' = 0.5/x^2 left of md */
  p->il = phil - 0.5 * (1.0 / md - 1.0 / xl) + p->rl * xl;      /* eta(xl) - eta'(xl) xl, :144 */
  /* tangent at xr (:198): delta = log(xr) - log(md), delta' = 1/xr */
  double tr = 0.5 * vr + 0.5 * Z2;
  double phir = lcZ + Lr - tr * xr;
  p->rr = tr + 1.0 / xr;
  p->ir = phir - (logxr - logmd) + p->rr * xr;
  double rt2rl = sqrt(2.0 * p->rl);                             /* :210 */
  p->mu = 1.0 / rt2rl;
  /* wl (:217-218): p_igauss(md, mu, n) = 1 - exp(-A1^2/2) [erfcx(A1/sqrt2) - erfcx(A2/sqrt2)] / 2,
   * A1 = sqrt(n/md)(md/mu - 1), A2 = sqrt(n/md)(md/mu + 1): the two exponents of the textbook formula
   * are both -A1^2/2.  Here through erfc (the oracle has no erfcx): 1 - erfc(a1)/2 + exp(2n/mu) erfc(a2)/2
   * with the second term in logs. */
  double s = sqrt(n / md);
  double A1 = s * (md / p->mu - 1.0), A2 = s * (md / p->mu + 1.0);
  double pig = 1.0 - 0.5 * erfc(A1 * M_SQRT1_2) + exp(2.0 * n / p->mu + bl_p_norm(-A2, 1));
  double lwl = p->lhal + n * (p->il - rt2rl + 0.5 / md) + log(pig);
  /* wr (:220-222): Gamma(n) Q(n, x) = exp(-x) x^n CF(n, x), x = n rr md, so that
   * -n log(n rr) - n log(md) + n log(x) = 0 and wr = sqrt(ar) sqrt(n / 2 pi) exp(n ir - x) CF */
  double x = n * p->rr * md;
  double lcn = 0.5 * log(0.5 * n / PG_PI);
  double lwr = p->lhar + lcn + n * p->ir - x + log(bl_upper_gamma_cf(n, x));
  p->pl = 1.0 / (1.0 + exp(lwr - lwl));                         /* wl / (wl + wr), :226-227 */
  p->ipl = 1.0 / p->pl;
  p->iql = 1.0 / (1.0 - p->pl);
  p->b = x;                                                     /* ltgamma(n, n rr, md): b = trunc * rate */
  dagpunar(n, p->b, &p->ic0, &p->omc, &p->log_m);
}

/* one attempt.  *state as in bl_alt_attempt.  Returns 1 when an outer iteration of :235-259 has
 * ended (a proposal X has met the test F U < sp_approx); *accepted says how. */
int bl_sp_attempt(int *state, double *X, int *accepted, const bl_sp_par *p, double u1, double u2)
{
  const double n = p->n, md = p->md;
  int left;
  double w;
  if (*state == 0) {
    left = u1 < p->pl;                                          /* :243 */
    w = left ? u1 * p->ipl : (u1 - p->pl) * p->iql;
  } else {
    left = *state == 1;
    w = u1;
  }
  w = clamp01(w);
  double Xc, v, logX, logF;
  if (left) {                                                   /* rtigauss(mu, n, md), :244 / :57-76, mu <= md */
    double mu = p->mu;
    double nu = bl_qnorm(w);
    double y = nu * nu;
    double muy = mu * y;
    double x0 = mu + 0.5 * mu * muy / n - 0.5 * mu / n * sqrt(4.0 * mu * n * y + muy * muy);
    double pk = mu / (mu + x0);
    int flip = u2 > pk;
    Xc = flip ? mu * mu / x0 : x0;
    if (Xc > md) { *state = 1; return 0; }
    v = clamp01(flip ? (u2 - pk) / (1.0 - pk) : u2 / pk);
    logX = log(Xc);
    logF = p->lhal - 1.5 * logX + n * (p->il - p->rl * Xc) + 0.5 * n * (1.0 / md - 1.0 / Xc);   /* :245-246 */
  } else {                                                      /* r.ltgamma(n, n rr, md), :250 */
    double E = -log(w);
    if (n == 1.0) {
      Xc = md + E / p->rr;
      v = u2;
    } else {
      double x = p->b + E * p->ic0;
      double rho = exp((n - 1.0) * log(x) - x * p->omc - p->log_m);
      if (u2 > rho) { *state = 2; return 0; }
      v = clamp01(u2 / rho);
      Xc = md * (x / p->b);
    }
    logX = log(Xc);
    logF = p->lhar + n * (p->ir - p->rr * Xc) + n * (logX - p->logmd) - logX;                    /* :251-252 */
  }
  double vv, L, lK2;
  bl_sp_vlk(Xc, logX, &vv, &L, &lK2);
  double phi = p->lcZ + L - (0.5 * vv + 0.5 * p->Z2) * Xc;      /* :157 */
  double logspa = -0.5 * lK2 + n * phi;                         /* :165 without lcn (it cancels against F's) */
  *state = 0;
  *X = Xc;
  *accepted = v < exp(logspa - logF);                           /* F U < spa, :257 */
  return 1;
}

/* PolyaGammaSP::draw in the attempt form: returns iter, :235-263 */
int bl_sp_draw_attempt(double *d, double n, double z, bl_rng *r, int maxiter)
{
  bl_sp_par p;
  bl_sp_par_of(&p, n, z);
  int state = 0, iter = 0, acc = 0;
  double X = 2.0;
  for (long blk = 0; blk < 4000000; ++blk) {
    if (state == 0) {
      if (iter >= maxiter) break;
      iter++;
    }
    double u1 = bl_unif(r), u2 = bl_unif(r);
    if (bl_sp_attempt(&state, &X, &acc, &p, u1, u2) && acc) break;
  }
  *d = n * 0.25 * X;
  return iter;
}

/* rpg_hybrid dispatch (LogitWrapper.cpp:140-162) on the attempt forms: what the HIP path computes */
double bl_pg_hybrid_attempt(double b, double z, bl_rng *r)
{
  double x;
  if (b > 170) {
    double m = bl_pg_m1(b, z);
    double v = bl_pg_m2(b, z) - m * m;
    x = bl_norm(r, m, sqrt(v));
  } else if (b > 13) {
    bl_sp_draw_attempt(&x, b, z, r, 200);
  } else if (b == 1 || b == 2) {
    x = bl_pg_draw_devroye((int)b, z, r);
  } else if (b > 1) {
    x = bl_alt_draw_attempt(b, z, r);
  } else if (b > 0) {
    x = bl_pg_draw_sum_of_gammas(b, z, 200, r);
  } else {
    x = 0.0;
  }
  return x;
}
