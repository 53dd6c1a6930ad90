/* rng.c -- oracle (test infrastructure, see bl_oracle.h): Philox4x32-10 counter
 * stream and the random primitives of the reference's absent RNG library.
 *
 * The reference draws every variate through `RNG& r` (jwindle/RNG, not in the
 * tree: INSTALL:14-33).  SURVEY.md Appendix B lists where the reference
 * states each primitive's semantics; those citations are repeated per function.
 * The stream itself (Philox) is ours: "parity unpinned" at stream level.
 */
#include "bl_oracle.h"
#include <math.h>

#define BL_PI 3.141592653589793238462643383279502884197

/* Philox4x32-10 (Salmon et al., SC'11): published algorithm, restated. */
void bl_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int round = 0; round < 10; ++round) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

uint64_t bl_chain_key(uint64_t seed, uint32_t call)
{
  uint32_t ctr[4] = {call, (uint32_t)BL_DOM_KEY << 24, 0, 0}, key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)}, out[4];
  bl_philox4x32_10(ctr, key, out);
  return ((uint64_t)out[1] << 32) | out[0];
}

void bl_rng_init(bl_rng *r, uint64_t seed, uint64_t idx, uint32_t domain, uint32_t epoch)
{
  r->key[0] = (uint32_t)seed;
  r->key[1] = (uint32_t)(seed >> 32);
  r->ctr[0] = (uint32_t)idx;
  r->ctr[1] = ((uint32_t)(idx >> 32) & 0x00FFFFFFu) | (domain << 24);
  r->ctr[2] = epoch;
  r->ctr[3] = 0;
  r->pos = 2;
  r->nunif = 0;
}

/* U(0,1) on the 52-bit grid (m + 1/2) 2^-52: exact in binary64, never 0 or 1.
 * r.unif() of the reference. */
double bl_unif(bl_rng *r)
{
  if (r->pos >= 2) {
    bl_philox4x32_10(r->ctr, r->key, r->buf);
    r->ctr[3] += 1;
    r->pos = 0;
  }
  uint32_t a = r->buf[2 * r->pos], b = r->buf[2 * r->pos + 1];
  r->pos += 1;
  r->nunif += 1;
  uint64_t m = (((uint64_t)a << 32) | b) >> 12;
  return ((double)m + 0.5) * 0x1.0p-52;
}

/* r.expon_rate(rate): rexp in Code/R/PG.R:86-90,152. */
double bl_expon_rate(bl_rng *r, double rate)
{
  return -log(bl_unif(r)) / rate;
}

/* r.norm(m, sd): Box-Muller on two uniforms (cosine branch only). */
double bl_norm(bl_rng *r, double mean, double sd)
{
  double u1 = bl_unif(r);
  double u2 = bl_unif(r);
  return mean + sd * sqrt(-2.0 * log(u1)) * cos(2.0 * BL_PI * u2);
}

/* r.flat(a,b): U(a,b).  Logit.hpp:375. */
double bl_flat(bl_rng *r, double a, double b)
{
  return a + (b - a) * bl_unif(r);
}

/* r.gamma_scale(shape, scale): rgamma in Code/R/PG.R:246-255.
 * Marsaglia & Tsang (2000); shape < 1 via the U^(1/shape) boost. */
double bl_gamma_scale(bl_rng *r, double shape, double scale)
{
  double boost = 1.0;
  double a = shape;
  if (a < 1.0) {
    boost = exp(log(bl_unif(r)) / a);
    a += 1.0;
  }
  double d = a - 1.0 / 3.0;
  double c = 1.0 / sqrt(9.0 * d);
  for (;;) {
    double x = bl_norm(r, 0.0, 1.0);
    double v = 1.0 + c * x;
    if (v <= 0.0) continue;
    v = v * v * v;
    double u = bl_unif(r);
    if (log(u) < 0.5 * x * x + d - d * v + d * log(v))
      return d * v * boost * scale;
  }
}

/* r.igauss(mu, lambda): Michael-Schucany-Haas, Code/R/PG.R:110-120,
 * Code/R/Ch.R:403-413. */
double bl_igauss(bl_rng *r, double mu, double lambda)
{
  double nu = bl_norm(r, 0.0, 1.0);
  double y = nu * nu;
  double muy = mu * y;
  double x = mu + 0.5 * mu * muy / lambda
           - 0.5 * mu / lambda * sqrt(4.0 * mu * lambda * y + muy * muy);
  if (bl_unif(r) > mu / (mu + x))
    x = mu * mu / x;
  return x;
}

/* r.ltgamma(shape, rate, trunc): Gamma(shape, rate) left-truncated at trunc.
 * Dagpunar's method as stated in Code/R/Ch.R:83-114. */
double bl_ltgamma(bl_rng *r, double shape, double rate, double trunc)
{
  double a = shape;
  double b = trunc * rate;
  if (a == 1.0)
    return trunc + bl_expon_rate(r, 1.0) / rate;
  double d1 = b - a;
  double d3 = a - 1.0;
  double c0 = 0.5 * (d1 + sqrt(d1 * d1 + 4.0 * b)) / b;
  double one_minus_c0 = 1.0 - c0;
  double x, log_m = d3 * (log(d3 / one_minus_c0) - 1.0);
  for (;;) {
    x = b + bl_expon_rate(r, 1.0) / c0;
    double log_rho = d3 * log(x) - x * one_minus_c0 - log_m;
    if (log(bl_unif(r)) <= log_rho) break;
  }
  return trunc * (x / b);
}

/* r.rtinvchi2(scale, trunc): scale/chi^2_1 restricted to (0, trunc].
 * Code/R/SPSample.R:534-550 states it as scale/E^2 with E a standard normal
 * left-truncated at 1/sqrt(trunc/scale); the in-tree C statement of the same
 * draw (pair of exponentials) is PolyaGammaAlt.cpp:6-22, which we follow. */
double bl_rtinvchi2(bl_rng *r, double scale, double trunc)
{
  double R = trunc / scale;
  double E1 = bl_expon_rate(r, 1.0);
  double E2 = bl_expon_rate(r, 1.0);
  while (E1 * E1 > 2.0 * E2 / R) {
    E1 = bl_expon_rate(r, 1.0);
    E2 = bl_expon_rate(r, 1.0);
  }
  double X = 1.0 + E1 * R;
  X = R / (X * X);
  return scale * X;
}

/* Phi^{-1}(p), lower tail: Wichura (1988) algorithm AS 241, routine PPND16
 * (published algorithm, relative accuracy ~1e-16).  Used by the fallback of bl_tnorm. */
double bl_qnorm(double p)
{
  double q = p - 0.5, r, val;
  if (fabs(q) <= 0.425) {
    r = 0.180625 - q * q;
    return q * (((((((r * 2509.0809287301226727 + 33430.575583588128105) * r + 67265.770927008700853) * r
                    + 45921.953931549871457) * r + 13731.693765509461125) * r + 1971.5909503065514427) * r
                 + 133.14166789178437745) * r + 3.387132872796366608)
           / (((((((r * 5226.495278852545925 + 28729.085735721942674) * r + 39307.89580009271061) * r
                  + 21213.794301586595867) * r + 5394.1960214247511077) * r + 687.1870074920579083) * r
               + 42.313330701600911252) * r + 1.0);
  }
  r = q < 0 ? p : 1.0 - p;
  r = sqrt(-log(r));
  if (r <= 5.0) {
    r -= 1.6;
    val = (((((((r * 7.7454501427834140764e-4 + 0.0227238449892691845833) * r + 0.24178072517745061177) * r
               + 1.27045825245236838258) * r + 3.64784832476320460504) * r + 5.7694972214606914055) * r
            + 4.6303378461565452959) * r + 1.42343711074968357734)
        / (((((((r * 1.05075007164441684324e-9 + 5.475938084995344946e-4) * r + 0.0151986665636164571966) * r
               + 0.14810397642748007459) * r + 0.68976733498510000455) * r + 1.6763848301838038494) * r
            + 2.05319162663775882187) * r + 1.0);
  } else {
    r -= 5.0;
    val = (((((((r * 2.01033439929228813265e-7 + 2.71155556874348757815e-5) * r + 0.0012426609473880784386) * r
               + 0.026532189526576123093) * r + 0.29656057182850489123) * r + 1.7848265399172913358) * r
            + 5.4637849111641143699) * r + 6.6579046435011037772)
        / (((((((r * 2.04426310338993978564e-15 + 1.4215117583164458887e-7) * r + 1.8463183175100546818e-5) * r
               + 7.868691311456132591e-4) * r + 0.0148753612908506148525) * r + 0.13692988092273580531) * r
            + 0.59983220655588793769) * r + 1.0);
  }
  return q < 0.0 ? -val : val;
}

/* upper-tail probability Q(x) = 1 - Phi(x), relative accuracy in the tail */
static double qtail(double x) { return 0.5 * erfc(x * M_SQRT1_2); }

/* X ~ N(0,1) | a <= X <= b for 0 <= a < b (b may be +inf), by inversion of the upper
 * tail: exact, one uniform.  Beyond a = 37 erfc underflows; there the tail is
 * exponential to within 1/a^2 and is inverted in closed form. */
static double tnorm_inv_right(double a, double b, double u)
{
  double x;
  if (a > 37.0) {
    double w = isinf(b) ? 1.0 : -expm1(-a * (b - a));
    x = a - log1p(-u * w) / a;
  } else {
    double qa = qtail(a), qb = isinf(b) ? 0.0 : qtail(b);
    double p = qa - u * (qa - qb);
    x = -bl_qnorm(p);
  }
  if (x < a) x = a;
  if (x > b) x = b;
  return x;
}

/* r.tnorm(lo, hi, 0, 1): standard normal restricted to (lo, hi), either bound may be
 * infinite.  Used by the constrained beta draw, Logit.hpp:393.  The reference's own
 * algorithm lives in the absent RNG library; this one is ours, designed so that a
 * call ALWAYS consumes exactly BL_TNORM_UNIFORMS = 9 uniforms whatever the bounds:
 * four rejection attempts (two uniforms each: Robert 1995 exponential-tail /
 * uniform / plain-normal proposals) and, if none accepts, one exact inverse-CDF
 * draw from the ninth.  Fixed consumption keeps the stream aligned even when the
 * bounds are rounding noise (chain sitting on the constraint boundary), so results
 * depend continuously on the bounds; it also lets a GPU pre-generate every random
 * input of a beta draw off the serial path. */
double bl_tnorm(bl_rng *r, double lo, double hi)
{
  double U[9];
  for (int k = 0; k < 9; ++k) U[k] = bl_unif(r);
  int lo_inf = isinf(lo) && lo < 0, hi_inf = isinf(hi) && hi > 0;
  if (!(hi - lo > 0.0)) return lo;                      /* empty / degenerate / NaN */
  /* (-inf, +inf) needs no special case: it is "wide", and its first Box-Muller attempt accepts */
  if (lo <= 0.0 && hi >= 0.0) {
    /* interval contains the mode */
    int wide = hi - lo > 2.5066282746310002;            /* sqrt(2 pi); true for infinite bounds */
    for (int k = 0; k < 4; ++k) {
      double ua = U[2 * k], ub = U[2 * k + 1];
      if (wide) {
        double x = sqrt(-2.0 * log(ua)) * cos(2.0 * BL_PI * ub);
        if (x >= lo && x <= hi) return x;
      } else {
        double x = lo + (hi - lo) * ua;
        if (log(ub) <= -0.5 * x * x) return x;
      }
    }
    double pl = lo_inf ? 0.0 : 0.5 * erfc(-lo * M_SQRT1_2);
    double ph = hi_inf ? 1.0 : 0.5 * erfc(-hi * M_SQRT1_2);
    double x = bl_qnorm(pl + U[8] * (ph - pl));
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return x;
  }
  /* interval on one side of zero: work on the positive side */
  int flip = hi < 0.0;
  double a = flip ? -hi : lo, b = flip ? -lo : hi, x = a;
  double alpha = 0.5 * (a + sqrt(a * a + 4.0));
  int tail = b - a > 1.0 / alpha;                       /* true for b = +inf */
  int done = 0;
  for (int k = 0; k < 4 && !done; ++k) {
    double ua = U[2 * k], ub = U[2 * k + 1];
    if (tail) {
      x = a - log(ua) / alpha;
      double d = x - alpha;
      done = x <= b && log(ub) <= -0.5 * d * d;
    } else {
      x = a + (b - a) * ua;
      done = log(ub) <= 0.5 * (a * a - x * x);
    }
  }
  if (!done) x = tnorm_inv_right(a, b, U[8]);
  return flip ? -x : x;
}
