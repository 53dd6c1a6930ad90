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

/* X ~ N(0,1) | X >= a.  Robert (1995) exponential rejection for a > 0. */
static double tail_norm(bl_rng *r, double a)
{
  if (a <= 0.0) {
    double x;
    do { x = bl_norm(r, 0.0, 1.0); } while (x < a);
    return x;
  }
  double alpha = 0.5 * (a + sqrt(a * a + 4.0));
  for (;;) {
    double x = a + bl_expon_rate(r, alpha);
    double d = x - alpha;
    if (log(bl_unif(r)) <= -0.5 * d * d) return x;
  }
}

/* r.tnorm(lo, hi, 0, 1): standard normal restricted to (lo, hi), either bound
 * may be infinite.  Used by the constrained beta draw, Logit.hpp:393. */
double bl_tnorm(bl_rng *r, double lo, double hi)
{
  int lo_inf = isinf(lo) && lo < 0, hi_inf = isinf(hi) && hi > 0;
  if (lo_inf && hi_inf) return bl_norm(r, 0.0, 1.0);
  if (hi_inf) return tail_norm(r, lo);
  if (lo_inf) return -tail_norm(r, -hi);
  /* Degenerate interval (the chain starts on the constraint boundary, beta = 0,
   * where cmin == cmax up to rounding): no room to move and no uniforms consumed.
   * The width test is deliberately coarse so that rounding noise in the bounds
   * cannot change how many uniforms a call consumes. */
  if (!(hi - lo > 1e-12)) return lo;
  /* both finite */
  if (lo <= 0.0 && hi >= 0.0) {
    if (hi - lo > 2.5066282746310002) { /* sqrt(2 pi) */
      double x;
      do { x = bl_norm(r, 0.0, 1.0); } while (x < lo || x > hi);
      return x;
    }
    for (;;) {
      double x = bl_flat(r, lo, hi);
      if (log(bl_unif(r)) <= -0.5 * x * x) return x;
    }
  }
  /* interval on one side of zero: work on the positive side */
  int flip = hi < 0.0;
  double a = flip ? -hi : lo, b = flip ? -lo : hi, x;
  double alpha = 0.5 * (a + sqrt(a * a + 4.0));
  if (b - a > 1.0 / alpha) {
    do { x = tail_norm(r, a); } while (x > b);
  } else {
    for (;;) {
      x = bl_flat(r, a, b);
      if (log(bl_unif(r)) <= 0.5 * (a * a - x * x)) break;
    }
  }
  return flip ? -x : x;
}
