/* pg_devroye.c -- oracle (test infrastructure, see bl_oracle.h).
 * Restates Code/C/PolyaGamma.{h,cpp}: the Devroye-style J*(1,z) sampler, the
 * truncated sum of gammas, and the closed-form moments.
 *
 * The PG(1,z) sampler appears twice:
 *   - bl_pg_draw_like_devroye_literal / bl_pg_rtigauss: the reference loops line by line
 *     (PolyaGamma.cpp:82-115, 151-202), one r.unif()/r.expon_rate()/r.norm() call where the
 *     reference has one.  This is the CPU baseline bench.py times and the distribution every
 *     other form is held to.
 *   - bl_pg1_attempt / bl_pg_draw_like_devroye: the SAME proposals and acceptance events with
 *     the uniforms of the stream spent two per attempt (one Philox block) and recycled, which
 *     is the form the HIP kernels execute (bayeslogit_amd/csrc/bl_pg1_sm.hpp states the
 *     derivation).  The HIP parity tests compare against this form draw for draw;
 *     tests/test_oracle_pg.py pins it to the literal form's distribution and to the exact
 *     PG(1,z) CDF.
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
double bl_pg_draw_like_devroye_literal(double Z, bl_rng *r)
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

/* Census of the literal loop above: counts[j] (j = 1..3) = number of proposals whose series test ended
 * at term j (counts[0]: later terms), *nprop = proposals made, over ndraws draws.  The reference's notes
 * state these frequencies analytically (Notes/notes.tex:1000-1027); tests/test_oracle_pins.py compares. */
void bl_pg_devroye_literal_census(double Z, int64_t ndraws, uint64_t seed, int64_t counts[4], int64_t *nprop)
{
  Z = fabs(Z) * 0.5;
  double fz = 0.125 * PG_PI * PG_PI + 0.5 * Z * Z;
  double mass = bl_pg_mass_texpon(Z);
  counts[0] = counts[1] = counts[2] = counts[3] = 0;
  *nprop = 0;
  for (int64_t i = 0; i < ndraws; ++i) {
    bl_rng rr, *r = &rr;
    bl_rng_init(r, seed, (uint64_t)i, BL_DOM_DRAW, 0);
    int accepted = 0;
    while (!accepted) {
      double X = bl_unif(r) < mass ? PG_TRUNC + bl_expon_rate(r, 1) / fz : bl_pg_rtigauss(Z, r);
      double S = bl_pg_a(0, X);
      double Y = bl_unif(r) * S;
      *nprop += 1;
      for (int n = 1;; ++n) {
        int stop;
        if (n % 2 == 1) { S = S - bl_pg_a(n, X); stop = Y <= S; accepted = stop; }
        else            { S = S + bl_pg_a(n, X); stop = Y > S; }
        if (stop) { counts[n <= 3 ? n : 0] += 1; break; }
      }
    }
  }
}

/* ---- the attempt form (one Philox block = two uniforms per proposal attempt) ----
 * Same events as the literal loops above:
 *   right piece (u1 < mass, PolyaGamma.cpp:170-171): E = -log(u1/mass), X = t + E/fz;
 *   left piece, mu > t (:87-101): E1 = -log(w), X = t/(1+t E1)^2; the pair test
 *     E1^2 <= 2 E2/t (:95) holds with probability exp(-t E1^2/2) over E2 ~ Exp(1), and the
 *     `unif <= alpha` test (:89,:100) with probability exp(-Z^2 X/2): both failures restart at
 *     a new pair, so one uniform decides both: u2 <= A = exp(-t E1^2/2 - Z^2 X/2);
 *   left piece, mu <= t (:103-113): Y = N(0,1)^2 from w by inversion, X0 as :107-109, the
 *     reciprocal flip on u2 vs mu/(mu+X0) (:110-111), retry while X > t (:105);
 *   accept (:175-199): first alternating-series test on the remainder of u2 (u2/A, or the
 *     part of u2 the flip left), then the literal series on that same uniform.
 * w = (u1 - mass)/(1 - mass) on the attempt that chose the left piece, u1 on retries. */
#define PG1_WMIN 0x1.0p-53
#define PG1_WMAX (1.0 - 0x1.0p-53)

void bl_pg1_par_of(bl_pg1_par *p, double z)
{
  p->Z = fabs(z) * 0.5;                                   /* :154 */
  p->fz = 0.125 * PG_PI * PG_PI + 0.5 * p->Z * p->Z;      /* :157 */
  p->mass = bl_pg_mass_texpon(p->Z);
  p->im = 1.0 / p->mass;
  p->il = 1.0 / (1.0 - p->mass);
}

/* literal series PolyaGamma.cpp:175-199 with Y = u a_0(X); 1 = accept */
static int pg1_series(double X, double u)
{
  double S = bl_pg_a(0, X);
  double Y = u * S;
  for (int n = 1; n < 100000; ++n) {
    if (n % 2 == 1) {
      S = S - bl_pg_a(n, X);
      if (Y <= S) return 1;
    } else {
      S = S + bl_pg_a(n, X);
      if (Y > S) return 0;
    }
  }
  return 1;
}

static int pg1_decide(int *fresh, double *Xout, double X, double A, double rarg, double u2)
{
  if (!(X == X)) { *fresh = 1; *Xout = X; return 1; }     /* z = NaN: the reference loops fall through with NaN */
  if (u2 > A) { *fresh = 0; return 0; }
  int ok = u2 <= A * (1.0 - 3.0 * exp(rarg));             /* U a_0 <= a_0 - a_1 */
  if (!ok) ok = pg1_series(X, u2 / A);
  *fresh = 1;
  *Xout = X;
  return ok;
}

/* one attempt; returns 1 when a draw completed (value 0.25 * *X) */
int bl_pg1_attempt(int *fresh, double *X, const bl_pg1_par *p, double u1, double u2)
{
  const double t = PG_TRUNC;
  int right = *fresh && u1 < p->mass;
  double w = *fresh ? (right ? u1 * p->im : (u1 - p->mass) * p->il) : u1;
  if (w < PG1_WMIN) w = PG1_WMIN;
  if (w > PG1_WMAX) w = PG1_WMAX;
  if (right) {
    double E = -log(w);
    double Xc = t + E / p->fz;
    return pg1_decide(fresh, X, Xc, 1.0, -PG_PI * PG_PI * Xc, u2);
  }
  if (PG_TRUNC_RECIP > p->Z) {
    double E1 = -log(w);
    double d = 1.0 + t * E1;
    double Xc = t / (d * d);
    double A = exp(-0.5 * (t * E1 * E1 + p->Z * p->Z * Xc));
    return pg1_decide(fresh, X, Xc, A, -(4.0 / t) * d * d, u2);
  }
  double mu = 1.0 / p->Z;
  double Y = bl_qnorm(w);
  Y *= Y;
  double half_mu = 0.5 * mu;
  double mu_Y = mu * Y;
  double X0 = mu + half_mu * mu_Y - half_mu * sqrt(4 * mu_Y + mu_Y * mu_Y);
  double pk = mu / (mu + X0);
  int flip = u2 > pk;
  double Xc = flip ? mu * mu / X0 : X0;
  if (Xc > t) { *fresh = 0; return 0; }
  double v = flip ? (u2 - pk) / (1.0 - pk) : u2 / pk;
  if (v < PG1_WMIN) v = PG1_WMIN;
  if (v > PG1_WMAX) v = PG1_WMAX;
  return pg1_decide(fresh, X, Xc, 1.0, -4.0 / Xc, v);
}

/* PolyaGamma::draw_like_devroye(Z, r) in the attempt form (what the HIP path computes) */
double bl_pg_draw_like_devroye(double z, bl_rng *r)
{
  bl_pg1_par p;
  bl_pg1_par_of(&p, z);
  int fresh = 1;
  double X = 0.0;
  for (;;) {
    double u1 = bl_unif(r);
    double u2 = bl_unif(r);
    if (bl_pg1_attempt(&fresh, &X, &p, u1, u2)) return 0.25 * X;
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

double bl_pg_draw_devroye_literal(int n, double z, bl_rng *r)
{
  if (n < 1) n = 1;
  double sum = 0.0;
  for (int i = 0; i < n; ++i)
    sum += bl_pg_draw_like_devroye_literal(z, r);
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
