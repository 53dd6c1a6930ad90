// bl_pg1_sm.hpp -- PG(1, z) by Devroye's method, one Philox block per proposal attempt.
//
// Same sampler as Code/C/PolyaGamma.cpp:151-202 (with rtigauss :82-115, mass_texpon :65-80,
// a() :41-55): the same mixture proposal (exponential piece right of t = 0.64 with probability
// mass_texpon(Z), inverse-Gaussian piece left of it), the same acceptance events with the same
// probabilities, the same alternating-series test.  What changes is how the observation's
// uniforms are spent, so that a 64-wide wavefront runs ONE short, state-free body per attempt
// instead of each lane spinning in its own nested rejection loops:
//
//   * one ATTEMPT consumes one Philox4x32-10 block = two uniforms (u1, u2).  u1 chooses the piece
//     and -- recycled: conditional on {u1 < p}, u1/p is again uniform and independent of the
//     event -- also drives the proposal variate; u2 decides every accept/reject event of the
//     attempt through nested thresholds (conditional on {u2 <= A}, u2/A is uniform):
//        right piece   E = -log(u1/mass),  X = t + E/fz                                 :171
//                      accept  u2 <= 1 - a_1(X)/a_0(X)                                   :175-192
//        left, mu > t  E1 = -log(w),  X = t/(1 + t E1)^2                                 :94-99
//                      pair kept (E1^2 <= 2 E2/t, E2 ~ Exp(1): probability exp(-t E1^2/2)) :95
//                      AND `unif <= alpha = exp(-Z^2 X/2)`                               :89,100
//                        <=>  u2 <= A = exp(-t E1^2/2 - Z^2 X/2);   else a new pair (both
//                             rejections restart at the same point of the reference's loops)
//                      accept  u2/A <= 1 - a_1(X)/a_0(X)
//        left, mu <= t Y = qnorm(w)^2, X0 as :107-109, reciprocal flip on u2 vs mu/(mu+X0) :110-111,
//                      retry while X > t :105, accept on the recycled remainder of u2
//     with w = (u1 - mass)/(1 - mass) on the attempt that chose the left piece and w = u1 on the
//     retries inside it.  E2 and the initial `unif() > 0` of :88-89 never reach the result and
//     are not drawn.
//   * the proposal mass (mass_texpon) is evaluated without log/exp/erfc: both exponents of
//     PolyaGamma.cpp:73-75 collapse to the constant  t pi^2/8 - 1/(2t)  once log Phi is written
//     with the scaled erfc, leaving two Chebyshev sums;
//   * the first alternating-series test  U a_0 <= a_0 - a_1  is done on the ratio a_1/a_0
//     (3 exp(-pi^2 X) right of t, 3 exp(-4/X) left of it; both < 0.006, so the exponential is
//     only evaluated when some lane of the wave has u2 within 0.006 of its threshold); only the
//     ~8e-4 of proposals that fail it walk the series, in the reference's literal arithmetic.
// The test suite's CPU checker holds the same attempt in plain C next to a call-for-call
// restatement of the reference loops, and pins the two to the same distribution.
#pragma once
#include "bl_erfcx.hpp"
#include "bl_fastmath.hpp"
#include "bl_masscheb2.hpp"
#include "bl_masspoly.hpp"
#include "bl_philox.hpp"
#include "bl_qnorm.hpp"

namespace bl {

constexpr double kSmPi = 3.141592653589793238462643383279502884197;
constexpr double kSmT = 0.64;                         // __TRUNC
constexpr double kSmTRecip = 1.0 / 0.64;              // __TRUNC_RECIP
constexpr double kSmPiSq8 = kSmPi * kSmPi / 8.0;
constexpr double kSmPiSq = kSmPi * kSmPi;
constexpr double kSmInvSqrt2T = 0.88388347648318440550105545263106;   // 1/sqrt(2 * 0.64)
constexpr double kSmLogHalfPi = 0.45158270528945486472619522989488;
constexpr double kSmRatioMax = 0.006;   // > 3 exp(-pi^2 t) = 0.00542 and > 3 exp(-4/t) = 0.00579
constexpr double kSmWMin = 0x1.0p-53, kSmWMax = 1.0 - 0x1.0p-53;   // recycled uniforms stay inside (0,1)
// The reference's loops are uncapped (PolyaGamma.cpp:167,181); a lane that never exits would hang its wave, so ONE PG(1,z)
// draw may spend at most this many Philox blocks (an attempt fails with probability < 0.3: the cap is a pathology detector,
// not a limit on n -- the count starts again with every draw of an observation's sum).  Overridable for the host tests.
#ifndef BL_PG1_BLK_CAP
#define BL_PG1_BLK_CAP 4000000u
#endif
constexpr uint32_t kPg1BlkCap = BL_PG1_BLK_CAP;

struct Pg1Par {      // per observation, from z
  double Z;          // |z|/2
  double fz;         // pi^2/8 + Z^2/2
  double mass;       // mass_texpon(Z)
  double im;         // 1/mass
  double il;         // 1/(1 - mass)
};

struct Pg1Lane {     // per lane
  bool fresh;        // true: the attempt starts a new proposal (u1 picks the piece);
                     // false: it is a retry inside the left piece
  double X;          // the accepted proposal when an attempt completes a draw
};

// true if any lane of the wavefront has `need` set (host build: the one caller)
BL_HD bool pg1_any(bool need)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return __ballot(need) != 0ull;
#else
  return need;
#endif
}

// mass_texpon(Z), PolyaGamma.cpp:65-80, rewritten (see header comment):
//   exp(x0 - Z + log Phi(b)) = fz C erfcx((1 - tZ)/sqrt(2t)) / 2              (tZ <= 1)
//                            = fz (exp(fz t - Z) - C erfcx((tZ - 1)/sqrt(2t))/2) (tZ > 1)
//   exp(x0 + Z + log Phi(a)) = fz C erfcx((1 + tZ)/sqrt(2t)) / 2
// with C = exp(t pi^2/8 - 1/(2t)).  The literal erfcx form (two 28-term Chebyshev sums):
BL_HD_COLD double pg1_mass_erfcx(double Z, double fz)
{
  const double tz = kSmT * Z;
  const double ea = 0.5 * kMassC * erfcx_pos((1.0 + tz) * kSmInvSqrt2T);
  double eb;
  if (tz <= 1.0)
    eb = 0.5 * kMassC * erfcx_pos((1.0 - tz) * kSmInvSqrt2T);
  else
    eb = bl_exp(fz * kSmT - Z) - 0.5 * kMassC * erfcx_pos((tz - 1.0) * kSmInvSqrt2T);
  return 4.0 / kSmPi * fz * (ea + eb);
}
// For 1 <= tZ <= 26 (the mu <= t class up to |z| = 81) the two erfcx terms are one smooth function
// H(Z) = C/2 [erfcx(k(tZ - 1)) - erfcx(k(tZ + 1))]: a 22-term Chebyshev series in r = s/(s + 4), s = tZ - 1
// (bl_masscheb2.hpp, 1e-17 relative to what it is subtracted from; Clenshaw) -- a divide, 44 vector instructions and the one
// exponential instead of two 28-term sums (round 3: the set-up was a quarter of the class's work).  Outside that range, and
// for NaN, the erfcx form.
BL_HD double pg1_mass(double Z, double fz)
{
  const double tz = kSmT * Z;
  const bool mid = tz >= 1.0 && tz <= 1.0 + kMassCheb2SMax;
  const double sv = tz - 1.0;
  const double r = bl_div(sv, sv + kMassCheb2C);
  const double x = fma(r, kMassCheb2Scale, -1.0), x2 = x + x;
  double b1 = 0.0, b2 = 0.0;
#pragma unroll
  for (int j = kMassCheb2N - 1; j >= 1; --j) {
    const double bn = fma(x2, b1, kMassCheb2[j] - b2);
    b2 = b1;
    b1 = bn;
  }
  const double H = fma(x, b1, kMassCheb2[0] - b2);
  double qdivp = 4.0 / kSmPi * fz * (bl_exp(fz * kSmT - Z) - H);
  if (pg1_any(!mid)) {
    if (!mid) qdivp = pg1_mass_erfcx(Z, fz);
  }
  return 1.0 / (1.0 + qdivp);
}

// mass_texpon(Z) for Z < 1/t (the mu > t class): there tZ < 1, so the two erfcx terms above are
// erfcx(k(1 + tZ)) + erfcx(k(1 - tZ)), k = 1/sqrt(2t): an even entire function of Z, i.e. a short
// polynomial in Z^2 (bl_masspoly.hpp, 12 terms, 2e-16) -- 13 FMAs and a divide instead of two
// 28-term Chebyshev sums.
BL_HD double pg1_mass_small(double Z, double fz)
{
  const double s = (2.0 * kSmT * kSmT) * Z * Z - 1.0;
  double g = kMassPoly[kMassPolyN - 1];
#pragma unroll
  for (int k = kMassPolyN - 2; k >= 0; --k) g = fma_vvs(g, s, kMassPoly[k]);
  const double qdivp = 4.0 / kSmPi * fz * g;
  return bl_div(1.0, 1.0 + qdivp);
}

// the reciprocals the recycling needs, from (Z, mass)
BL_HD void pg1_par_finish(Pg1Par& p)
{
  p.fz = kSmPiSq8 + 0.5 * p.Z * p.Z;                  // :157
  p.im = bl_div(1.0, p.mass);
  p.il = bl_div(1.0, 1.0 - p.mass);
}

BL_HD Pg1Par pg1_par(double z)
{
  Pg1Par p;
  p.Z = fabs(z) * 0.5;                                // :154
  p.fz = kSmPiSq8 + 0.5 * p.Z * p.Z;
  p.mass = kSmTRecip > p.Z ? pg1_mass_small(p.Z, p.fz) : pg1_mass(p.Z, p.fz);
  pg1_par_finish(p);
  return p;
}

// a_n(x), PolyaGamma.cpp:41-55 (literal form, used on the rare series path)
BL_HD double pg1_a(int n, double x, double logx)
{
  const double nh = n + 0.5;
  const double K = nh * kSmPi;
  if (x > kSmT) return K * exp(-0.5 * K * K * x);
  if (x > 0.0) return exp(-1.5 * (kSmLogHalfPi + logx) + log(K) - 2.0 * nh * nh / x);
  return 0.0;
}

// The alternating series from term 1 on, literal arithmetic of PolyaGamma.cpp:175-199.
// Returns 1 if the proposal X is accepted, 0 if not; bit 1 set: the iteration cap was hit (by value: a reference
// parameter of an out-of-line call would live in scratch).
BL_HD_COLD int pg1_series(double X, double u)
{
  const double logx = log(X);
  double S = pg1_a(0, X, logx);
  const double Y = u * S;
  for (int n = 1; n < 100000; ++n) {
    if (n & 1) {
      S = S - pg1_a(n, X, logx);
      if (Y <= S) return 1;
    } else {
      S = S + pg1_a(n, X, logx);
      if (Y > S) return 0;
    }
  }
  return 3;      // iteration cap (the reference loop is uncapped): accept and flag
}

// Outcome of an attempt once X, the threshold A of u2 and the exponent of a_1/a_0 are known.
// Sets s.fresh for the next attempt; returns true when the draw is complete (value 0.25 * s.X).
template <bool FAST>
BL_HD bool pg1_decide(Pg1Lane& s, double X, double A, double rarg, double u2, int& status)
{
  if (!(X == X)) {                   // z = NaN: the reference's loops fall through with X = NaN (:105, :191)
    s.fresh = true;
    s.X = X;
    return true;
  }
  if (u2 > A) {                      // left piece only (A = 1 on the right): a new pair / candidate
    s.fresh = false;
    return false;
  }
  bool ok = u2 <= A * (1.0 - kSmRatioMax);
  if (pg1_any(!ok)) {
    const double r3 = 3.0 * (FAST ? bl_exp(rarg) : exp(rarg));
    if (!ok) {
      ok = u2 <= A * (1.0 - r3);
      if (!ok) {                                                                 // rare
        const int r = pg1_series(X, u2 / A);
        ok = (r & 1) != 0;
        status |= r >> 1;
      }
    }
  }
  s.fresh = true;                    // accepted: next draw; rejected: new proposal, :167
  s.X = X;
  return ok;
}

// The exponential piece (right) and the mu > t left piece share one body -- one log, one divide, one exp: the
// proposal X (:171 / :98-99), the threshold A of u2 (1 on the right) and log(a_1/a_0) - log 3 at X.
template <bool FAST>
BL_HD void pg1_small_body(bool right, double w, double Z, double fz, double& X, double& A, double& rarg)
{
  const double E = -(FAST ? bl_log(w) : log(w));
  const double d = 1.0 + kSmT * E;
  const double q = FAST ? bl_div(right ? E : kSmT, right ? fz : d * d) : (right ? E / fz : kSmT / (d * d));
  X = right ? kSmT + q : q;
  const double aarg = -0.5 * (kSmT * E * E + Z * Z * X);
  A = right ? 1.0 : (FAST ? bl_exp(aarg) : exp(aarg));
  rarg = right ? -kSmPiSq * X : -(4.0 / kSmT) * d * d;
}

// One attempt of a |z|/2 < 1/t observation (finite z) whose state is KNOWN to the caller, without the rare
// path and without a branch: the arithmetic and the tests of pg1_attempt<true, 1> + pg1_decide up to the first
// series test, in STAGES, so that a caller can put other work between them (the single-pass Gibbs sweep issues
// its matrix instructions there).  Verdict 0: u2 > A, the next attempt is a retry inside the left piece; 1:
// accepted, the draw is 0.25 X; 2: the first series test failed (about 8e-4 of proposals) -- pg1_attempt would
// walk the series.  A caller that evaluates attempts ahead of time (lanes a = 0..3 of a row take blocks 0..3,
// block 0 fresh and the others as retries) hands such an observation to the full sampler, which replays its
// stream from block 0.
struct Pg1Staged {
  bool right;
  double w, E, d, X, aarg, rarg, A, r3;
};
// :170 and the proposal's uniform: u1 im resp. (u1 - mass) il of pg1_attempt, with the one reciprocal that is used
BL_HD void pg1_stage_w(Pg1Staged& s, bool fresh, double mass, double u1)
{
  s.right = fresh && u1 < mass;
  const double inv = bl_div(1.0, s.right ? mass : 1.0 - mass);                   // pg1_par_finish
  double w = fresh ? (s.right ? u1 : u1 - mass) * inv : u1;
  w = w < kSmWMin ? kSmWMin : w;
  s.w = w > kSmWMax ? kSmWMax : w;
}
BL_HD void pg1_stage_log(Pg1Staged& s) { s.E = -bl_log(s.w); }
BL_HD void pg1_stage_x(Pg1Staged& s, double Z, double fz)
{
  s.d = 1.0 + kSmT * s.E;
  const double q = bl_div(s.right ? s.E : kSmT, s.right ? fz : s.d * s.d);
  s.X = s.right ? kSmT + q : q;
  s.aarg = -0.5 * (kSmT * s.E * s.E + Z * Z * s.X);
  s.rarg = s.right ? -kSmPiSq * s.X : -(4.0 / kSmT) * s.d * s.d;
}
// exp(0) is exactly 1: no select on the result, so the exponential is not sunk into a branch
BL_HD void pg1_stage_A(Pg1Staged& s) { s.A = bl_exp_straight(s.right ? 0.0 : s.aarg); }
BL_HD void pg1_stage_r3(Pg1Staged& s) { s.r3 = 3.0 * bl_exp_straight(s.rarg); }   // a_1/a_0, whether or not the lane needs it
BL_HD int pg1_stage_verdict(const Pg1Staged& s, double u2)
{
  const bool inner = !(u2 > s.A);
  const bool ok = u2 <= s.A * (1.0 - kSmRatioMax) || u2 <= s.A * (1.0 - s.r3);
  return inner ? (ok ? 1 : 2) : 0;
}
// The stages in order.  STRAIGHT: no branch at all; otherwise a_1/a_0 (an exponential) is evaluated only when some lane of
// the wave has u2 within kSmRatioMax of its threshold, as pg1_decide does -- same verdicts, same X.
template <bool STRAIGHT = false>
BL_HD int pg1_attempt_small_known(bool fresh, double Z, double fz, double mass, double u1, double u2, double& X)
{
  Pg1Staged s;
  pg1_stage_w(s, fresh, mass, u1);
  pg1_stage_log(s);
  pg1_stage_x(s, Z, fz);
  pg1_stage_A(s);
  X = s.X;
  if (STRAIGHT) {
    pg1_stage_r3(s);
    return pg1_stage_verdict(s, u2);
  }
  const bool inner = !(u2 > s.A);
  bool ok = u2 <= s.A * (1.0 - kSmRatioMax);
  if (pg1_any(inner && !ok)) ok = ok || u2 <= s.A * (1.0 - 3.0 * bl_exp(s.rarg));
  return inner ? (ok ? 1 : 2) : 0;
}

// One attempt: consume the block (u1, u2).  Returns true when a draw has completed; the draw is
// then 0.25 * lane.X and the lane is ready for the next one.
// FAST: bl_fastmath log/exp (default) or libm's.
// ZC: what the caller guarantees about the lane's observation -- 0 nothing, 1 |z|/2 < 1/t (the
// mu > t inverse-Gaussian branch), 2 |z|/2 >= 1/t.  A wave whose lanes all hold observations of
// one class compiles and executes only that class's left piece.
template <bool FAST = true, int ZC = 0>
BL_HD bool pg1_attempt(Pg1Lane& s, const Pg1Par& p, double u1, double u2, int& status)
{
  const bool right = s.fresh && u1 < p.mass;                                     // :170
  double w = s.fresh ? (right ? u1 * p.im : (u1 - p.mass) * p.il) : u1;
  w = w < kSmWMin ? kSmWMin : w;
  w = w > kSmWMax ? kSmWMax : w;
  const bool small = ZC == 1 || (ZC == 0 && kSmTRecip > p.Z);                    // :87, mu > t

  if (ZC != 2 && (right || small)) {
    double X, A, rarg;
    pg1_small_body<FAST>(right, w, p.Z, p.fz, X, A, rarg);
    return pg1_decide<FAST>(s, X, A, rarg, u2, status);
  }
  if (right) {                                                                   // ZC == 2 only
    const double E = -(FAST ? bl_log(w) : log(w));
    const double X = kSmT + (FAST ? bl_div(E, p.fz) : E / p.fz);
    return pg1_decide<FAST>(s, X, 1.0, -kSmPiSq * X, u2, status);
  }
  // mu <= t: inverse-Gaussian candidate from one normal, PolyaGamma.cpp:103-113 (divides and the square
  // root in bl_fastmath's short forms when FAST: <= 1 ulp from the IEEE sequences)
  const double mu = FAST ? bl_div(1.0, p.Z) : 1.0 / p.Z;
  double Y = qnorm_t<FAST>(w);
  Y *= Y;
  const double half_mu = 0.5 * mu;
  const double mu_Y = mu * Y;
  const double rad = 4.0 * mu_Y + mu_Y * mu_Y;
  const double X0 = mu + half_mu * mu_Y - half_mu * ((FAST && rad > 1e-300 && rad < 1e300) ? bl_sqrt(rad) : sqrt(rad));
  const double pk = FAST ? bl_div(mu, mu + X0) : mu / (mu + X0);
  const bool flip = u2 > pk;                                                     // :110
  const double X = flip ? (FAST ? bl_div(mu * mu, X0) : mu * mu / X0) : X0;
  if (X > kSmT) {                                                                // :105
    s.fresh = false;
    return false;
  }
  const double vn = flip ? u2 - pk : u2, vd = flip ? 1.0 - pk : pk;              // the rest of u2
  double v = FAST ? bl_div(vn, vd) : vn / vd;
  v = v < kSmWMin ? kSmWMin : v;
  v = v > kSmWMax ? kSmWMax : v;
  return pg1_decide<FAST>(s, X, 1.0, FAST ? bl_div(-4.0, X) : -4.0 / X, v, status);
}

// Sum of n PG(1, z) draws (PolyaGamma::draw(int n, z, r), :126-140; n < 1 -> 1 in the NTHROW
// build) on the observation's own stream, as a per-lane loop over Philox blocks.  Lanes of a
// wave run it in lockstep; a lane that finishes early idles until its wave does (the work queue
// of bl_pg1_queue.hpp avoids that wait).
// blk0 > 0 (n == 1 only): the observation's blocks 0 .. blk0 - 1 are known to have been retries inside the left piece (the
// single-pass sweep's ahead-of-time attempts): the draw goes on at block blk0 as the retry it is -- an attempt is a function
// of its block and of `fresh` alone.
BL_HD double pg1_draw_n(int n, double z, uint64_t seed, uint64_t idx, uint32_t domain, uint32_t epoch, int& status,
                        uint32_t blk0 = 0)
{
  if (n < 1) {
    n = 1;
    status |= 2;
  }
  const Pg1Par p = pg1_par(z);
  Pg1Lane s{blk0 == 0, 0.0};
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  const uint32_t c0 = (uint32_t)idx, c1 = ctr1_of(idx, domain);
  double sum = 0.0;
  uint32_t cap = kPg1BlkCap;         // first block the current draw may not use
  for (uint32_t blk = blk0; blk != cap; ++blk) {
    const U4 o = philox4x32_10(c0, c1, epoch, blk, k0, k1);
    if (pg1_attempt(s, p, u52(o.x, o.y), u52(o.z, o.w), status)) {
      sum += 0.25 * s.X;
      if (--n == 0) return sum;
      cap = blk + 1u + kPg1BlkCap;   // per draw, not per observation: n is not limited by it
    }
  }
  status |= 1;                       // a draw ran into the cap: flagged, the sum so far returned
  return sum;
}

}  // namespace bl
