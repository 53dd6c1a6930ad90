// bl_alt_sm.hpp -- J*(h, z), h in [1, 4], by the alternating-series method, one Philox block per
// proposal attempt.  Portable (host + device).
//
// Same sampler as Code/C/PolyaGammaAlt.cpp:114-203 (draw_abridged): the same mixture proposal
// (Gamma(h, pi^2/8 + z^2/2) right of the truncation point t(h), the a_0 kernel -- a truncated
// inverse Gaussian -- left of it), the same acceptance events with the same probabilities, the
// same alternating-series test with its `decreasing` guard.  What changes is the shape of the
// computation, so that a 64-wide wavefront runs ONE state-free body per attempt under a work
// queue instead of every lane spinning in its own three levels of rejection loops:
//
//   * one ATTEMPT consumes one Philox4x32-10 block = two uniforms (u1, u2).
//       u1  picks the piece on a fresh proposal (`uu < prob_right`, :149) and is recycled --
//           conditional on {u1 < p}, u1/p is uniform and independent of the event -- into the
//           proposal variate: E = -log(w) for the truncated gamma's exponential (r.ltgamma,
//           Dagpunar's method as Code/R/Ch.R:83-114 states it) and for E1 of the
//           inverse-chi-square pair (:6-22); N(0,1) = qnorm(w) for the inverse-Gaussian
//           candidate (r.igauss, :91-94);
//       u2  decides the attempt's rejection event through a threshold A: Dagpunar's rho(x); or
//           exp(-R E1^2/2 - z^2 X/2), the pair test of :13 (which holds with probability
//           exp(-R E1^2/2) over E2 ~ Exp(1)) and the `unif > alpha` test of :82-88 in one
//           comparison (both failures restart at a new pair); or the reciprocal flip of igauss.
//           A rejected attempt keeps its piece (state LEFT / RIGHT), as the reference's inner
//           loops do.  Conditional on passing, u2/A (or what the flip left of u2) is uniform and
//           is the U of the series test (:160-161).
//   * the series runs on a_n / g_tilde: a_n/a_{n-1} = ((n+h-1)/n) ((2n+h)/(2n+h-2)) e_n with
//     e_n = exp(-2(2n+h-1)/X) = e_{n-1} exp(-4/X): two exponentials per proposal whatever the
//     number of terms (the reference: one exp and two logs per term, :37-49); left of t,
//     a_0/g_tilde = 1 (g_tilde IS a_0 there, :99-108).
//   * the mixture weights (:60-75) without their cancellations: both exponents of w_left's
//     inverse-Gaussian CDF collapse to -h^2/(2t) - t z^2/2 once Phi is written with the scaled
//     erfc; w_right's 1 - P(h, x) is Gamma(h, x)/Gamma(h) in closed form for integer h and by
//     Legendre's continued fraction otherwise.
// The test suite's CPU checker holds the same attempt in plain C next to a call-for-call
// restatement of the reference loops and pins the two to one distribution.
#pragma once
#include "bl_erfcx.hpp"
#include "bl_fastmath.hpp"
#include "bl_gammainc.hpp"
#include "bl_philox.hpp"
#include "bl_qnorm.hpp"

namespace bl {

constexpr double kAltPi = 3.141592653589793238462643383279502884197;
constexpr double kAltPiSq8 = kAltPi * kAltPi / 8.0;
constexpr double kAltLn2 = 0.69314718055994530941723212145818;
constexpr double kAltLogHalfPi = 0.45158270528945486472619522989488;    // log(pi/2)
constexpr double kAltLog4OverPi = 0.24156447527049044469103689059156;   // log(4/pi)
constexpr double kAltHalfLog2Pi = 0.91893853320467274178032973640562;   // log(2 pi)/2
constexpr double kAltWMin = 0x1.0p-53, kAltWMax = 1.0 - 0x1.0p-53;

struct AltPar {       // per (h, z): the part of draw_abridged that draws nothing, :117-140
  double h;           // shape of this abridged draw, in [1, 4]
  double Z;           // |z|/2, :122
  double t;           // trunc_schedule[floor((h-1) 100)], :124-125
  double fz;          // rate_z = pi^2/8 + Z^2/2, :128
  double lfz;         // log fz
  double p;           // prob_right, :131
  double R;           // t/h^2, :9
  double ic0, omc, log_m;   // Dagpunar's constants for Gamma(h, 1) left-truncated at t fz: 1/c0, 1 - c0, log M
  double cR;          // log[(4/pi)^h Gamma(h+1) / sqrt(2 pi)]: a_0/g_tilde right of t without its X-dependent part
};
constexpr int kAltParDoubles = 11;

struct AltLane {
  int state;          // 0: the attempt starts a new trial (:142-153); 1: retry inside the left piece; 2: inside the right piece
  double X;           // the accepted proposal when an attempt completes a draw (value 0.25 X, :191)
};

BL_HD double alt_clamp(double w)
{
  w = w < kAltWMin ? kAltWMin : w;
  return w > kAltWMax ? kAltWMax : w;
}

// Gamma(h, x)/Gamma(h) and log Gamma(h + 1) for non-integer h (rpg.alt with real shapes, or a remainder
// in (4, 5) halved, :219-220).
struct AltQ { double q, lg1; int status; };   // returned by value: reference parameters of an out-of-line call live in scratch
BL_HD_COLD AltQ alt_q_general(double h, double x)
{
  int st = 0;
  const double lg = lgamma(h);
  const double q = bl_exp(-x + h * bl_log(x) - lg) * upper_gamma_cf(h, x, st);
  return AltQ{q, lg + bl_log(h), st};
}

BL_HD AltPar alt_par(double h, double z, double t, int& status)
{
  AltPar p;
  p.h = h;
  p.Z = fabs(z) * 0.5;
  p.t = t;
  p.fz = kAltPiSq8 + 0.5 * p.Z * p.Z;
  p.lfz = bl_log(p.fz);
  // w_left, :60-68: 2^h [exp(-hZ) Phi((tZ - h)/sqrt t) + exp(hZ) Phi(-(tZ + h)/sqrt t)].  With
  // Phi(-a) = erfcx(a/sqrt 2) exp(-a^2/2)/2 both products have the exponent -h^2/(2t) - t Z^2/2 (the Z = 0
  // branch of :65-66, 2^h Q(1/2, h^2/(2t)) = 2^h erfc(h/sqrt(2t)), is the same expression).
  const double tz = t * p.Z;
  const double isq = bl_div(1.0, bl_sqrt(2.0 * t));
  const double Ex = bl_exp(-(bl_div(0.5 * h * h, t) + 0.5 * tz * p.Z));
  const double two_h1 = bl_exp((h - 1.0) * kAltLn2);
  const double e2 = erfcx_pos((h + tz) * isq);
  double wl;
  if (tz <= h)
    wl = two_h1 * Ex * (erfcx_pos((h - tz) * isq) + e2);
  else
    wl = 2.0 * two_h1 * bl_exp(-h * p.Z) + two_h1 * Ex * (e2 - erfcx_pos((tz - h) * isq));
  // w_right, :70-75: (pi/2 / fz)^h (1 - P(h, fz t)) = (pi/2 / fz)^h Gamma(h, x)/Gamma(h)
  const double x = p.fz * t;
  double q, lg1;   // Gamma(h, x)/Gamma(h), log Gamma(h + 1)
  const bool integer_h = h == 1.0 || h == 2.0 || h == 3.0 || h == 4.0;
  if (integer_h) {                                           // exp(-x) sum_{k<h} x^k/k!
    double s = 1.0;
    if (h >= 2.0) s += x;
    if (h >= 3.0) s += 0.5 * x * x;
    if (h >= 4.0) s += (1.0 / 6.0) * x * x * x;
    q = bl_exp(-x) * s;
    lg1 = h == 1.0 ? 0.0 : h == 2.0 ? kAltLn2 : h == 3.0 ? 1.7917594692280550008 : 3.1780538303479456196;
  }
  if (wave_any(!integer_h)) {
    const AltQ g = alt_q_general(integer_h ? 2.5 : h, integer_h ? 3.0 : x);
    if (!integer_h) { q = g.q; lg1 = g.lg1; status |= g.status; }
  }
  const double wr = bl_exp(h * (kAltLogHalfPi - p.lfz)) * q;
  p.p = bl_div(wr, wr + wl);                                 // :131
  p.R = bl_div(t, h * h);
  // Dagpunar (Code/R/Ch.R:83-114): a = h, b = t fz; a == 1: the exponential, :88-89
  if (h == 1.0) {
    p.ic0 = 1.0; p.omc = 0.0; p.log_m = 0.0;
  } else {
    const double b = x, d1 = b - h, d3 = h - 1.0;
    const double c0 = bl_div(0.5 * (d1 + bl_sqrt(d1 * d1 + 4.0 * b)), b);
    p.ic0 = bl_div(1.0, c0);
    p.omc = 1.0 - c0;
    p.log_m = d3 * (bl_log(bl_div(d3, p.omc)) - 1.0);
  }
  p.cR = h * kAltLog4OverPi + lg1 - kAltHalfLog2Pi;
  return p;
}

// One attempt: consume the block (u1, u2).  Returns true when an abridged draw has completed; the
// draw is then 0.25 * s.X and the lane is ready for the next one.
BL_HD bool alt_attempt(AltLane& s, const AltPar& p, double u1, double u2, int& status)
{
  const double h = p.h, t = p.t;
  const bool fresh = s.state == 0;
  const bool right = fresh ? u1 < p.p : s.state == 2;                                  // :149
  // the recycled uniform: u1/p given {u1 < p}, (u1 - p)/(1 - p) otherwise
  const double w = alt_clamp(fresh ? bl_div(right ? u1 : u1 - p.p, right ? p.p : 1.0 - p.p) : u1);
  const bool big = !right && !(p.Z * t < h);             // left piece with mu = h/Z <= t, :81 (Z = 0: mu = inf)
  double X = 0.0, logX = 0.0, vnum = u2, vden = 1.0;
  bool retry = false;
  if (!big) {
    // r.ltgamma(h, rate_z, trunc) (:150) and the mu > t rtigauss (:81-89 with :6-22) share one body:
    // E = -log w, one quotient, one exponential threshold
    const double E = -bl_log(w);
    double num, den, aarg;
    if (right) {
      const double x = fma(E, p.ic0, t * p.fz);          // b + E/c0
      const double lx = bl_log(x);
      aarg = (h - 1.0) * lx - x * p.omc - p.log_m;       // log rho(x)
      num = x;
      den = p.fz;                                        // X = trunc (x / b) = x / fz
      logX = lx - p.lfz;
    } else {
      const double d = fma(E, p.R, 1.0);
      num = t;
      den = d * d;                                       // X = h^2 R / (1 + E1 R)^2, :18-20
      aarg = -0.5 * p.R * E * E;
    }
    X = bl_div(num, den);
    if (!right) aarg = fma(-0.5 * p.Z * p.Z, X, aarg);   // alpha = exp(-z^2 X / 2), :87
    const double A = bl_exp(aarg);
    retry = u2 > A;
    vden = A;
  } else {
    // r.igauss(mu, h^2) until <= trunc, :91-94
    const double mu = bl_div(h, p.Z), lam = h * h;
    const double nu = qnorm_t<true>(w);
    const double y = nu * nu;
    const double muy = mu * y;
    const double hml = bl_div(0.5 * mu, lam);
    const double rad = 4.0 * mu * lam * y + muy * muy;
    const double x0 = mu + hml * muy - hml * ((rad > 1e-300 && rad < 1e300) ? bl_sqrt(rad) : sqrt(rad));
    const double pk = bl_div(mu, mu + x0);
    const bool flip = u2 > pk;
    X = flip ? bl_div(mu * mu, x0) : x0;
    retry = X > t;
    vnum = flip ? u2 - pk : u2;
    vden = flip ? 1.0 - pk : pk;
  }
  if (retry) {
    s.state = right ? 2 : 1;
    return false;
  }
  const double v = alt_clamp(bl_div(vnum, vden));
  const double iX = bl_div(1.0, X);
  // a_0/g_tilde: 1 left of t; right of it (4/pi)^h Gamma(h+1)/sqrt(2 pi) X^-(h+1/2) exp(pi^2 X/8 - h^2/(2X))
  const double s0 = right ? bl_exp(p.cR - (h + 0.5) * logX + kAltPiSq8 * X - 0.5 * h * h * iX) : 1.0;
  // the alternating series, :156-196, on a_n/g_tilde
  double e = bl_exp(-2.0 * (h + 1.0) * iX);
  double a_prev = s0, S = s0;
  bool decided = !(X == X), ok = false;                  // z = NaN: no test can hold; the trial is dropped
  {                                                      // n = 1: a_1/a_0 = (h + 2) e_1
    const double a = a_prev * ((h + 2.0) * e);
    S -= a;
    if (!decided && v <= S && a <= a_prev) { decided = true; ok = true; }
    a_prev = a;
  }
  if (wave_any(!decided)) {
    const double q2 = bl_exp(-4.0 * iX);
    double fn = 1.0;
    for (int n = 2; n <= 200; ++n) {                     // max_inner = 200, :114, :172
      fn += 1.0;
      e *= q2;
      const double rn = bl_div((fn + h - 1.0) * (2.0 * fn + h), fn * (2.0 * fn + h - 2.0)) * e;
      const double a = a_prev * rn;
      const bool dec = a <= a_prev;
      if (n & 1) {
        S -= a;
        if (!decided && v <= S && dec) { decided = true; ok = true; }
      } else {
        S += a;
        if (!decided && v > S && dec) decided = true;    // ok stays false: a new trial
      }
      a_prev = a;
      if (!wave_any(!decided)) break;
    }
  }
  s.state = 0;          // accepted: the next draw; rejected (or 200 terms without a verdict): a new trial, :142
  s.X = X;
  return ok;
}

// ---- a TASK: n abridged draws with one (h, z), summed -- one of the two groups PolyaGammaAlt::draw
// (:205-225) adds up: floor((h-1)/4) draws at shape 4 (group A), then the remainder (group B: one draw, or
// two at half the remainder when it exceeds 4).  A task reads the observation's Philox stream from block
// blk0 on: group A from block 0, group B from block 2^31, so the two groups of an observation can be drawn
// by different lanes at different times and x = sumA + sumB (one fp addition: commutative) is the same.
constexpr uint32_t kAltBlkGroupB = 0x80000000u;
#ifndef BL_ALT_BLK_CAP
#define BL_ALT_BLK_CAP 4000000u
#endif
constexpr uint32_t kAltBlkCap = BL_ALT_BLK_CAP;   // blocks per abridged DRAW (the reference's inner loops are uncapped); not per task
constexpr int kAltMaxTrials = 10000;           // :142

struct AltTask {
  AltPar par;
  AltLane sm;
  int nrem, trials;
  uint32_t c0, c1, blk, blk_end;
  double sum;
};

BL_HD void alt_task_start(AltTask& T, const AltPar& par, int ndraws, uint64_t idx, uint32_t domain, uint32_t blk0)
{
  T.par = par;
  T.sm.state = 0;
  T.sm.X = 0.0;
  T.nrem = ndraws;
  T.trials = 0;
  T.c0 = (uint32_t)idx;
  T.c1 = ctr1_of(idx, domain);
  T.blk = blk0;
  T.blk_end = blk0 + kAltBlkCap;
  T.sum = 0.0;
}

// One step of a task (at most one Philox block).  Returns true when the task's sum is complete.
BL_HD bool alt_task_step(AltTask& T, uint32_t epoch, uint32_t k0, uint32_t k1, int& status)
{
  bool got = false;
  double val = 0.0;
  if (T.sm.state == 0 && ++T.trials > kAltMaxTrials) {        // "We should never get here", :201-202
    status |= 4;
    got = true;
    val = -1.0;
  } else {
    const U4 o = philox4x32_10(T.c0, T.c1, epoch, T.blk, k0, k1);
    T.blk += 1;
    if (alt_attempt(T.sm, T.par, u52(o.x, o.y), u52(o.z, o.w), status)) {
      got = true;
      val = 0.25 * T.sm.X;                                    // :191
    } else if (T.blk == T.blk_end) {
      status |= 1;
      got = true;
      val = 0.25 * T.sm.X;
      T.sm.state = 0;
    }
  }
  if (got) {
    T.sum += val;
    T.trials = 0;
    T.blk_end = T.blk + kAltBlkCap;     // the cap counts from the start of each draw: a group of many draws is not cut short
    return --T.nrem <= 0;
  }
  return false;
}

// the two groups of PolyaGammaAlt::draw(h, ...), :211-222: shapes and counts (h >= 1)
BL_HD void alt_groups(double h, int& nA, double& hB, int& nB)
{
  const double n = floor((h - 1.0) * 0.25);
  const double remain = h - 4.0 * n;
  nA = (int)n;
  nB = remain > 4.0 ? 2 : 1;
  hB = remain > 4.0 ? 0.5 * remain : remain;
}

// t(h), :124-125
BL_HD double alt_trunc_of(const double* __restrict__ sched, double h)
{
  int idx = (int)floor((h - 1.0) * 100.0);
  idx = idx < 0 ? 0 : idx > 300 ? 300 : idx;
  return sched[idx];
}

}  // namespace bl
