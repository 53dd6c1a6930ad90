// bl_pg1_sm.hpp -- PG(1, z) by Devroye's method as a lane-uniform state machine.
//
// Same sampler as Code/C/PolyaGamma.cpp:151-202 (and rtigauss :82-115, mass_texpon
// :65-80, a() :41-55): same proposals, same accept/reject events, and the uniforms of
// the observation's stream are consumed in the same order with the same meaning, so a
// draw equals the straight-line restatement's to rounding.  What changes is the shape
// of the computation, for 64-wide wavefronts:
//   * one TRANSITION consumes one uniform and moves a lane from state to state; every
//     lane of a wave runs the same short transition body (one log, one divide, one exp;
//     bl_fastmath.hpp's 38- and 24-instruction forms)
//     whatever state it is in, instead of each lane spinning in its own nested
//     rejection loops while the others wait;
//   * the proposal mass (mass_texpon) is evaluated without log/exp/erfc: both exponents
//     of PolyaGamma.cpp:73-75 collapse to the constant  t pi^2/8 - 1/(2t)  once
//     log Phi is written with the scaled erfc, leaving two Chebyshev sums;
//   * the first alternating-series test  U a_0 <= a_0 - a_1  is done on the ratio
//     a_1/a_0 (one exp); only the ~8e-4 of proposals that fail it walk the series,
//     in the reference's literal arithmetic.
#pragma once
#include "bl_erfcx.hpp"
#include "bl_fastmath.hpp"
#include "bl_philox.hpp"

namespace bl {

constexpr double kSmPi = 3.141592653589793238462643383279502884197;
constexpr double kSmT = 0.64;                         // __TRUNC
constexpr double kSmTRecip = 1.0 / 0.64;              // __TRUNC_RECIP
constexpr double kSmPiSq8 = kSmPi * kSmPi / 8.0;
constexpr double kSmPiSq = kSmPi * kSmPi;
constexpr double kSmInvSqrt2T = 0.88388347648318440550105545263106;   // 1/sqrt(2 * 0.64)
constexpr double kSmLogHalfPi = 0.45158270528945486472619522989488;

enum Pg1St : int {
  SM_BRANCH = 0,   // u decides exponential (right) vs inverse-Gaussian (left) piece   :170
  SM_RIGHT_E,      // u -> X = t + Exp(1)/fz                                           :171
  SM_ACCEPT,       // u -> Y = u a_0(X); alternating series                            :175-199
  SM_L_TEST,       // u vs alpha (alpha = 0 on entry): `while (r.unif() > alpha)`      :89
  SM_L_E1,         // u -> E1                                                          :94
  SM_L_E2,         // u -> E2; pair accepted if E1^2 <= 2 E2 / t, then X, alpha        :94-100
  SM_G_N1,         // mu <= t branch: first uniform of the normal                      :106
  SM_G_N2,         // second uniform of the normal -> candidate X                      :106-109
  SM_G_U,          // u vs mu/(mu+X): reciprocal flip; loop while X > t                :110-111
  SM_DONE
};

struct Pg1Par {      // per observation, from z
  double Z;          // |z|/2
  double fz;         // pi^2/8 + Z^2/2
  double mass;       // mass_texpon(Z)
};

struct Pg1Lane {     // per lane
  int st;
  double X;          // current proposal
  double aux;        // E1 / alpha / log(u1), depending on state
};

// mass_texpon(Z), PolyaGamma.cpp:65-80, rewritten (see header comment):
//   exp(x0 - Z + log Phi(b)) = fz C erfcx((1 - tZ)/sqrt(2t)) / 2              (tZ <= 1)
//                            = fz (exp(fz t - Z) - C erfcx((tZ - 1)/sqrt(2t))/2) (tZ > 1)
//   exp(x0 + Z + log Phi(a)) = fz C erfcx((1 + tZ)/sqrt(2t)) / 2
// with C = exp(t pi^2/8 - 1/(2t)).
BL_HD double pg1_mass(double Z, double fz)
{
  const double tz = kSmT * Z;
  const double ea = 0.5 * kMassC * erfcx_pos((1.0 + tz) * kSmInvSqrt2T);
  double eb;
  if (tz <= 1.0)
    eb = 0.5 * kMassC * erfcx_pos((1.0 - tz) * kSmInvSqrt2T);
  else
    eb = bl_exp(fz * kSmT - Z) - 0.5 * kMassC * erfcx_pos((tz - 1.0) * kSmInvSqrt2T);
  const double qdivp = 4.0 / kSmPi * fz * (ea + eb);
  return 1.0 / (1.0 + qdivp);
}

BL_HD Pg1Par pg1_par(double z)
{
  Pg1Par p;
  p.Z = fabs(z) * 0.5;                                // :154
  p.fz = kSmPiSq8 + 0.5 * p.Z * p.Z;                  // :157
  p.mass = pg1_mass(p.Z, p.fz);
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
// Returns true if the proposal X is accepted.
BL_HD_COLD bool pg1_series(double X, double u, int& status)
{
  const double logx = log(X);
  double S = pg1_a(0, X, logx);
  const double Y = u * S;
  for (int n = 1; n < 100000; ++n) {
    if (n & 1) {
      S = S - pg1_a(n, X, logx);
      if (Y <= S) return true;
    } else {
      S = S + pg1_a(n, X, logx);
      if (Y > S) return false;
    }
  }
  status |= 1;   // iteration cap (the reference loop is uncapped)
  return true;
}

// The mu <= t inverse-Gaussian branch (|z| >= 3.125), PolyaGamma.cpp:103-113: out of line so
// that its square roots, cospi and divides do not inflate the registers of the common path.
BL_HD_COLD void pg1_advance_g(Pg1Lane& s, const Pg1Par& p, double u, double lu)
{
  const double mu = 1.0 / p.Z;
  if (s.st == SM_G_N1) {
    s.aux = lu;
    s.st = SM_G_N2;
  } else if (s.st == SM_G_N2) {                                                  // :104-109
    double Y = sqrt(-2.0 * s.aux) * BL_COSPI(2.0 * u);                           // r.norm(1.0)
    Y *= Y;
    const double half_mu = 0.5 * mu;
    const double mu_Y = mu * Y;
    s.X = mu + half_mu * mu_Y - half_mu * sqrt(4.0 * mu_Y + mu_Y * mu_Y);
    s.st = SM_G_U;
  } else {                                                                       // :110-111, :105
    if (u > mu / (mu + s.X)) s.X = mu * mu / s.X;
    s.st = (s.X > kSmT) ? SM_G_N1 : SM_ACCEPT;
  }
}

// same, inlined
BL_HD void pg1_advance_g_inl(Pg1Lane& s, const Pg1Par& p, double u, double lu)
{
  const double mu = 1.0 / p.Z;
  if (s.st == SM_G_N1) {
    s.aux = lu;
    s.st = SM_G_N2;
  } else if (s.st == SM_G_N2) {                                                  // :104-109
    double Y = sqrt(-2.0 * s.aux) * BL_COSPI(2.0 * u);                           // r.norm(1.0)
    Y *= Y;
    const double half_mu = 0.5 * mu;
    const double mu_Y = mu * Y;
    s.X = mu + half_mu * mu_Y - half_mu * sqrt(4.0 * mu_Y + mu_Y * mu_Y);
    s.st = SM_G_U;
  } else {                                                                       // :110-111, :105
    if (u > mu / (mu + s.X)) s.X = mu * mu / s.X;
    s.st = (s.X > kSmT) ? SM_G_N1 : SM_ACCEPT;
  }
}

// One transition: consume uniform u.  Returns true when a draw has completed; the draw
// is then 0.25 * lane.X and the lane is back in SM_BRANCH for the next one.
// FAST: bl_fastmath log/exp (default) or libm's; COLDG: large-|z| branch out of line or inline
// (inline measured faster on gfx950 at 3 waves/SIMD: 8.6 vs 9.7 ms per 1e8 draws, z ~ U(0,4)).
// ZC: what the caller guarantees about the lane's observation -- 0 nothing, 1 |z|/2 < 1/t (the
// mu > t inverse-Gaussian branch: states L_*), 2 |z|/2 >= 1/t (states G_*).  A wave whose lanes
// all hold observations of one class compiles and executes only that class's states.
template <bool FAST = true, bool COLDG = false, int ZC = 0>
BL_HD bool pg1_advance(Pg1Lane& s, const Pg1Par& p, double u, int& status)
{
  const int st = s.st;
  // ---- common body: one log, one divide, one exp, whatever the state ----
  const double lu = FAST ? bl_log(u) : log(u);
  // the one division: E/fz (RIGHT_E, :171), t/(1+t E1)^2 (L_E2, :98-99), -4/X (ACCEPT, left piece)
  double num = -lu, den = p.fz;
  if (ZC != 2 && st == SM_L_E2) {
    const double d = 1.0 + s.aux * kSmT;
    num = kSmT;
    den = d * d;
  }
  if (st == SM_ACCEPT) {
    num = -4.0;
    den = s.X;
  }
  const double q = FAST ? bl_div(num, den) : num / den;
  double Xc = s.X;
  if (st == SM_RIGHT_E) Xc = kSmT + q;
  if (ZC != 2 && st == SM_L_E2) Xc = q;
  // the one exponential: alpha = exp(-Z^2 X/2) (L_E2, :100) or a_1/a_0 = 3 exp(.) (ACCEPT)
  double earg = -0.5 * p.Z * p.Z * Xc;
  if (st == SM_ACCEPT) earg = Xc > kSmT ? -kSmPiSq * Xc : q;
  const double ex = FAST ? bl_exp(earg) : exp(earg);

  bool finished = false;
  switch (st) {
    case SM_BRANCH:
      if (u < p.mass) {
        s.st = SM_RIGHT_E;
      } else if (ZC == 1 || (ZC == 0 && kSmTRecip > p.Z)) {                      // :87, mu > t
        s.st = SM_L_TEST;
        s.aux = 0.0;                                                             // alpha = 0, :88
      } else {
        s.st = SM_G_N1;
      }
      break;
    case SM_RIGHT_E:
      s.X = Xc;
      s.st = SM_ACCEPT;
      break;
    case SM_L_TEST:                                                              // :89
      if (ZC != 2) s.st = (u > s.aux) ? SM_L_E1 : SM_ACCEPT;
      break;
    case SM_L_E1:
      if (ZC != 2) {
        s.aux = -lu;                                                             // E1
        s.st = SM_L_E2;
      }
      break;
    case SM_L_E2:
      if (ZC != 2) {
        const double E1 = s.aux, E2 = -lu;
        if (E1 * E1 > 2.0 * E2 / kSmT) {                                         // :95
          s.st = SM_L_E1;
        } else {
          s.X = Xc;
          s.aux = ex;                                                            // alpha
          s.st = SM_L_TEST;
        }
      }
      break;
    case SM_ACCEPT: {
      // U a_0 <= a_0 - a_1  <=>  U <= 1 - a_1/a_0 ;  X <= 0 cannot occur (X > 0 always)
      bool ok = u <= 1.0 - 3.0 * ex;
      if (!ok) ok = pg1_series(s.X, u, status);                                  // rare
      if (ok) {
        finished = true;
        s.st = SM_BRANCH;
      } else {
        s.st = SM_BRANCH;                                                        // new proposal, :167
      }
    } break;
    case SM_G_N1:
    case SM_G_N2:
    case SM_G_U:
      if (ZC != 1) {
        if (COLDG) pg1_advance_g(s, p, u, lu);
        else pg1_advance_g_inl(s, p, u, lu);
      }
      break;
    default: break;
  }
  return finished;
}

// Sum of n PG(1, z) draws (PolyaGamma::draw(int n, z, r), :126-140; n < 1 -> 1 in the NTHROW
// build) on the observation's own stream, as a per-lane loop over Philox blocks: two
// transitions per block.  Lanes of a wave run it in lockstep; a lane that finishes early
// idles until its wave does (the work-queue kernel in kernels_pg.hip avoids that wait).
BL_HD double pg1_draw_n(int n, double z, uint64_t seed, uint64_t idx, uint32_t domain, uint32_t epoch, int& status)
{
  if (n < 1) {
    n = 1;
    status |= 2;
  }
  const Pg1Par p = pg1_par(z);
  Pg1Lane s{SM_BRANCH, 0.0, 0.0};
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  const uint32_t c0 = (uint32_t)idx, c1 = ctr1_of(idx, domain);
  double sum = 0.0;
  for (uint32_t blk = 0; blk < 4000000u; ++blk) {
    const U4 o = philox4x32_10(c0, c1, epoch, blk, k0, k1);
    if (pg1_advance(s, p, u52(o.x, o.y), status)) {
      sum += 0.25 * s.X;
      if (--n == 0) return sum;
    }
    if (pg1_advance(s, p, u52(o.z, o.w), status)) {
      sum += 0.25 * s.X;
      if (--n == 0) return sum;
    }
  }
  status |= 1;
  return sum;
}

}  // namespace bl
