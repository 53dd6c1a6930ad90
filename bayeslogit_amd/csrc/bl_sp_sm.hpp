// bl_sp_sm.hpp -- J*(n, z) for large n by the saddle-point-approximation method, one Philox block
// per proposal attempt.  Portable (host + device).
//
// Same sampler as Code/C/PolyaGammaSP.cpp:169-264: the same envelope of the saddle-point density
// (two tangent lines to eta = phi - delta at xl = the mode and xr = 1.2 xl, :174-204), the same
// mixture proposal (inverse Gaussian right-truncated at md = 1.1 xl, gamma left-truncated at md),
// the same acceptance event F U < sp_approx(X), the same iteration count.  What changes:
//
//   * one ATTEMPT consumes one Philox4x32-10 block = two uniforms (u1, u2), as in bl_alt_sm.hpp:
//     u1 picks the piece (`r.unif() < pl`, :243) and, recycled, drives the proposal variate
//     (qnorm(w) for r.igauss inside rtigauss, :57-76; -log w for r.ltgamma, :250, Dagpunar's method
//     as Code/R/Ch.R:83-114 states it); u2 decides the inner rejection (reciprocal flip of igauss,
//     or Dagpunar's rho) and its remainder is the U of the final test (:257).  An inner rejection
//     keeps the piece (state LEFT / RIGHT), as the reference's inner loops do, and does not count
//     as an iteration (:241).
//   * v(x), the root of x = tan(sqrt v)/sqrt v that sp_approx (:148-167) gets from a table bracket
//     and a Newton loop with tan/tanh per step (Code/C/InvertY.cpp:57-99), comes with
//     -log cos(sqrt v) and log K2 from three fitted polynomials in log2 x (bl_vtab.hpp): no loop,
//     no trigonometric function.
//   * the set-up (:171-229) without its cancellations.  v(xl) = -z^2 exactly (xl = y_func(-z^2)),
//     so phi(xl) = 0, the left tangent is closed-form and the exponent of wl (:217) is ~0;
//     p_igauss's two exponents are both -A1^2/2 once Phi is written with the scaled erfc;
//     Gamma(n) (1 - P(n, x)) (:222) is exp(-x) x^n times Legendre's continued fraction, and x^n
//     cancels -n log(n rr) - n log(md) in wr's exponent exactly: no Gamma(n) (the reason the
//     reference cannot take n > 170, LogitWrapper.cpp:142-146), no 1 - P.
// The integer-division literals (1/3), (2/15) of :163,:188 are 0 in the compiled reference and are
// kept as 0 here: K2 = x^2 where |v| < 1e-6.
#pragma once
#include "bl_erfcx.hpp"
#include "bl_fastmath.hpp"
#include "bl_gammainc.hpp"
#include "bl_philox.hpp"
#include "bl_qnorm.hpp"

namespace bl {

constexpr double kSpPi = 3.141592653589793238462643383279502884197;
constexpr double kSpLn2 = 0.69314718055994530941723212145818;
constexpr double kSpInvLn2 = 1.4426950408889634073599246810019;
constexpr double kSpLog1p1 = 0.095310179804324860043952123280765;   // log 1.1
constexpr double kSpLog1p2 = 0.18232155679395462621171802515451;    // log 1.2
constexpr double kSpLog2Pi = 1.8378770664093454835606594728112;
constexpr double kSpSqrtHalf = 0.70710678118654752440084436210485;
constexpr double kSpWMin = 0x1.0p-53, kSpWMax = 1.0 - 0x1.0p-53;
constexpr int kSpVtInt = 16, kSpVtDeg = 10;

// Per (n, z): what the set-up of PolyaGammaSP.cpp:171-229 leaves for the proposals.  The test F U < sp_approx(X)
// (:245-257) is run on D = log sp_approx - log F, in which the constants of both sides are folded:
//   left  of md: D = -log(K2)/2 + n (L - (v/2 + Z2h) X) + 3/2 log X + cL1 X + n/(2X) - cL0
//                cL0 = log(al)/2 + n il + n/(2 md) - n log cosh Z,  cL1 = n rl                       (:245-246)
//   right of md: D = -log(K2)/2 + n (L - (v/2 + Z2h) X) - (n - 1) log X + cR1 X - cR0
//                cR0 = log(ar)/2 + n ir - n log md - n log cosh Z,  cR1 = n rr                       (:251-252)
// with (v, L = -log cos_rt v, K2) of X from sp_vlk; lcn = log(n / 2 pi)/2 is on both sides and drops out.
struct SpPar {
  double n;
  double Z2h;         // (|z|/2)^2 / 2
  double md;          // 1.1 x the mode: where the two pieces meet, :175
  double mu;          // 1/sqrt(2 rl): mean of the left piece's inverse Gaussian, :244
  double pl;          // wl/(wl + wr), :226-227
  double b, mdb, lmdb;     // ltgamma(n, n rr, md): b = md n rr; md/b; log(md/b)
  double ic0, omc, log_m;  // Dagpunar's constants for Gamma(n, 1) left-truncated at b
  double cL0, cL1, cR0, cR1;
};
constexpr int kSpParDoubles = 15;

struct SpLane {
  int state;          // 0: the attempt starts a new iteration (:241); 1: retry inside the left piece; 2: inside the right piece
  double X;           // the iteration's proposal
};

BL_HD double sp_clamp(double w)
{
  w = w < kSpWMin ? kSpWMin : w;
  return w > kSpWMax ? kSpWMax : w;
}

// outside the fitted range [2^-4, 2^4]: the asymptotic forms of InvertY.cpp:62-68 with the literal cos_rt and K2
struct SpVlk { double v, L, lK2; };   // returned by value: reference parameters of an out-of-line call live in scratch
BL_HD_COLD SpVlk sp_vlk_far(double x, bool below)
{
  double vv;
  if (below) {
    vv = -1.0 / (x * x);
  } else {
    vv = atan(0.5 * x * kSpPi);
    vv = vv * vv;
  }
  const double r = sqrt(fabs(vv));
  return SpVlk{vv, -log(vv >= 0.0 ? cos(r) : cosh(r)), log(x * x + (1.0 - x) / vv)};
}

// v(x), -log cos_rt(v(x)), log K2(x) (:153-163).  vt: the table of bl_vtab.hpp ([3][16][11] doubles; the
// kernels keep a copy in LDS).
BL_HD void sp_vlk(const double* __restrict__ vt, double x, double logx, double& v, double& L, double& lK2)
{
  const double s = logx * kSpInvLn2;
  const bool inside = s >= -4.0 && s <= 4.0;
  const double u = inside ? 2.0 * (s + 4.0) : 8.0;
  int k = (int)u;
  k = k > kSpVtInt - 1 ? kSpVtInt - 1 : k;
  const double tau = 2.0 * (u - (double)k) - 1.0;
  const double* c0 = vt + k * (kSpVtDeg + 1);
  const double* c1 = c0 + kSpVtInt * (kSpVtDeg + 1);
  const double* c2 = c1 + kSpVtInt * (kSpVtDeg + 1);
  double a0 = c0[kSpVtDeg], a1 = c1[kSpVtDeg], a2 = c2[kSpVtDeg];
#pragma unroll
  for (int j = kSpVtDeg - 1; j >= 0; --j) {
    a0 = fma(a0, tau, c0[j]);
    a1 = fma(a1, tau, c1[j]);
    a2 = fma(a2, tau, c2[j]);
  }
  v = s * a0;
  L = s * a1;
  lK2 = fabs(v) < 1e-6 ? 2.0 * logx : a2;                // K2 = x^2 - (1/3) - (2/15) v with integer literals, :163
  if (wave_any(!inside)) {
    const SpVlk f = sp_vlk_far(inside ? 32.0 : x, s < -4.0);
    if (!inside) { v = f.v; L = f.L; lK2 = f.lK2; }
  }
}

BL_HD SpPar sp_par(double n, double z, const double* __restrict__ vt, int& status)
{
  SpPar p;
  double lcZ, imd, logmd, lhal, lhar, rl, il, rr, ir;   // log cosh Z; 1/md, log md; log(al)/2, log(ar)/2 (:190-191); tangent lines (:201-204)
  const double Z = 0.5 * fabs(z);                                            // :172
  const double Z2 = Z * Z;
  const double e2 = bl_exp(-2.0 * Z);
  const bool zbig = Z2 > 1e-6;
  const double xl = zbig ? bl_div(1.0 - e2, (1.0 + e2) * Z) : 1.0;           // y_func(-z^2), :78-90, :174
  lcZ = Z + bl_log(1.0 + e2) - kSpLn2;
  const double md = xl * 1.1, xr = xl * 1.2;                                 // :175-176
  const double logxl = bl_log(xl);
  p.n = n;
  p.Z2h = 0.5 * Z2;
  p.md = md;
  imd = bl_div(1.0, md);
  logmd = logxl + kSpLog1p1;
  const double logxr = logxl + kSpLog1p2;
  double vmd, Lmd, lK2md, vr, Lr, lK2r;
  sp_vlk(vt, md, logmd, vmd, Lmd, lK2md);                                  // :182-188
  sp_vlk(vt, xr, logxr, vr, Lr, lK2r);
  lhal = 0.5 * (3.0 * logmd - lK2md);                                    // log(md^3 / K2md)/2, :190
  lhar = 0.5 * (2.0 * logmd - lK2md);                                    // :191
  // tangent to eta at xl, :197: v(xl) = -Z^2, t = 0, phi(xl) = 0; where y_func returned 1, v_eval(1) = 0
  const double ixl = zbig ? bl_div(1.0, xl) : 1.0;
  const double tl = zbig ? 0.0 : 0.5 * Z2;
  const double phil = zbig ? 0.0 : lcZ - tl * xl;
  rl = tl + 0.5 * ixl * ixl;                                               // delta' = 0.5/x^2 left of md, :109-112
  il = phil - 0.5 * (imd - ixl) + rl * xl;                             // :144
  // tangent at xr, :198: delta = log xr - log md, delta' = 1/xr, :105-107
  const double tr = 0.5 * vr + 0.5 * Z2;
  const double phir = lcZ + Lr - tr * xr;
  rr = tr + bl_div(1.0, xr);
  ir = phir - (kSpLog1p2 - kSpLog1p1) + rr * xr;
  const double rt2rl = bl_sqrt(2.0 * rl);                                  // :210
  p.mu = bl_div(1.0, rt2rl);
  // log wl, :217-218.  p_igauss(md; mu, n) = 1 - exp(-A1^2/2) [erfcx(A1/sqrt 2) - erfcx(A2/sqrt 2)]/2 with
  // A1 = sqrt(n/md)(md/mu - 1), A2 = sqrt(n/md)(md/mu + 1)
  const double sn = bl_sqrt(n * imd);
  const double A1 = sn * (md * rt2rl - 1.0), A2 = sn * (md * rt2rl + 1.0);
  const double pig =
      1.0 - 0.5 * bl_exp(-0.5 * A1 * A1) * (erfcx_pos(A1 * kSpSqrtHalf) - erfcx_pos(A2 * kSpSqrtHalf));
  const double logn = bl_log(n);
  const double lwl0 = lhal + n * (il - rt2rl + 0.5 * imd);                   // log wl = lwl0 + log pig
  // log wr, :220-222: Gamma(n) Q(n, x) = exp(-x) x^n CF(n, x), x = n rr md
  const double x = n * rr * md;
  const double cf = upper_gamma_cf(n, x, status);
  const double lwr0 = lhar + 0.5 * (logn - kSpLog2Pi) + n * ir - x;        // log wr = lwr0 + log cf
  // wl / (wl + wr), :226-227, with the two factors that are not exponentials kept out of the exponent (two logs fewer):
  // pig / (pig + exp(lwr0 - lwl0) cf)
  p.pl = bl_div(pig, pig + bl_exp(lwr0 - lwl0) * cf);
  p.b = x;
  p.mdb = bl_div(md, x);
  p.lmdb = -(logn + bl_log(rr));                                           // log(md / (n rr md))
  if (n == 1.0) {                                                            // a == 1: the exponential, Ch.R:88-89
    p.ic0 = 1.0; p.omc = 0.0; p.log_m = 0.0;
  } else {
    const double d1 = x - n, d3 = n - 1.0;
    const double c0 = bl_div(0.5 * (d1 + bl_sqrt(d1 * d1 + 4.0 * x)), x);
    p.ic0 = bl_div(1.0, c0);
    p.omc = 1.0 - c0;
    p.log_m = d3 * (bl_log(bl_div(d3, p.omc)) - 1.0);
  }
  p.cL0 = lhal + n * (il + 0.5 * imd - lcZ);
  p.cL1 = n * rl;
  p.cR0 = lhar + n * (ir - logmd - lcZ);
  p.cR1 = n * rr;
  return p;
}

// One attempt: consume the block (u1, u2).  Returns true when an iteration of :235-259 has ended: a
// proposal s.X has met the test, and `accepted` is its verdict.
BL_HD bool sp_attempt(SpLane& s, const SpPar& p, const double* __restrict__ vt, double u1, double u2, bool& accepted)
{
  const double n = p.n;
  const bool fresh = s.state == 0;
  const bool left = fresh ? u1 < p.pl : s.state == 1;                                  // :243
  // the recycled uniform: u1/pl given {u1 < pl}, (u1 - pl)/(1 - pl) otherwise
  const double w = sp_clamp(fresh ? bl_div(left ? u1 : u1 - p.pl, left ? p.pl : 1.0 - p.pl) : u1);
  double X, logX, D, vnum, vden;      // D: the side's part of log sp_approx - log F that does not need v(X)
  bool retry;
  if (left) {
    // rtigauss(mu, n, md), :244: mu <= md always (mu <= xl), so r.igauss until <= md, :69-73
    const double mu = p.mu;
    const double nu = qnorm_t<true>(w);
    const double y = nu * nu;
    const double muy = mu * y;
    const double hml = bl_div(0.5 * mu, n);
    const double rad = 4.0 * mu * n * y + muy * muy;
    const double x0 = mu + hml * muy - hml * ((rad > 1e-300 && rad < 1e300) ? bl_sqrt(rad) : sqrt(rad));
    const double pk = bl_div(mu, mu + x0);
    const bool flip = u2 > pk;
    X = flip ? bl_div(mu * mu, x0) : x0;
    retry = X > p.md;
    vnum = flip ? u2 - pk : u2;
    vden = flip ? 1.0 - pk : pk;
    const bool xok = X > 1e-300 && X < 1e300;
    logX = xok ? bl_log(X) : log(X);
    const double iX = xok ? bl_div(1.0, X) : 1.0 / X;
    D = 1.5 * logX + p.cL1 * X + 0.5 * n * iX - p.cL0;                                 // -log F + n log cosh Z, :245-246
  } else {
    // r.ltgamma(n, n rr, md), :250
    const double E = -bl_log(w);
    const double x = fma(E, p.ic0, p.b);
    const double lx = bl_log(x);
    const double rho = bl_exp((n - 1.0) * lx - x * p.omc - p.log_m);
    retry = u2 > rho;
    vnum = u2;
    vden = rho;
    X = x * p.mdb;                                                                      // trunc (x / b)
    logX = lx + p.lmdb;
    D = p.cR1 * X - (n - 1.0) * logX - p.cR0;                                          // -log F + n log cosh Z, :251-252
  }
  if (retry) {
    s.state = left ? 1 : 2;
    return false;
  }
  const double vu = sp_clamp(bl_div(vnum, vden));
  double v, L, lK2;
  sp_vlk(vt, X, logX, v, L, lK2);
  D += n * (L - (0.5 * v + p.Z2h) * X) - 0.5 * lK2;                                    // + log sp_approx - n log cosh Z, :157-165
  s.state = 0;
  s.X = X;
  accepted = vu < bl_exp(D);                                                           // F U < spa, :257
  return true;
}

// ---- a TASK: one draw, PolyaGammaSP::draw's loop :235-259 over the observation's Philox stream
constexpr uint32_t kSpBlkCap = 4000000u;

struct SpTask {
  SpPar par;
  SpLane sm;
  int iter;
  uint32_t c0, c1, blk;
};

BL_HD void sp_task_start(SpTask& T, const SpPar& par, uint64_t idx, uint32_t domain)
{
  T.par = par;
  T.sm.state = 0;
  T.sm.X = 2.0;                                             // :232
  T.iter = 0;
  T.c0 = (uint32_t)idx;
  T.c1 = ctr1_of(idx, domain);
  T.blk = 0;
}

// One step (at most one Philox block).  Returns true when the draw is complete: value n X / 4 (:262), T.iter
// iterations (:263).
BL_HD bool sp_task_step(SpTask& T, const double* __restrict__ vt, int maxiter, uint32_t epoch, uint32_t k0,
                        uint32_t k1, int& status)
{
  if (T.sm.state == 0) {
    if (T.iter >= maxiter) return true;                     // :235: the last proposal stands
    T.iter += 1;
  }
  const U4 o = philox4x32_10(T.c0, T.c1, epoch, T.blk, k0, k1);
  T.blk += 1;
  bool acc = false;
  if (sp_attempt(T.sm, T.par, vt, u52(o.x, o.y), u52(o.z, o.w), acc) && acc) return true;
  if (T.blk >= kSpBlkCap) {
    status |= 1;
    return true;
  }
  return false;
}

}  // namespace bl
