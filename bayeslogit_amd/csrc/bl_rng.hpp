// bl_rng.hpp -- device-side counter RNG and the random primitives the samplers
// pull from `RNG& r` in the reference (jwindle/RNG, absent from its tree;
// semantics per SURVEY.md Appendix B).  gfx950 only.
//
// Stream contract (DESIGN.md "RNG stream contract"): one Philox4x32-10 stream
// per (seed, index, domain, epoch); block b of the stream is
//   Philox(key = (seed_lo, seed_hi), ctr = (idx_lo, idx_hi | domain<<24, epoch, b))
// and yields two uniforms, from words (0,1) and (2,3):
//   u = ((w_a:w_b >> 12) + 1/2) * 2^-52   in (0,1), exact in binary64.
// Because a stream belongs to an observation (not to a lane, wave or GPU) the
// output is independent of launch geometry, lane assignment and GPU count.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bl_philox.hpp"
#include "bl_qnorm.hpp"

namespace bl {

// Stream domains (counter word 1, bits 24-31): disjoint counter spaces under one key.
//   DOM_DRAW   rpg_* draws            idx = observation, epoch = call number of the .C boundary (or the caller's)
//   DOM_BETA   beta draw of a chain   idx = 0,           epoch = sweep
//   DOM_DATA   synthetic data
//   DOM_OMEGA  omega draws of a chain idx = observation, epoch = sweep (mlogit: sweep (J-1) + category)
//   DOM_KEY    key derivation: the chain key of the k-th gibbs()/mult_gibbs() call after set_seed
enum : uint32_t { DOM_DRAW = 0, DOM_BETA = 1, DOM_DATA = 2, DOM_OMEGA = 3, DOM_KEY = 4 };

constexpr double kPi = 3.141592653589793238462643383279502884197;

struct Stream {
  uint32_t k0, k1, c0, c1, c2, blk;
  double spare;
  bool has;

  __device__ __forceinline__ void init(uint64_t seed, uint64_t idx, uint32_t domain, uint32_t epoch)
  {
    k0 = (uint32_t)seed;
    k1 = (uint32_t)(seed >> 32);
    c0 = (uint32_t)idx;
    c1 = ((uint32_t)(idx >> 32) & 0x00FFFFFFu) | (domain << 24);
    c2 = epoch;
    blk = 0;
    has = false;
    spare = 0.0;
  }

  __device__ __forceinline__ double unif()
  {
    if (has) {
      has = false;
      return spare;
    }
    const U4 o = philox4x32_10(c0, c1, c2, blk, k0, k1);
    blk += 1;
    spare = u52(o.z, o.w);
    has = true;
    return u52(o.x, o.y);
  }

  // r.expon_rate(rate)
  __device__ __forceinline__ double expon(double rate) { return -log(unif()) / rate; }

  // r.norm(m, sd): Box-Muller, cosine branch, two uniforms.
  __device__ __forceinline__ double norm(double mean, double sd)
  {
    const double u1 = unif();
    const double u2 = unif();
    return mean + sd * sqrt(-2.0 * log(u1)) * cospi(2.0 * u2);
  }

  // r.flat(a, b)
  __device__ __forceinline__ double flat(double a, double b) { return a + (b - a) * unif(); }
};

// r.gamma_scale(shape, scale): Marsaglia-Tsang, shape < 1 by the U^(1/a) boost.
__device__ inline double gamma_scale(Stream& r, double shape, double scale)
{
  double boost = 1.0;
  double a = shape;
  if (a < 1.0) {
    boost = exp(log(r.unif()) / a);
    a += 1.0;
  }
  const double d = a - 1.0 / 3.0;
  const double c = 1.0 / sqrt(9.0 * d);
  for (int it = 0; it < 100000; ++it) {
    const double x = r.norm(0.0, 1.0);
    double v = 1.0 + c * x;
    if (v <= 0.0) continue;
    v = v * v * v;
    const double u = r.unif();
    if (log(u) < 0.5 * x * x + d - d * v + d * log(v)) return d * v * boost * scale;
  }
  return d * boost * scale;
}

// r.igauss(mu, lambda): Michael-Schucany-Haas.
__device__ inline double igauss(Stream& r, double mu, double lambda)
{
  const double nu = r.norm(0.0, 1.0);
  const double y = nu * nu;
  const double muy = mu * y;
  double x = mu + 0.5 * mu * muy / lambda - 0.5 * mu / lambda * sqrt(4.0 * mu * lambda * y + muy * muy);
  if (r.unif() > mu / (mu + x)) x = mu * mu / x;
  return x;
}

// r.ltgamma(shape, rate, trunc): Dagpunar's left-truncated gamma.
__device__ inline double ltgamma(Stream& r, double shape, double rate, double trunc)
{
  const double a = shape;
  const double b = trunc * rate;
  if (a == 1.0) return trunc + r.expon(1.0) / rate;
  const double d1 = b - a;
  const double d3 = a - 1.0;
  const double c0 = 0.5 * (d1 + sqrt(d1 * d1 + 4.0 * b)) / b;
  const double omc = 1.0 - c0;
  const double log_m = d3 * (log(d3 / omc) - 1.0);
  double x = b;
  for (int it = 0; it < 100000; ++it) {
    x = b + r.expon(1.0) / c0;
    const double log_rho = d3 * log(x) - x * omc - log_m;
    if (log(r.unif()) <= log_rho) break;
  }
  return trunc * (x / b);
}

// r.rtinvchi2(scale, trunc): scale/chi^2_1 on (0, trunc], exponential-pair form.
__device__ inline double rtinvchi2(Stream& r, double scale, double trunc)
{
  const double R = trunc / scale;
  double E1 = r.expon(1.0);
  double E2 = r.expon(1.0);
  for (int it = 0; it < 100000 && E1 * E1 > 2.0 * E2 / R; ++it) {
    E1 = r.expon(1.0);
    E2 = r.expon(1.0);
  }
  double X = 1.0 + E1 * R;
  X = R / (X * X);
  return scale * X;
}

constexpr double kSqrtHalfR = 0.70710678118654752440084436210485;
constexpr int kTnormUniforms = 9;

// X ~ N(0,1) | a <= X <= b, 0 <= a < b (b may be +inf), by inversion of the upper tail.
__device__ inline double tnorm_inv_right(double a, double b, double u)
{
  double x;
  if (a > 37.0) {
    const double w = isinf(b) ? 1.0 : -expm1(-a * (b - a));
    x = a - log1p(-u * w) / a;
  } else {
    const double qa = 0.5 * erfc(a * kSqrtHalfR), qb = isinf(b) ? 0.0 : 0.5 * erfc(b * kSqrtHalfR);
    x = -qnorm(qa - u * (qa - qb));
  }
  x = x < a ? a : x;
  x = x > b ? b : x;
  return x;
}

// The four rejection attempts + inverse-CDF fallback of r.tnorm(lo, hi, 0, 1), given the
// call's nine uniforms (U[2k], U[2k+1] = attempt k; U[8] = fallback).  A call always
// owns exactly nine uniforms of its stream whatever the bounds, so the stream never
// desynchronises on rounding noise in the bounds and a whole beta draw's random input
// can be generated up front, off the serial coordinate loop.
__device__ inline double tnorm_from_uniforms(const double* U, double lo, double hi)
{
  const bool lo_inf = isinf(lo) && lo < 0, hi_inf = isinf(hi) && hi > 0;
  if (!(hi - lo > 0.0)) return lo;
  if (lo <= 0.0 && hi >= 0.0) {
    const bool wide = hi - lo > 2.5066282746310002;
    for (int k = 0; k < 4; ++k) {
      const double ua = U[2 * k], ub = U[2 * k + 1];
      if (wide) {
        const double x = sqrt(-2.0 * log(ua)) * cospi(2.0 * ub);
        if (x >= lo && x <= hi) return x;
      } else {
        const double x = lo + (hi - lo) * ua;
        if (log(ub) <= -0.5 * x * x) return x;
      }
    }
    const double pl = lo_inf ? 0.0 : 0.5 * erfc(-lo * kSqrtHalfR);
    const double ph = hi_inf ? 1.0 : 0.5 * erfc(-hi * kSqrtHalfR);
    double x = qnorm(pl + U[8] * (ph - pl));
    x = x < lo ? lo : x;
    x = x > hi ? hi : x;
    return x;
  }
  const bool flip = hi < 0.0;
  const double a = flip ? -hi : lo, b = flip ? -lo : hi;
  const double alpha = 0.5 * (a + sqrt(a * a + 4.0));
  const bool tail = b - a > 1.0 / alpha;
  double x = a;
  bool done = false;
  for (int k = 0; k < 4 && !done; ++k) {
    const double ua = U[2 * k], ub = U[2 * k + 1];
    if (tail) {
      x = a - log(ua) / alpha;
      const double d = x - alpha;
      done = x <= b && log(ub) <= -0.5 * d * d;
    } else {
      x = a + (b - a) * ua;
      done = log(ub) <= 0.5 * (a * a - x * x);
    }
  }
  if (!done) x = tnorm_inv_right(a, b, U[8]);
  return flip ? -x : x;
}

// r.tnorm(lo, hi, 0, 1)
__device__ inline double tnorm(Stream& r, double lo, double hi)
{
  double U[kTnormUniforms];
#pragma unroll
  for (int k = 0; k < kTnormUniforms; ++k) U[k] = r.unif();
  return tnorm_from_uniforms(U, lo, hi);
}

}  // namespace bl
