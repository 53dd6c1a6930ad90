// bl_pg_devroye.hpp -- the parts of Code/C/PolyaGamma.cpp other than the J*(1, z/2) sampler
// itself (which is bl_pg1_sm.hpp): status flags, the truncated sum of gammas and the
// closed-form moments.  Cited per function.  gfx950 only.
#pragma once
#include "bl_rng.hpp"
#include "bl_specfun.hpp"

namespace bl {

enum : int { ST_OK = 0, ST_ITER_CAP = 1, ST_BAD_SHAPE = 2, ST_ALT_FALLTHROUGH = 4, ST_NOT_PD = 8 };

// PolyaGamma::draw_sum_of_gammas, PolyaGamma.cpp:142-149 with bvec of :19-39.
__device__ inline double pg_draw_sum_of_gammas(double b, double z, int trunc, Stream& r)
{
  if (trunc < 1) trunc = 1;
  double x = 0.0;
  const double kappa = z * z;
  for (int k = 0; k < trunc; ++k) {
    const double d = (double)k + 0.5;
    const double bk = 4.0 * kPi * kPi * d * d;
    x += gamma_scale(r, b, 1.0) / (bk + kappa);
  }
  return 2.0 * x;
}

// PolyaGamma::jj_m1 / jj_m2 / pg_m1 / pg_m2, PolyaGamma.cpp:208-239.
__device__ __host__ inline double jj_m1(double b, double z)
{
  z = fabs(z);
  if (z > 1e-12) return b * tanh(z) / z;
  return b * (1 - (1.0 / 3) * pow(z, 2) + (2.0 / 15) * pow(z, 4) - (17.0 / 315) * pow(z, 6));
}
__device__ __host__ inline double jj_m2(double b, double z)
{
  z = fabs(z);
  if (z > 1e-12) return (b + 1) * b * pow(tanh(z) / z, 2) + b * ((tanh(z) - z) / pow(z, 3));
  return (b + 1) * b * pow(1 - (1.0 / 3) * pow(z, 2) + (2.0 / 15) * pow(z, 4) - (17.0 / 315) * pow(z, 6), 2) +
         b * ((-1.0 / 3) + (2.0 / 15) * pow(z, 2) - (17.0 / 315) * pow(z, 4));
}
__device__ __host__ inline double pg_m1(double b, double z) { return jj_m1(b, 0.5 * z) * 0.25; }
__device__ __host__ inline double pg_m2(double b, double z) { return jj_m2(b, 0.5 * z) * 0.0625; }

}  // namespace bl
