// bl_pg_hybrid.hpp -- device dispatch of one PG(b, z) draw by shape, as the
// reference's rpg_hybrid does (Code/C/LogitWrapper.cpp:140-162).  gfx950 only.
#pragma once
#include "bl_pg_devroye.hpp"
#include "bl_pg1_sm.hpp"

namespace bl {

enum : int { CLS_ZERO = 0, CLS_GAMMA = 1, CLS_DEVROYE = 2, CLS_ALT = 3, CLS_SP = 4, CLS_NORMAL = 5 };

// Which branch of LogitWrapper.cpp:142-161 a shape takes.
__device__ __host__ __forceinline__ int pg_class(double b)
{
  if (b > 170.0) return CLS_NORMAL;
  if (b > 13.0) return CLS_SP;
  if (b == 1.0 || b == 2.0) return CLS_DEVROYE;
  if (b > 1.0) return CLS_ALT;
  if (b > 0.0) return CLS_GAMMA;
  return CLS_ZERO;
}

// One draw of class `cls` on the observation's stream, for the two classes drawn one observation per lane
// (the normal approximation above b = 170, the sum of gammas below 1).  CLS_DEVROYE runs pg1_draw_n
// (bl_pg1_sm.hpp); CLS_ALT and CLS_SP run as tasks under the work queue of bl_task_queue.hpp.
__device__ inline double pg_hybrid_class(int cls, double b, double z, Stream& r, int& status)
{
  double x = 0.0;
  switch (cls) {
    case CLS_NORMAL: {
      const double m = pg_m1(b, z);
      const double v = pg_m2(b, z) - m * m;
      x = r.norm(m, sqrt(v));
    } break;
    case CLS_GAMMA: x = pg_draw_sum_of_gammas(b, z, 200, r); break;
    default: x = 0.0;
  }
  return x;
}

}  // namespace bl
