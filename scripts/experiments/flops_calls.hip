// flops_calls.hip -- one tiny kernel per arithmetic building block of the draw kernels, so that scripts/count_flops.py can
// count, in the gfx950 ISA the compiler emits for each, the vector instructions and the fp64 flops (v_fma_f64 = 2,
// v_mul/v_add_f64 = 1) of ONE call.  Nothing here runs; it is compiled with -S only.
#include "../../bayeslogit_amd/csrc/bl_pg1_sm.hpp"
using namespace bl;
#define K(name, expr)                                                                    \
  extern "C" __global__ void fc_##name(double* o, const double* a, const double* b)      \
  {                                                                                      \
    const int i = threadIdx.x;                                                           \
    const double x = a[i], y = b[i];                                                     \
    (void)y;                                                                             \
    o[i] = (expr);                                                                       \
  }
K(baseline, x)
K(log, bl_log(x))
K(exp, bl_exp(x))
K(div, bl_div(x, y))
K(sqrt, bl_sqrt(x))
K(qnorm, qnorm(x))
K(erfcx, erfcx_pos(x))
K(mass_small, pg1_mass_small(x, y))
K(mass_general, pg1_mass(x, y))
extern "C" __global__ void fc_philox(double* o, const unsigned* a)
{
  const int i = threadIdx.x;
  const U4 r = philox4x32_10(a[i], a[i + 64], a[i + 128], a[i + 192], a[256], a[257]);
  o[i] = u52(r.x, r.y) + u52(r.z, r.w);          // includes the two conversions to (0,1) doubles; the add is the harness's
}
// one attempt of the |z|/2 < 1/t class (fresh or retry), as the queue kernels run it, without Philox
extern "C" __global__ void fc_attempt_class1(double* o, const double* a, const double* b)
{
  const int i = threadIdx.x;
  Pg1Par p{a[i], a[i + 64], a[i + 128], a[i + 192], a[i + 256]};
  Pg1Lane s{b[i] > 0.5, 0.0};
  int st = 0;
  const bool done = pg1_attempt<true, 1>(s, p, b[i + 64], b[i + 128], st);
  o[i] = done ? s.X : -1.0;
}
extern "C" __global__ void fc_attempt_class2(double* o, const double* a, const double* b)
{
  const int i = threadIdx.x;
  Pg1Par p{a[i], a[i + 64], a[i + 128], a[i + 192], a[i + 256]};
  Pg1Lane s{b[i] > 0.5, 0.0};
  int st = 0;
  const bool done = pg1_attempt<true, 2>(s, p, b[i + 64], b[i + 128], st);
  o[i] = done ? s.X : -1.0;
}
