// Host build of the kernels' PG(1,z) attempt body (bayeslogit_amd/csrc/bl_pg1_sm.hpp) for
// CPU-side unit tests: the SAME header the HIP kernels inline, compiled as plain C++.
// Test scaffolding only -- the shipped library never contains this object.
#include "../../bayeslogit_amd/csrc/bl_pg1_sm.hpp"

extern "C" {

void sm_rpg_devroye(double* x, const int* n, const double* z, long num, unsigned long long seed, unsigned epoch,
                    unsigned long long idx0, int* status)
{
  int st = 0;
  for (long i = 0; i < num; ++i)
    x[i] = n[i] != 0 ? bl::pg1_draw_n(n[i], z[i], seed, idx0 + (unsigned long long)i, 0u, epoch, st) : 0.0;
  *status = st;
}

double sm_mass(double Z) { return bl::pg1_par(2.0 * Z).mass; }   // the class dispatch the kernels use
double sm_mass_general(double Z) { return bl::pg1_mass(Z, bl::kSmPiSq8 + 0.5 * Z * Z); }
double sm_erfcx(double x) { return bl::erfcx_pos(x); }

// attempt census: how many Philox blocks a draw consumed (diagnostic for DESIGN.md)
long sm_count_attempts(double z, long ndraws, unsigned long long seed)
{
  long total = 0;
  int st = 0;
  for (long i = 0; i < ndraws; ++i) {
    const bl::Pg1Par p = bl::pg1_par(z);
    bl::Pg1Lane s{true, 0.0};
    for (unsigned blk = 0;; ++blk) {
      const bl::U4 o = bl::philox4x32_10((unsigned)i, 0, 0, blk, (unsigned)seed, (unsigned)(seed >> 32));
      ++total;
      if (bl::pg1_attempt(s, p, bl::u52(o.x, o.y), bl::u52(o.z, o.w), st)) break;
    }
  }
  return total;
}
}

#include "../../bayeslogit_amd/csrc/bl_fastmath.hpp"
extern "C" {
double fm_log(double x) { return bl::bl_log(x); }
double fm_exp(double x) { return bl::bl_exp(x); }
double fm_sqrt(double x) { return bl::bl_sqrt(x); }
double fm_div(double a, double b) { return bl::bl_div(a, b); }
}
