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

// The single-pass Gibbs sweep's draw (kernels_sweep1.hip): attempts 0..3 of an observation evaluated ahead of time --
// block 0 as a fresh proposal, blocks 1..3 as retries inside the left piece -- and resolved as the kernel resolves its four
// lanes: the first attempt in block order that does not end in a retry decides; accepted -> the draw, first series test
// open or none of the four -> the observation goes to the full sampler.  Returns 1 and the draw in *x when settled, else 0.
int sm_ahead_of_time4(double z, unsigned long long seed, unsigned long long idx, unsigned domain, unsigned epoch, double* x)
{
  const double Z = fabs(z) * 0.5, fz = bl::kSmPiSq8 + 0.5 * Z * Z, mass = bl::pg1_mass_small(Z, fz);
  const unsigned c0 = (unsigned)idx, c1 = bl::ctr1_of(idx, domain);
  for (unsigned a = 0; a < 4; ++a) {
    const bl::U4 o = bl::philox4x32_10(c0, c1, epoch, a, (unsigned)seed, (unsigned)(seed >> 32));
    double X;
    const int verdict = bl::pg1_attempt_small_known(a == 0, Z, fz, mass, bl::u52(o.x, o.y), bl::u52(o.z, o.w), X);
    if (verdict == 1) {
      *x = 0.25 * X;
      return 1;
    }
    if (verdict == 2) return 0;
  }
  return 0;
}
// the staged Philox rounds against the block function
int sm_philox_staged_equal(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1)
{
  const bl::U4 ref = bl::philox4x32_10(c0, c1, c2, c3, k0, k1);
  bl::PhiloxState s{bl::U4{c0, c1, c2, c3}, k0, k1};
  bl::philox_rounds<3>(s);
  bl::philox_rounds<4>(s);
  bl::philox_rounds<3>(s);
  return s.c.x == ref.x && s.c.y == ref.y && s.c.z == ref.z && s.c.w == ref.w;
}

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
