// Host build of the kernels' alternating-series and saddle-point attempt bodies
// (bayeslogit_amd/csrc/bl_alt_sm.hpp, bl_sp_sm.hpp) for CPU-side unit tests: the SAME headers the HIP
// kernels inline, compiled as plain C++ and driven task by task as the kernels' work queue drives them.
// Test scaffolding only -- the shipped library never contains this object.
#include "../../bayeslogit_amd/csrc/bl_alt_sm.hpp"
#include "../../bayeslogit_amd/csrc/bl_sp_sm.hpp"
#include "../../bayeslogit_amd/csrc/bl_tables.hpp"
#include "../../bayeslogit_amd/csrc/bl_vtab.hpp"

using namespace bl;

extern "C" {

void hh_alt_par(double h, double z, double* out)
{
  int st = 0;
  const AltPar p = alt_par(h, z, alt_trunc_of(kTruncSchedule, h), st);
  const double v[kAltParDoubles] = {p.h, p.Z, p.t, p.fz, p.lfz, p.p, p.R, p.ic0, p.omc, p.log_m, p.cR};
  for (int i = 0; i < kAltParDoubles; ++i) out[i] = v[i];
}

void hh_sp_par(double n, double z, double* out)
{
  int st = 0;
  const SpPar p = sp_par(n, z, &kVtab[0][0][0], st);
  const double v[kSpParDoubles] = {p.n,   p.Z2h, p.md,    p.mu,  p.pl,  p.b,   p.mdb, p.lmdb,
                                   p.ic0, p.omc, p.log_m, p.cL0, p.cL1, p.cR0, p.cR1};
  for (int i = 0; i < kSpParDoubles; ++i) out[i] = v[i];
}

void hh_sp_vlk(double x, double* out)
{
  sp_vlk(&kVtab[0][0][0], x, log(x), out[0], out[1], out[2]);
}

double hh_cf(double a, double x)
{
  int st = 0;
  return upper_gamma_cf(a, x, st);
}

// rpg_alt over a vector: per observation the two task groups, each run to completion, x = sumA + sumB
void hh_rpg_alt(double* x, const double* h, const double* z, long num, unsigned long long seed, unsigned epoch,
                unsigned long long idx0, int* status)
{
  int st = 0;
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  for (long i = 0; i < num; ++i) {
    if (h[i] == 0.0) { x[i] = 0.0; continue; }
    if (!(h[i] >= 1.0)) { x[i] = 0.0; st |= 2; continue; }
    int nA, nB;
    double hB;
    alt_groups(h[i], nA, hB, nB);
    double acc = 0.0;
    AltTask T;
    if (nA > 0) {
      alt_task_start(T, alt_par(4.0, z[i], alt_trunc_of(kTruncSchedule, 4.0), st), nA, idx0 + (unsigned long long)i, 0u, 0u);
      while (!alt_task_step(T, epoch, k0, k1, st)) {}
      acc += T.sum;
    }
    alt_task_start(T, alt_par(hB, z[i], alt_trunc_of(kTruncSchedule, hB), st), nB, idx0 + (unsigned long long)i, 0u,
                   kAltBlkGroupB);
    while (!alt_task_step(T, epoch, k0, k1, st)) {}
    acc += T.sum;
    x[i] = acc;
  }
  *status = st;
}

void hh_rpg_sp(double* x, const double* h, const double* z, long num, int* iter, unsigned long long seed,
               unsigned epoch, unsigned long long idx0, int* status)
{
  int st = 0;
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  for (long i = 0; i < num; ++i) {
    if (h[i] == 0.0) { x[i] = 0.0; continue; }
    if (!(h[i] >= 1.0)) { x[i] = 0.0; st |= 2; continue; }
    SpTask T;
    sp_task_start(T, sp_par(h[i], z[i], &kVtab[0][0][0], st), idx0 + (unsigned long long)i, 0u);
    while (!sp_task_step(T, &kVtab[0][0][0], 200, epoch, k0, k1, st)) {}
    x[i] = h[i] * 0.25 * T.sm.X;
    if (iter) iter[i] = T.iter;
  }
  *status = st;
}
}
