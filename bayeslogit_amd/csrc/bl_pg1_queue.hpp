// bl_pg1_queue.hpp -- the wavefront work queue that drives the PG(1,z) attempts of
// bl_pg1_sm.hpp over a list of observations (shared by k_rpg_devroye and the Gibbs sweep's
// psi/omega pass).  gfx950 only.
//
// A lane whose draw has completed takes the next unstarted observation of the list (idle lanes
// found with __ballot, numbered with a prefix popcount), so the wave keeps all lanes on the same
// short attempt body instead of waiting for its slowest rejection loop.  The stream belongs
// to the observation (Philox counter = global index), so which lane draws it, and when, does
// not change the result.
//   ZC      : 1 = every listed observation has |z|/2 < 1/t, 2 = every one has |z|/2 >= 1/t
//             (PolyaGamma.cpp:87 vs :103: different left-piece samplers, i.e. different states)
//   ZSRC    : where a lane starting an observation reads z from: 0 = z[base + slot] (global, L2
//             re-read), 1 = x[base + slot] (z parked in the output until the draw overwrites it),
//             2 = z[slot] (a per-wave LDS array)
//   NT      : element type of the shape vector (int for rpg_devroye, double for Logit's n)
//   NLDS    : nvec is a per-wave LDS array indexed by slot (staged by the caller) instead of a
//             global array indexed by base + slot: nothing on the refill path then waits on HBM
#pragma once
#include "bl_pg1_sm.hpp"

namespace bl {

template <int ZC, int ZSRC, typename NT, bool NLDS = false>
__device__ __forceinline__ void devroye_queue(const unsigned short* __restrict__ list, int cnt,
                                              const double* __restrict__ z, const double* __restrict__ sM,
                                              double* __restrict__ x, const NT* __restrict__ nvec, int nscalar,
                                              int64_t base, uint64_t idx0, uint32_t epoch, uint32_t k0, uint32_t k1,
                                              uint64_t lt_mask, int& st_flags)
{
  int next = 0;      // wave-uniform: first unstarted entry of the list
  int q = -1;        // this lane's observation (slot in the chunk), -1 = idle
  int nrem = 0;
  uint32_t c0 = 0, c1 = 0, blk = 0;
  double sum = 0.0;
  Pg1Par par{0.0, 1.0, 0.5, 2.0, 2.0};
  Pg1Lane sm{true, 0.0};
  for (;;) {
    const bool idle = q < 0;
    const uint64_t im = __ballot(idle);
    if (im != 0 && next < cnt) {
      const int cand = next + __popcll(im & lt_mask);
      if (idle && cand < cnt) {
        const int slot = list[cand];
        int n = nvec ? (int)(NLDS ? nvec[slot] : nvec[base + slot]) : nscalar;      // (int) n(i), Logit.hpp:287
        if (n < 1) { n = 1; st_flags |= ST_BAD_SHAPE; }       // PolyaGamma.cpp:128-135 (NTHROW)
        q = slot;
        nrem = n;
        par.Z = fabs(ZSRC == 2 ? z[slot] : ZSRC == 1 ? x[base + slot] : z[base + slot]) * 0.5;
        par.mass = sM[slot];
        pg1_par_finish(par);
        const uint64_t idx = idx0 + (uint64_t)(base + slot);
        c0 = (uint32_t)idx;
        c1 = ctr1_of(idx, DOM_DRAW);
        blk = 0;
        sum = 0.0;
        sm.fresh = true;
      }
      next += __popcll(im);
    }
    if (__ballot(q >= 0) == 0) {
      if (next >= cnt) break;
      continue;
    }
    if (q >= 0) {
      const U4 o = philox4x32_10(c0, c1, epoch, blk, k0, k1);
      blk += 1;
      if (pg1_attempt<true, ZC>(sm, par, u52(o.x, o.y), u52(o.z, o.w), st_flags)) {
        sum += 0.25 * sm.X;
        if (--nrem == 0) { x[base + q] = sum; q = -1; }
      }
      if (blk > 4000000u) { st_flags |= ST_ITER_CAP; x[base + q] = sum; q = -1; }
    }
  }
}

}  // namespace bl
