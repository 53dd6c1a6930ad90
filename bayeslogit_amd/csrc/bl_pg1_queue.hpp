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
//   DOM     : stream domain (DOM_DRAW for rpg_devroye, DOM_OMEGA for a chain's omega draws)
//   NLDS    : nvec is a per-wave LDS array indexed by slot (staged by the caller) instead of a
//             global array indexed by base + slot: nothing on the refill path then waits on HBM
#pragma once
#include "bl_pg1_sm.hpp"

namespace bl {

// A lane's in-flight observation.  Self-contained: once an observation has been started nothing of
// the list it came from (index list, staged mass, z) is read again, so the state survives the caller
// rebuilding those for its next chunk.
// (kPg1BlkCap, Philox blocks per PG(1,z) DRAW: bl_pg1_sm.hpp)
struct Pg1Slot {
  int64_t row = -1;         // index into x[] / global observation offset; -1 = idle
  int nrem = 0;             // PG(1,z) draws still to add (PolyaGamma::draw(int n, ...), :126-140)
  uint32_t c0 = 0, c1 = 0, blk = 0;   // Philox counter words of the observation's stream, next block
  uint32_t cap = 0;         // first block the draw in progress may not use (blk at its start + kPg1BlkCap)
  double sum = 0.0;
  Pg1Par par{0.0, 1.0, 0.5, 2.0, 2.0};
  Pg1Lane sm{true, 0.0};
};

// Runs the queue over `list[0..cnt)`.  drain = false: return as soon as the list is exhausted, leaving
// the lanes that are still inside a draw in flight in L (the caller comes back with the next list, or
// with an empty list and drain = true); drain = true: run until every lane is idle.
template <int ZC, int ZSRC, typename NT, bool NLDS = false, uint32_t DOM = DOM_DRAW>
__device__ __forceinline__ void devroye_queue_run(Pg1Slot& L, bool drain, const unsigned short* __restrict__ list,
                                                  int cnt, const double* __restrict__ z,
                                                  const double* __restrict__ sM, double* __restrict__ x,
                                                  const NT* __restrict__ nvec, int nscalar, int64_t base,
                                                  uint64_t idx0, uint32_t epoch, uint32_t k0, uint32_t k1,
                                                  uint64_t lt_mask, int& st_flags,
                                                  const uint32_t* __restrict__ rowoff = nullptr)
{
  // rowoff (or null): the observation of list entry `slot` is row base + rowoff[slot] instead of base + slot
  // (a list gathered from scattered rows; z, mass and n are still indexed by slot)
  int next = 0;      // wave-uniform: first unstarted entry of the list
  for (;;) {
    const bool idle = L.row < 0;
    const uint64_t im = __ballot(idle);
    if (im != 0 && next < cnt) {
      const int cand = next + __popcll(im & lt_mask);
      if (idle && cand < cnt) {
        const int slot = list[cand];
        const int64_t row = base + (rowoff ? (int64_t)rowoff[slot] : (int64_t)slot);
        int n = nvec ? (int)(NLDS ? nvec[slot] : nvec[row]) : nscalar;              // (int) n(i), Logit.hpp:287
        if (n < 1) { n = 1; st_flags |= ST_BAD_SHAPE; }       // PolyaGamma.cpp:128-135 (NTHROW)
        L.row = row;
        L.nrem = n;
        L.par.Z = fabs(ZSRC == 2 ? z[slot] : ZSRC == 1 ? x[row] : z[row]) * 0.5;
        L.par.mass = sM[slot];
        pg1_par_finish(L.par);
        const uint64_t idx = idx0 + (uint64_t)row;
        L.c0 = (uint32_t)idx;
        L.c1 = ctr1_of(idx, DOM);
        L.blk = 0;
        L.cap = kPg1BlkCap;
        L.sum = 0.0;
        L.sm.fresh = true;
      }
      next += __popcll(im);
    }
    const uint64_t busy = __ballot(L.row >= 0);
    if (busy == 0) {
      if (next >= cnt) break;
      continue;
    }
    // The list is exhausted and some lanes found nothing to start: an attempt now would run the whole body for a partly
    // filled wave (measured, round 3: 38 % of the vector lane-slots of the PG(1,z) kernel were such idle lanes, most of
    // them in the small class-2 lists).  The lanes inside a draw stay in flight instead -- the caller's next list fills the
    // idle ones -- so every attempt but the last few of a launch runs on a full wave.  Same draws: a draw is a function of
    // its stream, not of when it is made.
    if (!drain && next >= cnt && busy != ~0ull) break;
    if (L.row >= 0) {
      const U4 o = philox4x32_10(L.c0, L.c1, epoch, L.blk, k0, k1);
      L.blk += 1;
      if (pg1_attempt<true, ZC>(L.sm, L.par, u52(o.x, o.y), u52(o.z, o.w), st_flags)) {
        L.sum += 0.25 * L.sm.X;
        if (--L.nrem == 0) { x[L.row] = L.sum; L.row = -1; }
        L.cap = L.blk + kPg1BlkCap;          // per draw: an observation's n is not limited by the cap
      }
      // the reference's loops are uncapped (PolyaGamma.cpp:167,181); a lane that never exits would hang its wave
      if (L.row >= 0 && L.blk == L.cap) { st_flags |= ST_ITER_CAP; x[L.row] = L.sum; L.row = -1; }
    }
    if (!drain && next >= cnt) break;
  }
}

// One list, start to finish (every lane idle on return).
template <int ZC, int ZSRC, typename NT, bool NLDS = false, uint32_t DOM = DOM_DRAW>
__device__ __forceinline__ void devroye_queue(const unsigned short* __restrict__ list, int cnt,
                                              const double* __restrict__ z, const double* __restrict__ sM,
                                              double* __restrict__ x, const NT* __restrict__ nvec, int nscalar,
                                              int64_t base, uint64_t idx0, uint32_t epoch, uint32_t k0, uint32_t k1,
                                              uint64_t lt_mask, int& st_flags)
{
  Pg1Slot L;
  devroye_queue_run<ZC, ZSRC, NT, NLDS, DOM>(L, true, list, cnt, z, sM, x, nvec, nscalar, base, idx0, epoch, k0, k1,
                                        lt_mask, st_flags);
}

}  // namespace bl
