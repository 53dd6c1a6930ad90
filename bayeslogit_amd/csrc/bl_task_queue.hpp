// bl_task_queue.hpp -- the wavefront work queue that drives the alternating-series and saddle-point
// attempt bodies (bl_alt_sm.hpp, bl_sp_sm.hpp) over the observations of rpg_alt / rpg_sp / rpg_hybrid
// (Code/C/LogitWrapper.cpp:87-167).  gfx950 only.
//
// A wave owns chunks of kTqChunk consecutive observations.  Per chunk:
//   scan    (all lanes busy) coalesced read of the shapes, the members of this launch's sampler class
//           COMPACTED into a task list in LDS (__ballot + prefix popcount).  A task is a run of draws with
//           one set-up: a saddle-point observation is one task; an alternating-series observation is one
//           or two (PolyaGammaAlt::draw's floor((h-1)/4) draws at shape 4, and its remainder);
//   set-up  (all lanes busy) 64 tasks at a time, lane j computes task j's constants -- mixture weight,
//           tangent lines, Dagpunar's constants: the part of the reference's draw() that draws nothing --
//           and stages them in LDS;
//   draw    the attempt body under the queue: a lane whose task has completed takes the next staged
//           task (idle lanes found with __ballot, numbered by prefix popcount) and copies its constants
//           into registers, so no lane waits on another lane's rejection loop; lanes still inside a task
//           when the 64 staged ones have all been started stay IN FLIGHT (their state is in registers)
//           while the wave stages the next 64, so the queue drains once per launch.
// The Philox stream belongs to the observation (counter = global index; an alternating-series
// observation's second task reads from block 2^31 on), so which lane draws a task, and when, does not
// change the result.  The two tasks of an alternating-series observation add their sums into x[] with
// one fp64 atomic each; the wave that scans the chunk stores the 0 they are added to (no zeroing launch over x), and
// 0 + a + b = 0 + b + a bit for bit.  Every element of x is written by exactly one kernel: members by their class's
// launch, h == 0 / refused shapes (rpg_alt, rpg_sp) and the b <= 0 branch (rpg_hybrid's first pass) by the scan.
#pragma once
#include "bl_alt_sm.hpp"
#include "bl_pg_hybrid.hpp"
#include "bl_sp_sm.hpp"
#include "bl_tables.hpp"
#include "bl_vtab.hpp"

namespace bl {

constexpr int kTqBlock = 256;      // 4 wavefronts
constexpr int kTqChunk = 512;      // observations per wave per chunk (task list: <= 2 per observation)

// ---- saddle-point policy: one task per observation
struct SpPolicy {
  static constexpr int kCls = CLS_SP;
  static constexpr int kWavesPerSimd = 3;      // 165 registers per lane
  static constexpr int kStageDoubles = kSpParDoubles;
  static constexpr int kMaxTasksPerObs = 1;
  static constexpr bool kNeedsVtab = true;
  using Task = SpTask;

  __device__ static __forceinline__ int groups(double h, int& nA) { nA = 0; return 1; }

  __device__ static __forceinline__ void setup(double* __restrict__ st, int slot, double h, double z, int /*group*/,
                                               const double* __restrict__ vt, int& ndraws, uint32_t& blk0, int& status)
  {
    const SpPar p = sp_par(h, z, vt, status);
    const double v[kSpParDoubles] = {p.n,   p.Z2h, p.md,    p.mu,  p.pl,  p.b,   p.mdb, p.lmdb,
                                     p.ic0, p.omc, p.log_m, p.cL0, p.cL1, p.cR0, p.cR1};
#pragma unroll
    for (int f = 0; f < kSpParDoubles; ++f) st[f * 64 + slot] = v[f];
    ndraws = 1;
    blk0 = 0;
  }

  __device__ static __forceinline__ void start(Task& T, const double* __restrict__ st, int slot, int /*ndraws*/,
                                               uint32_t /*blk0*/, uint64_t idx)
  {
    double v[kSpParDoubles];
#pragma unroll
    for (int f = 0; f < kSpParDoubles; ++f) v[f] = st[f * 64 + slot];
    const SpPar p{v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9], v[10], v[11], v[12], v[13], v[14]};
    sp_task_start(T, p, idx, DOM_DRAW);
  }

  __device__ static __forceinline__ bool step(Task& T, const double* __restrict__ vt, uint32_t epoch, uint32_t k0,
                                              uint32_t k1, int& status)
  {
    return sp_task_step(T, vt, 200, epoch, k0, k1, status);      // maxiter = 200, PolyaGammaSP.h:53
  }

  __device__ static __forceinline__ void emit(const Task& T, double* __restrict__ x, int* __restrict__ iter,
                                              int64_t row, bool /*two*/)
  {
    x[row] = T.par.n * 0.25 * T.sm.X;                            // PolyaGammaSP.cpp:262
    if (iter) iter[row] = T.iter;                                // LogitWrapper.cpp:117
  }
};

// ---- alternating-series policy: group B (the remainder) always, group A (draws at shape 4) when h >= 5
struct AltPolicy {
  static constexpr int kCls = CLS_ALT;
  static constexpr int kWavesPerSimd = 3;      // 147 registers per lane (at 4: 128 with 4 spilled doubles, and no faster)
  static constexpr int kStageDoubles = kAltParDoubles;
  static constexpr int kMaxTasksPerObs = 2;
  static constexpr bool kNeedsVtab = false;
  using Task = AltTask;

  __device__ static __forceinline__ int groups(double h, int& nA)
  {
    double hB;
    int nB;
    alt_groups(h, nA, hB, nB);
    return nA > 0 ? 2 : 1;
  }

  __device__ static __forceinline__ void setup(double* __restrict__ st, int slot, double h, double z, int group,
                                               const double* __restrict__ /*vt*/, int& ndraws, uint32_t& blk0,
                                               int& status)
  {
    int nA, nB;
    double hB;
    alt_groups(h, nA, hB, nB);
    const double hs = group ? 4.0 : hB;                          // group 1 = A, PolyaGammaAlt.cpp:216-217
    const AltPar p = alt_par(hs, z, alt_trunc_of(kTruncSchedule, hs), status);
    const double v[kAltParDoubles] = {p.h, p.Z, p.t, p.fz, p.lfz, p.p, p.R, p.ic0, p.omc, p.log_m, p.cR};
#pragma unroll
    for (int f = 0; f < kAltParDoubles; ++f) st[f * 64 + slot] = v[f];
    ndraws = group ? nA : nB;
    blk0 = group ? 0u : kAltBlkGroupB;
  }

  __device__ static __forceinline__ void start(Task& T, const double* __restrict__ st, int slot, int ndraws,
                                               uint32_t blk0, uint64_t idx)
  {
    double v[kAltParDoubles];
#pragma unroll
    for (int f = 0; f < kAltParDoubles; ++f) v[f] = st[f * 64 + slot];
    const AltPar p{v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9], v[10]};
    alt_task_start(T, p, ndraws, idx, DOM_DRAW, blk0);
  }

  __device__ static __forceinline__ bool step(Task& T, const double* __restrict__ /*vt*/, uint32_t epoch,
                                              uint32_t k0, uint32_t k1, int& status)
  {
    return alt_task_step(T, epoch, k0, k1, status);
  }

  __device__ static __forceinline__ void emit(const Task& T, double* __restrict__ x, int* __restrict__ /*iter*/,
                                              int64_t row, bool two)
  {
    if (two) atomicAdd(&x[row], T.sum);                          // x = sumA + sumB, PolyaGammaAlt.cpp:216-222
    else x[row] = T.sum;
  }
};

// hybrid != 0: the members are the observations whose shape takes this sampler in rpg_hybrid
// (LogitWrapper.cpp:142-161); the other classes' launches write the rest of x.  hybrid == 1 (the first class pass; the
// scan reads every shape anyway): the members of EVERY class are counted into cls_count[6] -- the later passes
// (hybrid == 2) return at once when their class is empty -- and the b <= 0 branch's zeros (:159-161) are written.
// hybrid == 0 (rpg_alt, rpg_sp): every h != 0 is a member (LogitWrapper.cpp:95-98, :116-120); h == 0 gives 0; a shape
// below 1 is refused (PolyaGammaAlt.cpp:207-210: message and 0; the saddle-point sampler only warns there,
// PolyaGammaSP.cpp:171, and goes on into a truncated gamma of shape < 1, which the reference's own prototype of that
// variate defines as NA, Code/R/Ch.R:91: refused here as well) and flagged; the scan writes those zeros.
template <class P>
__global__ __launch_bounds__(kTqBlock, P::kWavesPerSimd) void k_rpg_tasks(double* __restrict__ x, const double* __restrict__ h,
                                                          const double* __restrict__ z, int64_t num,
                                                          int* __restrict__ iter, uint64_t seed, uint32_t epoch,
                                                          uint64_t idx0, int hybrid,
                                                          unsigned long long* __restrict__ cls_count,
                                                          int* __restrict__ status)
{
  if (hybrid == 2 && cls_count[P::kCls] == 0) return;       // (uniform) nothing of this class in the vector
  constexpr int NW = kTqBlock / 64;
  constexpr int kList = kTqChunk * P::kMaxTasksPerObs;
  __shared__ unsigned short sList[NW][kList];               // (offset in chunk) << 1 | group
  __shared__ double sStage[NW][P::kStageDoubles * 64];
  __shared__ unsigned short sMetaK[NW][64];                 // per staged task: offset in chunk
  __shared__ int sMetaN[NW][64];                            //                  draws
  __shared__ uint32_t sMetaB[NW][64];                       //                  first Philox block | two-task flag
  __shared__ double sVt[P::kNeedsVtab ? kVtabDoubles : 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  if (P::kNeedsVtab) {
    const double* src = &kVtab[0][0][0];
    for (int i = threadIdx.x; i < kVtabDoubles; i += kTqBlock) sVt[i] = src[i];
    __syncthreads();
  }
  const double* vt = sVt;
  unsigned short* list = sList[wave];
  double* stage = sStage[wave];
  int st_flags = 0;
  typename P::Task T;
  int64_t row = -1;          // -1: idle
  bool two = false;
  __shared__ unsigned sCls[NW][8];                          // hybrid == 1: members of every class this wave has scanned
  if (lane < 8) sCls[wave][lane] = 0u;                      // (in LDS: six more live registers cost the saddle-point kernel a spill)

  const int64_t nchunks = (num + kTqChunk - 1) / kTqChunk;
  for (int64_t ch = (int64_t)blockIdx.x * NW + wave; ch < nchunks; ch += (int64_t)gridDim.x * NW) {
    const int64_t base = ch * kTqChunk;
    const int cnt = (int)((num - base) < kTqChunk ? (num - base) : kTqChunk);
    // ---- scan: task list of the chunk
    int nT = 0;
#pragma unroll 1
    for (int j0 = 0; j0 < kTqChunk / 64; j0 += 8) {
      if (j0 * 64 >= cnt) break;
      double hk[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = (j0 + j) * 64 + lane;
        hk[j] = k < cnt ? h[base + k] : 0.0;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = (j0 + j) * 64 + lane;
        bool mine, zero;
        if (hybrid) {
          const int cls = k < cnt ? pg_class(hk[j]) : -1;
          mine = cls == P::kCls;
          zero = hybrid == 1 && cls == CLS_ZERO;
          if (hybrid == 1) {
#pragma unroll
            for (int c = 0; c < 6; ++c) {
              const unsigned nc = (unsigned)__popcll(__ballot(cls == c));
              if (lane == c) sCls[wave][c] += nc;
            }
          }
        } else {
          mine = k < cnt && hk[j] != 0.0;
          if (mine && !(hk[j] >= 1.0)) { mine = false; st_flags |= ST_BAD_SHAPE; }
          zero = k < cnt && !mine;
        }
        int nA = 0;
        const bool second = mine && P::groups(hk[j], nA) == 2;
        if (zero || second) x[base + k] = 0.0;     // the value itself, or what the observation's two tasks add their sums to
        const uint64_t mB = __ballot(mine), mA = __ballot(second);
        if (mine) list[nT + __popcll(mB & lt_mask)] = (unsigned short)(k << 1);
        nT += __popcll(mB);
        if (P::kMaxTasksPerObs > 1) {
          if (second) list[nT + __popcll(mA & lt_mask)] = (unsigned short)((k << 1) | 1);
          nT += __popcll(mA);
        }
      }
    }
    // (workgroup scope: the zeros above have reached L2, where this wave's atomics on them execute, before any is issued)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // ---- 64 tasks at a time: set-up, then the queue
    for (int i0 = 0; i0 < nT; i0 += 64) {
      const int nb = nT - i0 < 64 ? nT - i0 : 64;
      if (lane < nb) {
        const int e = list[i0 + lane];
        const int k = e >> 1;
        int ndraws;
        uint32_t blk0;
        const double hh = h[base + k];
        P::setup(stage, lane, hh, z[base + k], e & 1, vt, ndraws, blk0, st_flags);
        int nA = 0;
        const bool tw = P::groups(hh, nA) == 2;
        sMetaK[wave][lane] = (unsigned short)k;
        sMetaN[wave][lane] = ndraws;
        sMetaB[wave][lane] = blk0 | (tw ? 1u : 0u);               // block numbers of a task stay far below 2^31 - 1
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      int next = 0;        // wave-uniform: first unstarted staged task
      for (;;) {
        const bool idle = row < 0;
        const uint64_t im = __ballot(idle);
        if (im != 0 && next < nb) {
          const int cand = next + __popcll(im & lt_mask);
          if (idle && cand < nb) {
            const uint32_t mb = sMetaB[wave][cand];
            row = base + (int64_t)sMetaK[wave][cand];
            two = (mb & 1u) != 0;
            P::start(T, stage, cand, sMetaN[wave][cand], mb & ~1u, idx0 + (uint64_t)row);
          }
          next += __popcll(im);
        }
        // every staged task has been started and some lanes found none: no step for a partly filled wave -- the lanes inside
        // a task stay in flight and the next 64 staged tasks fill the idle ones (bl_pg1_queue.hpp has the measurement)
        if (next >= nb && __ballot(row >= 0) != ~0ull) break;
        if (row >= 0) {
          if (P::step(T, vt, epoch, k0, k1, st_flags)) {
            P::emit(T, x, iter, row, two);
            row = -1;
          }
        }
        if (next >= nb) break;       // every staged task has been started; lanes inside a task stay in flight
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  // ---- drain
  while (__ballot(row >= 0) != 0) {
    if (row >= 0) {
      if (P::step(T, vt, epoch, k0, k1, st_flags)) {
        P::emit(T, x, iter, row, two);
        row = -1;
      }
    }
  }
  if (st_flags) atomicOr(status, st_flags);
  if (hybrid == 1 && lane < 6 && sCls[wave][lane]) atomicAdd(&cls_count[lane], (unsigned long long)sCls[wave][lane]);
}

}  // namespace bl
