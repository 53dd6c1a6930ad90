// kernels_pg.hip -- the rpg_* family on MI355X: one PG(b_i, z_i) draw per lane,
// each observation on its own Philox stream.  Replaces the serial loops of
// Code/C/LogitWrapper.cpp:39-167.  gfx950 only.
#include "bl_host.hpp"
#include <mutex>
#include "bl_pg_hybrid.hpp"
#include "bl_pg1_queue.hpp"

namespace {

using namespace bl;

constexpr int kBlock = 256;        // 4 wavefronts
constexpr int kMaxBlocks = 256 * 8;  // 256 CUs x 8 resident blocks, grid-stride beyond

// ---------------------------------------------------------------- rpg_devroye
// Wavefront work queue.  |z|/2 < 1/t and |z|/2 >= 1/t take different left-piece samplers (PolyaGamma.cpp:87 vs :103): two
// attempt bodies, two index lists per chunk, ONE launch (round 1 and most of round 2 used one launch per class: z was read
// and classified twice; 45.5 against 46.8 G draws/s on the same box).  Each wave owns chunks of kChunk consecutive
// observations.
//   phase 1 (all 64 lanes busy): coalesced load of z (the chunk's loads issued before the first is used), the observations
//            COMPACTED by class into two index lists (ballot + prefix popcount) and their proposal mass staged in LDS
//            (class 1: a 12-term polynomial in z^2 computed in the same sweep; class 2: two Chebyshev sums, evaluated
//            over its compacted list so that all lanes are busy); z and n are staged too, so a lane starting an
//            observation waits on nothing from HBM;
//   phase 2: the attempt bodies of bl_pg1_sm.hpp under the work queue of bl_pg1_queue.hpp, one class after the other.  A
//            lane whose draw has completed takes the next unstarted observation of the list (idle lanes found with
//            __ballot, numbered with a prefix popcount); lanes still inside a draw when a list runs out stay IN FLIGHT
//            (one in-flight slot per class and lane) while the wave goes on, so each queue drains once per launch, not
//            once per chunk.
// The stream belongs to the observation, so which lane draws it, and when, does not change the result.
// This file is built with machine-LICM off (bayeslogit_amd/build.py): hoisting the attempt bodies' fp64 polynomial
// constants out of the queue loop held 168 / 256 registers per lane in the per-class kernels (3 / 2 waves per SIMD);
// without it they took 68 / 78, and this kernel, with both bodies and two in-flight slots, 94: five workgroups per CU
// (six: 80 registers + 80 bytes of scratch, 42 G draws/s).  LDS: 24 KB per workgroup at 256-observation chunks.
constexpr int kChunk = 256;                          // observations per wave per chunk

constexpr int kDevOcc = 5;                           // resident workgroups per CU (94 registers; 6 spills: 42 G draws/s)
__global__ __launch_bounds__(kBlock, kDevOcc) void k_rpg_devroye(double* __restrict__ x,
                                                                          const int* __restrict__ nvec, int nscalar,
                                                                          const double* __restrict__ z, int64_t num,
                                                                          uint64_t seed, uint32_t epoch, uint64_t idx0,
                                                                          int* __restrict__ status)
{
  __shared__ double sM[kBlock / 64][kChunk];
  __shared__ double sZ[kBlock / 64][kChunk];
  __shared__ int sN[kBlock / 64][kChunk];
  __shared__ unsigned short sIdx1[kBlock / 64][kChunk];
  __shared__ unsigned short sIdx2[kBlock / 64][kChunk];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  const int64_t nchunks = (num + kChunk - 1) / kChunk;
  int st_flags = 0;
  Pg1Slot L1, L2;      // this lane's in-flight observation of either class
  for (int64_t ch = (int64_t)blockIdx.x * (kBlock / 64) + wave; ch < nchunks; ch += (int64_t)gridDim.x * (kBlock / 64)) {
    const int64_t base = ch * kChunk;
    const int cnt = (int)((num - base) < kChunk ? (num - base) : kChunk);
    int n1 = 0, n2 = 0;       // wave-uniform list lengths
    double zk[kChunk / 64];
    int nk[kChunk / 64];
#pragma unroll
    for (int j = 0; j < kChunk / 64; ++j) {
      const int k = j * 64 + lane;
      zk[j] = k < cnt ? z[base + k] : 0.0;
      nk[j] = nvec ? (k < cnt ? nvec[base + k] : 0) : nscalar;
    }
#pragma unroll
    for (int j = 0; j < kChunk / 64; ++j) {
      const int k = j * 64 + lane;
      bool m1 = false, m2 = false;
      if (k < cnt) {
        const int n = nk[j];
        if (n == 0) {
          x[base + k] = 0.0;                           // LogitWrapper.cpp:74-77
        } else {
          const double Z = fabs(zk[j]) * 0.5;          // PolyaGamma.cpp:154
          m1 = kSmTRecip > Z;                          // PolyaGamma.cpp:87
          m2 = !m1;
          sZ[wave][k] = zk[j];
          sN[wave][k] = n;
          if (m1) sM[wave][k] = pg1_mass_small(Z, kSmPiSq8 + 0.5 * Z * Z);
        }
      }
      const uint64_t b1 = __ballot(m1), b2 = __ballot(m2);
      if (m1) sIdx1[wave][n1 + __popcll(b1 & lt_mask)] = (unsigned short)k;
      if (m2) sIdx2[wave][n2 + __popcll(b2 & lt_mask)] = (unsigned short)k;
      n1 += __popcll(b1);
      n2 += __popcll(b2);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int i0 = 0; i0 < n2; i0 += 64) {               // class 2's mass over its compacted list
      const int i = i0 + lane;
      if (i < n2) {
        const int k = sIdx2[wave][i];
        const double Z = fabs(sZ[wave][k]) * 0.5;
        sM[wave][k] = pg1_mass(Z, kSmPiSq8 + 0.5 * Z * Z);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    devroye_queue_run<1, 2, int, true>(L1, false, &sIdx1[wave][0], n1, sZ[wave], sM[wave], x, sN[wave], 1, base, idx0, epoch,
                                       k0, k1, lt_mask, st_flags);
    devroye_queue_run<2, 2, int, true>(L2, false, &sIdx2[wave][0], n2, sZ[wave], sM[wave], x, sN[wave], 1, base, idx0, epoch,
                                       k0, k1, lt_mask, st_flags);
    __builtin_amdgcn_wave_barrier();
  }
  devroye_queue_run<1, 2, int, true>(L1, true, nullptr, 0, nullptr, nullptr, x, nullptr, 1, 0, idx0, epoch, k0, k1, lt_mask,
                                     st_flags);
  devroye_queue_run<2, 2, int, true>(L2, true, nullptr, 0, nullptr, nullptr, x, nullptr, 1, 0, idx0, epoch, k0, k1, lt_mask,
                                     st_flags);
  if (st_flags) atomicOr(status, st_flags);
}

// ------------------------------------------------------------------ rpg_gamma
// (rpg_alt and rpg_sp: k_rpg_tasks<AltPolicy> / <SpPolicy>, bl_task_queue.hpp)
__global__ __launch_bounds__(kBlock) void k_rpg_gamma(double* __restrict__ x, const double* __restrict__ h,
                                                      const double* __restrict__ z, int64_t num, int trunc,
                                                      uint64_t seed, uint32_t epoch, uint64_t idx0)
{
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < num; i += stride) {
    double out = 0.0;
    if (h[i] != 0.0) {
      Stream r;
      r.init(seed, idx0 + (uint64_t)i, DOM_DRAW, epoch);
      out = pg_draw_sum_of_gammas(h[i], z[i], trunc, r);
    }
    x[i] = out;
  }
}

// ----------------------------------------------------------------- rpg_hybrid
// The five branches of LogitWrapper.cpp:142-161 have very different register footprints and trip
// counts; a lane-per-observation kernel that switches per lane would keep the union of all branches'
// registers live and serialise every branch in every wavefront.  The launch is therefore split by class:
// each pass is a separate kernel that contains the code of ONE class only and skips the observations
// of the others.  Streams are per observation, so the result is identical to the unsplit loop.
//   b > 13  saddle point          k_rpg_tasks<SpPolicy>   (bl_task_queue.hpp)
//   b > 1   alternating series    k_rpg_tasks<AltPolicy>
//   b = 1,2 Devroye; b > 170 normal approximation; 0 < b < 1 sum of gammas: k_rpg_hybrid_class below --
//           the class's observations (scattered in the vector) COMPACTED per 4096-observation chunk into an
//           LDS index list (ballot + prefix popcount; Devroye: b = 1 before b = 2) and drawn 64 at a time.
//   b <= 0  0: the zeroing launch in front of the passes.
constexpr int kChunkH = 4096;

template <int CLS>
__global__ __launch_bounds__(kBlock, 3) void k_rpg_hybrid_class(double* __restrict__ x,
                                                             const double* __restrict__ h,
                                                             const double* __restrict__ z, int64_t num,
                                                             uint64_t seed, uint32_t epoch, uint64_t idx0,
                                                             const unsigned long long* __restrict__ cls_count,
                                                             int* __restrict__ status)
{
  if (cls_count && cls_count[CLS] == 0) return;          // (uniform) the first class pass found no member of this class
  static_assert(CLS == CLS_DEVROYE || CLS == CLS_NORMAL || CLS == CLS_GAMMA, "the other classes run as tasks");
  constexpr int NKEY = CLS == CLS_DEVROYE ? 2 : 1;      // Devroye: one draw (b = 1) or two (b = 2)
  __shared__ unsigned short sIdx[kBlock / 64][kChunkH];
  __shared__ unsigned char sKey[kBlock / 64][NKEY > 1 ? kChunkH : 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  int st = 0;
  const int64_t nchunks = (num + kChunkH - 1) / kChunkH;
  for (int64_t ch = (int64_t)blockIdx.x * (kBlock / 64) + wave; ch < nchunks; ch += (int64_t)gridDim.x * (kBlock / 64)) {
    const int64_t base = ch * kChunkH;
    const int cnt = (int)((num - base) < kChunkH ? (num - base) : kChunkH);
    // pass A: class membership and cost key of every observation; members counted per key
    int nkey[NKEY];
#pragma unroll
    for (int q = 0; q < NKEY; ++q) nkey[q] = 0;
#pragma unroll 1
    for (int j0 = 0; j0 < kChunkH / 64; j0 += 8) {
      if (j0 * 64 >= cnt) break;
      double hk[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = (j0 + j) * 64 + lane;
        hk[j] = k < cnt ? h[base + k] : -1.0;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = (j0 + j) * 64 + lane;
        const bool mine = k < cnt && pg_class(hk[j]) == CLS;
        const int key = mine ? (CLS == CLS_DEVROYE && hk[j] == 2.0 ? 1 : 0) : NKEY;
        if (NKEY > 1) {
          if (k < cnt) sKey[wave][k] = (unsigned char)key;
#pragma unroll
          for (int q = 0; q < NKEY; ++q) nkey[q] += __popcll(__ballot(key == q));
        } else {
          const uint64_t mm = __ballot(mine);
          if (mine) sIdx[wave][nkey[0] + __popcll(mm & lt_mask)] = (unsigned short)k;
          nkey[0] += __popcll(mm);
        }
      }
    }
    int total = nkey[0];
    if (NKEY > 1) {
      // pass B: fill the list, bucket after bucket
      int off[NKEY];
      off[0] = 0;
#pragma unroll
      for (int q = 1; q < NKEY; ++q) off[q] = off[q - 1] + nkey[q - 1];
      total = off[NKEY - 1] + nkey[NKEY - 1];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      for (int k0s = 0; k0s < cnt; k0s += 64) {
        const int k = k0s + lane;
        const int key = k < cnt ? (int)sKey[wave][k] : NKEY;
#pragma unroll
        for (int q = 0; q < NKEY; ++q) {
          const uint64_t mm = __ballot(key == q);
          if (key == q) sIdx[wave][off[q] + __popcll(mm & lt_mask)] = (unsigned short)k;
          off[q] += __popcll(mm);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // draw, 64 list entries at a time (a bucket boundary may fall inside a batch: only that batch mixes keys)
    for (int i0 = 0; i0 < total; i0 += 64) {
      const int i = i0 + lane;
      if (i < total) {
        const int k = sIdx[wave][i];
        const double b = h[base + k];
        double out;
        if (CLS == CLS_DEVROYE) {
          out = pg1_draw_n((int)b, z[base + k], seed, idx0 + (uint64_t)(base + k), DOM_DRAW, epoch, st);
        } else {
          Stream r;
          r.init(seed, idx0 + (uint64_t)(base + k), DOM_DRAW, epoch);
          out = pg_hybrid_class(CLS, b, z[base + k], r, st);
        }
        x[base + k] = out;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (st) atomicOr(status, st);
}

// ------------------------------------------------------------ synthetic data
__global__ __launch_bounds__(kBlock) void k_fill_unif(double* __restrict__ out, int64_t num, double lo, double hi,
                                                      uint64_t seed, uint32_t epoch, uint64_t idx0)
{
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < num; i += stride) {
    Stream r;
    r.init(seed, idx0 + (uint64_t)i, DOM_DATA, epoch);
    out[i] = r.flat(lo, hi);
  }
}

__global__ __launch_bounds__(kBlock) void k_fill_norm(double* __restrict__ out, int64_t num, double mean, double sd,
                                                      uint64_t seed, uint32_t epoch, uint64_t idx0)
{
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < num; i += stride) {
    Stream r;
    r.init(seed, idx0 + (uint64_t)i, DOM_DATA, epoch);
    out[i] = r.norm(mean, sd);
  }
}

__global__ __launch_bounds__(kBlock) void k_fill_shape(double* __restrict__ out, int64_t num, int kmax,
                                                       uint64_t seed, uint32_t epoch, uint64_t idx0)
{
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < num; i += stride) {
    const uint64_t idx = idx0 + (uint64_t)i;
    const U4 o = philox4x32_10((uint32_t)idx, ((uint32_t)(idx >> 32) & 0x00FFFFFFu) | (DOM_DATA << 24), epoch, 0,
                               (uint32_t)seed, (uint32_t)(seed >> 32));
    out[i] = 1.0 + (double)(o.x % (uint32_t)kmax);
  }
}

__global__ __launch_bounds__(kBlock) void k_fill_logit_y(double* __restrict__ y, const double* __restrict__ tX,
                                                         const double* __restrict__ beta, int64_t N, int P,
                                                         uint64_t seed, uint32_t epoch, uint64_t idx0)
{
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < N; i += stride) {
    const double* xr = tX + (size_t)i * P;
    double s = 0.0;
    for (int j = 0; j < P; ++j) s += xr[j] * beta[j];
    Stream r;
    r.init(seed, idx0 + (uint64_t)i, DOM_DATA, epoch);
    y[i] = r.unif() < 1.0 / (1.0 + exp(-s)) ? 1.0 : 0.0;
  }
}

int check_args(const void* a, const void* b, int64_t num)
{
  if (!blh::ensure_device()) return BL_ERR_NO_DEVICE;
  if (num < 0 || (num > 0 && (!a || !b))) {
    blh::set_error("null pointer or negative length");
    return BL_ERR_ARG;
  }
  return BL_OK;
}

// ------------------------------------------------ diagnostic: sustained v_mfma_f64_16x16x4_f64 rate
// Ten independent accumulators per wave (the X'Omega X pass's count), no memory traffic.  bench.py quotes the
// measured rate next to that pass's: the matrix pipe sustains well under its nominal 78.6 TFLOP/s.
typedef double diag_d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_diag_mfma_f64(double* __restrict__ out, int iters, double a0, double b0)
{
  diag_d4 acc[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) acc[i] = diag_d4{0.0, 0.0, 0.0, 0.0};
  const double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 10; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double sum = 0.0;
#pragma unroll
  for (int i = 0; i < 10; ++i) sum += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

// the same loop on v_mfma_f64_4x4x4_4b_f64 (512 flops), the instruction the P >= 64 X' Omega X kernels use
__global__ __launch_bounds__(256) void k_diag_mfma_f64_small(double* __restrict__ out, int iters, double a0, double b0)
{
  double acc[32];             // (a result comes back later than ten issue slots: 32 independent chains)
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = 0.0;
  const double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  double sum = 0.0;
#pragma unroll
  for (int i = 0; i < 32; ++i) sum += acc[i];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

}  // namespace

// =============================================================== Part 2 (device)
extern "C" {

int bl_rpg_devroye_dev(double* x, const int* n_vec, int n_scalar, const double* z, int64_t num, uint64_t seed,
                       uint32_t epoch, uint64_t idx0, void* stream)
{
  if (int rc = check_args(x, z, num)) return rc;
  if (num == 0) return BL_OK;
  hipStream_t s = (hipStream_t)stream;
  const int64_t wg_chunks = (num + 4 * kChunk - 1) / (4 * kChunk);
  hipLaunchKernelGGL(k_rpg_devroye, dim3(blh::grid_for(wg_chunks, 1, 256 * kDevOcc)), dim3(kBlock), 0, s, x, n_vec, n_scalar, z,
                     num, seed, epoch, idx0, blh::status_word(s));
  BL_HIP_TRY(hipGetLastError());
  return BL_OK;
}

int bl_rpg_gamma_dev(double* x, const double* h, const double* z, int64_t num, int trunc, uint64_t seed,
                     uint32_t epoch, uint64_t idx0, void* stream)
{
  if (int rc = check_args(x, z, num)) return rc;
  if (num == 0) return BL_OK;
  if (!h) { blh::set_error("h is null"); return BL_ERR_ARG; }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_rpg_gamma, dim3(blh::grid_for(num, kBlock, kMaxBlocks)), dim3(kBlock), 0, s, x, h, z, num,
                     trunc, seed, epoch, idx0);
  BL_HIP_TRY(hipGetLastError());
  return BL_OK;
}

int bl_rpg_hybrid_dev(double* x, const double* h, const double* z, int64_t num, uint64_t seed, uint32_t epoch,
                      uint64_t idx0, void* stream)
{
  if (int rc = check_args(x, z, num)) return rc;
  if (num == 0) return BL_OK;
  if (!h) { blh::set_error("h is null"); return BL_ERR_ARG; }
  hipStream_t s = (hipStream_t)stream;
  const dim3 g(blh::grid_for(num, kBlock, kMaxBlocks)), b(kBlock);
  int* st = blh::status_word(s);
  // Classified once: the first class pass (saddle point; its scan reads every shape) counts the members of every class and
  // writes the b <= 0 branch's zeros (LogitWrapper.cpp:159-161); a later pass whose class is empty returns at once.  No
  // zeroing launch over x: every element is written by exactly one pass.
  unsigned long long* cc = blh::class_counts_slot();
  BL_HIP_TRY(hipMemsetAsync(cc, 0, 8 * sizeof(unsigned long long), s));
  // The class passes write disjoint elements of x and read h and z only: the alternating-series kernel and the Devroye pass
  // go to two streams of their own beside the saddle-point kernel (all three scan every shape themselves), so that the
  // workgroups of one fill the CUs the last workgroups of another leave idle; the two classes that are usually empty wait
  // for the first pass's counts on the caller's stream.  BL_HYBRID_STREAMS=0: one stream, one pass after the other.
  static const bool fan = !(getenv("BL_HYBRID_STREAMS") && atoi(getenv("BL_HYBRID_STREAMS")) == 0);
  static std::mutex mu;               // the side streams and events are the library's: one call's record / wait pairs at a time
  std::lock_guard<std::mutex> lock(mu);
  static hipStream_t s1 = nullptr, s2 = nullptr;
  static hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
  static bool made = false, usable = false;
  if (fan && !made) {
    made = true;
    usable = hipStreamCreateWithFlags(&s1, hipStreamNonBlocking) == hipSuccess &&
             hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&e0, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&e1, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&e2, hipEventDisableTiming) == hipSuccess;
  }
  if (fan && usable) {
    BL_HIP_TRY(hipEventRecord(e0, s));                       // x, h, z are the caller's as of here
    BL_HIP_TRY(hipStreamWaitEvent(s1, e0, 0));
    BL_HIP_TRY(hipStreamWaitEvent(s2, e0, 0));
    if (int rc = blh::launch_rpg_tasks(true, x, h, z, num, nullptr, seed, epoch, idx0, blh::kHybFirst, cc, s)) return rc;     // saddle point
    if (int rc = blh::launch_rpg_tasks(false, x, h, z, num, nullptr, seed, epoch, idx0, blh::kHybPlain, nullptr, s1)) return rc;   // alternating series
    hipLaunchKernelGGL(k_rpg_hybrid_class<bl::CLS_DEVROYE>, g, b, 0, s2, x, h, z, num, seed, epoch, idx0,
                       (const unsigned long long*)nullptr, blh::status_word(s));
    BL_HIP_TRY(hipEventRecord(e1, s1));
    BL_HIP_TRY(hipEventRecord(e2, s2));
    hipLaunchKernelGGL(k_rpg_hybrid_class<bl::CLS_NORMAL>, g, b, 0, s, x, h, z, num, seed, epoch, idx0, cc, st);
    hipLaunchKernelGGL(k_rpg_hybrid_class<bl::CLS_GAMMA>, g, b, 0, s, x, h, z, num, seed, epoch, idx0, cc, st);
    BL_HIP_TRY(hipStreamWaitEvent(s, e1, 0));
    BL_HIP_TRY(hipStreamWaitEvent(s, e2, 0));
    BL_HIP_TRY(hipGetLastError());
    return BL_OK;
  }
  if (int rc = blh::launch_rpg_tasks(true, x, h, z, num, nullptr, seed, epoch, idx0, blh::kHybFirst, cc, s)) return rc;    // saddle point
  if (int rc = blh::launch_rpg_tasks(false, x, h, z, num, nullptr, seed, epoch, idx0, blh::kHybLater, cc, s)) return rc;   // alternating series
  hipLaunchKernelGGL(k_rpg_hybrid_class<bl::CLS_DEVROYE>, g, b, 0, s, x, h, z, num, seed, epoch, idx0, cc, st);
  hipLaunchKernelGGL(k_rpg_hybrid_class<bl::CLS_NORMAL>, g, b, 0, s, x, h, z, num, seed, epoch, idx0, cc, st);
  hipLaunchKernelGGL(k_rpg_hybrid_class<bl::CLS_GAMMA>, g, b, 0, s, x, h, z, num, seed, epoch, idx0, cc, st);
  BL_HIP_TRY(hipGetLastError());
  return BL_OK;
}

// diagnostic (bench.py's per-branch rates of C3): ONE class pass of rpg_hybrid alone -- cls = 4 saddle point, 3 alternating
// series, 2 Devroye, 5 normal approximation, 1 sum of gammas -- writing only that class's elements of x.
int bl_diag_rpg_hybrid_class_dev(double* x, const double* h, const double* z, int64_t num, int cls, uint64_t seed,
                                 uint32_t epoch, uint64_t idx0, void* stream)
{
  if (int rc = check_args(x, z, num)) return rc;
  if (num == 0) return BL_OK;
  if (!h) { blh::set_error("h is null"); return BL_ERR_ARG; }
  hipStream_t s = (hipStream_t)stream;
  const dim3 g(blh::grid_for(num, kBlock, kMaxBlocks)), b(kBlock);
  int* st = blh::status_word(s);
  const unsigned long long* none = nullptr;
  switch (cls) {
    case bl::CLS_SP: return blh::launch_rpg_tasks(true, x, h, z, num, nullptr, seed, epoch, idx0, blh::kHybLater + 1, nullptr, s);
    case bl::CLS_ALT: return blh::launch_rpg_tasks(false, x, h, z, num, nullptr, seed, epoch, idx0, blh::kHybLater + 1, nullptr, s);
    case bl::CLS_DEVROYE: hipLaunchKernelGGL(k_rpg_hybrid_class<bl::CLS_DEVROYE>, g, b, 0, s, x, h, z, num, seed, epoch, idx0, none, st); break;
    case bl::CLS_NORMAL: hipLaunchKernelGGL(k_rpg_hybrid_class<bl::CLS_NORMAL>, g, b, 0, s, x, h, z, num, seed, epoch, idx0, none, st); break;
    case bl::CLS_GAMMA: hipLaunchKernelGGL(k_rpg_hybrid_class<bl::CLS_GAMMA>, g, b, 0, s, x, h, z, num, seed, epoch, idx0, none, st); break;
    default: blh::set_error("bl_diag_rpg_hybrid_class_dev: cls must be 1..5"); return BL_ERR_ARG;
  }
  BL_HIP_TRY(hipGetLastError());
  return BL_OK;
}

int bl_fill_unif_dev(double* out, int64_t num, double lo, double hi, uint64_t seed, uint32_t epoch, uint64_t idx0,
                     void* stream)
{
  if (int rc = check_args(out, out, num)) return rc;
  if (num == 0) return BL_OK;
  hipLaunchKernelGGL(k_fill_unif, dim3(blh::grid_for(num, kBlock, kMaxBlocks)), dim3(kBlock), 0, (hipStream_t)stream,
                     out, num, lo, hi, seed, epoch, idx0);
  BL_HIP_TRY(hipGetLastError());
  return BL_OK;
}

int bl_fill_norm_dev(double* out, int64_t num, double mean, double sd, uint64_t seed, uint32_t epoch, uint64_t idx0,
                     void* stream)
{
  if (int rc = check_args(out, out, num)) return rc;
  if (num == 0) return BL_OK;
  hipLaunchKernelGGL(k_fill_norm, dim3(blh::grid_for(num, kBlock, kMaxBlocks)), dim3(kBlock), 0, (hipStream_t)stream,
                     out, num, mean, sd, seed, epoch, idx0);
  BL_HIP_TRY(hipGetLastError());
  return BL_OK;
}

int bl_fill_shape_dev(double* out, int64_t num, int kmax, uint64_t seed, uint32_t epoch, uint64_t idx0, void* stream)
{
  if (int rc = check_args(out, out, num)) return rc;
  if (kmax < 1) { blh::set_error("kmax < 1"); return BL_ERR_ARG; }
  if (num == 0) return BL_OK;
  hipLaunchKernelGGL(k_fill_shape, dim3(blh::grid_for(num, kBlock, kMaxBlocks)), dim3(kBlock), 0, (hipStream_t)stream,
                     out, num, kmax, seed, epoch, idx0);
  BL_HIP_TRY(hipGetLastError());
  return BL_OK;
}

int bl_diag_mfma_f64_dev(double* work, int64_t work_doubles, int waves_per_simd, int iters, double* flops,
                         void* stream)
{
  if (!blh::ensure_device()) return BL_ERR_NO_DEVICE;
  if (!work || !flops || waves_per_simd < 1 || waves_per_simd > 8 || iters < 1) {
    blh::set_error("bl_diag_mfma_f64_dev: bad argument");
    return BL_ERR_ARG;
  }
  int dev = 0, cus = 0;
  BL_HIP_TRY(hipGetDevice(&dev));
  BL_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const int grid = cus * waves_per_simd;      // 256-thread workgroups: one wave on each SIMD of a CU
  if (work_doubles < (int64_t)grid * 256) {
    blh::set_error("bl_diag_mfma_f64_dev: work buffer too small");
    return BL_ERR_ARG;
  }
  hipLaunchKernelGGL(k_diag_mfma_f64, dim3(grid), dim3(256), 0, (hipStream_t)stream, work, iters, 1.0, 1.0);
  BL_HIP_TRY(hipGetLastError());
  *flops = (double)grid * 4.0 * (double)iters * 10.0 * 2048.0;   // 16*16*4 multiply-adds per instruction
  return BL_OK;
}

int bl_diag_mfma_f64_small_dev(double* work, int64_t work_doubles, int waves_per_simd, int iters, double* flops,
                               void* stream)
{
  if (!blh::ensure_device()) return BL_ERR_NO_DEVICE;
  if (!work || !flops || waves_per_simd < 1 || waves_per_simd > 8 || iters < 1) {
    blh::set_error("bl_diag_mfma_f64_small_dev: bad argument");
    return BL_ERR_ARG;
  }
  int dev = 0, cus = 0;
  BL_HIP_TRY(hipGetDevice(&dev));
  BL_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const int grid = cus * waves_per_simd;
  if (work_doubles < (int64_t)grid * 256) {
    blh::set_error("bl_diag_mfma_f64_small_dev: work buffer too small");
    return BL_ERR_ARG;
  }
  hipLaunchKernelGGL(k_diag_mfma_f64_small, dim3(grid), dim3(256), 0, (hipStream_t)stream, work, iters, 1.0, 1.0);
  BL_HIP_TRY(hipGetLastError());
  *flops = (double)grid * 4.0 * (double)iters * 32.0 * 512.0;    // four blocks of 4*4*4 multiply-adds per instruction
  return BL_OK;
}

int bl_fill_logit_y_dev(double* y, const double* tX, const double* beta, int64_t N, int P, uint64_t seed,
                        uint32_t epoch, uint64_t idx0, void* stream)
{
  if (int rc = check_args(y, tX, N)) return rc;
  if (!beta || P < 1) { blh::set_error("beta null or P < 1"); return BL_ERR_ARG; }
  if (N == 0) return BL_OK;
  hipLaunchKernelGGL(k_fill_logit_y, dim3(blh::grid_for(N, kBlock, kMaxBlocks)), dim3(kBlock), 0, (hipStream_t)stream,
                     y, tX, beta, N, P, seed, epoch, idx0);
  BL_HIP_TRY(hipGetLastError());
  return BL_OK;
}

}  // extern "C"

// ================================================== Part 1 (.C boundary, host)
namespace {

template <class Launch>
void run_host_rpg(const char* name, double* x, const double* hd, const int* hi, const double* z, int num, int* iter,
                  Launch launch)
{
  if (!blh::ensure_device()) return;
  if (num <= 0) return;
  blh::DevBuf<double> dx, dh, dz;
  blh::DevBuf<int> dn, dit;
  hipError_t e = dx.alloc(num);
  if (e == hipSuccess) e = dz.alloc(num);
  if (e == hipSuccess && hd) e = dh.alloc(num);
  if (e == hipSuccess && hi) e = dn.alloc(num);
  if (e == hipSuccess && iter) e = dit.alloc(num);
  if (e == hipSuccess) e = dz.upload(z);
  if (e == hipSuccess && hd) e = dh.upload(hd);
  if (e == hipSuccess && hi) e = dn.upload(hi);
  if (e == hipSuccess && iter) e = dit.upload(iter);   // untouched entries keep the caller's value
  if (e != hipSuccess) {
    blh::set_error(std::string(name) + ": " + hipGetErrorString(e));
    return;
  }
  const int rc = launch(dx.p, dh.p, dn.p, dz.p, dit.p);
  if (rc == BL_OK) blh::collect_status(nullptr);   // prints a message on sampler flags, like the reference's Rprintf
  e = dx.download(x);
  if (e == hipSuccess && iter) e = dit.download(iter);
  if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
  if (e != hipSuccess) blh::set_error(std::string(name) + ": " + hipGetErrorString(e));
}

}  // namespace

extern "C" {

void rpg_devroye(double* x, int* n, double* z, int* num)
{
  const uint64_t seed = blh::global_seed();
  const uint32_t ep = blh::next_epoch();
  run_host_rpg("rpg_devroye", x, nullptr, n, z, *num, nullptr,
               [&](double* dx, double*, int* dn, double* dz, int*) {
                 return bl_rpg_devroye_dev(dx, dn, 1, dz, *num, seed, ep, 0, nullptr);
               });
}

void rpg_alt(double* x, double* h, double* z, int* num)
{
  const uint64_t seed = blh::global_seed();
  const uint32_t ep = blh::next_epoch();
  run_host_rpg("rpg_alt", x, h, nullptr, z, *num, nullptr, [&](double* dx, double* dh, int*, double* dz, int*) {
    return bl_rpg_alt_dev(dx, dh, dz, *num, seed, ep, 0, nullptr);
  });
}

void rpg_sp(double* x, double* h, double* z, int* num, int* iter)
{
  const uint64_t seed = blh::global_seed();
  const uint32_t ep = blh::next_epoch();
  run_host_rpg("rpg_sp", x, h, nullptr, z, *num, iter, [&](double* dx, double* dh, int*, double* dz, int* dit) {
    return bl_rpg_sp_dev(dx, dh, dz, *num, dit, seed, ep, 0, nullptr);
  });
}

void rpg_gamma(double* x, double* n, double* z, int* num, int* trunc)
{
  const uint64_t seed = blh::global_seed();
  const uint32_t ep = blh::next_epoch();
  run_host_rpg("rpg_gamma", x, n, nullptr, z, *num, nullptr, [&](double* dx, double* dh, int*, double* dz, int*) {
    return bl_rpg_gamma_dev(dx, dh, dz, *num, *trunc, seed, ep, 0, nullptr);
  });
}

void rpg_hybrid(double* x, double* h, double* z, int* num)
{
  const uint64_t seed = blh::global_seed();
  const uint32_t ep = blh::next_epoch();
  run_host_rpg("rpg_hybrid", x, h, nullptr, z, *num, nullptr, [&](double* dx, double* dh, int*, double* dz, int*) {
    return bl_rpg_hybrid_dev(dx, dh, dz, *num, seed, ep, 0, nullptr);
  });
}

}  // extern "C"
