// kernels_sweep1.hip -- the logistic Gibbs sweep over this rank's rows with X read ONCE (P = 64).
//
//   psi = X beta, omega_i ~ PG(n_i, psi_i), PPpart = sum_i omega_i x_i x_i'      (Logit.hpp:283-301,431)
//
// kernels_gibbs.hip does this in two streaming passes (psi/omega, then X' Omega X) because the draw's work
// queue wants hundreds of rows per wave while the rows wait on chip.  Here a wave takes 16 rows at a time --
// 8 KB, 32 registers per lane, the next tile's loads in flight in 32 more -- and removes the queue instead:
//
//   * the four lanes (k, 4g .. 4g+3) of an MFMA operand row all receive psi of row 4g+k from the 16-lane
//     butterfly anyway; lane a = 0..3 of that quad evaluates ATTEMPT a of the row's draw (Philox block a of the
//     row's stream) ahead of time: attempt 0 as a fresh proposal, attempts 1..3 as retries inside the left
//     piece -- which is what they are whenever they are reached at all: a fresh proposal is rejected by the
//     inner test of the mu > t inverse-Gaussian piece (u2 > A, PolyaGamma.cpp:89-101) in 14-26 % of the cases
//     and by the alternating series in < 0.6 %.  The first accepting attempt in block order is the draw, bit
//     for bit what the work queue of pass 1 returns (same blocks, same arithmetic: pg1_attempt_small_known);
//   * one attempt body per tile, no loop: a row that is not settled by its four attempts (0.3-2 %), or whose
//     first series test fails (8e-4), or with |psi|/2 >= 1/t (the other left-piece sampler), or with n_i != 1,
//     is DEFERRED: it enters this tile's MFMAs with weight 0 and goes to a per-wave list; every 16 deferred
//     rows are drawn by the full sampler (pg1_draw_n, out of line), their rows of X gathered again (L2 / HBM:
//     1-5 % extra traffic) and added by four more MFMA groups;
//   * the tile then goes through the fp64 matrix pipe exactly as in k_xwx_mfma (same lane <-> column
//     assignment, A = omega x_qa, B = x_qb, ten upper-triangle blocks): no LDS, no barrier, no other wave.
//
// Two waves per SIMD: one wave's attempt body (VALU) runs beside the other's MFMAs.  omega and psi are the
// values of the two-pass kernels bit for bit; PP differs from theirs in summation order only (fixed order:
// reproducible).  Slab layout and reduction are k_xwx_mfma's (k_reduce_fused).
#include "bl_gibbs_kernels.hpp"
#include "bl_pg_devroye.hpp"
#include "bl_pg1_sm.hpp"
#include <stdlib.h>

namespace {

using namespace bl;
typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int kBlock = 256;
constexpr int kDefCap = 32;      // a tile adds at most 16 entries to a list of at most 15

struct Tile {
  v2d v[4][2];     // group g (rows 4g .. 4g+3), half h: columns 32h + 2c, 32h + 2c + 1 of row 4g + k
  double nn;       // n of this lane's row (row 4 (c >> 2) + k)
};

// rows >= r1 (the end of the wave's range) read the range's last row; their weight is 0
__device__ __forceinline__ void tile_load(Tile& T, const double* __restrict__ tX, const double* __restrict__ nvec,
                                          int64_t base, int64_t r1, int k, int c)
{
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int64_t row = base + 4 * g + k;
    const double* p = tX + (size_t)(row < r1 ? row : r1 - 1) * 64;
#pragma unroll
    for (int h = 0; h < 2; ++h)
      T.v[g][h] = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(p + 32 * h + 2 * c));
  }
  const int64_t myrow = base + 4 * (c >> 2) + k;
  T.nn = __builtin_nontemporal_load(nvec + (myrow < r1 ? myrow : r1 - 1));
}

// the full sampler for a deferred row (any class, any n): the observation's stream from block 0
__device__ __attribute__((noinline)) double draw_full(int n, double psi, uint64_t seed, uint64_t idx, uint32_t epoch,
                                                      int* status)
{
  // arguments of an out-of-line function arrive in VGPRs; the key is wave-uniform and goes back to SGPRs
  const uint32_t s0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)seed);
  const uint32_t s1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(seed >> 32));
  epoch = (uint32_t)__builtin_amdgcn_readfirstlane((int)epoch);
  int st = 0;
  const double om = pg1_draw_n(n, psi, ((uint64_t)s1 << 32) | s0, idx, DOM_OMEGA, epoch, st);   // Logit.hpp:287
  if (st) atomicOr(status, st);
  return om;
}

__device__ __forceinline__ void mfma_group(v4d (&acc)[10], const double (&x)[4], double wg)
{
  double a[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) a[q] = wg * x[q];
  int blkid = 0;
#pragma unroll
  for (int qa = 0; qa < 4; ++qa)
#pragma unroll
    for (int qb = qa; qb < 4; ++qb) {
      acc[blkid] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[qa], x[qb], acc[blkid], 0, 0, 0);
      ++blkid;
    }
}

template <int VAR>
__global__ __launch_bounds__(kBlock, 2) void k_sweep_once64(const double* __restrict__ tX,
                                                            const double* __restrict__ nvec,
                                                            const double* __restrict__ beta, double* __restrict__ w,
                                                            int64_t N, uint64_t seed, uint32_t epoch, uint64_t idx0,
                                                            double* __restrict__ partial, int* __restrict__ status,
                                                            unsigned long long* __restrict__ stats)
{
  constexpr int NBLK = 10;
  __shared__ double red[2][NBLK * 4][64];
  __shared__ uint32_t sDefRow[kBlock / 64][kDefCap];
  __shared__ double sDefPsi[kBlock / 64][kDefCap];
  __shared__ double sDefN[kBlock / 64][kDefCap];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = lane >> 4, c = lane & 15;
  const int a = c & 3, gq = c >> 2;                      // attempt number, group whose row this lane draws
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  double bq[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) bq[q] = beta[(q >> 1) * 32 + 2 * c + (q & 1)];

  v4d acc[NBLK];
#pragma unroll
  for (int b = 0; b < NBLK; ++b) acc[b] = v4d{0.0, 0.0, 0.0, 0.0};

  // this wave's contiguous row range
  const int64_t nwaves = (int64_t)gridDim.x * (kBlock / 64);
  const int64_t per_wave = ((N + nwaves - 1) / nwaves + 15) / 16 * 16;
  const int64_t r0 = ((int64_t)blockIdx.x * (kBlock / 64) + wave) * per_wave;
  const int64_t r1 = (r0 + per_wave) < N ? (r0 + per_wave) : N;
  int nDef = 0;                                          // wave-uniform
  unsigned long long ndeferred = 0;

  // draw the first min(nDef, 16) rows of the list with the full sampler, add them, drop them from the list
  auto handle = [&]() __attribute__((always_inline)) {
    const int cnt = nDef < 16 ? nDef : 16;
    const int e = 4 * gq + k;
    const bool valid = e < cnt;
    const uint32_t roff = sDefRow[wave][valid ? e : 0];
    const double psi_e = sDefPsi[wave][valid ? e : 0];
    const int n_e = (int)sDefN[wave][valid ? e : 0];                  // (int) n(i), Logit.hpp:287
    double om = 0.0;
    if (valid) om = draw_full(n_e, psi_e, seed, idx0 + (uint64_t)(r0 + (int64_t)roff), epoch, status);
    if (w && valid && a == 0) w[r0 + (int64_t)roff] = om;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int eg = 4 * g + k;
      const int64_t rowg = r0 + (int64_t)sDefRow[wave][eg < cnt ? eg : 0];
      const double* p = tX + (size_t)rowg * 64;
      const v2d v0 = *reinterpret_cast<const v2d*>(p + 2 * c);
      const v2d v1 = *reinterpret_cast<const v2d*>(p + 32 + 2 * c);
      const double x[4] = {v0.x, v0.y, v1.x, v1.y};
      const double wg = __shfl(om, (lane & 48) | (4 * g));           // 0 for the entries past cnt
      mfma_group(acc, x, wg);
    }
    const int rest = nDef - cnt;
    uint32_t tr = 0;
    double tp = 0.0, tn = 0.0;
    if (lane < rest) {
      tr = sDefRow[wave][cnt + lane];
      tp = sDefPsi[wave][cnt + lane];
      tn = sDefN[wave][cnt + lane];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < rest) {
      sDefRow[wave][lane] = tr;
      sDefPsi[wave][lane] = tp;
      sDefN[wave][lane] = tn;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    nDef = rest;
  };

  unsigned long long tWait = 0, tValu = 0, tMfma = 0;
  auto step = [&](const Tile& T, int64_t base) __attribute__((always_inline)) {
    unsigned long long s0 = 0, s1 = 0, s2 = 0;
    if (VAR == 3) {
      s0 = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_waitcnt(0x0F70 | 9 | (0 << 14));   // vmcnt(9): everything but the prefetch just issued
      s1 = __builtin_amdgcn_s_memtime();
    }
    // psi of the tile's 16 rows (the arithmetic of k_psi_omega_nb: four products in column order, 16-lane butterfly)
    double psi = 0.0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const double xg[4] = {T.v[g][0].x, T.v[g][0].y, T.v[g][1].x, T.v[g][1].y};
      double part = 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) part += xg[q] * bq[q];
      part += __shfl_xor(part, 1);
      part += __shfl_xor(part, 2);
      part += __shfl_xor(part, 4);
      part += __shfl_xor(part, 8);
      psi = (gq == g) ? part : psi;
    }
    const int64_t row = base + 4 * gq + k;
    const bool inrange = row < r1;
    const double Z = fabs(psi) * 0.5;                                  // PolyaGamma.cpp:154
    const bool fast = inrange && (kSmTRecip > Z) && T.nn == 1.0;       // :87; n = 1: one PG(1, psi) draw
    // attempt a of the row, state known: a = 0 fresh, a > 0 a retry inside the left piece
    const double Zs = fast ? Z : 0.0;                                  // the others run the body on z = 0 and are ignored
    const double fz = kSmPiSq8 + 0.5 * Zs * Zs;                        // :157
    const double mass = pg1_mass_small(Zs, fz);
    const uint64_t idx = idx0 + (uint64_t)row;
    const U4 o = philox4x32_10((uint32_t)idx, ctr1_of(idx, DOM_OMEGA), epoch, (uint32_t)a, k0, k1);
    double X;
    int verdict;
    if (VAR == 1) { X = 1.0 + Zs; verdict = 1; }
    else verdict = pg1_attempt_small_known<true>(a == 0, Zs, fz, mass, u52(o.x, o.y), u52(o.z, o.w), X);
    // the row's first attempt (in block order) that does not end in a retry decides: accepted -> the draw;
    // series test open, or none of the four -> deferred
    const uint64_t bAcc = __ballot(fast && verdict == 1), bStop = __ballot(fast && verdict != 0);
    const int sh = lane & ~3;
    const uint32_t nAcc = (uint32_t)(bAcc >> sh) & 15u, nStop = (uint32_t)(bStop >> sh) & 15u;
    const uint32_t first = nStop & (0u - nStop);                       // lowest set bit
    const bool settled = (first & nAcc) != 0u;
    const int wl = (int)__builtin_ctz(first | 16u);                    // quad lane of the deciding attempt
    const double Xw = __shfl(X, sh | (wl & 3));
    const double om = settled ? 0.25 * Xw : 0.0;                       // :201
    if (w && settled && a == 0) w[row] = om;
    const bool defer = inrange && !settled && a == 0;
    const uint64_t dm = __ballot(defer);
    if (defer) {
      const int slot = nDef + __popcll(dm & lt_mask);
      sDefRow[wave][slot] = (uint32_t)(row - r0);
      sDefPsi[wave][slot] = psi;
      sDefN[wave][slot] = T.nn;
    }
    nDef += __popcll(dm);
    ndeferred += (unsigned long long)__popcll(dm);
    if (VAR == 3) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      s2 = __builtin_amdgcn_s_memtime();
    }
    // X' Omega X of the tile
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const double x[4] = {T.v[g][0].x, T.v[g][0].y, T.v[g][1].x, T.v[g][1].y};
      const double wg = __shfl(om, (lane & 48) | (4 * g));
      if (VAR == 2) { acc[g][0] += wg * x[0]; } else
      mfma_group(acc, x, wg);
    }
    if (VAR == 3) {
      const unsigned long long s3 = __builtin_amdgcn_s_memtime();
      tWait += s1 - s0;
      tValu += s2 - s1;
      tMfma += s3 - s2;
    }
    if (nDef >= 16) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      handle();
    }
  };

  const unsigned long long k0t = VAR == 3 ? __builtin_amdgcn_s_memtime() : 0, k0r = VAR == 3 ? __builtin_amdgcn_s_memrealtime() : 0;
  if (r0 < r1) {
    Tile A, B;
    tile_load(A, tX, nvec, r0, r1, k, c);
    for (int64_t base = r0; base < r1; base += 32) {
      if (base + 16 < r1) tile_load(B, tX, nvec, base + 16, r1, k, c);
      step(A, base);
      if (base + 16 >= r1) break;
      if (base + 32 < r1) tile_load(A, tX, nvec, base + 32, r1, k, c);
      step(B, base + 16);
    }
    if (nDef > 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      handle();
    }
  }
  if (stats && lane == 0 && ndeferred) atomicAdd(stats, ndeferred);
  if (VAR == 3 && stats && lane == 0) {
    atomicAdd(stats + 2, tWait);
    atomicAdd(stats + 3, tValu);
    atomicAdd(stats + 4, tMfma);
    atomicAdd(stats + 5, (unsigned long long)(__builtin_amdgcn_s_memtime() - k0t));
    atomicAdd(stats + 6, (unsigned long long)(__builtin_amdgcn_s_memrealtime() - k0r));
    atomicAdd(stats + 7, 1ull);
  }

  // fixed-order in-block reduction: (w0 + w2) + (w1 + w3)
  if (wave >= 2) {
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave - 2][b * 4 + r][lane] = acc[b][r];
  }
  __syncthreads();
  if (wave < 2) {
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[b][r] += red[wave][b * 4 + r][lane];
  }
  __syncthreads();
  if (wave == 1) {
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[0][b * 4 + r][lane] = acc[b][r];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        partial[(size_t)blockIdx.x * (NBLK * 4 * 64) + (b * 4 + r) * 64 + lane] = acc[b][r] + red[0][b * 4 + r][lane];
  }
}

}  // namespace

namespace blk {

void launch_sweep_once64(int nblocks, const double* tX, const double* n, const double* beta, double* w, int64_t N,
                         double* partial, uint64_t seed, uint32_t epoch, uint64_t idx0, int* status,
                         unsigned long long* stats, hipStream_t s)
{
  static const int var = getenv("BL_SWEEP1_VARIANT") ? atoi(getenv("BL_SWEEP1_VARIANT")) : 0;
  if (var == 1) hipLaunchKernelGGL(k_sweep_once64<1>, dim3(nblocks), dim3(kBlock), 0, s, tX, n, beta, w, N, seed, epoch, idx0, partial, status, stats);
  else if (var == 3) hipLaunchKernelGGL(k_sweep_once64<3>, dim3(nblocks), dim3(kBlock), 0, s, tX, n, beta, w, N, seed, epoch, idx0, partial, status, stats);
  else if (var == 2) hipLaunchKernelGGL(k_sweep_once64<2>, dim3(nblocks), dim3(kBlock), 0, s, tX, n, beta, w, N, seed, epoch, idx0, partial, status, stats);
  else
  hipLaunchKernelGGL(k_sweep_once64<0>, dim3(nblocks), dim3(kBlock), 0, s, tX, n, beta, w, N, seed, epoch, idx0, partial,
                     status, stats);
}

}  // namespace blk
