// kernels_sweep1.hip -- the logistic Gibbs sweep over this rank's rows with X read ONCE (P = 64).
//
//   psi = X beta, omega_i ~ PG(n_i, psi_i), PPpart = sum_i omega_i x_i x_i'      (Logit.hpp:283-301,431)
//
// kernels_gibbs.hip does this in two streaming passes (psi/omega, then X' Omega X) because the draw's work
// queue wants hundreds of rows per wave while the rows wait on chip.  Here a wave takes 16 rows at a time and
// removes the queue instead:
//
//   * tiles travel HBM -> LDS by LDS-DMA (global_load_lds_dwordx4: no registers hold data in flight), already in
//     MFMA operand order (piece (g, h) = lane (k, c)'s 16 bytes of row 4g+k, columns 32h+2c, 32h+2c+1), two
//     8 KB slots per wave: while tile t is in registers for its MFMAs and tile t+1 is in its slot, tile t+2 is
//     on its way into the slot tile t has just left;
//   * the four lanes (k, 4g .. 4g+3) of an MFMA operand row all receive psi of row 4g+k from the 16-lane
//     butterfly anyway; lane a = 0..3 of that quad evaluates ATTEMPT a of the row's draw (Philox block a of the
//     row's stream) ahead of time: attempt 0 as a fresh proposal, attempts 1..3 as retries inside the left
//     piece -- which is what they are whenever they are reached at all: a fresh proposal is rejected by the
//     inner test of the mu > t inverse-Gaussian piece (u2 > A, PolyaGamma.cpp:89-101) in 14-26 % of the cases
//     and by the alternating series in < 0.6 %.  The first accepting attempt in block order is the draw -- the
//     value the work queue of pass 1 returns (same blocks, same arithmetic: pg1_attempt_small_known);
//   * one attempt body per tile, no loop, no branch: a row that is not settled by its four attempts (0.3-2 %),
//     or whose first series test fails (8e-4), or with |psi|/2 >= 1/t (the other left-piece sampler), or with
//     n_i != 1, is DEFERRED: it enters its tile's MFMAs with weight 0 and goes to a per-wave list; every 16
//     deferred rows are drawn by the full sampler (pg1_draw_n, out of line), their rows of X gathered again
//     (L2 / HBM: 1-5 % extra traffic) and added by four more MFMA groups;
//   * SOFTWARE PIPELINE: the attempt body of tile t+1 (about 450 vector instructions) and the 40
//     v_mfma_f64_16x16x4_f64 of tile t (same lane <-> column assignment as k_xwx_mfma, A = omega x_qa, B = x_qb,
//     ten upper-triangle blocks) are one basic block, interleaved by the scheduler (sched_group_barrier: one
//     matrix instruction, then eleven vector instructions), so that every wave keeps the matrix pipe and the
//     vector ALU busy at the same time whatever its SIMD partner is doing.  (Un-pipelined -- body, then MFMAs --
//     the two waves of a SIMD fall into step, both in their matrix phase or both in their vector phase:
//     1.53 ms per C4 sweep against 1.87 for the two passes; stamps in DESIGN.md.)
//
// Two waves per SIMD.  psi and omega are those of the two-pass kernels (omega to the last bits: the full
// sampler is another instantiation of the same header); PP differs in summation order only (fixed order:
// reproducible).  Slab layout and reduction are k_xwx_mfma's (k_reduce_fused).
#include "bl_gibbs_kernels.hpp"
#include "bl_pg_devroye.hpp"
#include "bl_pg1_sm.hpp"

namespace {

using namespace bl;
typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int kBlock = 256;
constexpr int kDefCap = 32;      // a tile adds at most 16 entries to a list of at most 15

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

// s_waitcnt vmcnt(n) lgkmcnt(no wait) expcnt(no wait), n <= 15
#define BL_WAIT_VM(n) __builtin_amdgcn_s_waitcnt(0x0F70 | (n))

// MFMAs K0 .. K1-1 of a tile's 40 (number 10 g + b: group g, upper-triangle block b = (qa, qb)), fenced on both sides
__device__ constexpr int kQa[10] = {0, 0, 0, 0, 1, 1, 1, 2, 2, 3}, kQb[10] = {0, 1, 2, 3, 1, 2, 3, 2, 3, 3};
#define BL_MFMAS(K0, K1)                                                                                         \
  do {                                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    _Pragma("unroll") for (int kk = (K0); kk < (K1); ++kk)                                                       \
      acc[kk % 10] = __builtin_amdgcn_mfma_f64_16x16x4f64(am[kk / 10][kQa[kk % 10]], xm[kk / 10][kQb[kk % 10]],   \
                                                           acc[kk % 10], 0, 0, 0);                                \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
  } while (0)

__global__ __launch_bounds__(kBlock, 2) void k_sweep_once64(const double* __restrict__ tX,
                                                            const double* __restrict__ nvec,
                                                            const double* __restrict__ beta, double* __restrict__ w,
                                                            int64_t N, uint64_t seed, uint32_t epoch, uint64_t idx0,
                                                            double* __restrict__ partial, int* __restrict__ status,
                                                            unsigned long long* __restrict__ stats)
{
  constexpr int NBLK = 10;
  // per wave two tiles of 16 rows in MFMA operand order: [slot][piece 2g+h][lane] x 16 bytes (64 KB per workgroup;
  // the first 40 KB are the scratch of the final reduction)
  __shared__ __attribute__((aligned(16))) v2d sTile[kBlock / 64][2][8][64];
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

  // tile at `base` -> slot s: eight LDS-DMA pieces (rows past r1 read the range's last row; their weight is 0)
  auto dma_tile = [&](int64_t base, int s) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int64_t row = base + 4 * g + k;
      const double* p = tX + (size_t)(row < r1 ? row : r1 - 1) * 64 + 2 * c;
#pragma unroll
      for (int h = 0; h < 2; ++h) __builtin_amdgcn_global_load_lds(p + 32 * h, &sTile[wave][s][2 * g + h][0], 16, 0, 0);
    }
  };

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

  // psi of the 16 rows of the tile in slot s (the arithmetic of k_psi_omega_nb: four products in column order,
  // 16-lane butterfly); lane (k, c) keeps row 4 (c >> 2) + k
  auto psi_of = [&](int s) __attribute__((always_inline)) -> double {
    double psi = 0.0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const v2d v0 = sTile[wave][s][2 * g][lane], v1 = sTile[wave][s][2 * g + 1][lane];
      const double xg[4] = {v0.x, v0.y, v1.x, v1.y};
      double part = 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) part += xg[q] * bq[q];
      part += __shfl_xor(part, 1);
      part += __shfl_xor(part, 2);
      part += __shfl_xor(part, 4);
      part += __shfl_xor(part, 8);
      psi = (gq == g) ? part : psi;
    }
    return psi;
  };

  // the four attempts of every row of a tile (straight-line): verdict 0 / 1 / 2 and the proposal X
  auto attempts = [&](double psi, int64_t row, double& X) __attribute__((always_inline)) -> int {
    const double Z = fabs(psi) * 0.5;                                  // PolyaGamma.cpp:154
    const double fz = kSmPiSq8 + 0.5 * Z * Z;                          // :157
    const double mass = pg1_mass_small(Z, fz);                         // rows outside the class are masked by the caller
    const uint64_t idx = idx0 + (uint64_t)row;
    const U4 o = philox4x32_10((uint32_t)idx, ctr1_of(idx, DOM_OMEGA), epoch, (uint32_t)a, k0, k1);
    return pg1_attempt_small_known(a == 0, Z, fz, mass, u52(o.x, o.y), u52(o.z, o.w), X);
  };

  // the row's first attempt (in block order) that does not end in a retry decides: accepted -> the draw;
  // series test open, or none of the four, or not a fast row -> deferred.  Returns omega (0 if deferred).
  auto settle = [&](double psi, double nn, int64_t row, int verdict, double X) __attribute__((always_inline)) -> double {
    const bool inrange = row < r1;
    const bool fast = inrange && (kSmTRecip > fabs(psi) * 0.5) && nn == 1.0;   // :87; n = 1: one PG(1, psi) draw
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
      sDefN[wave][slot] = nn;
    }
    nDef += __popcll(dm);
    ndeferred += (unsigned long long)__popcll(dm);
    return om;
  };

  if (r0 < r1) {
    const int64_t ntiles = (r1 - r0 + 15) / 16;
    const int64_t myrow0 = r0 + 4 * gq + k;               // this lane's row of tile 0
    // prologue: tiles 0 and 1 on their way, the draw of tile 0
    dma_tile(r0, 0);
    if (ntiles > 1) dma_tile(r0 + 16, 1);
    double nn = nvec[myrow0 < r1 ? myrow0 : r1 - 1];
    if (ntiles > 1) BL_WAIT_VM(9); else BL_WAIT_VM(1);
    asm volatile("" ::: "memory");
    double om;
    {
      const double psi = psi_of(0);
      double X;
      const int verdict = attempts(psi, myrow0, X);
      om = settle(psi, nn, myrow0, verdict, X);
    }
    for (int64_t t = 0; t + 1 < ntiles; ++t) {
      const int s = (int)(t & 1);
      // tile t: slot -> registers (its DMA was waited for before its psi)
      v2d xt[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) xt[i] = sTile[wave][s][i][lane];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slot is free once the reads have returned
      const int64_t row1 = myrow0 + 16 * (t + 1);
      const double nn1 = nvec[row1 < r1 ? row1 : r1 - 1];
      if (t + 2 < ntiles) {
        dma_tile(r0 + 16 * (t + 2), s);
        BL_WAIT_VM(9);                                     // all but nn1 and the eight pieces just issued: tile t+1 has landed
      } else {
        BL_WAIT_VM(1);
      }
      asm volatile("" ::: "memory");
      const double psi1 = psi_of(s ^ 1);
      // One basic block: the attempt of tile t+1 in stages, three or four of tile t's 40 MFMAs after each stage
      // (the fences keep the scheduler from regrouping them).
      double am[4][4], xm[4][4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const double wg = __shfl(om, (lane & 48) | (4 * g));
        xm[g][0] = xt[2 * g].x, xm[g][1] = xt[2 * g].y, xm[g][2] = xt[2 * g + 1].x, xm[g][3] = xt[2 * g + 1].y;
#pragma unroll
        for (int q = 0; q < 4; ++q) am[g][q] = wg * xm[g][q];
      }
      BL_MFMAS(0, 3);
      const double Z = fabs(psi1) * 0.5;                               // PolyaGamma.cpp:154
      const double fz = kSmPiSq8 + 0.5 * Z * Z;                        // :157
      const uint64_t idx = idx0 + (uint64_t)row1;
      PhiloxState ph{U4{(uint32_t)idx, ctr1_of(idx, DOM_OMEGA), epoch, (uint32_t)a}, k0, k1};
      philox_rounds<2>(ph);
      BL_MFMAS(3, 6);
      philox_rounds<3>(ph);
      BL_MFMAS(6, 9);
      philox_rounds<3>(ph);
      BL_MFMAS(9, 12);
      philox_rounds<2>(ph);
      const double u1 = u52(ph.c.x, ph.c.y), u2 = u52(ph.c.z, ph.c.w);
      BL_MFMAS(12, 15);
      const double mass = pg1_mass_small(Z, fz);            // rows outside the class are masked in settle()
      BL_MFMAS(15, 19);
      Pg1Staged st;
      pg1_stage_w(st, a == 0, mass, u1);
      BL_MFMAS(19, 22);
      pg1_stage_log(st);
      BL_MFMAS(22, 26);
      pg1_stage_x(st, Z, fz);
      BL_MFMAS(26, 29);
      pg1_stage_A(st);
      BL_MFMAS(29, 33);
      pg1_stage_r3(st);
      BL_MFMAS(33, 37);
      const int verdict1 = pg1_stage_verdict(st, u2);
      const double X1 = st.X;
      BL_MFMAS(37, 40);
      om = settle(psi1, nn1, row1, verdict1, X1);
      if (nDef >= 16) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        handle();
      }
    }
    {   // the last tile's MFMAs
      const int s = (int)((ntiles - 1) & 1);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const v2d v0 = sTile[wave][s][2 * g][lane], v1 = sTile[wave][s][2 * g + 1][lane];
        const double x[4] = {v0.x, v0.y, v1.x, v1.y};
        const double wg = __shfl(om, (lane & 48) | (4 * g));
        mfma_group(acc, x, wg);
      }
    }
    if (nDef > 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      handle();
    }
  }
  if (stats && lane == 0 && ndeferred) atomicAdd(stats, ndeferred);

  // fixed-order in-block reduction: (w0 + w2) + (w1 + w3)
  __syncthreads();                                         // every wave is done with its slots
  double (*red)[NBLK * 4][64] = reinterpret_cast<double (*)[NBLK * 4][64]>(&sTile[0][0][0][0]);
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
  hipLaunchKernelGGL(k_sweep_once64, dim3(nblocks), dim3(kBlock), 0, s, tX, n, beta, w, N, seed, epoch, idx0, partial,
                     status, stats);
}

}  // namespace blk
