// kernels_sweep1.hip -- the logistic Gibbs sweep over this rank's rows with X read ONCE (P = 64).
//
//   psi = X beta, omega_i ~ PG(n_i, psi_i), PPpart = sum_i omega_i x_i x_i'      (Logit.hpp:283-301,431)
//
// kernels_gibbs.hip does this in two streaming passes (psi/omega, then X' Omega X) because the draw's work
// queue wants hundreds of rows per wave while the rows wait on chip.  Here a wave takes 16 rows at a time and
// removes the queue instead; and X' Omega X goes through the SMALL fp64 matrix instruction:
//
//   * v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 blocks, 512 flops) sustains 75 TFLOP/s on gfx950 (one per
//     16.5 cycles per SIMD), v_mfma_f64_16x16x4_f64 (2048 flops) 48 (one per ~101): scripts/gpu_mfma_rates.py.  Lane (K = lane >> 4, blk = (lane >> 2) & 3, e = lane & 3) holds A_blk[e][K],
//     B_blk[K][e] and D_blk[K][e] (scripts/experiments/mfma_f64_4x4_layout.hip).  With K = row of a 4-row group,
//     A_m = omega x (columns 16m + 4 blk + e) and B_n^r = x (columns 16n + 4 ((blk + r) & 3) + e), instruction
//     (m, n, r) adds the four 4x4 blocks PP[16m + 4blk + .][16n + 4((blk + r) & 3) + .]; 36 of them (m <= n; r = 0..3
//     for m < n; r = 0, 1, 2 for m = n) cover the upper triangle of the 16 x 16 grid of 4x4 blocks: 36 accumulator
//     doubles per lane, 144 matrix instructions per 16 rows = 2380 cycles against 4040 for the 40 big ones;
//   * a tile lives in LDS, row r at 512 r, its four 128-byte column chunks swapped pairwise in odd rows
//     (chunk n at 128 (n ^ (r & 1))): every B_n^r is then one conflict-free ds_read_b64 (the two rows a half-wave
//     reads sit in opposite halves of the 256-byte bank window), and the rotation r costs no vector instruction.
//     One 8 KB slot per wave; the next tile travels HBM -> 32 registers per lane meanwhile (two register sets taken
//     in turn, the load unconditional: a loop-carried set is copied -- and waited for -- in mid-iteration, and after
//     a conditional load the compiler can only wait for everything);
//   * the four lanes (K, blk, 0..3) all receive psi of row 4 blk + K from the 16-lane butterfly anyway; lane e
//     evaluates ATTEMPT e of that row's draw (Philox block e of the row's stream) ahead of time: attempt 0 as a
//     fresh proposal, attempts 1..3 as retries inside the left piece -- which is what they are whenever they are
//     reached at all: a fresh proposal is rejected by the inner test of the mu > t inverse-Gaussian piece
//     (u2 > A, PolyaGamma.cpp:89-101) in 14-26 % of the cases and by the alternating series in < 0.6 %.  The first
//     accepting attempt in block order is the draw -- the value the work queue of pass 1 returns (same blocks,
//     same arithmetic: pg1_attempt_small_known);
//   * one attempt body per tile, no loop, no branch: a row that is not settled by its four attempts (0.3-2 %), or
//     whose first series test fails (8e-4), or with |psi|/2 >= 1/t (the other left-piece sampler), or with n_i != 1,
//     is DEFERRED: it enters its tile's matrix instructions with weight 0 and its (row, psi) goes to the wave's list
//     in global memory (the list lives in the wave's own row range of two N-long arrays: it cannot overflow).
//     k_sweep_deferred64 then takes every wave's list 16 rows at a time: the full sampler (pg1_draw_n, out of line),
//     the rows of X gathered again by LDS-DMA (L2 / HBM: 0.3-2 % extra traffic), the same 144 matrix instructions,
//     slabs of its own.  (Inside the first kernel the out-of-line call cost every wave 50 registers: spills, and a
//     spill reload is a vector memory operation -- the wait for it is a wait for the prefetch.)
//   * fp64 matrix and fp64 vector instructions do NOT overlap on this part (their times add, whichever wave issues
//     them: mfma_f64_shapes.hip; a version that issued the attempt of tile t+1 in stages between tile t's matrix
//     instructions ran no faster), so what counts is the instruction total: 545 vector + 144 matrix instructions
//     per 16 rows against 336 + 40 big ones for the two passes.
//
// psi and omega are those of the two-pass kernels (omega to the last bits: the deferred rows' sampler is another
// instantiation of the same header); PP differs in summation order only (fixed order: reproducible).
#include "bl_dpp.hpp"
#include "bl_gibbs_kernels.hpp"
#include "bl_pg_devroye.hpp"
#include "bl_pg1_sm.hpp"

namespace {

using namespace bl;
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int kBlock = 256;

constexpr int kNAcc = 36;        // matrix instructions per 4-row group = accumulator doubles per lane

// the 36 instructions (m, n, r) of a 4-row group, ordered by r: 10, 10, 10, 6
struct Mnr { int m, n, r; };
__host__ __device__ constexpr Mnr mnr_of(int id)
{
  const int r = id < 30 ? id / 10 : 3;
  int q = id - 10 * r;
  if (r < 3) {       // m <= n in row-major order
    for (int m = 0; m < 4; ++m)
      for (int n = m; n < 4; ++n) {
        if (q == 0) return Mnr{m, n, r};
        --q;
      }
  } else {           // m < n
    for (int m = 0; m < 4; ++m)
      for (int n = m + 1; n < 4; ++n) {
        if (q == 0) return Mnr{m, n, r};
        --q;
      }
  }
  return Mnr{0, 0, 0};
}
constexpr int kSubStart[5] = {0, 10, 20, 30, 36};     // instructions of rotation r: [kSubStart[r], kSubStart[r+1])

// the full sampler for a deferred row (any class, any n): the observation's stream from block 0
__device__ __attribute__((noinline)) double draw_full(int n, double psi, uint64_t seed, uint64_t idx, uint32_t epoch,
                                                      int* status, uint32_t blk0)
{
  // arguments of an out-of-line function arrive in VGPRs; the key is wave-uniform and goes back to SGPRs
  const uint32_t s0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)seed);
  const uint32_t s1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(seed >> 32));
  epoch = (uint32_t)__builtin_amdgcn_readfirstlane((int)epoch);
  int st = 0;
  const double om = pg1_draw_n(n, psi, ((uint64_t)s1 << 32) | s0, idx, DOM_OMEGA, epoch, st, blk0);   // Logit.hpp:287
  if (st) atomicOr(status, st);
  return om;
}

// fixed-order in-block reduction (w0 + w2) + (w1 + w3) of the four waves' accumulators through `red` (2 x 36 x 64
// doubles of LDS that every wave has finished with), then the workgroup's slab
__device__ __forceinline__ void block_reduce_store(double (&acc)[kNAcc], double (*red)[kNAcc][64], double* __restrict__ partial,
                                                   int lane, int wave)
{
  __syncthreads();
  if (wave >= 2) {
#pragma unroll
    for (int b = 0; b < kNAcc; ++b) red[wave - 2][b][lane] = acc[b];
  }
  __syncthreads();
  if (wave < 2) {
#pragma unroll
    for (int b = 0; b < kNAcc; ++b) acc[b] += red[wave][b][lane];
  }
  __syncthreads();
  if (wave == 1) {
#pragma unroll
    for (int b = 0; b < kNAcc; ++b) red[0][b][lane] = acc[b];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int b = 0; b < kNAcc; ++b) partial[(size_t)blockIdx.x * (kNAcc * 64) + b * 64 + lane] = acc[b] + red[0][b][lane];
  }
}

#define BL_FENCE() __builtin_amdgcn_sched_barrier(0)
// s_waitcnt vmcnt(n) lgkmcnt(no wait) expcnt(no wait), n <= 15
#define BL_WAIT_VM(n) __builtin_amdgcn_s_waitcnt(0x0F70 | (n))

// The 144 matrix instructions of the 16-row tile in `slot` (chunk-swizzled rows, see the header) with row weights om
// (lane (k, 4g .. 4g+3) holds row 4g + k's); oB: this lane's operand offsets.
#define BL_MFMA_SUB(A, B, r)                                                                                      \
  _Pragma("unroll") for (int id = kSubStart[r]; id < kSubStart[(r) + 1]; ++id)                                    \
    acc[id] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[mnr_of(id).m], B[mnr_of(id).n], acc[id], 0, 0, 0)
__device__ __forceinline__ void mfma_tile(double (&acc)[kNAcc], const char* slot, const int (&oB)[4][2], double om, int lane)
{
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    BL_FENCE();                                            // one group's operands in registers at a time
    const double wg = __shfl(om, (lane & 48) | (4 * g));
    double A[4], B[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int n = 0; n < 4; ++n) B[n] = *reinterpret_cast<const double*>(slot + oB[r][n & 1] + 2048 * g + 128 * n);
      if (r == 0) {
#pragma unroll
        for (int m = 0; m < 4; ++m) A[m] = wg * B[m];
      }
      if (r == 0) BL_MFMA_SUB(A, B, 0);
      if (r == 1) BL_MFMA_SUB(A, B, 1);
      if (r == 2) BL_MFMA_SUB(A, B, 2);
      if (r == 3) BL_MFMA_SUB(A, B, 3);
    }
  }
}
// this lane's operand offsets: B_n^r (row 4g + k, column 16n + 4((blk + r) & 3) + e) lives at oB[r][n & 1] + 2048 g + 128 n
// (the chunk swap of odd rows folded into the base)
__device__ __forceinline__ void operand_offsets(int (&oB)[4][2], int lane)
{
  const int k = lane >> 4, blk = (lane >> 2) & 3, e = lane & 3;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int base = 512 * k + 32 * ((blk + r) & 3) + 8 * e;
    oB[r][0] = base + 128 * (k & 1);                    // even n: chunk n ^ (k & 1) = n + (k & 1)
    oB[r][1] = base - 128 * (k & 1);                    // odd n:  n - (k & 1)
  }
}


__global__ __launch_bounds__(kBlock, 2) void k_sweep_once64(const double* __restrict__ tX,
                                                            const double* __restrict__ nvec,
                                                            const double* __restrict__ beta, double* __restrict__ w,
                                                            int64_t N, uint64_t seed, uint32_t epoch, uint64_t idx0,
                                                            double* __restrict__ partial,
                                                            uint32_t* __restrict__ defRow, double* __restrict__ defPsi,
                                                            uint32_t* __restrict__ defCnt)
{
  // per wave one tile of 16 rows x 512 bytes, chunk-swizzled (see the header), and (after the loop) the 36 KB of
  // scratch of the final reduction
  __shared__ __attribute__((aligned(16))) char sTile[kBlock / 64][10240];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = lane >> 4, c = lane & 15;               // k: row of a 4-row group; c = 4 blk + e: column of a 16-column chunk
  const int a = c & 3, gq = c >> 2;                     // e = attempt number, blk = group whose row this lane draws
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  __shared__ __attribute__((aligned(16))) double sBeta[64];   // beta stays in LDS: eight registers the loop cannot spare
  if (threadIdx.x < 64) sBeta[threadIdx.x] = beta[threadIdx.x];
  __syncthreads();

  double acc[kNAcc];
#pragma unroll
  for (int b = 0; b < kNAcc; ++b) acc[b] = 0.0;

  // byte offsets into a slot.  Where this lane's 16-byte pieces (columns 32h + 2c, 32h + 2c + 1 of row 4g + k) live:
  // + 2048 g + 256 h; where its B_n^r operand (row 4g + k, column 16n + 4((blk + r) & 3) + e) lives:
  // + 2048 g + 128 n from oB[r][n & 1] (the chunk swap of odd rows folded into the base)
  const int oPiece = 512 * k + 128 * ((c >> 3) ^ (k & 1)) + 16 * (c & 7);
  int oB[4][2];
  operand_offsets(oB, lane);
  char* const myTiles = &sTile[wave][0];

  // this wave's contiguous row range
  const int64_t nwaves = (int64_t)gridDim.x * (kBlock / 64);
  const int64_t per_wave = ((N + nwaves - 1) / nwaves + 15) / 16 * 16;
  const int64_t r0 = ((int64_t)blockIdx.x * (kBlock / 64) + wave) * per_wave;
  const int64_t r1 = (r0 + per_wave) < N ? (r0 + per_wave) : N;
  int nDef = 0;                                          // wave-uniform

  // tile at `base`: HBM -> registers (rows past r1 read the range's last row; their weight is 0) ...
  auto load_tile = [&](v2d (&R)[8], double& nn, int64_t base) __attribute__((always_inline)) {
    const int64_t myrow = base + 4 * gq + k;               // n of the row this lane draws travels with the tile
    if (base + 16 <= r1) {                                 // (wave-uniform) a whole tile: one address, immediates
      const double* p = tX + (size_t)(base + k) * 64 + 2 * c;
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          R[2 * g + h] = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(p + 256 * g + 32 * h));
      nn = __builtin_nontemporal_load(nvec + myrow);
    } else {                                               // the range's last tile, and the load past it: rows clamped
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int64_t row = base + 4 * g + k;
        const double* p = tX + (size_t)(row < r1 ? row : r1 - 1) * 64 + 2 * c;
#pragma unroll
        for (int h = 0; h < 2; ++h) R[2 * g + h] = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(p + 32 * h));
      }
      nn = __builtin_nontemporal_load(nvec + (myrow < r1 ? myrow : r1 - 1));
    }
  };
  // ... -> slot s
  auto store_tile = [&](const v2d (&R)[8], int s) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int h = 0; h < 2; ++h) *reinterpret_cast<v2d*>(myTiles + oPiece + 2048 * g + 256 * h) = R[2 * g + h];
  };

  // psi of the 16 rows of the tile in slot s (the arithmetic of k_psi_omega_nb: four products in column order,
  // 16-lane butterfly); lane (k, c) keeps row 4 (c >> 2) + k
  auto psi_of = [&](int s) __attribute__((always_inline)) -> double {
    double psi = 0.0;
    const v2d b0 = *reinterpret_cast<const v2d*>(sBeta + 2 * c), b1 = *reinterpret_cast<const v2d*>(sBeta + 32 + 2 * c);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const v2d v0 = *reinterpret_cast<const v2d*>(myTiles + oPiece + 2048 * g);
      const v2d v1 = *reinterpret_cast<const v2d*>(myTiles + oPiece + 2048 * g + 256);
      const double xg[4] = {v0.x, v0.y, v1.x, v1.y};
      const double bq[4] = {b0.x, b0.y, b1.x, b1.y};
      double part = 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) part += xg[q] * bq[q];
      part = row16_allsum(part);          // (bl_dpp.hpp: the xor butterfly's bits, without LDS)
      psi = (gq == g) ? part : psi;
    }
    return psi;
  };

  // the four attempts of every row of a tile: verdict 0 / 1 / 2 and the proposal X
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
      const int64_t slot = r0 + nDef + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(dm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dm, 0u));   // the wave's list lives in its own row range
      // bit 31: a fast row none of whose four attempts stopped -- four retries inside the left piece: the deferred kernel goes
      // on at block 4 instead of taking the four again
      defRow[slot] = (uint32_t)(row - r0) | ((fast && nStop == 0u) ? 0x80000000u : 0u);
      defPsi[slot] = psi;
    }
    nDef += __popcll(dm);
    return om;
  };

  if (r0 < r1) {
    const int64_t ntiles = (r1 - r0 + 15) / 16;
    const int64_t myrow0 = r0 + 4 * gq + k;               // this lane's row of tile 0
    // one tile: `cur` (arrived, or arriving) into the slot, tile t+1 sets off into `nxt`, then psi, the attempts, the
    // matrix instructions.  Two register sets taken in turn: a loop-carried set would be copied, and waited for, early.
    auto step = [&](const v2d (&cur)[8], double nn, v2d (&nxt)[8], double& nn_nxt, int64_t t) __attribute__((always_inline)) {
      const int64_t row = myrow0 + 16 * t;
      store_tile(cur, 0);
      load_tile(nxt, nn_nxt, r0 + 16 * (t + 1));           // unconditional (past the range: its last row, unused): a
                                                           // conditional load would make every later wait a full one
      const double psi = psi_of(0);
      double X;
      const int verdict = attempts(psi, row, X);
      const double om = settle(psi, nn, row, verdict, X);
      mfma_tile(acc, myTiles, oB, om, lane);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the slot is read out before the next tile overwrites it
    };
    v2d Ra[8], Rb[8];
    double na = 1.0, nb = 1.0;
    load_tile(Ra, na, r0);
    for (int64_t t = 0; t < ntiles; t += 2) {
      step(Ra, na, Rb, nb, t);
      if (t + 1 < ntiles) step(Rb, nb, Ra, na, t + 1);
    }
  }
  if (lane == 0) defCnt[blockIdx.x * (kBlock / 64) + wave] = (uint32_t)nDef;

  block_reduce_store(acc, reinterpret_cast<double (*)[kNAcc][64]>(&sTile[0][0]), partial, lane, wave);
}

// The rows k_sweep_once64 left: wave w of this grid (the same grid) takes the list wave w of that kernel wrote into
// its own row range, 64 rows at a time: one lane per row runs the full sampler (the kernel's time is the longest wave's
// chain of draws, so they are taken 64 abreast), the weights go through LDS, and the rows of X are gathered 16 at a time
// into one of the wave's two slots by LDS-DMA (piece p = slot bytes [1024 p, 1024 p + 1024) = rows 2p, 2p + 1; lane i brings
// the 16 bytes stored at row 2p + (i >> 5), chunk (i >> 3) & 3, unit i & 7, i.e. columns of chunk ((i >> 3) & 3) ^ (row & 1))
// while the previous 16 go through the same 144 matrix instructions.  Its slabs follow the first kernel's.
__global__ __launch_bounds__(kBlock, 2) void k_sweep_deferred64(const double* __restrict__ tX,
                                                                const double* __restrict__ nvec, double* __restrict__ w,
                                                                int64_t N, uint64_t seed, uint32_t epoch, uint64_t idx0,
                                                                double* __restrict__ partial, int* __restrict__ status,
                                                                const uint32_t* __restrict__ defRow,
                                                                const double* __restrict__ defPsi,
                                                                const uint32_t* __restrict__ defCnt,
                                                                unsigned long long* __restrict__ stats,
                                                                unsigned long long* __restrict__ hstats)
{
  __shared__ __attribute__((aligned(16))) char sTile[kBlock / 64][2][8192];   // (the first 36 KB: scratch of the reduction)
  __shared__ double sOm[kBlock / 64][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = lane >> 4, gq = (lane >> 2) & 3;
  double acc[kNAcc];
#pragma unroll
  for (int b = 0; b < kNAcc; ++b) acc[b] = 0.0;
  int oB[4][2];
  operand_offsets(oB, lane);
  const int64_t nwaves = (int64_t)gridDim.x * (kBlock / 64);
  const int64_t per_wave = ((N + nwaves - 1) / nwaves + 15) / 16 * 16;
  const int64_t r0 = ((int64_t)blockIdx.x * (kBlock / 64) + wave) * per_wave;
  const int cnt = r0 < N ? (int)defCnt[blockIdx.x * (kBlock / 64) + wave] : 0;
  // the 16 rows of the list from position b on -> slot s
  auto gather = [&](int b, int s) __attribute__((always_inline)) {
    const int c16 = (cnt - b) < 16 ? (cnt - b) : 16;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int row = 2 * p + (lane >> 5);
      const int64_t grow = r0 + (int64_t)(defRow[r0 + b + (row < c16 ? row : 0)] & 0x7FFFFFFFu);
      const double* src = tX + (size_t)grow * 64 + 16 * (((lane >> 3) & 3) ^ (row & 1)) + 2 * (lane & 7);
      __builtin_amdgcn_global_load_lds(src, &sTile[wave][s][1024 * p], 16, 0, 0);
    }
  };
  for (int b0 = 0; b0 < cnt; b0 += 64) {
    const int c64 = (cnt - b0) < 64 ? (cnt - b0) : 64;
    gather(b0, 0);                                          // on its way while the draws run
    double om = 0.0;
    if (lane < c64) {
      const uint32_t dr = defRow[r0 + b0 + lane];
      const int64_t grow = r0 + (int64_t)(dr & 0x7FFFFFFFu);
      om = draw_full((int)nvec[grow] /* (int) n(i), Logit.hpp:287 */, defPsi[r0 + b0 + lane], seed, idx0 + (uint64_t)grow, epoch, status,
                     (dr >> 31) ? 4u : 0u);
      if (w) w[grow] = om;
    }
    sOm[wave][lane] = om;                                   // 0 past the end of the list
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int sub = 0; 16 * sub < c64; ++sub) {
      const int s = sub & 1;
      if (16 * (sub + 1) < c64) {
        gather(b0 + 16 * (sub + 1), s ^ 1);
        BL_WAIT_VM(8);                                      // all but the eight pieces just issued
      } else {
        BL_WAIT_VM(0);
      }
      asm volatile("" ::: "memory");
      mfma_tile(acc, &sTile[wave][s][0], oB, sOm[wave][16 * sub + 4 * gq + k], lane);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the slot is read out before it is overwritten
    }
  }
  // the deferred rows counted once per launch (k_reduce_q4 adds up defCnt): an atomic per wavefront here was 4096 atomics on
  // two words, served one after the other by their L2 channel -- 30 of this kernel's 38 us at N = 1.25e6
  (void)stats;
  (void)hstats;
  __syncthreads();
  block_reduce_store(acc, reinterpret_cast<double (*)[kNAcc][64]>(&sTile[0][0][0]), partial, lane, wave);
}

// PP = sum over workgroups (fixed order) of the slabs [36][64]; instruction id = (m, n, r), lane (i, blk, j) holds
// PP[16m + 4blk + i][16n + 4((blk + r) & 3) + j]: the i <= j half of the diagonal blocks and every block of an
// unordered pair once, mirrored: PP is exactly symmetric.
__global__ __launch_bounds__(1024) void k_reduce_q4(const double* __restrict__ partial, int nparts,
                                                    double* __restrict__ PP, const uint32_t* __restrict__ defCnt, int ncnt,
                                                    int64_t N, unsigned long long* __restrict__ stats,
                                                    unsigned long long* __restrict__ hstats)
{
  if (blockIdx.x == kNAcc) {
    // one more workgroup: the launch's deferred rows (k_sweep_once64's per-wave counts; a wave whose rows start past N has
    // none: its word is stale)
    __shared__ unsigned long long tot;
    if (threadIdx.x == 0) tot = 0ull;
    __syncthreads();
    const int64_t per_wave = ((N + ncnt - 1) / ncnt + 15) / 16 * 16;
    unsigned long long c = 0;
    for (int i = (int)threadIdx.x; i < ncnt; i += 1024)
      if ((int64_t)i * per_wave < N) c += defCnt[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&tot, c);
    __syncthreads();
    if (threadIdx.x == 0 && tot) {
      if (stats) atomicAdd(stats, tot);
      atomicAdd(hstats, tot);        // the handle's own count (fall-back policy)
    }
    return;
  }
  constexpr int E = kNAcc * 64;
  __shared__ double sm[16][64];
  const int e = blockIdx.x * 64 + (threadIdx.x & 63);
  const int s = threadIdx.x >> 6;            // 16 waves: wave s sums slabs s, s+16, ... (fixed order)
  const double sum = e < E ? blk::slab_sum16(partial, E, e, s, nparts) : 0.0;
  sm[s][threadIdx.x & 63] = sum;
  __syncthreads();
  if (s == 0 && e < E) {
    const int l = threadIdx.x & 63;
    double tot = sm[0][l];
#pragma unroll
    for (int q = 1; q < 16; ++q) tot += sm[q][l];
    const Mnr t = mnr_of(e >> 6);
    const int i = l >> 4, blk = (l >> 2) & 3, j = l & 3;
    const int qa = blk, qb = (blk + t.r) & 3;               // 4-column blocks inside the 16-column chunks m, n
    const int A = 16 * t.m + 4 * qa + i, B = 16 * t.n + 4 * qb + j;
    bool take = true;
    if (t.m == t.n) {
      if (t.r == 0) take = i <= j;
      if (t.r == 2) take = blk < 2;                          // blocks 2, 3 repeat blocks 0, 1 transposed
    }
    if (take) {
      PP[A + (size_t)B * 64] = tot;
      PP[B + (size_t)A * 64] = tot;
    }
  }
}

}  // namespace

namespace blk {

// workspace of the single-pass sweep: 2 x nblocks slabs of 36 x 64 doubles, the deferred rows' psi (N doubles) and row
// offsets (N uint32), one count per wave
size_t sweep_once64_ws_doubles(int nblocks, int64_t N)
{
  const size_t n = (size_t)(N > 0 ? N : 0);
  return (size_t)nblocks * 2 * kNAcc * 64 + n + (n + 1) / 2 + (size_t)nblocks * 2 + 8;
}

unsigned long long* sweep_once64_deferred_counter(double* ws, int nblocks, int64_t N)
{
  const size_t n = (size_t)(N > 0 ? N : 0);
  // behind the slabs, psi, row offsets (n uint32 = (n + 1) / 2 doubles) and counts (4 nblocks uint32 = 2 nblocks doubles)
  return reinterpret_cast<unsigned long long*>(ws + (size_t)nblocks * 2 * kNAcc * 64 + n + (n + 1) / 2 + (size_t)nblocks * 2);
}

void launch_sweep_once64(int nblocks, const double* tX, const double* n, const double* beta, double* w, int64_t N,
                         double* ws, double* PP, uint64_t seed, uint32_t epoch, uint64_t idx0, int* status,
                         unsigned long long* stats, hipStream_t s)
{
  const size_t nn = (size_t)(N > 0 ? N : 0);
  double* slabs = ws;
  double* defPsi = slabs + (size_t)nblocks * 2 * kNAcc * 64;
  uint32_t* defRow = reinterpret_cast<uint32_t*>(defPsi + nn);
  uint32_t* defCnt = defRow + 2 * ((nn + 1) / 2);
  unsigned long long* hstats = sweep_once64_deferred_counter(ws, nblocks, N);
  hipLaunchKernelGGL(k_sweep_once64, dim3(nblocks), dim3(kBlock), 0, s, tX, n, beta, w, N, seed, epoch, idx0, slabs, defRow,
                     defPsi, defCnt);
  hipLaunchKernelGGL(k_sweep_deferred64, dim3(nblocks), dim3(kBlock), 0, s, tX, n, w, N, seed, epoch, idx0,
                     slabs + (size_t)nblocks * kNAcc * 64, status, defRow, defPsi, defCnt, stats, hstats);
  hipLaunchKernelGGL(k_reduce_q4, dim3(kNAcc + 1), dim3(1024), 0, s, slabs, 2 * nblocks, PP, (const uint32_t*)defCnt,
                     nblocks * (kBlock / 64), N, stats, hstats);
}

}  // namespace blk
