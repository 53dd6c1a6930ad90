// kernels_xwx4.hip -- X' Omega X for 64 < P <= 256 (Logit.hpp:294-301) on the SMALL fp64 matrix instruction.
//
// The rank-N update is compute-bound here (N P^2 flops against 8 N P bytes), and on gfx950 v_mfma_f64_4x4x4_4b_f64
// (four independent 4x4x4 blocks, 512 flops, one per 16.5 cycles per SIMD: 75 TFLOP/s) outruns
// v_mfma_f64_16x16x4_f64 (2048 flops per ~101 cycles: 48 TFLOP/s) by half again (scripts/gpu_mfma_rates.py).  Operand layout (scripts/experiments/mfma_f64_4x4_layout.hip): lane
// (K = lane >> 4, blk = (lane >> 2) & 3, e = lane & 3) holds A_blk[e][K], B_blk[K][e], D_blk[K][e].  With K = row of a
// 4-row group, A_m = omega x (columns 16m + 4blk + e) and B_n^r = x (columns 16n + 4((blk + r) & 3) + e), instruction
// (m, n, r) adds the four 4x4 blocks PP[16m + 4blk + .][16n + 4((blk + r) & 3) + .]: r = 0..3 for a pair of 16-column
// chunks m < n, r = 0, 1, 2 for m = n.
//
//   * a workgroup of NC waves (NC = 8 or 16 chunks of 16 columns; four waves per SIMD at NC = 16: the small instruction has
//     little shadow to hide a wave's LDS waits and scalar branches in, so it wants many waves) shares 32-row (NC = 8: 64-row) tiles of X
//     staged in LDS by LDS-DMA, double buffered; the row stride is 128 NC + 128 bytes: every B_n^r is one conflict-free
//     ds_read_b64 (the two rows a half-wave reads sit in opposite halves of the 256-byte bank window) and the rotation r
//     costs no vector instruction;
//   * chunk-rows m1 = p and m2 = NC-1-p of the upper triangle (NC + 1 cells together) belong to the pair of waves p and
//     p + NC/2.  The off-diagonal cells are walked by chunk n = NC-1, NC-2, ...: position j serves row m1 and, while j < p,
//     row m2 as well from ONE read of B_n^r (the small instruction needs an operand double per lane per 512 flops); wave h
//     of the pair takes the positions j = h, h + 2, ... and one of the two diagonal cells: NC/2 accumulator slots of four
//     doubles (row m2's cells in the slots row m1 does not reach) + three.  A chain of nested scalar ifs (the wave
//     index is read into a scalar register: taken from threadIdx it is a vector value, the ifs become exec-masked and
//     every accumulator is kept twice) with static immediates, each cell's operands read one cell ahead.  The group loop
//     is not unrolled: the lane's ten base addresses advance by four rows instead;
//   * slabs [wave][2 NC + 3][64 lanes] per workgroup, summed in fixed order by k_reduce_q4_big: PP is reproducible and
//     exactly symmetric.
#include "bl_gibbs_kernels.hpp"
#include <type_traits>

namespace {

typedef double v2d __attribute__((ext_vector_type(2)));

#define BL_MF(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0)

template <int NC>
__global__ __launch_bounds__(NC * 64) void k_xwx_q4_big(const double* __restrict__ tX, const double* __restrict__ w,
                                                      int64_t N, double* __restrict__ partial)
{
  constexpr int NW = NC;                      // waves: four per SIMD at NC = 16
  constexpr int NLOC = NC / 2;                // off-diagonal accumulator slots of a wave
  constexpr int P = 16 * NC;                  // columns
  constexpr int ROWB = 128 * NC + 128;        // bytes per row of the tile (padded: consecutive rows half a bank window apart)
  constexpr int RT = NC == 16 ? 32 : 64;      // rows per tile (139 KB of LDS for the two buffers)
  constexpr int TILEB = RT * ROWB;            // bytes per tile
  constexpr int NACC = 4 * NLOC + 3;
  constexpr int HPR = P / 128;                // 1 KB pieces per row
  constexpr int PIECES = RT * HPR;            // per tile
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* const tiles = lds;                                        // [2][16][ROWB]
  double* const wt = reinterpret_cast<double*>(lds + 2 * TILEB);  // [2][RT]
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);          // a scalar: the branches on it below are scalar branches
  const int k = lane >> 4, blk = (lane >> 2) & 3, e = lane & 3;
  const int64_t ntiles = (N + RT - 1) / RT;

  // the wave's cells: pair p of chunk-rows (m1, m2), positions j = h, h + 2, ...
  const int p = wave & (NLOC - 1), h = wave / NLOC;
  const int m1 = p, m2 = NC - 1 - p, split = NC - 1 - p;
  double accD[3] = {0.0, 0.0, 0.0};
  double acc[NLOC][4];
#pragma unroll
  for (int s = 0; s < NLOC; ++s)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[s][r] = 0.0;

  // this lane's byte offsets into the tiles, for the 4-row group being worked on (they advance with it): B_n^r of
  // position j = 2 u + h (chunk n = NC-1-j) is read at b[r] + 128 (NC-2-2u), the immediate static (h is in the base)
  int b[4], d1[3], d2[3];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int lb = k * ROWB + 32 * ((blk + r) & 3) + 8 * e;   // (dynamic LDS starts at address 0: no static LDS here)
    b[r] = lb - 128 * h + 128;                       // never negative; the immediates carry the - 128
    if (r < 3) {
      d1[r] = lb + 128 * m1;
      d2[r] = lb + 128 * m2;
    }
  }
  int wrow = 8 * k;                                  // byte offset of this lane's row weight in wt[]
  auto advance = [&](int bytes, int wbytes) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      b[r] += bytes;
      if (r < 3) {
        d1[r] += bytes;
        d2[r] += bytes;
      }
    }
    wrow += wbytes;
#pragma unroll
    for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(b[r]));      // kept as registers, not recomputed per use
#pragma unroll
    for (int r = 0; r < 3; ++r) asm volatile("" : "+v"(d1[r]), "+v"(d2[r]));
  };

  // tile tl -> buffer buf by LDS-DMA: a piece (one wave instruction) is 1 KB = half a row; wave w brings pieces
  // w, w + NW, ...: lane i the 16 bytes at column 2 i of that half.  Rows past N read row N-1; their weight is 0.
  auto fetch = [&](int64_t tl, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < PIECES / NW; ++j) {
      const int piece = wave + j * NW;                             // row = piece / HPR, part = piece % HPR
      const int r = piece / HPR, part = piece % HPR;
      int64_t row = tl * RT + r;
      row = row < N ? row : N - 1;
      __builtin_amdgcn_global_load_lds(tX + (size_t)row * P + 128 * part + 2 * lane,
                                       tiles + buf * TILEB + r * ROWB + 1024 * part, 16, 0, 0);
    }
    if (t < RT) {
      const int64_t row = tl * RT + t;
      wt[buf * RT + t] = row < N ? w[row] : 0.0;
    }
  };

  // one 4-row group (the base offsets point at it)
#define BL_RD(base, imm) (*reinterpret_cast<const double*>(tiles + (base) + (imm)))
// The four operands of a cell by inline assembly: compiler-visible loads are sunk into the if of the cell that uses
// them, i.e. issued right before their use; these stay where they are written, one cell ahead.  The wait is explicit
// (lgkmcnt(4): all but the four reads just issued), tied to the registers it releases.
#define BL_LDS4(q, base, imm)                                                                                    \
  asm volatile("ds_read_b64 %0, %4 offset:%8\n\tds_read_b64 %1, %5 offset:%8\n\tds_read_b64 %2, %6 offset:%8\n\t"    \
               "ds_read_b64 %3, %7 offset:%8"                                                                    \
               : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3])                                              \
               : "v"(base[0]), "v"(base[1]), "v"(base[2]), "v"(base[3]), "n"(imm))
#define BL_WAIT4(q, n) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]))
#define BL_MF4(A, q, a)                                                                                          \
  a[0] = BL_MF(A, q[0], a[0]), a[1] = BL_MF(A, q[1], a[1]), a[2] = BL_MF(A, q[2], a[2]), a[3] = BL_MF(A, q[3], a[3])
// The wave's u-th position j = 2u + h is chunk n = NC-1-j: row m1's cell (m1, n) while j < split (slot u), and also row
// m2's cell (m2, n) while j < p (slot NLOC-1-u, which row m1 does not reach in this wave): one read of B_n^r serves both.
#define BL_CELL(u)                                                                                               \
  if (NLOC > (u) && 2 * (u) + h < split) {                                                                       \
    BL_LDS4(q[((u) + 1) & 1], b, 128 * (NC - 4 - 2 * (u) >= 0 ? NC - 4 - 2 * (u) : 0));                          \
    BL_WAIT4(q[(u) & 1], 4);                                                                                     \
    BL_MF4(A1, q[(u) & 1], acc[(u) < NLOC ? (u) : 0]);                                                           \
    if (2 * (u) + h < p) { BL_MF4(A2, q[(u) & 1], acc[(u) < NLOC ? NLOC - 1 - (u) : 0]); }
#define BL_CLOSE8 } } } } } } } }
  auto group = [&]() __attribute__((always_inline)) {
    const double wk = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(wt) + wrow);
    const double x10 = BL_RD(d1[0], 0), x20 = BL_RD(d2[0], 0);
    double q[2][4];
    BL_LDS4(q[0], b, 128 * (NC - 2));                // position h: chunk NC-1-h (the base carries 128 - 128 h)
    const double A1 = wk * x10, A2 = wk * x20;
    if (h == 0) {                                    // the diagonal cell of row m1 ...
      const double x11 = BL_RD(d1[1], 0), x12 = BL_RD(d1[2], 0);
      accD[0] = BL_MF(A1, x10, accD[0]);
      accD[1] = BL_MF(A1, x11, accD[1]);
      accD[2] = BL_MF(A1, x12, accD[2]);
    } else {                                         // ... or of row m2
      const double x21 = BL_RD(d2[1], 0), x22 = BL_RD(d2[2], 0);
      accD[0] = BL_MF(A2, x20, accD[0]);
      accD[1] = BL_MF(A2, x21, accD[1]);
      accD[2] = BL_MF(A2, x22, accD[2]);
    }
    BL_CELL(0) BL_CELL(1) BL_CELL(2) BL_CELL(3) BL_CELL(4) BL_CELL(5) BL_CELL(6) BL_CELL(7)
    BL_CLOSE8
    BL_WAIT4(q[0], 0);            // nothing in flight into registers the compiler believes settled
    BL_WAIT4(q[1], 0);
  };

  int64_t tl = blockIdx.x;
  if (tl < ntiles) fetch(tl, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);                            // vmcnt(0): this wave's pieces have landed
  __syncthreads();
  int buf = 0;
  for (; tl < ntiles; tl += gridDim.x) {
    const int64_t nxt = tl + gridDim.x;
    if (nxt < ntiles) fetch(nxt, buf ^ 1);
#pragma unroll 1
    for (int g = 0; g < RT / 4; ++g) {
      group();
      advance(4 * ROWB, 32);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                          // the next tile's pieces of this wave
    __syncthreads();
    // on to the other buffer's first group
    advance((buf ? -TILEB : TILEB) - RT * ROWB, (buf ? -8 * RT : 8 * RT) - 8 * RT);
    buf ^= 1;
  }
  // slab: [wave][NACC][lane], accumulator order: the diagonal cell's r = 0..2, then slot s, r
  double* out = partial + (size_t)blockIdx.x * (NW * NACC * 64) + (size_t)wave * (NACC * 64) + lane;
#pragma unroll
  for (int r = 0; r < 3; ++r) out[r * 64] = accD[r];
#pragma unroll
  for (int s = 0; s < NLOC; ++s)
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(3 + 4 * s + r) * 64] = acc[s][r];
}
#undef BL_CELL
#undef BL_LDS4
#undef BL_WAIT4
#undef BL_MF4
#undef BL_CLOSE8
#undef BL_RD

// PP from the slabs, fixed summation order.  Element (wave, a, lane), wave = (pair p, half h): a < 3: the diagonal cell
// of row m1 = p (h = 0) or m2 = NC-1-p (h = 1), r = a; else slot s = (a - 3) / 4, r = (a - 3) % 4: position j = 2s + h of row
// m1 (chunk NC-1-j) if j < NC-1-p, else position j = 2 (NC/2-1-s) + h of row m2 if j < p, else unused.  Lane (i, blk, j)
// holds PP[16m + 4blk + i][16n + 4((blk + r) & 3) + j]; of a diagonal cell, r = 0 gives the i <= j halves of the diagonal
// blocks, r = 1 the four neighbouring pairs, r = 2 blocks 0, 1 (2, 3 repeat them transposed).
template <int NC>
__global__ __launch_bounds__(1024) void k_reduce_q4_big(const double* __restrict__ partial, int nparts,
                                                        double* __restrict__ PP, int Pa)
{
  constexpr int NW = NC, NLOC = NC / 2, NACC = 4 * NLOC + 3;
  constexpr int E = NW * NACC * 64;
  __shared__ double sm[16][64];
  const int el = blockIdx.x * 64 + (threadIdx.x & 63);
  const int s16 = threadIdx.x >> 6;
  const double sum = el < E ? blk::slab_sum16(partial, E, el, s16, nparts) : 0.0;
  sm[s16][threadIdx.x & 63] = sum;
  __syncthreads();
  if (s16 == 0 && el < E) {
    const int l = threadIdx.x & 63;
    double tot = sm[0][l];
#pragma unroll
    for (int q = 1; q < 16; ++q) tot += sm[q][l];
    const int wave = el / (NACC * 64), a = (el / 64) % NACC;
    const int p = wave & (NLOC - 1), h = wave / NLOC;
    const int m1 = p, m2 = NC - 1 - p, split = NC - 1 - p;
    int m = 0, n = 0, r = 0;
    bool take = true;
    if (a < 3) {
      m = n = h ? m2 : m1;
      r = a;
    } else {
      const int s = (a - 3) >> 2;
      r = (a - 3) & 3;
      const int j1 = 2 * s + h, j2 = 2 * (NLOC - 1 - s) + h;
      if (j1 < split) { m = m1; n = NC - 1 - j1; }
      else if (j2 < p) { m = m2; n = NC - 1 - j2; }
      else take = false;
    }
    const int i = l >> 4, blk = (l >> 2) & 3, j = l & 3;
    const int A = 16 * m + 4 * blk + i, B = 16 * n + 4 * ((blk + r) & 3) + j;
    if (m == n) {
      if (r == 0) take = take && i <= j;
      if (r == 2) take = take && blk < 2;
    }
    if (take && A < Pa && B < Pa) {
      PP[A + (size_t)B * Pa] = tot;
      PP[B + (size_t)A * Pa] = tot;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// P = 256 by BLOCKS of 4 x 4 cells: the small matrix instruction is slower when an operand of it has just come from LDS
// (mfma_f64_shapes.hip), and the kernel above reads one per one or two instructions (6.4 ms per 4e6 rows; this one 5.9).  Here
// the 16 x 16 grid of cells (16-column chunks) is cut into 4 x 4 blocks: the six blocks above the diagonal go to waves 0..5
// (rows 4I..4I+3 against columns 4J..4J+3: 4 A operands, 16 reads of B_n^r, 64 instructions per 4-row group -- every read
// serves FOUR instructions), the four diagonal blocks two each to waves 6, 7 (4 + 16 reads, 36 instructions each).  Eight
// waves, two per SIMD, 64 / 72 accumulator doubles; tiles, staging and slabs as above.
constexpr int kBlkI[6] = {0, 0, 0, 1, 1, 2}, kBlkJ[6] = {1, 2, 3, 2, 3, 3};
constexpr int kBlkAcc = 72;
// accumulator index of cell (mi, ni), rotation r inside a diagonal block: per mi the diagonal cell's r = 0..2, then ni > mi
__host__ __device__ constexpr int diag_idx(int mi, int ni, int r)
{
  int base = 0;
  for (int q = 0; q < mi; ++q) base += 3 + 4 * (3 - q);
  return ni == mi ? base + r : base + 3 + 4 * (ni - mi - 1) + r;
}

__global__ __launch_bounds__(512, 2) void k_xwx_q4_blk16(const double* __restrict__ tX, const double* __restrict__ w,
                                                        int64_t N, double* __restrict__ partial)
{
  constexpr int NC = 16, NW = 8, P = 256;
  constexpr int ROWB = 128 * NC + 128, RT = 32, TILEB = RT * ROWB, HPR = P / 128, PIECES = RT * HPR;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* const tiles = lds;                                        // [2][RT][ROWB]
  double* const wt = reinterpret_cast<double*>(lds + 2 * TILEB);  // [2][RT]
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int k = lane >> 4, blk = (lane >> 2) & 3, e = lane & 3;
  const int64_t ntiles = (N + RT - 1) / RT;
  double acc[kBlkAcc];
#pragma unroll
  for (int i = 0; i < kBlkAcc; ++i) acc[i] = 0.0;

  // byte offsets of this lane for the group at hand: B_n^r of the block's column chunk ni is read at bB[r] + 128 ni, its row
  // chunk mi (unrotated) at bA + 128 mi; a diagonal block has both at the same chunks; waves 6, 7 walk two blocks
  const bool offd = wave < 6;
  const int I0 = offd ? kBlkI[wave < 6 ? wave : 0] : 2 * (wave - 6), J0 = offd ? kBlkJ[wave < 6 ? wave : 0] : I0;
  int bB[4], bA;
#pragma unroll
  for (int r = 0; r < 4; ++r) bB[r] = k * ROWB + 32 * ((blk + r) & 3) + 8 * e + 512 * J0;
  bA = k * ROWB + 32 * blk + 8 * e + 512 * I0;
  int wrow = 8 * k;
  auto advance = [&](int bytes, int wbytes) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 4; ++r) bB[r] += bytes;
    bA += bytes;
    wrow += wbytes;
#pragma unroll
    for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(bB[r]));
    asm volatile("" : "+v"(bA));
  };
  auto fetch = [&](int64_t tl, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < PIECES / NW; ++j) {
      const int piece = wave + j * NW;
      const int r = piece / HPR, part = piece % HPR;
      int64_t row = tl * RT + r;
      row = row < N ? row : N - 1;
      __builtin_amdgcn_global_load_lds(tX + (size_t)row * P + 128 * part + 2 * lane,
                                       tiles + buf * TILEB + r * ROWB + 1024 * part, 16, 0, 0);
    }
    if (t < RT) {
      const int64_t row = tl * RT + t;
      wt[buf * RT + t] = row < N ? w[row] : 0.0;
    }
  };
#define BL_RD(base, imm) (*reinterpret_cast<const double*>(tiles + (base) + (imm)))
#define BL_LDS4(q, base, imm)                                                                                    \
  asm volatile("ds_read_b64 %0, %4 offset:%8\n\tds_read_b64 %1, %5 offset:%8\n\tds_read_b64 %2, %6 offset:%8\n\t"    \
               "ds_read_b64 %3, %7 offset:%8"                                                                    \
               : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3])                                              \
               : "v"(base[0]), "v"(base[1]), "v"(base[2]), "v"(base[3]), "n"(imm))
#define BL_WAIT4(q, n) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]))
  // one 4-row group of an off-diagonal block: column chunk ni's four rotations against the four row chunks
  auto group_off = [&]() __attribute__((always_inline)) {
    const double wk = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(wt) + wrow);
    double q[2][4];
    BL_LDS4(q[0], bB, 0);
    double A[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) A[mi] = wk * BL_RD(bA, 128 * mi);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      if (ni < 3) BL_LDS4(q[(ni + 1) & 1], bB, 128 * (ni < 3 ? ni + 1 : 0));
      if (ni < 3) BL_WAIT4(q[ni & 1], 4); else BL_WAIT4(q[ni & 1], 0);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[(4 * mi + ni) * 4 + r] = BL_MF(A[mi], q[ni & 1][r], acc[(4 * mi + ni) * 4 + r]);
    }
  };
  // one 4-row group of the diagonal block whose accumulators start at `o` (0 or 36); the block's chunks: + boff bytes
  auto group_diag = [&](auto oc, int boff) __attribute__((always_inline)) {
    constexpr int o = decltype(oc)::value;
    const double wk = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(wt) + wrow);
    int bb[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bb[r] = bB[r] + boff;
    double q[2][4];
    BL_LDS4(q[0], bb, 0);
    double A[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) A[mi] = wk * BL_RD(bA + boff, 128 * mi);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      if (ni < 3) BL_LDS4(q[(ni + 1) & 1], bb, 128 * (ni < 3 ? ni + 1 : 0));
      if (ni < 3) BL_WAIT4(q[ni & 1], 4); else BL_WAIT4(q[ni & 1], 0);
#pragma unroll
      for (int mi = 0; mi <= ni; ++mi)
#pragma unroll
        for (int r = 0; r < (mi == ni ? 3 : 4); ++r)
          acc[o + diag_idx(mi, ni, r)] = BL_MF(A[mi], q[ni & 1][r], acc[o + diag_idx(mi, ni, r)]);
    }
  };

  int64_t tl = blockIdx.x;
  if (tl < ntiles) fetch(tl, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  int buf = 0;
  // (the two kinds of wave each run their own copy of the tile loop: one loop with an if inside merges the two
  // kinds' accumulators at every iteration, and the register allocator keeps both sets)
#define BL_TILE_LOOP(GROUPS)                                                                                     \
  for (; tl < ntiles; tl += gridDim.x) {                                                                         \
    const int64_t nxt = tl + gridDim.x;                                                                          \
    if (nxt < ntiles) fetch(nxt, buf ^ 1);                                                                       \
    _Pragma("unroll 1") for (int g = 0; g < RT / 4; ++g) {                                                       \
      GROUPS;                                                                                                    \
      advance(4 * ROWB, 32);                                                                                     \
    }                                                                                                            \
    __builtin_amdgcn_s_waitcnt(0x0F70);                                                                          \
    __syncthreads();                                                                                             \
    advance((buf ? -TILEB : TILEB) - RT * ROWB, (buf ? -8 * RT : 8 * RT) - 8 * RT);                              \
    buf ^= 1;                                                                                                    \
  }
  if (offd) {
    BL_TILE_LOOP(group_off())
  } else {
    BL_TILE_LOOP(group_diag(std::integral_constant<int, 0>{}, 0); group_diag(std::integral_constant<int, 36>{}, 512))
  }
#undef BL_TILE_LOOP
  double* out = partial + (size_t)blockIdx.x * (NW * kBlkAcc * 64) + (size_t)wave * (kBlkAcc * 64) + lane;
#pragma unroll
  for (int i = 0; i < kBlkAcc; ++i) out[i * 64] = acc[i];
}
#undef BL_LDS4
#undef BL_WAIT4
#undef BL_RD

// PP from the slabs [wave][72][64] of k_xwx_q4_blk16, fixed summation order.  Waves 0..5: accumulator (4 mi + ni) 4 + r =
// cell (4I + mi, 4J + ni), rotation r; waves 6, 7: a < 36 block I = 2 (wave - 6), else the next one, cell order diag_idx.
__global__ __launch_bounds__(1024) void k_reduce_q4_blk16(const double* __restrict__ partial, int nparts,
                                                          double* __restrict__ PP)
{
  constexpr int E = 8 * kBlkAcc * 64, Pa = 256;
  __shared__ double sm[16][64];
  const int el = blockIdx.x * 64 + (threadIdx.x & 63);
  const int s16 = threadIdx.x >> 6;
  const double sum = el < E ? blk::slab_sum16(partial, E, el, s16, nparts) : 0.0;
  sm[s16][threadIdx.x & 63] = sum;
  __syncthreads();
  if (s16 == 0 && el < E) {
    const int l = threadIdx.x & 63;
    double tot = sm[0][l];
#pragma unroll
    for (int q = 1; q < 16; ++q) tot += sm[q][l];
    const int wave = el / (kBlkAcc * 64), a = (el / 64) % kBlkAcc;
    int m = 0, n = 0, r = 0;
    bool take = true;
    if (wave < 6) {
      if (a >= 64) take = false;
      const int cell = a >> 2;
      r = a & 3;
      m = 4 * kBlkI[wave] + (cell >> 2);
      n = 4 * kBlkJ[wave] + (cell & 3);
    } else {
      const int I = 2 * (wave - 6) + (a >= 36 ? 1 : 0), aa = a >= 36 ? a - 36 : a;
      take = false;
      for (int mi = 0; mi < 4; ++mi)
        for (int ni = mi; ni < 4; ++ni)
          for (int rr = 0; rr < (mi == ni ? 3 : 4); ++rr)
            if (diag_idx(mi, ni, rr) == aa) {
              m = 4 * I + mi;
              n = 4 * I + ni;
              r = rr;
              take = true;
            }
    }
    const int i = l >> 4, blk = (l >> 2) & 3, j = l & 3;
    const int A = 16 * m + 4 * blk + i, B = 16 * n + 4 * ((blk + r) & 3) + j;
    if (m == n) {
      if (r == 0) take = take && i <= j;
      if (r == 2) take = take && blk < 2;
    }
    if (take) {
      PP[A + (size_t)B * Pa] = tot;
      PP[B + (size_t)A * Pa] = tot;
    }
  }
}

void launch_blk16(int nblocks, const double* tX, const double* w, int64_t N, double* partial, double* PP, hipStream_t s)
{
  constexpr size_t lds = 2 * 32 * (size_t)(128 * 16 + 128) + 2 * 32 * sizeof(double);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)k_xwx_q4_blk16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(k_xwx_q4_blk16, dim3(nblocks), dim3(512), lds, s, tX, w, N, partial);
  hipLaunchKernelGGL(k_reduce_q4_blk16, dim3((8 * kBlkAcc * 64 + 63) / 64), dim3(1024), 0, s, partial, nblocks, PP);
}

template <int NC>
void launch_x(int nblocks, const double* tX, const double* w, int64_t N, double* partial, double* PP, hipStream_t s)
{
  constexpr int NW = NC, NACC = 4 * (NC / 2) + 3;
  constexpr int RT = NC == 16 ? 32 : 64;
  constexpr size_t lds = 2 * RT * (size_t)(128 * NC + 128) + 2 * RT * sizeof(double);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)k_xwx_q4_big<NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((k_xwx_q4_big<NC>), dim3(nblocks), dim3(NW * 64), lds, s, tX, w, N, partial);
  hipLaunchKernelGGL((k_reduce_q4_big<NC>), dim3((NW * NACC * 64 + 63) / 64), dim3(1024), 0, s, partial, nblocks, PP, 16 * NC);
}

}  // namespace

namespace blk {

size_t xwx_q4_big_ws_doubles(int nblocks, int nc)
{
  const size_t a = (size_t)nblocks * nc * (2 * nc + 3) * 64, b = (size_t)nblocks * 8 * kBlkAcc * 64;
  return a > b ? a : b;
}

// P = 128 (nc = 8) or 256 (nc = 16) exactly; other P in (64, 256) stay on k_xwx_mfma_big (masked loads, padded columns)
void launch_xwx_q4_big(int nblocks, int nc, const double* tX, const double* w, int64_t N, double* partial, double* PP,
                       hipStream_t s)
{
  if (N <= 0) return;
  if (nc == 8) launch_x<8>(nblocks, tX, w, N, partial, PP, s);
  else launch_blk16(nblocks, tX, w, N, partial, PP, s);
}

}  // namespace blk
