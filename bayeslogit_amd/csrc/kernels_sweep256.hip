// kernels_sweep256.hip -- the logistic Gibbs sweep over this rank's rows with X read ONCE at P = 256 (config C5).
//
//   psi = X beta, omega_i ~ PG(n_i, psi_i), PPpart = sum_i omega_i x_i x_i'      (Logit.hpp:283-301,431)
//
// The two passes (k_psi_omega_nb<16> 4.2 ms + k_xwx_q4_blk16 16 ms per 12.5e6 x 256 shard) read X twice; the second is
// bound by the fp64 matrix pipe, which leaves the first -- a pure 6 TB/s stream plus the draws -- to be folded into it.
// One workgroup of eight waves per CU, as k_xwx_q4_blk16 (kernels_xwx4.hip: the small matrix instruction, the operand
// layout, tiles of X brought into LDS by LDS-DMA with a row stride of 128 NC + 128 bytes, every read of B_n^r conflict-free
// and serving four instructions), with three changes:
//
//   * THREE 16-row buffers instead of two of 32: while the matrix instructions run on tile i, tile i+2 is on its way from
//     HBM and tile i+1, already in LDS, is DRAWN: wave 0 forms psi of its 16 rows (the arithmetic of k_psi_omega_nb<16>:
//     the lane's sixteen products in column order, the 16-lane butterfly: same bits) and, as the P = 64 single-pass kernel
//     does (kernels_sweep1.hip), its four lanes per row evaluate attempts 0..3 of the row's draw ahead of time -- one
//     straight body, no queue, no loop; the first accepting attempt in block order is the draw (the value the work queue of
//     the two passes returns); the row weights go to LDS.  A row not settled by its four attempts, or whose first series
//     test is open, or with |psi|/2 >= 1/t, or n_i != 1, enters its tile with weight 0 and goes to the workgroup's list in
//     global memory; k_sweep_deferred256 draws those with the full sampler and gathers their rows of X again.
//   * fp64 vector and matrix instructions do not overlap on this part, so the draw is paid for in matrix-pipe time on wave 0's
//     SIMD (waves w and w + 4 of a workgroup always share one: scripts/experiments/wave_simd_map.hip), and the 528 matrix
//     instructions of a 4-row group are dealt out unevenly: per SIMD 100 + the draw / 144 / 144 / 140 instead of 128 / 128 /
//     136 / 136.  The 16 x 16 upper triangle of cells (16-column chunks) is cut into the four diagonal 4 x 4 blocks D0..D3 (36
//     instructions) and, right of block-row I, column strips of 4 x 1 cells (16 instructions: four A operands against the
//     four rotations of one chunk):
//        wave 0: the draw, nothing else          wave 4: block-row 2, chunks 12..15 + D3
//        wave 1: block-row 0, chunks 4..9        wave 5: block-row 1, chunks 8..10
//        wave 2: block-row 0, chunks 10..15      wave 6: block-row 1, chunks 11..13
//        wave 3: block-row 1, chunks 14, 15 + D1 wave 7: D0 + D2
//     (wave 0 with D3 beside the draw: 16.2 ms per 12.5e6-row shard against 15.7 -- a step waits for its slowest WAVE, and the
//     attempt body is a long dependent chain; without any draw the loop takes 15.15 ms, k_xwx_q4_blk16 alone 16.0.)
//   * a workgroup owns a contiguous range of tiles (its list of deferred rows lives in that range of two N-long arrays and
//     cannot overflow; fixed summation order: PP is reproducible and exactly symmetric).
//
// omega is that of the two passes to the last bit; PP differs in summation order only.
#include <type_traits>

#include "bl_dpp.hpp"
#include "bl_gibbs_kernels.hpp"
#include "bl_pg_devroye.hpp"
#include "bl_pg1_sm.hpp"

namespace {

using namespace bl;
typedef double v2d __attribute__((ext_vector_type(2)));

#define BL_MF(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0)

constexpr int kNC = 16, kP = 256, kNW = 8;
constexpr int kRowB = 128 * kNC + 128;       // bytes per row of a tile (consecutive rows half a bank window apart)
constexpr int kRT = 16;                      // rows per tile
constexpr int kTileB = kRT * kRowB;          // 34 816 bytes
constexpr int kNBuf = 3;
constexpr int kAccMax = 100;                 // accumulator doubles per lane of the widest role (four strips + a diagonal block)
constexpr int kSeg = 512;                    // deferred rows drawn at a time (their weights wait in LDS)

// what a wave multiplies: NS strips (chunks J0 .. J0+NS-1 against block-row I) and ND diagonal blocks (D[0..ND))
struct Role { int I, J0, NS, ND, D0, D1; };
__host__ __device__ constexpr Role role_of(int wave)
{
  switch (wave) {
    case 0: return Role{3, 12, 0, 0, 3, 3};
    case 1: return Role{0, 4, 6, 0, 0, 0};
    case 2: return Role{0, 10, 6, 0, 0, 0};
    case 3: return Role{1, 14, 2, 1, 1, 1};
    case 4: return Role{2, 12, 4, 1, 3, 3};
    case 5: return Role{1, 8, 3, 0, 0, 0};
    case 6: return Role{1, 11, 3, 0, 0, 0};
    default: return Role{0, 0, 0, 2, 0, 2};
  }
}
// accumulator index of cell (mi, ni), rotation r inside a diagonal block: per mi the diagonal cell's r = 0..2, then ni > mi
__host__ __device__ constexpr int diag_idx(int mi, int ni, int r)
{
  int base = 0;
  for (int q = 0; q < mi; ++q) base += 3 + 4 * (3 - q);
  return ni == mi ? base + r : base + 3 + 4 * (ni - mi - 1) + r;
}

// the full sampler for a deferred row (any class, any n): the observation's stream from block 0
__device__ __attribute__((noinline)) double draw_full256(int n, double psi, uint64_t seed, uint64_t idx, uint32_t epoch,
                                                         int* status, uint32_t blk0)
{
  const uint32_t s0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)seed);
  const uint32_t s1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(seed >> 32));
  epoch = (uint32_t)__builtin_amdgcn_readfirstlane((int)epoch);
  int st = 0;
  const double om = pg1_draw_n(n, psi, ((uint64_t)s1 << 32) | s0, idx, DOM_OMEGA, epoch, st, blk0);   // Logit.hpp:287
  if (st) atomicOr(status, st);
  return om;
}

struct Ctx {
  const double* tX;
  const double* nvec;
  const double* beta;
  double* w;
  int64_t N;
  uint64_t seed, idx0;
  uint32_t epoch;
  char* tiles;        // [kNBuf][kRT][kRowB]
  double* wt;         // [kNBuf][kRT] row weights of the tile in each buffer
  // this workgroup's range
  int64_t t0, ntl;    // first tile, number of tiles
  // deferred rows: the list (GATHER: what to multiply; else: where to append)
  uint32_t* defRow;   // offsets from row 16 t0, at index 16 t0 + position
  double* defPsi;
  const double* sOm;  // GATHER: weights of the segment's rows (LDS)
  const uint32_t* sRow;   // GATHER: their row offsets (LDS)
};

#define BL_RD(base, imm) (*reinterpret_cast<const double*>(cx.tiles + (base) + (imm)))
#define BL_LDS4(q, base, imm)                                                                                    \
  asm volatile("ds_read_b64 %0, %4 offset:%8\n\tds_read_b64 %1, %5 offset:%8\n\tds_read_b64 %2, %6 offset:%8\n\t"    \
               "ds_read_b64 %3, %7 offset:%8"                                                                    \
               : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3])                                              \
               : "v"(base[0]), "v"(base[1]), "v"(base[2]), "v"(base[3]), "n"(imm))
#define BL_WAIT4(q, n) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]))

// One wave's share of the workgroup's tile loop.  NS, ND: its role's sizes (role_of); DRAW: it also draws (wave 0 of the
// fused kernel); GATHER: the deferred kernel (rows from a list, weights from LDS, nothing drawn).  nDefOut: rows appended
// to the list (DRAW).
template <int NS, int ND, bool DRAW, bool GATHER>
__device__ __forceinline__ void wave_main(const Ctx& cx, const int wave, const int lane, double* __restrict__ slab,
                                          int& nDefOut, const bool first = true)
{
  constexpr int NACC = 16 * NS + 36 * ND;
  static_assert(NACC <= kAccMax, "slab too small");
  const Role ro = role_of(wave);
  const int k = lane >> 4, blk = (lane >> 2) & 3, e = lane & 3;
  double acc[NACC > 0 ? NACC : 1];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = first ? 0.0 : slab[(size_t)i * 64];   // (a further segment of the deferred kernel adds on)

  // byte offsets of this lane inside a tile, for the 4-row group at hand (they advance with it): the strips' B_n^r of chunk
  // J0 + ni at bB[r] + 128 ni, their A (unrotated, block-row I) at bA + 128 mi; diagonal block d: + boff[d] / + aoff[d]
  int bB[4], bA;
#pragma unroll
  for (int r = 0; r < 4; ++r) bB[r] = k * kRowB + 32 * ((blk + r) & 3) + 8 * e + 128 * ro.J0;
  bA = k * kRowB + 32 * blk + 8 * e + 512 * ro.I;
  const int aoff0 = 512 * (ro.D0 - ro.I), boff0 = 512 * ro.D0 - 128 * ro.J0;
  const int aoff1 = 512 * (ro.D1 - ro.I), boff1 = 512 * ro.D1 - 128 * ro.J0;
  auto advance = [&](int bytes) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 4; ++r) bB[r] += bytes;
    bA += bytes;
#pragma unroll
    for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(bB[r]));
    asm volatile("" : "+v"(bA));
  };

  // tile j of the range -> buffer b by LDS-DMA: a piece (one wave instruction) is 1 KB = half a row; wave w brings pieces
  // w, w + 8, w + 16, w + 24; lane i the 16 bytes at column 2 i of that half.  Rows past N read row N-1 (weight 0).
  auto fetch = [&](int64_t j, int b) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int piece = wave + 8 * q;
      const int r = piece >> 1, part = piece & 1;
      int64_t row;
      if (GATHER) {
        row = 16 * cx.t0 + (int64_t)cx.sRow[16 * j + r];          // (past the segment's end: entry 0's row, weight 0)
      } else {
        row = 16 * (cx.t0 + j) + r;
        row = row < cx.N ? row : cx.N - 1;
      }
      __builtin_amdgcn_global_load_lds(cx.tX + (size_t)row * kP + 128 * part + 2 * lane,
                                       cx.tiles + b * kTileB + r * kRowB + 1024 * part, 16, 0, 0);
    }
  };

  // ---- the draw of the tile in buffer b (rows 16 (t0 + j) ..): psi, four attempts per row, weights to wt[b]
  double bq[DRAW ? 16 : 1];
  const int c = lane & 15, a = c & 3, gq = c >> 2;
  const uint32_t k0 = (uint32_t)cx.seed, k1 = (uint32_t)(cx.seed >> 32);
  int nDef = 0;                                                    // wave-uniform
  if (DRAW) {
#pragma unroll
    for (int q = 0; q < 16; ++q) bq[q] = cx.beta[(q >> 1) * 32 + 2 * c + (q & 1)];   // colmap<16>(q, c) of kernels_gibbs.hip
  }
  auto load_n = [&](int64_t j) __attribute__((always_inline)) -> double {
    int64_t row = 16 * (cx.t0 + j) + 4 * gq + k;
    row = row < cx.N ? row : cx.N - 1;
    return cx.nvec[row];
  };
  auto draw_tile = [&](int64_t j, int b, double nn) __attribute__((always_inline)) {
#ifdef BL_S256_NODRAW            // timing experiment only (scripts/experiments): what the draw costs the tile loop
    if (a == 0) cx.wt[b * kRT + 4 * gq + k] = 0.125 + 0.0 * nn;
    return;
#endif
    const char* tile = cx.tiles + b * kTileB;
    double psi = 0.0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const char* rowp = tile + (4 * g + k) * kRowB + 16 * c;
      double part = 0.0;
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        const v2d v = *reinterpret_cast<const v2d*>(rowp + 256 * h);
        part += v.x * bq[2 * h];
        part += v.y * bq[2 * h + 1];
      }
      part = row16_allsum(part);          // (bl_dpp.hpp: the xor butterfly's bits, without LDS)
      psi = (gq == g) ? part : psi;
    }
    const int64_t row = 16 * (cx.t0 + j) + 4 * gq + k;               // the row this lane's quad draws
    const double Z = fabs(psi) * 0.5;                                // PolyaGamma.cpp:154
    const double fz = kSmPiSq8 + 0.5 * Z * Z;                        // :157
    const double mass = pg1_mass_small(Z, fz);                       // rows outside the class are masked below
    const uint64_t idx = cx.idx0 + (uint64_t)row;
    const U4 o = philox4x32_10((uint32_t)idx, ctr1_of(idx, DOM_OMEGA), cx.epoch, (uint32_t)a, k0, k1);
    double X;
    const int verdict = pg1_attempt_small_known(a == 0, Z, fz, mass, u52(o.x, o.y), u52(o.z, o.w), X);
    // the row's first attempt (in block order) that does not end in a retry decides: accepted -> the draw; series test
    // open, or none of the four, or not a fast row -> deferred (kernels_sweep1.hip: settle)
    const bool inrange = row < cx.N;
    const bool fast = inrange && (kSmTRecip > Z) && nn == 1.0;       // :87; n = 1: one PG(1, psi) draw
    const uint64_t bAcc = __ballot(fast && verdict == 1), bStop = __ballot(fast && verdict != 0);
    const int sh = lane & ~3;
    const uint32_t nAcc = (uint32_t)(bAcc >> sh) & 15u, nStop = (uint32_t)(bStop >> sh) & 15u;
    const uint32_t first = nStop & (0u - nStop);
    const bool settled = (first & nAcc) != 0u;
    const int wl = (int)__builtin_ctz(first | 16u);
    const double Xw = __shfl(X, sh | (wl & 3));
    const double om = settled ? 0.25 * Xw : 0.0;                     // :201
    if (a == 0) cx.wt[b * kRT + 4 * gq + k] = om;
    if (cx.w && settled && a == 0) cx.w[row] = om;
    const bool defer = inrange && !settled && a == 0;
    const uint64_t dm = __ballot(defer);
    if (defer) {
      const int64_t slot = 16 * cx.t0 + nDef +
                           (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(dm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dm, 0u));
      // bit 31: four retries inside the left piece -- the deferred kernel goes on at block 4 (kernels_sweep1.hip: settle)
      cx.defRow[slot] = (uint32_t)(row - 16 * cx.t0) | ((fast && nStop == 0u) ? 0x80000000u : 0u);
      cx.defPsi[slot] = psi;
    }
    nDef += __popcll(dm);
  };

  // ---- one 4-row group of the tile the offsets point at; wk = this lane's row weight
  auto group = [&](double wk) __attribute__((always_inline)) {
    if (NS > 0) {
      double q[2][4];
      BL_LDS4(q[0], bB, 0);
      double A[4];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) A[mi] = wk * BL_RD(bA, 128 * mi);
#pragma unroll
      for (int ni = 0; ni < NS; ++ni) {
        if (ni + 1 < NS) BL_LDS4(q[(ni + 1) & 1], bB, 128 * (ni + 1 < NS ? ni + 1 : 0));
        if (ni + 1 < NS) BL_WAIT4(q[ni & 1], 4); else BL_WAIT4(q[ni & 1], 0);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[(4 * ni + mi) * 4 + r] = BL_MF(A[mi], q[ni & 1][r], acc[(4 * ni + mi) * 4 + r]);
      }
    }
#pragma unroll
    for (int d = 0; d < ND; ++d) {
      constexpr int ob = 16 * NS;
      const int o = ob + 36 * d;
      const int ao = d ? aoff1 : aoff0, bo = d ? boff1 : boff0;
      int bb[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) bb[r] = bB[r] + bo;
      double q[2][4];
      BL_LDS4(q[0], bb, 0);
      double A[4];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) A[mi] = wk * BL_RD(bA + ao, 128 * mi);
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
    }
  };

  // ---- the pipeline over the range's tiles: fetch j+2 | draw j+1 | multiply j
  const int64_t ntl = cx.ntl;
  double nn_next = 1.0;
  if (ntl > 0) {
    fetch(0, 0);
    if (ntl > 1) fetch(1, 1);
    double nn0 = 1.0;
    if (DRAW) {
      nn0 = load_n(0);
      if (ntl > 1) nn_next = load_n(1);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                              // vmcnt(0): this wave's pieces have landed
    __syncthreads();
    if (DRAW) draw_tile(0, 0, nn0);
    __syncthreads();
  }
  int buf = 0;
#pragma unroll 1
  for (int64_t j = 0; j < ntl; ++j) {
    const int b1 = buf == kNBuf - 1 ? 0 : buf + 1, b2 = b1 == kNBuf - 1 ? 0 : b1 + 1;
    double nn_use = nn_next;
    if (j + 2 < ntl) {
      if (DRAW) nn_next = load_n(j + 2);
      fetch(j + 2, b2);
    }
    if (DRAW && j + 1 < ntl) draw_tile(j + 1, b1, nn_use);
#pragma unroll 1
    for (int g = 0; g < kRT / 4; ++g) {
      const double wk = GATHER ? cx.sOm[16 * j + 4 * g + k] : cx.wt[buf * kRT + 4 * g + k];
      group(wk);
      advance(4 * kRowB);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                              // the pieces of tile j+2 this wave brought
    __syncthreads();
    advance((b1 - buf) * kTileB - kRT * kRowB);                      // on to the next buffer's first group
    buf = b1;
  }
  // slab: [wave][kAccMax][lane]
#pragma unroll
  for (int i = 0; i < NACC; ++i) slab[(size_t)i * 64] = acc[i];
#pragma unroll
  for (int i = NACC; i < kAccMax; ++i) slab[(size_t)i * 64] = 0.0;
  nDefOut = nDef;
}

// the workgroup's share of the tiles: a contiguous range
__device__ __forceinline__ void block_range(int64_t N, int nblocks, int b, int64_t& t0, int64_t& ntl)
{
  const int64_t ntiles = (N + kRT - 1) / kRT;
  const int64_t per = (ntiles + nblocks - 1) / nblocks;
  t0 = (int64_t)b * per;
  const int64_t t1 = (t0 + per) < ntiles ? (t0 + per) : ntiles;
  ntl = t1 > t0 ? t1 - t0 : 0;
}

template <bool GATHER>
__device__ __forceinline__ void dispatch_roles(const Ctx& cx, int wave, int lane, double* slab, int& nDef, bool first = true)
{
  switch (wave) {          // (a scalar: the waves branch apart once)
    case 0: wave_main<0, 0, !GATHER, GATHER>(cx, 0, lane, slab, nDef, first); break;
    case 1: wave_main<6, 0, false, GATHER>(cx, 1, lane, slab, nDef, first); break;
    case 2: wave_main<6, 0, false, GATHER>(cx, 2, lane, slab, nDef, first); break;
    case 3: wave_main<2, 1, false, GATHER>(cx, 3, lane, slab, nDef, first); break;
    case 4: wave_main<4, 1, false, GATHER>(cx, 4, lane, slab, nDef, first); break;
    case 5: wave_main<3, 0, false, GATHER>(cx, 5, lane, slab, nDef, first); break;
    case 6: wave_main<3, 0, false, GATHER>(cx, 6, lane, slab, nDef, first); break;
    default: wave_main<0, 2, false, GATHER>(cx, 7, lane, slab, nDef, first); break;
  }
}

__global__ __launch_bounds__(512, 2) void k_sweep_once256(const double* __restrict__ tX, const double* __restrict__ nvec,
                                                          const double* __restrict__ beta, double* __restrict__ w,
                                                          int64_t N, uint64_t seed, uint32_t epoch, uint64_t idx0,
                                                          double* __restrict__ partial, uint32_t* __restrict__ defRow,
                                                          double* __restrict__ defPsi, uint32_t* __restrict__ defCnt)
{
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  Ctx cx;
  cx.tX = tX; cx.nvec = nvec; cx.beta = beta; cx.w = w; cx.N = N; cx.seed = seed; cx.idx0 = idx0; cx.epoch = epoch;
  cx.tiles = lds;
  cx.wt = reinterpret_cast<double*>(lds + kNBuf * kTileB);
  block_range(N, gridDim.x, blockIdx.x, cx.t0, cx.ntl);
  cx.defRow = defRow; cx.defPsi = defPsi; cx.sOm = nullptr; cx.sRow = nullptr;
  int nDef = 0;
  double* slab = partial + (size_t)blockIdx.x * (kNW * kAccMax * 64) + (size_t)wave * (kAccMax * 64) + lane;
  dispatch_roles<false>(cx, wave, lane, slab, nDef);
  if (wave == 0 && lane == 0) defCnt[blockIdx.x] = (uint32_t)nDef;
}

// The rows k_sweep_once256 left: workgroup b of this grid (the same grid) takes the list workgroup b of that kernel wrote,
// kSeg rows at a time: one thread per row runs the full sampler, the weights and row offsets wait in LDS, and the rows of X
// are gathered 16 at a time through the same three buffers and the same matrix instructions.  Its slabs follow the first
// kernel's.
__global__ __launch_bounds__(512, 2) void k_sweep_deferred256(const double* __restrict__ tX, const double* __restrict__ nvec,
                                                              double* __restrict__ w, int64_t N, uint64_t seed,
                                                              uint32_t epoch, uint64_t idx0, double* __restrict__ partial,
                                                              int* __restrict__ status, const uint32_t* __restrict__ defRow,
                                                              const double* __restrict__ defPsi,
                                                              const uint32_t* __restrict__ defCnt,
                                                              unsigned long long* __restrict__ stats,
                                                              unsigned long long* __restrict__ hstats)
{
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double* sOm = reinterpret_cast<double*>(lds + kNBuf * kTileB + kNBuf * kRT * 8);
  uint32_t* sRow = reinterpret_cast<uint32_t*>(sOm + kSeg);
  Ctx cx;
  cx.tX = tX; cx.nvec = nvec; cx.beta = nullptr; cx.w = w; cx.N = N; cx.seed = seed; cx.idx0 = idx0; cx.epoch = epoch;
  cx.tiles = lds;
  cx.wt = reinterpret_cast<double*>(lds + kNBuf * kTileB);
  int64_t ntl_all;
  block_range(N, gridDim.x, blockIdx.x, cx.t0, ntl_all);
  cx.defRow = nullptr; cx.defPsi = nullptr; cx.sOm = sOm; cx.sRow = sRow;
  const int cnt = ntl_all > 0 ? (int)defCnt[blockIdx.x] : 0;
  double* slab = partial + (size_t)blockIdx.x * (kNW * kAccMax * 64) + (size_t)wave * (kAccMax * 64) + lane;
  if (cnt == 0) {                              // (uniform) nothing was deferred here: an empty slab
    for (int i = 0; i < kAccMax; ++i) slab[(size_t)i * 64] = 0.0;
    return;
  }
  // kSeg rows = 32 tiles per pass of the pipeline; a further segment's wave_main starts from the slab the last one wrote
  const int64_t lbase = 16 * cx.t0;
  for (int s0 = 0; s0 < cnt; s0 += kSeg) {
    const int n = (cnt - s0) < kSeg ? (cnt - s0) : kSeg;
    __syncthreads();
    {
      const int i = threadIdx.x;               // kSeg == blockDim.x
      double om = 0.0;
      uint32_t ro = defRow[lbase + s0];        // past the end: the segment's first row, weight 0
      if (i < n) {
        ro = defRow[lbase + s0 + i];
        const int64_t grow = lbase + (int64_t)(ro & 0x7FFFFFFFu);
        om = draw_full256((int)nvec[grow] /* (int) n(i), Logit.hpp:287 */, defPsi[lbase + s0 + i], seed, idx0 + (uint64_t)grow,
                          epoch, status, (ro >> 31) ? 4u : 0u);
        if (w) w[grow] = om;
      }
      sOm[i] = om;
      sRow[i] = ro & 0x7FFFFFFFu;
    }
    __syncthreads();
    cx.ntl = (n + kRT - 1) / kRT;
    int nd = 0;
    dispatch_roles<true>(cx, wave, lane, slab, nd, s0 == 0);      // a further segment starts from the slab the last one wrote
  }
  if (threadIdx.x == 0) {
    if (stats) atomicAdd(stats, (unsigned long long)cnt);
    atomicAdd(hstats, (unsigned long long)cnt);      // the handle's own count (fall-back policy)
  }
}

// PP from the slabs [wave][kAccMax][64] of both kernels, fixed summation order.  Wave w, accumulator a: a < 16 NS: strip
// ni = a / 16, mi = (a / 4) % 4, r = a % 4: cell (4 I + mi, J0 + ni); else diagonal block d = (a - 16 NS) / 36 in the order
// of diag_idx.  Lane (i, blk, j) holds PP[16m + 4blk + i][16n + 4((blk + r) & 3) + j]; of a diagonal cell, r = 0 gives the
// i <= j halves of the diagonal blocks, r = 1 the four neighbouring pairs, r = 2 blocks 0, 1 (2, 3 repeat them transposed).
__global__ __launch_bounds__(1024) void k_reduce_256(const double* __restrict__ partial, int nparts, double* __restrict__ PP)
{
  constexpr int E = kNW * kAccMax * 64, Pa = 256;
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
    const int wave = el / (kAccMax * 64), a = (el / 64) % kAccMax;
    const Role ro = role_of(wave);
    int m = 0, n = 0, r = 0;
    bool take = false;
    if (a < 16 * ro.NS) {
      const int ni = a >> 4, mi = (a >> 2) & 3;
      r = a & 3;
      m = 4 * ro.I + mi;
      n = ro.J0 + ni;
      take = true;
    } else if (a < 16 * ro.NS + 36 * ro.ND) {
      const int d = (a - 16 * ro.NS) / 36, aa = (a - 16 * ro.NS) % 36;
      const int D = d ? ro.D1 : ro.D0;
      for (int mi = 0; mi < 4; ++mi)
        for (int ni = mi; ni < 4; ++ni)
          for (int rr = 0; rr < (mi == ni ? 3 : 4); ++rr)
            if (diag_idx(mi, ni, rr) == aa) {
              m = 4 * D + mi;
              n = 4 * D + ni;
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
#undef BL_LDS4
#undef BL_WAIT4
#undef BL_RD

constexpr size_t kLdsOnce = (size_t)kNBuf * kTileB + kNBuf * kRT * 8;
constexpr size_t kLdsDef = kLdsOnce + kSeg * 8 + kSeg * 4;

}  // namespace

namespace blk {

// workspace: 2 x nblocks slabs of 8 x 96 x 64 doubles, the deferred rows' psi (Npad doubles) and row offsets (Npad
// uint32), one count per workgroup, the handle's counter of deferred rows
static size_t npad256(int64_t N) { return (size_t)((N > 0 ? N : 0) + 15) / 16 * 16 + 16; }
size_t sweep_once256_ws_doubles(int nblocks, int64_t N)
{
  const size_t n = npad256(N);
  return (size_t)nblocks * 2 * kNW * kAccMax * 64 + n + (n + 1) / 2 + (size_t)(nblocks + 1) / 2 + 8;
}

unsigned long long* sweep_once256_deferred_counter(double* ws, int nblocks, int64_t N)
{
  const size_t n = npad256(N);
  return reinterpret_cast<unsigned long long*>(ws + (size_t)nblocks * 2 * kNW * kAccMax * 64 + n + (n + 1) / 2 +
                                               (size_t)(nblocks + 1) / 2);
}

void launch_sweep_once256(int nblocks, const double* tX, const double* n, const double* beta, double* w, int64_t N,
                          double* ws, double* PP, uint64_t seed, uint32_t epoch, uint64_t idx0, int* status,
                          unsigned long long* stats, hipStream_t s)
{
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)k_sweep_once256, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsOnce);
    (void)hipFuncSetAttribute((const void*)k_sweep_deferred256, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsDef);
    attr_set = true;
  }
  const size_t nn = npad256(N);
  double* slabs = ws;
  double* defPsi = slabs + (size_t)nblocks * 2 * kNW * kAccMax * 64;
  uint32_t* defRow = reinterpret_cast<uint32_t*>(defPsi + nn);
  uint32_t* defCnt = defRow + 2 * ((nn + 1) / 2);
  unsigned long long* hstats = sweep_once256_deferred_counter(ws, nblocks, N);
  hipLaunchKernelGGL(k_sweep_once256, dim3(nblocks), dim3(512), kLdsOnce, s, tX, n, beta, w, N, seed, epoch, idx0, slabs,
                     defRow, defPsi, defCnt);
  hipLaunchKernelGGL(k_sweep_deferred256, dim3(nblocks), dim3(512), kLdsDef, s, tX, n, w, N, seed, epoch, idx0,
                     slabs + (size_t)nblocks * kNW * kAccMax * 64, status, defRow, defPsi, defCnt, stats, hstats);
  hipLaunchKernelGGL(k_reduce_256, dim3((kNW * kAccMax * 64 + 63) / 64), dim3(1024), 0, s, slabs, 2 * nblocks, PP);
}

}  // namespace blk
