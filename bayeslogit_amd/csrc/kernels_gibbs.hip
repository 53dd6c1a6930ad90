// kernels_gibbs.hip -- kernels of the logistic Gibbs sweep on MI355X (gfx950).
// One sweep over this rank's rows of X (Logit.hpp:426-431) is two streaming passes:
//   pass 1  k_psi_omega_nb : psi = X beta by coalesced 16-byte loads + a 16-lane butterfly,
//           then omega ~ PG(n, psi) by the wavefront work queue of bl_pg1_queue.hpp (one draw per
//           lane, idle lanes refilled by ballot / prefix count); writes omega (8 B/row).
//   pass 2  k_xwx_mfma     : X' Omega X as a rank-N update on the fp64 matrix pipe
//           (v_mfma_f64_16x16x4_f64, upper-triangle 16x16 blocks), 2 waves per SIMD.
// replacing gemm(psi) + draw_w + the P x N temp + syrk of Logit.hpp:283-301,431.
// (A single-pass version that kept each 64-row tile in registers between the two uses was
// built first and measured 5.8 ms/sweep at N=1e7, P=64: it is confined to one wave per SIMD
// and the draw's dependent fp64 chains then run at latency, not throughput.  DESIGN.md.)
//   * any P <= 256 runs on these kernels: P <= 64 on the register-tile pair (columns padded to 16 NB with
//     zeros in registers when P is not a multiple of 16, 8-byte masked loads), 64 < P <= 256 on the LDS-tile
//     kernel k_xwx_mfma_big (128 or 256 columns); generic kernels (k_psi_omega + k_xwx_tiles) above that.
//   * fixed-order reductions (no float atomics): every bit of PP is reproducible.
//   * k_beta: the P x P stage (Cholesky, solves, both beta draws) in one workgroup.
#include "bl_gibbs_kernels.hpp"
#include "bl_pg_devroye.hpp"
#include "bl_pg1_queue.hpp"
#include "../../include/bayeslogit_hip.h"

namespace {

using namespace bl;
typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int kBlock = 256;

// omega for one observation
template <int MODE>
__device__ __forceinline__ double weight_of(double psi, double n, uint64_t seed, uint64_t idx, uint32_t epoch, int& st)
{
  if (MODE == blk::W_DRAW) {
    return pg1_draw_n((int)n, psi, seed, idx, DOM_OMEGA, epoch, st);   // Logit.hpp:287
  } else {
    const double hpsi = psi * 0.5;                      // Logit.hpp:509-519
    if (fabs(hpsi) < 0.01)
      return n / cosh(hpsi) * (1 + hpsi * hpsi / 6.0 + pow(hpsi, 4.0) / 120.0 + pow(hpsi, 6) / 5040.0) * 0.25;
    return n * tanh(hpsi) / hpsi * 0.25;
  }
}

// ===================================================== P in {16,32,48,64}: two streaming passes
// Column of X held by lane-column c in MFMA block q.  The assignment is chosen so
// that a lane's loads are 16-byte vectors and a wavefront's load instruction covers
// whole 128-byte lines; PP is un-permuted in the reduction epilogue.
template <int NB>
__device__ __host__ __forceinline__ int colmap(int q, int c)
{
  if (NB == 1) return c;
  if (NB == 2) return 2 * c + q;
  if (NB == 3) return q < 2 ? 2 * c + q : 32 + c;
  return (q >> 1) * 32 + 2 * c + (q & 1);        // NB even: block pair h = q>>1 covers columns 32h .. 32h+31
}

// One group = 4 consecutive rows; lane (k = lane>>4, c = lane&15) takes row k of the group and
// the NB columns colmap(q, c).  ok = row in range (out-of-range rows read row 0 and are zeroed).
// Pa = the matrix's real number of columns (row stride).  EXACT (Pa == 16 NB): 16-byte vector loads;
// otherwise the same lane-to-column assignment with 8-byte loads, columns >= Pa read as zero.  (A
// compile-time switch: with a runtime test the exact path carried the other one's registers.)
template <int NB, bool EXACT>
__device__ __forceinline__ void load_group(double (&xg)[NB], const double* __restrict__ tX, int64_t row, bool ok, int c,
                                           int Pa)
{
  constexpr int P = 16 * NB;
  const double* p = tX + (size_t)(ok ? row : 0) * (size_t)(EXACT ? P : Pa);
  if (!EXACT) {
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const int col = colmap<NB>(q, c);
      const double v = p[col < Pa ? col : Pa - 1];
      xg[q] = (ok && col < Pa) ? v : 0.0;
    }
  } else if (NB == 1) {
    const double v = p[c];
    xg[0] = ok ? v : 0.0;
  } else if (NB == 3) {
    const v2d v0 = *reinterpret_cast<const v2d*>(p + 2 * c);
    const double v = p[32 + c];
    xg[0] = ok ? v0.x : 0.0;
    xg[1] = ok ? v0.y : 0.0;
    xg[2] = ok ? v : 0.0;
  } else {
#pragma unroll
    for (int h = 0; h < NB / 2; ++h) {
      const v2d v = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(p + 32 * h + 2 * c));
      xg[2 * h] = ok ? v.x : 0.0;
      xg[2 * h + 1] = ok ? v.y : 0.0;
    }
  }
}

template <int CTRL>
__device__ __forceinline__ double dppmov_f64(double v)      // every lane has a valid source: no `old` copy
{
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// ---- pass 1: psi = X beta (- off), omega ~ PG(n, psi) -> w[]          (Logit.hpp:431,283-289)
// A wave takes chunks of `chunk` <= kSuper rows (make_plan sizes them so that every resident wave
// gets the same number of chunks).  Phase 1 streams the rows once (coalesced 16-byte loads),
// forms psi by a 16-lane butterfly, parks psi in w[] (re-read from L2 when a lane starts the
// row), leaves the proposal mass in LDS and compacts the rows by sampler class into two index
// lists; phase 2 is the work queue of bl_pg1_queue.hpp, once per class -- the same structure as
// kernels_pg.hip's k_rpg_devroye with X beta in place of a z vector.
constexpr int kSuper = 512;

// Phase 2 of the psi/omega pass as an out-of-line call: the draw's polynomial constants and state
// then live in registers only while a chunk is being drawn, not across the streaming loop of
// phase 1 (inlined, they were hoisted to kernel entry and starved that loop of registers: its
// loads serialised behind scratch traffic).
__device__ __forceinline__ uint32_t uni32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uni64(uint64_t v)
{
  return ((uint64_t)uni32((uint32_t)(v >> 32)) << 32) | uni32((uint32_t)v);
}
template <typename T>
__device__ __forceinline__ T* uniptr(T* p) { return reinterpret_cast<T*>(uni64(reinterpret_cast<uint64_t>(p))); }

// (arguments of an out-of-line device function arrive in VGPRs; all but lt_mask are wave-uniform and
// are moved back to SGPRs first.)  Returns the sampler status flags of the chunk.
template <int ZC>
__device__ __attribute__((noinline)) int draw_chunk(const unsigned short* list_, int n_, const double* sZw_,
                                                    const double* sMw_, const int* sNw_, double* w_, int64_t base_,
                                                    uint64_t idx0_, uint32_t epoch_, uint32_t k0_, uint32_t k1_,
                                                    uint64_t lt_mask)
{
  const unsigned short* list = uniptr(list_);
  const double* sZw = uniptr(sZw_);
  const double* sMw = uniptr(sMw_);
  const int* sNw = uniptr(sNw_);
  double* w = uniptr(w_);
  const int n = (int)uni32((uint32_t)n_);
  const int64_t base = (int64_t)uni64((uint64_t)base_);
  const uint64_t idx0 = uni64(idx0_);
  const uint32_t epoch = uni32(epoch_), k0 = uni32(k0_), k1 = uni32(k1_);
  int st_flags = 0;
  devroye_queue<ZC, 2, int, true, DOM_OMEGA>(list, n, sZw, sMw, w, sNw, 1, base, idx0, epoch, k0, k1, lt_mask, st_flags);
  return st_flags;
}

// The |psi|/2 >= 1/t rows are rare in a Gibbs sweep (0.5 % at C4) but their sampler is the heavy one, and an
// out-of-line call that uses most of the register file saves ~110 callee-saved registers per lane to scratch:
// paid once per 512-row chunk that was 1 GB of scratch traffic per pass.  Such rows are therefore DEFERRED:
// phase 1 parks their psi in w[] and their row offset in a per-wave LDS list, and this function draws the
// whole list at once (when it fills, and at the end of the wave's range), re-using the chunk's LDS arrays.
constexpr int kDefCap = 1024;    // >= 2 kSuper: a chunk adds at most kSuper entries and the list is flushed above kDefCap - kSuper
__device__ __attribute__((noinline)) int draw_deferred(const uint32_t* list_, int nd_, double* sZw_, double* sMw_,
                                                       int* sNw_, unsigned short* sIdxw_, double* w_,
                                                       const double* nvec_, int64_t r0_, uint64_t idx0_,
                                                       uint32_t epoch_, uint32_t k0_, uint32_t k1_, uint64_t lt_mask)
{
  const uint32_t* list = uniptr(list_);
  double* sZw = uniptr(sZw_);
  double* sMw = uniptr(sMw_);
  int* sNw = uniptr(sNw_);
  unsigned short* sIdxw = uniptr(sIdxw_);
  double* w = uniptr(w_);
  const double* nvec = uniptr(nvec_);
  const int nd = (int)uni32((uint32_t)nd_);
  const int64_t r0 = (int64_t)uni64((uint64_t)r0_);
  const uint64_t idx0 = uni64(idx0_);
  const uint32_t epoch = uni32(epoch_), k0 = uni32(k0_), k1 = uni32(k1_);
  const int lane = threadIdx.x & 63;
  int st_flags = 0;
  for (int seg = 0; seg < nd; seg += kSuper) {            // the chunk arrays hold kSuper entries at a time
    const int n = (nd - seg) < kSuper ? (nd - seg) : kSuper;
    for (int i0 = 0; i0 < n; i0 += 64) {
      const int i = i0 + lane;
      if (i < n) {
        const int64_t row = r0 + (int64_t)list[seg + i];
        const double psi = w[row];
        const double Z = fabs(psi) * 0.5;
        sZw[i] = psi;
        sMw[i] = pg1_mass(Z, kSmPiSq8 + 0.5 * Z * Z);
        sNw[i] = (int)nvec[row];                            // (int) n(i), Logit.hpp:287
        sIdxw[i] = (unsigned short)i;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    Pg1Slot L;
    devroye_queue_run<2, 2, int, true, DOM_OMEGA>(L, true, sIdxw, n, sZw, sMw, w, sNw, 1, r0, idx0, epoch, k0, k1, lt_mask, st_flags,
                                       list + seg);
    __builtin_amdgcn_wave_barrier();
  }
  return st_flags;
}

template <int NB, int MODE, bool EXACT>
__global__ __launch_bounds__(kBlock, 2) void k_psi_omega_nb(const double* __restrict__ tX,
                                                            const double* __restrict__ nvec,
                                                            const double* __restrict__ beta,
                                                            const double* __restrict__ off,
                                                            double* __restrict__ w, int64_t N, int Pa, int chunk,
                                                            uint64_t seed, uint32_t epoch, uint64_t idx0,
                                                            int* __restrict__ status)
{
  __shared__ double sM[kBlock / 64][kSuper];
  __shared__ double sZ[kBlock / 64][kSuper];
  __shared__ int sN[kBlock / 64][kSuper];
  __shared__ unsigned short sIdx[kBlock / 64][kSuper];   // the chunk's class-1 rows
  __shared__ uint32_t sDef[kBlock / 64][kDefCap];        // deferred class-2 rows (offsets from the wave's first row)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = lane >> 4, c = lane & 15;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  int st_flags = 0;
  int nDef = 0;                                          // wave-uniform
  double bq[NB];
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    const int col = colmap<NB>(q, c);
    bq[q] = (EXACT || col < Pa) ? beta[col] : 0.0;
  }

  // this wave's contiguous row range, cut into chunks
  const int64_t nwaves = (int64_t)gridDim.x * (kBlock / 64);
  const int64_t per_wave = ((N + nwaves - 1) / nwaves + 63) / 64 * 64;
  const int64_t r0 = ((int64_t)blockIdx.x * (kBlock / 64) + wave) * per_wave;
  const int64_t r1 = (r0 + per_wave) < N ? (r0 + per_wave) : N;
  for (int64_t base = r0; base < r1;) {
    const int cnt = (int)((r1 - base) < chunk ? (r1 - base) : chunk);
    int nA = 0;           // wave-uniform list length
    // phase 1: psi for the rows of the chunk, 64 rows (16 groups) at a time
    for (int t0 = 0; t0 < cnt; t0 += 64) {
      double psi = 0.0;
#pragma unroll 4
      for (int g = 0; g < 16; ++g) {
        const int64_t row = base + t0 + 4 * g + k;
        double xg[NB];
        load_group<NB, EXACT>(xg, tX, row, row < N, c, Pa);
        double part = 0.0;
#pragma unroll
        for (int q = 0; q < NB; ++q) part += xg[q] * bq[q];
        part += __shfl_xor(part, 1);
        part += __shfl_xor(part, 2);
        part += __shfl_xor(part, 4);
        part += __shfl_xor(part, 8);
        psi = (c == g) ? part : psi;
      }
      const int slot = t0 + 4 * c + k;          // the row whose psi this lane kept
      bool small = false, large = false;
      if (slot < cnt) {
        if (off) psi -= off[base + slot];
        if (MODE == blk::W_DRAW) {
          const double Z = fabs(psi) * 0.5;                     // PolyaGamma.cpp:154
          small = kSmTRecip > Z;                // PolyaGamma.cpp:87
          large = !small;
          if (small) {
            sZ[wave][slot] = psi;
            sM[wave][slot] = pg1_mass_small(Z, kSmPiSq8 + 0.5 * Z * Z);
            sN[wave][slot] = (int)nvec[base + slot];            // (int) n(i), Logit.hpp:287
          } else {
            w[base + slot] = psi;                               // parked until the deferred list is drawn
          }
        } else {
          w[base + slot] = weight_of<MODE>(psi, nvec[base + slot], seed, 0, epoch, st_flags);
        }
      }
      if (MODE == blk::W_DRAW) {
        const uint64_t ma = __ballot(small), mb = __ballot(large);
        if (small) sIdx[wave][nA + __popcll(ma & lt_mask)] = (unsigned short)slot;
        if (large) sDef[wave][nDef + __popcll(mb & lt_mask)] = (uint32_t)(base + slot - r0);
        nA += __popcll(ma);
        nDef += __popcll(mb);
      }
    }
    if (MODE != blk::W_DRAW) { base += cnt; continue; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // LDS lists complete; parked psi stores have left
    __builtin_amdgcn_wave_barrier();
    // phase 2: the work queue over the chunk's class-1 rows; the deferred class-2 rows when their list could
    // not take another chunk's worth, and after the wave's last chunk
    if (nA > 0)
      st_flags |= draw_chunk<1>(&sIdx[wave][0], nA, sZ[wave], sM[wave], sN[wave], w, base, idx0, epoch, k0, k1, lt_mask);
    base += cnt;
    if (nDef > 0 && (nDef > kDefCap - kSuper || base >= r1)) {
      st_flags |= draw_deferred(sDef[wave], nDef, sZ[wave], sM[wave], sN[wave], sIdx[wave], w, nvec, r0, idx0, epoch, k0, k1,
                                lt_mask);
      nDef = 0;
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (st_flags) atomicOr(status, st_flags);
}

// ---- pass 2: PPpart = sum_i w_i x_i x_i'  on the fp64 matrix pipe    (Logit.hpp:294-301)
// A wave streams 64-row tiles; per group of 4 rows the lane holds x(row k, colmap(q,c)) and
// issues one v_mfma_f64_16x16x4_f64 per upper-triangle block pair (A = w x_qa, B = x_qb).
template <int NB>
struct Acc {
  v4d a[NB * (NB + 1) / 2];
};

template <int NB, bool EXACT>
__global__ __launch_bounds__(kBlock, 2) void k_xwx_mfma(const double* __restrict__ tX, const double* __restrict__ w,
                                                        int64_t N, int Pa, double* __restrict__ partial)
{
  constexpr int NBLK = NB * (NB + 1) / 2;
  __shared__ double red[2][NBLK * 4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = lane >> 4, c = lane & 15;
  const int64_t ntiles = (N + 63) / 64;
  const int64_t W = (int64_t)gridDim.x * 4;

  Acc<NB> acc;
#pragma unroll
  for (int b = 0; b < NBLK; ++b) acc.a[b] = v4d{0.0, 0.0, 0.0, 0.0};

  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntiles; tile += W) {
    double x[16][NB];
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int64_t row = tile * 64 + 4 * g + k;
      load_group<NB, EXACT>(x[g], tX, row, row < N, c, Pa);
    }
    const int64_t myrow = tile * 64 + 4 * c + k;
    const double omega = myrow < N ? w[myrow] : 0.0;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const double wg = __shfl(omega, (lane & 48) | g);
      double a[NB];
#pragma unroll
      for (int q = 0; q < NB; ++q) a[q] = wg * x[g][q];
      int blkid = 0;
#pragma unroll
      for (int qa = 0; qa < NB; ++qa)
#pragma unroll
        for (int qb = qa; qb < NB; ++qb) {
          acc.a[blkid] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[qa], x[g][qb], acc.a[blkid], 0, 0, 0);
          ++blkid;
        }
    }
  }

  // fixed-order in-block reduction: (w0 + w2) + (w1 + w3)
  if (wave >= 2) {
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave - 2][b * 4 + r][lane] = acc.a[b][r];
  }
  __syncthreads();
  if (wave < 2) {
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc.a[b][r] += red[wave][b * 4 + r][lane];
  }
  __syncthreads();
  if (wave == 1) {
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[0][b * 4 + r][lane] = acc.a[b][r];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        partial[(size_t)blockIdx.x * (NBLK * 4 * 64) + (b * 4 + r) * 64 + lane] = acc.a[b][r] + red[0][b * 4 + r][lane];
  }
}

// PP = sum over workgroups (fixed order) of the permuted MFMA blocks; un-permute,
// take the i <= j half of diagonal blocks, mirror: PP is exactly symmetric.
template <int NB>
__global__ __launch_bounds__(1024) void k_reduce_fused(const double* __restrict__ partial, int nparts,
                                                       double* __restrict__ PP, int Pa)
{
  constexpr int NBLK = NB * (NB + 1) / 2;
  constexpr int E = NBLK * 4 * 64;
  __shared__ double sm[16][64];
  const int e = blockIdx.x * 64 + (threadIdx.x & 63);
  const int s = threadIdx.x >> 6;            // 16 waves: wave s sums slabs s, s+16, ... (fixed order)
  double sum = 0.0;
  if (e < E)
    for (int b = s; b < nparts; b += 16) sum += partial[(size_t)b * E + e];
  sm[s][threadIdx.x & 63] = sum;
  __syncthreads();
  if (s == 0 && e < E) {
    const int l = threadIdx.x & 63;
    double tot = sm[0][l];
#pragma unroll
    for (int q = 1; q < 16; ++q) tot += sm[q][l];
    const int blkid = e / 256, reg = (e >> 6) & 3, ln = e & 63;
    int qa = 0, qb = 0, id = 0;
    for (int a = 0; a < NB; ++a)
      for (int b = a; b < NB; ++b) {
        if (id == blkid) { qa = a; qb = b; }
        ++id;
      }
    const int i = (ln >> 4) + 4 * reg, j = ln & 15;
    const int A = colmap<NB>(qa, i), B = colmap<NB>(qb, j);
    if ((qa != qb || i <= j) && A < Pa && B < Pa) {
      PP[A + (size_t)B * Pa] = tot;
      PP[B + (size_t)A * Pa] = tot;
    }
  }
}

// ---- pass 2 for P = 128 / 256 (NB = 8 / 16): the rank-N update is compute-bound here
// (N P^2 flops against 8 N P bytes: 10.4 ms of fp64 MFMA vs 3.2 ms of HBM per 25.6 GB shard at
// P = 256), and the NB(NB+1)/2 accumulator blocks (136 at NB = 16) no longer fit one wave.  A
// workgroup of NW waves (two per SIMD) shares 16-row tiles of X staged in LDS (row stride padded by
// 128 B: the four rows of an MFMA operand land in disjoint bank halves) and splits the upper
// triangle by block-row: wave w owns block-rows w and NB-1-w, i.e. exactly NB+1 blocks each, so it
// loads two A fragments per 4-row group and one B fragment per MFMA.  Columns keep their natural
// order here (block q = columns 16q..16q+15).
template <int NB, int NW, bool EXACT>
__global__ __launch_bounds__(NW * 64, 2) void k_xwx_mfma_big(const double* __restrict__ tX,
                                                             const double* __restrict__ w, int64_t N, int Pa,
                                                             double* __restrict__ partial)
{
  constexpr int P = 16 * NB;
  constexpr int LDT = P + 16;                 // padded row stride (doubles)
  constexpr int RT = 16;                      // rows per tile
  constexpr int NT = NW * 64;
  constexpr int VPT = RT * P / 2 / NT;        // 16-byte vectors per thread per tile
  static_assert(NB % NW == 0 && (NB / NW == 1 || NB / NW == 2), "block-rows are dealt w and NB-1-w per wave");
  extern __shared__ double lds[];
  double* tile = lds;                         // [2][RT][LDT]
  double* wt = lds + 2 * RT * LDT;            // [2][RT]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int k = lane >> 4, c = lane & 15;
  const int64_t ntiles = (N + RT - 1) / RT;

  v4d acc[NB + 1];
#pragma unroll
  for (int b = 0; b < NB + 1; ++b) acc[b] = v4d{0.0, 0.0, 0.0, 0.0};

  // NW == NB/2: wave w owns block-rows {w, NB-1-w}; NW == NB: pairs of waves would split a row --
  // not needed for the instantiations built (NB = 8 with 4 waves, NB = 16 with 8 waves)
  const int ra = wave, rb = NB - 1 - wave;    // the wave's two block-rows; ra has NB-ra blocks, rb has wave+1
  const int na = NB - ra;

  v2d stage[VPT];
  double wstage = 0.0;
  auto fetch = [&](int64_t tl) {
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
      const int e = t + v * NT;                                    // vector index in the tile
      const int r = e / (P / 2), cv = e % (P / 2);
      const int64_t row = tl * RT + r;
      if (EXACT) {
        stage[v] = row < N ? *reinterpret_cast<const v2d*>(tX + (size_t)row * P + 2 * cv) : v2d{0.0, 0.0};
      } else {      // fewer real columns: row stride Pa, 8-byte loads, columns >= Pa are zero
        const double* rp = tX + (size_t)(row < N ? row : 0) * (size_t)Pa;
        const int c0 = 2 * cv, c1 = 2 * cv + 1;
        const double v0 = rp[c0 < Pa ? c0 : Pa - 1], v1 = rp[c1 < Pa ? c1 : Pa - 1];
        stage[v] = v2d{(row < N && c0 < Pa) ? v0 : 0.0, (row < N && c1 < Pa) ? v1 : 0.0};
      }
    }
    if (t < RT) {
      const int64_t row = tl * RT + t;
      wstage = row < N ? w[row] : 0.0;
    }
  };
  auto deposit = [&](int buf) {
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
      const int e = t + v * NT;
      const int r = e / (P / 2), cv = e % (P / 2);
      *reinterpret_cast<v2d*>(tile + (buf * RT + r) * LDT + 2 * cv) = stage[v];
    }
    if (t < RT) wt[buf * RT + t] = wstage;
  };

  int64_t tl = blockIdx.x;
  if (tl < ntiles) {
    fetch(tl);
    deposit(0);
  }
  __syncthreads();
  int buf = 0;
  for (; tl < ntiles; tl += gridDim.x) {
    const int64_t nxt = tl + gridDim.x;
    if (nxt < ntiles) fetch(nxt);
    const double* T = tile + buf * RT * LDT;
#pragma unroll
    for (int g = 0; g < RT / 4; ++g) {
      const double* rowp = T + (4 * g + k) * LDT + c;
      const double wk = wt[buf * RT + 4 * g + k];
      const double a0 = rowp[16 * ra] * wk, a1 = rowp[16 * rb] * wk;
#pragma unroll
      for (int sidx = 0; sidx < NB + 1; ++sidx) {
        const bool first = sidx < na;
        const int qb = first ? ra + sidx : rb + (sidx - na);
        const double b = rowp[16 * qb];
        acc[sidx] = __builtin_amdgcn_mfma_f64_16x16x4f64(first ? a0 : a1, b, acc[sidx], 0, 0, 0);
      }
    }
    if (nxt < ntiles) deposit(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  // slab layout: [block id in row-major upper-triangle order][reg][lane]
#pragma unroll
  for (int sidx = 0; sidx < NB + 1; ++sidx) {
    const bool first = sidx < na;
    const int qa = first ? ra : rb;
    const int qb = first ? ra + sidx : rb + (sidx - na);
    const int blkid = qa * NB - qa * (qa - 1) / 2 + (qb - qa);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      partial[(size_t)blockIdx.x * (NB * (NB + 1) / 2 * 256) + (size_t)blkid * 256 + r * 64 + lane] = acc[sidx][r];
  }
}

// PP from the slabs of k_xwx_mfma_big (natural column order), fixed summation order.
template <int NB>
__global__ __launch_bounds__(1024) void k_reduce_big(const double* __restrict__ partial, int nparts,
                                                     double* __restrict__ PP, int Pa)
{
  constexpr int E = NB * (NB + 1) / 2 * 256;
  __shared__ double sm[16][64];
  const int e = blockIdx.x * 64 + (threadIdx.x & 63);
  const int s = threadIdx.x >> 6;
  double sum = 0.0;
  if (e < E)
    for (int b = s; b < nparts; b += 16) sum += partial[(size_t)b * E + e];
  sm[s][threadIdx.x & 63] = sum;
  __syncthreads();
  if (s == 0 && e < E) {
    const int l = threadIdx.x & 63;
    double tot = sm[0][l];
#pragma unroll
    for (int q = 1; q < 16; ++q) tot += sm[q][l];
    const int blkid = e / 256, reg = (e >> 6) & 3, ln = e & 63;
    int qa = 0, rem = blkid;
    while (rem >= NB - qa) {
      rem -= NB - qa;
      ++qa;
    }
    const int qb = qa + rem;
    const int i = (ln >> 4) + 4 * reg, j = ln & 15;
    const int A = 16 * qa + i, B = 16 * qb + j;
    if ((qa != qb || i <= j) && A < Pa && B < Pa) {
      PP[A + (size_t)B * Pa] = tot;
      PP[B + (size_t)A * Pa] = tot;
    }
  }
}

// ======================================================= generic (any P) kernels
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_psi_omega(const double* __restrict__ tX, const double* __restrict__ nvec,
                                                      const double* __restrict__ beta,
                                                      const double* __restrict__ off, double* __restrict__ w_store,
                                                      int64_t N, int P, uint64_t seed, uint32_t epoch,
                                                      uint64_t idx0, int* __restrict__ status)
{
  extern __shared__ double sbeta[];
  for (int j = threadIdx.x; j < P; j += kBlock) sbeta[j] = beta[j];
  __syncthreads();
  int st = 0;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < N; i += stride) {
    const double* xr = tX + (size_t)i * P;
    double s = 0.0;
    for (int j = 0; j < P; ++j) s += xr[j] * sbeta[j];
    if (off) s -= off[i];
    w_store[i] = weight_of<MODE>(s, nvec[i], seed, idx0 + (uint64_t)i, epoch, st);
  }
  if (st) atomicOr(status, st);
}

// partial[chunk][tile][64x64] = sum over the chunk's rows of w_i x_i[A-tile] x_i[B-tile]'
constexpr int kRows = 32;
__global__ __launch_bounds__(kBlock) void k_xwx_tiles(const double* __restrict__ tX, const double* __restrict__ w,
                                                      int64_t N, int P, int T, int64_t rows_per_chunk,
                                                      double* __restrict__ partial)
{
  __shared__ double xa[kRows][64];
  __shared__ double xb[kRows][64];
  int ta = 0, tb = 0;
  {
    int id = 0;
    for (int a = 0; a < T; ++a)
      for (int b = a; b < T; ++b) {
        if (id == (int)blockIdx.y) { ta = a; tb = b; }
        ++id;
      }
  }
  const int a = threadIdx.x & 63, bq = threadIdx.x >> 6;
  double acc[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.0;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_chunk;
  const int64_t r1 = (r0 + rows_per_chunk < N) ? r0 + rows_per_chunk : N;
  for (int64_t rb = r0; rb < r1; rb += kRows) {
    for (int e = threadIdx.x; e < kRows * 64; e += kBlock) {
      const int r = e >> 6, cc = e & 63;
      const int64_t row = rb + r;
      const bool rok = row < r1;
      const int ca = 64 * ta + cc, cb = 64 * tb + cc;
      const double wr = rok ? w[row] : 0.0;
      xa[r][cc] = (rok && ca < P) ? tX[(size_t)row * P + ca] * wr : 0.0;
      xb[r][cc] = (rok && cb < P) ? tX[(size_t)row * P + cb] : 0.0;
    }
    __syncthreads();
#pragma unroll 4
    for (int r = 0; r < kRows; ++r) {
      const double va = xa[r][a];
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] += va * xb[r][bq + 4 * q];
    }
    __syncthreads();
  }
  double* out = partial + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 4096;
#pragma unroll
  for (int q = 0; q < 16; ++q) out[a + 64 * (bq + 4 * q)] = acc[q];
}

__global__ __launch_bounds__(kBlock) void k_reduce_tiles(const double* __restrict__ partial, int nchunks, int P,
                                                         int T, double* __restrict__ PP)
{
  const int ntile = T * (T + 1) / 2;
  const int64_t total = (int64_t)ntile * 4096;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlock) {
    const int tile = (int)(e / 4096), w = (int)(e % 4096);
    int ta = 0, tb = 0, id = 0;
    for (int a = 0; a < T; ++a)
      for (int b = a; b < T; ++b) {
        if (id == tile) { ta = a; tb = b; }
        ++id;
      }
    const int A = 64 * ta + (w & 63), B = 64 * tb + (w >> 6);
    if (A >= P || B >= P) continue;
    if (ta == tb && A > B) continue;   // take the upper half of diagonal tiles, mirror below
    double s = 0.0;
    for (int ch = 0; ch < nchunks; ++ch) s += partial[((size_t)ch * ntile + tile) * 4096 + w];
    PP[A + (size_t)B * P] = s;
    PP[B + (size_t)A * P] = s;
  }
}

// out_partial[blk][j] = sum over the block's rows of wgt_i x_ij  (deterministic)
__global__ __launch_bounds__(kBlock) void k_colsum(const double* __restrict__ tX, const double* __restrict__ y,
                                                   const double* __restrict__ nvec, const double* __restrict__ w,
                                                   const double* __restrict__ cvec, int64_t N, int P,
                                                   int64_t rows_per_block, double* __restrict__ ws)
{
  extern __shared__ double sm[];   // [rowlanes][P]
  const int cols = P < kBlock ? P : kBlock;
  const int rowlanes = kBlock / cols;
  const int rl = threadIdx.x / cols, cl = threadIdx.x % cols;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < N) ? r0 + rows_per_block : N;
  for (int j0 = 0; j0 < P; j0 += cols) {
    const int j = j0 + cl;
    double s = 0.0;
    if (rl < rowlanes && j < P)
      for (int64_t i = r0 + rl; i < r1; i += rowlanes) {
        const double wgt = w ? w[i] * (cvec ? cvec[i] : 1.0) : nvec[i] * (y[i] - 0.5);
        s += tX[(size_t)i * P + j] * wgt;
      }
    if (rl < rowlanes && j < P) sm[rl * P + j] = s;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < P; j += kBlock) {
    double s = 0.0;
    for (int q = 0; q < rowlanes; ++q) s += sm[q * P + j];
    ws[(size_t)blockIdx.x * P + j] = s;
  }
}

__global__ __launch_bounds__(kBlock) void k_colsum_reduce(const double* __restrict__ ws, int nblk, int P,
                                                          double* __restrict__ out)
{
  for (int j = blockIdx.x * kBlock + threadIdx.x; j < P; j += gridDim.x * kBlock) {
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += ws[(size_t)b * P + j];
    out[j] = s;
  }
}

__global__ __launch_bounds__(kBlock) void k_xbeta(const double* __restrict__ tX, const double* __restrict__ beta,
                                                  int64_t N, int P, double* __restrict__ out)
{
  extern __shared__ double sbeta[];
  for (int j = threadIdx.x; j < P; j += kBlock) sbeta[j] = beta[j];
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < N; i += stride) {
    const double* xr = tX + (size_t)i * P;
    double s = 0.0;
    for (int j = 0; j < P; ++j) s += xr[j] * sbeta[j];
    out[i] = s;
  }
}

// MultLogit.hpp:293-299: A = rowSums(exp(XB_no_j)); c_j = log A
__global__ __launch_bounds__(kBlock) void k_mlogit_offset(const double* __restrict__ XB, int64_t N, int J, int j,
                                                          double* __restrict__ c_out)
{
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < N; i += stride) {
    double A = 0.0;
    for (int k = 0; k < J; ++k)
      if (k != j) A += exp(XB[(size_t)k * N + i]);
    c_out[i] = log(A);
  }
}

// ================================================================ P x P stage
#define M_(M, i, j) ((M)[(size_t)(i) + (size_t)(j) * (size_t)P])

// In-place A = U'U (upper triangle holds U), right-looking, whole workgroup.
__device__ bool wg_chol_upper(double* A, int P, int* bad)
{
  const int t = threadIdx.x;
  for (int k = 0; k < P; ++k) {
    const double akk = M_(A, k, k);
    if (!(akk > 0.0)) {
      if (t == 0) *bad = 1;
      return false;
    }
    const double d = sqrt(akk);
    __syncthreads();
    if (t == 0) M_(A, k, k) = d;
    for (int j = k + 1 + t; j < P; j += (int)blockDim.x) M_(A, k, j) = M_(A, k, j) / d;
    __syncthreads();
    const int m = P - k - 1;
    for (int e = t; e < m * m; e += (int)blockDim.x) {
      const int i = k + 1 + e % m, j = k + 1 + e / m;
      if (i <= j) M_(A, i, j) -= M_(A, k, i) * M_(A, k, j);
    }
    __syncthreads();
  }
  return true;
}

// In-place S = L L' (lower triangle holds L; strict upper zeroed).
__device__ bool wg_chol_lower(double* S, int P, int* bad)
{
  const int t = threadIdx.x;
  for (int k = 0; k < P; ++k) {
    const double akk = M_(S, k, k);
    if (!(akk > 0.0)) {
      if (t == 0) *bad = 1;
      return false;
    }
    const double d = sqrt(akk);
    __syncthreads();
    if (t == 0) M_(S, k, k) = d;
    for (int i = k + 1 + t; i < P; i += (int)blockDim.x) M_(S, i, k) = M_(S, i, k) / d;
    __syncthreads();
    const int m = P - k - 1;
    for (int e = t; e < m * m; e += (int)blockDim.x) {
      const int i = k + 1 + e % m, j = k + 1 + e / m;
      if (i >= j) M_(S, i, j) -= M_(S, i, k) * M_(S, j, k);
    }
    __syncthreads();
  }
  for (int e = t; e < P * P; e += (int)blockDim.x) {
    const int i = e % P, j = e / P;
    if (i < j) M_(S, i, j) = 0.0;
  }
  __syncthreads();
  return true;
}

// B (P x nrhs, leading dim ldb) <- U'^{-1} B : forward substitution, all columns at once
__device__ void wg_solve_Ut(const double* U, double* B, int P, int nrhs, int ldb)
{
  const int t = threadIdx.x;
  for (int i = 0; i < P; ++i) {
    const double d = M_(U, i, i);
    for (int c = t; c < nrhs; c += (int)blockDim.x) B[i + (size_t)c * ldb] /= d;
    __syncthreads();
    const int m = P - i - 1;
    for (int e = t; e < m * nrhs; e += (int)blockDim.x) {
      const int j = i + 1 + e % m, c = e / m;
      B[j + (size_t)c * ldb] -= M_(U, i, j) * B[i + (size_t)c * ldb];
    }
    __syncthreads();
  }
}
// B <- U^{-1} B : backward substitution
__device__ void wg_solve_U(const double* U, double* B, int P, int nrhs, int ldb)
{
  const int t = threadIdx.x;
  for (int i = P - 1; i >= 0; --i) {
    const double d = M_(U, i, i);
    for (int c = t; c < nrhs; c += (int)blockDim.x) B[i + (size_t)c * ldb] /= d;
    __syncthreads();
    for (int e = t; e < i * nrhs; e += (int)blockDim.x) {
      const int j = e % i, c = e / i;
      B[j + (size_t)c * ldb] -= M_(U, j, i) * B[i + (size_t)c * ldb];
    }
    __syncthreads();
  }
}
// b <- L^{-1} b : forward substitution, single rhs
__device__ void wg_solve_L(const double* L, double* b, int P)
{
  const int t = threadIdx.x;
  for (int i = 0; i < P; ++i) {
    if (t == 0) b[i] /= M_(L, i, i);
    __syncthreads();
    for (int j = i + 1 + t; j < P; j += (int)blockDim.x) b[j] -= M_(L, j, i) * b[i];
    __syncthreads();
  }
}

__device__ __forceinline__ double wave_max(double v)
{
  for (int m = 32; m > 0; m >>= 1) v = fmax(v, __shfl_xor(v, m));
  return v;
}
__device__ __forceinline__ double wave_min(double v)
{
  for (int m = 32; m > 0; m >>= 1) v = fmin(v, __shfl_xor(v, m));
  return v;
}

__device__ void constrained_wide_prepare(const blk::BetaArgs& a, const double* __restrict__ Lg, double* Rg);

__global__ __launch_bounds__(1024) void k_beta(blk::BetaArgs a, int mode)
{
  extern __shared__ double lds[];          // constrained mode: L (P*P) when it fits, then beta, z (P each), perm
  // a Cholesky factorisation failed earlier in this chain (ST_NOT_PD is sticky until the host collects the status word):
  // the chain is dead, and for 64 < P <= 256 k_beta has not prepared the workspace k_beta_sweeps reads
  if (*a.status & ST_NOT_PD) return;
  const int P = a.P, t = threadIdx.x;
  double* A = a.work;                      // PP, then U
  double* S = a.work + (size_t)P * P;      // PP^{-1}
  double* mP = a.work + 2 * (size_t)P * P; // posterior mean
  double* zz = mP + P;
  __shared__ int bad;
  if (t == 0) bad = 0;
  for (int e = t; e < P * P; e += (int)blockDim.x) A[e] = a.PPsum[e] + a.P0[e];   // PP = P0 + X'OmX
  __syncthreads();
  if (!wg_chol_upper(A, P, &bad)) {
    __syncthreads();
    if (t == 0) atomicOr(a.status, ST_NOT_PD);
    return;
  }

  if (mode == blk::B_SOLVE || mode == blk::B_MVN) {
    for (int j = t; j < P; j += (int)blockDim.x) mP[j] = a.bP[j];
    if (mode == blk::B_MVN && t < P) {
      // eps_i = r.norm(0,1), i = 0..P-1 in stream order: normal i is exactly block i
      Stream r;
      r.init(a.seed, 0, DOM_BETA, a.epoch);
      for (int i = t; i < P; i += (int)blockDim.x) {
        r.blk = (uint32_t)i;
        r.has = false;
        zz[i] = r.norm(0.0, 1.0);
      }
    }
    __syncthreads();
    wg_solve_Ut(A, mP, P, 1, P);
    wg_solve_U(A, mP, P, 1, P);
    if (mode == blk::B_MVN) {
      wg_solve_U(A, zz, P, 1, P);
      for (int j = t; j < P; j += (int)blockDim.x) a.beta_out[j] = zz[j] + mP[j];
    } else {
      for (int j = t; j < P; j += (int)blockDim.x) a.beta_out[j] = mP[j];
    }
    return;
  }

  // S = PP^{-1}: two triangular solves on the identity
  for (int e = t; e < P * P; e += (int)blockDim.x) S[e] = (e % P == e / P) ? 1.0 : 0.0;
  __syncthreads();
  wg_solve_Ut(A, S, P, P, P);
  wg_solve_U(A, S, P, P, P);

  if (mode == blk::B_FROM_LIK) {
    // mean = V b ; lower = chol(V,'L') ; beta = mean + lower eps   (Normal.hpp:98-131)
    for (int i = t; i < P; i += (int)blockDim.x) {
      double s = 0.0;
      for (int k2 = 0; k2 < P; ++k2) s += M_(S, i, k2) * a.bP[k2];
      mP[i] = s;
    }
    if (t < P) {
      Stream r;
      r.init(a.seed, 0, DOM_BETA, a.epoch);
      for (int i = t; i < P; i += (int)blockDim.x) {
        r.blk = (uint32_t)i;
        r.has = false;
        zz[i] = r.norm(0.0, 1.0);
      }
    }
    __syncthreads();
    if (!wg_chol_lower(S, P, &bad)) {
      __syncthreads();
      if (t == 0) atomicOr(a.status, ST_NOT_PD);
      return;
    }
    for (int i = t; i < P; i += (int)blockDim.x) {
      double le = 0.0;
      for (int k2 = 0; k2 <= i; ++k2) le += M_(S, i, k2) * zz[k2];
      a.beta_out[i] = le + mP[i];
    }
    return;
  }

  // ---- B_CONSTRAINED: Logit.hpp:322-400 ----
  for (int j = t; j < P; j += (int)blockDim.x) mP[j] = a.bP[j];
  __syncthreads();
  wg_solve_Ut(A, mP, P, 1, P);
  wg_solve_U(A, mP, P, 1, P);
  if (!wg_chol_lower(S, P, &bad)) {     // L = chol(S,'L'), in place
    __syncthreads();
    if (t == 0) atomicOr(a.status, ST_NOT_PD);
    return;
  }
  if (P <= 256) {
    // z = L^{-1}(beta_prev - mP), then the coordinate sweeps with their random input pre-generated
    for (int j = t; j < P; j += (int)blockDim.x) zz[j] = a.beta_prev[j] - mP[j];
    __syncthreads();
    wg_solve_L(S, zz, P);
    constrained_wide_prepare(a, S, A);      // U in A is dead: A takes 1/L.  k_beta_sweeps follows.
    return;
  }
  // LDS layout: beta, z (P doubles each), perm (P ints), then L (P*P) when it fits.  The vectors
  // are exchanged between lanes of the serial wave, which LDS orders and global memory does not.
  const bool l_in_lds = (size_t)P * P * 8 <= 128 * 1024;
  double* sbeta = lds;
  double* sz = sbeta + P;
  int* perm = reinterpret_cast<int*>(sz + P);
  double* Lm = l_in_lds ? sz + P + (P + 1) / 2 + 1 : S;
  if (l_in_lds)
    for (int e = t; e < P * P; e += (int)blockDim.x) Lm[e] = S[e];
  for (int j = t; j < P; j += (int)blockDim.x) {
    zz[j] = a.beta_prev[j] - mP[j];     // z = beta_prev - mP
    sbeta[j] = a.beta_prev[j];
    perm[j] = j;
  }
  __syncthreads();
  wg_solve_L(S, zz, P);                 // z = L^{-1} z
  for (int j = t; j < P; j += (int)blockDim.x) sz[j] = zz[j];
  __syncthreads();

  if (t < 64) {                          // one wavefront runs the serial coordinate sweeps
    const int lane = t;
    Stream r;
    r.init(a.seed, 0, DOM_BETA, a.epoch);
    const double inf = __builtin_huge_val();
    for (int k = 0; k < P; ++k) {
      for (int i = 0; i < P - 1; ++i) {          // random sweep order, :375-377
        const int j = (int)(unsigned)r.flat((double)i, (double)P);
        if (lane == 0) {
          const int tmp = perm[i];
          perm[i] = perm[j];
          perm[j] = tmp;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      for (int i = 0; i < P; ++i) {              // :380-398
        const int c = perm[i];
        const double z1 = sz[c];
        double lo = -inf, hi = inf;
        for (int j = c + lane; j < P - 1; j += 64) {
          const double l1 = M_(Lm, j, c);
          const double c1 = z1 - sbeta[j] / l1;
          if (l1 > 0.0 && c1 > lo) lo = c1;
          else if (l1 < 0.0 && c1 < hi) hi = c1;
        }
        const double cmin = wave_max(lo), cmax = wave_min(hi);
        const double z2 = tnorm(r, cmin, cmax);
        const double dz = z2 - z1;
        for (int j = c + lane; j < P; j += 64) sbeta[j] += M_(Lm, j, c) * dz;
        if (lane == 0) sz[c] = z2;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  __syncthreads();
  for (int j = t; j < P; j += (int)blockDim.x) a.beta_out[j] = sbeta[j];
}

// ============================================= P x P stage, P <= 64: everything on chip
// k_beta64 does the same four jobs as k_beta with the three P x P matrices in LDS (leading
// dimension P+1: conflict-free rows and columns) and, for the constrained draw, a serial
// coordinate loop stripped to its dependent chain:
//   * every random input of the draw is generated BEFORE the loop, in parallel (possible
//     because a tnorm call owns exactly nine uniforms whatever its bounds): the P-1 swap
//     targets of each random scan and, per tnorm call, the four proposal pairs with their
//     logs and Box-Muller normal already taken;
//   * lane j of a wavefront owns row j (beta_j in a register, z in LDS); the moves are taken in speculative
//     groups on all four wavefronts (see the kernel), and a move that needs its bounds gets them as a 64-lane
//     max/min by DPP (no LDS round trip), with 1/L precomputed elementwise.
#define L_(M, i, j) ((M)[(i) + (j) * ld])

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int l)
{
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l),
                          __builtin_amdgcn_readlane(__double2loint(v), l));
}
// raw v_max_f64 / v_min_f64 (operands are never NaN here; skips the canonicalising pre-pass
// that fmax()/fmin() lower to)
__device__ __forceinline__ double vmax64(double a, double b)
{
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double vmin64(double a, double b)
{
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64_rm(double v)
{
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
// max of mx and min of mn over the 64 lanes, returned wave-uniform: butterfly inside each row
// of 16 lanes (4 DPP levels), then row_bcast15 / row_bcast31 fold the four rows into row 3.
// (A float-key fast path -- one VOP2-DPP op per level, winner located by ballot -- was tried and
// measured slower: its VALU->SGPR->branch crossings cost more than the 64-bit moves they save.)
__device__ __forceinline__ void wave_maxmin(double& mx, double& mn)
{
  mx = vmax64(mx, dppmov_f64<0xB1>(mx));  mn = vmin64(mn, dppmov_f64<0xB1>(mn));    // quad_perm [1,0,3,2]
  mx = vmax64(mx, dppmov_f64<0x4E>(mx));  mn = vmin64(mn, dppmov_f64<0x4E>(mn));    // quad_perm [2,3,0,1]
  mx = vmax64(mx, dppmov_f64<0x141>(mx)); mn = vmin64(mn, dppmov_f64<0x141>(mn));   // row_half_mirror
  mx = vmax64(mx, dppmov_f64<0x140>(mx)); mn = vmin64(mn, dppmov_f64<0x140>(mn));   // row_mirror
  mx = vmax64(mx, dpp_f64_rm<0x142, 0xa>(mx)); mn = vmin64(mn, dpp_f64_rm<0x142, 0xa>(mn));   // row_bcast15 -> rows 1,3
  mx = vmax64(mx, dpp_f64_rm<0x143, 0xc>(mx)); mn = vmin64(mn, dpp_f64_rm<0x143, 0xc>(mn));   // row_bcast31 -> rows 2,3
  mx = readlane_f64(mx, 63);
  mn = readlane_f64(mn, 63);
}

// tnorm from the pre-generated record of the call, attempts evaluated by lanes 0..3 at once.
// Lane g < 5 holds group g of the record: g < 4: (ua, log ua, log ub, Box-Muller normal) of
// attempt g; g = 4: (fallback uniform, -, -, -).  lo/hi are wave-uniform.  The first accepted
// attempt in attempt order wins, else the exact inverse-CDF draw: same decisions and values as
// bl::tnorm on the same nine uniforms.
__device__ __forceinline__ double tnorm_lanes(double r0, double r1, double r2, double r3, int lane, double lo, double hi)
{
  if (!(hi - lo > 0.0)) return lo;
  double x;
  bool ok;
  bool flip = false;
  double a = lo, b = hi;
  if (lo <= 0.0 && hi >= 0.0) {
    if (hi - lo > 2.5066282746310002) {
      x = r3;
      ok = x >= lo && x <= hi;
    } else {
      x = lo + (hi - lo) * r0;
      ok = r2 <= -0.5 * x * x;
    }
  } else {
    flip = hi < 0.0;
    a = flip ? -hi : lo;
    b = flip ? -lo : hi;
    // this branch is on the dependent chain of every coordinate move: the short sqrt / divide forms
    // (<= 1 ulp from the IEEE ones)
    const double s4 = a * a + 4.0;
    const double alpha = 0.5 * (a + (s4 < 1e300 ? bl_sqrt(s4) : sqrt(s4)));
    const double ialpha = bl_div(1.0, alpha);
    if (b - a > ialpha) {
      x = a - r1 * ialpha;
      const double d = x - alpha;
      ok = x <= b && r2 <= -0.5 * d * d;
    } else {
      x = a + (b - a) * r0;
      ok = r2 <= 0.5 * (a * a - x * x);
    }
  }
  const uint64_t m = __ballot(ok && lane < 4);
  if (m != 0) {
    const int first = __builtin_ctzll(m);
    x = readlane_f64(x, first);
    return flip ? -x : x;
  }
  const double u8 = readlane_f64(r0, 4);
  if (lo <= 0.0 && hi >= 0.0) {
    const double pl = isinf(lo) ? 0.0 : 0.5 * erfc(-lo * kSqrtHalfR);
    const double ph = isinf(hi) ? 1.0 : 0.5 * erfc(-hi * kSqrtHalfR);
    double xi = qnorm(pl + u8 * (ph - pl));
    xi = xi < lo ? lo : xi;
    xi = xi > hi ? hi : xi;
    return xi;
  }
  x = tnorm_inv_right(a, b, u8);
  return flip ? -x : x;
}

// uniform number `ui` of stream (seed, 0, DOM_BETA, epoch)
__device__ __forceinline__ double beta_stream_unif(uint64_t seed, uint32_t epoch, uint32_t ui)
{
  const U4 o = philox4x32_10(0u, ctr1_of(0, DOM_BETA), epoch, ui >> 1, (uint32_t)seed, (uint32_t)(seed >> 32));
  return (ui & 1u) ? u52(o.z, o.w) : u52(o.x, o.y);
}

// whole-workgroup dense helpers on LDS matrices with leading dimension ld
__device__ bool lds_chol_upper(double* A, int P, int ld, int* bad)
{
  const int t = threadIdx.x;
  for (int k = 0; k < P; ++k) {
    const double akk = L_(A, k, k);
    if (!(akk > 0.0)) {
      if (t == 0) *bad = 1;
      return false;
    }
    const double d = sqrt(akk);
    __syncthreads();
    if (t == 0) L_(A, k, k) = d;
    for (int j = k + 1 + t; j < P; j += kBlock) L_(A, k, j) = L_(A, k, j) / d;
    __syncthreads();
    const int m = P - k - 1;
    for (int e = t; e < m * m; e += kBlock) {
      const int i = k + 1 + e % m, j = k + 1 + e / m;
      if (i <= j) L_(A, i, j) -= L_(A, k, i) * L_(A, k, j);
    }
    __syncthreads();
  }
  return true;
}
__device__ bool lds_chol_lower(double* S, int P, int ld, int* bad)
{
  const int t = threadIdx.x;
  for (int k = 0; k < P; ++k) {
    const double akk = L_(S, k, k);
    if (!(akk > 0.0)) {
      if (t == 0) *bad = 1;
      return false;
    }
    const double d = sqrt(akk);
    __syncthreads();
    if (t == 0) L_(S, k, k) = d;
    for (int i = k + 1 + t; i < P; i += kBlock) L_(S, i, k) = L_(S, i, k) / d;
    __syncthreads();
    const int m = P - k - 1;
    for (int e = t; e < m * m; e += kBlock) {
      const int i = k + 1 + e % m, j = k + 1 + e / m;
      if (i >= j) L_(S, i, j) -= L_(S, i, k) * L_(S, j, k);
    }
    __syncthreads();
  }
  for (int e = t; e < P * P; e += kBlock) {
    const int i = e % P, j = e / P;
    if (i < j) L_(S, i, j) = 0.0;
  }
  __syncthreads();
  return true;
}
__device__ void lds_solve_Ut(const double* U, double* B, int P, int ld, int nrhs, int ldb)
{
  const int t = threadIdx.x;
  for (int i = 0; i < P; ++i) {
    const double d = L_(U, i, i);
    for (int c = t; c < nrhs; c += kBlock) B[i + c * ldb] /= d;
    __syncthreads();
    const int m = P - i - 1;
    for (int e = t; e < m * nrhs; e += kBlock) {
      const int j = i + 1 + e % m, c = e / m;
      B[j + c * ldb] -= L_(U, i, j) * B[i + c * ldb];
    }
    __syncthreads();
  }
}
__device__ void lds_solve_U(const double* U, double* B, int P, int ld, int nrhs, int ldb)
{
  const int t = threadIdx.x;
  for (int i = P - 1; i >= 0; --i) {
    const double d = L_(U, i, i);
    for (int c = t; c < nrhs; c += kBlock) B[i + c * ldb] /= d;
    __syncthreads();
    for (int e = t; e < i * nrhs; e += kBlock) {
      const int j = e % i, c = e / i;
      B[j + c * ldb] -= L_(U, j, i) * B[i + c * ldb];
    }
    __syncthreads();
  }
}
__device__ void lds_solve_L(const double* Lm, double* b, int P, int ld)
{
  const int t = threadIdx.x;
  for (int i = 0; i < P; ++i) {
    if (t == 0) b[i] /= L_(Lm, i, i);
    __syncthreads();
    for (int j = i + 1 + t; j < P; j += kBlock) b[j] -= L_(Lm, j, i) * b[i];
    __syncthreads();
  }
}

constexpr int kRec = 20;   // doubles per pre-generated tnorm record: 4 attempts x (ua, log ua, log ub, normal) + (u8,0,0,0)

// ---- single-wavefront dense kernels on LDS matrices (P <= 64, lane = column or row) ----
// One wave needs no s_barrier: LDS operations of a wave execute in program order, so a
// wave-level scheduling fence between a phase's writes and the next phase's reads is enough.
// The workgroup versions above pay two or three barriers per column (~1.4 us per column
// measured); these leave the other three waves free to generate the draw's random input, build the scan
// tables and solve for mP at the same time.
#define WAVE_SYNC()                                        \
  do {                                                     \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                       \
  } while (0)

__device__ __forceinline__ double bcast_f64(double v, int l)
{
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l),
                          __builtin_amdgcn_readlane(__double2loint(v), l));
}

// A = U'U in place (upper triangle holds U); lane j owns column j.  Same operation order per
// element as the reference's LAPACK-style column Cholesky (subtractions in ascending k).  The
// pivot row is passed between lanes by readlane, so the trailing update touches LDS only for the
// lane's own column (independent addresses: the loads of several rows are in flight together).
__device__ bool w_chol_upper(double* A, int P, int ld, int lane)
{
  for (int k = 0; k < P; ++k) {
    const double akk = L_(A, k, k);
    if (!(akk > 0.0)) return false;
    const double d = sqrt(akk);
    double ukj = 0.0;
    if (lane > k && lane < P) {
      ukj = L_(A, k, lane) / d;
      L_(A, k, lane) = ukj;
    }
    if (lane == k) L_(A, k, k) = d;
    double* col = A + lane * ld;
    int i = k + 1;
    for (; i + 3 < P; i += 4) {
      const double u0 = bcast_f64(ukj, i), u1 = bcast_f64(ukj, i + 1), u2 = bcast_f64(ukj, i + 2), u3 = bcast_f64(ukj, i + 3);
      if (lane < P) {
        const double a0 = col[i], a1 = col[i + 1], a2 = col[i + 2], a3 = col[i + 3];
        if (lane >= i) col[i] = a0 - u0 * ukj;
        if (lane >= i + 1) col[i + 1] = a1 - u1 * ukj;
        if (lane >= i + 2) col[i + 2] = a2 - u2 * ukj;
        if (lane >= i + 3) col[i + 3] = a3 - u3 * ukj;
      }
    }
    for (; i < P; ++i) {
      const double u0 = bcast_f64(ukj, i);
      if (lane >= i && lane < P) col[i] -= u0 * ukj;
    }
    WAVE_SYNC();
  }
  return true;
}

// S = L L' in place (lower triangle holds L, strict upper zeroed); lane i owns row i.
__device__ bool w_chol_lower(double* S, int P, int ld, int lane)
{
  for (int k = 0; k < P; ++k) {
    const double akk = L_(S, k, k);
    if (!(akk > 0.0)) return false;
    const double d = sqrt(akk);
    double lik = 0.0;
    if (lane > k && lane < P) {
      lik = L_(S, lane, k) / d;
      L_(S, lane, k) = lik;
    }
    if (lane == k) L_(S, k, k) = d;
    int j = k + 1;
    for (; j + 3 < P; j += 4) {
      const double l0 = bcast_f64(lik, j), l1 = bcast_f64(lik, j + 1), l2 = bcast_f64(lik, j + 2), l3 = bcast_f64(lik, j + 3);
      if (lane < P) {
        const double a0 = L_(S, lane, j), a1 = L_(S, lane, j + 1), a2 = L_(S, lane, j + 2), a3 = L_(S, lane, j + 3);
        if (lane >= j) L_(S, lane, j) = a0 - lik * l0;
        if (lane >= j + 1) L_(S, lane, j + 1) = a1 - lik * l1;
        if (lane >= j + 2) L_(S, lane, j + 2) = a2 - lik * l2;
        if (lane >= j + 3) L_(S, lane, j + 3) = a3 - lik * l3;
      }
    }
    for (; j < P; ++j) {
      const double l0 = bcast_f64(lik, j);
      if (lane >= j && lane < P) L_(S, lane, j) -= lik * l0;
    }
    WAVE_SYNC();
  }
  for (int j = 1; j < P; ++j)
    if (lane < j && lane < P) L_(S, lane, j) = 0.0;
  WAVE_SYNC();
  return true;
}

// Register-resident Cholesky on one wavefront (P <= 64): lane i keeps row i of the (upper) triangle in 64
// registers, so a step touches LDS only to pass the pivot row around (one 64-double buffer), not to update
// the trailing matrix.  Same arithmetic as w_chol_upper / w_chol_lower -- u_kj = a_kj / sqrt(a_kk), then
// a_ij -= u_ki u_kj, k ascending (a product of the same two numbers either way round) -- so the factor is
// bit-identical; LOWER = false: A = U'U, reads and writes the upper triangle of M; LOWER = true: M = L L', reads
// the lower triangle of M (row i of the transposed problem is column i of the lower triangle), writes L = U'
// into it and zeroes the strict upper triangle.  buf: 64 doubles of LDS.  Measured in k_beta64: 65 us each
// against 130 for w_chol_upper / w_chol_lower.
template <bool LOWER>
__device__ bool w_chol_reg(double* M, int P, int ld, int lane, double* buf)
{
  double r[64];
#pragma unroll
  for (int j = 0; j < 64; ++j) {
    const bool in = lane < P && j < P && j >= lane;
    r[j] = in ? (LOWER ? L_(M, j, lane) : L_(M, lane, j)) : 0.0;
  }
  for (int k = 0; k < P; ++k) {
    if (lane == k) {
#pragma unroll
      for (int j = 0; j < 64; ++j) buf[j] = r[j];
    }
    WAVE_SYNC();
    const double akk = buf[k];
    if (!(akk > 0.0)) return false;
    const double d = sqrt(akk);
    const double a_kl = buf[lane];
    WAVE_SYNC();
    double u = 0.0;
    if (lane > k && lane < P) u = a_kl / d;
    if (lane >= k && lane < P) {
      const double v = lane == k ? d : u;
      if (LOWER) L_(M, lane, k) = v;
      else L_(M, k, lane) = v;
    }
    buf[lane] = u;                                 // u_kj for j > k, 0 for j <= k and outside the matrix
    WAVE_SYNC();
#pragma unroll
    for (int j = 0; j < 64; ++j) r[j] = fma(-u, buf[j], r[j]);     // rows i <= k have u = 0: unchanged
    WAVE_SYNC();
  }
  if (LOWER) {
    for (int j = 1; j < P; ++j)
      if (lane < j && lane < P) L_(M, lane, j) = 0.0;
    WAVE_SYNC();
  }
  return true;
}

// S <- PP^{-1} given U (PP = U'U): S starts as I; lane c solves U'y = e_c then U x = y on its own
// column of S (dot-product form, ascending k as the reference's trsm), four products in flight.
__device__ void w_inverse_from_U(const double* U, double* S, int P, int ld, int lane)
{
  const int c = lane < P ? lane : 0;
  double* col = S + c * ld;
  for (int i = 0; i < P; ++i) {                            // forward: U' y = e_c
    const double* ui = U + i * ld;                         // column i of U: U[k][i], k < i
    double acc = col[i];
    int k = 0;
    for (; k + 3 < i; k += 4) {
      const double p0 = ui[k] * col[k], p1 = ui[k + 1] * col[k + 1], p2 = ui[k + 2] * col[k + 2], p3 = ui[k + 3] * col[k + 3];
      acc = (((acc - p0) - p1) - p2) - p3;
    }
    for (; k < i; ++k) acc -= ui[k] * col[k];
    const double y = acc / ui[i];
    if (lane < P) col[i] = y;
  }
  for (int i = P - 1; i >= 0; --i) {                       // backward: U x = y
    double acc = col[i];
    int k = i + 1;
    for (; k + 3 < P; k += 4) {
      const double p0 = L_(U, i, k) * col[k], p1 = L_(U, i, k + 1) * col[k + 1], p2 = L_(U, i, k + 2) * col[k + 2],
                   p3 = L_(U, i, k + 3) * col[k + 3];
      acc = (((acc - p0) - p1) - p2) - p3;
    }
    for (; k < P; ++k) acc -= L_(U, i, k) * col[k];
    const double x = acc / L_(U, i, i);
    if (lane < P) col[i] = x;
  }
  WAVE_SYNC();
}

// b <- U'^{-1} b, b_j in lane j's register
__device__ double w_solve_Ut_vec(const double* U, double b, int P, int ld, int lane)
{
  for (int i = 0; i < P; ++i) {
    const double bi = bcast_f64(b, i) / L_(U, i, i);
    if (lane == i) b = bi;
    if (lane > i && lane < P) b -= L_(U, i, lane) * bi;
  }
  return b;
}
// b <- U^{-1} b
__device__ double w_solve_U_vec(const double* U, double b, int P, int ld, int lane)
{
  for (int i = P - 1; i >= 0; --i) {
    const double bi = bcast_f64(b, i) / L_(U, i, i);
    if (lane == i) b = bi;
    if (lane < i) b -= L_(U, lane, i) * bi;
  }
  return b;
}
// b <- L^{-1} b (L lower)
__device__ double w_solve_L_vec(const double* Lm, double b, int P, int ld, int lane)
{
  for (int i = 0; i < P; ++i) {
    const double bi = bcast_f64(b, i) / L_(Lm, i, i);
    if (lane == i) b = bi;
    if (lane > i && lane < P) b -= L_(Lm, lane, i) * bi;
  }
  return b;
}

// One speculative group of the constrained sweeps (see k_beta64): beta after each of the group's moves on the
// fast path (bs_out: after all of them) and the bound test of moves 2W and 2W+1 of the group.  Returns true if
// both pass in every lane.
template <int W, int G>
__device__ __forceinline__ bool spec_group(const double* S, const double* Ri, const double* A, int ld, int lane, int i0,
                                           int cvec, double svec, double z1v, double dzv, double bj, double& bs_out)
{
  constexpr int M = G / 4, u0 = W * M;          // this wavefront tests moves u0 .. u0 + M - 1 of the group
  double lg[G];
#pragma unroll
  for (int u = 0; u < G; ++u) lg[u] = L_(S, lane, __builtin_amdgcn_readlane(cvec, i0 + u));
  double rl[M], rh[M], sm[M], zm[M], bm[M];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const int c = __builtin_amdgcn_readlane(cvec, i0 + u0 + m);
    rl[m] = L_(Ri, lane, c);
    rh[m] = L_(A, lane, c);
    sm[m] = readlane_f64(svec, i0 + u0 + m);
    zm[m] = readlane_f64(z1v, i0 + u0 + m);
    bm[m] = bj;
  }
  double bs = bj;
#pragma unroll
  for (int u = 0; u < G; ++u) {
    if (u >= u0 && u < u0 + M) bm[u - u0] = bs;
    bs += lg[u] * readlane_f64(dzv, i0 + u);
  }
  bs_out = bs;
  uint64_t acc = 0ull;
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const double lo = zm[m] - bm[m] * rl[m], hi = zm[m] - bm[m] * rh[m];   // NaN: the row does not bound that side
    // v_cmp masks (inactive lanes and NaN operands give 0), ORed on the scalar unit; 2: ogt, 4: olt
    acc |= __builtin_amdgcn_fcmp(lo, sm[m], 2) | __builtin_amdgcn_fcmp(hi, sm[m], 4) |
           __builtin_amdgcn_fcmp(lo, -1.26, 2) | __builtin_amdgcn_fcmp(hi, 1.26, 4);
  }
  return acc == 0ull;
}
template <int G>
__device__ __forceinline__ bool spec_group_w(int wave, const double* S, const double* Ri, const double* A, int ld, int lane,
                                             int i0, int cvec, double svec, double z1v, double dzv, double bj,
                                             double& bs_out)
{
  switch (wave) {
    case 0: return spec_group<0, G>(S, Ri, A, ld, lane, i0, cvec, svec, z1v, dzv, bj, bs_out);
    case 1: return spec_group<1, G>(S, Ri, A, ld, lane, i0, cvec, svec, z1v, dzv, bj, bs_out);
    case 2: return spec_group<2, G>(S, Ri, A, ld, lane, i0, cvec, svec, z1v, dzv, bj, bs_out);
    default: return spec_group<3, G>(S, Ri, A, ld, lane, i0, cvec, svec, z1v, dzv, bj, bs_out);
  }
}

__global__ __launch_bounds__(kBlock) void k_beta64(blk::BetaArgs a, int mode)
{
  extern __shared__ double lds[];
  // a Cholesky factorisation failed earlier in this chain (ST_NOT_PD is sticky until the host collects the status word):
  // the chain is dead, and for 64 < P <= 256 k_beta has not prepared the workspace k_beta_sweeps reads
  if (*a.status & ST_NOT_PD) return;
  const int P = a.P, t = threadIdx.x, ld = P + 1;
  double* A = lds;                       // PP -> U
  double* S = A + P * ld;                // PP^{-1} -> L
  double* Ri = S + P * ld;               // 1/L elementwise (constrained mode)
  double* mP = Ri + P * ld;
  double* zz = mP + P;
  int* perm = reinterpret_cast<int*>(zz + P);       // P ints
  int* ptab = perm + P + (P & 1);                    // ptab[k][i]: coordinate visited at step i of scan k (P*P ints)
  double* recL = reinterpret_cast<double*>(ptab + P * P + ((P * P) & 1));   // 2 x P records: the scan in progress / next
  double* rec = a.work;                             // P*P records of kRec doubles
  __shared__ int bad;
  __shared__ int uflag;                              // 0: U not ready; 1: U = chol(PP) is in A; 2: PP not positive definite
  if (t == 0) bad = 0;
  if (t == 0) uflag = 0;
  if (a.dbg && t == 0) a.dbg[0] = wall_clock64();
  const bool need_inverse = mode == blk::B_CONSTRAINED || mode == blk::B_FROM_LIK;
  for (int e = t; e < P * P; e += kBlock) {
    const int i = e % P, j = e / P;
    L_(A, i, j) = a.PPsum[e] + a.P0[e];              // PP = P0 + X'OmX
    if (need_inverse) L_(S, i, j) = (i == j) ? 1.0 : 0.0;
  }
  __syncthreads();

  if (mode == blk::B_SOLVE || mode == blk::B_MVN) {
    // nothing to overlap with: the whole workgroup factors and solves (measured 0.136 ms vs 0.161 ms
    // for the single-wave routines below)
    if (!lds_chol_upper(A, P, ld, &bad)) {
      __syncthreads();
      if (t == 0) atomicOr(a.status, ST_NOT_PD);
      return;
    }
    for (int j = t; j < P; j += kBlock) mP[j] = a.bP[j];
    if (mode == blk::B_MVN)
      for (int i = t; i < P; i += kBlock) {
        // eps_i = r.norm(0,1) in stream order: normal i is exactly Philox block i     (Logit.hpp:311)
        const double u1 = beta_stream_unif(a.seed, a.epoch, 2 * i), u2 = beta_stream_unif(a.seed, a.epoch, 2 * i + 1);
        zz[i] = sqrt(-2.0 * log(u1)) * cospi(2.0 * u2);
      }
    __syncthreads();
    lds_solve_Ut(A, mP, P, ld, 1, P);
    lds_solve_U(A, mP, P, ld, 1, P);
    if (mode == blk::B_MVN) {
      lds_solve_U(A, zz, P, ld, 1, P);
      for (int j = t; j < P; j += kBlock) a.beta_out[j] = zz[j] + mP[j];
    } else {
      for (int j = t; j < P; j += kBlock) a.beta_out[j] = mP[j];
    }
    return;
  }

  if (t < 64) {
    // ================= wave 0: the dense stage, alone, no workgroup barriers =================
    const int lane = t;
    bool ok = w_chol_reg<false>(A, P, ld, lane, recL);                        // U = chol(PP,'U') (recL is idle until the sweeps)
    // wave 1 solves for mP from U once its own work is done (it idles otherwise): hand U over
    __threadfence_block();
    if (lane == 0) __hip_atomic_store(&uflag, ok ? 1 : 2, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (a.dbg && t == 0) a.dbg[3] = wall_clock64();
    if (ok && need_inverse) {
      w_inverse_from_U(A, S, P, ld, lane);                                     // S = PP^{-1}
      if (a.dbg && t == 0) a.dbg[4] = wall_clock64();
      if (mode == blk::B_FROM_LIK) {
        // mean = V b ; lower = chol(V,'L') ; beta = mean + lower eps               (Normal.hpp:98-131)
        double mean = 0.0, e = 0.0;
        if (lane < P) {
          for (int k2 = 0; k2 < P; ++k2) mean += L_(S, lane, k2) * a.bP[k2];
          const double u1 = beta_stream_unif(a.seed, a.epoch, 2 * lane), u2 = beta_stream_unif(a.seed, a.epoch, 2 * lane + 1);
          e = sqrt(-2.0 * log(u1)) * cospi(2.0 * u2);
          zz[lane] = e;
        }
        WAVE_SYNC();
        ok = w_chol_lower(S, P, ld, lane);
        if (ok && lane < P) {
          double le = 0.0;
          for (int k2 = 0; k2 <= lane; ++k2) le += L_(S, lane, k2) * zz[k2];
          a.beta_out[lane] = le + mean;
        }
      } else {
        // B_CONSTRAINED set-up, Logit.hpp:335-366 (mP: wave 1; z: after the barrier, when mP is there)
        if (a.dbg && t == 0) a.dbg[1] = wall_clock64();
        ok = w_chol_reg<true>(S, P, ld, lane, recL);                          // L = chol(S,'L')
        if (a.dbg && t == 0) a.dbg[2] = wall_clock64();
      }
    }
    if (!ok && t == 0) bad = 1;
  } else if (mode == blk::B_CONSTRAINED) {
    // ====== waves 1-3, meanwhile: every random input of the draw, in stream order
    // (per scan k, P-1 r.flat for the shuffle, then P tnorm calls of 9 uniforms) ======
    const int tt = t - 64;
    const uint32_t per_scan = (uint32_t)(10 * P - 1);
    // the tnorm records: waves 2-3 take the first 25/32 of them, wave 1 the rest once the scan tables are built
    const int nsplit = (P * P * 25) / 32;
    auto records = [&](int e0, int e1, int first, int stride) {
      for (int e = e0 + first; e < e1; e += stride) {
        const int k = e / P, i = e % P;
        const uint32_t base = (uint32_t)k * per_scan + (uint32_t)(P - 1) + 9u * (uint32_t)i;
        double* R = rec + (size_t)e * kRec;
        for (int m = 0; m < 4; ++m) {
          const double ua = beta_stream_unif(a.seed, a.epoch, base + 2 * m);
          const double ub = beta_stream_unif(a.seed, a.epoch, base + 2 * m + 1);
          const double lua = log(ua);
          R[4 * m] = ua;
          R[4 * m + 1] = lua;
          R[4 * m + 2] = log(ub);
          R[4 * m + 3] = sqrt(-2.0 * lua) * cospi(2.0 * ub);
        }
        R[16] = beta_stream_unif(a.seed, a.epoch, base + 8);
        R[17] = R[18] = R[19] = 0.0;
      }
    };
    if (tt < 64) {
      // wave 1: the scan tables.  Scan k's P-1 swaps (r.flat(i, P), Logit.hpp:375-377) applied to the identity,
      // all scans in parallel (lane k, its row of ptab as scratch), then composed in scan order: `is` persists
      // across scans (Logit.hpp:368-377); the composition is in place, row by row.  One wavefront: no
      // workgroup barrier, wave 0 is in the dense stage.
      if (tt < P) {
        int* sg = ptab + tt * P;
        for (int i = 0; i < P; ++i) sg[i] = i;
        for (int i = 0; i < P - 1; ++i) {
          const double u = beta_stream_unif(a.seed, a.epoch, (uint32_t)tt * per_scan + (uint32_t)i);
          const int j = (int)(unsigned)((double)i + ((double)P - (double)i) * u);     // r.flat(i, P)
          const int tmp = sg[i];
          sg[i] = sg[j];
          sg[j] = tmp;
        }
      }
      WAVE_SYNC();
      for (int k = 1; k < P; ++k) {
        int v = 0;
        if (tt < P) v = ptab[(k - 1) * P + ptab[k * P + tt]];
        WAVE_SYNC();
        if (tt < P) ptab[k * P + tt] = v;
        WAVE_SYNC();
      }
      records(nsplit, P * P, tt, 64);
      // mP = U^{-1} U^{-T} bP (Logit.hpp:335-340), as soon as wave 0 has published U (long since, normally)
      int f;
      do {
        f = __hip_atomic_load(&uflag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (!f) __builtin_amdgcn_s_sleep(8);
      } while (!f);
      if (f == 1) {
        double m = tt < P ? a.bP[tt] : 0.0;
        m = w_solve_Ut_vec(A, m, P, ld, tt);
        m = w_solve_U_vec(A, m, P, ld, tt);
        if (tt < P) mP[tt] = m;
      }
    } else {
      records(0, nsplit, tt - 64, kBlock - 128);
    }
    // the records are read back (staged into LDS) by every wave after the barrier below
    __threadfence_block();
    if (a.dbg && t == 64) a.dbg[11] = wall_clock64();
  }
  if (a.dbg && t == 0) a.dbg[5] = wall_clock64();
  __syncthreads();
  if (bad) {
    if (t == 0) atomicOr(a.status, ST_NOT_PD);
    return;
  }
  if (mode != blk::B_CONSTRAINED) return;

  // 1/L split by the sign test of Logit.hpp:384-391 (see constrained_wide_prepare): Ri where L > 0, A (U is
  // dead by now) where L < 0, NaN elsewhere -- v_max/v_min and the compare masks ignore NaN.  Waves 1-3; wave 0
  // solves for z meanwhile.
  for (int e = t - 64; e >= 0 && e < P * P; e += kBlock - 64) {
    const int i = e % P, j = e / P;
    const double l = L_(S, i, j), r = 1.0 / l;
    const double nan = __builtin_nan("");
    L_(Ri, i, j) = (l > 0.0 && i < P - 1) ? r : nan;
    L_(A, i, j) = (l < 0.0 && i < P - 1) ? r : nan;
  }
  if (t < 64) {
    double z = t < P ? a.beta_prev[t] - mP[t] : 0.0;                         // z = L^{-1}(beta_prev - mP)
    z = w_solve_L_vec(S, z, P, ld, t);
    if (t < P) zz[t] = z;
  }
  __syncthreads();                                 // 1/L complete

  if (a.dbg && t == 0) a.dbg[7] = wall_clock64();
  if (a.dbg && t == 0) a.dbg[9] = clock64();
  // The coordinate sweeps: a move reads only LDS and registers (the next scan's random records travel from
  // global scratch to LDS while a scan runs: loaded into registers at its start, stored at its end).
  //
  // Fast path.  A move's value is almost always the Box-Muller normal s of its first attempt: the bounds
  // contain 0, are wider than sqrt(2 pi), and s falls inside (tnorm_lanes' first branch, attempt 0).  That
  // is decided WITHOUT reducing the bounds: if every lane's lower candidate is <= min(s, -a) and every upper
  // candidate >= max(s, b) for some a, b >= 0 with a + b > sqrt(2 pi), then lo <= 0 <= hi, hi - lo > sqrt(2 pi)
  // and lo <= s <= hi -- three __ballot tests ((a, b) = (1.26, 1.26), (0, 2.51), (2.51, 0): one-sided bounds,
  // the usual case, pass the second or third whatever their finite side is).  Only a move that fails all
  // three pays the 64-lane max/min and tnorm_lanes (0.3 % of the moves on C4).  Same values either way.
  //
  // Speculative groups on four wavefronts.  Moves are taken 16 (or 8) at a time on the fast path: every
  // move of the group is assumed to take attempt 0's normal s (which is what a move whose bounds contain 0,
  // are wider than sqrt(2 pi) and contain s does).  A group is straight-line code: its loads and broadcasts
  // first, one dependent FMA per move (beta after u moves), and a move's test -- every lane's lower
  // candidate <= min(s, -1.26) and upper candidate >= max(s, 1.26), the first of the three tests above --
  // only ORs compare masks into a scalar.  One wavefront issues an instruction every ~8 cycles here, so the
  // FOUR wavefronts of the workgroup (one per SIMD) each keep a replica of beta (lane j = row j), all run
  // the one-FMA-per-move chain, and each tests a quarter of the group's moves; the verdicts meet in LDS at one
  // barrier per group.  A group with a failing move (a few per draw on C4) is redone move by move with all
  // three tests and the full tnorm, by every wavefront alike (same inputs, same arithmetic: the replicas
  // stay identical without another exchange).  Same values as the move-by-move loop either way.
  // z lives in LDS (zz): within a scan every coordinate is visited once, so the z_c of all of a scan's
  // moves are gathered at its start (z1v) and a group's new values are scattered at its end.
  const int lane = t & 63, wave = t >> 6;
  const bool row = lane < P;
  double bj = row ? a.beta_prev[lane] : 0.0;       // beta_j, replicated in every wavefront
  const double inf = __builtin_huge_val();
  const int nrec = P * kRec;
  __shared__ int gflag[2][4];
  for (int e = t; e < nrec; e += kBlock) recL[e] = rec[e];
  __syncthreads();
  bool spec_on = true;
  int gmax = 32;                                   // largest group of the scan
  unsigned gi = 0;                                 // speculative groups so far (flag slot parity)
  for (int k = 0; k < P; ++k) {
    const double* Rk = recL + (k & 1) * nrec;
    // next scan's records: global -> registers now, -> LDS at the end of this scan
    constexpr int kStage = (64 * kRec + kBlock - 1) / kBlock;      // P <= 64
    double stage[kStage];
    if (k + 1 < P) {
      const double* src = rec + (size_t)(k + 1) * nrec;
#pragma unroll
      for (int q = 0; q < kStage; ++q) {
        const int e = t + q * kBlock;
        stage[q] = e < nrec ? src[e] : 0.0;
      }
    }
    const int g4 = (lane < 5 ? lane : 0) * 4;
    const double qnan = __builtin_nan("");
    const int cvec = row ? ptab[k * P + lane] : 0;               // lane i: coordinate of move i
    const double svec = row ? Rk[lane * kRec + 3] : 0.0;         // lane i: attempt 0's normal of move i
    const double z1v = row ? zz[cvec] : 0.0;                     // lane i: z_c before move i
    const double dzv = svec - z1v;
    int nfail = 0;                                                 // moves redone move by move in this scan
    for (int i0 = 0; i0 < P;) {
      const int left = P - i0;
      // groups of 32, 16 or 8 moves (a quarter of them tested by each wavefront); a shorter tail goes move by move
      const int ng = !spec_on ? (left < 8 ? left : 8) : (left >= 32 && gmax >= 32) ? 32 : (left >= 16 && gmax >= 16) ? 16 : left >= 8 ? 8 : left;
      const bool spec = spec_on && ng >= 8;
      double bs = bj;
      bool all_ok = false;
      if (spec) {
        bool ok_l = true;
        if (row) {
          if (ng == 32)
            ok_l = spec_group_w<32>(wave, S, Ri, A, ld, lane, i0, cvec, svec, z1v, dzv, bj, bs);
          else if (ng == 16)
            ok_l = spec_group_w<16>(wave, S, Ri, A, ld, lane, i0, cvec, svec, z1v, dzv, bj, bs);
          else
            ok_l = spec_group_w<8>(wave, S, Ri, A, ld, lane, i0, cvec, svec, z1v, dzv, bj, bs);
        }
        const bool okw = __ballot(!ok_l) == 0ull;                  // this wavefront's moves
        if (lane == 0) gflag[gi & 1][wave] = okw ? 0 : 1;
        __syncthreads();
        const int* gf = gflag[gi & 1];
        all_ok = __builtin_amdgcn_readfirstlane(gf[0] | gf[1] | gf[2] | gf[3]) == 0;
        ++gi;
      }
      if (all_ok) {
        bj = bs;
        if (wave == 0 && row && lane >= i0 && lane < i0 + ng) zz[cvec] = svec;
        i0 += ng;
        continue;
      }
      nfail += ng;
      const int lr = row ? lane : 0;                               // lanes outside the matrix read row 0, masked below
      for (int i = i0; i < i0 + ng; ++i) {
        const int c = __builtin_amdgcn_readlane(cvec, i);
        const double l1 = row ? L_(S, lr, c) : 0.0, rl = row ? L_(Ri, lr, c) : qnan, rh = row ? L_(A, lr, c) : qnan;
        const double* Rn = Rk + i * kRec + g4;
        const double r0 = Rn[0], r1 = Rn[1], r2 = Rn[2], r3 = Rn[3];
        const double z1 = readlane_f64(z1v, i);
        double lo = z1 - bj * rl;                                    // NaN: this row does not bound the move from below
        double hi = z1 - bj * rh;
        const double s = readlane_f64(r3, 0);                        // attempt 0's normal
        const double l0 = s < 0.0 ? s : 0.0, h0 = s > 0.0 ? s : 0.0;
        const double l1s = s < -1.26 ? s : -1.26, h1s = s > 1.26 ? s : 1.26;
        const double l2s = s < -2.51 ? s : -2.51, h2s = s > 2.51 ? s : 2.51;
        double z2 = s;
        if (!(__ballot(lo > l1s || hi < h1s) == 0ull || __ballot(lo > l0 || hi < h2s) == 0ull ||
              __ballot(lo > l2s || hi < h0) == 0ull)) {
          wave_maxmin(lo, hi);           // v_max_f64 / v_min_f64 return the other operand for a NaN
          lo = lo == lo ? lo : -inf;
          hi = hi == hi ? hi : inf;
          z2 = tnorm_lanes(r0, r1, r2, r3, lane, lo, hi);
        }
        const double dz = z2 - z1;
        bj += l1 * dz;                 // L(j, c) = 0 for j < c and l1 = 0 outside the matrix: those rows do not move
        if (wave == 0 && lane == 0) zz[c] = z2;
      }
      i0 += ng;
    }
    // a chain pressed against its bounds fails most groups: stop speculating, look again every 8th scan
    spec_on = 2 * nfail < P || ((k + 1) & 7) == 0;
    gmax = nfail == 0 ? 32 : 8;                                    // failures come in runs (a coordinate at its bound): small groups then
    if (a.dbg && t == 0) a.dbg[8] += (unsigned long long)nfail;
    if (k + 1 < P) {
      double* Rn = recL + ((k + 1) & 1) * nrec;
#pragma unroll
      for (int q = 0; q < kStage; ++q) {
        const int e = t + q * kBlock;
        if (e < nrec) Rn[e] = stage[q];
      }
    }
    __syncthreads();
  }
  if (wave == 0 && row) a.beta_out[lane] = bj;
  if (a.dbg && t == 0) a.dbg[6] = wall_clock64();
  if (a.dbg && t == 0) a.dbg[10] = clock64();
}

// ---- constrained coordinate sweeps for 64 < P <= 256 (Logit.hpp:368-399), same design as k_beta64's:
// every random input generated up front by the whole workgroup (a tnorm call owns nine uniforms whatever
// its bounds), the P^2 moves on ONE wavefront, lane l owning rows l + 64 r (beta in registers), bounds by a
// per-lane fold then the DPP max/min, the four tnorm attempts on lanes 0-3.  L stays in global memory
// (512 KB at P = 256: it lives in L2) next to its elementwise reciprocal; the columns of a move are fetched two
// moves ahead, the records of the next scan are staged into LDS by the idle waves.
// Scratch layout in a.work after the generic stage's 2 P^2 + 2 P doubles: records, then swap targets.
__device__ void constrained_wide_prepare(const blk::BetaArgs& a, const double* __restrict__ Lg, double* Rg)
{
  const int P = a.P, t = threadIdx.x, nthr = (int)blockDim.x;
  double* rec = a.work + 2 * (size_t)P * P + 2 * (size_t)P;
  int* swp = reinterpret_cast<int*>(rec + (size_t)P * P * kRec);
  const uint32_t per_scan = (uint32_t)(10 * P - 1);
  for (int e = t; e < P * (P - 1); e += nthr) {
    const int k = e / (P - 1), i = e % (P - 1);
    const double u = beta_stream_unif(a.seed, a.epoch, (uint32_t)k * per_scan + (uint32_t)i);
    swp[e] = (int)(unsigned)((double)i + ((double)P - (double)i) * u);       // r.flat(i, P), Logit.hpp:375
  }
  for (int e = t; e < P * P; e += nthr) {
    const int k = e / P, i = e % P;
    const uint32_t base = (uint32_t)k * per_scan + (uint32_t)(P - 1) + 9u * (uint32_t)i;
    double* R = rec + (size_t)e * kRec;
    for (int m = 0; m < 4; ++m) {
      const double ua = beta_stream_unif(a.seed, a.epoch, base + 2 * m);
      const double ub = beta_stream_unif(a.seed, a.epoch, base + 2 * m + 1);
      const double lua = log(ua);
      R[4 * m] = ua;
      R[4 * m + 1] = lua;
      R[4 * m + 2] = log(ub);
      R[4 * m + 3] = sqrt(-2.0 * lua) * cospi(2.0 * ub);
    }
    R[16] = beta_stream_unif(a.seed, a.epoch, base + 8);
    R[17] = R[18] = R[19] = 0.0;
  }
  // 1/L elementwise, off the serial loop, split by the sign test of Logit.hpp:384-391: Rlo holds 1/L where
  // L > 0 (those rows bound the move from below), Rhi where L < 0, NaN elsewhere -- above the diagonal
  // (L = 0: rows j < c are outside the loop of :383) and in row P-1 (the loop stops at P-2).  v_max_f64 /
  // v_min_f64 return the other operand for a NaN, so the sweeps need no compares or selects.
  double* Rhi = rec + (size_t)P * P * kRec + ((size_t)P * P + 1) / 2;
  const double nan = __builtin_nan("");
  for (int e = t; e < P * P; e += nthr) {
    const int j = e % P;
    const double l = Lg[e];
    const double r = 1.0 / l;
    Rg[e] = (l > 0.0 && j < P - 1) ? r : nan;
    Rhi[e] = (l < 0.0 && j < P - 1) ? r : nan;
  }
}

// The sweeps themselves: one 4-wave workgroup (512 registers per lane available: the pipeline's register
// sets do not spill), launched behind k_beta on the same stream.  Reads L, 1/L, z from k_beta's scratch.
template <int RPL>
__global__ __launch_bounds__(kBlock) void k_beta_sweeps(blk::BetaArgs a)
{
  extern __shared__ double lds[];
  // a Cholesky factorisation failed earlier in this chain (ST_NOT_PD is sticky until the host collects the status word):
  // the chain is dead, and for 64 < P <= 256 k_beta has not prepared the workspace k_beta_sweeps reads
  if (*a.status & ST_NOT_PD) return;
  const int P = a.P, t = threadIdx.x, nthr = (int)blockDim.x;
  const int nrec = P * kRec;
  const double* __restrict__ Rg = a.work;                          // 1/L where L > 0, else NaN (k_beta's A)
  const double* __restrict__ Lg = a.work + (size_t)P * P;          // L     (k_beta's S)
  const double* zz = a.work + 2 * (size_t)P * P + P;               // z
  double* recL = lds;                                  // 2 x (P records): the scan in progress / next
  double* sz = recL + 2 * nrec;                        // z
  unsigned char* ptab = reinterpret_cast<unsigned char*>(sz + P);   // ptab[k][i]: coordinate of move i of scan k
  double* rec = a.work + 2 * (size_t)P * P + 2 * (size_t)P;
  int* swp = reinterpret_cast<int*>(rec + (size_t)P * P * kRec);
  const double* __restrict__ Rh = rec + (size_t)P * P * kRec + ((size_t)P * P + 1) / 2;   // 1/L where L < 0, else NaN
  for (int j = t; j < P; j += nthr) sz[j] = zz[j];
  __syncthreads();
  // scan permutations: each scan's swaps on the identity (thread k), then composed in scan order
  if (t < P) {
    unsigned char* sg = ptab + t * P;
    for (int i = 0; i < P; ++i) sg[i] = (unsigned char)i;
    for (int i = 0; i < P - 1; ++i) {
      const int j = swp[t * (P - 1) + i];
      const unsigned char tmp = sg[i];
      sg[i] = sg[j];
      sg[j] = tmp;
    }
  }
  __syncthreads();
  for (int k = 1; k < P; ++k) {
    int v = 0;
    if (t < P) v = ptab[(k - 1) * P + ptab[k * P + t]];
    __syncthreads();
    if (t < P) ptab[k * P + t] = (unsigned char)v;
    __syncthreads();
  }

  const int lane = t & 63;
  const bool serial = t < 64;
  const double inf = __builtin_huge_val();
  double bj[RPL];
#pragma unroll
  for (int r = 0; r < RPL; ++r) {
    const int j = lane + 64 * r;
    bj[r] = (serial && j < P) ? a.beta_prev[j] : 0.0;
  }
  for (int e = t; e < nrec; e += nthr) recL[e] = rec[e];
  __syncthreads();
  for (int k = 0; k < P; ++k) {
    const double* Rk = recL + (k & 1) * nrec;
    if (!serial) {
      if (k + 1 < P) {
        double* Rn = recL + ((k + 1) & 1) * nrec;
        const double* src = rec + (size_t)(k + 1) * nrec;
        for (int e = t - 64; e < nrec; e += nthr - 64) Rn[e] = src[e];
      }
    } else {
      const int g4 = (lane < 5 ? lane : 0) * 4;
      const unsigned char* pk = ptab + k * P;
      // software pipeline, kDepth register sets used round-robin (no copies: a copy would wait on the
      // load): the columns of L and 1/L of move i + kDepth are requested as soon as move i has used its
      // set; z and the record come from LDS one move ahead
      constexpr int kDepth = 4;
      double lq[kDepth][RPL], rlo[kDepth][RPL], rhi[kDepth][RPL];
      int jr[RPL];           // this lane's rows, clamped to P-1 (a clamped row sees NaN reciprocals: no effect)
#pragma unroll
      for (int r = 0; r < RPL; ++r) jr[r] = (lane + 64 * r) < P ? lane + 64 * r : P - 1;
#pragma unroll
      for (int u = 0; u < kDepth; ++u) {
        const size_t co = (size_t)__builtin_amdgcn_readfirstlane((int)pk[u < P ? u : P - 1]) * P;
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
          lq[u][r] = (Lg + co)[jr[r]];
          rlo[u][r] = (Rg + co)[jr[r]];
          rhi[u][r] = (Rh + co)[jr[r]];
        }
      }
      int c_n = pk[0];
      double z1_n = sz[c_n];
      double q0 = Rk[g4], q1 = Rk[g4 + 1], q2 = Rk[g4 + 2], q3 = Rk[g4 + 3];
      for (int i0 = 0; i0 < P; i0 += kDepth) {
#pragma unroll
        for (int u = 0; u < kDepth; ++u) {
          const int i = i0 + u;
          if (i < P) {
            const int c = __builtin_amdgcn_readfirstlane(c_n);
            const double r0 = q0, r1 = q1, r2 = q2, r3 = q3;
            double l1[RPL], rl[RPL], rh[RPL];
#pragma unroll
            for (int r = 0; r < RPL; ++r) {
              l1[r] = lq[u][r];
              rl[r] = rlo[u][r];
              rh[r] = rhi[u][r];
            }
            if (i + kDepth < P) {
              const size_t co = (size_t)__builtin_amdgcn_readfirstlane((int)pk[i + kDepth]) * P;
#pragma unroll
              for (int r = 0; r < RPL; ++r) {
                lq[u][r] = (Lg + co)[jr[r]];
                rlo[u][r] = (Rg + co)[jr[r]];
                rhi[u][r] = (Rh + co)[jr[r]];
              }
            }
            if (i + 1 < P) {
              c_n = pk[i + 1];
              const double* Rn = Rk + (i + 1) * kRec + g4;
              q0 = Rn[0];
              q1 = Rn[1];
              q2 = Rn[2];
              q3 = Rn[3];
            }
            // fast path of k_beta64: attempt 0's normal is the move's value if three ballots say so
            const double z1 = z1_n;
            double lo = -inf, hi = inf;
#pragma unroll
            for (int r = 0; r < RPL; ++r) {
              lo = vmax64(lo, z1 - bj[r] * rl[r]);      // NaN (row not in the lower set) leaves lo as it is
              hi = vmin64(hi, z1 - bj[r] * rh[r]);
            }
            const double s0 = readlane_f64(r3, 0);
            const double l0s = s0 < 0.0 ? s0 : 0.0, h0s = s0 > 0.0 ? s0 : 0.0;
            const double l1s = s0 < -1.26 ? s0 : -1.26, h1s = s0 > 1.26 ? s0 : 1.26;
            const double l2s = s0 < -2.51 ? s0 : -2.51, h2s = s0 > 2.51 ? s0 : 2.51;
            double z2 = s0;
            if (!(__ballot(lo > l1s || hi < h1s) == 0ull || __ballot(lo > l0s || hi < h2s) == 0ull ||
                  __ballot(lo > l2s || hi < h0s) == 0ull)) {
              wave_maxmin(lo, hi);
              z2 = tnorm_lanes(r0, r1, r2, r3, lane, lo, hi);
            }
            const double dz = z2 - z1;
#pragma unroll
            for (int r = 0; r < RPL; ++r) bj[r] += l1[r] * dz;    // L(j, c) = 0 for j < c: rows above c do not move
            if (lane == 0) sz[c] = z2;
            if (i + 1 < P) z1_n = sz[__builtin_amdgcn_readfirstlane(c_n)];   // after the stores above in program order
          }
        }
      }
    }
    __syncthreads();
  }
  if (serial) {
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
      const int j = lane + 64 * r;
      if (j < P) a.beta_out[j] = bj[r];
    }
  }
}
#undef L_

// Running moments of a chain: Welford update of (mean, M2) with the count-th sample x.
__global__ __launch_bounds__(kBlock) void k_welford(const double* __restrict__ x, double* __restrict__ mean,
                                                    double* __restrict__ m2, int64_t n, double count)
{
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  const double inv = 1.0 / count;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const double v = x[i], m = mean[i];
    const double d = v - m;
    const double mn = m + d * inv;
    mean[i] = mn;
    m2[i] += d * (v - mn);
  }
}
// M2 -> sample variance M2 / (count - 1) (0 when count < 2)
__global__ __launch_bounds__(kBlock) void k_welford_finish(double* __restrict__ m2, int64_t n, double count)
{
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  const double s = count > 1.0 ? 1.0 / (count - 1.0) : 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) m2[i] *= s;
}

__global__ void k_maxabsdiff(const double* a, const double* b, int P, double* out)
{
  __shared__ double sm[kBlock];
  double m = 0.0;
  for (int j = threadIdx.x; j < P; j += kBlock) m = fmax(m, fabs(a[j] - b[j]));
  sm[threadIdx.x] = m;
  __syncthreads();
  for (int s = kBlock / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sm[threadIdx.x] = fmax(sm[threadIdx.x], sm[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = sm[0];
}

__global__ void k_vec_add(double* dst, const double* a, const double* b, int P)
{
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < P; j += gridDim.x * blockDim.x)
    dst[j] = a[j] + (b ? b[j] : 0.0);
}

__global__ void k_matvec(double* dst, const double* M, const double* v, int P)
{
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += gridDim.x * blockDim.x) {
    double s = 0.0;
    for (int j = 0; j < P; ++j) s += M_(M, i, j) * v[j];
    dst[i] = s;
  }
}

inline int grid_for(int64_t n, int block, int maxb)
{
  int64_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > maxb) g = maxb;
  return (int)g;
}

template <int NB, bool EXACT>
void launch_draw_pass(const blk::SweepPlan& plan, const double* tX, const double* n, const double* beta,
                      const double* off, double* w, int64_t N, uint64_t seed, uint32_t epoch, uint64_t idx0, int mode,
                      int* status, hipStream_t s)
{
  if (mode == blk::W_DRAW)
    hipLaunchKernelGGL((k_psi_omega_nb<NB, blk::W_DRAW, EXACT>), dim3(plan.nblocks_draw), dim3(kBlock), 0, s, tX, n,
                       beta, off, w, N, plan.P, plan.chunk_rows, seed, epoch, idx0, status);
  else
    hipLaunchKernelGGL((k_psi_omega_nb<NB, blk::W_EM, EXACT>), dim3(plan.nblocks_draw), dim3(kBlock), 0, s, tX, n, beta,
                       off, w, N, plan.P, plan.chunk_rows, seed, epoch, idx0, status);
}

template <int NB, bool EXACT>
void launch_nb_x(const blk::SweepPlan& plan, const double* tX, const double* n, const double* beta, const double* off,
                 double* w, int64_t N, double* partial, double* PP, uint64_t seed, uint32_t epoch, uint64_t idx0,
                 int mode, int* status, hipStream_t s)
{
  constexpr int E = NB * (NB + 1) / 2 * 4 * 64;
  launch_draw_pass<NB, EXACT>(plan, tX, n, beta, off, w, N, seed, epoch, idx0, mode, status, s);
  hipLaunchKernelGGL((k_xwx_mfma<NB, EXACT>), dim3(plan.nblocks), dim3(kBlock), 0, s, tX, w, N, plan.P, partial);
  hipLaunchKernelGGL((k_reduce_fused<NB>), dim3((E + 63) / 64), dim3(1024), 0, s, partial, plan.nblocks, PP, plan.P);
}
template <int NB>
void launch_nb(const blk::SweepPlan& plan, const double* tX, const double* n, const double* beta, const double* off,
               double* w, int64_t N, double* partial, double* PP, uint64_t seed, uint32_t epoch, uint64_t idx0,
               int mode, int* status, hipStream_t s)
{
  if (plan.P == 16 * NB)
    launch_nb_x<NB, true>(plan, tX, n, beta, off, w, N, partial, PP, seed, epoch, idx0, mode, status, s);
  else
    launch_nb_x<NB, false>(plan, tX, n, beta, off, w, N, partial, PP, seed, epoch, idx0, mode, status, s);
}

template <int NB, int NW, bool EXACT>
void launch_nb_big_x(const blk::SweepPlan& plan, const double* tX, const double* n, const double* beta,
                     const double* off, double* w, int64_t N, double* partial, double* PP, uint64_t seed,
                     uint32_t epoch, uint64_t idx0, int mode, int* status, hipStream_t s)
{
  constexpr int E = NB * (NB + 1) / 2 * 256;
  constexpr int P = 16 * NB;
  constexpr size_t lds = (2 * 16 * (size_t)(P + 16) + 2 * 16) * sizeof(double);
  launch_draw_pass<NB, EXACT>(plan, tX, n, beta, off, w, N, seed, epoch, idx0, mode, status, s);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)k_xwx_mfma_big<NB, NW, EXACT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((k_xwx_mfma_big<NB, NW, EXACT>), dim3(plan.nblocks), dim3(NW * 64), lds, s, tX, w, N, plan.P,
                     partial);
  hipLaunchKernelGGL((k_reduce_big<NB>), dim3((E + 63) / 64), dim3(1024), 0, s, partial, plan.nblocks, PP, plan.P);
}
template <int NB, int NW>
void launch_nb_big(const blk::SweepPlan& plan, const double* tX, const double* n, const double* beta,
                   const double* off, double* w, int64_t N, double* partial, double* PP, uint64_t seed, uint32_t epoch,
                   uint64_t idx0, int mode, int* status, hipStream_t s)
{
  if (plan.P == 16 * NB)
    launch_nb_big_x<NB, NW, true>(plan, tX, n, beta, off, w, N, partial, PP, seed, epoch, idx0, mode, status, s);
  else
    launch_nb_big_x<NB, NW, false>(plan, tX, n, beta, off, w, N, partial, PP, seed, epoch, idx0, mode, status, s);
}

}  // namespace

namespace blk {

// psi/omega pass: two 4-wave workgroups per CU are resident (launch bounds of k_psi_omega_nb);
// every wave takes r chunks of chunk_rows rows, r the smallest count that keeps chunk_rows <= kSuper,
// so the grid has no partial last round.
static void plan_draw_pass(SweepPlan& p, int64_t N, int num_cus)
{
  const int64_t waves = 2 * 4 * (int64_t)num_cus;
  const int64_t r = (N + waves * kSuper - 1) / (waves * kSuper);
  int64_t chunk = (N + waves * r - 1) / (waves * (r < 1 ? 1 : r));
  chunk = (chunk + 63) / 64 * 64;
  if (chunk < 64) chunk = 64;
  if (chunk > kSuper) chunk = kSuper;
  const int64_t nchunks = (N + chunk - 1) / chunk;
  int64_t nd = (nchunks + 3) / 4;
  if (nd < 1) nd = 1;
  if (nd > 2 * (int64_t)num_cus) nd = 2 * (int64_t)num_cus;
  p.chunk_rows = (int)chunk;
  p.nblocks_draw = (int)nd;
}

SweepPlan make_plan(int64_t N, int P, int num_cus)
{
  SweepPlan p;
  p.P = P;
  if (P > 64 && P <= 256) {
    p.fused = 2;                                 // LDS-tiled MFMA kernel (compute-bound at P = 256)
    p.nb = P <= 128 ? 8 : 16;                    // columns padded (as zeros, in registers) to 128 / 256
    const int64_t ntiles = (N + 15) / 16;
    int64_t nb = ntiles < 1 ? 1 : ntiles;
    if (nb > (int64_t)num_cus) nb = num_cus;     // one workgroup (two waves per SIMD) per CU
    p.nblocks = (int)nb;
    plan_draw_pass(p, N, num_cus);
    p.partial_doubles = (size_t)p.nblocks * (p.nb * (p.nb + 1) / 2) * 256;
  } else if (P >= 1 && P <= 64) {
    p.fused = 1;
    p.nb = (P + 15) / 16;                        // columns padded (as zeros, in registers) to 16 nb
    const int64_t ntiles = (N + 63) / 64;
    int64_t nb = (ntiles + 3) / 4;
    if (nb < 1) nb = 1;
    if (nb > 2 * (int64_t)num_cus) nb = 2 * (int64_t)num_cus;   // pass 2: two 4-wave workgroups per CU
    p.nblocks = (int)nb;
    plan_draw_pass(p, N, num_cus);
    p.partial_doubles = (size_t)p.nblocks * (p.nb * (p.nb + 1) / 2) * 4 * 64;
  } else {
    p.fused = 0;
    const int T = (P + 63) / 64;
    p.ntile = T * (T + 1) / 2;
    int64_t chunks = (N + 4095) / 4096;
    if (chunks < 1) chunks = 1;
    if (chunks > 512) chunks = 512;
    p.nblocks = (int)chunks;
    p.partial_doubles = (size_t)p.nblocks * p.ntile * 4096;
  }
  return p;
}

void launch_sweep(const SweepPlan& plan, const double* tX, const double* n, const double* beta, const double* off,
                  double* w_store, double* w_scratch, int64_t N, double* partial, double* PPpart, uint64_t seed,
                  uint32_t epoch, uint64_t idx0, int mode, int* status, hipStream_t s)
{
  const int P = plan.P;
  double* w = w_store ? w_store : w_scratch;
  if (plan.fused == 2) {
    if (plan.nb == 8)
      launch_nb_big<8, 4>(plan, tX, n, beta, off, w, N, partial, PPpart, seed, epoch, idx0, mode, status, s);
    else
      launch_nb_big<16, 8>(plan, tX, n, beta, off, w, N, partial, PPpart, seed, epoch, idx0, mode, status, s);
    return;
  }
  if (plan.fused) {
    switch (plan.nb) {
      case 1: launch_nb<1>(plan, tX, n, beta, off, w, N, partial, PPpart, seed, epoch, idx0, mode, status, s); break;
      case 2: launch_nb<2>(plan, tX, n, beta, off, w, N, partial, PPpart, seed, epoch, idx0, mode, status, s); break;
      case 3: launch_nb<3>(plan, tX, n, beta, off, w, N, partial, PPpart, seed, epoch, idx0, mode, status, s); break;
      default: launch_nb<4>(plan, tX, n, beta, off, w, N, partial, PPpart, seed, epoch, idx0, mode, status, s);
    }
    return;
  }
  const int g = grid_for(N, kBlock, 256 * 8);
  if (mode == W_DRAW)
    hipLaunchKernelGGL((k_psi_omega<W_DRAW>), dim3(g), dim3(kBlock), sizeof(double) * P, s, tX, n, beta, off, w, N, P,
                       seed, epoch, idx0, status);
  else
    hipLaunchKernelGGL((k_psi_omega<W_EM>), dim3(g), dim3(kBlock), sizeof(double) * P, s, tX, n, beta, off, w, N, P,
                       seed, epoch, idx0, status);
  const int T = (P + 63) / 64;
  int64_t rpc = (N + plan.nblocks - 1) / plan.nblocks;
  rpc = (rpc + kRows - 1) / kRows * kRows;
  if (rpc < kRows) rpc = kRows;
  hipLaunchKernelGGL(k_xwx_tiles, dim3(plan.nblocks, plan.ntile), dim3(kBlock), 0, s, tX, w, N, P, T, rpc, partial);
  hipLaunchKernelGGL(k_reduce_tiles, dim3(grid_for((int64_t)plan.ntile * 4096, kBlock, 1024)), dim3(kBlock), 0, s,
                     partial, plan.nblocks, P, T, PPpart);
}

static int colsum_blocks(int64_t N)
{
  int64_t b = (N + 2047) / 2048;
  if (b < 1) b = 1;
  if (b > 1024) b = 1024;
  return (int)b;
}
size_t colsum_ws_doubles(int64_t N, int P) { return (size_t)colsum_blocks(N) * P; }

void launch_colsum(const double* tX, const double* y, const double* n, const double* w, const double* c, int64_t N,
                   int P, double* ws, double* out, hipStream_t s)
{
  const int nb = colsum_blocks(N);
  const int64_t rpb = (N + nb - 1) / nb;
  const int cols = P < kBlock ? P : kBlock;
  const int rowlanes = kBlock / cols;
  hipLaunchKernelGGL(k_colsum, dim3(nb), dim3(kBlock), sizeof(double) * rowlanes * P, s, tX, y, n, w, c, N, P,
                     rpb > 0 ? rpb : 1, ws);
  hipLaunchKernelGGL(k_colsum_reduce, dim3(grid_for(P, kBlock, 64)), dim3(kBlock), 0, s, ws, nb, P, out);
}

void launch_xbeta(const double* tX, const double* beta, int64_t N, int P, double* out, hipStream_t s)
{
  if (N <= 0) return;
  hipLaunchKernelGGL(k_xbeta, dim3(grid_for(N, kBlock, 256 * 8)), dim3(kBlock), sizeof(double) * P, s, tX, beta, N, P,
                     out);
}

void launch_mlogit_offset(const double* XB, int64_t N, int J, int j, double* c_out, hipStream_t s)
{
  if (N <= 0) return;
  hipLaunchKernelGGL(k_mlogit_offset, dim3(grid_for(N, kBlock, 256 * 8)), dim3(kBlock), 0, s, XB, N, J, j, c_out);
}

size_t beta_work_doubles(int P)
{
  size_t generic = 2 * (size_t)P * P + 6 * (size_t)P + 64;
  if (P > 64 && P <= 256)    // constrained_sweeps_wide: tnorm records + swap targets after the dense stage's matrices
    generic += (size_t)P * P * kRec + ((size_t)P * P + 1) / 2 + (size_t)P * P;   // + the second reciprocal matrix
  const size_t small = (size_t)P * P * kRec + 2 * (((size_t)P * P + 1) / 2) + 64;   // tnorm records + int tables
  return generic > small ? generic : small;
}

void launch_beta(const BetaArgs& a, int mode, hipStream_t s)
{
  if (a.P <= 64) {
    const int ld = a.P + 1;
    const size_t lds = (3 * (size_t)a.P * ld + 2 * (size_t)a.P) * 8 + ((size_t)a.P + 1 + (size_t)a.P * a.P + 1) * 4 +
                       2 * (size_t)a.P * kRec * 8 + 32;
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)k_beta64, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_beta64, dim3(1), dim3(kBlock), lds, s, a, mode);
    return;
  }
  size_t lds = 0;
  if (mode == B_CONSTRAINED) {
    const size_t pp = (size_t)a.P * a.P * 8;
    if (a.P <= 256)   // constrained_sweeps_wide: two scans of records, z, byte permutation table
      lds = (2 * (size_t)a.P * kRec + (size_t)a.P) * 8 + (size_t)a.P * a.P + 64;
    else
      lds = (2 * (size_t)a.P + (a.P + 1) / 2 + 1) * 8 + (pp <= 128 * 1024 ? pp : 0);
  }
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)k_beta, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const bool wide = mode == B_CONSTRAINED && a.P <= 256;
  hipLaunchKernelGGL(k_beta, dim3(1), dim3(a.P > 128 ? 1024 : 256), wide ? 0 : lds, s, a, mode);
  if (wide) {
    if (a.P <= 128) {
      if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)k_beta_sweeps<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(k_beta_sweeps<2>, dim3(1), dim3(kBlock), lds, s, a);
    } else {
      if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)k_beta_sweeps<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(k_beta_sweeps<4>, dim3(1), dim3(kBlock), lds, s, a);
    }
  }
}

void launch_welford(const double* x, double* mean, double* m2, int64_t n, int64_t count, hipStream_t s)
{
  if (n <= 0) return;
  int64_t g = (n + kBlock - 1) / kBlock;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(k_welford, dim3((int)g), dim3(kBlock), 0, s, x, mean, m2, n, (double)count);
}
void launch_welford_finish(double* m2, int64_t n, int64_t count, hipStream_t s)
{
  if (n <= 0) return;
  int64_t g = (n + kBlock - 1) / kBlock;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(k_welford_finish, dim3((int)g), dim3(kBlock), 0, s, m2, n, (double)count);
}

void launch_maxabsdiff(const double* a, const double* b, int P, double* out, hipStream_t s)
{
  hipLaunchKernelGGL(k_maxabsdiff, dim3(1), dim3(kBlock), 0, s, a, b, P, out);
}
void launch_vec_add(double* dst, const double* a, const double* b, int P, hipStream_t s)
{
  hipLaunchKernelGGL(k_vec_add, dim3(1), dim3(kBlock), 0, s, dst, a, b, P);
}
void launch_matvec(double* dst, const double* M, const double* v, int P, hipStream_t s)
{
  hipLaunchKernelGGL(k_matvec, dim3(1), dim3(kBlock), 0, s, dst, M, v, P);
}

}  // namespace blk
