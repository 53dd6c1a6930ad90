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
#include "bl_dpp.hpp"
#include "bl_gibbs_kernels.hpp"
#include <stdlib.h>
#include "bl_host.hpp"
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
        part = row16_allsum(part);          // (bl_dpp.hpp: the xor butterfly's bits, without LDS)
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
// VEC (mlogit, MultLogit.hpp:246-247): X' Omega c in the same pass -- one more MFMA per block-row against an operand
// whose only non-zero column is c (column 0 of NB extra accumulator blocks), instead of a second pass over X.
template <int NB, bool EXACT, bool VEC>
__global__ __launch_bounds__(kBlock, 2) void k_xwx_mfma(const double* __restrict__ tX, const double* __restrict__ w,
                                                        const double* __restrict__ cvec, int64_t N, int Pa,
                                                        double* __restrict__ partial)
{
  constexpr int NBLK = NB * (NB + 1) / 2 + (VEC ? NB : 0);
  __shared__ double red[2][NBLK * 4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = lane >> 4, c = lane & 15;
  const int64_t ntiles = (N + 63) / 64;
  const int64_t W = (int64_t)gridDim.x * 4;

  struct {
    v4d a[NBLK];
  } acc;
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
    const double cmy = (VEC && myrow < N) ? cvec[myrow] : 0.0;
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
      if (VEC) {
        const double cg = __shfl(cmy, (lane & 48) | g);
        const double bv = c == 0 ? cg : 0.0;             // B[k][0] = c of row k of the group
#pragma unroll
        for (int qa = 0; qa < NB; ++qa)
          acc.a[NB * (NB + 1) / 2 + qa] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[qa], bv, acc.a[NB * (NB + 1) / 2 + qa], 0, 0, 0);
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
template <int NB, bool VEC>
__global__ __launch_bounds__(1024) void k_reduce_fused(const double* __restrict__ partial, int nparts,
                                                       double* __restrict__ PP, int Pa, double* __restrict__ xoc)
{
  constexpr int NTRI = NB * (NB + 1) / 2;
  constexpr int NBLK = NTRI + (VEC ? NB : 0);
  constexpr int E = NBLK * 4 * 64;
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
    const int blkid = e / 256, reg = (e >> 6) & 3, ln = e & 63;
    if (VEC && blkid >= NTRI) {                  // column 0 of block-row blkid - NTRI of X' Omega c
      const int A = colmap<NB>(blkid - NTRI, (ln >> 4) + 4 * reg);
      if ((ln & 15) == 0 && A < Pa) xoc[A] = tot;
      return;
    }
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
  const double sum = e < E ? blk::slab_sum16(partial, E, e, s, nparts) : 0.0;
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
                                                          double* __restrict__ c_out, double* __restrict__ eta_out)
{
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < N; i += stride) {
    double A = 0.0;
    for (int k = 0; k < J; ++k)
      if (k != j) A += exp(XB[(size_t)k * N + i]);
    const double cj = log(A);
    c_out[i] = cj;
    eta_out[i] = XB[(size_t)j * N + i] - cj;        // eta_j = XB.col(j) - c_j, MultLogit.hpp:300
  }
}


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

// dst[i] = src[i * stride + off]  (row j of a U x N column-major matrix: mlogit's y_j, MultLogit.hpp:214-219)
__global__ void k_gather_stride(double* __restrict__ dst, const double* __restrict__ src, int64_t n, int stride, int off)
{
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = src[(size_t)i * stride + off];
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
    for (int j = 0; j < P; ++j) s += M[(size_t)i + (size_t)j * (size_t)P] * v[j];
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
                 int mode, int* status, hipStream_t s, int parts, double* xoc)
{
  constexpr int E = NB * (NB + 1) / 2 * 4 * 64;
  if (parts & 1) launch_draw_pass<NB, EXACT>(plan, tX, n, beta, off, w, N, seed, epoch, idx0, mode, status, s);
  if (!(parts & 2)) return;
  if (xoc) {
    constexpr int EV = E + NB * 4 * 64;
    hipLaunchKernelGGL((k_xwx_mfma<NB, EXACT, true>), dim3(plan.nblocks), dim3(kBlock), 0, s, tX, w, off, N, plan.P,
                       partial);
    hipLaunchKernelGGL((k_reduce_fused<NB, true>), dim3((EV + 63) / 64), dim3(1024), 0, s, partial, plan.nblocks, PP,
                       plan.P, xoc);
    return;
  }
  hipLaunchKernelGGL((k_xwx_mfma<NB, EXACT, false>), dim3(plan.nblocks), dim3(kBlock), 0, s, tX, w, nullptr, N, plan.P,
                     partial);
  hipLaunchKernelGGL((k_reduce_fused<NB, false>), dim3((E + 63) / 64), dim3(1024), 0, s, partial, plan.nblocks, PP,
                     plan.P, nullptr);
}
template <int NB>
void launch_nb(const blk::SweepPlan& plan, const double* tX, const double* n, const double* beta, const double* off,
               double* w, int64_t N, double* partial, double* PP, uint64_t seed, uint32_t epoch, uint64_t idx0,
               int mode, int* status, hipStream_t s, int parts, double* xoc)
{
  if (plan.P == 16 * NB)
    launch_nb_x<NB, true>(plan, tX, n, beta, off, w, N, partial, PP, seed, epoch, idx0, mode, status, s, parts, xoc);
  else
    launch_nb_x<NB, false>(plan, tX, n, beta, off, w, N, partial, PP, seed, epoch, idx0, mode, status, s, parts, xoc);
}

template <int NB, int NW, bool EXACT>
void launch_nb_big_x(const blk::SweepPlan& plan, const double* tX, const double* n, const double* beta,
                     const double* off, double* w, int64_t N, double* partial, double* PP, uint64_t seed,
                     uint32_t epoch, uint64_t idx0, int mode, int* status, hipStream_t s, int parts)
{
  constexpr int E = NB * (NB + 1) / 2 * 256;
  constexpr int P = 16 * NB;
  constexpr size_t lds = (2 * 16 * (size_t)(P + 16) + 2 * 16) * sizeof(double);
  if (parts & 1) launch_draw_pass<NB, EXACT>(plan, tX, n, beta, off, w, N, seed, epoch, idx0, mode, status, s);
  if (!(parts & 2)) return;
  if (EXACT && N > 0) {      // P = 128 / 256: the small matrix instruction (kernels_xwx4.hip); padded P: the kernel below
    blk::launch_xwx_q4_big(plan.nblocks, NB, tX, w, N, partial, PP, s);
    return;
  }
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
                   uint64_t idx0, int mode, int* status, hipStream_t s, int parts)
{
  if (plan.P == 16 * NB)
    launch_nb_big_x<NB, NW, true>(plan, tX, n, beta, off, w, N, partial, PP, seed, epoch, idx0, mode, status, s, parts);
  else
    launch_nb_big_x<NB, NW, false>(plan, tX, n, beta, off, w, N, partial, PP, seed, epoch, idx0, mode, status, s, parts);
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

unsigned long long* sweep_once_deferred_counter(const SweepPlan& plan, double* ws, int64_t N)
{
  if (plan.P == 64) return sweep_once64_deferred_counter(ws, plan.nblocks, N);
  if (plan.P == 256) return sweep_once256_deferred_counter(ws, plan.nblocks, N);
  return nullptr;
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
    if (xwx_q4_big_ws_doubles(p.nblocks, p.nb) > p.partial_doubles) p.partial_doubles = xwx_q4_big_ws_doubles(p.nblocks, p.nb);
    if (P == 256 && sweep_once256_ws_doubles(p.nblocks, N) > p.partial_doubles)       // the single-pass sweep's workspace
      p.partial_doubles = sweep_once256_ws_doubles(p.nblocks, N);
  } else if (P >= 1 && P <= 64) {
    p.fused = 1;
    p.nb = (P + 15) / 16;                        // columns padded (as zeros, in registers) to 16 nb
    const int64_t ntiles = (N + 63) / 64;
    int64_t nb = (ntiles + 3) / 4;
    if (nb < 1) nb = 1;
    if (nb > 2 * (int64_t)num_cus) nb = 2 * (int64_t)num_cus;   // pass 2: two 4-wave workgroups per CU
    p.nblocks = (int)nb;
    plan_draw_pass(p, N, num_cus);
    p.partial_doubles = (size_t)p.nblocks * (p.nb * (p.nb + 1) / 2 + p.nb) * 4 * 64;   // + the X' Omega c blocks of mlogit
    if (P == 64) {                                                                     // the single-pass sweep's workspace
      const size_t once = sweep_once64_ws_doubles(p.nblocks, N);
      if (once > p.partial_doubles) p.partial_doubles = once;
    }
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
                  uint32_t epoch, uint64_t idx0, int mode, int* status, hipStream_t s, int parts, double* xoc)
{
  const int P = plan.P;
  double* w = w_store ? w_store : w_scratch;
  if (plan.fused == 2 && plan.P == 256 && mode == W_DRAW && !off && parts == 3 && !xoc && N > 0 &&
      (plan.single_pass < 0 ? blh::sweep_single_pass() : plan.single_pass)) {
    // X read once (kernels_sweep256.hip); omega is stored only if the caller wants it
    launch_sweep_once256(plan.nblocks, tX, n, beta, w_store, N, partial, PPpart, seed, epoch, idx0, status,
                         blh::sweep_stats(), s);
    return;
  }
  if (plan.fused == 2) {
    if (plan.nb == 8)
      launch_nb_big<8, 4>(plan, tX, n, beta, off, w, N, partial, PPpart, seed, epoch, idx0, mode, status, s, parts);
    else
      launch_nb_big<16, 8>(plan, tX, n, beta, off, w, N, partial, PPpart, seed, epoch, idx0, mode, status, s, parts);
    return;
  }
  if (plan.fused == 1 && plan.P == 64 && mode == W_DRAW && !off && parts == 3 && !xoc && N > 0 &&
      (plan.single_pass < 0 ? blh::sweep_single_pass() : plan.single_pass)) {
    // X read once (kernels_sweep1.hip); omega is stored only if the caller wants it
    launch_sweep_once64(plan.nblocks, tX, n, beta, w_store, N, partial, PPpart, seed, epoch, idx0, status,
                        blh::sweep_stats(), s);
    return;
  }
  if (plan.fused) {
    switch (plan.nb) {
      case 1: launch_nb<1>(plan, tX, n, beta, off, w, N, partial, PPpart, seed, epoch, idx0, mode, status, s, parts, xoc); break;
      case 2: launch_nb<2>(plan, tX, n, beta, off, w, N, partial, PPpart, seed, epoch, idx0, mode, status, s, parts, xoc); break;
      case 3: launch_nb<3>(plan, tX, n, beta, off, w, N, partial, PPpart, seed, epoch, idx0, mode, status, s, parts, xoc); break;
      default: launch_nb<4>(plan, tX, n, beta, off, w, N, partial, PPpart, seed, epoch, idx0, mode, status, s, parts, xoc);
    }
    return;
  }
  const int g = grid_for(N, kBlock, 256 * 8);
  if (!(parts & 1)) {
  } else if (mode == W_DRAW)
    hipLaunchKernelGGL((k_psi_omega<W_DRAW>), dim3(g), dim3(kBlock), sizeof(double) * P, s, tX, n, beta, off, w, N, P,
                       seed, epoch, idx0, status);
  else
    hipLaunchKernelGGL((k_psi_omega<W_EM>), dim3(g), dim3(kBlock), sizeof(double) * P, s, tX, n, beta, off, w, N, P,
                       seed, epoch, idx0, status);
  if (!(parts & 2)) return;
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

void launch_mlogit_offset(const double* XB, int64_t N, int J, int j, double* c_out, double* eta_out, hipStream_t s)
{
  if (N <= 0) return;
  hipLaunchKernelGGL(k_mlogit_offset, dim3(grid_for(N, kBlock, 256 * 8)), dim3(kBlock), 0, s, XB, N, J, j, c_out, eta_out);
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
void launch_gather_stride(double* dst, const double* src, int64_t n, int stride, int off, hipStream_t s)
{
  if (n <= 0) return;
  int64_t g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(k_gather_stride, dim3((unsigned)g), dim3(256), 0, s, dst, src, n, stride, off);
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
