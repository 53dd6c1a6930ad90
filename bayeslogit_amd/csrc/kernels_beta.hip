// kernels_beta.hip -- the replicated P x P stage of a Gibbs sweep on MI355X: PP = P0 + X'Omega X, Cholesky, and the beta
// draw -- the unconstrained MVN draw (Code/C/Logit.hpp:291-320), the fork's active coordinate-wise constrained draw
// (Logit.hpp:322-400), Normal::set_from_likelihood + draw (include/Normal.hpp:98-131) and the EM solve
// (Logit.hpp:533-541).  One workgroup, redundantly on every rank from the (seed, sweep) stream.  gfx950 only.
// Built with machine-LICM off (bayeslogit_amd/build.py): these are latency-bound serial routines whose cold paths
// (erfc / inverse normal CDF of the tnorm fallback) otherwise park ~100 hoisted constants in registers and spill
// SGPRs through v_writelane in the move loops.
#include "bl_gibbs_kernels.hpp"
#include "bl_host.hpp"
#include <mutex>
#include "bl_pg_devroye.hpp"
#include "bl_pg1_queue.hpp"
#include "../../include/bayeslogit_hip.h"

namespace {

using namespace bl;

constexpr int kBlock = 256;

template <int CTRL>
__device__ __forceinline__ double dppmov_f64(double v)      // every lane has a valid source: no `old` copy
{
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// ================================================================ P x P stage
#define M_(M, i, j) ((M)[(size_t)(i) + (size_t)(j) * (size_t)P])

// ---- Blocked dense routines for P > 64: one workgroup of 512 threads (a 32 x 16 grid), matrices in global memory
// (they live in L2), panels of 32 rows.  A panel is first brought up to date with everything before it (a tiled
// product through LDS: no barrier per pivot, k runs inside a thread), then finished on chip: the 32 x 32 diagonal
// block by one wavefront in registers, the rest of the panel by one thread per column against that block in LDS.
// Every element receives the products of the textbook loops in the same order (k ascending for the factorisations
// and the forward solve, descending for the backward solve; one product subtracted at a time, then the division), so
// the results do not depend on the blocking.
constexpr int kPB = 32;             // panel height = depth of a k tile
constexpr int kCT = 128;            // columns of one update chunk
constexpr int kDenseThreads = 512;  // 2 waves / SIMD: 256 registers per lane (the panel routines keep 32-double columns in registers)
constexpr int kNY = kDenseThreads / 32, kCPT = kCT / kNY;   // thread grid 32 x kNY, kCPT columns of a chunk per thread

struct DenseLds {
  double A[kPB][kPB + 1];           // A[kk][ii]: the panel's own columns of the factor, rows of the k tile
  double B[kPB][kCT + 1];           // B[kk][cc]: the k tile's rows of the chunk's columns
  double D[kPB][kPB + 1];           // the diagonal block of the factor, D[i][j], i <= j (zero below, and past the matrix)
  double Dt[kPB][kPB + 1];          // its transpose (backward solves read the block by columns)
  int bad;
};

struct View {                       // element (r, c) at p[r sr + c sc]
  double* p;
  int sr, sc;
  __device__ __forceinline__ double& at(int r, int c) const { return p[(size_t)r * (size_t)sr + (size_t)c * (size_t)sc]; }
};

__device__ __forceinline__ double rl64(double v, int l)
{
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// C(i0 + i, c) -= sum_k A(k, i0 + i) B(k, c) over k in [k_lo, k_hi) (REV: from k_hi - 1 down), i < pb, c in [c_lo, c_hi);
// tri: only c >= i0 + i (the triangle of a factorisation).
template <bool REV>
__device__ __forceinline__ void panel_update(DenseLds& L, View A, View B, View C, int i0, int pb, int c_lo, int c_hi, int k_lo, int k_hi,
                             bool tri)
{
  const int t = threadIdx.x, tx = t & 31, ty = t >> 5;
  const int ntile = (k_hi - k_lo + kPB - 1) / kPB;
  for (int cc = c_lo; cc < c_hi; cc += kCT) {
    const int cn = c_hi - cc < kCT ? c_hi - cc : kCT;
    double acc[kCPT];
    bool own[kCPT];
#pragma unroll
    for (int m = 0; m < kCPT; ++m) {
      const int c = ty + kNY * m;
      own[m] = tx < pb && c < cn && !(tri && cc + c < i0 + tx);
      acc[m] = own[m] ? C.at(i0 + tx, cc + c) : 0.0;
    }
    // a tile's operands are fetched into registers while the tile before it is being used (an L2 round trip is ~2 us)
    constexpr int kQA = kPB * kPB / kDenseThreads, kQB = kPB * kCT / kDenseThreads;
    double ra[kQA], rb[kQB];
    auto tile_of = [&](int ti, int& k0, int& kt) {
      if (!REV) {
        k0 = k_lo + kPB * ti;
        kt = k_hi - k0 < kPB ? k_hi - k0 : kPB;
      } else {
        const int hi = k_hi - kPB * ti;
        k0 = hi - kPB > k_lo ? hi - kPB : k_lo;
        kt = hi - k0;
      }
    };
    auto fetch = [&](int ti) {
      int k0, kt;
      tile_of(ti, k0, kt);
#pragma unroll
      for (int q = 0; q < kQA; ++q) {
        const int e = t + kDenseThreads * q;
        const int kk = A.sr <= A.sc ? e & 31 : e >> 5, ii = A.sr <= A.sc ? e >> 5 : e & 31;
        ra[q] = (kk < kt && ii < pb) ? A.at(k0 + kk, i0 + ii) : 0.0;
      }
#pragma unroll
      for (int q = 0; q < kQB; ++q) {
        const int e = t + kDenseThreads * q;
        const int kk = B.sr <= B.sc ? e & 31 : e / kCT, c = B.sr <= B.sc ? e >> 5 : e & (kCT - 1);
        rb[q] = (kk < kt && c < cn) ? B.at(k0 + kk, cc + c) : 0.0;
      }
    };
    if (ntile > 0) fetch(0);
    for (int ti = 0; ti < ntile; ++ti) {
      int k0, kt;
      tile_of(ti, k0, kt);
      __syncthreads();                       // the tile before this one has been used
#pragma unroll
      for (int q = 0; q < kQA; ++q) {
        const int e = t + kDenseThreads * q;
        const int kk = A.sr <= A.sc ? e & 31 : e >> 5, ii = A.sr <= A.sc ? e >> 5 : e & 31;
        L.A[kk][ii] = ra[q];
      }
#pragma unroll
      for (int q = 0; q < kQB; ++q) {
        const int e = t + kDenseThreads * q;
        const int kk = B.sr <= B.sc ? e & 31 : e / kCT, c = B.sr <= B.sc ? e >> 5 : e & (kCT - 1);
        L.B[kk][c] = rb[q];
      }
      __syncthreads();
      if (ti + 1 < ntile) fetch(ti + 1);
      if (!REV) {
        for (int kk = 0; kk < kt; ++kk) {
          const double av = L.A[kk][tx];
#pragma unroll
          for (int m = 0; m < kCPT; ++m) acc[m] -= av * L.B[kk][ty + kNY * m];
        }
      } else {
        for (int kk = kt - 1; kk >= 0; --kk) {
          const double av = L.A[kk][tx];
#pragma unroll
          for (int m = 0; m < kCPT; ++m) acc[m] -= av * L.B[kk][ty + kNY * m];
        }
      }
    }
#pragma unroll
    for (int m = 0; m < kCPT; ++m)
      if (own[m]) C.at(i0 + tx, cc + ty + kNY * m) = acc[m];
  }
}

// The pb x pb diagonal block of a factorisation on one wavefront: lane j keeps column j (rows i <= j) in registers, the
// pivot row goes round by v_readlane.  u_kk = sqrt(a_kk), u_kj = a_kj / u_kk, a_ij -= u_ki u_kj: the reference's LAPACK
// order.  Leaves the block in T and in L.D; a non-positive pivot sets L.bad.
__device__ __forceinline__ void diag_chol(DenseLds& L, View T, int i0, int pb, int lane)
{
  double col[kPB];
  const bool act = lane < pb;
#pragma unroll
  for (int i = 0; i < kPB; ++i) col[i] = (act && i <= lane) ? T.at(i0 + i, i0 + lane) : 0.0;
  bool ok = true;
#pragma unroll
  for (int k = 0; k < kPB; ++k) {
    if (k < pb && ok) {
      const double akk = rl64(col[k], k);
      if (!(akk > 0.0)) {
        ok = false;
      } else {
        const double d = sqrt(akk);
        col[k] = lane == k ? d : col[k] / d;
#pragma unroll
        for (int i = k + 1; i < kPB; ++i) {
          const double uki = rl64(col[k], i);
          col[i] -= uki * col[k];
        }
      }
    }
  }
  if (!ok && lane == 0) L.bad = 1;
#pragma unroll
  for (int i = 0; i < kPB; ++i)
    if (act && i <= lane) {
      T.at(i0 + i, i0 + lane) = col[i];
      L.D[i][lane] = col[i];
    }
}

// Rows i0 .. i0 + pb - 1 of the columns [c_lo, c_hi) of B against the diagonal block, one thread per column (32 values
// in registers): forward (U' y = b: for i ascending, y_i = b_i / u_ii, then b_j -= u_ij y_i for j > i, from L.D's row i)
// or REV (U x = b: i descending, b_j -= u_ji x_i for j < i, from L.Dt's row i).  Column-oriented so that the updates of a
// step are independent; each element still receives its products in the order of the dot-product form.
template <bool REV>
__device__ __forceinline__ void panel_solve(const DenseLds& L, View B, int i0, int pb, int c_lo, int c_hi)
{
  for (int c = c_lo + (int)threadIdx.x; c < c_hi; c += kDenseThreads) {
    double v[kPB];
#pragma unroll
    for (int i = 0; i < kPB; ++i) v[i] = i < pb ? B.at(i0 + i, c) : 0.0;
    if (!REV) {
#pragma unroll
      for (int i = 0; i < kPB; ++i) {
        if (i < pb) {
          v[i] = v[i] / L.D[i][i];
#pragma unroll
          for (int j = i + 1; j < kPB; ++j) v[j] -= L.D[i][j] * v[i];     // rows >= pb: zeros, never stored
        }
        __builtin_amdgcn_sched_barrier(0);       // keep the LDS reads of later steps where they are: 32 steps of them do not fit the registers
      }
    } else {
#pragma unroll
      for (int i = kPB - 1; i >= 0; --i) {
        if (i < pb) {
          v[i] = v[i] / L.Dt[i][i];
#pragma unroll
          for (int j = i - 1; j >= 0; --j) v[j] -= L.Dt[i][j] * v[i];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int i = 0; i < kPB; ++i)
      if (i < pb) B.at(i0 + i, c) = v[i];
  }
}

// In-place factorisation of the symmetric matrix whose triangle element (i, j), i <= j, is T.at(i, j): T <- U, A = U'U.
// View{A, 1, P}: the upper triangle of a column-major matrix (chol 'U'); View{S, P, 1}: its lower triangle, S = L L'
// with L = U' (chol 'L').
__device__ __forceinline__ bool wg_chol(DenseLds& L, View T, int P)
{
  const int t = threadIdx.x;
  if (t == 0) L.bad = 0;
  __syncthreads();
  for (int i0 = 0; i0 < P; i0 += kPB) {
    const int pb = P - i0 < kPB ? P - i0 : kPB;
    if (i0 > 0) panel_update<false>(L, T, T, T, i0, pb, i0, P, 0, i0, true);
    __syncthreads();
    if (t < 64) diag_chol(L, T, i0, pb, t);
    __syncthreads();
    if (L.bad) return false;
    if (i0 + pb < P) panel_solve<false>(L, T, i0, pb, i0 + pb, P);
    __syncthreads();
  }
  return true;
}

// Columns [c_lo, c_hi) of B <- U'^{-1} B (forward) or, REV, U^{-1} B (backward); U.at(i, j), i <= j, the factor.
template <bool REV>
__device__ __forceinline__ void wg_trsm(DenseLds& L, View U, View B, int P, int c_lo, int c_hi)
{
  const int np = (P + kPB - 1) / kPB, t = threadIdx.x;
  for (int pi = 0; pi < np; ++pi) {
    const int i0 = (REV ? np - 1 - pi : pi) * kPB;
    const int pb = P - i0 < kPB ? P - i0 : kPB;
    if (!REV) {
      if (i0 > 0) panel_update<false>(L, U, B, B, i0, pb, c_lo, c_hi, 0, i0, false);
    } else {
      if (i0 + pb < P) panel_update<true>(L, View{U.p, U.sc, U.sr}, B, B, i0, pb, c_lo, c_hi, i0 + pb, P, false);
    }
    __syncthreads();
    for (int e = t; e < kPB * kPB; e += kDenseThreads) {
      const int i = e & 31, j = e >> 5;
      const double u = (i <= j && j < pb) ? U.at(i0 + i, i0 + j) : 0.0;
      if (!REV) L.D[i][j] = u;
      else L.Dt[j][i] = u;
    }
    __syncthreads();
    panel_solve<REV>(L, B, i0, pb, c_lo, c_hi);
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

__device__ void constrained_wide_records(const blk::BetaArgs& a, int t, int nthr);
__device__ void constrained_wide_reciprocals(const blk::BetaArgs& a, const double* __restrict__ Lg, double* Rg);

// P > 64, stage 1 (one workgroup): PP = P0 + X'Omega X, U = chol(PP); the EM solve and the unconstrained draw end here.
__global__ __launch_bounds__(kDenseThreads) void k_beta_factor(blk::BetaArgs a, int mode)
{
  __shared__ DenseLds L;
  // a Cholesky factorisation failed earlier in this chain (the chain's own flag, BetaArgs::dead: another handle's failure does not stop this one):
  // the chain is dead, and the stages behind this one would read a workspace nobody prepared
  if (*a.dead != 0) return;
  const int P = a.P, t = threadIdx.x;
  double* A = a.work;                      // PP, then U
  double* mP = a.work + 2 * (size_t)P * P; // posterior mean
  double* zz = mP + P;
  for (int e = t; e < P * P; e += kDenseThreads) A[e] = a.PPsum[e] + a.P0[e];   // PP = P0 + X'OmX
  __syncthreads();
  const View U{A, 1, P};
  if (!wg_chol(L, U, P)) {
    if (t == 0) (atomicOr(a.status, ST_NOT_PD), atomicOr(a.dead, 1));
    return;
  }
  if (mode != blk::B_SOLVE && mode != blk::B_MVN) return;

  for (int j = t; j < P; j += kDenseThreads) mP[j] = a.bP[j];
  if (mode == blk::B_MVN && t < P) {
    // eps_i = r.norm(0,1), i = 0..P-1 in stream order: normal i is exactly block i
    Stream r;
    r.init(a.seed, 0, DOM_BETA, a.epoch);
    for (int i = t; i < P; i += kDenseThreads) {
      r.blk = (uint32_t)i;
      r.has = false;
      zz[i] = r.norm(0.0, 1.0);
    }
  }
  __syncthreads();
  const View B{mP, 1, P};                  // columns: mP, eps
  wg_trsm<false>(L, U, B, P, 0, 1);
  wg_trsm<true>(L, U, B, P, 0, mode == blk::B_MVN ? 2 : 1);
  if (mode == blk::B_MVN) {
    for (int j = t; j < P; j += kDenseThreads) a.beta_out[j] = zz[j] + mP[j];
  } else {
    for (int j = t; j < P; j += kDenseThreads) a.beta_out[j] = mP[j];
  }
}

// Stage 2 (a workgroup per 64 columns): S = PP^{-1} by two triangular solves on the identity and, for the constrained
// draw, mP = PP^{-1} bP as column P of the same matrix (mP follows S in the workspace).  Columns are independent.
__global__ __launch_bounds__(kDenseThreads) void k_beta_inverse(blk::BetaArgs a, int nc, int cpw)
{
  __shared__ DenseLds L;
  if (*a.dead != 0) return;
  const int P = a.P, t = threadIdx.x;
  double* A = a.work;
  double* S = a.work + (size_t)P * P;
  const int c_lo = (int)blockIdx.x * cpw, c_hi = c_lo + cpw < nc ? c_lo + cpw : nc;
  for (int e = t; e < (c_hi - c_lo) * P; e += kDenseThreads) {
    const int i = e % P, c = c_lo + e / P;
    S[(size_t)c * P + i] = c < P ? (i == c ? 1.0 : 0.0) : a.bP[i];
  }
  __syncthreads();
  const View U{A, 1, P}, B{S, 1, P};
  wg_trsm<false>(L, U, B, P, c_lo, c_hi);
  wg_trsm<true>(L, U, B, P, c_lo, c_hi);
}

// The random input of the constrained sweeps for 64 < P <= 256 (independent of the matrices: any number of workgroups).
__global__ __launch_bounds__(256) void k_beta_records(blk::BetaArgs a)
{
  constrained_wide_records(a, (int)(blockIdx.x * blockDim.x + threadIdx.x), (int)(gridDim.x * blockDim.x));
}

// Stage 3 (one workgroup): L = chol(S, 'L') and what the draw needs of it.
__global__ __launch_bounds__(kDenseThreads) void k_beta_finish(blk::BetaArgs a, int mode)
{
  __shared__ DenseLds L;
  extern __shared__ double lds[];          // P > 256 constrained: beta, z (P each), perm
  if (*a.dead != 0) return;
  const int P = a.P, t = threadIdx.x;
  double* A = a.work;                      // U
  double* S = a.work + (size_t)P * P;      // PP^{-1}, then L
  double* mP = a.work + 2 * (size_t)P * P; // posterior mean
  double* zz = mP + P;

  if (mode == blk::B_FROM_LIK) {
    // mean = V b ; lower = chol(V,'L') ; beta = mean + lower eps   (Normal.hpp:98-131)
    for (int i = t; i < P; i += kDenseThreads) {
      double s = 0.0;
      for (int k2 = 0; k2 < P; ++k2) s += M_(S, i, k2) * a.bP[k2];
      mP[i] = s;
    }
    if (t < P) {
      Stream r;
      r.init(a.seed, 0, DOM_BETA, a.epoch);
      for (int i = t; i < P; i += kDenseThreads) {
        r.blk = (uint32_t)i;
        r.has = false;
        zz[i] = r.norm(0.0, 1.0);
      }
    }
    __syncthreads();
  }
  if (!wg_chol(L, View{S, P, 1}, P)) {     // L = chol(S,'L'), in place
    if (t == 0) (atomicOr(a.status, ST_NOT_PD), atomicOr(a.dead, 1));
    return;
  }
  for (int e = t; e < P * P; e += kDenseThreads) {
    const int i = e % P, j = e / P;
    if (i < j) S[e] = 0.0;
  }
  __syncthreads();
  if (mode == blk::B_FROM_LIK) {
    for (int i = t; i < P; i += kDenseThreads) {
      double le = 0.0;
      for (int k2 = 0; k2 <= i; ++k2) le += M_(S, i, k2) * zz[k2];
      a.beta_out[i] = le + mP[i];
    }
    return;
  }

  // ---- B_CONSTRAINED: Logit.hpp:322-400 ----  z = L^{-1}(beta_prev - mP)
  for (int j = t; j < P; j += kDenseThreads) zz[j] = a.beta_prev[j] - mP[j];
  __syncthreads();
  wg_trsm<false>(L, View{S, P, 1}, View{zz, 1, P}, P, 0, 1);
  if (P <= 256) {
    constrained_wide_reciprocals(a, S, A);  // U in A is dead: A takes 1/L.  The sweeps kernel follows.
    return;
  }
  // P > 256: LDS holds beta, z (P doubles each) and perm (P ints).  The vectors are exchanged between lanes of the
  // serial wave, which LDS orders and global memory does not.
  double* sbeta = lds;
  double* sz = sbeta + P;
  int* perm = reinterpret_cast<int*>(sz + P);
  const double* Lm = S;
  for (int j = t; j < P; j += (int)blockDim.x) {
    sbeta[j] = a.beta_prev[j];
    perm[j] = j;
  }
  __syncthreads();
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

constexpr int kRec = 20;   // doubles per pre-generated tnorm record: 4 attempts x (ua, log ua, log ub, normal) + (u8,0,0,0)

// ---- single-wavefront dense kernels (P <= 64) ----
// One wave needs no s_barrier: LDS operations of a wave execute in program order, so a
// wave-level scheduling fence between a phase's writes and the next phase's reads is enough.
// (Workgroup versions paid two or three barriers per column, ~1.4 us per column measured, and are gone.)  In the
// constrained draw these leave the other waves free to build the scan tables and solve for mP at the same time.
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

// ---- the dense routines of one wavefront, P <= 64, matrix rows in registers (round 3) ----
// What one wavefront alone on its SIMD pays (scripts/experiments/wave_issue_probe.hip, shader cycles): 4.7 per fp64 FMA,
// dependent or not; 14.5 - 20 per FMA whose operand comes by two v_readlane into a scalar pair; 114 for an LDS write and
// the read behind it; 7 to ISSUE a broadcast ds_read_b128 (3.5 a double: the reads of a wave do not overlap its own
// FMAs, they queue in front of them); 190 for a square root and a divide; 16 for an indexed register read; a taken branch
// tens.  So: the row of the matrix in registers, the pivot's multipliers through ONE LDS slot read back as broadcasts
// twelve reads deep, no branch tree, no branch at all inside the pivot loops.
//
// A register array is not addressable by a run-time index; a 16-double vector is (s_set_gpr_idx): four of them hold the row,
// the wave-uniform k picks the element (two moves against a 64-way branch tree's 340 cycles).
typedef double d16 __attribute__((ext_vector_type(16)));
// (four named vectors, neither an array nor a struct of them: only then does each stay a register tuple; and no branch
// inside a loop that updates them: at a join the compiler copies whole vectors.  Hence the PHASES below: the pivots of one
// quarter form one loop whose body is straight-line code over the quarters still live.)
template <int Q, typename F>
__device__ __forceinline__ void fill_quarter(d16& v, F&& f)
{
#pragma unroll
  for (int jj = 0; jj < 16; ++jj) v[jj] = f(16 * Q + jj);
}
#define BL_FILL_ROW(q0, q1, q2, q3, f) (fill_quarter<0>(q0, f), fill_quarter<1>(q1, f), fill_quarter<2>(q2, f), fill_quarter<3>(q3, f))
template <int Q>
__device__ __forceinline__ d16& quarter_of(d16& q0, d16& q1, d16& q2, d16& q3)
{
  if constexpr (Q == 0) return q0;
  else if constexpr (Q == 1) return q1;
  else if constexpr (Q == 2) return q2;
  else return q3;
}
// Broadcast reads of LDS with the order of issue spelled out.  Left to the scheduler, three reads are in flight and a read
// takes ~100 cycles to come back: 33 cycles per pair of columns against the 9.4 of its two FMAs (and the scheduling
// builtins serialised it altogether).  So: chunks of eight doubles (four reads), three chunks in flight -- twelve of the
// fifteen the counter can tell apart --, each waited for by count (LDS returns in order).
typedef double d2 __attribute__((ext_vector_type(2)));
struct Bc8 {
  d2 a, b, c, d;
};
template <bool ALIGNED, int OFF>      // OFF: in doubles from addr
__device__ __forceinline__ void bcast8_issue(Bc8& t, unsigned addr)
{
  if constexpr (ALIGNED)
    asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8"
                 : "=&v"(t.a), "=&v"(t.b), "=&v"(t.c), "=&v"(t.d)
                 : "v"(addr), "n"(8 * OFF), "n"(8 * OFF + 16), "n"(8 * OFF + 32), "n"(8 * OFF + 48)
                 : "memory");
  else
    asm volatile("ds_read2_b64 %0, %4 offset0:%5 offset1:%6\n\tds_read2_b64 %1, %4 offset0:%7 offset1:%8\n\t"
                 "ds_read2_b64 %2, %4 offset0:%9 offset1:%10\n\tds_read2_b64 %3, %4 offset0:%11 offset1:%12"
                 : "=&v"(t.a), "=&v"(t.b), "=&v"(t.c), "=&v"(t.d)
                 : "v"(addr), "n"(OFF), "n"(OFF + 1), "n"(OFF + 2), "n"(OFF + 3), "n"(OFF + 4), "n"(OFF + 5), "n"(OFF + 6), "n"(OFF + 7)
                 : "memory");
}
template <int LEFT>                   // wait until at most LEFT reads issued after t's are outstanding
__device__ __forceinline__ void bcast8_wait(Bc8& t)
{
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(t.a), "+v"(t.b), "+v"(t.c), "+v"(t.d) : "n"(LEFT));
}
// v[8 H .. 8 H + 7] -= t * s
template <int H>
__device__ __forceinline__ void axpy8(d16& v, const Bc8& t, double s)
{
  v[8 * H + 0] = fma(-s, t.a.x, v[8 * H + 0]);
  v[8 * H + 1] = fma(-s, t.a.y, v[8 * H + 1]);
  v[8 * H + 2] = fma(-s, t.b.x, v[8 * H + 2]);
  v[8 * H + 3] = fma(-s, t.b.y, v[8 * H + 3]);
  v[8 * H + 4] = fma(-s, t.c.x, v[8 * H + 4]);
  v[8 * H + 5] = fma(-s, t.c.y, v[8 * H + 5]);
  v[8 * H + 6] = fma(-s, t.d.x, v[8 * H + 6]);
  v[8 * H + 7] = fma(-s, t.d.y, v[8 * H + 7]);
}
// quarters QLO..QHI of the row -= m[16 QLO .. 16 QHI + 15] * s, m at LDS byte address addr (16-byte aligned if ALIGNED;
// the multiplier is rounded as fma(-s, m, v) == fma(-m, s, v))
template <int QLO, int QHI, bool ALIGNED>
__device__ __forceinline__ void axpy_quarters(d16& q0, d16& q1, d16& q2, d16& q3, unsigned addr, double s)
{
  constexpr int nc = 2 * (QHI - QLO + 1);          // chunks of eight
  Bc8 t[nc];
  bcast8_issue<ALIGNED, 16 * QLO>(t[0], addr);
  if constexpr (nc > 1) bcast8_issue<ALIGNED, 16 * QLO + 8>(t[1], addr);
  if constexpr (nc > 2) bcast8_issue<ALIGNED, 16 * QLO + 16>(t[2], addr);
#define BL_CHUNK(C)                                                                                          \
  if constexpr ((C) < nc) {                                                                                  \
    bcast8_wait<(nc - 1 - (C) >= 2 ? 8 : 4 * (nc - 1 - (C)))>(t[(C)]);                                        \
    if constexpr ((C) + 3 < nc) bcast8_issue<ALIGNED, 16 * QLO + 8 * ((C) + 3)>(t[(C) + 3 < nc ? (C) + 3 : 0], addr); \
    constexpr int q = QLO + (C) / 2;                                                                         \
    axpy8<(C) % 2>(quarter_of<(q < 4 ? q : 3)>(q0, q1, q2, q3), t[(C)], s);                                   \
  }
  BL_CHUNK(0) BL_CHUNK(1) BL_CHUNK(2) BL_CHUNK(3) BL_CHUNK(4) BL_CHUNK(5) BL_CHUNK(6) BL_CHUNK(7)
#undef BL_CHUNK
}
__device__ __forceinline__ unsigned lds_addr(const double* p) { return (unsigned)(size_t)p; }   // the LDS offset of a shared pointer

// Cholesky on one wavefront: lane i keeps row i of the LOWER triangle (a_ij, j <= i) in registers.  At pivot k lane i's own
// a_ik is the element it divides (u_i = a_ik / d: column k of the lower triangle is row k of the upper one), the u of all
// rows go to LDS once and come back as broadcasts, and the update is a_ij -= u_i u_j over the quarters from k's on.  Same
// arithmetic as round 2's w_chol_reg element for element (the same quotient, the same products subtracted in ascending k
// by the same fused multiply-add): the factor is bit-identical (round 2: 65 us at P = 64 with three LDS round trips a pivot;
// its LDS-resident predecessors 130; this one 26).  LOWER = false: M = U'U, reads the upper triangle, writes
// U into it; LOWER = true: M = L L', reads the lower triangle, writes L and zeroes the strict upper triangle.
// MIRROR (with LOWER = false): U' is written into the lower triangle as well, so that a ROW of U is contiguous too
// (w_inverse_rl).  buf: 64 doubles of LDS, 16-byte aligned.
// progress (optional, LDS): pivots finished, published behind each pivot's writes for a wave that follows this one
// (w_inverse_rl's forward solve needs row k of U and no more at its step k); kCholFailed when a pivot is not positive.
constexpr int kCholFailed = 1 << 20;
template <bool LOWER, bool MIRROR, int Q>
__device__ __forceinline__ bool chol_phase(d16& r0, d16& r1, d16& r2, d16& r3, double* M, int P, int ld, int lane, double* buf, int* progress)
{
  const int k_end = P < 16 * Q + 16 ? P : 16 * Q + 16;
  bool bad = false;
#pragma nounroll
  for (int k = 16 * Q; k < k_end; ++k) {
    const double c = quarter_of<Q>(r0, r1, r2, r3)[k & 15];
    const double akk = bcast_f64(c, k);
    bad |= !(akk > 0.0);                             // looked at behind the quarter's last pivot: no branch on the pivot's path
    const double d = sqrt(akk);
    double u = 0.0;
    if (lane > k && lane < P) u = c / d;
    buf[lane] = u;                                   // u_j = 0 for j <= k and outside the matrix: those columns do not move
    if (lane >= k && lane < P) {
      const double v = lane == k ? d : u;
      if (LOWER || MIRROR) L_(M, lane, k) = v;
      if (!LOWER) L_(M, k, lane) = v;
    }
    WAVE_SYNC();
    // (LDS executes a wavefront's operations in order: the word lands behind the pivot's writes without a wait)
    if (progress && lane == 0) __hip_atomic_store(progress, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    axpy_quarters<Q, 3, true>(r0, r1, r2, r3, lds_addr(buf), u);
    WAVE_SYNC();
  }
  if (bad) {                                         // (NaNs since the pivot that was not positive: nothing reads them as results)
    if (progress && lane == 0) __hip_atomic_store(progress, kCholFailed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return false;
  }
  return true;
}
template <bool LOWER, bool MIRROR = false>
__device__ __forceinline__ bool w_chol_rl(double* M, int P, int ld, int lane, double* buf, int* progress = nullptr)
{
  d16 r0, r1, r2, r3;
  auto load = [&](int j) {
    const bool in = lane < P && j <= lane;
    return in ? (LOWER ? L_(M, lane, j) : L_(M, j, lane)) : 0.0;
  };
  BL_FILL_ROW(r0, r1, r2, r3, load);
  if (!chol_phase<LOWER, MIRROR, 0>(r0, r1, r2, r3, M, P, ld, lane, buf, progress)) return false;
  if (!chol_phase<LOWER, MIRROR, 1>(r0, r1, r2, r3, M, P, ld, lane, buf, progress)) return false;
  if (!chol_phase<LOWER, MIRROR, 2>(r0, r1, r2, r3, M, P, ld, lane, buf, progress)) return false;
  if (!chol_phase<LOWER, MIRROR, 3>(r0, r1, r2, r3, M, P, ld, lane, buf, progress)) return false;
  if (LOWER) {
    for (int j = 1; j < P; ++j)
      if (lane < j && lane < P) L_(M, lane, j) = 0.0;
    WAVE_SYNC();
  }
  return true;
}

// S <- PP^{-1} given U and its mirror (PP = U'U): lane c solves U'y = e_c, then U x = y, on column c of S held in registers,
// both solves in the column-oriented form of the reference BLAS (DTRSM, left, upper: once an entry is solved its multiple is
// taken off every entry still open -- independent fused multiply-adds, where the dot-product form is one dependent chain per
// lane and an LDS round trip per eight terms).  Per element that is the order of round 2's routine in the forward solve
// (ascending) and the reverse of it in the backward solve (descending, as DTRSM has it).  Column i of the mirrored factor
// holds what both need: U_mi above the diagonal (backward), U_im below it (forward).  Entries already solved are parked in S
// and their registers are dead (a phase runs over whole quarters: whatever it leaves in a dead entry is never read).
template <int Q>
__device__ __forceinline__ bool inverse_fwd_phase(d16& y0, d16& y1, d16& y2, d16& y3, const double* U, double* col, int P, int ld, int lane,
                                                  const int* progress)
{
  const int i_end = P < 16 * Q + 16 ? P : 16 * Q + 16;
#pragma nounroll
  for (int i = 16 * Q; i < i_end; ++i) {
    if (progress) {                       // the factorisation runs on another wavefront: step i needs its pivots 0 .. i
      int done;
      while ((done = __hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) <= i) __builtin_amdgcn_s_sleep(2);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (done >= kCholFailed) return false;
    }
    const double* ui = U + i * ld;
    const double v = quarter_of<Q>(y0, y1, y2, y3)[i & 15] / ui[i];
    if (lane < P) col[i] = v;
    axpy_quarters<Q, 3, false>(y0, y1, y2, y3, lds_addr(ui), v);
  }
  return true;
}
template <int Q>
__device__ __forceinline__ void inverse_bwd_phase(d16& y0, d16& y1, d16& y2, d16& y3, const double* U, double* col, int P, int ld, int lane)
{
#pragma nounroll
  for (int i = (P < 16 * Q + 16 ? P : 16 * Q + 16) - 1; i >= 16 * Q; --i) {
    const double* ui = U + i * ld;
    const double v = quarter_of<Q>(y0, y1, y2, y3)[i & 15] / ui[i];
    if (lane < P) col[i] = v;
    axpy_quarters<0, Q, false>(y0, y1, y2, y3, lds_addr(ui), v);
  }
}
// progress (optional): see chol_phase -- the forward solve then runs a step behind the factorisation of another wavefront and
// costs no time of its own; false: that factorisation failed.
__device__ __forceinline__ bool w_inverse_rl(const double* U, double* S, int P, int ld, int lane, const int* progress = nullptr)
{
  const int c = lane < P ? lane : 0;
  double* col = S + c * ld;
  d16 y0, y1, y2, y3;
  auto unit = [&](int j) { return j == lane ? 1.0 : 0.0; };
  BL_FILL_ROW(y0, y1, y2, y3, unit);
  if (!inverse_fwd_phase<0>(y0, y1, y2, y3, U, col, P, ld, lane, progress)) return false;     // forward: U' y = e_c
  if (!inverse_fwd_phase<1>(y0, y1, y2, y3, U, col, P, ld, lane, progress)) return false;
  if (!inverse_fwd_phase<2>(y0, y1, y2, y3, U, col, P, ld, lane, progress)) return false;
  if (!inverse_fwd_phase<3>(y0, y1, y2, y3, U, col, P, ld, lane, progress)) return false;
  WAVE_SYNC();
  auto parked = [&](int j) { return j < P ? col[j] : 0.0; };
  BL_FILL_ROW(y0, y1, y2, y3, parked);
  inverse_bwd_phase<3>(y0, y1, y2, y3, U, col, P, ld, lane);                  // backward: U x = y
  inverse_bwd_phase<2>(y0, y1, y2, y3, U, col, P, ld, lane);
  inverse_bwd_phase<1>(y0, y1, y2, y3, U, col, P, ld, lane);
  inverse_bwd_phase<0>(y0, y1, y2, y3, U, col, P, ld, lane);
  WAVE_SYNC();
  return true;
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

// the cheap pass of a scan of P = 64 moves on the four wavefronts, 16 moves each (defined with k_beta_sweeps_run's pieces, below)
template <int HB0, int HB1>
__device__ __forceinline__ uint32_t quad_pass(const double* S, int ld, int lane, int cvec, double svec, double z1v, double* zw,
                                              double bj, double& bs_out, double (&cur)[64], double (&cp)[8], const double* Rk);
// ... and what is left of such a scan when some move needs its bounds, on the wavefront that walked the whole chain
__device__ __forceinline__ int quad_slow(const double (&cur)[64], const double (&cp)[8], double bs_end, uint32_t Fb0, int lane,
                                         int cvec, double svec, double z1v, const double* Rk, double* zw, double* zk, double* zz,
                                         double& bj);
// one scan of the constrained sweeps on one wavefront (defined with k_beta_sweeps_run's pieces, below)
__device__ __forceinline__ int solo_scan(const double* S, int ld, int P, int lane, int cvec, double svec, double z1v,
                                         const double* Rk, double* zw, double* zk, double* zz, double& bj);

// The tnorm records of a constrained draw at P <= 64: record e = (scan k, move i) from its nine uniforms (per scan k, P-1
// r.flat for the shuffle, then P tnorm calls of 9 uniforms; a call owns its nine whatever its bounds), one thread a record.
// They depend on (seed, epoch, P) alone.
__global__ __launch_bounds__(256) void k_beta64_records(blk::BetaArgs a)
{
  if (*a.dead != 0) return;
  const int P = a.P, e = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (e >= P * P) return;
  const uint32_t per_scan = (uint32_t)(10 * P - 1);
  const int k = e / P, i = e % P;
  const uint32_t base = (uint32_t)k * per_scan + (uint32_t)(P - 1) + 9u * (uint32_t)i;
  double* R = a.work + (size_t)e * kRec;
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

__global__ __launch_bounds__(kBlock) void k_beta64(blk::BetaArgs a, int mode, int quad_ok)
{
  extern __shared__ double lds[];
  // a Cholesky factorisation failed earlier in this chain (the chain's own flag, BetaArgs::dead: another handle's failure does not stop this one):
  // the chain is dead, and for 64 < P <= 256 k_beta has not prepared the workspace k_beta_sweeps reads
  if (*a.dead != 0) return;
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
  __shared__ __attribute__((aligned(16))) double s_piv[64];   // w_chol_rl's multipliers
  __shared__ int bad;
  __shared__ int progress;                           // wavefront 0's pivots done (kCholFailed: not positive definite)
  __shared__ __attribute__((aligned(16))) double s_piv2[64];  // the multipliers of wavefront 2's Cholesky
  __shared__ int progress2;                          // wavefront 2's pivots of chol(S) done
  if (t == 0) bad = 0;
  if (t == 0) progress = 0;
  if (t == 0) progress2 = 0;
  if (a.dbg && t == 0) a.dbg[0] = wall_clock64();
  for (int e = t; e < P * P; e += kBlock) {
    const int i = e % P, j = e / P;
    L_(A, i, j) = a.PPsum[e] + a.P0[e];              // PP = P0 + X'OmX
  }
  if (t < P) L_(A, P, t) = 0.0;                      // the padding row of every column (read into dead entries only)
  __syncthreads();

  if (mode == blk::B_SOLVE || mode == blk::B_MVN) {
    // one wavefront, no workgroup barriers: the register-resident factorisation and the vector solves (75 us against 136
    // for the workgroup routines with their two or three barriers per pivot; same arithmetic, same bits)
    if (t < 64) {
      const int lane = t;
      if (!w_chol_rl<false, true>(A, P, ld, lane, s_piv)) {
        if (lane == 0) (atomicOr(a.status, ST_NOT_PD), atomicOr(a.dead, 1));
      } else {
        double m = lane < P ? a.bP[lane] : 0.0;
        m = w_solve_Ut_vec(A, m, P, ld, lane);
        m = w_solve_U_vec(A, m, P, ld, lane);
        if (mode == blk::B_MVN) {
          // eps_i = r.norm(0,1) in stream order: normal i is exactly Philox block i     (Logit.hpp:311)
          double e = 0.0;
          if (lane < P) {
            const double u1 = beta_stream_unif(a.seed, a.epoch, 2 * lane), u2 = beta_stream_unif(a.seed, a.epoch, 2 * lane + 1);
            e = sqrt(-2.0 * log(u1)) * cospi(2.0 * u2);
          }
          e = w_solve_U_vec(A, e, P, ld, lane);
          if (lane < P) a.beta_out[lane] = e + m;
        } else if (lane < P) {
          a.beta_out[lane] = m;
        }
      }
    }
    return;
  }

  // ================= the dense stage, no workgroup barriers =================
  // wavefront 0: U = chol(PP,'U'), one pivot after the other, each published; then mP (constrained draw)
  // wavefront 2: S = PP^{-1} -- the forward solve a step behind wavefront 0's pivots (its step i needs row i of U), the
  //              backward solve when U is whole -- and L = chol(S,'L')
  // wavefront 1: the scan tables (constrained draw)
  if (t < 64) {
    const int lane = t;
    const bool ok = w_chol_rl<false, true>(A, P, ld, lane, s_piv, &progress);  // U = chol(PP,'U'), and U' below it
    if (a.dbg && t == 0) a.dbg[3] = wall_clock64();
    if (!ok && t == 0) bad = 1;
    if (ok && mode == blk::B_CONSTRAINED) {
      // mP = U^{-1} U^{-T} bP (Logit.hpp:335-340)
      double m = lane < P ? a.bP[lane] : 0.0;
      m = w_solve_Ut_vec(A, m, P, ld, lane);
      m = w_solve_U_vec(A, m, P, ld, lane);
      // z = L^{-1}(beta_prev - mP) (Logit.hpp:359-365), a step behind wavefront 2's pivots of chol(S) (step i needs column i
      // of L), and beside it 1/L where L < 0 into A (U is dead: this wavefront and wavefront 2's inverse were its last
      // readers) -- see the table of wavefront 1 below
      double z = lane < P ? a.beta_prev[lane] - m : 0.0;
      const double nan = __builtin_nan("");
      bool live = true;
      for (int i = 0; i < P && live; ++i) {
        int done;
        while ((done = __hip_atomic_load(&progress2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) <= i) __builtin_amdgcn_s_sleep(2);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (done >= kCholFailed) {
          live = false;
          break;
        }
        const double l = (lane >= i && lane < P) ? L_(S, lane, i) : 0.0;
        const double bi = bcast_f64(z, i) / bcast_f64(l, i);
        if (lane == i) z = bi;
        if (lane > i) z -= l * bi;                       // (l = 0 outside the matrix)
        if (lane < P) L_(A, lane, i) = (l < 0.0 && lane < P - 1) ? 1.0 / l : nan;
      }
      if (live && lane < P) zz[lane] = z;
    }
  } else if (t >= 128 && t < 192) {
    const int lane = t - 128;
    bool ok = w_inverse_rl(A, S, P, ld, lane, &progress);                      // S = PP^{-1}
    if (a.dbg && lane == 0) a.dbg[4] = wall_clock64();
    if (ok) {
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
        ok = w_chol_rl<true>(S, P, ld, lane, s_piv2);
        if (ok && lane < P) {
          double le = 0.0;
          for (int k2 = 0; k2 <= lane; ++k2) le += L_(S, lane, k2) * zz[k2];
          a.beta_out[lane] = le + mean;
        }
      } else {
        // B_CONSTRAINED set-up, Logit.hpp:335-366 (z: after the barrier, when mP is there)
        if (a.dbg && lane == 0) a.dbg[1] = wall_clock64();
        ok = w_chol_rl<true>(S, P, ld, lane, s_piv2, &progress2);             // L = chol(S,'L')
        if (a.dbg && lane == 0) a.dbg[2] = wall_clock64();
      }
      if (!ok && lane == 0) bad = 1;
    }
    // (a factorisation that failed: the wavefronts that follow this one's must not wait for pivots that will not come)
    if (!ok && lane == 0) __hip_atomic_store(&progress2, kCholFailed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  } else if (mode == blk::B_CONSTRAINED && t >= 64 && t < 128) {
    // ====== wave 1, meanwhile: the scan tables.  (The tnorm records -- every other random input of the draw, in stream
    // order: DESIGN.md section 2 -- come from k_beta64_records, launched in front of this kernel: three wavefronts took 118 us
    // over them here, longer than the dense stage beside them takes since round 3.) ======
    const int tt = t - 64;
    const uint32_t per_scan = (uint32_t)(10 * P - 1);
    // Scan k's P-1 swaps (r.flat(i, P), Logit.hpp:375-377) applied to the identity, all scans in parallel (lane k, its row
    // of ptab as scratch), then composed in scan order: `is` persists across scans (Logit.hpp:368-377); the composition is in
    // place, row by row.  One wavefront: no workgroup barrier.
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
    if (a.dbg && t == 64) a.dbg[11] = wall_clock64();
    // 1/L split by the sign test of Logit.hpp:384-391 (see constrained_wide_prepare): Ri where L > 0 (here), A where L < 0
    // (wavefront 0), NaN elsewhere -- v_max/v_min and the compare masks ignore NaN.  Column by column behind the pivots of
    // chol(S), as wavefront 0's z.
    const double nan = __builtin_nan("");
    for (int i = 0; i < P; ++i) {
      int done;
      while ((done = __hip_atomic_load(&progress2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) <= i) __builtin_amdgcn_s_sleep(2);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (done >= kCholFailed) break;
      const double l = (tt >= i && tt < P) ? L_(S, tt, i) : 0.0;
      if (tt < P) L_(Ri, tt, i) = (l > 0.0 && tt < P - 1) ? 1.0 / l : nan;
    }
  }
  if (a.dbg && t == 0) a.dbg[5] = wall_clock64();
  __syncthreads();
  if (bad) {
    if (t == 0) (atomicOr(a.status, ST_NOT_PD), atomicOr(a.dead, 1));
    return;
  }
  if (mode != blk::B_CONSTRAINED) return;


  if (a.dbg && t == 0) a.dbg[7] = wall_clock64();
  if (a.dbg && t == 0) a.dbg[9] = clock64();
  // The coordinate sweeps: a move reads only LDS and registers.
  //
  // Fast path.  A move's value is almost always the Box-Muller normal s of its first attempt: the bounds
  // contain 0, are wider than sqrt(2 pi), and s falls inside (tnorm_lanes' first branch, attempt 0).  That
  // is decided WITHOUT reducing the bounds: if every lane's lower candidate is <= min(s, -a) and every upper
  // candidate >= max(s, b) for some a, b >= 0 with a + b > sqrt(2 pi), then lo <= 0 <= hi, hi - lo > sqrt(2 pi)
  // and lo <= s <= hi -- three __ballot tests ((a, b) = (1.26, 1.26), (0, 2.51), (2.51, 0): one-sided bounds,
  // the usual case, pass the second or third whatever their finite side is).  Only a move that fails all
  // three pays the 64-lane max/min and tnorm_lanes (0.3 % of the moves on C4).  Same values either way.
  //
  // One speculative segment per scan: every move of the scan is assumed to take attempt 0's normal s, so dz = s - z_c is known
  // for the whole scan up front; the chain beta_j + L_jc dz runs move by move in registers with the first of the three tests
  // above in feasibility form beside it (three FMAs and an OR per move, sign bits per half-block of 8), a failing half-block
  // goes through the three tests, a move that fails those is redone with its bounds and everything behind it is taken again.
  // P = 64: the cheap pass on the four wavefronts (quad_pass), the rest on the one that walked the whole chain (quad_slow);
  // P < 64: all of it on wavefront 0 (solo_scan), beta reaching the others through LDS at the scan's end.  A chain pressed
  // against its bounds gains nothing from that: its scans go move by move on every wavefront alike (same inputs, same
  // arithmetic: the replicas of beta stay identical without an exchange).  Same values as the move-by-move loop either way.
  // z lives in LDS (zz): within a scan every coordinate is visited once, so the z_c of all of a scan's
  // moves are gathered at its start (z1v) and the new values are scattered at its end.
  const int lane = t & 63, wave = t >> 6;
  const bool row = lane < P;
  double bj = row ? a.beta_prev[lane] : 0.0;       // beta_j, replicated in every wavefront
  const double inf = __builtin_huge_val();
  const int nrec = P * kRec;
  __shared__ __attribute__((aligned(16))) double s_zw[4][128], s_zk[64], s_bj[2][64];  // the scans' hand-over slots; beta and the count left for the other waves
  __shared__ int s_nf[2];                                                            // (by scan parity: one barrier per scan)
  __shared__ __attribute__((aligned(16))) uint32_t s_fb[2][4];                       // quad_pass: the four verdicts, by scan parity
  // the cheap pass on the four wavefronts (a full matrix: every lane a row and a move); bl_diag_beta_sweeps(0): on one, as for P < 64
  const bool quad = P == 64 && quad_ok != 0;
  // The records stay in global memory (L2: k_beta64_records wrote them just now) while the scans speculate -- a scan then needs
  // one number of each, the first normal, fetched a scan ahead; a move redone with its bounds reads its record from there
  // (4 moves in 4096 on C4).  A chain pressed against its bounds reads four numbers a move: its scans stage their records
  // into LDS first.
  bool spec_on = true;
  double svec_n = row ? rec[lane * kRec + 3] : 0.0;                 // lane i: attempt 0's normal of move i of scan 0
  int cvec_n = row ? ptab[lane] : 0;                                 //         coordinate of move i of scan 0
  for (int k = 0; k < P; ++k) {
    const double* Rk = rec + (size_t)k * nrec;
    const double svec = svec_n;
    const int cvec = cvec_n;                                         // (both a scan ahead: the scan's first LDS read is z's)
    if (k + 1 < P) svec_n = row ? rec[(size_t)(k + 1) * nrec + lane * kRec + 3] : 0.0;
    if (k + 1 < P) cvec_n = row ? ptab[(k + 1) * P + lane] : 0;
    if (!spec_on) {
      for (int e = t; e < nrec; e += kBlock) recL[e] = Rk[e];
      __syncthreads();
      Rk = recL;
    }
    const int g4 = (lane < 5 ? lane : 0) * 4;
    const double qnan = __builtin_nan("");
    const double z1v = row ? zz[cvec] : 0.0;                     // lane i: z_c before move i
    int nfail = 0;                                                 // moves taken with their exact bounds in this scan
    if (spec_on && quad) {
      // the scan's cheap pass on the four wavefronts, 16 moves each; if every move takes its first normal (15 scans in 16
      // on C4) that was the scan
      double bs, cur[64], cp[8];
      uint32_t vb;
      switch (wave) {
        // (a wave pays 36 cycles for a move it only walks -- its column of L, its term of the chain -- and 77 for one it
        // tests: 32 / 16 / 8 / 8 moves even the four out)
        case 0: vb = quad_pass<0, 4>(S, ld, lane, cvec, svec, z1v, s_zw[0], bj, bs, cur, cp, Rk); break;
        case 1: vb = quad_pass<4, 6>(S, ld, lane, cvec, svec, z1v, s_zw[1], bj, bs, cur, cp, Rk); break;
        case 2: vb = quad_pass<6, 7>(S, ld, lane, cvec, svec, z1v, s_zw[2], bj, bs, cur, cp, Rk); break;
        default: vb = quad_pass<7, 8>(S, ld, lane, cvec, svec, z1v, s_zw[3], bj, bs, cur, cp, Rk); break;
      }
      if (lane == 0) s_fb[k & 1][wave] = vb;
      if (wave == 3) s_bj[k & 1][lane] = bs;
      __syncthreads();
      const uint4 fb = *reinterpret_cast<const uint4*>(s_fb[k & 1]);
      const uint32_t Fb = __builtin_amdgcn_readfirstlane((int)(fb.x | fb.y | fb.z | fb.w));
      if (Fb == 0u) {
        // The one barrier of such a scan.  Every wavefront writes the scan's z (the same values to the same words: each reads
        // its own later), the slots go by the scan's parity, and nobody can be two scans ahead of anybody.
        bj = s_bj[k & 1][lane];
        zz[cvec] = svec;
        continue;
      }
      // some move needs its bounds: the wavefront that holds the whole scan finishes it (solo_scan's second part), the others wait
      if (wave == 3) {
        const long long tq0 = a.dbg ? clock64() : 0;
        nfail = quad_slow(cur, cp, bs, Fb, lane, cvec, svec, z1v, Rk, s_zw[3], s_zk, zz, bj);
        if (a.dbg && lane == 0) a.dbg[14] += (unsigned long long)(clock64() - tq0);
        s_bj[k & 1][lane] = bj;
        if (lane == 0) s_nf[k & 1] = nfail;
      }
    } else
    if (spec_on) {
      // the scan as one speculative segment on wavefront 0 (the others wait)
      if (wave == 0) {
        const long long tq0 = a.dbg ? clock64() : 0;
        nfail = solo_scan(S, ld, P, lane, cvec, svec, z1v, Rk, s_zw[0], s_zk, zz, bj);
        if (a.dbg && t == 0) a.dbg[14] += (unsigned long long)(clock64() - tq0);
        s_bj[k & 1][lane] = bj;
        if (lane == 0) s_nf[k & 1] = nfail;
      }
    } else {
      // a chain pressed against its bounds: move by move, every wavefront alike (same inputs, same arithmetic: the replicas
      // of beta stay identical without an exchange)
      const int lr = row ? lane : 0;                               // lanes outside the matrix read row 0, masked below
      for (int i = 0; i < P; ++i) {
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
          ++nfail;
        }
        const double dz = z2 - z1;
        bj += l1 * dz;                 // L(j, c) = 0 for j < c and l1 = 0 outside the matrix: those rows do not move
        if (wave == 0 && lane == 0) zz[c] = z2;
      }
    }
    __syncthreads();
    const bool was_solo = spec_on;
    if (was_solo) nfail = s_nf[k & 1];   // wavefront 0's scan: its count to every wavefront
    // a chain pressed against its bounds gains nothing from speculating: move by move then, look again every 8th scan
    spec_on = 2 * nfail < P || ((k + 1) & 7) == 0;
    if (was_solo && (quad || !spec_on)) bj = s_bj[k & 1][lane];   // the replicas are needed again: the beta of the wavefront that made the scan
    if (a.dbg && t == 0) a.dbg[8] += (unsigned long long)nfail;
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
__device__ void constrained_wide_records(const blk::BetaArgs& a, int t, int nthr)
{
  const int P = a.P;
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
}

__device__ void constrained_wide_reciprocals(const blk::BetaArgs& a, const double* __restrict__ Lg, double* Rg)
{
  const int P = a.P, t = threadIdx.x, nthr = (int)blockDim.x;
  double* rec = a.work + 2 * (size_t)P * P + 2 * (size_t)P;
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

// Two words per chain behind the scan tables of k_beta_sweeps_run (zero when the chain is created): [0] = 1 while the chain
// is pressed against its bounds (most moves need them: the one-wavefront sweeps below, which have no barrier per move,
// are then twice as fast as the row-split ones), [1] = k_beta_sweeps_run's verdict on the draw it has just made.
__device__ __forceinline__ uint32_t* beta_mode_word(const blk::BetaArgs& a)
{
  const size_t P = (size_t)a.P;
  double* rec = a.work + 2 * P * P + 2 * P;
  return reinterpret_cast<uint32_t*>(rec + P * P * kRec + (P * P + 1) / 2 + P * P) + P * P + 128;
}

// The sweeps themselves: one 4-wave workgroup (512 registers per lane available: the pipeline's register
// sets do not spill), launched behind k_beta on the same stream.  Reads L, 1/L, z from k_beta's scratch.
template <int RPL>
__global__ __launch_bounds__(kBlock) void k_beta_sweeps(blk::BetaArgs a, bool paired)
{
  extern __shared__ double lds[];
  // a Cholesky factorisation failed earlier in this chain (the chain's own flag, BetaArgs::dead: another handle's failure does not stop this one):
  // the chain is dead, and for 64 < P <= 256 k_beta has not prepared the workspace k_beta_sweeps reads
  if (*a.dead != 0) return;
  const int P = a.P, t = threadIdx.x, nthr = (int)blockDim.x;
  // paired: launched behind k_beta_sweeps_run (launch_beta); the chain's mode word says which of the two takes this draw
  uint32_t* mode = paired ? beta_mode_word(a) : nullptr;
  const uint32_t pressed = paired ? mode[0] : 1u;
  __syncthreads();                           // every thread has read the word before thread 0 rewrites it
  if (pressed == 0u) {                       // k_beta_sweeps_run has made the draw and left its verdict in mode[1]
    if (t == 0) mode[0] = mode[1];
    return;
  }
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
  int nbound = 0;              // moves that needed their bounds (wave-uniform)
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
              ++nbound;
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
    // still pressed against its bounds?  (a sixth of the moves needing them; k_beta_sweeps_run hands over at a third)
    if (paired && lane == 0) mode[0] = (6 * nbound >= P * P) ? 1u : 0u;
  }
}

// OR of v over the 64 lanes, returned wave-uniform (same DPP ladder as wave_maxmin)
__device__ __forceinline__ uint32_t wave_or_u32(uint32_t v)
{
  int x = (int)v;
  x |= __builtin_amdgcn_mov_dpp(x, 0xB1, 0xf, 0xf, true);      // quad_perm [1,0,3,2]
  x |= __builtin_amdgcn_mov_dpp(x, 0x4E, 0xf, 0xf, true);      // quad_perm [2,3,0,1]
  x |= __builtin_amdgcn_mov_dpp(x, 0x141, 0xf, 0xf, true);     // row_half_mirror
  x |= __builtin_amdgcn_mov_dpp(x, 0x140, 0xf, 0xf, true);     // row_mirror
  x |= __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);   // row_bcast15 -> rows 1, 3
  x |= __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);   // row_bcast31 -> rows 2, 3
  return (uint32_t)__builtin_amdgcn_readlane(x, 63);
}

// ---- the same sweeps with the ROWS split over the wavefronts: wave w owns rows 64 w + lane (one row per lane, beta_j in
// a register); k_beta_sweeps_run below.  split_exact is one move taken with its exact bounds by the four wavefronts
// together (reciprocal columns, DPP max/min per wavefront, four results through LDS, the tnorm record's attempts evaluated
// redundantly by every wavefront): what a scan of a chain pressed against its bounds runs move by move.
__device__ __forceinline__ void split_exact(const double* __restrict__ Lg, const double* __restrict__ Rg,
                                            const double* __restrict__ Rh, const double* __restrict__ RkSeg, int P, int jr,
                                            int lane, int wave, int nq, int i, int cq, double z1q, double& bj, double* sz,
                                            double* xl, unsigned& par, int g4)
{
  const int c = __builtin_amdgcn_readlane(cq, i);
  const size_t off = (size_t)c * P + jr;
  const double l1 = Lg[off], rl = Rg[off], rh = Rh[off];
  const double z1 = readlane_f64(z1q, i);
  const double* Rn = RkSeg + (size_t)i * kRec + g4;
  const double r0 = Rn[0], r1 = Rn[1], r2 = Rn[2], r3 = Rn[3];
  double lo = z1 - bj * rl, hi = z1 - bj * rh;       // NaN: this row does not bound the move on that side
  wave_maxmin(lo, hi);
  double* sl = xl + par * 8;
  if (lane == 0) {
    sl[wave * 2] = lo;
    sl[wave * 2 + 1] = hi;
  }
  __syncthreads();
  double glo = -__builtin_huge_val(), ghi = __builtin_huge_val();
  for (int w = 0; w < nq; ++w) {
    glo = vmax64(glo, sl[w * 2]);            // v_max_f64 / v_min_f64 return the other operand for a NaN
    ghi = vmin64(ghi, sl[w * 2 + 1]);
  }
  par ^= 1u;
  const double z2 = tnorm_lanes(r0, r1, r2, r3, lane, glo, ghi);
  bj += l1 * (z2 - z1);
  if (wave == 0 && lane == 0) sz[c] = z2;
}

// ---- the row-split sweeps: the P^2 moves of a draw in speculative SEGMENTS of 64 (assuming every move takes its first
// Box-Muller normal s, dz = s - z_c is known for the whole segment up front, because a scan visits every coordinate once:
// each wavefront walks the segment on its own rows and the four meet once per segment), with little but the dependent chain
// left in the walk:
//   * the scan tables (the coordinate of move i of scan k) come from k_beta_scan_tables in global memory; a lane holds the
//     column offsets of the segment after next (loaded two segments ahead), so fetching a move's column of L is one
//     v_readlane and one buffer load (row offset in a register, column offset as the scalar offset);
//   * the two column sets (this segment's, the next one's) swap roles from segment to segment: no register copies;
//   * the cheap test is k_beta64's test (1.26, 1.26) in feasibility form: [-1.26, 1.26] and s feasible for the row -- beta_j -
//     L z_c - 1.26 |L| >= 0 covers both ends at once, and beta_j after the move is the value the chain needs anyway --
//     hence an interval wider than sqrt(2 pi) around 0 with s inside, tnorm's first branch, z' = s: three FMAs and an OR
//     per row and move) runs on whole blocks of 16 with the sign bits OR-ed per block; (dz, z_c) reach the wave as uniform
//     operands through the wave's own LDS slot (broadcast reads, half a block ahead of the arithmetic); the four
//     wavefronts meet once per segment (one word each);
//   * a half-block that fails (or is cut by the end of the scan) runs k_beta64's three sufficient tests in their
//     feasibility form (beta_j + L_jc (p - z_c) >= 0 at p in {s, +-1.26, 0, +-2.51}: no reciprocals) from the chain value
//     the cheap pass left at its start; a move that fails those is redone with its exact bounds (split_exact), the rest of
//     its block goes through the three tests again and the blocks behind it through a new cheap pass.
// Same decisions as the one-wavefront kernel up to the rounding of the tests' left-hand sides; same arithmetic for beta_j.
template <int H>      // H: half-block of 8 moves
__device__ __forceinline__ void run_load(const double* zw, double (&dz)[8], double (&z1)[8])
{
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const double2 v = *reinterpret_cast<const double2*>(zw + 2 * (8 * H + u));
    dz[u] = v.x;
    z1[u] = v.y;
  }
}
template <int H>
__device__ __forceinline__ void run_calc(const double (&l1)[64], const double (&dz)[8], const double (&z1)[8], double& bs,
                                         uint32_t& acc)
{
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const double l = l1[8 * H + u];
    const double g = fma(-l, z1[u], bs);        // beta_j + L (p - z_c) at p = 0
    double e;                                   // ... at the worse end of [-1.26, 1.26] (by hand: the compiler would keep
    asm("v_fma_f64 %0, -|%1|, %2, %3" : "=v"(e) : "v"(l), "s"(1.26), "v"(g));   // |l| of the whole segment in registers)
    bs = fma(l, dz[u], bs);                     // ... at p = s: beta_j after the move
    acc |= (uint32_t)__double2hiint(e) | (uint32_t)__double2hiint(bs);
  }
}

// the same with the test's half-width per move (1.26, or 0 for a move that is not to be tested: with dz = z_c = 0 it then
// changes nothing and passes): the cheap pass taken again behind a move redone exactly, whose half-block is partly done
template <int H>
__device__ __forceinline__ void run_load_k(const double* zk, double (&kk)[8])
{
#pragma unroll
  for (int u = 0; u < 8; u += 2) {
    const double2 v = *reinterpret_cast<const double2*>(zk + 8 * H + u);
    kk[u] = v.x;
    kk[u + 1] = v.y;
  }
}
template <int H>
__device__ __forceinline__ void run_calc_k(const double (&l1)[64], const double (&dz)[8], const double (&z1)[8],
                                           const double (&kk)[8], double& bs, uint32_t& acc)
{
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const double l = l1[8 * H + u];
    const double g = fma(-l, z1[u], bs);
    double e;
    asm("v_fma_f64 %0, -|%1|, %2, %3" : "=v"(e) : "v"(l), "v"(kk[u]), "v"(g));
    bs = fma(l, dz[u], bs);
    acc |= (uint32_t)__double2hiint(e) | (uint32_t)__double2hiint(bs);
  }
}

// What a move redone with its exact bounds reads from memory (requested as soon as the move is known)
struct ExactIn {
  double l1, r0, r1, r2, r3;
};

// Half-block H (moves 8 H .. 8 H + 7; its entries of L copied to l8: one body of code for the eight, the path is rare and
// its code cold) of a segment the slow way: the three sufficient tests from the
// chain value bs at its start (a move before `start` carries dz = 0 in the wave's LDS slot and is not tested).  Returns the
// mask of the moves that fail all three somewhere in the matrix (wave-uniform, the same in every wave) and, in bb, the
// chain value just before the first of them (after the half-block if there is none); for that move the exact path's
// inputs go into x (its tnorm record is requested here, its entry of L is in the registers).
template <bool SOLO>     // SOLO: one wavefront holds every row (P <= 64): nothing to exchange, no barrier
__device__ __forceinline__ uint32_t run_half(const double (&l8)[8], int H, double bs, bool lastrow, int lane, int wave,
                                             int start, int mcnt, const double* zw, uint32_t* x1, unsigned& par,
                                             const double* __restrict__ RkSeg, int g4, double& bb, ExactIn& x)
{
  const int lo_m = start > 8 * H ? start : 8 * H, hi_m = mcnt < 8 * H + 8 ? mcnt : 8 * H + 8;
  const uint32_t valid = hi_m <= lo_m ? 0u : (((1u << (hi_m - 8 * H)) - 1u) & ~((1u << (lo_m - 8 * H)) - 1u));
  double dz[8], z1[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const double2 v = *reinterpret_cast<const double2*>(zw + 2 * (8 * H + u));
    dz[u] = v.x;
    z1[u] = v.y;
  }
  uint32_t pA = 0, pB = 0, pC = 0;
  double b = bs;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const double l1 = l8[u];
    const double g = fma(-l1, z1[u], b);                             // beta_j + L (p - z_c) at p = 0
    const double es = fma(l1, dz[u], b);                             //                      at p = s: beta_j after the move
    const double e1 = fma(l1, -2.51, g), e2 = fma(l1, -1.26, g), e3 = fma(l1, 1.26, g), e4 = fma(l1, 2.51, g);
    const uint32_t hs = (uint32_t)__double2hiint(es), h0 = (uint32_t)__double2hiint(g);
    pA |= ((hs | (uint32_t)__double2hiint(e2) | (uint32_t)__double2hiint(e3)) >> 31) << u;          // (1.26, 1.26)
    pB |= ((hs | h0 | (uint32_t)__double2hiint(e4)) >> 31) << (8 + u);                              // (0, 2.51)
    pC |= ((hs | (uint32_t)__double2hiint(e1) | h0) >> 31) << (16 + u);                             // (2.51, 0)
    b = es;
  }
  uint32_t pk = lastrow ? 0u : (pA | pB | pC);                       // row P-1 is not constrained (Logit.hpp:383: j < P-1)
  pk = wave_or_u32(pk) & (valid * 0x010101u);
  uint32_t m = pk;
  if (!SOLO) {
    uint32_t* slot = x1 + par * 4;
    if (lane == 0) slot[wave] = pk;
    __syncthreads();
    const uint4 sv = *reinterpret_cast<const uint4*>(slot);
    par ^= 1u;
    m = sv.x | sv.y | sv.z | sv.w;
  }
  const uint32_t f = (uint32_t)__builtin_amdgcn_readfirstlane((int)(m & (m >> 8) & (m >> 16) & 0xFFu));
  int n = 8;
  if (f != 0u) {
    n = __builtin_ctz(f);
    const double* Rn = RkSeg + (size_t)(8 * H + n) * kRec + g4;
    x.r0 = Rn[0];
    x.r1 = Rn[1];
    x.r2 = Rn[2];
    x.r3 = Rn[3];
  }
  b = bs;
  double lsel = 0.0;                           // the redone move's entry of L: in the registers already
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    if (u < n) b = fma(l8[u], dz[u], b);
    if (u == n) lsel = l8[u];
  }
  x.l1 = lsel;
  bb = b;
  return f;
}

// move mf of the segment with its exact bounds (Logit.hpp:383-397), its inputs in x: split_exact's arithmetic
template <bool SOLO>
__device__ __forceinline__ void run_exact(const ExactIn& x, int mf, int cq, double z1q, bool lastrow, int lane, int wave,
                                          int nq, double& bj, double* sz, double* xl, unsigned& par)
{
  const int c = __builtin_amdgcn_readlane(cq, mf);
  const double z1 = readlane_f64(z1q, mf);
  // constrained_wide_reciprocals' tables, made here from the entry of L (the same IEEE quotient; a table read would put an L2
  // round trip at the head of the move)
  const double rcp = 1.0 / x.l1, nan = __builtin_nan("");
  const double rl = (x.l1 > 0.0 && !lastrow) ? rcp : nan, rh = (x.l1 < 0.0 && !lastrow) ? rcp : nan;
  double lo = z1 - bj * rl, hi = z1 - bj * rh;       // NaN: this row does not bound the move on that side
  wave_maxmin(lo, hi);
  double glo = -__builtin_huge_val(), ghi = __builtin_huge_val();
  if (SOLO) {
    glo = vmax64(glo, lo);                   // v_max_f64 / v_min_f64 return the other operand for a NaN
    ghi = vmin64(ghi, hi);
  } else {
    double* sl = xl + par * 8;
    if (lane == 0) {
      sl[wave * 2] = lo;
      sl[wave * 2 + 1] = hi;
    }
    __syncthreads();
    for (int w = 0; w < nq; ++w) {
      glo = vmax64(glo, sl[w * 2]);
      ghi = vmin64(ghi, sl[w * 2 + 1]);
    }
    par ^= 1u;
  }
  const double z2 = tnorm_lanes(x.r0, x.r1, x.r2, x.r3, lane, glo, ghi);
  bj += x.l1 * (z2 - z1);
  if (wave == 0 && lane == 0) sz[c] = z2;
}

// Scan tables for k_beta_sweeps_run: tab[k P + i] = coordinate of move i of scan k.  Scan k applies its P-1 swaps to the
// order scan k-1 left (Logit.hpp:375-377: the permutation persists), i.e. order_k = g_0 o g_1 o ... o g_k with g_k the
// scan's own swaps on the identity: a prefix product, taken here in log2 P doubling steps over two LDS copies.
__global__ __launch_bounds__(1024) void k_beta_scan_tables(blk::BetaArgs a)
{
  extern __shared__ unsigned char tb8[];
  const int P = a.P, t = threadIdx.x, nthr = (int)blockDim.x;
  const double* rec = a.work + 2 * (size_t)P * P + 2 * (size_t)P;
  const int* swp = reinterpret_cast<const int*>(rec + (size_t)P * P * kRec);
  uint32_t* tab = reinterpret_cast<uint32_t*>(const_cast<double*>(rec) + (size_t)P * P * kRec + ((size_t)P * P + 1) / 2 + (size_t)P * P);
  unsigned char* A = tb8;
  unsigned char* Bf = tb8 + (size_t)P * P;
  if (t < P) {
    unsigned char* sg = A + t * P;
    for (int i = 0; i < P; ++i) sg[i] = (unsigned char)i;
    for (int i = 0; i < P - 1; ++i) {
      const int j = swp[t * (P - 1) + i];
      const unsigned char tmp = sg[i];
      sg[i] = sg[j];
      sg[j] = tmp;
    }
  }
  __syncthreads();
  const int k0 = t / P, kstep = nthr / P, istep = nthr - kstep * P;      // e = k P + i walks by nthr without a division
  for (int d = 1; d < P; d <<= 1) {
    int k = k0, i = t - k0 * P;
    for (int e = t; e < P * P; e += nthr) {
      const unsigned char v = A[e];
      Bf[e] = k >= d ? A[(k - d) * P + v] : v;
      k += kstep;
      i += istep;
      if (i >= P) {
        i -= P;
        ++k;
      }
    }
    __syncthreads();
    unsigned char* tmp = A;
    A = Bf;
    Bf = tmp;
  }
  for (int e = t; e < P * P + 128; e += nthr) tab[e] = e < P * P ? (uint32_t)A[e] : 0u;
}

// one cheap pass over the whole half-blocks h0 .. nbw-1 of the segment: the chain value at the end (bs) and at the
// half-blocks' starts (cp1..cp7; one not run leaves the value as it is), the half-blocks that failed somewhere in the
// matrix (Fb, the same in every wave)
#define BL_CALC(cur, h0, H, dX, zX, aX, FETCH, nxt)                                                                         \
    if ((h0) <= (H) && nbw > (H)) run_calc<H>(cur, dX, zX, bs, aX);                                                         \
    FETCH(nxt, H)
#define BL_CALC_K(cur, h0, H, dX, zX, aX, FETCH, nxt)                                                                       \
    if ((h0) <= (H) && nbw > (H)) {                                                                                         \
      double kk_[8];                                                                                                        \
      run_load_k<H>(zk, kk_);                                                                                               \
      run_calc_k<H>(cur, dX, zX, kk_, bs, aX);                                                                              \
    }
#define BL_PASS(cur, h0, CALC, FETCH, nxt, EXCH)                                                                            \
  {                                                                                                                         \
    double dA[8], zA[8], dB[8], zB[8];                                                                                      \
    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;                                                \
    bs = bj;                                                                                                                \
    run_load<0>(zw, dA, zA);                                                                                                \
    run_load<1>(zw, dB, zB);                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                                      \
    CALC(cur, h0, 0, dA, zA, a0, FETCH, nxt)                                                                                \
    cp1 = bs;                                                                                                               \
    run_load<2>(zw, dA, zA);                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                                      \
    CALC(cur, h0, 1, dB, zB, a1, FETCH, nxt)                                                                                \
    cp2 = bs;                                                                                                               \
    run_load<3>(zw, dB, zB);                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                                      \
    CALC(cur, h0, 2, dA, zA, a2, FETCH, nxt)                                                                                \
    cp3 = bs;                                                                                                               \
    run_load<4>(zw, dA, zA);                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                                      \
    CALC(cur, h0, 3, dB, zB, a3, FETCH, nxt)                                                                                \
    cp4 = bs;                                                                                                               \
    run_load<5>(zw, dB, zB);                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                                      \
    CALC(cur, h0, 4, dA, zA, a4, FETCH, nxt)                                                                                \
    cp5 = bs;                                                                                                               \
    run_load<6>(zw, dA, zA);                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                                      \
    CALC(cur, h0, 5, dB, zB, a5, FETCH, nxt)                                                                                \
    cp6 = bs;                                                                                                               \
    run_load<7>(zw, dB, zB);                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                                      \
    CALC(cur, h0, 6, dA, zA, a6, FETCH, nxt)                                                                                \
    cp7 = bs;                                                                                                               \
    CALC(cur, h0, 7, dB, zB, a7, FETCH, nxt)                                                                                \
    uint32_t vb = (a0 >> 31) | ((a1 >> 31) << 1) | ((a2 >> 31) << 2) | ((a3 >> 31) << 3) | ((a4 >> 31) << 4) |             \
                  ((a5 >> 31) << 5) | ((a6 >> 31) << 6) | ((a7 >> 31) << 7);                                                \
    if (lastrow) vb = 0u;                                                                                                   \
    vb = wave_or_u32(vb);                                                                                                   \
    EXCH(vb)                                                                                                                \
  }
#define BL_HALF(H, cp)                                                                                                      \
    {                                                                                                                       \
      _Pragma("unroll") for (int u = 0; u < 8; ++u) l8[u] = cur_[8 * (H) + u];                                              \
      bs0 = start <= 8 * (H) ? cp : bj;                                                                                     \
    }
// the waves' verdicts of a pass meet in LDS (one word each, one barrier); SOLO: there is one wave
#define BL_EXCH_WG(vb)                                                                                                      \
    uint32_t* slot1 = x1 + par * 4;                                                                                         \
    if (lane == 0) slot1[wave] = vb;                                                                                        \
    __syncthreads();                                                                                                        \
    const uint4 sv = *reinterpret_cast<const uint4*>(slot1);                                                                \
    par ^= 1u;                                                                                                              \
    Fb = (uint32_t)__builtin_amdgcn_readfirstlane((int)(sv.x | sv.y | sv.z | sv.w)) | tailbit;
#define BL_EXCH_SOLO(vb) Fb = (uint32_t)__builtin_amdgcn_readfirstlane((int)vb) | tailbit;
#define BL_NOFETCH(dst, G_)

// the rare part of a segment: a cheap pass has left failing half-blocks in Fb.  In their order: the three tests on the
// half-block; a move that fails them with its exact bounds; then everything behind that move again.  On exit bj is the
// chain value at the segment's end, exm the moves redone exactly, nslow their count added up.
#define BL_SLOW_LOOP(cur, SOLO, EXCH, NQW)                                                                                  \
  {                                                                                                                         \
  const double (&cur_)[64] = cur;                                                                                           \
  int start = 0;                                                                                                            \
  for (;;) {                                                                                                                \
    if (Fb == 0u) {                                                                                                         \
      bj = bs;                                                                                                              \
      break;                                                                                                                \
    }                                                                                                                       \
    const int Hf = __builtin_ctz(Fb);                                                                                       \
    const int bend = mcnt < 8 * Hf + 8 ? mcnt : 8 * Hf + 8;                                                                 \
    uint32_t f;                                                                                                             \
    double bb;                                                                                                              \
    ExactIn xin;                                                                                                            \
    if (prof) ++ncareful;                                                                                                   \
    double l8[8], bs0;                                                                                                      \
    switch (Hf) {                                                                                                           \
      case 0: BL_HALF(0, bj) break;                                                                                         \
      case 1: BL_HALF(1, cp1) break;                                                                                        \
      case 2: BL_HALF(2, cp2) break;                                                                                        \
      case 3: BL_HALF(3, cp3) break;                                                                                        \
      case 4: BL_HALF(4, cp4) break;                                                                                        \
      case 5: BL_HALF(5, cp5) break;                                                                                        \
      case 6: BL_HALF(6, cp6) break;                                                                                        \
      default: BL_HALF(7, cp7) break;                                                                                       \
    }                                                                                                                       \
    f = run_half<SOLO>(l8, Hf, bs0, lastrow, lane, wave, start, mcnt, zw, x1, par, RkSeg, g4, bb, xin);                     \
    bj = bb;                                                                                                                \
    if (f == 0u) {                           /* the chain is as the cheap pass had it: its other verdicts stand */          \
      start = bend;                                                                                                         \
      Fb &= ~(1u << Hf);                                                                                                    \
      if (start >= mcnt) break;              /* that was the segment's last half-block: bj is the chain's end */            \
      continue;                                                                                                             \
    }                                                                                                                       \
    const int nf = __builtin_ctz(f), mf = 8 * Hf + nf;     /* the first move that needs its bounds */                       \
    const long long te0 = prof ? clock64() : 0;                                                                             \
    run_exact<SOLO>(xin, mf, cq, z1q, lastrow, lane, wave, NQW, bj, sz, xl, par);                                           \
    if (prof) tExact += clock64() - te0;                                                                                    \
    exm |= 1ull << mf;                                                                                                      \
    nslow = __builtin_amdgcn_readfirstlane(nslow + 1);                                                                      \
    start = mf + 1;                                                                                                         \
    if (start >= mcnt) break;                                                                                               \
    /* everything behind the move again, from its chain value, in one cheap pass: the moves done retire (dz = z_c = 0, nothing */\
    /* to test), the others are tested as ever */                                                                           \
    {                                                                                                                       \
      const bool inr = lane >= start && lane < mcnt;                                                                        \
      zk[lane] = inr ? 1.26 : 0.0;                                                                                          \
      if (!inr) *reinterpret_cast<double2*>(zw + 2 * lane) = make_double2(0.0, 0.0);                                        \
      WAVE_SYNC();                                                                                                          \
    }                                                                                                                       \
    if (prof) ++nrepass;                                                                                                    \
    const long long tr0 = prof ? clock64() : 0;                                                                             \
    BL_PASS(cur, Hf, BL_CALC_K, BL_NOFETCH, cur, EXCH)                                                                      \
    if (prof) tRepass += clock64() - tr0;                                                                                   \
  }                                                                                                                         \
  }

// The cheap pass of a scan at P = 64 on FOUR wavefronts.  With every move of the scan taking its first normal, dz is known
// for the whole scan up front, so the chain value before move u is beta_j plus the first u terms of one sum: a wavefront walks
// the chain alone through the moves of the wavefronts before it (one FMA a move, in the one-wavefront order: the same bits)
// and tests its own moves -- solo_scan's pass, element for element.  Returns the wave's verdict (bit h: half-block h
// failed somewhere in the matrix) and in bs_out the chain value behind its last move (the last wavefront's is the scan's end).
// zw: the wave's own 128-double slot.
template <int H>      // the dz alone of half-block H (a wave that only walks the chain through it needs no z_c)
__device__ __forceinline__ void run_load_dz(const double* zw, double (&dz)[8])
{
#pragma unroll
  for (int u = 0; u < 8; ++u) dz[u] = zw[2 * (8 * H + u)];
}
template <int H>
__device__ __forceinline__ void run_chain(const double (&l1)[64], const double (&dz)[8], double& bs)
{
#pragma unroll
  for (int u = 0; u < 8; ++u) bs = fma(l1[8 * H + u], dz[u], bs);
}
template <int HB0, int HB1>      // the wave tests half-blocks HB0 .. HB1 - 1 (of eight moves each) and walks the chain up to them
__device__ __forceinline__ uint32_t quad_pass(const double* S, int ld, int lane, int cvec, double svec, double z1v, double* zw,
                                              double bj, double& bs_out, double (&cur)[64], double (&cp)[8], const double* Rk)
{
  const bool lastrow = lane == 63;                           // row P-1 is not constrained (Logit.hpp:383: j < P-1)
  *reinterpret_cast<double2*>(zw + 2 * lane) = make_double2(svec - z1v, z1v);
  const int co = cvec * ld;                                  // cur: L(lane, c_u) for the moves u < 8 HB1
#pragma unroll
  for (int u = 0; u < 64; ++u) cur[u] = u < 8 * HB1 ? S[__builtin_amdgcn_readlane(co, u) + lane] : 0.0;
  WAVE_SYNC();
  double bs = bj;
  double dA[8], zA[8], dB[8], zB[8];
  uint32_t vb = 0u;
  // (cp[h]: the chain value in front of half-block h -- the last wavefront has walked them all: quad_slow's starting points;
  // the (dz, z_c) of half-block h + 1 are requested before half-block h is worked on)
#define BL_QSTEP(H, dX, zX, dY, zY)                      \
  if constexpr ((H) < HB1) {                             \
    if constexpr ((H) + 1 < HB0) run_load_dz<((H) + 1 < 8 ? (H) + 1 : 7)>(zw, dY);  \
    else if constexpr ((H) + 1 < HB1) run_load<((H) + 1 < 8 ? (H) + 1 : 7)>(zw, dY, zY); \
    __builtin_amdgcn_sched_barrier(0);                   \
    cp[(H)] = bs;                                        \
    if constexpr ((H) < HB0) {                           \
      run_chain<(H)>(cur, dX, bs);                       \
    } else {                                             \
      uint32_t acc_ = 0u;                                \
      run_calc<(H)>(cur, dX, zX, bs, acc_);              \
      vb |= (acc_ >> 31) << (H);                         \
    }                                                    \
  }
  if constexpr (HB0 > 0) run_load_dz<0>(zw, dA);
  else run_load<0>(zw, dA, zA);
  BL_QSTEP(0, dA, zA, dB, zB) BL_QSTEP(1, dB, zB, dA, zA) BL_QSTEP(2, dA, zA, dB, zB) BL_QSTEP(3, dB, zB, dA, zA)
  BL_QSTEP(4, dA, zA, dB, zB) BL_QSTEP(5, dB, zB, dA, zA) BL_QSTEP(6, dA, zA, dB, zB) BL_QSTEP(7, dB, zB, dA, zA)
#undef BL_QSTEP
  if (lastrow) vb = 0u;
  bs_out = bs;
  vb = wave_or_u32(vb);                                      // bit h: half-block h failed the cheap test somewhere in the matrix
  // A half-block that fails the cheap test goes through the three sufficient tests here, on the wavefront that holds it (a
  // coordinate that sits within 1.26 of its bound fails the cheap test in every scan of the draw and passes one of the
  // one-sided ones: such a scan used to cost 2 600 cycles on the slow path, a draw of them 0.08 ms).  Only a half-block
  // with a move that needs its bounds is reported.
  for (uint32_t rem = vb; rem != 0u; rem &= rem - 1u) {
    const int h = __builtin_ctz(rem);
    double l8[8], bs0 = 0.0;
#define BL_QHALF(H)                                                      \
  if constexpr (HB0 <= (H) && (H) < HB1)                                  \
    if (h == (H)) {                                                       \
      _Pragma("unroll") for (int u = 0; u < 8; ++u) l8[u] = cur[8 * (H) + u]; \
      bs0 = cp[(H)];                                                      \
    }
    BL_QHALF(0) BL_QHALF(1) BL_QHALF(2) BL_QHALF(3) BL_QHALF(4) BL_QHALF(5) BL_QHALF(6) BL_QHALF(7)
#undef BL_QHALF
    double bb;
    ExactIn xin;
    unsigned par = 0;
    const uint32_t f = run_half<true>(l8, h, bs0, lastrow, lane, 0, 0, 64, zw, nullptr, par, Rk, (lane < 5 ? lane : 0) * 4, bb, xin);
    if (f == 0u) vb &= ~(1u << h);
  }
  return vb;
}

// One scan of the constrained sweeps for P <= 64 (Logit.hpp:380-398) on ONE wavefront -- lane = row, beta_j in a register; L,
// z and the scan's records in LDS: k_beta_sweeps_run's speculative segment (the scan's P moves are one segment) with nothing
// to exchange and nothing to fetch ahead.  Returns the number of moves redone with their exact bounds.
__device__ __forceinline__ int solo_scan(const double* S, int ld, int P, int lane, int cvec, double svec, double z1v,
                                         const double* Rk, double* zw, double* zk, double* zz, double& bj)
{
  const int mcnt = P, wave = 0;
  const bool has = lane < P, lastrow = lane >= P - 1;       // row P-1 is not constrained (Logit.hpp:383: j < P-1)
  const int lr = has ? lane : P - 1;                         // a lane past the matrix rides on row P-1, which no test looks at
  const int cq = has ? cvec : 0, g4 = (lane < 5 ? lane : 0) * 4;
  const double sq = svec, z1q = has ? z1v : 0.0, dzq = has ? sq - z1q : 0.0;
  const double* RkSeg = Rk;
  double* sz = zz;
  uint32_t* x1 = nullptr;                                    // (the exchange slots of the four-wave kernel: unused)
  double* xl = nullptr;
  unsigned par = 0;
  constexpr bool prof = false;
  unsigned long long ncareful = 0, nrepass = 0;
  long long tExact = 0, tRepass = 0;
  int nslow = 0;
  // the scan's columns of L, in the order of its moves: L(lr, c_u) for move u
  double cur[64];
  const int co = cq * ld;
#pragma unroll
  for (int u = 0; u < 64; ++u) cur[u] = S[__builtin_amdgcn_readlane(co, u) + lr];
  *reinterpret_cast<double2*>(zw + 2 * lane) = make_double2(dzq, z1q);
  WAVE_SYNC();
  const int nbw = mcnt >> 3;                                      // whole half-blocks
  const uint32_t tailbit = (mcnt & 7) ? (1u << nbw) : 0u;         // the one the scan's end cuts: the slow way
  double bs, cp1, cp2, cp3, cp4, cp5, cp6, cp7;
  uint32_t Fb;
  BL_PASS(cur, 0, BL_CALC, BL_NOFETCH, cur, BL_EXCH_SOLO)
  unsigned long long exm = 0ull;                                  // moves redone exactly (they wrote their own z)
  if (Fb == 0u) {
    bj = bs;
  } else {
    BL_SLOW_LOOP(cur, true, BL_EXCH_SOLO, 1)
  }
  if (has && !((exm >> lane) & 1ull)) sz[cq] = sq;
  (void)ncareful; (void)nrepass; (void)tExact; (void)tRepass; (void)x1; (void)xl;
  return nslow;
}

// What is left of a scan of P = 64 moves whose cheap pass (quad_pass) found half-blocks that need a closer look (Fb0), on
// the wavefront that holds every column of the scan and every chain value: solo_scan's second part.
__device__ __forceinline__ int quad_slow(const double (&cur)[64], const double (&cp)[8], double bs_end, uint32_t Fb0, int lane,
                                         int cvec, double svec, double z1v, const double* Rk, double* zw, double* zk, double* zz,
                                         double& bj)
{
  const int mcnt = 64, wave = 0;
  const bool lastrow = lane >= 63;
  const int cq = cvec, g4 = (lane < 5 ? lane : 0) * 4;
  const double sq = svec, z1q = z1v;
  const double* RkSeg = Rk;
  double* sz = zz;
  uint32_t* x1 = nullptr;
  double* xl = nullptr;
  unsigned par = 0;
  constexpr bool prof = false;
  unsigned long long ncareful = 0, nrepass = 0;
  long long tExact = 0, tRepass = 0;
  int nslow = 0;
  const int nbw = 8;
  const uint32_t tailbit = 0u;
  double bs = bs_end, cp1 = cp[1], cp2 = cp[2], cp3 = cp[3], cp4 = cp[4], cp5 = cp[5], cp6 = cp[6], cp7 = cp[7];
  uint32_t Fb = Fb0;
  unsigned long long exm = 0ull;
  BL_SLOW_LOOP(cur, true, BL_EXCH_SOLO, 1)
  if (!((exm >> lane) & 1ull)) sz[cq] = sq;
  (void)ncareful; (void)nrepass; (void)tExact; (void)tRepass; (void)x1; (void)xl;
  return nslow;
}

typedef uint32_t v2u32 __attribute__((ext_vector_type(2)));

template <int NQ>
__global__ __launch_bounds__(kBlock) void k_beta_sweeps_run(blk::BetaArgs a)
{
  extern __shared__ double lds[];
  if (*a.dead != 0) return;       // see k_beta
  uint32_t* mode = beta_mode_word(a);
  if (mode[0] != 0u) return;               // a chain pressed against its bounds: k_beta_sweeps, launched behind, takes the draw
  const int P = a.P, t = threadIdx.x, nthr = (int)blockDim.x;
  const double* __restrict__ Rg = a.work;                          // 1/L where L > 0, else NaN (k_beta's A)
  const double* __restrict__ Lg = a.work + (size_t)P * P;          // L     (k_beta's S)
  const double* zz = a.work + 2 * (size_t)P * P + P;               // z
  const double* rec = a.work + 2 * (size_t)P * P + 2 * (size_t)P;
  const double* __restrict__ Rh = rec + (size_t)P * P * kRec + ((size_t)P * P + 1) / 2;   // 1/L where L < 0, else NaN
  const uint32_t* __restrict__ tab = reinterpret_cast<const uint32_t*>(Rh + (size_t)P * P);
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int Pe = (P + 1) & ~1;                                   // 16-byte alignment of what follows
  double* sz = lds;                                              // z (P)
  double* xl = sz + Pe + 16;                                     // [2][4 waves][2]: a move's bound candidates (16 doubles before it: spare)
  uint32_t* x1 = reinterpret_cast<uint32_t*>(xl + 16);           // [2][4 waves]: a pass's verdicts / a half-block's three test masks
  double* zw = xl + 16 + 4 + 128 * wave;                         // [4 waves][64][2]: the wave's (dz, z_c), read back uniform
  double* zk = xl + 16 + 4 + 512 + 64 * wave;                    // [4 waves][64]: the test's half-width per move (a pass taken again)
  for (int j = t; j < P; j += nthr) sz[j] = zz[j];
  if (t < 8) x1[t] = 0u;
  __syncthreads();

  const int j = 64 * wave + lane;
  const int jr = j < P ? j : P - 1;          // a lane past the matrix rides on row P-1, which no test looks at
  const bool lastrow = jr == P - 1;
  double bj = j < P ? a.beta_prev[j] : 0.0;
  const int g4 = (lane < 5 ? lane : 0) * 4;
  const uint32_t jr8 = (uint32_t)jr * 8u, P8 = (uint32_t)P * 8u;
  unsigned par = 0;
  int spec_on = 1;      // (kept wave-uniform by hand: a branch the compiler takes for divergent drags the exact path's waits into the fast one)
  // L is lower triangular: the rows of wave w are zero in the columns from 64 (w + 1) on.  The buffer ends there, and a load
  // past the end of a buffer returns zero without going to memory (3/8 of the loads of P = 256)
  const int ncol = 64 * (wave + 1) < P ? 64 * (wave + 1) : P;
  const __amdgpu_buffer_rsrc_t Lrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(Lg), (short)0, (int)((uint32_t)ncol * P8), 0x00020000);
  const int nseg = (P + 63) / 64;            // segments per scan (<= NQ)
  const int total = P * nseg;
  // per lane, one and two segments ahead: the coordinate of move `lane` (its column offset, for the fetch) and its first normal
  int cq_n, cq_nn, co_n;
  double sq_n;
#define BL_TABLE(dst, gg, kk_, qq_)   /* lane i: coordinate of move i of segment gg = (scan kk_, moves 64 qq_ ...) */       \
  {                                                                                                                         \
    dst = ((gg) < total && lane < P - 64 * qq_) ? (int)tab[(size_t)kk_ * P + 64 * qq_ + lane] : 0;                          \
  }
#define BL_FETCH_HEAD(gg, kk_, qq_)   /* segment gg = (scan kk_, moves 64 qq_ ...): its normals; its column offsets (coordinates in cq_n) */\
  {                                                                                                                         \
    sq_n = ((gg) < total && lane < P - 64 * qq_) ? rec[((size_t)kk_ * P + 64 * qq_ + lane) * kRec + 3] : 0.0;               \
    co_n = cq_n * (int)P8;                                                                                                  \
  }
#define BL_FETCH_GROUP(dst, G_)       /* eight column offsets to scalars, then eight loads */                               \
  {                                                                                                                         \
    int so_[8];                                                                                                             \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) so_[u] = __builtin_amdgcn_readlane(co_n, 8 * (G_) + u);                   \
    __builtin_amdgcn_sched_barrier(0);                                                                                      \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                                         \
      const v2u32 v_ = __builtin_amdgcn_raw_buffer_load_b64(Lrs, (int)jr8, so_[u], 0);                                      \
      dst[8 * (G_) + u] = __hiloint2double((int)v_.y, (int)v_.x);                                                           \
    }                                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                                      \
  }
#define BL_FETCH_ALL(dst)                                                                                                   \
  {                                                                                                                         \
    BL_FETCH_GROUP(dst, 0) BL_FETCH_GROUP(dst, 1) BL_FETCH_GROUP(dst, 2) BL_FETCH_GROUP(dst, 3)                             \
    BL_FETCH_GROUP(dst, 4) BL_FETCH_GROUP(dst, 5) BL_FETCH_GROUP(dst, 6) BL_FETCH_GROUP(dst, 7)                             \
  }
  const bool prof = a.dbg != nullptr;
  unsigned long long nexact = 0, ncareful = 0, nrepass = 0;
  long long tPass = 0, tSlow = 0, tExact = 0, tRepass = 0, tAll = prof ? clock64() : 0;
  int nslow = 0, noff = 0;     // moves of the scan redone exactly; scans in which a third of the moves were

  // one segment: `cur` holds its columns, `nxt` receives the next segment's
#define BL_SEGMENT(cur, nxt, g, k, q)                                                                                       \
  {                                                                                                                         \
    const int m0 = 64 * (q);                                                                                                \
    const int mcnt = (P - m0) < 64 ? (P - m0) : 64;                                                                         \
    const double* RkSeg = rec + ((size_t)(k) * P + m0) * kRec;                                                              \
    const bool has = lane < mcnt;                                                                                           \
    /* what the last segment fetched has had a segment's time to land: say so before the next fetch is issued, or the */    \
    /* in-order load counter makes every later use of it wait for the new fetch instead */                                  \
    __builtin_amdgcn_s_waitcnt(0x0F70);     /* vmcnt(0) */                                                                  \
    const int cq = cq_n;                                                                                                    \
    const double sq = sq_n;                                                                                                 \
    cq_n = cq_nn;                                                                                                           \
    const double z1q = has ? sz[cq] : 0.0;                                                                                  \
    const double dzq = has ? sq - z1q : 0.0;                                                                                \
    *reinterpret_cast<double2*>(zw + 2 * lane) = make_double2(dzq, z1q);                                                    \
    const int q1_ = (q) + 1 == nseg ? 0 : (q) + 1, k1_ = (q) + 1 == nseg ? (k) + 1 : (k);      /* the next segment, the one after */\
    const int q2_ = q1_ + 1 == nseg ? 0 : q1_ + 1, k2_ = q1_ + 1 == nseg ? k1_ + 1 : k1_;                                   \
    BL_TABLE(cq_nn, (g) + 2, k2_, q2_)                                                                                      \
    BL_FETCH_HEAD((g) + 1, k1_, q1_)   /* (past the last segment: column 0 again, harmless) */                              \
    WAVE_SYNC();                                                                                                            \
    if (!spec_on) {                                                                                                         \
      BL_FETCH_ALL(nxt)                                                                                                     \
      for (int i = 0; i < mcnt; ++i) split_exact(Lg, Rg, Rh, RkSeg, P, jr, lane, wave, NQ, i, cq, z1q, bj, sz, xl, par, g4); \
      nslow += mcnt;                                                                                                        \
    } else {                                                                                                                \
      const int nbw = mcnt >> 3;                                      /* whole half-blocks */                               \
      const uint32_t tailbit = (mcnt & 7) ? (1u << nbw) : 0u;         /* the one the scan's end cuts: the slow way */       \
      double bs, cp1, cp2, cp3, cp4, cp5, cp6, cp7;                                                                         \
      uint32_t Fb;                                                                                                          \
      /* the first pass in line (a loop header here would wait for the fetch just issued); what follows it is rare */       \
      const long long tp0 = prof ? clock64() : 0;                                                                           \
      BL_PASS(cur, 0, BL_CALC, BL_FETCH_GROUP, nxt, BL_EXCH_WG)   /* the next segment's columns are requested between the half-blocks */\
      if (prof) tPass += clock64() - tp0;                                                                                   \
      unsigned long long exm = 0ull;                                  /* moves redone exactly (they wrote their own z) */   \
      if (Fb == 0u) {                                                                                                       \
        bj = bs;                                                                                                            \
      } else {                                                                                                              \
        const long long ts0 = prof ? clock64() : 0;                                                                         \
        BL_SLOW_LOOP(cur, false, BL_EXCH_WG, NQ)                                                                            \
        if (prof) tSlow += clock64() - ts0;                                                                                 \
      }                                                                                                                     \
      if (wave == 0 && has && !((exm >> lane) & 1ull)) sz[cq] = sq;                                                         \
    }                                                                                                                       \
  }

  double colA[64], colB[64];
  BL_TABLE(cq_n, 0, 0, 0)
  BL_TABLE(cq_nn, 1, (nseg == 1 ? 1 : 0), (nseg == 1 ? 0 : 1))
  __builtin_amdgcn_s_waitcnt(0x0F70);
  BL_FETCH_HEAD(0, 0, 0)
  BL_FETCH_ALL(colA)
  int k = 0, q = 0;
  for (int g = 0; g < total; g += 2) {
    BL_SEGMENT(colA, colB, g, k, q)
    if (++q == nseg) {
      q = 0;
      noff += 3 * nslow >= P ? 1 : 0;
      spec_on = __builtin_amdgcn_readfirstlane((3 * nslow < P || ((k + 1) & 7) == 0) ? 1 : 0);   // a chain pressed against its bounds: every move exactly; look again every 8th scan
      nexact += (unsigned long long)nslow;
      nslow = 0;
      ++k;
      __syncthreads();         // the scan's z are in LDS before the next scan gathers them
    }
    if (g + 1 < total) {
      BL_SEGMENT(colB, colA, g + 1, k, q)
      if (++q == nseg) {
        q = 0;
        noff += 3 * nslow >= P ? 1 : 0;
        spec_on = __builtin_amdgcn_readfirstlane((3 * nslow < P || ((k + 1) & 7) == 0) ? 1 : 0);
        nexact += (unsigned long long)nslow;
        nslow = 0;
        ++k;
        __syncthreads();
      }
    }
  }
  if (j < P) a.beta_out[j] = bj;
  if (t == 0) mode[1] = (2 * noff > P) ? 1u : 0u;     // most scans pressed against the bounds: the next draw goes to k_beta_sweeps
  if (prof && t == 0) {
    a.dbg[8] += nexact;
    a.dbg[12] += ncareful;
    a.dbg[13] += nrepass;
    a.dbg[19] = (unsigned long long)(clock64() - tAll);
    a.dbg[16] = 0ull;
    a.dbg[20] = (unsigned long long)tExact;
    a.dbg[21] = (unsigned long long)tRepass;
    a.dbg[17] = (unsigned long long)tPass;
    a.dbg[18] = (unsigned long long)tSlow;
  }
#undef BL_SEGMENT
#undef BL_FETCH_ALL
#undef BL_FETCH_GROUP
#undef BL_FETCH_HEAD
#undef BL_TABLE
}
#undef L_

}  // namespace

namespace blk {

size_t beta_work_doubles(int P)
{
  size_t generic = 2 * (size_t)P * P + 6 * (size_t)P + 64;
  if (P > 64 && P <= 256)    // constrained_sweeps_wide: tnorm records + swap targets after the dense stage's matrices
    generic += (size_t)P * P * kRec + ((size_t)P * P + 1) / 2 + (size_t)P * P +   // + the second reciprocal matrix
               ((size_t)P * P + 128 + 1) / 2 + 2;                                 // + the scan tables (u32, 128 entries of slack) + the mode words
  const size_t small = (size_t)P * P * kRec + 2 * (((size_t)P * P + 1) / 2) + 64;   // tnorm records + int tables
  return generic > small ? generic : small;
}

// bl_diag_beta_sweeps / BL_BETA_SPLIT: 1 (default) = the row-split sweeps in segments of 64 (k_beta_sweeps_run) with the
// one-wavefront ones (k_beta_sweeps) behind them for a pressed chain; 0 = the one-wavefront sweeps alone.  For comparison:
// both give the same beta.
static bool beta_row_split() { return blh::beta_sweeps_kind() != 0; }

void launch_beta(const BetaArgs& a, int mode, hipStream_t s)
{
  if (a.P <= 64) {
    const int ld = a.P + 1;
    const size_t lds = (3 * (size_t)a.P * ld + 2 * (size_t)a.P) * 8 + ((size_t)a.P + 1 + (size_t)a.P * a.P + 1) * 4 +
                       2 * (size_t)a.P * kRec * 8 + 32;
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)k_beta64, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (mode == B_CONSTRAINED) hipLaunchKernelGGL(k_beta64_records, dim3((a.P * a.P + 255) / 256), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_beta64, dim3(1), dim3(kBlock), lds, s, a, mode, beta_row_split() ? 1 : 0);
    return;
  }
  // P > 64: factor (one workgroup) -> inverse (a workgroup per 64 columns) -> finish (one workgroup) [-> sweeps]
  const bool wide = mode == B_CONSTRAINED && a.P <= 256;
  // The draw's random inputs (records, scan tables: 0.16 ms at P = 256) depend on (seed, epoch, P) alone and the dense stage
  // (1.4 ms of one-workgroup kernels) on X'Omega X alone: the inputs go to a stream of the library's own beside it, behind
  // everything the caller's stream holds so far (the last draw's sweeps read the same workspace); the sweeps wait for both.
  static std::mutex mu;               // the side stream and its events are the library's: one call's record / wait pairs at a time
  std::unique_lock<std::mutex> lock(mu, std::defer_lock);
  if (wide) lock.lock();
  static hipStream_t side = nullptr;
  static hipEvent_t e0 = nullptr, e1 = nullptr;
  static bool made = false, usable = false;
  if (wide && !made) {
    made = true;
    usable = hipStreamCreateWithFlags(&side, hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&e0, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&e1, hipEventDisableTiming) == hipSuccess;
  }
  const bool fan = wide && usable && hipEventRecord(e0, s) == hipSuccess && hipStreamWaitEvent(side, e0, 0) == hipSuccess;
  hipStream_t si = fan ? side : s;
  if (wide) hipLaunchKernelGGL(k_beta_records, dim3(64), dim3(256), 0, si, a);
  if (wide && beta_row_split()) {
    const size_t lt = 2 * (size_t)a.P * a.P;
    if (lt > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)k_beta_scan_tables, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lt);
    hipLaunchKernelGGL(k_beta_scan_tables, dim3(1), dim3(1024), lt, si, a);
  }
  if (fan) (void)hipEventRecord(e1, side);
  hipLaunchKernelGGL(k_beta_factor, dim3(1), dim3(kDenseThreads), 0, s, a, mode);
  if (mode == B_SOLVE || mode == B_MVN) return;
  {
    const int nc = mode == B_CONSTRAINED ? a.P + 1 : a.P, cpw = 64;
    hipLaunchKernelGGL(k_beta_inverse, dim3((nc + cpw - 1) / cpw), dim3(kDenseThreads), 0, s, a, nc, cpw);
  }
  size_t lds = 0;
  if (mode == B_CONSTRAINED && !wide) {    // the serial sweeps of P > 256: beta, z, perm
    lds = (2 * (size_t)a.P + (a.P + 1) / 2 + 1) * 8;
    (void)hipFuncSetAttribute((const void*)k_beta_finish, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  hipLaunchKernelGGL(k_beta_finish, dim3(1), dim3(kDenseThreads), lds, s, a, mode);
  if (fan) (void)hipStreamWaitEvent(s, e1, 0);
  if (wide) lds = (2 * (size_t)a.P * kRec + (size_t)a.P) * 8 + (size_t)a.P * a.P + 64;   // the one-wavefront sweeps' LDS
  if (wide && beta_row_split()) {
    const size_t l2 = ((((size_t)a.P + 1) & ~(size_t)1) + 16 + 16 + 4 + 512 + 256) * 8;
    const int nq = (a.P + 63) / 64;
    auto fn = nq == 2 ? k_beta_sweeps_run<2> : nq == 3 ? k_beta_sweeps_run<3> : k_beta_sweeps_run<4>;
    hipLaunchKernelGGL(fn, dim3(1), dim3(kBlock), l2, s, a);
    // and behind it the one-wavefront sweeps, which take the draw instead when the chain is pressed against its bounds (one
    // of the two returns at once: see beta_mode_word)
    if (a.P <= 128) {
      if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)k_beta_sweeps<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(k_beta_sweeps<2>, dim3(1), dim3(kBlock), lds, s, a, true);
    } else {
      if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)k_beta_sweeps<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(k_beta_sweeps<4>, dim3(1), dim3(kBlock), lds, s, a, true);
    }
  } else if (wide) {
    if (a.P <= 128) {
      if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)k_beta_sweeps<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(k_beta_sweeps<2>, dim3(1), dim3(kBlock), lds, s, a, false);
    } else {
      if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)k_beta_sweeps<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(k_beta_sweeps<4>, dim3(1), dim3(kBlock), lds, s, a, false);
    }
  }
}

}  // namespace blk
