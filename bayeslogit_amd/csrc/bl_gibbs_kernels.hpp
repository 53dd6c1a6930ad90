// bl_gibbs_kernels.hpp -- launch interface between capi_gibbs.hip (host logic of
// the Gibbs/EM/mlogit drivers) and kernels_gibbs.hip (the kernels).  Host-callable
// C++ functions; every pointer is a device pointer.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace blk {

enum WeightMode : int {
  W_DRAW = 0,   // omega_i ~ PG((int) n_i, psi_i)           Logit.hpp:283-289
  W_EM = 1      // omega_i = n_i tanh(psi_i/2)/(2 psi_i)    Logit.hpp:509-519
};

// Geometry of the partial-sum workspace for X' Omega X (fixed per (N, P) so that
// summation order -- and therefore every bit of PP -- is reproducible).
struct SweepPlan {
  int P = 0;
  int fused = 0;          // 1: MFMA path, registers (P in {16,32,48,64}); 2: MFMA path, LDS tiles (P in {128,256}); 0: generic
  int nblocks = 0;        // workgroups of the X'WX kernel (= number of partial slabs)
  int nblocks_draw = 0;   // workgroups of the psi/omega kernel
  int chunk_rows = 0;     // rows a wave of the psi/omega kernel takes at a time (multiple of 64, <= 512)
  int nb = 0;             // fused: P/16
  int ntile = 0;          // generic: number of 64x64 output tiles (upper triangle)
  size_t partial_doubles = 0;   // workspace size
  int single_pass = -1;         // P = 64, 256: -1 follow bl_set_sweep_mode, 0 two passes, 1 one pass (a handle turns it off when most
                                // of its rows leave the single pass's fast path: bl_gibbs_sweep_local)
};
SweepPlan make_plan(int64_t N, int P, int num_cus);

// One sweep over this rank's rows: psi_i = x_i.beta - off_i, omega_i by `mode`,
// PPpart = sum_i omega_i x_i x_i' (full symmetric P x P, column-major).
//   off     : per-row offset subtracted from psi (mlogit c_j), or nullptr
//   w_store : where omega_i is written (N doubles), or nullptr: `w_scratch` (N doubles)
//             is used then (omega always passes through memory between the two passes).
//   parts   : 1 = the psi/omega pass only, 2 = the X' Omega X pass only (omega as left in w), 3 = both
//   xoc     : with parts == 2 and plan.fused == 1 (P <= 64): also xoc = X' Omega off (P doubles) from the same pass
void launch_sweep(const SweepPlan& plan, const double* tX, const double* n, const double* beta, const double* off,
                  double* w_store, double* w_scratch, int64_t N, double* partial, double* PPpart, uint64_t seed,
                  uint32_t epoch, uint64_t idx0, int mode, int* status, hipStream_t s, int parts = 3,
                  double* xoc = nullptr);

// The same sweep with X read once (kernels_sweep1.hip): P = 64, W_DRAW, no offset.  launch_sweep takes this path
// when sweep_single_pass() is on (default; bl_set_sweep_mode) and the call is eligible; same omega (to the last bits),
// PPpart in another (fixed) summation order.  ws: sweep_once64_ws_doubles(nblocks, N) doubles (make_plan sizes the
// plan's workspace for it); stats (or nullptr): += the number of rows that left the fast path.
size_t sweep_once64_ws_doubles(int nblocks, int64_t N);
// device counter inside that workspace: rows the second kernel drew, accumulated over the sweeps since it was last zeroed
unsigned long long* sweep_once64_deferred_counter(double* ws, int nblocks, int64_t N);
void launch_sweep_once64(int nblocks, const double* tX, const double* n, const double* beta, double* w, int64_t N,
                         double* ws, double* PP, uint64_t seed, uint32_t epoch, uint64_t idx0, int* status,
                         unsigned long long* stats, hipStream_t s);

// The same at P = 256 (kernels_sweep256.hip): one workgroup of eight waves per CU multiplies tile i while wave 0 draws tile i+1
// and tile i+2 arrives; same contract as launch_sweep_once64 (omega to the last bits, PPpart in another fixed order).
size_t sweep_once256_ws_doubles(int nblocks, int64_t N);
unsigned long long* sweep_once256_deferred_counter(double* ws, int nblocks, int64_t N);
void launch_sweep_once256(int nblocks, const double* tX, const double* n, const double* beta, double* w, int64_t N,
                          double* ws, double* PP, uint64_t seed, uint32_t epoch, uint64_t idx0, int* status,
                          unsigned long long* stats, hipStream_t s);
// the handle's counter of deferred rows of whichever single-pass sweep the plan has (P = 64, 256), or nullptr
unsigned long long* sweep_once_deferred_counter(const SweepPlan& plan, double* ws, int64_t N);

// X' Omega X for P = 128 (nc = 8 chunks of 16 columns) or 256 (nc = 16) on v_mfma_f64_4x4x4_4b_f64 (kernels_xwx4.hip),
// one workgroup of nc/2 waves per block; partial: xwx_q4_big_ws_doubles(nblocks, nc) doubles.
size_t xwx_q4_big_ws_doubles(int nblocks, int nc);
void launch_xwx_q4_big(int nblocks, int nc, const double* tX, const double* w, int64_t N, double* partial, double* PP,
                       hipStream_t s);

// out[j] = sum_i wgt_i x_ij, with wgt_i = n_i (y_i - 1/2) (kappa, Logit.hpp:174-183)
// when w == nullptr, else wgt_i = w_i * c_i (c may be nullptr => 1).
// ws: workspace of colsum_ws_doubles(N, P) doubles.
size_t colsum_ws_doubles(int64_t N, int P);
void launch_colsum(const double* tX, const double* y, const double* n, const double* w, const double* c, int64_t N,
                   int P, double* ws, double* out, hipStream_t s);

// psi = X beta (gemm(psi, tX, beta, 'T'), Logit.hpp:421,431) into out[N]
void launch_xbeta(const double* tX, const double* beta, int64_t N, int P, double* out, hipStream_t s);

// mlogit: c_i = log sum_{k != j} exp(XB[i,k]) over the J columns of XB (N x J, last
// column zero), MultLogit.hpp:293-299, and eta_i = XB[i,j] - c_i (:300).
void launch_mlogit_offset(const double* XB, int64_t N, int J, int j, double* c_out, double* eta_out, hipStream_t s);

// ---- P x P stage (one workgroup; redundant on every rank) ----
struct BetaArgs {
  int P;
  const double* PPsum;     // sum over ranks of X' Omega X            (P*P)
  const double* P0;        // prior precision                         (P*P)
  const double* bP;        // P0 m0 + X' kappa  (or b1 for mlogit)    (P)
  const double* beta_prev; // previous beta                           (P)
  double* beta_out;        // new beta (may alias beta_prev)          (P)
  double* work;            // beta_work_doubles(P) doubles of scratch
  uint64_t seed;
  uint32_t epoch;
  int* status;             // BL_ERR_NOT_PD flag word (host-visible int, device memory): reporting only
  int* dead;               // this chain's own sticky flag (device int, zero while the chain is alive): set with ST_NOT_PD when a
                           // factorisation fails; every later beta kernel of THIS chain returns at once.  Never null.
  unsigned long long* dbg = nullptr;   // optional: 8 phase stamps (100 MHz wall clock), development aid
};
size_t beta_work_doubles(int P);
enum BetaMode : int {
  B_MVN = 0,          // Logit::draw_beta(beta, w, r)            Logit.hpp:291-320
  B_CONSTRAINED = 1,  // Logit::draw_beta(beta, w, beta_prev, r) Logit.hpp:322-400
  B_FROM_LIK = 2,     // Normal::set_from_likelihood + draw      Normal.hpp:98-131
  B_SOLVE = 3         // EM M-step: beta = PP^{-1} bP            Logit.hpp:537-540
};
void launch_beta(const BetaArgs& a, int mode, hipStream_t s);

// dist = max_j |a_j - b_j| into *out (device double)
void launch_maxabsdiff(const double* a, const double* b, int P, double* out, hipStream_t s);
// dst[i] = src[i * stride + off], i < n
void launch_gather_stride(double* dst, const double* src, int64_t n, int stride, int off, hipStream_t s);
// dst[j] = a[j] + (b ? b[j] : 0)
void launch_vec_add(double* dst, const double* a, const double* b, int P, hipStream_t s);
// dst = P0 * m0  (P x P times P)
void launch_matvec(double* dst, const double* M, const double* v, int P, hipStream_t s);

// ---- duplicate-row merge (Logit::compress, Logit.hpp:192-270; MultLogit::set_data
// merge, MultLogit.hpp:137-208): rows with identical covariates are folded into
// their first occurrence in index order; first-occurrence order is kept.
// ty is U x N column-major (U = 1 for the binomial case).  Returns new N in *N_out.
int combine_rows(double* ty, double* tX, double* n, int64_t N, int P, int U, int64_t* N_out, hipStream_t s);

// running moments over the samples of a chain: (mean, M2) += sample number `count` (1-based); finish: M2 -> variance
void launch_welford(const double* x, double* mean, double* m2, int64_t n, int64_t count, hipStream_t s);
void launch_welford_finish(double* m2, int64_t n, int64_t count, hipStream_t s);


#if defined(__HIPCC__)
// The slabs' sum of one element for one of a reduction kernel's 16 wavefronts: slabs first, first + 16, ... added in that order
// (what makes X'Omega X reproducible), 32 of them requested at a time: with one dependent load after the other a slab cost
// 0.33 us (k_reduce_q4: 21 us for the 1024 slabs of a sweep, 9 us this way).
__device__ __forceinline__ double slab_sum16(const double* __restrict__ partial, size_t stride, size_t e, int first, int nparts)
{
  double sum = 0.0;
  int b = first;
  for (; b + 16 * 31 < nparts; b += 16 * 32) {
    double v[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) v[q] = partial[(size_t)(b + 16 * q) * stride + e];
#pragma unroll
    for (int q = 0; q < 32; ++q) sum += v[q];
  }
  for (; b < nparts; b += 16) sum += partial[(size_t)b * stride + e];
  return sum;
}
#endif

}  // namespace blk
