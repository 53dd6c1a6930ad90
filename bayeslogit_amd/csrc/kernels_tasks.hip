// kernels_tasks.hip -- rpg_alt, rpg_sp and the two large classes of rpg_hybrid
// (Code/C/LogitWrapper.cpp:87-167) on the task work queue of bl_task_queue.hpp.  gfx950 only.
// Built with machine-LICM off (bayeslogit_amd/build.py): hoisting the fp64 polynomial constants of the
// attempt body and of the set-up out of the queue loop costs 30-80 registers per lane and a wave per SIMD.
#include "bl_host.hpp"
#include "bl_task_queue.hpp"

namespace blh {

int launch_rpg_tasks(bool sp, double* x, const double* h, const double* z, int64_t num, int* iter, uint64_t seed,
                     uint32_t epoch, uint64_t idx0, int hybrid, unsigned long long* cls_count, hipStream_t s)
{
  // the resident grid (3 workgroups per CU), or fewer when the vector is short
  const int64_t chunks = (num + bl::kTqChunk - 1) / bl::kTqChunk;
  const dim3 g(grid_for(chunks, bl::kTqBlock / 64, 256 * 3)), b(bl::kTqBlock);
  if (sp)
    hipLaunchKernelGGL(bl::k_rpg_tasks<bl::SpPolicy>, g, b, 0, s, x, h, z, num, iter, seed, epoch, idx0, hybrid, cls_count, status_word(s));
  else
    hipLaunchKernelGGL(bl::k_rpg_tasks<bl::AltPolicy>, g, b, 0, s, x, h, z, num, iter, seed, epoch, idx0, hybrid, cls_count, status_word(s));
  BL_HIP_TRY(hipGetLastError());
  return BL_OK;
}

}  // namespace blh

namespace bl {
// diagnostic: the fitted saddle-point inversion (sp_vlk: v(x), -log cos_rt v, log K2 -- what the attempt body evaluates where
// the reference runs InvertY.cpp:57-99's table bracket + Newton solve), table staged in LDS exactly as k_rpg_tasks<SpPolicy> does
__global__ __launch_bounds__(256) void k_diag_sp_vlk(double* __restrict__ out, const double* __restrict__ x, int64_t num)
{
  __shared__ double sVt[kVtabDoubles];
  const double* src = &kVtab[0][0][0];
  for (int i = threadIdx.x; i < kVtabDoubles; i += 256) sVt[i] = src[i];
  __syncthreads();
  for (int64_t i0 = (int64_t)blockIdx.x * 256; i0 < num; i0 += (int64_t)gridDim.x * 256) {
    const int64_t i = i0 + threadIdx.x;
    const double xx = i < num ? x[i] : 1.0;
    double v, L, lK2;
    sp_vlk(sVt, xx, log(xx), v, L, lK2);
    if (i < num) { out[3 * i] = v; out[3 * i + 1] = L; out[3 * i + 2] = lK2; }
  }
}

// diagnostic: how much work the draws of a vector are -- per sampler class {observations, PG draws (Devroye: PG(1,z) draws;
// alternating series: abridged draws; saddle point: 1), Philox blocks = proposal attempts}.  A replay of every observation's
// stream by the same attempt bodies, one observation per lane, no queue (slow, exact): the counts are those of the
// production kernels, whose draws are a function of the stream alone.  h == nullptr: rpg_devroye with n = 1 (C2).
__global__ __launch_bounds__(256) void k_diag_blocks(const double* __restrict__ h, const double* __restrict__ z, int64_t num,
                                                     uint64_t seed, uint32_t epoch, uint64_t idx0,
                                                     unsigned long long* __restrict__ out)
{
  __shared__ double sVt[kVtabDoubles];
  const double* src = &kVtab[0][0][0];
  for (int i = threadIdx.x; i < kVtabDoubles; i += 256) sVt[i] = src[i];
  __syncthreads();
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  int st = 0;
  for (int64_t i0 = (int64_t)blockIdx.x * 256; i0 < num; i0 += (int64_t)gridDim.x * 256) {
    const int64_t i = i0 + threadIdx.x;
    int cls = -1;
    bool wide = false;                       // Devroye with |z|/2 >= 1/t: the other left-piece sampler (PolyaGamma.cpp:103)
    unsigned long long draws = 0, blocks = 0;
    if (i < num) {
      const double hh = h ? h[i] : 1.0, zz = z[i];
      const uint64_t idx = idx0 + (uint64_t)i;
      cls = h ? pg_class(hh) : CLS_DEVROYE;
      if (cls == CLS_DEVROYE) {
        const Pg1Par p = pg1_par(zz);
        wide = !(kSmTRecip > p.Z);
        Pg1Lane s{true, 0.0};
        int n = (int)hh;
        draws = (unsigned long long)n;
        for (uint32_t blk = 0; n > 0 && blk < 100000u; ++blk) {
          const U4 o = philox4x32_10((uint32_t)idx, ctr1_of(idx, DOM_DRAW), epoch, blk, k0, k1);
          ++blocks;
          if (pg1_attempt(s, p, u52(o.x, o.y), u52(o.z, o.w), st)) --n;
        }
      } else if (cls == CLS_ALT) {
        int nA, nB;
        double hB;
        alt_groups(hh, nA, hB, nB);
        draws = (unsigned long long)(nA + nB);
        AltTask T;
        if (nA > 0) {
          alt_task_start(T, alt_par(4.0, zz, alt_trunc_of(kTruncSchedule, 4.0), st), nA, idx, DOM_DRAW, 0u);
          while (!alt_task_step(T, epoch, k0, k1, st)) {}
          blocks += T.blk;
        }
        alt_task_start(T, alt_par(hB, zz, alt_trunc_of(kTruncSchedule, hB), st), nB, idx, DOM_DRAW, kAltBlkGroupB);
        while (!alt_task_step(T, epoch, k0, k1, st)) {}
        blocks += T.blk - kAltBlkGroupB;
      } else if (cls == CLS_SP) {
        draws = 1;
        SpTask T;
        sp_task_start(T, sp_par(hh, zz, sVt, st), idx, DOM_DRAW);
        while (!sp_task_step(T, sVt, 200, epoch, k0, k1, st)) {}
        blocks = T.blk;
      } else if (cls == CLS_NORMAL || cls == CLS_GAMMA) {
        draws = 1;
      }
    }
    // per class: wave totals, one atomic each
#pragma unroll 1
    for (int c = 0; c < 7; ++c) {
      const bool mine = c < 6 ? cls == c : (cls == CLS_DEVROYE && wide);
      const uint64_t m = __ballot(mine);
      if (m == 0) continue;
      unsigned long long d = mine ? draws : 0, b = mine ? blocks : 0;
      for (int off = 32; off > 0; off >>= 1) {
        d += __shfl_down(d, off);
        b += __shfl_down(b, off);
      }
      if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out[3 * c], (unsigned long long)__popcll(m));
        atomicAdd(&out[3 * c + 1], d);
        atomicAdd(&out[3 * c + 2], b);
      }
    }
  }
}
}  // namespace bl

extern "C" {

int bl_diag_count_blocks_dev(const double* h, const double* z, int64_t num, uint64_t seed, uint32_t epoch, uint64_t idx0,
                             unsigned long long* out21, void* stream)
{
  if (!blh::ensure_device()) return BL_ERR_NO_DEVICE;
  if (num < 0 || !out21 || (num > 0 && !z)) { blh::set_error("null pointer or negative length"); return BL_ERR_ARG; }
  hipStream_t s = (hipStream_t)stream;
  BL_HIP_TRY(hipMemsetAsync(out21, 0, 21 * sizeof(unsigned long long), s));
  if (num == 0) return BL_OK;
  const int64_t blocks = (num + 255) / 256;
  hipLaunchKernelGGL(bl::k_diag_blocks, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, s, h, z, num, seed, epoch,
                     idx0, out21);
  BL_HIP_TRY(hipGetLastError());
  return BL_OK;
}

int bl_diag_sp_vlk_dev(double* out3, const double* x, int64_t num, void* stream)
{
  if (!blh::ensure_device()) return BL_ERR_NO_DEVICE;
  if (num < 0 || (num > 0 && (!out3 || !x))) { blh::set_error("null pointer or negative length"); return BL_ERR_ARG; }
  if (num == 0) return BL_OK;
  const int64_t blocks = (num + 255) / 256;
  hipLaunchKernelGGL(bl::k_diag_sp_vlk, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, (hipStream_t)stream, out3, x, num);
  BL_HIP_TRY(hipGetLastError());
  return BL_OK;
}

int bl_rpg_alt_dev(double* x, const double* h, const double* z, int64_t num, uint64_t seed, uint32_t epoch,
                   uint64_t idx0, void* stream)
{
  if (!blh::ensure_device()) return BL_ERR_NO_DEVICE;
  if (num < 0 || (num > 0 && (!x || !h || !z))) { blh::set_error("null pointer or negative length"); return BL_ERR_ARG; }
  if (num == 0) return BL_OK;
  hipStream_t s = (hipStream_t)stream;
  // h == 0 -> 0 (LogitWrapper.cpp:95-98): written by the kernel's own scan, like the zero a two-task observation's sums are added to
  return blh::launch_rpg_tasks(false, x, h, z, num, nullptr, seed, epoch, idx0, blh::kHybNone, nullptr, s);
}

int bl_rpg_sp_dev(double* x, const double* h, const double* z, int64_t num, int* iter, uint64_t seed,
                  uint32_t epoch, uint64_t idx0, void* stream)
{
  if (!blh::ensure_device()) return BL_ERR_NO_DEVICE;
  if (num < 0 || (num > 0 && (!x || !h || !z))) { blh::set_error("null pointer or negative length"); return BL_ERR_ARG; }
  if (num == 0) return BL_OK;
  hipStream_t s = (hipStream_t)stream;
  // h == 0 -> 0, iter untouched (LogitWrapper.cpp:116-120): the zeros are written by the kernel's own scan
  return blh::launch_rpg_tasks(true, x, h, z, num, iter, seed, epoch, idx0, blh::kHybNone, nullptr, s);
}

}  // extern "C"
