// kernels_tasks.hip -- rpg_alt, rpg_sp and the two large classes of rpg_hybrid
// (Code/C/LogitWrapper.cpp:87-167) on the task work queue of bl_task_queue.hpp.  gfx950 only.
// Built with machine-LICM off (bayeslogit_amd/build.py): hoisting the fp64 polynomial constants of the
// attempt body and of the set-up out of the queue loop costs 30-80 registers per lane and a wave per SIMD.
#include "bl_host.hpp"
#include "bl_task_queue.hpp"

namespace blh {

int launch_rpg_tasks(bool sp, double* x, const double* h, const double* z, int64_t num, int* iter, uint64_t seed,
                     uint32_t epoch, uint64_t idx0, int hybrid, hipStream_t s)
{
  // the resident grid (3 workgroups per CU), or fewer when the vector is short
  const int64_t chunks = (num + bl::kTqChunk - 1) / bl::kTqChunk;
  const dim3 g(grid_for(chunks, bl::kTqBlock / 64, 256 * 3)), b(bl::kTqBlock);
  if (sp)
    hipLaunchKernelGGL(bl::k_rpg_tasks<bl::SpPolicy>, g, b, 0, s, x, h, z, num, iter, seed, epoch, idx0, hybrid, status_word(s));
  else
    hipLaunchKernelGGL(bl::k_rpg_tasks<bl::AltPolicy>, g, b, 0, s, x, h, z, num, iter, seed, epoch, idx0, hybrid, status_word(s));
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
}  // namespace bl

extern "C" {

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
  // h == 0 -> 0 (LogitWrapper.cpp:95-98); the tasks of an observation add their sums into x
  BL_HIP_TRY(hipMemsetAsync(x, 0, sizeof(double) * (size_t)num, s));
  return blh::launch_rpg_tasks(false, x, h, z, num, nullptr, seed, epoch, idx0, 0, s);
}

int bl_rpg_sp_dev(double* x, const double* h, const double* z, int64_t num, int* iter, uint64_t seed,
                  uint32_t epoch, uint64_t idx0, void* stream)
{
  if (!blh::ensure_device()) return BL_ERR_NO_DEVICE;
  if (num < 0 || (num > 0 && (!x || !h || !z))) { blh::set_error("null pointer or negative length"); return BL_ERR_ARG; }
  if (num == 0) return BL_OK;
  hipStream_t s = (hipStream_t)stream;
  // h == 0 -> 0, iter untouched (LogitWrapper.cpp:116-120)
  BL_HIP_TRY(hipMemsetAsync(x, 0, sizeof(double) * (size_t)num, s));
  return blh::launch_rpg_tasks(true, x, h, z, num, iter, seed, epoch, idx0, 0, s);
}

}  // extern "C"
