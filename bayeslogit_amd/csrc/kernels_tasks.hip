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

extern "C" {

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
