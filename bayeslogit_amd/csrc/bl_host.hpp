// bl_host.hpp -- host-side plumbing shared by the translation units of
// libbayeslogit_hip.so: error/status reporting, process-global seed state, a
// device scratch helper.  No CPU compute path lives here or anywhere in csrc/:
// every entry point either runs HIP kernels or fails with a status.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/bayeslogit_hip.h"
#include "../../include/bayeslogit_hip_diag.h"

namespace blh {

void set_error(const std::string& msg);
bool ensure_device();                       // false => BL_ERR_NO_DEVICE recorded
int* status_word(hipStream_t s);            // device int, zeroed per sync
int  collect_status(hipStream_t s);         // sync + read + reset flags

// kernels_tasks.hip: the alternating-series (sp == false) or saddle-point work-queue kernel over (h, z).  hybrid: 0 =
// rpg_alt / rpg_sp (every h != 0 is a member; the kernel writes every element of x itself: no zeroing launch in front);
// kHybFirst = the first class pass of rpg_hybrid: it also counts the members of every class into cls_count[6] and writes
// the zeros of the b <= 0 branch; kHybLater = a later class pass: returns at once when cls_count says its class is empty.
enum : int { kHybNone = 0, kHybFirst = 1, kHybLater = 2, kHybPlain = 3 };   // kHybPlain: a class pass that does not look at the counts
int launch_rpg_tasks(bool sp, double* x, const double* h, const double* z, int64_t num, int* iter, uint64_t seed,
                     uint32_t epoch, uint64_t idx0, int hybrid, unsigned long long* cls_count, hipStream_t s);
unsigned long long* class_counts_slot();    // 8 device counters for one rpg_hybrid call (a ring of 64 slots)

uint64_t global_seed();
uint32_t next_epoch();                      // returns current, then increments
int      global_constrain();
int      sweep_single_pass();               // bl_set_sweep_mode / BL_SWEEP_SINGLE_PASS (default 1)
int      beta_sweeps_kind();                // bl_diag_beta_sweeps / BL_BETA_SPLIT (default 1: row-split sweeps)
unsigned long long* sweep_stats();          // device counters of the single-pass sweep (bl_diag_sweep_deferred)

// host (pageable, caller-owned) -> device; large buffers go through pinned staging (host_state.hip).  Synchronous for
// large copies (returns when the data is on the device), asynchronous on `s` for small ones -- as hipMemcpyAsync from
// pageable memory is.
hipError_t upload_staged(void* dst_dev, const void* src_host, size_t bytes, hipStream_t s);

#define BL_HIP_TRY(expr)                                                            \
  do {                                                                              \
    hipError_t _e = (expr);                                                         \
    if (_e != hipSuccess) {                                                         \
      blh::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));            \
      return BL_ERR_HIP;                                                            \
    }                                                                               \
  } while (0)

// RAII device buffer for the host-pointer (.C) entry points
template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t count)
  {
    n = count;
    return hipMalloc((void**)&p, sizeof(T) * (count ? count : 1));
  }
  hipError_t upload(const T* host, hipStream_t s = nullptr)
  {
    return upload_staged(p, host, sizeof(T) * n, s);
  }
  hipError_t download(T* host, hipStream_t s = nullptr) const
  {
    return hipMemcpyAsync(host, p, sizeof(T) * n, hipMemcpyDeviceToHost, s);
  }
};

inline int grid_for(int64_t n, int block, int max_blocks)
{
  int64_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > max_blocks) g = max_blocks;
  return (int)g;
}

}  // namespace blh
