// host_state.hip -- process-global state of libbayeslogit_hip.so: last error text,
// sampler status word, seed/epoch of the .C-style entry points.
#include "bl_host.hpp"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace {
std::mutex g_mu;
std::string g_err;
int g_flags = 0;
uint64_t initial_seed()
{
  if (const char* s = getenv("BAYESLOGIT_SEED")) return strtoull(s, nullptr, 0);
  return 0x42A7E5105EEDull;
}
std::atomic<uint64_t> g_seed{initial_seed()};
std::atomic<uint32_t> g_epoch{0};
std::atomic<int> g_constrain{1};
int initial_sweep_mode()
{
  if (const char* s = getenv("BL_SWEEP_SINGLE_PASS")) return atoi(s) ? 1 : 0;
  return 1;
}
static int initial_beta_sweeps()
{
  const char* e = getenv("BL_BETA_SPLIT");
  return (e && e[0] == '0') ? 0 : 1;
}
std::atomic<int> g_beta_sweeps{initial_beta_sweeps()};
std::atomic<int> g_sweep_mode{initial_sweep_mode()};
unsigned long long* g_sweep_stats_dev = nullptr;     // [0] rows that left the single-pass sweep's fast path
int* g_status_dev = nullptr;
unsigned long long* g_class_counts_dev = nullptr;    // 64 slots of 8: per-class member counts of one rpg_hybrid call
std::atomic<unsigned> g_class_slot{0};
bool g_dev_ok = false, g_dev_tried = false;
}  // namespace

namespace blh {

void set_error(const std::string& msg)
{
  std::lock_guard<std::mutex> l(g_mu);
  g_err = msg;
  fprintf(stderr, "bayeslogit_hip: %s\n", msg.c_str());
}

bool ensure_device()
{
  std::lock_guard<std::mutex> l(g_mu);
  if (g_dev_tried) return g_dev_ok;
  g_dev_tried = true;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n < 1) {
    g_err = "no HIP device available (libbayeslogit_hip has no CPU fallback)";
    fprintf(stderr, "bayeslogit_hip: %s\n", g_err.c_str());
    return false;
  }
  e = hipMalloc((void**)&g_status_dev, sizeof(int));
  if (e == hipSuccess) e = hipMemset(g_status_dev, 0, sizeof(int));
  if (e == hipSuccess) e = hipMalloc((void**)&g_sweep_stats_dev, 16 * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMemset(g_sweep_stats_dev, 0, 16 * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMalloc((void**)&g_class_counts_dev, 64 * 8 * sizeof(unsigned long long));
  if (e != hipSuccess) {
    g_err = std::string("hipMalloc(status): ") + hipGetErrorString(e);
    fprintf(stderr, "bayeslogit_hip: %s\n", g_err.c_str());
    return false;
  }
  g_dev_ok = true;
  return true;
}

int* status_word(hipStream_t) { return g_status_dev; }

int collect_status(hipStream_t s)
{
  int st = 0;
  hipError_t e = hipStreamSynchronize(s);
  if (e == hipSuccess) e = hipMemcpy(&st, g_status_dev, sizeof(int), hipMemcpyDeviceToHost);
  if (e == hipSuccess && st != 0) e = hipMemset(g_status_dev, 0, sizeof(int));
  if (e != hipSuccess) {
    set_error(std::string("status sync: ") + hipGetErrorString(e));
    return BL_ERR_HIP;
  }
  {
    std::lock_guard<std::mutex> l(g_mu);
    g_flags = st;
  }
  if (st & BL_ST_NOT_PD) {
    set_error("posterior precision matrix is not positive definite (Cholesky factorisation failed)");
    return BL_ERR_NOT_PD;
  }
  if (st != 0) {
    set_error("sampler flags raised: " + std::to_string(st) +
              " (1 = iteration cap, 2 = bad shape, 4 = alt sampler fall-through)");
    return BL_ERR_SAMPLER;
  }
  return BL_OK;
}

uint64_t global_seed() { return g_seed.load(); }
uint32_t next_epoch() { return g_epoch.fetch_add(1); }
int global_constrain() { return g_constrain.load(); }
int sweep_single_pass() { return g_sweep_mode.load(); }
int beta_sweeps_kind() { return g_beta_sweeps.load(); }
// ---- host -> device upload of a caller-owned (pageable) buffer, as the .C boundary hands them over.
// hipMemcpy from pageable memory reaches 24 GB/s on the MI355X box; copying 64 MB chunks into one of two pinned
// staging buffers with four threads while the DMA engine drains the other reaches 54 GB/s
// (scripts/experiments/h2d_staging.hip, profiles/r03_h2d_staging.txt; device -> host is 53 GB/s either way, and
// hipHostRegister of the caller's pages costs more than it saves).  Small copies take the plain path.
namespace {
constexpr size_t kStageChunk = (size_t)64 << 20;
constexpr int kStageThreads = 4;
std::mutex g_stage_mu;
char* g_stage[2] = {nullptr, nullptr};
hipEvent_t g_stage_ev[2];
hipStream_t g_stage_stream = nullptr;
void par_memcpy(char* d, const char* s, size_t n)
{
  std::vector<std::thread> th;
  const size_t per = ((n + kStageThreads - 1) / kStageThreads + 4095) & ~(size_t)4095;
  for (int t = 1; t < kStageThreads; ++t) {
    const size_t o = (size_t)t * per;
    if (o >= n) break;
    th.emplace_back([=] { memcpy(d + o, s + o, (o + per <= n) ? per : n - o); });
  }
  memcpy(d, s, per < n ? per : n);
  for (auto& x : th) x.join();
}
}  // namespace

hipError_t upload_staged(void* dst_dev, const void* src_host, size_t bytes, hipStream_t user_stream)
{
  if (bytes < 2 * kStageChunk) return hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, user_stream);
  std::lock_guard<std::mutex> l(g_stage_mu);
  hipError_t e = hipSuccess;
  if (!g_stage[0]) {
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
      e = hipHostMalloc((void**)&g_stage[i], kStageChunk, 0);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&g_stage_ev[i], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&g_stage_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {                                  // no pinned memory to be had: the plain path
      g_stage[0] = nullptr;
      return hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, user_stream);
    }
  }
  // work already queued on the caller's stream may still read the destination: the copies start behind it
  hipEvent_t before;
  e = hipEventCreateWithFlags(&before, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventRecord(before, user_stream);
  if (e == hipSuccess) e = hipStreamWaitEvent(g_stage_stream, before, 0);
  int b = 0;
  for (size_t o = 0; o < bytes && e == hipSuccess; o += kStageChunk, b ^= 1) {
    const size_t n = o + kStageChunk <= bytes ? kStageChunk : bytes - o;
    e = hipEventSynchronize(g_stage_ev[b]);                 // the buffer's previous copy has left
    if (e != hipSuccess) break;
    par_memcpy(g_stage[b], (const char*)src_host + o, n);
    e = hipMemcpyAsync((char*)dst_dev + o, g_stage[b], n, hipMemcpyHostToDevice, g_stage_stream);
    if (e == hipSuccess) e = hipEventRecord(g_stage_ev[b], g_stage_stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(g_stage_stream);   // the staging buffers are free again; the data is on the device
  (void)hipEventDestroy(before);
  return e;
}

unsigned long long* sweep_stats() { return g_sweep_stats_dev; }
unsigned long long* class_counts_slot() { return g_class_counts_dev + 8 * (g_class_slot.fetch_add(1) & 63u); }

}  // namespace blh

extern "C" {

const char* bl_last_error(void)
{
  static thread_local std::string copy;
  std::lock_guard<std::mutex> l(g_mu);
  copy = g_err;
  return copy.c_str();
}

int bl_last_sampler_flags(void)
{
  std::lock_guard<std::mutex> l(g_mu);
  return g_flags;
}

int bl_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int bl_set_device(int device)
{
  if (g_dev_tried && g_dev_ok) {
    blh::set_error("bl_set_device must be called before any other entry point");
    return BL_ERR_ARG;
  }
  if (hipSetDevice(device) != hipSuccess) {
    blh::set_error("hipSetDevice failed");
    return BL_ERR_NO_DEVICE;
  }
  return blh::ensure_device() ? BL_OK : BL_ERR_NO_DEVICE;
}

void bl_set_seed(uint64_t seed)
{
  g_seed = seed;
  g_epoch = 0;
}
uint64_t bl_get_seed(void) { return g_seed.load(); }
void bl_set_seed_from_unif(double* u)
{
  const uint64_t hi = (uint64_t)(u[0] * 4294967296.0) & 0xFFFFFFFFull, lo = (uint64_t)(u[1] * 4294967296.0) & 0xFFFFFFFFull;
  bl_set_seed((hi << 32) | lo);
}
uint32_t bl_get_epoch(void) { return g_epoch.load(); }
void bl_set_constrain(int c) { g_constrain = c ? 1 : 0; }
void bl_set_constrain_R(int* c) { g_constrain = (c && *c) ? 1 : 0; }
void bl_set_sweep_mode(int single_pass) { g_sweep_mode = single_pass ? 1 : 0; }
void bl_diag_beta_sweeps(int row_split) { g_beta_sweeps = row_split ? 1 : 0; }
int bl_diag_sweep_deferred(uint64_t* rows)
{
  if (!blh::ensure_device()) return BL_ERR_NO_DEVICE;
  unsigned long long v[16] = {0};
  if (hipDeviceSynchronize() != hipSuccess ||
      hipMemcpy(v, g_sweep_stats_dev, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemset(g_sweep_stats_dev, 0, sizeof(v)) != hipSuccess) {
    blh::set_error("bl_diag_sweep_deferred: HIP call failed");
    return BL_ERR_HIP;
  }
  if (rows) *rows = v[0];
  return BL_OK;
}
void bl_set_device_R(int* device, int* rc)
{
  const int r = bl_set_device(device ? *device : 0);
  if (rc) *rc = r;
}

int bl_sync_status(void* stream)
{
  if (!blh::ensure_device()) return BL_ERR_NO_DEVICE;
  return blh::collect_status((hipStream_t)stream);
}

}  // extern "C"
