// h2d_staging.hip -- what a host -> device upload of a large pageable buffer costs on the GPU box, three ways:
//   (a) hipMemcpy from the pageable buffer (what the .C entry points did through round 2),
//   (b) chunks through two pinned staging buffers: memcpy by T threads into one while the DMA engine drains the other,
//   (c) hipHostRegister of the caller's buffer, one async copy, unregister.
// and the same for device -> host.   hipcc --offload-arch=gfx950 -O2 h2d_staging.hip -o h2d_staging -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void par_memcpy(char* d, const char* s, size_t n, int T)
{
  if (T <= 1) { memcpy(d, s, n); return; }
  std::vector<std::thread> th;
  const size_t per = (n + T - 1) / T;
  for (int t = 0; t < T; ++t) {
    const size_t o = (size_t)t * per;
    if (o >= n) break;
    th.emplace_back([=] { memcpy(d + o, s + o, (o + per <= n) ? per : n - o); });
  }
  for (auto& x : th) x.join();
}
int main(int argc, char** argv)
{
  const size_t bytes = (size_t)(argc > 1 ? atof(argv[1]) : 2e9);
  char* h = (char*)malloc(bytes);
  memset(h, 1, bytes);
  char* d;
  hipMalloc((void**)&d, bytes);
  hipMemcpy(d, h, 1 << 20, hipMemcpyHostToDevice);
  double t0 = now();
  hipMemcpy(d, h, bytes, hipMemcpyHostToDevice);
  printf("H2D pageable hipMemcpy:            %.2f GB/s\n", bytes / (now() - t0) / 1e9);
  t0 = now();
  hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost);
  printf("D2H pageable hipMemcpy:            %.2f GB/s\n", bytes / (now() - t0) / 1e9);
  for (int T : {1, 4, 8}) {
    for (size_t chunk : {(size_t)16 << 20, (size_t)64 << 20}) {
      char* pin[2];
      hipEvent_t ev[2];
      hipStream_t s;
      hipStreamCreate(&s);
      for (int i = 0; i < 2; ++i) { hipHostMalloc((void**)&pin[i], chunk, 0); hipEventCreate(&ev[i]); }
      t0 = now();
      int b = 0;
      for (size_t o = 0; o < bytes; o += chunk, b ^= 1) {
        const size_t n = o + chunk <= bytes ? chunk : bytes - o;
        hipEventSynchronize(ev[b]);
        par_memcpy(pin[b], h + o, n, T);
        hipMemcpyAsync(d + o, pin[b], n, hipMemcpyHostToDevice, s);
        hipEventRecord(ev[b], s);
      }
      hipStreamSynchronize(s);
      printf("H2D staged %2d threads, %3zu MB chunks: %.2f GB/s\n", T, chunk >> 20, bytes / (now() - t0) / 1e9);
      t0 = now();
      b = 0;
      size_t prev_o[2] = {0, 0}, prev_n[2] = {0, 0};
      for (size_t o = 0; o < bytes; o += chunk, b ^= 1) {
        const size_t n = o + chunk <= bytes ? chunk : bytes - o;
        if (prev_n[b]) { hipEventSynchronize(ev[b]); par_memcpy(h + prev_o[b], pin[b], prev_n[b], T); }
        hipMemcpyAsync(pin[b], d + o, n, hipMemcpyDeviceToHost, s);
        hipEventRecord(ev[b], s);
        prev_o[b] = o; prev_n[b] = n;
      }
      for (int i = 0; i < 2; ++i, b ^= 1) if (prev_n[b]) { hipEventSynchronize(ev[b]); par_memcpy(h + prev_o[b], pin[b], prev_n[b], T); }
      printf("D2H staged %2d threads, %3zu MB chunks: %.2f GB/s\n", T, chunk >> 20, bytes / (now() - t0) / 1e9);
      for (int i = 0; i < 2; ++i) { hipHostFree(pin[i]); hipEventDestroy(ev[i]); }
      hipStreamDestroy(s);
    }
  }
  t0 = now();
  hipHostRegister(h, bytes, hipHostRegisterDefault);
  double t1 = now();
  hipMemcpy(d, h, bytes, hipMemcpyHostToDevice);
  double t2 = now();
  hipHostUnregister(h);
  printf("H2D register %.3f s + copy %.3f s (%.2f GB/s) + unregister %.3f s: %.2f GB/s overall\n", t1 - t0, t2 - t1,
         bytes / (t2 - t1) / 1e9, now() - t2, bytes / (now() - t0) / 1e9);
  return 0;
}
