// Does a wave's re-read of a chunk it streamed a moment ago come from the Infinity Cache at a rate that adds to the HBM
// stream?  (Design question of the fused Gibbs sweep: pass 2 re-reading X behind pass 1 in the same launch.)
//   mode 0: every wave streams its slice once                      (bytes = B)
//   mode 1: chunk by chunk, read the chunk then read it again       (bytes = 2B, second read <= 1 chunk old)
//   mode 2: read chunk c+1, then re-read chunk c                    (bytes = 2B, second read 1-2 chunks old)
// hipcc --offload-arch=gfx950 -O3 -o mall_reread mall_reread.hip && ./mall_reread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef double v2d __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256, 3) void k(const v2d* __restrict__ x, size_t n_per_wave, int chunk_v2, double* out)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t w = (size_t)blockIdx.x * 4 + wave;
  const v2d* p = x + w * n_per_wave;
  double s0 = 0, s1 = 0;
  const size_t nchunks = n_per_wave / chunk_v2;
  for (size_t c = 0; c < nchunks; ++c) {
    const v2d* q = p + c * chunk_v2;
#pragma unroll 8
    for (int i = lane; i < chunk_v2; i += 64) { v2d v = q[i]; s0 += v.x; s1 += v.y; }
    if (MODE == 1) {
#pragma unroll 8
      for (int i = lane; i < chunk_v2; i += 64) { v2d v = q[i]; s0 -= v.x; s1 -= v.y; }
    }
    if (MODE == 2 && c > 0) {
      const v2d* r = q - chunk_v2;
#pragma unroll 8
      for (int i = lane; i < chunk_v2; i += 64) { v2d v = r[i]; s0 -= v.x; s1 -= v.y; }
    }
  }
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s0 + s1;
}

int main()
{
  const size_t bytes = 5120000000ull;
  v2d* x; double* out;
  hipMalloc(&x, bytes); hipMalloc(&out, 8 * 256 * 4096);
  hipMemset(x, 0, bytes);
  for (int wgs_per_cu : {2, 3}) for (int chunk_kb : {32, 64, 128}) {
    const int grid = 256 * wgs_per_cu;
    const size_t waves = (size_t)grid * 4;
    const int chunk_v2 = chunk_kb * 1024 / 16;
    size_t n_per_wave = bytes / 16 / waves / chunk_v2 * chunk_v2;
    for (int mode = 0; mode < 3; ++mode) {
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      float best = 1e9;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(a);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, x, n_per_wave, chunk_v2, out);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, x, n_per_wave, chunk_v2, out);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, x, n_per_wave, chunk_v2, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
      }
      const double gb = (double)n_per_wave * 16 * waves / 1e9;
      printf("wg/cu %d chunk %3d KB mode %d: %.3f ms  first-read %.2f TB/s  total %.2f TB/s\n", wgs_per_cu, chunk_kb, mode, best,
             gb / best, gb * (mode ? 2 : 1) / best);
    }
  }
  return 0;
}
