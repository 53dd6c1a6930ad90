// Which lane holds which element of v_mfma_f64_4x4x4_4b_f64's operands (gfx950): unit vectors in A (lane p) and B
// (lane q), one wave per (p, q); for every lane of D the (p, q) pairs that reach it are printed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k(double* d)
{
  const int lane = threadIdx.x, p = blockIdx.x >> 6, q = blockIdx.x & 63;
  d[(size_t)blockIdx.x * 64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(lane == p ? 1.0 : 0.0, lane == q ? 1.0 : 0.0, 0.0, 0, 0, 0);
}
int main()
{
  double* d;
  double* h = (double*)malloc(4096 * 64 * 8);
  (void)hipMalloc(&d, 4096 * 64 * 8);
  hipLaunchKernelGGL(k, dim3(4096), dim3(64), 0, 0, d);
  (void)hipMemcpy(h, d, 4096 * 64 * 8, hipMemcpyDeviceToHost);
  for (int L = 0; L < 64; ++L) {
    printf("D lane %2d <-", L);
    for (int pq = 0; pq < 4096; ++pq)
      if (h[(size_t)pq * 64 + L] != 0.0) printf(" (A%d,B%d)", pq >> 6, pq & 63);
    printf("\n");
  }
  return 0;
}
