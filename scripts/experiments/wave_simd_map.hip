// wave_simd_map.hip -- which SIMD of its CU each wave of a 512-thread workgroup lands on (HW_REG_HW_ID bits 5:4 on gfx9 parts),
// for workgroups of 8 waves with ~110 KB of LDS (one per CU), as the P = 256 sweep kernels are launched.
//   hipcc --offload-arch=gfx950 -O2 wave_simd_map.hip -o wave_simd_map && ./wave_simd_map
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512, 2) void k(unsigned* out)
{
  extern __shared__ char lds[];
  lds[threadIdx.x] = 0;
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
}
int main()
{
  unsigned* d;
  const int nb = 512;
  hipMalloc(&d, nb * 8 * sizeof(unsigned));
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 110000);
  hipLaunchKernelGGL(k, dim3(nb), dim3(512), 110000, 0, d);
  unsigned h[nb * 8];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int hist[8][4] = {};
  for (int b = 0; b < nb; ++b)
    for (int w = 0; w < 8; ++w) hist[w][(h[b * 8 + w] >> 4) & 3]++;
  for (int w = 0; w < 8; ++w) printf("wave %d: SIMD0 %d SIMD1 %d SIMD2 %d SIMD3 %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
  for (int b = 0; b < 3; ++b) {
    printf("block %d:", b);
    for (int w = 0; w < 8; ++w) printf(" w%d->simd%u(cu%u,wave_slot%u)", w, (h[b * 8 + w] >> 4) & 3, (h[b * 8 + w] >> 8) & 15, h[b * 8 + w] & 15);
    printf("\n");
  }
  return 0;
}
