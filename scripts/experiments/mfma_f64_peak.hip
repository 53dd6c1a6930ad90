// Micro-benchmark: sustained issue rate of v_mfma_f64_16x16x4_f64 on gfx950, no memory traffic.
// NACC independent accumulators per wave, W waves per SIMD; prints cycles per MFMA per SIMD and TFLOP/s.
// Build and run on the GPU box:  hipcc --offload-arch=gfx950 -O3 mfma_f64_peak.hip -o /tmp/mfma && /tmp/mfma
// (DESIGN.md 4.3 quotes its output next to pass 2's measured rate.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC, bool MUL>
__global__ __launch_bounds__(256) void k_peak(double* out, int iters, double a0, double b0)
{
  d4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  double w = 1.0 + threadIdx.x * 1e-12;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      double ai = a;
      if (MUL && (i & 3) == 0) {   // pass 2 forms A = omega * x: four multiplies per ten MFMAs
        a = a * w;
        ai = a;
      }
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, b, acc[i], 0, 0, 0);
    }
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, bool MUL>
static void run(int wg_per_cu, int iters)
{
  int dev = 0;
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, dev);
  const int cus = prop.multiProcessorCount;
  const int grid = cus * wg_per_cu;
  double* out;
  hipMalloc(&out, (size_t)grid * 256 * sizeof(double));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k_peak<NACC, MUL>), dim3(grid), dim3(256), 0, 0, out, iters / 10, 1.0, 1.0);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k_peak<NACC, MUL>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 1.0);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = (double)iters * NACC * wg_per_cu;   // one wave of each workgroup per SIMD
  const double clk_ghz = prop.clockRate * 1e-6;
  const double cyc = ms * 1e-3 * clk_ghz * 1e9 / mfma_per_simd;
  const double tflops = mfma_per_simd * cus * 4 * 2048.0 / (ms * 1e-3) / 1e12;
  printf("NACC=%2d mul=%d waves/SIMD=%d: %.3f ms, %.1f cycles/MFMA/SIMD at %.2f GHz nominal, %.1f TFLOP/s\n", NACC,
         (int)MUL, wg_per_cu, ms, cyc, clk_ghz, tflops);
  hipFree(out);
}

int main()
{
  const int iters = 20000;
  run<10, false>(1, iters);
  run<10, false>(2, iters);
  run<4, false>(1, iters);
  run<4, false>(2, iters);
  run<2, false>(2, iters);
  run<1, false>(1, iters);
  run<10, true>(2, iters);
  run<16, false>(2, iters);
  run<10, false>(3, iters);
  run<10, false>(4, iters);
  run<4, false>(4, iters);
  run<4, false>(8, iters);
  return 0;
}
