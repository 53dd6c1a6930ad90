// Micro-benchmark (gfx950, no memory traffic): issue cost of the two fp64 matrix instructions and whether fp64
// vector work runs beside them.
//   (a) v_mfma_f64_16x16x4_f64 (2048 flops) and v_mfma_f64_4x4x4_4b_f64 (512 flops), NACC independent accumulators,
//       W waves per SIMD: shader cycles per instruction per SIMD (s_memtime) and the clock the chip holds
//       (s_memtime / s_memrealtime);
//   (c) the 4x4x4 loop with its B operand read from LDS by one ds_read_b64 per matrix instruction (the P = 256 kernel's mix);
//   (b) the 16x16x4 loop with F independent v_fma_f64 per matrix instruction in the same wave: if the vector FMAs ran
//       in the matrix instruction's shadow the cycles per iteration would stay put; they add up instead.
// The cycle counts are each wave's own (s_memtime around its loop) divided by the waves per SIMD: the waves of a SIMD are
// not in step, so the absolute figures flatter both instructions (event-timed rates: scripts/gpu_mfma_rates.py: 48 and
// 75 TFLOP/s); the comparisons between variants hold.
// Build here (hipcc cross-compiles), run on the GPU box:  scripts/experiments/mfma_f64_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int NACC, int F, int L = 0>
__global__ __launch_bounds__(256) void k(double* out, unsigned long long* clk, int iters, double a0)
{
  __shared__ double sb[4][64 * 17];
  if (L) { for (int i = threadIdx.x; i < 4 * 64 * 17; i += 256) (&sb[0][0])[i] = 1.0 + i * 1e-9; __syncthreads(); }
  const double* lp = &sb[threadIdx.x >> 6][threadIdx.x & 63];
  d4 acc16[SHAPE == 16 ? NACC : 1];
  double acc4[SHAPE == 4 ? NACC : 1];
  double f[F > 0 ? F : 1];
#pragma unroll
  for (int i = 0; i < (SHAPE == 16 ? NACC : 1); ++i) acc16[i] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int i = 0; i < (SHAPE == 4 ? NACC : 1); ++i) acc4[i] = 0.0;
#pragma unroll
  for (int i = 0; i < (F > 0 ? F : 1); ++i) f[i] = a0 + i;
  const double a = a0 + threadIdx.x * 1e-9, b = 1.0 + threadIdx.x * 1e-10;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (SHAPE == 16) acc16[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc16[i], 0, 0, 0);
      if (SHAPE == 4) acc4[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, L ? lp[64 * ((i + it) % 16)] : b, acc4[i], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < F; ++j) f[j] = __builtin_fma(f[j], b, a);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < (SHAPE == 16 ? NACC : 1); ++i) s += acc16[i][0] + acc16[i][1] + acc16[i][2] + acc16[i][3];
#pragma unroll
  for (int i = 0; i < (SHAPE == 4 ? NACC : 1); ++i) s += acc4[i];
#pragma unroll
  for (int i = 0; i < (F > 0 ? F : 1); ++i) s += f[i];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) {
    clk[2 * blockIdx.x] = t1 - t0;
    clk[2 * blockIdx.x + 1] = r1 - r0;
  }
}

template <int SHAPE, int NACC, int F, int L = 0>
static void run(int wg_per_cu, int iters)
{
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int grid = prop.multiProcessorCount * wg_per_cu;
  double* out;
  unsigned long long *clk, *h = (unsigned long long*)malloc(sizeof(unsigned long long) * 2 * grid);
  hipMalloc(&out, (size_t)grid * 256 * sizeof(double));
  hipMalloc(&clk, sizeof(unsigned long long) * 2 * grid);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<SHAPE, NACC, F, L>), dim3(grid), dim3(256), 0, 0, out, clk, iters, 1.0);
  hipDeviceSynchronize();
  hipMemcpy(h, clk, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost);
  double cyc = 0, real = 0;
  for (int i = 0; i < grid; ++i) cyc += (double)h[2 * i], real += (double)h[2 * i + 1];
  cyc /= grid, real /= grid;
  const double per_iter_simd = cyc / ((double)iters * NACC) / wg_per_cu;   // per matrix instruction (and its F FMAs) per SIMD
  const double ghz = cyc / real * 0.1;
  const double flops = SHAPE == 16 ? 2048.0 : 512.0;
  printf("shape %2dx%2dx4  acc %2d  fma/mfma %2d  lds reads/mfma %d  waves/SIMD %d: %6.1f cycles per matrix instruction per SIMD, clock %.2f GHz, %.1f TFLOP/s matrix\n",
         SHAPE, SHAPE, NACC, F, L, wg_per_cu, per_iter_simd, ghz,
         flops / per_iter_simd * ghz * 1e9 * prop.multiProcessorCount * 4 / 1e12);
  hipFree(out), hipFree(clk), free(h);
}

int main()
{
  const int it = 4000;
  run<16, 10, 0>(1, it);
  run<16, 10, 0>(2, it);
  run<4, 10, 0>(1, it);
  run<4, 10, 0>(2, it);
  run<4, 16, 0>(2, it);
  run<4, 32, 0>(2, it);
  run<4, 2, 0>(2, it);
  run<16, 10, 4>(2, it);
  run<16, 10, 8>(2, it);
  run<16, 10, 16>(2, it);
  run<4, 16, 2>(2, it);
  run<4, 16, 4>(2, it);
  run<16, 0 + 1, 0>(1, it);
  run<4, 64, 0>(2, it);
  run<4, 16, 0>(4, it);
  run<16, 10, 0>(4, it);
  run<4, 16, 0, 1>(2, it);
  run<4, 16, 0, 1>(4, it);
  run<4, 32, 0, 1>(2, it);
  return 0;
}
