// barrier_probe.hip -- what a workgroup barrier costs a 4-wave workgroup on gfx950 (one wave per SIMD), in shader cycles:
// bare, and as the hand-over of one LDS value from one wave to the others (write, barrier, read).
//   hipcc --offload-arch=gfx950 -O3 -o barrier_probe barrier_probe.hip && ./barrier_probe
#include <hip/hip_runtime.h>

#include <cstdio>

__device__ __forceinline__ long long now()
{
  long long t;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

template <int NT>
__global__ __launch_bounds__(NT) void probe(long long* cyc, double* out)
{
  __shared__ double buf[128];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  buf[t & 127] = t;
  __syncthreads();
  const int R = 512;
  long long t0 = now();
  for (int r = 0; r < R; ++r) asm volatile("s_barrier" ::: "memory");
  long long t1 = now();
  if (t == 0) cyc[0] = t1 - t0;
  double acc = 0.0;
  t0 = now();
  for (int r = 0; r < R; ++r) {
    if (wave == (r & (NT / 64 - 1))) buf[64 * (r & 1) + lane] = acc + r;
    asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory");
    acc += buf[64 * (r & 1) + lane];
  }
  t1 = now();
  if (t == 0) cyc[1] = t1 - t0;
  // one wave works 300 cycles before each barrier, the others wait
  t0 = now();
  for (int r = 0; r < R; ++r) {
    if (wave == 0) asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
                                "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n" ::);
    asm volatile("s_barrier" ::: "memory");
  }
  t1 = now();
  if (t == 0) cyc[2] = t1 - t0;
  t0 = now();
  for (int r = 0; r < R; ++r) {
    asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
                 "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n" ::);
  }
  t1 = now();
  if (t == 0) cyc[3] = t1 - t0;
  out[t] = acc;
}

int main()
{
  long long* cyc;
  double* out;
  hipMalloc(&cyc, 64);
  hipMalloc(&out, 8 * 1024);
  long long h[8];
  for (int nt : {64, 128, 256, 512}) {
    for (int rep = 0; rep < 2; ++rep) {
      if (nt == 64) probe<64><<<1, 64>>>(cyc, out);
      if (nt == 128) probe<128><<<1, 128>>>(cyc, out);
      if (nt == 256) probe<256><<<1, 256>>>(cyc, out);
      if (nt == 512) probe<512><<<1, 512>>>(cyc, out);
    }
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    std::printf("%d waves: bare s_barrier %.0f cycles; LDS write + barrier + read %.0f; one wave 19 x s_nop 15 then barrier %.0f (the nops alone %.0f)\n",
                nt / 64, h[0] / 512.0, h[1] / 512.0, h[2] / 512.0, h[3] / 512.0);
  }
  return 0;
}
