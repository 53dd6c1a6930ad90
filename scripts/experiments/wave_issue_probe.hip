// wave_issue_probe.hip -- what ONE wavefront (alone on its SIMD) pays per instruction on gfx950, in shader cycles: the numbers the
// single-wave dense routines of the beta stage (kernels_beta.hip) are designed around.
//   hipcc --offload-arch=gfx950 -O3 -o wave_issue_probe wave_issue_probe.hip && ./wave_issue_probe
#include <hip/hip_runtime.h>

#include <cstdio>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

__device__ __forceinline__ long long now()
{
  long long t;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

__global__ __launch_bounds__(64) void probe(double* out, long long* cyc, const double* in)
{
  __shared__ __attribute__((aligned(16))) double buf[128];
  const int lane = threadIdx.x;
  double a0 = in[lane], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  double u = in[64 + lane] * 1e-3;
  buf[lane] = u;
  buf[64 + lane] = u;
  __syncthreads();
  long long t0, t1;
  const int R = 64;

  // A: independent fp64 FMAs (8 accumulators)
  t0 = now();
  for (int r = 0; r < R; ++r)
    asm volatile(REP4("v_fma_f64 %0, %8, %0, %0\n v_fma_f64 %1, %8, %1, %1\n v_fma_f64 %2, %8, %2, %2\n v_fma_f64 %3, %8, %3, %3\n"
                      "v_fma_f64 %4, %8, %4, %4\n v_fma_f64 %5, %8, %5, %5\n v_fma_f64 %6, %8, %6, %6\n v_fma_f64 %7, %8, %7, %7\n")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : "v"(u));
  t1 = now();
  if (lane == 0) cyc[0] = (t1 - t0);

  // B: a dependent chain of fp64 FMAs
  t0 = now();
  for (int r = 0; r < R; ++r) asm volatile(REP16("v_fma_f64 %0, %1, %0, %0\n") REP16("v_fma_f64 %0, %1, %0, %0\n") : "+v"(a0) : "v"(u));
  t1 = now();
  if (lane == 0) cyc[1] = (t1 - t0);

  // B2: two interleaved dependent chains
  t0 = now();
  for (int r = 0; r < R; ++r) asm volatile(REP16("v_fma_f64 %0, %2, %0, %0\n v_fma_f64 %1, %2, %1, %1\n") : "+v"(a0), "+v"(a1) : "v"(u));
  t1 = now();
  if (lane == 0) cyc[2] = (t1 - t0);

  // C: readlane pair -> scalar operand of one FMA, the same scalar pair every time (the compiler's pattern)
  t0 = now();
  for (int r = 0; r < R; ++r)
    asm volatile(REP4("v_readlane_b32 s20, %8, 3\n v_readlane_b32 s21, %9, 3\n s_nop 1\n v_fma_f64 %0, -%10, s[20:21], %0\n"
                      "v_readlane_b32 s20, %8, 4\n v_readlane_b32 s21, %9, 4\n s_nop 1\n v_fma_f64 %1, -%10, s[20:21], %1\n"
                      "v_readlane_b32 s20, %8, 5\n v_readlane_b32 s21, %9, 5\n s_nop 1\n v_fma_f64 %2, -%10, s[20:21], %2\n"
                      "v_readlane_b32 s20, %8, 6\n v_readlane_b32 s21, %9, 6\n s_nop 1\n v_fma_f64 %3, -%10, s[20:21], %3\n"
                      "v_readlane_b32 s20, %8, 7\n v_readlane_b32 s21, %9, 7\n s_nop 1\n v_fma_f64 %4, -%10, s[20:21], %4\n"
                      "v_readlane_b32 s20, %8, 8\n v_readlane_b32 s21, %9, 8\n s_nop 1\n v_fma_f64 %5, -%10, s[20:21], %5\n"
                      "v_readlane_b32 s20, %8, 9\n v_readlane_b32 s21, %9, 9\n s_nop 1\n v_fma_f64 %6, -%10, s[20:21], %6\n"
                      "v_readlane_b32 s20, %8, 10\n v_readlane_b32 s21, %9, 10\n s_nop 1\n v_fma_f64 %7, -%10, s[20:21], %7\n")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : "v"(__double2loint(u)), "v"(__double2hiint(u)), "v"(u)
                 : "s20", "s21");
  t1 = now();
  if (lane == 0) cyc[3] = (t1 - t0);

  // D: the eight readlane pairs first (distinct scalar pairs), then the eight FMAs
  t0 = now();
  for (int r = 0; r < R; ++r)
    asm volatile(REP4("v_readlane_b32 s20, %8, 3\n v_readlane_b32 s21, %9, 3\n v_readlane_b32 s22, %8, 4\n v_readlane_b32 s23, %9, 4\n"
                      "v_readlane_b32 s24, %8, 5\n v_readlane_b32 s25, %9, 5\n v_readlane_b32 s26, %8, 6\n v_readlane_b32 s27, %9, 6\n"
                      "v_readlane_b32 s28, %8, 7\n v_readlane_b32 s29, %9, 7\n v_readlane_b32 s30, %8, 8\n v_readlane_b32 s31, %9, 8\n"
                      "v_readlane_b32 s32, %8, 9\n v_readlane_b32 s33, %9, 9\n v_readlane_b32 s34, %8, 10\n v_readlane_b32 s35, %9, 10\n"
                      "s_nop 1\n"
                      "v_fma_f64 %0, -%10, s[20:21], %0\n v_fma_f64 %1, -%10, s[22:23], %1\n v_fma_f64 %2, -%10, s[24:25], %2\n"
                      "v_fma_f64 %3, -%10, s[26:27], %3\n v_fma_f64 %4, -%10, s[28:29], %4\n v_fma_f64 %5, -%10, s[30:31], %5\n"
                      "v_fma_f64 %6, -%10, s[32:33], %6\n v_fma_f64 %7, -%10, s[34:35], %7\n")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : "v"(__double2loint(u)), "v"(__double2hiint(u)), "v"(u)
                 : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35");
  t1 = now();
  if (lane == 0) cyc[4] = (t1 - t0);

  // E: broadcast LDS reads (every lane the same address), two operands a read, then the FMAs (the compiler's schedule)
  t0 = now();
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      asm volatile("" ::: "memory");
      const double2 v0 = *reinterpret_cast<const double2*>(buf + 8 * q), v1 = *reinterpret_cast<const double2*>(buf + 8 * q + 2),
                    v2 = *reinterpret_cast<const double2*>(buf + 8 * q + 4), v3 = *reinterpret_cast<const double2*>(buf + 8 * q + 6);
      a0 = fma(-u, v0.x, a0), a1 = fma(-u, v0.y, a1), a2 = fma(-u, v1.x, a2), a3 = fma(-u, v1.y, a3);
      a4 = fma(-u, v2.x, a4), a5 = fma(-u, v2.y, a5), a6 = fma(-u, v3.x, a6), a7 = fma(-u, v3.y, a7);
    }
  }
  t1 = now();
  if (lane == 0) cyc[5] = (t1 - t0);

  // F: sqrt then divide, dependent (a pivot of the Cholesky)
  t0 = now();
  for (int r = 0; r < R; ++r) {
    asm volatile("" : "+v"(a0));
    const double d = sqrt(a0 * a0 + 1.0);
    a0 = u / d + 1.0;
  }
  t1 = now();
  if (lane == 0) cyc[6] = (t1 - t0);

  // G: v_mov_b64
  t0 = now();
  for (int r = 0; r < R; ++r)
    asm volatile(REP4("v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %4\n v_mov_b64 %4, %5\n v_mov_b64 %5, %6\n"
                      "v_mov_b64 %6, %7\n v_mov_b64 %7, %0\n")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  t1 = now();
  if (lane == 0) cyc[7] = (t1 - t0);

  // H: LDS round trip (a write, then a read that depends on it), and I / J: broadcast reads twelve deep, 8 doubles a chunk
  {
    const unsigned addr = (unsigned)(size_t)buf;
    t0 = now();
    for (int r = 0; r < R; ++r) {
      asm volatile("ds_write_b64 %1, %0\n s_waitcnt lgkmcnt(0)\n ds_read_b64 %0, %1 offset:8\n s_waitcnt lgkmcnt(0)" : "+v"(a0) : "v"(addr + 8 * lane) : "memory");
    }
    t1 = now();
    if (lane == 0) cyc[8] = (t1 - t0);
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2 p0, p1, p2, p3, p4, p5, p6, p7, p8, p9, p10, p11;
#define RD(x, o) "ds_read_b128 %" #x ", %20 offset:" #o "\n"
#define RD2(x, o0, o1) "ds_read2_b64 %" #x ", %20 offset0:" #o0 " offset1:" #o1 "\n"
#define FM(acc, x) "v_fma_f64 %" #acc ", -%21, %L" #x ", %" #acc "\n"
    // (the two halves of a 128-bit operand cannot be named in inline asm: the FMAs are left to the compiler below)
    t0 = now();
    for (int r = 0; r < R; ++r) {
      asm volatile("ds_read_b128 %0, %12\n ds_read_b128 %1, %12 offset:16\n ds_read_b128 %2, %12 offset:32\n ds_read_b128 %3, %12 offset:48\n"
                   "ds_read_b128 %4, %12 offset:64\n ds_read_b128 %5, %12 offset:80\n ds_read_b128 %6, %12 offset:96\n ds_read_b128 %7, %12 offset:112\n"
                   "ds_read_b128 %8, %12 offset:128\n ds_read_b128 %9, %12 offset:144\n ds_read_b128 %10, %12 offset:160\n ds_read_b128 %11, %12 offset:176\n"
                   : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "=&v"(p4), "=&v"(p5), "=&v"(p6), "=&v"(p7), "=&v"(p8), "=&v"(p9), "=&v"(p10), "=&v"(p11)
                   : "v"(addr) : "memory");
      asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
      a0 = fma(-u, p0.x, a0), a1 = fma(-u, p0.y, a1), a2 = fma(-u, p1.x, a2), a3 = fma(-u, p1.y, a3);
      a4 = fma(-u, p2.x, a4), a5 = fma(-u, p2.y, a5), a6 = fma(-u, p3.x, a6), a7 = fma(-u, p3.y, a7);
      asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7));
      a0 = fma(-u, p4.x, a0), a1 = fma(-u, p4.y, a1), a2 = fma(-u, p5.x, a2), a3 = fma(-u, p5.y, a3);
      a4 = fma(-u, p6.x, a4), a5 = fma(-u, p6.y, a5), a6 = fma(-u, p7.x, a6), a7 = fma(-u, p7.y, a7);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(p8), "+v"(p9), "+v"(p10), "+v"(p11));
      a0 = fma(-u, p8.x, a0), a1 = fma(-u, p8.y, a1), a2 = fma(-u, p9.x, a2), a3 = fma(-u, p9.y, a3);
      a4 = fma(-u, p10.x, a4), a5 = fma(-u, p10.y, a5), a6 = fma(-u, p11.x, a6), a7 = fma(-u, p11.y, a7);
    }
    t1 = now();
    if (lane == 0) cyc[9] = (t1 - t0);
    // K: the twelve reads alone
    t0 = now();
    for (int r = 0; r < R; ++r) {
      asm volatile("ds_read_b128 %0, %12\n ds_read_b128 %1, %12 offset:16\n ds_read_b128 %2, %12 offset:32\n ds_read_b128 %3, %12 offset:48\n"
                   "ds_read_b128 %4, %12 offset:64\n ds_read_b128 %5, %12 offset:80\n ds_read_b128 %6, %12 offset:96\n ds_read_b128 %7, %12 offset:112\n"
                   "ds_read_b128 %8, %12 offset:128\n ds_read_b128 %9, %12 offset:144\n ds_read_b128 %10, %12 offset:160\n ds_read_b128 %11, %12 offset:176\n"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "=&v"(p4), "=&v"(p5), "=&v"(p6), "=&v"(p7), "=&v"(p8), "=&v"(p9), "=&v"(p10), "=&v"(p11)
                   : "v"(addr) : "memory");
    }
    t1 = now();
    if (lane == 0) cyc[10] = (t1 - t0);
    a0 += p0.x + p11.y;
    // L: the same twelve reads as ds_read_b64 pairs would be 24 reads: here 12 x ds_read_b64
    double s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11;
    t0 = now();
    for (int r = 0; r < R; ++r) {
      asm volatile("ds_read_b64 %0, %12\n ds_read_b64 %1, %12 offset:8\n ds_read_b64 %2, %12 offset:16\n ds_read_b64 %3, %12 offset:24\n"
                   "ds_read_b64 %4, %12 offset:32\n ds_read_b64 %5, %12 offset:40\n ds_read_b64 %6, %12 offset:48\n ds_read_b64 %7, %12 offset:56\n"
                   "ds_read_b64 %8, %12 offset:64\n ds_read_b64 %9, %12 offset:72\n ds_read_b64 %10, %12 offset:80\n ds_read_b64 %11, %12 offset:88\n"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3), "=&v"(s4), "=&v"(s5), "=&v"(s6), "=&v"(s7), "=&v"(s8), "=&v"(s9), "=&v"(s10), "=&v"(s11)
                   : "v"(addr) : "memory");
    }
    t1 = now();
    if (lane == 0) cyc[11] = (t1 - t0);
    a1 += s0 + s11;
  }
  // M: s_set_gpr_idx_on / two moves / off
  {
    int i2 = lane, i3 = lane + 1;
    t0 = now();
    for (int r = 0; r < R; ++r)
      asm volatile(REP16("s_set_gpr_idx_on %2, gpr_idx(SRC0)\n v_mov_b32 %0, %1\n v_mov_b32 %0, %1\n s_set_gpr_idx_off\n") : "+v"(i2) : "v"(i3), "s"(0));
    t1 = now();
    if (lane == 0) cyc[12] = (t1 - t0);
    a2 += i2;
  }
  out[lane] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main()
{
  double *out, *in;
  long long* cyc;
  hipMalloc(&out, 64 * 8);
  hipMalloc(&in, 128 * 8);
  hipMalloc(&cyc, 16 * 8);
  double hin[128];
  for (int i = 0; i < 128; ++i) hin[i] = 0.5 + 0.001 * i;
  hipMemcpy(in, hin, sizeof(hin), hipMemcpyHostToDevice);
  hipMemset(cyc, 0, 16 * 8);
  for (int rep = 0; rep < 2; ++rep) probe<<<1, 64>>>(out, cyc, in);
  long long h[16];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const double n = 64 * 32.0;
  std::printf("shader cycles per fp64 FMA, one wavefront alone on its SIMD (gfx950):\n");
  std::printf("  A  independent FMAs                               %.1f\n", h[0] / n);
  std::printf("  B  one dependent chain                            %.1f\n", h[1] / n);
  std::printf("  B2 two interleaved chains                         %.1f\n", h[2] / n);
  std::printf("  C  2 x v_readlane -> s[..] -> FMA, same pair      %.1f\n", h[3] / n);
  std::printf("  D  16 readlanes (distinct pairs), then 8 FMAs     %.1f\n", h[4] / n);
  std::printf("  E  4 x ds_read_b128 broadcast, then 8 FMAs        %.1f\n", h[5] / n);
  std::printf("one sqrt and one divide, dependent: %.0f cycles; v_mov_b64: %.1f\n", h[6] / 64.0, h[7] / n);
  std::printf("LDS write -> dependent read: %.0f cycles; 12 broadcast ds_read_b128 in flight + their 24 FMAs: %.1f per FMA; the 12 reads alone: %.0f (b128), %.0f (b64)\n",
              h[8] / 64.0, h[9] / (64 * 24.0), h[10] / 64.0, h[11] / 64.0);
  std::printf("s_set_gpr_idx_on + 2 v_mov + off: %.1f cycles\n", h[12] / (64 * 16.0));
  return 0;
}
