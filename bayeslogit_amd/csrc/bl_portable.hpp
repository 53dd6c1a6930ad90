// bl_portable.hpp -- lets the arithmetic headers that hold no HIP-specific code compile
// both as device code (hipcc, gfx950) and as plain host C++ for CPU-side unit tests of
// the kernels' building blocks (tests/host_harness).  The host build is test scaffolding
// only; the shipped library contains the device build alone.
#pragma once
#include <math.h>
#include <stdint.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BL_HD __host__ __device__ __forceinline__
#define BL_HD_COLD __host__ __device__ inline __attribute__((noinline))
#else
#define BL_HD inline
#define BL_HD_COLD inline
#endif
// constant tables: device memory under hipcc (the kernels index them per lane), plain statics in the host harness
#if defined(__HIPCC__)
#define BL_DEVCONST __device__ const
#else
#define BL_DEVCONST static const
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define BL_COSPI(x) cospi(x)
#else
#define BL_COSPI(x) cos(3.141592653589793238462643383279502884197 * (x))
#endif

// a*b + c as ONE v_fma_f64 with all three operands in VGPRs.  For Horner steps whose addend is a
// constant the compiler otherwise emits v_mov_b64 (copy the constant) + v_fmac_f64: two VALU
// instructions per step in loops that are VALU-issue-bound.  Same rounding as fma().
#if defined(__HIP_DEVICE_COMPILE__)
namespace bl {
__device__ __forceinline__ double fma_vvv(double a, double b, double c)
{
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// The same with the addend in a scalar register pair: for a compile-time constant c the two halves are then set by
// scalar moves, which issue beside other waves' vector instructions, instead of two v_mov_b32 per step.
__device__ __forceinline__ double fma_vvs(double a, double b, double c)
{
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
  return r;
}
}  // namespace bl
#else
namespace bl {
inline double fma_vvv(double a, double b, double c) { return fma(a, b, c); }
inline double fma_vvs(double a, double b, double c) { return fma(a, b, c); }
}  // namespace bl
#endif
