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
#if defined(__HIP_DEVICE_COMPILE__)
#define BL_COSPI(x) cospi(x)
#else
#define BL_COSPI(x) cos(3.141592653589793238462643383279502884197 * (x))
#endif
