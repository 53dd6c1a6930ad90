// bl_philox.hpp -- Philox4x32-10 block function and the 52-bit uniform of the stream
// contract (DESIGN.md "RNG stream contract").  Portable (host + device).
#pragma once
#include "bl_portable.hpp"
#include <string.h>

namespace bl {

struct U4 { uint32_t x, y, z, w; };

BL_HD U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#if defined(__HIP_DEVICE_COMPILE__)
  // Keep only (k0, k1) live across the caller's loop: without this the ten bumped key pairs are
  // hoisted as loop invariants (20 SGPRs), the draw kernels run out of SGPRs and the spilled ones
  // come back through v_readlane -- VALU issue slots in a VALU-bound loop.  The bumps below are
  // SALU adds, which issue beside other waves' VALU work.
  asm volatile("" : "+s"(k0), "+s"(k1));
#endif
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}

// The same block function a few rounds at a time, for a caller that puts other work between the rounds:
// philox_rounds(s, 10) on s = {ctr, key} leaves philox4x32_10(ctr, key) in s.c.
struct PhiloxState { U4 c; uint32_t k0, k1; };
template <int R>
BL_HD void philox_rounds(PhiloxState& s)
{
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * s.c.x;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * s.c.z;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ s.c.y ^ s.k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ s.c.w ^ s.k1;
    s.c.y = (uint32_t)p1;
    s.c.w = (uint32_t)p0;
    s.c.x = n0;
    s.c.z = n2;
    s.k0 += 0x9E3779B9u;
    s.k1 += 0xBB67AE85u;
  }
}

// (m + 1/2) 2^-52 for the 52-bit integer m = (hi:lo) >> 12, built without an int->fp conversion:
// 1.m (exponent 0 | mantissa m) minus 1 is m 2^-52 exactly, and adding 2^-53 is exact too
// (2m+1 < 2^53), so this is bit-for-bit ((double)m + 0.5) * 2^-52.
BL_HD double u52(uint32_t hi, uint32_t lo)
{
  const uint64_t bits = 0x3FF0000000000000ull | ((((uint64_t)hi << 32) | lo) >> 12);
  double d;
  memcpy(&d, &bits, 8);
  return (d - 1.0) + 0x1.0p-53;
}

// Key of the chain started by the `call`-th gibbs()/mult_gibbs() of the .C boundary after set_seed(seed): a
// Philox block keyed by the seed in a domain of its own (4 = DOM_KEY), so that chains started by successive
// calls, and the rpg_* calls around them, never read the same (key, counter) pair.
BL_HD uint64_t chain_key(uint64_t seed, uint32_t call)
{
  const U4 o = philox4x32_10(call, 4u << 24, 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
  return ((uint64_t)o.y << 32) | o.x;
}

// counter words 0..2 of stream (idx, domain, epoch); word 3 is the block number
BL_HD uint32_t ctr1_of(uint64_t idx, uint32_t domain) { return ((uint32_t)(idx >> 32) & 0x00FFFFFFu) | (domain << 24); }

}  // namespace bl
