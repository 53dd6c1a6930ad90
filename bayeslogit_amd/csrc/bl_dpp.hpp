// bl_dpp.hpp -- cross-lane moves of fp64 values by DPP (data-parallel primitives: a source modifier of a vector move, no LDS
// round trip) for the reductions of the sweep kernels.  gfx950 only (device code).
#pragma once
#include <hip/hip_runtime.h>

namespace bl {

template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v)      // every lane has a valid source: no `old` copy
{
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// Sum over the 16 lanes of a row, every lane receiving it: the butterfly  part += part[lane ^ 1], ^ 2, ^ 4, ^ 8  of psi = X beta
// (Logit.hpp:421,431) -- with the SAME bits.  After the first two steps the four lanes of a quad hold one value (a + b = b + a),
// so lane ^ 4 may be any lane of the other quad of the eight, and after the third any lane of the other half: quad
// permutations, row_half_mirror and row_mirror pick such lanes.  Eight v_mov_dpp instead of eight ds_bpermute (an LDS round trip
// each, four of them in a dependent chain: 42 of the 84 LDS operations of a 16-row tile of the single-pass sweep).
__device__ __forceinline__ double row16_allsum(double part)
{
  part += dpp_mov_f64<0xB1>(part);     // quad_perm(1, 0, 3, 2): lane ^ 1
  part += dpp_mov_f64<0x4E>(part);     // quad_perm(2, 3, 0, 1): lane ^ 2
  part += dpp_mov_f64<0x141>(part);    // row_half_mirror: the other quad of the eight
  part += dpp_mov_f64<0x140>(part);    // row_mirror: the other half of the row
  return part;
}

}  // namespace bl
