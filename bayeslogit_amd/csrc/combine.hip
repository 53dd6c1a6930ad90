// combine.hip -- duplicate-covariate-row merge on the device.
// Same result as Logit::compress (Code/C/Logit.hpp:192-270) and the merge inside
// MultLogit::set_data (Code/C/MultLogit.hpp:137-208): every row is folded into the
// FIRST row (in index order) with identical covariates, folds happen in index order
//   y_i <- (n_i/s) y_i + (n_j/s) y_j ; n_i <- s = n_i + n_j
// and survivors keep first-occurrence order.  The reference does this with an
// O(N^2 P) list walk; here rows are hashed, radix-sorted by hash (stable, so index
// order survives inside a hash run), matched exactly inside each run, folded by one
// thread per surviving row, and compacted with a prefix sum.  gfx950 only.
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "bl_gibbs_kernels.hpp"
#include "bl_host.hpp"

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ uint64_t mix64(uint64_t h)
{
  h ^= h >> 33;
  h *= 0xff51afd7ed558ccdull;
  h ^= h >> 33;
  h *= 0xc4ceb9fe1a85ec53ull;
  h ^= h >> 33;
  return h;
}

__global__ void k_hash_rows(const double* __restrict__ tX, int64_t N, int P, uint64_t* __restrict__ keys,
                            uint32_t* __restrict__ idx)
{
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < N; i += (int64_t)gridDim.x * kBlock) {
    uint64_t h = 0x9E3779B97F4A7C15ull;
    for (int j = 0; j < P; ++j) {
      double v = tX[(size_t)i * P + j];
      if (v == 0.0) v = 0.0;                       // -0.0 == +0.0 must hash alike
      h = mix64(h ^ (uint64_t)__double_as_longlong(v)) + (uint64_t)j;
    }
    keys[i] = h;
    idx[i] = (uint32_t)i;
  }
}

__global__ void k_run_start(const uint64_t* __restrict__ keys, int64_t N, uint32_t* __restrict__ rs)
{
  for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < N; p += (int64_t)gridDim.x * kBlock)
    rs[p] = (p == 0 || keys[p] != keys[p - 1]) ? (uint32_t)p : 0u;
}

__device__ __forceinline__ bool rows_equal(const double* __restrict__ tX, int P, uint32_t a, uint32_t b)
{
  const double* xa = tX + (size_t)a * P;
  const double* xb = tX + (size_t)b * P;
  for (int j = 0; j < P; ++j)
    if (!(xa[j] == xb[j])) return false;
  return true;
}

// rep[i] = index of the first row (in index order) equal to row i
__global__ void k_find_rep(const double* __restrict__ tX, int P, int64_t N, const uint32_t* __restrict__ idx,
                           const uint32_t* __restrict__ rs, uint32_t* __restrict__ rep)
{
  for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < N; p += (int64_t)gridDim.x * kBlock) {
    const uint32_t me = idx[p];
    uint32_t r = me;
    for (int64_t q = rs[p]; q < p; ++q)
      if (rows_equal(tX, P, idx[q], me)) {
        r = idx[q];
        break;
      }
    rep[me] = r;
  }
}

// one thread per surviving row folds its duplicates in index order
__global__ void k_fold(double* __restrict__ ty, double* __restrict__ nvec, int U, int64_t N,
                       const uint64_t* __restrict__ keys, const uint32_t* __restrict__ idx,
                       const uint32_t* __restrict__ rep)
{
  for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < N; p += (int64_t)gridDim.x * kBlock) {
    const uint32_t me = idx[p];
    if (rep[me] != me) continue;
    const uint64_t key = keys[p];
    double ni = nvec[me];
    for (int64_t q = p + 1; q < N && keys[q] == key; ++q) {
      const uint32_t j = idx[q];
      if (rep[j] != me) continue;
      const double nj = nvec[j];
      const double sum = ni + nj;
      for (int k = 0; k < U; ++k)
        ty[(size_t)me * U + k] = (ni / sum) * ty[(size_t)me * U + k] + (nj / sum) * ty[(size_t)j * U + k];
      ni = sum;
    }
    nvec[me] = ni;
  }
}

__global__ void k_keep(const uint32_t* __restrict__ rep, int64_t N, uint32_t* __restrict__ keep)
{
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < N; i += (int64_t)gridDim.x * kBlock)
    keep[i] = rep[i] == (uint32_t)i ? 1u : 0u;
}

__global__ void k_compact(const double* __restrict__ ty, const double* __restrict__ tX, const double* __restrict__ nvec,
                          int U, int P, int64_t N, const uint32_t* __restrict__ keep,
                          const uint32_t* __restrict__ pos, double* __restrict__ ty2, double* __restrict__ tX2,
                          double* __restrict__ n2)
{
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < N; i += (int64_t)gridDim.x * kBlock) {
    if (!keep[i]) continue;
    const uint32_t d = pos[i];
    n2[d] = nvec[i];
    for (int k = 0; k < U; ++k) ty2[(size_t)d * U + k] = ty[(size_t)i * U + k];
    for (int j = 0; j < P; ++j) tX2[(size_t)d * P + j] = tX[(size_t)i * P + j];
  }
}

struct MaxOp {
  __device__ __host__ uint32_t operator()(uint32_t a, uint32_t b) const { return a > b ? a : b; }
};

}  // namespace

namespace blk {

int combine_rows(double* ty, double* tX, double* n, int64_t N, int P, int U, int64_t* N_out, hipStream_t s)
{
  if (N <= 0) {
    *N_out = N;
    return BL_OK;
  }
  if (N >= (int64_t)1 << 32) {
    blh::set_error("combine_rows: N >= 2^32");
    return BL_ERR_ARG;
  }
  blh::DevBuf<uint64_t> k0, k1;
  blh::DevBuf<uint32_t> i0, i1, rs, rep, keep, pos;
  blh::DevBuf<double> ty2, tX2, n2;
  blh::DevBuf<char> tmp;
  hipError_t e = k0.alloc(N);
  if (e == hipSuccess) e = k1.alloc(N);
  if (e == hipSuccess) e = i0.alloc(N);
  if (e == hipSuccess) e = i1.alloc(N);
  if (e == hipSuccess) e = rs.alloc(N);
  if (e == hipSuccess) e = rep.alloc(N);
  if (e == hipSuccess) e = keep.alloc(N);
  if (e == hipSuccess) e = pos.alloc(N);
  if (e == hipSuccess) e = ty2.alloc((size_t)N * U);
  if (e == hipSuccess) e = tX2.alloc((size_t)N * P);
  if (e == hipSuccess) e = n2.alloc(N);
  size_t t_sort = 0, t_scan1 = 0, t_scan2 = 0;
  if (e == hipSuccess)
    e = rocprim::radix_sort_pairs(nullptr, t_sort, k0.p, k1.p, i0.p, i1.p, (size_t)N, 0, 64, s);
  if (e == hipSuccess) e = rocprim::inclusive_scan(nullptr, t_scan1, rs.p, rs.p, (size_t)N, MaxOp(), s);
  if (e == hipSuccess)
    e = rocprim::exclusive_scan(nullptr, t_scan2, keep.p, pos.p, 0u, (size_t)N, rocprim::plus<uint32_t>(), s);
  size_t tbytes = t_sort > t_scan1 ? t_sort : t_scan1;
  if (t_scan2 > tbytes) tbytes = t_scan2;
  if (e == hipSuccess) e = tmp.alloc(tbytes);
  if (e != hipSuccess) {
    blh::set_error(std::string("combine_rows: ") + hipGetErrorString(e));
    return BL_ERR_HIP;
  }
  const int g = blh::grid_for(N, kBlock, 256 * 8);
  hipLaunchKernelGGL(k_hash_rows, dim3(g), dim3(kBlock), 0, s, tX, N, P, k0.p, i0.p);
  e = rocprim::radix_sort_pairs(tmp.p, t_sort, k0.p, k1.p, i0.p, i1.p, (size_t)N, 0, 64, s);
  hipLaunchKernelGGL(k_run_start, dim3(g), dim3(kBlock), 0, s, k1.p, N, rs.p);
  if (e == hipSuccess) e = rocprim::inclusive_scan(tmp.p, t_scan1, rs.p, rs.p, (size_t)N, MaxOp(), s);
  hipLaunchKernelGGL(k_find_rep, dim3(g), dim3(kBlock), 0, s, tX, P, N, i1.p, rs.p, rep.p);
  hipLaunchKernelGGL(k_fold, dim3(g), dim3(kBlock), 0, s, ty, n, U, N, k1.p, i1.p, rep.p);
  hipLaunchKernelGGL(k_keep, dim3(g), dim3(kBlock), 0, s, rep.p, N, keep.p);
  if (e == hipSuccess)
    e = rocprim::exclusive_scan(tmp.p, t_scan2, keep.p, pos.p, 0u, (size_t)N, rocprim::plus<uint32_t>(), s);
  hipLaunchKernelGGL(k_compact, dim3(g), dim3(kBlock), 0, s, ty, tX, n, U, P, N, keep.p, pos.p, ty2.p, tX2.p, n2.p);
  uint32_t lastpos = 0, lastkeep = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&lastpos, pos.p + (N - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipMemcpyAsync(&lastkeep, keep.p + (N - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  const int64_t M = (int64_t)lastpos + lastkeep;
  if (e == hipSuccess) e = hipMemcpyAsync(ty, ty2.p, sizeof(double) * (size_t)M * U, hipMemcpyDeviceToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(tX, tX2.p, sizeof(double) * (size_t)M * P, hipMemcpyDeviceToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(n, n2.p, sizeof(double) * (size_t)M, hipMemcpyDeviceToDevice, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e == hipSuccess) e = hipGetLastError();
  if (e != hipSuccess) {
    blh::set_error(std::string("combine_rows: ") + hipGetErrorString(e));
    return BL_ERR_HIP;
  }
  *N_out = M;
  return BL_OK;
}

}  // namespace blk
