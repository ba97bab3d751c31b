// grid.hip -- voxel-grid subsampling in PACK mode (points (total,3) + lengths (B,)).
// Replaces `grid_subsample_pack_mode` of the reference (vision3d/ops/grid_subsample.py:49-99): floor(p / voxel) ->
// per-cloud minimum subtracted in a Python loop with .item() syncs -> raveled hash -> torch.unique -> scatter_add_
// (float atomics on CUDA) -> divide -> bincount.  Same result definition: one point per occupied voxel = mean of the
// voxel's points, output ordered by (cloud, vx, vy, vz) like torch.unique of the hash.
//
// Design (gfx950): keys (cloud | vx | vy | vz, 10 + 3 x 18 bits) are sorted with ONE stable device radix sort
// (rocPRIM -- a plain library primitive, like a GEMM from hipBLASLt) carrying the point index as the value, so equal
// voxels stay in input order; a head flag + exclusive scan gives every voxel its output slot; one thread per voxel
// then sums its run SEQUENTIALLY IN INPUT ORDER and divides.  No float atomics: the result is bit-reproducible and
// equals the single-threaded CPU loop, which the reference's CUDA scatter_add_ cannot promise.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include <algorithm>

#include "common.h"

namespace v3d {

constexpr int kGsThreads = 256;
constexpr int kGsMaxClouds = 1024;       // 10 key bits
constexpr int kGsCoordBits = 18;         // voxel coordinates relative to the cloud's minimum: < 262,144 per axis

__device__ __forceinline__ int gs_voxel(float p, float inv) { return (int)floorf(p * inv); }  // :64-65

// One workgroup per cloud: its first point and its minimum voxel (grid_subsample.py:75).
__global__ __launch_bounds__(kGsThreads) void gs_minvox_kernel(const float* __restrict__ pts,
                                                               const int64_t* __restrict__ lengths,
                                                               long long* __restrict__ offsets, int* __restrict__ minv,
                                                               int B, float inv) {
  __shared__ int red[3][kGsThreads / 64];
  const int b = blockIdx.x, tid = threadIdx.x;
  long long off = 0;
  for (int i = 0; i < b; ++i) off += lengths[i];
  const long long n = lengths[b];
  int lo[3] = {INT32_MAX, INT32_MAX, INT32_MAX};
  for (long long i = tid; i < n; i += kGsThreads) {
#pragma unroll
    for (int a = 0; a < 3; ++a) lo[a] = min(lo[a], gs_voxel(pts[(off + i) * 3 + a], inv));
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    for (int d = 32; d >= 1; d >>= 1) lo[a] = min(lo[a], __shfl_xor(lo[a], d, 64));
    if ((tid & 63) == 0) red[a][tid >> 6] = lo[a];
  }
  __syncthreads();
  if (tid < 3) {
    int v = red[tid][0];
    for (int w = 1; w < kGsThreads / 64; ++w) v = min(v, red[tid][w]);
    minv[b * 3 + tid] = v;
  }
  if (tid == 0) {
    offsets[b] = off;
    if (b == B - 1) offsets[B] = off + n;
  }
}

__global__ __launch_bounds__(kGsThreads) void gs_keys_kernel(const float* __restrict__ pts,
                                                             const long long* __restrict__ offsets,
                                                             const int* __restrict__ minv,
                                                             unsigned long long* __restrict__ keys,
                                                             int* __restrict__ vals, int* __restrict__ error, int B,
                                                             long long total, float inv) {
  const long long i = (long long)blockIdx.x * kGsThreads + threadIdx.x;
  if (i >= total) return;
  int lo = 0, hi = B - 1;  // last cloud whose offset is <= i
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (offsets[mid] <= i) lo = mid; else hi = mid - 1;
  }
  unsigned long long key = (unsigned long long)lo;
  bool bad = false;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const long long v = (long long)gs_voxel(pts[i * 3 + a], inv) - (long long)minv[lo * 3 + a];
    bad |= v < 0 || v >= (1ll << kGsCoordBits);
    key = (key << kGsCoordBits) | (unsigned long long)(v & ((1ll << kGsCoordBits) - 1));
  }
  if (bad) *error = 1;  // NaN coordinates or a cloud wider than 2^18 voxels
  keys[i] = key;
  vals[i] = (int)i;
}

__global__ __launch_bounds__(kGsThreads) void gs_heads_kernel(const unsigned long long* __restrict__ keys,
                                                              int* __restrict__ flags, long long total) {
  const long long i = (long long)blockIdx.x * kGsThreads + threadIdx.x;
  if (i >= total) return;
  flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}

// One thread per voxel: sum its points in input order (the sort is stable), divide by the count (:88-89).
__global__ __launch_bounds__(kGsThreads) void gs_reduce_kernel(const float* __restrict__ pts,
                                                               const int* __restrict__ vals,
                                                               const int* __restrict__ flags,
                                                               const int* __restrict__ slots, const int* __restrict__ error,
                                                               float* __restrict__ out, int64_t* __restrict__ num_out,
                                                               long long total) {
  const long long i = (long long)blockIdx.x * kGsThreads + threadIdx.x;
  if (i >= total) return;
  if (i == total - 1) *num_out = *error ? -1 : (int64_t)slots[i] + flags[i];
  if (!flags[i]) return;
  float sx = 0.f, sy = 0.f, sz = 0.f;
  long long j = i;
  do {
    const long long p = vals[j];
    sx += pts[p * 3 + 0];
    sy += pts[p * 3 + 1];
    sz += pts[p * 3 + 2];
    ++j;
  } while (j < total && !flags[j]);
  const float c = (float)(j - i);
  const long long m = slots[i];
  out[m * 3 + 0] = sx / c;
  out[m * 3 + 1] = sy / c;
  out[m * 3 + 2] = sz / c;
}

// Per-cloud counts without atomics: the first record of a cloud is a voxel head, so the exclusive head count there is
// the number of voxels of all earlier clouds (:92-94 computes the same with bincount).
__global__ void gs_lengths_kernel(const unsigned long long* __restrict__ keys, const int* __restrict__ flags,
                                  const int* __restrict__ slots, int64_t* __restrict__ out_len, int B,
                                  long long total) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  long long bound[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {  // first record whose cloud is >= b + e
    const unsigned long long want = (unsigned long long)(b + e);
    long long lo = 0, hi = total;
    while (lo < hi) {
      const long long mid = (lo + hi) >> 1;
      if ((keys[mid] >> (3 * kGsCoordBits)) < want) lo = mid + 1; else hi = mid;
    }
    bound[e] = lo < total ? (long long)slots[lo] : (long long)slots[total - 1] + flags[total - 1];
  }
  out_len[b] = bound[1] - bound[0];
}

}  // namespace v3d

extern "C" {

int v3d_grid_subsample_pack_f32(const float* points, const int64_t* lengths, float* sampled_points,
                                int64_t* sampled_lengths, int64_t* num_sampled, int B, int64_t total, float voxel_size,
                                int device, void* stream_) {
  using namespace v3d;
  const char* what = "grid_subsample_pack_mode";
  V3D_REQUIRE(B >= 0 && total >= 0, what, "negative dimension");
  V3D_REQUIRE(voxel_size > 0.f, what, "voxel_size must be positive");  // NaN fails too
  V3D_REQUIRE(num_sampled && (B == 0 || (lengths && sampled_lengths)) && (total == 0 || (points && sampled_points)),
              what, "null pointer");
  if (B > kGsMaxClouds || total >= (int64_t)INT32_MAX) {
    set_last_error("%s: at most %d clouds and 2^31-1 points per call", what, kGsMaxClouds);
    return V3D_ERR_SIZE_LIMIT;
  }
  DeviceGuard guard(device);
  if (guard.rc) return guard.rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (B > 0) V3D_HIP_TRY(hipMemsetAsync(sampled_lengths, 0, sizeof(int64_t) * (size_t)B, stream), what);
  V3D_HIP_TRY(hipMemsetAsync(num_sampled, 0, sizeof(int64_t), stream), what);
  if (total == 0 || B == 0) return 0;
  const float inv = (float)(1.0 / (double)voxel_size);  // :64: a Python float, rounded to float32 by the product

  const size_t n = (size_t)total;
  size_t sort_bytes = 0, scan_bytes = 0;
  unsigned long long* nullk = nullptr;
  int* nulli = nullptr;
  const unsigned end_bit = 3 * kGsCoordBits + 10;
  V3D_HIP_TRY(rocprim::radix_sort_pairs(nullptr, sort_bytes, nullk, nullk, nulli, nulli, n, 0, end_bit, stream), what);
  V3D_HIP_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, nulli, nulli, 0, n, rocprim::plus<int>(), stream), what);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t at = off;
    off += align_up(bytes, 256);
    return at;
  };
  const size_t o_off = take(sizeof(long long) * ((size_t)B + 1)), o_min = take(sizeof(int) * 3 * (size_t)B),
               o_err = take(sizeof(int)), o_k0 = take(8 * n), o_k1 = take(8 * n), o_v0 = take(4 * n), o_v1 = take(4 * n),
               o_flag = take(4 * n), o_slot = take(4 * n), o_tmp = take(std::max(sort_bytes, scan_bytes));
  char* base = nullptr;
  V3D_HIP_TRY(hipMallocAsync(reinterpret_cast<void**>(&base), off, stream), what);
  long long* offsets = reinterpret_cast<long long*>(base + o_off);
  int* minv = reinterpret_cast<int*>(base + o_min);
  int* error = reinterpret_cast<int*>(base + o_err);
  unsigned long long* k0 = reinterpret_cast<unsigned long long*>(base + o_k0);
  unsigned long long* k1 = reinterpret_cast<unsigned long long*>(base + o_k1);
  int* v0 = reinterpret_cast<int*>(base + o_v0);
  int* v1 = reinterpret_cast<int*>(base + o_v1);
  int* flags = reinterpret_cast<int*>(base + o_flag);
  int* slots = reinterpret_cast<int*>(base + o_slot);
  void* tmp = base + o_tmp;
  int rc = 0;
  do {
    hipError_t e = hipMemsetAsync(error, 0, sizeof(int), stream);
    if (e != hipSuccess) {
      set_last_error("%s: hipMemsetAsync: %s", what, hipGetErrorString(e));
      rc = (int)e;
      break;
    }
    const unsigned blocks = (unsigned)ceil_div(total, kGsThreads);
    hipLaunchKernelGGL(gs_minvox_kernel, dim3(B), dim3(kGsThreads), 0, stream, points, lengths, offsets, minv, B, inv);
    hipLaunchKernelGGL(gs_keys_kernel, dim3(blocks), dim3(kGsThreads), 0, stream, points, offsets, minv, k0, v0, error,
                       B, (long long)total, inv);
    e = rocprim::radix_sort_pairs(tmp, sort_bytes, k0, k1, v0, v1, n, 0, end_bit, stream);
    if (e != hipSuccess) {
      set_last_error("%s: radix sort: %s", what, hipGetErrorString(e));
      rc = (int)e;
      break;
    }
    hipLaunchKernelGGL(gs_heads_kernel, dim3(blocks), dim3(kGsThreads), 0, stream, k1, flags, (long long)total);
    e = rocprim::exclusive_scan(tmp, scan_bytes, flags, slots, 0, n, rocprim::plus<int>(), stream);
    if (e != hipSuccess) {
      set_last_error("%s: scan: %s", what, hipGetErrorString(e));
      rc = (int)e;
      break;
    }
    hipLaunchKernelGGL(gs_reduce_kernel, dim3(blocks), dim3(kGsThreads), 0, stream, points, v1, flags, slots, error,
                       sampled_points, num_sampled, (long long)total);
    hipLaunchKernelGGL(gs_lengths_kernel, dim3((unsigned)ceil_div(B, 64)), dim3(64), 0, stream, k1, flags, slots,
                       sampled_lengths, B, (long long)total);
    rc = launch_status(what);
  } while (0);
  hipError_t e = hipFreeAsync(base, stream);
  if (rc == 0 && e != hipSuccess) {
    set_last_error("%s: hipFreeAsync: %s", what, hipGetErrorString(e));
    rc = (int)e;
  }
  return rc;
}

}  // extern "C"
