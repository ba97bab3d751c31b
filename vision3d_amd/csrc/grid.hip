// grid.hip -- voxel-grid subsampling in PACK mode (points (total,3) + lengths (B,)).
// Replaces `grid_subsample_pack_mode` of the reference (vision3d/ops/grid_subsample.py:49-99): floor(p / voxel) ->
// per-cloud minimum subtracted in a Python loop with .item() syncs -> raveled hash -> torch.unique -> scatter_add_
// (float atomics on CUDA) -> divide -> bincount.  Same result definition: one point per occupied voxel = mean of the
// voxel's points, output ordered by (cloud, vx, vy, vz) like torch.unique of the hash.
//
// `voxelize_pack_mode` (vision3d/ops/voxelize.py:6-63) is the same reduction with three more outputs -- the integer voxel
// coordinates, the voxel means of an optional (N,C) feature tensor, and floor(p / voxel) instead of floor(p * (1/voxel)) --
// and runs on the same kernels (`v3d_voxelize_pack_f32`).
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

// scale > 0: floor(p * scale), grid_subsample.py:64-65 (scale = 1 / voxel_size rounded to fp32);
// scale < 0: floor(p / -scale), voxelize.py:22 (correctly rounded fp32 division)
__device__ __forceinline__ int gs_voxel(float p, float scale) {
  return (int)floorf(scale > 0.f ? p * scale : p / -scale);
}

constexpr int kGsMinSplit = 16;  // workgroups per cloud in the minimum-voxel pass

// one launch for every small initialisation of a call: minimum voxels, the error flag, the per-cloud counts, the total
__global__ void gs_init_kernel(int* __restrict__ minv, int* __restrict__ error, int64_t* __restrict__ out_len,
                               int64_t* __restrict__ num_out, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * 3) minv[i] = INT32_MAX;
  if (i < B) out_len[i] = 0;
  if (i == 0) {
    *error = 0;
    *num_out = 0;
  }
}

// kGsMinSplit workgroups per cloud: its first point and its minimum voxel (grid_subsample.py:75), combined with
// integer atomics (exact, order-free).
__global__ __launch_bounds__(kGsThreads) void gs_minvox_kernel(const float* __restrict__ pts,
                                                               const int64_t* __restrict__ lengths,
                                                               long long* __restrict__ offsets, int* __restrict__ minv,
                                                               int B, float inv) {
  __shared__ int red[3][kGsThreads / 64];
  const int b = blockIdx.x, tid = threadIdx.x;
  long long off = 0;
  for (int i = 0; i < b; ++i) off += lengths[i];
  const long long n = lengths[b];
  if (tid == 0 && blockIdx.y == 0) {
    offsets[b] = off;
    if (b == B - 1) offsets[B] = off + n;
  }
  if ((long long)blockIdx.y * kGsThreads >= n) return;
  int lo[3] = {INT32_MAX, INT32_MAX, INT32_MAX};
  for (long long i = (long long)blockIdx.y * kGsThreads + tid; i < n; i += (long long)kGsThreads * kGsMinSplit) {
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
    atomicMin(&minv[b * 3 + tid], v);
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
                                                               long long total,
                                                               const unsigned long long* __restrict__ keys,
                                                               const int* __restrict__ minv,
                                                               int64_t* __restrict__ out_voxels,
                                                               int* __restrict__ head_pos) {
  const long long i = (long long)blockIdx.x * kGsThreads + threadIdx.x;
  if (i >= total) return;
  if (i == total - 1) *num_out = *error ? -1 : (int64_t)slots[i] + flags[i];
  if (!flags[i]) return;
  if (head_pos) head_pos[slots[i]] = (int)i;
  if (out_voxels) {  // voxelize.py:43-44: the voxel's integer coordinates
    const unsigned long long key = keys[i];
    const int b = (int)(key >> (3 * kGsCoordBits));
#pragma unroll
    for (int a = 0; a < 3; ++a)
      out_voxels[(long long)slots[i] * 3 + a] =
          (int64_t)((key >> ((2 - a) * kGsCoordBits)) & ((1ull << kGsCoordBits) - 1)) + (int64_t)minv[b * 3 + a];
  }
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

// Voxel means of a feature tensor (voxelize.py:57-60): thread = (voxel, channel), the voxel's points summed in input
// order like the coordinates; consecutive threads read consecutive channels of a row.
__global__ __launch_bounds__(kGsThreads) void gs_feats_kernel(const float* __restrict__ feats, const int* __restrict__ vals,
                                                              const int* __restrict__ flags,
                                                              const int* __restrict__ head_pos,
                                                              const int64_t* __restrict__ num_out, int C, long long total,
                                                              float* __restrict__ out_feats) {
  const long long t = (long long)blockIdx.x * kGsThreads + threadIdx.x;
  const long long m = t / C;
  const int c = (int)(t % C);
  if (m >= *num_out) return;  // also covers the error value -1
  long long j = head_pos[m];
  const long long first = j;
  float sum = 0.f;
  do {
    sum += feats[(long long)vals[j] * C + c];
    ++j;
  } while (j < total && !flags[j]);
  out_feats[m * C + c] = sum / (float)(j - first);
}

// Per-cloud counts without atomics: the first record of a cloud is a voxel head, so the exclusive head count there is
// the number of voxels of all earlier clouds (:92-94 computes the same with bincount).  Where a cloud's records start in
// the sorted order needs no search: the keys carry the cloud number in their top bits and gs_keys_kernel numbered record
// i with the last cloud whose offset is <= i, so the records of clouds < c are the first min(offsets[c], total).
__global__ void gs_lengths_kernel(const long long* __restrict__ offsets, const int* __restrict__ flags,
                                  const int* __restrict__ slots, int64_t* __restrict__ out_len, int B,
                                  long long total) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  long long bound[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {  // first record whose cloud is >= b + e
    long long lo = b + e >= B ? total : offsets[b + e];
    lo = lo < 0 ? 0 : (lo > total ? total : lo);
    bound[e] = lo < total ? (long long)slots[lo] : (long long)slots[total - 1] + flags[total - 1];
  }
  out_len[b] = bound[1] - bound[0];
}

}  // namespace v3d

namespace v3d {
struct GridLayout {
  size_t o_off, o_min, o_err, o_k0, o_k1, o_v0, o_v1, o_flag, o_slot, o_head, o_tmp, sort_bytes, scan_bytes, total;
};
// the sort only has to look at the key bits that can differ: 3 x 18 coordinate bits + the bits of the largest cloud number
static unsigned grid_end_bit(int B) {
  unsigned cloud_bits = 1;
  while (((unsigned)(B - 1) >> cloud_bits) != 0) ++cloud_bits;
  return 3 * kGsCoordBits + cloud_bits;
}

static int grid_layout(const char* what, int B, int64_t total, int C, GridLayout& g) {
  const size_t n = (size_t)total;
  g.sort_bytes = g.scan_bytes = 0;
  unsigned long long* nullk = nullptr;
  int* nulli = nullptr;
  const unsigned end_bit = grid_end_bit(B);
  V3D_HIP_TRY(rocprim::radix_sort_pairs(nullptr, g.sort_bytes, nullk, nullk, nulli, nulli, n, 0, end_bit, nullptr), what);
  V3D_HIP_TRY(rocprim::exclusive_scan(nullptr, g.scan_bytes, nulli, nulli, 0, n, rocprim::plus<int>(), nullptr), what);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t at = off;
    off += align_up(bytes, 256);
    return at;
  };
  g.o_off = take(sizeof(long long) * ((size_t)B + 1));
  g.o_min = take(sizeof(int) * 3 * (size_t)B);
  g.o_err = take(sizeof(int));
  g.o_k0 = take(8 * n);
  g.o_k1 = take(8 * n);
  g.o_v0 = take(4 * n);
  g.o_v1 = take(4 * n);
  g.o_flag = take(4 * n);
  g.o_slot = take(4 * n);
  g.o_head = take(C > 0 ? 4 * n : 0);
  g.o_tmp = take(std::max(g.sort_bytes, g.scan_bytes));
  g.total = off;
  return 0;
}

static int grid_reduce(const char* what, const float* points, const int64_t* lengths, const float* feats, int C,
                       float* sampled_points, int64_t* out_voxels, float* out_feats, int64_t* sampled_lengths,
                       int64_t* num_sampled, int B, int64_t total, float voxel_size, bool divide, void* workspace,
                       size_t workspace_bytes, int device, void* stream_) {
  V3D_REQUIRE(B >= 0 && total >= 0, what, "negative dimension");
  V3D_REQUIRE(voxel_size > 0.f, what, "voxel_size must be positive");  // NaN fails too
  V3D_REQUIRE(num_sampled && (B == 0 || (lengths && sampled_lengths)) && (total == 0 || (points && sampled_points)),
              what, "null pointer");
  V3D_REQUIRE(C >= 0 && (C == 0 || total == 0 || (feats && out_feats)), what, "feature buffers missing");
  if (B > kGsMaxClouds || total >= (int64_t)INT32_MAX) {
    set_last_error("%s: at most %d clouds and 2^31-1 points per call", what, kGsMaxClouds);
    return V3D_ERR_SIZE_LIMIT;
  }
  DeviceGuard guard(device);
  if (guard.rc) return guard.rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (total == 0 || B == 0) {
    if (B > 0) V3D_HIP_TRY(hipMemsetAsync(sampled_lengths, 0, sizeof(int64_t) * (size_t)B, stream), what);
    V3D_HIP_TRY(hipMemsetAsync(num_sampled, 0, sizeof(int64_t), stream), what);
    return 0;
  }
  // :64: 1 / voxel_size is a Python float, rounded to float32 by the product; voxelize.py:22 divides instead
  const float inv = divide ? -voxel_size : (float)(1.0 / (double)voxel_size);

  GridLayout g;
  if (int rc = grid_layout(what, B, total, C, g)) return rc;
  const size_t n = (size_t)total;
  size_t sort_bytes = g.sort_bytes, scan_bytes = g.scan_bytes;
  const unsigned end_bit = grid_end_bit(B);
  const size_t o_off = g.o_off, o_min = g.o_min, o_err = g.o_err, o_k0 = g.o_k0, o_k1 = g.o_k1, o_v0 = g.o_v0,
               o_v1 = g.o_v1, o_flag = g.o_flag, o_slot = g.o_slot, o_head = g.o_head, o_tmp = g.o_tmp;
  Scratch scratch;
  if (int rc = scratch.acquire(what, workspace, workspace_bytes, g.total, stream)) return rc;
  char* base = scratch.base;
  long long* offsets = reinterpret_cast<long long*>(base + o_off);
  int* minv = reinterpret_cast<int*>(base + o_min);
  int* error = reinterpret_cast<int*>(base + o_err);
  unsigned long long* k0 = reinterpret_cast<unsigned long long*>(base + o_k0);
  unsigned long long* k1 = reinterpret_cast<unsigned long long*>(base + o_k1);
  int* v0 = reinterpret_cast<int*>(base + o_v0);
  int* v1 = reinterpret_cast<int*>(base + o_v1);
  int* flags = reinterpret_cast<int*>(base + o_flag);
  int* slots = reinterpret_cast<int*>(base + o_slot);
  int* head_pos = C > 0 ? reinterpret_cast<int*>(base + o_head) : nullptr;
  void* tmp = base + o_tmp;
  int rc = 0;
  do {
    hipError_t e;
    const unsigned blocks = (unsigned)ceil_div(total, kGsThreads);
    hipLaunchKernelGGL(gs_init_kernel, dim3((unsigned)ceil_div((int64_t)B * 3, 256)), dim3(256), 0, stream, minv, error,
                       sampled_lengths, num_sampled, B);
    hipLaunchKernelGGL(gs_minvox_kernel, dim3(B, kGsMinSplit), dim3(kGsThreads), 0, stream, points, lengths, offsets, minv,
                       B, inv);
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
                       sampled_points, num_sampled, (long long)total, k1, minv, out_voxels, head_pos);
    if (C > 0)
      hipLaunchKernelGGL(gs_feats_kernel, dim3((unsigned)ceil_div(total * C, kGsThreads)), dim3(kGsThreads), 0, stream,
                         feats, v1, flags, head_pos, num_sampled, C, (long long)total, out_feats);
    hipLaunchKernelGGL(gs_lengths_kernel, dim3((unsigned)ceil_div(B, 64)), dim3(64), 0, stream, offsets, flags, slots,
                       sampled_lengths, B, (long long)total);
    rc = launch_status(what);
  } while (0);
  return scratch.release(what, rc);
}
}  // namespace v3d

extern "C" {

size_t v3d_grid_subsample_workspace_bytes(int B, int64_t total, int C) {
  if (B <= 0 || total <= 0 || C < 0 || B > v3d::kGsMaxClouds || total >= (int64_t)INT32_MAX) return 0;
  v3d::GridLayout g;
  return v3d::grid_layout("grid_subsample_workspace_bytes", B, total, C, g) == 0 ? g.total : 0;
}

int v3d_grid_subsample_pack_f32(const float* points, const int64_t* lengths, float* sampled_points,
                                int64_t* sampled_lengths, int64_t* num_sampled, int B, int64_t total, float voxel_size,
                                void* workspace, size_t workspace_bytes, int device, void* stream) {
  return v3d::grid_reduce("grid_subsample_pack_mode", points, lengths, nullptr, 0, sampled_points, nullptr, nullptr,
                          sampled_lengths, num_sampled, B, total, voxel_size, false, workspace, workspace_bytes, device,
                          stream);
}

int v3d_voxelize_pack_f32(const float* points, const int64_t* lengths, const float* feats, int C, int64_t* voxels,
                          float* nodes, float* voxel_feats, int64_t* voxel_lengths, int64_t* num_voxels, int B,
                          int64_t total, float voxel_size, void* workspace, size_t workspace_bytes, int device,
                          void* stream) {
  using namespace v3d;
  const char* what = "voxelize_pack_mode";
  V3D_REQUIRE(total <= 0 || voxels, what, "null pointer");
  if (C > 0 && total > 0 && total * (int64_t)C >= ((int64_t)INT32_MAX - 1) * kGsThreads) {
    set_last_error("%s: total * C too large for one launch", what);
    return V3D_ERR_SIZE_LIMIT;
  }
  return grid_reduce(what, points, lengths, feats, C, nodes, voxels, voxel_feats, voxel_lengths, num_voxels, B, total,
                     voxel_size, true, workspace, workspace_bytes, device, stream);
}

}  // extern "C"
