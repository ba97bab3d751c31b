// partition.hip -- point-to-node partition (ops/point_cloud_partition.py:40-104).
// The reference materialises the (M,N) node-to-point distance matrix, takes its column argmin, scatters a boolean (M,N)
// matching matrix, masks the distance matrix with it and runs a row top-k over all N columns per node -- four (M,N)
// tensors for what is a nearest-node assignment followed by a per-node sort of the node's OWN points.
//
// Design (gfx950): nothing of size M x N exists.
//   1. assign: one thread per point, nodes broadcast from LDS in tiles; nearest node by exact squared distance in the
//      reference order ((dx*dx + dy*dy) + dz*dz, strict <: the lowest node index wins ties) -> 64-bit key
//      (node << 32 | float bits of d^2), value = point index;
//   2. ONE stable device radix sort (rocPRIM primitive) orders the points by (node, distance, index);
//   3. node starts by bisection -> sizes, masks, maximum size; a second entry point cuts the first K of every node out
//      of the sorted order as node_knn_indices / node_knn_masks (padding = N, like the reference's masked_fill).
#include <cstring>

#include <rocprim/rocprim.hpp>

#include <algorithm>

#include "common.h"

namespace v3d {

constexpr int kP2nThreads = 256;
constexpr int kP2nTile = 1024;  // nodes per LDS tile

__global__ __launch_bounds__(kP2nThreads) void p2n_assign_kernel(const float* __restrict__ points,
                                                                 const float* __restrict__ nodes, int64_t N, int M,
                                                                 int64_t* __restrict__ point_to_node,
                                                                 unsigned long long* __restrict__ keys,
                                                                 int* __restrict__ vals) {
  __shared__ float4 tile[kP2nTile];
  const int64_t p = (int64_t)blockIdx.x * kP2nThreads + threadIdx.x;
  float px = 0.f, py = 0.f, pz = 0.f;
  if (p < N) {
    px = points[p * 3 + 0];
    py = points[p * 3 + 1];
    pz = points[p * 3 + 2];
  }
  float best = INFINITY;
  int best_i = 0;
  for (int m0 = 0; m0 < M; m0 += kP2nTile) {
    const int cnt = min(kP2nTile, M - m0);
    __syncthreads();
    for (int i = threadIdx.x; i < cnt; i += kP2nThreads)
      tile[i] = make_float4(nodes[(int64_t)(m0 + i) * 3 + 0], nodes[(int64_t)(m0 + i) * 3 + 1],
                            nodes[(int64_t)(m0 + i) * 3 + 2], 0.f);
    __syncthreads();
    for (int i = 0; i < cnt; ++i) {
      const float4 nd = tile[i];  // same address in every lane: LDS broadcast
      const float dx = nd.x - px, dy = nd.y - py, dz = nd.z - pz;
      const float d = (dx * dx + dy * dy) + dz * dz;
      if (d < best) {
        best = d;
        best_i = m0 + i;
      }
    }
  }
  if (p < N) {
    point_to_node[p] = best_i;
    keys[p] = ((unsigned long long)(unsigned)best_i << 32) | (unsigned long long)__float_as_uint(best);
    vals[p] = (int)p;
  }
}

// starts[m] = first sorted position whose node is >= m (m in [0, M]); sizes / masks / maximum from them.
__global__ __launch_bounds__(256) void p2n_starts_kernel(const unsigned long long* __restrict__ sorted, int64_t N, int M,
                                                         int* __restrict__ starts) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m > M) return;
  int64_t lo = 0, hi = N;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((unsigned)(sorted[mid] >> 32) < (unsigned)m) lo = mid + 1; else hi = mid;
  }
  starts[m] = (int)lo;
}
__global__ __launch_bounds__(256) void p2n_sizes_kernel(const int* __restrict__ starts, int M,
                                                        int64_t* __restrict__ node_sizes,
                                                        unsigned char* __restrict__ node_masks,
                                                        unsigned long long* __restrict__ max_size) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  int sz = 0;
  if (m < M) {
    sz = starts[m + 1] - starts[m];
    if (node_sizes) node_sizes[m] = sz;
    if (node_masks) node_masks[m] = sz > 0 ? 1 : 0;
  }
  for (int d = 32; d >= 1; d >>= 1) sz = max(sz, __shfl_xor(sz, d, 64));
  if ((threadIdx.x & 63) == 0 && sz > 0 && max_size) atomicMax(max_size, (unsigned long long)sz);  // integer atomic
}

__global__ __launch_bounds__(256) void p2n_knn_kernel(const int* __restrict__ starts, const int* __restrict__ order,
                                                      int64_t N, int M, int K, int64_t* __restrict__ knn_indices,
                                                      unsigned char* __restrict__ knn_masks) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (int64_t)M * K) return;
  const int m = (int)(t / K), j = (int)(t - (int64_t)m * K);
  const int s = starts[m], e = starts[m + 1];
  const bool own = s + j < e;
  knn_indices[t] = own ? (int64_t)order[s + j] : N;  // point_cloud_partition.py:100
  if (knn_masks) knn_masks[t] = own ? 1 : 0;         // :99
}

struct P2nLayout {
  size_t o_starts, o_order, o_k0, o_k1, o_v0, o_tmp, sort_bytes, total;
};
static int p2n_layout(int64_t N, int M, P2nLayout& c, hipStream_t stream) {
  unsigned long long* nk = nullptr;
  int* nv = nullptr;
  c.sort_bytes = 0;
  unsigned mbits = 1;
  while (((unsigned long long)M >> mbits) != 0) ++mbits;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, c.sort_bytes, nk, nk, nv, nv, (size_t)N, 0, 32 + mbits, stream);
  if (e != hipSuccess) return (int)e;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t at = off;
    off += align_up(bytes, 256);
    return at;
  };
  c.o_starts = take(4 * ((size_t)M + 1));
  c.o_order = take(4 * (size_t)N);
  c.o_k0 = take(8 * (size_t)N);
  c.o_k1 = take(8 * (size_t)N);
  c.o_v0 = take(4 * (size_t)N);
  c.o_tmp = take(c.sort_bytes);
  c.total = off;
  return 0;
}

}  // namespace v3d

extern "C" {

size_t v3d_point_to_node_workspace_bytes(int64_t N, int64_t M) {
  using namespace v3d;
  if (N <= 0 || M <= 0 || N >= (int64_t)INT32_MAX || M >= (int64_t)INT32_MAX) return 0;
  P2nLayout c;
  if (p2n_layout(N, (int)M, c, nullptr) != 0) return 0;
  return c.total;
}

int v3d_point_to_node_partition_f32(const float* points, const float* nodes, int64_t N, int64_t M,
                                    int64_t* point_to_node, int64_t* node_sizes, uint8_t* node_masks, int64_t* max_size,
                                    void* workspace, size_t workspace_bytes, int device, void* stream_) {
  using namespace v3d;
  const char* what = "point_to_node_partition";
  V3D_REQUIRE(N >= 0 && M >= 0, what, "negative dimension");
  V3D_REQUIRE(M > 0 || N == 0, what, "no nodes");
  if (N >= (int64_t)INT32_MAX || M >= (int64_t)INT32_MAX) {
    set_last_error("%s: N and M must stay below 2^31", what);
    return V3D_ERR_SIZE_LIMIT;
  }
  DeviceGuard guard(device);
  if (guard.rc) return guard.rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (max_size) V3D_HIP_TRY(hipMemsetAsync(max_size, 0, sizeof(int64_t), stream), what);
  if (M > 0 && node_sizes) V3D_HIP_TRY(hipMemsetAsync(node_sizes, 0, sizeof(int64_t) * (size_t)M, stream), what);
  if (M > 0 && node_masks) V3D_HIP_TRY(hipMemsetAsync(node_masks, 0, (size_t)M, stream), what);
  if (N == 0) return 0;
  V3D_REQUIRE(points && nodes && point_to_node && workspace, what, "null pointer");
  P2nLayout c;
  int rc = p2n_layout(N, (int)M, c, stream);
  if (rc) {
    set_last_error("%s: rocPRIM size query failed", what);
    return rc;
  }
  if (workspace_bytes < c.total || ((uintptr_t)workspace & 255) != 0) {
    set_last_error("%s: workspace needs %zu bytes, 256-B aligned (got %zu)", what, c.total, workspace_bytes);
    return V3D_ERR_WORKSPACE;
  }
  char* base = static_cast<char*>(workspace);
  unsigned long long* k0 = reinterpret_cast<unsigned long long*>(base + c.o_k0);
  unsigned long long* k1 = reinterpret_cast<unsigned long long*>(base + c.o_k1);
  int* v0 = reinterpret_cast<int*>(base + c.o_v0);
  int* order = reinterpret_cast<int*>(base + c.o_order);
  int* starts = reinterpret_cast<int*>(base + c.o_starts);
  unsigned mbits = 1;
  while (((unsigned long long)M >> mbits) != 0) ++mbits;
  hipLaunchKernelGGL(p2n_assign_kernel, dim3((unsigned)ceil_div(N, kP2nThreads)), dim3(kP2nThreads), 0, stream, points,
                     nodes, N, (int)M, point_to_node, k0, v0);
  size_t sort_bytes = c.sort_bytes;
  hipError_t e = rocprim::radix_sort_pairs(base + c.o_tmp, sort_bytes, k0, k1, v0, order, (size_t)N, 0, 32 + mbits, stream);
  if (e != hipSuccess) {
    set_last_error("%s: radix sort: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  hipLaunchKernelGGL(p2n_starts_kernel, dim3((unsigned)ceil_div(M + 1, 256)), dim3(256), 0, stream, k1, N, (int)M, starts);
  hipLaunchKernelGGL(p2n_sizes_kernel, dim3((unsigned)ceil_div(M, 256)), dim3(256), 0, stream, starts, (int)M, node_sizes,
                     node_masks, reinterpret_cast<unsigned long long*>(max_size));
  return launch_status(what);
}

int v3d_point_to_node_knn(const void* workspace, int64_t N, int64_t M, int K, int64_t* node_knn_indices,
                          uint8_t* node_knn_masks, int device, void* stream_) {
  using namespace v3d;
  const char* what = "point_to_node_partition";
  V3D_REQUIRE(N >= 0 && M >= 0 && K >= 0, what, "negative dimension");
  if (M == 0 || K == 0) return 0;
  V3D_REQUIRE(workspace && node_knn_indices && N > 0, what, "null pointer");
  DeviceGuard guard(device);
  if (guard.rc) return guard.rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  P2nLayout c;
  int rc = p2n_layout(N, (int)M, c, stream);
  if (rc) {
    set_last_error("%s: rocPRIM size query failed", what);
    return rc;
  }
  const char* base = static_cast<const char*>(workspace);
  hipLaunchKernelGGL(p2n_knn_kernel, dim3((unsigned)ceil_div(M * K, 256)), dim3(256), 0, stream,
                     reinterpret_cast<const int*>(base + c.o_starts), reinterpret_cast<const int*>(base + c.o_order), N,
                     (int)M, K, node_knn_indices, node_knn_masks);
  return launch_status(what);
}

}  // extern "C"
