// packops.hip -- row gathers of PACK mode (features row-major (N,C), neighbour tables (M,K) int64 whose entries equal
// to N mean "no neighbour") -- SURVEY 8f rank 3.  The reference builds all of them from
//   padded = cat(feats, zeros(1,C));  index_select(padded, neighbor_indices, dim=0)        (ops/index_select.py:4-33)
// followed by a reduction over K that re-reads the (M,K,C) intermediate: ops/pooling.py:6-19 (max),
// ops/nearest_interpolate.py:6-20 (first column), ops/knn_interpolate.py:43-76 (inverse-distance weights), and the
// same pattern inside layers/kpconv.py:110-124 and layers/edge_conv.py:113-122.
//
// Design (gfx950): a row of C elements is C*e/16 16-byte pieces; a wave maps lanes to (row, piece) so that one
// wave-wide load instruction moves whole rows (256 B - 1 KiB contiguous per row), the index of a row is one broadcast
// load for its lanes, and the reductions over K happen in registers -- the (M,K,C) tensor is never written.  The cat()
// with a zero row is not materialised either: index N (or anything outside [0,N)) reads as zeros.
// Backward passes are ATOMIC-FREE: the neighbour table is inverted once (stable radix sort of (source, position),
// rocPRIM primitive) into a CSR that a model can keep for every layer that shares the table; a wave then sums the rows
// of one source in ascending position order -- bit-reproducible, equal to the sequential CPU loop.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include <algorithm>

#include "common.h"

namespace v3d {

struct RowShape {
  int pieces;      // 16-byte (or unit-sized) pieces per row
  int sub_shift;   // log2 of lanes that serve one row (pieces rounded up to a power of two, at most 64)
};

static RowShape row_shape(int64_t pieces) {
  RowShape s;
  s.pieces = (int)pieces;
  int sh = 0;
  while ((1 << sh) < pieces && sh < 6) ++sh;
  s.sub_shift = sh;
  return s;
}

template <typename V>
__device__ __forceinline__ V zero_of();
template <>
__device__ __forceinline__ uint4 zero_of<uint4>() { return make_uint4(0, 0, 0, 0); }
template <>
__device__ __forceinline__ unsigned zero_of<unsigned>() { return 0u; }
template <>
__device__ __forceinline__ unsigned char zero_of<unsigned char>() { return 0; }

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_stream(uint4* dst, const uint4& v) {
  __builtin_nontemporal_store(u32x4_t{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4_t*>(dst));
}
__device__ __forceinline__ void store_stream(unsigned* dst, const unsigned& v) { __builtin_nontemporal_store(v, dst); }
__device__ __forceinline__ void store_stream(unsigned char* dst, const unsigned char& v) { *dst = v; }

// out[l, :] = src[idx[l * idx_stride], :] (zeros when the index is outside [0, N)).  Lanes: (row in wave, piece).
template <typename V>
__global__ __launch_bounds__(256) void rows_gather_kernel(const V* __restrict__ src, const int64_t* __restrict__ idx,
                                                          int64_t idx_stride, V* __restrict__ out, int64_t L, int N,
                                                          RowShape rs) {
  const int lane = threadIdx.x & 63;
  const int rows_per_wave = 64 >> rs.sub_shift;
  const int sub = lane >> rs.sub_shift, p0 = lane & ((1 << rs.sub_shift) - 1);
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  constexpr int U = 8;  // row groups in flight per wave
  const int64_t l0 = wave * (int64_t)(rows_per_wave * U) + sub;
  int64_t src_row[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t l = l0 + (int64_t)u * rows_per_wave;
    int64_t i = -1;
    if (l < L) i = idx[l * idx_stride];
    src_row[u] = (i >= 0 && i < N) ? i : -1;
  }
  for (int p = p0; p < rs.pieces; p += 64) {
    V v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = src_row[u] >= 0 ? src[src_row[u] * rs.pieces + p] : zero_of<V>();
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t l = l0 + (int64_t)u * rows_per_wave;
      if (l < L) store_stream(&out[l * rs.pieces + p], v[u]);  // written once, never re-read here
    }
  }
}

// ---- local max pooling: out[m, c] = max_k padded[idx[m,k], c]  (pooling.py:16-18) ------------------------------------
// Lanes: (query in wave, 4-channel piece).  `argk` (optional, (M,C) bytes) records the FIRST k that attains the maximum.
__global__ __launch_bounds__(256) void rows_maxpool_kernel(const float4* __restrict__ src, const int64_t* __restrict__ idx,
                                                           float4* __restrict__ out, uchar4* __restrict__ argk, int64_t M,
                                                           int N, int K, RowShape rs) {
  const int lane = threadIdx.x & 63;
  const int rows_per_wave = 64 >> rs.sub_shift;
  const int sub = lane >> rs.sub_shift, p0 = lane & ((1 << rs.sub_shift) - 1);
  const int64_t m = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * rows_per_wave + sub;
  if (m >= M) return;
  const int64_t* row = idx + m * K;
  for (int p = p0; p < rs.pieces; p += 64) {
    float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    uchar4 arg = make_uchar4(0, 0, 0, 0);
    for (int k0 = 0; k0 < K; k0 += 4) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = k0 + u;
        int64_t i = -1;
        if (k < K) i = row[k];
        v[u] = (i >= 0 && i < N) ? src[i * rs.pieces + p] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = k0 + u;
        if (k < K) {
          if (v[u].x > best.x) { best.x = v[u].x; arg.x = (unsigned char)k; }
          if (v[u].y > best.y) { best.y = v[u].y; arg.y = (unsigned char)k; }
          if (v[u].z > best.z) { best.z = v[u].z; arg.z = (unsigned char)k; }
          if (v[u].w > best.w) { best.w = v[u].w; arg.w = (unsigned char)k; }
        }
      }
    }
    out[m * rs.pieces + p] = best;
    if (argk) argk[m * rs.pieces + p] = arg;
  }
}

// ---- inverse-distance weights of knn_interpolate_pack_mode (knn_interpolate.py:65-74) -------------------------------
// One thread per query: w_k = mask_k / (|q - s_k|^2 + eps), normalised by (sum_k w_k + eps); padded entries weigh 0.
__global__ __launch_bounds__(256) void knn_interp_weights_kernel(const float* __restrict__ q, const float* __restrict__ s,
                                                                 const int64_t* __restrict__ idx, int64_t M, int N,
                                                                 int K_row, int k_used, float eps,
                                                                 float* __restrict__ w) {
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const float qx = q[m * 3 + 0], qy = q[m * 3 + 1], qz = q[m * 3 + 2];
  float total = 0.f;
  for (int k = 0; k < k_used; ++k) {
    const int64_t i = idx[m * K_row + k];
    float wk = 0.f;
    if (i >= 0 && i < N) {
      const float dx = qx - s[i * 3 + 0], dy = qy - s[i * 3 + 1], dz = qz - s[i * 3 + 2];
      wk = 1.f / (((dx * dx + dy * dy) + dz * dz) + eps);  // :70-72
    }
    w[m * k_used + k] = wk;
    total += wk;
  }
  const float denom = total + eps;  // :73
  for (int k = 0; k < k_used; ++k) w[m * k_used + k] = w[m * k_used + k] / denom;
}

// ---- out[m, :] = sum_k w[m,k] * padded[idx[m,k], :], k ascending, one rounding per multiply and per add (:74-75) ------
__global__ __launch_bounds__(256) void rows_weighted_sum_kernel(const float4* __restrict__ src,
                                                                const int64_t* __restrict__ idx, int K_row,
                                                                const float* __restrict__ w, int k_used,
                                                                float4* __restrict__ out, int64_t M, int N, RowShape rs) {
  const int lane = threadIdx.x & 63;
  const int rows_per_wave = 64 >> rs.sub_shift;
  const int sub = lane >> rs.sub_shift, p0 = lane & ((1 << rs.sub_shift) - 1);
  const int64_t m = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * rows_per_wave + sub;
  if (m >= M) return;
  const int64_t* row = idx + m * K_row;
  const float* wr = w + m * k_used;
  for (int p = p0; p < rs.pieces; p += 64) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k0 = 0; k0 < k_used; k0 += 4) {
      float4 v[4];
      float wk[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = k0 + u;
        int64_t i = -1;
        wk[u] = 0.f;
        if (k < k_used) {
          i = row[k];
          wk[u] = wr[k];
        }
        v[u] = (i >= 0 && i < N) ? src[i * rs.pieces + p] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (k0 + u < k_used) {
          acc.x = acc.x + v[u].x * wk[u];
          acc.y = acc.y + v[u].y * wk[u];
          acc.z = acc.z + v[u].z * wk[u];
          acc.w = acc.w + v[u].w * wk[u];
        }
      }
    }
    out[m * rs.pieces + p] = acc;
  }
}

// ---- global average pooling per cloud (pooling.py:22-39): one workgroup per (cloud, 64 channels) ---------------------
__global__ __launch_bounds__(1024) void rows_avgpool_kernel(const float* __restrict__ feats,
                                                            const int64_t* __restrict__ lengths, float* __restrict__ out,
                                                            int B, int C) {
  __shared__ float part[16][64];
  const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
  int64_t start = 0;
  for (int i = 0; i < b; ++i) start += lengths[i];
  const int64_t n = lengths[b];
  float acc = 0.f;
  if (c < C)
    for (int64_t r = ph; r < n; r += 16) acc += feats[(start + r) * C + c];
  part[ph][threadIdx.x & 63] = acc;
  __syncthreads();
  if (ph == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) t += part[j][threadIdx.x & 63];
    out[(int64_t)b * C + c] = t / (float)n;  // n == 0 -> NaN like torch.mean of an empty slice
  }
}

// ---- inverse of a neighbour table: CSR over sources ------------------------------------------------------------------
__global__ __launch_bounds__(256) void csr_keys_kernel(const int64_t* __restrict__ idx, int64_t L, int N,
                                                       unsigned* __restrict__ keys, int* __restrict__ vals) {
  const int64_t l = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (l >= L) return;
  const int64_t i = idx[l];
  keys[l] = (i >= 0 && i < N) ? (unsigned)i : (unsigned)N;
  vals[l] = (int)l;
}
// starts[n] = first sorted position whose source is >= n, n in [0, N]; entries [starts[N], L) are the padded ones.
__global__ __launch_bounds__(256) void csr_starts_kernel(const unsigned* __restrict__ sorted, int64_t L, int N,
                                                         int* __restrict__ starts) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n > N) return;
  int64_t lo = 0, hi = L;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (sorted[mid] < (unsigned)n) lo = mid + 1; else hi = mid;
  }
  starts[n] = (int)lo;
}

struct CsrLayout {
  size_t o_starts, o_order, o_k0, o_k1, o_v0, o_tmp, sort_bytes, total;
};
static int csr_layout(int N, int64_t L, CsrLayout& c, hipStream_t stream) {
  unsigned* nk = nullptr;
  int* nv = nullptr;
  c.sort_bytes = 0;
  unsigned end_bit = 1;
  while (((unsigned long long)N >> end_bit) != 0) ++end_bit;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, c.sort_bytes, nk, nk, nv, nv, (size_t)L, 0, end_bit, stream);
  if (e != hipSuccess) return (int)e;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t at = off;
    off += align_up(bytes, 256);
    return at;
  };
  c.o_starts = take(4 * ((size_t)N + 1));
  c.o_order = take(4 * (size_t)L);
  c.o_k0 = take(4 * (size_t)L);
  c.o_k1 = take(4 * (size_t)L);
  c.o_v0 = take(4 * (size_t)L);
  c.o_tmp = take(c.sort_bytes);
  c.total = off;
  return 0;
}

// grad_src[n, :] = sum over the table positions l that name n, ascending l, of w[l] * grad_out[row(l), :], where
//   mode 0: row(l) = l                      (index_select backward; (L,C) upstream gradient)
//   mode 1: row(l) = l / K                  (weighted-sum backward: (M,C) upstream gradient, weight per (m,k))
//   mode 2: row(l) = l / K, only channels whose argk[m,c] == l % K   (max-pool backward)
template <int MODE>
__global__ __launch_bounds__(256) void rows_scatter_kernel(const float4* __restrict__ grad_out,
                                                           const int* __restrict__ starts, const int* __restrict__ order,
                                                           const float* __restrict__ w, const uchar4* __restrict__ argk,
                                                           int K, int k_used, float4* __restrict__ grad_src, int N,
                                                           RowShape rs) {
  const int lane = threadIdx.x & 63;
  const int rows_per_wave = 64 >> rs.sub_shift;
  const int sub = lane >> rs.sub_shift, p0 = lane & ((1 << rs.sub_shift) - 1);
  const int64_t n = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * rows_per_wave + sub;
  if (n >= N) return;
  const int e0 = starts[n], e1 = starts[n + 1];
  for (int p = p0; p < rs.pieces; p += 64) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int e = e0; e < e1; e += 4) {
      float4 v[4];
      float wk[4];
      int kk[4];
      bool use[4];
      uchar4 a[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        use[u] = e + u < e1;
        const int l = use[u] ? order[e + u] : 0;
        int64_t r = l;
        kk[u] = 0;
        if (MODE != 0) {
          r = l / K;
          kk[u] = l - (int)r * K;
        }
        if (MODE == 1 && kk[u] >= k_used) use[u] = false;  // columns beyond k_used take no part (knn_interpolate.py:67)
        wk[u] = 1.f;
        if (MODE != 2 && w != nullptr && use[u]) wk[u] = (MODE == 1) ? w[r * k_used + kk[u]] : w[l];
        v[u] = use[u] ? grad_out[r * rs.pieces + p] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (MODE == 2) a[u] = use[u] ? argk[r * rs.pieces + p] : make_uchar4(255, 255, 255, 255);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (use[u]) {
          if (MODE == 2) {
            if (a[u].x == kk[u]) acc.x += v[u].x;
            if (a[u].y == kk[u]) acc.y += v[u].y;
            if (a[u].z == kk[u]) acc.z += v[u].z;
            if (a[u].w == kk[u]) acc.w += v[u].w;
          } else if (w != nullptr) {
            acc.x = acc.x + v[u].x * wk[u];
            acc.y = acc.y + v[u].y * wk[u];
            acc.z = acc.z + v[u].z * wk[u];
            acc.w = acc.w + v[u].w * wk[u];
          } else {
            acc.x += v[u].x;
            acc.y += v[u].y;
            acc.z += v[u].z;
            acc.w += v[u].w;
          }
        }
      }
    }
    grad_src[n * rs.pieces + p] = acc;
  }
}

static inline unsigned row_blocks(int64_t rows, const RowShape& rs, int groups_per_wave) {
  const int64_t per_block = (int64_t)(64 >> rs.sub_shift) * 4 * groups_per_wave;
  return (unsigned)ceil_div(rows, per_block);
}

}  // namespace v3d

extern "C" {

int v3d_index_select_rows(const void* sources, const int64_t* indices, int64_t index_stride, void* targets, int64_t L,
                          int64_t N, int64_t row_bytes, int device, void* stream_) {
  using namespace v3d;
  const char* what = "index_select_rows";
  V3D_REQUIRE(L >= 0 && N >= 0 && row_bytes >= 0 && index_stride >= 1, what, "bad dimension");
  if (L == 0 || row_bytes == 0) return 0;
  V3D_REQUIRE(indices && targets && (N == 0 || sources), what, "null pointer");
  if (N >= (int64_t)INT32_MAX || row_bytes >= (int64_t)INT32_MAX) {
    set_last_error("%s: N and row_bytes must stay below 2^31", what);
    return V3D_ERR_SIZE_LIMIT;
  }
  DeviceGuard guard(device);
  if (guard.rc) return guard.rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const uintptr_t both = (uintptr_t)sources | (uintptr_t)targets;
  if (row_bytes % 16 == 0 && both % 16 == 0) {
    const RowShape rs = row_shape(row_bytes / 16);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(rows_gather_kernel<uint4>), dim3(row_blocks(L, rs, 8)), dim3(256), 0, stream,
                       static_cast<const uint4*>(sources), indices, index_stride, static_cast<uint4*>(targets), L, (int)N,
                       rs);
  } else if (row_bytes % 4 == 0 && both % 4 == 0) {
    const RowShape rs = row_shape(row_bytes / 4);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(rows_gather_kernel<unsigned>), dim3(row_blocks(L, rs, 8)), dim3(256), 0, stream,
                       static_cast<const unsigned*>(sources), indices, index_stride, static_cast<unsigned*>(targets), L,
                       (int)N, rs);
  } else {
    const RowShape rs = row_shape(row_bytes);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(rows_gather_kernel<unsigned char>), dim3(row_blocks(L, rs, 8)), dim3(256), 0,
                       stream, static_cast<const unsigned char*>(sources), indices, index_stride,
                       static_cast<unsigned char*>(targets), L, (int)N, rs);
  }
  return launch_status(what);
}

static int check_rows_f32(const char* what, int64_t rows, int64_t N, int C, const void* a, const void* b) {
  using namespace v3d;
  V3D_REQUIRE(rows >= 0 && N >= 0 && C >= 0, what, "negative dimension");
  V3D_REQUIRE(C % 4 == 0, what, "the channel count must be a multiple of 4 (16-byte row pieces)");
  V3D_REQUIRE((((uintptr_t)a | (uintptr_t)b) & 15) == 0, what, "feature buffers must be 16-byte aligned");
  if (N >= (int64_t)INT32_MAX) {
    set_last_error("%s: N must stay below 2^31", what);
    return V3D_ERR_SIZE_LIMIT;
  }
  return 0;
}

int v3d_local_maxpool_pack_f32(const float* feats, const int64_t* neighbor_indices, float* pooled, uint8_t* argk,
                               int64_t M, int64_t N, int K, int C, int device, void* stream_) {
  using namespace v3d;
  const char* what = "local_maxpool_pack_mode";
  int rc = check_rows_f32(what, M, N, C, feats, pooled);
  if (rc) return rc;
  V3D_REQUIRE(K >= 1, what, "K must be positive");  // torch: max over an empty dimension raises
  V3D_REQUIRE(argk == nullptr || (K <= 255 && ((uintptr_t)argk & 3) == 0), what, "argk needs K <= 255, 4-byte aligned");
  if (M == 0 || C == 0) return 0;
  V3D_REQUIRE(neighbor_indices && pooled && (N == 0 || feats), what, "null pointer");
  DeviceGuard guard(device);
  if (guard.rc) return guard.rc;
  const RowShape rs = row_shape(C / 4);
  hipLaunchKernelGGL(rows_maxpool_kernel, dim3(row_blocks(M, rs, 1)), dim3(256), 0, static_cast<hipStream_t>(stream_),
                     reinterpret_cast<const float4*>(feats), neighbor_indices, reinterpret_cast<float4*>(pooled),
                     reinterpret_cast<uchar4*>(argk), M, (int)N, K, rs);
  return launch_status(what);
}

int v3d_knn_interp_weights_pack_f32(const float* q_points, const float* s_points, const int64_t* neighbor_indices,
                                    float* weights, int64_t M, int64_t N, int K, int k_used, float eps, int device,
                                    void* stream_) {
  using namespace v3d;
  const char* what = "knn_interpolate_pack_mode";
  V3D_REQUIRE(M >= 0 && N >= 0 && K >= 0 && k_used >= 0 && k_used <= K, what, "bad dimension");
  if (M == 0 || k_used == 0) return 0;
  V3D_REQUIRE(q_points && neighbor_indices && weights && (N == 0 || s_points), what, "null pointer");
  DeviceGuard guard(device);
  if (guard.rc) return guard.rc;
  hipLaunchKernelGGL(knn_interp_weights_kernel, dim3((unsigned)ceil_div(M, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream_), q_points, s_points, neighbor_indices, M, (int)N, K, k_used, eps,
                     weights);
  return launch_status(what);
}

int v3d_rows_weighted_sum_f32(const float* feats, const int64_t* neighbor_indices, const float* weights, float* out,
                              int64_t M, int64_t N, int K, int k_used, int C, int device, void* stream_) {
  using namespace v3d;
  const char* what = "rows_weighted_sum";
  int rc = check_rows_f32(what, M, N, C, feats, out);
  if (rc) return rc;
  V3D_REQUIRE(K >= 0 && k_used >= 0 && k_used <= K, what, "bad dimension");
  if (M == 0 || C == 0) return 0;
  V3D_REQUIRE(out && (k_used == 0 || (neighbor_indices && weights)) && (N == 0 || feats), what, "null pointer");
  DeviceGuard guard(device);
  if (guard.rc) return guard.rc;
  const RowShape rs = row_shape(C / 4);
  hipLaunchKernelGGL(rows_weighted_sum_kernel, dim3(row_blocks(M, rs, 1)), dim3(256), 0,
                     static_cast<hipStream_t>(stream_), reinterpret_cast<const float4*>(feats), neighbor_indices, K,
                     weights, k_used, reinterpret_cast<float4*>(out), M, (int)N, rs);
  return launch_status(what);
}

int v3d_global_avgpool_pack_f32(const float* feats, const int64_t* lengths, float* pooled, int B, int C, int device,
                                void* stream_) {
  using namespace v3d;
  const char* what = "global_avgpool_pack_mode";
  V3D_REQUIRE(B >= 0 && C >= 0, what, "negative dimension");
  if (B == 0 || C == 0) return 0;
  V3D_REQUIRE(feats && lengths && pooled, what, "null pointer");
  if (B > kMaxGridYZ) {
    set_last_error("%s: at most 65535 clouds", what);
    return V3D_ERR_SIZE_LIMIT;
  }
  DeviceGuard guard(device);
  if (guard.rc) return guard.rc;
  hipLaunchKernelGGL(rows_avgpool_kernel, dim3((unsigned)ceil_div(C, 64), B), dim3(1024), 0,
                     static_cast<hipStream_t>(stream_), feats, lengths, pooled, B, C);
  return launch_status(what);
}

size_t v3d_rows_csr_bytes(int64_t N, int64_t L) {
  using namespace v3d;
  if (N < 0 || L <= 0 || N >= (int64_t)INT32_MAX || L >= (int64_t)INT32_MAX) return 0;
  CsrLayout c;
  if (csr_layout((int)N, L, c, nullptr) != 0) return 0;
  return c.total;
}

int v3d_rows_csr_build(const int64_t* indices, int64_t L, int64_t N, void* csr, size_t csr_bytes, int device,
                       void* stream_) {
  using namespace v3d;
  const char* what = "rows_csr_build";
  V3D_REQUIRE(L >= 0 && N >= 0, what, "negative dimension");
  if (N >= (int64_t)INT32_MAX || L >= (int64_t)INT32_MAX) {
    set_last_error("%s: N and L must stay below 2^31", what);
    return V3D_ERR_SIZE_LIMIT;
  }
  if (L == 0) return 0;
  V3D_REQUIRE(indices && csr, what, "null pointer");
  DeviceGuard guard(device);
  if (guard.rc) return guard.rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  CsrLayout c;
  int rc = csr_layout((int)N, L, c, stream);
  if (rc) {
    set_last_error("%s: rocPRIM size query failed", what);
    return rc;
  }
  if (csr_bytes < c.total || ((uintptr_t)csr & 255) != 0) {
    set_last_error("%s: csr buffer needs %zu bytes, 256-B aligned (got %zu)", what, c.total, csr_bytes);
    return V3D_ERR_WORKSPACE;
  }
  char* base = static_cast<char*>(csr);
  unsigned* k0 = reinterpret_cast<unsigned*>(base + c.o_k0);
  unsigned* k1 = reinterpret_cast<unsigned*>(base + c.o_k1);
  int* v0 = reinterpret_cast<int*>(base + c.o_v0);
  int* order = reinterpret_cast<int*>(base + c.o_order);
  int* starts = reinterpret_cast<int*>(base + c.o_starts);
  unsigned end_bit = 1;
  while (((unsigned long long)N >> end_bit) != 0) ++end_bit;
  hipLaunchKernelGGL(csr_keys_kernel, dim3((unsigned)ceil_div(L, 256)), dim3(256), 0, stream, indices, L, (int)N, k0, v0);
  size_t sort_bytes = c.sort_bytes;
  hipError_t e = rocprim::radix_sort_pairs(base + c.o_tmp, sort_bytes, k0, k1, v0, order, (size_t)L, 0, end_bit, stream);
  if (e != hipSuccess) {
    set_last_error("%s: radix sort: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  hipLaunchKernelGGL(csr_starts_kernel, dim3((unsigned)ceil_div(N + 1, 256)), dim3(256), 0, stream, k1, L, (int)N,
                     starts);
  return launch_status(what);
}

int v3d_rows_scatter_f32(const float* target_grads, const void* csr, const float* weights, const uint8_t* argk, int mode,
                         float* source_grads, int64_t L, int64_t N, int K, int k_used, int C, int device, void* stream_) {
  using namespace v3d;
  const char* what = "rows_scatter";
  int rc = check_rows_f32(what, L, N, C, target_grads, source_grads);
  if (rc) return rc;
  V3D_REQUIRE(mode >= 0 && mode <= 2, what, "mode must be 0 (rows), 1 (weighted sum) or 2 (max pool)");
  V3D_REQUIRE(mode == 0 || K >= 1, what, "K must be positive");
  V3D_REQUIRE(mode != 2 || (argk && ((uintptr_t)argk & 3) == 0 && K <= 255), what, "max-pool backward needs argk");
  V3D_REQUIRE(mode != 1 || (weights && k_used >= 1 && k_used <= K), what, "weighted-sum backward needs weights");
  if (N >= (int64_t)INT32_MAX || L >= (int64_t)INT32_MAX) {
    set_last_error("%s: N and L must stay below 2^31", what);
    return V3D_ERR_SIZE_LIMIT;
  }
  if (N == 0 || C == 0) return 0;
  V3D_REQUIRE(source_grads, what, "null pointer");
  DeviceGuard guard(device);
  if (guard.rc) return guard.rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (L == 0) {
    V3D_HIP_TRY(hipMemsetAsync(source_grads, 0, sizeof(float) * (size_t)N * (size_t)C, stream), what);
    return 0;
  }
  V3D_REQUIRE(target_grads && csr, what, "null pointer");
  CsrLayout c;
  rc = csr_layout((int)N, L, c, stream);
  if (rc) {
    set_last_error("%s: rocPRIM size query failed", what);
    return rc;
  }
  const char* base = static_cast<const char*>(csr);
  const int* starts = reinterpret_cast<const int*>(base + c.o_starts);
  const int* order = reinterpret_cast<const int*>(base + c.o_order);
  const RowShape rs = row_shape(C / 4);
  const dim3 grid(row_blocks(N, rs, 1)), block(256);
  const float4* go = reinterpret_cast<const float4*>(target_grads);
  float4* gs = reinterpret_cast<float4*>(source_grads);
  const uchar4* ak = reinterpret_cast<const uchar4*>(argk);
  if (mode == 0)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(rows_scatter_kernel<0>), grid, block, 0, stream, go, starts, order, weights, ak, K,
                       k_used, gs, (int)N, rs);
  else if (mode == 1)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(rows_scatter_kernel<1>), grid, block, 0, stream, go, starts, order, weights, ak, K,
                       k_used, gs, (int)N, rs);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(rows_scatter_kernel<2>), grid, block, 0, stream, go, starts, order, weights, ak, K,
                       k_used, gs, (int)N, rs);
  return launch_status(what);
}

}  // extern "C"
