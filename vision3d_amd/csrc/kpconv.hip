// kpconv.hip -- fused neighbourhood aggregation of KPConv ("fused gather + weighted sum").
// The reference (vision3d/layers/kpconv.py:110-141) materialises, per layer:
//   index_select(padded_points) (M,H,3) -> differences (M,H,K,3) -> sq_distances / neighbor_weights (M,H,K) ->
//   index_select(padded_feats) (M,H,C) -> matmul(neighbor_weights^T, neighbor_feats) (M,K,C)  [M tiny GEMMs]
//   -> sum(neighbor_feats, -1) > 0 -> neighbor_num (M,)
// i.e. the (M,H,C) gather (4.7 GB at 480k points, 38 neighbours, 64 channels) and the (M,H,K,3) differences (3.3 GB)
// go to HBM and come back.  Here one kernel produces weighted_feats (M,K,C) and neighbor_num (M,) from the inputs.
//
// Design (gfx950): one wave per query.  Lane l first owns neighbour l (index, offset from the query: one coalesced
// index load and one point gather per query, H <= 64).  The kernel-point influences ARE the A operand of
// v_mfma_f32_16x16x4_f32 (rows = 16 kernel points, k = 4 neighbours per step): lane l computes
// A[l&15][l>>4] = max(0, 1 - |offset_h - kernel_point_kp| / sigma) in place, nothing is staged.  The B operand is the
// gathered feature row read as ONE 16-byte load per lane (4 neighbours x 256 contiguous bytes per instruction); the four
// floats of a lane feed four MFMAs whose column n stands for channel 4n + j, so the four accumulators of a lane are
// again 4 consecutive channels and the result leaves as 16-byte stores, 256 contiguous bytes per kernel point.
// Exact fp32 products (fp32-input MFMA); the sum over neighbours runs in the MFMA's order (like the reference's batched
// GEMM, whose order is unspecified).
#include <algorithm>

#include "common.h"

namespace v3d {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Results are written once and read by a later kernel: streaming stores keep them from displacing the gathered rows.
__device__ __forceinline__ void kp_store(float4* dst, float a, float b, float c, float d) {
  __builtin_nontemporal_store(f32x4{a, b, c, d}, reinterpret_cast<f32x4*>(dst));
}

constexpr int kKpMaxNeighbors = 64;
constexpr int kKpMaxKernelPoints = 16;

// rowsum[n] = sum_c feats[n,c]: the density normalisation counts neighbours whose feature SUM is positive (:137-138).
__global__ __launch_bounds__(256) void kp_rowsum_kernel(const float4* __restrict__ feats, float* __restrict__ rowsum,
                                                        int64_t N, int C4) {
  const int lane = threadIdx.x & 63;
  const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float s = 0.f;
  for (int p = lane; p < C4; p += 64) {
    const float4 v = feats[n * C4 + p];
    s += (v.x + v.y) + (v.z + v.w);
  }
  for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
  if (lane == 0) rowsum[n] = s;
}

// MAXS = ceil(H / 4) rounded up to 8, 12 or 16 steps (sizes the register arrays: 4 waves per SIMD up to H = 48).
// The grid-stride loop over queries is software-pipelined: the neighbour indices of the query two strides ahead and
// the point gather of the query one stride ahead are in flight while the current query multiplies.
template <int MAXS>
__global__ __launch_bounds__(256) void kpconv_gather_kernel(const float* __restrict__ q_points,
                                                            const float* __restrict__ s_points,
                                                            const float4* __restrict__ feats,
                                                            const int64_t* __restrict__ idx,
                                                            const float* __restrict__ kernel_points,
                                                            const float* __restrict__ rowsum,
                                                            float4* __restrict__ out, int64_t* __restrict__ neighbor_num,
                                                            int64_t M, int N, int H, int K, int C4, float sigma) {
  const int lane = threadIdx.x & 63;
  const int col = lane & 15, grp = lane >> 4;
  const int64_t waves = (int64_t)gridDim.x * 4;
  const int S = (H + 3) >> 2;
  float kx = 0.f, ky = 0.f, kz = 0.f;
  const bool kvalid = col < K;
  if (kvalid) {
    kx = kernel_points[col * 3 + 0];
    ky = kernel_points[col * 3 + 1];
    kz = kernel_points[col * 3 + 2];
  }
  const int chunks = (C4 + 15) >> 4;
  const int64_t m0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  auto load_index = [&](int64_t m) -> int {  // lane l owns neighbour l of query m; -1 = none
    if (m >= M || lane >= H) return -1;
    const int64_t i = idx[m * H + lane];
    return (i >= 0 && i < N) ? (int)i : -1;
  };
  // stage 1 of query m0, stage 0 of query m0 + waves
  int row = load_index(m0);
  int row_next = load_index(m0 + waves);
  float sx = 0.f, sy = 0.f, sz = 0.f, cnt_sum = 0.f;
  auto load_point = [&](int r) {
    sx = sy = sz = cnt_sum = 0.f;
    if (r >= 0) {
      sx = s_points[(int64_t)r * 3 + 0];
      sy = s_points[(int64_t)r * 3 + 1];
      sz = s_points[(int64_t)r * 3 + 2];
      if (rowsum) cnt_sum = rowsum[r];
    }
  };
  load_point(row);
  for (int64_t m = m0; m < M; m += waves) {
    // this query's operands (requested one stride ago) ...
    const int cur_row = row;
    const float qx = q_points[m * 3 + 0], qy = q_points[m * 3 + 1], qz = q_points[m * 3 + 2];
    const float ox = sx - qx, oy = sy - qy, oz = sz - qz;             // :101-103
    const bool counted = cur_row >= 0 && cnt_sum > 0.f;               // :137-138
    // ... and the next queries' requests
    row = row_next;
    row_next = load_index(m + 2 * waves);
    load_point(row);
    if (neighbor_num) {
      const int cnt = __builtin_popcountll(__ballot(counted));
      if (lane == 0) neighbor_num[m] = cnt > 1 ? cnt : 1;             // :139
    }
    // A operand of every step: kernel point `col`, neighbour 4*st + grp
    float A[MAXS];
#pragma unroll
    for (int st = 0; st < MAXS; ++st) {
      A[st] = 0.f;
      if (st < S) {
        const int h = 4 * st + grp;
        const float nx = __shfl(ox, h, 64), ny = __shfl(oy, h, 64), nz = __shfl(oz, h, 64);
        const int r = __shfl(cur_row, h, 64);
        const float dx = nx - kx, dy = ny - ky, dz = nz - kz;        // :106-107
        const float sq = (dx * dx + dy * dy) + dz * dz;              // :108
        // :109 -- correctly rounded sqrt and division like the reference (a hardware sqrt * (1/sigma) was measured:
        // 5 % faster, but the influence loses its last bits next to the clamp, where 1 - x cancels)
        const float w = fmaxf(1.f - sqrtf(sq) / sigma, 0.f);
        A[st] = (r >= 0 && kvalid) ? w : 0.f;                        // absent neighbours sit at `inf`: influence 0
      }
    }
    for (int ch = 0; ch < chunks; ++ch) {
      const int p = ch * 16 + col;
      const bool pin = p < C4;
      float4 Bv[MAXS];
#pragma unroll
      for (int st = 0; st < MAXS; ++st) {
        Bv[st] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (st < S) {
          const int r = __shfl(cur_row, 4 * st + grp, 64);
          if (pin && r >= 0) Bv[st] = feats[(int64_t)r * C4 + p];   // :113-114
        }
      }
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
#pragma unroll
      for (int st = 0; st < MAXS; ++st) {
        if (st < S) {                                                                      // :115
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[st], Bv[st].x, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[st], Bv[st].y, acc1, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[st], Bv[st].z, acc2, 0, 0, 0);
          acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[st], Bv[st].w, acc3, 0, 0, 0);
        }
      }
      if (pin) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int kp = grp * 4 + i;  // C/D layout: row = 4*(lane>>4) + register, column = lane & 15
          if (kp < K) kp_store(&out[(m * K + kp) * C4 + p], acc0[i], acc1[i], acc2[i], acc3[i]);
        }
      }
    }
  }
}

// Backward w.r.t. the features, first half: grad_neighbors[m,h,:] = sum_kp influence[m,kp,h] * grad_weighted[m,kp,:]
// (the transpose of the forward product; the second half is the ordered row scatter of packops.hip).  Same mapping as
// the forward with the roles swapped: MFMA rows = 16 neighbours, k = 4 kernel points per step (4 steps), columns =
// channels; the influences are recomputed in the A-operand layout, the B operand is grad_weighted read as 16-byte
// pieces (256 contiguous bytes per kernel point), the result leaves as 256 contiguous bytes per neighbour.
__global__ __launch_bounds__(256) void kpconv_gather_bwd_kernel(const float* __restrict__ q_points,
                                                                const float* __restrict__ s_points,
                                                                const int64_t* __restrict__ idx,
                                                                const float* __restrict__ kernel_points,
                                                                const float4* __restrict__ grad_weighted,
                                                                float4* __restrict__ grad_neighbors, int64_t M, int N,
                                                                int H, int K, int C4, float sigma) {
  const int lane = threadIdx.x & 63;
  const int col = lane & 15, grp = lane >> 4;
  const int64_t waves = (int64_t)gridDim.x * 4;
  const int RT = (H + 15) >> 4;
  float kx[4], ky[4], kz[4];
  bool kv[4];
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    const int kp = 4 * st + grp;
    kv[st] = kp < K;
    kx[st] = kv[st] ? kernel_points[kp * 3 + 0] : 0.f;
    ky[st] = kv[st] ? kernel_points[kp * 3 + 1] : 0.f;
    kz[st] = kv[st] ? kernel_points[kp * 3 + 2] : 0.f;
  }
  const int chunks = (C4 + 15) >> 4;
  const int64_t m0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  // software pipeline over the wave's queries: neighbour indices two strides ahead, point gather one stride ahead,
  // and the grad_weighted pieces of the next (query, channel chunk) item requested before this item's MFMAs
  auto load_index = [&](int64_t m) -> int {
    if (m >= M || lane >= H) return -1;
    const int64_t i = idx[m * H + lane];
    return (i >= 0 && i < N) ? (int)i : -1;
  };
  int row = load_index(m0);
  int row_next = load_index(m0 + waves);
  float sx = 0.f, sy = 0.f, sz = 0.f;
  auto load_point = [&](int r) {
    sx = sy = sz = 0.f;
    if (r >= 0) {
      sx = s_points[(int64_t)r * 3 + 0];
      sy = s_points[(int64_t)r * 3 + 1];
      sz = s_points[(int64_t)r * 3 + 2];
    }
  };
  load_point(row);
  float4 Bn[4];
  auto load_grad = [&](int64_t m, int ch) {
    const int p = ch * 16 + col;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      Bn[st] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < M && p < C4 && kv[st]) Bn[st] = grad_weighted[(m * K + 4 * st + grp) * C4 + p];
    }
  };
  load_grad(m0, 0);
  for (int64_t m = m0; m < M; m += waves) {
    const int cur_row = row;
    const float ox = sx - q_points[m * 3 + 0], oy = sy - q_points[m * 3 + 1], oz = sz - q_points[m * 3 + 2];
    row = row_next;
    row_next = load_index(m + 2 * waves);
    load_point(row);
    float A[4][4];  // [row tile][step]: neighbour 16*rt + col, kernel point 4*st + grp
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
#pragma unroll
      for (int st = 0; st < 4; ++st) A[rt][st] = 0.f;
      if (rt < RT) {  // wave-uniform: row tiles past the table width cost nothing
        const int h = 16 * rt + col;
        const float nx = __shfl(ox, h, 64), ny = __shfl(oy, h, 64), nz = __shfl(oz, h, 64);
        const int r = __shfl(cur_row, h, 64);
#pragma unroll
        for (int st = 0; st < 4; ++st) {
          const float dx = nx - kx[st], dy = ny - ky[st], dz = nz - kz[st];
          const float sq = (dx * dx + dy * dy) + dz * dz;
          const float w = fmaxf(1.f - sqrtf(sq) / sigma, 0.f);
          A[rt][st] = (r >= 0 && kv[st]) ? w : 0.f;
        }
      }
    }
    for (int ch = 0; ch < chunks; ++ch) {
      const int p = ch * 16 + col;
      const bool pin = p < C4;
      float4 Bv[4];
#pragma unroll
      for (int st = 0; st < 4; ++st) Bv[st] = Bn[st];
      if (ch + 1 < chunks) load_grad(m, ch + 1); else load_grad(m + waves, 0);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        if (rt < RT) {
          f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
#pragma unroll
          for (int st = 0; st < 4; ++st) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[rt][st], Bv[st].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[rt][st], Bv[st].y, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[rt][st], Bv[st].z, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[rt][st], Bv[st].w, acc3, 0, 0, 0);
          }
          if (pin) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int h = 16 * rt + grp * 4 + i;
              if (h < H) kp_store(&grad_neighbors[(m * H + h) * C4 + p], acc0[i], acc1[i], acc2[i], acc3[i]);
            }
          }
        }
      }
    }
  }
}

}  // namespace v3d

extern "C" {

size_t v3d_kpconv_gather_workspace_bytes(int64_t N) { return N > 0 ? v3d::align_up(sizeof(float) * (size_t)N, 256) : 0; }

int v3d_kpconv_gather_f32(const float* q_points, const float* s_points, const float* s_feats,
                          const int64_t* neighbor_indices, const float* kernel_points, float* weighted_feats,
                          int64_t* neighbor_num, int64_t M, int64_t N, int H, int K, int C, float sigma, void* workspace,
                          size_t workspace_bytes, int device, void* stream_) {
  using namespace v3d;
  const char* what = "kpconv_gather";
  V3D_REQUIRE(M >= 0 && N >= 0 && H >= 0 && K >= 0 && C >= 0, what, "negative dimension");
  V3D_REQUIRE(H <= kKpMaxNeighbors, what, "at most 64 neighbours per query");
  V3D_REQUIRE(K <= kKpMaxKernelPoints, what, "at most 16 kernel points");
  V3D_REQUIRE(C % 4 == 0, what, "the channel count must be a multiple of 4 (16-byte row pieces)");
  V3D_REQUIRE(sigma > 0.f, what, "sigma must be positive");
  if (N >= (int64_t)INT32_MAX) {
    set_last_error("%s: N must stay below 2^31", what);
    return V3D_ERR_SIZE_LIMIT;
  }
  if (M == 0) return 0;
  V3D_REQUIRE(q_points && (H == 0 || neighbor_indices) && (K == 0 || kernel_points) && (N == 0 || (s_points && s_feats)),
              what, "null pointer");
  V3D_REQUIRE(K == 0 || C == 0 || weighted_feats, what, "null pointer");
  V3D_REQUIRE((((uintptr_t)s_feats | (uintptr_t)weighted_feats) & 15) == 0, what, "feature buffers must be 16-byte aligned");
  DeviceGuard guard(device);
  if (guard.rc) return guard.rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  float* rowsum = nullptr;
  bool own = false;
  if (neighbor_num && N > 0) {
    const size_t need = v3d_kpconv_gather_workspace_bytes(N);
    if (workspace == nullptr) {
      V3D_HIP_TRY(hipMallocAsync(reinterpret_cast<void**>(&rowsum), need, stream), what);
      own = true;
    } else if (workspace_bytes < need || ((uintptr_t)workspace & 15) != 0) {
      set_last_error("%s: workspace needs %zu bytes (got %zu)", what, need, workspace_bytes);
      return V3D_ERR_WORKSPACE;
    } else {
      rowsum = static_cast<float*>(workspace);
    }
    hipLaunchKernelGGL(kp_rowsum_kernel, dim3((unsigned)ceil_div(N, 4)), dim3(256), 0, stream,
                       reinterpret_cast<const float4*>(s_feats), rowsum, N, C / 4);
  }
  // resident grid (4 workgroups of 4 waves per CU): every wave walks its queries with the pipeline above
  const int64_t blocks = std::min<int64_t>(ceil_div(M, 4), 256 * 4);
#define V3D_KP_LAUNCH(MAXS_)                                                                                         \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(kpconv_gather_kernel<MAXS_>), dim3((unsigned)blocks), dim3(256), 0, stream,    \
                     q_points, s_points, reinterpret_cast<const float4*>(s_feats), neighbor_indices, kernel_points, \
                     rowsum, reinterpret_cast<float4*>(weighted_feats), neighbor_num, M, (int)N, H, K, C / 4, sigma)
  if (H <= 32) V3D_KP_LAUNCH(8);
  else if (H <= 48) V3D_KP_LAUNCH(12);
  else V3D_KP_LAUNCH(16);
#undef V3D_KP_LAUNCH
  int rc = launch_status(what);
  if (own) {
    hipError_t e = hipFreeAsync(rowsum, stream);
    if (rc == 0 && e != hipSuccess) {
      set_last_error("%s: hipFreeAsync: %s", what, hipGetErrorString(e));
      rc = (int)e;
    }
  }
  return rc;
}

int v3d_kpconv_gather_bwd_f32(const float* q_points, const float* s_points, const int64_t* neighbor_indices,
                              const float* kernel_points, const float* grad_weighted, float* grad_neighbors, int64_t M,
                              int64_t N, int H, int K, int C, float sigma, int device, void* stream_) {
  using namespace v3d;
  const char* what = "kpconv_gather_backward";
  V3D_REQUIRE(M >= 0 && N >= 0 && H >= 0 && K >= 0 && C >= 0, what, "negative dimension");
  V3D_REQUIRE(H <= kKpMaxNeighbors, what, "at most 64 neighbours per query");
  V3D_REQUIRE(K <= kKpMaxKernelPoints, what, "at most 16 kernel points");
  V3D_REQUIRE(C % 4 == 0, what, "the channel count must be a multiple of 4 (16-byte row pieces)");
  V3D_REQUIRE(sigma > 0.f, what, "sigma must be positive");
  if (N >= (int64_t)INT32_MAX) {
    set_last_error("%s: N must stay below 2^31", what);
    return V3D_ERR_SIZE_LIMIT;
  }
  if (M == 0 || H == 0 || C == 0) return 0;
  V3D_REQUIRE(q_points && neighbor_indices && grad_neighbors && (K == 0 || (kernel_points && grad_weighted)) &&
                  (N == 0 || s_points),
              what, "null pointer");
  V3D_REQUIRE((((uintptr_t)grad_weighted | (uintptr_t)grad_neighbors) & 15) == 0, what,
              "gradient buffers must be 16-byte aligned");
  DeviceGuard guard(device);
  if (guard.rc) return guard.rc;
  const int64_t blocks = std::min<int64_t>(ceil_div(M, 4), 256 * 8);
  hipLaunchKernelGGL(kpconv_gather_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream_),
                     q_points, s_points, neighbor_indices, kernel_points, reinterpret_cast<const float4*>(grad_weighted),
                     reinterpret_cast<float4*>(grad_neighbors), M, (int)N, H, K, C / 4, sigma);
  return launch_status(what);
}

}  // extern "C"
