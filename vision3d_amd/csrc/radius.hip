// radius.hip -- radius neighbours in PACK mode (variable-length clouds concatenated: points (total,3) + lengths (B,)).
// Replaces `radius_search_pack_mode` of the reference (vision3d/ops/radius_search.py:44-71): pack -> padded batch ->
// pykeops brute-force kNN(limit) -> mask (dist > radius) -> batch -> pack, with Python loops and .item() syncs in the
// conversions (ops/conversion.py:7-12,69-81).  Same result definition: per query, the supports of its own cloud with
// NOT(sqrt(d2) > radius), ascending by (d2, support index), the first `limit` of them, padded with total_s.
//
// Design (gfx950): no padding, no host synchronisation, no brute force.  Per call:
//   1. rs_bbox / rs_plan : bounding box per support cloud; a uniform grid per cloud with cell edge h >= 1.001 * radius
//                          (enlarged until the grid has at most max(4096, 4*points) cells and <= 1024 cells per axis);
//   2. rs_count/scan/fill: counting sort of the supports by cell (32-bit integer atomics on the cell counters only),
//                          writing (x, y, z, index) as one 16-byte record per support in cell order;
//   3. rs_query          : one WAVE per query.  The 27 neighbouring cells are 9 contiguous runs of records (cells are
//                          x-major); 64 lanes test 64 candidates at once; hits are inserted into a lane-distributed
//                          sorted list keyed by (d2, index) (whole-wave DPP shift, no LDS), bounded by the current
//                          `limit`-th entry.  The result does not depend on the order candidates arrive in.
// Distances use the reference expression order (dx*dx + dy*dy) + dz*dz with one rounding per operation.
#include <algorithm>

#include "common.h"

namespace v3d {

constexpr int kRsThreads = 256;
constexpr int kRsMaxDim = 1024;
constexpr int kRsMaxClouds = 4096;  // plan rows are searched linearly / by bisection per query

struct RsPlan {  // one per cloud
  float ox, oy, oz, inv_h;
  int dx, dy, dz, cell_off;
  long long s_off, q_off, s_len, q_len;
};

static int64_t rs_cell_capacity(int B, int64_t total_s) { return 4096ll * B + 4ll * total_s + 1; }

struct RsWorkspace {
  RsPlan* plan;     // [B]
  float* bbox;      // [B][6]
  int* cell_count;  // [cap]
  int* cell_start;  // [cap + 1]
  float4* sorted;   // [total_s]
  size_t bytes;
};

static RsWorkspace rs_carve(void* base, int B, int64_t total_s) {
  RsWorkspace w;
  size_t off = 0;
  auto take = [&](size_t n) {
    void* p = base ? static_cast<char*>(base) + off : nullptr;
    off += align_up(n, 256);
    return p;
  };
  const size_t cap = (size_t)rs_cell_capacity(B, total_s);
  w.plan = static_cast<RsPlan*>(take(sizeof(RsPlan) * (size_t)B));
  w.bbox = static_cast<float*>(take(sizeof(float) * 6 * (size_t)B));
  w.cell_count = static_cast<int*>(take(sizeof(int) * cap));
  w.cell_start = static_cast<int*>(take(sizeof(int) * (cap + 1)));
  w.sorted = static_cast<float4*>(take(sizeof(float4) * (size_t)std::max<int64_t>(total_s, 1)));
  w.bytes = off;
  return w;
}

// One workgroup per cloud: min / max corner of its supports.
__global__ __launch_bounds__(kRsThreads) void rs_bbox_kernel(const float* __restrict__ s, const int64_t* __restrict__ s_len,
                                                             float* __restrict__ bbox) {
  __shared__ float red[6][kRsThreads / 64];
  const int b = blockIdx.x, tid = threadIdx.x;
  long long off = 0;
  for (int i = 0; i < b; ++i) off += s_len[i];
  const long long n = s_len[b];
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (long long i = tid; i < n; i += kRsThreads) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = s[(off + i) * 3 + a];
      lo[a] = fminf(lo[a], v);
      hi[a] = fmaxf(hi[a], v);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    for (int d = 32; d >= 1; d >>= 1) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], d, 64));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], d, 64));
    }
    if ((tid & 63) == 0) {
      red[a][tid >> 6] = lo[a];
      red[3 + a][tid >> 6] = hi[a];
    }
  }
  __syncthreads();
  if (tid < 6) {
    float v = red[tid][0];
    for (int w = 1; w < kRsThreads / 64; ++w) v = tid < 3 ? fminf(v, red[tid][w]) : fmaxf(v, red[tid][w]);
    bbox[b * 6 + tid] = v;
  }
}

// Cell coordinate along one axis; the same expression for supports, grid dimensions and queries.
__device__ __forceinline__ int rs_cell_coord(float p, float o, float inv_h) {
  const float c = floorf((p - o) * inv_h);
  return (int)fminf(fmaxf(c, -2.0f), (float)(kRsMaxDim + 2));
}

// One thread: grids, cell offsets and point offsets of all clouds (B is small; everything else waits on this anyway).
__global__ void rs_plan_kernel(const float* __restrict__ bbox, const int64_t* __restrict__ q_len,
                               const int64_t* __restrict__ s_len, RsPlan* __restrict__ plan, int* __restrict__ cell_start,
                               int B, float radius, long long total_s) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  long long s_off = 0, q_off = 0;
  int cell_off = 0;
  for (int b = 0; b < B; ++b) {
    RsPlan p;
    p.s_off = s_off;
    p.q_off = q_off;
    p.s_len = s_len[b];
    p.q_len = q_len[b];
    p.ox = p.oy = p.oz = 0.f;
    p.inv_h = 1.f;
    p.dx = p.dy = p.dz = 1;
    if (p.s_len > 0) {
      const float* bb = bbox + b * 6;
      p.ox = bb[0];
      p.oy = bb[1];
      p.oz = bb[2];
      const long long cap = p.s_len * 4 > 4096 ? p.s_len * 4 : 4096;
      float h = fmaxf(radius, 1e-30f) * 1.001f;
      for (int it = 0; it < 400; ++it) {
        p.inv_h = 1.0f / h;
        p.dx = rs_cell_coord(bb[3], p.ox, p.inv_h) + 1;
        p.dy = rs_cell_coord(bb[4], p.oy, p.inv_h) + 1;
        p.dz = rs_cell_coord(bb[5], p.oz, p.inv_h) + 1;
        if (p.dx <= kRsMaxDim && p.dy <= kRsMaxDim && p.dz <= kRsMaxDim && (long long)p.dx * p.dy * p.dz <= cap) break;
        h *= 1.25f;
      }
    }
    p.cell_off = cell_off;
    plan[b] = p;
    cell_off += p.dx * p.dy * p.dz;
    s_off += p.s_len;
    q_off += p.q_len;
  }
  cell_start[cell_off] = (int)total_s;  // end of the last cloud's last cell
}

__device__ __forceinline__ int rs_find_cloud(const RsPlan* __restrict__ plan, int B, long long i, bool query) {
  int lo = 0, hi = B - 1;  // last cloud whose offset is <= i (empty clouds share an offset: the last one wins, skip them)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    const long long off = query ? plan[mid].q_off : plan[mid].s_off;
    if (off <= i) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__device__ __forceinline__ int rs_support_cell(const RsPlan& p, float x, float y, float z) {
  const int cx = min(max(rs_cell_coord(x, p.ox, p.inv_h), 0), p.dx - 1);
  const int cy = min(max(rs_cell_coord(y, p.oy, p.inv_h), 0), p.dy - 1);
  const int cz = min(max(rs_cell_coord(z, p.oz, p.inv_h), 0), p.dz - 1);
  return p.cell_off + (cz * p.dy + cy) * p.dx + cx;
}

__global__ __launch_bounds__(kRsThreads) void rs_count_kernel(const float* __restrict__ s, const RsPlan* __restrict__ plan,
                                                              int* __restrict__ cell_count, int B, long long total_s) {
  const long long i = (long long)blockIdx.x * kRsThreads + threadIdx.x;
  if (i >= total_s) return;
  const RsPlan p = plan[rs_find_cloud(plan, B, i, false)];
  atomicAdd(&cell_count[rs_support_cell(p, s[i * 3], s[i * 3 + 1], s[i * 3 + 2])], 1);
}

// One workgroup per cloud: exclusive scan of its cells' counts (offset by the cloud's first support), cursors reset.
__global__ __launch_bounds__(1024) void rs_scan_kernel(const RsPlan* __restrict__ plan, int* __restrict__ cell_count,
                                                       int* __restrict__ cell_start) {
  __shared__ int wave_tot[16];
  __shared__ int carry_s;
  const RsPlan p = plan[blockIdx.x];
  const int ncell = p.dx * p.dy * p.dz;
  int* cnt = cell_count + p.cell_off;
  int* start = cell_start + p.cell_off;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = (int)p.s_off;
  __syncthreads();
  constexpr int kItems = 8;
  for (int base = 0; base < ncell; base += 1024 * kItems) {
    const int i0 = base + tid * kItems;
    int v[kItems], sum = 0;
#pragma unroll
    for (int e = 0; e < kItems; ++e) {
      v[e] = i0 + e < ncell ? cnt[i0 + e] : 0;
      sum += v[e];
    }
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d, 64);
      if (lane >= d) incl += t;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int run = carry_s + incl - sum;
    for (int q = 0; q < wave; ++q) run += wave_tot[q];
#pragma unroll
    for (int e = 0; e < kItems; ++e) {
      if (i0 + e < ncell) {
        start[i0 + e] = run;
        cnt[i0 + e] = 0;
      }
      run += v[e];
    }
    __syncthreads();
    if (tid == 1023) carry_s = run;
    __syncthreads();
  }
}

__global__ __launch_bounds__(kRsThreads) void rs_fill_kernel(const float* __restrict__ s, const RsPlan* __restrict__ plan,
                                                             const int* __restrict__ cell_start, int* __restrict__ cursor,
                                                             float4* __restrict__ sorted, int B, long long total_s) {
  const long long i = (long long)blockIdx.x * kRsThreads + threadIdx.x;
  if (i >= total_s) return;
  const RsPlan p = plan[rs_find_cloud(plan, B, i, false)];
  const float x = s[i * 3], y = s[i * 3 + 1], z = s[i * 3 + 2];
  const int c = rs_support_cell(p, x, y, z);
  const int slot = cell_start[c] + atomicAdd(&cursor[c], 1);
  sorted[slot] = make_float4(x, y, z, __int_as_float((int)i));
}

// ---- lane-distributed list sorted ascending by (d2, index); 64*R entries; sentinel = (+inf, INT_MAX) -----------------
__device__ __forceinline__ int rs_shr1(int v, int fill) {  // lane l <- lane l-1, lane 0 <- fill
  return __builtin_amdgcn_update_dpp(fill, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ float rs_shr1(float v, float fill) {
  return __int_as_float(rs_shr1(__float_as_int(v), __float_as_int(fill)));
}
__device__ __forceinline__ float rs_lane(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

template <int R>
struct RsList {
  float d[R];
  int i[R];
  __device__ __forceinline__ void insert(float cd, int ci, int lane) {
    int pos = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) pos += __builtin_popcountll(__ballot(d[r] < cd || (d[r] == cd && i[r] < ci)));
    float carry_d = cd;
    int carry_i = ci;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int lp = pos - 64 * r;
      const float last_d = rs_lane(d[r], 63);
      const int last_i = __builtin_amdgcn_readlane(i[r], 63);
      const float sd = rs_shr1(d[r], carry_d);
      const int si = rs_shr1(i[r], carry_i);
      const bool take_new = lane == lp, take_shift = lane > lp;
      d[r] = take_new ? cd : (take_shift ? sd : d[r]);
      i[r] = take_new ? ci : (take_shift ? si : i[r]);
      carry_d = last_d;
      carry_i = last_i;
    }
  }
  __device__ __forceinline__ void at(int rank, float& od, int& oi) const {  // rank is wave-uniform
    od = rs_lane(d[0], rank & 63);
    oi = __builtin_amdgcn_readlane(i[0], rank & 63);
#pragma unroll
    for (int r = 1; r < R; ++r) {
      const float d2 = rs_lane(d[r], rank & 63);
      const int i2 = __builtin_amdgcn_readlane(i[r], rank & 63);
      if ((rank >> 6) == r) {
        od = d2;
        oi = i2;
      }
    }
  }
};

template <int R>
__global__ __launch_bounds__(kRsThreads) void rs_query_kernel(const float* __restrict__ q, const RsPlan* __restrict__ plan,
                                                              const int* __restrict__ cell_start,
                                                              const float4* __restrict__ sorted,
                                                              int64_t* __restrict__ out, int B, long long total_q,
                                                              long long total_s, float radius, int limit) {
  const int lane = threadIdx.x & 63;
  const long long qi = (long long)blockIdx.x * (kRsThreads / 64) + (threadIdx.x >> 6);
  if (qi >= total_q) return;
  const RsPlan p = plan[rs_find_cloud(plan, B, qi, true)];
  const float qx = q[qi * 3], qy = q[qi * 3 + 1], qz = q[qi * 3 + 2];
  RsList<R> list;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    list.d[r] = INFINITY;
    list.i[r] = INT32_MAX;
  }
  float wd = INFINITY;  // current `limit`-th entry: a candidate must come before it
  int wi = INT32_MAX;
  if (p.s_len > 0) {
    const int cx = rs_cell_coord(qx, p.ox, p.inv_h), cy = rs_cell_coord(qy, p.oy, p.inv_h),
              cz = rs_cell_coord(qz, p.oz, p.inv_h);
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, p.dx - 1);
    if (x0 <= x1) {
      for (int zz = max(cz - 1, 0); zz <= min(cz + 1, p.dz - 1); ++zz) {
        for (int yy = max(cy - 1, 0); yy <= min(cy + 1, p.dy - 1); ++yy) {
          const int c0 = p.cell_off + (zz * p.dy + yy) * p.dx;
          const int beg = cell_start[c0 + x0], end = cell_start[c0 + x1 + 1];
          for (int j0 = beg; j0 < end; j0 += 64) {
            const int j = j0 + lane;
            const bool valid = j < end;
            const float4 s = sorted[valid ? j : beg];
            const float dx = qx - s.x, dy = qy - s.y, dz = qz - s.z;
            const float d2 = (dx * dx + dy * dy) + dz * dz;
            const int si = __float_as_int(s.w);
            const bool inside = valid && !(sqrtf(d2) > radius);  // radius_search.py:66 masks dist > radius
            unsigned long long m = __ballot(inside && (d2 < wd || (d2 == wd && si < wi)));
            while (m) {
              const int l = __builtin_ffsll((long long)m) - 1;
              list.insert(rs_lane(d2, l), __builtin_amdgcn_readlane(si, l), lane);
              list.at(limit - 1, wd, wi);
              m = __ballot(inside && (d2 < wd || (d2 == wd && si < wi))) & ~((2ull << l) - 1ull);
            }
          }
        }
      }
    }
  }
  int64_t* o = out + qi * limit;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int rank = 64 * r + lane;
    if (rank < limit) o[rank] = list.i[r] == INT32_MAX ? (int64_t)total_s : (int64_t)list.i[r];
  }
}

}  // namespace v3d

extern "C" {

size_t v3d_radius_search_workspace_bytes(int B, int64_t total_s) {
  if (B <= 0 || total_s < 0) return 0;
  return v3d::rs_carve(nullptr, B, total_s).bytes;
}

int v3d_radius_search_pack_f32(const float* q_points, const float* s_points, const int64_t* q_lengths,
                               const int64_t* s_lengths, int64_t* neighbor_indices, int B, int64_t total_q,
                               int64_t total_s, float radius, int neighbor_limit, void* workspace,
                               size_t workspace_bytes, int device, void* stream_) {
  using namespace v3d;
  const char* what = "radius_search_pack_mode";
  V3D_REQUIRE(B >= 0 && total_q >= 0 && total_s >= 0 && neighbor_limit >= 0, what, "negative dimension");
  if (total_q == 0 || neighbor_limit == 0) return 0;
  V3D_REQUIRE(B > 0, what, "points without clouds");
  V3D_REQUIRE(q_points && q_lengths && s_lengths && neighbor_indices && (total_s == 0 || s_points), what,
              "null pointer");
  V3D_REQUIRE(radius >= 0.f, what, "negative radius");  // NaN fails too
  if (neighbor_limit > 128) {
    set_last_error("%s: neighbor_limit=%d exceeds 128", what, neighbor_limit);
    return V3D_ERR_K_TOO_LARGE;
  }
  if (B > kRsMaxClouds || total_s >= (int64_t)INT32_MAX - 1 ||
      ceil_div(total_q, kRsThreads / 64) > (int64_t)INT32_MAX) {
    set_last_error("%s: at most %d clouds, 2^31-2 supports and 2^33 queries per call", what, kRsMaxClouds);
    return V3D_ERR_SIZE_LIMIT;
  }
  DeviceGuard guard(device);
  if (guard.rc) return guard.rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const size_t need = rs_carve(nullptr, B, total_s).bytes;
  void* base = workspace;
  bool own = false;
  if (base == nullptr) {
    V3D_HIP_TRY(hipMallocAsync(&base, need, stream), what);
    own = true;
  } else if (workspace_bytes < need || ((uintptr_t)base & 255) != 0) {
    set_last_error("%s: workspace needs %zu bytes, 256-B aligned (got %zu)", what, need, workspace_bytes);
    return V3D_ERR_WORKSPACE;
  }
  RsWorkspace ws = rs_carve(base, B, total_s);
  int rc = 0;
  do {
    hipError_t e = hipMemsetAsync(ws.cell_count, 0, sizeof(int) * (size_t)rs_cell_capacity(B, total_s), stream);
    if (e != hipSuccess) {
      set_last_error("%s: hipMemsetAsync: %s", what, hipGetErrorString(e));
      rc = (int)e;
      break;
    }
    hipLaunchKernelGGL(rs_bbox_kernel, dim3(B), dim3(kRsThreads), 0, stream, s_points, s_lengths, ws.bbox);
    hipLaunchKernelGGL(rs_plan_kernel, dim3(1), dim3(64), 0, stream, ws.bbox, q_lengths, s_lengths, ws.plan,
                       ws.cell_start, B, radius, (long long)total_s);
    if (total_s > 0) {
      const unsigned sb = (unsigned)ceil_div(total_s, kRsThreads);
      hipLaunchKernelGGL(rs_count_kernel, dim3(sb), dim3(kRsThreads), 0, stream, s_points, ws.plan, ws.cell_count, B,
                         (long long)total_s);
      hipLaunchKernelGGL(rs_scan_kernel, dim3(B), dim3(1024), 0, stream, ws.plan, ws.cell_count, ws.cell_start);
      hipLaunchKernelGGL(rs_fill_kernel, dim3(sb), dim3(kRsThreads), 0, stream, s_points, ws.plan, ws.cell_start,
                         ws.cell_count, ws.sorted, B, (long long)total_s);
    }
    const unsigned qb = (unsigned)ceil_div(total_q, kRsThreads / 64);
    if (neighbor_limit <= 64)
      hipLaunchKernelGGL(rs_query_kernel<1>, dim3(qb), dim3(kRsThreads), 0, stream, q_points, ws.plan, ws.cell_start,
                         ws.sorted, neighbor_indices, B, (long long)total_q, (long long)total_s, radius,
                         neighbor_limit);
    else
      hipLaunchKernelGGL(rs_query_kernel<2>, dim3(qb), dim3(kRsThreads), 0, stream, q_points, ws.plan, ws.cell_start,
                         ws.sorted, neighbor_indices, B, (long long)total_q, (long long)total_s, radius,
                         neighbor_limit);
    rc = launch_status(what);
  } while (0);
  if (own) {
    hipError_t e = hipFreeAsync(base, stream);
    if (rc == 0 && e != hipSuccess) {
      set_last_error("%s: hipFreeAsync: %s", what, hipGetErrorString(e));
      rc = (int)e;
    }
  }
  return rc;
}

}  // extern "C"
