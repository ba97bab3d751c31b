// radius.hip -- radius neighbours, radius counts and kNN in PACK mode (variable-length clouds concatenated: points (total,3) + lengths (B,)).
// Replaces `radius_search_pack_mode` of the reference (vision3d/ops/radius_search.py:44-71): pack -> padded batch ->
// pykeops brute-force kNN(limit) -> mask (dist > radius) -> batch -> pack, with Python loops and .item() syncs in the
// conversions (ops/conversion.py:7-12,69-81).  Same result definition: per query, the supports of its own cloud with
// NOT(sqrt(d2) > radius), ascending by (d2, support index), the first `limit` of them, padded with total_s.
//
// Design (gfx950): no padding, no host synchronisation, no brute force.  Per call:
//   1. rs_bbox / rs_plan : bounding box per support cloud (several workgroups per cloud, integer atomics on an
//                          order-preserving encoding); a uniform grid per cloud with cell edge h >= 1.001 * radius
//                          (enlarged until the grid has at most max(4096, 4*points) cells and <= 1024 cells per axis);
//   2. rs_count/scan/fill: counting sort of the supports by cell (32-bit integer atomics on the cell counters only, one
//                          rocprim scan over the counters of all clouds), writing (x, y, z, index) as one 16-byte
//                          record per support in cell order;
//   3. rs_query_*        : one WAVE per query.  The 27 neighbouring cells are 9 contiguous runs of records (cells are
//                          x-major) whose bounds lanes 0..8 fetch in one round trip; 64 lanes test 64 candidates of the
//                          concatenated runs at once.  limit <= 64: hits are batched in LDS as 64-bit (d2, index) keys
//                          and merged into the kept list by a bitonic network; limit <= 512: lane-distributed sorted
//                          list with a whole-wave DPP shift per hit.  rs_count_query: popcount of the hits.
//                          The result does not depend on the order candidates arrive in.
//   4. kp_query_kernel   : pack-mode kNN (ops/knn.py:110-145) on the same cell list, cells sized by the cloud's density:
//                          a first box of 3x3x3 (k <= 3: 2x2x2) cells, then one more layer of cells at a time until the
//                          k-th distance lies inside the explored box (v3d_knn_pack_f32 in query.hip sends clouds of
//                          >= 2048 supports here).
// Steps 1-2 can be kept by the caller and searched several times (v3d_radius_grid_build_f32 / v3d_radius_search_grid_f32:
// the pyramid builder searches the points of a stage three times at one radius).
// Distances use the reference expression order (dx*dx + dy*dy) + dz*dz with one rounding per operation; the radius
// test on sqrt(d2) is made on d2 against a threshold found on the host (rs_sq_threshold).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.h"
#include "wave_sort.h"

namespace v3d {

constexpr int kRsThreads = 256;
constexpr int kRsMaxDim = 1024;
constexpr int kRsMaxClouds = kKnnCellListMaxClouds;  // (4096) plan rows are searched linearly / by bisection per query

struct RsPlan {  // one per cloud
  float ox, oy, oz, inv_h;
  int dx, dy, dz, cell_off;
  long long s_off, q_off, s_len, q_len;
};

static int64_t rs_cell_capacity(int B, int64_t total_s) { return 4096ll * B + 4ll * total_s + 1; }

struct RsWorkspace {
  RsPlan* plan;     // [B]
  int* bbox;        // [B][6], order-preserving int encoding of the corners
  int* cell_count;  // [cap]
  int* cell_start;  // [cap + 1]
  float4* sorted;   // [total_s]
  void* scan_tmp;   // rocprim exclusive_scan over the cell counters
  size_t scan_bytes;
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
  w.bbox = static_cast<int*>(take(sizeof(int) * 6 * (size_t)B));
  w.cell_count = static_cast<int*>(take(sizeof(int) * cap));
  w.cell_start = static_cast<int*>(take(sizeof(int) * (cap + 1)));
  w.sorted = static_cast<float4*>(take(sizeof(float4) * (size_t)std::max<int64_t>(total_s, 1)));
  w.scan_bytes = 0;
  int* ni = nullptr;
  if (rocprim::exclusive_scan(nullptr, w.scan_bytes, ni, ni, 0, cap, rocprim::plus<int>(), nullptr) != hipSuccess)
    w.scan_bytes = 0;  // reported by the scan itself
  w.scan_tmp = take(std::max<size_t>(w.scan_bytes, 256));
  w.bytes = off;
  return w;
}

// Corner of a cloud's bounding box as an int whose signed order is the float order (NaN never gets here: fminf / fmaxf
// drop it), so that several workgroups per cloud can combine their corners with integer atomics.
__device__ __forceinline__ int rs_ordered(float f) {
  const int b = __float_as_int(f);
  return b >= 0 ? b : b ^ 0x7fffffff;
}
__device__ __forceinline__ float rs_unordered(int o) { return __int_as_float(o >= 0 ? o : o ^ 0x7fffffff); }

constexpr int kRsBboxSplit = 16;  // workgroups per cloud

__global__ void rs_bbox_init_kernel(int* __restrict__ bbox, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * 6) bbox[i] = rs_ordered(i % 6 < 3 ? INFINITY : -INFINITY);
}

// kRsBboxSplit workgroups per cloud: min / max corner of its supports.
// (lengths that overrun total_s -- device data nobody validated -- are cut like in rs_plan_kernel: no read leaves s)
__device__ __forceinline__ long long rs_cut(long long len, long long off, long long total) {
  return len < 0 ? 0 : (len > total - off ? total - off : len);
}
__global__ __launch_bounds__(kRsThreads) void rs_bbox_kernel(const float* __restrict__ s, const int64_t* __restrict__ s_len,
                                                             int* __restrict__ bbox, long long total_s) {
  __shared__ float red[6][kRsThreads / 64];
  const int b = blockIdx.x, tid = threadIdx.x;
  long long off = 0;
  for (int i = 0; i < b; ++i) off += rs_cut(s_len[i], off, total_s);
  const long long n = rs_cut(s_len[b], off, total_s);
  if ((long long)blockIdx.y * kRsThreads >= n) return;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (long long i = (long long)blockIdx.y * kRsThreads + tid; i < n; i += (long long)kRsThreads * kRsBboxSplit) {
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
    if (tid < 3) atomicMin(&bbox[b * 6 + tid], rs_ordered(v));
    else atomicMax(&bbox[b * 6 + tid], rs_ordered(v));
  }
}

// Cell coordinate along one axis; the same expression for supports, grid dimensions and queries.
__device__ __forceinline__ int rs_cell_coord(float p, float o, float inv_h) {
  const float c = floorf((p - o) * inv_h);
  return (int)fminf(fmaxf(c, -2.0f), (float)(kRsMaxDim + 2));
}

// One wave: grids, cell offsets and point offsets of all clouds, 64 clouds at a time (a cloud per lane; the offsets are
// wave scans with a carry).  Lengths that overrun the totals -- device data nobody validated -- are cut so that no
// access leaves the arrays: cloud b starts at min(sum of the lengths before it, total) and ends at most at total.
// per_cell > 0 (kNN): the cell edge comes from the cloud's density instead of a radius -- `per_cell` supports per cell of
// a cloud spread evenly over its bounding box (axes without extent do not count: a flat cloud gets square cells).
__device__ __forceinline__ long long rs_wave_inclusive(long long v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const long long t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}
__global__ __launch_bounds__(64) void rs_plan_kernel(const int* __restrict__ bbox, const int64_t* __restrict__ q_len,
                                                     const int64_t* __restrict__ s_len, RsPlan* __restrict__ plan,
                                                     int B, float radius, long long total_s, long long total_q,
                                                     float per_cell) {
  if (blockIdx.x != 0) return;
  const int lane = threadIdx.x;
  long long s_carry = 0, q_carry = 0, cell_carry = 0;  // raw (uncut) sums of the lengths / cells before this chunk
  for (int base = 0; base < B; base += 64) {
    const int b = base + lane;
    const bool live = b < B;
    long long sl = live ? (long long)s_len[b] : 0, ql = (live && q_len) ? (long long)q_len[b] : 0;
    sl = sl < 0 ? 0 : (sl > total_s ? total_s : sl);  // (also keeps the raw sums far from overflow)
    ql = ql < 0 ? 0 : (ql > total_q ? total_q : ql);
    const long long s_end = s_carry + rs_wave_inclusive(sl, lane), q_end = q_carry + rs_wave_inclusive(ql, lane);
    RsPlan p;
    p.s_off = s_end - sl < total_s ? s_end - sl : total_s;
    p.q_off = q_end - ql < total_q ? q_end - ql : total_q;
    p.s_len = (s_end < total_s ? s_end : total_s) - p.s_off;
    p.q_len = (q_end < total_q ? q_end : total_q) - p.q_off;
    p.ox = p.oy = p.oz = 0.f;
    p.inv_h = 1.f;
    p.dx = p.dy = p.dz = 1;
    if (live && p.s_len > 0) {
      float bb[6];
      for (int a = 0; a < 6; ++a) bb[a] = rs_unordered(bbox[b * 6 + a]);
      p.ox = bb[0];
      p.oy = bb[1];
      p.oz = bb[2];
      const long long cap = p.s_len * 4 > 4096 ? p.s_len * 4 : 4096;
      float h = fmaxf(radius, 1e-30f) * 1.001f;
      if (per_cell > 0.f) {
        float vol = 1.f;
        int axes = 0;
        for (int a = 0; a < 3; ++a) {
          const float ext = bb[3 + a] - bb[a];
          if (ext > 0.f && ext < INFINITY) {
            vol *= ext;
            ++axes;
          }
        }
        const float per = vol * per_cell / (float)p.s_len;  // volume (area, length) of a cell
        h = axes == 3 ? cbrtf(per) : (axes == 2 ? sqrtf(per) : (axes == 1 ? per : 1.f));
        h = (h > 1e-30f && h < INFINITY) ? h : 1.f;  // (overflowing products fall through to the enlarging loop / one cell)
      }
      for (int it = 0; it < 400; ++it) {
        p.inv_h = 1.0f / h;
        p.dx = rs_cell_coord(bb[3], p.ox, p.inv_h) + 1;
        p.dy = rs_cell_coord(bb[4], p.oy, p.inv_h) + 1;
        p.dz = rs_cell_coord(bb[5], p.oz, p.inv_h) + 1;
        if (p.dx <= kRsMaxDim && p.dy <= kRsMaxDim && p.dz <= kRsMaxDim && (long long)p.dx * p.dy * p.dz <= cap) break;
        h *= 1.25f;
      }
      // A cloud whose bounding box is not finite (infinite coordinates; all-NaN supports leave it at +inf / -inf) has no
      // usable grid: the loop above then ends with dimensions outside [1, kRsMaxDim] or a cell count beyond the
      // workspace.  Such a cloud gets ONE cell (inv_h = 0: every finite coordinate maps to cell 0, non-finite ones are
      // clamped into it), i.e. every support is a candidate of every query -- correct, merely brute force.
      bool finite = true;
      for (int a = 0; a < 6; ++a) finite = finite && (fabsf(bb[a]) <= 3.0e38f);  // false for inf and NaN
      if (!finite || p.dx < 1 || p.dy < 1 || p.dz < 1 || p.dx > kRsMaxDim || p.dy > kRsMaxDim || p.dz > kRsMaxDim ||
          (long long)p.dx * p.dy * p.dz > cap) {
        p.ox = p.oy = p.oz = 0.f;
        p.inv_h = 0.f;
        p.dx = p.dy = p.dz = 1;
      }
    }
    const long long cells = live ? (long long)p.dx * p.dy * p.dz : 0;
    const long long cell_end = cell_carry + rs_wave_inclusive(cells, lane);
    p.cell_off = (int)(cell_end - cells);  // < 2^31: rs_run refuses sizes whose cell capacity does not fit
    if (live) plan[b] = p;
    s_carry = __shfl(s_end, 63, 64);
    q_carry = __shfl(q_end, 63, 64);
    cell_carry = __shfl(cell_end, 63, 64);
  }
}

// A grid built once (v3d_radius_grid_build_f32) and searched by several query sets: the query offsets of the plan are
// rewritten per search (one wave; the support side of the plan is left alone).
__global__ __launch_bounds__(64) void rs_qplan_kernel(const int64_t* __restrict__ q_len, RsPlan* __restrict__ plan, int B,
                                                      long long total_q) {
  if (blockIdx.x != 0) return;
  const int lane = threadIdx.x;
  long long q_carry = 0;
  for (int base = 0; base < B; base += 64) {
    const int b = base + lane;
    const bool live = b < B;
    long long ql = live ? (long long)q_len[b] : 0;
    ql = ql < 0 ? 0 : (ql > total_q ? total_q : ql);
    const long long q_end = q_carry + rs_wave_inclusive(ql, lane);
    if (live) {
      const long long off = q_end - ql < total_q ? q_end - ql : total_q;
      plan[b].q_off = off;
      plan[b].q_len = (q_end < total_q ? q_end : total_q) - off;
    }
    q_carry = __shfl(q_end, 63, 64);
  }
}

// The cloud of support `i`: the whole wave finds the cloud of ITS first support (lane l looks at cloud l * stride, then
// inside the stride: one or two loads instead of a bisection's log2(B) dependent ones), every lane then walks on while
// the next cloud starts at or before its own support -- supports are consecutive, so almost never.  Empty clouds share
// an offset and the last one wins.
__device__ __forceinline__ int rs_find_support_cloud(const RsPlan* __restrict__ plan, int B, long long i, int lane) {
  const long long first = __shfl(i, 0, 64);
  const int stride = (B + 63) >> 6;
  const int i1 = lane * stride;
  bool ok = i1 < B && plan[i1 < B ? i1 : 0].s_off <= first;
  int c = __builtin_popcountll(__ballot(ok)) - 1;  // >= 0: cloud 0 starts at 0
  if (stride > 1) {
    const int i2 = c * stride + lane;
    const bool in = lane < stride && i2 < B;
    ok = in && plan[in ? i2 : 0].s_off <= first;
    c = c * stride + __builtin_popcountll(__ballot(ok)) - 1;
  }
  while (c + 1 < B && plan[c + 1].s_off <= i) ++c;
  return c;
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
  const int c = rs_find_support_cloud(plan, B, i < total_s ? i : total_s - 1, threadIdx.x & 63);  // (whole waves)
  if (i >= total_s) return;
  const RsPlan p = plan[c];
  atomicAdd(&cell_count[rs_support_cell(p, s[i * 3], s[i * 3 + 1], s[i * 3 + 2])], 1);
}

// Cell starts = one exclusive scan over the counters of ALL clouds (rocprim): the cells of cloud b follow those of cloud
// b-1 and so do its supports, so the global running sum is the cloud's first support plus the sum within the cloud;
// counters past the last cell are zero, so every later entry (the end marker included) reads total_s.  The fill kernel
// counts the same counters back down to zero to hand out the slots of a cell (their order within a cell is free).
__global__ __launch_bounds__(kRsThreads) void rs_fill_kernel(const float* __restrict__ s, const RsPlan* __restrict__ plan,
                                                             const int* __restrict__ cell_start, int* __restrict__ cursor,
                                                             float4* __restrict__ sorted, int B, long long total_s) {
  const long long i = (long long)blockIdx.x * kRsThreads + threadIdx.x;
  const int cl = rs_find_support_cloud(plan, B, i < total_s ? i : total_s - 1, threadIdx.x & 63);  // (whole waves)
  if (i >= total_s) return;
  const RsPlan p = plan[cl];
  const float x = s[i * 3], y = s[i * 3 + 1], z = s[i * 3 + 2];
  const int c = rs_support_cell(p, x, y, z);
  const int slot = cell_start[c] + atomicSub(&cursor[c], 1) - 1;
  sorted[slot] = make_float4(x, y, z, __int_as_float((int)i));
}

// The cloud of query `qi` (wave-uniform), found by the whole wave at once: lane l looks at cloud l * stride, the number
// of lanes whose cloud starts at or before qi picks the stride, a second round picks the cloud inside it.  Two dependent
// loads at most (one for B <= 64) instead of a bisection's log2(B); empty clouds share an offset and the last one wins,
// as in rs_find_cloud.
__device__ __forceinline__ int rs_find_query_cloud(const RsPlan* __restrict__ plan, int B, long long qi, int lane) {
  const int stride = (B + 63) >> 6;
  const int i1 = lane * stride;
  bool ok = i1 < B && plan[i1 < B ? i1 : 0].q_off <= qi;
  int c = __builtin_popcountll(__ballot(ok)) - 1;  // >= 0: cloud 0 starts at 0
  if (stride > 1) {
    const int i2 = c * stride + lane;
    const bool in = lane < stride && i2 < B;
    ok = in && plan[in ? i2 : 0].q_off <= qi;
    c = c * stride + __builtin_popcountll(__ballot(ok)) - 1;
  }
  return __builtin_amdgcn_readfirstlane(c);
}

// The 27 cells around a query are 9 runs of consecutive records (cells are x-major).  Lanes 0..8 fetch the bounds of
// one run each (ONE memory round trip for all of them), the bounds move to scalar registers, and the wave then walks
// the concatenation of the runs 64 candidates at a time: candidate t lies in the first run whose end exceeds t.
struct RsRuns {
  int beg[9], pre[10], total;
  __device__ __forceinline__ void load(const RsPlan& p, const int* __restrict__ cell_start, float qx, float qy,
                                       float qz, int lane) {
    int b = 0, n = 0;
    if (p.s_len > 0 && lane < 9) {
      const int cx = rs_cell_coord(qx, p.ox, p.inv_h), cy = rs_cell_coord(qy, p.oy, p.inv_h),
                cz = rs_cell_coord(qz, p.oz, p.inv_h);
      const int x0 = max(cx - 1, 0), x1 = min(cx + 1, p.dx - 1);
      const int zz = cz - 1 + lane / 3, yy = cy - 1 + lane % 3;
      if (x0 <= x1 && zz >= 0 && zz < p.dz && yy >= 0 && yy < p.dy) {
        const int c0 = p.cell_off + (zz * p.dy + yy) * p.dx;
        b = cell_start[c0 + x0];
        n = cell_start[c0 + x1 + 1] - b;
      }
    }
    pre[0] = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      beg[k] = __builtin_amdgcn_readlane(b, k);
      pre[k + 1] = pre[k] + __builtin_amdgcn_readlane(n, k);
    }
    total = pre[9];
  }
  __device__ __forceinline__ int slot(int t) const {  // 0 <= t < total
    int j = beg[8] + (t - pre[8]);
#pragma unroll
    for (int k = 7; k >= 0; --k) j = t < pre[k + 1] ? beg[k] + (t - pre[k]) : j;
    return j;
  }
};

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
                                                              long long total_s, float radius_sq, int limit) {
  const int lane = threadIdx.x & 63;
  const long long qi = (long long)blockIdx.x * (kRsThreads / 64) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (qi >= total_q) return;
  const float qx = q[qi * 3], qy = q[qi * 3 + 1], qz = q[qi * 3 + 2];
  const RsPlan p = plan[rs_find_query_cloud(plan, B, qi, lane)];
  RsList<R> list;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    list.d[r] = INFINITY;
    list.i[r] = INT32_MAX;
  }
  float wd = INFINITY;  // current `limit`-th entry: a candidate must come before it
  int wi = INT32_MAX;
  RsRuns runs;
  runs.load(p, cell_start, qx, qy, qz, lane);
  for (int t0 = 0; t0 < runs.total; t0 += 64) {
    const int t = t0 + lane;
    const bool valid = t < runs.total;
    const float4 s = sorted[runs.slot(valid ? t : 0)];
    const float dx = qx - s.x, dy = qy - s.y, dz = qz - s.z;
    const float d2 = (dx * dx + dy * dy) + dz * dz;
    const int si = __float_as_int(s.w);
    const bool inside = valid && d2 <= radius_sq;  // radius_search.py:66 masks dist > radius (rs_sq_threshold)
    unsigned long long m = __ballot(inside && (d2 < wd || (d2 == wd && si < wi)));
    while (m) {
      const int l = __builtin_ffsll((long long)m) - 1;
      list.insert(rs_lane(d2, l), __builtin_amdgcn_readlane(si, l), lane);
      list.at(limit - 1, wd, wi);
      m = __ballot(inside && (d2 < wd || (d2 == wd && si < wi))) & ~((2ull << l) - 1ull);
    }
  }
  int64_t* o = out + qi * limit;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int rank = 64 * r + lane;
    if (rank < limit) o[rank] = list.i[r] == INT32_MAX ? (int64_t)total_s : (int64_t)list.i[r];
  }
}

// ---- limit <= 64: batch the hits, sort them with a bitonic network, merge into the kept list ---------------------------
// Serial insertion costs ~30 dependent instructions per hit and a realistic neighbourhood (mean count ~ limit) has
// dozens of hits per query.  Here the hits that beat the current limit-th entry are compacted into a 64-slot per-wave
// LDS buffer as 64-bit keys (d2 bits << 32 | index: d2 >= +0 and finite, so the unsigned order is the (d2, index)
// order); a full buffer (and the last, partial one) is sorted DESCENDING across the lanes (21 compare-exchange steps),
// min-ed lane by lane against the kept ASCENDING list -- the 64 smallest of the union, as a bitonic sequence -- and
// re-sorted with 6 merge steps (wave_sort.h); a buffer of at most kWaveSerialKeys keys is inserted key by key instead (wave shift).
// The result is the `limit` smallest keys, whatever order the candidates arrive in.
constexpr wkey_t kRsSentKey = kWaveKeyMax;

__global__ __launch_bounds__(kRsThreads) void rs_query_sort_kernel(const float* __restrict__ q,
                                                                   const RsPlan* __restrict__ plan,
                                                                   const int* __restrict__ cell_start,
                                                                   const float4* __restrict__ sorted,
                                                                   int64_t* __restrict__ out, int B, long long total_q,
                                                                   long long total_s, float radius_sq, int limit) {
  __shared__ unsigned long long hits[kRsThreads / 64][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long qi = (long long)blockIdx.x * (kRsThreads / 64) + wave;
  if (qi >= total_q) return;
  const float qx = q[qi * 3], qy = q[qi * 3 + 1], qz = q[qi * 3 + 2];
  const RsPlan p = plan[rs_find_query_cloud(plan, B, qi, lane)];
  unsigned long long* buf = hits[wave];
  unsigned long long kept = kRsSentKey;   // ascending across the lanes
  unsigned long long worst = kRsSentKey;  // kept[limit - 1]: a candidate must come before it
  int pending = 0;                        // keys in buf
  bool empty = true;  // nothing kept yet: the first merge is a plain sort
  auto flush = [&](int n) {
    kept = wave_merge_buffer(kept, buf, n, lane, empty);
    empty = false;
    worst = wave_lane64(kept, limit - 1);
  };
  RsRuns runs;
  runs.load(p, cell_start, qx, qy, qz, lane);
  for (int t0 = 0; t0 < runs.total; t0 += 64) {
    const int t = t0 + lane;
    const bool valid = t < runs.total;
    const float4 s = sorted[runs.slot(valid ? t : 0)];
    const float dx = qx - s.x, dy = qy - s.y, dz = qz - s.z;
    const float d2 = (dx * dx + dy * dy) + dz * dz;
    const unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)__float_as_int(s.w);
    // radius_search.py:66 masks dist > radius; a NaN distance never enters the list
    const bool take = valid && d2 <= radius_sq && key < worst;  // (rs_sq_threshold: finite; NaN fails)
    const unsigned long long m = __ballot(take);
    if (m == 0) continue;
    const int n = __builtin_popcountll(m);
    const int dest = pending + wave_prefix_count(m);
    if (take && dest < 64) buf[dest] = key;
    if (pending + n >= 64) {
      flush(64);
      if (take && dest >= 64) buf[dest - 64] = key;
      pending += n - 64;
    } else {
      pending += n;
    }
  }
  if (pending > 0) flush(pending);
  if (lane < limit) out[qi * limit + lane] = kept == kRsSentKey ? (int64_t)total_s : (int64_t)(unsigned)kept;
}

// Number of supports strictly inside `radius` of every query (ops/radius_search.py:74-107: relu(radius - d).sign().sum()
// over the query's own cloud), on the same cell list: a wave per query, 64 candidates per step, one popcount.
__global__ __launch_bounds__(kRsThreads) void rs_count_query_kernel(const float* __restrict__ q,
                                                                    const RsPlan* __restrict__ plan,
                                                                    const int* __restrict__ cell_start,
                                                                    const float4* __restrict__ sorted,
                                                                    int64_t* __restrict__ counts, int B, long long total_q,
                                                                    float radius_sq) {
  const int lane = threadIdx.x & 63;
  const long long qi = (long long)blockIdx.x * (kRsThreads / 64) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (qi >= total_q) return;
  const float qx = q[qi * 3], qy = q[qi * 3 + 1], qz = q[qi * 3 + 2];
  const RsPlan p = plan[rs_find_query_cloud(plan, B, qi, lane)];
  RsRuns runs;
  runs.load(p, cell_start, qx, qy, qz, lane);
  int cnt = 0;
  for (int t0 = 0; t0 < runs.total; t0 += 64) {
    const int t = t0 + lane;
    const bool valid = t < runs.total;
    const float4 s = sorted[runs.slot(valid ? t : 0)];
    const float dx = qx - s.x, dy = qy - s.y, dz = qz - s.z;
    const float d2 = (dx * dx + dy * dy) + dz * dz;
    cnt += __builtin_popcountll(__ballot(valid && d2 <= radius_sq));  // relu(radius - d).sign() (rs_sq_threshold, strict)
  }
  if (lane == 0) counts[qi] = (int64_t)cnt;
}


// ---- pack-mode kNN on the same cell list (ops/knn.py:110-145; launched by knn_pack_cell_list below) -------------------
// k nearest supports of the query's own cloud, ascending by (d2, support index) -- the order pykeops' Kmin_argKmin leaves
// (strict-< insertion while scanning the supports in index order), which is also the radius search's.  One WAVE per
// query.  The 27 cells around the query are 9 runs of consecutive records; a first walk keeps every lane's minimum
// distance (the first 64 records of every run stay in registers), the k-th smallest of the 64 minima bounds the k-th
// distance from above, and only candidates under it are batched as 64-bit keys and sorted (wave_rank_sort / the bitonic
// merge of wave_sort.h).  The search is exact: the k-th distance is compared with the distance to the nearest face of
// the explored box that still has cells behind it (minus a rounding margin) and the box grows ring by ring -- the
// bounds of a ring's rows fetched by the lanes in one round trip -- until it is below; a query that is still not done
// after kKpMaxRing rings (far from its cloud, or a cloud of distant clusters) scans its cloud's records linearly.
constexpr int kKpMaxRing = 6;
static float kp_supports_per_cell(int k) { return std::max(12.0f, 0.36f * (float)k); }

__device__ __forceinline__ float kp_sq_dist(float qx, float qy, float qz, float sx, float sy, float sz) {
  const float dx = qx - sx, dy = qy - sy, dz = qz - sz;
  return (dx * dx + dy * dy) + dz * dz;  // reference order, one rounding per operation (-ffp-contract=off)
}
// (k+1)-th smallest (k = 0..63) of the lanes' non-negative values, rounded up at bit 17 of the pattern (<= 1.6 % above;
// 14 ballots); +inf when fewer than k+1 lanes hold a finite value.  (query.hip's kth_smallest_upper.)
__device__ __forceinline__ float kp_kth_upper(float v, int k) {
  const unsigned bits = __float_as_uint(v);
  unsigned prefix = 0;
  for (int b = 30; b >= 17; --b) {
    const unsigned trial = prefix | (1u << b);
    if (__builtin_popcountll(__ballot(bits < trial)) <= k) prefix = trial;
  }
  return __uint_as_float(prefix | 0x1ffffu);
}

__device__ __forceinline__ wkey_t kp_wave_min64(wkey_t v) {  // minimum over the lanes, in every lane
#pragma unroll
  for (int j = 1; j < 64; j <<= 1) {
    const wkey_t o = wave_xor64(v, j);
    v = o < v ? o : v;
  }
  return v;
}

// NS = cells per axis of the first box: 3 (the cell of the query and its neighbours, 9 runs of 3 cells) or, for k <= 3,
// 2 (per axis the query's cell and the neighbour on the side of the query: the 8 cells nearest to it, 4 runs of 2 cells
// -- at >= 12 supports per cell the third neighbour of a uniform cloud lies within 0.4 cell edges, the nearest face of
// that box at >= 0.5).
template <int NS>
__global__ __launch_bounds__(kRsThreads) void kp_query_kernel(const float* __restrict__ q, const RsPlan* __restrict__ plan,
                                                              const int* __restrict__ cell_start,
                                                              const float4* __restrict__ sorted,
                                                              float* __restrict__ out_d, int64_t* __restrict__ out_i,
                                                              int B, long long total_q, int k) {
  constexpr int NR = NS * NS;
  __shared__ wkey_t hits[kRsThreads / 64][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long qi = (long long)blockIdx.x * (kRsThreads / 64) + wave;
  if (qi >= total_q) return;
  const float qx = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(q[qi * 3])));
  const float qy = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(q[qi * 3 + 1])));
  const float qz = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(q[qi * 3 + 2])));
  const RsPlan p = plan[rs_find_query_cloud(plan, B, qi, lane)];
  const float sentinel = 1e10f;  // as v3d_knn_points_f32: nothing at or beyond 1e10 enters; unfilled slots read (1e5, 0)
  const wkey_t sent_key = (wkey_t)__float_as_uint(sentinel) << 32;
  wkey_t* buf = hits[wave];
  wkey_t kept = sent_key, worst = sent_key;
  int pending = 0;
  bool kept_empty = true;
  auto offer_key = [&](wkey_t key, bool take) {
    const unsigned long long m = __ballot(take);
    if (m == 0ull) return;
    const int n = __builtin_popcountll(m);
    const int dest = pending + wave_prefix_count(m);
    if (take && dest < 64) buf[dest] = key;
    if (pending + n >= 64) {
      kept = wave_merge_buffer(kept, buf, 64, lane);
      kept_empty = false;
      worst = wave_lane64(kept, k - 1);
      if (take && dest >= 64) buf[dest - 64] = key;
      pending += n - 64;
    } else {
      pending += n;
    }
  };
  auto offer = [&](int j, bool valid, float cap) {  // record j (lane-varying) of the sorted list
    const float4 s = sorted[valid ? j : 0];
    const float d = kp_sq_dist(qx, qy, qz, s.x, s.y, s.z);
    const wkey_t key = ((wkey_t)__float_as_uint(d) << 32) | (unsigned)__float_as_int(s.w);
    offer_key(key, valid && key < worst && d <= cap);  // NaN / +inf patterns lie above the sentinel's
  };
  auto flush = [&]() {
    if (pending > 0) {
      if (kept_empty) {
        const wkey_t h = wave_rank_sort(buf, pending, lane);
        kept = h < sent_key ? h : sent_key;
      } else {
        kept = wave_merge_buffer(kept, buf, pending, lane);
      }
      kept_empty = false;
      worst = wave_lane64(kept, k - 1);
      pending = 0;
    }
  };
  if (p.s_len > 0) {
    const int dim[3] = {p.dx, p.dy, p.dz};
    const float org[3] = {p.ox, p.oy, p.oz}, qv[3] = {qx, qy, qz};
    int p0[3], p1[3];  // the explored box of cells
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float t = (qv[a] - org[a]) * p.inv_h;  // rs_cell_coord's expression
      const int cc = min(max(rs_cell_coord(qv[a], org[a], p.inv_h), 0), dim[a] - 1);
      int lo;
      if (NS == 3) {
        lo = max(cc - 1, 0);
        p1[a] = min(cc + 1, dim[a] - 1);
      } else {
        lo = max(min(cc - ((t - floorf(t)) < 0.5f ? 1 : 0), dim[a] - 2), 0);  // (which side only matters for speed)
        p1[a] = min(lo + 1, dim[a] - 1);
      }
      p0[a] = __builtin_amdgcn_readfirstlane(lo);
      p1[a] = __builtin_amdgcn_readfirstlane(p1[a]);
    }
    {  // ---- the first box: NR runs of consecutive records
      int rb = 0, rn = 0;
      if (lane < NR) {
        const int zz = p0[2] + lane / NS, yy = p0[1] + lane % NS;
        if (zz <= p1[2] && yy <= p1[1]) {
          const int c0 = p.cell_off + (zz * p.dy + yy) * p.dx;
          rb = cell_start[c0 + p0[0]];
          rn = cell_start[c0 + p1[0] + 1] - rb;
        }
      }
      int beg[NR], len[NR], pre = 0;
      float dr[NR];
      unsigned wr[NR];
      float lmin = INFINITY;
      bool long_run = false;
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        beg[r] = __builtin_amdgcn_readlane(rb, r);
        len[r] = __builtin_amdgcn_readlane(rn, r);
      }
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        // run r is dealt to the lanes from lane (records before it) mod 64 on, as if the runs were concatenated
        const int jr = (lane - pre) & 63;
        const bool valid = jr < len[r];
        const float4 s = sorted[valid ? beg[r] + jr : 0];
        const float d = kp_sq_dist(qx, qy, qz, s.x, s.y, s.z);
        dr[r] = valid ? d : __uint_as_float(0x7fc00000u);  // an idle lane's entry never passes a comparison
        wr[r] = (unsigned)__float_as_int(s.w);
        lmin = (valid && d < lmin) ? d : lmin;  // +inf and NaN never replace a minimum
        for (int j = 64; j < len[r]; j += 64) {  // (a crowded run: more than 64 records)
          const int jj = j + jr;
          const float4 s2 = sorted[jj < len[r] ? beg[r] + jj : 0];
          const float d2 = kp_sq_dist(qx, qy, qz, s2.x, s2.y, s2.z);
          lmin = (jj < len[r] && d2 < lmin) ? d2 : lmin;
          long_run = true;
        }
        pre += len[r];
      }
      if (k == 1 && !long_run) {
        // the nearest one: every lane's smallest key, then the wave's -- no bound, no batch, no sort
        wkey_t best = kWaveKeyMax;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const wkey_t key = ((wkey_t)__float_as_uint(dr[r]) << 32) | wr[r];
          best = key < best ? key : best;
        }
        best = kp_wave_min64(best);
        worst = best < sent_key ? best : sent_key;
        kept = lane == 0 ? worst : sent_key;
        kept_empty = false;
      } else {
        const float capk = pre > 64 ? kp_kth_upper(lmin, k - 1) : INFINITY;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const wkey_t key = ((wkey_t)__float_as_uint(dr[r]) << 32) | wr[r];
          offer_key(key, key < worst && dr[r] <= capk);
        }
        if (long_run) {
          int pre2 = 0;
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            const int jr = (lane - pre2) & 63;
            for (int j = 64; j < len[r]; j += 64) offer(beg[r] + j + jr, j + jr < len[r], capk);
            pre2 += len[r];
          }
        }
      }
    }
    // ---- one more layer of cells around the box at a time, until the k-th distance lies inside it
    const float hc = 1.0f / p.inv_h;  // (inv_h = 0: the one-cell grid of a cloud without a finite box; never `open`)
    for (int ring = 1;; ++ring) {
      flush();  // the bound below needs the true k-th key
      float bound = INFINITY;
      bool open = false;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float rel = qv[a] - org[a];
        const float margin = 1e-5f * fmaxf((float)dim[a] * hc, fabsf(rel));
        if (p0[a] > 0) {
          open = true;
          bound = fminf(bound, fmaxf(rel - (float)p0[a] * hc - margin, 0.f));
        }
        if (p1[a] < dim[a] - 1) {
          open = true;
          bound = fminf(bound, fmaxf((float)(p1[a] + 1) * hc - rel - margin, 0.f));
        }
      }
      const float dk = __uint_as_float((unsigned)(worst >> 32));
      if (!open || dk < bound * bound) break;  // (NaN bounds compare false: the search goes on to the whole cloud)
      if (ring > kKpMaxRing) {
        // start over on the whole cloud, admitting only what is at most as far as the k-th found so far
        kept = worst = sent_key;
        kept_empty = true;
        for (long long j0 = p.s_off; j0 < p.s_off + p.s_len; j0 += 64)
          offer((int)(j0 + lane), j0 + lane < p.s_off + p.s_len, dk);
        flush();
        break;
      }
      int r0[3], r1[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        r0[a] = max(p0[a] - 1, 0);
        r1[a] = min(p1[a] + 1, dim[a] - 1);
      }
      // rows (z, y) of the grown box, 64 at a time: lane l fetches the bounds of row l -- the whole row outside the
      // explored box's (z, y) range, else the part left and the part right of it
      const int ny = r1[1] - r0[1] + 1, rows = (r1[2] - r0[2] + 1) * ny;
      for (int base = 0; base < rows; base += 64) {
        const int row = base + lane;
        int b0 = 0, n0 = 0, b1 = 0, n1 = 0;
        if (row < rows) {
          const int z = r0[2] + row / ny, y = r0[1] + row % ny;
          const int c0 = p.cell_off + (z * p.dy + y) * p.dx;
          const bool inner = z >= p0[2] && z <= p1[2] && y >= p0[1] && y <= p1[1];
          if (!inner) {
            b0 = cell_start[c0 + r0[0]];
            n0 = cell_start[c0 + r1[0] + 1] - b0;
          } else {
            if (r0[0] < p0[0]) {
              b0 = cell_start[c0 + r0[0]];
              n0 = cell_start[c0 + p0[0]] - b0;
            }
            if (r1[0] > p1[0]) {
              b1 = cell_start[c0 + p1[0] + 1];
              n1 = cell_start[c0 + r1[0] + 1] - b1;
            }
          }
        }
        unsigned long long busy = __ballot(n0 > 0 || n1 > 0);
        while (busy) {
          const int l = __builtin_ffsll((long long)busy) - 1;
          busy &= busy - 1ull;
          const int sb0 = __builtin_amdgcn_readlane(b0, l), sn0 = __builtin_amdgcn_readlane(n0, l);
          const int sb1 = __builtin_amdgcn_readlane(b1, l), sn1 = __builtin_amdgcn_readlane(n1, l);
          for (int j = 0; j < sn0; j += 64) offer(sb0 + j + lane, j + lane < sn0, INFINITY);
          for (int j = 0; j < sn1; j += 64) offer(sb1 + j + lane, j + lane < sn1, INFINITY);
        }
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        p0[a] = r0[a];
        p1[a] = r1[a];
      }
    }
  }
  if (lane < k) {
    const bool real = kept < sent_key;
    out_d[qi * k + lane] = sqrtf(__uint_as_float((unsigned)(kept >> 32)));
    out_i[qi * k + lane] = real ? (int64_t)(unsigned)kept - (int64_t)p.s_off : 0;
  }
}

}  // namespace v3d

extern "C" {

size_t v3d_radius_search_workspace_bytes(int B, int64_t total_s) {
  if (B <= 0 || total_s < 0) return 0;
  return v3d::rs_carve(nullptr, B, total_s).bytes;
}

}  // extern "C"

namespace v3d {
// The reference masks on the ROOT of the squared distance (`dist > radius`, radius_search.py:66; the count on
// `radius - dist > 0`, :104).  sqrtf is correctly rounded, hence monotone: NOT(sqrtf(d2) > radius) holds exactly for the
// d2 up to the largest float whose root is <= radius (strict: < radius) -- found here once, on the host, so that the
// kernels compare squared distances and skip a correctly rounded root (~15 instructions) per candidate.  Returns -1 when
// no d2 >= 0 qualifies; never +inf (an infinite d2 fails the reference's test too: it is beyond every real radius, and
// against radius = inf the list drops it as it drops NaN).
static float rs_sq_threshold(float radius, bool strict) {
  auto ok = [&](float v) {
    const float root = sqrtf(v);
    return strict ? root < radius : root <= radius;
  };
  float t = radius * radius;
  if (!(t <= FLT_MAX)) t = FLT_MAX;  // overflow (or radius = inf)
  while (t > 0.f && !ok(t)) t = nextafterf(t, -INFINITY);
  if (!ok(t)) return -1.f;
  while (t < FLT_MAX && ok(nextafterf(t, INFINITY))) t = nextafterf(t, INFINITY);
  return t;
}

// neighbor_limit < 0: count mode (`out` = counts (total_q,)), else search mode (`out` = neighbor_indices);
// knn_distances != nullptr: kNN mode (neighbor_limit = k <= 64, radius unused, cells sized by density)
// phase: the whole call, or one half of it around a caller-held grid (`cell_radius` = the radius the grid was built for:
// its cells serve every search radius up to it)
enum RsPhase { kRsWhole = 0, kRsBuildOnly = 1, kRsQueryOnly = 2 };
static int rs_run(const char* what, const float* q_points, const float* s_points, const int64_t* q_lengths,
                  const int64_t* s_lengths, int64_t* neighbor_indices, int B, int64_t total_q, int64_t total_s,
                  float radius, int neighbor_limit, void* workspace, size_t workspace_bytes, int device, void* stream_,
                  float* knn_distances = nullptr, RsPhase phase = kRsWhole, float cell_radius = 0.f) {
  const bool count_mode = neighbor_limit < 0;
  const float per_cell = knn_distances ? kp_supports_per_cell(neighbor_limit) : 0.f;
  const bool build = phase != kRsQueryOnly, query = phase != kRsBuildOnly;
  V3D_REQUIRE(B >= 0 && total_q >= 0 && total_s >= 0, what, "negative dimension");
  if (query && (total_q == 0 || neighbor_limit == 0)) return 0;
  V3D_REQUIRE(B > 0, what, "points without clouds");
  V3D_REQUIRE((!query || (q_points && q_lengths && neighbor_indices)) &&
                  (!build || (s_lengths && (total_s == 0 || s_points))),
              what, "null pointer");
  V3D_REQUIRE(radius >= 0.f, what, "negative radius");  // NaN fails too
  if (phase != kRsWhole) {
    V3D_REQUIRE(workspace != nullptr, what, "a grid is held by the caller: NULL workspace");
    V3D_REQUIRE(cell_radius >= 0.f && radius <= cell_radius, what, "search radius beyond the radius the grid was built for");
  }
  if (neighbor_limit > V3D_RADIUS_MAX_LIMIT) {
    set_last_error("%s: neighbor_limit=%d exceeds %d", what, neighbor_limit, V3D_RADIUS_MAX_LIMIT);
    return V3D_ERR_K_TOO_LARGE;
  }
  if (B > kRsMaxClouds || rs_cell_capacity(B, total_s) >= (int64_t)INT32_MAX ||
      ceil_div(total_q, kRsThreads / 64) > (int64_t)INT32_MAX) {
    set_last_error("%s: at most %d clouds, ~2^29 supports (4096 B + 4 total_s < 2^31 cells) and 2^33 queries per call",
                   what, kRsMaxClouds);
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
    hipError_t e = hipSuccess;
    if (!build) {
      hipLaunchKernelGGL(rs_qplan_kernel, dim3(1), dim3(64), 0, stream, q_lengths, ws.plan, B, (long long)total_q);
    } else {
    e = hipMemsetAsync(ws.cell_count, 0, sizeof(int) * (size_t)rs_cell_capacity(B, total_s), stream);
    if (e != hipSuccess) {
      set_last_error("%s: hipMemsetAsync: %s", what, hipGetErrorString(e));
      rc = (int)e;
      break;
    }
    hipLaunchKernelGGL(rs_bbox_init_kernel, dim3((unsigned)ceil_div((int64_t)B * 6, 256)), dim3(256), 0, stream, ws.bbox,
                       B);
    hipLaunchKernelGGL(rs_bbox_kernel, dim3(B, kRsBboxSplit), dim3(kRsThreads), 0, stream, s_points, s_lengths, ws.bbox,
                       (long long)total_s);
    hipLaunchKernelGGL(rs_plan_kernel, dim3(1), dim3(64), 0, stream, ws.bbox, query ? q_lengths : nullptr, s_lengths,
                       ws.plan, B, phase == kRsWhole ? radius : cell_radius, (long long)total_s, (long long)total_q,
                       per_cell);
    const size_t cap = (size_t)rs_cell_capacity(B, total_s);
    if (total_s > 0) {
      const unsigned sb = (unsigned)ceil_div(total_s, kRsThreads);
      hipLaunchKernelGGL(rs_count_kernel, dim3(sb), dim3(kRsThreads), 0, stream, s_points, ws.plan, ws.cell_count, B,
                         (long long)total_s);
    }
    size_t scan_bytes = ws.scan_bytes;
    e = rocprim::exclusive_scan(ws.scan_tmp, scan_bytes, ws.cell_count, ws.cell_start, 0, cap, rocprim::plus<int>(),
                                stream);
    if (e != hipSuccess) {
      set_last_error("%s: exclusive_scan: %s", what, hipGetErrorString(e));
      rc = (int)e;
      break;
    }
    if (total_s > 0) {
      const unsigned sb = (unsigned)ceil_div(total_s, kRsThreads);
      hipLaunchKernelGGL(rs_fill_kernel, dim3(sb), dim3(kRsThreads), 0, stream, s_points, ws.plan, ws.cell_start,
                         ws.cell_count, ws.sorted, B, (long long)total_s);
    }
    }  // build
    if (!query) {
      rc = launch_status(what);
      break;
    }
    const unsigned qb = (unsigned)ceil_div(total_q, kRsThreads / 64);
    if (knn_distances && neighbor_limit <= 3)
      hipLaunchKernelGGL(kp_query_kernel<2>, dim3(qb), dim3(kRsThreads), 0, stream, q_points, ws.plan, ws.cell_start,
                         ws.sorted, knn_distances, neighbor_indices, B, (long long)total_q, neighbor_limit);
    else if (knn_distances)
      hipLaunchKernelGGL(kp_query_kernel<3>, dim3(qb), dim3(kRsThreads), 0, stream, q_points, ws.plan, ws.cell_start,
                         ws.sorted, knn_distances, neighbor_indices, B, (long long)total_q, neighbor_limit);
    else if (count_mode)
      hipLaunchKernelGGL(rs_count_query_kernel, dim3(qb), dim3(kRsThreads), 0, stream, q_points, ws.plan, ws.cell_start,
                         ws.sorted, neighbor_indices, B, (long long)total_q, rs_sq_threshold(radius, true));
    else if (neighbor_limit <= 64)
      hipLaunchKernelGGL(rs_query_sort_kernel, dim3(qb), dim3(kRsThreads), 0, stream, q_points, ws.plan, ws.cell_start,
                         ws.sorted, neighbor_indices, B, (long long)total_q, (long long)total_s,
                         rs_sq_threshold(radius, false), neighbor_limit);
    else if (neighbor_limit <= 128)
      hipLaunchKernelGGL(rs_query_kernel<2>, dim3(qb), dim3(kRsThreads), 0, stream, q_points, ws.plan, ws.cell_start,
                         ws.sorted, neighbor_indices, B, (long long)total_q, (long long)total_s,
                         rs_sq_threshold(radius, false), neighbor_limit);
    else if (neighbor_limit <= 256)
      hipLaunchKernelGGL(rs_query_kernel<4>, dim3(qb), dim3(kRsThreads), 0, stream, q_points, ws.plan, ws.cell_start,
                         ws.sorted, neighbor_indices, B, (long long)total_q, (long long)total_s,
                         rs_sq_threshold(radius, false), neighbor_limit);
    else
      hipLaunchKernelGGL(rs_query_kernel<8>, dim3(qb), dim3(kRsThreads), 0, stream, q_points, ws.plan, ws.cell_start,
                         ws.sorted, neighbor_indices, B, (long long)total_q, (long long)total_s,
                         rs_sq_threshold(radius, false), neighbor_limit);
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
// Pack-mode kNN on the cell list (declared in common.h, called by query.hip's knn_pack): k <= 64.
int knn_pack_cell_list(const float* q_points, const float* s_points, const int64_t* q_lengths, const int64_t* s_lengths,
                       float* distances, int64_t* indices, int B, int64_t total_q, int64_t total_s, int k,
                       void* workspace, size_t workspace_bytes, int device, void* stream) {
  return rs_run("knn_pack_mode", q_points, s_points, q_lengths, s_lengths, indices, B, total_q, total_s, 0.f, k,
                workspace, workspace_bytes, device, stream, distances);
}
size_t knn_pack_cell_list_workspace_bytes(int B, int64_t total_s) { return rs_carve(nullptr, B, total_s).bytes; }
}  // namespace v3d

extern "C" {

int v3d_radius_search_pack_f32(const float* q_points, const float* s_points, const int64_t* q_lengths,
                               const int64_t* s_lengths, int64_t* neighbor_indices, int B, int64_t total_q,
                               int64_t total_s, float radius, int neighbor_limit, void* workspace,
                               size_t workspace_bytes, int device, void* stream) {
  if (neighbor_limit < 0) return v3d::invalid("radius_search_pack_mode", "negative dimension");
  return v3d::rs_run("radius_search_pack_mode", q_points, s_points, q_lengths, s_lengths, neighbor_indices, B, total_q,
                     total_s, radius, neighbor_limit, workspace, workspace_bytes, device, stream);
}

int v3d_radius_grid_build_f32(const float* s_points, const int64_t* s_lengths, int B, int64_t total_s, float cell_radius,
                              void* grid, size_t grid_bytes, int device, void* stream) {
  return v3d::rs_run("radius_grid_build", nullptr, s_points, nullptr, s_lengths, nullptr, B, 0, total_s, cell_radius, 1,
                     grid, grid_bytes, device, stream, nullptr, v3d::kRsBuildOnly, cell_radius);
}

int v3d_radius_search_grid_f32(const float* q_points, const int64_t* q_lengths, int64_t* neighbor_indices, int B,
                               int64_t total_q, int64_t total_s, float cell_radius, float radius, int neighbor_limit,
                               void* grid, size_t grid_bytes, int device, void* stream) {
  if (neighbor_limit < 0) return v3d::invalid("radius_search_grid", "negative dimension");
  return v3d::rs_run("radius_search_grid", q_points, nullptr, q_lengths, nullptr, neighbor_indices, B, total_q, total_s,
                     radius, neighbor_limit, grid, grid_bytes, device, stream, nullptr, v3d::kRsQueryOnly, cell_radius);
}

int v3d_radius_count_pack_f32(const float* q_points, const float* s_points, const int64_t* q_lengths,
                              const int64_t* s_lengths, int64_t* counts, int B, int64_t total_q, int64_t total_s,
                              float radius, void* workspace, size_t workspace_bytes, int device, void* stream) {
  return v3d::rs_run("radius_count_pack_mode", q_points, s_points, q_lengths, s_lengths, counts, B, total_q, total_s,
                     radius, -1, workspace, workspace_bytes, device, stream);
}

}  // extern "C"
