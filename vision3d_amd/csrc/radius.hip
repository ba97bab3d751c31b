// radius.hip -- radius neighbours and radius counts in PACK mode (variable-length clouds concatenated: points (total,3) + lengths (B,)).
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
// Distances use the reference expression order (dx*dx + dy*dy) + dz*dz with one rounding per operation.
#include <algorithm>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.h"
#include "wave_sort.h"

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
__global__ __launch_bounds__(kRsThreads) void rs_bbox_kernel(const float* __restrict__ s, const int64_t* __restrict__ s_len,
                                                             int* __restrict__ bbox) {
  __shared__ float red[6][kRsThreads / 64];
  const int b = blockIdx.x, tid = threadIdx.x;
  long long off = 0;
  for (int i = 0; i < b; ++i) off += s_len[i];
  const long long n = s_len[b];
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

// One thread: grids, cell offsets and point offsets of all clouds (B is small; everything else waits on this anyway).
__global__ void rs_plan_kernel(const int* __restrict__ bbox, const int64_t* __restrict__ q_len,
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
      float bb[6];
      for (int a = 0; a < 6; ++a) bb[a] = rs_unordered(bbox[b * 6 + a]);
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
    p.cell_off = cell_off;
    plan[b] = p;
    cell_off += p.dx * p.dy * p.dz;
    s_off += p.s_len;
    q_off += p.q_len;
  }
  (void)cell_start;  // the scan over the whole counter array also writes the end of the last cloud's last cell
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

// Cell starts = one exclusive scan over the counters of ALL clouds (rocprim): the cells of cloud b follow those of cloud
// b-1 and so do its supports, so the global running sum is the cloud's first support plus the sum within the cloud;
// counters past the last cell are zero, so every later entry (the end marker included) reads total_s.  The fill kernel
// counts the same counters back down to zero to hand out the slots of a cell (their order within a cell is free).
__global__ __launch_bounds__(kRsThreads) void rs_fill_kernel(const float* __restrict__ s, const RsPlan* __restrict__ plan,
                                                             const int* __restrict__ cell_start, int* __restrict__ cursor,
                                                             float4* __restrict__ sorted, int B, long long total_s) {
  const long long i = (long long)blockIdx.x * kRsThreads + threadIdx.x;
  if (i >= total_s) return;
  const RsPlan p = plan[rs_find_cloud(plan, B, i, false)];
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
                                                              long long total_s, float radius, int limit) {
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
    const bool inside = valid && !(sqrtf(d2) > radius);  // radius_search.py:66 masks dist > radius
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
                                                                   long long total_s, float radius, int limit) {
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
    const bool take = valid && !(sqrtf(d2) > radius) && d2 < INFINITY && key < worst;
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
                                                                    float radius) {
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
    cnt += __builtin_popcountll(__ballot(valid && (radius - sqrtf(d2) > 0.f)));  // relu(radius - d).sign()
  }
  if (lane == 0) counts[qi] = (int64_t)cnt;
}

}  // namespace v3d

extern "C" {

size_t v3d_radius_search_workspace_bytes(int B, int64_t total_s) {
  if (B <= 0 || total_s < 0) return 0;
  return v3d::rs_carve(nullptr, B, total_s).bytes;
}

}  // extern "C"

namespace v3d {
// neighbor_limit < 0: count mode (`out` = counts (total_q,)), else search mode (`out` = neighbor_indices)
static int rs_run(const char* what, const float* q_points, const float* s_points, const int64_t* q_lengths,
                  const int64_t* s_lengths, int64_t* neighbor_indices, int B, int64_t total_q, int64_t total_s,
                  float radius, int neighbor_limit, void* workspace, size_t workspace_bytes, int device, void* stream_) {
  const bool count_mode = neighbor_limit < 0;
  V3D_REQUIRE(B >= 0 && total_q >= 0 && total_s >= 0, what, "negative dimension");
  if (total_q == 0 || neighbor_limit == 0) return 0;
  V3D_REQUIRE(B > 0, what, "points without clouds");
  V3D_REQUIRE(q_points && q_lengths && s_lengths && neighbor_indices && (total_s == 0 || s_points), what,
              "null pointer");
  V3D_REQUIRE(radius >= 0.f, what, "negative radius");  // NaN fails too
  if (neighbor_limit > V3D_RADIUS_MAX_LIMIT) {
    set_last_error("%s: neighbor_limit=%d exceeds %d", what, neighbor_limit, V3D_RADIUS_MAX_LIMIT);
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
    hipLaunchKernelGGL(rs_bbox_init_kernel, dim3((unsigned)ceil_div((int64_t)B * 6, 256)), dim3(256), 0, stream, ws.bbox,
                       B);
    hipLaunchKernelGGL(rs_bbox_kernel, dim3(B, kRsBboxSplit), dim3(kRsThreads), 0, stream, s_points, s_lengths, ws.bbox);
    hipLaunchKernelGGL(rs_plan_kernel, dim3(1), dim3(64), 0, stream, ws.bbox, q_lengths, s_lengths, ws.plan,
                       ws.cell_start, B, radius, (long long)total_s);
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
    const unsigned qb = (unsigned)ceil_div(total_q, kRsThreads / 64);
    if (count_mode)
      hipLaunchKernelGGL(rs_count_query_kernel, dim3(qb), dim3(kRsThreads), 0, stream, q_points, ws.plan, ws.cell_start,
                         ws.sorted, neighbor_indices, B, (long long)total_q, radius);
    else if (neighbor_limit <= 64)
      hipLaunchKernelGGL(rs_query_sort_kernel, dim3(qb), dim3(kRsThreads), 0, stream, q_points, ws.plan, ws.cell_start,
                         ws.sorted, neighbor_indices, B, (long long)total_q, (long long)total_s, radius,
                         neighbor_limit);
    else if (neighbor_limit <= 128)
      hipLaunchKernelGGL(rs_query_kernel<2>, dim3(qb), dim3(kRsThreads), 0, stream, q_points, ws.plan, ws.cell_start,
                         ws.sorted, neighbor_indices, B, (long long)total_q, (long long)total_s, radius,
                         neighbor_limit);
    else if (neighbor_limit <= 256)
      hipLaunchKernelGGL(rs_query_kernel<4>, dim3(qb), dim3(kRsThreads), 0, stream, q_points, ws.plan, ws.cell_start,
                         ws.sorted, neighbor_indices, B, (long long)total_q, (long long)total_s, radius,
                         neighbor_limit);
    else
      hipLaunchKernelGGL(rs_query_kernel<8>, dim3(qb), dim3(kRsThreads), 0, stream, q_points, ws.plan, ws.cell_start,
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

int v3d_radius_count_pack_f32(const float* q_points, const float* s_points, const int64_t* q_lengths,
                              const int64_t* s_lengths, int64_t* counts, int B, int64_t total_q, int64_t total_s,
                              float radius, void* workspace, size_t workspace_bytes, int device, void* stream) {
  return v3d::rs_run("radius_count_pack_mode", q_points, s_points, q_lengths, s_lengths, counts, B, total_q, total_s,
                     radius, -1, workspace, workspace_bytes, device, stream);
}

}  // extern "C"
