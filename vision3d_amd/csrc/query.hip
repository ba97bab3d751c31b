// query.hip -- brute-force neighbour queries on 3-D points: ball_query, three_nn, knn_points.
// Semantics: ball_query_cuda.cuh:11-55, three_nn_cuda.cuh:11-75, knn_points_cuda.cuh:9-93 of the reference.
// Design (gfx950): support points of one cloud are staged once per workgroup in LDS and shared by all its
// queries; all three kernels are VALU/latency bound (a few MB of I/O against 10^8 pair evaluations).
//   * ball_query : one WAVE per group of G queries.  The 64 lanes test 64 consecutive supports at once; a
//                  ballot + prefix popcount gives every hit its slot in INDEX ORDER, so the "first k in scan
//                  order" rule of the reference is kept while hits are stored coalesced; a wave stops as soon
//                  as its queries are full.
//   * three_nn   : one thread per query, supports broadcast from LDS as 16-byte points, branch-free
//                  strict-< insertion cascade (ties keep the lower index first, like the reference).
//   * knn_points : one wave per group of 4 queries.  Lanes evaluate 64 supports at a time; per query the k+1 best
//                  so far live sorted, one entry per lane, and a chunk is filtered by one compare + ballot against
//                  the (k+1)-th best; survivors are inserted with a whole-wave DPP shift.  If the final k+1
//                  distances are all different the reference's heap can only yield this list; otherwise (exact
//                  ties) that query is replayed through the reference's max-heap (one slot per lane, sifted by the
//                  whole wave), so ties come out in exactly the reference's order.  Clouds of 2048..4096 supports
//                  (fp32, k <= 63) are searched on a cell list each workgroup builds in LDS
//                  (knn_points_grid_kernel) -- same k+1 keys, same replay.  v3d_knn_lex_f32 (what ops.knn runs for
//                  3-D points) and v3d_knn_pack_f32 leave equal distances lower index first (pykeops' order): the
//                  list itself, no replay; pack mode sends clouds of >= 2048 supports to radius.hip's cell list.
// Squared distances use the reference's expression order with one rounding per operation (the library is
// compiled with -ffp-contract=off).
#include <cmath>
#include <mutex>
#include <type_traits>

#include "common.h"
#include "wave_sort.h"

namespace v3d {

constexpr int kQueryThreads = 256;
constexpr int kWavesPerBlock = kQueryThreads / kWave;

template <typename T>
__device__ __forceinline__ T sq_dist(T qx, T qy, T qz, T sx, T sy, T sz) {
  T dx = qx - sx, dy = qy - sy, dz = qz - sz;
  return (dx * dx + dy * dy) + dz * dz;
}

// Two queries against one support in packed fp32 (v_pk_add_f32 / v_pk_mul_f32: per-element IEEE results identical
// to the scalar form, half the instructions); the double variant is two scalar evaluations.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned pd_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void sq_dist_pair(float qx0, float qy0, float qz0, float qx1, float qy1, float qz1, float sx,
                                             float sy, float sz, float& d0, float& d1) {
  const f32x2_t dx = f32x2_t{qx0, qx1} - f32x2_t{sx, sx};
  const f32x2_t dy = f32x2_t{qy0, qy1} - f32x2_t{sy, sy};
  const f32x2_t dz = f32x2_t{qz0, qz1} - f32x2_t{sz, sz};
  const f32x2_t d = (dx * dx + dy * dy) + dz * dz;
  d0 = d[0];
  d1 = d[1];
}
__device__ __forceinline__ void sq_dist_pair(double qx0, double qy0, double qz0, double qx1, double qy1, double qz1,
                                             double sx, double sy, double sz, double& d0, double& d1) {
  d0 = sq_dist(qx0, qy0, qz0, sx, sy, sz);
  d1 = sq_dist(qx1, qy1, qz1, sx, sy, sz);
}

__device__ __forceinline__ float read_lane(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ double read_lane(double v, int lane) {
  long long bits = __double_as_longlong(v);
  int lo = __builtin_amdgcn_readlane((int)(bits & 0xffffffffll), lane);
  int hi = __builtin_amdgcn_readlane((int)(bits >> 32), lane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Cooperative copy of `count` support points (AoS, 3*T each) of one cloud into LDS, unchanged layout.
// Lane i later reads words 3i, 3i+1, 3i+2: stride 3 is co-prime with the bank count, so conflict-free.
template <typename T>
__device__ __forceinline__ void stage_supports_aos(const T* __restrict__ s_cloud, int first, int count, T* lds) {
  const T* g = s_cloud + (int64_t)first * 3;
  const int nbytes = count * 3 * (int)sizeof(T);
  const int tid = threadIdx.x, nthr = blockDim.x;
  if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
    // 16-byte copies, eight in flight per thread before the first LDS store (a one-load-one-store loop would pay
    // the full L2 latency per iteration)
    const uint4* g4 = reinterpret_cast<const uint4*>(g);
    uint4* l4 = reinterpret_cast<uint4*>(lds);
    const int nvec = nbytes >> 4;
    for (int base = 0; base < nvec; base += 8 * nthr) {
      uint4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + u * nthr + tid;
        v[u] = i < nvec ? g4[i] : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + u * nthr + tid;
        if (i < nvec) l4[i] = v[u];
      }
    }
    for (int i = (nvec << 4) / (int)sizeof(T) + tid; i < count * 3; i += nthr) lds[i] = g[i];
  } else {
    for (int base = 0; base < count * 3; base += 8 * nthr) {
      T v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + u * nthr + tid;
        v[u] = i < count * 3 ? g[i] : T(0);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + u * nthr + tid;
        if (i < count * 3) lds[i] = v[u];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// ball query
// ------------------------------------------------------------------------------------------------------
template <typename T, int G, int TS>
__global__ __launch_bounds__(kQueryThreads) void ball_query_kernel(const T* __restrict__ q, const T* __restrict__ s,
                                                                   int64_t* __restrict__ out, int M, int N, int k,
                                                                   T r2) {
  __shared__ __align__(16) T tile[TS * 3];
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q_first = (blockIdx.x * kWavesPerBlock + wave) * G;
  const T* s_cloud = s + (int64_t)b * N * 3;

  T qx[G], qy[G], qz[G];
  int cnt[G], first[G];
  int nopen = 0;                    // queries of this wave that still take hits
  __amdgpu_buffer_rsrc_t row[G];    // the output rows of the G queries: k entries each, stores beyond are dropped
#pragma unroll
  for (int g = 0; g < G; ++g) {
    int qi = q_first + g;
    bool live = qi < M;
    const T* qp = q + ((int64_t)b * M + (live ? qi : 0)) * 3;
    qx[g] = qp[0];
    qy[g] = qp[1];
    qz[g] = qp[2];
    cnt[g] = live ? 0 : k;  // dead slots count as full
    nopen += (live && k > 0) ? 1 : 0;
    first[g] = INT_MAX;
    row[g] = __builtin_amdgcn_make_buffer_rsrc(out + ((int64_t)b * M + (live ? qi : 0)) * k, 0, live ? k * 8 : 0, 0x00020000);
  }

  for (int t0 = 0; t0 < N; t0 += TS) {
    const int tcount = min(TS, N - t0);
    __syncthreads();
    stage_supports_aos(s_cloud, t0, tcount, tile);
    {  // pad the last chunk to 64 supports with points at +inf: their distance is +inf, never below the radius
      const int padded = (tcount + 63) & ~63;
      for (int i = tcount * 3 + (int)threadIdx.x; i < padded * 3; i += kQueryThreads) tile[i] = static_cast<T>(INFINITY);
    }
    __syncthreads();
    if (nopen == 0) continue;  // wave-uniform; keep taking part in the barriers
    // A full (or dead) query compares against -1 so it can never hit again: the common no-hit chunk is then
    // branch-free (G distance evaluations, G compares, one combined test) and only chunks with a hit pay for slot
    // assignment and stores.
    T lim[G];
#pragma unroll
    for (int g = 0; g < G; ++g) lim[g] = cnt[g] < k ? r2 : static_cast<T>(-1);
    const T* sp = tile + lane * 3;
    int pos = t0;  // index of the chunk's first support
    const int pos_end = t0 + tcount;
    do {
      const T sx = sp[0], sy = sp[1], sz = sp[2];  // tail lanes read +inf padding
      sp += 64 * 3;
      unsigned long long m[G];
      bool hit[G];
      unsigned long long any = 0ull;
#pragma unroll
      for (int g = 0; g < G; g += 2) {
        T d0, d1;
        sq_dist_pair(qx[g], qy[g], qz[g], qx[g + 1], qy[g + 1], qz[g + 1], sx, sy, sz, d0, d1);
        hit[g] = d0 < lim[g];  // strict, ball_query_cuda.cuh:43
        hit[g + 1] = d1 < lim[g + 1];
        m[g] = __ballot(hit[g]);
        m[g + 1] = __ballot(hit[g + 1]);
        any |= m[g] | m[g + 1];
      }
      if (any != 0ull) {
        // The hit path is where this kernel spends a third of its time at the radii of the MSG layers (two of three
        // chunks hold a hit at r = 0.1, every chunk at r = 0.2 / 0.4), so it is kept to ~20 instructions per query:
        // the slot is the running count + the hits in the lower lanes (two `v_mbcnt`, the count rides in as the
        // addend); the store is a buffer store into the query's row of k entries, whose range check drops the slots
        // >= k, and the lanes without a hit are sent out of range as well; everything else is scalar arithmetic.
        const int64_t idx = (int64_t)(pos + lane);
        const pd_u32x2 idx_bits = {(unsigned)idx, (unsigned)(idx >> 32)};
#pragma unroll
        for (int g = 0; g < G; ++g) {
          if (m[g]) {  // wave-uniform
            const unsigned slot = __builtin_amdgcn_mbcnt_hi(
                (unsigned)(m[g] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m[g], (unsigned)cnt[g]));
            __builtin_amdgcn_raw_buffer_store_b64(idx_bits, row[g], hit[g] ? slot * 8u : 0xfffffff8u, 0, 0);
            first[g] = min(first[g], pos + (int)__builtin_ctzll(m[g]));  // chunks ascend: the minimum is the first hit
            cnt[g] += __builtin_popcountll(m[g]);
            const bool full = cnt[g] >= k;  // happens once per query: a full query never hits again
            lim[g] = full ? static_cast<T>(-1) : lim[g];
            nopen -= full ? 1 : 0;
          }
        }
        pos = nopen == 0 ? INT_MAX - 64 : pos;  // every query full: leave (a scalar select, no second loop exit)
      }
      pos += 64;
    } while (pos < pos_end);
  }
  // A query with 0 < cnt < k repeats its first hit in the remaining slots (ball_query_cuda.cuh:44-48);
  // cnt == 0 leaves the caller's zero fill untouched.
#pragma unroll
  for (int g = 0; g < G; ++g) {
    if (q_first + g < M && cnt[g] > 0 && cnt[g] < k) {
      int64_t* r = out + ((int64_t)b * M + q_first + g) * k;
      for (int slot = cnt[g] + lane; slot < k; slot += 64) r[slot] = (int64_t)first[g];
    }
  }
}

template <typename T>
static int ball_query(const T* q, const T* s, int64_t* out, int B, int M, int N, float max_radius, int k, int device,
                      void* stream_) {
  const char* what = "ball_query";
  V3D_REQUIRE(B >= 0 && M >= 0 && N >= 0 && k >= 0, what, "negative dimension");
  if (B == 0 || M == 0 || k == 0 || N == 0) return 0;
  V3D_REQUIRE(q && s && out, what, "null pointer");
  if (B > kMaxGridYZ || k > (1 << 27)) {  // a row of k int64 entries is addressed as a buffer of k * 8 bytes
    set_last_error("%s: B <= 65535 and k <= 2^27 required", what);
    return V3D_ERR_SIZE_LIMIT;
  }
  DeviceGuard guard(device);
  if (guard.rc) return guard.rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const T r2 = static_cast<T>(max_radius) * static_cast<T>(max_radius);  // ball_query_cuda.cuh:28
  constexpr int G = 4;
  constexpr int TS = 2048;
  dim3 grid((unsigned)ceil_div(M, kWavesPerBlock * G), (unsigned)B), block(kQueryThreads);
  hipLaunchKernelGGL((ball_query_kernel<T, G, TS>), grid, block, 0, stream, q, s, out, M, N, k, r2);
  return launch_status(what);
}

// Multi-scale grouping (PointNet++ MSG, models/pointnet2/modules.py:47-57, calls ball_query once per (radius, k) on the
// same queries and supports): one pass evaluates every pair ONCE and tests it against all S radii; each scale keeps its
// own counters, first hit and output tensor, so every output equals the single-scale kernel's bit for bit.  A query
// leaves the scan when all of its scales are full.
constexpr int kBallMaxScales = 4;
struct BallScales {
  int64_t* out[kBallMaxScales];
  float r2[kBallMaxScales];
  int k[kBallMaxScales];
};

template <int S, int G, int TS>
__global__ __launch_bounds__(kQueryThreads) void ball_query_multi_kernel(const float* __restrict__ q,
                                                                         const float* __restrict__ s, BallScales sc, int M,
                                                                         int N) {
  __shared__ __align__(16) float tile[TS * 3];
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q_first = (blockIdx.x * kWavesPerBlock + wave) * G;
  const float* s_cloud = s + (int64_t)b * N * 3;

  float qx[G], qy[G], qz[G];
  int cnt[G][S], first[G][S];
  float lim[G][S];
  int nopen = 0;  // (query, scale) lists of this wave that still take hits
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int qi = q_first + g;
    const bool live = qi < M;
    const float* qp = q + ((int64_t)b * M + (live ? qi : 0)) * 3;
    qx[g] = qp[0];
    qy[g] = qp[1];
    qz[g] = qp[2];
#pragma unroll
    for (int i = 0; i < S; ++i) {
      cnt[g][i] = live ? 0 : sc.k[i];  // dead slots count as full
      nopen += (live && sc.k[i] > 0) ? 1 : 0;
      first[g][i] = INT_MAX;
    }
  }

  for (int t0 = 0; t0 < N; t0 += TS) {
    const int tcount = min(TS, N - t0);
    __syncthreads();
    stage_supports_aos(s_cloud, t0, tcount, tile);
    {
      const int padded = (tcount + 63) & ~63;
      for (int i = tcount * 3 + (int)threadIdx.x; i < padded * 3; i += kQueryThreads) tile[i] = INFINITY;
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
      for (int i = 0; i < S; ++i) lim[g][i] = cnt[g][i] < sc.k[i] ? sc.r2[i] : -1.0f;  // a full (or dead) list never hits
    }
    if (nopen == 0) continue;  // wave-uniform; keep taking part in the barriers
    const float* sp = tile + lane * 3;
    int pos = t0;  // index of the chunk's first support
    const int pos_end = t0 + tcount;
    do {
      const float sx = sp[0], sy = sp[1], sz = sp[2];
      sp += 64 * 3;
      unsigned long long m[G][S];
      bool hit[G][S];
      unsigned long long any = 0ull;
#pragma unroll
      for (int g = 0; g < G; g += 2) {
        float d0, d1;
        sq_dist_pair(qx[g], qy[g], qz[g], qx[g + 1], qy[g + 1], qz[g + 1], sx, sy, sz, d0, d1);
#pragma unroll
        for (int i = 0; i < S; ++i) {
          hit[g][i] = d0 < lim[g][i];  // strict, ball_query_cuda.cuh:43
          hit[g + 1][i] = d1 < lim[g + 1][i];
          m[g][i] = __ballot(hit[g][i]);
          m[g + 1][i] = __ballot(hit[g + 1][i]);
          any |= m[g][i] | m[g + 1][i];
        }
      }
      if (any != 0ull) {  // the hit path of ball_query_kernel, per (query, scale) list
        const int64_t idx = (int64_t)(pos + lane);
#pragma unroll
        for (int g = 0; g < G; ++g) {
#pragma unroll
          for (int i = 0; i < S; ++i) {
            const unsigned long long mm = m[g][i];
            if (mm) {  // wave-uniform
              const unsigned slot = __builtin_amdgcn_mbcnt_hi(
                  (unsigned)(mm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mm, (unsigned)cnt[g][i]));
              if (hit[g][i] && slot < (unsigned)sc.k[i]) sc.out[i][((int64_t)b * M + q_first + g) * sc.k[i] + slot] = idx;
              first[g][i] = min(first[g][i], pos + (int)__builtin_ctzll(mm));  // chunks ascend: the minimum is the first hit
              cnt[g][i] += __builtin_popcountll(mm);
              const bool full = cnt[g][i] >= sc.k[i];  // happens once per list: a full list never hits again
              lim[g][i] = full ? -1.0f : lim[g][i];
              nopen -= full ? 1 : 0;
            }
          }
        }
        pos = nopen == 0 ? INT_MAX - 64 : pos;  // every list full: leave (a scalar select, no second loop exit)
      }
      pos += 64;
    } while (pos < pos_end);
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {
#pragma unroll
    for (int i = 0; i < S; ++i) {
      if (q_first + g < M && cnt[g][i] > 0 && cnt[g][i] < sc.k[i]) {  // ball_query_cuda.cuh:44-48
        int64_t* row = sc.out[i] + ((int64_t)b * M + q_first + g) * sc.k[i];
        for (int slot = cnt[g][i] + lane; slot < sc.k[i]; slot += 64) row[slot] = (int64_t)first[g][i];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// three_nn
// ------------------------------------------------------------------------------------------------------
template <typename T>
struct alignas(sizeof(T) * 4) Point4 {
  T x, y, z, pad;
};

template <typename T, int TS>
__global__ __launch_bounds__(kQueryThreads) void three_nn_kernel(const T* __restrict__ q, const T* __restrict__ s,
                                                                 T* __restrict__ out_d, int64_t* __restrict__ out_i,
                                                                 int Q, int S) {
  __shared__ Point4<T> tile[TS];
  const int b = blockIdx.y;
  const int qi = blockIdx.x * kQueryThreads + threadIdx.x;
  const bool live = qi < Q;
  const T* qp = q + ((int64_t)b * Q + (live ? qi : 0)) * 3;
  const T qx = qp[0], qy = qp[1], qz = qp[2];
  const T* s_cloud = s + (int64_t)b * S * 3;
  // the reference keeps the bests as double 1e40 (three_nn_cuda.cuh:37-39): any finite T compares below it
  // and +inf / NaN never do -- +inf in T behaves identically under the strict '<'.
  T b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;
  int i1 = -1, i2 = -1, i3 = -1;
  for (int t0 = 0; t0 < S; t0 += TS) {
    const int tcount = min(TS, S - t0);
    __syncthreads();
    for (int i = threadIdx.x; i < tcount; i += kQueryThreads) {
      const T* g = s_cloud + (int64_t)(t0 + i) * 3;
      Point4<T> p;
      p.x = g[0];
      p.y = g[1];
      p.z = g[2];
      p.pad = T(0);
      tile[i] = p;
    }
    __syncthreads();
#pragma unroll 4
    for (int j = 0; j < tcount; ++j) {
      const Point4<T> p = tile[j];  // same address in every lane: LDS broadcast
      const T d = sq_dist(qx, qy, qz, p.x, p.y, p.z);
      const bool lt1 = d < b1, lt2 = d < b2, lt3 = d < b3;  // b1 <= b2 <= b3 always, so lt1 => lt2 => lt3
      const int id = t0 + j;
      b3 = lt2 ? b2 : (lt3 ? d : b3);
      i3 = lt2 ? i2 : (lt3 ? id : i3);
      b2 = lt1 ? b1 : (lt2 ? d : b2);
      i2 = lt1 ? i1 : (lt2 ? id : i2);
      b1 = lt1 ? d : b1;
      i1 = lt1 ? id : i1;
    }
  }
  if (live) {
    // unfilled slots: sqrt(1e40) = 1e20 cast to T, index -1 (three_nn_cuda.cuh:67-73)
    const T unfilled = static_cast<T>(1e20);
    T* od = out_d + ((int64_t)b * Q + qi) * 3;
    int64_t* oi = out_i + ((int64_t)b * Q + qi) * 3;
    od[0] = i1 >= 0 ? static_cast<T>(sqrt(b1)) : unfilled;
    od[1] = i2 >= 0 ? static_cast<T>(sqrt(b2)) : unfilled;
    od[2] = i3 >= 0 ? static_cast<T>(sqrt(b3)) : unfilled;
    oi[0] = i1;
    oi[1] = i2;
    oi[2] = i3;
  }
}

// Cell-list variant (fp32, 64 <= S <= 2048).  Every workgroup (256 queries of one cloud) sorts the cloud's supports into
// a G^3 uniform grid over their bounding box, all in LDS (count, scan, scatter: integer LDS atomics only), then each
// thread walks the cells around its query ring by ring -- 27 cells hold ~2 supports each for PointNet++'s 1024 sampled
// centroids, against 1024 candidates of the brute-force scan -- and stops as soon as its third-best distance is below
// the distance to the nearest unexplored cell face (minus a safety margin that covers the rounding of the cell
// assignment).  Candidates arrive in cell order, not index order, so insertion compares (distance, index)
// lexicographically: the result is the brute-force one bit for bit, ties included (lower index first).
template <int G>
__global__ __launch_bounds__(kQueryThreads) void three_nn_grid_kernel(const float* __restrict__ q,
                                                                      const float* __restrict__ s,
                                                                      float* __restrict__ out_d,
                                                                      int64_t* __restrict__ out_i, int Q, int S) {
  constexpr int NC = G * G * G;
  constexpr int kMaxPer = 2048 / kQueryThreads;  // supports per thread
  extern __shared__ float4 nn_sorted[];          // S entries: x, y, z, index bits
  __shared__ int cell_start[NC + 1];
  __shared__ int cursor[NC];
  __shared__ float red[6][kWavesPerBlock];
  __shared__ int wave_sum[kWavesPerBlock];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const float* s_cloud = s + (int64_t)b * S * 3;

  // ---- supports of this thread, bounding box --------------------------------------------------------------------
  float sx[kMaxPer], sy[kMaxPer], sz[kMaxPer];
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
  for (int u = 0; u < kMaxPer; ++u) {
    const int i = u * kQueryThreads + tid;
    if (i < S) {
      sx[u] = s_cloud[i * 3 + 0];
      sy[u] = s_cloud[i * 3 + 1];
      sz[u] = s_cloud[i * 3 + 2];
      lo[0] = fminf(lo[0], sx[u]); hi[0] = fmaxf(hi[0], sx[u]);
      lo[1] = fminf(lo[1], sy[u]); hi[1] = fmaxf(hi[1], sy[u]);
      lo[2] = fminf(lo[2], sz[u]); hi[2] = fmaxf(hi[2], sz[u]);
    } else {
      sx[u] = sy[u] = sz[u] = 0.f;
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    for (int d = 32; d >= 1; d >>= 1) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], d, 64));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], d, 64));
    }
    if (lane == 0) {
      red[a][wave] = lo[a];
      red[3 + a][wave] = hi[a];
    }
  }
  for (int i = tid; i < NC; i += kQueryThreads) cell_start[i] = 0;
  __syncthreads();
  float cell[3], inv[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    lo[a] = red[a][0];
    hi[a] = red[3 + a][0];
    for (int w = 1; w < kWavesPerBlock; ++w) {
      lo[a] = fminf(lo[a], red[a][w]);
      hi[a] = fmaxf(hi[a], red[3 + a][w]);
    }
    const float ext = hi[a] - lo[a];
    cell[a] = ext > 0.f ? ext / (float)G : 1.0f;  // a flat axis: everything lands in cell 0 of that axis
    inv[a] = 1.0f / cell[a];
  }
  auto coord = [&](float v, int a) { return max(0, min(G - 1, (int)floorf((v - lo[a]) * inv[a]))); };

  // ---- counting sort of the supports by cell (LDS only) ---------------------------------------------------------
  int cid[kMaxPer];
#pragma unroll
  for (int u = 0; u < kMaxPer; ++u) {
    const int i = u * kQueryThreads + tid;
    cid[u] = (coord(sz[u], 2) * G + coord(sy[u], 1)) * G + coord(sx[u], 0);
    if (i < S) atomicAdd(&cell_start[cid[u]], 1);
  }
  __syncthreads();
  {  // exclusive scan of the NC counts: thread t owns cells 2t, 2t+1 (NC <= 512)
    const int c0 = 2 * tid < NC ? cell_start[2 * tid] : 0, c1 = 2 * tid + 1 < NC ? cell_start[2 * tid + 1] : 0;
    const int mine = c0 + c1;
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d, 64);
      if (lane >= d) incl += t;
    }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    int before = incl - mine;
    for (int w = 0; w < wave; ++w) before += wave_sum[w];
    if (2 * tid < NC) {
      cell_start[2 * tid] = before;
      cursor[2 * tid] = before;
    }
    if (2 * tid + 1 < NC) {
      cell_start[2 * tid + 1] = before + c0;
      cursor[2 * tid + 1] = before + c0;
    }
    if (tid == 0) cell_start[NC] = S;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kMaxPer; ++u) {
    const int i = u * kQueryThreads + tid;
    if (i < S) {
      const int pos = atomicAdd(&cursor[cid[u]], 1);
      nn_sorted[pos] = make_float4(sx[u], sy[u], sz[u], __int_as_float(i));
    }
  }
  __syncthreads();

  // ---- queries ---------------------------------------------------------------------------------------------------
  const int qi = blockIdx.x * kQueryThreads + tid;
  if (qi >= Q) return;  // no barrier below
  const float* qp = q + ((int64_t)b * Q + qi) * 3;
  const float qv[3] = {qp[0], qp[1], qp[2]};
  const int cc[3] = {coord(qv[0], 0), coord(qv[1], 1), coord(qv[2], 2)};
  // the three best as 64-bit keys (distance bits << 32 | index): distances are >= 0, so the unsigned order of the keys
  // IS the lexicographic (distance, index) order; unfilled = (+inf, 0xffffffff); a NaN distance sorts above +inf and is
  // never inserted, like under the brute-force kernel's strict '<'
  const unsigned long long kEmpty = ((unsigned long long)0x7f800000u << 32) | 0xffffffffull;
  unsigned long long k1 = kEmpty, k2 = kEmpty, k3 = kEmpty;
  auto offer = [&](const float4 p, bool on) {
    const float d = sq_dist(qv[0], qv[1], qv[2], p.x, p.y, p.z);
    // d < inf: an infinite (or NaN) distance is never below the brute-force kernel's initial +inf either
    const unsigned long long key = (on && d < INFINITY)
        ? (((unsigned long long)__float_as_uint(d) << 32) | (unsigned)__float_as_int(p.w)) : ~0ull;
    const bool lt1 = key < k1, lt2 = key < k2, lt3 = key < k3;  // k1 <= k2 <= k3, so lt1 => lt2 => lt3
    k3 = lt2 ? k2 : (lt3 ? key : k3);
    k2 = lt1 ? k1 : (lt2 ? key : k2);
    k1 = lt1 ? key : k1;
  };
  auto scan = [&](int first_cell, int last_cell) {
    const int end = cell_start[last_cell + 1];
    for (int j = cell_start[first_cell]; j < end; j += 4) {  // four candidates per step, all reads issued first
      const float4 pa = nn_sorted[j];
      const float4 pb = nn_sorted[min(j + 1, end - 1)];
      const float4 pc = nn_sorted[min(j + 2, end - 1)];
      const float4 pd = nn_sorted[min(j + 3, end - 1)];
      offer(pa, true);
      offer(pb, j + 1 < end);
      offer(pc, j + 2 < end);
      offer(pd, j + 3 < end);
    }
  };
  int p0[3] = {1, 1, 1}, p1[3] = {0, 0, 0};  // previous region (empty)
  for (int R = 1; R <= G; ++R) {
    int r0[3], r1[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      r0[a] = max(cc[a] - R, 0);
      r1[a] = min(cc[a] + R, G - 1);
    }
    for (int z = r0[2]; z <= r1[2]; ++z) {
      for (int y = r0[1]; y <= r1[1]; ++y) {
        const int row = (z * G + y) * G;
        const bool inner = z >= p0[2] && z <= p1[2] && y >= p0[1] && y <= p1[1];
        if (!inner) {
          scan(row + r0[0], row + r1[0]);  // cells of one (z, y) row are contiguous in the sorted order
        } else {
          if (cc[0] - R >= 0) scan(row + cc[0] - R, row + cc[0] - R);
          if (cc[0] + R <= G - 1) scan(row + cc[0] + R, row + cc[0] + R);
        }
      }
    }
    // nearest face of the explored box that still has cells behind it
    float bound = INFINITY;
    bool open = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      // everything relative to the box origin (clouds far from the coordinate origin: lo + j * cell would be rounded at
      // the magnitude of lo); the margin covers the rounding of the cell assignment and of the two sides of the test
      const float rel = qv[a] - lo[a];
      const float margin = 1e-5f * fmaxf(hi[a] - lo[a], fabsf(rel));
      if (r0[a] > 0) {
        open = true;
        bound = fminf(bound, fmaxf(rel - (float)r0[a] * cell[a] - margin, 0.f));
      }
      if (r1[a] < G - 1) {
        open = true;
        bound = fminf(bound, fmaxf((float)(r1[a] + 1) * cell[a] - rel - margin, 0.f));
      }
    }
    const float b3 = __uint_as_float((unsigned)(k3 >> 32));
    if (!open || b3 < bound * bound) break;  // (NaN bounds compare false: the search widens to the whole grid)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      p0[a] = r0[a];
      p1[a] = r1[a];
    }
  }
  const float unfilled = 1e20f;
  float* od = out_d + ((int64_t)b * Q + qi) * 3;
  int64_t* oi = out_i + ((int64_t)b * Q + qi) * 3;
  const int i1 = (int)(unsigned)k1, i2 = (int)(unsigned)k2, i3 = (int)(unsigned)k3;  // -1 where unfilled
  od[0] = i1 >= 0 ? sqrtf(__uint_as_float((unsigned)(k1 >> 32))) : unfilled;
  od[1] = i2 >= 0 ? sqrtf(__uint_as_float((unsigned)(k2 >> 32))) : unfilled;
  od[2] = i3 >= 0 ? sqrtf(__uint_as_float((unsigned)(k3 >> 32))) : unfilled;
  oi[0] = i1;
  oi[1] = i2;
  oi[2] = i3;
}

template <typename T>
static int three_nn(const T* q, const T* s, T* od, int64_t* oi, int B, int Q, int S, int device, void* stream_) {
  const char* what = "three_nn";
  V3D_REQUIRE(B >= 0 && Q >= 0 && S >= 0, what, "negative dimension");
  if (B == 0 || Q == 0) return 0;
  V3D_REQUIRE(q && od && oi && (S == 0 || s), what, "null pointer");
  if (B > kMaxGridYZ) {
    set_last_error("%s: B <= 65535 required", what);
    return V3D_ERR_SIZE_LIMIT;
  }
  DeviceGuard guard(device);
  if (guard.rc) return guard.rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  constexpr int TS = 1024;
  dim3 grid((unsigned)ceil_div(Q, kQueryThreads), (unsigned)B), block(kQueryThreads);
  if constexpr (std::is_same<T, float>::value) {
    if (S >= 64 && S <= 2048) {  // cell list in LDS; smaller sets are launch-bound either way, larger ones do not fit
      const size_t lds = (size_t)S * sizeof(float4);
      // a few supports per cell: the 27 cells around a query then hold its three nearest almost always (one ring);
      // measured at S = 1024: G = 4 79 us, 5 51-53, 6 51-57, 7 55-62, 8 63 (brute force: 92)
      const int g = S >= 1600 ? 7 : (S >= 900 ? 6 : (S >= 384 ? 5 : 4));  // ~5-8 supports per cell
      if (g == 7) hipLaunchKernelGGL((three_nn_grid_kernel<7>), grid, block, lds, stream, q, s, od, oi, Q, S);
      else if (g == 6) hipLaunchKernelGGL((three_nn_grid_kernel<6>), grid, block, lds, stream, q, s, od, oi, Q, S);
      else if (g == 5) hipLaunchKernelGGL((three_nn_grid_kernel<5>), grid, block, lds, stream, q, s, od, oi, Q, S);
      else hipLaunchKernelGGL((three_nn_grid_kernel<4>), grid, block, lds, stream, q, s, od, oi, Q, S);
      return launch_status(what);
    }
  }
  hipLaunchKernelGGL((three_nn_kernel<T, TS>), grid, block, 0, stream, q, s, od, oi, Q, S);
  return launch_status(what);
}

// ------------------------------------------------------------------------------------------------------
// radius count: thread per query, supports broadcast from LDS (same staging as three_nn)
// ------------------------------------------------------------------------------------------------------
template <int TS>
__global__ __launch_bounds__(kQueryThreads) void radius_count_kernel(const float* __restrict__ q,
                                                                     const float* __restrict__ s,
                                                                     int64_t* __restrict__ counts, int M, int N,
                                                                     float radius) {
  __shared__ Point4<float> tile[TS];
  const int b = blockIdx.y;
  const int qi = blockIdx.x * kQueryThreads + threadIdx.x;
  const bool live = qi < M;
  const float* qp = q + ((int64_t)b * M + (live ? qi : 0)) * 3;
  const float qx = qp[0], qy = qp[1], qz = qp[2];
  const float* s_cloud = s + (int64_t)b * N * 3;
  int cnt = 0;
  for (int t0 = 0; t0 < N; t0 += TS) {
    const int tcount = min(TS, N - t0);
    __syncthreads();
    for (int i = threadIdx.x; i < tcount; i += kQueryThreads) {
      const float* g = s_cloud + (int64_t)(t0 + i) * 3;
      Point4<float> p;
      p.x = g[0];
      p.y = g[1];
      p.z = g[2];
      p.pad = 0.f;
      tile[i] = p;
    }
    __syncthreads();
#pragma unroll 4
    for (int j = 0; j < tcount; ++j) {
      const Point4<float> p = tile[j];
      const float d = sqrtf(sq_dist(qx, qy, qz, p.x, p.y, p.z));
      cnt += (radius - d > 0.f) ? 1 : 0;  // relu(radius - d).sign()
    }
  }
  if (live) counts[(int64_t)b * M + qi] = cnt;
}

// ------------------------------------------------------------------------------------------------------
// knn_points
// ------------------------------------------------------------------------------------------------------
// ---- lane-distributed sorted list (ascending by (distance, arrival)), CAP = 64*R entries ----------------------
__device__ __forceinline__ int shr1_fill(int v, int fill) {  // lane l <- lane l-1, lane 0 <- fill
  return __builtin_amdgcn_update_dpp(fill, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ float shr1_fill(float v, float fill) {
  return __int_as_float(shr1_fill(__float_as_int(v), __float_as_int(fill)));
}
__device__ __forceinline__ double shr1_fill(double v, double fill) {
  const long long b = __double_as_longlong(v), f = __double_as_longlong(fill);
  const int lo = shr1_fill((int)(b & 0xffffffffll), (int)(f & 0xffffffffll));
  const int hi = shr1_fill((int)(b >> 32), (int)(f >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ int shl1_fill(int v, int fill) {  // lane l <- lane l+1, lane 63 <- fill
  return __builtin_amdgcn_update_dpp(fill, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ float shl1_fill(float v, float fill) {
  return __int_as_float(shl1_fill(__float_as_int(v), __float_as_int(fill)));
}
__device__ __forceinline__ double shl1_fill(double v, double fill) {
  const long long b = __double_as_longlong(v), f = __double_as_longlong(fill);
  const int lo = shl1_fill((int)(b & 0xffffffffll), (int)(f & 0xffffffffll));
  const int hi = shl1_fill((int)(b >> 32), (int)(f >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ float lane_permute(float v, int dst_lane) {  // lane l sends v to lane dst_lane
  return __int_as_float(__builtin_amdgcn_ds_permute(dst_lane << 2, __float_as_int(v)));
}
__device__ __forceinline__ int lane_permute(int v, int dst_lane) { return __builtin_amdgcn_ds_permute(dst_lane << 2, v); }
__device__ __forceinline__ double lane_permute(double v, int dst_lane) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_ds_permute(dst_lane << 2, (int)(b & 0xffffffffll));
  const int hi = __builtin_amdgcn_ds_permute(dst_lane << 2, (int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// (k+1)-th smallest (k = 0..63) of the 64 lanes' non-negative values, or an upper bound of it that is exact in the
// leading 32 bits: bit-wise binary search on the IEEE pattern (order-preserving for non-negative numbers), one
// compare + ballot + popcount per bit.  Returns +inf when fewer than k+1 lanes hold a finite value.
// `low_bit` > 0 stops the search there and rounds the undecided bits up: a looser bound for fewer ballots.
__device__ __forceinline__ float kth_smallest_upper(float v, int k, int low_bit = 0) {
  const unsigned bits = __float_as_uint(v);
  unsigned prefix = 0;
  for (int b = 30; b >= low_bit; --b) {  // sign bit is 0
    const unsigned trial = prefix | (1u << b);
    if (__builtin_popcountll(__ballot(bits < trial)) <= k) prefix = trial;  // the answer is >= trial
  }
  return __uint_as_float(prefix | ((1u << low_bit) - 1u));
}
__device__ __forceinline__ double kth_smallest_upper(double v, int k) {
  const unsigned hi = (unsigned)(__double_as_longlong(v) >> 32);
  unsigned prefix = 0;
  for (int b = 30; b >= 0; --b) {
    const unsigned trial = prefix | (1u << b);
    if (__builtin_popcountll(__ballot(hi < trial)) <= k) prefix = trial;
  }
  return __longlong_as_double(((long long)prefix << 32) | 0xffffffffll);  // largest double with these high bits
}

template <typename T, int R>
struct LaneList {
  T d[R];
  int i[R];
  __device__ __forceinline__ void insert(T cd, int ci, int lane) {
    int pos = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) pos += __builtin_popcountll(__ballot(d[r] <= cd));  // equal keys: earlier arrival first
    T carry_d = cd;  // value entering register r at its lane 0 when everything shifts
    int carry_i = ci;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int lp = pos - 64 * r;  // insertion lane inside register r (may be < 0: all shift, >= 64: untouched)
      const T last_d = read_lane(d[r], 63);
      const int last_i = __builtin_amdgcn_readlane(i[r], 63);
      const T sd = shr1_fill(d[r], carry_d);
      const int si = shr1_fill(i[r], carry_i);
      const bool take_new = lane == lp;
      const bool take_shift = lane > lp;
      d[r] = take_new ? cd : (take_shift ? sd : d[r]);
      i[r] = take_new ? ci : (take_shift ? si : i[r]);
      carry_d = last_d;
      carry_i = last_i;
    }
  }
  __device__ __forceinline__ T at(int rank) const {  // rank is wave-uniform
    T v = read_lane(d[0], rank & 63);
#pragma unroll
    for (int r = 1; r < R; ++r) {
      const T o = read_lane(d[r], rank & 63);
      v = (rank >> 6) == r ? o : v;
    }
    return v;
  }
};

// The reference's max-heap (knn_points_cuda.cuh:16-28) with its slots IN THE LANES (one per lane up to k = 64, two up
// to the reference's limit of 100), sifted by the whole wave
// at once instead of level by level.  Replacing the root by v and sifting down moves along the "bigger child" path
// from the root for as long as the child is >= v; on that path every mover takes its bigger child's slot and the first
// non-mover takes v.  So: every lane fetches its two children (four ds_bpermute, one round trip), picks the bigger
// (right only if strictly greater, :21); the path is followed with six dependent v_readlane (leaves point to
// themselves); a lane on the path moves iff its bigger child is not below v (:22 strict; distances never increase
// along the path, so the movers are a prefix of it); the lane on the path that does not move and whose parent does --
// or the root -- receives v.  No LDS heap, no per-level branch: ~350 cycles per insertion against ~1,500 for the
// branchy level-by-level walk that one wave alone executes at a branch every few instructions.
template <typename T, int R>
__device__ __forceinline__ void lane_heap_sift(T (&hd)[R], int (&hi)[R], int size, T vd, int vi, int lane) {
  // slot n = lane n & 63 of register n >> 6 (R = 2: up to 128 slots; slots 64.. have no children below 128: leaves)
  const int c0 = 2 * lane + 1, c1 = c0 + 1;  // children of this lane's register-0 slot
  T d0 = __shfl(hd[0], c0 & 63, 64), d1 = __shfl(hd[0], c1 & 63, 64);
  int i0 = __shfl(hi[0], c0 & 63, 64), i1 = __shfl(hi[0], c1 & 63, 64);
  if constexpr (R == 2) {
    const T e0 = __shfl(hd[1], c0 & 63, 64), e1 = __shfl(hd[1], c1 & 63, 64);
    const int j0 = __shfl(hi[1], c0 & 63, 64), j1 = __shfl(hi[1], c1 & 63, 64);
    d0 = c0 >= 64 ? e0 : d0;
    i0 = c0 >= 64 ? j0 : i0;
    d1 = c1 >= 64 ? e1 : d1;
    i1 = c1 >= 64 ? j1 : i1;
  }
  const bool has0 = c0 < size, has1 = c1 < size;
  const bool right = has1 && (d1 > d0);  // :21 strict
  const int big = has0 ? (right ? c1 : c0) : lane;
  const T cd = right ? d1 : d0;
  const int ci = right ? i1 : i0;
  int p = 0;
  unsigned long long path[R];  // bit n & 63 of word n >> 6: slot n lies on the bigger-child path from the root
  path[0] = 1ull;
  if constexpr (R == 2) path[1] = 0ull;
#pragma unroll
  for (int j = 0; j < 6; ++j) {  // slots 0 .. 126: at most 6 levels below the root
    if constexpr (R == 2) {
      p = p < 64 ? __builtin_amdgcn_readlane(big, p & 63) : p;  // a register-1 slot is a leaf
      if (p < 64) path[0] |= 1ull << p; else path[1] |= 1ull << (p - 64);
    } else {
      p = __builtin_amdgcn_readlane(big, p);
      path[0] |= 1ull << p;
    }
  }
  const bool on_path = (path[0] >> lane) & 1ull;
  const bool moves = on_path && has0 && !(vd > cd);  // :22 strict
  const unsigned long long moved = __ballot(moves);
  const bool parent_moved = lane == 0 || ((moved >> ((lane - 1) >> 1)) & 1ull);
  const bool takes_v = on_path && !moves && parent_moved;
  hd[0] = moves ? cd : (takes_v ? vd : hd[0]);
  hi[0] = moves ? ci : (takes_v ? vi : hi[0]);
  if constexpr (R == 2) {  // slot 64 + lane: never a mover; its parent is slot (63 + lane) / 2 of register 0
    const bool takes_v1 = ((path[1] >> lane) & 1ull) && ((moved >> ((63 + lane) >> 1)) & 1ull);
    hd[1] = takes_v1 ? vd : hd[1];
    hi[1] = takes_v1 ? vi : hi[1];
  }
}

// One slot per lane (size <= 64), no dependent walk at all.  Slot n's sibling is a neighbouring lane (children of p are
// 2p+1 and 2p+2), so "n is the child its parent would descend into" (:21: the right one only if strictly greater) is two
// whole-wave DPP shifts and two compares; the ballot of that, G, describes every bigger-child edge of the heap at once.
// A slot lies on the root's bigger-child path iff G holds it and all its ancestors: one AND + compare against a per-lane
// constant (`anc`, the slot's ancestor set incl. itself -- computed once per replay).  A path slot's path child is the one
// of its two children G holds: ONE pull (ds_bpermute of distance and index) instead of fetching both children, and the
// movers / the slot that takes v follow as in lane_heap_sift.  ~35 instructions and one LDS-crossbar round trip per
// insertion; the six dependent v_readlane steps (VALU -> SGPR -> lane select, four wait states each) are gone.
template <typename T>
__device__ __forceinline__ void lane_heap_sift1(T& hd, int& hi, int size, T vd, int vi, int lane,
                                                unsigned long long anc) {
  const T left = shr1_fill(hd, hd), right = shl1_fill(hd, hd);
  const unsigned long long in_heap = size >= 64 ? ~0ull : ((1ull << size) - 1ull);
  const unsigned long long kOdd = 0xAAAAAAAAAAAAAAAAull;                    // left children (and bit 1...)
  const unsigned long long right_bigger = __ballot(right > hd);            // slot n+1 strictly above slot n
  const unsigned long long above_left = __ballot(hd > left);               // slot n strictly above slot n-1
  // odd n (left child): chosen unless its right sibling exists and is strictly greater; even n >= 2 (right child):
  // chosen iff strictly greater than its left sibling; bit 0 = the root
  const unsigned long long good =
      ((kOdd & ~(right_bigger & (in_heap >> 1))) | (~kOdd & above_left) | 1ull) & in_heap;
  const bool on_path = (good & anc) == anc;
  const int c0 = 2 * lane + 1;
  const bool has0 = c0 < size;
  const int big = ((good >> (c0 & 63)) & 1ull) ? c0 : c0 + 1;  // (lanes without children pull from a harmless lane)
  const T cd = __shfl(hd, big & 63, 64);
  const int ci = __shfl(hi, big & 63, 64);
  const bool moves = on_path && has0 && !(vd > cd);  // :22 strict
  const unsigned long long moved = __ballot(moves);
  const bool parent_moved = lane == 0 || ((moved >> ((lane - 1) >> 1)) & 1ull);
  const bool takes_v = on_path && !moves && parent_moved;
  hd = moves ? cd : (takes_v ? vd : hd);
  hi = moves ? ci : (takes_v ? vi : hi);
}

// Replay on the lane heap (R = 1: k <= 64, R = 2: k <= 127): the heap, the heap sort and the output never leave the
// registers.
template <typename T, int R>
__device__ void knn_exact_replay_lanes(const T* __restrict__ s_cloud, int N, int k, T qx, T qy, T qz,
                                       T* __restrict__ od, int64_t* __restrict__ oi) {
  const int lane = threadIdx.x & 63;
  T hd[R];
  int hi[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    hd[r] = static_cast<T>(1e10);  // knn_points_cuda.cuh:70-73
    hi[r] = 0;
  }
  auto slot_d = [&](int n) {  // n wave-uniform
    if constexpr (R == 2) return n < 64 ? read_lane(hd[0], n & 63) : read_lane(hd[1], n & 63);
    else return read_lane(hd[0], n);
  };
  auto slot_i = [&](int n) {
    if constexpr (R == 2)
      return n < 64 ? __builtin_amdgcn_readlane(hi[0], n & 63) : __builtin_amdgcn_readlane(hi[1], n & 63);
    else return __builtin_amdgcn_readlane(hi[0], n);
  };
  unsigned long long anc = 0;  // this lane's slot and its ancestors in the heap (lane_heap_sift1)
  for (int a = lane;; a = (a - 1) >> 1) {
    anc |= 1ull << a;
    if (a == 0) break;
  }
  auto sift = [&](int size, T vd, int vi) {
    if constexpr (R == 1) lane_heap_sift1<T>(hd[0], hi[0], size, vd, vi, lane, anc);
    else lane_heap_sift<T, R>(hd, hi, size, vd, vi, lane);
  };
  T root_d = static_cast<T>(1e10);
  // Supports come from global memory (L2) in groups of kAhead chunks of 64, the next group requested before this one
  // goes through the heap: late in the scan most chunks hold no candidate, and with one chunk in flight the loop ran at
  // one memory round trip per 64 supports (~20 us of a 59 us replay at N = 4096).
  constexpr int kAhead = 4;
  T cx[kAhead], cy[kAhead], cz[kAhead];
  auto request = [&](int base) {
#pragma unroll
    for (int u = 0; u < kAhead; ++u) {
      const int jn = base + u * 64 + lane;
      const int jj = jn < N ? jn : 0;
      cx[u] = s_cloud[(int64_t)jj * 3 + 0];
      cy[u] = s_cloud[(int64_t)jj * 3 + 1];
      cz[u] = s_cloud[(int64_t)jj * 3 + 2];
    }
  };
  request(0);
  for (int g0 = 0; g0 < N; g0 += 64 * kAhead) {
    T dv[kAhead];
#pragma unroll
    for (int u = 0; u < kAhead; ++u) dv[u] = sq_dist(qx, qy, qz, cx[u], cy[u], cz[u]);
    request(g0 + 64 * kAhead);
#pragma unroll
    for (int u = 0; u < kAhead; ++u) {
      const int c0 = g0 + u * 64;
      const bool valid = c0 + lane < N;
      const T d = dv[u];
      unsigned long long m = __ballot(valid && d < root_d);  // :80 strict
      while (m) {
        const int l = __builtin_ffsll((long long)m) - 1;
        sift(k, read_lane(d, l), c0 + l);
        root_d = read_lane(hd[0], 0);
        m = __ballot(valid && d < root_d) & ~((2ull << l) - 1ull);
      }
    }
  }
  for (int i = k - 1; i > 0; --i) {  // heap_sort :30-37: root <-> slot i, then sift the old slot i down a heap of i
    const T top_d = slot_d(0), last_d = slot_d(i);
    const int top_i = slot_i(0), last_i = slot_i(i);
    if (lane == (i & 63)) {
      if (R == 1 || i < 64) {
        hd[0] = top_d;
        hi[0] = top_i;
      } else {
        hd[R - 1] = top_d;
        hi[R - 1] = top_i;
      }
    }
    sift(i, last_d, last_i);
  }
  if (lane < k) {  // :88-91
    od[lane] = sqrt(hd[0]);
    oi[lane] = (int64_t)hi[0];
  }
  if constexpr (R == 2) {
    if (lane + 64 < k) {
      od[lane + 64] = sqrt(hd[1]);
      oi[lane + 64] = (int64_t)hi[1];
    }
  }
}

// Exact replay of the reference heap for ONE query by the whole wave (used only when the fast path sees equal
// distances among the k+1 nearest, where the heap's mechanics decide identity and order): supports are read from
// global memory (L2-resident) one chunk ahead, candidates that beat the root are replayed in index order.  A replay is
// a serial chain (~190 insertions at N = 4096, k = 32) on one wave and the launch ends when it does
// (tools/knn_tail_probe.py), hence the lane heap: with the heap in LDS and a level-by-level walk it took ~130 us
// (k = 32) / ~220 us (k = 100).  (A lane heap walked level by level with v_readlane / v_writelane was slower than
// LDS: every step a VALU -> SGPR -> VALU hop.)
template <typename T>
__device__ void knn_exact_replay(const T* __restrict__ s_cloud, int N, int k, T qx, T qy, T qz, T* __restrict__ od,
                                 int64_t* __restrict__ oi) {
  static_assert(V3D_KNN_MAX_K <= 127, "two slots per lane");
  if (k <= 64) knn_exact_replay_lanes<T, 1>(s_cloud, N, k, qx, qy, qz, od, oi);  // wave-uniform
  else knn_exact_replay_lanes<T, 2>(s_cloud, N, k, qx, qy, qz, od, oi);
}

// ---- pack mode (ops/knn.py:110-145): clouds of different sizes concatenated ------------------------------------------
// The reference pads every cloud to the longest with points at 1e10 and searches the padded batch; here a workgroup
// serves 16 consecutive queries of ONE cloud against that cloud's own supports.  A one-workgroup kernel turns the two
// length vectors into a plan (offsets + the first workgroup of every cloud; device side, no host synchronisation); the
// search kernels are the batch kernels with the cloud looked up in the plan instead of blockIdx.y.
struct KnnPackPlan {
  long long q_off, s_off;
  int q_len, s_len, wg_first, pad_;
};

__global__ __launch_bounds__(64) void knn_pack_plan_kernel(const int64_t* __restrict__ q_lengths,
                                                           const int64_t* __restrict__ s_lengths,
                                                           KnnPackPlan* __restrict__ plan, int B, long long total_q,
                                                           long long total_s, int queries_per_wg) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  long long qo = 0, so = 0;
  int wg = 0;
  for (int b = 0; b <= B; ++b) {
    // lengths that overrun the totals are cut (the caller promised they sum to the totals): no access leaves the arrays
    long long ql = b < B ? (long long)q_lengths[b] : 0, sl = b < B ? (long long)s_lengths[b] : 0;
    ql = ql < 0 ? 0 : (ql > total_q - qo ? total_q - qo : ql);
    sl = sl < 0 ? 0 : (sl > total_s - so ? total_s - so : sl);
    KnnPackPlan p;
    p.q_off = qo;
    p.s_off = so;
    p.q_len = (int)ql;
    p.s_len = (int)sl;
    p.wg_first = wg;
    p.pad_ = 0;
    plan[b] = p;
    qo += ql;
    so += sl;
    wg += (int)((ql + queries_per_wg - 1) / queries_per_wg);
  }
}

struct KnnWhere {
  int M, N, blk;           // queries / supports of the cloud, this workgroup's position among the cloud's workgroups
  int64_t q_row0, s_row0;  // first query / support row of the cloud
  bool live;
};
template <bool PACK>
__device__ __forceinline__ KnnWhere knn_where(int M, int N, const KnnPackPlan* __restrict__ plan, int B) {
  KnnWhere w;
  if constexpr (!PACK) {
    w.M = M;
    w.N = N;
    w.blk = blockIdx.x;
    w.q_row0 = (int64_t)blockIdx.y * M;
    w.s_row0 = (int64_t)blockIdx.y * N;
    w.live = true;
  } else {
    const int id = blockIdx.x;
    w.live = id < plan[B].wg_first;
    // last cloud whose first workgroup is <= id: clouds without queries own no workgroup and share their start with the
    // next cloud, which the bisection then prefers
    int lo = 0, hi = B;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (plan[mid].wg_first <= id) lo = mid; else hi = mid;
    }
    const KnnPackPlan p = plan[lo];
    w.M = p.q_len;
    w.N = p.s_len;
    w.blk = id - p.wg_first;
    w.q_row0 = p.q_off;
    w.s_row0 = p.s_off;
  }
  return w;
}

// Stage supports [t0, t0 + tcount) of a cloud and pad the tile's last chunk of 64 with points at +inf (their distance is
// +inf or NaN: below no bound), so that the kNN loops read whole chunks without a validity test.  Whole workgroup.
template <typename T>
__device__ __forceinline__ void knn_stage_padded(const T* __restrict__ s_cloud, int t0, int tcount, T* tile) {
  __syncthreads();
  stage_supports_aos(s_cloud, t0, tcount, tile);
  const int padded = (tcount + 63) & ~63;
  for (int i = tcount * 3 + (int)threadIdx.x; i < padded * 3; i += kQueryThreads) tile[i] = static_cast<T>(INFINITY);
  __syncthreads();
}

// Lane-minimum bound of a cloud for the QPW queries of a wave (every wave of the workgroup calls it: the tiles are
// staged by the whole workgroup): capk[g] = an upper bound of query g's (k+1)-th smallest squared distance.
template <typename T, int QPW, int TS>
__device__ __forceinline__ void knn_lane_min_bound(const T* __restrict__ s_cloud, int N, int k, const T (&qx)[QPW],
                                                   const T (&qy)[QPW], const T (&qz)[QPW], T* tile, int lane,
                                                   T (&capk)[QPW]) {
  static_assert(QPW % 2 == 0, "queries are taken in pairs");
  T lmin[QPW];
#pragma unroll
  for (int g = 0; g < QPW; ++g) lmin[g] = INFINITY;
  for (int t0 = 0; t0 < N; t0 += TS) {
    const int tcount = min(TS, N - t0);
    knn_stage_padded(s_cloud, t0, tcount, tile);
    for (int c0 = 0; c0 < tcount; c0 += 64) {
      const int j = c0 + lane;
      const T sx = tile[j * 3 + 0], sy = tile[j * 3 + 1], sz = tile[j * 3 + 2];
#pragma unroll
      for (int g = 0; g < QPW; g += 2) {
        T d0, d1;
        sq_dist_pair(qx[g], qy[g], qz[g], qx[g + 1], qy[g + 1], qz[g + 1], sx, sy, sz, d0, d1);
        lmin[g] = d0 < lmin[g] ? d0 : lmin[g];  // +inf and NaN never replace a minimum
        lmin[g + 1] = d1 < lmin[g + 1] ? d1 : lmin[g + 1];
      }
    }
  }
#pragma unroll
  for (int g = 0; g < QPW; ++g) capk[g] = kth_smallest_upper(lmin[g], k);
}

// Fast path: one wave serves QPW queries against LDS-staged supports.  Per query the k+1 best so far sit sorted in a
// lane-distributed list; a chunk of 64 supports is filtered with one compare + ballot against the (k+1)-th best and
// only the survivors are inserted (whole-wave DPP shift).  When the final k+1 distances are pairwise different the
// reference heap can only produce this very list; otherwise the query is replayed through the exact heap.
template <typename T, int R, int QPW, int TS, bool PACK = false>
__global__ __launch_bounds__(kQueryThreads) void knn_points_kernel(const T* __restrict__ q, const T* __restrict__ s,
                                                                   T* __restrict__ out_d, int64_t* __restrict__ out_i,
                                                                   int M_arg, int N_arg, int k,
                                                                   const KnnPackPlan* __restrict__ plan, int B, int lex) {
  __shared__ __align__(16) T tile[TS * 3];
  const KnnWhere where = knn_where<PACK>(M_arg, N_arg, plan, B);
  if (!where.live) return;  // the whole workgroup
  const int M = where.M, N = where.N;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q_first = (where.blk * kWavesPerBlock + wave) * QPW;
  const T* s_cloud = s + where.s_row0 * 3;
  const T sentinel = static_cast<T>(1e10);  // knn_points_cuda.cuh:71: nothing >= 1e10 ever enters

  T qx[QPW], qy[QPW], qz[QPW], tau[QPW], capk[QPW];
  LaneList<T, R> list[QPW];
#pragma unroll
  for (int g = 0; g < QPW; ++g) {
    const int qi = q_first + g;
    const bool live = qi < M;
    const T* qp = q + (where.q_row0 + (live ? qi : 0)) * 3;
    qx[g] = qp[0];
    qy[g] = qp[1];
    qz[g] = qp[2];
    tau[g] = live ? sentinel : static_cast<T>(-1);  // dead slot: nothing ever beats -1
    capk[g] = INFINITY;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      list[g].d[r] = sentinel;
      list[g].i[r] = 0;
    }
  }

  // Pre-pass over the WHOLE cloud: every lane takes the minimum distance of its own supports (j = lane mod 64); the
  // (k+1)-th smallest of those 64 minima bounds the (k+1)-th smallest distance from above (64 distinct supports attain
  // them) -- at N = 4096 about 1 % of the supports lie under it, so the search proper admits a few dozen candidates
  // instead of the ~k ln(N/k) + stale ones of a bound that tightens as it goes.  Everything <= the bound is still
  // admitted, in index order, so the k+1 best and their order among equals are unchanged.  One more distance pass
  // (branch-free, ~6 VALU per query and chunk) for far fewer whole-wave insertions / merges.
  const bool prepass = R == 1 && N >= 256;
  if (prepass) knn_lane_min_bound<T, QPW, TS>(s_cloud, N, k, qx, qy, qz, tile, lane, capk);
  for (int t0 = 0; t0 < N; t0 += TS) {
    const int tcount = min(TS, N - t0);
    if (!(prepass && N <= TS)) knn_stage_padded(s_cloud, t0, tcount, tile);  // one tile: still staged from the pre-pass
    // common chunk (no candidate for any query): packed distances, one compare per query against min(tau, capk)
    T bound[QPW];
#pragma unroll
    for (int g = 0; g < QPW; ++g) bound[g] = tau[g] < capk[g] ? tau[g] : capk[g];
    for (int c0 = 0; c0 < tcount; c0 += 64) {
      const int j = c0 + lane;
      const T sx = tile[j * 3 + 0], sy = tile[j * 3 + 1], sz = tile[j * 3 + 2];  // tail lanes read the +inf padding
      const bool first_chunk = (t0 == 0 && c0 == 0);
      T dq[QPW];
#pragma unroll
      for (int g = 0; g < QPW; g += 2)
        sq_dist_pair(qx[g], qy[g], qz[g], qx[g + 1], qy[g + 1], qz[g + 1], sx, sy, sz, dq[g], dq[g + 1]);
      unsigned long long pm[QPW], any = 0ull;
#pragma unroll
      for (int g = 0; g < QPW; ++g) {
        pm[g] = __ballot(dq[g] <= bound[g]);
        any |= pm[g];
      }
      if (any == 0ull) continue;  // one scalar test per chunk on the common path
#pragma unroll
      for (int g = 0; g < QPW; ++g) {
        const T d = dq[g];
        if (pm[g] == 0ull) continue;
        if (first_chunk && tau[g] == sentinel && !(capk[g] < INFINITY)) {
          // empty list: rank-sort the whole chunk straight into it instead of 64 serial insertions
          const bool real = d < sentinel;  // padding is +inf
          const T dm = real ? d : sentinel;
          const int im = real ? j : 0;
          int rank = 0;
          for (int o = 0; o < 64; ++o) {
            const T dv = read_lane(dm, o);
            rank += (dv < dm || (dv == dm && o < lane)) ? 1 : 0;
          }
          list[g].d[0] = lane_permute(dm, rank);
          list[g].i[0] = lane_permute(im, rank);
          tau[g] = list[g].at(k);
          bound[g] = tau[g] < capk[g] ? tau[g] : capk[g];
          continue;
        }
        unsigned long long m = __ballot(d < tau[g] && d <= capk[g]);  // padding (+inf) and NaN pass neither
        while (m) {
          const int l = __builtin_ffsll((long long)m) - 1;
          list[g].insert(read_lane(d, l), t0 + c0 + l, lane);
          tau[g] = list[g].at(k);
          m = __ballot(d < tau[g] && d <= capk[g]) & ~((2ull << l) - 1ull);
        }
        bound[g] = tau[g] < capk[g] ? tau[g] : capk[g];
      }
    }
  }

#pragma unroll
  for (int g = 0; g < QPW; ++g) {
    if (q_first + g >= M) continue;
    T* od = out_d + (where.q_row0 + q_first + g) * k;
    int64_t* oi = out_i + (where.q_row0 + q_first + g) * k;
    // equal real distances among ranks 0..k -> the heap's mechanics matter
    unsigned long long ties = 0ull;
    T prev_last = static_cast<T>(-1);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const T below = shr1_fill(list[g].d[r], prev_last);
      const int rank = 64 * r + lane;
      ties |= __ballot(rank >= 1 && rank <= k && list[g].d[r] == below && list[g].d[r] < sentinel);
      prev_last = read_lane(list[g].d[r], 63);
    }
    if (!lex && ties != 0ull) {  // (lex: pykeops' order -- ascending (d2, index) -- is the list itself)
      knn_exact_replay(s_cloud, N, k, qx[g], qy[g], qz[g], od, oi);
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int rank = 64 * r + lane;
        if (rank < k) {
          od[rank] = sqrt(list[g].d[r]);  // :90
          oi[rank] = (int64_t)list[g].i[r];
        }
      }
    }
  }
}

// fp32, k + 1 <= 64: the same search with the candidates batched.  The k+1 best so far are kept as 64-bit keys
// (d2 bits << 32 | support index: d2 >= +0, so the unsigned order is the (distance, arrival) order of LaneList),
// ascending across the lanes; survivors of the filter are compacted into a 64-slot LDS buffer per (wave, query) and a
// full buffer -- or the last, partial one -- is merged by a bitonic network (wave_sort.h) instead of one whole-wave
// insertion per survivor.  A survivor admitted against a stale bound is dropped by the merge, so the list is the same.
template <int QPW, int TS, bool PACK = false>
__global__ __launch_bounds__(kQueryThreads) void knn_points_merge_kernel(const float* __restrict__ q,
                                                                         const float* __restrict__ s,
                                                                         float* __restrict__ out_d,
                                                                         int64_t* __restrict__ out_i, int M_arg,
                                                                         int N_arg, int k,
                                                                         const KnnPackPlan* __restrict__ plan, int B,
                                                                         int lex) {
  __shared__ __align__(16) float tile[TS * 3];
  __shared__ wkey_t hits[kWavesPerBlock][QPW][64];
  const KnnWhere where = knn_where<PACK>(M_arg, N_arg, plan, B);
  if (!where.live) return;  // the whole workgroup
  const int M = where.M, N = where.N;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q_first = (where.blk * kWavesPerBlock + wave) * QPW;
  const float* s_cloud = s + where.s_row0 * 3;
  const float sentinel = 1e10f;  // knn_points_cuda.cuh:71: nothing >= 1e10 ever enters
  const wkey_t sent_key = (wkey_t)__float_as_uint(sentinel) << 32;  // (1e10, index 0): the reference's empty slot

  float qx[QPW], qy[QPW], qz[QPW], capk[QPW];
  wkey_t kept[QPW], worst[QPW];
  int pending[QPW];
#pragma unroll
  for (int g = 0; g < QPW; ++g) {
    const int qi = q_first + g;
    const bool live = qi < M;
    const float* qp = q + (where.q_row0 + (live ? qi : 0)) * 3;
    qx[g] = qp[0];
    qy[g] = qp[1];
    qz[g] = qp[2];
    kept[g] = sent_key;
    worst[g] = live ? sent_key : 0ull;  // dead slot: nothing ever beats key 0
    capk[g] = INFINITY;
    pending[g] = 0;
  }

  // same whole-cloud pre-pass as knn_points_kernel: an upper bound of the (k+1)-th distance before the search proper
  const bool prepass = N >= 256;
  if (prepass) knn_lane_min_bound<float, QPW, TS>(s_cloud, N, k, qx, qy, qz, tile, lane, capk);
  for (int t0 = 0; t0 < N; t0 += TS) {
    const int tcount = min(TS, N - t0);
    if (!(prepass && N <= TS)) knn_stage_padded(s_cloud, t0, tcount, tile);  // one tile: still staged from the pre-pass
    // The common chunk holds no candidate for any of the wave's queries: it costs the packed distances, ONE float
    // compare per query against min(pre-pass bound, distance of the current (k+1)-th key) and a scalar test.  Only a
    // chunk that passes builds keys and applies the exact (distance, index) test.
    float bound[QPW];
#pragma unroll
    for (int g = 0; g < QPW; ++g) bound[g] = fminf(capk[g], __uint_as_float((unsigned)(worst[g] >> 32)));
    for (int c0 = 0; c0 < tcount; c0 += 64) {
      const int j = c0 + lane;
      const float sx = tile[j * 3 + 0], sy = tile[j * 3 + 1], sz = tile[j * 3 + 2];  // tail lanes read the +inf padding
      float dq[QPW];
#pragma unroll
      for (int g = 0; g < QPW; g += 2)
        sq_dist_pair(qx[g], qy[g], qz[g], qx[g + 1], qy[g + 1], qz[g + 1], sx, sy, sz, dq[g], dq[g + 1]);
      unsigned long long pm[QPW], any = 0ull;
#pragma unroll
      for (int g = 0; g < QPW; ++g) {
        pm[g] = __ballot(dq[g] <= bound[g]);
        any |= pm[g];
      }
      if (any == 0ull) continue;  // one scalar test per chunk on the common path
#pragma unroll
      for (int g = 0; g < QPW; ++g) {
        const float d = dq[g];
        if (pm[g] == 0ull) continue;
        const wkey_t key = ((wkey_t)__float_as_uint(d) << 32) | (unsigned)(t0 + j);
        // padding (+inf) and NaN patterns lie above the sentinel's, i.e. above every possible worst[g]
        const bool take = key < worst[g] && d <= capk[g];
        const unsigned long long m = __ballot(take);
        if (m == 0ull) continue;
        wkey_t* buf = hits[wave][g];
        const int n = __builtin_popcountll(m);
        const int dest = pending[g] + wave_prefix_count(m);
        if (take && dest < 64) buf[dest] = key;
        if (pending[g] + n >= 64) {
          kept[g] = wave_merge_buffer(kept[g], buf, 64, lane);
          worst[g] = wave_lane64(kept[g], k);
          bound[g] = fminf(capk[g], __uint_as_float((unsigned)(worst[g] >> 32)));
          if (take && dest >= 64) buf[dest - 64] = key;
          pending[g] += n - 64;
        } else {
          pending[g] += n;
        }
      }
    }
  }

#pragma unroll
  for (int g = 0; g < QPW; ++g) {
    if (q_first + g >= M) continue;
    if (pending[g] > 0) kept[g] = wave_merge_buffer(kept[g], hits[wave][g], pending[g], lane);
    float* od = out_d + (where.q_row0 + q_first + g) * k;
    int64_t* oi = out_i + (where.q_row0 + q_first + g) * k;
    const float d = __uint_as_float((unsigned)(kept[g] >> 32));
    const float below = shr1_fill(d, -1.0f);
    // equal real distances among ranks 0..k -> the heap's mechanics matter
    const unsigned long long ties = __ballot(lane >= 1 && lane <= k && d == below && d < sentinel);
    if (!lex && ties != 0ull) {
      knn_exact_replay_lanes<float, 1>(s_cloud, N, k, qx[g], qy[g], qz[g], od, oi);  // k + 1 <= 64 here
    } else if (lane < k) {
      od[lane] = sqrtf(d);  // :90
      oi[lane] = (int64_t)(unsigned)kept[g];
    }
  }
}

// ---- knn_points on a per-workgroup cell list (fp32, k + 1 <= 64, 2048 <= N <= 4096) -----------------------------------
// The brute-force kernels above evaluate every (query, support) pair twice (lane-minimum pre-pass + search): ~53 us of
// VALU issue at C2 before any candidate is handled.  Here every workgroup (QB queries of one cloud, 512 threads) first
// sorts the cloud's supports into a G^3 uniform grid over their bounding box, entirely in LDS (count -> scan -> scatter,
// integer LDS atomics; the three_nn_grid_kernel's builder; ~9 us at N = 4096), then a WAVE serves one query at a time:
// the 27 cells around it are 9 runs of consecutive records.  A first walk keeps every lane's minimum distance and bounds
// the (k+1)-th distance from above, a second one batches the candidates under that bound in a 64-slot LDS buffer, and a
// rank sort (or, once a list exists, the bitonic network of wave_sort.h as in knn_points_merge_kernel) turns them into
// the sorted list.  The search is exact, not approximate: after the 27 cells (and after every further ring) the
// (k+1)-th distance is compared with the distance to the nearest face of the explored box that still has cells behind
// it (minus a rounding margin), and the box grows ring by ring until it is below it or the grid is exhausted.  The
// result is the k+1 smallest (d2, index) keys of the whole cloud -- the list the other kernels keep -- so ties among
// them go to the same exact heap replay, and outputs are bit-identical.  G is chosen on the host (knn_points below) so
// that the ball of the k + 1 nearest fits the 27 cells for most queries of a roughly uniform cloud; a clustered cloud
// only costs more rings.  Measured at C2 (B 32, N 4096, M 1024): k = 16 113 -> 47 us, k = 32 129 -> 58 us without and
// 89 us with one tie replay in the launch (profiles/r03_knn_cell_list_sweep.txt).
constexpr int kKgThreads = 512;
constexpr int kKgWaves = kKgThreads / 64;
constexpr int kKgMaxG = 10;
constexpr int kKgMaxCells = kKgMaxG * kKgMaxG * kKgMaxG;
constexpr int kKgMaxN = 4096;
constexpr int kKgMinN = 2048;
constexpr int kKgPer = kKgMaxN / kKgThreads;  // supports per thread in the builder

__global__ __launch_bounds__(kKgThreads) void knn_points_grid_kernel(const float* __restrict__ q,
                                                                     const float* __restrict__ s,
                                                                     float* __restrict__ out_d,
                                                                     int64_t* __restrict__ out_i, int M, int N, int k,
                                                                     int G, int QB, int lex) {
  extern __shared__ float4 kg_sorted[];  // N records: x, y, z, index bits -- ordered by cell (x fastest)
  __shared__ int cell_start[kKgMaxCells + 2];
  __shared__ int cursor[kKgMaxCells];
  __shared__ float red[6][kKgWaves];
  __shared__ int wave_sum[kKgWaves];
  __shared__ wkey_t hits[kKgWaves][64];
  __shared__ int next_query;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y;
  const float* s_cloud = s + (int64_t)b * N * 3;
  const int NC = G * G * G;
  if (tid == 0) next_query = 0;

  // ---- bounding box of the supports ----------------------------------------------------------------------------------
  float sx[kKgPer], sy[kKgPer], sz[kKgPer];
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
  for (int u = 0; u < kKgPer; ++u) {
    const int i = u * kKgThreads + tid;
    const int ii = i < N ? i : 0;
    sx[u] = s_cloud[ii * 3 + 0];
    sy[u] = s_cloud[ii * 3 + 1];
    sz[u] = s_cloud[ii * 3 + 2];
    if (i < N) {  // fminf / fmaxf drop NaN: a non-finite support does not poison the box
      lo[0] = fminf(lo[0], sx[u]); hi[0] = fmaxf(hi[0], sx[u]);
      lo[1] = fminf(lo[1], sy[u]); hi[1] = fmaxf(hi[1], sy[u]);
      lo[2] = fminf(lo[2], sz[u]); hi[2] = fmaxf(hi[2], sz[u]);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    for (int d = 32; d >= 1; d >>= 1) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], d, 64));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], d, 64));
    }
    if (lane == 0) {
      red[a][wave] = lo[a];
      red[3 + a][wave] = hi[a];
    }
  }
  for (int i = tid; i < NC; i += kKgThreads) cell_start[i] = 0;
  __syncthreads();
  float cell[3], inv[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    lo[a] = red[a][0];
    hi[a] = red[3 + a][0];
    for (int w = 1; w < kKgWaves; ++w) {
      lo[a] = fminf(lo[a], red[a][w]);
      hi[a] = fmaxf(hi[a], red[3 + a][w]);
    }
    const float ext = hi[a] - lo[a];
    cell[a] = (ext > 0.f && ext < INFINITY) ? ext / (float)G : 1.0f;  // flat or unbounded axis: one layer of cells
    inv[a] = 1.0f / cell[a];
  }
  auto coord = [&](float v, int a) {  // NaN and anything outside the box clamp to the border cells
    const float c = floorf((v - lo[a]) * inv[a]);
    return c >= (float)(G - 1) ? G - 1 : (c > 0.f ? (int)c : 0);
  };

  // ---- counting sort of the supports by cell (LDS only) --------------------------------------------------------------
  int cid[kKgPer];
#pragma unroll
  for (int u = 0; u < kKgPer; ++u) {
    const int i = u * kKgThreads + tid;
    cid[u] = (coord(sz[u], 2) * G + coord(sy[u], 1)) * G + coord(sx[u], 0);
    if (i < N) atomicAdd(&cell_start[cid[u]], 1);
  }
  __syncthreads();
  {  // exclusive scan of the NC <= 1000 counts: thread t owns cells 2t, 2t+1
    const int c0 = 2 * tid < NC ? cell_start[2 * tid] : 0, c1 = 2 * tid + 1 < NC ? cell_start[2 * tid + 1] : 0;
    const int mine = c0 + c1;
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d, 64);
      if (lane >= d) incl += t;
    }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    int before = incl - mine;
    for (int w = 0; w < wave; ++w) before += wave_sum[w];
    if (2 * tid < NC) {
      cell_start[2 * tid] = before;
      cursor[2 * tid] = before;
    }
    if (2 * tid + 1 < NC) {
      cell_start[2 * tid + 1] = before + c0;
      cursor[2 * tid + 1] = before + c0;
    }
    if (tid == 0) cell_start[NC] = N;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kKgPer; ++u) {
    const int i = u * kKgThreads + tid;
    if (i < N) {
      const int pos = atomicAdd(&cursor[cid[u]], 1);
      kg_sorted[pos] = make_float4(sx[u], sy[u], sz[u], __int_as_float(i));
    }
  }
  __syncthreads();

  // ---- queries: one per wave at a time -------------------------------------------------------------------------------
  const float sentinel = 1e10f;  // knn_points_cuda.cuh:71: nothing >= 1e10 ever enters
  const wkey_t sent_key = (wkey_t)__float_as_uint(sentinel) << 32;
  wkey_t* buf = hits[wave];
  // Queries are handed out through an LDS counter, one ahead: the coordinates of a wave's next query arrive while it
  // serves the current one, and a wave held up by a tie replay leaves the rest of the block to the others.
  const int q_begin = blockIdx.x * QB, q_count = min(M, q_begin + QB) - q_begin;
  auto grab = [&]() {
    int v = 0;
    if (lane == 0) v = atomicAdd(&next_query, 1);
    return __builtin_amdgcn_readfirstlane(v);
  };
  auto fetch = [&](int t, float& x, float& y, float& z) {
    const float* qp = q + ((int64_t)b * M + q_begin + min(t, q_count - 1)) * 3;
    x = qp[0];
    y = qp[1];
    z = qp[2];
  };
  int cur = grab();
  float nx, ny, nz;
  fetch(cur, nx, ny, nz);
  while (cur < q_count) {
    const int qi = q_begin + cur;
    // wave-uniform values in scalar registers: the cell walk below then runs on scalar loop bounds
    const float qx = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(nx)));
    const float qy = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(ny)));
    const float qz = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(nz)));
    cur = grab();
    fetch(cur, nx, ny, nz);
    const float qv[3] = {qx, qy, qz};
    const int cc[3] = {__builtin_amdgcn_readfirstlane(coord(qx, 0)), __builtin_amdgcn_readfirstlane(coord(qy, 1)),
                       __builtin_amdgcn_readfirstlane(coord(qz, 2))};
    wkey_t kept = sent_key, worst = sent_key;
    int pending = 0;
    bool kept_empty = true;
    // 64 candidate keys, one per lane, `take` marks the live ones: batched in the wave's LDS buffer, merged when it fills
    auto offer_key = [&](wkey_t key, bool take) {
      const unsigned long long m = __ballot(take);
      if (m == 0ull) return;
      const int n = __builtin_popcountll(m);
      const int dest = pending + wave_prefix_count(m);
      if (take && dest < 64) buf[dest] = key;
      if (pending + n >= 64) {
        kept = wave_merge_buffer(kept, buf, 64, lane);
        kept_empty = false;
        worst = wave_lane64(kept, k);
        if (take && dest >= 64) buf[dest - 64] = key;
        pending += n - 64;
      } else {
        pending += n;
      }
    };
    auto offer = [&](int j, bool valid, float cap) {  // records j (lane-varying) of the sorted list
      const float4 p = kg_sorted[valid ? j : 0];
      const float d = sq_dist(qx, qy, qz, p.x, p.y, p.z);
      const wkey_t key = ((wkey_t)__float_as_uint(d) << 32) | (unsigned)__float_as_int(p.w);
      offer_key(key, valid && key < worst && d <= cap);  // NaN / +inf patterns lie above the sentinel's
    };
    auto scan_run = [&](int j0_, int j1_) {  // wave-uniform bounds (read from LDS: moved to scalar registers)
      const int j0 = __builtin_amdgcn_readfirstlane(j0_), j1 = __builtin_amdgcn_readfirstlane(j1_);
      for (int j = j0; j < j1; j += 64) offer(j + lane, j + lane < j1, INFINITY);
    };
    // The 27 cells are 9 runs of consecutive records; their bounds are fetched by lanes 0..8 in one LDS round trip and
    // moved to scalar registers.  Run r is dealt to the lanes from lane (records before it) mod 64 on, as if the runs
    // were concatenated, so every lane sees its share of the candidates.  First walk: the distances (one step of 64 per
    // run stays in registers) and every lane's minimum.  The (k+1)-th smallest of the 64 minima is attained by k+1
    // distinct supports, i.e. bounds the (k+1)-th distance from above (knn_lane_min_bound's argument; the search stops
    // at bit 17 of the pattern and rounds up: <= 1.6 % looser, 14 ballots instead of 31), and admits little more than
    // k+1 candidates to the second walk: one merge per query.
    {
      int rb = 0, rn = 0;
      if (lane < 9) {
        const int x0 = max(cc[0] - 1, 0), x1 = min(cc[0] + 1, G - 1);
        const int zz = cc[2] - 1 + lane / 3, yy = cc[1] - 1 + lane % 3;
        if (zz >= 0 && zz < G && yy >= 0 && yy < G) {
          const int c0 = (zz * G + yy) * G;
          rb = cell_start[c0 + x0];
          rn = cell_start[c0 + x1 + 1] - rb;
        }
      }
      int beg[9], len[9], pre = 0;
      float dr[9];
      unsigned wr[9];
      float lmin = INFINITY;
      bool long_run = false;
#pragma unroll
      for (int r = 0; r < 9; ++r) {
        beg[r] = __builtin_amdgcn_readlane(rb, r);
        len[r] = __builtin_amdgcn_readlane(rn, r);
        const int jr = (lane - pre) & 63;
        const bool valid = jr < len[r];
        const float4 p = kg_sorted[valid ? beg[r] + jr : 0];
        const float d = sq_dist(qx, qy, qz, p.x, p.y, p.z);
        dr[r] = valid ? d : __uint_as_float(0x7fc00000u);  // an idle lane's entry never passes a comparison
        wr[r] = (unsigned)__float_as_int(p.w);
        lmin = (valid && d < lmin) ? d : lmin;  // +inf and NaN never replace a minimum
        for (int j = 64; j < len[r]; j += 64) {  // (a crowded run: more than 64 records)
          const int jj = j + jr;
          const float4 p2 = kg_sorted[jj < len[r] ? beg[r] + jj : 0];
          const float d2 = sq_dist(qx, qy, qz, p2.x, p2.y, p2.z);
          lmin = (jj < len[r] && d2 < lmin) ? d2 : lmin;
          long_run = true;
        }
        pre += len[r];
      }
      const float capk = pre > 64 ? kth_smallest_upper(lmin, k, 17) : INFINITY;
#pragma unroll
      for (int r = 0; r < 9; ++r) {
        const wkey_t key = ((wkey_t)__float_as_uint(dr[r]) << 32) | wr[r];
        offer_key(key, key < worst && dr[r] <= capk);
      }
      if (long_run) {
        int pre2 = 0;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
          const int jr = (lane - pre2) & 63;
          for (int j = 64; j < len[r]; j += 64) offer(beg[r] + j + jr, j + jr < len[r], capk);
          pre2 += len[r];
        }
      }
    }
    int p0[3], p1[3];  // explored box
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      p0[a] = max(cc[a] - 1, 0);
      p1[a] = min(cc[a] + 1, G - 1);
    }
    for (int R = 2;; ++R) {
      if (pending > 0) {  // the bound below needs the true (k+1)-th key
        if (kept_empty) {
          const wkey_t h = wave_rank_sort(buf, pending, lane);
          kept = h < sent_key ? h : sent_key;
        } else {
          kept = wave_merge_buffer(kept, buf, pending, lane);
        }
        kept_empty = false;
        worst = wave_lane64(kept, k);
        pending = 0;
      }
      // nearest face of the explored box that still has cells behind it
      float bound = INFINITY;
      bool open = false;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float rel = qv[a] - lo[a];
        const float margin = 1e-5f * fmaxf(hi[a] - lo[a], fabsf(rel));
        if (p0[a] > 0) {
          open = true;
          bound = fminf(bound, fmaxf(rel - (float)p0[a] * cell[a] - margin, 0.f));
        }
        if (p1[a] < G - 1) {
          open = true;
          bound = fminf(bound, fmaxf((float)(p1[a] + 1) * cell[a] - rel - margin, 0.f));
        }
      }
      const float dk = __uint_as_float((unsigned)(worst >> 32));
      if (!open || dk < bound * bound) break;  // (NaN bounds compare false: the search widens to the whole grid)
      int r0[3], r1[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        r0[a] = max(cc[a] - R, 0);
        r1[a] = min(cc[a] + R, G - 1);
      }
      for (int z = r0[2]; z <= r1[2]; ++z) {
        for (int y = r0[1]; y <= r1[1]; ++y) {
          const int row = (z * G + y) * G;
          const bool inner = z >= p0[2] && z <= p1[2] && y >= p0[1] && y <= p1[1];
          if (!inner) {
            scan_run(cell_start[row + r0[0]], cell_start[row + r1[0] + 1]);
          } else {
            if (r0[0] < p0[0]) scan_run(cell_start[row + r0[0]], cell_start[row + p0[0]]);
            if (r1[0] > p1[0]) scan_run(cell_start[row + p1[0] + 1], cell_start[row + r1[0] + 1]);
          }
        }
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        p0[a] = r0[a];
        p1[a] = r1[a];
      }
    }
    float* od = out_d + ((int64_t)b * M + qi) * k;
    int64_t* oi = out_i + ((int64_t)b * M + qi) * k;
    const float d = __uint_as_float((unsigned)(kept >> 32));
    const float below = shr1_fill(d, -1.0f);
    // equal real distances among ranks 0..k -> the heap's mechanics matter
    const unsigned long long ties = __ballot(lane >= 1 && lane <= k && d == below && d < sentinel);
    if (!lex && ties != 0ull) {
      knn_exact_replay_lanes<float, 1>(s_cloud, N, k, qx, qy, qz, od, oi);
    } else if (lane < k) {
      od[lane] = sqrtf(d);  // :90
      oi[lane] = (int64_t)(unsigned)kept;
    }
  }
}

template <typename T>
static int knn_points(const T* q, const T* s, T* od, int64_t* oi, int B, int M, int N, int k, int device,
                      void* stream_, int lex = 0) {
  const char* what = lex ? "knn (index order)" : "knn_points";
  V3D_REQUIRE(B >= 0 && M >= 0 && N >= 0 && k >= 0, what, "negative dimension");
  if (k > V3D_KNN_MAX_K) {
    set_last_error("%s: num_neighbors=%d exceeds %d (reference: best_dist[100])", what, k, V3D_KNN_MAX_K);
    return V3D_ERR_K_TOO_LARGE;
  }
  if (B == 0 || M == 0 || k == 0) return 0;
  V3D_REQUIRE(q && od && oi && (N == 0 || s), what, "null pointer");
  if (B > kMaxGridYZ) {
    set_last_error("%s: B <= 65535 required", what);
    return V3D_ERR_SIZE_LIMIT;
  }
  DeviceGuard guard(device);
  if (guard.rc) return guard.rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  constexpr int QPW = 4;
  constexpr int TS = 1024;
  dim3 grid((unsigned)ceil_div(M, kWavesPerBlock * QPW), (unsigned)B), block(kQueryThreads);
  if constexpr (std::is_same<T, float>::value) {
    // per-workgroup cell list: clouds whose supports fit LDS and that are large enough to pay for sorting them (at
    // N = 1024 the ~9 us builder eats the gain: 22.7 us at best against 24.5 us brute force, profiles/r03_knn_cell_list_sweep.txt)
    if (k + 1 <= 64 && N >= kKgMinN && N <= kKgMaxN && M >= 64) {
      // >= 12 supports per cell: a run of three cells then fills most of a 64-lane step; from k ~ 33 up 0.36 (k + 1),
      // so that the ball of the k + 1 nearest of a uniform cloud stays inside the 27 cells (cell edge >= 1.15 radii)
      const double per_cell = std::max(12.0, 0.36 * (k + 1));
      int G = (int)std::lround(std::cbrt((double)N / per_cell));
      G = std::max(3, std::min(G, kKgMaxG));
      // queries per workgroup: every workgroup sorts the whole cloud, so as many as still leave ~2 workgroups per CU
      int QB = 64;
      while (QB > 16 && (int64_t)B * ceil_div(M, QB) < 512) QB >>= 1;
      static std::mutex attr_mutex;
      static bool attr_set[64] = {};
      {
        std::lock_guard<std::mutex> lock(attr_mutex);
        if (device < 0 || device >= 64 || !attr_set[device]) {
          V3D_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_points_grid_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, kKgMaxN * (int)sizeof(float4)),
                      what);
          if (device >= 0 && device < 64) attr_set[device] = true;
        }
      }
      dim3 ggrid((unsigned)ceil_div(M, QB), (unsigned)B);
      hipLaunchKernelGGL(knn_points_grid_kernel, ggrid, dim3(kKgThreads), (size_t)N * sizeof(float4), stream, q, s, od, oi,
                         M, N, k, G, QB, lex);
      return launch_status(what);
    }
    // batched survivors + bitonic merge: 14-25 % faster than the list from k ~ 24 up, slower below k ~ 12
    if (k >= 24 && k + 1 <= 64) {
      hipLaunchKernelGGL((knn_points_merge_kernel<QPW, TS>), grid, block, 0, stream, q, s, od, oi, M, N, k, nullptr, 0, lex);
      return launch_status(what);
    }
  }
  if (k + 1 <= 64)  // the list holds the k+1 best: one or two entries per lane
    hipLaunchKernelGGL((knn_points_kernel<T, 1, QPW, TS>), grid, block, 0, stream, q, s, od, oi, M, N, k, nullptr, 0, lex);
  else
    hipLaunchKernelGGL((knn_points_kernel<T, 2, QPW, TS>), grid, block, 0, stream, q, s, od, oi, M, N, k, nullptr, 0, lex);
  return launch_status(what);
}

constexpr int kKnnPackMaxClouds = 1 << 20;
// The cell list of radius.hip can serve the call (k <= 64, <= 4096 clouds averaging >= 2048 supports, 32-bit cell
// counters): what v3d_knn_pack_workspace_bytes sizes the workspace for.
static bool knn_pack_may_use_cell_list(int B, int64_t total_s, int k) {
  return k <= kKnnCellListMaxK && B >= 1 && B <= kKnnCellListMaxClouds && total_s / B >= 2048 &&
         4096ll * B + 4ll * total_s + 1 < (int64_t)INT32_MAX;
}
// ... and is worth it: pair by pair costs ~1/2400 ns per (query, support) pair of a cloud, the cell list ~50 us of
// launches + ~0.1 ns per support (counting sort) + ~0.6 ns per query (profiles/r03_knn_pack_cell_list.txt) -- a handful
// of queries against a large cloud stay pair by pair.  The host only knows the totals: the average cloud decides.
static bool knn_pack_uses_cell_list(int B, int64_t total_q, int64_t total_s, int k) {
  if (!knn_pack_may_use_cell_list(B, total_s, k)) return false;
  const double pairwise_ns = (double)total_q * ((double)total_s / B) / 2400.0;
  const double cell_list_ns = 50000.0 + 0.1 * (double)total_s + 0.6 * (double)total_q;
  return cell_list_ns < pairwise_ns;
}

static int knn_pack(const float* q, const float* s, const int64_t* q_lengths, const int64_t* s_lengths, float* od,
                    int64_t* oi, int B, int64_t total_q, int64_t total_s, int k, void* workspace,
                    size_t workspace_bytes, int device, void* stream_) {
  const char* what = "knn_pack_mode";
  V3D_REQUIRE(B >= 0 && total_q >= 0 && total_s >= 0 && k >= 0, what, "negative dimension");
  if (k > V3D_KNN_MAX_K) {
    set_last_error("%s: k=%d exceeds %d", what, k, V3D_KNN_MAX_K);
    return V3D_ERR_K_TOO_LARGE;
  }
  if (total_q == 0 || k == 0) return 0;
  V3D_REQUIRE(B > 0, what, "points without clouds");
  V3D_REQUIRE(q && od && oi && q_lengths && s_lengths && (total_s == 0 || s), what, "null pointer");
  constexpr int QPW = 4;
  constexpr int TS = 1024;
  constexpr int kPerWg = kWavesPerBlock * QPW;
  // a cloud's query count lives in an int inside the plan (KnnPackPlan::q_len): the same 31-bit bound as the supports
  if (B > kKnnPackMaxClouds || total_s > (int64_t)INT32_MAX || total_q > (int64_t)INT32_MAX) {
    set_last_error("%s: at most 2^20 clouds, 2^31-1 supports and 2^31-1 queries per call", what);
    return V3D_ERR_SIZE_LIMIT;
  }
  // Clouds of a few thousand supports and more: the cell list of the radius search (radius.hip) instead of every
  // (query, support) pair -- 480,000 queries in 30,000-point clouds: 6.0 ms -> see profiles/r03_knn_pack_cell_list.txt.
  // The host only knows the totals: the average cloud decides.
  if (knn_pack_uses_cell_list(B, total_q, total_s, k))
    return knn_pack_cell_list(q, s, q_lengths, s_lengths, od, oi, B, total_q, total_s, k, workspace, workspace_bytes, device,
                              stream_);
  DeviceGuard guard(device);
  if (guard.rc) return guard.rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const size_t need = align_up(sizeof(KnnPackPlan) * ((size_t)B + 1), 256);
  void* base = workspace;
  bool own = false;
  if (base == nullptr) {
    V3D_HIP_TRY(hipMallocAsync(&base, need, stream), what);
    own = true;
  } else if (workspace_bytes < need || ((uintptr_t)base & 255) != 0) {
    set_last_error("%s: workspace needs %zu bytes, 256-B aligned (got %zu)", what, need, workspace_bytes);
    return V3D_ERR_WORKSPACE;
  }
  KnnPackPlan* plan = static_cast<KnnPackPlan*>(base);
  hipLaunchKernelGGL(knn_pack_plan_kernel, dim3(1), dim3(64), 0, stream, q_lengths, s_lengths, plan, B,
                     (long long)total_q, (long long)total_s, kPerWg);
  // every cloud rounds its queries up to whole workgroups: at most one partly filled workgroup per cloud
  dim3 grid((unsigned)(ceil_div(total_q, kPerWg) + B)), block(kQueryThreads);
  if (k >= 24 && k + 1 <= 64)
    hipLaunchKernelGGL((knn_points_merge_kernel<QPW, TS, true>), grid, block, 0, stream, q, s, od, oi, 0, 0, k, plan, B, 1);
  else if (k + 1 <= 64)
    hipLaunchKernelGGL((knn_points_kernel<float, 1, QPW, TS, true>), grid, block, 0, stream, q, s, od, oi, 0, 0, k, plan,
                       B, 1);
  else
    hipLaunchKernelGGL((knn_points_kernel<float, 2, QPW, TS, true>), grid, block, 0, stream, q, s, od, oi, 0, 0, k, plan,
                       B, 1);
  int rc = launch_status(what);
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

size_t v3d_knn_pack_workspace_bytes(int B, int64_t total_s, int k) {
  if (B < 0 || total_s < 0) return 0;
  if (v3d::knn_pack_may_use_cell_list(B, total_s, k)) return v3d::knn_pack_cell_list_workspace_bytes(B, total_s);
  return v3d::align_up(sizeof(v3d::KnnPackPlan) * ((size_t)B + 1), 256);
}

int v3d_knn_pack_uses_cell_list(int B, int64_t total_q, int64_t total_s, int k) {
  return (B >= 0 && total_q >= 0 && total_s >= 0 && v3d::knn_pack_uses_cell_list(B, total_q, total_s, k)) ? 1 : 0;
}

int v3d_knn_pack_f32(const float* q_points, const float* s_points, const int64_t* q_lengths, const int64_t* s_lengths,
                     float* distances, int64_t* indices, int B, int64_t total_q, int64_t total_s, int k,
                     void* workspace, size_t workspace_bytes, int device, void* stream) {
  return v3d::knn_pack(q_points, s_points, q_lengths, s_lengths, distances, indices, B, total_q, total_s, k, workspace,
                       workspace_bytes, device, stream);
}

int v3d_ball_query_f32(const float* q, const float* s, int64_t* indices, int B, int M, int N, float max_radius, int k,
                       int device, void* stream) {
  return v3d::ball_query<float>(q, s, indices, B, M, N, max_radius, k, device, stream);
}
int v3d_ball_query_multi_f32(const float* q, const float* s, int64_t* const* indices, const float* max_radii,
                             const int* num_samples, int num_scales, int B, int M, int N, int device, void* stream_) {
  using namespace v3d;
  const char* what = "ball_query_multi";
  V3D_REQUIRE(B >= 0 && M >= 0 && N >= 0, what, "negative dimension");
  V3D_REQUIRE(num_scales >= 1 && num_scales <= kBallMaxScales, what, "1 to 4 scales per call");
  V3D_REQUIRE(indices && max_radii && num_samples, what, "null pointer");
  BallScales sc{};
  int S = 0;
  for (int i = 0; i < num_scales; ++i) {
    V3D_REQUIRE(num_samples[i] >= 0, what, "negative dimension");
    if (num_samples[i] == 0) continue;  // nothing to write for this scale
    V3D_REQUIRE(indices[i] || B == 0 || M == 0, what, "null pointer");
    sc.out[S] = indices[i];
    sc.k[S] = num_samples[i];
    sc.r2[S] = max_radii[i] * max_radii[i];  // float(r) * float(r), ball_query_cuda.cuh:28
    ++S;
  }
  if (B == 0 || M == 0 || N == 0 || S == 0) return 0;
  V3D_REQUIRE(q && s, what, "null pointer");
  if (B > kMaxGridYZ) {
    set_last_error("%s: B <= 65535 required", what);
    return V3D_ERR_SIZE_LIMIT;
  }
  DeviceGuard guard(device);
  if (guard.rc) return guard.rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  constexpr int G = 4, TS = 2048;
  const dim3 grid((unsigned)ceil_div(M, kWavesPerBlock * G), (unsigned)B), block(kQueryThreads);
  if (S == 1) hipLaunchKernelGGL((ball_query_kernel<float, G, TS>), grid, block, 0, stream, q, s, sc.out[0], M, N, sc.k[0], sc.r2[0]);
  else if (S == 2) hipLaunchKernelGGL((ball_query_multi_kernel<2, G, TS>), grid, block, 0, stream, q, s, sc, M, N);
  else if (S == 3) hipLaunchKernelGGL((ball_query_multi_kernel<3, G, TS>), grid, block, 0, stream, q, s, sc, M, N);
  else hipLaunchKernelGGL((ball_query_multi_kernel<4, G, TS>), grid, block, 0, stream, q, s, sc, M, N);
  return launch_status(what);
}
int v3d_ball_query_f64(const double* q, const double* s, int64_t* indices, int B, int M, int N, float max_radius,
                       int k, int device, void* stream) {
  return v3d::ball_query<double>(q, s, indices, B, M, N, max_radius, k, device, stream);
}
int v3d_radius_count_f32(const float* q, const float* s, int64_t* counts, int B, int M, int N, float radius, int device,
                         void* stream_) {
  using namespace v3d;
  const char* what = "radius_count";
  V3D_REQUIRE(B >= 0 && M >= 0 && N >= 0, what, "negative dimension");
  if (B == 0 || M == 0) return 0;
  V3D_REQUIRE(q && counts && (N == 0 || s), what, "null pointer");
  if (B > kMaxGridYZ) {
    set_last_error("%s: B <= 65535 required", what);
    return V3D_ERR_SIZE_LIMIT;
  }
  DeviceGuard guard(device);
  if (guard.rc) return guard.rc;
  dim3 grid((unsigned)ceil_div(M, kQueryThreads), (unsigned)B), block(kQueryThreads);
  hipLaunchKernelGGL((radius_count_kernel<1024>), grid, block, 0, static_cast<hipStream_t>(stream_), q, s, counts, M, N,
                     radius);
  return launch_status(what);
}
int v3d_knn_points_f32(const float* q, const float* s, float* d, int64_t* i, int B, int M, int N, int k, int device,
                       void* stream) {
  return v3d::knn_points<float>(q, s, d, i, B, M, N, k, device, stream);
}
int v3d_knn_lex_f32(const float* q, const float* s, float* d, int64_t* i, int B, int M, int N, int k, int device,
                    void* stream) {
  return v3d::knn_points<float>(q, s, d, i, B, M, N, k, device, stream, 1);
}
int v3d_knn_points_f64(const double* q, const double* s, double* d, int64_t* i, int B, int M, int N, int k, int device,
                       void* stream) {
  return v3d::knn_points<double>(q, s, d, i, B, M, N, k, device, stream);
}
int v3d_three_nn_f32(const float* q, const float* s, float* d, int64_t* i, int B, int Q, int S, int device,
                     void* stream) {
  return v3d::three_nn<float>(q, s, d, i, B, Q, S, device, stream);
}
int v3d_three_nn_f64(const double* q, const double* s, double* d, int64_t* i, int B, int Q, int S, int device,
                     void* stream) {
  return v3d::three_nn<double>(q, s, d, i, B, Q, S, device, stream);
}

}  // extern "C"
