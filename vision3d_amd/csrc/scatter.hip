// scatter.hip -- atomic-free, bit-reproducible backward of gather / group_gather / three_interpolate.
// Replaces the atomicAdd kernels of gather_cuda.cuh:44-66, group_gather_cuda.cuh:48-75 and
// three_interpolate_cuda.cuh:48-83 (semantics only; the design is gfx950-first).
//
//   grad_src[b,c,n] = sum over the target positions t with idx[b,t] == n of value(b,c,t), ascending t
//     MODE 0: value = grad_out[b,c,t]                           (gather, group_gather)
//     MODE 1: value = grad_out[b,c,t/3] * weight[b,t]           (three_interpolate; product rounded first)
//
// grad_out is channel-major (positions contiguous) while the sum runs over positions grouped by source: the
// transposition goes through LDS.  Per call, once for all channels, the index tensor is inverted into a CSR keyed by
// (column chunk, source): count -> scan -> fill -> rank-sort (32-bit integer atomics on the small count/cursor arrays
// only).  The main kernel gives a workgroup CT channels of one cloud; per column chunk it stages the CT x P tile of
// grad_out with 16-byte loads, the chunk's entry list (16-bit local columns, already in (source, t) order), its
// weights and its row of offsets in LDS, and every thread then walks the entries of its own sources with LDS
// latency only, carrying its accumulators across chunks in registers.  One thread sums one (b,c,n) in ascending t: no
// float atomics, and the result equals the sequential CPU loop bit for bit.
#include <algorithm>

#include "common.h"

namespace v3d {

constexpr int kChunkCols = 2048;        // columns of grad_out per chunk (MODE 1: three entries per column)
constexpr int kMaxLdsSources = 4096;    // above this the sources of a cloud do not fit one workgroup's registers
constexpr int kScatterThreads = 1024;     // CSR build workgroup
constexpr int kSumThreads = 512;          // summing workgroup: two of them share a CU

template <int MODE>
struct ChunkGeom {
  static constexpr int P = kChunkCols;
  static constexpr int EMAX = MODE == 1 ? 3 * P : P;  // entries (target positions) a chunk can hold
};

static int offrow_stride(int N) { return (int)align_up((size_t)N + 1, 8); }  // uint16 elements, 16-byte rows

// Two workspace layouts share one allocation (sized for the larger, so that v3d_scatter_workspace_bytes needs
// neither mode nor dtype):
//   chunked (N <= kMaxLdsSources): offrow u16 [B][nchunks][stride(N)] | ent u16 [B][nchunks][EMAX] | wsorted T [..]
//   generic                      : offsets i32 [B][N+1] | cursor i32 [B][N] | unsorted i32 [B][L] | sorted i32 [B][L]
struct ScatterWorkspace {
  uint16_t* offrow;
  uint16_t* ent;
  void* wsorted;
  int32_t* offsets;
  int32_t* cursor;
  int32_t* unsorted;
  int32_t* sorted;
  size_t bytes;
};

static ScatterWorkspace carve_workspace(void* base, int B, int N, int64_t L) {
  ScatterWorkspace w{};
  size_t off = 0;
  auto take = [&](size_t n) {
    void* p = base ? static_cast<char*>(base) + off : nullptr;
    off += align_up(n, 256);
    return p;
  };
  if (N <= kMaxLdsSources) {
    const size_t nch = (size_t)std::max<int64_t>(1, ceil_div(L, kChunkCols));  // MODE 0 has the most chunks
    const size_t emax_total = (size_t)L + 3 * kChunkCols;                       // nchunks*EMAX <= L + EMAX, any mode
    w.offrow = static_cast<uint16_t*>(take(sizeof(uint16_t) * (size_t)B * nch * offrow_stride(N)));
    w.ent = static_cast<uint16_t*>(take(sizeof(uint16_t) * (size_t)B * emax_total));
    w.wsorted = take(sizeof(double) * (size_t)B * emax_total);
  } else {
    w.offsets = static_cast<int32_t*>(take(sizeof(int32_t) * (size_t)B * ((size_t)N + 1)));
    w.cursor = static_cast<int32_t*>(take(sizeof(int32_t) * (size_t)B * (size_t)N));
    w.unsorted = static_cast<int32_t*>(take(sizeof(int32_t) * (size_t)B * (size_t)L));
    w.sorted = static_cast<int32_t*>(take(sizeof(int32_t) * (size_t)B * (size_t)L));
  }
  w.bytes = off;
  return w;
}

// ------------------------------------------------------------------------------------------------------
// chunked path: one workgroup inverts one column chunk of one cloud entirely in LDS (no global atomics)
// ------------------------------------------------------------------------------------------------------
// LDS: offs i32 [N+1] | cursor i32 [N] | srcl u16 [EMAX] | ul u16 [EMAX]
template <typename T, int MODE>
__global__ __launch_bounds__(kScatterThreads) void csr_chunk_build_kernel(
    const int64_t* __restrict__ idx, const T* __restrict__ w, uint16_t* __restrict__ offrow_g,
    uint16_t* __restrict__ ent_g, T* __restrict__ wsorted_g, int N, int64_t L, int nchunks, int ostride) {
  constexpr int EMAX = ChunkGeom<MODE>::EMAX;
  constexpr int kItems = kMaxLdsSources / kScatterThreads;  // scan items per thread
  extern __shared__ __align__(16) unsigned char smem_raw[];
  int* offs = reinterpret_cast<int*>(smem_raw);
  int* cursor = offs + (N + 1);
  uint16_t* srcl = reinterpret_cast<uint16_t*>(cursor + N);
  uint16_t* ul = srcl + EMAX;
  __shared__ int wave_tot[kScatterThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k = blockIdx.x, b = blockIdx.y;
  const int64_t t0 = (int64_t)k * EMAX;
  const int ne = (int)max((int64_t)0, min((int64_t)EMAX, L - t0));
  const int64_t* bidx = idx + (int64_t)b * L + t0;
  for (int n = tid; n <= N; n += kScatterThreads) offs[n] = 0;
  __syncthreads();
  for (int e = tid; e < ne; e += kScatterThreads) {
    const int src = (int)bidx[e];
    const bool ok = src >= 0 && src < N;
    srcl[e] = ok ? (uint16_t)src : (uint16_t)0xFFFF;
    if (ok) atomicAdd(&offs[src], 1);
  }
  __syncthreads();
  // exclusive scan of the N counts, kItems consecutive sources per thread
  int v[kItems];
  int sum = 0;
#pragma unroll
  for (int i = 0; i < kItems; ++i) {
    const int n = tid * kItems + i;
    v[i] = n < N ? offs[n] : 0;
    sum += v[i];
  }
  int incl = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  int run = incl - sum;
  for (int q = 0; q < wave; ++q) run += wave_tot[q];
  uint16_t* orow = offrow_g + ((int64_t)b * nchunks + k) * ostride;
#pragma unroll
  for (int i = 0; i < kItems; ++i) {
    const int n = tid * kItems + i;
    if (n < N) {
      offs[n] = run;
      cursor[n] = 0;
      orow[n] = (uint16_t)run;
    }
    run += v[i];
  }
  if (tid == kScatterThreads - 1) {
    offs[N] = run;  // entries in this chunk
    orow[N] = (uint16_t)run;
  }
  __syncthreads();
  for (int e = tid; e < ne; e += kScatterThreads) {
    const int src = srcl[e];
    if (src != 0xFFFF) ul[offs[src] + atomicAdd(&cursor[src], 1)] = (uint16_t)e;
  }
  __syncthreads();
  // rank sort inside every group (local positions are distinct: rank = number of smaller members)
  const int total = offs[N];
  uint16_t* eg = ent_g + ((int64_t)b * nchunks + k) * EMAX;
  T* wg = MODE == 1 ? wsorted_g + ((int64_t)b * nchunks + k) * EMAX : nullptr;
  for (int s = tid; s < total; s += kScatterThreads) {
    const int e = ul[s];
    const int src = srcl[e];
    const int start = offs[src], end = offs[src + 1];
    int rank = 0;
    for (int j = start; j < end; ++j) rank += (ul[j] < e) ? 1 : 0;
    eg[start + rank] = (uint16_t)(MODE == 1 ? e / 3 : e);
    if (MODE == 1) wg[start + rank] = w[(int64_t)b * L + t0 + e];
  }
}

// Summing kernel of the chunked path, see the file header.  LDS: tile[CT][P] | wl[EMAX] (MODE 1) | ent[EMAX] |
// offl[stride(N)].  NPT sources per thread (source n = j*TH + tid).  The next chunk's 16-byte loads are issued
// before the current chunk is summed and land in LDS after it.
template <typename T, int MODE, int CT, int NPT, int TH>
__global__ __launch_bounds__(TH, 4) void segment_sum_chunked_kernel(
    const T* __restrict__ gout, const uint16_t* __restrict__ offrow_g, const uint16_t* __restrict__ ent_g,
    const T* __restrict__ wsorted_g, T* __restrict__ gsrc, int C, int N, int64_t Lrow, int nchunks, int ostride,
    int vec_ok, GridMap gm) {
  constexpr int P = ChunkGeom<MODE>::P;
  constexpr int EMAX = ChunkGeom<MODE>::EMAX;
  constexpr int kVec = 16 / sizeof(T);
  constexpr int kRowV = P / kVec;                                  // 16-byte vectors per tile row
  constexpr int kTileV = CT * kRowV / TH;                          // per thread
  constexpr int kEntV = (EMAX * 2 / 16 + TH - 1) / TH;
  constexpr int kWV = MODE == 1 ? (EMAX * (int)sizeof(T) / 16 + TH - 1) / TH : 1;
  constexpr int kOffV = ((kMaxLdsSources + 8) / 8 + TH - 1) / TH;  // offset row: up to 4104 uint16
  static_assert(CT * kRowV % TH == 0, "tile must split evenly over the workgroup");
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* tile = reinterpret_cast<T*>(smem_raw);
  T* wl = tile + CT * P;
  uint16_t* ent = reinterpret_cast<uint16_t*>(wl + (MODE == 1 ? EMAX : 0));
  uint16_t* offl = ent + EMAX;
  const int tid = threadIdx.x;
  int bx, by, b;
  decode_block(gm, bx, by, b);
  const int c0 = bx * CT;
  const int offv = ostride / 8;  // 16-byte vectors per offset row
  T acc[NPT][CT];
#pragma unroll
  for (int j = 0; j < NPT; ++j)
#pragma unroll
    for (int i = 0; i < CT; ++i) acc[j][i] = T(0);

  uint4 rt[kTileV], re[kEntV], rw[kWV], ro[kOffV];
  // issue the loads of chunk k into registers
#define V3D_LOAD_CHUNK(k)                                                                                      \
  do {                                                                                                         \
    const int64_t col0_ = (int64_t)(k) * P;                                                                    \
    const int pcv_ = (int)(min((int64_t)P, Lrow - col0_) / kVec);                                              \
    const int64_t chunk_ = (int64_t)b * nchunks + (k);                                                         \
    _Pragma("unroll") for (int i = 0; i < kTileV; ++i) {                                                       \
      const int v = i * TH + tid, row = v / kRowV, within = v % kRowV;                                         \
      const bool ok = (c0 + row < C) && within < pcv_;                                                         \
      const uint4* src = reinterpret_cast<const uint4*>(gout + ((int64_t)b * C + (ok ? c0 + row : 0)) * Lrow + \
                                                         col0_);                                               \
      rt[i] = ok ? src[within] : make_uint4(0, 0, 0, 0);                                                       \
    }                                                                                                          \
    _Pragma("unroll") for (int i = 0; i < kEntV; ++i) {                                                        \
      const int v = i * TH + tid;                                                                              \
      re[i] = v < EMAX * 2 / 16 ? reinterpret_cast<const uint4*>(ent_g + chunk_ * EMAX)[v]                     \
                                : make_uint4(0, 0, 0, 0);                                                      \
    }                                                                                                          \
    if (MODE == 1) {                                                                                           \
      _Pragma("unroll") for (int i = 0; i < kWV; ++i) {                                                        \
        const int v = i * TH + tid;                                                                            \
        rw[i] = v < EMAX * (int)sizeof(T) / 16 ? reinterpret_cast<const uint4*>(wsorted_g + chunk_ * EMAX)[v]  \
                                                : make_uint4(0, 0, 0, 0);                                      \
      }                                                                                                        \
    }                                                                                                          \
    _Pragma("unroll") for (int i = 0; i < kOffV; ++i) {                                                        \
      const int v = i * TH + tid;                                                                              \
      ro[i] = v < offv ? reinterpret_cast<const uint4*>(offrow_g + chunk_ * ostride)[v] : make_uint4(0, 0, 0, 0); \
    }                                                                                                          \
  } while (0)

  if (vec_ok) V3D_LOAD_CHUNK(0);
  for (int k = 0; k < nchunks; ++k) {
    __syncthreads();  // the previous chunk has been consumed
    if (vec_ok) {
#pragma unroll
      for (int i = 0; i < kTileV; ++i) reinterpret_cast<uint4*>(tile)[i * TH + tid] = rt[i];
#pragma unroll
      for (int i = 0; i < kEntV; ++i) {
        const int v = i * TH + tid;
        if (v < EMAX * 2 / 16) reinterpret_cast<uint4*>(ent)[v] = re[i];
      }
      if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < kWV; ++i) {
          const int v = i * TH + tid;
          if (v < EMAX * (int)sizeof(T) / 16) reinterpret_cast<uint4*>(wl)[v] = rw[i];
        }
      }
#pragma unroll
      for (int i = 0; i < kOffV; ++i) {
        const int v = i * TH + tid;
        if (v < offv) reinterpret_cast<uint4*>(offl)[v] = ro[i];
      }
    } else {  // rows that are not 16-byte tileable: element-wise staging, no prefetch
      const int64_t col0 = (int64_t)k * P;
      const int pc = (int)min((int64_t)P, Lrow - col0);
      const int64_t chunk = (int64_t)b * nchunks + k;
#pragma unroll
      for (int i = 0; i < CT; ++i) {
        if (c0 + i < C) {
          const T* row = gout + ((int64_t)b * C + c0 + i) * Lrow + col0;
          for (int e = tid; e < pc; e += TH) tile[i * P + e] = row[e];
        }
      }
      for (int e = tid; e < EMAX; e += TH) ent[e] = ent_g[chunk * EMAX + e];
      if (MODE == 1)
        for (int e = tid; e < EMAX; e += TH) wl[e] = wsorted_g[chunk * EMAX + e];
      for (int n = tid; n <= N; n += TH) offl[n] = offrow_g[chunk * ostride + n];
    }
    __syncthreads();
    if (vec_ok && k + 1 < nchunks) V3D_LOAD_CHUNK(k + 1);
    // Round-robin over the thread's NPT sources: round r adds entry r of every source that still has one, so the
    // LDS reads of different sources are independent and in flight together (a source-by-source walk is a chain of
    // dependent LDS latencies: offsets -> entry -> tile value).  The next round's entries are fetched a round ahead.
    // fl[j] = first entry | (number of entries << 16) of source j in this chunk
    unsigned fl[NPT];
    int lcn[NPT];
    T wn[NPT];
    int maxlen = 0;
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
      const int n = j * TH + tid;
      fl[j] = 0;
      if (n < N) {
        const unsigned f = offl[n], e = offl[n + 1];
        fl[j] = f | ((e - f) << 16);
      }
      maxlen = max(maxlen, (int)(fl[j] >> 16));
    }
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
      lcn[j] = 0;
      wn[j] = T(1);
      if ((fl[j] >> 16) > 0) {
        lcn[j] = ent[fl[j] & 0xFFFFu];
        if (MODE == 1) wn[j] = wl[fl[j] & 0xFFFFu];
      }
    }
    for (int r = 0; r < maxlen; ++r) {
      constexpr int G = NPT < 4 ? NPT : (NPT > 4 ? 2 : 4);  // sources whose tile values are in flight together
#pragma unroll
      for (int g = 0; g < NPT; g += G) {
        T v[G][CT];
        T wt[G];
#pragma unroll
        for (int j = 0; j < G; ++j) {
          const int cnt = (int)(fl[g + j] >> 16);
          wt[j] = wn[g + j];
          if (r < cnt) {
            const int lc = lcn[g + j];
#pragma unroll
            for (int i = 0; i < CT; ++i) v[j][i] = tile[i * P + lc];
          }
          if (r + 1 < cnt) {
            const int q = (int)(fl[g + j] & 0xFFFFu) + r + 1;
            lcn[g + j] = ent[q];
            if (MODE == 1) wn[g + j] = wl[q];
          }
        }
#pragma unroll
        for (int j = 0; j < G; ++j) {
          if (r < (int)(fl[g + j] >> 16)) {
#pragma unroll
            for (int i = 0; i < CT; ++i) {
              T x = v[j][i];
              if (MODE == 1) x = x * wt[j];
              acc[g + j][i] = acc[g + j][i] + x;
            }
          }
        }
      }
    }
  }
#undef V3D_LOAD_CHUNK
#pragma unroll
  for (int j = 0; j < NPT; ++j) {
    const int n = j * TH + tid;
    if (n < N) {
#pragma unroll
      for (int i = 0; i < CT; ++i)
        if (c0 + i < C) gsrc[((int64_t)b * C + c0 + i) * N + n] = acc[j][i];
    }
  }
}

template <typename T, int MODE, int CT, int NPT, int TH>
static int launch_chunked(const char* what, const T* gout, const ScatterWorkspace& ws, T* gsrc, int B, int C, int N,
                          int64_t Lrow, int nchunks, hipStream_t stream) {
  constexpr int P = ChunkGeom<MODE>::P;
  constexpr int EMAX = ChunkGeom<MODE>::EMAX;
  const int ostride = offrow_stride(N);
  const size_t lds = sizeof(T) * CT * P + (MODE == 1 ? sizeof(T) * EMAX : 0) + sizeof(uint16_t) * EMAX +
                     sizeof(uint16_t) * (size_t)ostride;
  auto kernel = segment_sum_chunked_kernel<T, MODE, CT, NPT, TH>;
  static bool attr_set[64] = {};  // per device: the LDS request exceeds the 64 KiB default for some instantiations
  int dev = 0;
  V3D_HIP_TRY(hipGetDevice(&dev), what);
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    constexpr size_t kLdsMax = sizeof(T) * CT * P + sizeof(T) * 3 * P + sizeof(uint16_t) * 3 * P +
                               sizeof(uint16_t) * (kMaxLdsSources + 8);
    V3D_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax),
                what);
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  const int vec_ok = (Lrow % (16 / sizeof(T)) == 0) && ((uintptr_t)gout % 16 == 0);
  const int64_t ctiles = ceil_div(C, CT);
  const GridMap gm = make_grid_map(ctiles, 1, B);
  hipLaunchKernelGGL(kernel, dim3((unsigned)(ctiles * B)), dim3(TH), lds, stream, gout, ws.offrow,
                     ws.ent, static_cast<const T*>(ws.wsorted), gsrc, C, N, Lrow, nchunks, ostride, vec_ok, gm);
  return 0;
}

// ------------------------------------------------------------------------------------------------------
// generic path (more than kMaxLdsSources sources per cloud): global CSR keyed by source + thread per source
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void csr_count_kernel(const int64_t* __restrict__ idx, int32_t* cursor, int N,
                                                        int64_t L) {
  const int b = blockIdx.y;
  const int64_t* bidx = idx + (int64_t)b * L;
  int32_t* cur = cursor + (int64_t)b * N;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < L; p += (int64_t)gridDim.x * blockDim.x) {
    const int key = (int)bidx[p];
    if (key >= 0 && key < N) atomicAdd(&cur[key], 1);
  }
}

// One workgroup per cloud: exclusive scan of the counts (8 per thread and sweep), reset of the cursors.
constexpr int kScanItems = 8;
__global__ __launch_bounds__(1024) void csr_scan_kernel(int32_t* cursor, int32_t* offsets, int N) {
  __shared__ int wave_tot[16];
  __shared__ int carry_s;
  const int b = blockIdx.x;
  int32_t* cur = cursor + (int64_t)b * N;
  int32_t* off = offsets + (int64_t)b * (N + 1);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < N; base += 1024 * kScanItems) {
    const int i0 = base + tid * kScanItems;
    int v[kScanItems];
    int sum = 0;
#pragma unroll
    for (int e = 0; e < kScanItems; ++e) {
      v[e] = i0 + e < N ? cur[i0 + e] : 0;
      sum += v[e];
    }
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      int t = __shfl_up(incl, d, 64);
      if (lane >= d) incl += t;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int wave_base = 0;
    for (int q = 0; q < wave; ++q) wave_base += wave_tot[q];
    int run = carry_s + wave_base + incl - sum;
#pragma unroll
    for (int e = 0; e < kScanItems; ++e) {
      if (i0 + e < N) {
        off[i0 + e] = run;
        cur[i0 + e] = 0;
      }
      run += v[e];
    }
    __syncthreads();
    if (tid == 1023) carry_s = run;
    __syncthreads();
  }
  if (tid == 0) off[N] = carry_s;
}

__global__ __launch_bounds__(256) void csr_fill_kernel(const int64_t* __restrict__ idx,
                                                       const int32_t* __restrict__ offsets, int32_t* cursor,
                                                       int32_t* unsorted, int N, int64_t L) {
  const int b = blockIdx.y;
  const int64_t* bidx = idx + (int64_t)b * L;
  const int32_t* off = offsets + (int64_t)b * (N + 1);
  int32_t* cur = cursor + (int64_t)b * N;
  int32_t* uns = unsorted + (int64_t)b * L;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < L; p += (int64_t)gridDim.x * blockDim.x) {
    const int key = (int)bidx[p];
    if (key >= 0 && key < N) uns[off[key] + atomicAdd(&cur[key], 1)] = (int32_t)p;
  }
}

// Rank sort inside every group: positions are distinct, so rank = number of smaller members.
__global__ __launch_bounds__(256) void csr_rank_kernel(const int64_t* __restrict__ idx,
                                                       const int32_t* __restrict__ offsets,
                                                       const int32_t* __restrict__ unsorted, int32_t* sorted,
                                                       int N, int64_t L) {
  const int b = blockIdx.y;
  const int64_t* bidx = idx + (int64_t)b * L;
  const int32_t* off = offsets + (int64_t)b * (N + 1);
  const int32_t* uns = unsorted + (int64_t)b * L;
  int32_t* srt = sorted + (int64_t)b * L;
  const int filled = off[N];
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < filled;
       s += (int64_t)gridDim.x * blockDim.x) {
    const int e = uns[s];
    const int key = (int)bidx[e];
    const int start = off[key], end = off[key + 1];
    int rank = 0;
    for (int j = start; j < end; ++j) rank += (uns[j] < e) ? 1 : 0;
    srt[start + rank] = e;
  }
}

template <typename T, int MODE, int CT>
__global__ __launch_bounds__(256) void segment_sum_kernel(const T* __restrict__ gout, const T* __restrict__ w,
                                                          const int32_t* __restrict__ offsets,
                                                          const int32_t* __restrict__ sorted, T* __restrict__ gsrc,
                                                          int C, int N, int64_t L, int64_t Lrow) {
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * CT;
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const int32_t* off = offsets + (int64_t)b * (N + 1);
  const int32_t* srt = sorted + (int64_t)b * L;
  const T* wb = MODE == 1 ? w + (int64_t)b * L : nullptr;
  const int start = off[n], end = off[n + 1];
  T acc[CT];
#pragma unroll
  for (int i = 0; i < CT; ++i) acc[i] = T(0);
  const T* g0 = gout + ((int64_t)b * C + c0) * Lrow;
  for (int j = start; j < end; ++j) {
    const int t = srt[j];
    const int64_t col = MODE == 1 ? (int64_t)(t / 3) : (int64_t)t;
    const T wt = MODE == 1 ? wb[t] : T(1);
#pragma unroll
    for (int i = 0; i < CT; ++i) {
      if (c0 + i < C) {
        T v = g0[(int64_t)i * Lrow + col];
        if (MODE == 1) v = v * wt;
        acc[i] = acc[i] + v;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < CT; ++i)
    if (c0 + i < C) gsrc[((int64_t)b * C + c0 + i) * N + n] = acc[i];
}

template <typename T, int MODE>
int scatter_backward(const char* what, const T* gout, const int64_t* idx, const T* w, T* gsrc, int B, int C, int N,
                     int64_t L, void* workspace, size_t workspace_bytes, int device, void* stream_) {
  V3D_REQUIRE(B >= 0 && C >= 0 && N >= 0 && L >= 0, what, "negative dimension");
  if (B == 0 || C == 0 || N == 0) return 0;
  V3D_REQUIRE(gsrc && (L == 0 || (gout && idx)) && (MODE == 0 || L == 0 || w), what, "null pointer");
  if (B > kMaxGridYZ || L > (int64_t)INT32_MAX) {
    set_last_error("%s: B <= 65535 and targets per cloud < 2^31 required", what);
    return V3D_ERR_SIZE_LIMIT;
  }
  DeviceGuard guard(device);
  if (guard.rc) return guard.rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (L == 0) {
    V3D_HIP_TRY(hipMemsetAsync(gsrc, 0, sizeof(T) * (size_t)B * C * N, stream), what);
    return 0;
  }
  const int64_t Lrow = MODE == 1 ? L / 3 : L;
  const bool chunked = N <= kMaxLdsSources;
  const int nchunks = (int)std::max<int64_t>(1, ceil_div(Lrow, kChunkCols));
  if ((int64_t)ceil_div(C, 2) * B > (int64_t)INT32_MAX) {
    set_last_error("%s: launch exceeds the grid limit", what);
    return V3D_ERR_SIZE_LIMIT;
  }
  const size_t need = carve_workspace(nullptr, B, N, L).bytes;
  void* base = workspace;
  bool own = false;
  if (base == nullptr) {
    V3D_HIP_TRY(hipMallocAsync(&base, need, stream), what);
    own = true;
  } else if (workspace_bytes < need || ((uintptr_t)base & 255) != 0) {
    set_last_error("%s: workspace needs %zu bytes, 256-B aligned (got %zu)", what, need, workspace_bytes);
    return V3D_ERR_WORKSPACE;
  }
  ScatterWorkspace ws = carve_workspace(base, B, N, L);
  int rc = 0;
  do {
    if (chunked) {
      constexpr int EMAX = ChunkGeom<MODE>::EMAX;
      const int ostride = offrow_stride(N);
      const size_t lds = sizeof(int) * (2 * (size_t)N + 1) + 2 * sizeof(uint16_t) * EMAX;
      hipLaunchKernelGGL((csr_chunk_build_kernel<T, MODE>), dim3((unsigned)nchunks, (unsigned)B),
                         dim3(kScatterThreads), lds, stream, idx, w, ws.offrow, ws.ent, static_cast<T*>(ws.wsorted), N,
                         L, nchunks, ostride);
      constexpr int CT = sizeof(T) == 4 ? 4 : 2;  // 32 KiB tile
      // two 512-thread workgroups per CU while a thread can hold its sources' accumulators; beyond that one
      // 1024-thread workgroup (NPT = 8 at 512 threads spills)
      if (N <= 2 * kSumThreads)
        rc = launch_chunked<T, MODE, CT, 2, kSumThreads>(what, gout, ws, gsrc, B, C, N, Lrow, nchunks, stream);
      else
        rc = launch_chunked<T, MODE, CT, kMaxLdsSources / 1024, 1024>(what, gout, ws, gsrc, B, C, N, Lrow, nchunks,
                                                                     stream);
      if (rc != 0) break;
    } else {
      hipError_t e = hipMemsetAsync(ws.cursor, 0, sizeof(int32_t) * (size_t)B * N, stream);
      if (e != hipSuccess) {
        set_last_error("%s: hipMemsetAsync: %s", what, hipGetErrorString(e));
        rc = (int)e;
        break;
      }
      const unsigned xb = (unsigned)std::min<int64_t>(ceil_div(L, 256), 1024);
      hipLaunchKernelGGL(csr_count_kernel, dim3(xb, B), dim3(256), 0, stream, idx, ws.cursor, N, L);
      hipLaunchKernelGGL(csr_scan_kernel, dim3(B), dim3(1024), 0, stream, ws.cursor, ws.offsets, N);
      hipLaunchKernelGGL(csr_fill_kernel, dim3(xb, B), dim3(256), 0, stream, idx, ws.offsets, ws.cursor, ws.unsorted,
                         N, L);
      hipLaunchKernelGGL(csr_rank_kernel, dim3(xb, B), dim3(256), 0, stream, idx, ws.offsets, ws.unsorted, ws.sorted,
                         N, L);
      constexpr int CT = 8;
      dim3 grid((unsigned)ceil_div(N, 256), (unsigned)ceil_div(C, CT), (unsigned)B);
      hipLaunchKernelGGL((segment_sum_kernel<T, MODE, CT>), grid, dim3(256), 0, stream, gout, w, ws.offsets,
                         ws.sorted, gsrc, C, N, L, Lrow);
    }
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

// used by interp.hip
int three_interp_bwd_f32_impl(const float* gout, const int64_t* idx, const float* w, float* gsrc, int B, int C,
                              int S, int Q, void* ws, size_t wsb, int device, void* stream) {
  return scatter_backward<float, 1>("three_interpolate_backward", gout, idx, w, gsrc, B, C, S, (int64_t)Q * 3, ws,
                                    wsb, device, stream);
}
int three_interp_bwd_f64_impl(const double* gout, const int64_t* idx, const double* w, double* gsrc, int B, int C,
                              int S, int Q, void* ws, size_t wsb, int device, void* stream) {
  return scatter_backward<double, 1>("three_interpolate_backward", gout, idx, w, gsrc, B, C, S, (int64_t)Q * 3, ws,
                                     wsb, device, stream);
}

}  // namespace v3d

extern "C" {

size_t v3d_scatter_workspace_bytes(int B, int num_sources, int64_t positions_per_batch) {
  if (B <= 0 || num_sources < 0 || positions_per_batch < 0) return 0;
  return v3d::carve_workspace(nullptr, B, num_sources, positions_per_batch).bytes;
}

int v3d_gather_bwd_f32(const float* g, const int64_t* idx, float* gs, int B, int C, int N, int M, void* ws,
                       size_t wsb, int device, void* stream) {
  return v3d::scatter_backward<float, 0>("gather_backward", g, idx, nullptr, gs, B, C, N, (int64_t)M, ws, wsb,
                                         device, stream);
}
int v3d_gather_bwd_f64(const double* g, const int64_t* idx, double* gs, int B, int C, int N, int M, void* ws,
                       size_t wsb, int device, void* stream) {
  return v3d::scatter_backward<double, 0>("gather_backward", g, idx, nullptr, gs, B, C, N, (int64_t)M, ws, wsb,
                                          device, stream);
}
int v3d_group_gather_bwd_f32(const float* g, const int64_t* idx, float* gs, int B, int C, int N, int M, int K,
                             void* ws, size_t wsb, int device, void* stream) {
  if (M < 0 || K < 0) return v3d::invalid("group_gather_backward", "negative dimension");
  return v3d::scatter_backward<float, 0>("group_gather_backward", g, idx, nullptr, gs, B, C, N, (int64_t)M * K, ws,
                                         wsb, device, stream);
}
int v3d_group_gather_bwd_f64(const double* g, const int64_t* idx, double* gs, int B, int C, int N, int M, int K,
                             void* ws, size_t wsb, int device, void* stream) {
  if (M < 0 || K < 0) return v3d::invalid("group_gather_backward", "negative dimension");
  return v3d::scatter_backward<double, 0>("group_gather_backward", g, idx, nullptr, gs, B, C, N, (int64_t)M * K, ws,
                                          wsb, device, stream);
}

}  // extern "C"
