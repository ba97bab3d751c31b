// scatter.hip -- atomic-free, bit-reproducible backward of gather / group_gather / three_interpolate.
// Replaces the atomicAdd kernels of gather_cuda.cuh:44-66, group_gather_cuda.cuh:48-75 and
// three_interpolate_cuda.cuh:48-83 (semantics only; the design is gfx950-first).
//
//   grad_src[b,c,n] = sum over the target positions t with idx[b,t] == n of value(b,c,t), ascending t
//     MODE 0: value = grad_out[b,c,t]                           (gather, group_gather)
//     MODE 1: value = grad_out[b,c,t/3] * weight[b,t]           (three_interpolate; product rounded first)
//
// grad_out is channel-major (positions contiguous) while the sum runs over positions grouped by source: the
// transposition goes through LDS, in column chunks of P = 2048.
//   1. csr_chunk_build_kernel, once per call for all channels: one workgroup inverts one chunk of one cloud entirely
//      in LDS (histogram, scan, fill, rank sort; LDS integer atomics only) and writes the chunk's entry list (16-bit
//      local columns in (source, t) order, bit 15 marks the last entry of a source), the weights in the same order,
//      and a compact list of the non-empty sources ("segments": source | first entry << 16).
//   2. segment_sum_kernel: a workgroup owns CT channels of one cloud and keeps acc[N][CT] in LDS.  Per chunk it
//      stages the CT x P tile of grad_out transposed to [column][CT] (one 16-byte LDS read per entry and 4
//      channels), the entry list and the weights; then ONE THREAD PER SEGMENT reads acc[n], adds its entries in
//      order and writes acc[n] back: every lane has work, whatever the index distribution.  Chunks are two deep in
//      flight through two register sets, since one workgroup fills a CU.
// One thread sums one (b,c,n) in ascending t: no float atomics, and the result equals the sequential CPU loop bit
// for bit.  Clouds with more than 4096 sources take a generic path (global CSR, thread per source, L2 gathers).
#include <algorithm>

#include <mutex>

#include "common.h"

namespace v3d {

constexpr int kChunkCols = 2048;      // columns of grad_out per chunk (MODE 1: three entries per column)
constexpr int kInterpChunkCols = 1024;  // MODE 1: a target row is short (Q ~ 4096), smaller chunks = a deeper load pipeline
constexpr int kMaxLdsSources = 4096;  // acc[N][CT] must fit LDS beside the tile
constexpr int kScatterThreads = 1024;
constexpr unsigned kNoSegment = 0xFFFFFFFFu;

template <int MODE>
struct ChunkGeom {
  static constexpr int P = MODE == 1 ? kInterpChunkCols : kChunkCols;
  static constexpr int EMAX = MODE == 1 ? 3 * P : P;  // entries (target positions) a chunk can hold
};

// segments a chunk can hold: one per source that has an entry in it (multiple of 4)
static int seg_stride(int N, int emax) { return (int)align_up((size_t)std::min(N, emax), 4); }

// Two workspace layouts share one allocation (sized for the larger, so that v3d_scatter_workspace_bytes needs
// neither mode nor dtype):
//   chunked (N <= kMaxLdsSources): seg u32 [B][nchunks][seg_stride] | ent u16 [B][nchunks][EMAX] | wsorted T [..]
//   generic                      : offsets i32 [B][N+1] | cursor i32 [B][N] | unsorted i32 [B][L] | sorted i32 [B][L]
struct ScatterWorkspace {
  uint32_t* seg;
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
    w.seg = static_cast<uint32_t*>(take(sizeof(uint32_t) * (size_t)B * nch * seg_stride(N, 3 * kChunkCols)));
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
// chunked path
// ------------------------------------------------------------------------------------------------------
// LDS: offs i32 [N+1] | cursor i32 [N] | srcl u16 [EMAX] | ul u16 [EMAX]
template <typename T, int MODE>
__global__ __launch_bounds__(kScatterThreads) void csr_chunk_build_kernel(
    const int64_t* __restrict__ idx, const T* __restrict__ w, uint32_t* __restrict__ seg_g,
    uint16_t* __restrict__ ent_g, T* __restrict__ wsorted_g, int N, int64_t L, int nchunks, int segs) {
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
  // One exclusive scan for two things: entries before source n (low 16 bits; a chunk holds < 2^16 entries) and
  // non-empty sources before n (high bits).  kItems consecutive sources per thread.
  int v[kItems];
  int sum = 0;
#pragma unroll
  for (int i = 0; i < kItems; ++i) {
    const int n = tid * kItems + i;
    const int c = n < N ? offs[n] : 0;
    v[i] = c + (c > 0 ? (1 << 16) : 0);
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
  uint32_t* sg = seg_g + ((int64_t)b * nchunks + k) * segs;
  // (Ordering the segment list by length, longest first, so that a wave of the sum kernel runs segments of one length:
  // measured 242 -> 217 us on the k = 128 sum kernel, but every way of computing the order here -- class counters with
  // LDS atomics, per wave or per workgroup; ballots; counters packed through the scan -- cost this kernel 17 us or more.)
#pragma unroll
  for (int i = 0; i < kItems; ++i) {
    const int n = tid * kItems + i;
    if (n < N) {
      const int first = run & 0xFFFF;
      offs[n] = first;
      cursor[n] = 0;
      if (v[i] != 0) sg[run >> 16] = (uint32_t)n | ((uint32_t)first << 16);
    }
    run += v[i];
  }
  if (tid == kScatterThreads - 1) offs[N] = run;  // entries | segments << 16 of this chunk
  __syncthreads();
  const int total = offs[N] & 0xFFFF, nseg = offs[N] >> 16;
  for (int i = nseg + tid; i < segs; i += kScatterThreads) sg[i] = kNoSegment;
  __syncthreads();
  if (tid == 0) offs[N] = total;  // end of the last source's entries for the rank pass
  for (int e = tid; e < ne; e += kScatterThreads) {
    const int src = srcl[e];
    if (src != 0xFFFF) ul[offs[src] + atomicAdd(&cursor[src], 1)] = (uint16_t)e;
  }
  __syncthreads();
  // rank sort inside every group (local positions are distinct: rank = number of smaller members)
  uint16_t* eg = ent_g + ((int64_t)b * nchunks + k) * EMAX;
  T* wg = MODE == 1 ? wsorted_g + ((int64_t)b * nchunks + k) * EMAX : nullptr;
  for (int s = tid; s < total; s += kScatterThreads) {
    const int e = ul[s];
    const int src = srcl[e];
    const int start = offs[src], end = offs[src + 1];
    int rank = 0;
    for (int j = start; j < end; ++j) rank += (ul[j] < e) ? 1 : 0;
    const int col = MODE == 1 ? e / 3 : e;
    eg[start + rank] = (uint16_t)(col | (start + rank == end - 1 ? 0x8000 : 0));
    if (MODE == 1) wg[start + rank] = w[(int64_t)b * L + t0 + e];
  }
}

// Registers that hold one chunk on its way from global memory to LDS.  Every thread issues the same number of
// loads for every chunk (addresses are clamped, never predicated), so that the compiler can wait for the older of
// the two register sets with a counted vmcnt while the younger set's loads stay in flight.
template <typename T, int MODE, int CT>
struct ChunkRegs {
  static constexpr int kVec = 16 / sizeof(T);
  static constexpr int kGroups = ChunkGeom<MODE>::P / kVec;            // 16-byte column groups per grad_out row
  static constexpr int kRows = CT * kGroups / kScatterThreads;         // rows (channels) a thread loads
  static constexpr int kParts = CT / kRows;                            // threads that share one column group
  // Column c's record (CT values) lives at slot swz(c): the low bits are XORed with the column-group number, so that
  // the kVec-strided columns a wave stores together fall into different banks (records of 16 or 32 bytes at a stride
  // of kVec records would otherwise hit 2-4 banks with all 64 lanes: measured 0.7 of the kernel's LDS-busy cycles).
  static constexpr int kRecsPerBankCycle = 256 / (CT * (int)sizeof(T));
  static constexpr int kVecShift = kVec == 4 ? 2 : 1;
  __device__ static __forceinline__ int swz(int c) { return c ^ ((c >> kVecShift) & (kRecsPerBankCycle - 1)); }
  static constexpr int kEntVecs = ChunkGeom<MODE>::EMAX * 2 / 16;
  static constexpr int kEntV = (kEntVecs + kScatterThreads - 1) / kScatterThreads;
  static constexpr int kWVecs = ChunkGeom<MODE>::EMAX * (int)sizeof(T) / 16;
  static constexpr int kWV = MODE == 1 ? (kWVecs + kScatterThreads - 1) / kScatterThreads : 1;
  static constexpr int kSegMax = ChunkGeom<MODE>::EMAX < kMaxLdsSources ? ChunkGeom<MODE>::EMAX : kMaxLdsSources;
  static constexpr int kSeg = (kSegMax + kScatterThreads - 1) / kScatterThreads;  // segment slots per thread
  static constexpr int kDepth = 2;  // chunks in flight besides the one being summed (3 and 4 measured no faster)
  static_assert(kRows >= 1 && CT * kGroups % kScatterThreads == 0, "tile must split evenly over the workgroup");
  static_assert(CT % kRows == 0 && kParts * kGroups == kScatterThreads && (kVec == 4 || kVec == 2), "thread -> tile piece map");
  static_assert(ChunkGeom<MODE>::P % kRecsPerBankCycle == 0 && (kRecsPerBankCycle & (kRecsPerBankCycle - 1)) == 0, "swizzle");
  uint4 t[kRows];
  uint4 e[kEntV];
  uint4 w[kWV];
  unsigned s[kSeg];
};

// LDS: acc[N][CT] | tile[P][CT] | wl[EMAX] (MODE 1) | ent[EMAX] (+ pad).  CT * sizeof(T) is 16 or 32 bytes.
// A thread loads one 16-byte column group (kVec columns) of kRows channels and writes it transposed, kRows
// channels of one column at a time.
template <typename T, int MODE, int CT, bool VEC_OK>
__global__ __launch_bounds__(kScatterThreads) void segment_sum_kernel_lds(
    const T* __restrict__ gout, const uint32_t* __restrict__ seg_g, const uint16_t* __restrict__ ent_g,
    const T* __restrict__ wsorted_g, T* __restrict__ gsrc, int C, int N, int64_t Lrow, int nchunks, int segs,
    GridMap gm) {
  using Regs = ChunkRegs<T, MODE, CT>;
  constexpr int P = ChunkGeom<MODE>::P;
  constexpr int EMAX = ChunkGeom<MODE>::EMAX;
  constexpr int TH = kScatterThreads;
  constexpr int kVec = Regs::kVec;
  constexpr int kRows = Regs::kRows;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* acc = reinterpret_cast<T*>(smem_raw);
  T* tile = acc + (size_t)N * CT;
  T* wl = tile + P * CT;
  uint16_t* ent = reinterpret_cast<uint16_t*>(wl + (MODE == 1 ? EMAX : 0));
  const int tid = threadIdx.x;
  int bx, by, b;
  decode_block(gm, bx, by, b);
  const int c0 = bx * CT;
  for (int i = tid; i < N * CT; i += TH) acc[i] = T(0);

  // this thread's tile piece: column group grp, rows part*kRows .. +kRows-1.  Neighbouring lanes take the parts of one
  // column group (together one whole record per column), then the next group: with the slot swizzle the transposed
  // stores below are free of bank conflicts, and a row is still read in runs of 64 / kParts adjacent 16-byte pieces.
  const int part = tid % Regs::kParts, grp = tid / Regs::kParts;

#define V3D_LOAD_CHUNK(r, kk)                                                                                   \
  do {                                                                                                          \
    const int k_ = min((kk), nchunks - 1);                                                                      \
    const int64_t chunk_ = (int64_t)b * nchunks + k_;                                                           \
    const int64_t col_ = (int64_t)k_ * P + (int64_t)grp * kVec;                                                 \
    _Pragma("unroll") for (int i = 0; i < kRows; ++i) {                                                         \
      const int c = c0 + part * kRows + i;                                                                      \
      const T* src = gout + ((int64_t)b * C + min(c, C - 1)) * Lrow;                                            \
      uint4 v; /* columns past the row end and channels past C are never referenced / written: clamp only */    \
      if (VEC_OK) {                                                                                             \
        v = *reinterpret_cast<const uint4*>(src + (col_ < Lrow ? col_ : 0));                                    \
      } else {                                                                                                  \
        T tmp[kVec];                                                                                            \
        _Pragma("unroll") for (int e = 0; e < kVec; ++e) tmp[e] = src[col_ + e < Lrow ? col_ + e : 0];          \
        v = *reinterpret_cast<const uint4*>(tmp);                                                               \
      }                                                                                                         \
      (r).t[i] = v;                                                                                             \
    }                                                                                                           \
    _Pragma("unroll") for (int i = 0; i < Regs::kEntV; ++i)                                                     \
      (r).e[i] = reinterpret_cast<const uint4*>(ent_g + chunk_ * EMAX)[min(i * TH + tid, Regs::kEntVecs - 1)];  \
    if (MODE == 1) {                                                                                            \
      _Pragma("unroll") for (int i = 0; i < Regs::kWV; ++i)                                                     \
        (r).w[i] = reinterpret_cast<const uint4*>(wsorted_g + chunk_ * EMAX)[min(i * TH + tid, Regs::kWVecs - 1)]; \
    }                                                                                                           \
    _Pragma("unroll") for (int i = 0; i < Regs::kSeg; ++i) {                                                    \
      const int v = i * TH + tid;                                                                               \
      (r).s[i] = seg_g[chunk_ * segs + min(v, segs - 1)];                                                       \
    }                                                                                                           \
  } while (0)

#define V3D_STORE_CHUNK(r)                                                                                      \
  do {                                                                                                          \
    /* (r).t[i] holds kVec columns of channel part*kRows+i: write kRows channels of one column together */      \
    _Pragma("unroll") for (int e = 0; e < kVec; ++e) {                                                          \
      T piece[kRows];                                                                                           \
      _Pragma("unroll") for (int i = 0; i < kRows; ++i) piece[i] = reinterpret_cast<const T*>(&(r).t[i])[e];    \
      T* dst = tile + (size_t)Regs::swz(grp * kVec + e) * CT + part * kRows;                                    \
      _Pragma("unroll") for (int i = 0; i < kRows; ++i) dst[i] = piece[i];                                      \
    }                                                                                                           \
    _Pragma("unroll") for (int i = 0; i < Regs::kEntV; ++i) {                                                   \
      const int v = i * TH + tid;                                                                               \
      if (v < Regs::kEntVecs) reinterpret_cast<uint4*>(ent)[v] = (r).e[i];                                      \
    }                                                                                                           \
    if (MODE == 1) {                                                                                            \
      _Pragma("unroll") for (int i = 0; i < Regs::kWV; ++i) {                                                   \
        const int v = i * TH + tid;                                                                             \
        if (v < Regs::kWVecs) reinterpret_cast<uint4*>(wl)[v] = (r).w[i];                                       \
      }                                                                                                         \
    }                                                                                                           \
  } while (0)

  // one thread per segment: acc[n] += its entries, in order; the next entry is fetched while this one's tile
  // values are on their way
#define V3D_SUM_CHUNK(segv)                                                                                     \
  do {                                                                                                          \
    _Pragma("unroll") for (int i = 0; i < Regs::kSeg; ++i) {                                                    \
      const unsigned sv = (segv)[i];                                                                            \
      if (sv != kNoSegment) {                                                                                   \
        const int n = (int)(sv & 0xFFFFu);                                                                      \
        int q = (int)(sv >> 16);                                                                                \
        unsigned en = ent[q];                                                                                   \
        T wt = MODE == 1 ? wl[q] : T(1);                                                                        \
        T a[CT];                                                                                                \
        _Pragma("unroll") for (int c = 0; c < CT; ++c) a[c] = acc[(size_t)n * CT + c];                          \
        bool last;                                                                                              \
        do {                                                                                                    \
          const T* tv = tile + (size_t)Regs::swz((int)(en & 0x7FFFu)) * CT;                                     \
          T x[CT];                                                                                              \
          _Pragma("unroll") for (int c = 0; c < CT; ++c) x[c] = tv[c];                                          \
          last = (en & 0x8000u) != 0;                                                                           \
          const T wcur = wt;                                                                                    \
          ++q;                                                                                                  \
          en = ent[q]; /* one past the chunk's last entry at most: the LDS allocation is padded */              \
          if (MODE == 1) wt = wl[min(q, EMAX - 1)];                                                             \
          _Pragma("unroll") for (int c = 0; c < CT; ++c) {                                                      \
            T y = x[c];                                                                                         \
            if (MODE == 1) y = y * wcur;                                                                        \
            a[c] = a[c] + y;                                                                                    \
          }                                                                                                     \
        } while (!last && q < EMAX); /* the bound only matters if the lists were corrupted */                   \
        _Pragma("unroll") for (int c = 0; c < CT; ++c) acc[(size_t)n * CT + c] = a[c];                          \
      }                                                                                                         \
    }                                                                                                           \
  } while (0)

  // kDepth register sets: chunks k+1 .. k+kDepth are in flight while chunk k is summed (one workgroup fills a CU, so
  // the loads in flight are all that hides the memory latency).  The loop starts kDepth chunks early (the first pass
  // only loads) so that each set is defined at ONE place: with a peeled prologue the register allocator joins the two
  // definitions through copies at the loop latch, and a copy waits for the load it copies.  Every set issues the same
  // number of loads, oldest set first, so the wait before a store is a counted vmcnt that leaves the younger sets in
  // flight.  A tail that is not a multiple of kDepth runs empty sets (no segments).
  constexpr int kDepth = Regs::kDepth;
  Regs r[kDepth];
  for (int k = -kDepth; k < nchunks; k += kDepth) {
#pragma unroll
    for (int j = 0; j < kDepth; ++j) {
      unsigned segv[Regs::kSeg];
      if (k >= 0) {
        __syncthreads();  // chunk k+j-1 consumed (first pass: acc zeroed)
        V3D_STORE_CHUNK(r[j]);
#pragma unroll
        for (int i = 0; i < Regs::kSeg; ++i)
          segv[i] = (i * TH + tid < segs && k + j < nchunks) ? r[j].s[i] : kNoSegment;
        __syncthreads();
      }
      V3D_LOAD_CHUNK(r[j], k + j + kDepth);
      if (k >= 0) V3D_SUM_CHUNK(segv);
    }
  }
#undef V3D_LOAD_CHUNK
#undef V3D_STORE_CHUNK
#undef V3D_SUM_CHUNK
  __syncthreads();
  for (int i = tid; i < N * CT; i += TH) {
    const int c = i / N, n = i - c * N;  // consecutive threads -> consecutive n of one channel: coalesced stores
    if (c0 + c < C) gsrc[((int64_t)b * C + c0 + c) * N + n] = acc[(size_t)n * CT + c];
  }
}

template <typename T, int MODE, int CT>
static int launch_chunked(const char* what, const T* gout, const ScatterWorkspace& ws, T* gsrc, int B, int C, int N,
                          int64_t Lrow, int nchunks, hipStream_t stream) {
  constexpr int P = ChunkGeom<MODE>::P;
  constexpr int EMAX = ChunkGeom<MODE>::EMAX;
  const size_t lds =
      sizeof(T) * CT * ((size_t)N + P) + (MODE == 1 ? sizeof(T) * EMAX : 0) + sizeof(uint16_t) * EMAX + 16;
  const bool vec_ok = (Lrow % (16 / sizeof(T)) == 0) && ((uintptr_t)gout % 16 == 0);
  auto kernel = vec_ok ? segment_sum_kernel_lds<T, MODE, CT, true> : segment_sum_kernel_lds<T, MODE, CT, false>;
  static bool attr_set[64][2] = {};  // per device and variant: the LDS request exceeds the 64 KiB default
  int dev = 0;
  V3D_HIP_TRY(hipGetDevice(&dev), what);
  if (dev < 0 || dev >= 64 || !attr_set[dev][vec_ok]) {
    V3D_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                what);
    if (dev >= 0 && dev < 64) attr_set[dev][vec_ok] = true;
  }
  const int64_t ctiles = ceil_div(C, CT);
  const GridMap gm = make_grid_map(ctiles, 1, B);
  hipLaunchKernelGGL(kernel, dim3((unsigned)(ctiles * B)), dim3(kScatterThreads), lds, stream, gout, ws.seg, ws.ent,
                     static_cast<const T*>(ws.wsorted), gsrc, C, N, Lrow, nchunks, seg_stride(N, EMAX), gm);
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

// The backward in two steps, so that the lists of one index tensor serve every gradient that flows through it (the xyz
// and the feature gather of one set-abstraction layer share `indices`, layers/pointnet2.py:44-46; every channel tile
// does anyway): scatter_build inverts the indices into the workspace, scatter_apply sums with them.
template <typename T, int MODE>
static int scatter_check(const char* what, int B, int C, int N, int64_t L) {
  V3D_REQUIRE(B >= 0 && C >= 0 && N >= 0 && L >= 0, what, "negative dimension");
  if (B > kMaxGridYZ || L > (int64_t)INT32_MAX) {
    set_last_error("%s: B <= 65535 and targets per cloud < 2^31 required", what);
    return V3D_ERR_SIZE_LIMIT;
  }
  if ((int64_t)ceil_div(C, 2) * B > (int64_t)INT32_MAX) {
    set_last_error("%s: launch exceeds the grid limit", what);
    return V3D_ERR_SIZE_LIMIT;
  }
  return 0;
}

static int scatter_check_workspace(const char* what, const void* base, size_t workspace_bytes, size_t need) {
  if (base == nullptr || workspace_bytes < need || ((uintptr_t)base & 255) != 0) {
    set_last_error("%s: workspace needs %zu bytes, 256-B aligned (got %zu)", what, need, workspace_bytes);
    return V3D_ERR_WORKSPACE;
  }
  return 0;
}

// Enqueues the inversion of idx (B, L) over N sources into `base` (carve_workspace layout).  MODE 1 also stores the
// weights in list order (chunked layout).  The device is current, sizes are checked, L > 0.
template <typename T, int MODE>
static int scatter_build_launch(const char* what, const int64_t* idx, const T* w, int B, int N, int64_t L, void* base,
                                hipStream_t stream) {
  ScatterWorkspace ws = carve_workspace(base, B, N, L);
  const int64_t Lrow = MODE == 1 ? L / 3 : L;
  const int nchunks = (int)std::max<int64_t>(1, ceil_div(Lrow, ChunkGeom<MODE>::P));
  if (N <= kMaxLdsSources) {
    constexpr int EMAX = ChunkGeom<MODE>::EMAX;
    const size_t lds = sizeof(int) * (2 * (size_t)N + 1) + 2 * sizeof(uint16_t) * EMAX;
    hipLaunchKernelGGL((csr_chunk_build_kernel<T, MODE>), dim3((unsigned)nchunks, (unsigned)B), dim3(kScatterThreads),
                       lds, stream, idx, w, ws.seg, ws.ent, static_cast<T*>(ws.wsorted), N, L, nchunks,
                       seg_stride(N, EMAX));
  } else {
    V3D_HIP_TRY(hipMemsetAsync(ws.cursor, 0, sizeof(int32_t) * (size_t)B * N, stream), what);
    const unsigned xb = (unsigned)std::min<int64_t>(ceil_div(L, 256), 1024);
    hipLaunchKernelGGL(csr_count_kernel, dim3(xb, B), dim3(256), 0, stream, idx, ws.cursor, N, L);
    hipLaunchKernelGGL(csr_scan_kernel, dim3(B), dim3(1024), 0, stream, ws.cursor, ws.offsets, N);
    hipLaunchKernelGGL(csr_fill_kernel, dim3(xb, B), dim3(256), 0, stream, idx, ws.offsets, ws.cursor, ws.unsorted, N,
                       L);
    hipLaunchKernelGGL(csr_rank_kernel, dim3(xb, B), dim3(256), 0, stream, idx, ws.offsets, ws.unsorted, ws.sorted, N,
                       L);
  }
  return launch_status(what);
}

// Enqueues the sums: gsrc (B, C, N) fully written from gout with the lists `base` holds.
template <typename T, int MODE>
static int scatter_apply_launch(const char* what, const T* gout, const T* w, T* gsrc, int B, int C, int N, int64_t L,
                                void* base, hipStream_t stream) {
  ScatterWorkspace ws = carve_workspace(base, B, N, L);
  const int64_t Lrow = MODE == 1 ? L / 3 : L;
  const int nchunks = (int)std::max<int64_t>(1, ceil_div(Lrow, ChunkGeom<MODE>::P));
  int rc = 0;
  if (N <= kMaxLdsSources) {
    // acc[N][CT] + tile[P][CT] + lists must fit 160 KiB: 32-byte channel vectors up to 1024 sources, else 16-byte
    constexpr int kQuad = 16 / sizeof(T);
    bool wide = false;
    if constexpr (sizeof(T) == 4 && MODE == 1) {
      // interpolation (short target rows, few chunks per workgroup): 16 channels per workgroup halve the list reads
      // and the per-entry bookkeeping, and C/16 x B workgroups fill the chip in one round at C2
      if (N <= 1024 && C >= 16) {
        wide = true;
        rc = launch_chunked<T, MODE, 16>(what, gout, ws, gsrc, B, C, N, Lrow, nchunks, stream);
      }
    }
    if constexpr (sizeof(T) == 4) {
      if (!wide && N <= 1024) {
        wide = true;
        rc = launch_chunked<T, MODE, 8>(what, gout, ws, gsrc, B, C, N, Lrow, nchunks, stream);
      }
    }
    if (!wide) rc = launch_chunked<T, MODE, kQuad>(what, gout, ws, gsrc, B, C, N, Lrow, nchunks, stream);
    if (rc != 0) return rc;
  } else {
    constexpr int CT = 8;
    dim3 grid((unsigned)ceil_div(N, 256), (unsigned)ceil_div(C, CT), (unsigned)B);
    hipLaunchKernelGGL((segment_sum_kernel<T, MODE, CT>), grid, dim3(256), 0, stream, gout, w, ws.offsets, ws.sorted,
                       gsrc, C, N, L, Lrow);
  }
  return launch_status(what);
}

template <typename T, int MODE>
int scatter_backward(const char* what, const T* gout, const int64_t* idx, const T* w, T* gsrc, int B, int C, int N,
                     int64_t L, void* workspace, size_t workspace_bytes, int device, void* stream_) {
  if (int rc = scatter_check<T, MODE>(what, B, C, N, L)) return rc;
  if (B == 0 || C == 0 || N == 0) return 0;
  V3D_REQUIRE(gsrc && (L == 0 || (gout && idx)) && (MODE == 0 || L == 0 || w), what, "null pointer");
  DeviceGuard guard(device);
  if (guard.rc) return guard.rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (L == 0) {
    V3D_HIP_TRY(hipMemsetAsync(gsrc, 0, sizeof(T) * (size_t)B * C * N, stream), what);
    return 0;
  }
  const size_t need = carve_workspace(nullptr, B, N, L).bytes;
  void* base = workspace;
  bool own = false;
  if (base == nullptr) {
    V3D_HIP_TRY(hipMallocAsync(&base, need, stream), what);
    own = true;
  } else if (int rc = scatter_check_workspace(what, base, workspace_bytes, need)) {
    return rc;
  }
  int rc = scatter_build_launch<T, MODE>(what, idx, w, B, N, L, base, stream);
  if (rc == 0) rc = scatter_apply_launch<T, MODE>(what, gout, w, gsrc, B, C, N, L, base, stream);
  if (own) {
    hipError_t e = hipFreeAsync(base, stream);
    if (rc == 0 && e != hipSuccess) {
      set_last_error("%s: hipFreeAsync: %s", what, hipGetErrorString(e));
      rc = (int)e;
    }
  }
  return rc;
}

template <typename T>
static int scatter_build_entry(const int64_t* idx, const T* w, int mode, int B, int N, int64_t L, void* workspace,
                               size_t workspace_bytes, int device, void* stream_) {
  const char* what = "scatter_build";
  V3D_REQUIRE(mode == 0 || mode == 1, what, "mode is 0 (gather / group_gather) or 1 (three_interpolate)");
  V3D_REQUIRE(mode == 0 || L % 3 == 0, what, "three_interpolate lists hold 3 entries per target");
  if (int rc = scatter_check<T, 0>(what, B, 1, N, L)) return rc;
  if (B == 0 || N == 0 || L == 0) return 0;
  V3D_REQUIRE(idx && (mode == 0 || w), what, "null pointer");
  if (int rc = scatter_check_workspace(what, workspace, workspace_bytes, carve_workspace(nullptr, B, N, L).bytes)) return rc;
  DeviceGuard guard(device);
  if (guard.rc) return guard.rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  return mode == 0 ? scatter_build_launch<T, 0>(what, idx, w, B, N, L, workspace, stream)
                   : scatter_build_launch<T, 1>(what, idx, w, B, N, L, workspace, stream);
}

template <typename T>
static int scatter_apply_entry(const T* gout, const T* w, T* gsrc, int mode, int B, int C, int N, int64_t L,
                               const void* workspace, size_t workspace_bytes, int device, void* stream_) {
  const char* what = "scatter_apply";
  V3D_REQUIRE(mode == 0 || mode == 1, what, "mode is 0 (gather / group_gather) or 1 (three_interpolate)");
  V3D_REQUIRE(mode == 0 || L % 3 == 0, what, "three_interpolate lists hold 3 entries per target");
  if (int rc = scatter_check<T, 0>(what, B, C, N, L)) return rc;
  if (B == 0 || C == 0 || N == 0) return 0;
  V3D_REQUIRE(gsrc && (L == 0 || gout) && (mode == 0 || L == 0 || w), what, "null pointer");
  DeviceGuard guard(device);
  if (guard.rc) return guard.rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (L == 0) {
    V3D_HIP_TRY(hipMemsetAsync(gsrc, 0, sizeof(T) * (size_t)B * C * N, stream), what);
    return 0;
  }
  if (int rc = scatter_check_workspace(what, workspace, workspace_bytes, carve_workspace(nullptr, B, N, L).bytes)) return rc;
  void* base = const_cast<void*>(workspace);  // read only by the kernels
  return mode == 0 ? scatter_apply_launch<T, 0>(what, gout, w, gsrc, B, C, N, L, base, stream)
                   : scatter_apply_launch<T, 1>(what, gout, w, gsrc, B, C, N, L, base, stream);
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

// ------------------------------------------------------------------------------------------------------
// The entry point that allocates nothing (v3d_three_interp_bwd_noalloc_f32): its scratch is a file-scope device
// array instead of a caller workspace / stream-ordered allocation.  Every GPU holds its own copy of the array at its
// own address until the process ends; the array's NAME is only a host-side stand-in, so the device address is looked
// up once per device with hipGetSymbolAddress (that device current) and kept.  Calls on one GPU share the array:
// overlapping calls on one GPU race (documented in include/v3d_hip.h).
// ------------------------------------------------------------------------------------------------------
constexpr size_t kNoAllocScratchBytes = (size_t)V3D_NOALLOC_SCRATCH_BYTES;
constexpr int kNoAllocMaxDevices = 64;
static __device__ __align__(256) unsigned char scratch[kNoAllocScratchBytes];
static void* g_noalloc_ptr[kNoAllocMaxDevices] = {};
static std::mutex g_noalloc_mutex;

// the device's address of `scratch`; `device` must be current (the caller holds a DeviceGuard)
static int noalloc_scratch_ptr(const char* what, int device, void** out) {
  if (device < 0 || device >= kNoAllocMaxDevices) {
    set_last_error("%s: device ordinal %d outside [0, %d)", what, device, kNoAllocMaxDevices);
    return V3D_ERR_INVALID_ARGUMENT;
  }
  std::lock_guard<std::mutex> lock(g_noalloc_mutex);
  if (g_noalloc_ptr[device] == nullptr) {
    void* ptr = nullptr;
    V3D_HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)), what);
    if (ptr == nullptr || ((uintptr_t)ptr & 255) != 0) {
      set_last_error("%s: hipGetSymbolAddress returned an unusable address", what);
      return V3D_ERR_INVALID_ARGUMENT;
    }
    g_noalloc_ptr[device] = ptr;
  }
  *out = g_noalloc_ptr[device];
  return 0;
}

}  // namespace v3d

extern "C" {

int v3d_three_interp_bwd_noalloc_f32(const float* g, const int64_t* idx, const float* w, float* gs, int B, int C, int S,
                                     int Q, int device, void* stream) {
  using namespace v3d;
  const char* what = "three_interpolate_backward (no-alloc)";
  V3D_REQUIRE(B >= 0 && C >= 0 && S >= 0 && Q >= 0, what, "negative dimension");
  if (B == 0 || C == 0 || S == 0) return 0;
  const size_t need = carve_workspace(nullptr, B, S, (int64_t)Q * 3).bytes;
  if (need > kNoAllocScratchBytes) {
    set_last_error("%s: B=%d, S=%d, Q=%d needs %zu bytes of scratch, the built-in array holds %zu; use "
                   "v3d_three_interp_bwd_f32 with a workspace", what, B, S, Q, need, kNoAllocScratchBytes);
    return V3D_ERR_SIZE_LIMIT;
  }
  void* ptr = nullptr;
  {
    DeviceGuard guard(device);  // hipGetSymbolAddress answers for the CURRENT device
    if (guard.rc) return guard.rc;
    const int rc = noalloc_scratch_ptr(what, device, &ptr);
    if (rc) return rc;
  }
  // from here on the array is `ptr`, never its name: the kernels take it as the workspace argument
  return scatter_backward<float, 1>(what, g, idx, w, gs, B, C, S, (int64_t)Q * 3, ptr, kNoAllocScratchBytes, device,
                                    stream);
}

int v3d_scatter_build(const int64_t* indices, const void* weights, int mode, int elem_bytes, int B, int num_sources,
                      int64_t positions_per_batch, void* workspace, size_t workspace_bytes, int device, void* stream) {
  if (mode == 1 && elem_bytes == 8)
    return v3d::scatter_build_entry<double>(indices, static_cast<const double*>(weights), mode, B, num_sources,
                                            positions_per_batch, workspace, workspace_bytes, device, stream);
  if (mode == 1 && elem_bytes != 4) return v3d::invalid("scatter_build", "elem_bytes is 4 (f32) or 8 (f64)");
  return v3d::scatter_build_entry<float>(indices, static_cast<const float*>(weights), mode, B, num_sources,
                                         positions_per_batch, workspace, workspace_bytes, device, stream);
}
int v3d_scatter_apply_f32(const float* target_grads, const float* weights, float* source_grads, int mode, int B, int C,
                          int num_sources, int64_t positions_per_batch, const void* workspace, size_t workspace_bytes,
                          int device, void* stream) {
  return v3d::scatter_apply_entry<float>(target_grads, weights, source_grads, mode, B, C, num_sources,
                                         positions_per_batch, workspace, workspace_bytes, device, stream);
}
int v3d_scatter_apply_f64(const double* target_grads, const double* weights, double* source_grads, int mode, int B, int C,
                          int num_sources, int64_t positions_per_batch, const void* workspace, size_t workspace_bytes,
                          int device, void* stream) {
  return v3d::scatter_apply_entry<double>(target_grads, weights, source_grads, mode, B, C, num_sources,
                                          positions_per_batch, workspace, workspace_bytes, device, stream);
}

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
