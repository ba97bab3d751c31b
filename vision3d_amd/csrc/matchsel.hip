// matchsel.hip -- the two remaining selectors of vision3d/ops/matching.py:
//   * threshold selection, `extract_correspondences_from_scores_threshold` (:64-86): torch.gt -> torch.nonzero -> gather;
//   * global top-k, `extract_correspondences_from_scores_topk` (:89-121) and the flat top-k at the end of
//     SuperPointMatchingGlobalTopk (models/geotransformer/superpoint_matching_global_topk.py:61-64):
//     score_mat.view(-1).topk(k) -> div / mod -> dustbin masks.
//
// Design (gfx950).  Both are HBM-read-bound compactions of an (N, M) fp32 matrix; nothing of the matrix's size is written.
//   threshold: a thread owns 16 CONSECUTIVE cells (four 16-byte loads), a workgroup 4096; pass 1 counts the hits of every
//     tile, a scan of the tile counts gives each tile its first output slot, pass 2 re-reads and writes the hits in cell
//     order = torch.nonzero order.  Matrices of up to 64 Ki cells are one launch of one workgroup.
//   top-k: an exact radix select on the monotone integer image of the score (11 + 11 + 10 bits: three histogram passes,
//     LDS histograms flushed with integer atomics; the workgroup that finishes last picks the bin) finds the k-th best
//     value T; if more cells equal T than are still needed, three more passes select on the CELL INDEX among them (so that
//     equal scores are taken lowest index first -- torch.topk leaves that order open); otherwise those passes return at
//     once.  A last pass appends the k winners as (key << 32 | cell) words through one atomic per wave, and the k words
//     are sorted (one workgroup in LDS up to 4096, the device radix sort above) = best score first, lower cell first.
//     Dustbin rows / columns are dropped after the sort, as the reference does.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cmath>

#include "common.h"

namespace v3d {

constexpr int kMsTile = 4096;       // cells per workgroup in the threshold passes (256 threads x 16)
constexpr int kMsSmallCells = 65536;
constexpr int kMsBins = 2048;
constexpr int kMsSortSmall = 4096;

// ascending unsigned image: a better score has a SMALLER key; -0 and +0 share one key
__device__ __forceinline__ unsigned ms_key(float f, bool largest) {
  f = (largest ? -f : f) + 0.0f;
  const int b = __float_as_int(f);
  return (unsigned)(b ^ ((b >> 31) & 0x7fffffff)) ^ 0x80000000u;
}

// The 16 consecutive cells [base, base + 16) of a thread; cells past `total` read as `fill`.
__device__ __forceinline__ void ms_load16(const float* __restrict__ score, int64_t base, int64_t total, bool vec,
                                          float fill, float (&v)[16]) {
  if (vec && base + 16 <= total) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 t = *reinterpret_cast<const float4*>(score + base + 4 * j);
      v[4 * j + 0] = t.x;
      v[4 * j + 1] = t.y;
      v[4 * j + 2] = t.z;
      v[4 * j + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = base + j < total ? score[base + j] : fill;
  }
}

__device__ __forceinline__ unsigned ms_hits16(const float (&v)[16], float threshold, bool largest) {
  unsigned m = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) m |= ((largest ? v[j] > threshold : v[j] < threshold) ? 1u : 0u) << j;
  return m;
}

// exclusive prefix of `mine` over the workgroup (any multiple of 64 threads up to 1024); `total` = sum over all threads
__device__ __forceinline__ int ms_block_scan(int mine, int* wave_tot, int& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
  int incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  __syncthreads();  // wave_tot may still be read from a previous use
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  int run = incl - mine;
  total = 0;
  for (int w = 0; w < waves; ++w) {
    if (w < wave) run += wave_tot[w];
    total += wave_tot[w];
  }
  return run;
}

__device__ __forceinline__ void ms_emit_cell(int64_t cell, int M, float value, int64_t at, int64_t capacity,
                                             int64_t* __restrict__ out_r, int64_t* __restrict__ out_c,
                                             float* __restrict__ out_s) {
  if (at >= capacity) return;
  if (out_r) out_r[at] = cell / M;
  if (out_c) out_c[at] = cell % M;
  if (out_s) out_s[at] = value;
}

// ---- threshold selection --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ms_thr_count_kernel(const float* __restrict__ score, int64_t total, float threshold,
                                                           int largest, int vec, int* __restrict__ tile_counts) {
  __shared__ int wave_tot[4];
  float v[16];
  const int64_t base = (int64_t)blockIdx.x * kMsTile + (int64_t)threadIdx.x * 16;
  ms_load16(score, base, total, vec != 0, threshold, v);  // the threshold itself is never a hit
  int sum = 0;
  (void)ms_block_scan(__builtin_popcount(ms_hits16(v, threshold, largest != 0)), wave_tot, sum);
  if (threadIdx.x == 0) tile_counts[blockIdx.x] = sum;
}

__global__ __launch_bounds__(256) void ms_thr_emit_kernel(const float* __restrict__ score, int64_t total, int M,
                                                          float threshold, int largest, int vec,
                                                          const int* __restrict__ tile_counts,
                                                          const int* __restrict__ tile_slots, int64_t* __restrict__ out_r,
                                                          int64_t* __restrict__ out_c, float* __restrict__ out_s,
                                                          int64_t* __restrict__ num_out, int64_t capacity) {
  __shared__ int wave_tot[4];
  float v[16];
  const int64_t base = (int64_t)blockIdx.x * kMsTile + (int64_t)threadIdx.x * 16;
  ms_load16(score, base, total, vec != 0, threshold, v);
  const unsigned hits = ms_hits16(v, threshold, largest != 0);
  int sum = 0;
  int64_t at = (int64_t)tile_slots[blockIdx.x] + ms_block_scan(__builtin_popcount(hits), wave_tot, sum);
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0 && num_out)
    *num_out = (int64_t)tile_slots[blockIdx.x] + tile_counts[blockIdx.x];
#pragma unroll
  for (int j = 0; j < 16; ++j)
    if (hits & (1u << j)) ms_emit_cell(base + j, M, v[j], at++, capacity, out_r, out_c, out_s);
}

// up to 64 Ki cells: one workgroup, thread t owns the cells [64 t, 64 t + 64)
__global__ __launch_bounds__(1024) void ms_thr_small_kernel(const float* __restrict__ score, int total, int M,
                                                            float threshold, int largest, int vec,
                                                            int64_t* __restrict__ out_r, int64_t* __restrict__ out_c,
                                                            float* __restrict__ out_s, int64_t* __restrict__ num_out,
                                                            int64_t capacity) {
  __shared__ int wave_tot[16];
  unsigned hits[4];
  int mine = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    float v[16];
    ms_load16(score, (int64_t)threadIdx.x * 64 + u * 16, total, vec != 0, threshold, v);
    hits[u] = ms_hits16(v, threshold, largest != 0);
    mine += __builtin_popcount(hits[u]);
  }
  int sum = 0;
  int64_t at = ms_block_scan(mine, wave_tot, sum);
  if (threadIdx.x == 0 && num_out) *num_out = sum;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    for (unsigned m = hits[u]; m; m &= m - 1) {
      const int64_t cell = (int64_t)threadIdx.x * 64 + u * 16 + __builtin_ctz(m);
      ms_emit_cell(cell, M, score[cell], at++, capacity, out_r, out_c, out_s);
    }
  }
}

// ---- global top-k -----------------------------------------------------------------------------------------------
constexpr int kMsSamples = 8192;  // keys of the pivot sample (32 KiB of LDS)

struct MsState {
  unsigned key;        // resolved high bits of the k-th best key T
  unsigned cell;       // resolved high bits of the last cell taken among the cells equal to T
  unsigned remaining;  // entries still to take among the cells that match the resolved bits
  unsigned all_equal;  // 1: every cell equal to T is taken -- the three index passes return at once
  unsigned taken;      // append cursor of the last pass
  unsigned ticket[6];
  unsigned pivot;      // sampled key bound of the short list
  unsigned listed;     // cells with key <= pivot (all of them, also past the list's capacity)
  unsigned pad[3];
};
static_assert(sizeof(MsState) == 64, "MsState layout");

__device__ __forceinline__ void ms_pass_bits(int pass, int& shift, unsigned& mask) {
  const int p = pass % 3;
  shift = p == 0 ? 21 : (p == 1 ? 10 : 0);
  mask = p == 2 ? 1023u : 2047u;
}

// Short list (k << N*M).  ONE workgroup reads 8192 jittered, evenly spread cells and finds the rank-th smallest key of
// the sample by an 8-bit radix select in LDS: the pivot.  `rank` is chosen on the host so that the k-th best cell of the
// matrix is below the pivot with overwhelming probability and about rank/8192 of all cells are.
__global__ __launch_bounds__(1024) void ms_sample_kernel(const float* __restrict__ score, int64_t total, int largest,
                                                         unsigned rank, MsState* __restrict__ st) {
  __shared__ unsigned keys[kMsSamples];
  __shared__ unsigned hist[256];
  __shared__ unsigned s_prefix, s_rem;
  const int tid = threadIdx.x;
  const int64_t stride = total / kMsSamples;  // >= 32 (host)
#pragma unroll
  for (int u = 0; u < kMsSamples / 1024; ++u) {
    const unsigned i = (unsigned)(tid + u * 1024);
    const unsigned mix = (i * 2654435761u) ^ ((i * 2246822519u) >> 13);
    keys[i] = ms_key(score[(int64_t)i * stride + (int64_t)(mix % (unsigned long long)stride)], largest != 0);
  }
  unsigned prefix = 0, rem = rank;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kMsSamples / 1024; ++u) {
      const unsigned key = keys[tid + u * 1024];
      if (pass == 0 || (key >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid < 64) {  // lane l owns bins 4l .. 4l+3
      unsigned c[4], mine = 0;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        c[u] = hist[tid * 4 + u];
        mine += c[u];
      }
      unsigned incl = mine;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const unsigned t = __shfl_up(incl, d, 64);
        if (tid >= d) incl += t;
      }
      const unsigned before = incl - mine;
      if (before < rem && rem <= incl) {  // exactly one lane
        unsigned run = before;
        int u = 0;
        while (run + c[u] < rem) run += c[u++];
        s_prefix = prefix | ((unsigned)(tid * 4 + u) << shift);
        s_rem = rem - run;
      }
    }
    __syncthreads();
    prefix = s_prefix;
    rem = s_rem;
    __syncthreads();
  }
  if (tid == 0) st->pivot = prefix;
}

// one atomic per wave: every lane's first slot for its `mine` entries
__device__ __forceinline__ unsigned ms_wave_append(unsigned* cursor, int mine) {
  const int lane = threadIdx.x & 63;
  int incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  const int wave_sum = __shfl(incl, 63, 64);
  unsigned first = 0;
  if (wave_sum) {
    if (lane == 0) first = atomicAdd(cursor, (unsigned)wave_sum);
    first = __shfl(first, 0, 64);
  }
  return first + (unsigned)(incl - mine);
}

// Workgroup-buffered append: hits collect in LDS and reach the global list through ONE atomic per flush instead of one
// per wave and step (65k atomics on one address cost more than reading a 256 MiB matrix).  Entries that find the LDS
// buffer full go to the global list directly (one more wave-wide atomic); a flush happens when the buffer is half full.
constexpr int kMsBuf = 2048;
struct MsAppender {
  unsigned long long buf[kMsBuf];
  unsigned count, base;
};
__device__ __forceinline__ void ms_push16(MsAppender& A, const unsigned long long (&w)[16], unsigned hits,
                                          unsigned* cursor, unsigned long long* __restrict__ out, unsigned cap) {
  if (!__any(hits != 0)) return;
  const int mine = __builtin_popcount(hits);
  unsigned at = ms_wave_append(&A.count, mine);
  const int over = (int)min((unsigned)mine, at + mine > (unsigned)kMsBuf ? at + mine - max(at, (unsigned)kMsBuf) : 0u);
  unsigned gat = 0;
  if (__any(over != 0)) gat = ms_wave_append(cursor, over);
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if (hits & (1u << j)) {
      if (at < (unsigned)kMsBuf) {
        A.buf[at] = w[j];
      } else {
        if (gat < cap) out[gat] = w[j];
        ++gat;
      }
      ++at;
    }
  }
}
// all threads of the workgroup; returns the first global slot of this flush (the caller's overflow test)
__device__ __forceinline__ unsigned ms_flush(MsAppender& A, unsigned* cursor, unsigned long long* __restrict__ out,
                                             unsigned cap) {
  __syncthreads();
  const unsigned n = min(A.count, (unsigned)kMsBuf);
  if (threadIdx.x == 0) A.base = n ? atomicAdd(cursor, n) : 0u;
  __syncthreads();
  const unsigned base = A.base;
  for (unsigned i = threadIdx.x; i < n; i += blockDim.x)
    if (base + i < cap) out[base + i] = A.buf[i];
  __syncthreads();
  if (threadIdx.x == 0) A.count = 0;
  __syncthreads();
  return base + n;
}

// The one full read of the short-list route: cells with key <= pivot become (key << 32 | cell) words.
__global__ __launch_bounds__(512) void ms_list_kernel(const float* __restrict__ score, int64_t total, int largest, int vec,
                                                      MsState* __restrict__ st, unsigned long long* __restrict__ list,
                                                      unsigned list_cap) {
  __shared__ MsAppender A;
  if (threadIdx.x == 0) A.count = 0;
  __syncthreads();
  const unsigned pivot = st->pivot;
  const int64_t stride = (int64_t)gridDim.x * 512 * 16;
  for (int64_t first = (int64_t)blockIdx.x * 512 * 16; first < total; first += stride) {  // uniform over the workgroup
    const int64_t base = first + (int64_t)threadIdx.x * 16;
    float v[16];
    ms_load16(score, base, total, vec != 0, 0.f, v);
    unsigned long long w[16];
    unsigned hits = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const unsigned key = ms_key(v[j], largest != 0);
      hits |= ((base + j < total && key <= pivot) ? 1u : 0u) << j;
      w[j] = ((unsigned long long)key << 32) | (unsigned)(base + j);
    }
    ms_push16(A, w, hits, &st->listed, list, list_cap);
    __syncthreads();
    if (A.count >= (unsigned)kMsBuf / 2 && ms_flush(A, &st->listed, list, list_cap) > list_cap)
      return;  // the list has overflowed (ties, or a pivot that admits too much): it will be ignored, stop feeding it
  }
  (void)ms_flush(A, &st->listed, list, list_cap);
}

// The selection runs on the short list when it holds the answer: at least k cells, none lost to the capacity.
__device__ __forceinline__ bool ms_use_list(const MsState* st, unsigned k, unsigned list_cap) {
  return list_cap != 0 && st->listed >= k && st->listed <= list_cap;
}

// pass 0..2: histogram of the next key bits over the cells whose higher key bits equal state.key;
// pass 3..5: histogram of the next CELL-INDEX bits over the cells with key == T whose higher index bits match.
// Source: the short list if it is valid, else the matrix.
__global__ __launch_bounds__(512) void ms_hist_kernel(const float* __restrict__ score, int64_t total, int largest, int vec,
                                                      int pass, unsigned k, MsState* __restrict__ st,
                                                      unsigned* __restrict__ hist,
                                                      const unsigned long long* __restrict__ list, unsigned list_cap) {
  __shared__ unsigned h[kMsBins];
  __shared__ unsigned wave_tot[8];
  __shared__ int is_last;
  const unsigned skey = st->key, scell = st->cell;
  if (pass >= 3 && st->all_equal) return;  // uniform over the grid: written by the pass-2 pick before this launch began
  // The grid is sized for the matrix.  On the short list only a few workgroups take part: each one ends with same-address
  // global atomics (histogram flush, ticket) that serialise in the L2 -- 256 workgroups spent 30 us per pass there, on a
  // list that sixteen read in five.
  const bool on_list = ms_use_list(st, k, list_cap);
  const unsigned nblk = on_list ? min(gridDim.x, max(1u, st->listed / (512u * 16u))) : gridDim.x;
  if (blockIdx.x >= nblk) return;
  int shift;
  unsigned mask;
  ms_pass_bits(pass, shift, mask);
  const int hi = shift + (mask == 2047u ? 11 : 10);  // bits above the ones this pass resolves (32 in passes 0 and 3)
  for (int i = threadIdx.x; i < kMsBins; i += 512) h[i] = 0;
  __syncthreads();
  unsigned* gh = hist + (size_t)pass * kMsBins;
  auto count = [&](unsigned key, unsigned cell, bool valid) {  // called by whole waves (ballots inside)
    bool on;
    unsigned bin;
    if (pass < 3) {
      on = hi >= 32 || (key >> hi) == (skey >> hi);
      bin = (key >> shift) & mask;
    } else {
      on = key == skey && (hi >= 32 || (cell >> hi) == (scell >> hi));
      bin = (cell >> shift) & mask;
    }
    on = on && valid;
    // keys under one pivot share their high bits, tied scores share all of them: when every counting lane of the wave
    // names the same bin, one lane adds the count (same-address LDS atomics serialise)
    const unsigned long long voters = __ballot(on);
    if (voters == 0) return;
    const unsigned bin0 = (unsigned)__shfl((int)bin, __builtin_ctzll(voters), 64);
    if (__ballot(on && bin != bin0) == 0) {
      if ((threadIdx.x & 63) == (unsigned)__builtin_ctzll(voters)) atomicAdd(&h[bin0], (unsigned)__builtin_popcountll(voters));
    } else if (on) {
      atomicAdd(&h[bin], 1u);
    }
  };
  if (on_list) {
    const unsigned n = st->listed;
    for (unsigned i0 = blockIdx.x * 512; i0 < n; i0 += nblk * 512) {
      const unsigned i = i0 + threadIdx.x;
      const unsigned long long w = i < n ? list[i] : 0ull;
      count((unsigned)(w >> 32), (unsigned)w, i < n);
    }
  } else {
    const int64_t stride = (int64_t)gridDim.x * 512 * 16;
    for (int64_t base = ((int64_t)blockIdx.x * 512 + threadIdx.x) * 16; __any(base < total); base += stride) {
      float v[16];
      ms_load16(score, base, total, vec != 0, 0.f, v);
#pragma unroll
      for (int j = 0; j < 16; ++j) count(ms_key(v[j], largest != 0), (unsigned)(base + j), base + j < total);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kMsBins; i += 512)
    if (h[i]) atomicAdd(&gh[i], h[i]);
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) is_last = atomicAdd(&st->ticket[pass], 1u) == nblk - 1 ? 1 : 0;
  __syncthreads();
  if (!is_last) return;
  __threadfence();
  // the last workgroup: thread t owns bins 4t .. 4t+3; find the bin in which the running count reaches `remaining`
  const unsigned remaining = pass == 0 ? k : st->remaining;
  unsigned c[4], mine = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    c[u] = __hip_atomic_load(&gh[threadIdx.x * 4 + u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    mine += c[u];
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  unsigned before = incl - mine;
  for (int w = 0; w < wave; ++w) before += wave_tot[w];
  if (before < remaining && remaining <= before + mine) {  // exactly one thread
    unsigned run = before;
    int u = 0;
    while (run + c[u] < remaining) run += c[u++];
    const unsigned bin = threadIdx.x * 4 + u, left = remaining - run;
    if (pass < 3) {
      st->key = skey | (bin << shift);
      if (pass == 2) {
        const bool all = c[u] == left;
        st->all_equal = all ? 1u : 0u;
        st->cell = all ? 0xffffffffu : 0u;
      }
    } else {
      st->cell = scell | (bin << shift);
    }
    st->remaining = left;
  }
}

// cells with key < T, and cells with key == T up to cell index `state.cell`, are the k winners
__global__ __launch_bounds__(512) void ms_take_kernel(const float* __restrict__ score, int64_t total, int largest, int vec,
                                                      unsigned k, MsState* __restrict__ st,
                                                      unsigned long long* __restrict__ cand,
                                                      const unsigned long long* __restrict__ list, unsigned list_cap) {
  const unsigned T = st->key, tcell = st->cell;
  const unsigned long long bound = ((unsigned long long)T << 32) | tcell;  // words compare like (key, cell) pairs
  if (ms_use_list(st, k, list_cap)) {
    const unsigned n = st->listed;
    const unsigned nblk = min(gridDim.x, max(1u, n / (512u * 16u)));
    if (blockIdx.x >= nblk) return;
    for (unsigned i0 = blockIdx.x * 512; i0 < n; i0 += nblk * 512) {
      const unsigned i = i0 + threadIdx.x;
      const unsigned long long w = i < n ? list[i] : ~0ull;
      const bool on = i < n && w <= bound;
      const unsigned at = ms_wave_append(&st->taken, on ? 1 : 0);
      if (on && at < k) cand[at] = w;
    }
    return;
  }
  __shared__ MsAppender A;
  if (threadIdx.x == 0) A.count = 0;
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * 512 * 16;
  for (int64_t first = (int64_t)blockIdx.x * 512 * 16; first < total; first += stride) {  // uniform over the workgroup
    const int64_t base = first + (int64_t)threadIdx.x * 16;
    float v[16];
    ms_load16(score, base, total, vec != 0, 0.f, v);
    unsigned long long w[16];
    unsigned hits = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      w[j] = ((unsigned long long)ms_key(v[j], largest != 0) << 32) | (unsigned)(base + j);
      hits |= ((base + j < total && w[j] <= bound) ? 1u : 0u) << j;
    }
    ms_push16(A, w, hits, &st->taken, cand, k);
    __syncthreads();
    if (A.count >= (unsigned)kMsBuf / 2) (void)ms_flush(A, &st->taken, cand, k);
  }
  (void)ms_flush(A, &st->taken, cand, k);
}

// k <= 4096 winners: bitonic sort in LDS, dustbin filter, scan, emit -- one workgroup
__global__ __launch_bounds__(1024) void ms_sort_emit_small_kernel(const unsigned long long* __restrict__ cand, int k,
                                                                  const float* __restrict__ score, int N, int M,
                                                                  int has_dustbin, int64_t* __restrict__ out_r,
                                                                  int64_t* __restrict__ out_c, float* __restrict__ out_s,
                                                                  int64_t* __restrict__ num_out, int64_t capacity) {
  __shared__ unsigned long long s[kMsSortSmall];
  __shared__ int wave_tot[16];
  const int tid = threadIdx.x;
  int P = 64;
  while (P < k) P <<= 1;
  for (int e = tid; e < P; e += 1024) s[e] = e < k ? cand[e] : ~0ull;
  __syncthreads();
  for (int size = 2; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (P >> 1); t += 1024) {
        const int lo = ((t & ~(stride - 1)) << 1) | (t & (stride - 1)), hi = lo | stride;
        const bool up = (lo & size) == 0;
        const unsigned long long a = s[lo], b = s[hi];
        if ((a > b) == up) {
          s[lo] = b;
          s[hi] = a;
        }
      }
      __syncthreads();
    }
  }
  int keep[4], mine = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = tid * 4 + u;
    bool on = i < k;
    if (on && has_dustbin) {  // matching.py:110-115
      const unsigned cell = (unsigned)s[i];
      on = (int)(cell / (unsigned)M) != N - 1 && (int)(cell % (unsigned)M) != M - 1;
    }
    keep[u] = on ? 1 : 0;
    mine += keep[u];
  }
  int sum = 0;
  int64_t at = ms_block_scan(mine, wave_tot, sum);
  if (tid == 0 && num_out) *num_out = sum;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (!keep[u]) continue;
    const int64_t cell = (int64_t)(unsigned)s[tid * 4 + u];
    ms_emit_cell(cell, M, out_s ? score[cell] : 0.f, at++, capacity, out_r, out_c, out_s);
  }
}

__global__ __launch_bounds__(256) void ms_keep_kernel(const unsigned long long* __restrict__ sorted, int64_t k, int N, int M,
                                                      int has_dustbin, int* __restrict__ keep) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= k) return;
  bool on = true;
  if (has_dustbin) {
    const unsigned cell = (unsigned)sorted[i];
    on = (int)(cell / (unsigned)M) != N - 1 && (int)(cell % (unsigned)M) != M - 1;
  }
  keep[i] = on ? 1 : 0;
}

__global__ __launch_bounds__(256) void ms_emit_sorted_kernel(const unsigned long long* __restrict__ sorted,
                                                             const int* __restrict__ keep, const int* __restrict__ slots,
                                                             int64_t k, const float* __restrict__ score, int M,
                                                             int64_t* __restrict__ out_r, int64_t* __restrict__ out_c,
                                                             float* __restrict__ out_s, int64_t* __restrict__ num_out,
                                                             int64_t capacity) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= k) return;
  if (i == k - 1 && num_out) *num_out = (int64_t)slots[i] + keep[i];
  if (!keep[i]) return;
  const int64_t cell = (int64_t)(unsigned)sorted[i];
  ms_emit_cell(cell, M, out_s ? score[cell] : 0.f, slots[i], capacity, out_r, out_c, out_s);
}

static inline int ms_vec_ok(const float* score) { return ((uintptr_t)score & 15) == 0 ? 1 : 0; }

struct ThrLayout {
  size_t o_slots, o_tmp, scan_bytes, total;
};
static int thr_layout(int64_t tiles, ThrLayout& t) {
  t.scan_bytes = 0;
  int* ni = nullptr;
  hipError_t e = rocprim::exclusive_scan(nullptr, t.scan_bytes, ni, ni, 0, (size_t)tiles, rocprim::plus<int>(), nullptr);
  if (e != hipSuccess) return (int)e;
  t.o_slots = align_up(4 * (size_t)tiles, 256);
  t.o_tmp = 2 * t.o_slots;
  t.total = t.o_tmp + align_up(t.scan_bytes, 256);
  return 0;
}

// scratch of v3d_global_topk_select_f32 and the short-list parameters (see the entry point)
struct GtkLayout {
  bool small_sort;
  unsigned rank, list_cap;
  size_t sort_bytes, scan_bytes, o_state, o_hist, zero_bytes, o_cand, o_list, o_sorted, o_keep, o_slot, o_tmp, total;
};
static int gtk_layout(int64_t total, int64_t k, GtkLayout& g) {
  g.small_sort = k <= kMsSortSmall;
  g.sort_bytes = g.scan_bytes = 0;
  if (!g.small_sort) {
    unsigned long long* nk = nullptr;
    int* ni = nullptr;
    hipError_t e = rocprim::radix_sort_keys(nullptr, g.sort_bytes, nk, nk, (size_t)k, 0, 64, nullptr);
    if (e == hipSuccess)
      e = rocprim::exclusive_scan(nullptr, g.scan_bytes, ni, ni, 0, (size_t)k, rocprim::plus<int>(), nullptr);
    if (e != hipSuccess) return (int)e;
  }
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t at = off;
    off += align_up(bytes, 256);
    return at;
  };
  g.o_state = take(sizeof(MsState));
  g.o_hist = take(6 * kMsBins * sizeof(unsigned));
  g.zero_bytes = off;
  // Short-list route (k a small fraction of the matrix): rank of the pivot in the 8192-cell sample = expected number of
  // sampled cells among the k best, plus six standard deviations, plus 16; the list holds three times the expected
  // number of cells under the pivot.  Whatever the data does (ties, adversarial layouts), a list that overflows or holds
  // fewer than k cells is ignored and the selection reads the matrix: the route changes the speed, never the result.
  g.rank = g.list_cap = 0;
  if (total >= (int64_t)kMsSamples * 128) {
    const double expect = (double)kMsSamples * (double)k / (double)total;
    const double r = std::ceil(expect + 6.0 * std::sqrt(expect) + 16.0);
    const double cap = 3.0 * r * (double)total / kMsSamples + 4096.0;
    if (r <= kMsSamples / 24 && cap <= (double)total / 8) {
      g.rank = (unsigned)r;
      g.list_cap = (unsigned)cap;
    }
  }
  g.o_cand = take(8 * (size_t)k);
  g.o_list = take(8 * (size_t)g.list_cap);
  g.o_sorted = g.small_sort ? 0 : take(8 * (size_t)k);
  g.o_keep = g.small_sort ? 0 : take(4 * (size_t)k);
  g.o_slot = g.small_sort ? 0 : take(4 * (size_t)k);
  g.o_tmp = g.small_sort ? 0 : take(std::max(g.sort_bytes, g.scan_bytes));
  g.total = off;
  return 0;
}

}  // namespace v3d

extern "C" {

size_t v3d_threshold_select_workspace_bytes(int64_t N, int64_t M) {
  if (N <= 0 || M <= 0 || N * M <= v3d::kMsSmallCells) return 0;
  v3d::ThrLayout t;
  return v3d::thr_layout(v3d::ceil_div(N * M, v3d::kMsTile), t) == 0 ? t.total : 0;
}
size_t v3d_global_topk_select_workspace_bytes(int64_t N, int64_t M, int64_t k) {
  if (N <= 0 || M <= 0 || k <= 0 || k > N * M) return 0;
  v3d::GtkLayout g;
  return v3d::gtk_layout(N * M, k, g) == 0 ? g.total : 0;
}

int v3d_threshold_select_f32(const float* score, int64_t N, int64_t M, float threshold, int flags, int64_t* out_row,
                             int64_t* out_col, float* out_score, int64_t* num_out, int64_t capacity, void* workspace,
                             size_t workspace_bytes, int device, void* stream_) {
  using namespace v3d;
  const char* what = "threshold_select";
  V3D_REQUIRE(N >= 0 && M >= 0 && capacity >= 0, what, "negative dimension");
  V3D_REQUIRE(num_out, what, "num_out is required");
  V3D_REQUIRE(M <= INT32_MAX, what, "M must fit 31 bits");
  const int64_t total = N * M;
  const int64_t tiles = ceil_div(total, kMsTile);
  if (tiles >= (int64_t)INT32_MAX) {
    set_last_error("%s: N*M must stay below 2^43", what);
    return V3D_ERR_SIZE_LIMIT;
  }
  DeviceGuard guard(device);
  if (guard.rc) return guard.rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (total == 0) {
    V3D_HIP_TRY(hipMemsetAsync(num_out, 0, sizeof(int64_t), stream), what);
    return 0;
  }
  V3D_REQUIRE(score, what, "null pointer");
  const int largest = (flags & V3D_SEL_LARGEST) ? 1 : 0, vec = ms_vec_ok(score);
  if (total <= kMsSmallCells) {
    hipLaunchKernelGGL(ms_thr_small_kernel, dim3(1), dim3(1024), 0, stream, score, (int)total, (int)M, threshold, largest,
                       vec, out_row, out_col, out_score, num_out, capacity);
    return launch_status(what);
  }
  ThrLayout tl;
  if (thr_layout(tiles, tl) != 0) {
    set_last_error("%s: rocPRIM size query failed", what);
    return (int)hipErrorUnknown;
  }
  size_t scan_bytes = tl.scan_bytes;
  const size_t o_slots = tl.o_slots, o_tmp = tl.o_tmp;
  Scratch scratch;
  if (int rc0 = scratch.acquire(what, workspace, workspace_bytes, tl.total, stream)) return rc0;
  char* base = scratch.base;
  hipError_t e;
  int* counts = reinterpret_cast<int*>(base);
  int* slots = reinterpret_cast<int*>(base + o_slots);
  hipLaunchKernelGGL(ms_thr_count_kernel, dim3((unsigned)tiles), dim3(256), 0, stream, score, total, threshold, largest, vec,
                     counts);
  int rc = 0;
  e = rocprim::exclusive_scan(base + o_tmp, scan_bytes, counts, slots, 0, (size_t)tiles, rocprim::plus<int>(), stream);
  if (e != hipSuccess) {
    set_last_error("%s: scan: %s", what, hipGetErrorString(e));
    rc = (int)e;
  } else {
    hipLaunchKernelGGL(ms_thr_emit_kernel, dim3((unsigned)tiles), dim3(256), 0, stream, score, total, (int)M, threshold,
                       largest, vec, counts, slots, out_row, out_col, out_score, num_out, capacity);
    rc = launch_status(what);
  }
  return scratch.release(what, rc);
}

int v3d_global_topk_select_f32(const float* score, int64_t N, int64_t M, int64_t k, int flags, int64_t* out_row,
                               int64_t* out_col, float* out_score, int64_t* num_out, int64_t capacity, void* workspace,
                               size_t workspace_bytes, int device, void* stream_) {
  using namespace v3d;
  const char* what = "global_topk_select";
  V3D_REQUIRE(N >= 0 && M >= 0 && k >= 0 && capacity >= 0, what, "negative dimension");
  V3D_REQUIRE(M <= INT32_MAX && N <= INT32_MAX, what, "N and M must fit 31 bits");
  const int64_t total = N * M;
  V3D_REQUIRE(k <= total, what, "selected index k out of range");  // torch.topk's own error
  if (total > (int64_t)UINT32_MAX) {
    set_last_error("%s: N*M must stay below 2^32", what);
    return V3D_ERR_SIZE_LIMIT;
  }
  DeviceGuard guard(device);
  if (guard.rc) return guard.rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (k == 0) {
    if (num_out) V3D_HIP_TRY(hipMemsetAsync(num_out, 0, sizeof(int64_t), stream), what);
    return 0;
  }
  V3D_REQUIRE(score, what, "null pointer");
  const int largest = (flags & V3D_SEL_LARGEST) ? 1 : 0, dustbin = (flags & V3D_SEL_DUSTBIN) ? 1 : 0;
  const int vec = ms_vec_ok(score);
  GtkLayout gl;
  if (gtk_layout(total, k, gl) != 0) {
    set_last_error("%s: rocPRIM size query failed", what);
    return (int)hipErrorUnknown;
  }
  const bool small_sort = gl.small_sort;
  size_t sort_bytes = gl.sort_bytes, scan_bytes = gl.scan_bytes;
  const unsigned rank = gl.rank, list_cap = gl.list_cap;
  const size_t o_state = gl.o_state, o_hist = gl.o_hist, zero_bytes = gl.zero_bytes, o_cand = gl.o_cand, o_list = gl.o_list,
               o_sorted = gl.o_sorted, o_keep = gl.o_keep, o_slot = gl.o_slot, o_tmp = gl.o_tmp;
  Scratch scratch;
  if (int rc0 = scratch.acquire(what, workspace, workspace_bytes, gl.total, stream)) return rc0;
  char* base = scratch.base;
  int rc = 0;
  hipError_t e = hipMemsetAsync(base, 0, zero_bytes, stream);
  if (e != hipSuccess) {
    set_last_error("%s: hipMemsetAsync: %s", what, hipGetErrorString(e));
    rc = (int)e;
  }
  if (rc == 0) {
    MsState* st = reinterpret_cast<MsState*>(base + o_state);
    unsigned* hist = reinterpret_cast<unsigned*>(base + o_hist);
    unsigned long long* cand = reinterpret_cast<unsigned long long*>(base + o_cand);
    unsigned long long* list = list_cap ? reinterpret_cast<unsigned long long*>(base + o_list) : nullptr;
    const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(total, 512 * 16), 2048));
    if (list_cap) {
      hipLaunchKernelGGL(ms_sample_kernel, dim3(1), dim3(1024), 0, stream, score, total, largest, rank, st);
      hipLaunchKernelGGL(ms_list_kernel, dim3(blocks), dim3(512), 0, stream, score, total, largest, vec, st, list, list_cap);
    }
    const unsigned sel_blocks = blocks;
    for (int pass = 0; pass < 6; ++pass)
      hipLaunchKernelGGL(ms_hist_kernel, dim3(sel_blocks), dim3(512), 0, stream, score, total, largest, vec, pass,
                         (unsigned)k, st, hist, list, list_cap);
    hipLaunchKernelGGL(ms_take_kernel, dim3(sel_blocks), dim3(512), 0, stream, score, total, largest, vec, (unsigned)k, st,
                       cand, list, list_cap);
    if (small_sort) {
      hipLaunchKernelGGL(ms_sort_emit_small_kernel, dim3(1), dim3(1024), 0, stream, cand, (int)k, score, (int)N, (int)M,
                         dustbin, out_row, out_col, out_score, num_out, capacity);
      rc = launch_status(what);
    } else {
      unsigned long long* sorted = reinterpret_cast<unsigned long long*>(base + o_sorted);
      int* keep = reinterpret_cast<int*>(base + o_keep);
      int* slots = reinterpret_cast<int*>(base + o_slot);
      e = rocprim::radix_sort_keys(base + o_tmp, sort_bytes, cand, sorted, (size_t)k, 0, 64, stream);
      const unsigned kb = (unsigned)ceil_div(k, 256);
      if (e == hipSuccess) {
        hipLaunchKernelGGL(ms_keep_kernel, dim3(kb), dim3(256), 0, stream, sorted, k, (int)N, (int)M, dustbin, keep);
        e = rocprim::exclusive_scan(base + o_tmp, scan_bytes, keep, slots, 0, (size_t)k, rocprim::plus<int>(), stream);
      }
      if (e != hipSuccess) {
        set_last_error("%s: sort / scan: %s", what, hipGetErrorString(e));
        rc = (int)e;
      } else {
        hipLaunchKernelGGL(ms_emit_sorted_kernel, dim3(kb), dim3(256), 0, stream, sorted, keep, slots, k, score, (int)M,
                           out_row, out_col, out_score, num_out, capacity);
        rc = launch_status(what);
      }
    }
  }
  return scratch.release(what, rc);
}

}  // extern "C"
