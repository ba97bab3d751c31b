// select.hip -- correspondence selectors downstream of pairwise_distance.
// Replaces `mutual_topk_select` / `batch_mutual_topk_select` of the reference (vision3d/ops/mutual_topk_select.py:7-133):
// two torch.topk passes over the score matrix, two boolean (B,N,M) scatter matrices, logical_and/or, threshold and mask
// passes over the matrix again, torch.nonzero and a final gather -- the N x M matrix is read or written 8-10 times.
//
// Design (gfx950).  The matrix is read exactly twice (once per side), nothing of its size is written:
//   1. row side: one wave per row keeps the running top-k as a lane-distributed sorted list (one entry per lane,
//      whole-wave DPP shift insert), 64 columns per step, four steps of loads in flight;
//   2. column side: a workgroup owns 64 columns of one row range; 64 x 64 tiles are staged through LDS (coalesced row
//      reads, conflict-free transposed reads: pitch 65) and every wave runs the same list insert for 16 columns;
//      row ranges (so that small M still fills the chip) are merged by a third small kernel;
//   3. both lists become 64-bit keys (b, r, c); ONE device radix sort (rocPRIM primitive) puts them in
//      torch.nonzero order; a key present twice is a mutual pair, heads are the union; threshold and masks are
//      applied to the <= k(N+M) candidates only; flags -> exclusive scan -> compacted (batch, row, col, score).
// Scores are compared through a monotone integer image of the float (so -inf / +inf entries are ordinary values);
// among equal scores the lower index wins (torch.topk leaves this unspecified).
// The same back end serves lists that never came from a matrix: `v3d_mutual_select_lists_f32` takes the two
// feature-kNN results of (x, y) and (y, x), which is how 100k x 100k mutual matching runs without a 40 GB matrix.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include <algorithm>

#include "common.h"

namespace v3d {

constexpr int kSelMaxK = 64;
constexpr int kSelLargest = 1, kSelMutual = 2, kSelThreshold = 4;
constexpr int kSelEmptyKey = INT32_MAX;  // above the image of +inf: an unfilled list slot

__device__ __forceinline__ int sel_shr1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, false); }

// Monotone image of a float: a < b  <=>  sel_key(a) < sel_key(b); -0 and +0 share one key.
__device__ __forceinline__ int sel_key(float f, bool largest) {
  f = (largest ? -f : f) + 0.0f;
  const int b = __float_as_int(f);
  return b ^ ((b >> 31) & 0x7fffffff);
}
__device__ __forceinline__ float sel_unkey(int key, bool largest) {
  const float f = __int_as_float(key ^ ((key >> 31) & 0x7fffffff));
  return largest ? -f : f;
}

// Lane j of the wave holds the j-th best (key, index) so far.  Offers one candidate per lane (lanes in ascending index
// order); a candidate enters behind every entry with key <= its own, so equal keys stay in index order.
__device__ __forceinline__ void sel_push(int& lk, int& li, const int key, const int idx, const int k, const int lane) {
  int tau = __builtin_amdgcn_readlane(lk, k - 1);
  unsigned long long m = __ballot(key < tau);
  while (m) {
    const int l = __builtin_ffsll((long long)m) - 1;
    const int ck = __builtin_amdgcn_readlane(key, l);
    const int ci = __builtin_amdgcn_readlane(idx, l);
    const int pos = __builtin_popcountll(__ballot(lk <= ck));
    const int sk = sel_shr1(lk), si = sel_shr1(li);
    if (lane > pos) {
      lk = sk;
      li = si;
    } else if (lane == pos) {
      lk = ck;
      li = ci;
    }
    tau = __builtin_amdgcn_readlane(lk, k - 1);
    m = __ballot(key < tau) & ~((2ull << l) - 1ull);
  }
}

// Same list, candidates in ANY index order: position and admission compare (key, index) lexicographically.
__device__ __forceinline__ void sel_push_any(int& lk, int& li, const int key, const int idx, const int k, const int lane) {
  int tau = __builtin_amdgcn_readlane(lk, k - 1), taui = __builtin_amdgcn_readlane(li, k - 1);
  if (taui < 0) taui = INT32_MAX;  // an unfilled slot loses against any index
  unsigned long long m = __ballot(key < tau || (key == tau && idx < taui));
  while (m) {
    const int l = __builtin_ffsll((long long)m) - 1;
    const int ck = __builtin_amdgcn_readlane(key, l);
    const int ci = __builtin_amdgcn_readlane(idx, l);
    const int pos = __builtin_popcountll(__ballot(li >= 0 && (lk < ck || (lk == ck && li < ci))));
    const int sk = sel_shr1(lk), si = sel_shr1(li);
    if (lane > pos) {
      lk = sk;
      li = si;
    } else if (lane == pos) {
      lk = ck;
      li = ci;
    }
    tau = __builtin_amdgcn_readlane(lk, k - 1);
    taui = __builtin_amdgcn_readlane(li, k - 1);
    if (taui < 0) taui = INT32_MAX;
    m = __ballot(key < tau || (key == tau && idx < taui)) & ~((2ull << l) - 1ull);
  }
}

// ---- row side: (rows, M) -> lists (rows, k) --------------------------------------------------------------------
__global__ __launch_bounds__(256) void sel_rows_kernel(const float* __restrict__ score, int64_t rows, int M, int k,
                                                       int largest, int vec4, int* __restrict__ out_i,
                                                       float* __restrict__ out_v) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* srow = score + row * M;
  int lk = kSelEmptyKey, li = -1;
  if (vec4) {
    // 16 bytes per lane, two loads in flight: 2 KiB of the row per step.  Lane l holds columns c0 + 4l .. 4l+3, so the
    // four pushes of a load are not in index order: the order-free push keeps ties by index.
    for (int c0 = 0; c0 < M; c0 += 512) {
      float4 v[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int c = c0 + u * 256 + lane * 4;
        v[u] = c < M ? *reinterpret_cast<const float4*>(srow + c) : make_float4(0.f, 0.f, 0.f, 0.f);  // M % 4 == 0
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int c = c0 + u * 256 + lane * 4;
        const float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
        for (int q = 0; q < 4; ++q)
          sel_push_any(lk, li, c < M ? sel_key(e[q], largest) : kSelEmptyKey, c + q, k, lane);
      }
    }
  } else {
    for (int c0 = 0; c0 < M; c0 += 256) {
      int key[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + u * 64 + lane;
        key[u] = c < M ? sel_key(srow[c], largest) : kSelEmptyKey;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) sel_push(lk, li, key[u], c0 + u * 64 + lane, k, lane);
    }
  }
  if (lane < k) {
    out_i[row * k + lane] = li;
    out_v[row * k + lane] = li >= 0 ? sel_unkey(lk, largest) : 0.f;
  }
}

// Column side for small matrices (L2-resident): one wave per column, lane l reads row r0 + l of that column (stride M);
// the tile kernel below needs M / 64 x row-range workgroups and is launch-latency bound under a few hundred columns.
__global__ __launch_bounds__(256) void sel_cols_strided_kernel(const float* __restrict__ score, int B, int N, int M, int k,
                                                               int largest, int* __restrict__ out_i,
                                                               float* __restrict__ out_v) {
  const int lane = threadIdx.x & 63;
  const int64_t colid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (colid >= (int64_t)B * M) return;
  const float* base = score + (colid / M) * (int64_t)N * M + (colid % M);
  int lk = kSelEmptyKey, li = -1;
  for (int r0 = 0; r0 < N; r0 += 256) {
    int key[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = r0 + u * 64 + lane;
      key[u] = r < N ? sel_key(base[(int64_t)r * M], largest) : kSelEmptyKey;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) sel_push(lk, li, key[u], r0 + u * 64 + lane, k, lane);
  }
  if (lane < k) {
    out_i[colid * k + lane] = li;
    out_v[colid * k + lane] = li >= 0 ? sel_unkey(lk, largest) : 0.f;
  }
}

// ---- column side: workgroup = 64 columns x one row range of one matrix -----------------------------------------
constexpr int kColTile = 64;
__global__ __launch_bounds__(256) void sel_cols_kernel(const float* __restrict__ score, int B, int N, int M, int k,
                                                       int largest, int rows_per_split, int vec4,
                                                       int* __restrict__ part_k, int* __restrict__ part_i,
                                                       int* __restrict__ fin_i, float* __restrict__ fin_v) {
  __shared__ int tile[2][kColTile][kColTile + 1];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c0 = blockIdx.x * kColTile, s = blockIdx.y, b = blockIdx.z;
  const int r_begin = s * rows_per_split, r_end = min(N, r_begin + rows_per_split);
  const float* mat = score + (int64_t)b * N * M;
  int lk[16], li[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    lk[j] = kSelEmptyKey;
    li[j] = -1;
  }
  const int col = c0 + lane;
  int buf = 0;
  for (int rt = r_begin; rt < r_end; rt += kColTile, buf ^= 1) {
    if (vec4) {  // 16 bytes per lane: one instruction brings 4 rows x 256 B (M % 4 == 0, 16-byte aligned rows)
      float4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = rt + w * 16 + i * 4 + (lane >> 4), c = c0 + (lane & 15) * 4;
        v[i] = (r < r_end && c < M) ? *reinterpret_cast<const float4*>(mat + (int64_t)r * M + c)
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rl = w * 16 + i * 4 + (lane >> 4), cl = (lane & 15) * 4;
        const bool in = rt + rl < r_end && c0 + cl < M;
        tile[buf][rl][cl + 0] = in ? sel_key(v[i].x, largest) : kSelEmptyKey;
        tile[buf][rl][cl + 1] = in ? sel_key(v[i].y, largest) : kSelEmptyKey;
        tile[buf][rl][cl + 2] = in ? sel_key(v[i].z, largest) : kSelEmptyKey;
        tile[buf][rl][cl + 3] = in ? sel_key(v[i].w, largest) : kSelEmptyKey;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int r = rt + w * 16 + i;
        const bool in = r < r_end && col < M;
        tile[buf][w * 16 + i][lane] = in ? sel_key(mat[(int64_t)r * M + col], largest) : kSelEmptyKey;
      }
    }
    __syncthreads();  // the other buffer is free again once every wave has passed this barrier of the NEXT tile
#pragma unroll
    for (int j = 0; j < 16; ++j) sel_push(lk[j], li[j], tile[buf][lane][w * 16 + j], rt + lane, k, lane);
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int c = c0 + w * 16 + j;
    if (c < M && lane < k) {
      const int64_t o = (((int64_t)s * B + b) * M + c) * k + lane;
      if (fin_i) {  // a single row range: these ARE the column lists
        fin_i[o] = li[j];
        fin_v[o] = li[j] >= 0 ? sel_unkey(lk[j], largest) : 0.f;
      } else {
        part_k[o] = lk[j];
        part_i[o] = li[j];
      }
    }
  }
}

// Merge the row-range lists of one column (ranges ascend, each list is in (key, index) order) and write the final list.
__global__ __launch_bounds__(256) void sel_cols_merge_kernel(const int* __restrict__ part_k,
                                                             const int* __restrict__ part_i, int64_t cols, int S, int k,
                                                             int largest, int* __restrict__ out_i,
                                                             float* __restrict__ out_v) {
  const int lane = threadIdx.x & 63;
  const int64_t col = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (col >= cols) return;
  int lk = kSelEmptyKey, li = -1;
  for (int s = 0; s < S; ++s) {
    const int64_t o = ((int64_t)s * cols + col) * k + lane;
    const int key = lane < k ? part_k[o] : kSelEmptyKey;
    const int idx = lane < k ? part_i[o] : -1;
    if (s == 0) {
      lk = key;
      li = idx;
    } else {
      sel_push(lk, li, key, idx, k, lane);
    }
  }
  if (lane < k) {
    out_i[col * k + lane] = li;
    out_v[col * k + lane] = li >= 0 ? sel_unkey(lk, largest) : 0.f;
  }
}

// ---- lists -> keys ---------------------------------------------------------------------------------------------
// entry e < nR is slot (b, r, j) of the row lists and names column row_i[e]; the rest are slots (b, c, j) of the column
// lists and name row col_i[e - nR].  key = (b*N + r)*M + c, or N*M*B for an unfilled slot.
template <typename IndexT>
__global__ __launch_bounds__(256) void sel_keys_kernel(const IndexT* __restrict__ row_i, const IndexT* __restrict__ col_i,
                                                       int B, int N, int M, int k, unsigned long long* __restrict__ keys,
                                                       int* __restrict__ vals) {
  const int64_t nR = (int64_t)B * N * k, T = nR + (int64_t)B * M * k;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= T) return;
  int64_t b, r, c;
  if (e < nR) {
    const int64_t slot = e / k;
    b = slot / N;
    r = slot % N;
    c = (int64_t)row_i[e];
  } else {
    const int64_t slot = (e - nR) / k;
    b = slot / M;
    c = slot % M;
    r = (int64_t)col_i[e - nR];
  }
  const bool valid = r >= 0 && r < N && c >= 0 && c < M;
  keys[e] = valid ? (unsigned long long)((b * N + r) * M + c) : (unsigned long long)B * N * M;
  vals[e] = (int)e;
}

__global__ __launch_bounds__(256) void sel_flags_kernel(const unsigned long long* __restrict__ keys,
                                                        const int* __restrict__ vals, const float* __restrict__ row_v,
                                                        const float* __restrict__ col_v,
                                                        const unsigned char* __restrict__ row_masks,
                                                        const unsigned char* __restrict__ col_masks, int B, int N, int M,
                                                        int k, int flags, float threshold, int* __restrict__ keep) {
  const int64_t nR = (int64_t)B * N * k, T = nR + (int64_t)B * M * k;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= T) return;
  const unsigned long long key = keys[i], none = (unsigned long long)B * N * M;
  bool on = key < none;
  if (on) on = (flags & kSelMutual) ? (i + 1 < T && keys[i + 1] == key) : (i == 0 || keys[i - 1] != key);
  if (on && (flags & kSelThreshold)) {
    const int e = vals[i];
    const float v = e < nR ? row_v[e] : col_v[e - nR];
    on = (flags & kSelLargest) ? v > threshold : v < threshold;  // mutual_topk_select.py:48-53
  }
  if (on && (row_masks || col_masks)) {
    const int64_t c = (int64_t)(key % (unsigned long long)M), br = (int64_t)(key / (unsigned long long)M);
    if (row_masks) on = row_masks[br] != 0;                                  // :123-124
    if (on && col_masks) on = col_masks[(br / N) * (int64_t)M + c] != 0;     // :125-126
  }
  keep[i] = on ? 1 : 0;
}

__global__ __launch_bounds__(256) void sel_emit_kernel(const unsigned long long* __restrict__ keys,
                                                       const int* __restrict__ vals, const int* __restrict__ keep,
                                                       const int* __restrict__ slots, const float* __restrict__ row_v,
                                                       const float* __restrict__ col_v, int B, int N, int M, int k,
                                                       unsigned char* __restrict__ corr_mat,
                                                       int64_t* __restrict__ out_b, int64_t* __restrict__ out_r,
                                                       int64_t* __restrict__ out_c, float* __restrict__ out_s,
                                                       int64_t* __restrict__ num_out, int64_t capacity) {
  const int64_t nR = (int64_t)B * N * k, T = nR + (int64_t)B * M * k;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= T) return;
  if (i == T - 1 && num_out) *num_out = (int64_t)slots[i] + keep[i];
  if (!keep[i]) return;
  const unsigned long long key = keys[i];
  if (corr_mat) corr_mat[key] = 1;
  const int64_t at = slots[i];
  if (at >= capacity) return;
  const int64_t c = (int64_t)(key % (unsigned long long)M), br = (int64_t)(key / (unsigned long long)M);
  if (out_b) out_b[at] = br / N;
  if (out_r) out_r[at] = br % N;
  if (out_c) out_c[at] = c;
  if (out_s) {
    const int e = vals[i];
    out_s[at] = e < nR ? row_v[e] : col_v[e - nR];
  }
}

// ---- small problems (<= 4096 candidates: superpoint matching, a few hundred rows and columns) -------------------------
// ONE workgroup does what the five launches below do: keys into LDS, bitonic sort of (key, entry) pairs, duplicate /
// head flags with threshold and masks, block-wide exclusive scan, emit.  The launch count, not the work, bounds these
// sizes (512 x 512, k = 3: 140 us through the general path, of which ~100 us are launch gaps).
constexpr int kSelSmallMax = 4096;
template <typename IndexT>
__global__ __launch_bounds__(1024) void sel_small_kernel(const IndexT* __restrict__ row_i, const float* __restrict__ row_v,
                                                         const IndexT* __restrict__ col_i, const float* __restrict__ col_v,
                                                         const unsigned char* __restrict__ row_masks,
                                                         const unsigned char* __restrict__ col_masks, int B, int N, int M,
                                                         int k, int flags, float threshold,
                                                         unsigned char* __restrict__ corr_mat, int64_t* __restrict__ out_b,
                                                         int64_t* __restrict__ out_r, int64_t* __restrict__ out_c,
                                                         float* __restrict__ out_s, int64_t* __restrict__ num_out,
                                                         int64_t capacity) {
  __shared__ unsigned long long skey[kSelSmallMax];
  __shared__ int sval[kSelSmallMax];
  __shared__ int wave_tot[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nR = B * N * k, T = nR + B * M * k;
  const unsigned long long none = (unsigned long long)B * N * M;
  int P = 64;
  while (P < T) P <<= 1;
  for (int e = tid; e < P; e += 1024) {
    unsigned long long key = ~0ull;
    if (e < T) {
      long long b, r, c;
      if (e < nR) {
        const int slot = e / k;
        b = slot / N;
        r = slot % N;
        c = (long long)row_i[e];
      } else {
        const int slot = (e - nR) / k;
        b = slot / M;
        c = slot % M;
        r = (long long)col_i[e - nR];
      }
      const bool valid = r >= 0 && r < N && c >= 0 && c < M;
      key = valid ? (unsigned long long)((b * N + r) * M + c) : none;
    }
    skey[e] = key;
    sval[e] = e;
  }
  __syncthreads();
  for (int size = 2; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (P >> 1); t += 1024) {
        const int lo = ((t & ~(stride - 1)) << 1) | (t & (stride - 1)), hi = lo | stride;
        const bool up = (lo & size) == 0;
        const unsigned long long a = skey[lo], bb = skey[hi];
        if ((a > bb) == up) {
          skey[lo] = bb;
          skey[hi] = a;
          const int va = sval[lo];
          sval[lo] = sval[hi];
          sval[hi] = va;
        }
      }
      __syncthreads();
    }
  }
  // flags (4 consecutive candidates per thread), then an exclusive scan over the workgroup
  int keep[4], sum = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = tid * 4 + u;
    bool on = false;
    if (i < T) {
      const unsigned long long key = skey[i];
      on = key < none;
      if (on) on = (flags & kSelMutual) ? (i + 1 < T && skey[i + 1] == key) : (i == 0 || skey[i - 1] != key);
      if (on && (flags & kSelThreshold)) {
        const int e = sval[i];
        const float v = e < nR ? row_v[e] : col_v[e - nR];
        on = (flags & kSelLargest) ? v > threshold : v < threshold;
      }
      if (on && (row_masks || col_masks)) {
        const long long c = (long long)(key % (unsigned long long)M), br = (long long)(key / (unsigned long long)M);
        if (row_masks) on = row_masks[br] != 0;
        if (on && col_masks) on = col_masks[(br / N) * (long long)M + c] != 0;
      }
    }
    keep[u] = on ? 1 : 0;
    sum += keep[u];
  }
  int incl = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  int run = incl - sum, total = 0;
  for (int w = 0; w < 16; ++w) {
    if (w < wave) run += wave_tot[w];
    total += wave_tot[w];
  }
  if (tid == 0 && num_out) *num_out = total;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (keep[u]) {
      const int i = tid * 4 + u;
      const unsigned long long key = skey[i];
      if (corr_mat) corr_mat[key] = 1;
      if (run < capacity) {
        const long long c = (long long)(key % (unsigned long long)M), br = (long long)(key / (unsigned long long)M);
        if (out_b) out_b[run] = br / N;
        if (out_r) out_r[run] = br % N;
        if (out_c) out_c[run] = c;
        if (out_s) {
          const int e = sval[i];
          out_s[run] = e < nR ? row_v[e] : col_v[e - nR];
        }
      }
      ++run;
    }
  }
}

// Mutual selection of small problems needs no sort at all: a row's candidates are its own k list entries, a candidate
// (r, c) is kept iff r is in column c's list, and nonzero order is "rows ascending, columns ascending inside a row".
// One workgroup: thread t owns rows 4t .. 4t+3 (keep masks of k <= 64 bits), block scan of the row counts, emit.
template <typename IndexT>
__global__ __launch_bounds__(1024) void sel_mutual_small_kernel(const IndexT* __restrict__ row_i,
                                                                const float* __restrict__ row_v,
                                                                const IndexT* __restrict__ col_i,
                                                                const unsigned char* __restrict__ row_masks,
                                                                const unsigned char* __restrict__ col_masks, int B, int N,
                                                                int M, int k, int flags, float threshold,
                                                                unsigned char* __restrict__ corr_mat,
                                                                int64_t* __restrict__ out_b, int64_t* __restrict__ out_r,
                                                                int64_t* __restrict__ out_c, float* __restrict__ out_s,
                                                                int64_t* __restrict__ num_out, int64_t capacity) {
  __shared__ int wave_tot[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rows = B * N;
  unsigned long long keep[4];
  int sum = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int row = tid * 4 + u;
    keep[u] = 0ull;
    if (row < rows) {
      const int b = row / N, r = row - b * N;
      const bool row_on = row_masks == nullptr || row_masks[row] != 0;
      for (int j = 0; j < k && row_on; ++j) {
        const long long c = (long long)row_i[(long long)row * k + j];
        if (c < 0 || c >= M) continue;
        const IndexT* cl = col_i + ((long long)b * M + c) * k;
        bool on = false;
        for (int t = 0; t < k; ++t) on = on || ((long long)cl[t] == r);
        if (on && (flags & kSelThreshold)) {
          const float v = row_v[(long long)row * k + j];
          on = (flags & kSelLargest) ? v > threshold : v < threshold;
        }
        if (on && col_masks) on = col_masks[(long long)b * M + c] != 0;
        if (on) keep[u] |= 1ull << j;
      }
      sum += __builtin_popcountll(keep[u]);
    }
  }
  int incl = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  int run = incl - sum, total = 0;
  for (int w = 0; w < 16; ++w) {
    if (w < wave) run += wave_tot[w];
    total += wave_tot[w];
  }
  if (tid == 0 && num_out) *num_out = total;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int row = tid * 4 + u;
    const int b = row / N, r = row - b * N;
    for (unsigned long long m = keep[u]; m; m &= m - 1) {
      const int j = __builtin_ctzll(m);
      const long long c = (long long)row_i[(long long)row * k + j];
      int rank = 0;  // kept entries of this row with a smaller column come first
      for (unsigned long long o = keep[u]; o; o &= o - 1)
        rank += (long long)row_i[(long long)row * k + __builtin_ctzll(o)] < c ? 1 : 0;
      const long long at = run + rank;
      if (corr_mat) corr_mat[((long long)row) * M + c] = 1;
      if (at < capacity) {
        if (out_b) out_b[at] = b;
        if (out_r) out_r[at] = r;
        if (out_c) out_c[at] = c;
        if (out_s) out_s[at] = row_v[(long long)row * k + j];
      }
    }
    run += __builtin_popcountll(keep[u]);
  }
}

struct SelectOut {
  unsigned char* corr_mat;
  int64_t *out_b, *out_r, *out_c;
  float* out_s;
  int64_t* num_out;
  int64_t capacity;
};

static inline unsigned bits_for(unsigned long long v) {  // bits needed to hold every value in [0, v]
  unsigned n = 1;
  while ((v >> n) != 0 && n < 64) ++n;
  return n;
}

struct SelectSizes {
  size_t T, sort_bytes, scan_bytes;
  size_t o_k0, o_k1, o_v0, o_v1, o_keep, o_slot, o_tmp, total;
};

static int select_sizes(int B, int N, int M, int k, size_t base_off, SelectSizes& z, hipStream_t stream) {
  z.T = (size_t)B * (size_t)k * ((size_t)N + (size_t)M);
  unsigned long long* nk = nullptr;
  int* ni = nullptr;
  z.sort_bytes = z.scan_bytes = 0;
  const unsigned end_bit = bits_for((unsigned long long)B * N * M);
  hipError_t e = rocprim::radix_sort_pairs(nullptr, z.sort_bytes, nk, nk, ni, ni, z.T, 0, end_bit, stream);
  if (e == hipSuccess) e = rocprim::exclusive_scan(nullptr, z.scan_bytes, ni, ni, 0, z.T, rocprim::plus<int>(), stream);
  if (e != hipSuccess) return (int)e;
  size_t off = base_off;
  auto take = [&](size_t bytes) {
    const size_t at = off;
    off += align_up(bytes, 256);
    return at;
  };
  z.o_k0 = take(8 * z.T);
  z.o_k1 = take(8 * z.T);
  z.o_v0 = take(4 * z.T);
  z.o_v1 = take(4 * z.T);
  z.o_keep = take(4 * z.T);
  z.o_slot = take(4 * z.T);
  z.o_tmp = take(std::max(z.sort_bytes, z.scan_bytes));
  z.total = off;
  return 0;
}

// keys -> sort -> flags -> scan -> emit, all on `stream`, scratch carved from `base` per `z`.
template <typename IndexT>
static int select_from_lists(const IndexT* row_i, const float* row_v, const IndexT* col_i, const float* col_v,
                             const unsigned char* row_masks, const unsigned char* col_masks, int B, int N, int M, int k,
                             int flags, float threshold, const SelectOut& o, char* base, const SelectSizes& z,
                             hipStream_t stream, const char* what) {
  unsigned long long* k0 = reinterpret_cast<unsigned long long*>(base + z.o_k0);
  unsigned long long* k1 = reinterpret_cast<unsigned long long*>(base + z.o_k1);
  int* v0 = reinterpret_cast<int*>(base + z.o_v0);
  int* v1 = reinterpret_cast<int*>(base + z.o_v1);
  int* keep = reinterpret_cast<int*>(base + z.o_keep);
  int* slots = reinterpret_cast<int*>(base + z.o_slot);
  void* tmp = base + z.o_tmp;
  const unsigned blocks = (unsigned)ceil_div((int64_t)z.T, 256);
  const unsigned end_bit = bits_for((unsigned long long)B * N * M);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(sel_keys_kernel<IndexT>), dim3(blocks), dim3(256), 0, stream, row_i, col_i, B, N, M,
                     k, k0, v0);
  size_t sort_bytes = z.sort_bytes, scan_bytes = z.scan_bytes;
  hipError_t e = rocprim::radix_sort_pairs(tmp, sort_bytes, k0, k1, v0, v1, z.T, 0, end_bit, stream);
  if (e != hipSuccess) {
    set_last_error("%s: radix sort: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  hipLaunchKernelGGL(sel_flags_kernel, dim3(blocks), dim3(256), 0, stream, k1, v1, row_v, col_v, row_masks, col_masks, B,
                     N, M, k, flags, threshold, keep);
  e = rocprim::exclusive_scan(tmp, scan_bytes, keep, slots, 0, z.T, rocprim::plus<int>(), stream);
  if (e != hipSuccess) {
    set_last_error("%s: scan: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  hipLaunchKernelGGL(sel_emit_kernel, dim3(blocks), dim3(256), 0, stream, k1, v1, keep, slots, row_v, col_v, B, N, M, k,
                     o.corr_mat, o.out_b, o.out_r, o.out_c, o.out_s, o.num_out, o.capacity);
  return launch_status(what);
}

template <typename IndexT>
static int select_small(const IndexT* row_i, const float* row_v, const IndexT* col_i, const float* col_v,
                        const unsigned char* row_masks, const unsigned char* col_masks, int B, int N, int M, int k,
                        int flags, float threshold, const SelectOut& o, hipStream_t stream, const char* what) {
  const bool needs_values = o.out_s != nullptr || (flags & kSelThreshold);
  if ((flags & kSelMutual) && (int64_t)B * N <= kSelSmallMax && (row_v != nullptr || !needs_values))
    hipLaunchKernelGGL(HIP_KERNEL_NAME(sel_mutual_small_kernel<IndexT>), dim3(1), dim3(1024), 0, stream, row_i, row_v,
                       col_i, row_masks, col_masks, B, N, M, k, flags, threshold, o.corr_mat, o.out_b, o.out_r, o.out_c,
                       o.out_s, o.num_out, o.capacity);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(sel_small_kernel<IndexT>), dim3(1), dim3(1024), 0, stream, row_i, row_v, col_i,
                       col_v, row_masks, col_masks, B, N, M, k, flags, threshold, o.corr_mat, o.out_b, o.out_r, o.out_c,
                       o.out_s, o.num_out, o.capacity);
  return launch_status(what);
}
static inline bool select_is_small(int B, int N, int M, int k) {
  return (int64_t)B * k * ((int64_t)N + M) <= kSelSmallMax && (int64_t)B * N * M < (1ll << 40);
}

static int select_prologue(const char* what, int B, int N, int M, int k, const SelectOut& o, hipStream_t stream,
                           bool& done) {
  done = false;
  if (o.num_out) V3D_HIP_TRY(hipMemsetAsync(o.num_out, 0, sizeof(int64_t), stream), what);
  const size_t cells = (size_t)B * (size_t)N * (size_t)M;
  if (o.corr_mat && cells) V3D_HIP_TRY(hipMemsetAsync(o.corr_mat, 0, cells, stream), what);
  if (cells == 0 || k == 0) done = true;
  return 0;
}

// scratch of v3d_mutual_topk_select_f32: row / column lists, partial column lists of the row splits, the back end's buffers
struct MutualLayout {
  int col_tiles, S, rows_per_split;
  size_t o_ri, o_rv, o_ci, o_cv, o_pk, o_pi;
  SelectSizes z;
};
static int mutual_layout(int B, int N, int M, int k, MutualLayout& m) {
  // row ranges for the column side: enough workgroups to fill 256 CUs, at least 256 rows each
  m.col_tiles = (int)ceil_div(M, kColTile);
  int S = (int)ceil_div(1024, std::max<int64_t>(1, (int64_t)B * m.col_tiles));
  S = std::max(1, std::min({S, (int)ceil_div(N, 256), 64}));
  m.rows_per_split = (int)(ceil_div(ceil_div(N, S), kColTile) * kColTile);
  m.S = (int)ceil_div(N, m.rows_per_split);
  const size_t nR = (size_t)B * N * k, nC = (size_t)B * M * k;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t at = off;
    off += align_up(bytes, 256);
    return at;
  };
  m.o_ri = take(4 * nR);
  m.o_rv = take(4 * nR);
  m.o_ci = take(4 * nC);
  m.o_cv = take(4 * nC);
  m.o_pk = take(4 * nC * m.S);
  m.o_pi = take(4 * nC * m.S);
  return select_sizes(B, N, M, k, off, m.z, nullptr);
}

}  // namespace v3d

extern "C" {

size_t v3d_mutual_topk_select_workspace_bytes(int B, int N, int M, int k) {
  if (B <= 0 || N <= 0 || M <= 0 || k <= 0) return 0;
  v3d::MutualLayout m;
  return v3d::mutual_layout(B, N, M, k, m) == 0 ? m.z.total : 0;
}
size_t v3d_mutual_select_lists_workspace_bytes(int B, int N, int M, int k) {
  if (B <= 0 || N <= 0 || M <= 0 || k <= 0 || v3d::select_is_small(B, N, M, k)) return 0;
  v3d::SelectSizes z;
  return v3d::select_sizes(B, N, M, k, 0, z, nullptr) == 0 ? z.total : 0;
}

int v3d_mutual_topk_select_f32(const float* score, const uint8_t* row_masks, const uint8_t* col_masks, int B, int N,
                               int M, int k, int flags, float threshold, uint8_t* corr_mat, int64_t* out_batch,
                               int64_t* out_row, int64_t* out_col, float* out_score, int64_t* num_out, int64_t capacity,
                               void* workspace, size_t workspace_bytes, int device, void* stream_) {
  using namespace v3d;
  const char* what = "mutual_topk_select";
  V3D_REQUIRE(B >= 0 && N >= 0 && M >= 0 && k >= 0 && capacity >= 0, what, "negative dimension");
  V3D_REQUIRE(k <= kSelMaxK, what, "k must be <= 64");
  V3D_REQUIRE(k <= N && k <= M, what, "selected index k out of range");  // torch.topk's own error
  V3D_REQUIRE(corr_mat || num_out, what, "one of corr_mat / num_out is required");
  if ((unsigned long long)B * (unsigned long long)k * ((unsigned long long)N + M) >= (unsigned long long)INT32_MAX ||
      B > kMaxGridYZ) {
    set_last_error("%s: B*k*(N+M) must stay below 2^31 and B <= 65535", what);
    return V3D_ERR_SIZE_LIMIT;
  }
  DeviceGuard guard(device);
  if (guard.rc) return guard.rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const SelectOut o{corr_mat, out_batch, out_row, out_col, out_score, num_out, capacity};
  bool done = false;
  int rc = select_prologue(what, B, N, M, k, o, stream, done);
  if (rc || done) return rc;
  V3D_REQUIRE(score, what, "null pointer");

  MutualLayout ml;
  rc = mutual_layout(B, N, M, k, ml);
  if (rc) {
    set_last_error("%s: rocPRIM size query failed", what);
    return rc;
  }
  const int col_tiles = ml.col_tiles, S = ml.S, rows_per_split = ml.rows_per_split;
  const size_t o_ri = ml.o_ri, o_rv = ml.o_rv, o_ci = ml.o_ci, o_cv = ml.o_cv, o_pk = ml.o_pk, o_pi = ml.o_pi;
  const SelectSizes& z = ml.z;
  Scratch scratch;
  if ((rc = scratch.acquire(what, workspace, workspace_bytes, z.total, stream)) != 0) return rc;
  char* base = scratch.base;
  int* row_i = reinterpret_cast<int*>(base + o_ri);
  float* row_v = reinterpret_cast<float*>(base + o_rv);
  int* col_i = reinterpret_cast<int*>(base + o_ci);
  float* col_v = reinterpret_cast<float*>(base + o_cv);
  int* part_k = reinterpret_cast<int*>(base + o_pk);
  int* part_i = reinterpret_cast<int*>(base + o_pi);
  const int largest = flags & kSelLargest;
  const int64_t rows = (int64_t)B * N, cols = (int64_t)B * M;
  const int vec4 = (M % 4 == 0 && ((uintptr_t)score & 15) == 0 && M >= 1024) ? 1 : 0;
  hipLaunchKernelGGL(sel_rows_kernel, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, stream, score, rows, M, k, largest,
                     vec4, row_i, row_v);
  if ((int64_t)B * N * M <= (4ll << 20) && M <= 4096) {  // small and L2-resident: one wave per column
    hipLaunchKernelGGL(sel_cols_strided_kernel, dim3((unsigned)ceil_div(cols, 4)), dim3(256), 0, stream, score, B, N, M, k,
                       largest, col_i, col_v);
  } else {
    const int cvec4 = (M % 4 == 0 && ((uintptr_t)score & 15) == 0) ? 1 : 0;
    hipLaunchKernelGGL(sel_cols_kernel, dim3(col_tiles, S, B), dim3(256), 0, stream, score, B, N, M, k, largest,
                       rows_per_split, cvec4, part_k, part_i, S == 1 ? col_i : nullptr, S == 1 ? col_v : nullptr);
    if (S > 1)
      hipLaunchKernelGGL(sel_cols_merge_kernel, dim3((unsigned)ceil_div(cols, 4)), dim3(256), 0, stream, part_k, part_i,
                         cols, S, k, largest, col_i, col_v);
  }
  rc = launch_status(what);
  if (rc == 0) {
    if (select_is_small(B, N, M, k))
      rc = select_small<int>(row_i, row_v, col_i, col_v, row_masks, col_masks, B, N, M, k, flags, threshold, o, stream,
                             what);
    else
      rc = select_from_lists<int>(row_i, row_v, col_i, col_v, row_masks, col_masks, B, N, M, k, flags, threshold, o, base,
                                  z, stream, what);
  }
  return scratch.release(what, rc);
}

int v3d_mutual_select_lists_f32(const int64_t* row_idx, const float* row_val, const int64_t* col_idx,
                                const float* col_val, const uint8_t* row_masks, const uint8_t* col_masks, int B, int N,
                                int M, int k, int flags, float threshold, uint8_t* corr_mat, int64_t* out_batch,
                                int64_t* out_row, int64_t* out_col, float* out_score, int64_t* num_out, int64_t capacity,
                                void* workspace, size_t workspace_bytes, int device, void* stream_) {
  using namespace v3d;
  const char* what = "mutual_select_lists";
  V3D_REQUIRE(B >= 0 && N >= 0 && M >= 0 && k >= 0 && capacity >= 0, what, "negative dimension");
  V3D_REQUIRE(corr_mat || num_out, what, "one of corr_mat / num_out is required");
  if ((unsigned long long)B * (unsigned long long)k * ((unsigned long long)N + M) >= (unsigned long long)INT32_MAX) {
    set_last_error("%s: B*k*(N+M) must stay below 2^31", what);
    return V3D_ERR_SIZE_LIMIT;
  }
  DeviceGuard guard(device);
  if (guard.rc) return guard.rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const SelectOut o{corr_mat, out_batch, out_row, out_col, out_score, num_out, capacity};
  bool done = false;
  int rc = select_prologue(what, B, N, M, k, o, stream, done);
  if (rc || done) return rc;
  V3D_REQUIRE(row_idx && col_idx, what, "null pointer");
  V3D_REQUIRE((row_val && col_val) || (!out_score && !(flags & kSelThreshold)), what,
              "list values are required for scores / threshold");
  if (select_is_small(B, N, M, k))
    return select_small<int64_t>(row_idx, row_val, col_idx, col_val, row_masks, col_masks, B, N, M, k, flags, threshold, o,
                                 stream, what);
  SelectSizes z;
  rc = select_sizes(B, N, M, k, 0, z, stream);
  if (rc) {
    set_last_error("%s: rocPRIM size query failed", what);
    return rc;
  }
  Scratch scratch;
  if ((rc = scratch.acquire(what, workspace, workspace_bytes, z.total, stream)) != 0) return rc;
  rc = select_from_lists<int64_t>(row_idx, row_val, col_idx, col_val, row_masks, col_masks, B, N, M, k, flags, threshold,
                                  o, scratch.base, z, stream, what);
  return scratch.release(what, rc);
}

}  // extern "C"
