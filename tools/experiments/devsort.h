// devsort.h -- the library's own device-wide stable radix sort and exclusive scan (gfx950), used by the pack-mode rows
// (grid subsampling / voxelisation, CSR builds, selectors, point-to-node partition).  Replaces the rocPRIM primitives those
// rows called in r01/r02: same calling convention (caller-provided temporary storage, asynchronous on a stream), different
// design goals -- few launches, no decoupled look-back (no workgroup ever waits for another: nothing here can hang), and
// DIGIT SKIPPING: the keys of these rows are packed fields whose upper bits are mostly constant (a KPConv voxel key uses
// ~25 of 64 bits), but how many is only known on the device.  One up-front pass histograms every 8-bit digit of every key;
// a pass whose histogram has one full bin moves nothing and is skipped by every workgroup on its own (its launches exit
// after reading 1 KiB), and the ping-pong between the output and the temporary buffer is planned from the same
// histograms so that the last active pass lands in the output.
//
//   sort:  memset(hist) -> hist_all -> per 8-bit pass { count (per-workgroup histogram), scatter } -> fix-up (only copies
//          when no pass was active).  LSD, stable: a workgroup's elements keep their order inside a digit -- every wave
//          owns a contiguous run of the tile and ranks 64 elements per step with a ballot match (8 ballots), running digit
//          counters in LDS; a workgroup's global offset per digit is the digit's base (scan of the global histogram) plus
//          the histograms of the workgroups before it, summed by the workgroup itself (at most kMaxBlocks of them).
//   scan:  per-workgroup sums -> every workgroup adds up the sums before it and scans its tile.  Two launches.
// Header-only (templates on the key type: uint32_t and uint64_t keys, optional int32 values).
#ifndef V3D_DEVSORT_H_
#define V3D_DEVSORT_H_

#include <algorithm>

#include "common.h"

namespace v3d {
namespace devsort {

constexpr int kThreads = 256;   // 4 waves
constexpr int kWaves = kThreads / 64;
constexpr int kPerWave = 1024;  // elements a wave ranks per sub-tile: 16 steps of 64
constexpr int kSteps = kPerWave / 64;
constexpr int kSubTile = kWaves * kPerWave;  // 4096
constexpr int kMaxBlocks = 512;  // a workgroup sums the histograms of the workgroups before it
constexpr int kMaxPasses = 8;

inline size_t tile_of(size_t n) {
  size_t tile = kSubTile;
  if (ceil_div((int64_t)n, (int64_t)tile) > kMaxBlocks) tile = align_up((size_t)ceil_div((int64_t)n, kMaxBlocks), kSubTile);
  return tile;
}
inline int blocks_of(size_t n) { return (int)std::max<int64_t>(1, ceil_div((int64_t)n, (int64_t)tile_of(n))); }

template <typename K>
struct Layout {
  size_t o_hist, o_blockhist, o_ktmp, o_vtmp, total;
};
template <typename K>
inline Layout<K> layout(size_t n, bool with_values) {
  Layout<K> l;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t at = off;
    off += align_up(bytes, 256);
    return at;
  };
  l.o_hist = take(sizeof(unsigned) * kMaxPasses * 256);
  l.o_blockhist = take(sizeof(unsigned) * 256 * (size_t)blocks_of(n));
  l.o_ktmp = take(sizeof(K) * n);
  l.o_vtmp = take(with_values ? sizeof(int) * n : 0);
  l.total = off;
  return l;
}
template <typename K>
inline size_t sort_bytes(size_t n, bool with_values) {
  return layout<K>(n, with_values).total;
}

template <typename K>
struct Bufs {
  const K* kin;
  K* kout;
  K* ktmp;
  const int* vin;
  int* vout;
  int* vtmp;
};

struct Pass {
  bool active;
  int src;  // 0 = input, 1 = output buffer, 2 = temporary
  int dst;  // 1 or 2
};

// Which passes move anything, and where pass p reads and writes: decided by every workgroup from the global digit
// histograms (a pass with one full bin is the identity permutation).  `flags` is a shared int; all threads call.
__device__ __forceinline__ Pass plan_pass(const unsigned* __restrict__ hist, unsigned n, int P, int p, int* flags) {
  if (threadIdx.x == 0) *flags = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < P * 256; i += kThreads)
    if (hist[i] == n) atomicOr(flags, 1 << (i >> 8));
  __syncthreads();
  const int active = ~*flags & ((1 << P) - 1);
  __syncthreads();
  Pass r;
  r.active = p >= 0 && ((active >> p) & 1);
  const int total = __builtin_popcount(active);
  const int before = p >= 0 ? __builtin_popcount(active & ((1 << p) - 1)) : 0;
  r.dst = ((total - 1 - before) & 1) ? 2 : 1;  // the last active pass writes the output
  r.src = before == 0 ? 0 : (r.dst == 1 ? 2 : 1);
  if (p < 0) r.active = total == 0;  // fix-up: copy input -> output when nothing moved
  return r;
}

template <typename K>
__device__ __forceinline__ unsigned digit_of(K key, int shift, unsigned mask) {
  return (unsigned)(key >> shift) & mask;
}

// One read of the keys: global histogram of every digit of every pass.
template <typename K>
__global__ __launch_bounds__(kThreads) void hist_all_kernel(const K* __restrict__ keys, unsigned* __restrict__ hist,
                                                            size_t n, size_t tile, int begin_bit, int end_bit, int P) {
  __shared__ unsigned h[kMaxPasses * 256];
  for (int i = threadIdx.x; i < P * 256; i += kThreads) h[i] = 0;
  __syncthreads();
  const size_t first = (size_t)blockIdx.x * tile;
  const size_t last = first + tile < n ? first + tile : n;
  for (size_t e0 = first; e0 < last; e0 += kThreads * 8) {  // eight loads in flight per thread
    K key[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const size_t e = e0 + (size_t)u * kThreads + threadIdx.x;
      key[u] = keys[e < last ? e : last - 1];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (e0 + (size_t)u * kThreads + threadIdx.x < last) {
        for (int p = 0; p < P; ++p) {
          const int shift = begin_bit + 8 * p;
          const int bits = end_bit - shift < 8 ? end_bit - shift : 8;
          atomicAdd(&h[p * 256 + digit_of(key[u], shift, (1u << bits) - 1u)], 1u);
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < P * 256; i += kThreads)
    if (h[i]) atomicAdd(&hist[i], h[i]);
}

template <typename K>
__device__ __forceinline__ const K* key_src(const Bufs<K>& b, int which) {
  return which == 0 ? b.kin : (which == 1 ? b.kout : b.ktmp);
}
template <typename K>
__device__ __forceinline__ const int* val_src(const Bufs<K>& b, int which) {
  return which == 0 ? b.vin : (which == 1 ? b.vout : b.vtmp);
}

template <typename K>
__global__ __launch_bounds__(kThreads) void count_kernel(Bufs<K> bufs, const unsigned* __restrict__ hist,
                                                         unsigned* __restrict__ blockhist, size_t n, size_t tile,
                                                         int begin_bit, int end_bit, int P, int p) {
  __shared__ unsigned h[256];
  __shared__ int flags;
  const Pass pass = plan_pass(hist, (unsigned)n, P, p, &flags);
  if (!pass.active) return;
  const K* keys = key_src(bufs, pass.src);
  h[threadIdx.x] = 0;
  __syncthreads();
  const int shift = begin_bit + 8 * p;
  const unsigned mask = (1u << (end_bit - shift < 8 ? end_bit - shift : 8)) - 1u;
  const size_t first = (size_t)blockIdx.x * tile;
  const size_t last = first + tile < n ? first + tile : n;
  for (size_t e0 = first; e0 < last; e0 += kThreads * 8) {
    K key[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const size_t e = e0 + (size_t)u * kThreads + threadIdx.x;
      key[u] = keys[e < last ? e : last - 1];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (e0 + (size_t)u * kThreads + threadIdx.x < last) atomicAdd(&h[digit_of(key[u], shift, mask)], 1u);
  }
  __syncthreads();
  blockhist[(size_t)blockIdx.x * 256 + threadIdx.x] = h[threadIdx.x];
}

template <typename K, bool VALUES>
__global__ __launch_bounds__(kThreads) void scatter_kernel(Bufs<K> bufs, const unsigned* __restrict__ hist,
                                                           const unsigned* __restrict__ blockhist, size_t n, size_t tile,
                                                           int begin_bit, int end_bit, int P, int p) {
  __shared__ unsigned gbase[256];          // next free output slot of every digit for this workgroup
  __shared__ unsigned run[kWaves][256];    // per wave: elements of the digit seen so far in the sub-tile
  __shared__ unsigned wbase[kWaves][256];  // per wave: output slot of its first element of the digit
  __shared__ unsigned wave_tot[kWaves];
  __shared__ int flags;
  const Pass pass = plan_pass(hist, (unsigned)n, P, p, &flags);
  if (!pass.active) return;
  const K* ksrc = key_src(bufs, pass.src);
  K* kdst = pass.dst == 1 ? bufs.kout : bufs.ktmp;
  const int* vsrc = VALUES ? val_src(bufs, pass.src) : nullptr;
  int* vdst = VALUES ? (pass.dst == 1 ? bufs.vout : bufs.vtmp) : nullptr;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int shift = begin_bit + 8 * p;
  const unsigned mask = (1u << (end_bit - shift < 8 ? end_bit - shift : 8)) - 1u;
  {  // digit `tid`: base = digits below it in the whole array + this digit in the workgroups before this one
    const unsigned mine = hist[p * 256 + tid];
    unsigned incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned t = __shfl_up(incl, d, 64);
      if (lane >= d) incl += t;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    unsigned base = incl - mine;
    for (int w = 0; w < wave; ++w) base += wave_tot[w];
    const unsigned* col = blockhist + tid;
    const int nb = (int)blockIdx.x;
    int b = 0;
    for (; b + 16 <= nb; b += 16) {  // 16 independent loads in flight
      unsigned part[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) part[u] = col[(size_t)(b + u) * 256];
#pragma unroll
      for (int u = 0; u < 16; ++u) base += part[u];
    }
    for (; b < nb; ++b) base += col[(size_t)b * 256];
    gbase[tid] = base;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) run[w][tid] = 0;
  }
  __syncthreads();
  const size_t first = (size_t)blockIdx.x * tile;
  const size_t last = first + tile < n ? first + tile : n;
  for (size_t sub = first; sub < last; sub += kSubTile) {
    const size_t w0 = sub + (size_t)wave * kPerWave;
    K key[kSteps];
    unsigned where[kSteps];  // digit << 16 | rank among the wave's elements of that digit
#pragma unroll
    for (int i = 0; i < kSteps; ++i) {
      const size_t e = w0 + (size_t)i * 64 + lane;
      key[i] = ksrc[e < last ? e : last - 1];
    }
#pragma unroll
    for (int i = 0; i < kSteps; ++i) {
      const size_t e = w0 + (size_t)i * 64 + lane;
      const bool valid = e < last;
      const unsigned d = digit_of(key[i], shift, mask);
      unsigned long long peers = __ballot(valid);
#pragma unroll
      for (int bit = 0; bit < 8; ++bit) {
        const bool set = (d >> bit) & 1u;
        const unsigned long long m = __ballot(set);
        peers &= set ? m : ~m;
      }
      const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(peers >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)peers, 0u));
      const unsigned r = run[wave][d];
      where[i] = (d << 16) | (r + below);
      if (valid && below == 0) run[wave][d] = r + (unsigned)__builtin_popcountll(peers);  // one leader per digit
    }
    __syncthreads();
    {  // digit `tid`: where each wave's elements of it go; advance the workgroup's slot; reset the counters
      unsigned at = gbase[tid];
#pragma unroll
      for (int w = 0; w < kWaves; ++w) {
        wbase[w][tid] = at;
        at += run[w][tid];
        run[w][tid] = 0;
      }
      gbase[tid] = at;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kSteps; ++i) {
      const size_t e = w0 + (size_t)i * 64 + lane;
      if (e < last) {
        const unsigned slot = wbase[wave][where[i] >> 16] + (where[i] & 0xFFFFu);
        kdst[slot] = key[i];
        if (VALUES) vdst[slot] = vsrc[e];
      }
    }
    __syncthreads();
  }
}

template <typename K, bool VALUES>
__global__ __launch_bounds__(kThreads) void fixup_kernel(Bufs<K> bufs, const unsigned* __restrict__ hist, size_t n, int P) {
  __shared__ int flags;
  const Pass pass = plan_pass(hist, (unsigned)n, P, -1, &flags);
  if (!pass.active) return;  // some pass moved the data into the output already
  for (size_t e = (size_t)blockIdx.x * kThreads + threadIdx.x; e < n; e += (size_t)gridDim.x * kThreads) {
    bufs.kout[e] = bufs.kin[e];
    if (VALUES) bufs.vout[e] = bufs.vin[e];
  }
}

// Stable ascending sort of keys_in[0..n) by bits [begin_bit, end_bit) into keys_out, values (int32, optional) carried
// along.  tmp: sort_bytes<K>(n, values) bytes, 256-B aligned.  keys_in / values_in are left untouched and may not alias
// the outputs.  n < 2^32.  Returns 0 or a hipError_t.
template <typename K>
inline int sort(void* tmp, size_t tmp_bytes, const K* keys_in, K* keys_out, const int* values_in, int* values_out,
                size_t n, int begin_bit, int end_bit, hipStream_t stream) {
  if (n == 0) return 0;
  const bool values = values_in != nullptr;
  const Layout<K> l = layout<K>(n, values);
  if (tmp == nullptr || tmp_bytes < l.total || n >= (1ull << 32) || end_bit <= begin_bit ||
      end_bit > (int)sizeof(K) * 8)
    return (int)hipErrorInvalidValue;
  char* base = static_cast<char*>(tmp);
  unsigned* hist = reinterpret_cast<unsigned*>(base + l.o_hist);
  unsigned* blockhist = reinterpret_cast<unsigned*>(base + l.o_blockhist);
  Bufs<K> bufs{keys_in, keys_out, reinterpret_cast<K*>(base + l.o_ktmp), values_in, values_out,
               values ? reinterpret_cast<int*>(base + l.o_vtmp) : nullptr};
  const int P = (end_bit - begin_bit + 7) / 8;
  const size_t tile = tile_of(n);
  const unsigned blocks = (unsigned)blocks_of(n);
  hipError_t e = hipMemsetAsync(hist, 0, sizeof(unsigned) * kMaxPasses * 256, stream);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL((hist_all_kernel<K>), dim3(blocks), dim3(kThreads), 0, stream, keys_in, hist, n, tile, begin_bit,
                     end_bit, P);
  for (int p = 0; p < P; ++p) {
    hipLaunchKernelGGL((count_kernel<K>), dim3(blocks), dim3(kThreads), 0, stream, bufs, hist, blockhist, n, tile,
                       begin_bit, end_bit, P, p);
    if (values)
      hipLaunchKernelGGL((scatter_kernel<K, true>), dim3(blocks), dim3(kThreads), 0, stream, bufs, hist, blockhist, n,
                         tile, begin_bit, end_bit, P, p);
    else
      hipLaunchKernelGGL((scatter_kernel<K, false>), dim3(blocks), dim3(kThreads), 0, stream, bufs, hist, blockhist, n,
                         tile, begin_bit, end_bit, P, p);
  }
  const unsigned fb = (unsigned)std::min<int64_t>(ceil_div((int64_t)n, kThreads), 1024);
  if (values)
    hipLaunchKernelGGL((fixup_kernel<K, true>), dim3(fb), dim3(kThreads), 0, stream, bufs, hist, n, P);
  else
    hipLaunchKernelGGL((fixup_kernel<K, false>), dim3(fb), dim3(kThreads), 0, stream, bufs, hist, n, P);
  return (int)hipGetLastError();
}

// ---- exclusive scan of int32 ------------------------------------------------------------------------------------------
constexpr int kScanPer = 16;  // elements per thread and step
constexpr int kScanTile = kThreads * kScanPer;
constexpr int kScanMaxBlocks = 2048;

inline size_t scan_tile_of(size_t n) {
  size_t tile = kScanTile;
  if (ceil_div((int64_t)n, (int64_t)tile) > kScanMaxBlocks)
    tile = align_up((size_t)ceil_div((int64_t)n, kScanMaxBlocks), kScanTile);
  return tile;
}
inline size_t scan_bytes(size_t n) {
  return align_up(sizeof(int) * (size_t)std::max<int64_t>(1, ceil_div((int64_t)n, (int64_t)scan_tile_of(n))), 256);
}

__device__ __forceinline__ int block_exclusive(int v, int* wave_tot, int& total) {  // all threads call
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  __syncthreads();  // wave_tot may still be read from the previous call
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  int before = 0;
  total = 0;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) {
    before += w < wave ? wave_tot[w] : 0;
    total += wave_tot[w];
  }
  return before + incl - v;
}

__global__ __launch_bounds__(kThreads) void scan_sums_kernel(const int* __restrict__ in, int* __restrict__ sums, size_t n,
                                                             size_t tile) {
  __shared__ int wave_tot[kWaves];
  const size_t first = (size_t)blockIdx.x * tile;
  const size_t last = first + tile < n ? first + tile : n;
  int s = 0;
  for (size_t e = first + threadIdx.x; e < last; e += kThreads) s += in[e];
  int total;
  block_exclusive(s, wave_tot, total);
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(kThreads) void scan_apply_kernel(const int* __restrict__ in, const int* __restrict__ sums,
                                                              int* __restrict__ out, size_t n, size_t tile, int init) {
  __shared__ int wave_tot[kWaves];
  int s = 0;
  for (int b = threadIdx.x; b < (int)blockIdx.x; b += kThreads) s += sums[b];
  int carry;
  block_exclusive(s, wave_tot, carry);
  carry += init;
  const size_t first = (size_t)blockIdx.x * tile;
  const size_t last = first + tile < n ? first + tile : n;
  for (size_t step = first; step < last; step += kScanTile) {  // a thread owns kScanPer consecutive elements
    const size_t e0 = step + (size_t)threadIdx.x * kScanPer;
    int v[kScanPer];
    int mine = 0;
#pragma unroll
    for (int i = 0; i < kScanPer; ++i) {
      v[i] = e0 + i < last ? in[e0 + i] : 0;
      mine += v[i];
    }
    int total;
    int at = carry + block_exclusive(mine, wave_tot, total);
#pragma unroll
    for (int i = 0; i < kScanPer; ++i) {
      if (e0 + i < last) out[e0 + i] = at;
      at += v[i];
    }
    carry += total;
  }
}

// out[i] = init + in[0] + ... + in[i-1]; in == out allowed.  tmp: scan_bytes(n) bytes.
inline int exclusive_scan(void* tmp, size_t tmp_bytes, const int* in, int* out, int init, size_t n, hipStream_t stream) {
  if (n == 0) return 0;
  if (tmp == nullptr || tmp_bytes < scan_bytes(n)) return (int)hipErrorInvalidValue;
  const size_t tile = scan_tile_of(n);
  const unsigned blocks = (unsigned)std::max<int64_t>(1, ceil_div((int64_t)n, (int64_t)tile));
  int* sums = static_cast<int*>(tmp);
  hipLaunchKernelGGL(scan_sums_kernel, dim3(blocks), dim3(kThreads), 0, stream, in, sums, n, tile);
  hipLaunchKernelGGL(scan_apply_kernel, dim3(blocks), dim3(kThreads), 0, stream, in, sums, out, n, tile, init);
  return (int)hipGetLastError();
}

}  // namespace devsort
}  // namespace v3d
#endif  // V3D_DEVSORT_H_
