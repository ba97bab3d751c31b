// wave_sort.h -- 64-bit keys, one per lane of a wave64: bitonic sort / merge networks and a shifted insertion.
// Used where a wave keeps the best few candidates of a query sorted across its lanes (radius.hip, query.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace v3d {

typedef unsigned long long wkey_t;
constexpr wkey_t kWaveKeyMax = ~0ull;

// Partner exchange through ds_bpermute.  (Measured: the same network on DPP quad permutes / row shifts and the gfx950
// v_permlane16/32_swap is no faster -- the hazard no-ops and extra selects cost what the crossbar round trip does -- and
// made the radius search 12-15 % slower, so the plain form stays.)
__device__ __forceinline__ wkey_t wave_xor64(wkey_t v, int j) {  // lane l <- lane l ^ j
  const int lo = __shfl_xor((int)(unsigned)v, j, 64), hi = __shfl_xor((int)(v >> 32), j, 64);
  return ((wkey_t)(unsigned)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ wkey_t wave_cex(wkey_t v, int j, bool keep_min) {
  const wkey_t o = wave_xor64(v, j);
  return ((v < o) == keep_min) ? v : o;
}
// Full bitonic sort across the lanes (21 compare-exchange steps), descending or ascending (`desc` wave-uniform).
__device__ __forceinline__ wkey_t wave_sort(wkey_t v, int lane, bool desc) {
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) v = wave_cex(v, j, ((lane & j) == 0) == (((lane & k) != 0) == desc));
  }
  return v;
}
__device__ __forceinline__ wkey_t wave_sort_desc(wkey_t v, int lane) { return wave_sort(v, lane, true); }
// v bitonic across the lanes -> ascending (6 steps).
__device__ __forceinline__ wkey_t wave_merge_asc(wkey_t v, int lane) {
#pragma unroll
  for (int j = 32; j > 0; j >>= 1) v = wave_cex(v, j, (lane & j) == 0);
  return v;
}
__device__ __forceinline__ wkey_t wave_lane64(wkey_t v, int l) {  // l wave-uniform
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(v >> 32), l);
  return ((wkey_t)hi << 32) | lo;
}
__device__ __forceinline__ int wave_shr1(int v) {  // lane l <- lane l-1, lane 0 <- 0
  return __builtin_amdgcn_update_dpp(0, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
// `kept` ascending across the lanes; insert the wave-uniform `key` (the last lane's entry falls off).
__device__ __forceinline__ wkey_t wave_insert_asc(wkey_t kept, wkey_t key, int lane) {
  const int pos = __builtin_popcountll(__ballot(kept < key));
  const unsigned lo = (unsigned)wave_shr1((int)(unsigned)kept), hi = (unsigned)wave_shr1((int)(kept >> 32));
  const wkey_t shifted = ((wkey_t)hi << 32) | lo;
  return lane == pos ? key : (lane > pos ? shifted : kept);
}
// The first n (wave-uniform, 0..64) DISTINCT keys of a per-wave LDS buffer -> ascending across the lanes, kWaveKeyMax behind
// them.  Every lane counts the keys below its own (n broadcast reads, no cross-lane round trips) and the keys change
// places through the buffer: cheaper than the 21-step network while n is a few dozen.  The buffer is left sorted.
__device__ __forceinline__ wkey_t wave_rank_sort(wkey_t* buf, int n, int lane) {
  __builtin_amdgcn_wave_barrier();
  const wkey_t mine = lane < n ? buf[lane] : kWaveKeyMax;
  int rank = 0;
#pragma unroll 8
  for (int e = 0; e < n; ++e) rank += buf[e] < mine ? 1 : 0;
  __builtin_amdgcn_wave_barrier();
  if (lane < n) buf[rank] = mine;
  __builtin_amdgcn_wave_barrier();
  const wkey_t out = lane < n ? buf[lane] : kWaveKeyMax;
  __builtin_amdgcn_wave_barrier();
  return out;
}
// Merge the first n (wave-uniform, 1..64) keys of a per-wave LDS buffer into `kept` (ascending, 64 smallest survive).
constexpr int kWaveSerialKeys = 12;  // up to here one shifted insertion per key beats the network
// `kept_empty` (wave-uniform): every entry of `kept` is kWaveKeyMax, so sorting the buffer ascending is the whole merge
// (the keys of one buffer are distinct: they carry the support index).
constexpr int kWaveRankKeys = 48;    // up to here (and nothing kept yet) counting ranks beats the 21-step network ...
constexpr int kWaveRankMinKeys = 2;  // ... and from here on the shifted insertions (3 n + ~20 instructions against ~25 n)
__device__ __forceinline__ wkey_t wave_merge_buffer(wkey_t kept, wkey_t* buf, int n, int lane,
                                                    bool kept_empty = false) {
  if (kept_empty && n >= kWaveRankMinKeys && n <= kWaveRankKeys) return wave_rank_sort(buf, n, lane);
  __builtin_amdgcn_wave_barrier();
  wkey_t h = lane < n ? buf[lane] : kWaveKeyMax;
  __builtin_amdgcn_wave_barrier();
  if (n <= kWaveSerialKeys) {
    for (int e = 0; e < n; ++e) kept = wave_insert_asc(kept, wave_lane64(h, e), lane);
  } else {
    h = wave_sort(h, lane, !kept_empty);
    kept = kept_empty ? h : wave_merge_asc(h < kept ? h : kept, lane);
  }
  return kept;
}
__device__ __forceinline__ int wave_prefix_count(unsigned long long m) {  // set bits of m below this lane
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

}  // namespace v3d
