// fps.hip -- iterative furthest point sampling, one workgroup per cloud.
// Semantics: furthest_point_sample_cuda.cuh:29-88 of the reference (start at point 0; each round updates the
// running min squared distance of every point to the chosen set and picks the arg-max).
//
// gfx950 design: the M-1 rounds are a serial dependency chain, so the only lever is the latency of one round.
//   * The cloud stays ON CHIP for the whole kernel: every thread keeps PPT points (x, y, z, running
//     min-distance, tie priority) in VGPRs -- a CU has 512 KiB of them, 3x its LDS -- so a round touches no
//     global memory (the reference re-reads 20 B/point/round through L2).
//   * arg-max: one 64-bit key per point = (float bits of the distance) << 32 | ~priority; the per-thread scan,
//     a DPP butterfly inside each 16-lane row, readlane across the four rows, then ONE LDS slot per wave
//     carrying (key, x, y, z) of its winner and ONE barrier per round (slots double-buffered by round parity).
//     The winner's coordinates travel with the key, so no thread ever looks a point up by index.
//   * Ties: the reference's winner among equally-far points depends on ITS launch geometry (block =
//     opt_n_threads(N), per-thread strided first-max, then a shared-memory tree that prefers the lower slot):
//     the winner is the point with the smallest (bit-reversed (p mod block), p div block).  That pair is the
//     `priority` above, so our result equals the reference's whatever OUR block size is.
//   * Clouds larger than the register budget fall back to a streaming variant (points from L2, distances in
//     the caller's scratch buffer), same reduction.
#include <math.h>

#include "common.h"

namespace v3d {

__device__ __forceinline__ uint32_t fps_priority(int p, int lb, int qs) {
  uint32_t low = (uint32_t)p & ((1u << lb) - 1u);
  uint32_t br = lb ? (__brev(low) >> (32 - lb)) : 0u;
  return br * (uint32_t)qs + ((uint32_t)p >> lb);
}
__device__ __forceinline__ int fps_point_of_priority(uint32_t prio, int lb, int qs) {
  uint32_t br = prio / (uint32_t)qs, stride_i = prio % (uint32_t)qs;
  uint32_t low = lb ? (__brev(br) >> (32 - lb)) : 0u;
  return (int)((stride_i << lb) + low);
}

template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v) {
  int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, 0xf, 0xf, false);
  int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, 0xf, 0xf, false);
  return ((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo;
}
__device__ __forceinline__ unsigned long long umax64(unsigned long long a, unsigned long long b) {
  return a > b ? a : b;
}
// max over the 16 lanes of each DPP row, result in every lane of the row
__device__ __forceinline__ unsigned long long row_max_u64(unsigned long long v) {
  v = umax64(v, dpp_u64<0xB1>(v));   // quad_perm [1,0,3,2]
  v = umax64(v, dpp_u64<0x4E>(v));   // quad_perm [2,3,0,1]
  v = umax64(v, dpp_u64<0x141>(v));  // row_half_mirror
  v = umax64(v, dpp_u64<0x140>(v));  // row_mirror
  return v;
}
__device__ __forceinline__ unsigned long long read_lane_u64(unsigned long long v, int lane) {
  uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
  uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
  v = row_max_u64(v);
  unsigned long long a = read_lane_u64(v, 0), b = read_lane_u64(v, 16), c = read_lane_u64(v, 32),
                     d = read_lane_u64(v, 48);
  return umax64(umax64(a, b), umax64(c, d));
}

struct alignas(16) FpsSlot {
  unsigned long long key;
  float x, y, z, pad;
};

// RESIDENT: points + distances in VGPRs (PPT per thread).  Otherwise streamed from global every round.
template <int THREADS, int PPT, bool RESIDENT>
__global__ __launch_bounds__(THREADS) void fps_kernel(const float* __restrict__ pts, float* __restrict__ dist_io,
                                                      int64_t* __restrict__ out_idx, int N, int M, int lb, int qs) {
  constexpr int W = THREADS / 64;
  __shared__ FpsSlot slots[2][W];
  if (M <= 0) return;  // furthest_point_sample_cuda.cuh:37
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  pts += (int64_t)b * N * 3;
  dist_io += (int64_t)b * N;
  out_idx += (int64_t)b * M;

  float px[PPT], py[PPT], pz[PPT], pd[PPT];
  uint32_t np[PPT];  // ~priority; 0 for padding lanes so they lose every comparison
  if (RESIDENT) {
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      int p = i * THREADS + tid;
      bool ok = p < N;
      int pp = ok ? p : 0;
      px[i] = pts[pp * 3 + 0];
      py[i] = pts[pp * 3 + 1];
      pz[i] = pts[pp * 3 + 2];
      pd[i] = ok ? dist_io[pp] : 0.0f;
      np[i] = ok ? ~fps_priority(p, lb, qs) : 0u;
    }
  }

  float x1 = pts[0], y1 = pts[1], z1 = pts[2];  // old = 0  (:50,:57-59)
  if (tid == 0) out_idx[0] = 0;                 // :51

  for (int s = 1; s < M; ++s) {
    unsigned long long best = 0ull;
    float bx = 0.f, by = 0.f, bz = 0.f;
    if (RESIDENT) {
#pragma unroll
      for (int i = 0; i < PPT; ++i) {
        float dx = x1 - px[i], dy = y1 - py[i], dz = z1 - pz[i];
        float d = (dx * dx + dy * dy) + dz * dz;  // :64, reference order, no contraction
        d = fminf(d, pd[i]);                      // :65
        pd[i] = d;                                // :66 (kept in registers; written back once at the end)
        unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | np[i];
        bool gt = key > best;
        best = gt ? key : best;
        bx = gt ? px[i] : bx;
        by = gt ? py[i] : by;
        bz = gt ? pz[i] : bz;
      }
    } else {
      for (int p = tid; p < N; p += THREADS) {
        float x2 = pts[p * 3 + 0], y2 = pts[p * 3 + 1], z2 = pts[p * 3 + 2];
        float dx = x1 - x2, dy = y1 - y2, dz = z1 - z2;
        float d = (dx * dx + dy * dy) + dz * dz;
        d = fminf(d, dist_io[p]);
        dist_io[p] = d;
        unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (~fps_priority(p, lb, qs));
        bool gt = key > best;
        best = gt ? key : best;
        bx = gt ? x2 : bx;
        by = gt ? y2 : by;
        bz = gt ? z2 : bz;
      }
    }
    // wave arg-max; keys of real points are unique, so exactly one lane matches (several only if the whole
    // wave holds padding, key 0 -- then any of them may publish the losing slot)
    const unsigned long long wmax = wave_max_u64(best);
    const unsigned long long match = __ballot(best == wmax);
    FpsSlot* round_slots = slots[s & 1];
    if (lane == __builtin_ffsll((long long)match) - 1) {
      FpsSlot v;
      v.key = wmax;
      v.x = bx;
      v.y = by;
      v.z = bz;
      v.pad = 0.f;
      round_slots[wave] = v;
    }
    __syncthreads();
    // every wave reduces the W published slots on its own (no second barrier, no broadcast step)
    unsigned long long k2 = lane < W ? round_slots[lane].key : 0ull;
    unsigned long long gmax;
    if (W > 1) {
      unsigned long long r = row_max_u64(k2);  // W <= 16: one DPP row
      gmax = read_lane_u64(r, 0);
    } else {
      gmax = read_lane_u64(k2, 0);
    }
    const unsigned long long m2 = __ballot(lane < W && k2 == gmax);
    const int wwin = __builtin_ffsll((long long)m2) - 1;
    const FpsSlot win = round_slots[wwin];
    x1 = win.x;
    y1 = win.y;
    z1 = win.z;
    if (tid == 0) out_idx[s] = (int64_t)fps_point_of_priority(~(uint32_t)gmax, lb, qs);  // :85-86
  }

  if (RESIDENT) {  // leave the scratch buffer as the reference does: final running min-distances
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      int p = i * THREADS + tid;
      if (p < N) dist_io[p] = pd[i];
    }
  }
}

static int ilog2_floor(int v) {
  int l = 0;
  while ((1 << (l + 1)) <= v) ++l;
  return l;
}

}  // namespace v3d

extern "C" {

int v3d_fps_ref_block(int num_points) {
  // opt_n_threads, furthest_point_sample_cuda.cuh:9-12 -- same double expression, same truncation
  if (num_points <= 0) return 1;
  int pow_2 = static_cast<int>(std::log(static_cast<double>(num_points)) / std::log(2.0));
  int b = 1 << pow_2;
  if (b > 512) b = 512;
  if (b < 1) b = 1;
  return b;
}

int v3d_fps_f32(const float* points, float* distances, int64_t* indices, int B, int N, int M, int device,
                void* stream_) {
  using namespace v3d;
  const char* what = "furthest_point_sample";
  V3D_REQUIRE(B >= 0 && N >= 0 && M >= 0, what, "negative dimension");
  if (B == 0 || M == 0) return 0;
  V3D_REQUIRE(N > 0, what, "cannot sample from an empty cloud");
  V3D_REQUIRE(points && distances && indices, what, "null pointer");
  if ((int64_t)N * 3 > (int64_t)INT32_MAX) {
    set_last_error("%s: N*3 must fit in 31 bits", what);
    return V3D_ERR_SIZE_LIMIT;
  }
  DeviceGuard guard(device);
  if (guard.rc) return guard.rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int bs = v3d_fps_ref_block(N);
  const int lb = ilog2_floor(bs);
  const int qs = (int)ceil_div(N, bs);
#define V3D_FPS(T_, P_, R_)                                                                                \
  hipLaunchKernelGGL((fps_kernel<T_, P_, R_>), dim3(B), dim3(T_), 0, stream, points, distances, indices, N, M, \
                     lb, qs)
  if (N <= 64) V3D_FPS(64, 1, true);
  else if (N <= 256) V3D_FPS(64, 4, true);
  else if (N <= 1024) V3D_FPS(256, 4, true);
  else if (N <= 2048) V3D_FPS(512, 4, true);
  else if (N <= 4096) V3D_FPS(1024, 4, true);
  else if (N <= 8192) V3D_FPS(1024, 8, true);
  else if (N <= 16384) V3D_FPS(1024, 16, true);
  else V3D_FPS(1024, 1, false);
#undef V3D_FPS
  return launch_status(what);
}

}  // extern "C"
