// fps.hip -- iterative furthest point sampling, one workgroup per cloud.
// Semantics: furthest_point_sample_cuda.cuh:29-88 of the reference (start at point 0; each round updates the
// running min squared distance of every point to the chosen set and picks the arg-max).
//
// gfx950 design: the M-1 rounds are a serial dependency chain, so the only lever is the latency of one round
// (a wave issues one instruction every ~4.5 cycles, so a round costs ~4.5 x its per-wave instruction count plus
// the LDS / barrier round trips; measured with tools/fps_probe.cpp).
//   * The cloud stays ON CHIP for the whole kernel: every thread keeps PPT points (x, y, z, running
//     min-distance) in VGPRs -- a CU has 512 KiB of them, 3x its LDS -- so a round touches no global memory
//     (the reference re-reads 20 B/point/round through L2).  A float4 copy of the points sits in LDS only to
//     look the round's winner up by index.
//   * arg-max per round: per-thread strict-'>' scan (packed fp32 math), a six-step DPP max butterfly per wave
//     (one VALU instruction per step), then the wave's winner lane publishes ONE 64-bit key
//     (distance bits << 32 | ~priority) with an LDS atomic max; after ONE barrier every thread reads the winning
//     key back, decodes the point index and fetches its coordinates from the LDS table.  No second reduction
//     stage, no broadcast step; the key cell is triple-buffered so it can be reset without a second barrier.
//   * Ties: the reference's winner among equally-far points depends on ITS launch geometry (block =
//     opt_n_threads(N), per-thread strided first-max, then a shared-memory tree that prefers the lower slot):
//     the winner is the point with the smallest (bit-reversed (p mod block), p div block).  That pair, packed as
//     (bitrev << 22 | p div block), is the `priority`: inside a thread it grows with the scan order (our block
//     size is a multiple of the reference's), across lanes it is only consulted on exact ties, and across waves
//     it rides in the low word of the atomic key -- so our result equals the reference's whatever OUR geometry.
//   * Clouds beyond the register / LDS budget (N > 8192) take the slot variant (winner coordinates travel
//     through a per-wave LDS slot) or, past 16384 points, the streaming variant (points from L2, distances in
//     the caller's scratch buffer); same reduction, same tie rule.
#include <math.h>
#include <stdlib.h>

#include "common.h"

namespace v3d {

constexpr int kPrioShift = 22;  // priority = bitrev(p mod block) << 22 | (p div block); p div block < 2^22

// One DPP max step as a single VALU instruction (hipcc expands fmaxf(v, dpp(v)) into five: a zeroing move, the DPP
// move, a NaN-canonicalising max and the max).  `s_nop 1` covers the two wait states a DPP read needs after the
// VALU write of the same register.  Lanes a row_mask leaves out keep their value (the destination is the source).
#define V3D_DPP_MAX(v, CTRL_TEXT) asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 " CTRL_TEXT : "+v"(v))
#define V3D_DPP_UMIN(v, CTRL_TEXT) asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 " CTRL_TEXT : "+v"(v))

// max over the 64 lanes, returned wave-uniform.  Inputs are >= 0 and never NaN.
__device__ __forceinline__ float wave_max_f32(float v) {
  V3D_DPP_MAX(v, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
  V3D_DPP_MAX(v, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");
  V3D_DPP_MAX(v, "row_half_mirror row_mask:0xf bank_mask:0xf");
  V3D_DPP_MAX(v, "row_mirror row_mask:0xf bank_mask:0xf");
  V3D_DPP_MAX(v, "row_bcast:15 row_mask:0xa bank_mask:0xf");  // rows 1,3 <- max with last lane of rows 0,2
  V3D_DPP_MAX(v, "row_bcast:31 row_mask:0xc bank_mask:0xf");  // rows 2,3 <- max with lane 31
  asm volatile("s_nop 1" ::: "memory");
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
  V3D_DPP_UMIN(v, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
  V3D_DPP_UMIN(v, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");
  V3D_DPP_UMIN(v, "row_half_mirror row_mask:0xf bank_mask:0xf");
  V3D_DPP_UMIN(v, "row_mirror row_mask:0xf bank_mask:0xf");
  V3D_DPP_UMIN(v, "row_bcast:15 row_mask:0xa bank_mask:0xf");
  V3D_DPP_UMIN(v, "row_bcast:31 row_mask:0xc bank_mask:0xf");
  asm volatile("s_nop 1" ::: "memory");
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// Diagnostic build only (tools/fps_probe.cpp defines V3D_FPS_STAMPS): per-segment cycle shares of one wave.
#ifdef V3D_FPS_STAMPS
__device__ unsigned long long g_fps_stamps[16];
#define V3D_STAMP_DECL unsigned long long st_prev = __builtin_amdgcn_s_memtime(), st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define V3D_STAMP(k)                                          \
  do {                                                        \
    __builtin_amdgcn_sched_barrier(0);                        \
    unsigned long long st_now = __builtin_amdgcn_s_memtime(); \
    __builtin_amdgcn_s_waitcnt(0xC07F);                       \
    st_acc[k] += st_now - st_prev;                            \
    st_prev = st_now;                                         \
    __builtin_amdgcn_sched_barrier(0);                        \
  } while (0)
#define V3D_STAMP_FLUSH                    \
  if (blockIdx.x == 0 && threadIdx.x == 0) \
    for (int k_ = 0; k_ < 8; ++k_) g_fps_stamps[k_] = st_acc[k_]
#else
#define V3D_STAMP_DECL
#define V3D_STAMP(k)
#define V3D_STAMP_FLUSH
#endif

typedef float f32x2 __attribute__((ext_vector_type(2)));
// min of two non-NaN floats as ONE v_min_f32 (fminf costs an extra NaN-canonicalising v_max per operand)
__device__ __forceinline__ float fps_min(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

enum FpsMode { kFpsTable = 0, kFpsSlots = 1, kFpsStream = 2 };

// kFpsTable : points + distances in VGPRs, winner coordinates from an LDS float4 table (N*16 B of dynamic LDS)
// kFpsSlots : points + distances in VGPRs, winner coordinates through a per-wave LDS slot
// kFpsStream: points streamed from L2 every round, distances in the caller's buffer, per-wave slots
// THREADS is a multiple of the reference block size, so all points of one thread share (p mod block): their tie
// priority grows with the scan order and the strict '>' of the scan already picks the reference's winner.
template <int THREADS, int PPT, int MODE>
__global__ __launch_bounds__(THREADS) void fps_kernel(const float* __restrict__ pts, float* __restrict__ dist_io,
                                                      int64_t* __restrict__ out_idx, int N, int M, int lb) {
  constexpr int W = THREADS / 64;
  constexpr bool RESIDENT = MODE != kFpsStream;
  extern __shared__ float4 table[];               // kFpsTable only
  __shared__ unsigned long long cell[3];          // winning key of round s lives in cell[s % 3]
  __shared__ float4 slots[2][W];                  // kFpsSlots / kFpsStream: winner coordinates of each wave
  if (M <= 0) return;                             // furthest_point_sample_cuda.cuh:37
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  pts += (int64_t)b * N * 3;
  dist_io += (int64_t)b * N;
  out_idx += (int64_t)b * M;

  // priority of this thread's i-th point: prio_base + i * prio_step (see the header comment)
  const uint32_t low = (uint32_t)tid & ((1u << lb) - 1u);
  const uint32_t prio_base = ((lb ? (__brev(low) >> (32 - lb)) : 0u) << kPrioShift) | ((uint32_t)tid >> lb);
  const uint32_t prio_step = (uint32_t)THREADS >> lb;

  float px[PPT], py[PPT], pz[PPT], pd[PPT];
  if (RESIDENT) {
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int p = i * THREADS + tid;
      const bool ok = p < N;
      const int pp = ok ? p : 0;
      px[i] = pts[pp * 3 + 0];
      py[i] = pts[pp * 3 + 1];
      pz[i] = pts[pp * 3 + 2];
      pd[i] = ok ? dist_io[pp] : -1.0f;  // padding: min(d, -1) = -1 never beats a real point (d >= 0)
      if (MODE == kFpsTable && ok) table[p] = make_float4(px[i], py[i], pz[i], 0.f);
    }
  }
  if (tid < 3) cell[tid] = 0ull;
  float x1 = pts[0], y1 = pts[1], z1 = pts[2];  // old = 0  (:50,:57-59)
  if (tid == 0) out_idx[0] = 0;                 // :51
  __syncthreads();

  V3D_STAMP_DECL;
  for (int s = 1; s < M; ++s) {
    V3D_STAMP(0);
    float bd = -1.0f, bx = 0.f, by = 0.f, bz = 0.f;  // best = -1 (:56)
    int bi = 0;
    if (RESIDENT) {
      // two points per packed fp32 instruction (v_pk_add/v_pk_mul: same IEEE results as the scalar forms)
#pragma unroll
      for (int i = 0; i < PPT; i += 2) {
        constexpr bool PAIR = (PPT % 2 == 0);
        if (PAIR) {
          const f32x2 dx = f32x2{x1, x1} - f32x2{px[i], px[i + 1]};
          const f32x2 dy = f32x2{y1, y1} - f32x2{py[i], py[i + 1]};
          const f32x2 dz = f32x2{z1, z1} - f32x2{pz[i], pz[i + 1]};
          const f32x2 d2 = (dx * dx + dy * dy) + dz * dz;  // :64, reference order, no contraction
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const float d = fps_min(d2[e], pd[i + e]);  // :65
            pd[i + e] = d;                              // :66 (kept in registers; written back once at the end)
            const bool gt = d > bd;                     // :67 strict: the first maximum in scan order wins
            bd = gt ? d : bd;
            bi = gt ? i + e : bi;
            if (MODE == kFpsSlots) {
              bx = gt ? px[i + e] : bx;
              by = gt ? py[i + e] : by;
              bz = gt ? pz[i + e] : bz;
            }
          }
        } else {  // PPT == 1
          const float dx = x1 - px[i], dy = y1 - py[i], dz = z1 - pz[i];
          float d = (dx * dx + dy * dy) + dz * dz;
          d = fps_min(d, pd[i]);
          pd[i] = d;
          const bool gt = d > bd;
          bd = gt ? d : bd;
          bi = gt ? i : bi;
          if (MODE == kFpsSlots) {
            bx = gt ? px[i] : bx;
            by = gt ? py[i] : by;
            bz = gt ? pz[i] : bz;
          }
        }
      }
    } else {
      int i = 0;
      for (int p = tid; p < N; p += THREADS, ++i) {
        const float x2 = pts[p * 3 + 0], y2 = pts[p * 3 + 1], z2 = pts[p * 3 + 2];
        const float dx = x1 - x2, dy = y1 - y2, dz = z1 - z2;
        float d = (dx * dx + dy * dy) + dz * dz;
        d = fminf(d, dist_io[p]);
        dist_io[p] = d;
        const bool gt = d > bd;
        bd = gt ? d : bd;
        bi = gt ? i : bi;
        bx = gt ? x2 : bx;
        by = gt ? y2 : by;
        bz = gt ? z2 : bz;
      }
    }
    V3D_STAMP(1);
    // wave arg-max.  Padding-only lanes carry -1 (clamped to 0 for the butterfly) and never equal the maximum.
    const float wmax = wave_max_f32(fmaxf(bd, 0.0f));
    const unsigned long long match = __ballot(bd == wmax);
    const uint32_t prio = prio_base + (uint32_t)bi * prio_step;
    V3D_STAMP(2);
    if (match != 0ull) {  // (a wave that holds padding only publishes nothing)
      int wl = __builtin_ffsll((long long)match) - 1;
      if (__builtin_popcountll(match) > 1) {  // exact tie inside the wave: lowest priority value wins
        const bool in = (match >> lane) & 1ull;
        const uint32_t mine = in ? prio : 0xffffffffu;
        const uint32_t best = wave_min_u32(mine);
        wl = __builtin_ffsll((long long)__ballot(in && mine == best)) - 1;
      }
      if (lane == wl) {
        const unsigned long long key = ((unsigned long long)__float_as_uint(bd) << 32) | (unsigned long long)(~prio);
#ifdef V3D_FPS_EXPERIMENT_STORE
        cell[s % 3] = key;
#else
        atomicMax(&cell[s % 3], key);  // ds_max_u64: equal distances in different waves resolve by priority here
#endif
        if (MODE != kFpsTable) slots[s & 1][wave] = make_float4(bx, by, bz, 0.f);
      }
    }
    if (tid == 0) cell[(s + 1) % 3] = 0ull;  // last read two rounds ago, next written after this round's barrier
    V3D_STAMP(3);
    __syncthreads();
    V3D_STAMP(4);
    const unsigned long long key = cell[s % 3];
    const uint32_t wprio = ~(uint32_t)key;
    const uint32_t wq = wprio & ((1u << kPrioShift) - 1u), wbr = wprio >> kPrioShift;
    const int wp = (int)((wq << lb) + (lb ? (__brev(wbr) >> (32 - lb)) : 0u));  // the round's winner (:85)
    V3D_STAMP(5);
    float4 w4;
    if (MODE == kFpsTable) w4 = table[wp];
    else w4 = slots[s & 1][(wp & (THREADS - 1)) >> 6];  // the wave that owns point wp
    x1 = w4.x;
    y1 = w4.y;
    z1 = w4.z;
    if (tid == 0) out_idx[s] = (int64_t)wp;  // :86
    V3D_STAMP(6);
  }
  V3D_STAMP_FLUSH;

  if (RESIDENT) {  // leave the scratch buffer as the reference does: final running min-distances
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int p = i * THREADS + tid;
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
  const size_t table_bytes = (size_t)N * sizeof(float4);
#define V3D_FPS(T_, P_, M_)                                                                               \
  hipLaunchKernelGGL((fps_kernel<T_, P_, M_>), dim3(B), dim3(T_), (M_) == kFpsTable ? table_bytes : 0, stream, \
                     points, distances, indices, N, M, lb)
  // THREADS must be a multiple of the reference block `bs` (see the kernel comment): bs <= 512 is a power of two
  if (N <= 64) V3D_FPS(64, 1, kFpsTable);          // bs <= 64
  else if (N < 128) V3D_FPS(64, 2, kFpsTable);     // bs == 64
  else if (N < 256) V3D_FPS(128, 2, kFpsTable);    // bs == 128
  else if (N < 512) V3D_FPS(256, 2, kFpsTable);    // bs == 256
  else if (N <= 1024) V3D_FPS(512, 2, kFpsTable);  // bs == 512 from here on
  else if (N <= 2048) V3D_FPS(512, 4, kFpsTable);
  else if (N <= 4096) {
    static const int variant = getenv("V3D_FPS_VARIANT") ? atoi(getenv("V3D_FPS_VARIANT")) : 0;  // tuning aid
    if (variant == 1) V3D_FPS(1024, 4, kFpsTable);
    else if (variant == 2) V3D_FPS(512, 8, kFpsSlots);
    else V3D_FPS(512, 8, kFpsTable);
  } else if (N <= 8192) {
    static bool attr_set[64] = {};  // 128 KiB of dynamic LDS needs the opt-in, once per device
    if (device < 0 || device >= 64 || !attr_set[device]) {
      V3D_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fps_kernel<1024, 8, kFpsTable>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * (int)sizeof(float4)),
                  what);
      if (device >= 0 && device < 64) attr_set[device] = true;
    }
    V3D_FPS(1024, 8, kFpsTable);
  } else if (N <= 16384) V3D_FPS(1024, 16, kFpsSlots);
  else V3D_FPS(1024, 1, kFpsStream);
#undef V3D_FPS
  return launch_status(what);
}

}  // extern "C"
