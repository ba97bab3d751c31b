// fps_kernel.h -- the furthest-point-sampling kernel (included by fps.hip, and by tools/fps_probe.cpp with a stamping
// probe policy; the shipped library instantiates it with FpsNoProbe only).
// Semantics: furthest_point_sample_cuda.cuh:29-88 of the reference (start at point 0; each round updates the running min
// squared distance of every point to the chosen set and picks the arg-max).
//
// gfx950 design: the M-1 rounds are a serial dependency chain, so the only lever is the latency of one round.
//   * The cloud stays ON CHIP for the whole kernel: every thread keeps PPT points (x, y, z, running min-distance) in
//     VGPRs -- a CU has 512 KiB of them, 3x its LDS -- so a round touches no global memory (the reference re-reads
//     20 B/point/round through L2).
//   * arg-max per round = per-thread strict-'>' scan (packed fp32 math) -> every lane asks an LDS coordinate table for
//     ITS best point (one 16-byte read per lane, issued before the wave reduction and in flight under it: only the
//     winner lane's answer is used, but which lane wins is not known yet and the read costs no VALU time) -> six-step
//     DPP max butterfly per wave -> the winner lane writes ONE 20-byte record (x, y, z, distance | priority) into the
//     wave's LDS slot -> ONE barrier -> every lane reads slot (lane mod W) with one 16-byte + one 4-byte read issued
//     together -> log2(W) DPP steps on the distance, log2(W) on the priority among the equally-far -> three readlanes
//     hand the winner's coordinates to the next scan as scalars.  One dependent LDS round trip per round after the
//     barrier; no atomic, no second dependent read, no key reset (the slots are double-buffered by round parity: a wave
//     can be at most one barrier ahead of the slowest reader).
//     (r01 published a 64-bit key with ds_max_u64, read it back, decoded the index and then read an LDS coordinate
//     table: three dependent LDS round trips, ~1,400 cycles per round for a ~330-cycle scan.)
//   * Ties: the reference's winner among equally-far points depends on ITS launch geometry (block = opt_n_threads(N),
//     per-thread strided first-max, then a shared-memory tree that prefers the lower slot): the winner is the point with
//     the smallest (bit-reversed (p mod block), p div block).  That pair, packed as (bitrev << 22 | p div block), is the
//     `priority`: inside a thread it grows with the scan order (our block size is a multiple of the reference's), across
//     lanes it is only consulted on exact ties, across waves it decides among equal distances -- so our result equals
//     the reference's whatever OUR geometry.
//   * Clouds whose table does not fit LDS (N > 8192) select the winner's coordinates out of the lane's own registers
//     instead (kFpsRegs); past the register budget (N > 16384) the streaming variant takes over (points from L2 each
//     round, distances in the caller's scratch buffer); same reduction, same tie rule.
#ifndef V3D_FPS_KERNEL_H_
#define V3D_FPS_KERNEL_H_

#include <math.h>

#include <type_traits>

#include "common.h"

namespace v3d {

constexpr int kPrioShift = 22;  // priority = bitrev(p mod block) << 22 | (p div block); p div block < 2^22

// DPP reductions as single asm blocks: hipcc expands fmaxf(v, dpp(v)) into five instructions (a zeroing move, the DPP
// move, a NaN-canonicalising max and the max) and pads every separate asm statement with a wait state of its own; a
// round's serial chain is paid per INSTRUCTION ISSUED (~4.5 cycles each, times the waves sharing a SIMD), so the chains
// are spelled out.  `s_nop 1` = the two wait states a DPP read needs after the VALU write of the same register.  Lanes a
// row_mask leaves out keep their value (the destination is the source).  Inputs are never NaN.
#define V3D_DPP_STEP(OP, CTRL) "s_nop 1\n\t" OP " %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\t"
#define V3D_DPP_STEP_M(OP, CTRL, MASK) "s_nop 1\n\t" OP " %1, %1, %1 " CTRL " row_mask:" MASK " bank_mask:0xf\n\t"
#define V3D_WAVE_REDUCE(OP)                                                                      \
  V3D_DPP_STEP(OP, "quad_perm:[1,0,3,2]")                                                        \
  V3D_DPP_STEP(OP, "quad_perm:[2,3,0,1]")                                                        \
  V3D_DPP_STEP(OP, "row_half_mirror")                                                            \
  V3D_DPP_STEP(OP, "row_mirror")                                                                 \
  V3D_DPP_STEP_M(OP, "row_bcast:15", "0xa") /* rows 1,3 <- op with the last lane of rows 0,2 */ \
  V3D_DPP_STEP_M(OP, "row_bcast:31", "0xc") /* rows 2,3 <- op with lane 31 */                   \
  "s_nop 1\n\tv_readlane_b32 %0, %1, 63"

// max over the 64 lanes (any sign, no NaN), returned wave-uniform
__device__ __forceinline__ float wave_max_f32(float v) {
  int out;
  asm volatile(V3D_WAVE_REDUCE("v_max_f32_dpp") : "=s"(out), "+v"(v));
  return __int_as_float(out);
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
  uint32_t out;
  asm volatile(V3D_WAVE_REDUCE("v_min_u32_dpp") : "=s"(out), "+v"(v));
  return out;
}
// max / min over aligned groups of G consecutive lanes (G = 1, 2, 4, 8, 16), result in every lane of the group
#define V3D_GROUP_REDUCE(OP, G, v)                                                                                  \
  do {                                                                                                              \
    int unused_;                                                                                                    \
    if (G == 2) asm volatile(V3D_DPP_STEP(OP, "quad_perm:[1,0,3,2]") "s_nop 1" : "=s"(unused_), "+v"(v));           \
    if (G == 4)                                                                                                     \
      asm volatile(V3D_DPP_STEP(OP, "quad_perm:[1,0,3,2]") V3D_DPP_STEP(OP, "quad_perm:[2,3,0,1]") "s_nop 1"        \
                   : "=s"(unused_), "+v"(v));                                                                       \
    if (G == 8)                                                                                                     \
      asm volatile(V3D_DPP_STEP(OP, "quad_perm:[1,0,3,2]") V3D_DPP_STEP(OP, "quad_perm:[2,3,0,1]")                  \
                       V3D_DPP_STEP(OP, "row_half_mirror") "s_nop 1"                                                \
                   : "=s"(unused_), "+v"(v));                                                                       \
    if (G == 16)                                                                                                    \
      asm volatile(V3D_DPP_STEP(OP, "quad_perm:[1,0,3,2]") V3D_DPP_STEP(OP, "quad_perm:[2,3,0,1]")                  \
                       V3D_DPP_STEP(OP, "row_half_mirror") V3D_DPP_STEP(OP, "row_mirror") "s_nop 1"                 \
                   : "=s"(unused_), "+v"(v));                                                                       \
  } while (0)
template <int G>
__device__ __forceinline__ float group_max_f32(float v) {
  V3D_GROUP_REDUCE("v_max_f32_dpp", G, v);
  return v;
}
template <int G>
__device__ __forceinline__ uint32_t group_min_u32(uint32_t v) {
  V3D_GROUP_REDUCE("v_min_u32_dpp", G, v);
  return v;
}

}  // namespace v3d
#include "fps_chain.h"
namespace v3d {

typedef float f32x2 __attribute__((ext_vector_type(2)));
// min of two non-NaN floats as ONE v_min_f32 (fminf costs an extra NaN-canonicalising v_max per operand)
__device__ __forceinline__ float fps_min(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float fps_max3(float a, float b, float c) {  // no NaN: plain v_max3_f32
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float fps_max(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// Probe policy of the shipped library: nothing.  tools/fps_probe.cpp supplies one that stamps s_memtime per segment.
struct FpsNoProbe {
  __device__ __forceinline__ void begin() {}
  __device__ __forceinline__ void mark(int) {}
  __device__ __forceinline__ void flush() {}
};

// A thread's PPT values as ONE register tuple, so that element [i] with a wave-uniform i is a single indexed move
// (s_set_gpr_idx / v_movrels) instead of a select chain.
template <int P>
struct FpsRegs {
  typedef float type __attribute__((ext_vector_type(P)));
};
template <>
struct FpsRegs<1> {
  struct type {
    float v;
    __device__ __forceinline__ float& operator[](int) { return v; }
    __device__ __forceinline__ const float& operator[](int) const { return v; }
  };
};

enum FpsMode { kFpsTable = 0, kFpsRegs = 1, kFpsStream = 2 };

__device__ __forceinline__ int sgpr_popcount64(unsigned long long m) {  // one s_bcnt1 instead of the (m & (m-1)) idiom
  int c;
  asm("s_bcnt1_i32_b64 %0, %1" : "=s"(c) : "s"(m) : "scc");
  return c;
}
__device__ __forceinline__ uint32_t fps_decode_prio(uint32_t prio, int lb) {  // priority -> point index (:85-86)
  const uint32_t wq = prio & ((1u << kPrioShift) - 1u), wbr = prio >> kPrioShift;
  return (wq << lb) + (lb ? (__brev(wbr) >> (32 - lb)) : 0u);
}

// kFpsTable : points + distances in VGPRs; a lane's best point is looked up in an LDS float4 table (dynamic LDS of
//             THREADS * PPT * 16 bytes); scan by v_max3 + fps_pick
// kFpsRegs  : the same scan, but the winner lane re-reads its best point's coordinates from the cloud in L2: clouds
//             whose table does not fit LDS
// kFpsStream: points streamed from L2 every round, distances in the caller's buffer (PPT unused)
// THREADS is a multiple of the reference block size, so all points of one thread share (p mod block): their tie
// priority grows with the scan order and the first maximum in scan order is the reference's winner.
// LOG_SPLIT = j > 0: THREADS = (reference block) / 2^j, i.e. a thread's points fall into 2^j residue classes of the
// reference block.  Register slot i then holds point tid + THREADS * k(i) with k(i) = (q << j) | bitrev_j(c) for
// c = i / Q, q = i % Q, Q = PPT >> j: slots ordered by (bit-reversed class, p div block), which is again the reference's
// tie priority order, so "first maximum in slot order" stays the reference's winner.  (j = 0: k(i) = i.)
template <int J>
__host__ __device__ constexpr int fps_bitrev(int c) {
  int r = 0;
  for (int b = 0; b < J; ++b) r |= ((c >> b) & 1) << (J - 1 - b);
  return r;
}
template <int PPT, int J>
__host__ __device__ constexpr int fps_slot_k(int i) {
  return J == 0 ? i : (((i % (PPT >> J)) << J) | fps_bitrev<J>(i / (PPT >> J)));
}

template <int THREADS, int PPT, int MODE, class PROBE = FpsNoProbe, int LOG_SPLIT = 0>
__global__ __launch_bounds__(THREADS) void fps_kernel(const float* __restrict__ pts, float* __restrict__ dist_io,
                                                      int64_t* __restrict__ out_idx, int N, int M, int lb,
                                                      int first, int ld) {
  constexpr int W = THREADS / 64;
  constexpr bool STREAM = MODE == kFpsStream;
  static_assert(W == 1 || W == 2 || W == 4 || W == 8 || W == 16, "waves per workgroup");
  extern __shared__ float4 table[];                        // kFpsTable only
  __shared__ float4 slot_xyzp[2][W];                       // round parity x wave: the wave's winner coordinates + tie priority
  __shared__ __align__(16) float slot_d[2][W < 4 ? 4 : W];  // ... and its distance (contiguous: read as a whole)
  if (M <= 0) return;                                      // furthest_point_sample_cuda.cuh:37
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform, in a scalar register
  const int b = blockIdx.x;
  pts += (int64_t)b * N * 3;
  dist_io += (int64_t)b * N;
  out_idx += (int64_t)b * ld;  // rounds [first, M) of a row of ld samples

  // priority of this thread's slot i (see the header comment): LOG_SPLIT == 0: prio_base + i * prio_step;
  // LOG_SPLIT == j: prio_base + ((i / Q) << kPrioShift) + (i % Q), Q = PPT >> j (tid < block / 2^j, so tid >> lb == 0)
  constexpr int J = LOG_SPLIT, LQ = J ? __builtin_ctz(PPT >> J) : 0;
  static_assert(J == 0 || (PPT >> J) >= 1, "more residue classes than slots");
  static_assert(J == 0 || MODE == kFpsTable, "split blocks are a table-mode geometry");
  const int lbt = lb - J;  // bits of the block residue that come from tid
  const uint32_t low = (uint32_t)tid & ((1u << lbt) - 1u);
  const uint32_t prio_base =
      (((lbt ? (__brev(low) >> (32 - lbt)) : 0u) << J) << kPrioShift) | (J ? 0u : ((uint32_t)tid >> lb));
  const uint32_t prio_step = (uint32_t)THREADS >> lb;
  auto slot_prio = [&](int i) -> uint32_t {
    return J == 0 ? prio_base + (uint32_t)i * prio_step
                  : prio_base + (((uint32_t)i >> LQ) << kPrioShift) + ((uint32_t)i & ((1u << LQ) - 1u));
  };

  typename FpsRegs<PPT>::type px, py, pz, pd;
  if (!STREAM) {
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int p = fps_slot_k<PPT, J>(i) * THREADS + tid;
      const bool ok = p < N;
      const int pp = ok ? p : 0;
      px[i] = pts[pp * 3 + 0];
      py[i] = pts[pp * 3 + 1];
      pz[i] = pts[pp * 3 + 2];
      pd[i] = ok ? dist_io[pp] : -1.0f;  // padding: min(d, -1) = -1 never beats a real point (d >= 0)
      if (MODE == kFpsTable) table[i * THREADS + tid] = make_float4(px[i], py[i], pz[i], 0.f);  // by slot
    }
  }
  if (MODE == kFpsTable) __syncthreads();
  // old = 0 (:50,:57-59) -- or, resuming at round `first` (a multiple of 64; the running min-distances arrive in
  // dist_io as the previous launch left them), the sample of round first - 1
  // A resumed call trusts indices[:, :first] to come from the same (points, N) run; a value outside [0, N) (a buffer
  // that was never sampled into) is read as point 0 instead of being used as an address.
  int old = 0;
  if (first > 0) {
    const int64_t prev = out_idx[first - 1];
    old = (prev >= 0 && prev < (int64_t)N) ? (int)prev : 0;
  }
  float x1 = pts[old * 3 + 0], y1 = pts[old * 3 + 1], z1 = pts[old * 3 + 2];
  int hist = 0;  // wave 0: lane j <-> the winner's priority of round (chunk * 64 + j); round 0 = priority 0 = point 0 (:51)

  PROBE probe;
  probe.begin();
  // One round.  PAR = s & 1 is a compile-time constant (the loop below alternates two instantiations) so that every LDS
  // address of the double-buffered slots is loop-invariant: no per-round address arithmetic on the serial chain.
  auto round = [&](auto par_tag, int s) {
    constexpr int PAR = decltype(par_tag)::value;
    probe.mark(0);
    float bd = -1.0f, bx = 0.f, by = 0.f, bz = 0.f;  // best = -1 (:56)
    int bi = 0;
    float wmax;
    if constexpr (MODE != kFpsStream) {
      // Scan: two points per packed fp32 instruction (v_pk_add / v_pk_mul: same IEEE results as the scalar forms), the
      // running maximum by v_max3 (no compare, no mask hazard on the chain).  WHICH point holds the maximum -- the first
      // one in scan order, :67's strict '>' -- and the wave's maximum come out of fps_pick's interleaved block.
#pragma unroll
      for (int i = 0; i < PPT; i += 2) {
        if (PPT % 2 == 0) {
          const f32x2 dx = f32x2{x1, x1} - f32x2{px[i], px[i + 1]};
          const f32x2 dy = f32x2{y1, y1} - f32x2{py[i], py[i + 1]};
          const f32x2 dz = f32x2{z1, z1} - f32x2{pz[i], pz[i + 1]};
          const f32x2 d2 = (dx * dx + dy * dy) + dz * dz;  // :64, reference order, no contraction
          pd[i] = fps_min(d2[0], pd[i]);                   // :65-66 (kept in registers; written back once at the end)
          pd[i + 1] = fps_min(d2[1], pd[i + 1]);
          bd = fps_max3(bd, pd[i], pd[i + 1]);
        } else {  // PPT == 1
          const float dx = x1 - px[i], dy = y1 - py[i], dz = z1 - pz[i];
          pd[i] = fps_min((dx * dx + dy * dy) + dz * dz, pd[i]);
          bd = fps_max(bd, pd[i]);
        }
      }
      probe.mark(1);
      fps_pick(FpsCount<PPT>(), bd, pd, bi, wmax);
    } else {
      {
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
      probe.mark(1);
      wmax = wave_max_f32(bd);
    }
    // Padding lanes carry -1: a wave of padding only publishes -1 and never wins (wave 0 owns point 0).
    float4 mine4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE == kFpsTable) mine4 = table[bi * THREADS + tid];  // every lane asks for ITS best point (table is padded)
    const uint32_t prio = slot_prio(bi);
    bool win = bd == wmax;
    probe.mark(2);
    if (sgpr_popcount64(__ballot(win)) > 1) {  // exact tie inside the wave (rare): lowest priority value wins
      const uint32_t best = wave_min_u32(win ? prio : 0xffffffffu);
      win = win && prio == best;
    }
    if (win) {
      if (MODE == kFpsTable) {
        bx = mine4.x;
        by = mine4.y;
        bz = mine4.z;
      } else if (MODE == kFpsRegs) {
        // The winner lane re-reads its point from the cloud (L2-resident: ~200 cycles on a ~3,000-cycle round at these
        // sizes).  Taking it out of the lane's own registers needs either a dynamic index (pins the tuples to aligned
        // 16-register blocks) or a 16-way branch tree: both spill under the 128-register budget of 1024 threads.
        const int p = (bi * THREADS + tid) * 3;  // LOG_SPLIT == 0 in this mode: slot i holds point i * THREADS + tid
        bx = pts[p + 0];
        by = pts[p + 1];
        bz = pts[p + 2];
      }
      slot_xyzp[PAR][wave] = make_float4(bx, by, bz, __uint_as_float(prio));
      slot_d[PAR][wave] = bd;
    }
    probe.mark(3);
    __syncthreads();
    probe.mark(4);
    // Lane l holds record (l mod W); every lane also reads all W distances (one broadcast read) and takes their
    // maximum with v_max3 -- no cross-lane step: a DPP reduction would pay its wait states here with nothing to fill them.
    const float4 cand = slot_xyzp[PAR][lane & (W - 1)];
    const float cd = slot_d[PAR][lane & (W - 1)];
    float dmax;
    {
      const float4* dv = reinterpret_cast<const float4*>(slot_d[PAR]);
      const float4 a = dv[0];
      if (W == 1) dmax = a.x;
      else if (W == 2) dmax = fps_max(a.x, a.y);
      else {
        dmax = fps_max(fps_max3(a.x, a.y, a.z), a.w);
#pragma unroll
        for (int j = 1; j < W / 4; ++j) {
          const float4 c = dv[j];
          dmax = fps_max3(fps_max3(dmax, c.x, c.y), c.z, c.w);
        }
      }
    }
    uint32_t top = (uint32_t)__ballot(cd == dmax) & ((1u << W) - 1u);  // the first group of W lanes speaks for all
    if (W > 1 && (top & (top - 1u)) != 0u) {  // equally far winners in several waves (rare): lowest priority value wins
      const uint32_t cp = __float_as_uint(cand.w);
      const uint32_t best = group_min_u32<W>(cd == dmax ? cp : 0xffffffffu);
      top = (uint32_t)__ballot(cd == dmax && cp == best) & ((1u << W) - 1u);
    }
    const int src = __builtin_ctz(top);  // never 0: some record attains the maximum
    probe.mark(5);
    x1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cand.x), src));
    y1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cand.y), src));
    z1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cand.z), src));
    if (wave == 0) {
      // The winner's priority is parked in lane (s mod 64) of `hist`; every 64 rounds the 64 parked priorities are
      // decoded into point indices and stored with one coalesced write -- three instructions per round on the serial
      // chain of one wave instead of a compare, an exec-mask round trip, the decode and a store in every wave.
      asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0"  // (two different SGPR operands exceed the constant bus)
          : "+v"(hist)
          : "s"(__builtin_amdgcn_readlane(__float_as_int(cand.w), src)), "s"(s & 63)
          : "m0");
      if ((s & 63) == 63) out_idx[(s & ~63) + lane] = (int64_t)fps_decode_prio((uint32_t)hist, lb);
    }
    probe.mark(6);
  };
  int s = 1;
  if (first > 0) {  // even: its round uses the parity-0 slots, then the pairs go on as below
    round(std::integral_constant<int, 0>(), first);
    s = first + 1;
  }
  for (; s < M; s += 2) {
    round(std::integral_constant<int, 1>(), s);
    if (s + 1 < M) round(std::integral_constant<int, 0>(), s + 1);
  }
  probe.flush();

  if (wave == 0 && ((M - 1) & 63) != 63 && lane <= ((M - 1) & 63))  // the rounds parked since the last full flush
    out_idx[((M - 1) & ~63) + lane] = (int64_t)fps_decode_prio((uint32_t)hist, lb);

  if (!STREAM) {  // leave the scratch buffer as the reference does: final running min-distances
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int p = fps_slot_k<PPT, J>(i) * THREADS + tid;
      if (p < N) dist_io[p] = pd[i];
    }
  }
}

}  // namespace v3d
#endif  // V3D_FPS_KERNEL_H_
