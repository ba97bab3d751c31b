// interp.hip -- three_interpolate forward (and the C entry points of its backward, which shares the
// atomic-free CSR machinery of gather.hip).  Semantics: three_interpolate_cuda.cuh:11-44 / :48-83.
//
// out[b,c,q] = src[b,c,i0]*w0 + src[b,c,i1]*w1 + src[b,c,i2]*w2, products rounded before the adds, left to
// right (the source expression without FMA contraction == the CPU oracle, bit for bit).
// HBM-write bound: each thread owns 4 consecutive targets (16-byte store), keeps their 12 indices and 12
// weights in registers across a tile of channels (the reference re-reads both per channel), and the (b,c)
// source row is staged through LDS, double buffered, one barrier per channel.
#include "common.h"

namespace v3d {

int three_interp_bwd_f32_impl(const float*, const int64_t*, const float*, float*, int, int, int, int, void*, size_t,
                              int, void*);
int three_interp_bwd_f64_impl(const double*, const int64_t*, const double*, double*, int, int, int, int, void*,
                              size_t, int, void*);

constexpr int kInterpThreads = 256;

template <typename T, int VEC>
struct alignas(sizeof(T) * VEC) TPack {
  T v[VEC];
};

// Indices and weights of the VEC consecutive targets starting at q0 (3 neighbours each).  Interior, 16-byte aligned
// tiles use 16-byte loads; out-of-range indices (undefined behaviour in the reference) are redirected to `zero_slot`.
template <typename T, int VEC>
__device__ __forceinline__ void load_neighbours(const int64_t* __restrict__ idx_b, const T* __restrict__ w_b, int q0,
                                                int Q, int S, int zero_slot, int (&id)[VEC][3], T (&wt)[VEC][3]) {
  const int64_t* ip = idx_b + (int64_t)q0 * 3;
  const T* wp = w_b + (int64_t)q0 * 3;
  const bool interior = q0 + VEC <= Q;
  if (interior && ((reinterpret_cast<uintptr_t>(ip) | reinterpret_cast<uintptr_t>(wp)) & 15) == 0) {
    constexpr int kIdxLoads = 3 * VEC / 2;  // two int64 per load
    int flat[3 * VEC];
#pragma unroll
    for (int i = 0; i < kIdxLoads; ++i) {
      const longlong2 t = reinterpret_cast<const longlong2*>(ip)[i];
      flat[2 * i] = (int)t.x;
      flat[2 * i + 1] = (int)t.y;
    }
    TPack<T, VEC> wv[3];  // 3 * VEC weights = three 16-byte packs
#pragma unroll
    for (int i = 0; i < 3; ++i) wv[i] = reinterpret_cast<const TPack<T, VEC>*>(wp)[i];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int f = e * 3 + j;
        const int v = flat[f];
        id[e][j] = (v >= 0 && v < S) ? v : zero_slot;
        wt[e][j] = wv[f / VEC].v[f % VEC];
      }
    }
  } else {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const bool ok = q0 + e < Q;
        const int v = ok ? (int)ip[e * 3 + j] : -1;
        id[e][j] = (v >= 0 && v < S) ? v : zero_slot;
        wt[e][j] = ok ? wp[e * 3 + j] : T(0);
      }
    }
  }
}

// LDS path: a workgroup owns 256*VEC*R targets of one cloud and CT channels.  Indices and weights stay in registers
// for the whole channel tile; the (b,c) source row (S+1 elements, slot S = 0 for out-of-range indices) is double
// buffered in LDS with the next row's 16-byte loads in flight while the current row is interpolated and stored.
// NSTAGE = 16-byte vectors per thread and row (0: element-wise staging for rows that are not 16-byte tileable).
typedef unsigned interp_u32x4 __attribute__((ext_vector_type(4)));

// One channel of one thread's targets: 3 LDS gathers, 3 products, 2 adds per target (reference order, no contraction).
template <typename T, int VEC, int R>
__device__ __forceinline__ void interp_channel(const T* cur, const int (&id)[R][VEC][3], const T (&wt)[R][VEC][3],
                                               TPack<T, VEC> (&pk)[R]) {
#pragma unroll
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const T p0 = cur[id[r][e][0]] * wt[r][e][0];
      const T p1 = cur[id[r][e][1]] * wt[r][e][1];
      const T p2 = cur[id[r][e][2]] * wt[r][e][2];
      pk[r].v[e] = (p0 + p1) + p2;
    }
  }
}

// Codegen notes (same as group.hip's kernel): `src` / `out` are not __restrict__ (the no-alias promise lets the compiler
// sink the staging load below the stores, where it waits with vmcnt(0) = for every store in flight); the staging
// registers are a native vector type; the staging load is clamped, not predicated, and the last channel re-stages
// itself, so whole tiles run a loop without conditionals in which the wait before the LDS write is a counted vmcnt.
template <typename T, int VEC, int R, int NSTAGE, bool VEC_STORE>
__global__ __launch_bounds__(kInterpThreads) void three_interp_fwd_lds_kernel(
    const T* src, const int64_t* __restrict__ idx, const T* __restrict__ w, T* out, int C, int S, int Q, int CT,
    int buf_elems, GridMap gm) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  const int tid = threadIdx.x;
  int bx, by, b;
  decode_block(gm, bx, by, b);
  const int c0 = by * CT;
  const int c1 = min(c0 + CT, C);
  const int tile0 = bx * (kInterpThreads * VEC * R);

  int id[R][VEC][3];
  T wt[R][VEC][3];
  const int64_t* idx_b = idx + (int64_t)b * Q * 3;
  const T* w_b = w + (int64_t)b * Q * 3;
#pragma unroll
  for (int r = 0; r < R; ++r)
    load_neighbours<T, VEC>(idx_b, w_b, tile0 + (r * kInterpThreads + tid) * VEC, Q, S, S, id[r], wt[r]);
  if (tid == 0) {
    smem[S] = T(0);
    smem[buf_elems + S] = T(0);
  }
  const int nvec = (int)(((size_t)S * sizeof(T)) / 16);
  interp_u32x4 stg[NSTAGE > 0 ? NSTAGE : 1];
  {
    const T* row = src + ((int64_t)b * C + c0) * S;
    if (NSTAGE > 0) {
#pragma unroll
      for (int i = 0; i < NSTAGE; ++i) {
        const int v = i * kInterpThreads + tid;
        if (v < nvec) reinterpret_cast<interp_u32x4*>(smem)[v] = reinterpret_cast<const interp_u32x4*>(row)[v];
      }
    } else {
      for (int i = tid; i < S; i += kInterpThreads) smem[i] = row[i];
    }
  }
  const bool full = VEC_STORE && NSTAGE > 0 && tile0 + kInterpThreads * VEC * R <= Q;
  if (full) {  // whole tile, 16-byte stores, vector staging: nothing in the loop is conditional
    for (int c = c0; c < c1; ++c) {
      __syncthreads();
      const int which = (c - c0) & 1;
      const T* cur = smem + which * buf_elems;
      const interp_u32x4* nrow = reinterpret_cast<const interp_u32x4*>(src + ((int64_t)b * C + min(c + 1, c1 - 1)) * S);
#pragma unroll
      for (int i = 0; i < (NSTAGE > 0 ? NSTAGE : 1); ++i) stg[i] = nrow[min(i * kInterpThreads + tid, nvec - 1)];
      TPack<T, VEC> pk[R];
      interp_channel<T, VEC, R>(cur, id, wt, pk);
      T* orow = out + ((int64_t)b * C + c) * Q + tile0 + tid * VEC;
#pragma unroll
      for (int r = 0; r < R; ++r) *reinterpret_cast<TPack<T, VEC>*>(orow + r * kInterpThreads * VEC) = pk[r];
      interp_u32x4* nxt = reinterpret_cast<interp_u32x4*>(smem + (which ^ 1) * buf_elems);
#pragma unroll
      for (int i = 0; i < (NSTAGE > 0 ? NSTAGE : 1); ++i) {
        const int v = i * kInterpThreads + tid;
        if (v < nvec) nxt[v] = stg[i];
      }
    }
    return;
  }
  for (int c = c0; c < c1; ++c) {
    __syncthreads();
    const int which = (c - c0) & 1;
    const T* cur = smem + which * buf_elems;
    T* nxt = smem + (which ^ 1) * buf_elems;
    const bool have_next = c + 1 < c1;
    const T* nrow = src + ((int64_t)b * C + (have_next ? c + 1 : c)) * S;
    if (NSTAGE > 0) {
#pragma unroll
      for (int i = 0; i < NSTAGE; ++i)
        stg[i] = reinterpret_cast<const interp_u32x4*>(nrow)[min(i * kInterpThreads + tid, nvec - 1)];
    }
    T* orow = out + ((int64_t)b * C + c) * Q;
    TPack<T, VEC> pk[R];
    interp_channel<T, VEC, R>(cur, id, wt, pk);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int q0 = tile0 + (r * kInterpThreads + tid) * VEC;
      if (q0 < Q) {
        if (VEC_STORE) {
          *reinterpret_cast<TPack<T, VEC>*>(orow + q0) = pk[r];
        } else {
#pragma unroll
          for (int e = 0; e < VEC; ++e)
            if (q0 + e < Q) orow[q0 + e] = pk[r].v[e];
        }
      }
    }
    if (have_next) {
      if (NSTAGE > 0) {
#pragma unroll
        for (int i = 0; i < NSTAGE; ++i) {
          const int v = i * kInterpThreads + tid;
          if (v < nvec) reinterpret_cast<interp_u32x4*>(nxt)[v] = stg[i];
        }
      } else {
        for (int i = tid; i < S; i += kInterpThreads) nxt[i] = nrow[i];
      }
    }
  }
}

// Direct path for source rows that do not fit LDS: gathers straight from global memory (L2).
template <typename T, int VEC, bool VEC_STORE>
__global__ __launch_bounds__(kInterpThreads) void three_interp_fwd_direct_kernel(
    const T* __restrict__ src, const int64_t* __restrict__ idx, const T* __restrict__ w, T* __restrict__ out, int C,
    int S, int Q, int CT, GridMap gm) {
  const int tid = threadIdx.x;
  int bx, by, b;
  decode_block(gm, bx, by, b);
  const int c0 = by * CT;
  const int c1 = min(c0 + CT, C);
  const int q0 = (bx * kInterpThreads + tid) * VEC;
  int id[VEC][3];
  T wt[VEC][3];
  load_neighbours<T, VEC>(idx + (int64_t)b * Q * 3, w + (int64_t)b * Q * 3, q0, Q, S, -1, id, wt);
  if (q0 >= Q) return;
  for (int c = c0; c < c1; ++c) {
    const T* cur = src + ((int64_t)b * C + c) * S;
    TPack<T, VEC> pk;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const T s0 = id[e][0] >= 0 ? cur[id[e][0]] : T(0);
      const T s1 = id[e][1] >= 0 ? cur[id[e][1]] : T(0);
      const T s2 = id[e][2] >= 0 ? cur[id[e][2]] : T(0);
      const T p0 = s0 * wt[e][0];
      const T p1 = s1 * wt[e][1];
      const T p2 = s2 * wt[e][2];
      pk.v[e] = (p0 + p1) + p2;
    }
    T* orow = out + ((int64_t)b * C + c) * Q;
    if (VEC_STORE) {
      *reinterpret_cast<TPack<T, VEC>*>(orow + q0) = pk;
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e)
        if (q0 + e < Q) orow[q0 + e] = pk.v[e];
    }
  }
}

template <typename T>
static int three_interp_fwd(const T* src, const int64_t* idx, const T* w, T* out, int B, int C, int S, int Q,
                            int device, void* stream_) {
  const char* what = "three_interpolate_forward";
  V3D_REQUIRE(B >= 0 && C >= 0 && S >= 0 && Q >= 0, what, "negative dimension");
  if (B == 0 || C == 0 || Q == 0) return 0;
  V3D_REQUIRE(src && idx && w && out, what, "null pointer");
  if (B > kMaxGridYZ) {
    set_last_error("%s: B <= 65535 required", what);
    return V3D_ERR_SIZE_LIMIT;
  }
  DeviceGuard guard(device);
  if (guard.rc) return guard.rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  constexpr int VEC = 16 / sizeof(T);
  const bool vec_store = (Q % VEC == 0) && ((uintptr_t)out % 16 == 0);
  const size_t row_bytes = (size_t)S * sizeof(T);
  const bool row_vec_ok = (row_bytes % 16 == 0) && ((uintptr_t)src % 16 == 0);
  const int buf_elems = (int)(align_up(row_bytes + sizeof(T), 16) / sizeof(T));
  const bool use_lds = row_bytes <= 32768;
  const int R = (use_lds && vec_store && Q >= kInterpThreads * VEC * 2) ? 2 : 1;
  const int64_t xblocks = ceil_div(Q, (int64_t)kInterpThreads * VEC * R);
  // channel tile: large (index/weight reuse) while the launch keeps >= 2 waves per SIMD
  int CT = 32;
  while (CT > 4 && xblocks * ceil_div(C, CT) * B * (kInterpThreads / 64) < 2048) CT >>= 1;
  if (CT > C) CT = C;
  const int64_t yblocks = ceil_div(C, CT);
  if (xblocks * yblocks * B > (int64_t)INT32_MAX) {
    set_last_error("%s: launch exceeds the grid limit", what);
    return V3D_ERR_SIZE_LIMIT;
  }
  const GridMap gm = make_grid_map(xblocks, yblocks, B);
  dim3 grid((unsigned)(xblocks * yblocks * B)), block(kInterpThreads);
  if (use_lds) {
    const size_t lds = 2 * (size_t)buf_elems * sizeof(T);
    const int nvt = row_vec_ok ? (int)ceil_div((int64_t)(row_bytes / 16), kInterpThreads) : 0;
#define V3D_LAUNCH_TI(RR, NS, VS)                                                                               \
  hipLaunchKernelGGL((three_interp_fwd_lds_kernel<T, VEC, RR, NS, VS>), grid, block, lds, stream, src, idx, w, \
                     out, C, S, Q, CT, buf_elems, gm)
#define V3D_LAUNCH_TI_R(RR)                       \
  do {                                            \
    if (nvt == 0) V3D_LAUNCH_TI(RR, 0, true);     \
    else if (nvt <= 1) V3D_LAUNCH_TI(RR, 1, true);\
    else if (nvt <= 4) V3D_LAUNCH_TI(RR, 4, true);\
    else V3D_LAUNCH_TI(RR, 8, true);              \
  } while (0)
    if (!vec_store) V3D_LAUNCH_TI(1, 0, false);
    else if (R == 2) V3D_LAUNCH_TI_R(2);
    else V3D_LAUNCH_TI_R(1);
#undef V3D_LAUNCH_TI_R
#undef V3D_LAUNCH_TI
  } else {
    if (vec_store)
      hipLaunchKernelGGL((three_interp_fwd_direct_kernel<T, VEC, true>), grid, block, 0, stream, src, idx, w, out, C,
                         S, Q, CT, gm);
    else
      hipLaunchKernelGGL((three_interp_fwd_direct_kernel<T, VEC, false>), grid, block, 0, stream, src, idx, w, out, C,
                         S, Q, CT, gm);
  }
  return launch_status(what);
}

}  // namespace v3d

extern "C" {

int v3d_three_interp_fwd_f32(const float* s, const int64_t* i, const float* w, float* t, int B, int C, int S, int Q,
                             int device, void* stream) {
  return v3d::three_interp_fwd<float>(s, i, w, t, B, C, S, Q, device, stream);
}
int v3d_three_interp_fwd_f64(const double* s, const int64_t* i, const double* w, double* t, int B, int C, int S,
                             int Q, int device, void* stream) {
  return v3d::three_interp_fwd<double>(s, i, w, t, B, C, S, Q, device, stream);
}
int v3d_three_interp_bwd_f32(const float* g, const int64_t* i, const float* w, float* gs, int B, int C, int S, int Q,
                             void* ws, size_t wsb, int device, void* stream) {
  return v3d::three_interp_bwd_f32_impl(g, i, w, gs, B, C, S, Q, ws, wsb, device, stream);
}
int v3d_three_interp_bwd_f64(const double* g, const int64_t* i, const double* w, double* gs, int B, int C, int S,
                             int Q, void* ws, size_t wsb, int device, void* stream) {
  return v3d::three_interp_bwd_f64_impl(g, i, w, gs, B, C, S, Q, ws, wsb, device, stream);
}

}  // extern "C"
