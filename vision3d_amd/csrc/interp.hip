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

template <typename T, int VEC, bool USE_LDS, bool VEC_STORE>
__global__ __launch_bounds__(kInterpThreads) void three_interp_fwd_kernel(
    const T* __restrict__ src, const int64_t* __restrict__ idx, const T* __restrict__ w, T* __restrict__ out, int C,
    int S, int Q, int CT) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int tid = threadIdx.x;
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * CT;
  const int c1 = min(c0 + CT, C);
  const int q0 = (blockIdx.x * kInterpThreads + tid) * VEC;
  const size_t row_bytes = ((size_t)S * sizeof(T) + 15) & ~(size_t)15;
  T* buf[2] = {reinterpret_cast<T*>(smem_raw), reinterpret_cast<T*>(smem_raw + row_bytes)};

  int id[VEC][3];
  T wt[VEC][3];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      int64_t p = ((int64_t)b * Q + q0 + e) * 3 + j;
      bool ok = q0 + e < Q;
      int v = ok ? (int)idx[p] : -1;
      id[e][j] = (v >= 0 && v < S) ? v : -1;
      wt[e][j] = ok ? w[p] : T(0);
    }
  }

  auto stage = [&](int c, T* dst) {
    const T* row = src + ((int64_t)b * C + c) * S;
    for (int i = tid; i < S; i += kInterpThreads) dst[i] = row[i];
  };
  if (USE_LDS) stage(c0, buf[0]);
  for (int c = c0; c < c1; ++c) {
    if (USE_LDS) __syncthreads();
    const T* cur = USE_LDS ? buf[(c - c0) & 1] : src + ((int64_t)b * C + c) * S;
    if (q0 < Q) {
      TPack<T, VEC> pk;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        T s0 = id[e][0] >= 0 ? cur[id[e][0]] : T(0);
        T s1 = id[e][1] >= 0 ? cur[id[e][1]] : T(0);
        T s2 = id[e][2] >= 0 ? cur[id[e][2]] : T(0);
        T p0 = s0 * wt[e][0];
        T p1 = s1 * wt[e][1];
        T p2 = s2 * wt[e][2];
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
    if (USE_LDS && c + 1 < c1) stage(c + 1, buf[(c - c0 + 1) & 1]);
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
  const size_t row_bytes = align_up((size_t)S * sizeof(T), 16);
  const bool use_lds = row_bytes <= 32768;
  const int64_t xblocks = ceil_div(Q, (int64_t)kInterpThreads * VEC);
  int CT = 32;
  while (CT > 4 && xblocks * ceil_div(C, CT) * B < 2048) CT >>= 1;
  if (CT > C) CT = C;
  dim3 grid((unsigned)xblocks, (unsigned)ceil_div(C, CT), (unsigned)B), block(kInterpThreads);
#define V3D_LAUNCH_TI(UL, VS)                                                                                  \
  hipLaunchKernelGGL((three_interp_fwd_kernel<T, VEC, UL, VS>), grid, block, (UL) ? 2 * row_bytes : 0, stream, \
                     src, idx, w, out, C, S, Q, CT)
  if (use_lds) {
    if (vec_store) V3D_LAUNCH_TI(true, true); else V3D_LAUNCH_TI(true, false);
  } else {
    if (vec_store) V3D_LAUNCH_TI(false, true); else V3D_LAUNCH_TI(false, false);
  }
#undef V3D_LAUNCH_TI
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
