// gather.hip -- gather / group_gather forward (any element width).  Replaces the forward kernels of gather_cuda.cuh:11-33
// and group_gather_cuda.cuh:11-33 of the reference (semantics only; the design is gfx950-first).  The
// backward lives in scatter.hip.
//
// out[b,c,t] = src[b,c,idx[b,t]]  (t runs over the M resp. M*K flat target positions): HBM-write bound.
//   One workgroup owns a tile of target positions of one cloud and walks a TILE OF CHANNELS with the indices held in
//   registers (the reference re-reads the int64 index tensor once per channel).  Stores are 16 B per lane, 1 KiB
//   contiguous per wave.  Two source paths:
//     * lds   : the (b,c) source row is staged in LDS (double buffered, the next row's loads in flight, one barrier
//               per channel) and the random 4-byte reads hit LDS banks instead of the vector L1;
//     * direct: rows too large for LDS are gathered straight from global memory (L2-resident).
//   Workgroups of one cloud are placed on one XCD (GridMap, common.h) so that source rows and index tiles are fetched
//   from HBM once.
#include "common.h"

namespace v3d {

// ------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------
template <typename E, int VEC>
struct alignas(sizeof(E) * VEC) Pack {
  E v[VEC];
};

constexpr int kGatherThreads = 256;

// Streaming store of one packed vector: the gathered output is written once and never re-read by this kernel, so it
// is stored non-temporally (V3D_GATHER_NT=0 switches to plain stores; tuning aid).
template <typename E, int VEC>
__device__ __forceinline__ void store_pack(E* dst, const Pack<E, VEC>& pk, bool nt) {
  constexpr int kBytes = sizeof(E) * VEC;
  if (nt) {
    if constexpr (kBytes == 16) {
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      __builtin_nontemporal_store(*reinterpret_cast<const u32x4*>(&pk), reinterpret_cast<u32x4*>(dst));
    } else if constexpr (kBytes == 8) {
      typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
      __builtin_nontemporal_store(*reinterpret_cast<const u32x2*>(&pk), reinterpret_cast<u32x2*>(dst));
    } else {
      __builtin_nontemporal_store(*reinterpret_cast<const unsigned int*>(&pk), reinterpret_cast<unsigned int*>(dst));
    }
  } else {
    *reinterpret_cast<Pack<E, VEC>*>(dst) = pk;
  }
}

// Index fetch with the reference's long -> int narrowing (group_gather_cuda.cuh:30); out-of-range values
// (undefined behaviour in the reference) are mapped to -1 and produce zeros instead of a fault.
__device__ __forceinline__ int fetch_index(const int64_t* __restrict__ idx, int64_t pos, int64_t limit, int N) {
  if (pos >= limit) return -1;
  int v = (int)idx[pos];
  return (v >= 0 && v < N) ? v : -1;
}

// VEC consecutive indices of one thread: two int64 per 16-byte load when the tile is interior and aligned.
template <int VEC>
__device__ __forceinline__ void fetch_indices(const int64_t* __restrict__ bidx, int64_t p0, int64_t L, int N,
                                              bool idx_vec, int (&v)[VEC]) {
  static_assert(VEC % 2 == 0, "VEC must be even");
  if (idx_vec && p0 + VEC <= L) {
    const longlong2* src2 = reinterpret_cast<const longlong2*>(bidx + p0);
#pragma unroll
    for (int e = 0; e < VEC / 2; ++e) {
      const longlong2 t = src2[e];
      const int a = (int)t.x, c = (int)t.y;
      v[2 * e] = (a >= 0 && a < N) ? a : -1;
      v[2 * e + 1] = (c >= 0 && c < N) ? c : -1;
    }
  } else {
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] = fetch_index(bidx, p0 + e, L, N);
  }
}

// LDS path.  NSTAGE = 16-byte vectors each thread moves per source row (0: rows that are not 16-byte
// tileable are staged element-wise).  Row buffers hold N+1 elements: slot N is a zero that out-of-range
// indices are redirected to, so the gather loop carries no per-element range check.  Requires 16-byte stores
// (L % VEC == 0, aligned output); everything else takes the direct kernel.
template <typename E, int VEC, int R, int NSTAGE, int T>
__global__ __launch_bounds__(T) void gather_fwd_lds_kernel(
    const E* __restrict__ src, const int64_t* __restrict__ idx, E* __restrict__ out, int C, int N, int64_t L,
    int CT, int buf_elems, int nt, GridMap gm) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  E* smem = reinterpret_cast<E*>(smem_raw);
  const int tid = threadIdx.x;
  int bx, by, b;
  decode_block(gm, bx, by, b);
  const int c0 = by * CT;
  const int c1 = min(c0 + CT, C);
  const int64_t tile0 = (int64_t)bx * (T * VEC * R);

  int my[R][VEC];
  const int64_t* bidx = idx + (int64_t)b * L;
  const bool idx_vec = (reinterpret_cast<uintptr_t>(bidx) & 15) == 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    int64_t p0 = tile0 + ((int64_t)r * T + tid) * VEC;
    int v[VEC];
    fetch_indices<VEC>(bidx, p0, L, N, idx_vec, v);
#pragma unroll
    for (int e = 0; e < VEC; ++e) my[r][e] = v[e] < 0 ? N : v[e];
  }
  if (tid == 0) {
    smem[N] = E(0);
    smem[buf_elems + N] = E(0);
  }

  const int nvec = (int)(((size_t)N * sizeof(E)) / 16);
  uint4 stg[NSTAGE > 0 ? NSTAGE : 1];
  // first row: straight into buffer 0
  {
    const E* row = src + ((int64_t)b * C + c0) * N;
    if (NSTAGE > 0) {
      const uint4* row4 = reinterpret_cast<const uint4*>(row);
      uint4* dst4 = reinterpret_cast<uint4*>(smem);
#pragma unroll
      for (int i = 0; i < NSTAGE; ++i) {
        int v = i * T + tid;
        if (v < nvec) dst4[v] = row4[v];
      }
    } else {
      for (int i = tid; i < N; i += T) smem[i] = row[i];
    }
  }
  for (int c = c0; c < c1; ++c) {
    __syncthreads();
    const int which = (c - c0) & 1;
    const E* cur = smem + which * buf_elems;
    E* nxt = smem + (which ^ 1) * buf_elems;
    const bool have_next = (c + 1 < c1);
    const E* nrow = src + ((int64_t)b * C + (have_next ? c + 1 : c)) * N;
    if (NSTAGE > 0) {  // next row's loads are in flight while this row is gathered and stored
      const uint4* row4 = reinterpret_cast<const uint4*>(nrow);
#pragma unroll
      for (int i = 0; i < NSTAGE; ++i) {
        int v = i * T + tid;
        stg[i] = (have_next && v < nvec) ? row4[v] : make_uint4(0, 0, 0, 0);
      }
    }
    E* orow = out + ((int64_t)b * C + c) * L;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int64_t p0 = tile0 + ((int64_t)r * T + tid) * VEC;
      if (p0 < L) {
        Pack<E, VEC> pk;
#pragma unroll
        for (int e = 0; e < VEC; ++e) pk.v[e] = cur[my[r][e]];
        store_pack<E, VEC>(orow + p0, pk, nt);
      }
    }
    if (have_next) {
      if (NSTAGE > 0) {
        uint4* dst4 = reinterpret_cast<uint4*>(nxt);
#pragma unroll
        for (int i = 0; i < NSTAGE; ++i) {
          int v = i * T + tid;
          if (v < nvec) dst4[v] = stg[i];
        }
      } else {
        for (int i = tid; i < N; i += T) nxt[i] = nrow[i];
      }
    }
  }
}

template <typename E, int VEC, int R, bool VEC_STORE>
__global__ __launch_bounds__(kGatherThreads) void gather_fwd_direct_kernel(
    const E* __restrict__ src, const int64_t* __restrict__ idx, E* __restrict__ out, int C, int N, int64_t L,
    int CT, GridMap gm) {
  const int tid = threadIdx.x;
  int bx, by, b;
  decode_block(gm, bx, by, b);
  const int c0 = by * CT;
  const int c1 = min(c0 + CT, C);
  const int64_t tile0 = (int64_t)bx * (kGatherThreads * VEC * R);
  int my[R][VEC];
  const int64_t* bidx = idx + (int64_t)b * L;
  const bool idx_vec = (reinterpret_cast<uintptr_t>(bidx) & 15) == 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    int64_t p0 = tile0 + ((int64_t)r * kGatherThreads + tid) * VEC;
    fetch_indices<VEC>(bidx, p0, L, N, idx_vec, my[r]);
  }
#pragma unroll 2
  for (int c = c0; c < c1; ++c) {
    const E* row = src + ((int64_t)b * C + c) * N;
    E* orow = out + ((int64_t)b * C + c) * L;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int64_t p0 = tile0 + ((int64_t)r * kGatherThreads + tid) * VEC;
      if (p0 >= L) continue;
      Pack<E, VEC> pk;
#pragma unroll
      for (int e = 0; e < VEC; ++e) pk.v[e] = my[r][e] >= 0 ? row[my[r][e]] : E(0);
      if (VEC_STORE) {
        *reinterpret_cast<Pack<E, VEC>*>(orow + p0) = pk;
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          if (p0 + e < L) orow[p0 + e] = pk.v[e];
      }
    }
  }
}

template <typename E, int VEC>
static int launch_gather_fwd(const void* src_, const int64_t* idx, void* out_, int B, int C, int N, int64_t L,
                             hipStream_t stream) {
  const E* src = static_cast<const E*>(src_);
  E* out = static_cast<E*>(out_);
  const bool vec_store = (L % VEC == 0) && ((uintptr_t)out % (sizeof(E) * VEC) == 0);
  const size_t row_bytes = (size_t)N * sizeof(E);
  const bool row_vec_ok = (row_bytes % 16 == 0) && ((uintptr_t)src % 16 == 0);
  const int buf_elems = (int)(align_up(row_bytes + sizeof(E), 16) / sizeof(E));  // N + zero slot, 16-B multiple
  // positions per workgroup: T threads x R vectors of VEC (512- and 1024-thread tiles were measured: no gain).
  constexpr int R_BIG = 8, R_SMALL = 2;
  const int nt = 1;  // streaming (nontemporal) stores: measured +-1 % against plain stores, kept
  const bool lds_fits = row_bytes <= 32768;
  int T = kGatherThreads, R = R_SMALL, CT = 16;
  int64_t xblocks = 0;
  auto plan = [&](int threads) {
    T = threads;
    R = (L >= (int64_t)T * VEC * R_BIG) ? R_BIG : R_SMALL;
    xblocks = ceil_div(L, (int64_t)T * VEC * R);
    // channel tile: as large as possible (index reuse, staging amortisation) while keeping >= 2 waves per SIMD
    // (measured on MI355X: re-loading the index tile per channel tile costs more than lower occupancy)
    CT = 16;
    while (CT > 2 && xblocks * ceil_div(C, CT) * B * (T / 64) < 2048) CT >>= 1;
    if (CT > C) CT = C;
  };
  plan(kGatherThreads);
  bool use_lds = vec_store && lds_fits && (int64_t)N * 2 <= (int64_t)T * VEC * R * CT;
  const int64_t yblocks = ceil_div(C, CT);
  if (xblocks * yblocks * B > (int64_t)INT32_MAX) {
    set_last_error("gather_forward: launch of %lld workgroups exceeds the grid limit", (long long)(xblocks * yblocks * B));
    return V3D_ERR_SIZE_LIMIT;
  }
  const GridMap gm = make_grid_map(xblocks, yblocks, B);
  dim3 grid((unsigned)(xblocks * yblocks * B));
  dim3 block(T);
  if (use_lds) {
    const size_t lds = 2 * (size_t)buf_elems * sizeof(E);
    const int nvec_per_thread = row_vec_ok ? (int)ceil_div((int64_t)(row_bytes / 16), T) : 0;
#define V3D_LAUNCH_LDS(RR, NS, TT)                                                                                 \
  hipLaunchKernelGGL((gather_fwd_lds_kernel<E, VEC, RR, NS, TT>), grid, block, lds, stream, src, idx, out, C, N, L, \
                     CT, buf_elems, nt, gm)
    {
#define V3D_LAUNCH_LDS_R(RR)                                     \
  do {                                                           \
    if (nvec_per_thread == 0) V3D_LAUNCH_LDS(RR, 0, 256);        \
    else if (nvec_per_thread <= 1) V3D_LAUNCH_LDS(RR, 1, 256);   \
    else if (nvec_per_thread <= 4) V3D_LAUNCH_LDS(RR, 4, 256);   \
    else V3D_LAUNCH_LDS(RR, 8, 256);                             \
  } while (0)
      if (R == R_BIG) V3D_LAUNCH_LDS_R(R_BIG); else V3D_LAUNCH_LDS_R(R_SMALL);
#undef V3D_LAUNCH_LDS_R
    }
#undef V3D_LAUNCH_LDS
  } else {
#define V3D_LAUNCH_DIRECT(RR, VS) \
  hipLaunchKernelGGL((gather_fwd_direct_kernel<E, VEC, RR, VS>), grid, block, 0, stream, src, idx, out, C, N, L, CT, gm)
    if (R == R_BIG) {
      if (vec_store) V3D_LAUNCH_DIRECT(R_BIG, true); else V3D_LAUNCH_DIRECT(R_BIG, false);
    } else {
      if (vec_store) V3D_LAUNCH_DIRECT(R_SMALL, true); else V3D_LAUNCH_DIRECT(R_SMALL, false);
    }
#undef V3D_LAUNCH_DIRECT
  }
  return launch_status("gather_forward");
}

int group_fused_plain_u32(const void* src, const int64_t* idx, void* out, int B, int C, int N, int64_t L,
                          hipStream_t stream);  // group.hip: 1 = shape not eligible

static int gather_fwd_dispatch(const char* what, const void* src, const int64_t* idx, void* out, int B, int C,
                               int N, int64_t L, int elem, int device, void* stream_) {
  V3D_REQUIRE(B >= 0 && C >= 0 && N >= 0 && L >= 0, what, "negative dimension");
  V3D_REQUIRE(elem == 1 || elem == 2 || elem == 4 || elem == 8, what, "elem_bytes must be 1, 2, 4 or 8");
  if (B == 0 || C == 0 || L == 0) return 0;
  V3D_REQUIRE(src && idx && out, what, "null pointer");
  if (B > kMaxGridYZ || L > (int64_t)INT32_MAX) {
    set_last_error("%s: B <= 65535 and targets per cloud < 2^31 required", what);
    return V3D_ERR_SIZE_LIMIT;
  }
  DeviceGuard guard(device);
  if (guard.rc) return guard.rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (elem == 4) {  // 4-byte rows >= 8 KiB run on group.hip's kernel (returns 1 when the shape is not its case)
    const int rc = group_fused_plain_u32(src, idx, out, B, C, N, L, stream);
    if (rc != 1) return rc;
  }
  switch (elem) {
    case 1: return launch_gather_fwd<uint8_t, 4>(src, idx, out, B, C, N, L, stream);
    case 2: return launch_gather_fwd<uint16_t, 4>(src, idx, out, B, C, N, L, stream);
    case 4: return launch_gather_fwd<uint32_t, 4>(src, idx, out, B, C, N, L, stream);
    default: return launch_gather_fwd<uint64_t, 2>(src, idx, out, B, C, N, L, stream);
  }
}

}  // namespace v3d

extern "C" {

int v3d_gather_fwd(const void* sources, const int64_t* indices, void* targets, int B, int C, int N, int M,
                   int elem_bytes, int device, void* stream) {
  return v3d::gather_fwd_dispatch("gather_forward", sources, indices, targets, B, C, N, (int64_t)M, elem_bytes,
                                  device, stream);
}

int v3d_group_gather_fwd(const void* sources, const int64_t* indices, void* targets, int B, int C, int N, int M,
                         int K, int elem_bytes, int device, void* stream) {
  if (M < 0 || K < 0) return v3d::invalid("group_gather_forward", "negative dimension");
  return v3d::gather_fwd_dispatch("group_gather_forward", sources, indices, targets, B, C, N, (int64_t)M * K,
                                  elem_bytes, device, stream);
}

}  // extern "C"
