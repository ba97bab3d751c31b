// group.hip -- fused neighbourhood grouping of the batch-mode layers.
// The reference builds the input of a set-abstraction MLP in three passes over the (B,C,M,K) tensor
// (layers/pointnet2.py:44-47):  group_gather(points) -> subtract the centroid -> group_gather(feats) -> torch.cat,
// and the input of an EdgeConv MLP the same way (layers/edge_conv.py:59-61): group_gather(feats) -> expand(q) ->
// cat([q, neighbours - q]).  Each intermediate is a full HBM round trip of the largest tensor of the layer.
//
// Here ONE kernel writes the MLP input directly.  Output channels come in three kinds, in this order:
//   [ plain : out = src0[b,c,idx] ] [ broadcast : out = ctr[b,c,m] ] [ offset : out = src1[b,c,idx] - ctr[b,c,m] ]
//   set abstraction : plain = C feature channels, broadcast = 0, offset = 3 coordinates        (C+3 channels)
//   EdgeConv        : plain = 0, broadcast = C, offset = C                                      (2C channels)
// Structure = the group-gather forward (gather.hip): a workgroup owns 8192 target positions of one cloud and walks a
// tile of output channels with its 32 indices in registers; the source row of the channel is staged in LDS (double
// buffered, next row in flight), the centroid values of the tile's queries sit in LDS beside it; 16-byte streaming
// stores.  The subtraction is the single fp32 operation the reference performs, so results are bit-identical to it.
#include <algorithm>

#include "common.h"

namespace v3d {

constexpr int kGrpThreads = 256;
constexpr int kGrpVec = 4;
constexpr int kGrpR = 8;                                   // vectors per thread
constexpr int kGrpTile = kGrpThreads * kGrpVec * kGrpR;    // 8192 target positions per workgroup
constexpr int kGrpRowVecs = 8;                             // 16-byte pieces of a source row per thread: N <= 8192
constexpr int kGrpCtrVals = 4;                             // centre values per thread: <= 1024 queries per tile

struct GroupArgs {
  const float* src0;  // (B, C0, N) plain channels (may be null when C0 == 0)
  const float* src1;  // (B, C1, N) offset channels
  const float* ctr;   // (B, Cb + C1 or C1, M) centre values: broadcast channels first when share_ctr == 0
  int C0, Cb, C1;
  int share_ctr;      // 1: broadcast channel c and offset channel c read the same centre row (EdgeConv)
  int N, M, K;
};

// kind / source row / centre row of output channel oc
__device__ __forceinline__ void grp_channel(const GroupArgs& a, int b, int oc, const float*& row, const float*& crow) {
  row = nullptr;
  crow = nullptr;
  if (oc < a.C0) {
    row = a.src0 + ((int64_t)b * a.C0 + oc) * a.N;
  } else if (oc < a.C0 + a.Cb) {
    const int c = oc - a.C0;
    const int cc = a.share_ctr ? a.C1 : a.Cb + a.C1;
    crow = a.ctr + ((int64_t)b * cc + c) * a.M;
  } else {
    const int c = oc - a.C0 - a.Cb;
    const int cc = a.share_ctr ? a.C1 : a.Cb + a.C1;
    row = a.src1 + ((int64_t)b * a.C1 + c) * a.N;
    crow = a.ctr + ((int64_t)b * cc + (a.share_ctr ? c : a.Cb + c)) * a.M;
  }
}

// Source row / centre row of output channel oc, as plain values: the three base pointers are direct kernel arguments
// (so the loads are global_load, not flat_load) and nothing is selected through memory.  An earlier version kept them in
// the by-value GroupArgs and chose the row inside lambdas: the compiler passed the chosen pointer through a scratch slot,
// and the scratch_load's s_waitcnt vmcnt(0) at the top of every channel drained all stores of the previous channel.
struct GrpSel {
  const float* row;   // valid iff has_row
  const float* crow;  // valid iff has_ctr
  bool has_row, has_ctr;
};
template <bool CTR>
__device__ __forceinline__ GrpSel grp_select(const float* src0, const float* src1, const float* ctr, const GroupArgs& a,
                                             int b, int oc) {
  GrpSel s;
  if (!CTR) {
    s.row = src0 + ((int64_t)b * a.C0 + oc) * a.N;
    s.crow = src0;
    s.has_row = true;
    s.has_ctr = false;
    return s;
  }
  const int cc = a.share_ctr ? a.C1 : a.Cb + a.C1;
  const bool plain = oc < a.C0, bcast = !plain && oc < a.C0 + a.Cb;
  const int c = plain ? oc : (bcast ? oc - a.C0 : oc - a.C0 - a.Cb);
  const int64_t r0 = ((int64_t)b * a.C0 + c) * a.N, r1 = ((int64_t)b * a.C1 + c) * a.N;
  // broadcast channels have no source row: they stage a valid dummy row (never read) so that the row loads are
  // unconditional -- conditional loads make the compiler wait for ALL outstanding stores before it reuses the registers
  const float* dummy = a.C1 ? src1 + (int64_t)b * a.C1 * a.N : src0 + (int64_t)b * a.C0 * a.N;
  s.row = plain ? src0 + r0 : (bcast ? dummy : src1 + r1);
  s.crow = ctr + ((int64_t)b * cc + (plain ? 0 : ((bcast || a.share_ctr) ? c : a.Cb + c))) * a.M;  // plain: dummy
  s.has_row = !bcast;
  s.has_ctr = !plain;
  return s;
}

// (a native vector, not HIP's uint4 struct: struct copies between address spaces become memcpy calls that keep the staging
// array in scratch memory)
typedef unsigned grp_u32x4 __attribute__((ext_vector_type(4)));
// Staging of one channel: loads into registers (addresses clamped, not predicated), later written to an LDS buffer.
template <bool CTR, int NSTAGE>
__device__ __forceinline__ void grp_issue(const GrpSel& sel, grp_u32x4 (&stg)[NSTAGE], float (&cst)[kGrpCtrVals], int tid,
                                          int nvec, int m_first, int m_count) {
  const grp_u32x4* row4 = reinterpret_cast<const grp_u32x4*>(sel.row);
#pragma unroll
  for (int i = 0; i < NSTAGE; ++i) stg[i] = row4[min(i * kGrpThreads + tid, nvec - 1)];
  if (CTR) {
#pragma unroll
    for (int i = 0; i < kGrpCtrVals; ++i) cst[i] = sel.crow[m_first + min(i * kGrpThreads + tid, m_count - 1)];
  }
}
template <bool CTR, int NSTAGE>
__device__ __forceinline__ void grp_commit(const grp_u32x4 (&stg)[NSTAGE], const float (&cst)[kGrpCtrVals], float* rows_buf,
                                           float* ctrs_buf, int tid, int nvec, int m_count) {
  grp_u32x4* dr = reinterpret_cast<grp_u32x4*>(rows_buf);
#pragma unroll
  for (int i = 0; i < NSTAGE; ++i) {
    const int v = i * kGrpThreads + tid;
    if (v < nvec) dr[v] = stg[i];
  }
  if (CTR) {
#pragma unroll
    for (int i = 0; i < kGrpCtrVals; ++i) {
      const int v = i * kGrpThreads + tid;
      if (v < m_count) ctrs_buf[v] = cst[i];
    }
  }
}

typedef float grp_f32x4 __attribute__((ext_vector_type(4)));
// The 32 values of one thread for one output channel (8 vectors of 4 consecutive positions).
template <bool KV, bool CTR>
__device__ __forceinline__ void grp_gather(const float* cur, const float* cc, bool has_row, bool has_ctr,
                                           const int (&my)[kGrpR][kGrpVec], const int (&mq)[kGrpR][KV ? 1 : kGrpVec],
                                           grp_f32x4 (&pk)[kGrpR]) {
  if (!CTR || (has_row && !has_ctr)) {  // plain gather: 32 LDS reads, no tests
#pragma unroll
    for (int r = 0; r < kGrpR; ++r) pk[r] = grp_f32x4{cur[my[r][0]], cur[my[r][1]], cur[my[r][2]], cur[my[r][3]]};
  } else if (has_row) {  // gather minus centre
#pragma unroll
    for (int r = 0; r < kGrpR; ++r) {
#pragma unroll
      for (int e = 0; e < kGrpVec; ++e) pk[r][e] = cur[my[r][e]] - cc[mq[r][KV ? 0 : e]];
    }
  } else {  // broadcast centre
#pragma unroll
    for (int r = 0; r < kGrpR; ++r) {
#pragma unroll
      for (int e = 0; e < kGrpVec; ++e) pk[r][e] = cc[mq[r][KV ? 0 : e]];
    }
  }
}
__device__ __forceinline__ void grp_store(const grp_f32x4 (&pk)[kGrpR], float* orow) {
#pragma unroll
  for (int r = 0; r < kGrpR; ++r)
    __builtin_nontemporal_store(pk[r], reinterpret_cast<grp_f32x4*>(orow + (int64_t)r * kGrpThreads * kGrpVec));
}

// KV: K % 4 == 0, so the four positions of a vector share one query.
// CTR: the launch has broadcast / offset channels (false = plain gather: no query bookkeeping at all).
template <bool KV, bool CTR, int NSTAGE>
__global__ __launch_bounds__(kGrpThreads) void group_fused_lds_kernel(const float* src0, const float* src1,
                                                                      const float* ctr, GroupArgs a,
                                                                      const int64_t* __restrict__ idx, float* out, int CT,
                                                                      int row_elems, int ctr_elems, GridMap gm) {
  // src0 / src1 / ctr / out are deliberately NOT __restrict__: with the no-alias promise the compiler sinks the staging
  // loads of the next row below this channel's stores, next to the LDS writes that consume them, and each load then
  // waits with vmcnt(0) -- i.e. for every store in flight.  As possibly-aliasing accesses they stay where they are
  // written: above the stores.
  extern __shared__ __align__(16) float grp_smem[];
  float* rows = grp_smem;                         // 2 x row_elems: source row (+ zero slot at N)
  float* ctrs = grp_smem + 2 * row_elems;         // 2 x ctr_elems: centre values of this tile's queries
  const int tid = threadIdx.x;
  int bx, by, b;
  decode_block(gm, bx, by, b);
  const int Cout = a.C0 + a.Cb + a.C1;
  const int c0 = by * CT, c1 = min(c0 + CT, Cout);
  const int64_t L = (int64_t)a.M * a.K;
  const int64_t tile0 = (int64_t)bx * kGrpTile;
  const int m_first = (int)(tile0 / a.K);
  const int m_count = (int)(min(L, tile0 + kGrpTile) - 1) / a.K - m_first + 1;
  const bool full = tile0 + kGrpTile <= L;  // every vector of every thread is inside the row: no per-store test

  int my[kGrpR][kGrpVec];
  int mq[kGrpR][KV ? 1 : kGrpVec];
  const int64_t* bidx = idx + (int64_t)b * L;
  const bool idx_vec = (reinterpret_cast<uintptr_t>(bidx) & 15) == 0;
#pragma unroll
  for (int r = 0; r < kGrpR; ++r) {
    const int64_t p0 = tile0 + ((int64_t)r * kGrpThreads + tid) * kGrpVec;
    int64_t raw[kGrpVec];
    if (idx_vec && p0 + kGrpVec <= L) {  // two int64 per 16-byte load
      const longlong2* src2 = reinterpret_cast<const longlong2*>(bidx + p0);
      const longlong2 t0 = src2[0], t1 = src2[1];
      raw[0] = t0.x; raw[1] = t0.y; raw[2] = t1.x; raw[3] = t1.y;
    } else {
#pragma unroll
      for (int e = 0; e < kGrpVec; ++e) raw[e] = (p0 + e < L) ? bidx[p0 + e] : -1;
    }
#pragma unroll
    for (int e = 0; e < kGrpVec; ++e) {
      my[r][e] = (raw[e] >= 0 && raw[e] < a.N) ? (int)raw[e] : a.N;
      if (CTR && (!KV || e == 0)) mq[r][KV ? 0 : e] = (p0 + e < L) ? (int)((p0 + e) / a.K) - m_first : 0;
    }
  }
  if (tid == 0) {
    rows[a.N] = 0.f;
    rows[row_elems + a.N] = 0.f;
  }
  // register staging: the next channel's source row (16-byte pieces) and centre values are requested before this
  // channel's gathers and stores and written to the other LDS buffer after them; the wait before that write counts the
  // stores issued in between, so they stay in flight
  const int nvec = a.N >> 2;
  grp_u32x4 stg[NSTAGE];
  float cst[kGrpCtrVals];
  typedef grp_f32x4 f32x4;
  GrpSel nxt = grp_select<CTR>(src0, src1, ctr, a, b, c0);
  grp_issue<CTR, NSTAGE>(nxt, stg, cst, tid, nvec, m_first, m_count);
  grp_commit<CTR, NSTAGE>(stg, cst, rows, ctrs, tid, nvec, m_count);
  // Two loops, chosen once: whole tiles (every path through the body issues exactly eight stores between the staging
  // loads and the LDS writes that consume them, so the compiler's wait there is a COUNTED vmcnt and the stores stay in
  // flight) and the last, partial tile of a row (per-store tests; its waits drain the queue, it is one tile per row).
  if (full) {
    // Stores run ONE CHANNEL BEHIND the gathers, from two alternating register sets: the LDS reads of channel c write
    // registers whose last stores (channel c - 2) were issued a whole iteration ago, so the wait the compiler puts in
    // front of them leaves channel c - 1's eight stores in flight instead of draining the queue every channel.
    // (Not for K % 4 != 0: the per-position query table of that variant plus two value sets need 343 registers -- one
    // wave per SIMD -- and ran 10-30 % slower than the plain loop below.)
    f32x4 pkA[kGrpR], pkB[kGrpR];
#define V3D_GRP_STEP(OC, DST, PREV, STORE_PREV)                                                                        \
  do {                                                                                                                 \
    __syncthreads();                                                                                                   \
    const int which_ = ((OC) - c0) & 1;                                                                                \
    const float* cur_ = rows + which_ * row_elems;                                                                     \
    const float* cc_ = ctrs + which_ * ctr_elems;                                                                      \
    const bool has_row_ = !CTR || nxt.has_row, has_ctr_ = CTR && nxt.has_ctr; /* of channel OC */                      \
    /* the next channel's row is requested now and written to the other buffer at the end of the step; the last     */ \
    /* channel re-stages itself (one redundant L2 read per workgroup): nothing here is conditional                  */ \
    nxt = grp_select<CTR>(src0, src1, ctr, a, b, min((OC) + 1, c1 - 1));                                               \
    grp_issue<CTR, NSTAGE>(nxt, stg, cst, tid, nvec, m_first, m_count);                                                \
    grp_gather<KV, CTR>(cur_, cc_, has_row_, has_ctr_, my, mq, DST);                                                   \
    if (STORE_PREV) grp_store(PREV, out + ((int64_t)b * Cout + (OC) - 1) * L + tile0 + (int64_t)tid * kGrpVec);        \
    grp_commit<CTR, NSTAGE>(stg, cst, rows + (which_ ^ 1) * row_elems, ctrs + (which_ ^ 1) * ctr_elems, tid, nvec,     \
                            m_count);                                                                                  \
  } while (0)
    if constexpr (KV) {
      int oc = c0;
      V3D_GRP_STEP(oc, pkA, pkB, false);
      for (oc = c0 + 1; oc + 1 < c1; oc += 2) {
        V3D_GRP_STEP(oc, pkB, pkA, true);
        V3D_GRP_STEP(oc + 1, pkA, pkB, true);
      }
      if (oc < c1) {
        V3D_GRP_STEP(oc, pkB, pkA, true);
        grp_store(pkB, out + ((int64_t)b * Cout + oc) * L + tile0 + (int64_t)tid * kGrpVec);
      } else {
        grp_store(pkA, out + ((int64_t)b * Cout + c1 - 1) * L + tile0 + (int64_t)tid * kGrpVec);
      }
    } else {
      for (int oc = c0; oc < c1; ++oc) {  // one register set, stores right behind the gathers
        V3D_GRP_STEP(oc, pkA, pkA, false);
        grp_store(pkA, out + ((int64_t)b * Cout + oc) * L + tile0 + (int64_t)tid * kGrpVec);
      }
    }
#undef V3D_GRP_STEP
  } else {
    for (int oc = c0; oc < c1; ++oc) {
      __syncthreads();
      const int which = (oc - c0) & 1;
      const float* cur = rows + which * row_elems;
      const float* cc = ctrs + which * ctr_elems;
      const bool has_row = !CTR || nxt.has_row, has_ctr = CTR && nxt.has_ctr;
      nxt = grp_select<CTR>(src0, src1, ctr, a, b, min(oc + 1, c1 - 1));
      grp_issue<CTR, NSTAGE>(nxt, stg, cst, tid, nvec, m_first, m_count);
      float* orow = out + ((int64_t)b * Cout + oc) * L + tile0 + (int64_t)tid * kGrpVec;
#pragma unroll
      for (int r = 0; r < kGrpR; ++r) {
        const int64_t p0 = tile0 + ((int64_t)r * kGrpThreads + tid) * kGrpVec;
        if (p0 < L) {
          f32x4 pk;
#pragma unroll
          for (int e = 0; e < kGrpVec; ++e) {
            const float g = has_row ? cur[my[r][e]] : 0.f;
            const float q = has_ctr ? cc[mq[r][KV ? 0 : e]] : 0.f;
            pk[e] = has_row ? (has_ctr ? g - q : g) : q;
          }
          __builtin_nontemporal_store(pk, reinterpret_cast<f32x4*>(orow + (int64_t)r * kGrpThreads * kGrpVec));
        }
      }
      grp_commit<CTR, NSTAGE>(stg, cst, rows + (which ^ 1) * row_elems, ctrs + (which ^ 1) * ctr_elems, tid, nvec,
                              m_count);
    }
  }
}

// Generic path (any N, any L): one thread per target position, everything from global memory / L2.
__global__ __launch_bounds__(kGrpThreads) void group_fused_direct_kernel(GroupArgs a, const int64_t* __restrict__ idx,
                                                                         float* __restrict__ out) {
  const int b = blockIdx.z, oc = blockIdx.y;
  const int64_t L = (int64_t)a.M * a.K;
  const int64_t p = (int64_t)blockIdx.x * kGrpThreads + threadIdx.x;
  if (p >= L) return;
  const int Cout = a.C0 + a.Cb + a.C1;
  const float *row, *crow;
  grp_channel(a, b, oc, row, crow);
  const int64_t raw = idx[(int64_t)b * L + p];
  const bool ok = raw >= 0 && raw < a.N;
  const float g = (row && ok) ? row[raw] : 0.f;
  const float q = crow ? crow[p / a.K] : 0.f;
  out[((int64_t)b * Cout + oc) * L + p] = row ? (crow ? g - q : g) : q;
}

static bool group_lds_ok(const GroupArgs& a, const float* out) {
  const int64_t L = (int64_t)a.M * a.K;
  const size_t row_elems = align_up((size_t)(a.N + 1) * 4, 16) / 4;
  const size_t ctr_elems = align_up((size_t)(kGrpTile / std::max(a.K, 1) + 2) * 4, 16) / 4;
  const size_t lds = sizeof(float) * 2 * (row_elems + ctr_elems);
  const uintptr_t srcs = (uintptr_t)a.src0 | (uintptr_t)a.src1;
  return a.N % 4 == 0 && a.N >= 4 && a.C0 + a.C1 > 0 && a.N <= kGrpRowVecs * kGrpThreads * 4 && (srcs & 15) == 0 &&
         kGrpTile / std::max(a.K, 1) + 2 <= kGrpCtrVals * kGrpThreads && lds <= 65536 && L % kGrpVec == 0 &&
         ((uintptr_t)out & 15) == 0 && L >= kGrpTile / 4;
}

static int launch_group_fused(const char* what, const GroupArgs& a, const int64_t* idx, float* out, int B,
                              hipStream_t stream) {
  const int Cout = a.C0 + a.Cb + a.C1;
  const int64_t L = (int64_t)a.M * a.K;
  if (B == 0 || Cout == 0 || L == 0) return 0;
  if (B > kMaxGridYZ || Cout > kMaxGridYZ || L > (int64_t)INT32_MAX) {
    set_last_error("%s: B, channels <= 65535 and M*K < 2^31 required", what);
    return V3D_ERR_SIZE_LIMIT;
  }
  const int row_elems = (int)(align_up((size_t)(a.N + 1) * 4, 16) / 4);
  const int ctr_elems = (int)(align_up((size_t)(kGrpTile / std::max(a.K, 1) + 2) * 4, 16) / 4);
  const size_t lds = sizeof(float) * 2 * ((size_t)row_elems + ctr_elems);
  const bool lds_ok = group_lds_ok(a, out);
  if (!lds_ok) {
    hipLaunchKernelGGL(group_fused_direct_kernel, dim3((unsigned)ceil_div(L, kGrpThreads), Cout, B), dim3(kGrpThreads), 0,
                       stream, a, idx, out);
    return launch_status(what);
  }
  const int64_t xblocks = ceil_div(L, kGrpTile);
  int CT = 16;
  while (CT > 2 && xblocks * ceil_div(Cout, CT) * B * (kGrpThreads / 64) < 2048) CT >>= 1;
  if (CT > Cout) CT = Cout;
  const int64_t yblocks = ceil_div(Cout, CT);
  const GridMap gm = make_grid_map(xblocks, yblocks, B);
  const dim3 grid((unsigned)(xblocks * yblocks * B)), block(kGrpThreads);
#define V3D_GRP_LAUNCH_NS(KV, CTR, NS)                                                                                \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(group_fused_lds_kernel<KV, CTR, NS>), grid, block, lds, stream, a.src0, a.src1, \
                     a.ctr, a, idx, out, CT, row_elems, ctr_elems, gm)
#define V3D_GRP_LAUNCH(KV, CTR)                            \
  do {                                                     \
    if (per_thread <= 1) V3D_GRP_LAUNCH_NS(KV, CTR, 1);    \
    else if (per_thread <= 2) V3D_GRP_LAUNCH_NS(KV, CTR, 2); \
    else if (per_thread <= 4) V3D_GRP_LAUNCH_NS(KV, CTR, 4); \
    else V3D_GRP_LAUNCH_NS(KV, CTR, 8);                    \
  } while (0)
  const int per_thread = (int)ceil_div(a.N / 4, kGrpThreads);  // 16-byte pieces of a source row per thread
  const bool ctr = a.Cb + a.C1 > 0;
  if (!ctr) V3D_GRP_LAUNCH(true, false);
  else if (a.K % kGrpVec == 0) V3D_GRP_LAUNCH(true, true);
  else V3D_GRP_LAUNCH(false, true);
#undef V3D_GRP_LAUNCH_NS
#undef V3D_GRP_LAUNCH
  return launch_status(what);
}

// Plain 4-byte group gather through the same kernel: gather.hip routes there whenever the LDS path applies (it is
// the faster of the two forward kernels on MI355X: 5.7 against 5.0 TB/s on the k = 128 grouping of BASELINE config 2).
// Returns 1 when the shape is not eligible (nothing launched), else the launch status.
int group_fused_plain_u32(const void* src, const int64_t* idx, void* out, int B, int C, int N, int64_t L,
                          hipStream_t stream) {
  // rows under 8 KiB: gather.hip's kernel is as fast or faster (measured: N=1024,K=20 36 vs 42 us)
  if (L % 16 != 0 || L >= (int64_t)INT32_MAX || N < 2048) return 1;
  GroupArgs a{static_cast<const float*>(src), nullptr, nullptr, C, 0, 0, 0, N, (int)(L / 16), 16};
  if (!group_lds_ok(a, static_cast<float*>(out)) || B > kMaxGridYZ) return 1;
  return launch_group_fused("group_gather_forward", a, idx, static_cast<float*>(out), B, stream);
}

}  // namespace v3d

extern "C" {

int v3d_sa_group_f32(const float* q_points, const float* s_points, const float* s_feats, const int64_t* indices,
                     float* grouped, int B, int C, int N, int M, int K, int device, void* stream_) {
  using namespace v3d;
  const char* what = "sa_group";
  V3D_REQUIRE(B >= 0 && C >= 0 && N >= 0 && M >= 0 && K >= 0, what, "negative dimension");
  if (B == 0 || M == 0 || K == 0) return 0;
  V3D_REQUIRE(q_points && indices && grouped && (N == 0 || s_points) && (C == 0 || N == 0 || s_feats), what,
              "null pointer");
  DeviceGuard guard(device);
  if (guard.rc) return guard.rc;
  GroupArgs a{s_feats, s_points, q_points, C, 0, 3, 0, N, M, K};
  return launch_group_fused(what, a, indices, grouped, B, static_cast<hipStream_t>(stream_));
}

int v3d_edge_group_f32(const float* q_feats, const float* s_feats, const int64_t* indices, float* grouped, int B, int C,
                       int N, int M, int K, int device, void* stream_) {
  using namespace v3d;
  const char* what = "edge_group";
  V3D_REQUIRE(B >= 0 && C >= 0 && N >= 0 && M >= 0 && K >= 0, what, "negative dimension");
  if (B == 0 || M == 0 || K == 0 || C == 0) return 0;
  V3D_REQUIRE(q_feats && indices && grouped && (N == 0 || s_feats), what, "null pointer");
  DeviceGuard guard(device);
  if (guard.rc) return guard.rc;
  GroupArgs a{nullptr, s_feats, q_feats, 0, C, C, 1, N, M, K};
  return launch_group_fused(what, a, indices, grouped, B, static_cast<hipStream_t>(stream_));
}

}  // extern "C"
