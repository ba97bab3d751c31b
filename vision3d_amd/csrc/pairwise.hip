// pairwise.hip -- feature-space pairwise distance on fp32-input MFMA, row top-k selection, and the
// partial-top-k merge used by reference-set sharding.
// Semantics: ops/pairwise_distance.py:39-56 (expansion branch) and the call-site semantics of
// ops/knn.py:10-27 (pykeops Kmin_argKmin of the Euclidean norm; third-party, see DESIGN.md "parity unpinned").
//
// pairwise kernel: out[b,n,m] = clamp((x2[n] - 2*xy[n,m]) + y2[m], 0), xy on v_mfma_f32_32x32x2_f32 (exact fp32
// FMA chain in k order -- bf16/fp8 MFMA cannot hold the 1e-5 bar, MI355X has no TF32).  128x128 output tile per
// 256-thread workgroup, 64x64 per wave (2x2 MFMA tiles, 64 accumulator VGPRs), K staged through LDS in 32-deep
// steps, tiles stored k-major ([k][row]) so an operand fragment is one conflict-free ds_read_b32.  Row/column
// squared norms are accumulated from the same LDS tiles and applied with clamp / sqrt in the epilogue, so the
// N x M matrix is written exactly once (the reference makes 3-4 elementwise passes over it).
#include <math.h>

#include "common.h"

namespace v3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int PD_BM = 128, PD_BN = 128, PD_BK = 32, PD_THREADS = 256;
constexpr int PD_LD = PD_BM + 1;          // k-major LDS tiles [k][row], 129 floats per k (see pd_addr)
constexpr int PD_TILE = PD_BK * PD_LD;    // floats per operand tile (16.1 KiB)
static_assert(2 * PD_TILE >= 64 * PD_BN, "the operand tiles double as the 64 x 128 output staging area");

// A (128 rows x 32 k) slab of a row-major (rows x C) matrix, 4 float4 per thread.  Held in registers between the
// global load (issued before the MFMA loop of the previous slab) and the transposing LDS store (after it).
struct PdSlab {
  float4 v[4];
};
// Generic slab load: any C / alignment / edge, one guarded scalar load per element.
__device__ __forceinline__ void pd_load(PdSlab& s, const float* __restrict__ base, int rows, int C, int row0, int k0) {
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int f = it * PD_THREADS + threadIdx.x;
    const int r = f >> 3, k4 = (f & 7) * 4;
    const int row = row0 + r, k = k0 + k4;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < rows) {
      const float* p = base + (int64_t)row * C + k;
      if (k + 0 < C) t.x = p[0];
      if (k + 1 < C) t.y = p[1];
      if (k + 2 < C) t.z = p[2];
      if (k + 3 < C) t.w = p[3];
    }
    s.v[it] = t;
  }
}
// FAST slab load (C % 32 == 0, 16-byte aligned rows, at least 128 rows so that every tile is a whole one -- the last
// tile of a dimension is moved back to end at the matrix edge and recomputes a few rows of its neighbour): four
// buffer_load_dwordx4 whose address is a tile descriptor (scalar registers) + ONE per-thread byte offset + a scalar
// offset carrying the row group and k0.  Nothing per-load lives in vector registers: eight 64-bit pointers (or their
// spills) were what kept the k loop above 128 registers.
typedef unsigned pd_u32x4 __attribute__((ext_vector_type(4)));
struct PdTileSrc {
  __amdgpu_buffer_rsrc_t rsrc;  // the tile's 128 rows: base + row0 * C, 128 * C * 4 bytes
  unsigned voff;                // this thread's (row tid/8, k group tid%8) inside it, bytes
};
__device__ __forceinline__ PdTileSrc pd_tile_src(const float* __restrict__ base, int C, int row0) {
  PdTileSrc t;
  t.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + (int64_t)row0 * C), 0, PD_BM * C * 4, 0x00020000);
  t.voff = ((threadIdx.x >> 3) * (unsigned)C + (threadIdx.x & 7) * 4) * 4u;
  return t;
}
__device__ __forceinline__ void pd_load_fast(PdSlab& s, const PdTileSrc& src, int C, int k0) {
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const pd_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(src.rsrc, src.voff, (it * 32 * C + k0) * 4, 0);
    s.v[it] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
  }
}
// LDS element (k, r) lives at k*129 + r.  Fragment reads (fixed k, 32 consecutive r) are conflict-free under any pitch;
// the odd pitch spreads the transposing stores of a half-wave (4 rows x 8 k-groups, bank = (k + r) mod 32) over all 32
// banks, and -- unlike an XOR swizzle -- leaves every address of the k loop at a compile-time distance from one base
// register per operand (an XOR swizzle cost ~30 address registers, i.e. a workgroup per CU).
__device__ __forceinline__ int pd_addr(int k, int r) { return k * PD_LD + r; }
__device__ __forceinline__ void pd_store(const PdSlab& s, float* lds) {
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int f = it * PD_THREADS + threadIdx.x;
    const int r = f >> 3, k4 = (f & 7) * 4;
    lds[pd_addr(k4 + 0, r)] = s.v[it].x;
    lds[pd_addr(k4 + 1, r)] = s.v[it].y;
    lds[pd_addr(k4 + 2, r)] = s.v[it].z;
    lds[pd_addr(k4 + 3, r)] = s.v[it].w;
  }
}

// One 32-deep slab of the MFMA form: 16 steps of four v_mfma_f32_32x32x2_f32 (2x2 tiles of 32x32 per wave).  The
// operand fragments of step kk+2 are requested before the four MFMAs of step kk are issued, so the LDS latency runs
// under 256 cycles of matrix work instead of in front of it; the two values per step of this thread's norm row travel
// with them (the norm's FMA chain stays in k order).
__device__ __forceinline__ void pd_mfma_slab(const float* xs, const float* ys, f32x16 (&acc)[2][2], float& norm, int tid,
                                             int wr, int wc, int l31, int lh) {
  const float* nt = tid < 128 ? xs : ys;
  const int nr = tid & 127;
  float a[2][2], bb[2][2], nv[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) a[0][i] = xs[pd_addr(lh, wr * 64 + i * 32 + l31)];
#pragma unroll
  for (int j = 0; j < 2; ++j) bb[0][j] = ys[pd_addr(lh, wc * 64 + j * 32 + l31)];
#pragma unroll
  for (int u = 0; u < 2; ++u) nv[0][u] = nt[pd_addr(u, nr)];
#pragma unroll
  for (int kk = 0; kk < PD_BK; kk += 2) {
    const int cur = (kk >> 1) & 1, nxt = cur ^ 1;
    if (kk + 2 < PD_BK) {
#pragma unroll
      for (int i = 0; i < 2; ++i) a[nxt][i] = xs[pd_addr(kk + 2 + lh, wr * 64 + i * 32 + l31)];
#pragma unroll
      for (int j = 0; j < 2; ++j) bb[nxt][j] = ys[pd_addr(kk + 2 + lh, wc * 64 + j * 32 + l31)];
#pragma unroll
      for (int u = 0; u < 2; ++u) nv[nxt][u] = nt[pd_addr(kk + 2 + u, nr)];
    }
    __builtin_amdgcn_sched_barrier(0);  // hipcc otherwise sinks the reads back below the MFMAs they should overlap
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], bb[cur][j], acc[i][j], 0, 0, 0);
    norm = __builtin_fmaf(nv[cur][0], nv[cur][0], norm);
    norm = __builtin_fmaf(nv[cur][1], nv[cur][1], norm);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// STRICT: the exact difference form sum_c (x_c - y_c)^2 (pairwise_distance.py:28-37) on the VALU, accumulated in
// the same per-lane output layout as the MFMA path so both share staging and epilogue.
// Pipeline per 32-deep k-slab: the next slab's global loads are issued first and stay in flight under the 64 MFMAs of
// the current slab; the transposing LDS stores follow between two barriers.  Epilogue: distances go through LDS
// (reusing the operand tiles) so every lane stores 16 contiguous bytes and a wave 512-byte row segments.
template <bool STRICT, bool FAST>
__device__ __forceinline__ void pairwise_distance_tile(const float* __restrict__ x, const float* __restrict__ y,
                                                       float* __restrict__ out, int N, int M, int C,
                                                       int64_t y_batch_stride, int64_t out_ld, int flags, float eps,
                                                       int vec_out) {
  __shared__ __align__(16) float smem[2 * PD_TILE];  // xs | ys, later the 64 x 128 output staging area
  __shared__ float x2s[PD_BM];
  __shared__ float y2s[PD_BN];
  float* xs = smem;
  float* ys = smem + PD_TILE;
  const int b = blockIdx.z;
  // FAST: N, M >= 128 and the last tile of each dimension ends at the edge (overlapping its neighbour, which computes
  // the same values for the shared cells), so no tile has rows or columns to mask
  const int row0 = FAST ? min((int)blockIdx.y * PD_BM, N - PD_BM) : blockIdx.y * PD_BM;
  const int col0 = FAST ? min((int)blockIdx.x * PD_BN, M - PD_BN) : blockIdx.x * PD_BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;  // 2x2 waves, 64x64 each
  const float* xb = x + (int64_t)b * N * C;
  const float* yb = y + (int64_t)b * y_batch_stride;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float norm = 0.f;  // threads 0..127: x row `tid`; 128..255: y row `tid-128`

  const int l31 = lane & 31, lh = lane >> 5;
  PdSlab sx, sy;
  PdTileSrc tx, ty;
  if (FAST) {
    tx = pd_tile_src(xb, C, row0);
    ty = pd_tile_src(yb, C, col0);
    pd_load_fast(sx, tx, C, 0);
    pd_load_fast(sy, ty, C, 0);
  } else {
    pd_load(sx, xb, N, C, row0, 0);
    pd_load(sy, yb, M, C, col0, 0);
  }
  for (int k0 = 0; k0 < C; k0 += PD_BK) {
    __syncthreads();  // every wave is done reading the previous slab
    pd_store(sx, xs);
    pd_store(sy, ys);
    __syncthreads();
    if (k0 + PD_BK < C) {  // next slab: in flight while this one is multiplied
      if (FAST) {
        pd_load_fast(sx, tx, C, k0 + PD_BK);
        pd_load_fast(sy, ty, C, k0 + PD_BK);
      } else {
        pd_load(sx, xb, N, C, row0, k0 + PD_BK);
        pd_load(sy, yb, M, C, col0, k0 + PD_BK);
      }
    }
    if (STRICT) {
      for (int k = 0; k < PD_BK; ++k) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const float yv = ys[pd_addr(k, wc * 64 + j * 32 + l31)];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float xv = xs[pd_addr(k, wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh)];
              const float df = xv - yv;
              acc[i][j][r] = __builtin_fmaf(df, df, acc[i][j][r]);
            }
          }
        }
      }
      continue;
    }
    pd_mfma_slab(xs, ys, acc, norm, tid, wr, wc, l31, lh);
  }
  if (tid < 128) x2s[tid] = norm; else y2s[tid - 128] = norm;
  __syncthreads();  // norms visible; operand tiles free for the output staging

  const bool normalized = flags & V3D_PD_NORMALIZED;
  const bool do_sqrt = flags & V3D_PD_SQRT;
  float* ob = out + (int64_t)b * N * out_ld;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int cl = wc * 64 + j * 32 + l31;
      const int col = col0 + cl;
      const float y2 = y2s[cl];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int r32 = (r & 3) + 8 * (r >> 2) + 4 * lh;  // row inside the 32 x 32 MFMA tile
        const int rl = wr * 64 + i * 32 + r32;
        const int row = row0 + rl;
        const float xy = acc[i][j][r];
        float d;
        if (STRICT) {
          d = do_sqrt ? sqrtf(xy) : xy;  // :34-37 (norm, then optional square): no eps, no clamp
        } else {
          d = normalized ? (2.0f - 2.0f * xy) : ((x2s[rl] - 2.0f * xy) + y2);  // pairwise_distance.py:47,:51
          d = fmaxf(d, 0.0f);                                                   // :53
          if (do_sqrt) d = sqrtf(d + eps);                                      // :56
        }
        if (vec_out) {
          smem[(wr * 32 + r32) * PD_BN + cl] = d;  // staging row = 32*wr + r32 for this i
        } else {
          if (row < N && col < M) ob[(int64_t)row * out_ld + col] = d;
        }
      }
    }
    if (vec_out) {  // 64 rows x 128 columns staged: 8 float4 per thread, 512 contiguous bytes per row
      __syncthreads();
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int f = it * PD_THREADS + tid;
        const int srow = f >> 5, c4 = (f & 31) * 4;
        const int row = row0 + (srow >> 5) * 64 + i * 32 + (srow & 31);
        const int col = col0 + c4;
        if (row < N && col < M) {  // M % 4 == 0 here, so col < M covers the whole vector
          const float4 v = *reinterpret_cast<const float4*>(&smem[srow * PD_BN + c4]);
          *reinterpret_cast<float4*>(&ob[(int64_t)row * out_ld + col]) = v;
        }
      }
      __syncthreads();
    }
  }
}

// Occupancy is what the MFMA form is built around: a workgroup in its store / barrier phases or in its epilogue leaves
// the matrix pipe to the others on its CU.  Left to itself hipcc takes 184-208 registers (two workgroups per CU); the
// fast form fits 128 (64 of them accumulators) = four workgroups, the generic form 168 = three.  Measured: three
// instead of two, C=64 shapes 9-16 % faster and C=256 5-8 %.  The strict (VALU) form needs its 230 registers.
__global__ __launch_bounds__(PD_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void pairwise_distance_kernel(
    const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out, int N, int M, int C,
    int64_t y_batch_stride, int64_t out_ld, int flags, float eps, int vec_out) {
  pairwise_distance_tile<false, true>(x, y, out, N, M, C, y_batch_stride, out_ld, flags, eps, vec_out);
}
// The MFMA form for whole tiles and vector stores, as a sweep: at most four workgroups per CU, each taking tiles
// blockIdx.x, blockIdx.x + gridDim.x, ...  What the one-tile-per-workgroup form cannot do: (1) the first slab of the
// next tile is requested during the last slab of this one, so no tile starts by waiting for memory; (2) the epilogue's
// global stores are only ISSUED -- the workgroup goes on with the next tile while they drain, instead of ending and
// holding its CU slot until the last store is acknowledged.  (As compiled, the wait in front of the next tile's first
// LDS stores is the loop's vmcnt(<= 7), which the stores issued after those loads fall under too: the wave stalls
// there until its own stores are acknowledged, the other three workgroups of the CU carry on.  DESIGN.md section 9.)  With
// one tile per workgroup the workgroups of a CU run in lock-step (they start together and take equally long), so
// their k loops and their epilogues did not overlap: total time was the SUM of the k-loop time and the store time.
constexpr int PD_SWEEP_WGS_PER_CU = 4;  // 128 registers, 34 KB of LDS each (MI355X: 256 CUs -> 1024 workgroups)
static int pd_sweep_workgroups() {     // per current device; a partitioned GPU shows fewer CUs
  static int cached[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return PD_SWEEP_WGS_PER_CU * 256;
  if (cached[dev] == 0) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    cached[dev] = PD_SWEEP_WGS_PER_CU * cus;
  }
  return cached[dev];
}
struct PdTileAt {
  int b, row0, col0;
};
__device__ __forceinline__ PdTileAt pd_tile_at(int t, int tiles_x, int tiles_y, int N, int M) {
  PdTileAt a;
  const int per_b = tiles_x * tiles_y;
  a.b = t / per_b;
  const int rem = t - a.b * per_b;
  const int ty = rem / tiles_x;
  a.row0 = min(ty * PD_BM, N - PD_BM);  // the last tile of a dimension ends at the edge
  a.col0 = min((rem - ty * tiles_x) * PD_BN, M - PD_BN);
  return a;
}
template <bool SQRT>
__global__ __launch_bounds__(PD_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void pairwise_distance_sweep_kernel(
    const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out, int N, int M, int C,
    int64_t y_batch_stride, int64_t out_ld, int flags, float eps, int tiles_x, int tiles_y, int total) {
  __shared__ __align__(16) float smem[2 * PD_TILE];  // xs | ys, later the 64 x 128 output staging area
  __shared__ __align__(16) float x2s[PD_BM];
  __shared__ float y2s[PD_BN];
  float* xs = smem;
  float* ys = smem + PD_TILE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;  // 2x2 waves, 64x64 each
  const int l31 = lane & 31, lh = lane >> 5;
  const bool normalized = flags & V3D_PD_NORMALIZED;
  // every LDS address of the epilogue is one of these bases plus a compile-time offset (written as (row, column) index
  // arithmetic hipcc builds -- and keeps across the k loop -- a register per staged row)
  float* const stage = smem + (wr * 32 + 4 * lh) * PD_BN + wc * 64 + l31;
  const float* const x2p = x2s + wr * 64 + 4 * lh;
  const float* const y2p = y2s + wc * 64 + l31;
  const unsigned ovoff = ((tid >> 5) * (unsigned)out_ld + (tid & 31) * 4) * 4u;  // this thread's cell of an 8-row group

  int t = blockIdx.x;  // < total: the grid is min(total, four workgroups per CU)
  PdTileAt at = pd_tile_at(t, tiles_x, tiles_y, N, M);
  PdTileSrc tx = pd_tile_src(x + (int64_t)at.b * N * C, C, at.row0);
  PdTileSrc ty = pd_tile_src(y + (int64_t)at.b * y_batch_stride, C, at.col0);
  PdSlab sx, sy;
  pd_load_fast(sx, tx, C, 0);
  pd_load_fast(sy, ty, C, 0);
  for (;;) {
    const PdTileAt cur = at;
    const int t_next = t + (int)gridDim.x;
    const bool more = t_next < total;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float norm = 0.f;  // threads 0..127: x row `tid`; 128..255: y row `tid-128`
    for (int k0 = 0; k0 < C; k0 += PD_BK) {
      __syncthreads();  // every wave is done reading the previous slab (or the previous tile's staged output)
      pd_store(sx, xs);
      pd_store(sy, ys);
      __syncthreads();
      if (k0 + PD_BK < C) {  // next slab: in flight while this one is multiplied
        pd_load_fast(sx, tx, C, k0 + PD_BK);
        pd_load_fast(sy, ty, C, k0 + PD_BK);
      } else if (more) {  // first slab of the next tile: in flight under this tile's last slab and its epilogue
        at = pd_tile_at(t_next, tiles_x, tiles_y, N, M);
        tx = pd_tile_src(x + (int64_t)at.b * N * C, C, at.row0);
        ty = pd_tile_src(y + (int64_t)at.b * y_batch_stride, C, at.col0);
        pd_load_fast(sx, tx, C, 0);
        pd_load_fast(sy, ty, C, 0);
      }
      pd_mfma_slab(xs, ys, acc, norm, tid, wr, wc, l31, lh);
    }
    // normalized inputs: d = 2 - 2 x.y (pairwise_distance.py:47) = (2 - 2 x.y) + 0, the general form (:51) on norms 2, 0
    if (tid < 128) x2s[tid] = normalized ? 2.0f : norm; else y2s[tid - 128] = normalized ? 0.0f : norm;
    __syncthreads();  // norms visible; operand tiles free for the output staging

    // the tile's 128 output rows as a buffer: < 2^31 bytes (out_ld <= 2^21 on this path)
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
        out + ((int64_t)cur.b * N + cur.row0) * out_ld + cur.col0, 0, (PD_BM - 1) * (int)out_ld * 4 + PD_BN * 4, 0x00020000);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float y2v[2] = {y2p[0], y2p[32]};
#pragma unroll
      for (int q = 0; q < 4; ++q) {  // accumulator registers 4q..4q+3 = rows 8q + 4 lh + 0..3 of the MFMA tile
        const float4 x2v = *reinterpret_cast<const float4*>(x2p + i * 32 + 8 * q);
        const float x2[4] = {x2v.x, x2v.y, x2v.z, x2v.w};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            float d = (x2[u] - 2.0f * acc[i][j][4 * q + u]) + y2v[j];  // :51
            d = fmaxf(d, 0.0f);                                        // :53
            if (SQRT) d = sqrtf(d + eps);                              // :56
            stage[(8 * q + u) * PD_BN + j * 32] = d;  // staging row = 32*wr + (row in the MFMA tile) for this i
          }
        }
      }
      __syncthreads();
      // 64 staged rows x 128 columns: 8 float4 per thread, a wave-instruction stores two 512-byte row segments.
      // Buffer stores: tile descriptor + one per-thread offset + a scalar row offset, like the loads.
      const float* sp = smem + (tid >> 5) * PD_BN + (tid & 31) * 4;
#pragma unroll
      for (int it = 0; it < 8; ++it) {  // staged row it*8 + tid/32 = tile row (it/4)*64 + i*32 + (it%4)*8 + tid/32
        const float4 v = *reinterpret_cast<const float4*>(sp + it * 8 * PD_BN);
        const pd_u32x4 bits = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
        __builtin_amdgcn_raw_buffer_store_b128(bits, orsrc, ovoff,
                                               (((it >> 2) * 64 + i * 32 + (it & 3) * 8) * (int)out_ld) * 4, 0);
      }
      if (i == 0) __syncthreads();  // after i = 1 the next tile's first barrier stands between these reads and new writes
    }
    if (!more) break;
    t = t_next;
  }
}
__global__ __launch_bounds__(PD_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) void pairwise_distance_generic_kernel(
    const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out, int N, int M, int C,
    int64_t y_batch_stride, int64_t out_ld, int flags, float eps, int vec_out) {
  pairwise_distance_tile<false, false>(x, y, out, N, M, C, y_batch_stride, out_ld, flags, eps, vec_out);
}
template <bool FAST>
__global__ __launch_bounds__(PD_THREADS) void pairwise_distance_strict_kernel(
    const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out, int N, int M, int C,
    int64_t y_batch_stride, int64_t out_ld, int flags, float eps, int vec_out) {
  pairwise_distance_tile<true, FAST>(x, y, out, N, M, C, y_batch_stride, out_ld, flags, eps, vec_out);
}

static int launch_pairwise(const float* x, const float* y, float* out, int batch, int N, int M, int C,
                           int64_t y_batch_stride, int64_t out_ld, int flags, float eps, hipStream_t stream) {
  const bool fast = C > 0 && C <= (1 << 20) && (C % PD_BK == 0) && N >= PD_BM && M >= PD_BN && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0) && N > 0 && M > 0;
  const int vec_out = (M % 4 == 0) && (out_ld % 4 == 0) && ((uintptr_t)out % 16 == 0);
  dim3 grid((unsigned)ceil_div(M, PD_BN), (unsigned)ceil_div(N, PD_BM), (unsigned)batch);
#define V3D_LAUNCH_PD(K_)                                                                                         \
  hipLaunchKernelGGL(K_, grid, dim3(PD_THREADS), 0, stream, x, y, out, N, M, C, y_batch_stride, out_ld, flags, eps, \
                     vec_out)
  // out_ld <= 2^21: the 128 output rows of a tile (127 * out_ld * 4 + 512 bytes) and every row offset stay below 2^31
  if (fast && vec_out && out_ld <= (1 << 21) && !(flags & V3D_PD_STRICT)) {
    const int64_t tiles_x = ceil_div(M, PD_BN), tiles_y = ceil_div(N, PD_BM), total = tiles_x * tiles_y * batch;
    if (total <= (int64_t)INT32_MAX) {
      const dim3 sweep((unsigned)std::min<int64_t>(total, pd_sweep_workgroups()));
      if (flags & V3D_PD_SQRT)
        hipLaunchKernelGGL(pairwise_distance_sweep_kernel<true>, sweep, dim3(PD_THREADS), 0, stream, x, y, out, N, M, C,
                           y_batch_stride, out_ld, flags, eps, (int)tiles_x, (int)tiles_y, (int)total);
      else
        hipLaunchKernelGGL(pairwise_distance_sweep_kernel<false>, sweep, dim3(PD_THREADS), 0, stream, x, y, out, N, M, C,
                           y_batch_stride, out_ld, flags, eps, (int)tiles_x, (int)tiles_y, (int)total);
      return launch_status("pairwise_distance");
    }
  }
  if (flags & V3D_PD_STRICT) {
    if (fast) V3D_LAUNCH_PD(pairwise_distance_strict_kernel<true>); else V3D_LAUNCH_PD(pairwise_distance_strict_kernel<false>);
  } else {
    if (fast) V3D_LAUNCH_PD(pairwise_distance_kernel); else V3D_LAUNCH_PD(pairwise_distance_generic_kernel);
  }
#undef V3D_LAUNCH_PD
  return launch_status("pairwise_distance");
}

// ------------------------------------------------------------------------------------------------------
// row top-k over a materialised distance chunk: one wave per row, the running list (ascending by
// (distance, index), k <= 64) lives one entry per lane; candidates are inserted with a whole-wave shift.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_shr1(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ int wave_shr1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, false); }

constexpr int kTopkCand = 192;  // candidates ranked in LDS per row
constexpr int kTopkAhead = 8;   // 64-column groups requested together on the insertion path

__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ D, int64_t rows, int ncols,
                                                        int64_t ld, int col_base, int k, int first_chunk,
                                                        float* __restrict__ run_d, int* __restrict__ run_i) {
  // candidates as (distance bits << 32 | column) words; + 4: the ranking's halves read up to three sentinels past the end
  __shared__ __align__(16) unsigned long long cand_s[4][kTopkCand + 4];
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float ld_ = INFINITY;
  int li = -1;
  if (!first_chunk && lane < k) {
    ld_ = run_d[row * k + lane];
    li = run_i[row * k + lane];
  }
  const float* drow = D + row * ld;
  // Pre-pass (first chunk of a row only, when the list is still empty): every lane takes the minimum of its own
  // columns; the k-th smallest of those 64 minima is an upper bound of the row's k-th smallest distance, usually a
  // tight one, and caps the insertion threshold from the start (an empty list would otherwise admit every column
  // of the first chunks).  Costs one extra read of the row (L2 / Infinity-Cache resident).
  // Both passes read the row in tiles of 1024 columns, four 16-byte loads per lane issued together (a lane owns 4
  // consecutive columns of each 256-column slab; rows are 16-byte aligned and padded to a multiple of 128 columns).
  float cap = INFINITY;
  const int ntile = (ncols + 1023) >> 10;
  float4 keep[4];  // a row of one tile (<= 1024 columns: config 3, every chunk of a wide search) stays in registers
                   // between the two passes: the selection is bound by its reads, and this halves them
#pragma unroll
  for (int u = 0; u < 4; ++u) keep[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (first_chunk && ncols >= 256) {
    float lmin = INFINITY;
    for (int t = 0; t < ntile; ++t) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = (t << 10) + ((u << 6) + lane) * 4;
        v[u] = *reinterpret_cast<const float4*>(drow + (c < ncols ? c : 0));
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = (t << 10) + ((u << 6) + lane) * 4;
        float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          e[q] = c + q < ncols ? e[q] : INFINITY;  // columns past the end never qualify, here or below
          lmin = fminf(lmin, e[q]);
        }
        keep[u] = make_float4(e[0], e[1], e[2], e[3]);  // (the last tile's; used only when it is the only one)
      }
    }
    // k-th smallest of the 64 minima (distances are >= 0, so the IEEE pattern orders like an integer): bit-wise binary
    // search, one compare + ballot + popcount per bit
    const unsigned lbits = __float_as_uint(lmin);
    unsigned prefix = 0;
    for (int bit = 30; bit >= 15; --bit) {  // the low 15 bits are not searched: all ones keeps it an upper bound, at most
      const unsigned trial = prefix | (1u << bit);  // 2^-9 looser (about one more candidate per row at config 3)
      if (__builtin_popcountll(__ballot(lbits < trial)) <= k - 1) prefix = trial;
    }
    prefix |= 0x7fffu;
    const float kth = __uint_as_float(prefix);
    // strict '<' below must still admit values equal to the bound: step to the next float (kth >= 0)
    cap = kth < INFINITY ? __uint_as_float(__float_as_uint(kth) + 1u) : INFINITY;
  }
  // Whole row in one chunk and a finite bound: the columns below the bound (a few more than k, usually) are collected
  // in LDS and ranked against each other in one go -- every lane ranks one candidate by (distance, column) -- instead
  // of one whole-wave shift per candidate.  More than kTopkCand candidates: the insertion path below takes the row.
  if (first_chunk && cap < INFINITY) {
    unsigned long long* cand = cand_s[threadIdx.x >> 6];
    int count = 0;
    for (int t = 0; t < ntile && count <= kTopkCand; ++t) {
      float4 v[4];
      if (ntile == 1) {  // wave-uniform
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = keep[u];
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int c = (t << 10) + ((u << 6) + lane) * 4;
          v[u] = *reinterpret_cast<const float4*>(drow + (c < ncols ? c : 0));
          if (c + 3 >= ncols) {  // the row's last columns (rows are padded to 128 columns, not to a tile)
            v[u].x = c < ncols ? v[u].x : INFINITY;
            v[u].y = c + 1 < ncols ? v[u].y : INFINITY;
            v[u].z = c + 2 < ncols ? v[u].z : INFINITY;
            v[u].w = c + 3 < ncols ? v[u].w : INFINITY;
          }
        }
      }
      // The selection is bound by instruction issue, and this loop was half of it (26 instructions per element as
      // compiled from compare / mask / popcount / conditional store): a hit's slot is the running count + the hits in
      // the lower lanes (two `v_mbcnt`, the count rides in as the addend), clamped to the spare slot behind the list.
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = (t << 10) + ((u << 6) + lane) * 4;
        const float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool in = e[q] < cap;
          const unsigned long long m = __ballot(in);
          const unsigned slot = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32),
                                                          __builtin_amdgcn_mbcnt_lo((unsigned)m, (unsigned)count));
          // (distance, column) as one 64-bit word that orders by (distance, column); e + 0 turns -0 into +0
          if (in)
            cand[min(slot, (unsigned)kTopkCand)] =
                ((unsigned long long)__float_as_uint(e[q] + 0.0f) << 32) | (unsigned)(col_base + c + q);
          count += __builtin_popcountll(m);
        }
      }
    }
    if (count <= kTopkCand) {  // count >= k: the bound is attained by k distinct columns
      // Every candidate's rank = the candidates below it (distances are >= +0: the bit pattern orders like the value;
      // columns differ, so the order is strict).  32 candidates per pass: lanes l and l + 32 hold the same candidate and
      // count over one half of the list each (a wave-wide pass left half of the lanes idle at the ~30 candidates a row
      // has), two words per LDS read; the halves may run past the end into sentinels.
      if (lane < 4) cand[count + lane] = ~0ull;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int half = lane >> 5;
      const int mid = ((count + 3) >> 2) << 1;  // even (16-byte reads), 2 * mid >= count
      typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
      const u64x2* part = reinterpret_cast<const u64x2*>(cand + half * mid);
      for (int j0 = 0; j0 < count; j0 += 32) {
        const int j = j0 + (lane & 31);
        const unsigned long long me = cand[j < count ? j : 0];
        int rank = 0;
        for (int i = 0; i < (mid >> 1); ++i) {
          const u64x2 two = part[i];
          rank += two.x < me ? 1 : 0;
          rank += two.y < me ? 1 : 0;
        }
        rank += __shfl_xor(rank, 32, 64);
        if (half == 0 && j < count && rank < k) {
          run_d[row * k + rank] = __uint_as_float((unsigned)(me >> 32));
          run_i[row * k + rank] = (int)(unsigned)me;
        }
      }
      return;
    }
  }
  float tau = fminf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(ld_), k - 1)), cap);
  for (int c00 = 0; c00 < ncols; c00 += 64 * kTopkAhead) {
    // kTopkAhead loads in flight per lane (the row is read once, front to back: one load per 64 columns and its wait
    // would leave the pass latency-bound); the 64-column groups are then taken in ascending order, as the tie rule of
    // the insertion needs
    float dv[kTopkAhead];
#pragma unroll
    for (int u = 0; u < kTopkAhead; ++u) {
      const int c = c00 + u * 64 + lane;
      dv[u] = drow[c < ncols ? c : 0];
    }
#pragma unroll
    for (int u = 0; u < kTopkAhead; ++u) {
    const int c0 = c00 + u * 64;
    const int c = c0 + lane;
    const float d = c < ncols ? dv[u] : INFINITY;
    unsigned long long m = __ballot(d < tau);
    while (m) {
      const int l = __builtin_ffsll((long long)m) - 1;
      const float cd = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), l));
      const int ci = col_base + c0 + l;
      // entries <= cd stay (equal distances: earlier index first); the rest shift one lane up
      const unsigned long long keep = __ballot(ld_ <= cd);
      const int pos = __builtin_popcountll(keep);
      const float sd = wave_shr1(ld_);
      const int si = wave_shr1(li);
      if (lane > pos) {
        ld_ = sd;
        li = si;
      } else if (lane == pos) {
        ld_ = cd;
        li = ci;
      }
      tau = fminf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(ld_), k - 1)), cap);
      m = __ballot(d < tau) & ~((2ull << l) - 1ull);
    }
    }
  }
  if (lane < k) {
    run_d[row * k + lane] = ld_;
    run_i[row * k + lane] = li;
  }
}

#define V3D_DPP_ADD(v, CTRL_TEXT) asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 " CTRL_TEXT : "+v"(v))
__device__ __forceinline__ float wave_sum_f32(float v) {  // sum over the 64 lanes, wave-uniform result
  V3D_DPP_ADD(v, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
  V3D_DPP_ADD(v, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");
  V3D_DPP_ADD(v, "row_half_mirror row_mask:0xf bank_mask:0xf");
  V3D_DPP_ADD(v, "row_mirror row_mask:0xf bank_mask:0xf");
  V3D_DPP_ADD(v, "row_bcast:15 row_mask:0xa bank_mask:0xf");
  V3D_DPP_ADD(v, "row_bcast:31 row_mask:0xc bank_mask:0xf");
  asm volatile("s_nop 1" ::: "memory");
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ float row16_sum_f32(float v) {  // sum over each row of 16 lanes, in every lane of the row
  V3D_DPP_ADD(v, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
  V3D_DPP_ADD(v, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");
  V3D_DPP_ADD(v, "row_half_mirror row_mask:0xf bank_mask:0xf");
  V3D_DPP_ADD(v, "row_mirror row_mask:0xf bank_mask:0xf");
  return v;
}

// The same for four values at once: the four chains alternate, so each DPP read finds its operand written three
// instructions earlier and only the first needs the wait states (`s_nop 1`) that row16_sum_f32 pays per step.
__device__ __forceinline__ void row16_sum4_f32(float (&v)[4]) {
#define V3D_DPP_STEP4(CTRL_TEXT)                                                  \
  "v_add_f32_dpp %0, %0, %0 " CTRL_TEXT " row_mask:0xf bank_mask:0xf\n\t"         \
  "v_add_f32_dpp %1, %1, %1 " CTRL_TEXT " row_mask:0xf bank_mask:0xf\n\t"         \
  "v_add_f32_dpp %2, %2, %2 " CTRL_TEXT " row_mask:0xf bank_mask:0xf\n\t"         \
  "v_add_f32_dpp %3, %3, %3 " CTRL_TEXT " row_mask:0xf bank_mask:0xf\n\t"
  asm volatile("s_nop 1\n\t" V3D_DPP_STEP4("quad_perm:[1,0,3,2]") V3D_DPP_STEP4("quad_perm:[2,3,0,1]")
                   V3D_DPP_STEP4("row_half_mirror") V3D_DPP_STEP4("row_mirror") "s_nop 1"
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
#undef V3D_DPP_STEP4
}

// Candidates kept per row for k requested neighbours.  The lists are selected on the expansion-form distance
// (|x|^2 - 2xy + |y|^2 on the MFMA), whose rounding error (~1e-6 of |x|^2 + |y|^2) can swap neighbours whose exact
// distances are closer than that; the finalize pass re-ranks on the exact difference form.  Keeping a few spare
// candidates makes the k that come out the k best by EXACT (distance, index) unless more than `spare` columns crowd
// inside the rounding error of the k-th -- which is also what makes a reference-set-sharded search (partial lists per
// shard, merged) return exactly the single-device lists.
__host__ __device__ inline int knn_feat_select_k(int k) {
  const int spare = k / 4 > 4 ? k / 4 : 4;
  if (k > 64) return k + spare;  // the workgroup-per-row kernels: no lane-list limit
  return k + spare < 64 ? k + spare : 64;
}

// Final pass: recompute the k selected distances exactly as sqrt(sum_c (x_c - y_c)^2) (what pykeops evaluates,
// ops/knn.py:25), re-rank by (distance, index) and write fp32 distances + int64 (offset) indices.
// One wave per query row; for each selected neighbour the 64 lanes split the channels (coalesced row reads) and
// reduce with a DPP butterfly, lane j keeps neighbour j's distance.
__global__ __launch_bounds__(256) void knn_feat_finalize_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                const float* __restrict__ run_d,
                                                                const int* __restrict__ run_i, int batch, int N, int M,
                                                                int C, int k, int kout, int64_t index_offset, int vec4,
                                                                float* __restrict__ out_d, int64_t* __restrict__ out_i) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform
  if (row >= (int64_t)batch * N) return;
  const int b = (int)(row / N);
  const int idx = lane < k ? run_i[row * k + lane] : -1;
  float dist = INFINITY;
  const float* xr = x + row * C;
  if (vec4) {
    // a row of 16 lanes owns one neighbour and reads its features as 16-byte pieces; the four DPP steps that stay
    // inside a row of 16 lanes leave every lane of the row with its neighbour's sum.  Four such groups of four
    // neighbours go together (16 feature rows requested at once).  The kernel is bound by instruction issue (~900 per
    // query row at config 3 before, most of it bookkeeping): neighbour rows are addressed as a scalar cloud base + a
    // 32-bit offset (vec4 implies M * C < 2^31), the DPP chains alternate, the square root is taken once at the end.
    const int l16 = lane & 15, grp = lane >> 4;
    const float* yb = y + (int64_t)b * M * C;  // scalar registers
    for (int j0 = 0; j0 < k; j0 += 16) {
      unsigned yo[4];
      bool live[4];
      float part[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int jj = j0 + 4 * g + grp;
        const int ij = __shfl(idx, min(jj, 63), 64);
        live[g] = (jj < k) && ij >= 0;
        yo[g] = (unsigned)(live[g] ? ij : 0) * (unsigned)C + (unsigned)(l16 * 4);
        part[g] = 0.f;
      }
      for (int c = l16 * 4, c0 = 0; c < C; c += 64, c0 += 64) {
        const float4 xv = *reinterpret_cast<const float4*>(xr + c);
        float4 yv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) yv[g] = *reinterpret_cast<const float4*>(yb + (yo[g] + (unsigned)c0));
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float d0 = xv.x - yv[g].x, d1 = xv.y - yv[g].y, d2 = xv.z - yv[g].z, d3 = xv.w - yv[g].w;
          part[g] = __builtin_fmaf(d0, d0, part[g]);
          part[g] = __builtin_fmaf(d1, d1, part[g]);
          part[g] = __builtin_fmaf(d2, d2, part[g]);
          part[g] = __builtin_fmaf(d3, d3, part[g]);
        }
      }
      row16_sum4_f32(part);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float mine = __shfl(live[g] ? part[g] : INFINITY, (lane & 3) << 4, 64);  // neighbour j0 + 4g + (lane & 3)
        if ((lane >> 2) == ((j0 + 4 * g) >> 2)) dist = mine;                            // lane L keeps neighbour L's sum
      }
    }
    dist = sqrtf(dist);  // one square root per lane, after the sweep (sqrt(inf) = inf for the lanes without a neighbour)
    if (idx < 0) dist = INFINITY;
  } else {
    for (int j = 0; j < k; ++j) {
      const int ij = __builtin_amdgcn_readlane(idx, j);  // wave-uniform
      if (ij < 0) continue;
      const float* yr = y + ((int64_t)b * M + ij) * C;
      float part = 0.f;
      for (int c = lane; c < C; c += 64) {
        const float df = xr[c] - yr[c];
        part = __builtin_fmaf(df, df, part);
      }
      const float tot = wave_sum_f32(part);
      if (lane == j) dist = sqrtf(tot);
    }
  }
  // rank sort among the k entries (k <= 64): position = number of entries that precede mine.  (distance, index) as one
  // 64-bit key -- distances are sqrt results, >= +0, so the bit pattern orders like the value; unfilled entries and the
  // lanes >= k carry the largest key -- makes "precedes" a single compare per entry.  The keys go through LDS: 32 entries
  // per pass, lanes l and l + 32 hold the same entry and count over one half of the list each, two keys per read (a
  // `v_readlane` pair per key and a wave-wide pass, half of its lanes idle at k = 25, were a third of this kernel).
  __shared__ __align__(16) unsigned long long keys_s[4][64];
  unsigned long long* keys = keys_s[threadIdx.x >> 6];
  const unsigned long long key =
      idx < 0 ? ~0ull : (((unsigned long long)__float_as_uint(dist) << 32) | (unsigned long long)(unsigned)idx);
  keys[lane] = key;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  int rank = 0;
  {
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    const int half = lane >> 5;
    const int mid = ((k + 3) >> 2) << 1;  // even (16-byte reads), k <= 2 * mid <= 64: past k the keys are the largest
    const u64x2* part = reinterpret_cast<const u64x2*>(keys + half * mid);
    int mine_rank = 0;
    for (int j0 = 0; j0 < k; j0 += 32) {
      const unsigned long long me = keys[j0 + (lane & 31)];
      int r = 0;
      for (int i = 0; i < (mid >> 1); ++i) {
        const u64x2 two = part[i];
        r += two.x < me ? 1 : 0;
        r += two.y < me ? 1 : 0;
      }
      r += __shfl_xor(r, 32, 64);
      // entry j0 + (lane & 31) is lane `that`'s: hand the rank to its owner (a no-op move for the first pass)
      const int got = __shfl(r, lane & 31, 64);
      if ((lane >> 5) == (j0 >> 5)) mine_rank = got;
    }
    rank = mine_rank;
  }
  if (lane < k) {
    // unfilled entries (M < k) all tie: keep them behind the real ones in lane order
    if (idx < 0) {
      const unsigned long long empties = __ballot(idx < 0 && lane < k);
      const int filled = k - __builtin_popcountll(empties);
      rank = filled + __builtin_popcountll(empties & ((1ull << lane) - 1ull));
    }
    if (rank < kout) {  // k - kout spare candidates only guard the selection against the expansion form's rounding
      out_d[row * kout + rank] = dist;
      out_i[row * kout + rank] = idx >= 0 ? (int64_t)idx + index_offset : (int64_t)-1;
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// k > 64 (pykeops' Kmin_argKmin has no limit, ops/knn.py:26): one WORKGROUP per row, selection by value instead of a
// lane list.  The pool of a pass is the row's running list (k slots from earlier chunks) followed by this chunk's
// columns; an element's key is (distance bits, index), both unsigned -- distances are >= +0, so the bit pattern orders
// like the value; unfilled slots carry (0xffffffff, 0x80000000 + slot) and sort behind everything.  Two bit-wise binary
// searches (each step: count the keys below a trial, block-wide) find the k-th smallest distance and, among the
// columns that tie with it, the index that completes the k; the k elements at or below that pair are gathered in LDS
// and ranked against each other.  The pool is re-read from memory (L2 / Infinity Cache) on every step: this path
// trades speed for having no limit but LDS (kKnnLargeMaxK entries).
// ------------------------------------------------------------------------------------------------------
constexpr int kKnnLargeMaxK = 2560;  // candidates per row: 2048 requested + a quarter spare
static_assert(V3D_KNN_FEAT_MAX_K + V3D_KNN_FEAT_MAX_K / 4 <= kKnnLargeMaxK, "the selection's LDS list holds k + k/4 entries");
constexpr unsigned kEmptyKey = 0xffffffffu, kEmptyIdx = 0x80000000u;

__device__ __forceinline__ int block_sum_256(int v, int* red) {  // every thread of a 256-thread workgroup calls it
  for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void topk_rows_large_kernel(const float* __restrict__ D, int ncols, int64_t ld,
                                                              int col_base, int k, int first_chunk,
                                                              float* __restrict__ run_d, int* __restrict__ run_i) {
  extern __shared__ uint2 sel[];  // k gathered (distance bits, index)
  __shared__ int red[4];
  __shared__ int fill;
  const int tid = threadIdx.x;
  const int64_t row = blockIdx.x;
  const float* drow = D + row * ld;
  float* rd = run_d + row * k;
  int* ri = run_i + row * k;
  const int kprev = first_chunk ? 0 : k;
  const int pool = kprev + ncols;
  const int need = pool < k ? pool : k;
  auto element = [&](int e, unsigned& key, unsigned& idx) {
    if (e < kprev) {
      const int i = ri[e];
      const float d = rd[e];
      key = i < 0 ? kEmptyKey : (d == 0.f ? 0u : __float_as_uint(d));
      idx = i < 0 ? kEmptyIdx + (unsigned)e : (unsigned)i;
    } else {
      const float d = drow[e - kprev];
      key = d == 0.f ? 0u : __float_as_uint(d);
      idx = (unsigned)(col_base + (e - kprev));
    }
  };
  unsigned T = 0, I = 0xffffffffu;
  if (need > 0 && need < pool) {
    for (int bit = 31; bit >= 0; --bit) {  // T = need-th smallest distance pattern
      const unsigned trial = T | (1u << bit);
      int c = 0;
      for (int e = tid; e < pool; e += 256) {
        unsigned key, idx;
        element(e, key, idx);
        c += key < trial ? 1 : 0;
      }
      if (block_sum_256(c, red) <= need - 1) T = trial;
    }
    int below = 0, equal = 0;
    for (int e = tid; e < pool; e += 256) {
      unsigned key, idx;
      element(e, key, idx);
      below += key < T ? 1 : 0;
      equal += key == T ? 1 : 0;
    }
    below = block_sum_256(below, red);
    equal = block_sum_256(equal, red);
    const int ties = need - below;  // >= 1 columns at distance T belong to the k; lowest indices first
    if (ties < equal) {
      I = 0;
      for (int bit = 31; bit >= 0; --bit) {  // I = ties-th smallest index among the columns at distance T
        const unsigned trial = I | (1u << bit);
        int c = 0;
        for (int e = tid; e < pool; e += 256) {
          unsigned key, idx;
          element(e, key, idx);
          c += (key == T && idx < trial) ? 1 : 0;
        }
        if (block_sum_256(c, red) <= ties - 1) I = trial;
      }
    }
  } else {
    T = 0xffffffffu;  // the whole pool
  }
  if (tid == 0) fill = 0;
  __syncthreads();
  for (int e = tid; e < pool; e += 256) {
    unsigned key, idx;
    element(e, key, idx);
    if (key < T || (key == T && idx <= I)) {
      const int slot = atomicAdd(&fill, 1);
      if (slot < k) sel[slot] = make_uint2(key, idx);
    }
  }
  __syncthreads();  // the running list has been read for the last time: it may now be overwritten
  const int got = fill < k ? fill : k;  // == need
  for (int j = tid; j < k; j += 256) {
    if (j >= got) {
      rd[j] = INFINITY;
      ri[j] = -1;
    }
  }
  for (int j = tid; j < got; j += 256) {
    const uint2 me = sel[j];
    int rank = 0;
    for (int i = 0; i < got; ++i) {
      const uint2 o = sel[i];  // same address in every lane: LDS broadcast
      rank += (o.x < me.x || (o.x == me.x && o.y < me.y)) ? 1 : 0;
    }
    const bool empty = me.y >= kEmptyIdx;
    rd[rank] = empty ? INFINITY : __uint_as_float(me.x);
    ri[rank] = empty ? -1 : (int)me.y;
  }
}

// Final pass for k > 64: a workgroup per row, a wave per candidate for the exact distance, then a rank sort in LDS.
__global__ __launch_bounds__(256) void knn_feat_finalize_large_kernel(const float* __restrict__ x,
                                                                      const float* __restrict__ y,
                                                                      const int* __restrict__ run_i, int N, int M, int C,
                                                                      int k, int kout, int64_t index_offset,
                                                                      float* __restrict__ out_d,
                                                                      int64_t* __restrict__ out_i) {
  extern __shared__ uint2 sel[];  // (exact distance bits, index)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row = blockIdx.x;
  const int b = (int)(row / N);
  const float* xr = x + row * C;
  for (int j = wave; j < k; j += 4) {
    const int ij = run_i[row * k + j];  // wave-uniform
    float dist = INFINITY;
    if (ij >= 0) {
      const float* yr = y + ((int64_t)b * M + ij) * C;
      float part = 0.f;
      for (int c = lane; c < C; c += 64) {
        const float df = xr[c] - yr[c];
        part = __builtin_fmaf(df, df, part);
      }
      dist = sqrtf(wave_sum_f32(part));
    }
    if (lane == 0) sel[j] = make_uint2(__float_as_uint(dist), ij < 0 ? kEmptyIdx + (unsigned)j : (unsigned)ij);
  }
  __syncthreads();
  for (int j = tid; j < k; j += 256) {
    const uint2 me = sel[j];
    const float md = __uint_as_float(me.x);
    int rank = 0;
    for (int i = 0; i < k; ++i) {
      const uint2 o = sel[i];
      const float od = __uint_as_float(o.x);
      const bool o_real = o.y < kEmptyIdx, m_real = me.y < kEmptyIdx;
      bool before;
      if (o_real != m_real) before = o_real;
      else before = od < md || (od == md && o.y < me.y);
      rank += before ? 1 : 0;
    }
    if (rank < kout) {
      out_d[row * kout + rank] = md;
      out_i[row * kout + rank] = me.y < kEmptyIdx ? (int64_t)me.y + index_offset : (int64_t)-1;
    }
  }
}

// Merge of G partial lists per row: parts (G, rows, k) sorted ascending -> (rows, k) by (distance, index).
// One wave per row; lane j repeatedly... k <= 64 and G*k candidates: rank every candidate against all others.
__global__ __launch_bounds__(256) void topk_merge_kernel(const float* __restrict__ pd, const int64_t* __restrict__ pi,
                                                         float* __restrict__ od, int64_t* __restrict__ oi, int G,
                                                         int64_t rows, int k) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int total = G * k;
  for (int c = lane; c < total; c += 64) {
    const int g = c / k, j = c % k;
    const float d = pd[((int64_t)g * rows + row) * k + j];
    const int64_t id = pi[((int64_t)g * rows + row) * k + j];
    int rank = 0;
    for (int o = 0; o < total; ++o) {
      const int go = o / k, jo = o % k;
      const float d2 = pd[((int64_t)go * rows + row) * k + jo];
      const int64_t i2 = pi[((int64_t)go * rows + row) * k + jo];
      // invalid entries (index < 0) sort last; full order (distance, index, slot) so ranks are distinct
      const bool v1 = id >= 0, v2 = i2 >= 0;
      bool before;
      if (v1 != v2) before = v2;
      else if (d2 != d) before = d2 < d;
      else if (i2 != id) before = i2 < id;
      else before = o < c;
      rank += before ? 1 : 0;
    }
    if (rank < k) {
      od[row * k + rank] = d;
      oi[row * k + rank] = id;
    }
  }
}

struct KnnFeatWorkspace {
  float* chunk;  // (batch*N, MC)
  float* run_d;  // (batch*N, k)
  int* run_i;    // (batch*N, k)
  int mc;
  size_t bytes;
};

static KnnFeatWorkspace carve_knn_ws(void* base, int batch, int N, int M, int k) {
  KnnFeatWorkspace w;
  const int64_t rows = (int64_t)batch * N;
  // chunk of columns sized to stay inside the 256 MiB Infinity Cache (<= 128 MiB), multiple of the 128 tile -- but not
  // under 1024 columns while that fits 1 GiB: with very many rows a 128 MiB chunk is only a few tiles wide, and the
  // distance launches then run at 0.50 instead of 0.57 of the MFMA rate (100,000 rows: 9.4 -> 8.4 ms at 1280 columns,
  // although the selection pass then reads the chunk from HBM).  Wider chunks for FEW rows measured 3-4 % slower.
  int64_t mc = (128ll << 20) / (rows * 4);
  mc = mc / 128 * 128;
  if (mc < 1024) mc = std::max<int64_t>(mc, std::min<int64_t>(1024, (1ll << 30) / (rows * 4) / 128 * 128));
  if (mc < 128) mc = 128;
  const int64_t m_pad = ceil_div(M, 128) * 128;
  if (mc > m_pad) mc = m_pad;
  w.mc = (int)mc;
  size_t off = 0;
  auto take = [&](size_t n) {
    void* p = base ? static_cast<char*>(base) + off : nullptr;
    off += align_up(n, 256);
    return p;
  };
  w.chunk = static_cast<float*>(take(sizeof(float) * (size_t)rows * (size_t)mc));
  w.run_d = static_cast<float*>(take(sizeof(float) * (size_t)rows * (size_t)k));
  w.run_i = static_cast<int*>(take(sizeof(int) * (size_t)rows * (size_t)k));
  w.bytes = off;
  return w;
}

}  // namespace v3d

extern "C" {

int v3d_pairwise_distance_f32(const float* x, const float* y, float* out, int batch, int N, int M, int C, int flags,
                              float eps, int device, void* stream_) {
  using namespace v3d;
  const char* what = "pairwise_distance";
  V3D_REQUIRE(batch >= 0 && N >= 0 && M >= 0 && C >= 0, what, "negative dimension");
  if (batch == 0 || N == 0 || M == 0) return 0;
  V3D_REQUIRE(x && y && out, what, "null pointer");
  if (batch > kMaxGridYZ || ceil_div(N, PD_BM) > kMaxGridYZ) {
    set_last_error("%s: batch <= 65535 and N <= 8388480 required", what);
    return V3D_ERR_SIZE_LIMIT;
  }
  DeviceGuard guard(device);
  if (guard.rc) return guard.rc;
  return launch_pairwise(x, y, out, batch, N, M, C, (int64_t)M * C, (int64_t)M, flags, eps,
                         static_cast<hipStream_t>(stream_));
}

size_t v3d_knn_feat_workspace_bytes(int batch, int N, int M, int C, int k) {
  (void)C;
  if (batch <= 0 || N <= 0 || M <= 0 || k <= 0) return 0;
  return v3d::carve_knn_ws(nullptr, batch, N, M, v3d::knn_feat_select_k(k)).bytes;
}

int v3d_knn_feat_f32(const float* x, const float* y, float* distances, int64_t* indices, int batch, int N, int M,
                     int C, int k, int64_t index_offset, void* workspace, size_t workspace_bytes, int device,
                     void* stream_) {
  using namespace v3d;
  const char* what = "knn_feat";
  V3D_REQUIRE(batch >= 0 && N >= 0 && M >= 0 && C >= 0 && k >= 0, what, "negative dimension");
  if (k > V3D_KNN_FEAT_MAX_K) {
    set_last_error("%s: k=%d exceeds %d", what, k, V3D_KNN_FEAT_MAX_K);
    return V3D_ERR_K_TOO_LARGE;
  }
  if (batch == 0 || N == 0 || k == 0) return 0;
  V3D_REQUIRE(x && distances && indices && (M == 0 || y), what, "null pointer");
  if (batch > kMaxGridYZ || ceil_div(N, PD_BM) > kMaxGridYZ || (int64_t)batch * N > 0x7fffffffll) {
    set_last_error("%s: batch <= 65535, N <= 8388480 and batch * N < 2^31 required", what);
    return V3D_ERR_SIZE_LIMIT;
  }
  DeviceGuard guard(device);
  if (guard.rc) return guard.rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int64_t rows = (int64_t)batch * N;
  const int Mw = M > 0 ? M : 1;
  const int kout = k;
  k = knn_feat_select_k(kout);
  const size_t need = carve_knn_ws(nullptr, batch, N, Mw, k).bytes;
  void* base = workspace;
  bool own = false;
  if (base == nullptr) {
    V3D_HIP_TRY(hipMallocAsync(&base, need, stream), what);
    own = true;
  } else if (workspace_bytes < need || ((uintptr_t)base & 255) != 0) {
    set_last_error("%s: workspace needs %zu bytes, 256-B aligned (got %zu)", what, need, workspace_bytes);
    return V3D_ERR_WORKSPACE;
  }
  KnnFeatWorkspace ws = carve_knn_ws(base, batch, N, Mw, k);
  int rc = 0;
  const unsigned row_blocks = (unsigned)ceil_div(rows, 4);
  const bool large = kout > 64;  // a workgroup per row instead of a wave per row
  const size_t sel_bytes = sizeof(uint2) * (size_t)k;
  auto select_rows = [&](int ncols, int col_base, int first) {
    if (large)
      hipLaunchKernelGGL(topk_rows_large_kernel, dim3((unsigned)rows), dim3(256), sel_bytes, stream, ws.chunk, ncols,
                         (int64_t)ws.mc, col_base, k, first, ws.run_d, ws.run_i);
    else
      hipLaunchKernelGGL(topk_rows_kernel, dim3(row_blocks), dim3(256), 0, stream, ws.chunk, rows, ncols,
                         (int64_t)ws.mc, col_base, k, first, ws.run_d, ws.run_i);
  };
  if (M == 0) select_rows(0, 0, 1);  // no candidates: every slot is (inf, -1)
  for (int m0 = 0; m0 < M && rc == 0; m0 += ws.mc) {
    const int mc = std::min(ws.mc, M - m0);
    // distances of every row of every batch against columns [m0, m0+mc): batch b uses y + (b*M + m0)*C
    const float* ychunk = y + (int64_t)m0 * C;
    // only mc columns are visited per cloud, but consecutive clouds of y stay M*C apart
    rc = launch_pairwise(x, ychunk, ws.chunk, batch, N, mc, C, (int64_t)M * C, (int64_t)ws.mc, 0, 0.f, stream);
    if (rc) break;
    select_rows(mc, m0, m0 == 0 ? 1 : 0);
    rc = launch_status(what);
  }
  if (rc == 0) {
    const int vec4 = (C % 4 == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0) &&
                     (int64_t)Mw * C <= (int64_t)INT32_MAX;
    if (large)
      hipLaunchKernelGGL(knn_feat_finalize_large_kernel, dim3((unsigned)rows), dim3(256), sel_bytes, stream, x, y,
                         ws.run_i, N, Mw, C, k, kout, index_offset, distances, indices);
    else
      // M == 0: every candidate is "none" and lanes without one read row 0 of the cloud -- give them x's, y may be null
      hipLaunchKernelGGL(knn_feat_finalize_kernel, dim3(row_blocks), dim3(256), 0, stream, x, M > 0 ? y : x, ws.run_d,
                         ws.run_i, batch, N, Mw, C, k, kout, index_offset, vec4, distances, indices);
    rc = launch_status(what);
  }
  if (own) {
    hipError_t e = hipFreeAsync(base, stream);
    if (rc == 0 && e != hipSuccess) {
      set_last_error("%s: hipFreeAsync: %s", what, hipGetErrorString(e));
      rc = (int)e;
    }
  }
  return rc;
}

int v3d_topk_merge_f32(const float* parts_dist, const int64_t* parts_idx, float* out_dist, int64_t* out_idx, int G,
                       int64_t rows, int k, int device, void* stream_) {
  using namespace v3d;
  const char* what = "topk_merge";
  V3D_REQUIRE(G >= 1 && rows >= 0 && k >= 0, what, "bad dimension");
  if (rows == 0 || k == 0) return 0;
  V3D_REQUIRE(parts_dist && parts_idx && out_dist && out_idx, what, "null pointer");
  DeviceGuard guard(device);
  if (guard.rc) return guard.rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  hipLaunchKernelGGL(topk_merge_kernel, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, stream, parts_dist, parts_idx,
                     out_dist, out_idx, G, rows, k);
  return launch_status(what);
}

}  // extern "C"
