/*
 * v3d_oracle.c -- CPU oracle for the vision3d point-cloud operator hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under vision3d_amd/ may import, link or call this file; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and only as the checker.
 *
 * Each function is a sequential restatement of one reference CUDA kernel
 * (/root/reference/vision3d/csrc/cuda/<op>/<op>_cuda.cuh); the cited lines are the ones it follows.
 * "Reference order" of the squared distance is the source expression
 *     (qx-sx)*(qx-sx) + (qy-sy)*(qy-sy) + (qz-sz)*(qz-sz)
 * evaluated left to right in the element type, one rounding per operation, NO fused multiply-add
 * (build with -ffp-contract=off; see oracle/Makefile).
 *
 * Pinning status: the reference ships no golden vectors for these kernels and its CUDA
 * sources cannot be built here (no nvcc / no GPU).  The restatement is pinned by
 *   - scipy cKDTree through the reference's own array_ops/knn.py and array_ops/ball_query.py
 *     (neighbour sets and distances, tie-free inputs)                     -> tests/test_oracle_pin.py
 *   - hand-worked known-answer cases for the tie rules                    -> tests/test_oracle_kat.py
 * TIE ORDER (FPS tree reduction, knn heap mechanics) is "parity unpinned": it follows the source text only.
 *
 * Threads: v3d_oracle_set_threads(n) selects how many OpenMP threads split the (batch, query) loops;
 * results do not depend on it (every output element is produced by exactly one sequential loop).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int g_threads = 1;

void v3d_oracle_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int v3d_oracle_get_threads(void) { return g_threads; }
int v3d_oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------------------------------------
 * furthest point sampling
 * furthest_point_sample_cuda.cuh:9-12  opt_n_threads  (double log ratio, truncation, clamp to [1,512])
 * ---------------------------------------------------------------------------------------------- */
int v3d_oracle_fps_block(int work_size) {
  int pow_2 = (int)(log((double)work_size) / log(2.0));
  int b = 1 << pow_2;
  if (b > 512) b = 512;
  if (b < 1) b = 1;
  return b;
}

/* furthest_point_sample_cuda.cuh:29-88.  `block` is the reference launch's blockDim (opt_n_threads(N));
 * it decides which of several equally-far points wins (per-thread strided first-max :60-71, then the
 * shared-memory tree with strict v2>v1 :14-25,:76-83).  pts (B,N,3), dist (B,N) in/out, idx (B,M). */
static void fps_one_cloud(const float* pts, float* dist, int64_t* idx, int N, int M, int block,
                          float* sh_d, int* sh_i) {
  if (M <= 0) return;               /* :37 */
  int old = 0;                      /* :50 */
  idx[0] = old;                     /* :51 */
  for (int s = 1; s < M; ++s) {     /* :54 */
    float x1 = pts[old * 3 + 0], y1 = pts[old * 3 + 1], z1 = pts[old * 3 + 2];
    for (int tid = 0; tid < block; ++tid) {
      int besti = 0;                /* :55 */
      float best = -1.0f;           /* :56 */
      for (int p = tid; p < N; p += block) { /* :60 */
        float x2 = pts[p * 3 + 0], y2 = pts[p * 3 + 1], z2 = pts[p * 3 + 2];
        float d = (x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2) + (z1 - z2) * (z1 - z2); /* :64 */
        float dp = dist[p];
        d = d < dp ? d : dp;        /* :65  min() */
        dist[p] = d;                /* :66 */
        if (d > best) {             /* :67 */
          best = d;
          besti = p;
        }
      }
      sh_d[tid] = best;             /* :72-73 */
      sh_i[tid] = besti;
    }
    for (int thresh = 512; thresh >= 2; thresh >>= 1) { /* :77 */
      int half = thresh >> 1;
      if (block >= thresh) {
        for (int tid = 0; tid < half; ++tid) {          /* :79-80, __update :14-25 */
          if (sh_d[tid + half] > sh_d[tid]) {
            sh_d[tid] = sh_d[tid + half];
            sh_i[tid] = sh_i[tid + half];
          }
        }
      }
    }
    old = sh_i[0];                  /* :85 */
    idx[s] = old;                   /* :86 */
  }
}

void v3d_oracle_fps_f32(const float* pts, float* dist, int64_t* idx, int B, int N, int M) {
  if (B <= 0 || N <= 0) return;
  int block = v3d_oracle_fps_block(N);
#pragma omp parallel num_threads(g_threads)
  {
    float* sh_d = (float*)malloc(sizeof(float) * 512);
    int* sh_i = (int*)malloc(sizeof(int) * 512);
#pragma omp for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b)
      fps_one_cloud(pts + (size_t)b * N * 3, dist + (size_t)b * N, idx + (size_t)b * M, N, M, block, sh_d,
                    sh_i);
    free(sh_d);
    free(sh_i);
  }
}

/* ------------------------------------------------------------------------------------------------
 * ball query   ball_query_cuda.cuh:11-55     q (B,M,3)  s (B,N,3)  idx (B,M,k) arrives zero-filled
 * ---------------------------------------------------------------------------------------------- */
#define BALL_QUERY_IMPL(NAME, T)                                                                         \
  void NAME(const T* q, const T* s, int64_t* idx, int B, int M, int N, float max_radius, int k) {        \
    T r2 = (T)max_radius * (T)max_radius; /* :28 */                                                      \
    long total = (long)B * M;                                                                            \
    _Pragma("omp parallel for schedule(static) num_threads(g_threads)")                                  \
    for (long bq = 0; bq < total; ++bq) {                                                                \
      int b = (int)(bq / M);                                                                             \
      const T* qp = q + (size_t)bq * 3;                                                                  \
      const T* sp = s + (size_t)b * N * 3;                                                               \
      int64_t* out = idx + (size_t)bq * k;                                                               \
      T qx = qp[0], qy = qp[1], qz = qp[2];                                                              \
      int cnt = 0; /* :37 */                                                                             \
      for (int j = 0; j < N; ++j) {                                                                      \
        T sx = sp[j * 3 + 0], sy = sp[j * 3 + 1], sz = sp[j * 3 + 2];                                    \
        T d = (qx - sx) * (qx - sx) + (qy - sy) * (qy - sy) + (qz - sz) * (qz - sz); /* :42 */           \
        if (d < r2) { /* :43 strict */                                                                   \
          if (cnt == 0)                                                                                  \
            for (int i = 0; i < k; ++i) out[i] = j; /* :44-48 */                                         \
          if (cnt < k) out[cnt] = j; /* :49 (cnt<k always holds before the break when k>0) */            \
          ++cnt;                                                                                         \
          if (cnt >= k) break; /* :51 */                                                                 \
        }                                                                                                \
      }                                                                                                  \
    }                                                                                                    \
  }
BALL_QUERY_IMPL(v3d_oracle_ball_query_f32, float)
BALL_QUERY_IMPL(v3d_oracle_ball_query_f64, double)

/* ------------------------------------------------------------------------------------------------
 * knn_points   knn_points_cuda.cuh:9-93      dist (B,M,k) = sqrt(sq), idx (B,M,k); k <= 100 (:68-69)
 * ---------------------------------------------------------------------------------------------- */
#define KNN_IMPL(NAME, T, SQRT)                                                                          \
  static void NAME##_reheap(T* dist, int* idx, int k) { /* :16-28 */                                     \
    int root = 0, child = 1;                                                                             \
    while (child < k) {                                                                                  \
      if (child + 1 < k && dist[child + 1] > dist[child]) child++;                                       \
      if (dist[root] > dist[child]) return;                                                              \
      T td = dist[root]; dist[root] = dist[child]; dist[child] = td;                                     \
      int ti = idx[root]; idx[root] = idx[child]; idx[child] = ti;                                       \
      root = child;                                                                                      \
      child = root * 2 + 1;                                                                              \
    }                                                                                                    \
  }                                                                                                      \
  int NAME(const T* q, const T* s, T* kd, int64_t* ki, int B, int M, int N, int k) {                     \
    if (k > 100 || k < 0) return -1; /* the reference overruns best_dist[100] here; we refuse */         \
    long total = (long)B * M;                                                                            \
    _Pragma("omp parallel for schedule(static) num_threads(g_threads)")                                  \
    for (long bq = 0; bq < total; ++bq) {                                                                \
      int b = (int)(bq / M);                                                                             \
      const T* qp = q + (size_t)bq * 3;                                                                  \
      const T* sp = s + (size_t)b * N * 3;                                                               \
      T qx = qp[0], qy = qp[1], qz = qp[2];                                                              \
      T best_dist[100];                                                                                  \
      int best_idx[100];                                                                                 \
      for (int n = 0; n < k; ++n) { best_dist[n] = (T)1e10; best_idx[n] = 0; } /* :70-73 */              \
      for (int j = 0; j < N; ++j) {                                                                      \
        T sx = sp[j * 3 + 0], sy = sp[j * 3 + 1], sz = sp[j * 3 + 2];                                    \
        T d = (qx - sx) * (qx - sx) + (qy - sy) * (qy - sy) + (qz - sz) * (qz - sz); /* :79 */           \
        if (k > 0 && d < best_dist[0]) { /* :80 */                                                       \
          best_dist[0] = d;                                                                              \
          best_idx[0] = j;                                                                               \
          NAME##_reheap(best_dist, best_idx, k);                                                         \
        }                                                                                                \
      }                                                                                                  \
      for (int i = k - 1; i > 0; --i) { /* heap_sort :30-37 */                                           \
        T td = best_dist[0]; best_dist[0] = best_dist[i]; best_dist[i] = td;                             \
        int ti = best_idx[0]; best_idx[0] = best_idx[i]; best_idx[i] = ti;                               \
        NAME##_reheap(best_dist, best_idx, i);                                                           \
      }                                                                                                  \
      for (int n = 0; n < k; ++n) { /* :88-91 */                                                         \
        ki[(size_t)bq * k + n] = best_idx[n];                                                            \
        kd[(size_t)bq * k + n] = SQRT(best_dist[n]);                                                     \
      }                                                                                                  \
    }                                                                                                    \
    return 0;                                                                                            \
  }
KNN_IMPL(v3d_oracle_knn_points_f32, float, sqrtf)
KNN_IMPL(v3d_oracle_knn_points_f64, double, sqrt)

/* ------------------------------------------------------------------------------------------------
 * three_nn   three_nn_cuda.cuh:11-75     bests kept as double (:37-39), idx -1 (:40-42)
 * ---------------------------------------------------------------------------------------------- */
#define THREE_NN_IMPL(NAME, T)                                                                           \
  void NAME(const T* q, const T* s, T* td, int64_t* ti, int B, int Q, int S) {                           \
    long total = (long)B * Q;                                                                            \
    _Pragma("omp parallel for schedule(static) num_threads(g_threads)")                                  \
    for (long bq = 0; bq < total; ++bq) {                                                                \
      int b = (int)(bq / Q);                                                                             \
      const T* qp = q + (size_t)bq * 3;                                                                  \
      const T* sp = s + (size_t)b * S * 3;                                                               \
      T qx = qp[0], qy = qp[1], qz = qp[2];                                                              \
      double best1 = 1e40, best2 = 1e40, best3 = 1e40;                                                   \
      int i1 = -1, i2 = -1, i3 = -1;                                                                     \
      for (int j = 0; j < S; ++j) {                                                                      \
        T sx = sp[j * 3 + 0], sy = sp[j * 3 + 1], sz = sp[j * 3 + 2];                                    \
        T d = (qx - sx) * (qx - sx) + (qy - sy) * (qy - sy) + (qz - sz) * (qz - sz); /* :48 */           \
        if (d < best1) { /* :49-64 */                                                                    \
          best3 = best2; i3 = i2; best2 = best1; i2 = i1; best1 = d; i1 = j;                             \
        } else if (d < best2) {                                                                          \
          best3 = best2; i3 = i2; best2 = d; i2 = j;                                                     \
        } else if (d < best3) {                                                                          \
          best3 = d; i3 = j;                                                                             \
        }                                                                                                \
      }                                                                                                  \
      td[(size_t)bq * 3 + 0] = (T)sqrt(best1); /* :67-69 */                                              \
      td[(size_t)bq * 3 + 1] = (T)sqrt(best2);                                                           \
      td[(size_t)bq * 3 + 2] = (T)sqrt(best3);                                                           \
      ti[(size_t)bq * 3 + 0] = i1; /* :71-73 */                                                          \
      ti[(size_t)bq * 3 + 1] = i2;                                                                       \
      ti[(size_t)bq * 3 + 2] = i3;                                                                       \
    }                                                                                                    \
  }
THREE_NN_IMPL(v3d_oracle_three_nn_f32, float)
THREE_NN_IMPL(v3d_oracle_three_nn_f64, double)

/* ------------------------------------------------------------------------------------------------
 * three_interpolate   three_interpolate_cuda.cuh:11-44 (forward) and :48-83 (backward)
 * src (B,C,S)  idx (B,Q,3)  w (B,Q,3)  out (B,C,Q).  Backward adds in ascending (q, j) order, which is
 * ONE of the orders the reference's atomics may take; products are rounded before the add (:79-81).
 * ---------------------------------------------------------------------------------------------- */
#define THREE_INTERP_IMPL(FWD, BWD, T)                                                                   \
  void FWD(const T* src, const int64_t* idx, const T* w, T* out, int B, int C, int S, int Q) {           \
    long total = (long)B * C;                                                                            \
    _Pragma("omp parallel for schedule(static) num_threads(g_threads)")                                  \
    for (long bc = 0; bc < total; ++bc) {                                                                \
      int b = (int)(bc / C);                                                                             \
      const T* sp = src + (size_t)bc * S;                                                                \
      T* op = out + (size_t)bc * Q;                                                                      \
      for (int t = 0; t < Q; ++t) {                                                                      \
        const int64_t* ci = idx + ((size_t)b * Q + t) * 3;                                               \
        const T* cw = w + ((size_t)b * Q + t) * 3;                                                       \
        int a0 = (int)ci[0], a1 = (int)ci[1], a2 = (int)ci[2]; /* :38-40 long -> int */                  \
        op[t] = sp[a0] * cw[0] + sp[a1] * cw[1] + sp[a2] * cw[2]; /* :42 */                              \
      }                                                                                                  \
    }                                                                                                    \
  }                                                                                                      \
  void BWD(const T* gout, const int64_t* idx, const T* w, T* gsrc, int B, int C, int S, int Q) {         \
    long total = (long)B * C;                                                                            \
    _Pragma("omp parallel for schedule(static) num_threads(g_threads)")                                  \
    for (long bc = 0; bc < total; ++bc) {                                                                \
      int b = (int)(bc / C);                                                                             \
      const T* gp = gout + (size_t)bc * Q;                                                               \
      T* sp = gsrc + (size_t)bc * S;                                                                     \
      for (int t = 0; t < Q; ++t) {                                                                      \
        const int64_t* ci = idx + ((size_t)b * Q + t) * 3;                                               \
        const T* cw = w + ((size_t)b * Q + t) * 3;                                                       \
        for (int j = 0; j < 3; ++j) {                                                                    \
          T prod = gp[t] * cw[j];                                                                        \
          sp[(int)ci[j]] += prod; /* :79-81 */                                                           \
        }                                                                                                \
      }                                                                                                  \
    }                                                                                                    \
  }
THREE_INTERP_IMPL(v3d_oracle_three_interp_fwd_f32, v3d_oracle_three_interp_bwd_f32, float)
THREE_INTERP_IMPL(v3d_oracle_three_interp_fwd_f64, v3d_oracle_three_interp_bwd_f64, double)

/* ------------------------------------------------------------------------------------------------
 * group_gather / gather.   group_gather_cuda.cuh:11-33,37-59 ; gather_cuda.cuh:11-33,37-59.
 * gather is group_gather with one sample per target, so both share `L` = M*K (or M) flat positions.
 * Forward is a byte copy for any element size (the reference dispatches ALL_TYPES + Bool).
 * ---------------------------------------------------------------------------------------------- */
void v3d_oracle_group_gather_fwd(const void* src, const int64_t* idx, void* out, int B, int C, int N, long L,
                                 int elem) {
  long total = (long)B * C;
#pragma omp parallel for schedule(static) num_threads(g_threads)
  for (long bc = 0; bc < total; ++bc) {
    int b = (int)(bc / C);
    const char* sp = (const char*)src + (size_t)bc * N * elem;
    char* op = (char*)out + (size_t)bc * L * elem;
    const int64_t* ip = idx + (size_t)b * L;
    for (long t = 0; t < L; ++t) {
      int si = (int)ip[t]; /* group_gather_cuda.cuh:30 long -> int */
      memcpy(op + (size_t)t * elem, sp + (size_t)si * elem, (size_t)elem); /* :31 */
    }
  }
}

#define GATHER_BWD_IMPL(NAME, T)                                                                         \
  void NAME(const T* gout, const int64_t* idx, T* gsrc, int B, int C, int N, long L) {                   \
    long total = (long)B * C;                                                                            \
    _Pragma("omp parallel for schedule(static) num_threads(g_threads)")                                  \
    for (long bc = 0; bc < total; ++bc) {                                                                \
      int b = (int)(bc / C);                                                                             \
      const T* gp = gout + (size_t)bc * L;                                                               \
      T* sp = gsrc + (size_t)bc * N;                                                                     \
      const int64_t* ip = idx + (size_t)b * L;                                                           \
      for (long t = 0; t < L; ++t) sp[(int)ip[t]] += gp[t]; /* group_gather_cuda.cuh:56-57 */            \
    }                                                                                                    \
  }
GATHER_BWD_IMPL(v3d_oracle_group_gather_bwd_f32, float)
GATHER_BWD_IMPL(v3d_oracle_group_gather_bwd_f64, double)

/* ------------------------------------------------------------------------------------------------
 * pairwise_distance, expansion branch of ops/pairwise_distance.py:39-56, fp32, per batch:
 *   xy = x . y^T (k-ordered fp32 sum; torch's own GEMM order is unspecified, so parity here is a
 *   tolerance, see tests), x2/y2 = sum of squares, d = x2 - 2*xy + y2 (or 2 - 2*xy), clamp at 0,
 *   optional sqrt(d + eps).   x (batch,N,C)  y (batch,M,C)  out (batch,N,M)
 * ---------------------------------------------------------------------------------------------- */
void v3d_oracle_pairwise_distance_f32(const float* x, const float* y, float* out, int batch, int N, int M,
                                      int C, int normalized, int squared, float eps) {
  long total = (long)batch * N;
#pragma omp parallel for schedule(static) num_threads(g_threads)
  for (long bn = 0; bn < total; ++bn) {
    int b = (int)(bn / N);
    const float* xr = x + (size_t)bn * C;
    float x2 = 0.f;
    for (int c = 0; c < C; ++c) x2 += xr[c] * xr[c];
    for (int m = 0; m < M; ++m) {
      const float* yr = y + ((size_t)b * M + m) * C;
      float xy = 0.f, y2 = 0.f;
      for (int c = 0; c < C; ++c) {
        xy += xr[c] * yr[c];
        y2 += yr[c] * yr[c];
      }
      float d = normalized ? (2.0f - 2.0f * xy) : ((x2 - 2.0f * xy) + y2); /* :47,:51 */
      if (d < 0.f) d = 0.f;                                                 /* :53 */
      if (!squared) d = sqrtf(d + eps);                                     /* :56 */
      out[(size_t)bn * M + m] = d;
    }
  }
}

/* Feature-space KNN restated from its call site ops/knn.py:10-27 (pykeops Kmin_argKmin of the Euclidean
 * norm of the difference; third-party, absent here -> "parity unpinned"): k smallest by
 * (distance, index) ascending, distance = sqrt(sum_c (x_c - y_c)^2) accumulated in double.
 * x (batch,N,C) y (batch,M,C) -> dist (batch,N,k) idx (batch,N,k).  O(N*M*(C+k)); small cases only. */
void v3d_oracle_knn_feat_f32(const float* x, const float* y, float* kd, int64_t* ki, int batch, int N, int M,
                             int C, int k) {
  long total = (long)batch * N;
#pragma omp parallel for schedule(static) num_threads(g_threads)
  for (long bn = 0; bn < total; ++bn) {
    int b = (int)(bn / N);
    const float* xr = x + (size_t)bn * C;
    double* bd = (double*)malloc(sizeof(double) * (size_t)(k > 0 ? k : 1));
    int64_t* bi = (int64_t*)malloc(sizeof(int64_t) * (size_t)(k > 0 ? k : 1));
    int cnt = 0;
    for (int m = 0; m < M; ++m) {
      const float* yr = y + ((size_t)b * M + m) * C;
      double acc = 0.0;
      for (int c = 0; c < C; ++c) {
        double df = (double)xr[c] - (double)yr[c];
        acc += df * df;
      }
      if (cnt < k || acc < bd[cnt - 1]) { /* stable insertion: equal distances keep the lower index first */
        int pos = cnt < k ? cnt : k - 1;
        while (pos > 0 && bd[pos - 1] > acc) {
          bd[pos] = bd[pos - 1];
          bi[pos] = bi[pos - 1];
          --pos;
        }
        bd[pos] = acc;
        bi[pos] = m;
        if (cnt < k) ++cnt;
      }
    }
    for (int n = 0; n < k; ++n) {
      kd[(size_t)bn * k + n] = n < cnt ? (float)sqrt(bd[n]) : INFINITY;
      ki[(size_t)bn * k + n] = n < cnt ? bi[n] : -1;
    }
    free(bd);
    free(bi);
  }
}
