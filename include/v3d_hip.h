/*
 * v3d_hip.h -- C ABI of libv3d_hip.so, the gfx950 (MI355X) replacement for the CUDA half of the
 * reference's `vision3d.ext` extension (/root/reference/vision3d/csrc/pybind.cpp:71-158).
 *
 * Conventions (all entry points):
 *   - plain device pointers + sizes, no torch types; the caller owns every buffer and pre-allocates every
 *     output exactly as the reference's Python wrappers do (vision3d/ops/<op>.py);
 *   - `device` is the HIP device ordinal the buffers live on, `stream` a hipStream_t passed as void*
 *     (NULL = the device's default stream).  Work is enqueued asynchronously; nothing synchronises, nothing
 *     is retained after the enqueued work ends (same as the reference launchers: CUDAGuard +
 *     getCurrentCUDAStream + launch + cudaGetLastError, e.g. ball_query_cuda.cu:16-38);
 *   - return value: 0 = enqueued; > 0 = a hipError_t from launch / runtime; < 0 = a V3D_ERR_* validation
 *     code.  Never throws.  v3d_last_error_string() describes the calling thread's last failure;
 *   - index tensors are int64 ("long", torch_helper.h:21); values are narrowed to 32 bit inside the kernels
 *     exactly like the reference (`int source_index = indices[...]`, group_gather_cuda.cuh:30), while all
 *     ADDRESS arithmetic is 64 bit (the reference overflows `int` past 2^31 elements; we do not);
 *   - squared distances are evaluated in "reference order" ((dx*dx + dy*dy) + dz*dz, one rounding per
 *     operation, no FMA contraction), see DESIGN.md.
 */
#ifndef V3D_HIP_H_
#define V3D_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define V3D_OK 0
#define V3D_ERR_INVALID_ARGUMENT (-1) /* null pointer, negative size, unsupported element width ... */
#define V3D_ERR_K_TOO_LARGE (-2)      /* knn_points: k > 100 (reference: best_dist[100], knn_points_cuda.cuh:68); the
                                         other searches: k beyond the V3D_*_MAX_* stated at their declarations */
#define V3D_ERR_WORKSPACE (-3)        /* caller workspace smaller than v3d_scatter_workspace_bytes() */
#define V3D_ERR_SIZE_LIMIT (-4)       /* a dimension exceeds what the 32-bit index VALUES can express */

#define V3D_KNN_MAX_K 100
#define V3D_KNN_FEAT_MAX_K 2048       /* v3d_knn_feat_f32 */
#define V3D_RADIUS_MAX_LIMIT 512      /* v3d_radius_search_pack_f32 */

/* Library identification / diagnostics (no reference counterpart). */
int v3d_version(void);                       /* 10000*major + 100*minor + patch */
const char* v3d_last_error_string(void);     /* thread-local, never NULL */
const char* v3d_error_name(int code);        /* "V3D_OK", "V3D_ERR_K_TOO_LARGE", or the hipGetErrorName text */
int v3d_device_count(int* count);            /* hipGetDeviceCount wrapper used by the host shim's load check */

/* ---- furthest point sampling ------------------------------------------------------------------------
 * replaces vision3d::furthest_point_sample (csrc/cuda/furthest_point_sample/furthest_point_sample.cpp:5-8,
 * launcher furthest_point_sample_cuda.cu:5-73, kernel furthest_point_sample_cuda.cuh:29-88).
 * points (B,N,3) f32; distances (B,N) f32 in/out scratch, arrives filled with 1e10
 * (ops/furthest_point_sample.py:34) and leaves holding the final min-distances as in the reference;
 * indices (B,M) i64.  Ties between equally far points are resolved as the reference's launch geometry
 * does (block = v3d_fps_ref_block(N)), independent of our own geometry. */
int v3d_fps_ref_block(int num_points);       /* opt_n_threads, furthest_point_sample_cuda.cuh:9-12 */
int v3d_fps_f32(const float* points, float* distances, int64_t* indices, int B, int N, int M, int device,
                void* stream);
/* The same sampling in pieces: rounds [first, last) of a run of M samples per cloud, indices (B,M).  The rounds
 * [0, first) were written into indices[:, :first] by earlier calls, which left the running min-distances in
 * `distances`; first % 64 == 0, last <= M.  Calls with (first, last) = (0, m1), (m1, m2), ..., (mj, M) produce exactly
 * what v3d_fps_f32 produces.  No counterpart in the reference (its kernel always runs all M rounds,
 * furthest_point_sample_cuda.cuh:54): it is for callers that start the consumers of the first samples (ball query,
 * grouping) while the sampling goes on -- the sampling is a serial chain on B of the 256 CUs. */
int v3d_fps_resume_f32(const float* points, float* distances, int64_t* indices, int B, int N, int M, int first,
                       int last, int device, void* stream);

/* ---- ball query -------------------------------------------------------------------------------------
 * replaces vision3d::ball_query (csrc/cuda/ball_query/ball_query.h:7-12; kernel ball_query_cuda.cuh:11-55).
 * q (B,M,3), s (B,N,3), indices (B,M,k) i64 arriving zero-filled (ops/ball_query.py:38): slots of a query
 * with no neighbour are left untouched, as in the reference.  max_radius is a C float in both variants
 * (ball_query.h:11) and squared in the element type (ball_query_cuda.cuh:28). */
int v3d_ball_query_f32(const float* q, const float* s, int64_t* indices, int B, int M, int N, float max_radius,
                       int k, int device, void* stream);
int v3d_ball_query_f64(const double* q, const double* s, int64_t* indices, int B, int M, int N,
                       float max_radius, int k, int device, void* stream);

/* Multi-scale form for PointNet++ MSG grouping (models/pointnet2/modules.py:47-57 calls ball_query once per
 * (radius, k) on the same queries and supports): num_scales (1..4) outputs in ONE pass over the pairs.  indices /
 * max_radii / num_samples are HOST arrays of num_scales entries; indices[i] is the device tensor (B,M,num_samples[i]) of
 * scale i, arriving zero-filled like above.  Every output is bit-identical to v3d_ball_query_f32 with that scale's
 * arguments. */
int v3d_ball_query_multi_f32(const float* q, const float* s, int64_t* const* indices, const float* max_radii,
                             const int* num_samples, int num_scales, int B, int M, int N, int device, void* stream);

/* ---- knn_points (3-D, heap based) -------------------------------------------------------------------
 * replaces vision3d::knn_points (csrc/cuda/knn_points/knn_points.h; kernel knn_points_cuda.cuh:41-93),
 * also reachable as `knn` through the host shim (ops/knn_points.py:36 calls ext.knn).
 * distances (B,M,k) = sqrt of squared distance ascending, indices (B,M,k); N<k leaves trailing (1e5, 0).
 * Equal distances come out in the order the reference's heap produces.  k > 100 -> V3D_ERR_K_TOO_LARGE. */
int v3d_knn_points_f32(const float* q, const float* s, float* distances, int64_t* indices, int B, int M, int N,
                       int k, int device, void* stream);
int v3d_knn_points_f64(const double* q, const double* s, double* distances, int64_t* indices, int B, int M,
                       int N, int k, int device, void* stream);

/* ---- three_nn ---------------------------------------------------------------------------------------
 * replaces vision3d::three_nn (csrc/cuda/three_nn/three_nn.h; kernel three_nn_cuda.cuh:11-75).
 * q (B,Q,3), s (B,S,3) -> distances (B,Q,3) (sqrt), indices (B,Q,3); S<3 leaves (1e20, -1) in the
 * unfilled slots like the reference's 1e40 / -1 initialisers (:37-42,:67-73). */
int v3d_three_nn_f32(const float* q, const float* s, float* distances, int64_t* indices, int B, int Q, int S,
                     int device, void* stream);
int v3d_three_nn_f64(const double* q, const double* s, double* distances, int64_t* indices, int B, int Q, int S,
                     int device, void* stream);

/* Count of supports strictly inside `radius` of each query (no kernel in the reference: it evaluates
 * relu(radius - |q-s|).sign().sum() with pykeops, ops/radius_search.py:74-81).  counts (B,M) i64. */
int v3d_radius_count_f32(const float* q, const float* s, int64_t* counts, int B, int M, int N, float radius,
                         int device, void* stream);

/* ---- three_interpolate ------------------------------------------------------------------------------
 * replaces vision3d::three_interpolate_forward / _backward (csrc/cuda/three_interpolate/three_interpolate.h;
 * kernels three_interpolate_cuda.cuh:11-44 and :48-83).
 * sources (B,C,S), indices (B,Q,3) i64, weights (B,Q,3), targets (B,C,Q).
 * Backward: source_grads (B,C,S) is fully overwritten (the reference accumulates into a zero-filled buffer
 * with atomics; we write the sum once, atomic-free and bit-reproducible, in ascending (q,j) order).
 * It needs scratch: pass `workspace` of at least v3d_scatter_workspace_bytes(B, S, Q*3) bytes on the same
 * device (256-B aligned), or NULL to let the library hipMallocAsync/hipFreeAsync it on `stream`. */
int v3d_three_interp_fwd_f32(const float* sources, const int64_t* indices, const float* weights, float* targets,
                             int B, int C, int S, int Q, int device, void* stream);
int v3d_three_interp_fwd_f64(const double* sources, const int64_t* indices, const double* weights,
                             double* targets, int B, int C, int S, int Q, int device, void* stream);
size_t v3d_scatter_workspace_bytes(int B, int num_sources, int64_t positions_per_batch);
int v3d_three_interp_bwd_f32(const float* target_grads, const int64_t* indices, const float* weights,
                             float* source_grads, int B, int C, int S, int Q, void* workspace,
                             size_t workspace_bytes, int device, void* stream);
int v3d_three_interp_bwd_f64(const double* target_grads, const int64_t* indices, const double* weights,
                             double* source_grads, int B, int C, int S, int Q, void* workspace,
                             size_t workspace_bytes, int device, void* stream);

/* The same backward for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc) -- NO counterpart in the reference.  It calls no hipMalloc / hipFree / allocating routine: its
 * scratch is a V3D_NOALLOC_SCRATCH_BYTES array built into the library (one copy per GPU, held until the process ends).
 * Shapes whose v3d_scatter_workspace_bytes(B, S, Q*3) exceeds that size are refused with V3D_ERR_SIZE_LIMIT (use the
 * entry point above with a workspace).  All calls on one GPU share the array: calls that overlap on one GPU (two
 * streams, two threads) race -- serialise them.  Results are bit-identical to v3d_three_interp_bwd_f32. */
#define V3D_NOALLOC_SCRATCH_BYTES (16u << 20)
int v3d_three_interp_bwd_noalloc_f32(const float* target_grads, const int64_t* indices, const float* weights,
                                     float* source_grads, int B, int C, int S, int Q, int device, void* stream);

/* ---- gather / group_gather --------------------------------------------------------------------------
 * replace vision3d::gather_forward/_backward (csrc/cuda/gather/gather.h; gather_cuda.cuh:11-33,37-59) and
 * vision3d::group_gather_forward/_backward (csrc/cuda/group_gather/group_gather.h;
 * group_gather_cuda.cuh:11-33,37-59).
 * Forward: sources (B,C,N) of ANY element type of `elem_bytes` in {1,2,4,8} (the reference dispatches
 * AT_DISPATCH_ALL_TYPES_AND(Bool), gather_cuda.cu:22 / group_gather_cuda.cu:23 -- a gather is a byte copy);
 * indices (B,M) resp. (B,M,K) i64; targets (B,C,M) resp. (B,C,M,K).
 * Backward (f32/f64, gather_cuda.cu:53): source_grads (B,C,N) fully overwritten, atomic-free, summed in
 * ascending target order (== the CPU oracle's sequential loop, bit for bit).  Workspace as above with
 * positions_per_batch = M resp. M*K. */
int v3d_gather_fwd(const void* sources, const int64_t* indices, void* targets, int B, int C, int N, int M,
                   int elem_bytes, int device, void* stream);
int v3d_group_gather_fwd(const void* sources, const int64_t* indices, void* targets, int B, int C, int N, int M,
                         int K, int elem_bytes, int device, void* stream);
int v3d_gather_bwd_f32(const float* target_grads, const int64_t* indices, float* source_grads, int B, int C,
                       int N, int M, void* workspace, size_t workspace_bytes, int device, void* stream);
int v3d_gather_bwd_f64(const double* target_grads, const int64_t* indices, double* source_grads, int B, int C,
                       int N, int M, void* workspace, size_t workspace_bytes, int device, void* stream);
int v3d_group_gather_bwd_f32(const float* target_grads, const int64_t* indices, float* source_grads, int B,
                             int C, int N, int M, int K, void* workspace, size_t workspace_bytes, int device,
                             void* stream);
int v3d_group_gather_bwd_f64(const double* target_grads, const int64_t* indices, double* source_grads, int B,
                             int C, int N, int M, int K, void* workspace, size_t workspace_bytes, int device,
                             void* stream);

/* The backward in two steps (no counterpart in the reference, whose atomics need no lists): v3d_scatter_build inverts an
 * index tensor into `workspace` once -- indices (B, positions_per_batch) i64 over num_sources sources; mode 0 = gather /
 * group_gather, mode 1 = three_interpolate (positions_per_batch = 3*Q; weights (B,Q,3) of elem_bytes 4 or 8 are stored in
 * list order too) -- and v3d_scatter_apply_f32/_f64 writes source_grads (B,C,num_sources) from target_grads with those
 * lists, as often as wanted: the xyz and the feature gather of one set-abstraction layer share their indices
 * (layers/pointnet2.py:44-46), so does every channel width.  Results are bit-identical to the one-call entry points
 * above.  `workspace`: v3d_scatter_workspace_bytes(B, num_sources, positions_per_batch) bytes, 256-B aligned, REQUIRED
 * and owned by the caller for as long as the lists are used; apply only reads it.  `weights` in apply: the same tensor
 * the build saw (mode 1; read by clouds with more than 4096 sources), NULL in mode 0. */
int v3d_scatter_build(const int64_t* indices, const void* weights, int mode, int elem_bytes, int B, int num_sources,
                      int64_t positions_per_batch, void* workspace, size_t workspace_bytes, int device, void* stream);
int v3d_scatter_apply_f32(const float* target_grads, const float* weights, float* source_grads, int mode, int B, int C,
                          int num_sources, int64_t positions_per_batch, const void* workspace, size_t workspace_bytes,
                          int device, void* stream);
int v3d_scatter_apply_f64(const double* target_grads, const double* weights, double* source_grads, int mode, int B, int C,
                          int num_sources, int64_t positions_per_batch, const void* workspace, size_t workspace_bytes,
                          int device, void* stream);

/* ---- pairwise_distance / feature-space KNN ----------------------------------------------------------
 * v3d_pairwise_distance_f32 computes what ops/pairwise_distance.py:39-56 computes with torch.matmul and
 * 3-4 elementwise passes, in one kernel (fp32-input MFMA for the -2*x.y^T term, norms/clamp/sqrt in the
 * epilogue).  x (batch,N,C), y (batch,M,C) row-major ("transposed=False" layout), out (batch,N,M).
 * flags: bit0 normalized (d = 2 - 2xy, :46-47), bit1 non-squared (d = sqrt(d + eps), :55-56), bit2 strict (the
 * exact difference form of :28-37 on the VALU: sum_c (x_c-y_c)^2, or its sqrt with bit1; no clamp, no eps).
 * v3d_knn_feat_f32 is the fused form behind ops/knn.py:10-27 (pykeops Kmin_argKmin of the Euclidean norm):
 * k smallest by (distance, index), distances (batch,N,k) NOT squared, indices (batch,N,k) i64, offset by
 * `index_offset` (reference-set sharding).  k <= V3D_KNN_FEAT_MAX_K (pykeops has no limit; k <= 64 runs
 * a wave per query row, larger k a workgroup per row -- slower, same results); slots beyond M come out as (inf, -1).
 * `workspace` as above, size from v3d_knn_feat_workspace_bytes(); NULL = stream-ordered allocation. */
#define V3D_PD_NORMALIZED 1
#define V3D_PD_SQRT 2
#define V3D_PD_STRICT 4
int v3d_pairwise_distance_f32(const float* x, const float* y, float* out, int batch, int N, int M, int C,
                              int flags, float eps, int device, void* stream);
size_t v3d_knn_feat_workspace_bytes(int batch, int N, int M, int C, int k);
int v3d_knn_feat_f32(const float* x, const float* y, float* distances, int64_t* indices, int batch, int N,
                     int M, int C, int k, int64_t index_offset, void* workspace, size_t workspace_bytes,
                     int device, void* stream);
/* Merge G sorted partial top-k lists per row (the all-gathered shards of a reference-set partition) into the
 * global top-k: parts_* (G, rows, k) -> out_* (rows, k), ordered by (distance, index). */
int v3d_topk_merge_f32(const float* parts_dist, const int64_t* parts_idx, float* out_dist, int64_t* out_idx,
                       int G, int64_t rows, int k, int device, void* stream);

/* ---- "next" rows: pack-mode neighbourhood construction of the KPConv / GeoTransformer pyramid ----
 * Pack mode = variable-length clouds concatenated: points (total,3) fp32 + lengths (B,) int64, all in device memory.
 *
 * Radius search, replaces vision3d/ops/radius_search.py:44-71 (pack -> padded batch -> pykeops kNN -> mask -> pack):
 * neighbor_indices (total_q, neighbor_limit) = the supports of the query's own cloud with NOT(sqrt(d2) > radius),
 * ascending by (d2, index), first `neighbor_limit` (<= V3D_RADIUS_MAX_LIMIT), padded with total_s.  The lengths must sum
 * to the totals.
 * No host synchronisation.  Scratch: v3d_radius_search_workspace_bytes(B, total_s) bytes or NULL (stream-ordered
 * allocation). */
size_t v3d_radius_search_workspace_bytes(int B, int64_t total_s);
int v3d_radius_search_pack_f32(const float* q_points, const float* s_points, const int64_t* q_lengths,
                               const int64_t* s_lengths, int64_t* neighbor_indices, int B, int64_t total_q,
                               int64_t total_s, float radius, int neighbor_limit, void* workspace,
                               size_t workspace_bytes, int device, void* stream);
/* The same search with the cell list held by the caller: built once, searched by several query sets.  The pyramid
 * builder (models/graph_pyramid/graph_pyramid.py:31-45) searches the points of one level three times at one radius --
 * its neighbours, the subsampling of the level below into it, the upsampling of the level above -- and the reference
 * re-pads and re-searches each time (ops/radius_search.py:44-71); here the counting sort of the supports is shared.
 * v3d_radius_grid_build_f32 leaves in `grid` (v3d_radius_search_workspace_bytes(B, total_s) bytes, caller-owned, 256-B
 * aligned) a cell list that serves every radius <= cell_radius; v3d_radius_search_grid_f32 is
 * v3d_radius_search_pack_f32 on it -- identical result -- and takes the B, total_s and cell_radius of the build.
 * Searches on one grid must be ordered among themselves (each rewrites the query offsets kept in it). */
int v3d_radius_grid_build_f32(const float* s_points, const int64_t* s_lengths, int B, int64_t total_s, float cell_radius,
                              void* grid, size_t grid_bytes, int device, void* stream);
int v3d_radius_search_grid_f32(const float* q_points, const int64_t* q_lengths, int64_t* neighbor_indices, int B,
                               int64_t total_q, int64_t total_s, float cell_radius, float radius, int neighbor_limit,
                               void* grid, size_t grid_bytes, int device, void* stream);
/* Pack-mode kNN, replaces vision3d/ops/knn.py:110-145 (pack -> batch padded with points at 1e10 -> pykeops kNN -> pack):
 * for every query the k nearest supports of ITS OWN cloud: distances (total_q,k) = sqrt of the squared distance,
 * ascending; indices (total_q,k) i64 = positions inside the cloud (what the reference's padded batch yields).  Equal
 * distances come lower index first -- ascending (d2, index), the order of pykeops' Kmin_argKmin (strict-< insertion while
 * scanning the supports in index order) and of v3d_radius_search_pack_f32; NOT the heap order of v3d_knn_points_f32.
 * k <= V3D_KNN_MAX_K; a cloud with fewer than k supports (the reference asserts against it, :137) leaves (1e5, 0) in the
 * unfilled slots like v3d_knn_points_f32; lengths that overrun the totals are cut.  Clouds averaging >= 2048 supports
 * (k <= 64, B <= 4096, enough queries to pay for the build) are searched on the radius search's cell list, the rest pair
 * by pair.  No host
 * synchronisation.  Scratch: v3d_knn_pack_workspace_bytes(B, total_s, k) bytes or NULL (stream-ordered allocation). */
size_t v3d_knn_pack_workspace_bytes(int B, int64_t total_s, int k);
/* 1 if a call with these sizes is searched on the cell list, 0 if pair by pair (a handful of queries against large clouds
 * stay pair by pair: the cell list costs ~50 us of launches + a counting sort of the supports).  Informational -- the
 * result is the same either way; tests use it to know which kernels they exercise.  No counterpart in the reference. */
int v3d_knn_pack_uses_cell_list(int B, int64_t total_q, int64_t total_s, int k);
int v3d_knn_pack_f32(const float* q_points, const float* s_points, const int64_t* q_lengths, const int64_t* s_lengths,
                     float* distances, int64_t* indices, int B, int64_t total_q, int64_t total_s, int k,
                     void* workspace, size_t workspace_bytes, int device, void* stream);
/* Batch-mode kNN of 3-D points in pykeops' order: what vision3d/ops/knn.py:10-27 (Kmin_argKmin of the Euclidean norm)
 * yields for q (B,M,3) and s (B,N,3): distances (B,M,k) ascending, equal distances lower index first -- the kernels of
 * v3d_knn_points_f32 without the heap replay, with its limits (k <= V3D_KNN_MAX_K) and its (1e5, 0) in the slots a cloud
 * of fewer than k supports leaves unfilled.  ops.knn sends 3-D inputs here, any other channel count to v3d_knn_feat_f32. */
int v3d_knn_lex_f32(const float* q, const float* s, float* distances, int64_t* indices, int B, int M, int N, int k,
                    int device, void* stream);
/* Radius count, replaces vision3d/ops/radius_search.py:74-107 (pack -> padded batch -> pykeops sum of
 * relu(radius - d).sign() -> pack): counts (total_q,) = the number of supports of the query's own cloud with
 * radius - sqrt(d2) > 0 (strict).  Same cell list, workspace and limits as the search above. */
int v3d_radius_count_pack_f32(const float* q_points, const float* s_points, const int64_t* q_lengths,
                              const int64_t* s_lengths, int64_t* counts, int B, int64_t total_q, int64_t total_s,
                              float radius, void* workspace, size_t workspace_bytes, int device, void* stream);

/* Grid (voxel) subsampling, replaces vision3d/ops/grid_subsample.py:49-99: one output point per occupied voxel of
 * edge `voxel_size` = mean of the voxel's points (summed in input order, fp32), ordered by (cloud, vx, vy, vz) like
 * torch.unique of the reference's hash.  sampled_points must hold `total` rows; the number actually written goes to
 * *num_sampled (device scalar; -1 if a cloud spans more than 2^18 voxels along an axis or holds NaN), per-cloud
 * counts to sampled_lengths (B,).  Scratch: `workspace` of v3d_grid_subsample_workspace_bytes(B, total, C) bytes (C = 0
 * here; the feature channels of v3d_voxelize_pack_f32), 256-B aligned, owned by the caller like every other buffer -- or
 * NULL for a stream-ordered allocation inside (hipMallocAsync / hipFreeAsync on `stream`). */
size_t v3d_grid_subsample_workspace_bytes(int B, int64_t total, int C);
int v3d_grid_subsample_pack_f32(const float* points, const int64_t* lengths, float* sampled_points,
                                int64_t* sampled_lengths, int64_t* num_sampled, int B, int64_t total, float voxel_size,
                                void* workspace, size_t workspace_bytes, int device, void* stream);
/* v3d_voxelize_pack_f32 replaces voxelize_pack_mode (vision3d/ops/voxelize.py:6-63; its batch form `voxelize` :66-100 is
 * pack/batch conversion around it): the same per-voxel reduction with voxel = floor(p / voxel_size) (:22, a correctly
 * rounded fp32 division; grid subsampling multiplies by 1 / voxel_size instead), and three more outputs: voxels (M,3)
 * int64 = the integer voxel coordinates (:43-44), voxel_feats (M,C) = the voxel means of feats (N,C) (:57-60; feats = NULL,
 * C = 0 without features), nodes (M,3) = the voxel means of the points (:46-49), voxel_lengths (B,) (:51-54).  Output
 * order = torch.unique of the hash = ascending (cloud, vx, vy, vz); sums run in input order (no atomics): bit-reproducible.
 * Output buffers hold `total` rows; *num_voxels = M, or -1 as above.  Workspace as for grid subsampling (with this C). */
int v3d_voxelize_pack_f32(const float* points, const int64_t* lengths, const float* feats, int C, int64_t* voxels,
                          float* nodes, float* voxel_feats, int64_t* voxel_lengths, int64_t* num_voxels, int B,
                          int64_t total, float voxel_size, void* workspace, size_t workspace_bytes, int device,
                          void* stream);

/* ---- "next" rows: correspondence selectors downstream of pairwise_distance -----------------------
 * v3d_mutual_topk_select_f32 replaces vision3d/ops/mutual_topk_select.py:7-60 (B = 1) and :63-133 (batched):
 * score (B,N,M) fp32; the top-k of every row and of every column (k <= 64, k <= N, k <= M like torch.topk; among equal
 * scores the lower index wins) are intersected (V3D_SEL_MUTUAL) or united, then filtered by `threshold`
 * (V3D_SEL_THRESHOLD: score > threshold if V3D_SEL_LARGEST else score < threshold, :48-53) and by the optional
 * row_masks (B,N) / col_masks (B,M) (bool bytes, :123-126; NULL = none).
 * Results, either or both: corr_mat (B,N,M) bool bytes, fully written (reduce_result=False, :60/:133); and the
 * compacted correspondences in torch.nonzero order (ascending batch, row, col): out_batch / out_row / out_col int64 and
 * out_score fp32 (each may be NULL), at most `capacity` entries written, the true count in *num_out (device scalar;
 * k*B*min(N,M) bounds it for mutual selection, k*B*(N+M) otherwise).  The score matrix is read twice, nothing of its
 * size is written unless corr_mat is requested.  No host synchronisation.  Scratch (all four selectors): `workspace` of
 * the matching v3d_*_workspace_bytes() bytes, 256-B aligned, owned by the caller -- or NULL for a stream-ordered
 * allocation inside (hipMallocAsync / hipFreeAsync on `stream`); a size of 0 means the call needs none.
 *
 * v3d_mutual_select_lists_f32 is the same back end for top-k lists that were never a matrix: row_idx/row_val (B,N,k) =
 * for every row its k best columns, col_idx/col_val (B,M,k) = for every column its k best rows (entries < 0 are
 * unfilled), e.g. the two v3d_knn_feat_f32 results of (x,y) and (y,x): mutual nearest-neighbour matching of 100k x 100k
 * descriptors (ops/matching.py:124-151) without the 40 GB matrix.  Values are needed for out_score / threshold only. */
#define V3D_SEL_LARGEST 1
#define V3D_SEL_MUTUAL 2
#define V3D_SEL_THRESHOLD 4
size_t v3d_mutual_topk_select_workspace_bytes(int B, int N, int M, int k);
size_t v3d_mutual_select_lists_workspace_bytes(int B, int N, int M, int k);
int v3d_mutual_topk_select_f32(const float* score, const uint8_t* row_masks, const uint8_t* col_masks, int B, int N,
                               int M, int k, int flags, float threshold, uint8_t* corr_mat, int64_t* out_batch,
                               int64_t* out_row, int64_t* out_col, float* out_score, int64_t* num_out, int64_t capacity,
                               void* workspace, size_t workspace_bytes, int device, void* stream);
int v3d_mutual_select_lists_f32(const int64_t* row_idx, const float* row_val, const int64_t* col_idx,
                                const float* col_val, const uint8_t* row_masks, const uint8_t* col_masks, int B, int N,
                                int M, int k, int flags, float threshold, uint8_t* corr_mat, int64_t* out_batch,
                                int64_t* out_row, int64_t* out_col, float* out_score, int64_t* num_out, int64_t capacity,
                                void* workspace, size_t workspace_bytes, int device, void* stream);

/* The two other selectors of vision3d/ops/matching.py, on a 2-D score matrix (N,M) fp32.
 * v3d_threshold_select_f32 replaces extract_correspondences_from_scores_threshold (:64-86) after its torch.exp: the cells
 * with score > threshold (V3D_SEL_LARGEST; < threshold without it) in torch.nonzero order -> out_row / out_col int64,
 * out_score fp32 (each may be NULL), at most `capacity` entries written, the true count in *num_out (device scalar,
 * required).  The matrix is read twice (count, emit); up to 64 Ki cells it is one launch.
 * v3d_global_topk_select_f32 replaces extract_correspondences_from_scores_topk (:89-121) and the flat top-k of
 * SuperPointMatchingGlobalTopk (models/geotransformer/superpoint_matching_global_topk.py:61-64): the k best cells of the
 * flattened matrix (largest with V3D_SEL_LARGEST), best first like torch.topk(sorted=True), equal scores by ascending
 * flat index (torch leaves that open); k > N*M is torch.topk's "selected index k out of range"; with V3D_SEL_DUSTBIN the
 * entries of the last row / last column are dropped AFTER the selection (:110-115), so *num_out <= k (it equals k
 * without the flag; num_out may be NULL then).  Exact radix select (3 histogram passes, +3 only when the k-th value is
 * tied), one compaction pass, sort of the k winners; N*M < 2^32.  No host synchronisation in either; scratch as above
 * (v3d_threshold_select_workspace_bytes / v3d_global_topk_select_workspace_bytes, or NULL). */
#define V3D_SEL_DUSTBIN 8
size_t v3d_threshold_select_workspace_bytes(int64_t N, int64_t M);
size_t v3d_global_topk_select_workspace_bytes(int64_t N, int64_t M, int64_t k);
int v3d_threshold_select_f32(const float* score, int64_t N, int64_t M, float threshold, int flags, int64_t* out_row,
                             int64_t* out_col, float* out_score, int64_t* num_out, int64_t capacity, void* workspace,
                             size_t workspace_bytes, int device, void* stream);
int v3d_global_topk_select_f32(const float* score, int64_t N, int64_t M, int64_t k, int flags, int64_t* out_row,
                               int64_t* out_col, float* out_score, int64_t* num_out, int64_t capacity, void* workspace,
                               size_t workspace_bytes, int device, void* stream);

/* ---- "next" rows: PACK-mode row gathers (features row-major (N,C), neighbour table (M,K) int64,
 * entries equal to N -- or anything outside [0,N) -- mean "no neighbour" and read as a row of zeros, which is what the
 * reference obtains by concatenating a zero row: cat(feats, zeros(1,C)) + index_select) ------------------------------
 *
 * v3d_index_select_rows replaces vision3d/ops/index_select.py:4-33 for dim = 0 on a 2-D source of any element type:
 * targets[l,:] = sources[indices[l * index_stride], :], L rows of row_bytes bytes.  index_stride = K with L = M reads
 * only the first column of an (M,K) table = ops/nearest_interpolate.py:6-20.
 * v3d_local_maxpool_pack_f32 replaces ops/pooling.py:6-19: pooled[m,c] = max_k padded[idx[m,k],c]; the (M,K,C)
 * intermediate is never written.  argk (optional, (M,C) bytes, K <= 255) = first k attaining the maximum, for backward.
 * v3d_knn_interp_weights_pack_f32 + v3d_rows_weighted_sum_f32 replace ops/knn_interpolate.py:43-76: weights (M,k_used)
 * = mask/(|q-s|^2+eps) normalised by (sum+eps); out[m,:] = sum_k weights[m,k] * padded[idx[m,k],:] (k ascending, one
 * rounding per multiply and per add).  K = row length of the table, k_used <= K = columns that take part.
 * v3d_global_avgpool_pack_f32 replaces ops/pooling.py:22-39 (the reference loops over clouds with .item() syncs).
 * C must be a multiple of 4 and the fp32 buffers 16-byte aligned for the *_f32 entry points. */
int v3d_index_select_rows(const void* sources, const int64_t* indices, int64_t index_stride, void* targets, int64_t L,
                          int64_t N, int64_t row_bytes, int device, void* stream);
int v3d_local_maxpool_pack_f32(const float* feats, const int64_t* neighbor_indices, float* pooled, uint8_t* argk,
                               int64_t M, int64_t N, int K, int C, int device, void* stream);
int v3d_knn_interp_weights_pack_f32(const float* q_points, const float* s_points, const int64_t* neighbor_indices,
                                    float* weights, int64_t M, int64_t N, int K, int k_used, float eps, int device,
                                    void* stream);
int v3d_rows_weighted_sum_f32(const float* feats, const int64_t* neighbor_indices, const float* weights, float* out,
                              int64_t M, int64_t N, int K, int k_used, int C, int device, void* stream);
int v3d_global_avgpool_pack_f32(const float* feats, const int64_t* lengths, float* pooled, int B, int C, int device,
                                void* stream);
/* Backward of the three gathers above, atomic-free (the reference's autograd scatters with float atomics).  The table is
 * inverted ONCE into a CSR over sources (v3d_rows_csr_build: stable sort of (source, position); `csr` = caller buffer of
 * v3d_rows_csr_bytes(N, L) bytes, 256-B aligned, L = number of table entries) and can be kept for every layer that
 * shares the table.  v3d_rows_scatter_f32 then writes source_grads (N,C) in full, each row summed in ascending table
 * position (bit-reproducible):
 *   mode 0: target_grads (L,C), optional weights (L)                -- index_select backward
 *   mode 1: target_grads (L/K, C), weights (L/K, k_used)            -- weighted-sum backward w.r.t. the features
 *   mode 2: target_grads (L/K, C), argk (L/K, C) from the forward   -- max-pool backward */
size_t v3d_rows_csr_bytes(int64_t N, int64_t L);
int v3d_rows_csr_build(const int64_t* indices, int64_t L, int64_t N, void* csr, size_t csr_bytes, int device,
                       void* stream);
int v3d_rows_scatter_f32(const float* target_grads, const void* csr, const float* weights, const uint8_t* argk, int mode,
                         float* source_grads, int64_t L, int64_t N, int K, int k_used, int C, int device, void* stream);

/* ---- "next" rows: fused grouping of the batch-mode layers ----------------------------------------
 * v3d_sa_group_f32 writes the input of a set-abstraction MLP in one pass (vision3d/layers/pointnet2.py:44-47:
 * group_gather(points), subtract the centroid, group_gather(feats), torch.cat): q_points (B,3,M), s_points (B,3,N),
 * s_feats (B,C,N) (NULL with C = 0), indices (B,M,K) -> grouped (B,C+3,M,K) with channels [0,C) = s_feats[b,c,idx] and
 * [C,C+3) = s_points[b,c,idx] - q_points[b,c,m].
 * v3d_edge_group_f32 writes the input of an EdgeConv MLP (layers/edge_conv.py:59-61): q_feats (B,C,M), s_feats (B,C,N),
 * indices (B,M,K) -> grouped (B,2C,M,K) with channels [0,C) = q_feats[b,c,m] and [C,2C) = s_feats[b,c,idx] -
 * q_feats[b,c,m].  Same single fp32 subtraction as the reference: bit-identical.  Indices outside [0,N) read as 0. */
int v3d_sa_group_f32(const float* q_points, const float* s_points, const float* s_feats, const int64_t* indices,
                     float* grouped, int B, int C, int N, int M, int K, int device, void* stream);
int v3d_edge_group_f32(const float* q_feats, const float* s_feats, const int64_t* indices, float* grouped, int B, int C,
                       int N, int M, int K, int device, void* stream);

/* ---- KPConv neighbourhood aggregation (fused gather + weighted sum) -------------------------------
 * replaces the body of KPConv.forward up to the convolution weights (vision3d/layers/kpconv.py:100-115 and :137-139):
 * q_points (M,3), s_points (N,3), s_feats (N,C), neighbor_indices (M,H) (entry N = none), kernel_points (K,3) ->
 *   weighted_feats[m,k,:] = sum_h max(0, 1 - |s_points[idx[m,h]] - q_points[m] - kernel_points[k]| / sigma)
 *                                 * s_feats[idx[m,h],:]                                            (M,K,C)
 *   neighbor_num[m] = max(1, #{h : sum_c s_feats[idx[m,h],c] > 0})                                 (M,) int64, optional
 * without the (M,H,3), (M,H,K,3), (M,H,K) and (M,H,C) intermediates.  H <= 64, K <= 16, C % 4 == 0, fp32-input MFMA
 * (exact products; summation order over h is the MFMA's).  workspace: v3d_kpconv_gather_workspace_bytes(N) bytes when
 * neighbor_num is requested (NULL = stream-ordered allocation). */
size_t v3d_kpconv_gather_workspace_bytes(int64_t N);
int v3d_kpconv_gather_f32(const float* q_points, const float* s_points, const float* s_feats,
                          const int64_t* neighbor_indices, const float* kernel_points, float* weighted_feats,
                          int64_t* neighbor_num, int64_t M, int64_t N, int H, int K, int C, float sigma, void* workspace,
                          size_t workspace_bytes, int device, void* stream);
/* Backward of the aggregation w.r.t. s_feats, first half: grad_neighbors (M,H,C) = influences^T x grad_weighted (M,K,C),
 * influences recomputed (the reference's autograd keeps the (M,K,H) weights and the (M,H,C) gather alive instead).  The
 * second half is v3d_rows_scatter_f32 (mode 0) over the neighbour table's CSR: atomic-free, ordered. */
int v3d_kpconv_gather_bwd_f32(const float* q_points, const float* s_points, const int64_t* neighbor_indices,
                              const float* kernel_points, const float* grad_weighted, float* grad_neighbors, int64_t M,
                              int64_t N, int H, int K, int C, float sigma, int device, void* stream);

/* ---- point-to-node partition (vision3d/ops/point_cloud_partition.py:40-104) ------------------------
 * points (N,3), nodes (M,3) fp32.  v3d_point_to_node_partition_f32: point_to_node (N) = nearest node of every point
 * (exact squared differences, lowest node index on ties; the reference takes the argmin of its expanded-form distance
 * matrix, :70-72), node_sizes (M) int64 (:75-79), node_masks (M) bool bytes (:82-84), *max_size = largest node size
 * (device scalar; the reference reads it back with .item(), :91) -- each of the last three may be NULL.  `workspace`
 * (v3d_point_to_node_workspace_bytes(N, M) bytes, 256-B aligned, REQUIRED) keeps the points ordered by (node, distance,
 * index) for v3d_point_to_node_knn, which writes node_knn_indices (M,K) = the K nearest own points of every node,
 * ascending, padded with N (:96-100), and node_knn_masks (M,K) (:99).  No (M,N) tensor is ever formed. */
size_t v3d_point_to_node_workspace_bytes(int64_t N, int64_t M);
int v3d_point_to_node_partition_f32(const float* points, const float* nodes, int64_t N, int64_t M,
                                    int64_t* point_to_node, int64_t* node_sizes, uint8_t* node_masks, int64_t* max_size,
                                    void* workspace, size_t workspace_bytes, int device, void* stream);
int v3d_point_to_node_knn(const void* workspace, int64_t N, int64_t M, int K, int64_t* node_knn_indices,
                          uint8_t* node_knn_masks, int device, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* V3D_HIP_H_ */
