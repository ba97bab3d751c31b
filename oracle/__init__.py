"""NumPy front-end of the CPU oracle (``oracle/v3d_oracle.c``).

TEST INFRASTRUCTURE ONLY: imported by ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg -- never by ``vision3d_amd`` (the product path fails loudly without its HIP library).

Every function takes/returns C-contiguous NumPy arrays with the layouts of the reference's ``vision3d.ext``
entry points (``/root/reference/vision3d/csrc/pybind.cpp:72-158``) and allocates its outputs the way the
reference's Python wrappers do (zeros for indices, 1e10 for the FPS scratch distances).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libv3d_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile ``libv3d_oracle.so`` with gcc (see ``oracle/Makefile``)."""
    src = os.path.join(_HERE, "v3d_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B" if force else "-s"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.v3d_oracle_fps_block.restype = ctypes.c_int
        _lib.v3d_oracle_knn_points_f32.restype = ctypes.c_int
        _lib.v3d_oracle_knn_points_f64.restype = ctypes.c_int
    return _lib


def set_threads(n: int) -> None:
    lib().v3d_oracle_set_threads(ctypes.c_int(int(n)))


def max_threads() -> int:
    return int(lib().v3d_oracle_max_threads())


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype=None):
    return np.ascontiguousarray(a, dtype=dtype)


def _sfx(dtype):
    if dtype == np.float32:
        return "f32"
    if dtype == np.float64:
        return "f64"
    raise TypeError(f"oracle: unsupported dtype {dtype}")


def fps_block(n: int) -> int:
    return int(lib().v3d_oracle_fps_block(ctypes.c_int(int(n))))


def furthest_point_sample(points, num_samples, distances=None):
    """points (B,N,3) f32 -> indices (B,M) i64.  ``distances`` (B,N) is the in/out scratch (1e10 default)."""
    points = _c(points, np.float32)
    B, N, _ = points.shape
    if distances is None:
        distances = np.full((B, N), 1e10, dtype=np.float32)
    indices = np.zeros((B, num_samples), dtype=np.int64)
    lib().v3d_oracle_fps_f32(_p(points), _p(distances), _p(indices), B, N, int(num_samples))
    return indices


def ball_query(q_points, s_points, num_samples, radius):
    q_points = _c(q_points)
    s_points = _c(s_points, q_points.dtype)
    B, M, _ = q_points.shape
    N = s_points.shape[1]
    indices = np.zeros((B, M, num_samples), dtype=np.int64)
    fn = getattr(lib(), "v3d_oracle_ball_query_" + _sfx(q_points.dtype))
    fn(_p(q_points), _p(s_points), _p(indices), B, M, N, ctypes.c_float(radius), int(num_samples))
    return indices


def knn_points(q_points, s_points, k):
    q_points = _c(q_points)
    s_points = _c(s_points, q_points.dtype)
    B, M, _ = q_points.shape
    N = s_points.shape[1]
    dist = np.zeros((B, M, k), dtype=q_points.dtype)
    idx = np.zeros((B, M, k), dtype=np.int64)
    fn = getattr(lib(), "v3d_oracle_knn_points_" + _sfx(q_points.dtype))
    rc = fn(_p(q_points), _p(s_points), _p(dist), _p(idx), B, M, N, int(k))
    if rc != 0:
        raise ValueError("knn_points: k must be in [0, 100]")
    return dist, idx


def three_nn(q_points, s_points):
    q_points = _c(q_points)
    s_points = _c(s_points, q_points.dtype)
    B, Q, _ = q_points.shape
    S = s_points.shape[1]
    dist = np.zeros((B, Q, 3), dtype=q_points.dtype)
    idx = np.zeros((B, Q, 3), dtype=np.int64)
    fn = getattr(lib(), "v3d_oracle_three_nn_" + _sfx(q_points.dtype))
    fn(_p(q_points), _p(s_points), _p(dist), _p(idx), B, Q, S)
    return dist, idx


def three_interpolate_forward(sources, indices, weights):
    sources = _c(sources)
    indices = _c(indices, np.int64)
    weights = _c(weights, sources.dtype)
    B, C, S = sources.shape
    Q = indices.shape[1]
    out = np.zeros((B, C, Q), dtype=sources.dtype)
    fn = getattr(lib(), "v3d_oracle_three_interp_fwd_" + _sfx(sources.dtype))
    fn(_p(sources), _p(indices), _p(weights), _p(out), B, C, S, Q)
    return out


def three_interpolate_backward(target_grads, indices, weights, num_sources):
    target_grads = _c(target_grads)
    indices = _c(indices, np.int64)
    weights = _c(weights, target_grads.dtype)
    B, C, Q = target_grads.shape
    out = np.zeros((B, C, num_sources), dtype=target_grads.dtype)
    fn = getattr(lib(), "v3d_oracle_three_interp_bwd_" + _sfx(target_grads.dtype))
    fn(_p(target_grads), _p(indices), _p(weights), _p(out), B, C, int(num_sources), Q)
    return out


def group_gather_forward(sources, indices):
    """sources (B,C,N) any dtype, indices (B,M,K) or (B,M) -> (B,C,M,K) or (B,C,M)."""
    sources = _c(sources)
    indices = _c(indices, np.int64)
    B, C, N = sources.shape
    L = int(np.prod(indices.shape[1:]))
    out = np.zeros((B, C) + indices.shape[1:], dtype=sources.dtype)
    lib().v3d_oracle_group_gather_fwd(
        _p(sources), _p(indices), _p(out), B, C, N, ctypes.c_long(L), int(sources.dtype.itemsize)
    )
    return out


def group_gather_backward(target_grads, indices, num_sources):
    target_grads = _c(target_grads)
    indices = _c(indices, np.int64)
    B, C = target_grads.shape[:2]
    L = int(np.prod(indices.shape[1:]))
    out = np.zeros((B, C, num_sources), dtype=target_grads.dtype)
    fn = getattr(lib(), "v3d_oracle_group_gather_bwd_" + _sfx(target_grads.dtype))
    fn(_p(target_grads), _p(indices), _p(out), B, C, int(num_sources), ctypes.c_long(L))
    return out


gather_forward = group_gather_forward
gather_backward = group_gather_backward


def pairwise_distance(x, y, normalized=False, transposed=False, squared=True, strict=False, eps=1e-8):
    """Restates ``ops/pairwise_distance.py:5-58`` (fp32).  ``strict`` follows :28-37 in NumPy."""
    x = np.asarray(x, dtype=np.float32)
    y = np.asarray(y, dtype=np.float32)
    if transposed:
        x = np.swapaxes(x, -1, -2)
        y = np.swapaxes(y, -1, -2)
    lead = x.shape[:-2]
    N, C = x.shape[-2:]
    M = y.shape[-2]
    if strict:
        diff = x[..., :, None, :] - y[..., None, :, :]
        d = np.sqrt(np.sum(diff.astype(np.float32) ** 2, axis=-1, dtype=np.float32))
        return (d * d if squared else d).astype(np.float32)
    xb = _c(x.reshape((-1, N, C)))
    yb = _c(y.reshape((-1, M, C)))
    out = np.zeros((xb.shape[0], N, M), dtype=np.float32)
    lib().v3d_oracle_pairwise_distance_f32(
        _p(xb), _p(yb), _p(out), xb.shape[0], N, M, C, int(bool(normalized)), int(bool(squared)),
        ctypes.c_float(eps),
    )
    return out.reshape(lead + (N, M))


def knn_feat(x, y, k):
    """Feature-space KNN (ops/knn.py:10-27 call-site semantics): (dist (…,N,k) f32, idx (…,N,k) i64)."""
    x = np.asarray(x, dtype=np.float32)
    y = np.asarray(y, dtype=np.float32)
    lead = x.shape[:-2]
    N, C = x.shape[-2:]
    M = y.shape[-2]
    xb = _c(x.reshape((-1, N, C)))
    yb = _c(y.reshape((-1, M, C)))
    dist = np.zeros((xb.shape[0], N, k), dtype=np.float32)
    idx = np.zeros((xb.shape[0], N, k), dtype=np.int64)
    lib().v3d_oracle_knn_feat_f32(_p(xb), _p(yb), _p(dist), _p(idx), xb.shape[0], N, M, C, int(k))
    return dist.reshape(lead + (N, k)), idx.reshape(lead + (N, k))


def knn_pack(q_points, s_points, q_lengths, s_lengths, k):
    """Pack-mode kNN (ops/knn.py:110-145): (dist (total_q, k) f32, idx (total_q, k) i64 inside the cloud), equal
    distances lower index first (pykeops' insertion order; parity unpinned)."""
    q = _c(q_points, np.float32)
    s = _c(s_points, np.float32)
    ql = _c(q_lengths, np.int64)
    sl = _c(s_lengths, np.int64)
    dist = np.zeros((q.shape[0], int(k)), dtype=np.float32)
    idx = np.zeros((q.shape[0], int(k)), dtype=np.int64)
    lib().v3d_oracle_knn_pack_f32(_p(q), _p(s), _p(ql), _p(sl), _p(dist), _p(idx), int(ql.shape[0]), int(k))
    return dist, idx


def radius_search_pack(q_points, s_points, q_lengths, s_lengths, radius, neighbor_limit):
    """Pack-mode radius search (ops/radius_search.py:44-71): (total_q, limit) i64, absent = total_s."""
    q = _c(q_points, np.float32)
    s = _c(s_points, np.float32)
    ql = _c(q_lengths, np.int64)
    sl = _c(s_lengths, np.int64)
    out = np.zeros((q.shape[0], int(neighbor_limit)), dtype=np.int64)
    lib().v3d_oracle_radius_search_pack_f32(
        _p(q), _p(s), _p(ql), _p(sl), _p(out), int(ql.shape[0]), ctypes.c_int64(s.shape[0]), ctypes.c_float(radius),
        int(neighbor_limit),
    )
    return out


def grid_subsample_pack(points, lengths, voxel_size):
    """Pack-mode grid subsampling (ops/grid_subsample.py:49-99): (sampled_points (M,3) f32, sampled_lengths (B,) i64)."""
    p = _c(points, np.float32)
    ln = _c(lengths, np.int64)
    out = np.zeros((max(p.shape[0], 1), 3), dtype=np.float32)
    out_len = np.zeros((ln.shape[0],), dtype=np.int64)
    fn = lib().v3d_oracle_grid_subsample_pack_f32
    fn.restype = ctypes.c_int64
    m = fn(_p(p), _p(ln), int(ln.shape[0]), ctypes.c_float(voxel_size), _p(out), _p(out_len))
    return out[:m].copy(), out_len


from .next_rows import (edge_group, extract_correspondences_from_scores, extract_correspondences_from_scores_threshold,
                        extract_correspondences_from_scores_topk, radius_count_pack, threshold_select, voxelize_pack, kpconv_gather, sa_group, global_avgpool_pack, index_select_rows, knn_interp_weights_pack,  # noqa: E402,F401
                        knn_interpolate_pack, local_maxpool_pack, mutual_topk_corr_mat, mutual_topk_select, point_to_node_partition,
                        nearest_interpolate_pack, rows_scatter, rows_weighted_sum, sample_nodes_with_fps)  # SURVEY 8f ranks 2-4, NumPy
