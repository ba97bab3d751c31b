"""CPU restatements (NumPy) of the reference's pure-torch operators of SURVEY.md 8f ranks 2-4.

TEST INFRASTRUCTURE ONLY, like the rest of ``oracle/``: imported by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``; never by the product path.  Every function cites the reference lines it follows.
Pinned by fixtures generated from the reference's own Python (``tests/golden/make_golden.py``, ``ref_select.npz`` /
``ref_pack_ops.npz``).  Where the reference leaves an order unspecified (``torch.topk`` among equal scores) the rule is
"lower index first"; the fixtures are tie-free, so that rule is parity unpinned.
"""
import numpy as np


def _topk_mask(score, k, axis, largest):
    """Boolean mask of the k best entries along ``axis`` (ties: lower index first), mutual_topk_select.py:33-41."""
    key = -score if largest else score
    order = np.argsort(key + 0.0, axis=axis, kind="stable")
    top = np.take(order, np.arange(k), axis=axis)
    mask = np.zeros(score.shape, dtype=bool)
    np.put_along_axis(mask, top, True, axis=axis)
    return mask


def mutual_topk_corr_mat(score_mat, k, row_masks=None, col_masks=None, largest=True, threshold=None, mutual=True):
    """corr_mat (B,N,M) bool of batch_mutual_topk_select (mutual_topk_select.py:92-126)."""
    s = np.asarray(score_mat, dtype=np.float32)
    row = _topk_mask(s, k, 2, largest)  # :96-100
    col = _topk_mask(s, k, 1, largest)  # :103-107
    corr = np.logical_and(row, col) if mutual else np.logical_or(row, col)  # :110-113
    if threshold is not None:  # :116-121
        thr = np.float32(threshold)
        corr &= (s > thr) if largest else (s < thr)
    if row_masks is not None:  # :123-124
        corr &= np.asarray(row_masks, dtype=bool)[:, :, None]
    if col_masks is not None:  # :125-126
        corr &= np.asarray(col_masks, dtype=bool)[:, None, :]
    return corr


def mutual_topk_select(score_mat, k, **kw):
    """(batch, row, col, scores) in nonzero order (mutual_topk_select.py:128-131)."""
    s = np.asarray(score_mat, dtype=np.float32)
    corr = mutual_topk_corr_mat(s, k, **kw)
    b, r, c = np.nonzero(corr)
    return b.astype(np.int64), r.astype(np.int64), c.astype(np.int64), s[b, r, c]


def extract_correspondences_from_scores(score_mat, mutual=False, has_dustbin=False, threshold=0.0):
    """ops/matching.py:32-59: (rows, cols, exp scores) of the row / column maxima above the threshold."""
    s = np.exp(np.asarray(score_mat, dtype=np.float32))                                  # :32
    n, m = s.shape
    row_mat = np.zeros_like(s)
    row_mat[np.arange(n), s.argmax(axis=1)] = s.max(axis=1)                              # :35-38
    col_mat = np.zeros_like(s)
    col_mat[s.argmax(axis=0), np.arange(m)] = s.max(axis=0)                              # :41-44
    thr = np.float32(threshold)
    corr = (row_mat > thr) & (col_mat > thr) if mutual else (row_mat > thr) | (col_mat > thr)   # :39,:45,:47-50
    if has_dustbin:
        corr = corr[:-1, :-1]                                                            # :52-53
    r, c = np.nonzero(corr)
    return r.astype(np.int64), c.astype(np.int64), s[r, c]


# ---- rank 3: pack-mode row gathers ------------------------------------------------------------------------------
def extract_correspondences_from_scores_threshold(score_mat, threshold):
    """ops/matching.py:78-83: (rows, cols, exp scores) of the cells with exp(score) > threshold, nonzero order."""
    s = np.exp(np.asarray(score_mat, dtype=np.float32))                                  # :78
    r, c = np.nonzero(s > np.float32(threshold))                                         # :79-80
    return r.astype(np.int64), c.astype(np.int64), s[r, c]


def threshold_select(score_mat, threshold, largest=True):
    """The selection behind it without the exp: cells above (below) the threshold in nonzero order."""
    s = np.asarray(score_mat, dtype=np.float32)
    r, c = np.nonzero(s > np.float32(threshold) if largest else s < np.float32(threshold))
    return r.astype(np.int64), c.astype(np.int64), s[r, c]


def extract_correspondences_from_scores_topk(score_mat, k, has_dustbin=False, largest=True):
    """ops/matching.py:107-118: flat top-k (best first; ties: lower flat index, which torch leaves open), row = idx // M,
    col = idx % M, dustbin entries dropped afterwards."""
    s = np.asarray(score_mat, dtype=np.float32)
    n, m = s.shape
    if k > s.size:
        raise RuntimeError("selected index k out of range")
    flat = s.reshape(-1)
    order = np.argsort((-flat if largest else flat) + 0.0, kind="stable")[:k]            # :107
    r, c = order // m, order % m                                                         # :108-109
    if has_dustbin:                                                                      # :110-115
        keep = (r != n - 1) & (c != m - 1)
        r, c = r[keep], c[keep]
    return r.astype(np.int64), c.astype(np.int64), s[r, c]


def _padded(feats):
    feats = np.asarray(feats)
    return np.concatenate([feats, np.zeros_like(feats[:1])], axis=0)  # pooling.py:16, knn_interpolate.py:65-66


def _clamp(indices, n):
    idx = np.asarray(indices, dtype=np.int64)
    return np.where((idx >= 0) & (idx < n), idx, n)


def index_select_rows(feats, indices):
    """index_select(cat(feats, zeros), indices, dim=0) (index_select.py:25-33): (..., C)."""
    feats = np.asarray(feats)
    return _padded(feats)[_clamp(indices, feats.shape[0])]


def local_maxpool_pack(feats, neighbor_indices):
    """pooling.py:16-18 -> (pooled (M,C), argk (M,C) = first maximising column)."""
    g = index_select_rows(np.asarray(feats, dtype=np.float32), neighbor_indices)  # (M,K,C)
    return g.max(axis=1), g.argmax(axis=1).astype(np.uint8)


def nearest_interpolate_pack(inputs, neighbor_indices):
    """nearest_interpolate.py:18-20."""
    return index_select_rows(inputs, np.asarray(neighbor_indices)[:, 0])


def knn_interp_weights_pack(q_points, s_points, neighbor_indices, k=None, eps=1e-8):
    """knn_interpolate.py:67-73 in float32, one rounding per operation, sums in ascending k."""
    q = np.asarray(q_points, dtype=np.float32)
    s = _padded(np.asarray(s_points, dtype=np.float32))
    idx = np.asarray(neighbor_indices, dtype=np.int64)
    idx = idx if k is None else idx[:, :k]
    n = s.shape[0] - 1
    ci = _clamp(idx, n)
    d = q[:, None, :] - s[ci]
    sq = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    eps = np.float32(eps)
    w = np.where(ci != n, np.float32(1) / (sq + eps), np.float32(0)).astype(np.float32)
    total = np.zeros(w.shape[0], dtype=np.float32)
    for j in range(w.shape[1]):
        total = total + w[:, j]
    return (w / (total + eps)[:, None]).astype(np.float32)


def rows_weighted_sum(feats, neighbor_indices, weights):
    """knn_interpolate.py:74-75: sum_k w[m,k] * padded[idx[m,k],:], ascending k, float32."""
    feats = np.asarray(feats, dtype=np.float32)
    w = np.asarray(weights, dtype=np.float32)
    idx = np.asarray(neighbor_indices, dtype=np.int64)[:, : w.shape[1]]
    g = index_select_rows(feats, idx)
    acc = np.zeros((idx.shape[0], feats.shape[1]), dtype=np.float32)
    for j in range(w.shape[1]):
        acc = acc + g[:, j, :] * w[:, j : j + 1]
    return acc


def knn_interpolate_pack(q_points, s_points, s_feats, neighbor_indices, k=None, eps=1e-8):
    return rows_weighted_sum(s_feats, neighbor_indices, knn_interp_weights_pack(q_points, s_points, neighbor_indices, k, eps))


def global_avgpool_pack(feats, lengths):
    """pooling.py:22-39 (float64 accumulation: the order of torch.mean's sum is not part of the contract)."""
    feats = np.asarray(feats, dtype=np.float32)
    out, start = [], 0
    for n in np.asarray(lengths, dtype=np.int64):
        out.append(feats[start : start + n].astype(np.float64).mean(axis=0))
        start += n
    return np.stack(out).astype(np.float32)


def rows_scatter(target_grads, neighbor_indices, num_sources, mode=0, weights=None, argk=None):
    """Backward of the three gathers w.r.t. the features: every source row summed in ascending table position
    (np.add.at applies its updates sequentially, i.e. in exactly that order), float32."""
    idx = np.asarray(neighbor_indices, dtype=np.int64)
    g = np.asarray(target_grads, dtype=np.float32)
    n, c = int(num_sources), g.shape[-1]
    out = np.zeros((n + 1, c), dtype=np.float32)
    if mode == 0:
        contrib = g.reshape(-1, c)
        if weights is not None:
            contrib = contrib * np.asarray(weights, dtype=np.float32).reshape(-1, 1)
        np.add.at(out, _clamp(idx, n).reshape(-1), contrib)
    elif mode == 1:
        w = np.asarray(weights, dtype=np.float32)
        ku = w.shape[1]
        contrib = g[:, None, :] * w[:, :, None]
        np.add.at(out, _clamp(idx[:, :ku], n).reshape(-1), contrib.reshape(-1, c))
    else:
        m, k = idx.shape
        sel = np.asarray(argk)[:, None, :] == np.arange(k, dtype=np.int64)[None, :, None]  # (M,K,C)
        contrib = np.where(sel, g[:, None, :], np.float32(0))
        # adding an exact 0.0 leaves a float32 sum unchanged, so masked entries may take part in the ordered sum
        np.add.at(out, _clamp(idx, n).reshape(-1), contrib.reshape(-1, c))
    return out[:n]


# ---- rank 4: fused grouping of the batch-mode layers ------------------------------------------------------------
def _group_gather(src, idx):
    """out[b,c,m,k] = src[b,c,idx[b,m,k]] (group_gather_cuda.cuh:29-32)."""
    src, idx = np.asarray(src), np.asarray(idx, dtype=np.int64)
    b = np.arange(src.shape[0])[:, None, None, None]
    c = np.arange(src.shape[1])[None, :, None, None]
    return src[b, c, idx[:, None, :, :]]


def sa_group(q_points, s_points, s_feats, neighbor_indices):
    """layers/pointnet2.py:44-47: cat([group_gather(feats), group_gather(points) - q.unsqueeze(3)], dim=1)."""
    q = np.asarray(q_points, dtype=np.float32)
    offsets = _group_gather(np.asarray(s_points, dtype=np.float32), neighbor_indices) - q[:, :, :, None]
    if s_feats is None:
        return offsets
    return np.concatenate([_group_gather(np.asarray(s_feats, dtype=np.float32), neighbor_indices), offsets], axis=1)


def edge_group(q_feats, s_feats, neighbor_indices):
    """layers/edge_conv.py:59-61: cat([q.expand, group_gather(feats) - q.expand], dim=1)."""
    g = _group_gather(np.asarray(s_feats, dtype=np.float32), neighbor_indices)
    q = np.broadcast_to(np.asarray(q_feats, dtype=np.float32)[:, :, :, None], g.shape)
    return np.concatenate([q, g - q], axis=1)


# ---- KPConv aggregation (layers/kpconv.py:97-141) -------------------------------------------------------------
def kpconv_gather(q_points, s_points, s_feats, neighbor_indices, kernel_points, sigma, inf=1e6):
    """weighted_feats (M,K,C) float32 (accumulated in float64: the order of the reference's batched GEMM is not part
    of the contract) and neighbor_num (M,) int64, following kpconv.py:100-115 and :137-139 line by line."""
    q = np.asarray(q_points, dtype=np.float32)
    s = np.asarray(s_points, dtype=np.float32)
    f = np.asarray(s_feats, dtype=np.float32)
    kp = np.asarray(kernel_points, dtype=np.float32)
    n = s.shape[0]
    idx = _clamp(neighbor_indices, n)
    padded_points = np.concatenate([s, np.zeros_like(s[:1]) + np.float32(inf)], axis=0)      # :100
    neighbors = padded_points[idx] - q[:, None, :]                                             # :101-102
    differences = neighbors[:, :, None, :] - kp[None, None, :, :]                              # :105-106
    d2 = differences * differences
    sq = (d2[..., 0] + d2[..., 1]) + d2[..., 2]                                                # :107
    w = np.maximum(np.float32(1) - np.sqrt(sq) / np.float32(sigma), np.float32(0))             # :108
    w = np.transpose(w, (0, 2, 1))                                                             # :109  (M,K,H)
    neighbor_feats = _padded(f)[idx]                                                           # :112-113 (M,H,C)
    weighted = np.matmul(w.astype(np.float64), neighbor_feats.astype(np.float64))              # :114
    num = (neighbor_feats.sum(axis=-1, dtype=np.float64) > 0).sum(axis=-1)                     # :137-138
    return weighted.astype(np.float32), np.maximum(num, 1).astype(np.int64), w


# ---- point-to-node partition (ops/point_cloud_partition.py:40-104) ----------------------------------------------
def point_to_node_partition(points, nodes, point_limit=None):
    """(point_to_node (N,), node_sizes (M,), node_masks (M,), node_knn_indices (M,K), node_knn_masks (M,K)).

    Follows the reference's steps (:70-102) with two stated definitions where it leaves room: squared distances are the
    exact differences ((dx*dx + dy*dy) + dz*dz in float32) instead of the fp32 expansion of pairwise_distance, and
    ties go to the lower node / point index (torch.min / torch.topk: unspecified).  PARITY UNPINNED: the module imports
    pykeops at import time, so no reference-generated fixture exists; tests cross-check a torch restatement instead."""
    p = np.asarray(points, dtype=np.float32)
    nd = np.asarray(nodes, dtype=np.float32)
    n, m = p.shape[0], nd.shape[0]
    d = nd[:, None, :] - p[None, :, :]
    sq = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]  # (M, N)   :70
    point_to_node = np.argmin(sq, axis=0).astype(np.int64)                          # :72 (first minimum)
    node_sizes = np.bincount(point_to_node, minlength=m).astype(np.int64)           # :76-78
    node_masks = node_sizes > 0                                                     # :82-83
    k = int(node_sizes.max())                                                       # :91
    if point_limit is not None:
        k = min(k, int(point_limit))                                                # :92-93
    matching = np.zeros((m, n), dtype=bool)
    matching[point_to_node, np.arange(n)] = True                                    # :86-88
    masked = np.where(matching, sq, np.float32(np.inf))                             # :89
    order = np.argsort(masked, axis=1, kind="stable")[:, :k]                        # :96
    own = np.take_along_axis(matching, order, axis=1)                               # :97-99
    knn = np.where(own, order, n).astype(np.int64)                                  # :100
    return point_to_node, node_sizes, node_masks, knn, own


# ---- the reference extension's CPU node sampling (csrc/cpu/node_sampling/node_sampling.cpp:10-77) --------------
def sample_nodes_with_fps(points, min_distance, num_samples):
    """Line-by-line restatement incl. the swap-and-pop traversal (float32 sqrt distances, strict '>' / '<')."""
    pts = np.asarray(points, dtype=np.float32)
    n = pts.shape[0]
    selected = []
    unselected = list(range(n))
    num_unselected = n
    distances = np.full(n, np.float32(np.inf), dtype=np.float32)  # 1e40 overflows float to +inf (:33-35)
    best_index = 0
    while True:
        cur = unselected[best_index]
        selected.append(cur)
        unselected[best_index], unselected[num_unselected - 1] = unselected[num_unselected - 1], unselected[best_index]
        num_unselected -= 1                                                                   # :5-8, :43
        if num_samples > 0 and len(selected) >= num_samples:
            break                                                                             # :45
        best_index, best_distance, index = -1, np.float32(0), 0
        while index < num_unselected:
            pi = unselected[index]
            d = pts[cur] - pts[pi]
            dist = np.sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2], dtype=np.float32)        # :52
            dist = min(distances[pi], dist)
            distances[pi] = dist
            if dist > best_distance:
                best_index, best_distance = index, dist                                       # :56-59
            if dist < np.float32(min_distance):
                unselected[index], unselected[num_unselected - 1] = unselected[num_unselected - 1], unselected[index]
                num_unselected -= 1                                                           # :61-62
            else:
                index += 1
        if best_distance < np.float32(min_distance) or best_index < 0:
            break                                                                             # :68
    return np.asarray(selected, dtype=np.int64)


def voxelize_pack(points, lengths, voxel_size, feats=None):
    """ops/voxelize.py:20-63: voxels = floor(points / voxel_size) (fp32 division), one output per occupied (cloud, voxel)
    in ascending (cloud, vx, vy, vz) order = torch.unique of the hash (:37-38); means of the points (:46-49) and of the
    optional features (:57-60), summed in input order (what a sequential scatter_add_ does); per-cloud counts (:51-54)."""
    pts = np.asarray(points, dtype=np.float32)
    lengths = np.asarray(lengths, dtype=np.int64)
    vox = np.floor(pts / np.float32(voxel_size)).astype(np.int64)                          # :22
    cloud = np.repeat(np.arange(len(lengths)), lengths)                                    # :33-35
    keys = np.concatenate([cloud[:, None], vox], axis=1)
    uniq, inv, counts = np.unique(keys, axis=0, return_inverse=True, return_counts=True)   # :37-38
    inv = inv.reshape(-1)
    nodes = np.zeros((len(uniq), 3), dtype=np.float32)
    np.add.at(nodes, inv, pts)                                                             # :48 (sequential, input order)
    nodes /= counts[:, None].astype(np.float32)                                            # :49
    v_lengths = np.bincount(uniq[:, 0], minlength=len(lengths)).astype(np.int64)           # :51-54
    out = [uniq[:, 1:].astype(np.int64), nodes]
    if feats is not None:
        f = np.asarray(feats, dtype=np.float32)
        v_feats = np.zeros((len(uniq), f.shape[1]), dtype=np.float32)
        np.add.at(v_feats, inv, f)                                                         # :59
        v_feats /= counts[:, None].astype(np.float32)                                      # :60
        out.append(v_feats)
    return tuple(out) + (v_lengths,)


def radius_count_pack(q_points, s_points, q_lengths, s_lengths, radius):
    """ops/radius_search.py:74-107: per query, the number of supports of its own cloud with radius - d > 0, where
    d = sqrt((dx*dx + dy*dy) + dz*dz) in fp32 (the padded rows of the reference sit at 1e10 and never count)."""
    q = np.asarray(q_points, np.float32)
    s = np.asarray(s_points, np.float32)
    r = np.float32(radius)
    out = np.zeros((q.shape[0],), np.int64)
    qo = so = 0
    for ql, sl in zip(np.asarray(q_lengths).tolist(), np.asarray(s_lengths).tolist()):
        cq, cs = q[qo:qo + ql], s[so:so + sl]
        for a in range(0, ql, 1024):
            blk = cq[a:a + 1024, None, :] - cs[None, :, :]  # fp32
            d2 = (blk[..., 0] * blk[..., 0] + blk[..., 1] * blk[..., 1]) + blk[..., 2] * blk[..., 2]
            with np.errstate(invalid="ignore"):
                out[qo + a:qo + a + blk.shape[0]] = ((r - np.sqrt(d2)) > 0).sum(1)
        qo += ql
        so += sl
    return out

