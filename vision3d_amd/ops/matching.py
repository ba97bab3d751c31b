"""Correspondence extraction -- mirrors ``vision3d/ops/matching.py`` (all five functions, :10-151).

The reference materialises the (M, N) feature-distance matrix (``pairwise_distance``) and runs ``mutual_topk_select``
with k = 1 on it.  Here the matrix never exists: the nearest neighbour of every source descriptor among the targets and
of every target among the sources come from the fused MFMA distance + top-k kernel (``ext.knn_feat``, distance chunks
stay in the Infinity Cache), and ``ext.mutual_select_lists`` intersects / unites the two lists.  That is what makes
scene-scale matching (100k x 100k descriptors, BASELINE config 5) possible at all: the matrix would be 40 GB.
"""
import torch
from torch import Tensor

from .. import ext
from .mutual_topk_select import mutual_topk_select


@torch.no_grad()
def extract_correspondences_from_scores(
    score_mat: Tensor,
    mutual: bool = False,
    has_dustbin: bool = False,
    threshold: float = 0.0,
    return_score: bool = False,
):
    """Max selection on a log-probability matrix (matching.py:10-61): entries that are the maximum of their row and /
    or of their column with ``exp(score) > threshold``; (N+1, M+1) matrices with a dustbin drop their last row / column.

    One elementwise ``exp`` plus the native selection (``ext.mutual_topk_select``, k = 1) replace the reference's two
    ``torch.max``, four matrix-sized scatters / comparisons and ``nonzero``.
    """
    score_mat = torch.exp(score_mat)  # :32
    rows, cols, scores = mutual_topk_select(score_mat, 1, largest=True, threshold=threshold, mutual=mutual)
    if has_dustbin:  # :52-53
        keep = (rows != score_mat.shape[0] - 1) & (cols != score_mat.shape[1] - 1)
        rows, cols, scores = rows[keep], cols[keep], scores[keep]
    if return_score:
        return rows, cols, scores
    return rows, cols


@torch.no_grad()
def extract_correspondences_from_scores_threshold(score_mat: Tensor, threshold: float, return_score: bool = False):
    """Threshold selection on a log-probability matrix (matching.py:64-86): the cells with ``exp(score) > threshold`` in
    ``nonzero`` order.  One elementwise ``exp`` plus ``ext.threshold_select`` (count pass, emit pass; no boolean matrix,
    no ``nonzero``); the one host synchronisation is reading the count, which ``torch.nonzero`` needs as well."""
    if not score_mat.is_cuda:
        raise RuntimeError("extract_correspondences_from_scores_threshold: CUDA tensor required (no CPU fallback)")
    score = torch.exp(score_mat).float().contiguous()  # :78
    dev = score.device
    count = torch.zeros((), dtype=torch.long, device=dev)
    cap = max(min(score.numel(), 1 << 20), 1)  # every matrix of up to 1 Mi cells is one call; beyond, re-run on overflow
    while True:
        idx = torch.empty((2, cap), dtype=torch.long, device=dev)
        scores = torch.empty((cap,), device=dev) if return_score else None
        ext.threshold_select(score, threshold, out_row=idx[0], out_col=idx[1], out_score=scores, num_out=count)
        c = int(count.item())
        if c <= cap:
            break
        cap = c
    if return_score:
        return idx[0, :c], idx[1, :c], scores[:c].to(score_mat.dtype)
    return idx[0, :c], idx[1, :c]


@torch.no_grad()
def extract_correspondences_from_scores_topk(
    score_mat: Tensor, k: int, has_dustbin: bool = False, largest: bool = True, return_score: bool = False
):
    """Global top-k selection (matching.py:89-121): the k best cells of the flattened matrix, best first, entries of the
    dustbin row / column dropped afterwards.  ``ext.global_topk_select`` = exact radix select + sort of the k winners;
    equal scores are taken by ascending flat index (``torch.topk`` leaves that open).  Host synchronisation only with
    ``has_dustbin`` (the reference's boolean-mask indexing has it too)."""
    if not score_mat.is_cuda:
        raise RuntimeError("extract_correspondences_from_scores_topk: CUDA tensor required (no CPU fallback)")
    score = score_mat.float().contiguous()
    if k > score.numel():
        raise RuntimeError("selected index k out of range")  # torch.topk's message
    dev = score.device
    idx = torch.empty((2, max(k, 1)), dtype=torch.long, device=dev)
    scores = torch.empty((max(k, 1),), device=dev) if return_score else None
    count = torch.zeros((), dtype=torch.long, device=dev) if has_dustbin else None
    ext.global_topk_select(score, k, out_row=idx[0], out_col=idx[1], out_score=scores, num_out=count, largest=largest,
                           has_dustbin=has_dustbin)
    c = int(count.item()) if has_dustbin else k
    if return_score:
        return idx[0, :c], idx[1, :c], scores[:c].to(score_mat.dtype)
    return idx[0, :c], idx[1, :c]


@torch.no_grad()
def extract_correspondences_from_feats(
    src_feats: Tensor, tgt_feats: Tensor, mutual: bool = False, return_feat_dist: bool = False, normalized: bool = True
):
    """src_feats (M, C), tgt_feats (N, C) -> (src_corr_indices, tgt_corr_indices[, corr_feat_dists]).

    ``corr_feat_dists`` are SQUARED feature distances like the reference's ``pairwise_distance(..., squared=True)``;
    they come from the exact difference form (sum_c (x_c - y_c)^2), so they agree with the reference's expansion
    (2 - 2 x.y, or x^2 - 2 x.y + y^2) to its cancellation error, not bit for bit.
    """
    if not (src_feats.is_cuda and tgt_feats.is_cuda):
        raise RuntimeError("extract_correspondences_from_feats: CUDA tensors required (no CPU fallback)")
    x = src_feats.float().contiguous().unsqueeze(0)
    y = tgt_feats.float().contiguous().unsqueeze(0)
    m, n = x.shape[1], y.shape[1]
    dev = x.device
    row_d = torch.empty((1, m, 1), device=dev)
    row_i = torch.empty((1, m, 1), dtype=torch.long, device=dev)
    col_d = torch.empty((1, n, 1), device=dev)
    col_i = torch.empty((1, n, 1), dtype=torch.long, device=dev)
    ext.knn_feat(x, y, row_d, row_i, 1)
    ext.knn_feat(y, x, col_d, col_i, 1)
    bound = max(min(m, n) if mutual else m + n, 1)
    idx = torch.empty((2, bound), dtype=torch.long, device=dev)
    dist = torch.empty((bound,), device=dev)
    count = torch.zeros((), dtype=torch.long, device=dev)
    ext.mutual_select_lists(row_i, row_d, col_i, col_d, n, out_row=idx[0], out_col=idx[1], out_score=dist,
                            num_out=count, largest=False, mutual=mutual)
    c = int(count.item())
    if return_feat_dist:
        d = dist[:c]
        return idx[0, :c], idx[1, :c], (d * d).to(src_feats.dtype)
    return idx[0, :c], idx[1, :c]
