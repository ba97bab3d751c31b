"""Mutual top-k selection -- mirrors ``vision3d/ops/mutual_topk_select.py:7-133`` (same signatures and returns).

The reference runs two ``torch.topk`` passes, scatters two boolean matrices, combines / thresholds / masks them in
more matrix-sized passes and ends with ``torch.nonzero``.  Here one C-ABI call (``ext.mutual_topk_select``) reads the
score matrix twice and compacts the <= k(N+M) candidates; the only host synchronisation is reading the number of
correspondences, which ``torch.nonzero`` needs as well.  Among EQUAL scores the lower index is preferred
(``torch.topk`` leaves that order unspecified).
"""
from typing import Optional

import torch
from torch import Tensor

from .. import ext


def _select(score_mat: Tensor, k: int, row_masks, col_masks, largest, threshold, mutual, reduce_result):
    if not score_mat.is_cuda:
        raise RuntimeError("mutual_topk_select: score_mat must be a CUDA tensor (no CPU fallback in vision3d_amd)")
    score = score_mat.float().contiguous()
    B, N, M = score.shape
    if k > N or k > M:
        raise RuntimeError("selected index k out of range")  # torch.topk's message
    if row_masks is not None:
        row_masks = row_masks.to(torch.bool).contiguous()
    if col_masks is not None:
        col_masks = col_masks.to(torch.bool).contiguous()
    if not reduce_result:
        corr_mat = torch.empty((B, N, M), dtype=torch.bool, device=score.device)
        ext.mutual_topk_select(score, k, corr_mat=corr_mat, row_masks=row_masks, col_masks=col_masks, largest=largest,
                               threshold=threshold, mutual=mutual)
        return corr_mat
    bound = B * k * (min(N, M) if mutual else N + M)
    idx = torch.empty((3, max(bound, 1)), dtype=torch.long, device=score.device)
    scores = torch.empty((max(bound, 1),), dtype=torch.float32, device=score.device)
    count = torch.zeros((), dtype=torch.long, device=score.device)
    ext.mutual_topk_select(score, k, out_batch=idx[0], out_row=idx[1], out_col=idx[2], out_score=scores, num_out=count,
                           row_masks=row_masks, col_masks=col_masks, largest=largest, threshold=threshold,
                           mutual=mutual)
    n = int(count.item())  # the one sync torch.nonzero has too
    if score_mat.requires_grad and torch.is_grad_enabled():
        # the reference returns score_mat[rows, cols] (mutual_topk_select.py:57,:130), which autograd follows; the
        # kernel's copy of the same values is detached, so gather them through torch when a gradient can be asked for
        return idx[0, :n], idx[1, :n], idx[2, :n], score_mat[idx[0, :n], idx[1, :n], idx[2, :n]]
    return idx[0, :n], idx[1, :n], idx[2, :n], scores[:n].to(score_mat.dtype)


def mutual_topk_select(
    score_mat: Tensor,
    k: int,
    largest: bool = True,
    threshold: Optional[float] = None,
    mutual: bool = True,
    reduce_result: bool = True,
):
    """score_mat (N, M) -> (row_corr_indices, col_corr_indices, corr_scores) or corr_mat (N, M) bool."""
    out = _select(score_mat.unsqueeze(0), k, None, None, largest, threshold, mutual, reduce_result)
    if reduce_result:
        return out[1], out[2], out[3]
    return out[0]


def batch_mutual_topk_select(
    score_mat: Tensor,
    k: int,
    row_masks: Optional[Tensor] = None,
    col_masks: Optional[Tensor] = None,
    largest: bool = True,
    threshold: Optional[float] = None,
    mutual: bool = True,
    reduce_result: bool = True,
):
    """score_mat (B, N, M) -> (batch, row, col indices, scores) or corr_mat (B, N, M) bool."""
    return _select(score_mat, k, row_masks, col_masks, largest, threshold, mutual, reduce_result)
