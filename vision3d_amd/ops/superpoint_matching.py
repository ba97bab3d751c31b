"""Superpoint matching heads -- mirror ``vision3d/models/geotransformer/superpoint_matching_global_topk.py:10-83`` and
``superpoint_matching_mutual_topk.py:7-78`` (same constructor arguments, ``forward`` signatures and returns).

They are listed with the selectors in SURVEY 8f rank 2 because they are nothing but compositions of them:
``pairwise_distance`` (native, fp32 MFMA) -> elementwise score shaping (``exp``, dual normalisation, weights: left to
torch, a handful of passes over a superpoint-sized matrix) -> the native selection (``ext.global_topk_select`` for the
flat top-k, ``ops.mutual_topk_select`` for the mutual top-k).  The modules have no parameters.
"""
from typing import Optional

import torch
import torch.nn as nn
from torch import Tensor

from .. import ext
from .mutual_topk_select import mutual_topk_select
from .pairwise_distance import pairwise_distance


def _drop_empty(feats, masks, weights=None):
    """Keep the rows whose mask is True (both heads start by removing empty patches)."""
    if masks is None:
        return feats, weights
    return feats[masks], (weights[masks] if weights is not None else None)


def _restore(indices, masks):
    """Positions in the compacted set -> positions in the original one."""
    if masks is None:
        return indices
    return torch.nonzero(masks, as_tuple=True)[0][indices]


def superpoint_matching_global_topk(src_feats, tgt_feats, num_correspondences, dual_normalization=True, src_masks=None,
                                    tgt_masks=None, src_weights=None, tgt_weights=None):
    """global_topk.py:40-75 as a function: exp(-d) scores, optional dual normalisation and weights, flat top-k."""
    src_feats, src_weights = _drop_empty(src_feats, src_masks, src_weights)  # :41-49
    tgt_feats, tgt_weights = _drop_empty(tgt_feats, tgt_masks, tgt_weights)
    scores = torch.exp(-pairwise_distance(src_feats, tgt_feats, normalized=True))  # :52
    if dual_normalization:  # :53-56, same operation order: (s / row sums) * (s / column sums)
        scores = (scores / scores.sum(dim=1, keepdim=True)) * (scores / scores.sum(dim=0, keepdim=True))
    if src_weights is not None:  # :57-58
        scores = scores * src_weights.unsqueeze(1)
    if tgt_weights is not None:  # :59-60
        scores = scores * tgt_weights.unsqueeze(0)
    count = min(num_correspondences, scores.numel())  # :61
    # :62-64 view(-1).topk, div, mod -> ONE native selection that writes (row, col, score) directly
    mat = scores.float().contiguous()
    out = torch.empty((2, max(count, 1)), dtype=torch.long, device=mat.device)
    best = torch.empty((max(count, 1),), device=mat.device)
    ext.global_topk_select(mat, count, out_row=out[0], out_col=out[1], out_score=best, largest=True)
    return _restore(out[0, :count], src_masks), _restore(out[1, :count], tgt_masks), best[:count].to(scores.dtype)


def superpoint_matching_mutual_topk(src_feats, tgt_feats, num_correspondences, k, threshold=None, mutual=True, eps=1e-8,
                                    src_masks=None, tgt_masks=None):
    """mutual_topk.py:30-68 as a function: sqrt(d + eps) scores, mutual top-k, optional threshold with top-k fallback."""
    src_feats, _ = _drop_empty(src_feats, src_masks)  # :31-35
    tgt_feats, _ = _drop_empty(tgt_feats, tgt_masks)
    score_mat = torch.sqrt(pairwise_distance(src_feats, tgt_feats, normalized=True) + eps)  # :38
    src_idx, tgt_idx, corr_scores = mutual_topk_select(score_mat, k, largest=False, threshold=None, mutual=mutual)  # :40
    if threshold is not None:  # :45-57 -- on the <= k(N+M) candidates only
        wanted = min(num_correspondences, corr_scores.numel())
        good = corr_scores <= threshold
        if int(good.sum().item()) < wanted:  # not enough good ones: the `wanted` smallest instead
            corr_scores, pick = corr_scores.topk(k=wanted, largest=False)
        else:
            pick = torch.nonzero(good, as_tuple=True)[0]
            corr_scores = corr_scores[pick]
        src_idx, tgt_idx = src_idx[pick], tgt_idx[pick]
    return _restore(src_idx, src_masks), _restore(tgt_idx, tgt_masks), corr_scores  # :60-68


class SuperPointMatchingGlobalTopk(nn.Module):
    """Same constructor and forward arguments as the reference module (global_topk.py:11-24); no parameters."""

    def __init__(self, num_correspondences, dual_normalization=True):
        super().__init__()
        self.num_correspondences = num_correspondences
        self.dual_normalization = dual_normalization

    def forward(self, src_feats: Tensor, tgt_feats: Tensor, src_masks: Optional[Tensor] = None,
                tgt_masks: Optional[Tensor] = None, src_weights: Optional[Tensor] = None,
                tgt_weights: Optional[Tensor] = None):
        return superpoint_matching_global_topk(src_feats, tgt_feats, self.num_correspondences, self.dual_normalization,
                                               src_masks, tgt_masks, src_weights, tgt_weights)

    def extra_repr(self) -> str:
        return f"num_correspondences={self.num_correspondences}, dual_normalization={self.dual_normalization}"


class SuperPointMatchingMutualTopk(nn.Module):
    """Same constructor and forward arguments as the reference module (mutual_topk.py:8-16); no parameters."""

    def __init__(self, num_correspondences, k, threshold=None, mutual=True, eps=1e-8):
        super().__init__()
        self.num_correspondences = num_correspondences
        self.k = k
        self.threshold = threshold
        self.mutual = mutual
        self.eps = eps

    def forward(self, src_feats, tgt_feats, src_masks=None, tgt_masks=None):
        return superpoint_matching_mutual_topk(src_feats, tgt_feats, self.num_correspondences, self.k, self.threshold,
                                               self.mutual, self.eps, src_masks, tgt_masks)

    def extra_repr(self) -> str:
        return (f"num_correspondences={self.num_correspondences}, k={self.k}, threshold={self.threshold}, "
                f"mutual={self.mutual}")
