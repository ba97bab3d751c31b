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
