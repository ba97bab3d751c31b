"""Radius search in pack mode -- mirrors ``vision3d/ops/radius_search.py:44-107``.

Same definition as the reference: the ``neighbor_limit`` nearest supports of each query inside its own cloud,
those farther than ``radius`` replaced by the sentinel index N_total (= s_points.shape[0]).
"""
import torch

from .. import ext
from .conversion import batch_to_pack, pack_to_batch
from .knn import keops_knn


def _device_lengths(lengths, like):
    """The kernels read the lengths on the device as int64; the reference's loops accept them on either side and in any
    integer type (`lengths[i].item()`, ops/conversion.py:7-12)."""
    return lengths.to(device=like.device, dtype=torch.long).contiguous()


def radius_search_pack_mode(q_points, s_points, q_lengths, s_lengths, radius, neighbor_limit, inf=1e10):
    """q_points (M, 3), s_points (N, 3), lengths (B,) -> neighbor indices (M, neighbor_limit), N where absent.

    Native pack-mode kernel (uniform-grid cell list, no padding, no host synchronisation).  fp32 3-D points with
    ``neighbor_limit <= 512``; anything else takes the reference's route (kNN over padded batches + mask).  ``inf`` is
    only used by that route."""
    if (q_points.dtype == torch.float32 and s_points.dtype == torch.float32 and q_points.shape[-1] == 3
            and 0 < neighbor_limit <= 512):
        out = torch.empty((q_points.shape[0], neighbor_limit), dtype=torch.long, device=q_points.device)
        ext.radius_search_pack(q_points.contiguous(), s_points.contiguous(), _device_lengths(q_lengths, q_points),
                               _device_lengths(s_lengths, q_points), out, radius, neighbor_limit)
        return out
    return padded_radius_search_pack_mode(q_points, s_points, q_lengths, s_lengths, radius, neighbor_limit, inf)


def padded_radius_search_pack_mode(q_points, s_points, q_lengths, s_lengths, radius, neighbor_limit, inf=1e10):
    """The reference's formulation (radius_search.py:58-70): kNN over padded batches, then the distance mask."""
    batch_q_points, batch_q_masks = pack_to_batch(q_points, q_lengths, fill_value=inf)  # (B, M', 3)
    batch_s_points, _ = pack_to_batch(s_points, s_lengths, fill_value=inf)  # (B, N', 3)
    batch_knn_distances, batch_knn_indices = keops_knn(batch_q_points, batch_s_points, neighbor_limit)
    batch_start_index = torch.cumsum(s_lengths, dim=0) - s_lengths
    batch_knn_indices = batch_knn_indices + batch_start_index.view(-1, 1, 1)
    batch_knn_indices.masked_fill_(torch.gt(batch_knn_distances, radius), s_points.shape[0])
    knn_indices, _ = batch_to_pack(batch_knn_indices, batch_q_masks)  # (M, K)
    return knn_indices


def radius_count_pack_mode(q_points, s_points, q_lengths, s_lengths, radius, inf=1e10):
    """Number of supports strictly inside ``radius`` of each query (reference: relu(radius - d).sign().sum(),
    radius_search.py:74-107) -> (M,) int64.

    fp32 3-D points count on the native cell list (no padding, no host synchronisation); anything else takes the
    padded route.  ``inf`` is only used by that route."""
    if q_points.dtype == torch.float32 and s_points.dtype == torch.float32 and q_points.shape[-1] == 3:
        counts = torch.zeros((q_points.shape[0],), dtype=torch.long, device=q_points.device)
        ext.radius_count_pack(q_points.contiguous(), s_points.contiguous(), _device_lengths(q_lengths, q_points),
                              _device_lengths(s_lengths, q_points), counts, radius)
        return counts
    return padded_radius_count_pack_mode(q_points, s_points, q_lengths, s_lengths, radius, inf)


def padded_radius_count_pack_mode(q_points, s_points, q_lengths, s_lengths, radius, inf=1e10):
    """The reference's formulation (radius_search.py:98-107): brute-force count over padded batches.  The count kernel
    is fp32: other floating dtypes are converted first (a pair within fp32 rounding of the radius may count differently)."""
    batch_q_points, batch_q_masks = pack_to_batch(q_points, q_lengths, fill_value=inf)
    batch_s_points, _ = pack_to_batch(s_points, s_lengths, fill_value=inf)
    counts = torch.zeros(batch_q_points.shape[:2], dtype=torch.long, device=q_points.device)
    ext.radius_count(batch_q_points.float().contiguous(), batch_s_points.float().contiguous(), counts, radius)
    radius_counts, _ = batch_to_pack(counts, batch_q_masks)
    return radius_counts


def naive_radius_search_pack_mode(q_points, s_points, q_lengths, s_lengths, radius, neighbor_limit):
    """The reference's per-cloud loop (radius_search.py:8-41) computes the same table as the fast version; here both
    names reach the same native kernel."""
    return radius_search_pack_mode(q_points, s_points, q_lengths, s_lengths, radius, neighbor_limit)
