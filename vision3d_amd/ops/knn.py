"""Feature-space / general kNN -- mirrors ``vision3d/ops/knn.py:30-145`` (``knn``, ``knn_pack_mode``).

The reference delegates the search to pykeops (``keops_knn``, knn.py:10-27: k smallest Euclidean distances,
ascending, with their indices).  pykeops is a third-party JIT library absent from this build, so the search is
re-backed by our own kernels: 3-D points go through the exact-difference ``knn_points`` kernel, other channel
counts through the MFMA distance + top-k kernels (``ext.knn_feat``).  Post-processing (dilation,
remove_nearest, distance_limit padding, squeeze) follows the reference line by line in meaning.
"""
from typing import Tuple

import torch
from torch import Tensor

from .. import ext
from .conversion import batch_to_pack, pack_to_batch


def keops_knn(q_points: Tensor, s_points: Tensor, k: int) -> Tuple[Tensor, Tensor]:
    """(*, N, C), (*, M, C) -> distances (*, N, k) ascending, indices (*, N, k).  Name kept for drop-in use."""
    if not (q_points.is_cuda and s_points.is_cuda):
        raise RuntimeError("knn: q_points and s_points must be CUDA tensors (no CPU fallback in vision3d_amd)")
    if q_points.dtype != torch.float32 or s_points.dtype != torch.float32:
        raise RuntimeError("knn: q_points and s_points must be float tensors")
    lead = q_points.shape[:-2]
    n, c = q_points.shape[-2:]
    m = s_points.shape[-2]
    qb = q_points.reshape((-1, n, c)).contiguous()
    sb = s_points.reshape((-1, m, c)).contiguous()
    dist = qb.new_empty((qb.shape[0], n, k))
    idx = torch.empty((qb.shape[0], n, k), dtype=torch.long, device=qb.device)
    if c == 3 and k <= 100:
        ext.knn_points(qb, sb, dist, idx, k)
    else:
        ext.knn_feat(qb, sb, dist, idx, k)
    return dist.reshape(tuple(lead) + (n, k)), idx.reshape(tuple(lead) + (n, k))


def knn(
    q_points: Tensor,
    s_points: Tensor,
    k: int,
    dilation: int = 1,
    distance_limit: float = None,
    return_distance: bool = False,
    remove_nearest: bool = False,
    transposed: bool = False,
    padding_mode: str = "nearest",
    padding_value: float = 1e10,
    squeeze: bool = False,
):
    """kNN of ``q_points`` in ``s_points`` ((*, N, C) / (*, M, C), channel-first if ``transposed``).

    Returns indices (*, N, k), preceded by distances if ``return_distance``; see reference knn.py:43-63 for the
    meaning of dilation / distance_limit / padding_mode / remove_nearest / squeeze.
    """
    if transposed:
        q_points = q_points.transpose(-1, -2)
        s_points = s_points.transpose(-1, -2)
    q_points = q_points.contiguous()
    s_points = s_points.contiguous()
    num_s_points = s_points.shape[-2]

    dilated_k = (k - 1) * dilation + 1
    if remove_nearest:
        dilated_k += 1
    final_k = min(dilated_k, num_s_points)
    knn_distances, knn_indices = keops_knn(q_points, s_points, final_k)

    if remove_nearest:
        knn_distances = knn_distances[..., 1:]
        knn_indices = knn_indices[..., 1:]
    if dilation > 1:
        knn_distances = knn_distances[..., ::dilation]
        knn_indices = knn_indices[..., ::dilation]
    knn_distances = knn_distances.contiguous()
    knn_indices = knn_indices.contiguous()

    if distance_limit is not None:
        assert padding_mode in ["nearest", "empty"]
        too_far = torch.ge(knn_distances, distance_limit)
        if padding_mode == "nearest":
            knn_distances = torch.where(too_far, knn_distances[..., :1], knn_distances)
            knn_indices = torch.where(too_far, knn_indices[..., :1], knn_indices)
        else:
            knn_distances[too_far] = padding_value
            knn_indices[too_far] = num_s_points

    if squeeze and k == 1:
        knn_distances = knn_distances.squeeze(-1)
        knn_indices = knn_indices.squeeze(-1)
    if return_distance:
        return knn_distances, knn_indices
    return knn_indices


def knn_pack_mode(
    q_points: Tensor,
    s_points: Tensor,
    q_lengths: Tensor,
    s_lengths: Tensor,
    k: int,
    return_distance: bool = False,
    inf: float = 1e10,
):
    """kNN in pack mode: q_points (N, 3), s_points (M, 3) with per-cloud lengths (B,) -> (N, k) per-cloud indices."""
    assert torch.all(torch.ge(s_lengths, k)), f"The number of support points less than {k}."
    batch_q_points, batch_q_masks = pack_to_batch(q_points, q_lengths, fill_value=inf)
    batch_s_points, _ = pack_to_batch(s_points, s_lengths, fill_value=inf)
    batch_knn_distances, batch_knn_indices = keops_knn(batch_q_points, batch_s_points, k)
    knn_indices, _ = batch_to_pack(batch_knn_indices, masks=batch_q_masks)
    if not return_distance:
        return knn_indices
    knn_distances, _ = batch_to_pack(batch_knn_distances, masks=batch_q_masks)
    return knn_distances, knn_indices
