"""``knn_points`` -- mirrors ``vision3d/ops/knn_points.py:11-41`` (heap-order kNN on 3-D points, k <= 100)."""
from typing import Tuple, Union

import torch
from torch import Tensor

from .. import ext


def knn_points(
    q_points: Tensor, s_points: Tensor, k: int, transposed: bool = False, return_distance: bool = False
) -> Union[Tensor, Tuple[Tensor, Tensor]]:
    """q_points (B, N, 3), s_points (B, M, 3) (or channel-first if ``transposed``) -> indices (B, N, k)
    and, with ``return_distance``, the Euclidean distances (B, N, k) in front of them."""
    if transposed:
        q_points = q_points.transpose(1, 2)
        s_points = s_points.transpose(1, 2)
    q_points = q_points.contiguous()
    s_points = s_points.contiguous()
    # (the reference zero-fills both, ops/knn_points.py:33-34; the kernel writes every slot -- the (1e5, 0) sentinels of
    # M < k included -- so nothing is filled here)
    knn_distances = q_points.new_empty((q_points.shape[0], q_points.shape[1], k))
    knn_indices = torch.empty((q_points.shape[0], q_points.shape[1], k), dtype=torch.long, device=q_points.device)
    ext.knn(q_points, s_points, knn_distances, knn_indices, k)  # the reference calls `ext_module.knn` here (:36)
    if return_distance:
        return knn_distances, knn_indices
    return knn_indices
