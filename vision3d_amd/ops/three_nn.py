"""``three_nn`` -- mirrors ``vision3d/ops/three_nn.py:12-38``."""
from typing import Tuple

import torch
from torch import Tensor

from .. import ext


def three_nn(q_points: Tensor, s_points: Tensor, transposed: bool = False) -> Tuple[Tensor, Tensor]:
    """Three nearest support points of every query point: (distances (B, N, 3), indices (B, N, 3))."""
    if transposed:
        q_points = q_points.transpose(1, 2)
        s_points = s_points.transpose(1, 2)
    q_points = q_points.contiguous()
    s_points = s_points.contiguous()
    assert s_points.shape[1] >= 3, "s_points must have at least 3 points."
    tnn_distances = q_points.new_zeros((q_points.shape[0], q_points.shape[1], 3))
    tnn_indices = torch.zeros((q_points.shape[0], q_points.shape[1], 3), dtype=torch.long, device=q_points.device)
    ext.three_nn(q_points, s_points, tnn_distances, tnn_indices)
    return tnn_distances, tnn_indices
