"""``ball_query`` -- mirrors ``vision3d/ops/ball_query.py:10-42``."""
import torch
from torch import Tensor

from .. import ext


def ball_query(q_points: Tensor, s_points: Tensor, num_samples: int, radius: float, transposed: bool = False) -> Tensor:
    """First ``num_samples`` support points (in index order) inside the ball of each query point.

    q_points (B, M, 3) / s_points (B, N, 3), or (B, 3, M) / (B, 3, N) if ``transposed``.
    Returns a LongTensor (B, M, num_samples); short lists are padded with their first hit, empty ones are 0.
    """
    if transposed:
        q_points = q_points.transpose(1, 2)
        s_points = s_points.transpose(1, 2)
    q_points = q_points.contiguous()
    s_points = s_points.contiguous()
    indices = torch.zeros((q_points.shape[0], q_points.shape[1], num_samples), dtype=torch.long, device=q_points.device)
    ext.ball_query(q_points, s_points, indices, radius, num_samples)
    return indices
