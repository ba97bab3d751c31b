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


def ball_query_multi(q_points: Tensor, s_points: Tensor, num_samples_list, radius_list, transposed: bool = False):
    """The ball queries of a multi-scale grouping layer (PointNet++ MSG, models/pointnet2/modules.py:47-57 calls
    ``ball_query`` once per scale on the same clouds) in one launch: every pair of points is evaluated once and tested
    against all radii.  Returns a list with the result of ``ball_query(q_points, s_points, k, r, transposed)`` per scale,
    bit for bit.  fp32 points, at most four scales (more, or fp64, fall back to one call per scale)."""
    if transposed:
        q_points = q_points.transpose(1, 2)
        s_points = s_points.transpose(1, 2)
    q_points = q_points.contiguous()
    s_points = s_points.contiguous()
    if len(num_samples_list) != len(radius_list):
        raise ValueError("ball_query_multi: one radius per sample count")
    if q_points.dtype != torch.float32 or not (1 <= len(num_samples_list) <= 4):
        return [ball_query(q_points, s_points, k, r) for k, r in zip(num_samples_list, radius_list)]
    outs = [torch.zeros((q_points.shape[0], q_points.shape[1], int(k)), dtype=torch.long, device=q_points.device)
            for k in num_samples_list]
    ext.ball_query_multi(q_points, s_points, outs, list(radius_list), list(num_samples_list))
    return outs
