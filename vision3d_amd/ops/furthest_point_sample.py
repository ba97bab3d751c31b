"""``furthest_point_sample`` -- mirrors ``vision3d/ops/furthest_point_sample.py:12-49``."""
import torch
from torch import Tensor

from .. import ext
from .gather import gather


def furthest_point_sample(
    points: Tensor, num_samples: int, *input_tensors: Tensor, gather_points: bool = True, transposed: bool = False
):
    """Furthest point sampling, optionally followed by gathering the sampled coordinates and features.

    Args and return values are the reference's, quirk included: with ``gather_points=True`` the result is the
    tuple ``(sampled_points, *sampled_features)`` and the indices are NOT returned
    (reference ``furthest_point_sample.py:49``); with ``gather_points=False`` it is the (B, M) index tensor.
    """
    pts = points.transpose(1, 2) if transposed else points  # (B, 3, N) -> (B, N, 3)
    pts = pts.contiguous()
    batch, n = pts.shape[0], pts.shape[1]
    # allocated on the points' device directly (the reference builds them on the host and copies, :34-35)
    distances = torch.full((batch, n), 1e10, dtype=torch.float32, device=pts.device)
    indices = torch.zeros((batch, num_samples), dtype=torch.long, device=pts.device)
    ext.furthest_point_sample(pts, distances, indices, num_samples)
    if not gather_points:
        return indices

    channel_first = pts.transpose(1, 2).contiguous()  # (B, 3, N)
    sampled_points = gather(channel_first, indices)  # (B, 3, M)
    if not transposed:
        sampled_points = sampled_points.transpose(1, 2).contiguous()  # (B, M, 3)
    sampled_tensors = [gather(t, indices) for t in input_tensors]
    return (sampled_points, *sampled_tensors)
