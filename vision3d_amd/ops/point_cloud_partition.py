"""Point-to-node partition -- mirrors ``vision3d/ops/point_cloud_partition.py:40-124`` (same signatures and returns).

The reference forms the (M, N) node-to-point distance matrix and three more tensors of that size; here the nearest node of
every point, one device sort and a slice of the sorted order give the same outputs (``ext.point_to_node_partition`` /
``ext.point_to_node_knn``).  Distances are exact squared differences, so a point that is equidistant from two nodes to
within the reference's expansion error (x^2 - 2xy + y^2 in fp32) may be assigned to the other of the two; among exact
ties the lower node / point index wins (``torch.min`` / ``torch.topk`` leave that unspecified).
"""
from typing import Optional, Tuple

import torch
from torch import Tensor

from .. import ext
from .knn import knn


@torch.no_grad()
def point_to_node_partition(
    points: Tensor,
    nodes: Tensor,
    point_limit: Optional[int] = None,
    return_count: bool = False,
    gather_points: bool = True,
    inf: float = 1e12,
) -> Tuple[Tensor, ...]:
    """points (N, 3), nodes (M, 3) -> point_to_node (N,), [node_sizes (M,)], [node_masks (M,), node_knn_indices (M, K),
    node_knn_masks (M, K)] exactly as the reference orders them.  ``inf`` is accepted for signature parity only."""
    if not (points.is_cuda and nodes.is_cuda):
        raise RuntimeError("point_to_node_partition: CUDA tensors required (no CPU fallback in vision3d_amd)")
    points = points.float().contiguous()
    nodes = nodes.float().contiguous()
    n, m = points.shape[0], nodes.shape[0]
    dev = points.device
    point_to_node = torch.empty((n,), dtype=torch.long, device=dev)
    node_sizes = torch.empty((m,), dtype=torch.long, device=dev) if return_count else None
    node_masks = torch.empty((m,), dtype=torch.bool, device=dev) if gather_points else None
    max_size = torch.zeros((), dtype=torch.long, device=dev) if gather_points else None
    ws = ext.point_to_node_partition(points, nodes, point_to_node, node_sizes, node_masks, max_size)
    return_list = [point_to_node]
    if return_count:
        return_list.append(node_sizes)
    if gather_points:
        return_list.append(node_masks)
        max_points_per_node = int(max_size.item())  # the reference's own sync (:91): it sizes the output
        if point_limit is not None:
            max_points_per_node = min(max_points_per_node, point_limit)
        assert max_points_per_node > 0, "All nodes are empty."
        node_knn_indices = torch.empty((m, max_points_per_node), dtype=torch.long, device=dev)
        node_knn_masks = torch.empty((m, max_points_per_node), dtype=torch.bool, device=dev)
        ext.point_to_node_knn(ws, n, node_knn_indices, node_knn_masks)
        return_list.append(node_knn_indices)
        return_list.append(node_knn_masks)
    return tuple(return_list)


@torch.no_grad()
def ball_query_partition(
    points: Tensor,
    nodes: Tensor,
    radius: float,
    point_limit: int,
    return_count: bool = False,
) -> Tuple[Tensor, ...]:
    """The ``point_limit`` nearest points of every node that lie within ``radius`` (point_cloud_partition.py:107-124)."""
    node_knn_distances, node_knn_indices = knn(nodes, points, point_limit, return_distance=True)
    node_knn_masks = torch.lt(node_knn_distances, radius)
    node_knn_indices = torch.where(node_knn_masks, node_knn_indices, torch.full_like(node_knn_indices, points.shape[0]))
    node_sizes = node_knn_masks.sum(1)
    node_masks = torch.gt(node_sizes, 0)
    if return_count:
        return node_knn_indices, node_knn_masks, node_masks, node_sizes
    return node_knn_indices, node_knn_masks, node_masks


def get_point_to_node_indices(points: Tensor, nodes: Tensor, return_counts: bool = False):
    """Deprecated spelling kept by the reference (point_cloud_partition.py:13-37): nearest node of every point and,
    optionally, the node sizes -- the first outputs of ``point_to_node_partition``."""
    out = point_to_node_partition(points, nodes, return_count=return_counts, gather_points=False)
    return (out[0], out[1]) if return_counts else out[0]
