"""Grid + radius graph pyramid in pack mode -- mirrors ``vision3d/models/graph_pyramid/graph_pyramid.py:9-63``
(`build_grid_and_radius_graph_pyramid`), the caller of the pack-mode radius search and grid subsampling on the KPConv /
GeoTransformer path (BASELINE config 4).  Same arguments, same dictionary; every table comes from the native kernels."""
from typing import List

import torch
from torch import Tensor

from .. import ext
from .grid_subsample import grid_subsample_pack_mode
from .radius_search import _device_lengths, radius_search_pack_mode


def _native(points: Tensor, neighbor_limits: List[int]) -> bool:
    return (points.is_cuda and points.dtype == torch.float32 and points.dim() == 2 and points.shape[-1] == 3
            and all(0 < limit <= 512 for limit in neighbor_limits))


@torch.no_grad()
def build_grid_and_radius_graph_pyramid(points: Tensor, lengths: Tensor, num_stages: int, voxel_size: float,
                                        search_radius: float, neighbor_limits: List[int],
                                        voxelize_first: bool = False) -> dict:
    points_list, lengths_list = [], []
    for i in range(num_stages):
        if voxelize_first or i > 0:
            points, lengths = grid_subsample_pack_mode(points, lengths, voxel_size)
        points_list.append(points)
        lengths_list.append(lengths)
        voxel_size *= 2.0

    # The points of stage i are searched at ONE radius, search_radius * 2^i, up to three times: by themselves
    # (neighbours), by the points of stage i+1 (subsampling) and by the points of stage i-1 (the upsampling of stage i-1,
    # whose radius is twice that stage's).  The reference re-pads and re-searches per table (graph_pyramid.py:31-45);
    # here the cell list of a stage is built once (ext.radius_grid_build) and searched by the three query sets --
    # identical tables, 4 counting sorts instead of 10.
    native = _native(points_list[0], neighbor_limits)

    def search(grid, q_points, q_lengths, s_points, s_lengths, radius, limit):
        if grid is None:
            return radius_search_pack_mode(q_points, s_points, q_lengths, s_lengths, radius, limit)
        out = torch.empty((q_points.shape[0], limit), dtype=torch.long, device=q_points.device)
        ext.radius_search_grid(q_points, _device_lengths(q_lengths, q_points), grid, out, radius, limit)
        return out

    if native:
        points_list = [p.contiguous() for p in points_list]
        dev_lengths = [_device_lengths(ln, p) for ln, p in zip(lengths_list, points_list)]
        grids = [ext.radius_grid_build(p, ln, search_radius * 2.0 ** i)
                 for i, (p, ln) in enumerate(zip(points_list, dev_lengths))]
    else:
        grids = [None] * num_stages

    neighbors_list, subsampling_list, upsampling_list = [], [], []
    for i in range(num_stages):
        cur_points, cur_lengths = points_list[i], lengths_list[i]
        neighbors_list.append(search(grids[i], cur_points, cur_lengths, cur_points, cur_lengths, search_radius,
                                     neighbor_limits[i]))
        if i < num_stages - 1:
            sub_points, sub_lengths = points_list[i + 1], lengths_list[i + 1]
            subsampling_list.append(search(grids[i], sub_points, sub_lengths, cur_points, cur_lengths, search_radius,
                                           neighbor_limits[i]))
            upsampling_list.append(search(grids[i + 1], cur_points, cur_lengths, sub_points, sub_lengths,
                                          search_radius * 2, neighbor_limits[i + 1]))
        search_radius *= 2

    return {"points": points_list, "lengths": lengths_list, "neighbors": neighbors_list,
            "subsampling": subsampling_list, "upsampling": upsampling_list}
