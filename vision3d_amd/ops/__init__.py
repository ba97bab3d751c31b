"""Host-side mirror of the hot-path subset of the reference's ``vision3d.ops`` package.

Same names, signatures, return values and error behaviour as the reference wrappers
(``/root/reference/vision3d/ops/__init__.py`` and the per-op files cited in each module), re-backed by
``vision3d_amd.ext`` (hand-written gfx950 kernels behind a C ABI).  ``vision3d_amd.install()`` registers this
package as ``vision3d.ops`` / ``vision3d.ext`` so unchanged model code imports it.
"""
from .ball_query import ball_query, ball_query_multi
from .conversion import batch_to_pack, batch_to_pack_with_lengths, pack_to_batch
from .furthest_point_sample import furthest_point_sample
from .fused_group import edge_group, sa_group
from .gather import gather
from .group_gather import group_gather
from .knn import knn, knn_pack_mode
from .knn_points import knn_points
from .kpconv import kpconv, kpconv_gather
from .matching import (extract_correspondences_from_feats, extract_correspondences_from_scores,
                       extract_correspondences_from_scores_threshold, extract_correspondences_from_scores_topk)
from .superpoint_matching import SuperPointMatchingGlobalTopk, SuperPointMatchingMutualTopk
from .mutual_topk_select import batch_mutual_topk_select, mutual_topk_select
from .pack_gather import (global_avgpool_pack_mode, index_select, knn_interpolate, knn_interpolate_pack_mode,
                          local_maxpool_pack_mode, nearest_interpolate_pack_mode)
from .pairwise_distance import pairwise_distance
from .graph_pyramid import build_grid_and_radius_graph_pyramid
from .grid_subsample import grid_subsample_pack_mode
from .point_cloud_partition import ball_query_partition, get_point_to_node_indices, point_to_node_partition
from .radius_search import naive_radius_search_pack_mode, radius_count_pack_mode, radius_search_pack_mode
from .three_interpolate import three_interpolate
from .three_nn import three_nn
from .voxelize import voxelize, voxelize_pack_mode

__all__ = [
    "ball_query", "ball_query_multi", "batch_to_pack", "batch_to_pack_with_lengths", "pack_to_batch", "furthest_point_sample", "gather",
    "group_gather", "knn", "knn_pack_mode", "knn_points", "pairwise_distance", "radius_count_pack_mode",
    "grid_subsample_pack_mode", "radius_search_pack_mode", "three_interpolate", "three_nn",
    "mutual_topk_select", "batch_mutual_topk_select", "extract_correspondences_from_feats", "extract_correspondences_from_scores",
    "extract_correspondences_from_scores_threshold", "extract_correspondences_from_scores_topk",
    "SuperPointMatchingGlobalTopk", "SuperPointMatchingMutualTopk", "voxelize", "voxelize_pack_mode", "knn_interpolate", "get_point_to_node_indices",
    "naive_radius_search_pack_mode",
    "index_select", "local_maxpool_pack_mode", "global_avgpool_pack_mode", "nearest_interpolate_pack_mode",
    "knn_interpolate_pack_mode", "build_grid_and_radius_graph_pyramid", "sa_group", "edge_group", "kpconv_gather", "kpconv", "point_to_node_partition", "ball_query_partition",
]
