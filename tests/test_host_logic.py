"""Host-side logic that needs no GPU: argument validation (reference messages), pack/batch conversion,
kNN post-processing, and the `vision3d` aliasing."""
import importlib
import sys

import numpy as np
import pytest
import torch

import vision3d_amd
from vision3d_amd import ext
from vision3d_amd.ops import conversion

knn_mod = importlib.import_module("vision3d_amd.ops.knn")


def test_ext_exports_reference_names():
    # csrc/pybind.cpp:72-158 (+ the `knn` alias used by ops/knn_points.py:36)
    for name in ("ball_query", "furthest_point_sample", "gather_forward", "gather_backward", "group_gather_forward",
                 "group_gather_backward", "three_nn", "three_interpolate_forward", "three_interpolate_backward",
                 "knn_points", "knn"):
        assert callable(getattr(ext, name))
    assert ext.knn is ext.knn_points


def test_validation_messages_match_torch_helper():
    q = torch.zeros(1, 4, 3)
    idx = torch.zeros(1, 4, 2, dtype=torch.long)
    with pytest.raises(RuntimeError, match="q_points must be a CUDA tensor"):
        ext.ball_query(q, q, idx, 0.1, 2)
    with pytest.raises(RuntimeError, match="points must be a CUDA tensor"):
        ext.furthest_point_sample(q, torch.zeros(1, 4), torch.zeros(1, 2, dtype=torch.long), 2)
    with pytest.raises(RuntimeError, match="sources must be a CUDA tensor"):
        ext.group_gather_forward(torch.zeros(1, 2, 4), idx, torch.zeros(1, 2, 4, 2))
    with pytest.raises(RuntimeError, match="CUDA tensors"):
        from vision3d_amd.ops import pairwise_distance
        pairwise_distance(torch.zeros(3, 2), torch.zeros(4, 2))


def test_keyword_names_match_pybind():
    import inspect

    expect = {
        "ball_query": ["q_points", "s_points", "indices", "max_radius", "num_samples"],
        "furthest_point_sample": ["points", "distances", "indices", "num_samples"],
        "gather_forward": ["sources", "indices", "targets"],
        "gather_backward": ["target_grads", "indices", "source_grads", "num_sources"],
        "group_gather_forward": ["sources", "indices", "targets"],
        "group_gather_backward": ["target_grads", "indices", "source_grads", "num_sources"],
        "three_nn": ["q_points", "s_points", "tnn_distances", "tnn_indices"],
        "three_interpolate_forward": ["sources", "indices", "weights", "targets"],
        "three_interpolate_backward": ["target_grads", "indices", "weights", "source_grads", "num_sources"],
        "knn_points": ["q_points", "s_points", "knn_distances", "knn_indices", "num_neighbors"],
    }
    for name, args in expect.items():
        assert list(inspect.signature(getattr(ext, name)).parameters) == args, name


def test_pack_batch_round_trip():
    lengths = torch.tensor([3, 0, 5, 1])
    pack = torch.arange(9 * 2, dtype=torch.float32).view(9, 2)
    batch, masks = conversion.pack_to_batch(pack, lengths, fill_value=1e10)
    assert batch.shape == (4, 5, 2) and masks.sum().item() == 9
    assert torch.equal(masks.sum(1), lengths)
    assert torch.all(batch[~masks] == 1e10)
    back, l2 = conversion.batch_to_pack(batch, masks)
    assert torch.equal(back, pack) and torch.equal(l2, lengths)
    back2, _ = conversion.batch_to_pack_with_lengths(batch, lengths)
    assert torch.equal(back2, pack)
    b3, m3 = conversion.pack_to_batch(pack, lengths, max_length=7)
    assert b3.shape == (4, 7, 2) and torch.equal(m3.sum(1), lengths)


def test_knn_post_processing(monkeypatch):
    """dilation / remove_nearest / distance_limit / squeeze follow ops/knn.py:72-107; the search itself is faked."""
    def fake_search(q, s, k):
        d = torch.cdist(q, s)
        dist, idx = torch.sort(d, dim=-1, stable=True)
        return dist[..., :k].contiguous(), idx[..., :k].contiguous()

    monkeypatch.setattr(knn_mod, "keops_knn", fake_search)
    g = torch.Generator().manual_seed(0)
    q = torch.rand(2, 6, 3, generator=g)
    s = torch.rand(2, 20, 3, generator=g)
    full_d, full_i = fake_search(q, s, 20)
    idx = knn_mod.knn(q, s, 4)
    assert torch.equal(idx, full_i[..., :4])
    d, idx = knn_mod.knn(q, s, 3, dilation=2, return_distance=True)
    assert torch.equal(idx, full_i[..., 0:5:2]) and torch.equal(d, full_d[..., 0:5:2])
    idx = knn_mod.knn(q, s, 3, remove_nearest=True)
    assert torch.equal(idx, full_i[..., 1:4])
    lim = float(full_d[..., 2].median())
    d, idx = knn_mod.knn(q, s, 4, distance_limit=lim, return_distance=True)
    far = full_d[..., :4] >= lim
    assert torch.equal(idx[far], full_i[..., :1].expand(-1, -1, 4)[far])
    d, idx = knn_mod.knn(q, s, 4, distance_limit=lim, return_distance=True, padding_mode="empty")
    assert torch.all(idx[far] == 20) and torch.all(d[far] == 1e10)
    idx = knn_mod.knn(q, s, 1, squeeze=True)
    assert idx.shape == (2, 6)
    idx = knn_mod.knn(q.transpose(1, 2), s.transpose(1, 2), 2, transposed=True)
    assert torch.equal(idx, full_i[..., :2])
    idx = knn_mod.knn(q, s, 30)  # k larger than the support set is clipped (knn.py:75)
    assert idx.shape[-1] == 20


def test_install_aliases_modules():
    saved = {k: sys.modules.get(k) for k in ("vision3d", "vision3d.ext", "vision3d.ops")}
    try:
        for k in saved:
            sys.modules.pop(k, None)
        vision3d_amd.install()
        m = importlib.import_module("vision3d.ext")  # what the reference's load_ext does (utils/misc.py:23-27)
        for fn in ("ball_query", "knn_points", "knn", "three_nn", "group_gather_forward"):
            assert hasattr(m, fn)
        from vision3d.ops import ball_query, furthest_point_sample, group_gather  # noqa: F401
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from vision3d_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU / PyTorch fallback"):
        _lib.lib()


def test_next_row_ops_signatures_and_no_cpu_path():
    """SURVEY 8f mirrors: same parameter names as the reference's functions, and CPU tensors are refused (no fallback)."""
    import inspect

    from vision3d_amd import ops

    expect = {
        # vision3d/ops/mutual_topk_select.py:7-14, :63-72
        "mutual_topk_select": ["score_mat", "k", "largest", "threshold", "mutual", "reduce_result"],
        "batch_mutual_topk_select": ["score_mat", "k", "row_masks", "col_masks", "largest", "threshold", "mutual",
                                     "reduce_result"],
        # ops/matching.py:11-17, :125-127
        "extract_correspondences_from_scores": ["score_mat", "mutual", "has_dustbin", "threshold", "return_score"],
        "extract_correspondences_from_feats": ["src_feats", "tgt_feats", "mutual", "return_feat_dist", "normalized"],
        # ops/matching.py:65, :90-92
        "extract_correspondences_from_scores_threshold": ["score_mat", "threshold", "return_score"],
        "extract_correspondences_from_scores_topk": ["score_mat", "k", "has_dustbin", "largest", "return_score"],
        # ops/voxelize.py:6, :66
        "voxelize_pack_mode": ["points", "lengths", "voxel_size", "feats"],
        "voxelize": ["points", "voxel_size", "masks", "feats"],
        # ops/point_cloud_partition.py:41-48, :108-114
        "point_to_node_partition": ["points", "nodes", "point_limit", "return_count", "gather_points", "inf"],
        "ball_query_partition": ["points", "nodes", "radius", "point_limit", "return_count"],
        # ops/index_select.py:4, ops/pooling.py:6,22, ops/nearest_interpolate.py:6, ops/knn_interpolate.py:43-50
        "index_select": ["inputs", "indices", "dim"],
        "local_maxpool_pack_mode": ["feats", "neighbor_indices"],
        "global_avgpool_pack_mode": ["feats", "lengths"],
        "nearest_interpolate_pack_mode": ["inputs", "neighbor_indices"],
        "knn_interpolate_pack_mode": ["q_points", "s_points", "s_feats", "neighbor_indices", "k", "eps"],
        # ops/knn_interpolate.py:10-17, ops/point_cloud_partition.py:14, ops/radius_search.py:8
        "knn_interpolate": ["q_points", "s_points", "s_feats", "k", "eps", "distance_limit"],
        "get_point_to_node_indices": ["points", "nodes", "return_counts"],
        "naive_radius_search_pack_mode": ["q_points", "s_points", "q_lengths", "s_lengths", "radius", "neighbor_limit"],
        # ops/radius_search.py:44, :84
        "radius_search_pack_mode": ["q_points", "s_points", "q_lengths", "s_lengths", "radius", "neighbor_limit", "inf"],
        "radius_count_pack_mode": ["q_points", "s_points", "q_lengths", "s_lengths", "radius", "inf"],
    }
    for name, args in expect.items():
        assert list(inspect.signature(getattr(ops, name)).parameters) == args, name
    f, t = torch.zeros(5, 4), torch.zeros(3, 2, dtype=torch.long)
    for call in (lambda: ops.mutual_topk_select(torch.zeros(4, 4), 1),
                 lambda: ops.extract_correspondences_from_feats(f, f),
                 lambda: ops.extract_correspondences_from_scores_threshold(f, 0.1),
                 lambda: ops.voxelize_pack_mode(torch.zeros(5, 3), torch.tensor([5]), 0.1),
                 lambda: ops.extract_correspondences_from_scores_topk(f, 3),
                 lambda: ops.SuperPointMatchingGlobalTopk(4)(f, f),
                 lambda: ops.SuperPointMatchingMutualTopk(4, 1)(f, f),
                 lambda: ops.point_to_node_partition(torch.zeros(5, 3), torch.zeros(2, 3)),
                 lambda: ops.local_maxpool_pack_mode(f, t),
                 lambda: ops.global_avgpool_pack_mode(f, torch.tensor([5])),
                 lambda: ops.nearest_interpolate_pack_mode(f, t),
                 lambda: ops.knn_interpolate_pack_mode(torch.zeros(3, 3), torch.zeros(5, 3), f, t),
                 lambda: ops.knn_interpolate(torch.zeros(3, 3), torch.zeros(5, 3), f),
                 lambda: ops.get_point_to_node_indices(torch.zeros(5, 3), torch.zeros(2, 3)),
                 lambda: ops.radius_count_pack_mode(torch.zeros(5, 3), torch.zeros(5, 3), torch.tensor([5]),
                                                    torch.tensor([5]), 0.1),
                 lambda: ops.radius_search_pack_mode(torch.zeros(5, 3), torch.zeros(5, 3), torch.tensor([5]),
                                                     torch.tensor([5]), 0.1, 4),
                 lambda: ops.sa_group(torch.zeros(1, 3, 2), torch.zeros(1, 3, 5), None, torch.zeros(1, 2, 2, dtype=torch.long)),
                 lambda: ops.edge_group(torch.zeros(1, 4, 2), torch.zeros(1, 4, 5), torch.zeros(1, 2, 2, dtype=torch.long)),
                 lambda: ops.kpconv_gather(torch.zeros(3, 3), torch.zeros(5, 3), f, t, torch.zeros(15, 3), 0.1)):
        with pytest.raises(RuntimeError, match="CUDA"):
            call()
    # the generic index_select keeps the reference's semantics on any device for layouts outside the native kernel
    x = torch.arange(24.0).view(2, 3, 4)
    idx = torch.tensor([[0, 2], [1, 1]])
    out = ops.index_select(x, idx, dim=1)
    assert out.shape == (2, 2, 2, 4) and torch.equal(out[:, 1, 0], x[:, 1])


def test_select_shim_validates_before_touching_the_gpu():
    with pytest.raises(RuntimeError, match="score_mat must be a CUDA tensor"):
        ext.mutual_topk_select(torch.zeros(1, 3, 3), 1, num_out=torch.zeros((), dtype=torch.long))
    with pytest.raises(RuntimeError, match="feats must be a CUDA tensor"):
        ext.local_maxpool_pack(torch.zeros(3, 4), torch.zeros(2, 2, dtype=torch.long), torch.zeros(2, 4))
    with pytest.raises(RuntimeError, match="q_points must be a CUDA tensor"):
        ext.sa_group(torch.zeros(1, 3, 2), torch.zeros(1, 3, 4), None, torch.zeros(1, 2, 2, dtype=torch.long),
                     torch.zeros(1, 3, 2, 2))


def test_ext_has_every_name_of_the_reference_module():
    """csrc/pybind.cpp:20-158: every registered name exists; the deformation-graph / depth-image helpers raise."""
    for name in ("radius_neighbors", "grid_subsampling", "sample_nodes_with_fps"):
        assert callable(getattr(ext, name))
    for name in ("depth_to_mesh", "erode_mesh", "sample_nodes", "compute_edges_geodesic", "compute_edges_euclidean",
                 "compute_pixel_anchors_geodesic", "node_and_edge_clean_up", "update_pixel_anchors", "compute_clusters",
                 "build_deformation_graph_from_point_cloud"):
        with pytest.raises(NotImplementedError, match="outside the operator path"):
            getattr(ext, name)()
    import inspect

    assert list(inspect.signature(ext.radius_neighbors).parameters) == ["q_points", "s_points", "q_lengths", "s_lenghts",
                                                                        "radius"]
    assert list(inspect.signature(ext.grid_subsampling).parameters) == ["points", "lengths", "voxel_size"]
    assert list(inspect.signature(ext.sample_nodes_with_fps).parameters) == ["points", "min_distance", "num_samples"]
