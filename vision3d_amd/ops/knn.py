"""Feature-space / general kNN -- the interface of ``vision3d/ops/knn.py:30-145`` (``knn``, ``knn_pack_mode``).

The reference delegates the search to pykeops (``keops_knn``, knn.py:10-27: k smallest Euclidean distances,
ascending, with their indices).  pykeops is a third-party JIT library absent from this build, so the search is
re-backed by our own kernels:

* 3-D fp32 points, k <= 100: the exact-difference ``knn_points`` kernels without their heap replay (``ext.knn_lex``);
* any other channel count or larger k (<= 2048): MFMA distances + row selection + exact re-ranking (``ext.knn_feat``);
* pack mode (3-D fp32, k <= 100): the radius search's cell list for clouds of a few thousand points and more, the
  ``knn_points`` kernels on each cloud's own slice of the packed arrays below (``ext.knn_pack``) -- no padded batch, no
  ``.item()``.

Equal distances come lower index first on every route: pykeops' ``Kmin_argKmin`` inserts with a strict ``<`` while it
scans the supports in index order (``ops.knn_points``, the reference's CUDA kernel, keeps ITS order: a max-heap's).

``knn``'s options (dilation, remove_nearest, distance_limit / padding_mode, squeeze; reference knn.py:43-63) are expressed
as ONE strided column selection of the ascending list plus one masked fill.
"""
from typing import Tuple

import torch
from torch import Tensor

from .. import ext
from .conversion import batch_to_pack, pack_to_batch

_KNN_POINTS_MAX_K = 100  # knn_points_cuda.cuh:68 (best_dist[100]); kept by our kernels
_KNN_FEAT_MAX_K = 2048  # include/v3d_hip.h: V3D_KNN_FEAT_MAX_K
_KNN_CELL_LIST_MAX_K = 64  # v3d_knn_pack_f32: the cell-list route (k <= 64, <= 4096 clouds, >= 2048 supports per cloud)
_KNN_CELL_LIST_MAX_CLOUDS = 4096
_KNN_LDS_MAX_SUPPORTS = 4096  # up to here v3d_knn_lex_f32 builds its cell list per workgroup in LDS


def keops_knn(q_points: Tensor, s_points: Tensor, k: int) -> Tuple[Tensor, Tensor]:
    """(*, N, C), (*, M, C) -> distances (*, N, k) ascending, indices (*, N, k).  Name kept for drop-in use.

    pykeops has no limit on k; here k <= 2048 (k > M yields (inf, -1) in the extra slots of the feature path)."""
    if not (q_points.is_cuda and s_points.is_cuda):
        raise RuntimeError("knn: q_points and s_points must be CUDA tensors (no CPU fallback in vision3d_amd)")
    if q_points.dtype != torch.float32 or s_points.dtype != torch.float32:
        raise RuntimeError("knn: q_points and s_points must be float tensors")
    if k > _KNN_FEAT_MAX_K:
        raise RuntimeError(f"knn: k={k} exceeds {_KNN_FEAT_MAX_K}, the largest list the selection kernels keep per query")
    lead = tuple(q_points.shape[:-2])
    n, c = q_points.shape[-2:]
    m = s_points.shape[-2]
    qb = q_points.reshape((-1, n, c)).contiguous()
    sb = s_points.reshape((-1, m, c)).contiguous()
    dist = qb.new_empty((qb.shape[0], n, k))
    idx = torch.empty((qb.shape[0], n, k), dtype=torch.long, device=qb.device)
    if c == 3 and k <= _KNN_CELL_LIST_MAX_K and m > _KNN_LDS_MAX_SUPPORTS and qb.shape[0] <= _KNN_CELL_LIST_MAX_CLOUDS:
        # clouds too large for the per-workgroup cell list of the batch kernels: the pack-mode cell list, every cloud
        # of the batch one cloud of a pack with equal lengths (same order among equal distances)
        ql = torch.full((qb.shape[0],), n, dtype=torch.long, device=qb.device)
        sl = torch.full((qb.shape[0],), m, dtype=torch.long, device=qb.device)
        ext.knn_pack(qb.reshape(-1, 3), sb.reshape(-1, 3), ql, sl, dist.reshape(-1, k), idx.reshape(-1, k), k)
    elif c == 3 and k <= _KNN_POINTS_MAX_K:
        ext.knn_lex(qb, sb, dist, idx, k)
    else:
        ext.knn_feat(qb, sb, dist, idx, k)
    return dist.reshape(lead + (n, k)), idx.reshape(lead + (n, k))


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

    Returns indices (*, N, k), preceded by distances if ``return_distance``.  Of the ascending neighbour list the ranks
    ``first, first + dilation, ...`` are kept, ``first`` = 1 with ``remove_nearest`` (drop the query itself) else 0; the
    list is as long as that needs, cut at the number of supports.  Neighbours at or beyond ``distance_limit`` are replaced
    by the nearest one (``padding_mode="nearest"``) or by ``(padding_value, M)`` (``"empty"``).
    """
    if transposed:
        q_points, s_points = q_points.transpose(-1, -2), s_points.transpose(-1, -2)
    q_points, s_points = q_points.contiguous(), s_points.contiguous()
    num_supports = s_points.shape[-2]
    first = 1 if remove_nearest else 0
    searched = min(first + (k - 1) * dilation + 1, num_supports)
    distances, indices = keops_knn(q_points, s_points, searched)
    if first or dilation > 1:
        distances, indices = distances[..., first::dilation], indices[..., first::dilation]
    distances, indices = distances.contiguous(), indices.contiguous()

    if distance_limit is not None:
        assert padding_mode in ["nearest", "empty"]
        beyond = distances >= distance_limit
        if padding_mode == "nearest":
            fill_distance, fill_index = distances[..., :1], indices[..., :1]
        else:
            fill_distance = distances.new_full((), padding_value)
            fill_index = indices.new_full((), num_supports)
        distances = torch.where(beyond, fill_distance, distances)
        indices = torch.where(beyond, fill_index, indices)

    if squeeze and k == 1:
        distances, indices = distances.squeeze(-1), indices.squeeze(-1)
    return (distances, indices) if return_distance else indices


def knn_pack_mode(
    q_points: Tensor,
    s_points: Tensor,
    q_lengths: Tensor,
    s_lengths: Tensor,
    k: int,
    return_distance: bool = False,
    inf: float = 1e10,
):
    """kNN in pack mode: q_points (N, 3), s_points (M, 3) with per-cloud lengths (B,) -> (N, k) indices INSIDE the cloud
    (what the reference's padded batch yields, knn.py:138-141), preceded by the distances if ``return_distance``.

    The length check is the reference's own assertion (knn.py:137; one boolean read back from the device, gone under
    ``python -O``); the search itself neither pads nor synchronises.  ``inf`` only matters on the padded route taken by
    inputs the pack kernel does not cover (not fp32 3-D, or k > 100)."""
    assert torch.all(torch.ge(s_lengths, k)), f"The number of support points less than {k}."
    native = (q_points.is_cuda and q_points.dtype == torch.float32 and s_points.dtype == torch.float32
              and q_points.dim() == 2 and q_points.shape[-1] == 3 and s_points.shape[-1] == 3 and k <= _KNN_POINTS_MAX_K)
    if not native:
        return padded_knn_pack_mode(q_points, s_points, q_lengths, s_lengths, k, return_distance, inf)
    distances = q_points.new_empty((q_points.shape[0], k))
    indices = torch.empty((q_points.shape[0], k), dtype=torch.long, device=q_points.device)
    dev = q_points.device  # the reference's loops take the lengths on either side and in any integer type
    ext.knn_pack(q_points.contiguous(), s_points.contiguous(), q_lengths.to(device=dev, dtype=torch.long).contiguous(),
                 s_lengths.to(device=dev, dtype=torch.long).contiguous(), distances, indices, k)
    return (distances, indices) if return_distance else indices


def padded_knn_pack_mode(q_points, s_points, q_lengths, s_lengths, k, return_distance=False, inf=1e10):
    """The reference's formulation (knn.py:138-145): pad every cloud to the longest with points at ``inf``, search the
    batch, take the rows of the real queries."""
    batch_q_points, batch_q_masks = pack_to_batch(q_points, q_lengths, fill_value=inf)
    batch_s_points, _ = pack_to_batch(s_points, s_lengths, fill_value=inf)
    batch_distances, batch_indices = keops_knn(batch_q_points, batch_s_points, k)
    indices, _ = batch_to_pack(batch_indices, masks=batch_q_masks)
    if not return_distance:
        return indices
    distances, _ = batch_to_pack(batch_distances, masks=batch_q_masks)
    return distances, indices
