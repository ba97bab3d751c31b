"""Voxelisation -- mirrors ``vision3d/ops/voxelize.py:6-100`` (same signatures and return tuples).

The reference floors ``points / voxel_size``, hashes (cloud, voxel) into one integer, runs ``torch.unique`` and three
``scatter_add_`` passes (float atomics on a GPU) and builds the batch indices in a Python loop with a device->host copy
of ``lengths``.  Here ``ext.voxelize_pack`` does all of it on the grid-subsampling kernels: one stable radix sort, ordered
per-voxel sums (bit-reproducible), voxel coordinates / means / feature means / per-cloud counts written once; the only
host read is the number of voxels, which ``torch.unique`` needs as well.
"""
import torch

from .. import ext
from .conversion import batch_to_pack, pack_to_batch


def voxelize_pack_mode(points, lengths, voxel_size, feats=None):
    """points (N, 3), lengths (B,), optional feats (N, C) -> (voxels (M, 3) long, nodes (M, 3), [v_feats (M, C)],
    v_lengths (B,)), ordered by (cloud, vx, vy, vz)."""
    if not points.is_cuda:
        raise RuntimeError("voxelize_pack_mode: CUDA tensors required (no CPU fallback in vision3d_amd)")
    pts = points.float().contiguous()
    lengths = lengths.to(device=pts.device, dtype=torch.long).contiguous()
    n = pts.shape[0]
    voxels = torch.empty((n, 3), dtype=torch.long, device=pts.device)
    nodes = torch.empty((n, 3), device=pts.device)
    v_lengths = torch.empty_like(lengths)
    num = torch.zeros((), dtype=torch.long, device=pts.device)
    fts = v_feats = None
    if feats is not None:
        fts = feats.float().contiguous()
        v_feats = torch.empty((n, fts.shape[1]), device=pts.device)
    ext.voxelize_pack(pts, lengths, voxels, nodes, v_lengths, num, voxel_size, feats=fts, voxel_feats=v_feats)
    m = int(num.item())
    if m < 0:
        raise RuntimeError("voxelize_pack_mode: a cloud spans more than 2^18 voxels along an axis (or holds NaN)")
    if feats is not None:
        return voxels[:m], nodes[:m], v_feats[:m], v_lengths
    return voxels[:m], nodes[:m], v_lengths


def voxelize(points, voxel_size, masks=None, feats=None):
    """Batch form (voxelize.py:66-100): points (B, N, 3), masks (B, N), optional feats (B, N, C) ->
    (voxels (B, M, 3), nodes (B, M, 3), [v_feats (B, M, C)], v_masks (B, M))."""
    packed, lengths = batch_to_pack(points, masks=masks)
    if masks is None:
        packed = packed.reshape(-1, points.shape[-1])
    if feats is not None:
        packed_feats, _ = batch_to_pack(feats, masks=masks)
        if masks is None:
            packed_feats = packed_feats.reshape(-1, feats.shape[-1])
        voxels, nodes, v_feats, v_lengths = voxelize_pack_mode(packed, lengths, voxel_size, feats=packed_feats)
    else:
        voxels, nodes, v_lengths = voxelize_pack_mode(packed, lengths, voxel_size)
    voxels, v_masks = pack_to_batch(voxels, v_lengths)
    nodes, _ = pack_to_batch(nodes, v_lengths)
    if feats is not None:
        v_feats, _ = pack_to_batch(v_feats, v_lengths)
        return voxels, nodes, v_feats, v_masks
    return voxels, nodes, v_masks
