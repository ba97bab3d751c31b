"""Grid subsampling in pack mode -- mirrors ``vision3d/ops/grid_subsample.py:49-99``."""
import torch

from .. import ext


def grid_subsample_pack_mode(points, lengths, voxel_size):
    """points (N, 3), lengths (B,) -> sampled_points (M, 3), sampled_lengths (B,).

    One point per occupied voxel (the mean of its points), ordered by (cloud, vx, vy, vz) as the reference's
    ``torch.unique`` of the hash orders them.  One device->host read (the output size), like the reference."""
    points = points.contiguous()
    lengths = lengths.contiguous()
    out = torch.empty_like(points)
    out_lengths = torch.empty_like(lengths)
    num = torch.zeros((), dtype=torch.long, device=points.device)
    ext.grid_subsample_pack(points, lengths, out, out_lengths, num, voxel_size)
    m = int(num.item())
    if m < 0:
        raise RuntimeError("grid_subsample_pack_mode: a cloud spans more than 2^18 voxels along an axis (or holds NaN)")
    return out[:m], out_lengths
