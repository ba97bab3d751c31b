"""Pack-mode grid subsampling / voxelisation at the pyramid shape (16 clouds x 30000 points), for a kernel trace:
   rocprofv3 --kernel-trace --stats -d gpurun_out/grid -- python tools/grid_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision3d_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
nc, n = 16, 30000
pts = (torch.rand((nc * n, 3), generator=g) * torch.tensor([3.0, 3.0, 2.5])).to(DEV)
feats = torch.rand((nc * n, 32), generator=g).to(DEV)
lengths = torch.full((nc,), n, dtype=torch.long, device=DEV)
for _ in range(20):
    sp, sl = ops.grid_subsample_pack_mode(pts, lengths, 0.025)
torch.cuda.synchronize()
print("sampled", sp.shape[0])
for _ in range(20):
    out = ops.voxelize_pack_mode(pts, lengths, 0.025, feats=feats)
torch.cuda.synchronize()
print("voxels", out[0].shape[0])
