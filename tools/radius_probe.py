"""Pack-mode radius search / count at the KPConv-pyramid shape (16 clouds x 30000 points), for a kernel trace:
   rocprofv3 --kernel-trace --stats -d gpurun_out/rad -- python tools/radius_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision3d_amd import ext  # noqa: E402

DEV = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
nc, n = 16, 30000
pts = (torch.rand((nc * n, 3), generator=g) * torch.tensor([3.0, 3.0, 2.5])).to(DEV)
lengths = torch.full((nc,), n, dtype=torch.long, device=DEV)
for r, lim in ((0.0625, 38), (0.2, 38)):
    out = torch.empty((nc * n, lim), dtype=torch.long, device=DEV)
    cnt = torch.empty((nc * n,), dtype=torch.long, device=DEV)
    for _ in range(20):
        ext.radius_search_pack(pts, pts, lengths, lengths, out, r, lim)
        ext.radius_count_pack(pts, pts, lengths, lengths, cnt, r)
    torch.cuda.synchronize()
    print(r, "mean count", cnt.float().mean().item(), "max", cnt.max().item())
