"""(GPU box) ball_query time against support count and radius: a radius of 1e-4 never hits (the bare scan), 2.0 fills at
once; then the three MSG scales of config 2 in one multi-scale launch against three single-scale launches."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vision3d_amd import ext
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
from opbench import timeit
g = torch.Generator().manual_seed(0)
for (B, N, M, k, r) in ((32, 64, 1024, 16, 0.1), (32, 1024, 1024, 16, 0.1), (32, 2048, 1024, 16, 0.1), (32, 4096, 1024, 16, 0.1), (32, 4096, 1024, 16, 0.0001), (32, 4096, 1024, 16, 2.0), (32, 8192, 1024, 16, 0.1), (32, 4096, 4096, 16, 0.1),
                        (32, 4096, 1024, 32, 0.2), (32, 4096, 1024, 32, 0.0001), (32, 4096, 1024, 128, 0.4), (32, 4096, 1024, 128, 0.0001)):
    pts = torch.rand((B, N, 3), generator=g).cuda()
    q = torch.rand((B, M, 3), generator=g).cuda()
    bi = torch.zeros((B, M, k), dtype=torch.long, device="cuda")
    med, mn = timeit(lambda: ext.ball_query(q, pts, bi, r, k))
    print(f"B{B} N{N} M{M} k{k} r{r}: {med*1e3:.1f} us (min {mn*1e3:.1f})")
# the three MSG scales of config 2: one multi-scale launch against three single-scale launches
from vision3d_amd import ops
pts = torch.rand((32, 4096, 3), generator=g).cuda()
q = pts[:, :1024].contiguous()
ks, rs = (16, 32, 128), (0.1, 0.2, 0.4)
med, mn = timeit(lambda: ops.ball_query_multi(q, pts, ks, rs))
print(f"ball_query_multi C2: {med*1e3:.1f} us (min {mn*1e3:.1f})")
med, mn = timeit(lambda: [ops.ball_query(q, pts, k, r) for k, r in zip(ks, rs)])
print(f"three ball_query C2: {med*1e3:.1f} us (min {mn*1e3:.1f})")
