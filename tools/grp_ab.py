"""A/B of two builds of the library on fused-grouping shapes (tuning aid): python tools/grp_ab.py <lib.so>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision3d_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from vision3d_amd import ops

def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2] * 1e3

dev = "cuda:0"
g = torch.Generator().manual_seed(0)
for (B, C, N, M, K) in ((32, 64, 4096, 1024, 30), (32, 64, 4096, 1024, 126), (32, 64, 4096, 1024, 32), (32, 64, 4096, 1024, 128), (16, 32, 8192, 2048, 18)):
    sp = torch.rand((B, 3, N), generator=g).to(dev)
    sf = torch.randn((B, C, N), generator=g).to(dev)
    qp = sp[:, :, :M].contiguous()
    qf = sf[:, :, :M].contiguous()
    idx = torch.randint(0, N, (B, M, K), generator=g).to(dev)
    t_sa = timeit(lambda: ops.sa_group(qp, sp, sf, idx))
    t_ed = timeit(lambda: ops.edge_group(qf, sf, idx))
    print(f"B{B} C{C} N{N} M{M} K{K}: sa_group {t_sa:7.1f} us   edge_group {t_ed:7.1f} us", flush=True)
