"""knn_points fp32: batched bitonic-merge kernel (default) against the per-survivor list kernel (V3D_KNN_LIST=1)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision3d_amd import ext  # noqa: E402

DEV = torch.device("cuda:0")


def bench(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t)
    ts.sort()
    return ts[len(ts) // 2] * 1e6


g = torch.Generator().manual_seed(0)
for (B, N, M, k) in ((16, 1024, 512, 32), (32, 4096, 1024, 16), (32, 4096, 1024, 32), (32, 4096, 1024, 3), (32, 4096, 1024, 63),
                     (4, 32768, 4096, 16)):
    s = torch.rand((B, N, 3), generator=g).to(DEV)
    q = s[:, :M].contiguous()
    res = {}
    for mode in ("merge", "list"):
        if mode == "list":
            os.environ["V3D_KNN_LIST"] = "1"
        else:
            os.environ.pop("V3D_KNN_LIST", None)
        d = torch.empty((B, M, k), device=DEV)
        i = torch.empty((B, M, k), dtype=torch.long, device=DEV)
        us = bench(lambda: ext.knn_points(q, s, d, i, k))
        res[mode] = (us, d.clone(), i.clone())
    same = torch.equal(res["merge"][1], res["list"][1]) and torch.equal(res["merge"][2], res["list"][2])
    print(f"B{B} N{N} M{M} k{k}: merge {res['merge'][0]:8.1f} us  list {res['list'][0]:8.1f} us  identical={same}", flush=True)
