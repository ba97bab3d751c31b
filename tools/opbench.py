#!/usr/bin/env python3
"""Per-operator micro-benchmark (HIP events, median of N) -- a tuning aid, not the driver's bench.py.
usage: python tools/opbench.py [op-substring ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision3d_amd import ext  # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return float(np.median(ts)), float(np.min(ts))


def main():
    want = sys.argv[1:]
    g = torch.Generator().manual_seed(0)
    cases = []

    def add(name, fn, work=None, unit=""):
        if not want or any(w in name for w in want):
            cases.append((name, fn, work, unit))

    for (B, N, M) in ((16, 1024, 512), (32, 4096, 1024), (8, 16384, 2048), (2, 100000, 1024)):
        pts = torch.rand((B, N, 3), generator=g).to(DEV)
        dist = torch.empty((B, N), device=DEV)
        idx = torch.zeros((B, M), dtype=torch.long, device=DEV)

        def f(pts=pts, dist=dist, idx=idx, M=M):
            dist.fill_(1e10)
            ext.furthest_point_sample(pts, dist, idx, M)
        add(f"fps B{B} N{N} M{M}", f, B * N, "Mpts/s")
    for (B, N, M, k, r) in ((16, 1024, 512, 32, 0.2), (32, 4096, 1024, 16, 0.1), (32, 4096, 1024, 32, 0.2),
                            (32, 4096, 1024, 128, 0.4)):
        pts = torch.rand((B, N, 3), generator=g).to(DEV)
        q = pts[:, :M].contiguous()
        bi = torch.zeros((B, M, k), dtype=torch.long, device=DEV)
        add(f"ball_query B{B} N{N} M{M} k{k}", lambda q=q, pts=pts, bi=bi, r=r, k=k: ext.ball_query(q, pts, bi, r, k),
            B * M, "Mq/s")
        kd = torch.zeros((B, M, k), device=DEV)
        if k <= 100:
            add(f"knn_points B{B} N{N} M{M} k{k}",
                lambda q=q, pts=pts, kd=kd, bi=bi, k=k: ext.knn_points(q, pts, kd, bi, k), B * M, "Mq/s")
    B, Q, S = 32, 4096, 1024
    pts = torch.rand((B, Q, 3), generator=g).to(DEV)
    sp = pts[:, :S].contiguous()
    td = torch.zeros((B, Q, 3), device=DEV)
    ti = torch.zeros((B, Q, 3), dtype=torch.long, device=DEV)
    add("three_nn B32 Q4096 S1024", lambda pts=pts, sp=sp, td=td, ti=ti: ext.three_nn(pts, sp, td, ti), B * Q, "Mq/s")
    for (B, C, N, M, K) in ((16, 64, 1024, 512, 32), (32, 64, 4096, 1024, 16), (32, 64, 4096, 1024, 32),
                            (32, 64, 4096, 1024, 128), (32, 64, 1024, 1024, 20), (32, 3, 4096, 1024, 32)):
        src = torch.randn((B, C, N), generator=g).to(DEV)
        gi = torch.randint(0, N, (B, M, K), generator=g).to(DEV)
        out = torch.empty((B, C, M, K), device=DEV)
        nbytes = B * M * K * 8 + B * C * N * 4 + B * C * M * K * 4
        add(f"group_gather_fwd B{B} C{C} N{N} M{M} K{K}", lambda s=src, gi=gi, o=out: ext.group_gather_forward(s, gi, o),
            nbytes / 1e3, "GB/s")
        gs = torch.empty((B, C, N), device=DEV)
        add(f"group_gather_bwd B{B} C{C} N{N} M{M} K{K}",
            lambda o=out, gi=gi, gs=gs, N=N: ext.group_gather_backward(o, gi, gs, N), nbytes / 1e3, "GB/s")
    B, C, S, Q = 32, 128, 1024, 4096
    src = torch.randn((B, C, S), generator=g).to(DEV)
    ii = torch.randint(0, S, (B, Q, 3), generator=g).to(DEV)
    w = torch.rand((B, Q, 3), generator=g).to(DEV)
    out = torch.empty((B, C, Q), device=DEV)
    gs = torch.empty((B, C, S), device=DEV)
    nb = B * C * S * 4 + B * Q * 3 * 12 + B * C * Q * 4
    add("three_interp_fwd C2", lambda src=src, ii=ii, w=w, out=out: ext.three_interpolate_forward(src, ii, w, out),
        nb / 1e3, "GB/s")
    add("three_interp_bwd C2", lambda out=out, ii=ii, w=w, gs=gs, S=S: ext.three_interpolate_backward(out, ii, w, gs, S),
        nb / 1e3, "GB/s")
    for (B, N, M, C) in ((32, 1024, 1024, 64), (1, 8192, 8192, 256), (1, 16384, 16384, 64)):
        x = torch.randn((B, N, C), generator=g).to(DEV)
        y = torch.randn((B, M, C), generator=g).to(DEV)
        o = torch.empty((B, N, M), device=DEV)
        add(f"pairwise B{B} N{N} M{M} C{C}", lambda x=x, y=y, o=o: ext.pairwise_distance(x, y, o),
            2.0 * B * N * M * C / 1e6, "TFLOP/s")
    for (B, N, M, C, k) in ((32, 1024, 1024, 64, 20), (1, 12500, 100000, 256, 3), (1, 100000, 12500, 256, 1)):
        x = torch.randn((B, N, C), generator=g).to(DEV)
        y = torch.randn((B, M, C), generator=g).to(DEV)
        d = torch.empty((B, N, k), device=DEV)
        i = torch.empty((B, N, k), dtype=torch.long, device=DEV)
        add(f"knn_feat B{B} N{N} M{M} C{C} k{k}", lambda x=x, y=y, d=d, i=i, k=k: ext.knn_feat(x, y, d, i, k),
            2.0 * B * N * M * C / 1e6, "TFLOP/s")
    # config 4 (KPConv / GeoTransformer pyramid stage): 16 clouds of 30,000 points in a 3 x 3 x 2.5 m box, pack mode
    for (nc, n, r, lim) in ((16, 30000, 0.0625, 38), (16, 30000, 0.2, 38)):
        pts = (torch.rand((nc * n, 3), generator=g) * torch.tensor([3.0, 3.0, 2.5])).to(DEV)
        lengths = torch.full((nc,), n, dtype=torch.long, device=DEV)
        out = torch.empty((nc * n, lim), dtype=torch.long, device=DEV)
        add(f"radius_search_pack clouds{nc} n{n} r{r} limit{lim}",
            lambda pts=pts, lengths=lengths, out=out, r=r, lim=lim: ext.radius_search_pack(pts, pts, lengths, lengths,
                                                                                           out, r, lim),
            nc * n, "Mq/s")
    for (nc, n, vs) in ((16, 30000, 0.025), (16, 30000, 0.1)):
        pts = (torch.rand((nc * n, 3), generator=g) * torch.tensor([3.0, 3.0, 2.5])).to(DEV)
        lengths = torch.full((nc,), n, dtype=torch.long, device=DEV)
        o = torch.empty_like(pts)
        ol = torch.empty_like(lengths)
        num = torch.zeros((), dtype=torch.long, device=DEV)
        add(f"grid_subsample_pack clouds{nc} n{n} voxel{vs}",
            lambda pts=pts, lengths=lengths, o=o, ol=ol, num=num, vs=vs: ext.grid_subsample_pack(pts, lengths, o, ol,
                                                                                                 num, vs),
            nc * n, "Mpts/s")
    for name, fn, work, unit in cases:
        med, mn = timeit(fn)
        rate = (work / (med * 1e-3) / 1e6) if work else 0.0  # work/1e6 per second
        print(f"{name:48s} median {med*1e3:9.1f} us  min {mn*1e3:9.1f} us   {rate:10.2f} {unit}", flush=True)


if __name__ == "__main__":
    main()
