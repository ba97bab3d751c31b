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


BATCH = int(os.environ.get("OPBENCH_BATCH", "10"))


def timeit(fn, iters=30, warm=5):
    """Median / minimum time per call over `iters` samples; a sample = BATCH back-to-back calls between one event pair.
    (r01/r02 timed ONE call per pair: an event pair around a single launch also measures the ~4-5 us between the
    dispatch of the start marker and the kernel's first wave, which is a quarter of a 20 us kernel -- rocprofv3's kernel
    trace of `three_interp_fwd` reads 19.5 us where the single-call pair read 24-25.  OPBENCH_BATCH=1 restores that.)"""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(BATCH):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / BATCH)
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
        lists = ext.scatter_build(gi, N)  # what the autograd wrappers do: lists once per index tensor, then sums only
        add(f"group_gather_bwd_apply B{B} C{C} N{N} M{M} K{K}",
            lambda o=out, l=lists, gs=gs: ext.scatter_apply(o, l, gs), nbytes / 1e3, "GB/s")
        add(f"group_gather_bwd_build B{B} C{C} N{N} M{M} K{K}", lambda gi=gi, N=N: ext.scatter_build(gi, N))
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
    l3 = ext.scatter_build(ii, S, w)
    add("three_interp_bwd_apply C2", lambda out=out, l3=l3, gs=gs: ext.scatter_apply(out, l3, gs), nb / 1e3, "GB/s")
    add("three_interp_bwd_build C2", lambda ii=ii, w=w, S=S: ext.scatter_build(ii, S, w))
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
        cnt = torch.empty((nc * n,), dtype=torch.long, device=DEV)
        add(f"radius_count_pack clouds{nc} n{n} r{r}",
            lambda pts=pts, lengths=lengths, cnt=cnt, r=r: ext.radius_count_pack(pts, pts, lengths, lengths, cnt, r),
            nc * n, "Mq/s")
        if r < 0.1:
            from vision3d_amd.ops.radius_search import padded_radius_count_pack_mode
            add(f"radius_count padded brute-force route clouds{nc} n{n} r{r}",
                lambda pts=pts, lengths=lengths, r=r: padded_radius_count_pack_mode(pts, pts, lengths, lengths, r),
                nc * n, "Mq/s")
    # pack-mode kNN (ops/knn.py:110-145): decoder upsampling (fine <- coarse, k = 1 / 3) and same-level neighbourhoods
    for (nc, nq, ns, k) in ((16, 30000, 7500, 1), (16, 30000, 7500, 3), (16, 7500, 7500, 16), (16, 30000, 30000, 16)):
        box = torch.tensor([3.0, 3.0, 2.5])
        qp = (torch.rand((nc * nq, 3), generator=g) * box).to(DEV)
        sp = (torch.rand((nc * ns, 3), generator=g) * box).to(DEV)
        ql = torch.full((nc,), nq, dtype=torch.long, device=DEV)
        sl = torch.full((nc,), ns, dtype=torch.long, device=DEV)
        kd = torch.empty((nc * nq, k), device=DEV)
        ki = torch.empty((nc * nq, k), dtype=torch.long, device=DEV)
        add(f"knn_pack clouds{nc} q{nq} s{ns} k{k}",
            lambda qp=qp, sp=sp, ql=ql, sl=sl, kd=kd, ki=ki, k=k: ext.knn_pack(qp, sp, ql, sl, kd, ki, k), nc * nq, "Mq/s")
    # ops.knn on large 3-D clouds (single cloud, superpoint <- point style): pack cell list vs the batch kernels
    from vision3d_amd import ops as _ops
    for (nq, ns, k) in ((100000, 12500, 1), (12500, 100000, 16), (100000, 100000, 16)):
        qp = torch.rand((nq, 3), generator=g).to(DEV)
        sp = torch.rand((ns, 3), generator=g).to(DEV)
        kd = torch.empty((1, nq, k), device=DEV)
        ki = torch.empty((1, nq, k), dtype=torch.long, device=DEV)
        add(f"knn3d ops.knn q{nq} s{ns} k{k}", lambda qp=qp, sp=sp, k=k: _ops.knn(qp, sp, k), nq, "Mq/s")
        add(f"knn3d knn_lex (batch kernels) q{nq} s{ns} k{k}",
            lambda qp=qp, sp=sp, kd=kd, ki=ki, k=k: ext.knn_lex(qp[None], sp[None], kd, ki, k), nq, "Mq/s")
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
    # ---- mutual top-k selection (native vs the reference's torch formulation on the same GPU) -------
    from vision3d_amd import ops  # noqa: E402

    def torch_mutual_topk(score, k):  # the reference's formulation (ops/mutual_topk_select.py:31-58), for comparison only
        n, m = score.shape
        rt = score.topk(k=k, largest=False, dim=1)[1]
        rm = torch.zeros_like(score, dtype=torch.bool)
        rm[torch.arange(n, device=DEV).view(n, 1).expand(-1, k), rt] = True
        ct = score.topk(k=k, largest=False, dim=0)[1]
        cm = torch.zeros_like(score, dtype=torch.bool)
        cm[ct, torch.arange(m, device=DEV).view(1, m).expand(k, -1)] = True
        corr = torch.logical_and(rm, cm)
        r, c = torch.nonzero(corr, as_tuple=True)
        return r, c, score[r, c]

    for (n, m, k) in ((512, 512, 3), (4096, 4096, 3), (16384, 16384, 1)):
        score = torch.rand((n, m), generator=g).to(DEV)
        add(f"mutual_topk_select N{n} M{m} k{k}", lambda score=score, k=k: ops.mutual_topk_select(score, k, largest=False),
            n * m, "Mcells/s")
        add(f"mutual_topk_select_torchref N{n} M{m} k{k}", lambda score=score, k=k: torch_mutual_topk(score, k),
            n * m, "Mcells/s")
    for (n, m, c) in ((20000, 20000, 256), (100000, 100000, 256)):
        x = torch.nn.functional.normalize(torch.randn((n, c), generator=g), dim=1).to(DEV)
        y = torch.nn.functional.normalize(torch.randn((m, c), generator=g), dim=1).to(DEV)
        add(f"mutual_nn_from_feats N{n} M{m} C{c}",
            lambda x=x, y=y: ops.extract_correspondences_from_feats(x, y, mutual=True), 4.0 * n * m * c / 1e6, "TFLOP/s")
    # ---- the other selectors of ops/matching.py: threshold selection and global top-k (native vs torch formulation) ----
    for (n, k) in ((512, 256), (2048, 256), (8192, 256), (16384, 256)):
        score = torch.randn((n, n), generator=g).to(DEV)
        idx = torch.empty((2, 1 << 20), dtype=torch.long, device=DEV)
        sc = torch.empty((1 << 20,), device=DEV)
        cnt = torch.zeros((), dtype=torch.long, device=DEV)
        add(f"global_topk_select N{n} M{n} k{k}",
            lambda score=score, k=k, idx=idx, sc=sc: ext.global_topk_select(score, k, out_row=idx[0], out_col=idx[1], out_score=sc),
            n * n, "Mcells/s")
        add(f"global_topk_select_torchref N{n} M{n} k{k}", lambda score=score, k=k: score.view(-1).topk(k), n * n, "Mcells/s")
        add(f"threshold_select N{n} M{n} thr3.0",
            lambda score=score, idx=idx, sc=sc, cnt=cnt: ext.threshold_select(score, 3.0, out_row=idx[0], out_col=idx[1],
                                                                              out_score=sc, num_out=cnt),
            n * n, "Mcells/s")
        add(f"threshold_select_torchref N{n} M{n} thr3.0", lambda score=score: torch.nonzero(score > 3.0, as_tuple=True),
            n * n, "Mcells/s")
    # ---- voxelisation (ops/voxelize.py) at the config-4 size, with 32 feature channels ---------------------------------
    for (nc, n, vs, c) in ((16, 30000, 0.025, 32),):
        pts = (torch.rand((nc * n, 3), generator=g) * torch.tensor([3.0, 3.0, 2.5])).to(DEV)
        fts = torch.randn((nc * n, c), generator=g).to(DEV)
        lens = torch.full((nc,), n, dtype=torch.long).to(DEV)
        vox = torch.empty((nc * n, 3), dtype=torch.long, device=DEV)
        nodes, vf = torch.empty_like(pts), torch.empty_like(fts)
        vl, num = torch.empty_like(lens), torch.zeros((), dtype=torch.long, device=DEV)
        add(f"voxelize_pack B{nc} N{n} v{vs} C{c}",
            lambda pts=pts, lens=lens, vox=vox, nodes=nodes, vl=vl, num=num, fts=fts, vf=vf, vs=vs:
            ext.voxelize_pack(pts, lens, vox, nodes, vl, num, vs, feats=fts, voxel_feats=vf), nc * n, "Mpts/s")
    # ---- pack-mode row gathers (KPConv stage shapes: 16 x 30,000 points, 38 neighbours) ------------
    for (n, k, c) in ((480000, 38, 64), (120000, 36, 256)):
        feats = torch.randn((n, c), generator=g).to(DEV)
        table = torch.randint(0, n + 1, (n, k), generator=g).to(DEV)
        # neighbours of a point are near it in memory after grid subsampling: indices within +-2000 rows
        near = ((torch.arange(n).view(n, 1) + torch.randint(-2000, 2000, (n, k), generator=g)) % n).to(DEV)
        out3 = torch.empty((n, k, c), device=DEV)
        out2 = torch.empty((n, c), device=DEV)
        argk = torch.empty((n, c), dtype=torch.uint8, device=DEV)
        w = torch.rand((n, k), generator=g).to(DEV)
        gb_gather = (n * k * (8 + 4 * c) + 4 * n * c) / 1e3  # idx + written rows + source once, in KB -> GB/s via 1e6
        gb_pool = (n * k * 8 + 4 * n * c * 2) / 1e3
        for tname, tb in (("random", table), ("local", near)):
            add(f"index_select_rows {tname} N{n} K{k} C{c}", lambda feats=feats, tb=tb, out3=out3: ext.index_select_rows(feats, tb, out3),
                gb_gather, "GB/s")
            add(f"local_maxpool_pack {tname} N{n} K{k} C{c}",
                lambda feats=feats, tb=tb, out2=out2, argk=argk: ext.local_maxpool_pack(feats, tb, out2, argk), gb_pool, "GB/s(alg)")
            add(f"rows_weighted_sum {tname} N{n} K{k} C{c}",
                lambda feats=feats, tb=tb, out2=out2, w=w: ext.rows_weighted_sum(feats, tb, w, out2), gb_pool, "GB/s(alg)")
        padded = torch.cat([feats, torch.zeros_like(feats[:1])], 0)
        add(f"local_maxpool_torchref local N{n} K{k} C{c}",
            lambda padded=padded, near=near: padded.index_select(0, near.view(-1)).view(near.shape[0], near.shape[1], -1).max(1)[0],
            gb_pool, "GB/s(alg)")
        add(f"rows_csr_build N{n} K{k}", lambda near=near, n=n: ext.rows_csr_build(near, n), n * k, "Mentries/s")
        csr = ext.rows_csr_build(near, n) if (not want or any("rows_scatter" in w_ for w_ in want) or any("N%d" % n in w_ for w_ in want)) else None
        if csr is not None:
            gs = torch.empty((n, c), device=DEV)
            add(f"rows_scatter mode0 local N{n} K{k} C{c}",
                lambda out3=out3, csr=csr, gs=gs, n=n, k=k, c=c: ext.rows_scatter(out3.view(n * k, c), csr, gs, (n * k, 1), mode=0),
                gb_gather, "GB/s")
            add(f"rows_scatter mode2 local N{n} K{k} C{c}",
                lambda out2=out2, csr=csr, gs=gs, argk=argk, n=n, k=k: ext.rows_scatter(out2, csr, gs, (n, k), mode=2, argk=argk),
                gb_pool, "GB/s(alg)")
    # ---- fused grouping vs the reference's three-pass formulation built from our own ops ------------
    for (B, C, N, M, K) in ((32, 64, 4096, 1024, 32), (32, 64, 4096, 1024, 128), (32, 64, 1024, 1024, 20)):
        sp = torch.rand((B, 3, N), generator=g).to(DEV)
        qp = sp[:, :, :M].contiguous()
        sf = torch.randn((B, C, N), generator=g).to(DEV)
        qf = torch.randn((B, C, M), generator=g).to(DEV)
        gi = torch.randint(0, N, (B, M, K), generator=g).to(DEV)
        o_sa = torch.empty((B, C + 3, M, K), device=DEV)
        o_ed = torch.empty((B, 2 * C, M, K), device=DEV)
        by_sa = (B * M * K * 8 + 4 * B * (C + 3) * (N + M * K) + 4 * B * 3 * M) / 1e3
        by_ed = (B * M * K * 8 + 4 * B * C * (N + M) + 4 * B * 2 * C * M * K) / 1e3
        add(f"sa_group B{B} C{C} N{N} M{M} K{K}", lambda qp=qp, sp=sp, sf=sf, gi=gi, o=o_sa: ext.sa_group(qp, sp, sf, gi, o),
            by_sa, "GB/s")

        def three_pass_sa(qp=qp, sp=sp, sf=sf, gi=gi):
            return torch.cat([ops.group_gather(sf, gi), ops.group_gather(sp, gi) - qp.unsqueeze(3)], dim=1)
        add(f"sa_group_threepass B{B} C{C} N{N} M{M} K{K}", three_pass_sa, by_sa, "GB/s")
        add(f"edge_group B{B} C{C} N{N} M{M} K{K}", lambda qf=qf, sf=sf, gi=gi, o=o_ed: ext.edge_group(qf, sf, gi, o),
            by_ed, "GB/s")

        def three_pass_edge(qf=qf, sf=sf, gi=gi):
            nb = ops.group_gather(sf, gi)
            qe = qf.unsqueeze(3).expand_as(nb)
            return torch.cat([qe, nb - qe], dim=1)
        add(f"edge_group_threepass B{B} C{C} N{N} M{M} K{K}", three_pass_edge, by_ed, "GB/s")
    # ---- KPConv aggregation: fused kernel vs the reference formulation in torch (kpconv.py:100-115) --------------------
    for (n, h, c) in ((480000, 38, 64), (120000, 36, 128), (30000, 36, 256)):
        s_pts = (torch.rand((n, 3), generator=g) * torch.tensor([3.0, 3.0, 2.5])).to(DEV)
        feats = torch.randn((n, c), generator=g).to(DEV)
        near = ((torch.arange(n).view(n, 1) + torch.randint(-2000, 2000, (n, h), generator=g)) % n).to(DEV)
        kpts = (torch.randn((15, 3), generator=g) * 0.03).to(DEV)
        wf = torch.empty((n, 15, c), device=DEV)
        num = torch.empty((n,), dtype=torch.long, device=DEV)
        flops = 2.0 * n * h * 15 * c / 1e6
        add(f"kpconv_gather N{n} H{h} C{c}",
            lambda s_pts=s_pts, feats=feats, near=near, kpts=kpts, wf=wf, num=num: ext.kpconv_gather(
                s_pts, s_pts, feats, near, kpts, wf, num, 0.05), flops, "TFLOP/s")
        gw = torch.randn((n, 15, c), generator=g).to(DEV)
        gn = torch.empty((n, h, c), device=DEV)
        add(f"kpconv_gather_bwd N{n} H{h} C{c}",
            lambda s_pts=s_pts, near=near, kpts=kpts, gw=gw, gn=gn: ext.kpconv_gather_backward(
                s_pts, s_pts, near, kpts, gw, gn, 0.05), flops, "TFLOP/s")
        if n <= 120000:
            def torch_form(s_pts=s_pts, feats=feats, near=near, kpts=kpts):
                pp = torch.cat([s_pts, torch.zeros_like(s_pts[:1]) + 1e6], 0)
                nb = pp[near] - s_pts.unsqueeze(1)
                sq = ((nb.unsqueeze(2) - kpts) ** 2).sum(dim=3)
                nw = torch.clamp(1 - torch.sqrt(sq) / 0.05, min=0.0).transpose(1, 2)
                nf = torch.cat([feats, torch.zeros_like(feats[:1])], 0)[near]
                return torch.matmul(nw, nf), torch.gt(nf.sum(-1), 0).sum(-1)
            add(f"kpconv_gather_torchref N{n} H{h} C{c}", torch_form, flops, "TFLOP/s")
    for name, fn, work, unit in cases:
        med, mn = timeit(fn)
        rate = (work / (med * 1e-3) / 1e6) if work else 0.0  # work/1e6 per second
        print(f"{name:48s} median {med*1e3:9.1f} us  min {mn*1e3:9.1f} us   {rate:10.2f} {unit}", flush=True)


if __name__ == "__main__":
    main()
