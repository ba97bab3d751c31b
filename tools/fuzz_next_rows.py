#!/usr/bin/env python3
"""Randomised differential run of the SURVEY 8f entry points against the NumPy oracle (shapes drawn at random, small
enough for the oracle).  A hardening aid, run by hand on the GPU box: python tools/fuzz_next_rows.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from vision3d_amd import ext, ops  # noqa: E402

DEV = "cuda:0"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)  # noqa: E731
N_ = lambda t: t.detach().cpu().numpy()  # noqa: E731


def table(rng, m, k, n):
    t = rng.integers(0, max(n, 1), size=(m, k)).astype(np.int64)
    t[rng.random((m, k)) < rng.random() * 0.5] = n
    return t


def one_case(rng, i):
    kind = i % 19
    if kind == 17:  # single-scale ball query (the (a) row): both dtypes, lattices (pairs exactly at the radius), every k class
        dt = np.float32 if rng.integers(4) else np.float64
        b, m, n = int(rng.integers(1, 4)), int(rng.integers(1, 600)), int(rng.integers(1, 5000))
        sup = rng.random((b, n, 3)).astype(dt)
        if rng.integers(2):
            sup = (np.round(sup * 5) / 5).astype(dt)
        qs = (sup[:, rng.integers(0, n, size=m)] + (rng.random((b, m, 3)) < 0.1) * 3.0).astype(dt)
        k = int(rng.choice([1, 3, 16, 32, 63, 64, 65, 128, 200]))
        r = float(rng.choice([0.0, 0.01, 0.1, 0.2, 0.4, 2.0]))
        out = torch.zeros((b, m, k), dtype=torch.long, device=DEV)
        ext.ball_query(T(qs), T(sup), out, r, k)
        return np.array_equal(N_(out), oracle.ball_query(qs, sup, k, r)), f"ball_query {dt.__name__} b{b} m{m} n{n} k{k} r{r}"
    if kind == 18:  # furthest point sampling: sizes around the geometry switches, duplicate-heavy clouds (ties)
        b = int(rng.integers(1, 4))
        n = int(rng.choice([1, 2, 63, 64, 65, 500, 1024, 1025, 2048, 4096, 4097, 8192, 9000, 17000]))
        m = int(rng.integers(1, min(n, 300) + 1))
        pts = rng.random((b, n, 3), dtype=np.float32)
        if rng.integers(3) == 0:
            pts = (np.round(pts * 4) / 4).astype(np.float32)
        want = oracle.furthest_point_sample(pts, m)
        got = ops.furthest_point_sample(T(pts), m, gather_points=False)
        return np.array_equal(N_(got), want), f"fps b{b} n{n} m{m}"
    if kind == 16:  # pairwise distance: the tile sweep (whole tiles, shifted edge tiles), the one-tile kernels, every flag
        b = int(rng.integers(1, 4))
        n, m = int(rng.integers(1, 700)), int(rng.integers(1, 700))
        c = int(rng.choice([3, 32, 33, 64, 96, 160]))
        normalized, squared, strict = bool(rng.integers(2)), bool(rng.integers(2)), rng.integers(4) == 0
        x = rng.standard_normal((b, n, c)).astype(np.float32)
        y = rng.standard_normal((b, m, c)).astype(np.float32)
        if normalized:
            x /= np.maximum(np.linalg.norm(x, axis=-1, keepdims=True), 1e-6)
            y /= np.maximum(np.linalg.norm(y, axis=-1, keepdims=True), 1e-6)
        xd, yd = x.astype(np.float64), y.astype(np.float64)
        xy = np.einsum("bnc,bmc->bnm", xd, yd)
        sq = (xd ** 2).sum(-1)[:, :, None] - 2.0 * xy + (yd ** 2).sum(-1)[:, None, :]
        if strict:
            want = np.maximum(sq, 0.0)  # the difference form ignores `normalized`
        else:
            want = np.maximum((2.0 - 2.0 * xy) if normalized else sq, 0.0)
        got = N_(ops.pairwise_distance(T(x), T(y), normalized=normalized, squared=squared, strict=strict)).astype(np.float64)
        if not squared:
            got = got ** 2 - (0.0 if strict else 1e-8)
        tol = 1e-5 * ((x.astype(np.float64) ** 2).sum(-1)[:, :, None] + (y.astype(np.float64) ** 2).sum(-1)[:, None, :]) + 3e-7
        ok = got.shape == want.shape and bool(np.all(np.abs(got - want) <= tol))
        return ok, f"pairwise b{b} n{n} m{m} c{c} norm{normalized} sq{squared} strict{strict}"
    if kind == 12:  # backward scatters (scatter.hip): every channel-tile / chunk geometry, both dtypes, skewed indices
        dt = np.float32 if rng.integers(4) else np.float64
        b, c = int(rng.integers(1, 4)), int(rng.integers(1, 40))
        n = int(rng.choice([1, 7, 300, 1024, 1025, 4096, 4100]))
        if rng.integers(2):  # group gather
            m, k = int(rng.integers(1, 700)), int(rng.integers(1, 40))
            idx = rng.integers(0, n, size=(b, m, k)).astype(np.int64)
            if rng.integers(3) == 0:
                idx[:] = idx % max(1, n // 50)  # few sources collect everything
            g = rng.standard_normal((b, c, m, k)).astype(dt)
            want = oracle.group_gather_backward(g, idx, n)
            got = torch.empty((b, c, n), dtype=T(g).dtype, device=DEV)
            ext.group_gather_backward(T(g), T(idx), got, n)
            return np.array_equal(N_(got), want), f"group_gather_bwd {dt.__name__} b{b} c{c} n{n} m{m} k{k}"
        q = int(rng.integers(1, 6000))
        idx = rng.integers(0, n, size=(b, q, 3)).astype(np.int64)
        w = rng.random((b, q, 3)).astype(dt)
        g = rng.standard_normal((b, c, q)).astype(dt)
        want = oracle.three_interpolate_backward(g, idx, w, n)
        got = torch.empty((b, c, n), dtype=T(g).dtype, device=DEV)
        ext.three_interpolate_backward(T(g), T(idx), T(w), got, n)
        return np.array_equal(N_(got), want), f"three_interp_bwd {dt.__name__} b{b} c{c} n{n} q{q}"
    if kind == 13:  # pack-mode kNN on the packed arrays against the knn_points oracle per cloud
        b = int(rng.integers(1, 7))
        k = int(rng.choice([1, 2, 5, 16, 23, 24, 40, 63, 64, 100]))
        sl = rng.integers(0, 1500, size=b).astype(np.int64)
        ql = rng.integers(0, 300, size=b).astype(np.int64)
        s = rng.random((int(sl.sum()), 3), dtype=np.float32)
        if rng.integers(3) == 0:
            s = (np.round(s * 6) / 6).astype(np.float32)  # lattice: ties
        qq = rng.random((int(ql.sum()), 3), dtype=np.float32)
        d = torch.full((qq.shape[0], k), -1.0, device=DEV)
        i_ = torch.full((qq.shape[0], k), -7, dtype=torch.long, device=DEV)
        ext.knn_pack(T(qq), T(s), T(ql), T(sl), d, i_, k)
        ok, qo, so = True, 0, 0
        for a, c in zip(ql, sl):
            if a:
                wd, wi = oracle.knn_points(qq[None, qo:qo + a], s[None, so:so + c], k)
                ok = ok and np.array_equal(N_(i_)[qo:qo + a], wi[0]) and np.array_equal(N_(d)[qo:qo + a], wd[0])
            qo, so = qo + a, so + c
        return ok, f"knn_pack b{b} k{k} ql{ql.tolist()} sl{sl.tolist()}"
    if kind == 14:  # radius search with the wide lists (limits 129 .. 512)
        b = int(rng.integers(1, 4))
        sl = rng.integers(0, 2500, size=b).astype(np.int64)
        ql = rng.integers(0, 200, size=b).astype(np.int64)
        s = rng.random((int(sl.sum()), 3), dtype=np.float32)
        if rng.integers(3) == 0:
            s = (np.round(s * 8) / 8).astype(np.float32)
        qq = rng.random((int(ql.sum()), 3), dtype=np.float32)
        limit, r = int(rng.integers(129, 513)), float(rng.choice([0.125, 0.2, 0.35, 0.6]))
        want = oracle.radius_search_pack(qq, s, ql, sl, r, limit)
        got = torch.full((qq.shape[0], limit), -7, dtype=torch.long, device=DEV)
        ext.radius_search_pack(T(qq), T(s), T(ql), T(sl), got, r, limit)
        return np.array_equal(N_(got), want), f"radius_search b{b} limit{limit} r{r} ql{ql.tolist()} sl{sl.tolist()}"
    if kind == 15:  # feature kNN on both routes (wave per row up to k = 64, workgroup per row above)
        b, n, m = int(rng.integers(1, 3)), int(rng.integers(1, 200)), int(rng.integers(1, 1500))
        c, k = int(rng.choice([3, 8, 32, 64, 96])), int(rng.choice([1, 7, 20, 52, 64, 65, 100, 300]))
        x = rng.standard_normal((b, n, c)).astype(np.float32)
        y = rng.standard_normal((b, m, c)).astype(np.float32)
        wd, wi = oracle.knn_feat(x, y, k)
        d = torch.empty((b, n, k), device=DEV)
        i_ = torch.empty((b, n, k), dtype=torch.long, device=DEV)
        ext.knn_feat(T(x), T(y), d, i_, k)
        kk = min(k, m)
        gd, gi = N_(d), N_(i_)
        ok = np.all(gi[..., kk:] == -1) and np.all(np.abs(gd[..., :kk] - wd[..., :kk]) <= 1e-4 * (1 + wd[..., :kk]))
        ok = ok and (gi[..., :kk] != wi[..., :kk]).mean() < 5e-3 and np.all(np.diff(gd[..., :kk], axis=-1) >= 0)
        return bool(ok), f"knn_feat b{b} n{n} m{m} c{c} k{k}"
    if kind == 6:  # threshold selection and global top-k on one matrix (ties on purpose: quantised scores)
        n, m = int(rng.integers(1, 1300)), int(rng.integers(1, 1300))
        score = np.round(rng.standard_normal((n, m)) * rng.choice([1, 4, 50, 1000])).astype(np.float32) / 4
        largest = bool(rng.integers(2))
        thr = float(rng.standard_normal())
        wr, wc, ws = oracle.threshold_select(score, thr, largest)
        idx = torch.empty((2, max(score.size, 1)), dtype=torch.long, device=DEV)
        sc = torch.empty((max(score.size, 1),), device=DEV)
        cnt = torch.zeros((), dtype=torch.long, device=DEV)
        ext.threshold_select(T(score), thr, out_row=idx[0], out_col=idx[1], out_score=sc, num_out=cnt, largest=largest)
        c_ = int(cnt.item())
        ok = c_ == len(wr) and np.array_equal(N_(idx[0, :c_]), wr) and np.array_equal(N_(idx[1, :c_]), wc) and \
            np.array_equal(N_(sc[:c_]), ws)
        k = int(rng.integers(1, min(score.size, 6000) + 1))
        dust = bool(rng.integers(2))
        r, c, v = ops.extract_correspondences_from_scores_topk(T(score), k, has_dustbin=dust, largest=largest,
                                                               return_score=True)
        wr, wc, ws = oracle.extract_correspondences_from_scores_topk(score, k, dust, largest)
        ok = ok and np.array_equal(N_(r), wr) and np.array_equal(N_(c), wc) and np.array_equal(N_(v), ws)
        return ok, f"matchsel n{n} m{m} k{k} l{largest} thr{thr} dust{dust}"
    if kind == 7:  # voxelisation
        b = int(rng.integers(1, 6))
        lengths = rng.integers(0, 3000, size=b).astype(np.int64)
        total = int(lengths.sum())
        c = int(rng.integers(1, 40))
        pts = ((rng.random((total, 3), dtype=np.float32) - 0.5) * np.float32(rng.choice([1, 10, 100]))).astype(np.float32)
        if rng.integers(2) and total:
            pts = np.round(pts * 8) / 8  # points on voxel faces
        feats = rng.standard_normal((total, c)).astype(np.float32)
        vs = float(rng.choice([0.05, 0.125, 0.3, 1.0]))
        want = oracle.voxelize_pack(pts, lengths, vs, feats)
        got = ops.voxelize_pack_mode(T(pts), T(lengths), vs, feats=T(feats))
        ok = all(np.array_equal(N_(g), w) for g, w in zip(got, want))
        return ok, f"voxelize b{b} total{total} c{c} vs{vs}"
    if kind == 8:  # three_nn: every size class of the cell list, clustered / quantised supports
        b, q, s_ = int(rng.integers(1, 4)), int(rng.integers(1, 900)), int(rng.integers(1, 2300))
        sup = rng.random((b, s_, 3), dtype=np.float32) * np.float32(rng.choice([1e-3, 1, 1e3]))
        if rng.integers(2):
            sup = np.round(sup / sup.max() * 6) * np.float32(0.25)
        sup = (sup + np.float32(rng.choice([0, 5, 1e5]))).astype(np.float32)
        lo, hi = sup.min(), sup.max()
        qs = (lo + (rng.random((b, q, 3), dtype=np.float32) * 2 - 0.5) * (hi - lo + 1e-3)).astype(np.float32)
        wd, wi = oracle.three_nn(qs, sup)
        d = torch.empty((b, q, 3), device=DEV)
        i_ = torch.zeros((b, q, 3), dtype=torch.long, device=DEV)
        ext.three_nn(T(qs), T(sup), d, i_)
        ok = np.array_equal(N_(i_), wi) and np.array_equal(N_(d), wd)
        return ok, f"three_nn b{b} q{q} s{s_}"
    if kind == 9:  # multi-scale ball query
        b, m, n = int(rng.integers(1, 4)), int(rng.integers(1, 300)), int(rng.integers(1, 3000))
        sup = rng.random((b, n, 3), dtype=np.float32)
        if rng.integers(2):
            sup = (np.round(sup * 5) / 5).astype(np.float32)
        qs = sup[:, rng.integers(0, n, size=m)] + (rng.random((b, m, 3), dtype=np.float32) < 0.1) * np.float32(3.0)
        qs = qs.astype(np.float32)
        ns = int(rng.integers(1, 5))
        ks = [int(rng.integers(1, 140)) for _ in range(ns)]
        rs = [float(rng.choice([0.01, 0.1, 0.2, 0.4, 2.0])) for _ in range(ns)]
        got = ops.ball_query_multi(T(qs), T(sup), ks, rs)
        ok = all(np.array_equal(N_(g), oracle.ball_query(qs, sup, k, r)) for g, k, r in zip(got, ks, rs))
        return ok, f"ball_multi b{b} m{m} n{n} ks{ks} rs{rs}"
    if kind == 11:  # knn_points: list kernel (k < 24, k >= 64) and bitonic-merge kernel (24 <= k <= 63), ties on lattices
        b, m, n = int(rng.integers(1, 4)), int(rng.integers(1, 200)), int(rng.integers(1, 3000))
        sup = rng.random((b, n, 3), dtype=np.float32) * np.float32(rng.choice([1e-2, 1, 100]))
        if rng.integers(3) == 0:
            sup = (np.round(sup / max(float(sup.max()), 1e-6) * 7) * np.float32(0.125)).astype(np.float32)
        qs = sup[:, rng.integers(0, n, size=m)] + (rng.random((b, m, 3), dtype=np.float32) < 0.5) * np.float32(0.01)
        qs = qs.astype(np.float32)
        k = int(rng.choice([1, 8, 23, 24, 25, 32, 40, 63, 64, 100]))
        wd, wi = oracle.knn_points(qs, sup, k)
        d = torch.zeros((b, m, k), device=DEV)
        idx = torch.zeros((b, m, k), dtype=torch.long, device=DEV)
        ext.knn_points(T(qs), T(sup), d, idx, k)
        ok = np.array_equal(N_(idx), wi) and np.array_equal(N_(d), wd)
        return ok, f"knn_points b{b} m{m} n{n} k{k}"
    if kind == 10:  # pack-mode radius search (every limit class: <= 12 serial, <= 64 bitonic merge, <= 128 list) + count
        b = int(rng.integers(1, 5))
        ql = rng.integers(0, 400, size=b).astype(np.int64)
        sl = rng.integers(0, 2500, size=b).astype(np.int64)
        scale = np.float32(rng.choice([1e-2, 1, 50]))
        off = np.float32(rng.choice([0, -3, 1e3]))
        mode = int(rng.integers(3))

        def cloud(n):
            c = rng.random((n, 3), dtype=np.float32)
            if mode == 1:
                c = np.round(c * 6) / 6  # lattice: (d2, index) ties, pairs exactly at the radius
            if mode == 2:
                c = c * np.array([1, 1, 0.02], np.float32)  # slab: few cells along z, dense cells
            return (c * scale + off).astype(np.float32)

        q = np.concatenate([cloud(int(n)) for n in ql] + [np.zeros((0, 3), np.float32)])
        s_ = np.concatenate([cloud(int(n)) for n in sl] + [np.zeros((0, 3), np.float32)])
        r = float(np.float32(rng.choice([0.0, 0.04, 1 / 6, 0.3, 0.8, 3.0])) * scale)
        lim = int(rng.choice([1, 5, 12, 13, 38, 64, 65, 100, 128]))
        out = torch.full((q.shape[0], lim), -7, dtype=torch.long, device=DEV)
        ext.radius_search_pack(T(q), T(s_), T(ql), T(sl), out, r, lim)
        cnt = torch.full((q.shape[0],), -7, dtype=torch.long, device=DEV)
        ext.radius_count_pack(T(q), T(s_), T(ql), T(sl), cnt, r)
        ok = np.array_equal(N_(out), oracle.radius_search_pack(q, s_, ql, sl, r, lim)) and \
            np.array_equal(N_(cnt), oracle.radius_count_pack(q, s_, ql, sl, r))
        return ok, f"radius b{b} q{ql.tolist()} s{sl.tolist()} r{r} lim{lim} mode{mode} scale{scale} off{off}"
    if kind == 0:  # selection on a matrix
        b, n, m = int(rng.integers(1, 4)), int(rng.integers(1, 400)), int(rng.integers(1, 400))
        k = int(rng.integers(1, min(n, m, 64) + 1))
        score = np.round(rng.standard_normal((b, n, m)) * rng.choice([1, 4, 50])).astype(np.float32) / 4
        largest, mutual = bool(rng.integers(2)), bool(rng.integers(2))
        thr = None if rng.integers(2) else float(rng.standard_normal())
        rm = rng.random((b, n)) > 0.3 if rng.integers(2) else None
        cm = rng.random((b, m)) > 0.3 if rng.integers(2) else None
        want = oracle.mutual_topk_select(score, k, row_masks=rm, col_masks=cm, largest=largest, threshold=thr, mutual=mutual)
        got = ops.batch_mutual_topk_select(T(score), k, row_masks=None if rm is None else T(rm),
                                           col_masks=None if cm is None else T(cm), largest=largest, threshold=thr,
                                           mutual=mutual)
        ok = all(np.array_equal(N_(g), w) for g, w in zip(got, want))
        return ok, f"select b{b} n{n} m{m} k{k} l{largest} mu{mutual} thr{thr}"
    if kind == 1:  # pack gathers
        n, m, k, c = int(rng.integers(1, 600)), int(rng.integers(1, 600)), int(rng.integers(1, 70)), 4 * int(rng.integers(1, 70))
        feats = rng.standard_normal((n, c)).astype(np.float32)
        tb = table(rng, m, k, n)
        out = torch.empty((m, k, c), device=DEV)
        ext.index_select_rows(T(feats), T(tb), out)
        ok = np.array_equal(N_(out), oracle.index_select_rows(feats, tb))
        pooled = torch.empty((m, c), device=DEV)
        argk = torch.empty((m, c), dtype=torch.uint8, device=DEV)
        ext.local_maxpool_pack(T(feats), T(tb), pooled, argk)
        wp, wa = oracle.local_maxpool_pack(feats, tb)
        ok = ok and np.array_equal(N_(pooled), wp) and np.array_equal(N_(argk), wa)
        csr = ext.rows_csr_build(T(tb), n)
        up = rng.standard_normal((m, c)).astype(np.float32)
        gs = torch.empty((n, c), device=DEV)
        ext.rows_scatter(T(up), csr, gs, (m, k), mode=2, argk=argk)
        ok = ok and np.array_equal(N_(gs), oracle.rows_scatter(up, tb, n, mode=2, argk=wa))
        return ok, f"pack n{n} m{m} k{k} c{c}"
    if kind == 2:  # fused grouping
        b, c, n, m, k = (int(rng.integers(1, 5)), int(rng.integers(0, 40)), 4 * int(rng.integers(1, 700)),
                         int(rng.integers(1, 300)), int(rng.integers(1, 40)))
        sp = rng.random((b, 3, n), dtype=np.float32)
        qp = rng.random((b, 3, m), dtype=np.float32)
        idx = rng.integers(0, n, size=(b, m, k)).astype(np.int64)
        sf = rng.standard_normal((b, c, n)).astype(np.float32) if c else None
        out = torch.empty((b, c + 3, m, k), device=DEV)
        ext.sa_group(T(qp), T(sp), T(sf) if c else None, T(idx), out)
        ok = np.array_equal(N_(out), oracle.sa_group(qp, sp, sf, idx))
        if c:
            qf = rng.standard_normal((b, c, m)).astype(np.float32)
            out = torch.empty((b, 2 * c, m, k), device=DEV)
            ext.edge_group(T(qf), T(sf), T(idx), out)
            ok = ok and np.array_equal(N_(out), oracle.edge_group(qf, sf, idx))
            g = torch.empty((b, c, m, k), device=DEV)
            ext.group_gather_forward(T(sf), T(idx), g)
            ok = ok and np.array_equal(N_(g), oracle.group_gather_forward(sf, idx))
        return ok, f"group b{b} c{c} n{n} m{m} k{k}"
    if kind == 3:  # kpconv forward / backward
        n, m, h, k, c = (int(rng.integers(1, 300)), int(rng.integers(1, 300)), int(rng.integers(0, 65)),
                         int(rng.integers(1, 17)), 4 * int(rng.integers(1, 40)))
        s = rng.random((n, 3), dtype=np.float32) * 0.3
        q = rng.random((m, 3), dtype=np.float32) * 0.3
        feats = rng.standard_normal((n, c)).astype(np.float32)
        tb = table(rng, m, h, n)
        kpts = (rng.standard_normal((k, 3)) * 0.05).astype(np.float32)
        sigma = 0.1
        want, wnum, w = oracle.kpconv_gather(q, s, feats, tb, kpts, sigma)
        out = torch.empty((m, k, c), device=DEV)
        num = torch.empty((m,), dtype=torch.long, device=DEV)
        ext.kpconv_gather(T(q), T(s), T(feats), T(tb), T(kpts), out, num, sigma)
        scale = np.matmul(w.astype(np.float64), np.abs(oracle.index_select_rows(feats, tb)).astype(np.float64))
        ok = np.array_equal(N_(num), wnum) and np.all(np.abs(N_(out) - want) <= 1e-5 * scale + 1e-30)
        if h:
            gw = rng.standard_normal((m, k, c)).astype(np.float32)
            gn = torch.empty((m, h, c), device=DEV)
            ext.kpconv_gather_backward(T(q), T(s), T(tb), T(kpts), T(gw), gn, sigma)
            wt = np.transpose(w, (0, 2, 1)).astype(np.float64)
            ok = ok and np.all(np.abs(N_(gn) - np.matmul(wt, gw.astype(np.float64)))
                               <= 1e-5 * np.matmul(wt, np.abs(gw).astype(np.float64)) + 1e-30)
        return ok, f"kpconv n{n} m{m} h{h} k{k} c{c}"
    if kind == 4:  # partition
        n, m = int(rng.integers(1, 1500)), int(rng.integers(1, 200))
        pts = rng.random((n, 3), dtype=np.float32)
        nodes = rng.random((m, 3), dtype=np.float32)
        lim = None if rng.integers(2) else int(rng.integers(1, 64))
        want = oracle.point_to_node_partition(pts, nodes, lim)
        got = ops.point_to_node_partition(T(pts), T(nodes), point_limit=lim, return_count=True)
        ok = all(np.array_equal(N_(g), w) for g, w in zip(got, want))
        return ok, f"partition n{n} m{m} lim{lim}"
    # lists back end with a batch
    b, n, m, k = int(rng.integers(1, 4)), int(rng.integers(1, 200)), int(rng.integers(1, 200)), int(rng.integers(1, 6))
    k = min(k, n, m)
    score = rng.standard_normal((b, n, m)).astype(np.float32)
    ri = np.argsort(score, axis=2, kind="stable")[:, :, :k]
    ci = np.transpose(np.argsort(score, axis=1, kind="stable")[:, :k, :], (0, 2, 1))
    rv = np.take_along_axis(score, ri, axis=2)
    cv = np.take_along_axis(np.transpose(score, (0, 2, 1)), ci, axis=2)
    mutual = bool(rng.integers(2))
    want = oracle.mutual_topk_select(score, k, largest=False, mutual=mutual)
    bound = b * k * (n + m)
    idx = torch.empty((3, bound), dtype=torch.long, device=DEV)
    sc = torch.empty((bound,), device=DEV)
    cnt = torch.zeros((), dtype=torch.long, device=DEV)
    ext.mutual_select_lists(T(ri.astype(np.int64)), T(rv), T(ci.astype(np.int64)), T(cv), m, out_batch=idx[0],
                            out_row=idx[1], out_col=idx[2], out_score=sc, num_out=cnt, largest=False, mutual=mutual)
    c_ = int(cnt.item())
    ok = c_ == len(want[0]) and all(np.array_equal(N_(idx[j, :c_]), want[j]) for j in range(3)) and \
        np.array_equal(N_(sc[:c_]), want[3])
    return ok, f"lists b{b} n{n} m{m} k{k} mu{mutual}"


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for i in range(cases):
        ok, what = one_case(rng, i)
        if not ok:
            bad += 1
            print("MISMATCH", what, flush=True)
        if i % 50 == 49:
            print(f"{i + 1} cases, {bad} mismatches", flush=True)
    torch.cuda.synchronize()
    print(f"done: {cases} cases, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
