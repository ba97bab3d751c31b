"""GPU parity of the SURVEY 8f "next" rows (ranks 2-4) through the C ABI, against oracle/next_rows.py and the
reference-generated fixtures.  Indices / masks bit-exact; scores are copies of the inputs (bit-exact)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def ext():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from vision3d_amd import _lib, ext as ext_mod

    assert os.path.exists(_lib.LIB_PATH), "libv3d_hip.so must be built in-tree before the GPU run"
    _lib.lib()
    return ext_mod


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def N_(t):
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------------------------------------------
# rank 2: mutual top-k selection
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_mutual_topk_select_vs_reference_fixture(ext, golden, name):
    from vision3d_amd import ops

    g = golden("ref_select.npz")
    score, k = T(g[f"{name}_score"]), int(g[f"{name}_k"])
    rmask, cmask = T(g[f"{name}_rmask"]), T(g[f"{name}_cmask"])
    for largest in (True, False):
        for mutual in (True, False):
            for thr in (None, 0.25 if largest else -0.25):
                tag = f"{name}_l{int(largest)}_m{int(mutual)}_t{int(thr is not None)}"
                got = ops.batch_mutual_topk_select(score, k, largest=largest, threshold=thr, mutual=mutual,
                                                   reduce_result=False)
                assert got.dtype == torch.bool and np.array_equal(N_(got), g[tag]), tag
                got = ops.batch_mutual_topk_select(score, k, row_masks=rmask, col_masks=cmask, largest=largest,
                                                   threshold=thr, mutual=mutual, reduce_result=False)
                assert np.array_equal(N_(got), g[tag + "_masked"]), tag
                # the compacted form is torch.nonzero of the same matrix, with the scores gathered
                b, r, c, s = ops.batch_mutual_topk_select(score, k, row_masks=rmask, col_masks=cmask, largest=largest,
                                                          threshold=thr, mutual=mutual)
                wb, wr, wc = np.nonzero(g[tag + "_masked"])
                assert np.array_equal(N_(b), wb) and np.array_equal(N_(r), wr) and np.array_equal(N_(c), wc), tag
                assert np.array_equal(N_(s), g[f"{name}_score"][wb, wr, wc]), tag
    b, r, c, s = ops.batch_mutual_topk_select(score, k)
    assert np.array_equal(np.stack([N_(b), N_(r), N_(c)]), g[f"{name}_reduced"])
    assert np.array_equal(N_(s), g[f"{name}_reduced_scores"])


def test_mutual_topk_select_2d_signature(ext, golden):
    from vision3d_amd import ops

    g = golden("ref_select.npz")
    r, c, s = ops.mutual_topk_select(T(g["a_score"][0]), 3, largest=False, mutual=False)
    assert np.array_equal(np.stack([N_(r), N_(c)]), g["a2d_reduced"])
    assert np.array_equal(N_(s), g["a2d_reduced_scores"])
    corr = ops.mutual_topk_select(T(g["a_score"][0]), 3, reduce_result=False)
    assert corr.shape == (37, 53) and np.array_equal(N_(corr), g["a_l1_m1_t0"][0])
    with pytest.raises(RuntimeError, match="out of range"):
        ops.mutual_topk_select(T(g["a_score"][0]), 38)


@pytest.mark.parametrize("shape", [(1, 1, 1, 1), (2, 5, 300, 5), (1, 700, 3, 2), (4, 257, 129, 7), (1, 1500, 1100, 64),
                                   (8, 128, 128, 1), (1, 3000, 70, 3)])
def test_mutual_topk_select_vs_oracle_ties_and_infinities(ext, oracle_mod, shape):
    from vision3d_amd import ops

    b, n, m, k = shape
    rng = np.random.default_rng(sum(shape))
    # quantised scores: massive ties; plus -inf / +inf entries (log-space score matrices are masked with -inf)
    score = np.round(rng.standard_normal((b, n, m)) * 4).astype(np.float32) / 4
    score[rng.random((b, n, m)) < 0.05] = -np.inf
    score[rng.random((b, n, m)) < 0.01] = np.inf
    score[rng.random((b, n, m)) < 0.02] = -0.0
    for largest in (True, False):
        for mutual in (True, False):
            thr = 0.5 if largest else -0.5
            want = oracle_mod.mutual_topk_corr_mat(score, k, largest=largest, threshold=thr, mutual=mutual)
            got = ops.batch_mutual_topk_select(T(score), k, largest=largest, threshold=thr, mutual=mutual,
                                               reduce_result=False)
            assert np.array_equal(N_(got), want), (largest, mutual)
            want = oracle_mod.mutual_topk_select(score, k, largest=largest, mutual=mutual)
            got = ops.batch_mutual_topk_select(T(score), k, largest=largest, mutual=mutual)
            for w, g_ in zip(want, got):
                assert np.array_equal(N_(g_), w), (largest, mutual)


def test_mutual_select_lists_matches_matrix_route(ext, oracle_mod):
    """Feature route (two fused kNN passes + list selection) == matrix route on the same descriptors."""
    from vision3d_amd import ops

    rng = np.random.default_rng(5)
    x = rng.standard_normal((900, 32)).astype(np.float32)
    y = np.concatenate([x[:600] + 0.05 * rng.standard_normal((600, 32)).astype(np.float32),
                        rng.standard_normal((500, 32)).astype(np.float32)])
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    y /= np.linalg.norm(y, axis=1, keepdims=True)
    d = ((x[:, None, :].astype(np.float64) - y[None].astype(np.float64)) ** 2).sum(-1)
    for mutual in (True, False):
        _, wr, wc, _ = oracle_mod.mutual_topk_select(d[None].astype(np.float32), 1, largest=False, mutual=mutual)
        r, c, fd = ops.extract_correspondences_from_feats(T(x), T(y), mutual=mutual, return_feat_dist=True)
        assert np.array_equal(N_(r), wr) and np.array_equal(N_(c), wc)
        # squared feature distances: |d - d_ref| <= 1e-5 * (|x|^2 + |y|^2) = 2e-5 (unit descriptors)
        assert np.abs(N_(fd) - d[wr, wc]).max() <= 2e-5
        r2, c2 = ops.extract_correspondences_from_feats(T(x), T(y), mutual=mutual)
        assert np.array_equal(N_(r2), wr) and np.array_equal(N_(c2), wc)


# ------------------------------------------------------------------------------------------------------
# rank 3: pack-mode row gathers
# ------------------------------------------------------------------------------------------------------
def _table(rng, m, k, n, none_frac=0.2):
    t = rng.integers(0, n, size=(m, k)).astype(np.int64)
    t[rng.random((m, k)) < none_frac] = n
    if m > 3:
        t[3] = n
    return t


def test_pack_gathers_vs_reference_fixture(ext, golden):
    from vision3d_amd import ops

    g = golden("ref_pack_ops.npz")
    feats, table = T(g["feats"]), T(g["table"])
    padded = torch.cat([feats, torch.zeros_like(feats[:1])], 0)
    assert np.array_equal(N_(ops.index_select(padded, table, dim=0)), g["index_select"])
    assert np.array_equal(N_(ops.index_select(feats, table, dim=0)), g["index_select"])  # the zero row is implicit
    assert np.array_equal(N_(ops.local_maxpool_pack_mode(feats, table)), g["maxpool"])
    assert np.array_equal(N_(ops.nearest_interpolate_pack_mode(feats, table)), g["nearest"])
    got = ops.global_avgpool_pack_mode(feats, T(g["lengths"]))
    assert np.abs(N_(got) - g["avgpool"]).max() <= 1e-5 * np.abs(g["feats"]).max()
    # backward against the reference's autograd results (1e-5 of the summed magnitude: its order is not ours)
    scale = np.abs(g["index_select_upstream"]).max() * g["table"].shape[1]
    f = feats.clone().requires_grad_(True)
    ops.index_select(f, table, dim=0).backward(T(g["index_select_upstream"]))
    assert np.abs(N_(f.grad) - g["index_select_grad"]).max() <= 1e-5 * scale
    f = feats.clone().requires_grad_(True)
    ops.local_maxpool_pack_mode(f, table).backward(T(g["maxpool_upstream"]))
    assert np.abs(N_(f.grad) - g["maxpool_grad"]).max() <= 1e-5 * scale
    # a NON-contiguous input that requires grad (a transposed activation): same gradient, in the input's layout
    ft = feats.t().contiguous().requires_grad_(True)  # (C, N); ft.t() is the (N, C) view handed to the op
    ops.local_maxpool_pack_mode(ft.t(), table).backward(T(g["maxpool_upstream"]))
    assert np.abs(N_(ft.grad.t()) - g["maxpool_grad"]).max() <= 1e-5 * scale


@pytest.mark.parametrize("n,m,k,c", [(1, 1, 1, 4), (100, 77, 5, 4), (1000, 513, 38, 64), (5000, 3000, 16, 128),
                                     (300, 2000, 7, 20), (2000, 100, 64, 256), (700, 900, 3, 1028)])
def test_pack_gathers_vs_oracle_bit_exact(ext, oracle_mod, n, m, k, c):
    rng = np.random.default_rng(n + m + k + c)
    feats = rng.standard_normal((n, c)).astype(np.float32)
    feats[rng.random((n, c)) < 0.05] = 0.0  # ties against the implicit zero row
    table = _table(rng, m, k, n)
    tf, tt = T(feats), T(table)
    # gather (16-byte path) and its first-column form
    out = torch.empty((m, k, c), device=DEV)
    ext.index_select_rows(tf, tt, out)
    assert np.array_equal(N_(out), oracle_mod.index_select_rows(feats, table))
    out1 = torch.empty((m, c), device=DEV)
    ext.index_select_rows(tf, tt, out1, index_stride=k)
    assert np.array_equal(N_(out1), oracle_mod.nearest_interpolate_pack(feats, table))
    # max pool + argk
    pooled = torch.empty((m, c), device=DEV)
    argk = torch.empty((m, c), dtype=torch.uint8, device=DEV)
    ext.local_maxpool_pack(tf, tt, pooled, argk)
    wp, wa = oracle_mod.local_maxpool_pack(feats, table)
    assert np.array_equal(N_(pooled), wp) and np.array_equal(N_(argk), wa)
    # inverse-distance weights + weighted sum (bit-identical: same operation order, no contraction)
    s_pts = rng.random((n, 3), dtype=np.float32)
    q_pts = rng.random((m, 3), dtype=np.float32)
    ku = max(1, k - 2)
    w = torch.empty((m, ku), device=DEV)
    ext.knn_interp_weights_pack(T(q_pts), T(s_pts), tt, w, 1e-8)
    ww = oracle_mod.knn_interp_weights_pack(q_pts, s_pts, table, k=ku, eps=1e-8)
    assert np.array_equal(N_(w), ww)
    o = torch.empty((m, c), device=DEV)
    ext.rows_weighted_sum(tf, tt, w, o)
    assert np.array_equal(N_(o), oracle_mod.rows_weighted_sum(feats, table, ww))
    # the three backward modes: equal to the sequential loop bit for bit
    csr = ext.rows_csr_build(tt, n)
    up = rng.standard_normal((m, k, c)).astype(np.float32)
    gs = torch.empty((n, c), device=DEV)
    ext.rows_scatter(T(up).view(m * k, c), csr, gs, (m * k, 1), mode=0)
    assert np.array_equal(N_(gs), oracle_mod.rows_scatter(up, table, n, mode=0))
    up2 = rng.standard_normal((m, c)).astype(np.float32)
    ext.rows_scatter(T(up2), csr, gs, (m, k), mode=1, weights=w)
    assert np.array_equal(N_(gs), oracle_mod.rows_scatter(up2, table, n, mode=1, weights=ww))
    ext.rows_scatter(T(up2), csr, gs, (m, k), mode=2, argk=argk)
    assert np.array_equal(N_(gs), oracle_mod.rows_scatter(up2, table, n, mode=2, argk=wa))


@pytest.mark.parametrize("dtype,c", [(np.int64, 3), (np.float64, 5), (np.int32, 7), (np.uint8, 13), (np.int16, 6),
                                     (np.float32, 3)])
def test_index_select_rows_any_dtype_and_row_width(ext, oracle_mod, dtype, c):
    rng = np.random.default_rng(c)
    src = (rng.standard_normal((200, c)) * 50).astype(dtype)
    table = _table(rng, 150, 4, 200)
    out = torch.empty((150, 4, c), dtype=T(src).dtype, device=DEV)
    ext.index_select_rows(T(src), T(table), out)
    assert np.array_equal(N_(out), oracle_mod.index_select_rows(src, table))


def test_pack_ops_wrappers_and_autograd(ext, oracle_mod):
    from vision3d_amd import ops

    rng = np.random.default_rng(11)
    n, m, k, c = 400, 350, 12, 32
    feats = rng.standard_normal((n, c)).astype(np.float32)
    table = _table(rng, m, k, n)
    s_pts, q_pts = rng.random((n, 3), dtype=np.float32), rng.random((m, 3), dtype=np.float32)
    up = rng.standard_normal((m, c)).astype(np.float32)
    f = T(feats).requires_grad_(True)
    out = ops.knn_interpolate_pack_mode(T(q_pts), T(s_pts), f, T(table), k=5)
    assert np.array_equal(N_(out), oracle_mod.knn_interpolate_pack(q_pts, s_pts, feats, table, k=5))
    out.backward(T(up))
    ww = oracle_mod.knn_interp_weights_pack(q_pts, s_pts, table, k=5)
    assert np.array_equal(N_(f.grad), oracle_mod.rows_scatter(up, table, n, mode=1, weights=ww))
    # the CSR is cached on the table tensor and reused by every operator that shares the table
    tt = T(table)
    f2 = T(feats).requires_grad_(True)
    ops.local_maxpool_pack_mode(f2, tt).sum().backward()
    first = tt._v3d_csr[1]
    f3 = T(feats).requires_grad_(True)
    ops.local_maxpool_pack_mode(f3, tt).sum().backward()
    assert tt._v3d_csr[1] is first and torch.equal(f2.grad, f3.grad)
    # global average pooling: mean per cloud, gradient = upstream / length on every row of the cloud
    lengths = torch.tensor([100, 250, 50], device=DEV)
    f4 = T(feats).requires_grad_(True)
    pooled = ops.global_avgpool_pack_mode(f4, lengths)
    want = oracle_mod.global_avgpool_pack(feats, [100, 250, 50])
    assert np.abs(N_(pooled) - want).max() <= 1e-5 * np.abs(feats).max()
    pooled.sum().backward()
    assert torch.allclose(f4.grad[:100], torch.full((100, c), 1 / 100, device=DEV))
    # generic index_select semantics outside the native layout stay the reference's
    x = T(rng.standard_normal((4, 6, 5)).astype(np.float32))
    idx = torch.tensor([[0, 2], [5, 1]], device=DEV)
    assert ops.index_select(x, idx, dim=1).shape == (4, 2, 2, 5)
    with pytest.raises(RuntimeError, match="CUDA"):
        ops.local_maxpool_pack_mode(torch.zeros(4, 4), torch.zeros(2, 2, dtype=torch.long))


# ------------------------------------------------------------------------------------------------------
# rank 4: fused grouping (set abstraction / EdgeConv MLP inputs)
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("b,c,n,m,k", [(2, 0, 64, 16, 4), (3, 5, 100, 33, 7), (8, 64, 1024, 512, 32),
                                       (2, 16, 4096, 1024, 16), (1, 6, 8192, 700, 20), (2, 3, 10000, 300, 9),
                                       (4, 32, 1000, 1000, 20)])
def test_fused_group_bit_exact(ext, oracle_mod, b, c, n, m, k):
    rng = np.random.default_rng(b + c + n + m + k)
    s_points = rng.random((b, 3, n), dtype=np.float32)
    q_points = np.ascontiguousarray(s_points[:, :, :m]) if m <= n else rng.random((b, 3, m), dtype=np.float32)
    idx = rng.integers(0, n, size=(b, m, k)).astype(np.int64)
    s_feats = rng.standard_normal((b, c, n)).astype(np.float32) if c else None
    out = torch.empty((b, c + 3, m, k), device=DEV)
    ext.sa_group(T(q_points), T(s_points), T(s_feats) if c else None, T(idx), out)
    assert np.array_equal(N_(out), oracle_mod.sa_group(q_points, s_points, s_feats, idx))
    if c:
        q_feats = rng.standard_normal((b, c, m)).astype(np.float32)
        out = torch.empty((b, 2 * c, m, k), device=DEV)
        ext.edge_group(T(q_feats), T(s_feats), T(idx), out)
        assert np.array_equal(N_(out), oracle_mod.edge_group(q_feats, s_feats, idx))


def test_fused_group_equals_reference_formulation_and_autograd(ext):
    """The reference formulation itself (group_gather + subtract + cat, pointnet2.py:44-47 / edge_conv.py:59-61) built
    from the hot-path ops, forward and backward."""
    from vision3d_amd import ops

    rng = np.random.default_rng(21)
    b, c, n, m, k = 2, 8, 512, 128, 16
    s_points = T(rng.random((b, 3, n), dtype=np.float32)).requires_grad_(True)
    q_points = T(rng.random((b, 3, m), dtype=np.float32)).requires_grad_(True)
    s_feats = T(rng.standard_normal((b, c, n)).astype(np.float32)).requires_grad_(True)
    q_feats = T(rng.standard_normal((b, c, m)).astype(np.float32)).requires_grad_(True)
    idx = T(rng.integers(0, n, size=(b, m, k)).astype(np.int64))
    up = T(rng.standard_normal((b, c + 3, m, k)).astype(np.float32))
    ref = torch.cat([ops.group_gather(s_feats, idx), ops.group_gather(s_points, idx) - q_points.unsqueeze(3)], dim=1)
    ref.backward(up)
    want = [t.grad.clone() for t in (q_points, s_points, s_feats)]
    for t in (q_points, s_points, s_feats):
        t.grad = None
    got = ops.sa_group(q_points, s_points, s_feats, idx)
    assert torch.equal(got, ref)
    got.backward(up)
    assert torch.equal(s_points.grad, want[1]) and torch.equal(s_feats.grad, want[2])
    assert torch.allclose(q_points.grad, want[0], rtol=1e-5, atol=1e-5)
    s_feats.grad = None
    up2 = T(rng.standard_normal((b, 2 * c, m, k)).astype(np.float32))
    g = ops.group_gather(s_feats, idx)
    qe = q_feats.unsqueeze(3).expand_as(g)
    ref = torch.cat([qe, g - qe], dim=1)
    ref.backward(up2)
    want = [q_feats.grad.clone(), s_feats.grad.clone()]
    q_feats.grad = s_feats.grad = None
    got = ops.edge_group(q_feats, s_feats, idx)
    assert torch.equal(got, ref)
    got.backward(up2)
    assert torch.equal(s_feats.grad, want[1]) and torch.allclose(q_feats.grad, want[0], rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------------------------------------------
# KPConv aggregation (fused gather + kernel-point influences + weighted sum)
# ------------------------------------------------------------------------------------------------------
def _kpconv_case(rng, n, m, h, k, c, radius=0.12):
    s = rng.random((n, 3), dtype=np.float32)
    q = np.ascontiguousarray(s[:m]) if m <= n else rng.random((m, 3), dtype=np.float32)
    # neighbours = points within reach of the kernel, so that the influences are not all zero
    d = np.linalg.norm(q[:, None, :] - s[None], axis=-1)
    order = np.argsort(d, axis=1)[:, :h]
    table = np.where(np.take_along_axis(d, order, 1) < radius * 1.5, order, n).astype(np.int64)
    if table.shape[1] < h:
        table = np.concatenate([table, np.full((m, h - table.shape[1]), n, dtype=np.int64)], axis=1)
    feats = rng.standard_normal((n, c)).astype(np.float32)
    feats[rng.random(n) < 0.2] = 0.0  # rows that do not count for the density normalisation
    kpts = (rng.standard_normal((k, 3)) * radius * 0.5).astype(np.float32)
    kpts[0] = 0
    return q, s, feats, table, kpts


@pytest.mark.parametrize("n,m,h,k,c", [(300, 200, 38, 15, 64), (500, 500, 17, 15, 128), (64, 40, 64, 16, 32),
                                       (400, 333, 5, 7, 4), (1000, 800, 30, 15, 256), (200, 150, 1, 1, 68),
                                       (50, 30, 0, 15, 8)])
def test_kpconv_gather_vs_oracle(ext, oracle_mod, n, m, h, k, c):
    rng = np.random.default_rng(n + m + h + k + c)
    q, s, feats, table, kpts = _kpconv_case(rng, n, m, h, k, c)
    sigma = 0.06
    want, want_num, w = oracle_mod.kpconv_gather(q, s, feats, table, kpts, sigma)
    out = torch.full((m, k, c), float("nan"), device=DEV)
    num = torch.zeros((m,), dtype=torch.long, device=DEV)
    ext.kpconv_gather(T(q), T(s), T(feats), T(table), T(kpts), out, num, sigma)
    assert np.array_equal(N_(num), want_num)
    # tolerance: 1e-5 of the summed magnitude sum_h w |f| (the order of the sum over neighbours is the MFMA's)
    scale = np.matmul(w.astype(np.float64), np.abs(oracle_mod.index_select_rows(feats, table)).astype(np.float64))
    assert np.all(np.abs(N_(out).astype(np.float64) - want) <= 1e-5 * scale + 1e-30)


def test_kpconv_layer_forward_backward_matches_reference_formulation(ext, oracle_mod):
    """ops.kpconv against the reference's formulation (kpconv.py:100-149) written with torch on the same device."""
    from vision3d_amd import ops

    rng = np.random.default_rng(77)
    n, m, h, k, c, d = 600, 450, 24, 15, 32, 48
    q, s, feats, table, kpts = _kpconv_case(rng, n, m, h, k, c)
    sigma, inf = 0.06, 1e6
    tq, ts, tt, tk = T(q), T(s), T(table), T(kpts)
    weights = T((rng.standard_normal((k, c, d)) * 0.1).astype(np.float32)).requires_grad_(True)
    bias = T(rng.standard_normal(d).astype(np.float32))
    up = T(rng.standard_normal((m, d)).astype(np.float32))

    f_ref = T(feats).requires_grad_(True)
    padded_points = torch.cat([ts, torch.zeros_like(ts[:1]) + inf], 0)
    neighbors = padded_points[tt] - tq.unsqueeze(1)
    sq = ((neighbors.unsqueeze(2) - tk) ** 2).sum(dim=3)
    nw = torch.clamp(1 - torch.sqrt(sq) / sigma, min=0.0).transpose(1, 2)
    padded_feats = torch.cat([f_ref, torch.zeros_like(f_ref[:1])], 0)
    nf = padded_feats[tt]
    ref = torch.einsum("mkc,kcd->md", torch.matmul(nw, nf), weights)
    num = torch.gt(nf.sum(dim=-1), 0.0).sum(dim=-1)
    ref = ref / torch.max(num, torch.ones_like(num)).unsqueeze(1) + bias
    ref.backward(up)
    want_gf, want_gw = f_ref.grad.clone(), weights.grad.clone()
    weights.grad = None

    f = T(feats).requires_grad_(True)
    got = ops.kpconv(tq, ts, f, tt, tk, weights, sigma, bias=bias)
    assert torch.allclose(got, ref, rtol=1e-4, atol=1e-5)
    got.backward(up)
    assert torch.allclose(f.grad, want_gf, rtol=1e-4, atol=1e-5)
    assert torch.allclose(weights.grad, want_gw, rtol=1e-4, atol=1e-5)
    # shapes outside the fused kernel (C not a multiple of 4) keep working through the row-gather formulation
    f5 = T(rng.standard_normal((n, 5)).astype(np.float32))
    w5, n5 = ops.kpconv_gather(tq, ts, f5, tt, tk, sigma)
    want5, num5, _ = oracle_mod.kpconv_gather(q, s, N_(f5), table, kpts, sigma)
    assert np.allclose(N_(w5), want5, rtol=1e-4, atol=1e-5) and np.array_equal(N_(n5), num5)


# ------------------------------------------------------------------------------------------------------
# point-to-node partition
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,m,limit", [(1, 1, None), (500, 7, None), (3000, 64, 128), (5000, 200, 16), (2000, 1500, 4),
                                       (4096, 33, 64)])
def test_point_to_node_partition_vs_oracle(ext, oracle_mod, n, m, limit):
    from vision3d_amd import ops

    rng = np.random.default_rng(n + m)
    points = rng.random((n, 3), dtype=np.float32)
    nodes = points[rng.permutation(n)[:m]] + (0.01 * rng.standard_normal((m, 3))).astype(np.float32) if m <= n else \
        rng.random((m, 3), dtype=np.float32)
    if n >= 500:  # exact ties: duplicated points (equal distances inside a node) and a duplicated node
        points[10:20] = points[0:10]
        if m > 3:
            nodes[3] = nodes[2]
    p2n, sizes, masks, knn, kmask = oracle_mod.point_to_node_partition(points, nodes, limit)
    got = ops.point_to_node_partition(T(points), T(nodes), point_limit=limit, return_count=True)
    assert len(got) == 5
    assert np.array_equal(N_(got[0]), p2n) and np.array_equal(N_(got[1]), sizes) and np.array_equal(N_(got[2]), masks)
    assert got[3].dtype == torch.long and got[4].dtype == torch.bool
    assert np.array_equal(N_(got[3]), knn) and np.array_equal(N_(got[4]), kmask)
    only = ops.point_to_node_partition(T(points), T(nodes), gather_points=False)
    assert len(only) == 1 and np.array_equal(N_(only[0]), p2n)


def test_knn_ops_vs_reference_fixture(ext, golden):
    """ops/knn_interpolate.py and ops/point_cloud_partition.py as the reference's own files computed them on the CPU
    (tests/golden/ref_knn_ops.npz; their pykeops search replaced by the reference's pairwise_distance + topk)."""
    from vision3d_amd import ops

    g = golden("ref_knn_ops.npz")
    q, s, f, table = T(g["ki_q"]), T(g["ki_s"]), T(g["ki_feats"]), T(g["ki_table"])
    tol = 1e-5 * float(np.abs(g["ki_feats"]).max())
    for k_used in (None, 3):
        got = ops.knn_interpolate_pack_mode(q, s, f, table, k=k_used)
        assert np.abs(N_(got) - g[f"ki_pack_k{k_used or 0}"]).max() <= tol
    for lim, tag in ((None, 0), (0.15, 1)):
        got = ops.knn_interpolate(q, s, f, k=3, distance_limit=lim)
        assert np.abs(N_(got) - g[f"ki_knn_lim{tag}"]).max() <= tol
    pts, nodes = T(g["p2n_points"]), T(g["p2n_nodes"])
    for tag, limit in (("all", None), ("lim", 20)):
        res = ops.point_to_node_partition(pts, nodes, point_limit=limit, return_count=True)
        for key, got in zip(("p2n", "sizes", "masks", "knn_idx", "knn_masks"), res):
            assert np.array_equal(N_(got), g[f"p2n_{tag}_{key}"]), (tag, key)
    res = ops.ball_query_partition(pts, nodes, 0.45, 24, return_count=True)
    for key, got in zip(("knn_idx", "knn_masks", "masks", "sizes"), res):
        assert np.array_equal(N_(got), g[f"bqp_{key}"]), key


def test_point_to_node_partition_matches_reference_formulation(ext):
    """The reference's own steps (point_cloud_partition.py:70-102) in torch on the same device; compared where the
    assignment is unambiguous at the expansion's rounding error (margin between the two nearest nodes > 1e-5)."""
    from vision3d_amd import ops

    rng = np.random.default_rng(9)
    n, m, limit = 4000, 96, 48
    points = T(rng.random((n, 3), dtype=np.float32))
    nodes = T(rng.random((m, 3), dtype=np.float32))
    sq = ops.pairwise_distance(nodes, points)
    p2n = sq.min(dim=0)[1]
    two = sq.topk(k=2, dim=0, largest=False)[0]
    clear = (two[1] - two[0]) > 1e-5
    got_p2n, got_masks, got_knn, got_kmask = ops.point_to_node_partition(points, nodes, point_limit=limit)
    assert torch.equal(got_p2n[clear], p2n[clear]) and clear.float().mean() > 0.99
    if bool(clear.all()):
        matching = torch.zeros_like(sq, dtype=torch.bool)
        matching[p2n, torch.arange(n, device=DEV)] = True
        k = min(int(matching.sum(dim=1).max()), limit)
        knn = sq.masked_fill(~matching, 1e12).topk(k=k, dim=1, largest=False)[1]
        kmask = p2n[knn] == torch.arange(m, device=DEV).unsqueeze(1)
        knn = knn.masked_fill(~kmask, n)
        assert torch.equal(got_kmask, kmask)
        # same point sets per node; order may differ only between points whose distances agree to 1e-6
        assert torch.equal(got_knn.sort(dim=1)[0], knn.sort(dim=1)[0])


def test_ball_query_partition_contract(ext):
    from vision3d_amd import ops

    rng = np.random.default_rng(10)
    points = T(rng.random((1500, 3), dtype=np.float32))
    nodes = T(rng.random((40, 3), dtype=np.float32))
    idx, masks, node_masks, sizes = ops.ball_query_partition(points, nodes, 0.1, 32, return_count=True)
    d = torch.cdist(nodes, points)
    assert idx.shape == (40, 32) and torch.equal(sizes, masks.sum(1)) and torch.equal(node_masks, sizes > 0)
    assert torch.all(idx[~masks] == 1500)
    want = torch.minimum((d < 0.1).sum(1), torch.tensor(32, device=DEV))
    assert torch.equal(sizes, want)


# ------------------------------------------------------------------------------------------------------
# full BASELINE sizes (configs C2 / C4 / C5) -- checked on the device against independent torch formulations and through
# size-independent properties, since the CPU oracle does not finish in seconds there
# ------------------------------------------------------------------------------------------------------
def test_full_size_c5_mutual_matching_properties(ext):
    """100k x 100k descriptors, C = 256 (BASELINE config 5): mutual nearest neighbours without the 40 GB matrix."""
    from vision3d_amd import ops

    g = torch.Generator().manual_seed(5)
    n = m = 100000
    x = torch.nn.functional.normalize(torch.randn((n, 256), generator=g), dim=1).to(DEV)
    y = torch.nn.functional.normalize(torch.randn((m, 256), generator=g), dim=1).to(DEV)
    y[:30000] = torch.nn.functional.normalize(x[:30000] + 0.02 * torch.randn((30000, 256), generator=g).to(DEV), dim=1)
    row_d = torch.empty((1, n, 1), device=DEV)
    row_i = torch.empty((1, n, 1), dtype=torch.long, device=DEV)
    col_d = torch.empty((1, m, 1), device=DEV)
    col_i = torch.empty((1, m, 1), dtype=torch.long, device=DEV)
    ext.knn_feat(x.unsqueeze(0), y.unsqueeze(0), row_d, row_i, 1)
    ext.knn_feat(y.unsqueeze(0), x.unsqueeze(0), col_d, col_i, 1)
    r, c, d = ops.extract_correspondences_from_feats(x, y, mutual=True, return_feat_dist=True)
    ri, ci = row_i.view(-1), col_i.view(-1)
    # definition: (r, c) is kept iff c is r's nearest column and r is c's nearest row; nonzero order; planted pairs found
    assert torch.equal(ri[r], c) and torch.equal(ci[c], r)
    keep = ci[ri] == torch.arange(n, device=DEV)
    assert r.numel() == int(keep.sum()) and torch.equal(r, torch.nonzero(keep).view(-1))
    assert int((r < 30000).sum()) >= 29990 and torch.equal(c[r < 30000], r[r < 30000])
    exact = ((x[r] - y[c]) ** 2).sum(dim=1)
    assert torch.allclose(d, exact, atol=2e-5, rtol=0)
    # union: every row and every column appears; pairs are unique and sorted by (row, col)
    r2, c2 = ops.extract_correspondences_from_feats(x, y, mutual=False)
    key = r2 * m + c2
    assert torch.all(key[1:] > key[:-1])
    assert torch.unique(r2).numel() == n and torch.unique(c2).numel() == m
    assert r2.numel() == n + m - r.numel()


def test_full_size_c4_pack_gathers_against_torch(ext):
    """One KPConv pyramid stage of BASELINE config 4: 16 clouds x 30,000 points, 38 neighbours, 64 channels."""
    from vision3d_amd import ops

    g = torch.Generator().manual_seed(4)
    n, k, c = 480000, 38, 64
    feats = torch.randn((n, c), generator=g).to(DEV)
    table = ((torch.arange(n).view(n, 1) + torch.randint(-3000, 3000, (n, k), generator=g)) % n)
    table[torch.rand((n, k), generator=g) < 0.15] = n
    table = table.to(DEV)
    padded = torch.cat([feats, torch.zeros_like(feats[:1])], 0)
    gathered = padded[table]  # (n, k, c): the reference's intermediate, 4.7 GB
    assert torch.equal(ops.index_select(feats, table, dim=0), gathered)
    assert torch.equal(ops.local_maxpool_pack_mode(feats, table), gathered.max(1)[0])
    assert torch.equal(ops.nearest_interpolate_pack_mode(feats, table), gathered[:, 0])
    pts = (torch.rand((n, 3), generator=g) * torch.tensor([3.0, 3.0, 2.5])).to(DEV)
    got = ops.knn_interpolate_pack_mode(pts, pts, feats, table, k=3)
    t3 = table[:, :3]
    ppad = torch.cat([pts, torch.zeros_like(pts[:1])], 0)
    sq = (pts.unsqueeze(1) - ppad[t3]).pow(2).sum(dim=-1)
    w = (t3 != n).float() / (sq + 1e-8)
    w = w / (w.sum(dim=1, keepdim=True) + 1e-8)
    want = (gathered[:, :3] * w.unsqueeze(-1)).sum(dim=1)
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-5 * float(feats.abs().max()))
    lengths = torch.full((16,), 30000, dtype=torch.long, device=DEV)
    pooled = ops.global_avgpool_pack_mode(feats, lengths)
    assert torch.allclose(pooled, feats.view(16, 30000, c).mean(dim=1), atol=1e-5 * float(feats.abs().max()), rtol=0)
    # backward of the gather: linear in the upstream gradient and equal to torch's index_add_ up to summation order
    del gathered
    up = torch.randn((n, c), generator=g).to(DEV)
    f = feats.clone().requires_grad_(True)
    ops.local_maxpool_pack_mode(f, table).backward(up)
    f2 = feats.clone().requires_grad_(True)
    torch.cat([f2, torch.zeros_like(f2[:1])], 0)[table].max(1)[0].backward(up)
    assert torch.allclose(f.grad, f2.grad, rtol=1e-4, atol=1e-4)


def test_full_size_c2_fused_grouping_equals_three_pass(ext):
    from vision3d_amd import ops

    g = torch.Generator().manual_seed(2)
    b, c, n, m, k = 32, 64, 4096, 1024, 128
    s_points = torch.rand((b, 3, n), generator=g).to(DEV)
    q_points = s_points[:, :, :m].contiguous()
    s_feats = torch.randn((b, c, n), generator=g).to(DEV)
    idx = torch.randint(0, n, (b, m, k), generator=g).to(DEV)
    want = torch.cat([ops.group_gather(s_feats, idx), ops.group_gather(s_points, idx) - q_points.unsqueeze(3)], dim=1)
    assert torch.equal(ops.sa_group(q_points, s_points, s_feats, idx), want)
    # the plain group gather itself (now on the fused kernel) against torch's gather
    ref = torch.gather(s_feats.unsqueeze(2).expand(-1, -1, m, -1), 3, idx.unsqueeze(1).expand(-1, c, -1, -1))
    assert torch.equal(want[:, :c], ref)
    q_feats = torch.randn((b, c, m), generator=g).to(DEV)
    nb = ops.group_gather(s_feats, idx[:, :, :32].contiguous())
    qe = q_feats.unsqueeze(3).expand_as(nb)
    assert torch.equal(ops.edge_group(q_feats, s_feats, idx[:, :, :32].contiguous()), torch.cat([qe, nb - qe], dim=1))


def test_full_size_kpconv_stage_against_torch_formulation(ext):
    from vision3d_amd import ops

    g = torch.Generator().manual_seed(6)
    n, h, c = 120000, 36, 128
    pts = (torch.rand((n, 3), generator=g) * torch.tensor([3.0, 3.0, 2.5])).to(DEV)
    # neighbours that are actually near: sort by a space-filling key first so that index-near means space-near
    order = torch.argsort((pts[:, 0] * 8).floor() * 10000 + (pts[:, 1] * 8).floor() * 100 + (pts[:, 2] * 8).floor())
    pts = pts[order].contiguous()
    feats = torch.randn((n, c), generator=g).to(DEV)
    feats[torch.rand(n, generator=g).to(DEV) < 0.1] = 0.0
    table = ((torch.arange(n).view(n, 1) + torch.randint(-40, 40, (n, h), generator=g)) % n)
    table[torch.rand((n, h), generator=g) < 0.1] = n
    table = table.to(DEV)
    kpts = (torch.randn((15, 3), generator=g) * 0.05).to(DEV)
    sigma = 0.08
    weighted, num = ops.kpconv_gather(pts, pts, feats, table, kpts, sigma)
    pp = torch.cat([pts, torch.zeros_like(pts[:1]) + 1e6], 0)
    sq = ((pp[table] - pts.unsqueeze(1)).unsqueeze(2) - kpts).pow(2).sum(dim=3)
    nw = torch.clamp(1 - torch.sqrt(sq) / sigma, min=0.0).transpose(1, 2)
    nf = torch.cat([feats, torch.zeros_like(feats[:1])], 0)[table]
    want = torch.matmul(nw, nf)
    scale = torch.matmul(nw, nf.abs())
    assert float(nw.max()) > 0.5  # the influences are not all clamped away
    # 1e-5 of the summed magnitude, plus the influences' own rounding: torch evaluates 1 - sqrt(.)/sigma with its own
    # sqrt / division, and next to the clamp that difference (<= 1e-6 absolute) is not relative to the influence
    slack = 1e-6 * nf.abs().sum(dim=1).unsqueeze(1)
    assert torch.all((weighted - want).abs() <= 1e-5 * scale + slack)
    want_num = torch.gt(nf.sum(dim=-1), 0.0).sum(dim=-1).clamp_(min=1)
    assert float((num != want_num).float().mean()) < 1e-4  # feature sums within rounding of zero may differ


def test_extract_correspondences_from_scores_vs_reference_fixture(ext, golden):
    from vision3d_amd import ops

    g = golden("ref_select.npz")
    log_p = T(g["m_log_p"])
    for mutual in (False, True):
        for dustbin in (False, True):
            for thr in (0.0, 0.05):
                tag = f"m_m{int(mutual)}_d{int(dustbin)}_t{int(thr > 0)}"
                r, c, s = ops.extract_correspondences_from_scores(log_p, mutual=mutual, has_dustbin=dustbin,
                                                                  threshold=thr, return_score=True)
                assert np.array_equal(np.stack([N_(r), N_(c)]), g[tag]), tag
                assert np.allclose(N_(s), g[tag + "_scores"], rtol=1e-6), tag
                assert len(ops.extract_correspondences_from_scores(log_p, mutual=mutual, has_dustbin=dustbin)) == 2


def test_sharded_matching_default_kernels_single_rank(ext):
    """sharding.mutual_matching_sharded with its default (HIP) search / merge / select on one rank == the ops route."""
    from vision3d_amd import ops, sharding

    rng = np.random.default_rng(31)
    x = T(rng.standard_normal((700, 16)).astype(np.float32))
    y = T(rng.standard_normal((900, 16)).astype(np.float32))
    for mutual in (True, False):
        r, c, d = sharding.mutual_matching_sharded(x, y, 1, 0, 900, mutual=mutual)
        wr, wc, wd = ops.extract_correspondences_from_feats(x, y, mutual=mutual, return_feat_dist=True, normalized=False)
        assert torch.equal(r, wr) and torch.equal(c, wc) and torch.allclose(d * d, wd)


# ------------------------------------------------------------------------------------------------------
# edge cases and error codes of the new entry points
# ------------------------------------------------------------------------------------------------------
def test_next_rows_empty_inputs_and_error_codes(ext):
    from vision3d_amd import ops

    z = lambda *shape, dtype=torch.float32: torch.zeros(shape, dtype=dtype, device=DEV)  # noqa: E731
    # empty selections
    r, c, s = ops.mutual_topk_select(z(5, 7), 0)
    assert r.numel() == 0 and c.numel() == 0 and s.numel() == 0
    assert ops.mutual_topk_select(z(5, 7), 0, reduce_result=False).sum() == 0
    r, c, s = ops.mutual_topk_select(torch.full((4, 4), float("-inf"), device=DEV), 1, threshold=0.0)
    assert r.numel() == 0
    with pytest.raises(RuntimeError, match="k must be <= 64"):
        ext.mutual_topk_select(z(1, 100, 100), 65, num_out=z(dtype=torch.long))
    # empty tables / clouds
    feats = torch.randn((6, 8), device=DEV)
    assert ops.index_select(feats, z(0, 3, dtype=torch.long), 0).shape == (0, 3, 8)
    assert ops.local_maxpool_pack_mode(feats, z(0, 3, dtype=torch.long)).shape == (0, 8)
    none = torch.full((4, 3), 6, dtype=torch.long, device=DEV)  # every neighbour absent
    assert torch.all(ops.local_maxpool_pack_mode(feats, none) == 0)
    assert torch.all(ops.knn_interpolate_pack_mode(z(4, 3), z(6, 3), feats, none) == 0)
    pooled = ops.global_avgpool_pack_mode(feats, torch.tensor([6, 0], device=DEV))
    assert torch.allclose(pooled[0], feats.mean(0)) and torch.isnan(pooled[1]).all()  # torch.mean of an empty slice
    with pytest.raises(RuntimeError, match="multiple of 4"):
        ext.local_maxpool_pack(torch.randn((6, 6), device=DEV), none, z(4, 6))
    f = feats.clone().requires_grad_(True)  # backward with a table that names nobody: zero gradient
    ops.local_maxpool_pack_mode(f, none).sum().backward()
    assert torch.all(f.grad == 0)
    # kpconv: no neighbours at all -> zeros and a count of 1; too many neighbours for the fused kernel -> formulation path
    w, n = ops.kpconv_gather(z(4, 3), z(6, 3), feats, none, torch.randn((15, 3), device=DEV), 0.1)
    assert torch.all(w == 0) and torch.all(n == 1)
    with pytest.raises(RuntimeError, match="at most 64 neighbours"):
        ext.kpconv_gather(z(4, 3), z(6, 3), feats, z(4, 65, dtype=torch.long), z(15, 3), z(4, 15, 8), None, 0.1)
    wide = torch.randint(0, 6, (4, 70), device=DEV)
    w2, _ = ops.kpconv_gather(torch.rand((4, 3), device=DEV), torch.rand((6, 3), device=DEV), feats, wide,
                              torch.zeros((15, 3), device=DEV), 2.0)
    assert w2.shape == (4, 15, 8) and torch.isfinite(w2).all()
    # partition: a single node owns everything; grouping with K = 1 and an out-of-range index reading as zero
    p2n, sizes, masks, knn, kmask = ops.point_to_node_partition(torch.rand((9, 3), device=DEV), z(1, 3), return_count=True)
    assert torch.all(p2n == 0) and int(sizes[0]) == 9 and knn.shape == (1, 9) and bool(kmask.all())
    idx = torch.tensor([[[0], [9]]], device=DEV)
    out = ops.sa_group(z(1, 3, 2), torch.ones((1, 3, 4), device=DEV), None, idx)
    assert torch.all(out[0, :, 0, 0] == 1) and torch.all(out[0, :, 1, 0] == 0)


@pytest.mark.parametrize("n,m,h,k,c", [(300, 200, 38, 15, 64), (64, 40, 64, 16, 32), (400, 333, 5, 7, 4), (200, 150, 17, 15, 132)])
def test_kpconv_gather_backward_vs_oracle(ext, oracle_mod, n, m, h, k, c):
    rng = np.random.default_rng(n * 3 + m + h + k + c)
    q, s, feats, table, kpts = _kpconv_case(rng, n, m, h, k, c)
    sigma = 0.06
    _, _, w = oracle_mod.kpconv_gather(q, s, feats, table, kpts, sigma)  # (M,K,H) influences, float32 reference order
    gw = rng.standard_normal((m, k, c)).astype(np.float32)
    want = np.matmul(np.transpose(w, (0, 2, 1)).astype(np.float64), gw.astype(np.float64))
    scale = np.matmul(np.transpose(w, (0, 2, 1)).astype(np.float64), np.abs(gw).astype(np.float64))
    got = torch.full((m, h, c), float("nan"), device=DEV)
    ext.kpconv_gather_backward(T(q), T(s), T(table), T(kpts), T(gw), got, sigma)
    assert np.all(np.abs(N_(got).astype(np.float64) - want) <= 1e-5 * scale + 1e-30)


# ------------------------------------------------------------------------------------------------------
# the extension's NumPy entry points used by vision3d.array_ops (csrc/pybind.cpp:34-70)
# ------------------------------------------------------------------------------------------------------
def test_ext_numpy_entry_points(ext, oracle_mod):
    rng = np.random.default_rng(41)
    lengths = np.asarray([700, 0, 900], dtype=np.int64)
    pts = (rng.random((1600, 3), dtype=np.float32) * np.asarray([2.0, 2.0, 1.0], dtype=np.float32))
    # radius_neighbors: all neighbours within the radius, ascending, padded with total_s, width = largest count
    got = ext.radius_neighbors(pts, pts, lengths, lengths, 0.15)
    full = oracle_mod.radius_search_pack(pts, pts, lengths, lengths, 0.15, 128)
    width = int((full != 1600).sum(axis=1).max())
    assert isinstance(got, np.ndarray) and got.dtype == np.int64 and got.shape == (1600, width)
    assert np.array_equal(got, full[:, :width])
    with pytest.raises(RuntimeError, match="more than 512"):
        ext.radius_neighbors(pts, pts, lengths, lengths, 5.0)
    # grid_subsampling
    sp, sl = ext.grid_subsampling(pts, lengths, 0.1)
    wp, wl = oracle_mod.grid_subsample_pack(pts, lengths, 0.1)
    assert sp.dtype == np.float32 and np.array_equal(sp, wp) and np.array_equal(sl, wl)
    # sample_nodes_with_fps against the restated CPU code (tie-free cloud): by count, by distance, by both
    cloud = rng.random((3000, 3), dtype=np.float32)
    for min_d, num in ((-1.0, 64), (0.12, -1), (0.05, 40), (0.3, 500), (0.02, -1)):
        want = oracle_mod.sample_nodes_with_fps(cloud, min_d, num)
        got = ext.sample_nodes_with_fps(cloud, min_d, num)
        assert got.dtype == np.int64 and np.array_equal(got, want), (min_d, num, len(got), len(want))
    # the helpers outside the path say so
    with pytest.raises(NotImplementedError, match="outside the operator path"):
        ext.compute_clusters(None, None)


# ------------------------------------------------------------------------------------------------------
# rank 2, the remaining selectors of ops/matching.py: threshold selection and global top-k
# ------------------------------------------------------------------------------------------------------
def _thr_native(ext, score, thr, largest=True, cap=None):
    n = score.numel() if cap is None else cap
    idx = torch.full((2, max(n, 1)), -7, dtype=torch.long, device=DEV)
    sc = torch.full((max(n, 1),), -7.0, device=DEV)
    cnt = torch.zeros((), dtype=torch.long, device=DEV)
    ext.threshold_select(score, thr, out_row=idx[0], out_col=idx[1], out_score=sc, num_out=cnt, largest=largest)
    return idx, sc, int(cnt.item())


def _topk_native(ext, score, k, largest=True, dustbin=False):
    idx = torch.full((2, max(k, 1)), -7, dtype=torch.long, device=DEV)
    sc = torch.full((max(k, 1),), -7.0, device=DEV)
    cnt = torch.zeros((), dtype=torch.long, device=DEV)
    ext.global_topk_select(score, k, out_row=idx[0], out_col=idx[1], out_score=sc, num_out=cnt, largest=largest,
                           has_dustbin=dustbin)
    c = int(cnt.item())
    return N_(idx[0, :c]), N_(idx[1, :c]), N_(sc[:c])


def test_threshold_and_topk_ops_vs_reference_fixture(ext, golden):
    from vision3d_amd import ops

    g = golden("ref_matching.npz")
    log_p = T(g["log_p"])
    for i in range(3):
        thr = float(g[f"thr{i}_value"])
        r, c, s = ops.extract_correspondences_from_scores_threshold(log_p, thr, return_score=True)
        assert np.array_equal(np.stack([N_(r), N_(c)]), g[f"thr{i}"]), i
        assert np.allclose(N_(s), g[f"thr{i}_scores"], rtol=1e-6), i
        assert len(ops.extract_correspondences_from_scores_threshold(log_p, thr)) == 2
    for k in (1, 17, 300):
        for largest in (True, False):
            for dustbin in (False, True):
                tag = f"topk{k}_l{int(largest)}_d{int(dustbin)}"
                r, c, s = ops.extract_correspondences_from_scores_topk(log_p, k, has_dustbin=dustbin, largest=largest,
                                                                       return_score=True)
                assert np.array_equal(np.stack([N_(r), N_(c)]), g[tag]), tag
                assert np.array_equal(N_(s), g[tag + "_scores"]), tag
    with pytest.raises(RuntimeError, match="out of range"):
        ops.extract_correspondences_from_scores_topk(log_p, log_p.numel() + 1)


@pytest.mark.parametrize("shape", [(1, 1), (7, 9), (255, 257), (256, 256), (300, 1000), (1031, 517), (2048, 2048)])
def test_threshold_select_vs_oracle(ext, oracle_mod, shape):
    rng = np.random.default_rng(shape[0] * 7919 + shape[1])
    s = rng.standard_normal(shape).astype(np.float32)
    s[rng.random(shape) < 0.01] = np.inf
    s[rng.random(shape) < 0.01] = -np.inf
    score = T(s)
    for thr, largest in ((1.5, True), (-1.5, False), (-10.0, True), (10.0, True), (0.0, False)):
        wr, wc, ws = oracle_mod.threshold_select(s, thr, largest)
        idx, sc, c = _thr_native(ext, score, thr, largest)
        assert c == len(wr), (thr, largest)
        assert np.array_equal(N_(idx[0, :c]), wr) and np.array_equal(N_(idx[1, :c]), wc), (thr, largest)
        assert np.array_equal(N_(sc[:c]), ws)
        assert int((idx[:, c:] != -7).sum()) == 0  # nothing written past the count
    # a short output buffer: the true count is still reported, only `capacity` entries are written
    wr, wc, _ = oracle_mod.threshold_select(s, 0.0, True)
    cap = max(len(wr) // 3, 1)
    idx, sc, c = _thr_native(ext, score, 0.0, True, cap=cap)
    assert c == len(wr)
    m = min(c, cap)
    assert np.array_equal(N_(idx[0, :m]), wr[:m]) and np.array_equal(N_(idx[1, :m]), wc[:m])
    # an unaligned view (row offset of one float): scalar loads
    flat = torch.empty(s.size + 1, device=DEV)
    flat[1:] = score.reshape(-1)
    view = flat[1:].view(shape)
    wr, wc, ws = oracle_mod.threshold_select(s, 0.5, True)
    idx, sc, c = _thr_native(ext, view, 0.5, True)
    assert c == len(wr) and np.array_equal(N_(idx[0, :c]), wr) and np.array_equal(N_(sc[:c]), ws)


@pytest.mark.parametrize("shape,ks", [((1, 1), (1,)), ((7, 9), (1, 5, 63)), ((255, 257), (1, 100, 4096, 4097, 20000)),
                                      ((1031, 517), (3, 256, 5000)), ((2048, 2048), (1, 256, 4096, 100000))])
def test_global_topk_select_vs_oracle(ext, oracle_mod, shape, ks):
    rng = np.random.default_rng(shape[0] * 104729 + shape[1])
    s = rng.standard_normal(shape).astype(np.float32)
    score = T(s)
    for k in ks:
        for largest in (True, False):
            for dustbin in (False, True):
                wr, wc, ws = oracle_mod.extract_correspondences_from_scores_topk(s, k, dustbin, largest)
                r, c, v = _topk_native(ext, score, k, largest, dustbin)
                assert np.array_equal(r, wr) and np.array_equal(c, wc), (k, largest, dustbin)
                assert np.array_equal(v, ws), (k, largest, dustbin)
    # without has_dustbin the count is k and num_out may be omitted
    k = ks[-1]
    idx = torch.empty((2, k), dtype=torch.long, device=DEV)
    ext.global_topk_select(score, k, out_row=idx[0], out_col=idx[1])
    wr, wc, _ = oracle_mod.extract_correspondences_from_scores_topk(s, k)
    assert np.array_equal(N_(idx[0]), wr) and np.array_equal(N_(idx[1]), wc)


def test_global_topk_select_ties_take_lower_flat_index(ext, oracle_mod):
    """Heavily tied scores (a few distinct values, signed zeros, infinities): the k-th value is tied, so the index passes
    of the radix select decide -- lowest flat index first, exactly the oracle's stable order."""
    rng = np.random.default_rng(5)
    for shape in ((64, 64), (300, 700), (1000, 1000)):
        s = rng.integers(-2, 3, size=shape).astype(np.float32)
        s[rng.random(shape) < 0.05] = -0.0
        s[rng.random(shape) < 0.001] = np.inf
        s[rng.random(shape) < 0.001] = -np.inf
        score = T(s)
        for k in (1, 10, 1000, min(50000, s.size), s.size if s.size <= 4096 else 4096):
            for largest in (True, False):
                wr, wc, ws = oracle_mod.extract_correspondences_from_scores_topk(s, k, False, largest)
                r, c, v = _topk_native(ext, score, k, largest)
                assert np.array_equal(r, wr) and np.array_equal(c, wc), (shape, k, largest)
                assert np.array_equal(v, ws)
    # matrices large enough for the sampled short list (>= 2^20 cells) whose ties overflow it: the device-side fallback
    # to the full-matrix select must give the same answer (few distinct values; one dominant value; all equal)
    for shape, make in (((1100, 1100), lambda: rng.integers(-2, 3, size=(1100, 1100)).astype(np.float32)),
                        ((2048, 1024), lambda: np.where(rng.random((2048, 1024)) < 0.999, 1.0, rng.random((2048, 1024))).astype(np.float32)),
                        ((1500, 1500), lambda: np.full((1500, 1500), 0.25, dtype=np.float32))):
        s = make()
        score = T(s)
        for k in (1, 300, 5000):
            for largest in (True, False):
                wr, wc, ws = oracle_mod.extract_correspondences_from_scores_topk(s, k, False, largest)
                r, c, v = _topk_native(ext, score, k, largest)
                assert np.array_equal(r, wr) and np.array_equal(c, wc), (shape, k, largest)
                assert np.array_equal(v, ws)
    s = np.zeros((100, 100), dtype=np.float32)  # all equal: the first k cells
    r, c, _ = _topk_native(ext, T(s), 250)
    assert np.array_equal(r * 100 + c, np.arange(250))


def test_global_topk_select_matches_torch_topk_at_scale(ext):
    """8192 x 8192 (256 MiB): values and index set equal torch.topk's on tie-free scores; sorted best first."""
    g = torch.Generator(device=DEV).manual_seed(3)
    score = torch.randn((8192, 8192), device=DEV, generator=g)
    for k, largest in ((256, True), (5000, False)):
        r, c, v = _topk_native(ext, score, k, largest)
        tv, ti = score.view(-1).topk(k, largest=largest)
        assert np.array_equal(v, N_(tv))
        assert np.array_equal(r * 8192 + c, N_(ti))
    wr, wc = torch.nonzero(score > 3.5, as_tuple=True)
    idx, sc, cnt = _thr_native(ext, score, 3.5, True, cap=1 << 20)
    assert cnt == wr.numel() and torch.equal(idx[0, :cnt], wr) and torch.equal(idx[1, :cnt], wc)
    assert torch.equal(sc[:cnt], score[wr, wc])


def test_superpoint_matching_heads_vs_reference_fixture(ext, golden):
    """The two GeoTransformer heads against the reference modules run on the CPU.  Scores: the CPU / MFMA distance
    matrices differ by rounding, so 1e-5 of the score scale; indices exact (the fixture's candidates are well separated)."""
    from vision3d_amd import ops

    g = golden("ref_matching.npz")
    src, tgt = T(g["spm_src"]), T(g["spm_tgt"])
    sm, tm, sw, tw = T(g["spm_smask"]), T(g["spm_tmask"]), T(g["spm_sw"]), T(g["spm_tw"])
    for dual in (True, False):
        for masked in (False, True):
            for weighted in (False, True):
                tag = f"spmg_d{int(dual)}_m{int(masked)}_w{int(weighted)}"
                head = ops.SuperPointMatchingGlobalTopk(48, dual_normalization=dual)
                a, b, c = head(src, tgt, sm if masked else None, tm if masked else None, sw if weighted else None,
                               tw if weighted else None)
                want, ws = g[tag], g[tag + "_scores"]
                assert np.allclose(N_(c), ws, rtol=2e-5, atol=1e-7), tag
                assert np.array_equal(np.stack([N_(a), N_(b)]), want), tag
    for mutual in (True, False):
        for masked in (False, True):
            for thr in (None, 1.05, 0.9):
                tag = f"spmm_m{int(mutual)}_k{int(masked)}_t{'n' if thr is None else str(thr)}"
                head = ops.SuperPointMatchingMutualTopk(24, 3, threshold=thr, mutual=mutual)
                a, b, c = head(src, tgt, sm if masked else None, tm if masked else None)
                assert np.array_equal(np.stack([N_(a), N_(b)]), g[tag]), tag
                assert np.allclose(N_(c), g[tag + "_scores"], rtol=2e-5, atol=1e-6), tag


def test_threshold_and_topk_empty_inputs_and_errors(ext):
    cnt = torch.full((), 5, dtype=torch.long, device=DEV)
    ext.threshold_select(torch.empty((0, 7), device=DEV), 0.0, num_out=cnt)
    assert int(cnt.item()) == 0
    cnt.fill_(5)
    ext.global_topk_select(torch.empty((4, 4), device=DEV), 0, num_out=cnt)
    assert int(cnt.item()) == 0
    with pytest.raises(RuntimeError, match="out of range"):
        ext.global_topk_select(torch.zeros((4, 4), device=DEV), 17, num_out=cnt)
    with pytest.raises(RuntimeError, match="num_out"):
        ext.threshold_select(torch.zeros((4, 4), device=DEV), 0.0)
    with pytest.raises(RuntimeError, match="CUDA"):
        ext.threshold_select(torch.zeros((4, 4)), 0.0, num_out=cnt)


# ------------------------------------------------------------------------------------------------------
# voxelisation (ops/voxelize.py) on the grid-subsampling kernels
# ------------------------------------------------------------------------------------------------------
def test_voxelize_vs_reference_fixture(ext, golden):
    from vision3d_amd import ops

    g = golden("ref_voxelize.npz")
    pts, lengths, feats = T(g["points"]), T(g["lengths"]), T(g["feats"])
    v, n, f, l = ops.voxelize_pack_mode(pts, lengths, float(g["voxel_size"]), feats=feats)
    assert v.dtype == torch.long and np.array_equal(N_(v), g["voxels"])
    assert np.array_equal(N_(l), g["v_lengths"])
    assert np.array_equal(N_(n), g["nodes"]) and np.array_equal(N_(f), g["v_feats"])  # ordered sums: bit-exact
    v2, n2, l2 = ops.voxelize_pack_mode(pts, lengths, float(g["voxel_size"]))
    assert torch.equal(v, v2) and torch.equal(n, n2) and torch.equal(l, l2)
    bv, bn, bf, bm = ops.voxelize(T(g["b_points"]), 0.25, masks=T(g["b_masks"]), feats=T(g["b_feats"]))
    assert np.array_equal(N_(bm), g["b_v_masks"]) and np.array_equal(N_(bv), g["b_voxels"])
    assert np.array_equal(N_(bn), g["b_nodes"]) and np.array_equal(N_(bf), g["b_v_feats"])
    bv3, bn3, bm3 = ops.voxelize(T(g["b_points"]), 0.25)  # no masks, no features
    assert bv3.shape[0] == 3 and bv3.shape[:2] == bm3.shape and bn3.shape == bv3.shape


@pytest.mark.parametrize("case", ["random", "duplicates", "single_voxel", "ragged", "c4"])
def test_voxelize_vs_oracle(ext, oracle_mod, case):
    from vision3d_amd import ops

    rng = np.random.default_rng(["random", "duplicates", "single_voxel", "ragged", "c4"].index(case) + 77)
    if case == "random":
        lengths, vs, C = [5000, 3000], 0.05, 16
        pts = rng.random((8000, 3)).astype(np.float32)
    elif case == "duplicates":
        lengths, vs, C = [600, 600], 0.1, 3
        pts = np.repeat(rng.random((400, 3)).astype(np.float32), 3, axis=0)
    elif case == "single_voxel":
        lengths, vs, C = [1000], 10.0, 5
        pts = rng.random((1000, 3)).astype(np.float32)
    elif case == "ragged":
        lengths, vs, C = [1, 0, 257, 0, 4095, 2], 0.07, 1
        pts = (rng.random((sum(lengths), 3)).astype(np.float32) - 0.5) * 4
    else:  # config 4: 16 clouds of 30,000 points in a 3 x 3 x 2.5 m box, 2.5 cm voxels
        lengths, vs, C = [30000] * 16, 0.025, 32
        pts = rng.random((480000, 3)).astype(np.float32) * np.float32([3.0, 3.0, 2.5])
    feats = rng.standard_normal((pts.shape[0], C)).astype(np.float32)
    want = oracle_mod.voxelize_pack(pts, np.asarray(lengths), vs, feats)
    got = ops.voxelize_pack_mode(T(pts), T(np.asarray(lengths, dtype=np.int64)), vs, feats=T(feats))
    for w, g_, name in zip(want, got, ("voxels", "nodes", "v_feats", "v_lengths")):
        assert np.array_equal(N_(g_), w), (case, name)
    # consistency with grid subsampling where floor(p / v) == floor(p * (1 / v)): same voxel count per cloud
    assert int(got[3].sum()) == got[0].shape[0]


def test_voxelize_empty(ext):
    from vision3d_amd import ops

    v, n, l = ops.voxelize_pack_mode(torch.empty((0, 3), device=DEV), torch.zeros(2, dtype=torch.long, device=DEV), 0.1)
    assert v.shape == (0, 3) and n.shape == (0, 3) and N_(l).tolist() == [0, 0]


def test_knn_interpolate_and_deprecated_spellings(ext):
    """knn_interpolate (knn_interpolate.py:10-40) against the same lines written with torch on the device (the search
    replaced by a brute-force cdist + topk); get_point_to_node_indices / naive_radius_search_pack_mode against the
    functions they are spellings of."""
    from vision3d_amd import ops

    g = torch.Generator(device=DEV).manual_seed(11)
    q = torch.rand((3000, 3), device=DEV, generator=g)
    s = torch.rand((1200, 3), device=DEV, generator=g)
    for C in (32, 6):  # fused kernel / generic formulation
        f = torch.randn((1200, C), device=DEV, generator=g, requires_grad=True)
        for limit in (None, 0.05):
            out = ops.knn_interpolate(q, s, f, k=3, distance_limit=limit)
            d, i = torch.cdist(q.double(), s.double()).topk(3, dim=1, largest=False)
            d = d.float()
            if limit is not None:
                d = d.masked_fill(d > limit, 1e10)
            w = 1.0 / (d + 1e-10)
            w = w / w.sum(dim=1, keepdim=True)
            want = (f.detach()[i] * w.unsqueeze(-1)).sum(dim=1)
            assert torch.allclose(out, want, rtol=1e-4, atol=1e-5), (C, limit)
        up = torch.randn_like(out)
        (grad,) = torch.autograd.grad(out, f, up)
        want_grad = torch.zeros_like(f).index_add_(0, i.reshape(-1), (up.unsqueeze(1) * w.unsqueeze(-1)).reshape(-1, C))
        assert torch.allclose(grad, want_grad, rtol=1e-4, atol=1e-5), C
    nodes = s[:50].contiguous()
    a, sizes = ops.get_point_to_node_indices(q, nodes, return_counts=True)
    b = ops.point_to_node_partition(q, nodes, return_count=True, gather_points=False)
    assert torch.equal(a, b[0]) and torch.equal(sizes, b[1]) and torch.equal(ops.get_point_to_node_indices(q, nodes), a)
    ql = torch.tensor([1000, 2000], device=DEV)
    sl = torch.tensor([500, 700], device=DEV)
    assert torch.equal(ops.naive_radius_search_pack_mode(q, s, ql, sl, 0.1, 16),
                       ops.radius_search_pack_mode(q, s, ql, sl, 0.1, 16))


def test_next_row_entry_points_take_caller_owned_workspaces():
    """SURVEY 8b ownership contract ("ext never allocates or frees", ops/ball_query.py:38) on the (f) rows, straight through
    the C ABI: grid subsampling, voxelisation and the four selectors give the same outputs with a caller workspace of
    v3d_*_workspace_bytes() as with NULL (stream-ordered allocation inside), make no stream-ordered allocation when the
    workspace is given (the default memory pool's high-water mark stays at zero), and refuse a workspace that is too small."""
    import ctypes

    from vision3d_amd import _lib

    lib = _lib.lib()
    hip = ctypes.CDLL("libamdhip64.so")
    dev = torch.cuda.current_device()
    stream = torch.cuda.current_stream().cuda_stream
    pool = ctypes.c_void_p()
    assert hip.hipDeviceGetDefaultMemPool(ctypes.byref(pool), dev) == 0

    def pool_mark(reset=False):
        v = ctypes.c_uint64(0)
        if reset:
            assert hip.hipMemPoolSetAttribute(pool, 0x8, ctypes.byref(v)) == 0  # hipMemPoolAttrUsedMemHigh
        assert hip.hipMemPoolGetAttribute(pool, 0x8, ctypes.byref(v)) == 0
        return v.value

    def both(nbytes, call, outputs):
        """call(ws_ptr, ws_bytes) -> rc; outputs() -> list of tensors to compare."""
        for t in outputs():
            t.fill_(0)  # rows past the count are not written: the same start for both runs
        assert call(None, 0) == 0, lib.v3d_last_error_string()
        torch.cuda.synchronize()
        want = [t.clone() for t in outputs()]
        for t in outputs():
            t.fill_(0)
        ws = torch.empty((nbytes + 256,), dtype=torch.uint8, device=DEV)
        off = (-ws.data_ptr()) % 256
        torch.cuda.synchronize()
        pool_mark(reset=True)
        assert call(ws.data_ptr() + off, nbytes) == 0, lib.v3d_last_error_string()
        torch.cuda.synchronize()
        assert pool_mark() == 0, "a caller workspace was given, yet the call allocated"
        assert all(torch.equal(a, b) for a, b in zip(outputs(), want))
        if nbytes > 256:
            rc = call(ws.data_ptr() + off, nbytes - 256)
            assert rc != 0 and lib.v3d_error_name(rc).decode() == "V3D_ERR_WORKSPACE"
            rc = call(ws.data_ptr() + off + 8, nbytes)  # misaligned
            assert rc != 0 and lib.v3d_error_name(rc).decode() == "V3D_ERR_WORKSPACE"

    g = torch.Generator().manual_seed(77)
    # grid subsampling + voxelisation
    lens = torch.tensor([3000, 0, 2000], device=DEV)
    pts = (torch.rand((5000, 3), generator=g) * 2.0 - 0.5).to(DEV)
    out, out_l = torch.zeros_like(pts), torch.zeros_like(lens)
    num = torch.zeros((), dtype=torch.long, device=DEV)
    both(lib.v3d_grid_subsample_workspace_bytes(3, 5000, 0),
         lambda p, n: lib.v3d_grid_subsample_pack_f32(pts.data_ptr(), lens.data_ptr(), out.data_ptr(), out_l.data_ptr(),
                                                      num.data_ptr(), 3, 5000, 0.1, p, n, dev, stream),
         lambda: [out, out_l, num])
    feats = torch.randn((5000, 8), generator=g).to(DEV)
    vox = torch.zeros((5000, 3), dtype=torch.long, device=DEV)
    vf = torch.zeros_like(feats)
    both(lib.v3d_grid_subsample_workspace_bytes(3, 5000, 8),
         lambda p, n: lib.v3d_voxelize_pack_f32(pts.data_ptr(), lens.data_ptr(), feats.data_ptr(), 8, vox.data_ptr(),
                                                out.data_ptr(), vf.data_ptr(), out_l.data_ptr(), num.data_ptr(), 3, 5000,
                                                0.1, p, n, dev, stream),
         lambda: [vox, out, vf, out_l, num])
    # mutual top-k on a matrix large enough for the sorted back end, and on lists
    B, N, M, k = 1, 3000, 2500, 3
    score = torch.randn((B, N, M), generator=g).to(DEV)
    cap = k * B * min(N, M)
    ob, orow, ocol = (torch.zeros((cap,), dtype=torch.long, device=DEV) for _ in range(3))
    osc = torch.zeros((cap,), device=DEV)
    assert lib.v3d_mutual_topk_select_workspace_bytes(B, N, M, k) > 0
    both(lib.v3d_mutual_topk_select_workspace_bytes(B, N, M, k),
         lambda p, n: lib.v3d_mutual_topk_select_f32(score.data_ptr(), None, None, B, N, M, k, 1 | 2, 0.0, None,
                                                     ob.data_ptr(), orow.data_ptr(), ocol.data_ptr(), osc.data_ptr(),
                                                     num.data_ptr(), cap, p, n, dev, stream),
         lambda: [ob, orow, ocol, osc, num])
    rv, ri = score.topk(k, dim=2)
    cv, ci = score.topk(k, dim=1)
    ci, cv = ci.transpose(1, 2).contiguous(), cv.transpose(1, 2).contiguous()
    ri, rv = ri.contiguous(), rv.contiguous()
    both(lib.v3d_mutual_select_lists_workspace_bytes(B, N, M, k),
         lambda p, n: lib.v3d_mutual_select_lists_f32(ri.data_ptr(), rv.data_ptr(), ci.data_ptr(), cv.data_ptr(), None, None,
                                                      B, N, M, k, 1 | 2, 0.0, None, ob.data_ptr(), orow.data_ptr(),
                                                      ocol.data_ptr(), osc.data_ptr(), num.data_ptr(), cap, p, n, dev,
                                                      stream),
         lambda: [ob, orow, ocol, osc, num])
    # threshold and global top-k selection
    mat = score[0].contiguous()
    cap = 1 << 16
    orow2, ocol2 = (torch.zeros((cap,), dtype=torch.long, device=DEV) for _ in range(2))
    osc2 = torch.zeros((cap,), device=DEV)
    both(lib.v3d_threshold_select_workspace_bytes(N, M),
         lambda p, n: lib.v3d_threshold_select_f32(mat.data_ptr(), N, M, 3.0, 1, orow2.data_ptr(), ocol2.data_ptr(),
                                                   osc2.data_ptr(), num.data_ptr(), cap, p, n, dev, stream),
         lambda: [orow2, ocol2, osc2, num])
    for kk in (300, 20000):  # sorted in LDS / by the device sort
        both(lib.v3d_global_topk_select_workspace_bytes(N, M, kk),
             lambda p, n: lib.v3d_global_topk_select_f32(mat.data_ptr(), N, M, kk, 1, orow2.data_ptr(), ocol2.data_ptr(),
                                                         osc2.data_ptr(), num.data_ptr(), cap, p, n, dev, stream),
             lambda: [orow2, ocol2, osc2, num])
    assert lib.v3d_threshold_select_workspace_bytes(100, 100) == 0  # one-launch size: no scratch at all
    # pack-mode radius search and kNN: pair-by-pair route (small clouds) and cell-list route (clouds of >= 2048 supports)
    for lens_s, lens_q in (((300, 200), (100, 150)), ((3000, 2500), (70000, 60000))):
        ns, nq = sum(lens_s), sum(lens_q)
        sp = torch.rand((ns, 3), generator=g).to(DEV)
        qp = torch.rand((nq, 3), generator=g).to(DEV)
        sl = torch.tensor(lens_s, device=DEV)
        ql = torch.tensor(lens_q, device=DEV)
        kd = torch.zeros((nq, 5), device=DEV)
        ki = torch.zeros((nq, 5), dtype=torch.long, device=DEV)
        assert lib.v3d_knn_pack_uses_cell_list(2, nq, ns, 5) == (1 if ns > 4096 else 0)
        both(lib.v3d_knn_pack_workspace_bytes(2, ns, 5),
             lambda p, n: lib.v3d_knn_pack_f32(qp.data_ptr(), sp.data_ptr(), ql.data_ptr(), sl.data_ptr(), kd.data_ptr(),
                                               ki.data_ptr(), 2, nq, ns, 5, p, n, dev, stream),
             lambda: [kd, ki])
        tab = torch.zeros((nq, 12), dtype=torch.long, device=DEV)
        both(lib.v3d_radius_search_workspace_bytes(2, ns),
             lambda p, n: lib.v3d_radius_search_pack_f32(qp.data_ptr(), sp.data_ptr(), ql.data_ptr(), sl.data_ptr(),
                                                         tab.data_ptr(), 2, nq, ns, 0.1, 12, p, n, dev, stream),
             lambda: [tab])
    assert lib.v3d_knn_pack_workspace_bytes(2, 5500, 5) > lib.v3d_knn_pack_workspace_bytes(2, 500, 5)
    # a caller-held grid is never optional
    rc = lib.v3d_radius_grid_build_f32(sp.data_ptr(), sl.data_ptr(), 2, ns, 0.1, None, 0, dev, stream)
    assert rc != 0 and b"NULL workspace" in lib.v3d_last_error_string()


def test_radius_grid_built_once_serves_several_query_sets_and_radii(ext, oracle_mod):
    """v3d_radius_grid_build_f32 + v3d_radius_search_grid_f32: one cell list of the supports, searched by two query sets
    and at two radii up to the one it was built for -- every table identical to a search of its own and to the oracle."""
    rng = np.random.default_rng(5)
    sl = np.asarray([900, 0, 1500, 40], np.int64)
    s = rng.random((int(sl.sum()), 3), dtype=np.float32)
    qsets = [(s, sl), (rng.random((700, 3), dtype=np.float32), np.asarray([300, 50, 349, 1], np.int64))]
    ts, tsl = T(s), T(sl)
    grid = ext.radius_grid_build(ts, tsl, 0.2)
    for q, ql in qsets:
        for radius, limit in ((0.2, 40), (0.1, 16), (0.2, 100), (0.0, 3)):
            out = torch.full((q.shape[0], limit), -1, dtype=torch.long, device=DEV)
            ext.radius_search_grid(T(q), T(ql), grid, out, radius, limit)
            own = torch.full_like(out, -2)
            ext.radius_search_pack(T(q), ts, T(ql), tsl, own, radius, limit)
            assert torch.equal(out, own), (radius, limit)
            assert np.array_equal(N_(out), oracle_mod.radius_search_pack(q, s, ql, sl, radius, limit)), (radius, limit)
    out = torch.zeros((700, 8), dtype=torch.long, device=DEV)
    with pytest.raises(RuntimeError, match="beyond the radius the grid was built for"):
        ext.radius_search_grid(T(qsets[1][0]), T(qsets[1][1]), grid, out, 0.25, 8)


def test_graph_pyramid_tables_equal_table_by_table_searches(ext):
    """ops.build_grid_and_radius_graph_pyramid shares one cell list per stage among its searches
    (graph_pyramid.py:31-45 searches table by table): the tables are those of searches of their own."""
    from vision3d_amd import ops

    g = torch.Generator().manual_seed(9)
    lengths = torch.tensor([4000, 2500, 6000], device=DEV)
    pts = (torch.rand((int(lengths.sum()), 3), generator=g) * torch.tensor([2.0, 2.0, 1.0])).to(DEV)
    limits = [24, 20, 18]
    pyr = ops.build_grid_and_radius_graph_pyramid(pts, lengths, 3, 0.05, 0.12, limits)
    radius = 0.12
    for i in range(3):
        cp, cl = pyr["points"][i], pyr["lengths"][i]
        assert torch.equal(pyr["neighbors"][i], ops.radius_search_pack_mode(cp, cp, cl, cl, radius, limits[i]))
        if i < 2:
            sp, sl = pyr["points"][i + 1], pyr["lengths"][i + 1]
            assert torch.equal(pyr["subsampling"][i], ops.radius_search_pack_mode(sp, cp, sl, cl, radius, limits[i]))
            assert torch.equal(pyr["upsampling"][i], ops.radius_search_pack_mode(cp, sp, cl, sl, radius * 2, limits[i + 1]))
        radius *= 2
