"""Searches beyond the lane-list sizes, and pack-mode kNN without padding:
  * v3d_knn_feat_f32 with k > 64 (workgroup-per-row selection; pykeops' Kmin_argKmin, ops/knn.py:26, has no limit);
  * v3d_radius_search_pack_f32 with neighbor_limit > 128 (ops/radius_search.py:44-71 passes any limit to pykeops);
  * v3d_knn_pack_f32 behind ops.knn_pack_mode (ops/knn.py:110-145) against the pack-mode kNN oracle (equal distances
    lower index first), on the pair-by-pair kernels (small clouds) and on the cell list (clouds of >= 2048 supports).
Bars as in test_gpu_parity.py: indices bit-exact for the 3-D searches; the feature search to fp32 distance resolution."""
import importlib
import os

import numpy as np
import pytest
import torch

from conftest import lattice_cloud, unit_sphere_cloud

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
knn_mod = importlib.import_module("vision3d_amd.ops.knn")  # `ops.knn` is the function


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


def _check_knn_feat(ext, oracle_mod, x, y, k):
    b, n, _ = x.shape
    m = y.shape[1]
    oracle_mod.set_threads(oracle_mod.max_threads())
    try:
        wd, wi = oracle_mod.knn_feat(x, y, k)
    finally:
        oracle_mod.set_threads(1)
    d = torch.empty((b, n, k), device=DEV)
    idx = torch.empty((b, n, k), dtype=torch.long, device=DEV)
    ext.knn_feat(T(x), T(y), d, idx, k)
    gd, gi = N_(d), N_(idx)
    kk = min(k, m)
    assert np.all(gi[..., kk:] == -1) and np.all(np.isinf(gd[..., kk:]))
    scale = np.sqrt((x ** 2).sum(-1))[..., None] + np.sqrt((y ** 2).sum(-1)).max()
    tol = 1e-5 * np.broadcast_to(scale, gd[..., :kk].shape) + 1e-6
    assert np.all(np.abs(gd[..., :kk] - wd[..., :kk]) <= tol)
    assert np.all(np.diff(gd[..., :kk], axis=-1) >= 0)
    assert np.all((gi[..., :kk] >= 0) & (gi[..., :kk] < m))
    srt = np.sort(gi[..., :kk], axis=-1)
    assert np.all(np.diff(srt, axis=-1) > 0)  # no column twice
    mismatch = gi[..., :kk] != wi[..., :kk]
    assert mismatch.mean() < 2e-3  # any index mismatch is a near-tie: same distance at that rank (checked above)
    return gd, gi, wd, wi


@pytest.mark.parametrize("shape", [(1, 50, 300, 16, 65), (2, 40, 700, 32, 128), (1, 20, 5000, 8, 500), (1, 10, 90, 8, 128),
                                   (1, 4, 3000, 4, 2048), (1, 33, 200, 3, 150)])
def test_knn_feat_k_above_64(ext, oracle_mod, shape):
    b, n, m, c, k = shape
    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal((b, n, c)).astype(np.float32)
    y = rng.standard_normal((b, m, c)).astype(np.float32)
    _check_knn_feat(ext, oracle_mod, x, y, k)


def test_knn_feat_large_k_exact_ties_come_in_index_order(ext, oracle_mod):
    """Every reference row appears three times: equal distances (identical arithmetic) must come lowest index first."""
    rng = np.random.default_rng(5)
    base = rng.standard_normal((1, 120, 16)).astype(np.float32)
    y = np.concatenate([base, base, base], axis=1)
    x = rng.standard_normal((1, 30, 16)).astype(np.float32)
    gd, gi, wd, wi = _check_knn_feat(ext, oracle_mod, x, y, 100)
    same = gd[..., 1:] == gd[..., :-1]
    assert same.mean() > 0.5
    assert np.all(gi[..., 1:][same] > gi[..., :-1][same])
    assert np.array_equal(gi, wi)


def test_knn_feat_large_k_over_several_column_chunks(ext, oracle_mod):
    """4096 x 16384 distances exceed the 128 MiB chunk: the running list is carried from chunk to chunk."""
    rng = np.random.default_rng(6)
    x = rng.standard_normal((1, 4096, 8)).astype(np.float32)
    y = rng.standard_normal((1, 16384, 8)).astype(np.float32)
    k = 70
    d = torch.empty((1, 4096, k), device=DEV)
    idx = torch.empty((1, 4096, k), dtype=torch.long, device=DEV)
    ext.knn_feat(T(x), T(y), d, idx, k)
    gd, gi = N_(d), N_(idx)
    rows = np.arange(0, 4096, 16)  # the oracle on a sample of the rows
    oracle_mod.set_threads(oracle_mod.max_threads())
    try:
        wd, wi = oracle_mod.knn_feat(x[:, rows], y, k)
    finally:
        oracle_mod.set_threads(1)
    assert np.all(np.abs(gd[:, rows] - wd) <= 1e-4)
    assert (gi[:, rows] != wi).mean() < 2e-3
    assert np.all(np.diff(gd, axis=-1) >= 0) and np.all((gi >= 0) & (gi < 16384))
    # both chunks contribute
    assert (gi < 8192).any() and (gi >= 8192).any()


def test_knn_feat_sharded_equals_single_shot_large_k(ext):
    rng = np.random.default_rng(7)
    x = rng.standard_normal((1, 100, 24)).astype(np.float32)
    y = rng.standard_normal((1, 1200, 24)).astype(np.float32)
    k, G = 100, 3
    full_d = torch.empty((1, 100, k), device=DEV)
    full_i = torch.empty((1, 100, k), dtype=torch.long, device=DEV)
    ext.knn_feat(T(x), T(y), full_d, full_i, k)
    pd = torch.empty((G, 100, k), device=DEV)
    pi = torch.empty((G, 100, k), dtype=torch.long, device=DEV)
    for g in range(G):
        ys = np.ascontiguousarray(y[:, g * 400:(g + 1) * 400])
        ext.knn_feat(T(x), T(ys), pd[g:g + 1], pi[g:g + 1], k, index_offset=g * 400)
    od = torch.empty((100, k), device=DEV)
    oi = torch.empty((100, k), dtype=torch.long, device=DEV)
    ext.topk_merge(pd, pi, od, oi)
    assert torch.equal(oi, full_i[0]) and torch.equal(od, full_d[0])


def test_knn_limits_are_errors_not_crashes(ext):
    from vision3d_amd import ops

    x = torch.zeros((1, 4, 8), device=DEV)
    with pytest.raises(RuntimeError, match="V3D_ERR_K_TOO_LARGE"):
        ext.knn_feat(x, x, torch.zeros((1, 4, 2049), device=DEV), torch.zeros((1, 4, 2049), dtype=torch.long, device=DEV),
                     2049)
    with pytest.raises(RuntimeError, match="exceeds 2048"):
        knn_mod.keops_knn(x, x, 2049)
    q = torch.zeros((4, 3), device=DEV)
    ln = torch.tensor([4], device=DEV)
    with pytest.raises(RuntimeError, match="V3D_ERR_K_TOO_LARGE"):
        ext.radius_search_pack(q, q, ln, ln, torch.zeros((4, 513), dtype=torch.long, device=DEV), 0.1, 513)
    with pytest.raises(RuntimeError, match="V3D_ERR_K_TOO_LARGE"):
        ext.knn_pack(q, q, ln, ln, torch.zeros((4, 101), device=DEV), torch.zeros((4, 101), dtype=torch.long, device=DEV),
                     101)


def test_knn_wrapper_3d_large_k_and_dilation(ext, oracle_mod):
    """3-D points with k > 100 leave the knn_points kernels for the feature route; dilated lists need k*dilation."""
    from vision3d_amd import ops

    rng = np.random.default_rng(8)
    pts = rng.random((2, 600, 3), dtype=np.float32)
    d, i = ops.knn(T(pts), T(pts), 130, return_distance=True)
    wd, wi = oracle_mod.knn_feat(pts, pts, 130)
    assert (N_(i) != wi).mean() < 2e-3 and np.allclose(N_(d), wd, atol=1e-5)
    i2 = ops.knn(T(pts), T(pts), 40, dilation=4, remove_nearest=True)  # searches 1 + 39*4 + 1 = 158 neighbours
    wd2, wi2 = oracle_mod.knn_feat(pts, pts, 158)
    assert (N_(i2) != wi2[..., 1::4]).mean() < 2e-3


# ------------------------------------------------------------------------------------------------------
# radius search with neighbor_limit > 128
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", [
    dict(q=(200, 0, 100), s=(1500, 10, 900), r=0.25, k=200),       # R = 4 list; hits below and above the limit
    dict(q=(150,), s=(3000,), r=0.3, k=512),                       # R = 8 list
    dict(q=(300,), s=(512,), r=0.4, k=300, lattice=8),             # lattice: (d2, index) ties across the list
    dict(q=(64, 64), s=(100, 2000), r=5.0, k=129),                 # radius covers everything: the first 129 of each cloud
])
def test_radius_search_pack_large_limits(ext, oracle_mod, case):
    rng = np.random.default_rng(71)

    def cloud(n):
        if case.get("lattice"):
            return (rng.integers(0, case["lattice"], (n, 3)) / case["lattice"]).astype(np.float32)
        return rng.random((n, 3)).astype(np.float32)

    q = np.concatenate([cloud(n) for n in case["q"]] + [np.zeros((0, 3), np.float32)])
    s = np.concatenate([cloud(n) for n in case["s"]] + [np.zeros((0, 3), np.float32)])
    ql, sl = np.asarray(case["q"], np.int64), np.asarray(case["s"], np.int64)
    want = oracle_mod.radius_search_pack(q, s, ql, sl, case["r"], case["k"])
    out = torch.full((q.shape[0], case["k"]), -7, dtype=torch.long, device=DEV)
    ext.radius_search_pack(T(q), T(s), T(ql), T(sl), out, case["r"], case["k"])
    assert np.array_equal(N_(out), want)
    from vision3d_amd import ops

    got = ops.radius_search_pack_mode(T(q), T(s), T(ql), T(sl), case["r"], case["k"])
    assert np.array_equal(N_(got), want)


def test_radius_neighbors_numpy_entry_widens_its_table(ext, oracle_mod):
    rng = np.random.default_rng(72)
    pts = rng.random((1500, 3)).astype(np.float32)
    lengths = np.asarray([1000, 500], np.int64)
    table = ext.radius_neighbors(pts, pts, lengths, lengths, 0.3)  # up to ~150 neighbours per query
    assert 128 < table.shape[1] < 512
    want = oracle_mod.radius_search_pack(pts, pts, lengths, lengths, 0.3, table.shape[1])
    assert np.array_equal(table, want)


# ------------------------------------------------------------------------------------------------------
# pack-mode kNN on the packed arrays
# ------------------------------------------------------------------------------------------------------
def _knn_pack_oracle(oracle_mod, q, s, ql, sl, k):
    return oracle_mod.knn_pack(q, s, ql, sl, k)


@pytest.mark.parametrize("case", [
    dict(q=(150, 90, 200), s=(150, 90, 200)),
    dict(q=(17, 0, 1, 400, 16), s=(300, 50, 120, 2500, 101)),   # empty query cloud, single query, partly filled workgroups
    dict(q=(1000,), s=(1200,)),
    dict(q=(216, 100), s=(216, 512), lattice=True),               # equal distances: the lower index first
])
@pytest.mark.parametrize("k", [1, 5, 16, 30, 63, 64, 100])
def test_knn_pack_native_vs_per_cloud_oracle(ext, oracle_mod, case, k):
    rng = np.random.default_rng(73)
    if case.get("lattice"):
        sc = [lattice_cloud(6, 0.25)[0], lattice_cloud(8, 0.125, shuffle_seed=1)[0]]
        qc = [sc[0].copy(), sc[1][:100].copy()]
    else:
        sc = [rng.random((n, 3), dtype=np.float32) for n in case["s"]]
        qc = [rng.random((n, 3), dtype=np.float32) for n in case["q"]]
    q, s = np.concatenate(qc), np.concatenate(sc)
    ql, sl = np.asarray(case["q"], np.int64), np.asarray(case["s"], np.int64)
    from vision3d_amd import _lib

    assert _lib.lib().v3d_knn_pack_uses_cell_list(len(sl), q.shape[0], s.shape[0], k) == 0
    wd, wi = _knn_pack_oracle(oracle_mod, q, s, ql, sl, k)
    d = torch.full((q.shape[0], k), -1.0, device=DEV)
    idx = torch.full((q.shape[0], k), -7, dtype=torch.long, device=DEV)
    ext.knn_pack(T(q), T(s), T(ql), T(sl), d, idx, k)
    assert np.array_equal(N_(idx), wi)
    assert np.array_equal(N_(d), wd)


def _cell_list_pack_case(name, rng):
    """(query clouds, support clouds) whose supports average >= 2048 per cloud: the cell-list route of v3d_knn_pack_f32."""
    if name == "uniform":
        sc = [rng.random((n, 3), dtype=np.float32) * np.float32([3, 3, 2.5]) for n in (3000, 2500, 5000)]
        qc = [rng.random((n, 3), dtype=np.float32) * np.float32([3, 3, 2.5]) for n in (700, 0, 900)]
    elif name == "queries_are_supports":  # every query has itself at distance 0
        sc = [rng.random((n, 3), dtype=np.float32) for n in (4000, 2200)]
        qc = [c[::5].copy() for c in sc]
    elif name == "surface":  # points on a sphere and on two planes: most cells of the box are empty
        v = rng.standard_normal((5000, 3)).astype(np.float32)
        sphere = v / np.linalg.norm(v, axis=1, keepdims=True)
        planes = rng.random((4000, 3), dtype=np.float32) * 4
        planes[:2000, 2] = 0.0
        planes[2000:, 0] = 4.0
        sc = [sphere.astype(np.float32), planes]
        qc = [np.concatenate([sphere[::20] * np.float32(0.3), sphere[::25]]).astype(np.float32),
              (rng.random((300, 3), dtype=np.float32) * 4)]
    elif name == "clusters_and_far_queries":  # distant clusters: rings run out, the linear scan of the cloud finishes
        cen = rng.random((5, 3)).astype(np.float32) * 100
        c = (cen[rng.integers(0, 5, 6000)] + rng.normal(0, 0.05, (6000, 3))).astype(np.float32)
        far = (rng.random((150, 3)) * 300 - 100).astype(np.float32)
        sc = [c]
        qc = [np.concatenate([c[::40], far]).astype(np.float32)]
    elif name == "lattice_ties":  # integer lattice, shuffled: equal distances everywhere -> lower index first
        g = np.stack(np.meshgrid(*[np.arange(14)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float32) * np.float32(0.5)
        a = g[rng.permutation(len(g))]   # 2744
        b = g[rng.permutation(len(g))]
        sc = [a, b]
        qc = [a[:150].copy(), (b[:100] + np.float32(0.25)).astype(np.float32)]
    elif name == "duplicates_and_flat":
        d = np.repeat(rng.random((600, 3), dtype=np.float32), 5, axis=0)
        d = d[rng.permutation(len(d))]  # 3000, every point five times
        f = rng.random((2500, 3), dtype=np.float32)
        f[:, 1] = 0.5
        one = np.tile(np.float32([[1.0, 2.0, 3.0]]), (2100, 1))
        sc = [d, f, one]
        qc = [d[:120].copy(), rng.random((130, 3), dtype=np.float32), rng.random((40, 3), dtype=np.float32) * 3]
    elif name == "nonfinite":
        a = rng.random((2600, 3), dtype=np.float32)
        a[7] = np.nan
        a[100, 2] = np.inf       # the cloud's box is not finite: one cell, every support a candidate
        b = rng.random((2400, 3), dtype=np.float32)
        b[3, 0] = np.nan         # NaN only: dropped from the box, parked in a border cell
        b[50] = 2e5              # farther than the 1e10 sentinel from everything
        qa = rng.random((90, 3), dtype=np.float32)
        qa[4] = np.nan
        qb = rng.random((70, 3), dtype=np.float32)
        qb[5, 1] = np.inf
        qb[6] = 1e8
        sc, qc = [a, b], [qa, qb]
    else:
        raise KeyError(name)
    return qc, sc


@pytest.mark.parametrize("name", ["uniform", "queries_are_supports", "surface", "clusters_and_far_queries",
                                  "lattice_ties", "duplicates_and_flat", "nonfinite"])
def test_knn_pack_cell_list_vs_oracle(ext, oracle_mod, name):
    """Clouds of a few thousand supports: v3d_knn_pack_f32 searches them on the radius search's cell list.  The search
    is exact -- identical indices and distances to the pair-by-pair oracle, equal distances lower index first -- on
    uniform, surface-like, clustered, tie-heavy, degenerate and non-finite clouds."""
    from vision3d_amd import _lib

    qc, sc = _cell_list_pack_case(name, np.random.default_rng(91))
    q, s = np.concatenate(qc), np.concatenate(sc)
    ql = np.asarray([len(c) for c in qc], np.int64)
    sl = np.asarray([len(c) for c in sc], np.int64)
    # The cell list is only taken when there are enough queries to pay for its build: every cloud's queries are
    # repeated `rep` times (the oracle answers them once; every copy must get that answer).
    rep = -(-200000 // q.shape[0])
    q_rep = np.concatenate([np.tile(c, (rep, 1)) for c in qc])
    ql_rep = ql * rep
    oracle_mod.set_threads(oracle_mod.max_threads())
    try:
        for k in (1, 3, 16, 38, 64):
            assert _lib.lib().v3d_knn_pack_uses_cell_list(len(sl), q_rep.shape[0], s.shape[0], k) == 1
            with np.errstate(all="ignore"):
                wd, wi = oracle_mod.knn_pack(q, s, ql, sl, k)
            d = torch.full((q_rep.shape[0], k), -1.0, device=DEV)
            idx = torch.full((q_rep.shape[0], k), -7, dtype=torch.long, device=DEV)
            ext.knn_pack(T(q_rep), T(s), T(ql_rep), T(sl), d, idx, k)
            gd, gi = N_(d), N_(idx)
            qo = ro = 0
            for n in ql:
                n = int(n)
                for r in range(rep):
                    rows = slice(ro + r * n, ro + (r + 1) * n)
                    assert np.array_equal(gi[rows], wi[qo:qo + n]), (name, k, r)
                    assert np.array_equal(gd[rows], wd[qo:qo + n], equal_nan=True), (name, k, r)
                qo += n
                ro += n * rep
    finally:
        oracle_mod.set_threads(1)


def test_knn_pack_cell_list_equals_pairwise_kernels_at_pyramid_size(ext):
    """16 clouds x 30,000 supports (config C4's finest level), 7,500 queries each: the cell list and the pair-by-pair
    kernels (reached by cutting the same supports into one cloud per call below the cell list's size) agree."""
    g = torch.Generator().manual_seed(5)
    nc, ns, nq, k = 4, 30000, 2000, 16
    box = torch.tensor([3.0, 3.0, 2.5])
    s = (torch.rand((nc * ns, 3), generator=g) * box).to(DEV)
    q = (torch.rand((nc * nq, 3), generator=g) * box).to(DEV)
    ql = torch.full((nc,), nq, dtype=torch.long, device=DEV)
    sl = torch.full((nc,), ns, dtype=torch.long, device=DEV)
    d = torch.empty((nc * nq, k), device=DEV)
    i = torch.empty((nc * nq, k), dtype=torch.long, device=DEV)
    from vision3d_amd import _lib

    assert _lib.lib().v3d_knn_pack_uses_cell_list(nc, nc * nq, nc * ns, k) == 1
    ext.knn_pack(q, s, ql, sl, d, i, k)
    # brute force per cloud with torch (fp32 differences, same expression order)
    for c in range(nc):
        qq, ss = q[c * nq:(c + 1) * nq], s[c * ns:(c + 1) * ns]
        dx = qq[:, None, 0] - ss[None, :, 0]
        dy = qq[:, None, 1] - ss[None, :, 1]
        dz = qq[:, None, 2] - ss[None, :, 2]
        d2 = (dx * dx + dy * dy) + dz * dz
        wd, wi = torch.topk(d2, k, dim=1, largest=False, sorted=True)
        assert torch.equal(d[c * nq:(c + 1) * nq], wd.sqrt())
        same = i[c * nq:(c + 1) * nq] == wi
        # (torch.topk leaves the order of equal distances open: rows may differ there only)
        rows = ~same.all(dim=1)
        if rows.any():
            dd = wd[rows]
            assert bool(((dd[:, 1:] == dd[:, :-1]).any(dim=1)).all())


@pytest.mark.parametrize("name", ["lattice", "lattice_large", "dup", "uniform_cell_list", "few_supports"])
def test_knn_lex_index_order_vs_oracle(ext, oracle_mod, name):
    """v3d_knn_lex_f32 (what ops.knn runs for 3-D points): the knn_points kernels -- list, merge and per-workgroup cell
    list -- leaving equal distances lower index first instead of replaying the heap."""
    from vision3d_amd import ops

    rng = np.random.default_rng(17)
    if name == "lattice":
        s = np.concatenate([lattice_cloud(8, 0.125, shuffle_seed=3), lattice_cloud(8, 0.125, shuffle_seed=4)])
        q = np.ascontiguousarray(s[:, :70])
    elif name == "lattice_large":  # 2744 supports, 96 queries: the cell-list kernel
        g = np.stack(np.meshgrid(*[np.arange(14)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float32) * np.float32(0.5)
        s = np.stack([g[rng.permutation(len(g))] for _ in range(2)])
        q = np.ascontiguousarray(s[:, :96])
    elif name == "dup":
        s = np.repeat(unit_sphere_cloud(rng, 2, 300), 3, axis=1)
        q = np.ascontiguousarray(s[:, ::9])
    elif name == "uniform_cell_list":
        s = rng.random((2, 4096, 3), dtype=np.float32)
        q = np.ascontiguousarray(s[:, ::16]) + np.float32(1e-3)
    else:  # fewer supports than neighbours: (1e5, 0) in the unfilled slots
        s = rng.random((2, 5, 3), dtype=np.float32)
        q = rng.random((2, 9, 3), dtype=np.float32)
    B, M, N = q.shape[0], q.shape[1], s.shape[1]
    ql, sl = np.full((B,), M, np.int64), np.full((B,), N, np.int64)
    oracle_mod.set_threads(oracle_mod.max_threads())
    try:
        for k in (1, 7, 16, 32, 63, 64, 100):
            wd, wi = oracle_mod.knn_pack(q.reshape(-1, 3), s.reshape(-1, 3), ql, sl, k)
            d = torch.full((B, M, k), -1.0, device=DEV)
            idx = torch.full((B, M, k), -7, dtype=torch.long, device=DEV)
            ext.knn_lex(T(q), T(s), d, idx, k)
            assert np.array_equal(N_(idx).reshape(-1, k), wi), (name, k)
            assert np.array_equal(N_(d).reshape(-1, k), wd), (name, k)
            if k <= N:
                od, oi = ops.knn(T(q), T(s), k, return_distance=True)
                assert torch.equal(oi, idx) and torch.equal(od, d)
    finally:
        oracle_mod.set_threads(1)


def test_ops_knn_large_3d_clouds_take_the_cell_list_and_agree_with_the_batch_kernels(ext):
    """ops.knn on 3-D clouds beyond the batch kernels' LDS cell list (> 4096 supports) runs the pack-mode cell list with
    every cloud as one cloud of a pack: identical to v3d_knn_lex_f32 (pair by pair) on the same inputs."""
    from vision3d_amd import ops

    g = torch.Generator().manual_seed(11)
    q = torch.rand((2, 20000, 3), generator=g).to(DEV)
    s = torch.rand((2, 9000, 3), generator=g).to(DEV)
    from vision3d_amd import _lib

    assert _lib.lib().v3d_knn_pack_uses_cell_list(2, 40000, 18000, 20) == 1
    s[1, 100:200] = s[1, :100]  # duplicates: equal distances
    for k in (1, 3, 20):
        d, i = ops.knn(q, s, k, return_distance=True)
        wd = torch.empty_like(d)
        wi = torch.empty_like(i)
        ext.knn_lex(q, s, wd, wi, k)
        assert torch.equal(i, wi) and torch.equal(d, wd)
    # unbatched (N, 3) inputs
    assert torch.equal(ops.knn(q[0], s[0], 5), ops.knn(q[:1], s[:1], 5)[0])


def test_knn_pack_mode_op_matches_padded_route_and_keeps_the_assertion(ext, oracle_mod):
    from vision3d_amd import ops

    rng = np.random.default_rng(74)
    ql, sl = [130, 70, 260], [400, 90, 333]
    q = rng.random((sum(ql), 3), dtype=np.float32)
    s = rng.random((sum(sl), 3), dtype=np.float32)
    tq, ts = T(q), T(s)
    tql, tsl = torch.tensor(ql, device=DEV), torch.tensor(sl, device=DEV)
    for k in (1, 8, 32):
        d, i = ops.knn_pack_mode(tq, ts, tql, tsl, k, return_distance=True)
        pd, pi = knn_mod.padded_knn_pack_mode(tq, ts, tql, tsl, k, return_distance=True)
        assert torch.equal(i, pi) and torch.equal(d, pd)
        assert torch.equal(ops.knn_pack_mode(tq, ts, tql, tsl, k), i)
    with pytest.raises(AssertionError, match="less than 91"):
        ops.knn_pack_mode(tq, ts, tql, tsl, 91)
    # k above the knn_points limit: padded route through the feature kernels
    sl2 = torch.tensor([400, 190, 233], device=DEV)  # the same 823 supports cut differently: every cloud >= 120
    i = ops.knn_pack_mode(tq, ts, tql, sl2, 120)
    pi = knn_mod.padded_knn_pack_mode(tq, ts, tql, sl2, 120)
    assert i.shape == (sum(ql), 120) and torch.equal(i, pi)
    for c, (qo, n) in enumerate(zip((0, 130, 200), (400, 190, 233))):
        rows = i[qo:qo + ql[c]]
        assert int(rows.min()) >= 0 and int(rows.max()) < n


def test_knn_pack_lengths_that_overrun_the_arrays_are_cut(ext):
    """Lengths are device data nobody validated: a sum beyond the arrays must not read or write outside them."""
    q = torch.rand((100, 3), device=DEV)
    s = torch.rand((80, 3), device=DEV)
    d = torch.full((100, 4), -1.0, device=DEV)
    i = torch.full((100, 4), -7, dtype=torch.long, device=DEV)
    ext.knn_pack(q, s, torch.tensor([60, 500], device=DEV), torch.tensor([50, 10 ** 9], device=DEV), d, i, 4)
    torch.cuda.synchronize()
    assert int(i[:60].max()) < 50 and int(i[60:].max()) < 30 and int(i.min()) >= 0


def test_radius_neighbors_is_strict_at_the_radius(ext, oracle_mod):
    """Lattice with spacing 0.25 and radius 0.25: the six axis neighbours sit exactly AT the radius; the NumPy entry
    point mirrors the CPU code's strict nanoflann search (radius_neighbors.cpp), the search kernel itself is inclusive."""
    pts = lattice_cloud(5, 0.25)[0].astype(np.float32)
    lengths = np.asarray([pts.shape[0]], np.int64)
    table = ext.radius_neighbors(pts, pts, lengths, lengths, 0.25)
    strict = oracle_mod.radius_count_pack(pts, pts, lengths, lengths, 0.25)
    assert table.shape[1] == strict.max() == 1  # only the point itself is strictly inside
    assert np.array_equal(table[:, 0], np.arange(pts.shape[0]))
    inclusive = oracle_mod.radius_search_pack(pts, pts, lengths, lengths, 0.25, 8)
    assert (inclusive != pts.shape[0]).sum(axis=1).max() == 7


def test_selected_scores_carry_gradients_like_the_reference_indexing(ext):
    """mutual_topk_select.py:57 returns score_mat[rows, cols]: differentiable.  Same here when a gradient can be asked."""
    from vision3d_amd import ops

    g = torch.Generator().manual_seed(3)
    score = torch.rand((40, 50), generator=g).to(DEV).requires_grad_(True)
    rows, cols, vals = ops.mutual_topk_select(score, 2, largest=True)
    assert vals.requires_grad and torch.equal(vals.detach(), score.detach()[rows, cols])
    vals.sum().backward()
    want = torch.zeros_like(score)
    want[rows, cols] = 1.0
    assert torch.equal(score.grad, want)
    with torch.no_grad():
        r2, c2, v2 = ops.mutual_topk_select(score, 2, largest=True)
    assert not v2.requires_grad and torch.equal(r2, rows) and torch.equal(v2, vals.detach())


def test_pairwise_distance_backward_blocks_exactly_the_clipped_cells(ext):
    """clamp(min=0) (pairwise_distance.py:53): no gradient through cells the clamp clipped, with or without the sqrt."""
    from vision3d_amd import ops

    g = torch.Generator().manual_seed(4)
    x = torch.randn((1, 64, 32), generator=g).to(DEV)
    y = x.clone()
    y[:, 32:] += 0.5 * torch.randn((1, 32, 32), generator=g).to(DEV)  # rows 0..31 coincide with x: d_raw ~ +-1e-6
    for squared in (True, False):
        xr = x.clone().requires_grad_(True)
        out = ops.pairwise_distance(xr, y, squared=squared)
        go = torch.ones_like(out)
        (gx,) = torch.autograd.grad(out, xr, go)
        floor = 0.0 if squared else float(torch.sqrt(torch.tensor(1e-8)))
        clipped = out.detach() <= floor
        assert clipped[0, :32, :32].diagonal().float().mean() > 0.2  # the coinciding pairs do clip
        # reference formula with the same mask: d = |x|^2 - 2xy + |y|^2, grad_x = 2 (sum_m g) x - 2 g y
        gm = torch.where(clipped, torch.zeros_like(go), go if squared else go / (2.0 * out.detach()))
        want = 2.0 * (gm.sum(-1, keepdim=True) * x - torch.matmul(gm, y))
        torch.testing.assert_close(gx, want, rtol=1e-4, atol=1e-4)
        assert torch.isfinite(gx).all()


def test_pack_mode_ops_take_lengths_on_the_host_and_in_int32(ext):
    """The reference reads lengths with `.item()` loops (ops/conversion.py:7-12): any device, any integer type."""
    from vision3d_amd import ops

    g = torch.Generator().manual_seed(9)
    pts = torch.rand((700, 3), generator=g).to(DEV)
    dev_len = torch.tensor([300, 400], device=DEV)
    host_len = torch.tensor([300, 400], dtype=torch.int32)
    assert torch.equal(ops.knn_pack_mode(pts, pts, host_len, host_len, 6), ops.knn_pack_mode(pts, pts, dev_len, dev_len, 6))
    assert torch.equal(ops.radius_search_pack_mode(pts, pts, host_len, host_len, 0.1, 20),
                       ops.radius_search_pack_mode(pts, pts, dev_len, dev_len, 0.1, 20))
    assert torch.equal(ops.radius_count_pack_mode(pts, pts, host_len, host_len, 0.1),
                       ops.radius_count_pack_mode(pts, pts, dev_len, dev_len, 0.1))
