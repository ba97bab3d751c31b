"""GPU parity: every HIP entry point (called through the C ABI via vision3d_amd.ext) against the CPU oracle on
identical seeded inputs.  Bar (BASELINE.json north_star): indices bit-exact, ties included; float outputs
within 1e-5 relative (here they are in fact bit-identical to the oracle except where noted)."""
import os

import numpy as np
import pytest
import torch

from conftest import lattice_cloud, unit_sphere_cloud

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


def clouds(rng):
    """name -> (B, N, 3) float32 inputs: random, sphere-shaped, tie-heavy and odd sizes."""
    return {
        "uniform_2x1000": rng.random((2, 1000, 3), dtype=np.float32),
        "sphere_3x1024": unit_sphere_cloud(rng, 3, 1024),
        "lattice_216": lattice_cloud(6, 0.25),
        "lattice_shuffled_512": lattice_cloud(8, 0.125, shuffle_seed=1),
        "dup_120": np.repeat(unit_sphere_cloud(rng, 1, 40), 3, axis=1),
        "uniform_1x2500": rng.random((1, 2500, 3), dtype=np.float32),
        "tiny_1x5": rng.random((1, 5, 3), dtype=np.float32),
    }


# ------------------------------------------------------------------------------------------------------
# furthest point sampling
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,m", [(1, 1), (5, 5), (50, 20), (64, 64), (100, 37), (200, 64), (256, 100), (1000, 128),
                                 (1024, 512), (2000, 300), (4096, 256), (5000, 200), (10000, 128), (16384, 64),
                                 (20000, 48)])
def test_fps_random(ext, oracle_mod, n, m):
    rng = np.random.default_rng(n * 7 + m)
    pts = rng.random((3, n, 3), dtype=np.float32)
    want = oracle_mod.furthest_point_sample(pts, m)
    dist = torch.full((3, n), 1e10, dtype=torch.float32, device=DEV)
    idx = torch.zeros((3, m), dtype=torch.long, device=DEV)
    ext.furthest_point_sample(T(pts), dist, idx, m)
    assert np.array_equal(N_(idx), want)


@pytest.mark.parametrize("name", ["u100", "u257", "s1024", "u1000"])
def test_fps_vs_reference_native_fps_fixture(ext, golden, name):
    """Kernel == the reference's own NumPy FPS (array_ops/furthest_point_sample.py:22-66) on the committed fixture
    (tests/golden/ref_array_fps.npz: clouds with a clear winner in every round), all clouds of the fixture in one batch
    position each, and through the NumPy entry point `ext.sample_nodes_with_fps` the reference's array_ops calls."""
    g = golden("ref_array_fps.npz")
    pts, want = g[f"{name}_pts"], g[f"{name}_idx"]
    n, m = pts.shape[0], len(want)
    dist = torch.full((2, n), 1e10, dtype=torch.float32, device=DEV)
    idx = torch.zeros((2, m), dtype=torch.long, device=DEV)
    ext.furthest_point_sample(T(np.stack([pts, pts[::-1]])), dist, idx, m)
    assert np.array_equal(N_(idx)[0], want)
    assert np.array_equal(ext.sample_nodes_with_fps(pts, -1.0, m), want)


@pytest.mark.parametrize("side,scale", [(4, 0.5), (6, 0.25), (8, 0.125), (11, 0.5), (17, 0.25), (27, 1.0)])
def test_fps_lattice_ties(ext, oracle_mod, side, scale):
    """Integer lattices: almost every round has many equally-far candidates, so the reference's geometry-dependent
    tie rule decides nearly every pick.  Sizes cover every register-resident kernel variant and the streaming one."""
    for seed in (None, 3):
        pts = lattice_cloud(side, scale, shuffle_seed=seed)
        n = pts.shape[1]
        m = min(n, 96)
        want = oracle_mod.furthest_point_sample(pts, m)
        dist = torch.full((1, n), 1e10, dtype=torch.float32, device=DEV)
        idx = torch.zeros((1, m), dtype=torch.long, device=DEV)
        ext.furthest_point_sample(T(pts), dist, idx, m)
        assert np.array_equal(N_(idx), want), (side, seed)


def test_fps_more_samples_than_points_and_scratch_state(ext, oracle_mod):
    pts = np.random.default_rng(0).random((2, 40, 3), dtype=np.float32)
    want = oracle_mod.furthest_point_sample(pts, 55)
    dist = torch.full((2, 40), 1e10, dtype=torch.float32, device=DEV)
    idx = torch.zeros((2, 55), dtype=torch.long, device=DEV)
    ext.furthest_point_sample(T(pts), dist, idx, 55)
    assert np.array_equal(N_(idx), want)
    assert torch.all(dist == 0)  # every point was selected: the in/out scratch holds the final min-distances


@pytest.mark.parametrize("n", [100, 700, 1024, 3000, 4096, 8192, 12000, 20000])
def test_fps_in_pieces_equals_one_call(ext, oracle_mod, n):
    """`first` / `last` (v3d_fps_resume_f32): the rounds of one sampling spread over several launches -- every kernel
    geometry (split block, table, winner re-read, streaming), random and tie-heavy clouds -- give the indices and the
    final running distances of the single launch, which equal the oracle's."""
    rng = np.random.default_rng(n)
    m = min(n, 330)
    for lattice in (False, True):
        pts = rng.random((2, n, 3), dtype=np.float32)
        if lattice:
            pts = (np.round(pts * 5) / 5).astype(np.float32)
        want = oracle_mod.furthest_point_sample(pts, m)
        one_d = torch.full((2, n), 1e10, dtype=torch.float32, device=DEV)
        one_i = torch.zeros((2, m), dtype=torch.long, device=DEV)
        ext.furthest_point_sample(T(pts), one_d, one_i, m)
        assert np.array_equal(N_(one_i), want)
        dist = torch.full((2, n), 1e10, dtype=torch.float32, device=DEV)
        idx = torch.full((2, m), -7, dtype=torch.long, device=DEV)
        cuts = [c for c in (0, 64, 256, 320) if c < m] + [m]
        for a, b in zip(cuts[:-1], cuts[1:]):
            ext.furthest_point_sample_rounds(T(pts), dist, idx, m, a, b)
            assert torch.all(idx[:, b:] == -7)  # nothing beyond the piece is written
        assert torch.equal(idx, one_i), (n, lattice)
        assert torch.equal(dist, one_d), (n, lattice)


def test_fps_in_pieces_argument_checks(ext):
    pts = torch.rand((1, 200, 3), device=DEV)
    dist = torch.full((1, 200), 1e10, device=DEV)
    idx = torch.zeros((1, 128), dtype=torch.long, device=DEV)
    with pytest.raises(RuntimeError, match="multiple of 64"):
        ext.furthest_point_sample_rounds(pts, dist, idx, 128, 32, 128)
    with pytest.raises(RuntimeError, match="beyond"):
        ext.furthest_point_sample_rounds(pts, dist, idx, 128, 64, 129)
    ext.furthest_point_sample_rounds(pts, dist, idx, 128, 128, 128)  # nothing left to do: not an error


def test_fps_full_config_sizes(ext, oracle_mod):
    """BASELINE configs C1 (B=16, N=1024 -> 512) and C2 (B=32, N=4096 -> 1024), full size, bit-exact."""
    rng = np.random.default_rng(0)
    oracle_mod.set_threads(oracle_mod.max_threads())
    try:
        for b, n, m, make in ((16, 1024, 512, unit_sphere_cloud), (32, 4096, 1024, None)):
            pts = make(rng, b, n) if make else rng.random((b, n, 3), dtype=np.float32)
            want = oracle_mod.furthest_point_sample(pts, m)
            dist = torch.full((b, n), 1e10, dtype=torch.float32, device=DEV)
            idx = torch.zeros((b, m), dtype=torch.long, device=DEV)
            ext.furthest_point_sample(T(pts), dist, idx, m)
            got = N_(idx)
            assert np.array_equal(got, want)
            assert all(len(set(r.tolist())) == m for r in got)  # size-independent property: no repeats
    finally:
        oracle_mod.set_threads(1)


# ------------------------------------------------------------------------------------------------------
# ball query / knn_points / three_nn
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_ball_query(ext, oracle_mod, dtype):
    rng = np.random.default_rng(11)
    for name, pts in clouds(rng).items():
        s = pts.astype(dtype)
        q = np.concatenate([s[:, : min(70, s.shape[1])], s[:, :3] + 50.0], axis=1)  # last 3 queries: empty balls
        for k, r in ((1, 0.05), (8, 0.3), (32, 0.2), (70, 0.6), (130, 10.0)):
            want = oracle_mod.ball_query(q, s, k, r)
            idx = torch.zeros((q.shape[0], q.shape[1], k), dtype=torch.long, device=DEV)
            ext.ball_query(T(q), T(s), idx, r, k)
            assert np.array_equal(N_(idx), want), (name, k, r)


def test_ball_query_multi_equals_single_scale_queries(ext, oracle_mod):
    """The multi-scale pass (PointNet++ MSG) against the oracle scale by scale: sorted and unsorted radii, a scale that
    never fills (full scan) next to one that fills at once, lattice / duplicate ties, empty balls, ragged M and N."""
    from vision3d_amd import ops

    rng = np.random.default_rng(21)
    for name, pts in clouds(rng).items():
        s = pts.astype(np.float32)
        q = np.concatenate([s[:, : min(70, s.shape[1])], s[:, :3] + 50.0], axis=1)
        for ks, rs in (((16, 32, 128), (0.1, 0.2, 0.4)), ((8, 1), (0.3, 0.05)), ((130, 4, 32, 70), (10.0, 0.01, 0.2, 0.6)),
                       ((5,), (0.25,))):
            got = ops.ball_query_multi(T(q), T(s), ks, rs)
            assert len(got) == len(ks)
            for k, r, g in zip(ks, rs, got):
                assert np.array_equal(N_(g), oracle_mod.ball_query(q, s, k, r)), (name, k, r)
    # transposed layout and the single-call form agree
    s = rng.random((2, 300, 3), dtype=np.float32)
    a = ops.ball_query_multi(T(s).transpose(1, 2), T(s).transpose(1, 2), (8, 16), (0.1, 0.2), transposed=True)
    assert torch.equal(a[0], ops.ball_query(T(s), T(s), 8, 0.1)) and torch.equal(a[1], ops.ball_query(T(s), T(s), 16, 0.2))
    with pytest.raises(RuntimeError, match="scales"):
        ext.ball_query_multi(T(s), T(s), [], [], [])


def test_ball_query_multi_full_size_c2(ext):
    """BASELINE config 2 (B 32, N 4096, M 1024, the three MSG scales): identical to three single-scale launches."""
    from vision3d_amd import ops

    g = torch.Generator(device=DEV).manual_seed(2)
    s = torch.rand((32, 4096, 3), device=DEV, generator=g)
    q = s[:, ::4].contiguous()
    got = ops.ball_query_multi(q, s, (16, 32, 128), (0.1, 0.2, 0.4))
    for k, r, t in zip((16, 32, 128), (0.1, 0.2, 0.4), got):
        assert torch.equal(t, ops.ball_query(q, s, k, r)), (k, r)


def test_ball_query_leaves_unmatched_rows_untouched(ext):
    s = torch.zeros((1, 10, 3), device=DEV)
    q = torch.full((1, 4, 3), 100.0, device=DEV)
    idx = torch.full((1, 4, 5), 7, dtype=torch.long, device=DEV)  # not zero-filled on purpose
    ext.ball_query(q, s, idx, 0.5, 5)
    assert torch.all(idx == 7)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_knn_points(ext, oracle_mod, dtype):
    rng = np.random.default_rng(12)
    for name, pts in clouds(rng).items():
        s = pts.astype(dtype)
        q = s[:, : min(45, s.shape[1])] + (0.0 if "lattice" in name or "dup" in name else 0.001)
        q = q.astype(dtype)
        for k in (1, 3, 16, 20, 24, 32, 63, 64, 100):  # 24..63 fp32: batched bitonic-merge kernel
            wd, wi = oracle_mod.knn_points(q, s, k)
            d = torch.zeros((q.shape[0], q.shape[1], k), dtype=T(q).dtype, device=DEV)
            idx = torch.zeros((q.shape[0], q.shape[1], k), dtype=torch.long, device=DEV)
            ext.knn_points(T(q), T(s), d, idx, k)
            assert np.array_equal(N_(idx), wi), (name, k)
            assert np.array_equal(N_(d), wd), (name, k)


@pytest.mark.parametrize("n", [63, 64, 255, 256, 1023, 1024, 1025, 2048, 2049])
def test_knn_points_tile_and_prepass_boundaries(ext, oracle_mod, n):
    """Support counts either side of the lane-minimum pre-pass threshold (256) and of the 1024-point LDS tile (one
    tile: staged once for pre-pass and search; several: staged twice; last tile padded with +inf points), on the list
    kernel (k = 8), the merge kernel (k = 32), the two-register list (k = 80) and fp64."""
    rng = np.random.default_rng(n)
    s = rng.random((2, n, 3), dtype=np.float32)
    q = rng.random((2, 37, 3), dtype=np.float32)
    for dtype, ks in ((np.float32, (8, 32, 80)), (np.float64, (8, 32))):
        sd, qd = s.astype(dtype), q.astype(dtype)
        for k in ks:
            if k > n:
                continue
            wd, wi = oracle_mod.knn_points(qd, sd, k)
            d = torch.zeros((2, 37, k), dtype=T(qd).dtype, device=DEV)
            idx = torch.zeros((2, 37, k), dtype=torch.long, device=DEV)
            ext.knn_points(T(qd), T(sd), d, idx, k)
            assert np.array_equal(N_(idx), wi), (dtype, k)
            assert np.array_equal(N_(d), wd), (dtype, k)


def _cell_list_clouds(rng):
    """Clouds of 2048..4096 supports for the per-workgroup cell-list kernel (fp32, k + 1 <= 64, >= 64 queries):
    name -> (supports (B, N, 3), queries (B, M, 3))."""
    out = {}
    s = rng.random((2, 4096, 3), dtype=np.float32)
    out["uniform_4096"] = (s, np.ascontiguousarray(s[:, ::8]) + np.float32(1e-3))
    s = rng.random((2, 2048, 3), dtype=np.float32)
    out["uniform_2048_ragged_queries"] = (s, rng.random((2, 203, 3), dtype=np.float32))
    # a few tight clusters plus far outliers: almost every support in a handful of cells (runs of more than 64
    # records), most cells empty, queries between the clusters have to widen ring by ring
    centres = rng.random((1, 6, 3)).astype(np.float32) * 10
    c = centres[:, rng.integers(0, 6, 3000)] + rng.normal(0, 0.01, (1, 3000, 3)).astype(np.float32)
    c[0, :5] = [[100, 100, 100], [-50, 0, 0], [0, 200, 0], [3, 3, 3], [7, 1, 5]]
    qc = np.concatenate([c[:, ::30], rng.random((1, 60, 3)).astype(np.float32) * 10], axis=1)
    out["clustered_3000"] = (c.astype(np.float32), qc.astype(np.float32))
    # integer lattice, shuffled: every query has many equidistant supports -> the heap replay on every row
    g = np.stack(np.meshgrid(*[np.arange(13)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float32) * np.float32(0.5)
    g = g[rng.permutation(len(g))][None]  # 2197 supports
    out["lattice_2197"] = (np.ascontiguousarray(g), np.ascontiguousarray(g[:, :96]))
    # every support repeated 4 times
    d = np.repeat(unit_sphere_cloud(rng, 1, 640), 4, axis=1)
    d = d[:, rng.permutation(d.shape[1])]
    out["dup_2560"] = (np.ascontiguousarray(d), np.ascontiguousarray(d[:, :80]))
    # flat cloud (one layer of cells along z) with queries off the plane and outside the bounding box
    f = rng.random((1, 2500, 3), dtype=np.float32)
    f[..., 2] = 0.25
    qf = rng.random((1, 100, 3), dtype=np.float32) * 3 - 1
    out["flat_2500_queries_outside"] = (f, qf.astype(np.float32))
    # one cloud that is a single point 2100 times, one line
    one = np.tile(np.float32([[0.3, 0.6, 0.9]]), (1, 2100, 1))
    line = np.zeros((1, 2100, 3), np.float32)
    line[0, :, 0] = rng.random(2100, dtype=np.float32)
    out["degenerate_2100"] = (np.concatenate([one, line]), rng.random((2, 70, 3), dtype=np.float32))
    # non-finite supports and queries: NaN / +-inf coordinates never enter a list (knn_points_cuda.cuh:80: d < best)
    n = rng.random((1, 2300, 3), dtype=np.float32)
    n[0, 5] = np.nan
    n[0, 77, 1] = np.inf
    n[0, 1500, 0] = -np.inf
    n[0, 2000] = 1e6
    qn = rng.random((1, 64, 3), dtype=np.float32)
    qn[0, 3, 0] = np.nan
    qn[0, 9] = np.inf
    qn[0, 11] = 1e8  # every real distance above the 1e10 sentinel
    out["nonfinite_2300"] = (n, qn)
    return out


@pytest.mark.parametrize("name", ["uniform_4096", "uniform_2048_ragged_queries", "clustered_3000", "lattice_2197",
                                  "dup_2560", "flat_2500_queries_outside", "degenerate_2100", "nonfinite_2300"])
def test_knn_points_cell_list_kernel(ext, oracle_mod, name):
    """The per-workgroup cell-list kernel (knn_points_grid_kernel: fp32, 2048 <= N <= 4096, M >= 64, k <= 63) is an
    exact search: indices and distances identical to the oracle's heap on uniform, clustered, tie-heavy, flat,
    degenerate and non-finite clouds, for k either side of its internal thresholds."""
    s, q = _cell_list_clouds(np.random.default_rng(77))[name]
    oracle_mod.set_threads(oracle_mod.max_threads())
    try:
        for k in (1, 2, 12, 13, 16, 32, 63):
            with np.errstate(all="ignore"):
                wd, wi = oracle_mod.knn_points(q, s, k)
            d = torch.full((q.shape[0], q.shape[1], k), -1.0, device=DEV)
            idx = torch.full((q.shape[0], q.shape[1], k), -1, dtype=torch.long, device=DEV)
            ext.knn_points(T(q), T(s), d, idx, k)
            assert np.array_equal(N_(idx), wi), (name, k)
            assert np.array_equal(N_(d), wd, equal_nan=True), (name, k)
    finally:
        oracle_mod.set_threads(1)


def test_knn_points_k_too_large_raises(ext):
    q = torch.zeros((1, 2, 3), device=DEV)
    with pytest.raises(RuntimeError, match="V3D_ERR_K_TOO_LARGE"):
        ext.knn_points(q, q, torch.zeros((1, 2, 101), device=DEV), torch.zeros((1, 2, 101), dtype=torch.long, device=DEV),
                       101)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_three_nn(ext, oracle_mod, dtype):
    rng = np.random.default_rng(13)
    for name, pts in clouds(rng).items():
        s = pts.astype(dtype)
        q = np.concatenate([s[:, ::3], rng.random((s.shape[0], 300, 3)).astype(dtype)], axis=1)
        wd, wi = oracle_mod.three_nn(q, s)
        d = torch.zeros((q.shape[0], q.shape[1], 3), dtype=T(q).dtype, device=DEV)
        idx = torch.zeros((q.shape[0], q.shape[1], 3), dtype=torch.long, device=DEV)
        ext.three_nn(T(q), T(s), d, idx)
        assert np.array_equal(N_(idx), wi), name
        assert np.array_equal(N_(d), wd), name
    # fewer than three supports: (-1, 1e20) in the unfilled slots
    s2 = rng.random((1, 2, 3), dtype=np.float32)
    wd, wi = oracle_mod.three_nn(s2, s2)
    d = torch.zeros((1, 2, 3), device=DEV)
    idx = torch.zeros((1, 2, 3), dtype=torch.long, device=DEV)
    ext.three_nn(T(s2), T(s2), d, idx)
    assert np.array_equal(N_(idx), wi) and np.array_equal(N_(d), wd)


def test_query_full_config_sizes(ext, oracle_mod):
    """C1 / C2 shapes at full size against the (multi-threaded) oracle."""
    rng = np.random.default_rng(1)
    oracle_mod.set_threads(oracle_mod.max_threads())
    try:
        s = unit_sphere_cloud(rng, 16, 1024)
        q = s[:, :512]
        want = oracle_mod.ball_query(q, s, 32, 0.2)
        idx = torch.zeros((16, 512, 32), dtype=torch.long, device=DEV)
        ext.ball_query(T(q), T(s), idx, 0.2, 32)
        assert np.array_equal(N_(idx), want)

        s = rng.random((32, 4096, 3), dtype=np.float32)
        q = s[:, ::4]
        for r, k in ((0.1, 16), (0.2, 32), (0.4, 128)):
            want = oracle_mod.ball_query(q, s, k, r)
            idx = torch.zeros((32, 1024, k), dtype=torch.long, device=DEV)
            ext.ball_query(T(q), T(s), idx, r, k)
            assert np.array_equal(N_(idx), want), (r, k)
        wd, wi = oracle_mod.three_nn(s, q)  # feature propagation: 4096 queries <- 1024 supports
        d = torch.zeros((32, 4096, 3), device=DEV)
        idx = torch.zeros((32, 4096, 3), dtype=torch.long, device=DEV)
        ext.three_nn(T(s), T(q), d, idx)
        assert np.array_equal(N_(idx), wi) and np.array_equal(N_(d), wd)
        wd, wi = oracle_mod.knn_points(q[:8], s[:8], 16)
        d = torch.zeros((8, 1024, 16), device=DEV)
        idx = torch.zeros((8, 1024, 16), dtype=torch.long, device=DEV)
        ext.knn_points(T(q[:8]), T(s[:8]), d, idx, 16)
        assert np.array_equal(N_(idx), wi) and np.array_equal(N_(d), wd)
        assert torch.all(d[..., 1:] >= d[..., :-1])  # sortedness
        for k in (32, 63):  # batched bitonic-merge kernel at full cloud size
            wd, wi = oracle_mod.knn_points(q[:4], s[:4], k)
            d = torch.zeros((4, 1024, k), device=DEV)
            idx = torch.zeros((4, 1024, k), dtype=torch.long, device=DEV)
            ext.knn_points(T(q[:4]), T(s[:4]), d, idx, k)
            assert np.array_equal(N_(idx), wi) and np.array_equal(N_(d), wd)
    finally:
        oracle_mod.set_threads(1)


def test_golden_tie_fixture_on_gpu(ext, golden):
    g = golden("oracle_ties.npz")
    for name in ("lattice", "lattice_shuffled", "dup", "rand100"):
        pts = g[f"{name}_pts"]
        b, n = pts.shape[:2]
        m = min(n, 48)
        dist = torch.full((b, n), 1e10, dtype=torch.float32, device=DEV)
        idx = torch.zeros((b, m), dtype=torch.long, device=DEV)
        ext.furthest_point_sample(T(pts), dist, idx, m)
        assert np.array_equal(N_(idx), g[f"{name}_fps"]), name
        qs = pts[:, : min(n, 32)]
        bi = torch.zeros((b, qs.shape[1], 8), dtype=torch.long, device=DEV)
        ext.ball_query(T(qs), T(pts), bi, 0.3, 8)
        assert np.array_equal(N_(bi), g[f"{name}_ball"]), name
        kd = torch.zeros((b, qs.shape[1], 12), device=DEV)
        ki = torch.zeros((b, qs.shape[1], 12), dtype=torch.long, device=DEV)
        ext.knn_points(T(qs), T(pts), kd, ki, 12)
        assert np.array_equal(N_(ki), g[f"{name}_knn_i"]) and np.array_equal(N_(kd), g[f"{name}_knn_d"]), name
        td = torch.zeros((b, qs.shape[1], 3), device=DEV)
        ti = torch.zeros((b, qs.shape[1], 3), dtype=torch.long, device=DEV)
        ext.three_nn(T(qs), T(pts), td, ti)
        assert np.array_equal(N_(ti), g[f"{name}_tnn_i"]) and np.array_equal(N_(td), g[f"{name}_tnn_d"]), name


def test_query_matches_reference_scipy_fixture(ext, golden):
    """The reference CPU path itself (array_ops/knn.py via scipy, committed vectors) on the GPU kernels."""
    g = golden("ref_array_knn.npz")
    q, s, k = g["q"], g["s"], int(g["k"])
    d = torch.zeros((1, q.shape[0], k), device=DEV)
    idx = torch.zeros((1, q.shape[0], k), dtype=torch.long, device=DEV)
    ext.knn_points(T(q[None]), T(s[None]), d, idx, k)
    assert np.array_equal(N_(idx)[0], g["idx"])
    np.testing.assert_allclose(N_(d)[0], g["dist"], rtol=1e-5, atol=1e-7)
    g = golden("ref_array_ball_query.npz")
    q, s, radius, lens, flat = g["q"], g["s"], float(g["radius"]), g["lens"], g["flat"]
    kmax = int(lens.max()) + 1
    bi = torch.zeros((1, q.shape[0], kmax), dtype=torch.long, device=DEV)
    ext.ball_query(T(q[None]), T(s[None]), bi, radius, kmax)
    off = 0
    for row, n in zip(N_(bi)[0], lens):
        assert np.array_equal(row[:n], flat[off:off + n])
        off += n


# ------------------------------------------------------------------------------------------------------
# gather / group_gather / three_interpolate, forward and backward
# ------------------------------------------------------------------------------------------------------
GATHER_DTYPES = [np.uint8, np.int8, np.bool_, np.int16, np.int32, np.int64, np.float32, np.float64]


@pytest.mark.parametrize("dtype", GATHER_DTYPES)
def test_gather_forward_all_dtypes(ext, oracle_mod, dtype):
    rng = np.random.default_rng(21)
    for (b, c, n, m, k) in ((2, 3, 17, 9, 1), (2, 5, 1024, 128, 16), (1, 7, 333, 50, 7), (2, 4, 9000, 257, 5)):
        src = (rng.integers(0, 2, (b, c, n)) if dtype == np.bool_ else rng.integers(-100, 100, (b, c, n))).astype(dtype)
        idx = rng.integers(0, n, (b, m, k))
        want = oracle_mod.group_gather_forward(src, idx)
        out = torch.zeros((b, c, m, k), dtype=T(src).dtype, device=DEV)
        ext.group_gather_forward(T(src), T(idx), out)
        assert np.array_equal(N_(out), want), (dtype, n, m, k)
        want1 = oracle_mod.gather_forward(src, idx[:, :, 0])
        out1 = torch.zeros((b, c, m), dtype=T(src).dtype, device=DEV)
        ext.gather_forward(T(src), T(idx[:, :, 0].copy()), out1)
        assert np.array_equal(N_(out1), want1)


def test_group_gather_forward_config_shapes(ext, oracle_mod):
    """C1 (B=16,C=64,N=1024,M=512,K=32) and C2/C3-like shapes, plus a row too large for LDS (direct path)."""
    rng = np.random.default_rng(22)
    for (b, c, n, m, k) in ((16, 64, 1024, 512, 32), (4, 67, 4096, 1024, 16), (2, 64, 1024, 1024, 20),
                            (2, 16, 30000, 4096, 8)):
        src = rng.standard_normal((b, c, n)).astype(np.float32)
        idx = rng.integers(0, n, (b, m, k))
        out = torch.zeros((b, c, m, k), device=DEV)
        ext.group_gather_forward(T(src), T(idx), out)
        want = oracle_mod.group_gather_forward(src, idx)
        assert np.array_equal(N_(out), want)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_gather_backward_bit_exact(ext, oracle_mod, dtype):
    rng = np.random.default_rng(23)
    # chunked LDS path with 1..16 column chunks, both channel-tile widths (n <= 1024 / n <= 4096), partial channel
    # tiles, odd row lengths (element-wise staging), and the generic path (n > 4096)
    for (b, c, n, m, k) in ((2, 3, 17, 9, 4), (2, 13, 1024, 256, 16), (1, 9, 50, 700, 3), (2, 8, 3000, 100, 2),
                            (3, 10, 4096, 1024, 32), (2, 6, 5000, 300, 7), (1, 5, 4097, 33, 3)):
        idx = rng.integers(0, n, (b, m, k))
        idx[:, : m // 3] = 0  # a heavy group (ball-query padding pattern): one source spanning several chunks
        g = rng.standard_normal((b, c, m, k)).astype(dtype)
        want = oracle_mod.group_gather_backward(g, idx, n)
        out = torch.full((b, c, n), 7.0, dtype=T(g).dtype, device=DEV)  # must be fully overwritten
        ext.group_gather_backward(T(g), T(idx), out, n)
        assert np.array_equal(N_(out), want), (dtype, n, m, k)
        g1 = np.ascontiguousarray(g[..., 0])
        i1 = np.ascontiguousarray(idx[..., 0])
        want1 = oracle_mod.gather_backward(g1, i1, n)
        out1 = torch.zeros((b, c, n), dtype=T(g).dtype, device=DEV)
        ext.gather_backward(T(g1), T(i1), out1, n)
        assert np.array_equal(N_(out1), want1)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_three_interpolate(ext, oracle_mod, dtype):
    rng = np.random.default_rng(24)
    for (b, c, s, q) in ((2, 5, 11, 7), (2, 33, 1024, 4096), (1, 4, 9000, 130), (3, 128, 256, 1001)):
        src = rng.standard_normal((b, c, s)).astype(dtype)
        idx = rng.integers(0, s, (b, q, 3))
        w = rng.random((b, q, 3)).astype(dtype)
        w /= w.sum(-1, keepdims=True)
        want = oracle_mod.three_interpolate_forward(src, idx, w)
        out = torch.zeros((b, c, q), dtype=T(src).dtype, device=DEV)
        ext.three_interpolate_forward(T(src), T(idx), T(w), out)
        assert np.array_equal(N_(out), want), (dtype, s, q)
        g = rng.standard_normal((b, c, q)).astype(dtype)
        wantb = oracle_mod.three_interpolate_backward(g, idx, w, s)
        outb = torch.full((b, c, s), 3.0, dtype=T(src).dtype, device=DEV)
        ext.three_interpolate_backward(T(g), T(idx), T(w), outb, s)
        assert np.array_equal(N_(outb), wantb), (dtype, s, q)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_scatter_build_apply_equals_one_call_backward(ext, oracle_mod, dtype):
    """v3d_scatter_build + v3d_scatter_apply (lists built ONCE per index tensor, summed with as often as wanted) against
    the oracle and against the one-call entry points, bit for bit: both list layouts (<= 4096 sources in LDS, more in the
    global CSR), both modes, two channel widths on one set of lists, refusals."""
    from vision3d_amd import _lib

    rng = np.random.default_rng(26)
    tdt = T(np.zeros(1, dtype)).dtype
    for (b, n, m, k) in ((2, 1024, 256, 16), (3, 4096, 512, 32), (2, 5000, 300, 7), (1, 50, 700, 3)):
        idx = rng.integers(0, n, (b, m, k))
        idx[:, : m // 3] = 0
        ti = T(idx)
        lists = ext.scatter_build(ti, n)
        for c in (3, 10):  # the xyz and the feature gather of one layer share the lists
            g = rng.standard_normal((b, c, m, k)).astype(dtype)
            out = torch.full((b, c, n), 7.0, dtype=tdt, device=DEV)
            ext.scatter_apply(T(g), lists, out)
            assert np.array_equal(N_(out), oracle_mod.group_gather_backward(g, idx, n)), (dtype, n, m, k, c)
            one = torch.empty_like(out)
            ext.group_gather_backward(T(g), ti, one, n)
            assert torch.equal(out, one)
        i1 = np.ascontiguousarray(idx[..., 0])
        l1 = ext.scatter_build(T(i1), n)
        g1 = rng.standard_normal((b, 5, m)).astype(dtype)
        out1 = torch.empty((b, 5, n), dtype=tdt, device=DEV)
        ext.scatter_apply(T(g1), l1, out1)
        assert np.array_equal(N_(out1), oracle_mod.gather_backward(g1, i1, n))
        with pytest.raises(RuntimeError):  # lists of another index tensor
            ext.scatter_apply(T(g1), lists, out1)
    for (b, c, s_, q) in ((2, 33, 1024, 4096), (1, 4, 9000, 130), (3, 20, 256, 1001)):
        idx = rng.integers(0, s_, (b, q, 3))
        w = rng.random((b, q, 3)).astype(dtype)
        g = rng.standard_normal((b, c, q)).astype(dtype)
        tw = T(w)
        lists = ext.scatter_build(T(idx), s_, tw)
        out = torch.full((b, c, s_), 3.0, dtype=tdt, device=DEV)
        ext.scatter_apply(T(g), lists, out)
        assert np.array_equal(N_(out), oracle_mod.three_interpolate_backward(g, idx, w, s_)), (dtype, s_, q)
    # straight through the C ABI: a missing or short workspace is refused, not allocated behind the caller's back
    lib = _lib.lib()
    ti = T(rng.integers(0, 64, (1, 100)))
    rc = lib.v3d_scatter_build(ti.data_ptr(), None, 0, 4, 1, 64, 100, None, 0, torch.cuda.current_device(), None)
    assert rc != 0 and lib.v3d_error_name(rc).decode() == "V3D_ERR_WORKSPACE"
    small = torch.empty((256,), dtype=torch.uint8, device=DEV)
    rc = lib.v3d_scatter_build(ti.data_ptr(), None, 0, 4, 1, 64, 100, small.data_ptr(), 16, torch.cuda.current_device(), None)
    assert rc != 0 and lib.v3d_error_name(rc).decode() == "V3D_ERR_WORKSPACE"


def test_autograd_wrappers_share_one_list_build_per_index_tensor(ext, monkeypatch):
    """ops.group_gather / ops.gather / ops.three_interpolate backward: the lists are cached on the index tensor -- the xyz
    and the feature gather of one set-abstraction layer (layers/pointnet2.py:44-46) build them once -- and an in-place edit
    of the indices rebuilds them; gradients equal torch's own autograd of the same gathers."""
    from vision3d_amd import ops

    builds = []
    real = ext.scatter_build
    monkeypatch.setattr(ext, "scatter_build", lambda *a, **k: (builds.append(1), real(*a, **k))[1])
    g = torch.Generator().manual_seed(3)
    b, n, m, k = 2, 600, 128, 8
    idx = torch.randint(0, n, (b, m, k), generator=g).to(DEV)
    xyz = torch.randn((b, 3, n), generator=g).to(DEV).requires_grad_(True)
    feat = torch.randn((b, 16, n), generator=g).to(DEV).requires_grad_(True)
    (ops.group_gather(xyz, idx).sum() * 2 + (ops.group_gather(feat, idx) ** 2).sum()).backward()
    assert len(builds) == 1
    rx = xyz.detach().clone().requires_grad_(True)
    rf = feat.detach().clone().requires_grad_(True)
    flat = idx.view(b, 1, -1)
    (torch.gather(rx, 2, flat.expand(b, 3, -1)).sum() * 2 + (torch.gather(rf, 2, flat.expand(b, 16, -1)) ** 2).sum()).backward()
    torch.testing.assert_close(xyz.grad, rx.grad, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(feat.grad, rf.grad, rtol=1e-5, atol=1e-4)
    idx[0, 0, 0] = (idx[0, 0, 0] + 1) % n  # in-place edit: the cached lists are stale
    feat.grad = None
    ops.group_gather(feat, idx).sum().backward()
    assert len(builds) == 2
    want = torch.zeros((b, 16, n), device=DEV).scatter_add_(2, idx.view(b, 1, -1).expand(b, 16, -1),
                                                            torch.ones((b, 16, m * k), device=DEV))
    assert torch.equal(feat.grad, want)
    # three_interpolate: lists keyed on the weights too
    q = 500
    i3 = torch.randint(0, m, (b, q, 3), generator=g).to(DEV)
    w = torch.rand((b, q, 3), generator=g).to(DEV)
    sf = torch.randn((b, 8, m), generator=g).to(DEV).requires_grad_(True)
    ops.three_interpolate(sf, i3, w).sum().backward()
    ops.three_interpolate(sf, i3, w).sum().backward()
    assert len(builds) == 3
    w2 = w.clone()
    ops.three_interpolate(sf, i3, w2).sum().backward()
    assert len(builds) == 4


def test_backward_single_source_and_full_c2_interp(ext, oracle_mod):
    """Adversarial index tensors (every target on one source: one segment per chunk, carried across chunks) and the
    C2 three_interpolate backward at full size against the oracle."""
    rng = np.random.default_rng(25)
    b, c, n, m, k = 2, 5, 300, 700, 9
    idx = np.full((b, m, k), 123, dtype=np.int64)
    g = rng.standard_normal((b, c, m, k)).astype(np.float32)
    out = torch.empty((b, c, n), device=DEV)
    ext.group_gather_backward(T(g), T(idx), out, n)
    assert np.array_equal(N_(out), oracle_mod.group_gather_backward(g, idx, n))
    b, c, s, q = 32, 128, 1024, 4096
    idx = rng.integers(0, s, (b, q, 3))
    w = rng.random((b, q, 3)).astype(np.float32)
    g = rng.standard_normal((b, c, q)).astype(np.float32)
    outb = torch.empty((b, c, s), device=DEV)
    ext.three_interpolate_backward(T(g), T(idx), T(w), outb, s)
    assert np.array_equal(N_(outb), oracle_mod.three_interpolate_backward(g, idx, w, s))


def test_backward_is_linear_and_reproducible(ext):
    """Size-independent properties at C2 size: scatter(a+b) == scatter(a)+scatter(b) up to rounding, identical
    bits on repeated runs (the reference's atomics are not reproducible), and sum conservation."""
    g = torch.Generator(device="cpu").manual_seed(0)
    b, c, n, m, k = 8, 32, 4096, 1024, 32
    idx = torch.randint(0, n, (b, m, k), generator=g).to(DEV)
    ga = torch.randn((b, c, m, k), generator=g).to(DEV)
    outs = []
    for _ in range(3):
        o = torch.empty((b, c, n), device=DEV)
        ext.group_gather_backward(ga, idx, o, n)
        outs.append(o)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    ref = torch.zeros((b, c, n), device=DEV).scatter_add_(2, idx.view(b, 1, -1).expand(b, c, -1), ga.view(b, c, -1))
    torch.testing.assert_close(outs[0], ref, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(outs[0].sum(-1), ga.sum((-1, -2)), rtol=1e-4, atol=1e-3)


# ------------------------------------------------------------------------------------------------------
# ops wrappers (reference signatures) incl. autograd
# ------------------------------------------------------------------------------------------------------
def test_ops_wrappers_contract(ext, oracle_mod):
    from vision3d_amd import ops

    rng = np.random.default_rng(31)
    pts = unit_sphere_cloud(rng, 2, 512)
    feats = rng.standard_normal((2, 6, 512)).astype(np.float32)
    tp, tf = T(pts), T(feats)
    idx = ops.furthest_point_sample(tp, 64, gather_points=False)
    assert np.array_equal(N_(idx), oracle_mod.furthest_point_sample(pts, 64))
    res = ops.furthest_point_sample(tp.transpose(1, 2), 64, tf, transposed=True)  # (B,3,N) in -> (B,3,M) out
    assert isinstance(res, tuple) and len(res) == 2  # quirk Q1: tuple, indices not returned
    assert res[0].shape == (2, 3, 64) and res[1].shape == (2, 6, 64)
    assert torch.equal(res[0], torch.gather(tp.transpose(1, 2), 2, idx.unsqueeze(1).expand(-1, 3, -1)))
    res2 = ops.furthest_point_sample(tp, 64)
    assert res2[0].shape == (2, 64, 3)
    q = res2[0]
    bq = ops.ball_query(q, tp, 16, 0.25)
    assert np.array_equal(N_(bq), oracle_mod.ball_query(N_(q), pts, 16, 0.25))
    bq_t = ops.ball_query(q.transpose(1, 2), tp.transpose(1, 2), 16, 0.25, transposed=True)
    assert torch.equal(bq, bq_t)
    ki = ops.knn_points(q, tp, 8)
    kd, ki2 = ops.knn_points(q, tp, 8, return_distance=True)
    assert torch.equal(ki, ki2) and kd.shape == (2, 64, 8)
    d3, i3 = ops.three_nn(tp, q)
    wd, wi = oracle_mod.three_nn(pts, N_(q))
    assert np.array_equal(N_(i3), wi) and np.array_equal(N_(d3), wd)

    # autograd: group_gather / gather / three_interpolate against torch's own indexing
    x = tf.clone().requires_grad_(True)
    y = ops.group_gather(x, bq)
    ref = torch.gather(x.unsqueeze(2).expand(-1, -1, 64, -1), 3, bq.unsqueeze(1).expand(-1, 6, -1, -1))
    assert torch.equal(y, ref)
    gy = torch.randn_like(y)
    (gx,) = torch.autograd.grad(y, x, gy)
    (gref,) = torch.autograd.grad(ref, x, gy)
    torch.testing.assert_close(gx, gref, rtol=1e-5, atol=1e-5)
    yt = ops.group_gather(x.transpose(1, 2), bq, True)  # transposed: (B,N,C) -> (B,M,K,C)
    assert torch.equal(yt, y.permute(0, 2, 3, 1))
    w = 1.0 / (d3 + 1e-5)
    w = w / w.sum(-1, keepdim=True)
    sf = T(rng.standard_normal((2, 6, 64)).astype(np.float32)).requires_grad_(True)
    out = ops.three_interpolate(sf, i3, w)
    ref = (torch.gather(sf.unsqueeze(2).expand(-1, -1, 512, -1), 3, i3.unsqueeze(1).expand(-1, 6, -1, -1)) *
           w.unsqueeze(1)).sum(-1)
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-6)
    go = torch.randn_like(out)
    (g1,) = torch.autograd.grad(out, sf, go)
    (g2,) = torch.autograd.grad(ref, sf, go)
    torch.testing.assert_close(g1, g2, rtol=1e-5, atol=1e-5)
    z = ops.gather(x, idx)
    assert torch.equal(z, torch.gather(x, 2, idx.unsqueeze(1).expand(-1, 6, -1)))


# ------------------------------------------------------------------------------------------------------
# pairwise_distance / feature kNN / merge
# ------------------------------------------------------------------------------------------------------
def _pd_tol(x, y):
    return 1e-5 * ((x ** 2).sum(-1)[..., :, None] + (y ** 2).sum(-1)[..., None, :])


@pytest.mark.parametrize("normalized", [False, True])
@pytest.mark.parametrize("squared", [False, True])
@pytest.mark.parametrize("strict", [False, True])
@pytest.mark.parametrize("transposed", [False, True])
def test_pairwise_distance_vs_reference_fixture(ext, golden, normalized, squared, strict, transposed):
    from vision3d_amd import ops

    g = golden("ref_pairwise_distance.npz")
    x, y = (g["xn"], g["yn"]) if normalized else (g["x"], g["y"])
    ref = g[f"d_n{int(normalized)}_s{int(squared)}_t{int(strict)}_T{int(transposed)}"]
    a, b = (np.swapaxes(x, -1, -2), np.swapaxes(y, -1, -2)) if transposed else (x, y)
    got = N_(ops.pairwise_distance(T(a), T(b), normalized=normalized, transposed=transposed, squared=squared,
                                   strict=strict))
    if not squared:
        got, ref = got ** 2, ref ** 2
    assert got.shape == ref.shape
    assert np.all(np.abs(got - ref) <= _pd_tol(x, y) + 1e-7)


@pytest.mark.parametrize("shape", [(1, 1, 1, 1), (2, 130, 257, 64), (1, 128, 128, 32), (3, 100, 50, 7), (2, 300, 129, 256),
                                   (1, 1024, 1024, 64)])
def test_pairwise_distance_vs_oracle(ext, oracle_mod, shape):
    from vision3d_amd import ops

    b, n, m, c = shape
    rng = np.random.default_rng(n + m + c)
    x = rng.standard_normal((b, n, c)).astype(np.float32)
    y = rng.standard_normal((b, m, c)).astype(np.float32)
    oracle_mod.set_threads(oracle_mod.max_threads())
    try:
        want = oracle_mod.pairwise_distance(x, y)
    finally:
        oracle_mod.set_threads(1)
    got = N_(ops.pairwise_distance(T(x), T(y)))
    assert np.all(np.abs(got - want) <= _pd_tol(x, y) + 1e-7)
    assert np.all(got >= 0)
    gs = N_(ops.pairwise_distance(T(x), T(y), strict=True))
    assert np.all(np.abs(gs - want) <= _pd_tol(x, y) + 1e-7)


@pytest.mark.parametrize("shape", [(2, 200, 260, 64), (1, 128, 132, 96), (3, 385, 512, 32), (1, 1000, 1028, 160),
                                   (5, 129, 128, 64), (1, 2176, 1284, 64)])
@pytest.mark.parametrize("squared", [True, False])
@pytest.mark.parametrize("normalized", [False, True])
def test_pairwise_distance_tile_sweep_edges(ext, shape, squared, normalized):
    """The persistent tile sweep (N, M >= 128, C % 32 == 0, M % 4 == 0): last tiles moved back to end at the matrix edge
    (cells shared with the neighbouring tile are computed twice, identically), several tiles per workgroup, both
    epilogue forms.  Checked against the float64 evaluation of pairwise_distance.py:39-56, tolerance 1e-5 of the
    operand scale as everywhere."""
    from vision3d_amd import ops

    b, n, m, c = shape
    rng = np.random.default_rng(7 * n + m + c)
    x = rng.standard_normal((b, n, c)).astype(np.float32)
    y = rng.standard_normal((b, m, c)).astype(np.float32)
    if normalized:
        x /= np.linalg.norm(x, axis=-1, keepdims=True)
        y /= np.linalg.norm(y, axis=-1, keepdims=True)
    xd, yd = x.astype(np.float64), y.astype(np.float64)
    xy = np.einsum("bnc,bmc->bnm", xd, yd)
    want = (2.0 - 2.0 * xy) if normalized else ((xd ** 2).sum(-1)[:, :, None] - 2.0 * xy + (yd ** 2).sum(-1)[:, None, :])
    want = np.maximum(want, 0.0)
    got = N_(ops.pairwise_distance(T(x), T(y), normalized=normalized, squared=squared)).astype(np.float64)
    if not squared:
        got = got ** 2 - 1e-8  # sqrt(d + eps), eps = 1e-8
    assert got.shape == want.shape
    assert np.all(np.abs(got - want) <= _pd_tol(x, y) + 2e-7)
    # a second call reproduces the first bit for bit (overlapping tiles write the same values, no order dependence)
    again = N_(ops.pairwise_distance(T(x), T(y), normalized=normalized, squared=squared)).astype(np.float64)
    if not squared:
        again = again ** 2 - 1e-8
    assert np.array_equal(got, again)


def test_pairwise_distance_backward(ext):
    from vision3d_amd import ops

    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 40, 16, generator=g).to(DEV).requires_grad_(True)
    y = torch.randn(2, 50, 16, generator=g).to(DEV).requires_grad_(True)
    go = torch.randn(2, 40, 50, generator=g).to(DEV)
    for kw in ({}, {"squared": False}, {"normalized": True}):
        out = ops.pairwise_distance(x, y, **kw)
        gx, gy = torch.autograd.grad(out, (x, y), go)
        xy = torch.matmul(x, y.transpose(-1, -2))
        ref = (2.0 - 2.0 * xy) if kw.get("normalized") else ((x ** 2).sum(-1).unsqueeze(-1) - 2 * xy +
                                                             (y ** 2).sum(-1).unsqueeze(-2))
        ref = ref.clamp(min=0.0)
        if kw.get("squared") is False:
            ref = torch.sqrt(ref + 1e-8)
        rx, ry = torch.autograd.grad(ref, (x, y), go)
        torch.testing.assert_close(gx, rx, rtol=1e-3, atol=1e-3)
        torch.testing.assert_close(gy, ry, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("shape", [(2, 100, 300, 64, 20), (1, 257, 1000, 16, 1), (1, 64, 5, 8, 9), (2, 128, 128, 256, 3),
                                   (32, 64, 1024, 64, 20)])
def test_knn_feat_vs_oracle(ext, oracle_mod, shape):
    """Feature-space kNN ('parity unpinned' third-party semantics): same neighbour SETS as the float64 oracle except
    where two candidates are closer than the fp32 distance resolution; distances to 1e-5 of the operand scale."""
    b, n, m, c, k = shape
    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal((b, n, c)).astype(np.float32)
    y = rng.standard_normal((b, m, c)).astype(np.float32)
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
    assert np.all(np.abs(gd[..., :kk] - wd[..., :kk]) <= 1e-5 * scale + 1e-6)
    assert np.all(np.diff(gd[..., :kk], axis=-1) >= 0)
    mismatch = gi[..., :kk] != wi[..., :kk]
    # any index mismatch must be a near-tie: the distance at that rank agrees although the neighbour differs
    assert mismatch.mean() < 1e-3
    assert np.all(np.abs(gd[..., :kk][mismatch] - wd[..., :kk][mismatch]) <= 1e-5 * np.broadcast_to(scale, gd[..., :kk].shape)[mismatch] + 1e-6)


def test_knn_wrapper_and_pack_mode(ext, oracle_mod):
    from vision3d_amd import ops

    rng = np.random.default_rng(41)
    x = rng.standard_normal((2, 200, 32)).astype(np.float32)
    wd, wi = oracle_mod.knn_feat(x, x, 10)
    d, i = ops.knn(T(x), T(x), 10, return_distance=True)
    assert (N_(i) != wi).mean() < 1e-3
    assert np.all(N_(i)[..., 0] == np.arange(200))  # self first
    # pack mode on 3-D points: per-cloud search, indices local to the cloud
    lengths = torch.tensor([150, 90, 200], device=DEV)
    pts = rng.random((440, 3), dtype=np.float32)
    ki = ops.knn_pack_mode(T(pts), T(pts), lengths, lengths, 5)
    off = 0
    for L in (150, 90, 200):
        _, wi = oracle_mod.knn_points(pts[None, off:off + L], pts[None, off:off + L], 5)
        assert np.array_equal(N_(ki)[off:off + L], wi[0])
        off += L
    ri = ops.radius_search_pack_mode(T(pts), T(pts), lengths, lengths, 0.15, 12)
    rc = ops.radius_count_pack_mode(T(pts), T(pts), lengths, lengths, 0.15)
    off = 0
    for L in (150, 90, 200):
        p = pts[off:off + L].astype(np.float64)
        dd = np.sqrt(((p[:, None] - p[None]) ** 2).sum(-1))
        cnt = (dd < 0.15).sum(1)
        got_cnt = N_(rc)[off:off + L]
        assert np.abs(got_cnt - cnt).max() <= 1 and (got_cnt != cnt).mean() < 0.01  # fp32 vs fp64 at the boundary
        rows = N_(ri)[off:off + L]
        valid = rows < 440
        assert np.all(valid.sum(1) == np.minimum(got_cnt, 12)) or np.abs(valid.sum(1) - np.minimum(cnt, 12)).max() <= 1
        assert np.all(rows[valid] >= off) and np.all(rows[valid] < off + L)
        off += L


def test_topk_merge_equals_single_shot(ext, oracle_mod):
    """Reference-set sharding: per-shard top-k + merge == top-k over the whole set."""
    rng = np.random.default_rng(51)
    x = rng.standard_normal((1, 300, 48)).astype(np.float32)
    y = rng.standard_normal((1, 1000, 48)).astype(np.float32)
    k, G = 3, 4
    full_d = torch.empty((1, 300, k), device=DEV)
    full_i = torch.empty((1, 300, k), dtype=torch.long, device=DEV)
    ext.knn_feat(T(x), T(y), full_d, full_i, k)
    pd = torch.empty((G, 300, k), device=DEV)
    pi = torch.empty((G, 300, k), dtype=torch.long, device=DEV)
    for g in range(G):
        ys = np.ascontiguousarray(y[:, g * 250:(g + 1) * 250])
        ext.knn_feat(T(x), T(ys), pd[g:g + 1], pi[g:g + 1], k, index_offset=g * 250)
    od = torch.empty((300, k), device=DEV)
    oi = torch.empty((300, k), dtype=torch.long, device=DEV)
    ext.topk_merge(pd, pi, od, oi)
    assert torch.equal(oi, full_i[0]) and torch.equal(od, full_d[0])


# ------------------------------------------------------------------------------------------------------
# "next" rows: native pack-mode radius search
# ------------------------------------------------------------------------------------------------------
def _pack_clouds(rng, q_lens, s_lens, lattice=None, scale=1.0):
    def cloud(n):
        if lattice:
            return (rng.integers(0, lattice, (n, 3)) * (scale / lattice)).astype(np.float32)
        return (rng.random((n, 3)) * scale).astype(np.float32)

    q = np.concatenate([cloud(n) for n in q_lens] + [np.zeros((0, 3), np.float32)])
    s = np.concatenate([cloud(n) for n in s_lens] + [np.zeros((0, 3), np.float32)])
    return q, s, np.asarray(q_lens, np.int64), np.asarray(s_lens, np.int64)


@pytest.mark.parametrize("case", [
    dict(q=(300, 0, 200), s=(500, 10, 400), r=0.15, k=20),          # empty query cloud, tiny support cloud
    dict(q=(64, 1, 1000), s=(0, 7, 3000), r=0.1, k=38),             # empty support cloud -> all sentinel
    dict(q=(500,), s=(500,), r=0.3, k=100, lattice=8),              # lattice: massive (d2, index) ties, R = 2 list
    dict(q=(400, 400), s=(2000, 1500), r=0.05, k=5),                # hits << limit and hits >> limit mix
    dict(q=(300,), s=(4000,), r=2.0, k=64),                         # radius covers the whole cloud: one coarse cell
    dict(q=(200, 200), s=(3000, 3000), r=0.0625, k=38, scale=3.0),  # config-4-like proportions
    dict(q=(100,), s=(100,), r=0.0, k=4, lattice=4),                # zero radius: only coincident points
])
def test_radius_search_pack_native(ext, oracle_mod, case):
    rng = np.random.default_rng(61)
    q, s, ql, sl = _pack_clouds(rng, case["q"], case["s"], case.get("lattice"), case.get("scale", 1.0))
    want = oracle_mod.radius_search_pack(q, s, ql, sl, case["r"], case["k"])
    out = torch.full((q.shape[0], case["k"]), -7, dtype=torch.long, device=DEV)
    ext.radius_search_pack(T(q), T(s), T(ql), T(sl), out, case["r"], case["k"])
    assert np.array_equal(N_(out), want)


@pytest.mark.parametrize("case", [
    dict(q=(300, 0, 200), s=(500, 10, 400), r=0.15),          # empty query cloud, tiny support cloud
    dict(q=(64, 1, 1000), s=(0, 7, 3000), r=0.1),             # empty support cloud -> zero
    dict(q=(500,), s=(500,), r=0.25, lattice=8),              # lattice: pairs exactly at the radius (strict: out)
    dict(q=(400, 400), s=(2000, 1500), r=0.05),
    dict(q=(300,), s=(4000,), r=2.0),                         # radius covers the whole cloud: one coarse cell
    dict(q=(200, 200), s=(3000, 3000), r=0.0625, scale=3.0),  # config-4-like proportions
    dict(q=(100,), s=(100,), r=0.0, lattice=4),               # zero radius: nothing is strictly inside
])
def test_radius_count_pack_native(ext, oracle_mod, case):
    """v3d_radius_count_pack_f32 (cell list) == the NumPy restatement of radius_search.py:74-107, bit for bit, and ==
    the padded brute-force route the op used before."""
    from vision3d_amd import ops
    from vision3d_amd.ops.radius_search import padded_radius_count_pack_mode

    rng = np.random.default_rng(63)
    q, s, ql, sl = _pack_clouds(rng, case["q"], case["s"], case.get("lattice"), case.get("scale", 1.0))
    q[:10] += 0.7 * case.get("scale", 1.0)  # some queries outside the support box
    want = oracle_mod.radius_count_pack(q, s, ql, sl, case["r"])
    out = torch.full((q.shape[0],), -7, dtype=torch.long, device=DEV)
    ext.radius_count_pack(T(q), T(s), T(ql), T(sl), out, case["r"])
    assert np.array_equal(N_(out), want)
    got = ops.radius_count_pack_mode(T(q), T(s), T(ql), T(sl), case["r"])
    assert got.shape == (q.shape[0],) and np.array_equal(N_(got), want)
    if min(case["s"]) > 0 and min(case["q"]) > 0:
        assert torch.equal(padded_radius_count_pack_mode(T(q), T(s), T(ql), T(sl), case["r"]), got)
    if case["r"] > 0:  # count == number of hits of the search when the limit is not reached
        k = int(min(128, max(1, want.max())))
        idx = torch.empty((q.shape[0], k), dtype=torch.long, device=DEV)
        ext.radius_search_pack(T(q), T(s), T(ql), T(sl), idx, case["r"], k)
        hits = (N_(idx) < s.shape[0]).sum(1)
        assert np.all(hits >= np.minimum(want, k))  # the search keeps d == radius, the count does not


def test_radius_count_pack_nonfinite_and_refusals(ext, oracle_mod):
    rng = np.random.default_rng(64)
    q, s, ql, sl = _pack_clouds(rng, (200, 100), (700, 300))
    s[5] = np.nan
    s[17, 1] = np.inf
    q[3, 0] = np.nan
    q[9] = -np.inf
    want = oracle_mod.radius_count_pack(q, s, ql, sl, 0.2)
    out = torch.empty((300,), dtype=torch.long, device=DEV)
    ext.radius_count_pack(T(q), T(s), T(ql), T(sl), out, 0.2)
    assert np.array_equal(N_(out), want)
    with pytest.raises(RuntimeError):
        ext.radius_count_pack(T(q), T(s), T(ql), T(sl), out, -1.0)
    with pytest.raises(RuntimeError):
        ext.radius_count_pack(T(q), T(s), T(ql), T(sl), out[:10], 0.2)


def test_radius_pack_nonfinite_bounding_box_keeps_to_its_cells(ext, oracle_mod):
    """A cloud whose supports have no finite bounding box (all NaN; or one support at +inf on every axis) must not disturb
    the cell lists of its neighbours in the pack, nor leave the workspace: such a cloud falls back to a one-cell grid."""
    rng = np.random.default_rng(66)
    q, s, ql, sl = _pack_clouds(rng, (150, 120, 130), (600, 500, 700))
    nan_cloud = s.copy()
    nan_cloud[600:1100] = np.nan  # the middle cloud: every support NaN -> bbox stays (+inf, -inf)
    want = oracle_mod.radius_count_pack(q, nan_cloud, ql, sl, 0.15)
    out = torch.empty((400,), dtype=torch.long, device=DEV)
    ext.radius_count_pack(T(q), T(nan_cloud), T(ql), T(sl), out, 0.15)
    assert np.array_equal(N_(out), want) and np.all(want[150:270] == 0) and want[:150].sum() > 0
    for radius in (0.15, 1e-4):  # a tiny radius is what used to ask for ~1e9 cells
        inf_cloud = s.copy()
        inf_cloud[600 + 7] = np.inf  # one support at (inf, inf, inf) in the middle cloud, next to normal clouds
        want = oracle_mod.radius_count_pack(q, inf_cloud, ql, sl, radius)
        ext.radius_count_pack(T(q), T(inf_cloud), T(ql), T(sl), out, radius)
        assert np.array_equal(N_(out), want)
        idx = torch.empty((400, 20), dtype=torch.long, device=DEV)
        ext.radius_search_pack(T(q), T(inf_cloud), T(ql), T(sl), idx, radius, 20)
        assert np.array_equal(N_(idx), oracle_mod.radius_search_pack(q, inf_cloud, ql, sl, radius, 20))


def test_radius_pack_c_abi_null_workspace_and_bad_workspace(ext, oracle_mod):
    """Straight through the C ABI: workspace NULL (the library allocates stream-ordered scratch) gives the same tables
    as a caller-provided workspace; a workspace that is too small is refused with V3D_ERR_WORKSPACE."""
    from vision3d_amd import _lib

    lib = _lib.lib()
    rng = np.random.default_rng(65)
    q, s, ql, sl = _pack_clouds(rng, (300, 150), (2000, 900))
    tq, ts, tql, tsl = T(q), T(s), T(ql), T(sl)
    limit, r = 40, 0.12
    out = torch.full((450, limit), -7, dtype=torch.long, device=DEV)
    cnt = torch.full((450,), -7, dtype=torch.long, device=DEV)
    dev = torch.cuda.current_device()
    stream = torch.cuda.current_stream().cuda_stream
    rc = lib.v3d_radius_search_pack_f32(tq.data_ptr(), ts.data_ptr(), tql.data_ptr(), tsl.data_ptr(), out.data_ptr(), 2,
                                        450, 2900, r, limit, None, 0, dev, stream)
    assert rc == 0, lib.v3d_last_error_string()
    rc = lib.v3d_radius_count_pack_f32(tq.data_ptr(), ts.data_ptr(), tql.data_ptr(), tsl.data_ptr(), cnt.data_ptr(), 2,
                                       450, 2900, r, None, 0, dev, stream)
    assert rc == 0, lib.v3d_last_error_string()
    torch.cuda.synchronize()
    assert np.array_equal(N_(out), oracle_mod.radius_search_pack(q, s, ql, sl, r, limit))
    assert np.array_equal(N_(cnt), oracle_mod.radius_count_pack(q, s, ql, sl, r))
    small = torch.empty((1024,), dtype=torch.uint8, device=DEV)
    rc = lib.v3d_radius_count_pack_f32(tq.data_ptr(), ts.data_ptr(), tql.data_ptr(), tsl.data_ptr(), cnt.data_ptr(), 2,
                                       450, 2900, r, small.data_ptr(), small.numel(), dev, stream)
    assert rc != 0 and lib.v3d_error_name(rc).decode() == "V3D_ERR_WORKSPACE"


def test_radius_search_pack_mode_op_matches_padded_route(ext, oracle_mod):
    """ops.radius_search_pack_mode (native) == the reference's formulation (kNN over padded batches + mask) on
    tie-free inputs, and queries outside the supports' bounding box are handled."""
    from vision3d_amd import ops
    from vision3d_amd.ops.radius_search import padded_radius_search_pack_mode

    rng = np.random.default_rng(62)
    q, s, ql, sl = _pack_clouds(rng, (257, 300), (900, 1100))
    q[:20] += 0.9  # outside the support box
    a = ops.radius_search_pack_mode(T(q), T(s), T(ql), T(sl), 0.2, 30)
    b = padded_radius_search_pack_mode(T(q), T(s), T(ql), T(sl), 0.2, 30)
    assert torch.equal(a, b)
    assert np.array_equal(N_(a), oracle_mod.radius_search_pack(q, s, ql, sl, 0.2, 30))


@pytest.mark.parametrize("case", [
    dict(lens=(700, 0, 300), vs=0.25, scale=2.0, shift=-0.5),
    dict(lens=(5000,), vs=0.05, scale=1.0, shift=0.0),
    dict(lens=(1, 1, 1), vs=0.1, scale=1.0, shift=0.0),
    dict(lens=(3000, 2000), vs=10.0, scale=1.0, shift=-3.0),        # every cloud collapses into <= 8 voxels: long runs
    dict(lens=(4000,), vs=0.03, scale=1.0, shift=100.0),            # far from the origin
])
def test_grid_subsample_pack_native(ext, oracle_mod, case):
    from vision3d_amd import ops

    rng = np.random.default_rng(71)
    n = sum(case["lens"])
    pts = (rng.random((n, 3)) * case["scale"] + case["shift"]).astype(np.float32)
    lengths = np.asarray(case["lens"], np.int64)
    want_p, want_l = oracle_mod.grid_subsample_pack(pts, lengths, case["vs"])
    got_p, got_l = ops.grid_subsample_pack_mode(T(pts), T(lengths), case["vs"])
    assert np.array_equal(N_(got_l), want_l)
    assert np.array_equal(N_(got_p), want_p)  # same order, same summation order: bit-identical


@pytest.mark.parametrize("tag", ["a", "b"])
def test_grid_subsample_pack_vs_reference_fixture(ext, golden, tag):
    """The kernel against the REFERENCE's own ops/grid_subsample.py (naive :13-46 and fast :49-99, run on the CPU by
    tests/golden/make_golden.py): same order, lengths and barycentres bit for bit."""
    from vision3d_amd import ops

    g = golden("ref_pack_glue.npz")
    got_p, got_l = ops.grid_subsample_pack_mode(T(g["gs_points"]), T(g["gs_lengths"]), float(g[f"gs_{tag}_voxel"]))
    for version in ("fast", "naive"):
        assert np.array_equal(N_(got_l), g[f"gs_{tag}_{version}_lengths"])
        assert np.array_equal(N_(got_p), g[f"gs_{tag}_{version}_points"])


def test_radius_search_pack_vs_reference_glue_fixture(ext, golden):
    """The native pack-mode search, and the padded route through ops.conversion + keops_knn, against the reference's
    ops/radius_search.py:44-71 run around a stand-in kNN (pins padding / offsets / mask / sentinel, not pykeops)."""
    from vision3d_amd import ops
    from vision3d_amd.ops import conversion as conv
    from vision3d_amd.ops.radius_search import padded_radius_search_pack_mode

    g = golden("ref_pack_glue.npz")
    args = (T(g["rs_q"]), T(g["rs_s"]), T(g["rs_q_lengths"]), T(g["rs_s_lengths"]), float(g["rs_radius"]), int(g["rs_limit"]))
    assert np.array_equal(N_(ops.radius_search_pack_mode(*args)), g["rs_table"])
    assert np.array_equal(N_(padded_radius_search_pack_mode(*args)), g["rs_table"])
    # the conversion mirrors on the device
    bt, bm = conv.pack_to_batch(T(g["cv_packed"]), T(g["cv_lengths"]), max_length=12, fill_value=1e10)
    assert np.array_equal(N_(bt), g["cv_wide_batch"]) and np.array_equal(N_(bm), g["cv_wide_masks"])
    pk, pl = conv.batch_to_pack(bt, bm)
    assert np.array_equal(N_(pk), g["cv_packed"]) and np.array_equal(N_(pl), g["cv_wide_relengths"])


def test_knn_pack_and_knn_vs_reference_glue_fixture(ext, golden):
    """ops.knn_pack_mode (native and padded route) and ops.knn on 3-D points (plain, dilated + remove_nearest) against the
    reference's ops/knn.py:110-145 / :30-107 run around a stand-in kNN (tests/golden/ref_pack_glue.npz; tie-free clouds)."""
    from vision3d_amd import ops
    from vision3d_amd.ops.knn import padded_knn_pack_mode

    g = golden("ref_pack_glue.npz")
    q, s, ql, sl = T(g["rs_q"]), T(g["rs_s"]), T(g["rs_q_lengths"]), T(g["rs_s_lengths"])
    for k in (1, 3, 16):
        for fn in (ops.knn_pack_mode, padded_knn_pack_mode):
            d, i = fn(q, s, ql, sl, k, return_distance=True)
            assert np.array_equal(N_(i), g[f"kp_k{k}_idx"]), (k, fn.__name__)
            np.testing.assert_allclose(N_(d), g[f"kp_k{k}_dist"], rtol=0, atol=2e-6)
    bq, bs = q[:120].reshape(2, 60, 3), s[:400].reshape(2, 200, 3)
    d, i = ops.knn(bq, bs, 5, return_distance=True)
    assert np.array_equal(N_(i), g["kb_idx"])
    np.testing.assert_allclose(N_(d), g["kb_dist"], rtol=0, atol=2e-6)
    d, i = ops.knn(bq, bs, 4, dilation=2, remove_nearest=True, return_distance=True)
    assert np.array_equal(N_(i), g["kb_dil_idx"])
    np.testing.assert_allclose(N_(d), g["kb_dil_dist"], rtol=0, atol=2e-6)


def test_c4_full_size_radius_search_count_and_grid_subsample(ext, oracle_mod):
    """BASELINE config 4 at full size: 16 clouds x 30,000 points in a 3 x 3 x 2.5 m box, radius 0.0625 with
    limit 38 and a denser radius 0.2 (more hits than the limit), voxel 0.025.  Radius search / count: oracle comparison on
    a 2,000-query subsample per cloud (the oracle is brute force; a query's row depends on that query and the supports
    only) plus properties of the whole output; grid subsampling: whole output against the oracle, bit for bit.
    Reference: ops/radius_search.py:44-107, ops/grid_subsample.py:49-99."""
    from vision3d_amd import ops

    rng = np.random.default_rng(91)
    nc, n, limit, sub = 16, 30000, 38, 2000
    pts = (rng.random((nc * n, 3)) * np.float32([3.0, 3.0, 2.5])).astype(np.float32)
    lengths = np.full((nc,), n, np.int64)
    tp, tl = T(pts), T(lengths)
    total = nc * n
    pick = np.concatenate([c * n + np.sort(rng.choice(n, sub, replace=False)) for c in range(nc)])
    q_sub, ql_sub = pts[pick], np.full((nc,), sub, np.int64)
    oracle_mod.set_threads(oracle_mod.max_threads())
    try:
        for radius in (0.0625, 0.2):
            idx = torch.empty((total, limit), dtype=torch.long, device=DEV)
            ext.radius_search_pack(tp, tp, tl, tl, idx, radius, limit)
            cnt = torch.empty((total,), dtype=torch.long, device=DEV)
            ext.radius_count_pack(tp, tp, tl, tl, cnt, radius)
            got, gcnt = N_(idx), N_(cnt)
            assert np.array_equal(got[pick], oracle_mod.radius_search_pack(q_sub, pts, ql_sub, lengths, radius, limit))
            # (the count oracle is single-threaded NumPy: every fourth query of the subsample)
            assert np.array_equal(gcnt[pick[::4]], oracle_mod.radius_count_pack(q_sub[::4], pts, ql_sub // 4, lengths, radius))
            # whole output: sentinel padding only at the tail of a row, indices inside the query's own cloud, the query
            # itself first (distance 0), as many hits as the strict count allows, ascending distance
            valid = got < total
            hits = valid.sum(1)
            assert np.array_equal(valid, np.arange(limit)[None, :] < hits[:, None])
            cloud = np.repeat(np.arange(nc), n)
            assert np.all((got // n == cloud[:, None]) | ~valid)
            assert np.array_equal(got[:, 0], np.arange(total))
            assert np.all(hits >= np.minimum(gcnt, limit))  # the search keeps d == radius, the count is strict
            assert np.all(hits[gcnt < limit] <= gcnt[gcnt < limit] + 1)
            rows = rng.choice(total, 20000, replace=False)
            nb = np.where(valid[rows], got[rows], rows[:, None])
            d = np.linalg.norm(pts[nb].astype(np.float64) - pts[rows][:, None, :].astype(np.float64), axis=2)
            assert np.all(d[valid[rows]] <= radius * (1 + 1e-6))
            d[~valid[rows]] = np.inf
            assert np.all(np.diff(d, axis=1)[valid[rows][:, 1:]] >= -1e-7)
    finally:
        oracle_mod.set_threads(1)
    want_p, want_l = oracle_mod.grid_subsample_pack(pts, lengths, 0.025)
    got_p, got_l = ops.grid_subsample_pack_mode(tp, tl, 0.025)
    assert np.array_equal(N_(got_l), want_l) and np.array_equal(N_(got_p), want_p)
    assert 0.3 * total < want_p.shape[0] < total  # the voxel size does subsample


def test_feature_knn_full_config_sizes(ext, oracle_mod):
    """C3 (DGCNN EdgeConv: 32 x 1024 x 64, k = 20, self included) against the oracle at full size, and a C5-shaped
    reference-set shard (normalised 256-d features, 100k supports) through size-independent properties: the planted
    nearest neighbour is found, rows are sorted, and the distances equal the exact difference form."""
    rng = np.random.default_rng(81)
    oracle_mod.set_threads(oracle_mod.max_threads())
    try:
        x = rng.standard_normal((32, 1024, 64)).astype(np.float32)
        wd, wi = oracle_mod.knn_feat(x, x, 20)
    finally:
        oracle_mod.set_threads(1)
    d = torch.empty((32, 1024, 20), device=DEV)
    idx = torch.empty((32, 1024, 20), dtype=torch.long, device=DEV)
    ext.knn_feat(T(x), T(x), d, idx, 20)
    gd, gi = N_(d), N_(idx)
    scale = 2.0 * (x ** 2).sum(-1).max()
    assert np.all(gi[..., 0] == np.arange(1024)) and np.all(np.abs(gd - wd) <= 1e-5 * scale)
    assert (gi != wi).mean() < 1e-4  # only exact near-ties may swap
    # C5 shard: y = 100k unit vectors, x = 2048 of them plus noise -> neighbour 0 is the planted one
    n, m, c, k = 2048, 100000, 256, 3
    g = torch.Generator(device="cpu").manual_seed(5)
    y = torch.nn.functional.normalize(torch.randn((1, m, c), generator=g), dim=-1)
    planted = torch.randint(0, m, (n,), generator=g)
    xq = torch.nn.functional.normalize(y[0, planted] + 0.01 * torch.randn((n, c), generator=g), dim=-1)[None]
    yd, xd = y.to(DEV).contiguous(), xq.to(DEV).contiguous()
    d = torch.empty((1, n, k), device=DEV)
    idx = torch.empty((1, n, k), dtype=torch.long, device=DEV)
    ext.knn_feat(xd, yd, d, idx, k)
    assert torch.equal(idx[0, :, 0].cpu(), planted)
    assert torch.all(d[..., 1:] >= d[..., :-1])
    exact = (xd[0, :, None, :] - yd[0][idx[0]]).pow(2).sum(-1).sqrt()  # (n, k) exact difference form
    torch.testing.assert_close(d[0], exact, rtol=0, atol=2e-5)


@pytest.mark.parametrize("S", [3, 63, 64, 100, 383, 384, 900, 1024, 1600, 2048, 2049])
def test_three_nn_cell_list_sizes_and_degenerate_clouds(ext, oracle_mod, S):
    """three_nn switches to the in-LDS cell list for 64 <= S <= 2048 (fp32): every size class, clustered / flat /
    duplicated supports, queries far outside the supports' bounding box, infinite and NaN coordinates -- all equal to
    the oracle's sequential scan, ties included."""
    rng = np.random.default_rng(S)
    B, Qn = 2, 300
    cases = {
        "uniform": rng.random((B, S, 3), dtype=np.float32),
        "flat": np.concatenate([rng.random((B, S, 2), dtype=np.float32), np.full((B, S, 1), 0.5, np.float32)], axis=2),
        "clustered": (rng.random((B, S, 3), dtype=np.float32) * 1e-3 + (np.arange(S)[None, :, None] % 3)).astype(np.float32),
        "lattice": (rng.integers(0, 4, size=(B, S, 3)).astype(np.float32) * 0.25),  # massive ties and duplicates
        "one_point": np.full((B, S, 3), 0.25, dtype=np.float32),
        # far from the coordinate origin: coordinates quantised at 1/16, face positions must not be rounded at 1e6
        "offset": (rng.random((B, S, 3), dtype=np.float32) * 40 + np.float32(1e6)).astype(np.float32),
        "anisotropic": (rng.random((B, S, 3), dtype=np.float32) * np.float32([1000.0, 1.0, 1e-3])).astype(np.float32),
    }
    for name, s in cases.items():
        centre, span = s.mean(axis=1, keepdims=True), (s.max(axis=1, keepdims=True) - s.min(axis=1, keepdims=True))
        q = np.concatenate([s[:, : min(S, 100)], (centre + (rng.random((B, Qn - 100, 3), dtype=np.float32) * 3 - 1.5) * span),
                            np.full((B, 100 - min(S, 100), 3), 7.0, np.float32)], axis=1).astype(np.float32)
        want_d, want_i = oracle_mod.three_nn(q, s)
        d = torch.empty((B, q.shape[1], 3), device=DEV)
        i = torch.zeros((B, q.shape[1], 3), dtype=torch.long, device=DEV)
        ext.three_nn(T(q), T(s), d, i)
        assert np.array_equal(N_(i), want_i), (name, S)
        assert np.array_equal(N_(d), want_d), (name, S)
    if S >= 64:
        s = rng.random((1, S, 3), dtype=np.float32)
        s[0, 5] = np.inf
        s[0, 9, 1] = np.nan
        q = rng.random((1, 50, 3), dtype=np.float32)
        q[0, 3, 0] = np.nan
        q[0, 4, 2] = np.inf
        want_d, want_i = oracle_mod.three_nn(q, s)
        d = torch.empty((1, 50, 3), device=DEV)
        i = torch.zeros((1, 50, 3), dtype=torch.long, device=DEV)
        ext.three_nn(T(q), T(s), d, i)
        assert np.array_equal(N_(i), want_i), ("nonfinite", S)
        assert np.array_equal(N_(d), want_d, equal_nan=True), ("nonfinite", S)
