"""Known-answer tests of the CPU oracle: hand-worked cases for every rule the kernels must reproduce.

They pin the RESTATEMENT against the reference source text (cited per test); the reference ships no vectors
of its own for these kernels.
"""
import numpy as np


def test_fps_block_matches_opt_n_threads(oracle_mod):
    # furthest_point_sample_cuda.cuh:9-12: largest power of two <= N, clamped to [1, 512]
    expect = {1: 1, 2: 2, 3: 2, 63: 32, 64: 64, 100: 64, 511: 256, 512: 512, 1000: 512, 1024: 512, 4096: 512}
    for n, b in expect.items():
        assert oracle_mod.fps_block(n) == b, n


def test_fps_simple_line(oracle_mod):
    # points on a line: start at index 0, then the farthest, then the farthest from both ...
    xs = np.array([0.0, 1.0, 2.0, 3.0, 10.0, 4.0], dtype=np.float32)
    pts = np.stack([xs, np.zeros_like(xs), np.zeros_like(xs)], -1)[None]
    idx = oracle_mod.furthest_point_sample(pts, 4)
    # 0 -> 10 (idx 4) -> min-dist: [0,1,2,3,0,4] -> 4.0 (idx 5) -> [0,1,2,1,0,0]... -> 2.0 (idx 2)
    assert idx.tolist() == [[0, 4, 5, 2]]


def test_fps_tie_order_is_bit_reversed_thread_order(oracle_mod):
    # 4 points all at distance 1 from point 0 (N=5 -> reference block 4: threads own p = tid, tid+4).
    # Thread 0 owns p0 (d=0) and p4 (d=1), threads 1..3 own p1..p3 (d=1).  The shared-memory tree
    # (cuh:76-83) keeps the lower slot on ties: slot0=max(t0,t2)=t0 [p4], slot1=max(t1,t3)=t1 [p1];
    # then slot0 wins -> p4.  A "lowest index wins" rule would give p1 instead.
    pts = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, 0, 0]], dtype=np.float32)[None]
    idx = oracle_mod.furthest_point_sample(pts, 2)
    assert idx.tolist() == [[0, 4]]
    # N=8 -> block 8, one point per thread, all at distance 1 from p0: tree order 0,4,2,6,1,5,3,7 -> p4 first
    ring = np.zeros((8, 3), dtype=np.float32)
    ang = np.arange(1, 8) * 0.7
    ring[1:, 0], ring[1:, 1] = np.cos(ang), np.sin(ang)
    ring[1:] /= np.linalg.norm(ring[1:], axis=1, keepdims=True)
    d = ((ring[1:] - ring[0]) ** 2).sum(1)
    if np.all(d == d[0]):  # only meaningful when the 7 distances are bit-identical
        assert oracle_mod.furthest_point_sample(ring[None], 2)[0, 1] == 4


def test_fps_more_samples_than_points_repeats_rule(oracle_mod):
    pts = np.array([[0, 0, 0], [1, 0, 0], [3, 0, 0]], dtype=np.float32)[None]
    idx = oracle_mod.furthest_point_sample(pts, 5)[0].tolist()
    assert idx[:3] == [0, 2, 1]
    # all min-distances are now 0: N=3 -> block 2; thread0 owns p0,p2 -> first max p0; thread1 owns p1; tie -> slot 0
    assert idx[3:] == [0, 0]


def test_ball_query_first_k_in_index_order_and_padding(oracle_mod):
    s = np.array([[0, 0, 0], [0.05, 0, 0], [5, 5, 5], [0.1, 0, 0], [0, 0.15, 0], [0.19, 0, 0]], dtype=np.float32)[None]
    q = np.array([[0, 0, 0], [5, 5, 5], [9, 9, 9]], dtype=np.float32)[None]
    idx = oracle_mod.ball_query(q, s, 3, 0.2)[0]
    assert idx[0].tolist() == [0, 1, 3]  # first three in scan order, not the three nearest
    assert idx[1].tolist() == [2, 2, 2]  # one hit, padded with it (cuh:44-48)
    assert idx[2].tolist() == [0, 0, 0]  # no hit: the caller's zero fill is untouched
    # strict '<' at the boundary (cuh:43): a point exactly at radius is out.  r=0.5 -> r2=0.25 exactly.
    s2 = np.array([[0.5, 0, 0], [0.25, 0, 0]], dtype=np.float32)[None]
    idx2 = oracle_mod.ball_query(np.zeros((1, 1, 3), np.float32), s2, 2, 0.5)[0, 0]
    assert idx2.tolist() == [1, 1]


def test_knn_points_sorted_sqrt_and_sentinels(oracle_mod):
    s = np.array([[3, 0, 0], [1, 0, 0], [2, 0, 0]], dtype=np.float32)[None]
    q = np.zeros((1, 1, 3), dtype=np.float32)
    d, i = oracle_mod.knn_points(q, s, 5)
    assert i[0, 0].tolist() == [1, 2, 0, 0, 0]  # N<k: trailing (sqrt(1e10), 0) (cuh:70-73,:90)
    assert d[0, 0].tolist() == [1.0, 2.0, 3.0, 100000.0, 100000.0]


def test_knn_points_tie_order_follows_heap(oracle_mod):
    # six supports at the same distance, k=3: simulate knn_points_cuda.cuh:16-37 by hand.
    # heap after j=0,1,2: the three 1e10 sentinels are replaced one by one -> heap holds {0,1,2};
    # later equal distances never enter (strict '<', cuh:80).  heap_sort of three equal keys: swap(0,2) then
    # reheap(size 2): dist[0] > dist[1] is false -> swap -> order depends on the mechanics, not on the index.
    s = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, 0, 0], [0, -1, 0], [0, 0, -1]], dtype=np.float32)[None]
    q = np.zeros((1, 1, 3), dtype=np.float32)
    d, i = oracle_mod.knn_points(q, s, 3)

    def ref(dists, k):
        bd, bi = [1e10] * k, [0] * k

        def reheap(n):
            root, child = 0, 1
            while child < n:
                if child + 1 < n and bd[child + 1] > bd[child]:
                    child += 1
                if bd[root] > bd[child]:
                    return
                bd[root], bd[child] = bd[child], bd[root]
                bi[root], bi[child] = bi[child], bi[root]
                root, child = child, child * 2 + 1

        for j, dj in enumerate(dists):
            if dj < bd[0]:
                bd[0], bi[0] = dj, j
                reheap(k)
        for n in range(k - 1, 0, -1):
            bd[0], bd[n] = bd[n], bd[0]
            bi[0], bi[n] = bi[n], bi[0]
            reheap(n)
        return bi

    assert i[0, 0].tolist() == ref([1.0] * 6, 3)
    assert set(i[0, 0].tolist()) == {0, 1, 2}
    assert d[0, 0].tolist() == [1.0, 1.0, 1.0]


def test_three_nn_cascade_and_unfilled(oracle_mod):
    s = np.array([[2, 0, 0], [1, 0, 0], [1, 0, 0], [3, 0, 0]], dtype=np.float32)[None]
    q = np.zeros((1, 1, 3), dtype=np.float32)
    d, i = oracle_mod.three_nn(q, s)
    assert i[0, 0].tolist() == [1, 2, 0]  # equal distances keep the lower index first (strict '<', cuh:49-64)
    assert d[0, 0].tolist() == [1.0, 1.0, 2.0]
    d2, i2 = oracle_mod.three_nn(q, s[:, :2])
    assert i2[0, 0].tolist() == [1, 0, -1]  # S<3: index -1, distance sqrt(1e40) cast to float (cuh:37-42,:67-73)
    assert d2[0, 0, 2] == np.float32(1e20)


def test_three_interpolate_and_gathers_against_numpy(oracle_mod):
    rng = np.random.default_rng(0)
    src = rng.standard_normal((2, 5, 11)).astype(np.float32)
    idx = rng.integers(0, 11, (2, 7, 3))
    w = rng.random((2, 7, 3)).astype(np.float32)
    out = oracle_mod.three_interpolate_forward(src, idx, w)
    expect = np.zeros((2, 5, 7), np.float32)
    for b in range(2):
        g = src[b][:, idx[b]]  # (C, Q, 3)
        expect[b] = (g[..., 0] * w[b, :, 0] + g[..., 1] * w[b, :, 1]) + g[..., 2] * w[b, :, 2]
    assert np.array_equal(out, expect)
    gout = rng.standard_normal((2, 5, 7)).astype(np.float32)
    gsrc = oracle_mod.three_interpolate_backward(gout, idx, w, 11)
    e2 = np.zeros((2, 5, 11), np.float64)
    for b in range(2):
        for q in range(7):
            for j in range(3):
                e2[b, :, idx[b, q, j]] += (gout[b, :, q] * w[b, q, j]).astype(np.float32)
    np.testing.assert_allclose(gsrc, e2, rtol=1e-5, atol=1e-6)

    gidx = rng.integers(0, 11, (2, 4, 3))
    gg = oracle_mod.group_gather_forward(src, gidx)
    assert gg.shape == (2, 5, 4, 3)
    for b in range(2):
        assert np.array_equal(gg[b], src[b][:, gidx[b]])
    gb = oracle_mod.group_gather_backward(gg, gidx, 11)
    e3 = np.zeros((2, 5, 11), np.float64)
    for b in range(2):
        np.add.at(e3[b], (slice(None), gidx[b].reshape(-1)), gg[b].reshape(5, -1))
    np.testing.assert_allclose(gb, e3, rtol=1e-5, atol=1e-6)
    # every element width of AT_DISPATCH_ALL_TYPES_AND(Bool)
    for dt in (np.uint8, np.int8, np.int16, np.int32, np.int64, np.float64, np.bool_):
        a = (rng.integers(0, 2, (1, 3, 9)) if dt == np.bool_ else rng.integers(0, 100, (1, 3, 9))).astype(dt)
        ii = rng.integers(0, 9, (1, 6))
        assert np.array_equal(oracle_mod.gather_forward(a, ii)[0], a[0][:, ii[0]])
