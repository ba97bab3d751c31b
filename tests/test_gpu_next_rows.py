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
