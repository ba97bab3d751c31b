"""CPU: the NumPy restatements of the SURVEY 8f "next" rows (oracle/next_rows.py) against fixtures produced by the
reference's own Python (tests/golden/make_golden.py)."""
import numpy as np
import pytest


@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_mutual_topk_oracle_matches_reference_fixture(oracle_mod, golden, name):
    g = golden("ref_select.npz")
    score, k = g[f"{name}_score"], int(g[f"{name}_k"])
    for largest in (True, False):
        for mutual in (True, False):
            for thr in (None, 0.25 if largest else -0.25):
                tag = f"{name}_l{int(largest)}_m{int(mutual)}_t{int(thr is not None)}"
                got = oracle_mod.mutual_topk_corr_mat(score, k, largest=largest, threshold=thr, mutual=mutual)
                assert np.array_equal(got, g[tag]), tag
                got = oracle_mod.mutual_topk_corr_mat(score, k, row_masks=g[f"{name}_rmask"],
                                                      col_masks=g[f"{name}_cmask"], largest=largest, threshold=thr,
                                                      mutual=mutual)
                assert np.array_equal(got, g[tag + "_masked"]), tag
    b, r, c, s = oracle_mod.mutual_topk_select(score, k)
    assert np.array_equal(np.stack([b, r, c]), g[f"{name}_reduced"])
    assert np.array_equal(s, g[f"{name}_reduced_scores"])


def test_mutual_topk_oracle_2d_union(oracle_mod, golden):
    g = golden("ref_select.npz")
    _, r, c, s = oracle_mod.mutual_topk_select(g["a_score"][:1], 3, largest=False, mutual=False)
    assert np.array_equal(np.stack([r, c]), g["a2d_reduced"])
    assert np.array_equal(s, g["a2d_reduced_scores"])


def test_mutual_topk_oracle_tie_rule_lower_index_first(oracle_mod):
    s = np.zeros((1, 4, 5), dtype=np.float32)  # all equal: top-2 of a row = columns 0,1; of a column = rows 0,1
    corr = oracle_mod.mutual_topk_corr_mat(s, 2)
    want = np.zeros((1, 4, 5), dtype=bool)
    want[0, :2, :2] = True
    assert np.array_equal(corr, want)


# ---- rank 3: pack-mode row gathers ---------------------------------------------------------------------------
def test_pack_gather_oracle_matches_reference_fixture(oracle_mod, golden):
    g = golden("ref_pack_ops.npz")
    feats, table = g["feats"], g["table"]
    assert np.array_equal(oracle_mod.index_select_rows(feats, table), g["index_select"])
    pooled, argk = oracle_mod.local_maxpool_pack(feats, table)
    assert np.array_equal(pooled, g["maxpool"])
    assert np.array_equal(oracle_mod.nearest_interpolate_pack(feats, table), g["nearest"])
    # torch.mean's summation order is not part of the contract: 1e-5 of the operand scale
    assert np.abs(oracle_mod.global_avgpool_pack(feats, g["lengths"]) - g["avgpool"]).max() <= 1e-5 * np.abs(feats).max()
    # backward: torch autograd of the reference formulation (float atomics / unordered on GPU, sequential on CPU)
    got = oracle_mod.rows_scatter(g["index_select_upstream"], table, feats.shape[0], mode=0)
    scale = np.abs(g["index_select_upstream"]).max() * table.shape[1]
    assert np.abs(got - g["index_select_grad"]).max() <= 1e-5 * scale
    got = oracle_mod.rows_scatter(g["maxpool_upstream"], table, feats.shape[0], mode=2, argk=argk)
    assert np.abs(got - g["maxpool_grad"]).max() <= 1e-5 * scale


def test_knn_interpolate_oracle_properties(oracle_mod):
    """ops/knn_interpolate.py cannot be imported (pykeops at module level): parity unpinned, checked by properties."""
    rng = np.random.default_rng(3)
    s = rng.random((50, 3), dtype=np.float32)
    q = rng.random((40, 3), dtype=np.float32)
    feats = rng.standard_normal((50, 8)).astype(np.float32)
    table = rng.integers(0, 51, size=(40, 6)).astype(np.int64)
    table[0] = 50
    w = oracle_mod.knn_interp_weights_pack(q, s, table, k=4)
    assert w.shape == (40, 4) and np.all(w[0] == 0) and np.all(w[table[:, :4] == 50] == 0)
    has = (table[:, :4] != 50).any(axis=1)
    assert np.allclose(w[has].sum(axis=1), 1.0, atol=1e-5)
    const = np.ones((50, 8), dtype=np.float32) * 3
    out = oracle_mod.knn_interpolate_pack(q, s, const, table, k=4)
    assert np.allclose(out[has], 3.0, atol=1e-4) and np.all(out[0] == 0)
    # a query sitting on a support point reproduces that point's features
    q2 = s[:5].copy()
    t2 = np.stack([np.arange(5), (np.arange(5) + 7) % 50], axis=1).astype(np.int64)
    out = oracle_mod.knn_interpolate_pack(q2, s, feats, t2)
    assert np.allclose(out, feats[:5], atol=1e-4)


def test_extract_correspondences_from_scores_oracle_matches_reference_fixture(oracle_mod, golden):
    g = golden("ref_select.npz")
    for mutual in (False, True):
        for dustbin in (False, True):
            for thr in (0.0, 0.05):
                tag = f"m_m{int(mutual)}_d{int(dustbin)}_t{int(thr > 0)}"
                r, c, s = oracle_mod.extract_correspondences_from_scores(g["m_log_p"], mutual, dustbin, thr)
                assert np.array_equal(np.stack([r, c]), g[tag]), tag
                assert np.allclose(s, g[tag + "_scores"], rtol=1e-6), tag  # exp() of numpy vs torch: 1 ulp


def test_threshold_and_global_topk_oracle_match_reference_fixture(oracle_mod, golden):
    """ops/matching.py:64-121 run on the CPU -> ref_matching.npz."""
    g = golden("ref_matching.npz")
    for i in range(3):
        r, c, s = oracle_mod.extract_correspondences_from_scores_threshold(g["log_p"], float(g[f"thr{i}_value"]))
        assert np.array_equal(np.stack([r, c]), g[f"thr{i}"]), i
        assert np.allclose(s, g[f"thr{i}_scores"], rtol=1e-6), i  # exp() of numpy vs torch: 1 ulp
    for k in (1, 17, 300):
        for largest in (True, False):
            for dustbin in (False, True):
                tag = f"topk{k}_l{int(largest)}_d{int(dustbin)}"
                r, c, s = oracle_mod.extract_correspondences_from_scores_topk(g["log_p"], k, dustbin, largest)
                assert np.array_equal(np.stack([r, c]), g[tag]), tag
                assert np.array_equal(s, g[tag + "_scores"]), tag


def test_global_topk_oracle_tie_rule_and_range(oracle_mod):
    s = np.zeros((3, 4), dtype=np.float32)
    s[2, 1] = 1.0
    r, c, _ = oracle_mod.extract_correspondences_from_scores_topk(s, 4)
    assert list(zip(r, c)) == [(2, 1), (0, 0), (0, 1), (0, 2)]  # equal scores: ascending flat index
    with pytest.raises(RuntimeError, match="out of range"):
        oracle_mod.extract_correspondences_from_scores_topk(s, 13)


def test_voxelize_oracle_matches_reference_fixture(oracle_mod, golden):
    """ops/voxelize.py:6-63 run on the CPU -> ref_voxelize.npz; bit-exact (sequential sums in input order)."""
    g = golden("ref_voxelize.npz")
    v, n, f, l = oracle_mod.voxelize_pack(g["points"], g["lengths"], float(g["voxel_size"]), g["feats"])
    assert np.array_equal(v, g["voxels"]) and np.array_equal(l, g["v_lengths"])
    assert np.array_equal(n, g["nodes"]) and np.array_equal(f, g["v_feats"])
    assert int(g["lengths"][1]) == 0 and int(l[1]) == 0  # the empty cloud stays empty
