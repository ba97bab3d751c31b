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
