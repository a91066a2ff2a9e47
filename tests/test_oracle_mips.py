"""CPU tests: the MIPS oracle against (i) an independent big-integer restatement, (ii) the golden
outputs produced by running the reference's own search_mips_index (tests/golden/gen_mips_golden.py)."""
import os

import numpy as np
import pytest

import mips_cases
from oracle import mips_oracle as mo
from tests.parity import assert_bit_identical, assert_same_modulo_ties

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _gold(case):
    g = np.load(os.path.join(GOLD, "mips_ref_%s.npz" % case["name"]))
    assert str(g["digest"]) == mips_cases.digest(case), "input builder drifted from the fixture"
    return g


def test_exact_sum_vs_bigint_including_extremes():
    rng = np.random.default_rng(0)
    rows = rng.standard_normal((40, 24)).astype(np.float16)
    q = rng.standard_normal((3, 24)).astype(np.float16)
    rows[0, :4] = np.array([65504, -65504, 6e-8, 1e-4], dtype=np.float16)
    q[0, :4] = np.array([1.0, 1.0, 6e-8, 3.0], dtype=np.float16)
    rows[1, :] = 0
    rows[2, :3] = np.array([6e-8, 6e-8, -6e-8], dtype=np.float16)     # subnormal products
    q[1, :3] = np.array([6e-8, 0.5, 0.25], dtype=np.float16)
    a = mo.scores(rows, q)
    b = mo.scores_bigint(rows, q)
    assert np.array_equal(a.view(np.uint16), b.view(np.uint16))


def test_rounding_ties_to_even_and_overflow():
    # 2049 = 2048 + 1 is a tie between 2048 and 2050 -> even mantissa (2048); 2051 -> 2052
    rows = np.zeros((4, 8), dtype=np.float16)
    rows[0, :2] = [2048, 1]
    rows[1, :2] = [2048, 3]
    rows[2, :2] = [65504, 65504]      # overflow -> inf
    rows[3, :2] = [65504, 15]         # 65519 < 65520 -> 65504
    q = np.zeros((1, 8), dtype=np.float16)
    q[0, :2] = 1
    s = mo.scores(rows, q)[0].astype(np.float32)
    assert s[0] == 2048 and s[1] == 2052 and np.isinf(s[2]) and s[3] == 65504


def test_nonfinite_input_is_rejected():
    rows = np.zeros((2, 8), dtype=np.float16)
    rows[1, 0] = np.inf
    with pytest.raises(ValueError):
        mo.scores(rows, np.ones((1, 8), dtype=np.float16))


def test_topk_matches_explicit_sort():
    rng = np.random.default_rng(5)
    rows = rng.standard_normal((700, 64)).astype(np.float16)
    q = rng.standard_normal((9, 64)).astype(np.float16)
    ids = (rng.permutation(700) + 1).astype(np.int32)
    d, i = mo.topk(rows, q, 33, ids=ids)
    d2, i2 = mo.topk_from_scores(mo.scores(rows, q), 33, ids=ids)
    assert_bit_identical(d, i, d2, i2)


def test_k_larger_than_n_pads():
    rows = np.eye(4, 8, dtype=np.float16)
    q = np.ones((1, 8), dtype=np.float16)
    d, i = mo.topk(rows, q, 6)
    assert list(i[0]) == [0, 1, 2, 3, -1, -1] and np.isneginf(d[0, 4:].astype(np.float32)).all()


def test_golden_exact_distinct_is_bit_identical_to_reference():
    case = mips_cases.case_exact_distinct()
    g = _gold(case)
    d, i = mo.topk(case["rows"], case["queries"], case["k"], ids=case["ids"])
    assert_bit_identical(d, i, g["dist"].view(np.float16), g["idx"])
    assert_bit_identical(d, i, g["dist_3dev"].view(np.float16), g["idx_3dev"])


@pytest.mark.parametrize("fn", [mips_cases.case_exact_ties, mips_cases.case_realistic, mips_cases.case_realistic_k100])
def test_golden_matches_reference_modulo_tie_order(fn):
    """torch.topk's order among equal fp16 scores is unspecified; everything else must be identical."""
    case = fn()
    g = _gold(case)
    d, i = mo.topk(case["rows"], case["queries"], case["k"], ids=case["ids"])
    full = mo.scores(case["rows"], case["queries"])
    assert_same_modulo_ties(d, i, g["dist"].view(np.float16), g["idx"], all_scores=full, ids=case["ids"])


def test_canonical_tie_order_is_row_ascending():
    case = mips_cases.case_exact_ties()
    d, i, r = mo.topk(case["rows"], case["queries"], case["k"], ids=case["ids"], return_rows=True)
    s = d.astype(np.float32)
    for q in range(s.shape[0]):
        for j in range(1, s.shape[1]):
            assert s[q, j - 1] > s[q, j] or (s[q, j - 1] == s[q, j] and r[q, j - 1] < r[q, j])


def test_shard_merge_invariance():
    """Searching row shards separately and merging by (score desc, row asc) equals one search."""
    case = mips_cases.case_realistic_k100()
    rows, q, k = case["rows"], case["queries"], 50
    d, _, r = mo.topk(rows, q, k, return_rows=True)
    for nshard in (2, 3, 8):
        bounds = np.linspace(0, rows.shape[0], nshard + 1).astype(int)
        ds, rs = [], []
        for a, b in zip(bounds[:-1], bounds[1:]):
            dd, _, rr = mo.topk(rows[a:b], q, k, row_base=int(a), return_rows=True)
            ds.append(dd); rs.append(rr)
        dcat, rcat = np.concatenate(ds, 1), np.concatenate(rs, 1)
        order = np.lexsort((rcat, -dcat.astype(np.float32)), axis=1)[:, :k]
        assert np.array_equal(np.take_along_axis(rcat, order, 1), r)
        assert np.array_equal(np.take_along_axis(dcat, order, 1).view(np.uint16), d.view(np.uint16))


def test_fp32accum_port_agrees_except_rounding_boundaries():
    case = mips_cases.case_realistic()
    d, _, r = mo.topk(case["rows"], case["queries"], case["k"], return_rows=True)
    d2, r2 = mo.topk_fp32accum(case["rows"], case["queries"], case["k"])
    assert (d2.view(np.uint16) == d.view(np.uint16)).mean() > 0.995
    assert (r2 == r).mean() > 0.99


def test_flat_embedding_file_round_trips_and_converts_to_the_reference_pickle(tmp_path):
    """flat [N, D] fp16 + ids file <-> the reference's pickle store, same row order."""
    import numpy as np
    from emdr2_amd.data.emdr2_index import FlatEmbeddingFile, OpenRetreivalDataStore
    rng = np.random.default_rng(0)
    ids = (rng.permutation(100) + 1).astype(np.int64)
    rows = rng.standard_normal((100, 64)).astype(np.float16)
    store = OpenRetreivalDataStore(str(tmp_path / "emb.pkl"), load_from_path=False, rank=0)
    store.add_block_data(ids, rows)
    flat = FlatEmbeddingFile.from_store(store, str(tmp_path / "emb.flat"))
    assert flat.n == 100 and flat.dim == 64 and np.array_equal(flat.ids, ids) and np.array_equal(np.asarray(flat.rows), rows)
    assert flat._rows_off % 4096 == 0
    back = flat.to_store(str(tmp_path / "emb2.pkl"))
    assert list(back.embed_data.keys()) == ids.tolist()
    assert all(np.array_equal(back.embed_data[int(i)], rows[n]) and back.embed_data[int(i)].dtype == np.float16 for n, i in enumerate(ids))
    back.save_shard(); back.merge_shards_and_save()
    again = OpenRetreivalDataStore(str(tmp_path / "emb2.pkl"), load_from_path=True)
    i2, r2 = again.to_arrays()
    assert np.array_equal(i2, ids.astype(np.int32)) and np.array_equal(r2, rows)
    import pytest
    (tmp_path / "bad").write_bytes(b"x" * 64)
    with pytest.raises(ValueError):
        FlatEmbeddingFile(str(tmp_path / "bad"))
