"""Comparison helpers shared by the CPU and GPU parity tests."""
import numpy as np


def assert_same_modulo_ties(dist_a, idx_a, dist_b, idx_b, all_scores=None, ids=None):
    """Two top-k results over the same data agree up to the order of equal-score rows.

    - the sorted score lists are bit-identical;
    - for every score strictly above the k-th score, the id multisets are identical;
    - in the k-th score's bucket the ids may differ (a tie at the boundary), but if the full
      canonical score matrix is given, every returned id must carry exactly that score.
    """
    da, db = dist_a.view(np.uint16), dist_b.view(np.uint16)
    assert da.shape == db.shape and idx_a.shape == idx_b.shape
    assert np.array_equal(da, db), "score lists differ"
    nq, k = da.shape
    id_to_row = None
    if all_scores is not None:
        n = all_scores.shape[1]
        id_to_row = {int(i): r for r, i in enumerate(ids)} if ids is not None else None
    for q in range(nq):
        kth = da[q, -1]
        for s in np.unique(da[q]):
            sel = da[q] == s
            a, b = np.sort(idx_a[q][sel]), np.sort(idx_b[q][sel])
            if s != kth:
                assert np.array_equal(a, b), "query %d: ids differ above the boundary bucket" % q
        if all_scores is not None:
            for res in (idx_a[q], idx_b[q]):
                rows = np.array([id_to_row[int(i)] if id_to_row else int(i) for i in res])
                got = all_scores[q, rows].view(np.uint16)
                assert np.array_equal(got, da[q]), "query %d: an id does not carry its reported score" % q


def assert_bit_identical(dist_a, idx_a, dist_b, idx_b):
    assert np.array_equal(dist_a.view(np.uint16), dist_b.view(np.uint16)), "scores differ"
    assert np.array_equal(idx_a, idx_b), "ids differ"
