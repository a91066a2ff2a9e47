"""CPU: the token-assembly oracle reproduces the outputs of the reference's own functions bit for bit."""
import os

import numpy as np

import assembly_cases
from oracle import assembly_oracle as ao

GOLD = os.path.join(os.path.dirname(__file__), "golden", "assembly_ref.npz")


def run_oracle(case, cfg):
    corpus = ao.Corpus(case["passages"], case["titles"], case["group_of_doc"])
    return ao.postprocess(case["query_uid"].tolist(), case["q_t5"].tolist(), case["q_len"].tolist(), case["topk_ids"].tolist(),
                          corpus, cfg["topk"], cfg["seq_length_ret"], cfg["seq_length"], cfg["cls_id"], cfg["sep_id"], cfg["pad_id"])


def test_oracle_matches_reference_functions():
    case, cfg = assembly_cases.build(), assembly_cases.CFG
    g = np.load(GOLD)
    ctx, typ, ext, one, kept = run_oracle(case, cfg)
    flat = lambda x: np.array([row for per_q in x for row in per_q], dtype=np.int64)
    assert np.array_equal(flat(ctx), g["ctx"]) and np.array_equal(flat(typ), g["typ"])
    assert np.array_equal(np.array(ext, dtype=np.int64), g["ext"]) and np.array_equal(np.array(one, dtype=np.int64), g["one"])
    assert np.array_equal(np.array([e for k in kept for e in k], dtype=np.int32), g["kept"])


def test_case_covers_every_branch():
    """single-doc, main 0 / 1 / -1, the negative-slice quirk, both truncation offsets, the trivial-doc skip."""
    case = assembly_cases.build()
    mains, sizes = set(), set()
    for ids in case["topk_ids"]:
        for e in ids:
            docs, m = ao.get_neighbour_paragraphs(case["group_of_doc"][int(e)], int(e))
            mains.add(m); sizes.add(len(docs))
    assert mains == {0, 1, -1} and sizes == {1, 2, 3}
    assert ao.get_neighbour_paragraphs([5, 6], 6) == ([6], -1)        # doc_row[-1:2]
    assert int(case["query_uid"][2]) in case["topk_ids"][2].tolist()
