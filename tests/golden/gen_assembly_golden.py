"""Generate tests/golden/assembly_ref.npz by calling the REFERENCE's own token-assembly functions
(megatron/model/emdr2_model.py:306-376, megatron/data/orqa_wiki_dataset.py:86-120,
tools/inverted_title_index.py:23-38) on the synthetic corpus of assembly_cases.py.  Build container only."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402
import assembly_cases  # noqa: E402


def main():
    _ref_import.install_import_shims()
    from megatron.model.emdr2_model import query_extended_context_t5_format, query_single_context_t5_format
    from megatron.data.orqa_wiki_dataset import build_tokens_types_paddings_from_ids as context_bert_format
    from tools.inverted_title_index import WikiTitleDocMap
    case, cfg = assembly_cases.build(), assembly_cases.CFG
    wmap = WikiTitleDocMap.__new__(WikiTitleDocMap)            # skip the TSV parse; fill the two dicts it builds
    wmap.docid2title = {d: tuple(case["titles"][d - 1]) + (min(g),) for d, g in case["group_of_doc"].items()}
    wmap.title2docs = {}
    for d, g in case["group_of_doc"].items():
        wmap.title2docs[wmap.docid2title[d]] = list(g)
    ctx, typ, ext, one, kept = [], [], [], [], []
    for qid, q_t5, q_len, ids in zip(case["query_uid"].tolist(), case["q_t5"], case["q_len"], case["topk_ids"].tolist()):
        k = 0
        q = q_t5.tolist()[:q_len]
        for eid in ids:                                           # the loop body of postprocess (emdr2_model.py:263-296)
            if qid != eid and k < cfg["topk"]:
                k += 1
                doc_idxs, main = wmap.get_neighbour_paragraphs(eid)
                docs = [list(case["passages"][d - 1]) for d in doc_idxs]
                title = list(case["titles"][eid - 1])
                c, t, _ = context_bert_format(title + [cfg["sep_id"]] + docs[main], cfg["seq_length_ret"], cfg["cls_id"], cfg["sep_id"], cfg["pad_id"])
                ctx.append(c); typ.append(t); kept.append(eid)
                ext.append(query_extended_context_t5_format(q, title, docs, main, cfg["seq_length"], cfg["sep_id"], cfg["pad_id"]))
                one.append(query_single_context_t5_format(q, title, docs[main], cfg["seq_length"], cfg["sep_id"], cfg["pad_id"]))
    out = os.path.join(HERE, "assembly_ref.npz")
    np.savez_compressed(out, ctx=np.array(ctx, dtype=np.int64), typ=np.array(typ, dtype=np.int64), ext=np.array(ext, dtype=np.int64),
                        one=np.array(one, dtype=np.int64), kept=np.array(kept, dtype=np.int32))
    print("saved", out, np.array(ctx).shape, np.array(ext).shape)


if __name__ == "__main__":
    main()
