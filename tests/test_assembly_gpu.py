"""GPU: the HIP evidence fetch + token assembly equals the oracle (and hence the reference's functions) bit for bit."""
import os

import numpy as np
import pytest
import torch

import assembly_cases
from oracle import assembly_oracle as ao

gpu = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "assembly_ref.npz")


def _arena(case):
    from emdr2_amd.data.evidence_arena import EvidenceArena
    keys = [min(case["group_of_doc"][d + 1]) for d in range(len(case["passages"]))]
    return EvidenceArena(case["passages"], case["titles"], title_keys=keys).to_device()


@gpu
def test_golden_case_bit_identical():
    case, cfg = assembly_cases.build(), assembly_cases.CFG
    arena = _arena(case)
    ctx, typ, ext, one, kept = arena.assemble(torch.from_numpy(case["topk_ids"]).cuda(), cfg["topk"], torch.from_numpy(case["query_uid"]),
                                              torch.from_numpy(case["q_t5"]), torch.from_numpy(case["q_len"]), cfg["seq_length_ret"],
                                              cfg["seq_length"], cfg["cls_id"], cfg["sep_id"], cfg["pad_id"])
    torch.cuda.synchronize()
    g = np.load(GOLD)
    assert np.array_equal(ctx.cpu().numpy().reshape(-1, cfg["seq_length_ret"]), g["ctx"])
    assert np.array_equal(typ.cpu().numpy().reshape(-1, cfg["seq_length_ret"]), g["typ"])
    assert np.array_equal(ext.cpu().numpy(), g["ext"]) and np.array_equal(one.cpu().numpy(), g["one"])
    assert np.array_equal(kept.cpu().numpy().reshape(-1), g["kept"])


@gpu
def test_random_corpus_at_reference_shapes_vs_oracle():
    """S_ret 256, S 512, K 50 (NQ script shapes), 64 queries, random group sizes and lengths."""
    rng = np.random.default_rng(5)
    n_docs, b, K = 4000, 64, 50
    sizes = []
    while sum(sizes) < n_docs:
        sizes.append(int(min(n_docs - sum(sizes), rng.integers(1, 9))))
    groups, d = [], 1
    for s in sizes:
        groups.append(list(range(d, d + s))); d += s
    gmap = {doc: g for g in groups for doc in g}
    passages = [rng.integers(5, 30522, size=int(rng.integers(80, 200))).tolist() for _ in range(n_docs)]
    tby = [rng.integers(5, 30522, size=int(rng.integers(1, 12))).tolist() for _ in groups]
    titles = [None] * n_docs
    for g, t in zip(groups, tby):
        for doc in g:
            titles[doc - 1] = t
    qlen = rng.integers(6, 40, size=b).astype(np.int64)
    q = np.zeros((b, 256), dtype=np.int64)
    for i in range(b):
        q[i, :qlen[i]] = rng.integers(5, 30522, size=int(qlen[i]))
    ids = np.stack([rng.permutation(n_docs)[:K + 1] + 1 for _ in range(b)]).astype(np.int32)
    uid = -np.arange(1, b + 1, dtype=np.int64); uid[3] = int(ids[3, 0])
    case = dict(passages=passages, titles=titles, group_of_doc=gmap)
    arena = _arena(case)
    ctx, typ, ext, one, kept = arena.assemble(torch.from_numpy(ids).cuda(), K, torch.from_numpy(uid), torch.from_numpy(q),
                                              torch.from_numpy(qlen), 256, 512, 101, 102, 0)
    torch.cuda.synchronize()
    corpus = ao.Corpus(passages, titles, gmap)
    octx, otyp, oext, oone, okept = ao.postprocess(uid.tolist(), q.tolist(), qlen.tolist(), ids.tolist(), corpus, K, 256, 512, 101, 102, 0)
    assert np.array_equal(ctx.cpu().numpy(), np.array(octx, dtype=np.int64))
    assert np.array_equal(typ.cpu().numpy(), np.array(otyp, dtype=np.int64))
    assert np.array_equal(ext.cpu().numpy(), np.array(oext, dtype=np.int64))
    assert np.array_equal(one.cpu().numpy(), np.array(oone, dtype=np.int64))
    assert np.array_equal(kept.cpu().numpy(), np.array(okept, dtype=np.int32))


def test_host_views_match_reference_neighbour_rule():
    case = assembly_cases.build()
    from emdr2_amd.data.evidence_arena import EvidenceArena
    keys = [min(case["group_of_doc"][d + 1]) for d in range(len(case["passages"]))]
    arena = EvidenceArena(case["passages"], case["titles"], title_keys=keys)
    for doc in range(1, len(case["passages"]) + 1):
        assert arena.neighbour_paragraphs(doc) == ao.get_neighbour_paragraphs(case["group_of_doc"][doc], doc)
        assert arena.passage(doc) == list(case["passages"][doc - 1]) and arena.title(doc) == list(case["titles"][doc - 1])


@gpu
def test_retriever_plugin_end_to_end():
    """MIPS search -> evidence fetch -> token assembly through the retriever plug-in, both paths, vs the oracles."""
    import types
    from emdr2_amd.data.emdr2_index import OpenRetreivalDataStore
    from emdr2_amd.model.emdr2_model import PreComputedEvidenceDocsRetriever
    from oracle import mips_oracle as mo
    case, cfg = assembly_cases.build(), assembly_cases.CFG
    n = len(case["passages"])
    rng = np.random.default_rng(9)
    emb = rng.standard_normal((n, 64)).astype(np.float16)
    qe = rng.standard_normal((6, 64)).astype(np.float16)
    store = OpenRetreivalDataStore(embedding_path="/tmp/_emdr2_unused2.pkl", load_from_path=False, rank=0)
    store.add_block_data(list(range(1, n + 1)), emb)
    args = types.SimpleNamespace(topk_retrievals=cfg["topk"], hidden_size=64, allow_trivial_doc=False, embedding_path=None,
                                 faiss_use_gpu=True, seq_length=cfg["seq_length"], seq_length_ret=cfg["seq_length_ret"])
    retr = PreComputedEvidenceDocsRetriever(args, _arena(case), embed_data=store)
    assert retr.topk == cfg["topk"] + 1
    od, oi = mo.topk(emb, qe, cfg["topk"] + 1, ids=np.arange(1, n + 1, dtype=np.int32))
    topk_data, dist = retr.get_topk(torch.from_numpy(qe).cuda())
    assert [t[0] for t in topk_data] == oi.tolist() and np.array_equal(dist.cpu().numpy().view(np.uint16), od.view(np.uint16))
    corpus = ao.Corpus(case["passages"], case["titles"], case["group_of_doc"])
    for (ids, texts), oids in zip(topk_data, oi.tolist()):
        for e, (docs, main, title) in zip(ids, texts):
            assert (docs, main, title) == corpus.evidence(e)
    ctx, typ, ext, one, kept, _ = retr.get_topk_assembled(torch.from_numpy(qe).cuda(), torch.from_numpy(case["query_uid"]),
                                                          torch.from_numpy(case["q_t5"]), torch.from_numpy(case["q_len"]),
                                                          cfg["cls_id"], cfg["sep_id"], cfg["pad_id"])
    octx, otyp, oext, oone, okept = ao.postprocess(case["query_uid"].tolist(), case["q_t5"].tolist(), case["q_len"].tolist(), oi.tolist(),
                                                   corpus, cfg["topk"], cfg["seq_length_ret"], cfg["seq_length"], cfg["cls_id"],
                                                   cfg["sep_id"], cfg["pad_id"])
    assert np.array_equal(ctx.cpu().numpy(), np.array(octx, dtype=np.int64)) and np.array_equal(ext.cpu().numpy(), np.array(oext, dtype=np.int64))
    assert np.array_equal(one.cpu().numpy(), np.array(oone, dtype=np.int64)) and np.array_equal(kept.cpu().numpy(), np.array(okept, dtype=np.int32))
