"""Loads tests/golden/model_ref.npz (reference modules run on CPU) for the oracle and HIP parity tests."""
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CFG = dict(layers=2, hidden=32, heads=2, ffn=128, seq=48, seq_ret=24, dec=8, topk=3, batch=2)


def load():
    g = np.load(os.path.join(GOLD, "model_ref.npz"))
    P = {k[len("emdr2."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("emdr2.")}
    grads = {k[len("grad."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("grad.")}
    meta = dict(zip(["bert_vocab", "t5_vocab", "cls", "sep", "pad", "bos", "eos"], [int(x) for x in g["meta"]]))
    c = np.load(os.path.join(GOLD, "model_corpus.npz"), allow_pickle=True)
    return g, P, grads, meta, [list(x) for x in c["passages"]], [list(x) for x in c["titles"]]


def assembled_inputs(g, meta, passages, titles):
    """The four tensors `postprocess` builds for the F4 case, from the assembly oracle."""
    import assembly_cases
    from oracle import assembly_oracle as ao
    case = assembly_cases.build()
    corpus = ao.Corpus(passages, titles, case["group_of_doc"])
    ctx, typ, ext, one, _ = ao.postprocess(g["e_query_uid"].tolist(), g["e_query_ids"].tolist(), g["e_query_len"].tolist(),
                                           g["e_topk_ids"].tolist(), corpus, CFG["topk"], CFG["seq_ret"], CFG["seq"],
                                           meta["cls"], meta["sep"], meta["pad"])
    t = lambda x: torch.tensor(x, dtype=torch.int64)
    return t(ctx), t(typ), t(ext), t(one)
