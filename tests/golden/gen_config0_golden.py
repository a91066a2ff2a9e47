"""Fixture generator for BASELINE configs[0] AS WRITTEN (10,000-passage toy index, B = 8, K = 50, fp32, dropout 0; tools/config0.py has
the world): the CPU ORACLES' outputs for the first batch -- the query tower's fp16 query embeddings, the exact MIPS result (doc ids and
fp16 scores of the top 51), the assembled token tensors (context / reader / one-context inputs), the prior over the passages and both
losses -- so that `-m gpu` can check the HIP path against configs[0] without the 8-minute CPU run.

The oracles themselves are pinned on the reference (tests/golden/gen_model_golden.py, gen_assembly_golden.py, gen_mips_golden.py); this
script does not import /root/reference.  ~10 minutes on 8 cores (forward only).

    python tests/golden/gen_config0_golden.py"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import config0 as c0  # noqa: E402
from oracle import assembly_oracle as ao  # noqa: E402
from oracle import mips_oracle as mo  # noqa: E402
from oracle import transformer_oracle as to  # noqa: E402


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    rows, passages, titles, group_of_doc, groups, qa = c0.world()
    P = to.random_params(c0.CFG, c0.V_BERT, c0.V_T5, seed=1234)
    uid, qb, qlen, dec, labels, mask = c0.batch(qa, 0)
    t0 = time.time()
    with torch.no_grad():
        q_emb = to.bert_embed(P, "retriever_model.query_model", c0.CFG, qb, ~to.make_attention_mask_3d(qb, qb), torch.zeros_like(qb))
        q16 = q_emb.to(torch.float16).numpy()
        dist, ids = mo.topk(rows, q16, c0.K + 1, ids=np.arange(1, c0.N_DOCS + 1, dtype=np.int32))
        corpus = ao.Corpus(passages, titles, group_of_doc)
        ctx, typ, ext, one, kept = ao.postprocess(uid.tolist(), qb.tolist(), qlen.tolist(), ids.tolist(), corpus, c0.K, c0.S_RET, c0.S, c0.CLS, c0.SEP, c0.PAD)
        tt = lambda x: torch.tensor(x, dtype=torch.int64)
        ctx, typ, ext, one = tt(ctx), tt(typ), tt(ext), tt(one)
        lm, tlp, oc = to.emdr2_forward(P, c0.CFG, qb, torch.zeros_like(qb), ~to.make_attention_mask_3d(qb, qb), ctx, typ, ext, one, dec)
        lm_loss = to.reader_ce_loss(lm, labels, mask)
        r_loss = to.retriever_loss_and_utility(oc, tlp, labels, mask, c0.EOS)[0]
    assert int(ctx.max()) < 65536 and int(ext.max()) < 65536
    np.savez_compressed(os.path.join(HERE, "config0_ref.npz"), q16=q16, dist=dist, ids=ids, kept=np.asarray(kept, dtype=np.int32),
                        ctx=ctx.numpy().astype(np.uint16), typ=typ.numpy().astype(np.uint8), ext=ext.numpy().astype(np.uint16), one=one.numpy().astype(np.uint16),
                        tlp=tlp.numpy().astype(np.float32), lm_loss=np.float64(float(lm_loss)), retriever_loss=np.float64(float(r_loss)),
                        lm_gold=np.asarray(torch.log_softmax(lm, -1).gather(-1, labels[..., None])[..., 0], dtype=np.float32))
    print("config0 fixture: lm_loss %.6f retriever_loss %.6f in %.0f s" % (float(lm_loss), float(r_loss), time.time() - t0))


if __name__ == "__main__":
    main()
