"""BASELINE configs[0] AS WRITTEN under -m gpu: the 10,000-passage toy index, 128 synthetic QA pairs, B = 8, K = 50,
S_ret 256, S 512, L 32, 12-layer towers + the 12 + 12-layer reader, dropout 0, random-init weights (seed 1234) -- the world of
tools/config0.py -- through the HIP path, against what the CPU ORACLES produced for the same first batch
(tests/golden/config0_ref.npz, written by tests/golden/gen_config0_golden.py: ~8 minutes of fp32 CPU work the GPU box need not repeat):
retrieved doc ids and fp16 scores bit-identical, assembled token tensors identical, prior and gold log-probabilities within the bf16
tolerance north_star states (2e-2), both losses within 2e-3."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_config0_first_batch_against_the_oracle_fixture():
    import config0 as c0
    from oracle import transformer_oracle as to
    from emdr2_amd.data.emdr2_index import HipIndexShard
    from emdr2_amd.data.evidence_arena import EvidenceArena
    from emdr2_amd.model.emdr2_model import EMDR2Model, emdr2_loss
    from emdr2_amd.model.transformer import Config
    ref = np.load(os.path.join(ROOT, "tests", "golden", "config0_ref.npz"))
    rows, passages, titles, group_of_doc, groups, qa = c0.world()
    uid, qb, qlen, dec, labels, mask = c0.batch(qa, 0)
    P = to.random_params(c0.CFG, c0.V_BERT, c0.V_T5, seed=1234)                 # (the oracle module only as the weight generator of the world)
    K, N = c0.K, c0.N_DOCS

    cfg = Config(num_layers=12, hidden_size=768, num_attention_heads=12, ffn_hidden_size=3072, max_position_embeddings=512, init_method_std=0.02)
    m = EMDR2Model(None, cfg, c0.V_T5, c0.V_BERT, K, c0.S, c0.S_RET, cls_id=c0.CLS, sep_id=c0.SEP)
    m.load_state_dict({k: v for k, v in P.items()})
    m.train()

    # a5: the search of the ORACLE's fp16 queries (the fixture's) -- bit-exact ids and scores; and of the HIP query tower's own queries: the
    # same ids wherever the score gap to the next passage exceeds the towers' bf16 difference
    shard = HipIndexShard(c0.DIM, N, 0)
    shard.append_rows(rows)
    shard.set_ids(torch.arange(1, N + 1, dtype=torch.int32, device="cuda"))
    gd, gi, _, flags = shard.search(torch.from_numpy(ref["q16"]).cuda(), K + 1)
    assert int(flags.abs().sum()) == 0
    assert np.array_equal(gi.cpu().numpy(), ref["ids"]) and np.array_equal(gd.cpu().numpy().view(np.uint16), ref["dist"].view(np.uint16))
    with torch.no_grad():
        q_hip = m.retriever_embedder(qb.cuda(), None, torch.zeros_like(qb).cuda(), "query")
    assert float((q_hip.float().cpu() - torch.from_numpy(ref["q16"]).float()).abs().max()) < 2e-2 * float(np.abs(ref["q16"].astype(np.float32)).max())

    # a3 / a8: evidence fetch + token assembly on the device
    arena = EvidenceArena(passages, titles, title_keys=[tuple(group_of_doc[d + 1]) for d in range(N)])
    out = arena.assemble(gi, K, torch.from_numpy(uid).cuda(), qb.cuda(), torch.from_numpy(qlen).cuda(), c0.S_RET, c0.S, c0.CLS, c0.SEP, c0.PAD)
    g_ctx, g_typ, g_ext, g_one = out[0], out[1], out[2], out[3]
    for got, key in ((g_ctx, "ctx"), (g_typ, "typ"), (g_ext, "ext"), (g_one, "one")):
        assert np.array_equal(got.cpu().numpy().astype(np.int64).reshape(ref[key].shape), ref[key].astype(np.int64)), key

    # a2, a9-a14: towers, reader, one-context pass, both losses
    ql = m.retriever_embedder(qb.cuda(), None, torch.zeros_like(qb).cuda(), "query")
    lm, tlp, one = m.forward_assembled(ql, g_ctx, g_typ, g_ext, g_one, dec.cuda())
    loss, stats = emdr2_loss(lm, tlp, one, labels.cuda(), mask.cuda(), eos_id=c0.EOS)
    loss.backward()
    assert float((tlp.float().cpu() - torch.from_numpy(ref["tlp"])).abs().max()) < 2e-2 * float(np.abs(ref["tlp"]).max())
    gold = torch.log_softmax(lm.float(), -1).gather(-1, labels.cuda()[..., None])[..., 0].cpu()
    real = labels != 0
    assert float((gold[real] - torch.from_numpy(ref["lm_gold"])[real]).abs().max()) < 2e-2 * float(np.abs(ref["lm_gold"][real.numpy()]).max())
    assert abs(float(stats["lm_loss"]) - float(ref["lm_loss"])) < 2e-3 * float(ref["lm_loss"])
    assert abs(float(stats["retriever_loss"]) - float(ref["retriever_loss"])) < 2e-3 * float(ref["retriever_loss"])
    assert all(p.grad is None or bool(torch.isfinite(p.grad).all()) for p in m.parameters())


def test_config0_first_batch_in_the_fp32_mode_within_1e_3():
    """r06: BASELINE configs[0] is the reference's own CPU-runnable case and an fp32 one (no --fp16: megatron/training.py:55-56).  The same world
    and first batch through the VALIDATION-ONLY fp32 compute mode (Config(compute_dtype="fp32"), forward only: the score matrices of 400
    sequences x 12 heads x 512 x 512 are 5 GB per layer in this mode), against the fp32 oracle's fixture at north_star's fp32 bar of 1e-3:
    the query embeddings (so that the fp16 queries the search sees are the oracle's, and with them the retrieved ids), the prior over the K
    passages, the gold log-probabilities of the reader, both losses."""
    import config0 as c0
    from oracle import transformer_oracle as to
    from emdr2_amd.data.emdr2_index import HipIndexShard
    from emdr2_amd.data.evidence_arena import EvidenceArena
    from emdr2_amd.model import kernels as K
    from emdr2_amd.model.emdr2_model import EMDR2Model, emdr2_loss
    from emdr2_amd.model.transformer import Config
    ref = np.load(os.path.join(ROOT, "tests", "golden", "config0_ref.npz"))
    rows, passages, titles, group_of_doc, groups, qa = c0.world()
    uid, qb, qlen, dec, labels, mask = c0.batch(qa, 0)
    P = to.random_params(c0.CFG, c0.V_BERT, c0.V_T5, seed=1234)
    Kk, N = c0.K, c0.N_DOCS
    cfg = Config(num_layers=12, hidden_size=768, num_attention_heads=12, ffn_hidden_size=3072, max_position_embeddings=512, init_method_std=0.02,
                 compute_dtype="fp32")
    m = EMDR2Model(None, cfg, c0.V_T5, c0.V_BERT, Kk, c0.S, c0.S_RET, cls_id=c0.CLS, sep_id=c0.SEP)
    m.load_state_dict({k: v for k, v in P.items()})
    m.train()                                                          # (training-mode return triple; dropout is 0 in this world)
    try:
        with torch.no_grad():
            ql = m.retriever_embedder(qb.cuda(), None, torch.zeros_like(qb).cuda(), "query")
            assert ql.dtype == torch.float32
            q_ref = torch.from_numpy(ref["q16"]).float()
            assert float((ql.cpu() - q_ref).abs().max()) < 1e-3 * float(q_ref.abs().max())
            # the search over the HIP fp32 tower's OWN queries (rounded to fp16 like the reference's index does): the oracle's ids
            shard = HipIndexShard(c0.DIM, N, 0)
            shard.append_rows(rows)
            shard.set_ids(torch.arange(1, N + 1, dtype=torch.int32, device="cuda"))
            q16 = ql.to(torch.float16)
            same_bits = bool(torch.equal(q16.cpu().view(torch.int16), torch.from_numpy(ref["q16"]).view(torch.int16)))
            gd, gi, _, flags = shard.search(q16.contiguous(), Kk + 1)
            assert int(flags.abs().sum()) == 0
            agree = float((gi.cpu() == torch.from_numpy(ref["ids"])).float().mean())
            assert agree == 1.0 if same_bits else agree > 0.99, (same_bits, agree)
            arena = EvidenceArena(passages, titles, title_keys=[tuple(group_of_doc[d + 1]) for d in range(N)])
            out = arena.assemble(torch.from_numpy(ref["ids"]).cuda(), Kk, torch.from_numpy(uid).cuda(), qb.cuda(), torch.from_numpy(qlen).cuda(), c0.S_RET, c0.S,
                                 c0.CLS, c0.SEP, c0.PAD)
            lm, tlp, one = m.forward_assembled(ql, out[0], out[1], out[2], out[3], dec.cuda())
            loss, stats = emdr2_loss(lm, tlp, one, labels.cuda(), mask.cuda(), eos_id=c0.EOS)
        assert lm.dtype == torch.float32
        assert float((tlp.cpu() - torch.from_numpy(ref["tlp"])).abs().max()) < 1e-3 * float(np.abs(ref["tlp"]).max())
        gold = torch.log_softmax(lm, -1).gather(-1, labels.cuda()[..., None])[..., 0].cpu()
        real = labels != 0
        assert float((gold[real] - torch.from_numpy(ref["lm_gold"])[real]).abs().max()) < 1e-3 * float(np.abs(ref["lm_gold"][real.numpy()]).max())
        assert abs(float(stats["lm_loss"]) - float(ref["lm_loss"])) < 1e-3 * float(ref["lm_loss"])
        assert abs(float(stats["retriever_loss"]) - float(ref["retriever_loss"])) < 1e-3 * float(ref["retriever_loss"])
        print("config0 in fp32: tlp %.2e, gold log-probs %.2e, lm loss %.2e, retriever loss %.2e (relative); query bits equal the oracle's: %s, ids agree %.4f"
              % (float((tlp.cpu() - torch.from_numpy(ref["tlp"])).abs().max()) / float(np.abs(ref["tlp"]).max()),
                 float((gold[real] - torch.from_numpy(ref["lm_gold"])[real]).abs().max()) / float(np.abs(ref["lm_gold"][real.numpy()]).max()),
                 abs(float(stats["lm_loss"]) - float(ref["lm_loss"])) / float(ref["lm_loss"]),
                 abs(float(stats["retriever_loss"]) - float(ref["retriever_loss"])) / float(ref["retriever_loss"]), same_bits, agree))
    finally:
        K.WEIGHTS.invalidate()
