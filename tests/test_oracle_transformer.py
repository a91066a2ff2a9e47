"""CPU: the torch-fp32 transformer/loss oracle reproduces the reference's own modules (tests/golden/model_ref.npz)."""
import numpy as np
import torch

from oracle import transformer_oracle as to
from tests import model_fixture as mf

TOL = dict(rtol=2e-5, atol=2e-5)


def test_masks():
    g = mf.load()[0]
    a, b = torch.from_numpy(g["mask_src"]), torch.from_numpy(g["mask_tgt"])
    assert np.array_equal(to.make_attention_mask_3d(a, b).numpy(), g["mask_3d"])
    assert np.array_equal(to.make_history_mask_3d(a).numpy(), g["mask_hist"])


def test_bert_towers_cls_embedding():
    g, P, _, _, _, _ = mf.load()
    ids = torch.from_numpy(g["de_ids"])
    mask = ~to.make_attention_mask_3d(ids, ids)
    types = torch.zeros_like(ids)
    q = to.bert_embed(P, "retriever_model.query_model", mf.CFG, ids, mask, types)
    c = to.bert_embed(P, "retriever_model.context_model", mf.CFG, ids, mask, types)
    np.testing.assert_allclose(q.numpy(), g["de_query_emb"], **TOL)
    np.testing.assert_allclose(c.numpy(), g["de_context_emb"], **TOL)


def test_reader_encoder_and_full_logits():
    g, P, _, _, _, _ = mf.load()
    enc_ids, dec_ids = torch.from_numpy(g["t5_enc_ids"]), torch.from_numpy(g["t5_dec_ids"])
    enc = to.t5_encode(P, "language_model", mf.CFG, enc_ids, ~to.make_attention_mask_3d(enc_ids, enc_ids))
    np.testing.assert_allclose(enc.numpy(), g["t5_enc_out"], **TOL)
    d_mask = ~(to.make_attention_mask_3d(dec_ids, dec_ids) * to.make_history_mask_3d(dec_ids))
    logits = to.t5_decode(P, "language_model", mf.CFG, dec_ids, enc, d_mask, ~to.make_attention_mask_3d(dec_ids, enc_ids))
    np.testing.assert_allclose(logits.numpy(), g["t5_logits"], rtol=1e-4, atol=1e-4)


def test_emdr2_forward_loss_and_gradients():
    g, P, grads, meta, passages, titles = mf.load()
    ctx, typ, ext, one = mf.assembled_inputs(g, meta, passages, titles)
    P = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    qb = torch.from_numpy(g["e_query_ids"])
    lm_logits, tlp, one_logits = to.emdr2_forward(P, mf.CFG, qb, torch.zeros_like(qb), ~to.make_attention_mask_3d(qb, qb), ctx, typ, ext, one,
                                                  torch.from_numpy(g["e_dec_ids"]))
    np.testing.assert_allclose(lm_logits.detach().numpy(), g["e_lm_logits"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(tlp.detach().numpy(), g["e_topk_log_probs"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(one_logits.numpy(), g["e_one_context_logits"], rtol=1e-4, atol=1e-4)
    labels, loss_mask = torch.from_numpy(g["e_labels"]), torch.from_numpy(g["e_loss_mask"])
    lm_loss = to.reader_ce_loss(lm_logits, labels, loss_mask)
    r_loss, util, null = to.retriever_loss_and_utility(one_logits, tlp, labels, loss_mask, meta["eos"])
    np.testing.assert_allclose([lm_loss.item(), r_loss.item(), util.item(), null.item()], g["e_losses"], rtol=1e-5, atol=1e-6)
    (lm_loss + r_loss).backward()
    worst = 0.0
    for k, ref in grads.items():
        got = P[k].grad if P[k].grad is not None else torch.zeros_like(P[k])
        worst = max(worst, float((got - ref).abs().max()))
        np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=2e-3, atol=2e-6, err_msg=k)
    assert worst < 1e-3


def test_learning_rate_table():
    g = mf.load()[0]
    ours = [to.annealing_lr(i + 1, 2e-5, 10, 1000) for i in range(1000)]
    np.testing.assert_allclose(ours, g["lr_table"], rtol=1e-12, atol=0)


def test_kl_div_retriever_loss_matches_reference():
    import json
    import os
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ckpt_layout.json")))["kl"]
    got = to.retriever_kl_div_loss(torch.tensor(ref["one"]), torch.tensor(ref["tlp"]), torch.tensor(ref["labels"]), torch.tensor(ref["mask"]))
    assert abs(float(got) - ref["loss"]) < 1e-6 * max(1.0, abs(ref["loss"]))


def test_base_size_layer_fixture_from_the_reference():
    """F1 of SURVEY 8c at BASE size: the reference's ParallelTransformerLayer (transformer.py:422-563) as encoder layer (H 768, 12 heads,
    FFN 3072, s 512) and decoder layer (s 32 over 512 encoder positions): output, input gradients and every parameter gradient of the oracle
    against sampled values of the reference's own run (tests/golden/gen_layer_base_golden.py)."""
    import os
    import layer_base_case as lb
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "layer_base_ref.npz"))
    inp = lb.inputs()
    t = lambda k: torch.from_numpy(inp[k])
    enc_ids, dec_ids = t("enc_ids"), t("dec_ids")
    enc_out = None
    for kind, seed in (("encoder", 11), ("decoder", 12)):
        P = {"L." + k: torch.from_numpy(v).requires_grad_(True) for k, v in lb.layer_params(kind, seed).items()}
        if kind == "encoder":
            x = t("enc_x").clone().requires_grad_(True)
            y = to.transformer_layer(P, "L", lb.DIMS["heads"], x, (~to.make_attention_mask_3d(enc_ids, enc_ids))[:, None])
            w = t("w_enc")
            enc_out = y.detach()
        else:
            x = t("dec_x").clone().requires_grad_(True)
            enc = enc_out.clone().requires_grad_(True)
            mask = (~(to.make_attention_mask_3d(dec_ids, dec_ids) * to.make_history_mask_3d(dec_ids)))[:, None]
            ed = (~to.make_attention_mask_3d(dec_ids, enc_ids))[:, None]
            y = to.transformer_layer(P, "L", lb.DIMS["heads"], x, mask, encoder_output=enc, enc_dec_mask=ed)
            w = t("w_dec")
        (y * w).sum().backward()

        def check(name, got, tol=2e-4):
            ref = g[name]
            s = lb.sample(got.detach().numpy())
            scale = max(1e-6, float(np.abs(ref[2:]).max()))
            assert np.abs(s[2:] - ref[2:]).max() <= tol * scale, (name, np.abs(s[2:] - ref[2:]).max(), scale)
            assert abs(s[1] - ref[1]) <= 1e-3 * ref[1] + 1e-6, name                    # absolute sum of the whole tensor
        check(kind + ".out", y)
        check(kind + ".dx", x.grad)
        if kind == "decoder":
            check(kind + ".denc", enc.grad)
        for k in lb.layer_params(kind, seed):
            check(kind + ".grad." + k, P["L." + k].grad, tol=5e-4)


def _sharpen(name, w, gain):
    """The transformation tests/golden/gen_decode_golden.py applies to the stored toy weights (see there)."""
    if not name.startswith("language_model."):
        return w
    if name.endswith("lm_head.bias"):
        return torch.zeros_like(w)
    if name.endswith(".weight") and w.dim() == 2 and "layernorm" not in name and "embedding" not in name:
        return w * gain
    return w


def test_greedy_decode_matches_the_reference_search_strategy():
    """f4: the oracle's greedy decoder against the ids the reference's own SampleOrGreedySearch produced (search_strategy.py:185-240) on
    the toy model with an injected retriever (tests/golden/decode_ref.npz): identical token ids for all 16 questions."""
    import os
    import assembly_cases
    from oracle import assembly_oracle as ao
    g, P, _, meta, passages, titles = mf.load()
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "decode_ref.npz"))
    P = {k: _sharpen(k, v, float(d["reader_gain"])) for k, v in P.items()}
    corpus = ao.Corpus(passages, titles, assembly_cases.build()["group_of_doc"])
    _, _, ext, _, _ = ao.postprocess(d["query_uid"].tolist(), d["query_ids"].tolist(), d["query_len"].tolist(), d["topk_ids"].tolist(), corpus,
                                     mf.CFG["topk"], mf.CFG["seq_ret"], mf.CFG["seq"], meta["cls"], meta["sep"], meta["pad"])
    outs, margins = to.greedy_decode(P, mf.CFG, torch.tensor(ext, dtype=torch.int64), mf.CFG["topk"], int(d["max_decode_len"]), int(d["bos"]),
                                     int(d["eos"]), return_margins=True)
    ref = [[t for t in row if t >= 0] for row in d["decoded"].tolist()]
    assert outs == ref
    assert len(set(map(tuple, ref))) >= 8                                # the fixture depends on the evidence, it is not one constant answer
    np.testing.assert_allclose(margins.numpy(), d["margins"], rtol=2e-2, atol=2e-4)


def test_beam_decode_matches_the_reference_beam_search():
    """The oracle's beam search against the ids the reference's own BeamSearch produced (search_strategy.py:124-182) on the toy model of
    decode_ref.npz, beam sizes 2 and 3, without and with an LM-head bias on [EOS] that makes hypotheses end (tests/golden/decode_beam_ref.npz)."""
    import os
    import assembly_cases
    from oracle import assembly_oracle as ao
    g, P, _, meta, passages, titles = mf.load()
    here = os.path.join(os.path.dirname(__file__), "golden")
    d, b = np.load(os.path.join(here, "decode_ref.npz")), np.load(os.path.join(here, "decode_beam_ref.npz"))
    P = {k: _sharpen(k, v, float(d["reader_gain"])) for k, v in P.items()}
    corpus = ao.Corpus(passages, titles, assembly_cases.build()["group_of_doc"])
    _, _, ext, _, _ = ao.postprocess(d["query_uid"].tolist(), d["query_ids"].tolist(), d["query_len"].tolist(), d["topk_ids"].tolist(), corpus,
                                     mf.CFG["topk"], mf.CFG["seq_ret"], mf.CFG["seq"], meta["cls"], meta["sep"], meta["pad"])
    ext = torch.tensor(ext, dtype=torch.int64)
    ended = 0
    for tag, eos_bias in (("", 0.0), ("_eos", float(b["eos_bias"]))):
        P2 = dict(P)
        P2["language_model.lm_head.bias"] = P["language_model.lm_head.bias"].clone()
        P2["language_model.lm_head.bias"][int(d["eos"])] = eos_bias
        for k in (2, 3):
            outs = to.beam_decode(P2, mf.CFG, ext, mf.CFG["topk"], int(d["max_decode_len"]), int(d["bos"]), int(d["eos"]), k, float(b["alpha"]))
            ref = [[t for t in row if t >= 0] for row in b["beam%d%s" % (k, tag)].tolist()]
            assert outs == ref, (tag, k)
            ended += sum(len(r) < int(d["max_decode_len"]) for r in ref)
    assert ended >= 4                                                    # the [EOS] variants do end hypotheses early
    greedy = to.greedy_decode(P, mf.CFG, ext, mf.CFG["topk"], int(d["max_decode_len"]), int(d["bos"]), int(d["eos"]))
    assert to.beam_decode(P, mf.CFG, ext, mf.CFG["topk"], int(d["max_decode_len"]), int(d["bos"]), int(d["eos"]), 1) == greedy   # beam 1 = greedy


def test_bf16_faithful_mode_is_pinned_on_the_fp32_oracle():
    """The bf16-faithful form of the oracle (rounding where the HIP kernels store bf16) against the fp32 form -- itself pinned on the
    reference above -- on the reference fixture: logits / losses / gradients differ by bf16 round-off only, weight gradients stay fp32
    (not multiples of a bf16 ulp), and the switch leaves the fp32 form untouched."""
    g, P0, grads, meta, passages, titles = mf.load()
    ctx, typ, ext, one = mf.assembled_inputs(g, meta, passages, titles)
    qb, dec = torch.from_numpy(g["e_query_ids"]), torch.from_numpy(g["e_dec_ids"])
    labels, loss_mask = torch.from_numpy(g["e_labels"]), torch.from_numpy(g["e_loss_mask"])

    def run(bf16):
        P = {k: v.clone().requires_grad_(True) for k, v in P0.items()}
        with to.bf16_faithful(bf16):
            lm, tlp, oc = to.emdr2_forward(P, mf.CFG, qb, torch.zeros_like(qb), ~to.make_attention_mask_3d(qb, qb), ctx, typ, ext, one, dec)
            loss = to.reader_ce_loss(lm, labels, loss_mask) + to.retriever_loss_and_utility(oc, tlp, labels, loss_mask, meta["eos"])[0]
            loss.backward()
        return lm.detach(), tlp.detach(), oc, float(loss), {k: v.grad for k, v in P.items() if v.grad is not None}
    lm32, tlp32, oc32, loss32, g32 = run(False)
    lmb, tlpb, ocb, lossb, gb = run(True)
    assert not to._Mode.bf16
    np.testing.assert_allclose(lm32.numpy(), g["e_lm_logits"], rtol=1e-4, atol=1e-4)                   # fp32 form unchanged by the switch
    rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-12))
    rms = lambda a, b: float((a - b).norm() / (b.norm() + 1e-12))
    assert torch.equal(lmb, lmb.bfloat16().float()) and not torch.equal(lm32, lm32.bfloat16().float())    # outputs ARE bf16 values
    # consumed positions (a padded decoder query averages over the keys its layout keeps: the bf16 form follows the packed layout, which
    # does not store padded encoder rows; nothing reads such a row -- loss_mask / ignore_index 0)
    dreal = dec != 0
    dk = dreal[:, None, :].expand(-1, ocb.shape[1], -1)
    assert 1e-5 < rel(lmb[dreal], lm32[dreal]) < 2e-2 and rel(ocb[dk], oc32[dk]) < 2e-2 and rel(tlpb, tlp32) < 2e-2
    assert abs(lossb - loss32) < 1e-2 * abs(loss32)
    gmax = max(float(v.abs().max()) for v in g32.values())
    assert set(gb) == set(g32)
    for k in g32:
        assert float((gb[k] - g32[k]).abs().max()) < 5e-2 * max(float(g32[k].abs().max()), 1e-2 * gmax), k
        assert rms(gb[k], g32[k]) < 5e-2 or float(g32[k].abs().max()) < 1e-2 * gmax, (k, rms(gb[k], g32[k]))
    w = gb["language_model.language_model.encoder.layers.0.mlp.dense_h_to_4h.weight"]
    assert not torch.equal(w, w.bfloat16().float())                                                     # weight gradients are fp32


def test_bf16_faithful_layer_backward_is_pinned_on_the_fp32_form_at_base_size():
    """VERDICT r03 weak #2b: the bf16-faithful oracle's attention backward follows the KERNELS' formula (D = rowsum(dO o O) from the stored
    bf16 output, probabilities rebuilt from (m, l)) and was pinned on the fp32 form -- the one the reference fixture above pins -- only at
    the 2-layer toy size.  Here at BASE size (H 768, 12 heads, FFN 3072, s 512 / 32 over 512), one encoder and one decoder layer of
    tests/golden/layer_base_case.py: output, input gradients (encoder states included) and every parameter gradient of the bf16-faithful
    form stay within bf16 round-off of the fp32 form, so the checker cannot drift with the checked."""
    import layer_base_case as lb
    inp = lb.inputs()
    t = lambda k: torch.from_numpy(inp[k])
    enc_ids, dec_ids = t("enc_ids"), t("dec_ids")
    enc_real, dec_real = enc_ids != 0, dec_ids != 0
    OUT_TOL, GRAD_TOL = (1.2e-2, 6e-3), (1.5e-2, 1.2e-2)        # (max-, RMS-normalised); measured r04: 5.7e-3 / 3.3e-3 and 5.8e-3 / 6.1e-3

    def run(kind, seed, bf16, enc_out):
        P = {"L." + k: torch.from_numpy(v).requires_grad_(True) for k, v in lb.layer_params(kind, seed).items()}
        with to.bf16_faithful(bf16):
            if kind == "encoder":
                x = t("enc_x").clone().requires_grad_(True)
                y = to.transformer_layer(P, "L", lb.DIMS["heads"], x, (~to.make_attention_mask_3d(enc_ids, enc_ids))[:, None])
                (y * t("w_enc") * enc_real[..., None]).sum().backward()      # (no upstream gradient on pad rows: nothing consumes them)
                return y.detach(), {"dx": x.grad, **{"grad " + k: v.grad for k, v in P.items()}}
            x = t("dec_x").clone().requires_grad_(True)
            enc = enc_out.clone().requires_grad_(True)
            mask = (~(to.make_attention_mask_3d(dec_ids, dec_ids) * to.make_history_mask_3d(dec_ids)))[:, None]
            ed = (~to.make_attention_mask_3d(dec_ids, enc_ids))[:, None]
            y = to.transformer_layer(P, "L", lb.DIMS["heads"], x, mask, encoder_output=enc, enc_dec_mask=ed)
            (y * t("w_dec") * dec_real[..., None]).sum().backward()
            return y.detach(), {"dx": x.grad, "denc": enc.grad, **{"grad " + k: v.grad for k, v in P.items()}}
    enc32, g_enc32 = run("encoder", 11, False, None)
    encb, g_encb = run("encoder", 11, True, None)
    dec32, g_dec32 = run("decoder", 12, False, enc32)
    decb, g_decb = run("decoder", 12, True, enc32)
    rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-12))
    rms = lambda a, b: float((a - b).norm() / (b.norm() + 1e-12))
    # consumed rows only: a padded query row attends uniformly over whatever keys its layout keeps and is never read
    worst = {"out": [0.0, 0.0], "grad": [0.0, 0.0]}
    for a, b in ((encb[enc_real], enc32[enc_real]), (decb[dec_real], dec32[dec_real])):
        worst["out"] = [max(worst["out"][0], rel(a, b)), max(worst["out"][1], rms(a, b))]
    for name, gb, g32, real in (("encoder", g_encb, g_enc32, enc_real), ("decoder", g_decb, g_dec32, dec_real)):
        assert set(gb) == set(g32)
        for k in g32:
            a, b = gb[k], g32[k]
            if k == "dx":
                a, b = a[real], b[real]
            elif k == "denc":
                a, b = a[enc_real], b[enc_real]
            worst["grad"] = [max(worst["grad"][0], rel(a, b)), max(worst["grad"][1], rms(a, b))]
            assert rel(a, b) < GRAD_TOL[0] and rms(a, b) < GRAD_TOL[1], (name, k, rel(a, b), rms(a, b))
    print("bf16-faithful vs fp32 at base size: outputs max %.3g rms %.3g, gradients max %.3g rms %.3g" % tuple(worst["out"] + worst["grad"]))
    assert worst["out"][0] < OUT_TOL[0] and worst["out"][1] < OUT_TOL[1], worst
