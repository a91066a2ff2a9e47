"""oracle/transformer_oracle.py -- TEST INFRASTRUCTURE ONLY.

Plain PyTorch fp32 (CPU) restatement of the encoder / reader / loss arithmetic of the EMDR2 hot path, written
functionally over a flat {name: tensor} parameter dict that uses the reference's parameter names:

  embedding, transformer layer, stack     megatron/model/language_model.py:169-181,319-353; transformer.py:94-108,
                                          212-394 (QKV stored interleaved [np, hn, 3], KV [np, hn, 2]), 474-563, 648-699
  BERT tower -> [CLS] embedding           megatron/model/dualencoder_model.py:166-181
  T5-style reader (enc, dec, tied head)   megatron/model/t5_model.py:112-154, language_model.py:28-41
  EMDR2 forward                           megatron/model/emdr2_model.py:87-214
  EMDR2 loss                              tasks/openqa/e2eqa/train_e2eqa.py:72-123,152-181
  masks                                   megatron/data/mask_creation_utils.py:17-42
  learning rate                           megatron/learning_rates.py:51-71

Pinned by tests/golden/model_ref.npz (the reference's own modules run on CPU, tests/golden/gen_model_golden.py).
Tolerance: fp32 round-off (1e-5 relative); the HIP path is compared to this oracle at 1e-3 (fp32) / 2e-2 (bf16).
"""
import math

import torch
import torch.nn.functional as F


# ---- masks (True = masked after the `< 0.5` the reference applies) -----------------------------------------
def make_attention_mask_3d(source_block, target_block):
    """[b, s_src, s_tgt] 1 where both tokens are real (id >= 1)."""
    return ((target_block[:, None, :] >= 1) * (source_block[:, :, None] >= 1))


def make_history_mask_3d(block):
    b, s = block.shape
    ar = torch.arange(s, device=block.device)
    return (ar[None, ] <= ar[:, None])[None, ].expand(b, s, s)


def position_ids(ids):
    return torch.arange(ids.shape[1], device=ids.device)[None, :].expand_as(ids)


# ---- building blocks ----------------------------------------------------------------------------------------
def layer_norm(x, P, prefix, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), P[prefix + ".weight"], P[prefix + ".bias"], eps)


def embedding(P, prefix, ids, tokentype_ids=None):
    e = P[prefix + ".word_embeddings.weight"][ids] + P[prefix + ".position_embeddings.weight"][position_ids(ids)]
    if tokentype_ids is not None:
        e = e + P[prefix + ".tokentype_embeddings.weight"][tokentype_ids]
    return e


def attention(P, prefix, heads, x, mask, encoder_output=None):
    """x [b, sq, h]; mask bool [b, 1 or np, sq, sk] True = masked.  Self-attention if encoder_output is None."""
    b, sq, h = x.shape
    hn = h // heads
    if encoder_output is None:
        mixed = F.linear(x, P[prefix + ".query_key_value.weight"], P[prefix + ".query_key_value.bias"])
        mixed = mixed.view(b, sq, heads, hn, 3)
        q, k, v = mixed[..., 0], mixed[..., 1], mixed[..., 2]
    else:
        kv = F.linear(encoder_output, P[prefix + ".key_value.weight"], P[prefix + ".key_value.bias"])
        kv = kv.view(b, encoder_output.shape[1], heads, hn, 2)
        k, v = kv[..., 0], kv[..., 1]
        q = F.linear(x, P[prefix + ".query.weight"], P[prefix + ".query.bias"]).view(b, sq, heads, hn)
    scores = torch.einsum("bqnd,bknd->bnqk", q, k) / math.sqrt(hn)
    scores = scores.masked_fill(mask, -10000.0)
    probs = torch.softmax(scores, dim=-1)
    ctx = torch.einsum("bnqk,bknd->bqnd", probs, v).reshape(b, sq, h)
    return F.linear(ctx, P[prefix + ".dense.weight"]), P[prefix + ".dense.bias"]      # bias added by the caller (skip_bias_add)


def mlp(P, prefix, x):
    inter = F.gelu(F.linear(x, P[prefix + ".dense_h_to_4h.weight"]) + P[prefix + ".dense_h_to_4h.bias"])
    return F.linear(inter, P[prefix + ".dense_4h_to_h.weight"]), P[prefix + ".dense_4h_to_h.bias"]


def transformer_layer(P, prefix, heads, x, mask, encoder_output=None, enc_dec_mask=None):
    ln = layer_norm(x, P, prefix + ".input_layernorm")
    a, ab = attention(P, prefix + ".self_attention", heads, ln, mask)
    x = x + (a + ab)
    ln = layer_norm(x, P, prefix + ".post_attention_layernorm")
    if encoder_output is not None:
        a, ab = attention(P, prefix + ".inter_attention", heads, ln, enc_dec_mask, encoder_output=encoder_output)
        x = x + (a + ab)
        ln = layer_norm(x, P, prefix + ".post_inter_attention_layernorm")
    m, mb = mlp(P, prefix + ".mlp", ln)
    return x + (m + mb)


def transformer(P, prefix, layers, heads, x, mask, encoder_output=None, enc_dec_mask=None):
    for i in range(layers):
        x = transformer_layer(P, "%s.layers.%d" % (prefix, i), heads, x, mask, encoder_output, enc_dec_mask)
    return layer_norm(x, P, prefix + ".final_layernorm")


# ---- models ---------------------------------------------------------------------------------------------------
def bert_embed(P, prefix, cfg, ids, mask3d, types):
    """PretrainedBertModel.forward: hidden state of token 0 (no pooler).  mask3d bool [b, s, s] True = masked."""
    lm = prefix + ".language_model"
    x = embedding(P, lm + ".embedding", ids, types)
    x = transformer(P, lm + ".encoder", cfg["layers"], cfg["heads"], x, mask3d[:, None])
    return x[:, 0, :]


def t5_encode(P, prefix, cfg, enc_ids, enc_mask):
    lm = prefix + ".language_model"
    x = embedding(P, lm + ".embedding", enc_ids)
    return transformer(P, lm + ".encoder", cfg["layers"], cfg["heads"], x, enc_mask[:, None])


def t5_decode(P, prefix, cfg, dec_ids, enc_hidden, dec_mask, enc_dec_mask):
    lm = prefix + ".language_model"
    y = embedding(P, lm + ".embedding", dec_ids)
    y = transformer(P, lm + ".decoder", cfg["layers"], cfg["heads"], y, dec_mask[:, None], enc_hidden, enc_dec_mask[:, None])
    return F.linear(y, P[lm + ".embedding.word_embeddings.weight"], P[prefix + ".lm_head.bias"])


def emdr2_forward(P, cfg, query_ids_bert, query_types, query_mask, ctx_ids, ctx_types, qext_ids, qone_ids, dec_ids,
                  update_retriever=True, score_scaling=True):
    """EMDR2Model.forward in training mode after `postprocess`: ctx_ids/types [B,K,S_ret], qext/qone [B*K,S]."""
    B, K = ctx_ids.shape[:2]
    H = cfg["hidden"]
    q = bert_embed(P, "retriever_model.query_model", cfg, query_ids_bert, query_mask, query_types)
    c_ids, c_types = ctx_ids.reshape(B * K, -1), ctx_types.reshape(B * K, -1)
    c_mask = ~make_attention_mask_3d(c_ids, c_ids)
    c = bert_embed(P, "retriever_model.context_model", cfg, c_ids, c_mask, c_types).reshape(B, K, H)
    sim = torch.bmm(q[:, None, :].float(), c.float().transpose(1, 2))
    if score_scaling:
        sim = sim / math.sqrt(H)
    topk_log_probs = F.log_softmax(sim, dim=2).squeeze(1)

    enc_mask = ~make_attention_mask_3d(qext_ids, qext_ids)
    enc = t5_encode(P, "language_model", cfg, qext_ids, enc_mask).reshape(B, K * qext_ids.shape[1], H)
    unflat = qext_ids.reshape(B, -1)
    ed_mask = ~make_attention_mask_3d(dec_ids, unflat)
    d_mask = ~(make_attention_mask_3d(dec_ids, dec_ids) * make_history_mask_3d(dec_ids))
    lm_logits = t5_decode(P, "language_model", cfg, dec_ids, enc, d_mask, ed_mask)

    one = None
    if update_retriever:
        with torch.no_grad():
            dec_rep = torch.repeat_interleave(dec_ids, K, dim=0)
            e1 = t5_encode(P, "language_model", cfg, qone_ids, ~make_attention_mask_3d(qone_ids, qone_ids))
            ed1 = ~make_attention_mask_3d(dec_rep, qone_ids)
            d1 = ~(make_attention_mask_3d(dec_rep, dec_rep) * make_history_mask_3d(dec_rep))
            one = t5_decode(P, "language_model", cfg, dec_rep, e1, d1, ed1).reshape(B, K, dec_ids.shape[1], -1)
    return lm_logits, topk_log_probs, one


# ---- losses -----------------------------------------------------------------------------------------------------
def reader_ce_loss(lm_logits, labels, loss_mask):
    ce = F.cross_entropy(lm_logits.float().reshape(-1, lm_logits.shape[-1]), labels.reshape(-1), reduction="none", ignore_index=0)
    return torch.sum(ce * loss_mask.reshape(-1)) / loss_mask.sum()


def retriever_loss_and_utility(one_context_logits, topk_log_probs, labels, loss_mask, eos_id):
    logp = F.log_softmax(one_context_logits.float(), dim=-1)
    labels = labels.masked_fill(~loss_mask.to(torch.bool), 0)
    K = one_context_logits.shape[1]
    gold = torch.gather(logp, -1, labels[:, None, :, None].expand(-1, K, -1, 1)).squeeze(-1)
    marginal = torch.logsumexp(topk_log_probs.float().unsqueeze(-1) + gold, dim=1)
    loss = -torch.sum(marginal * loss_mask) / torch.sum(loss_mask)
    util_mask = loss_mask.masked_fill(labels >= eos_id, 0)
    utility = torch.sum((marginal - gold[:, -1, :]) * util_mask) / torch.sum(util_mask)
    null_loss = -torch.sum(gold[:, -1, :] * loss_mask) / torch.sum(loss_mask)
    return loss, utility, null_loss


def retriever_kl_div_loss(one_context_logits, topk_log_probs, labels, loss_mask):
    """--ret-kldiv variant (train_e2eqa.py:184-214): KL(teacher || retriever prior), teacher = softmax over K of the length-normalised
    gold log-likelihood under each single-context reader pass."""
    logp = F.log_softmax(one_context_logits.float(), dim=-1)
    labels = labels.masked_fill(~loss_mask.to(torch.bool), 0)
    K = one_context_logits.shape[1]
    gold = torch.gather(logp, -1, labels[:, None, :, None].expand(-1, K, -1, 1)).squeeze(-1)
    teacher_log = torch.sum(gold * loss_mask.unsqueeze(1), dim=2) / torch.sum(loss_mask.unsqueeze(1), dim=2)
    return F.kl_div(topk_log_probs.float(), torch.softmax(teacher_log, dim=1), reduction='batchmean')


def annealing_lr(num_iters, start_lr, warmup_iter, end_iter, min_lr=0.0):
    """AnnealingLR.get_lr with decay_style 'linear' (learning_rates.py:51-71), incl. its clamp quirk."""
    n_ = min(num_iters, end_iter - warmup_iter)
    if warmup_iter > 0 and num_iters <= warmup_iter:
        return float(start_lr) * n_ / warmup_iter
    n_ = n_ - warmup_iter
    return max(start_lr * (end_iter - n_) / end_iter, min_lr)


# ---- random parameters with the reference's names (bench.py cpu_baseline; shapes: language_model.py:98-181, transformer.py:58-563) --------
def random_params(cfg, bert_vocab, t5_vocab, max_pos=512, std=0.02, seed=1234):
    """Flat {name: fp32 tensor} for the two BERT towers and the reader, random-normal(0, std) weights / zero biases / unit LayerNorm gains,
    cfg = dict(layers, hidden, heads, ffn)."""
    g = torch.Generator().manual_seed(seed)
    H, F_, L = cfg["hidden"], cfg["ffn"], cfg["layers"]
    P = {}

    def w(name, *shape):
        P[name] = torch.randn(shape, generator=g) * std

    def lin(prefix, n_out, n_in):
        w(prefix + ".weight", n_out, n_in)
        P[prefix + ".bias"] = torch.zeros(n_out)

    def ln(prefix):
        P[prefix + ".weight"] = torch.ones(H)
        P[prefix + ".bias"] = torch.zeros(H)

    def stack(prefix, decoder):
        for i in range(L):
            lp = "%s.layers.%d" % (prefix, i)
            ln(lp + ".input_layernorm")
            lin(lp + ".self_attention.query_key_value", 3 * H, H)
            lin(lp + ".self_attention.dense", H, H)
            ln(lp + ".post_attention_layernorm")
            if decoder:
                lin(lp + ".inter_attention.query", H, H)
                lin(lp + ".inter_attention.key_value", 2 * H, H)
                lin(lp + ".inter_attention.dense", H, H)
                ln(lp + ".post_inter_attention_layernorm")
            lin(lp + ".mlp.dense_h_to_4h", F_, H)
            lin(lp + ".mlp.dense_4h_to_h", H, F_)
        ln(prefix + ".final_layernorm")

    def lm(prefix, vocab, decoder):
        w(prefix + ".embedding.word_embeddings.weight", vocab, H)
        w(prefix + ".embedding.position_embeddings.weight", max_pos, H)
        w(prefix + ".embedding.tokentype_embeddings.weight", 2, H)
        stack(prefix + ".encoder", False)
        if decoder:
            stack(prefix + ".decoder", True)

    lm("retriever_model.query_model.language_model", bert_vocab, False)
    lm("retriever_model.context_model.language_model", bert_vocab, False)
    lm("language_model.language_model", t5_vocab, True)
    P["language_model.lm_head.bias"] = torch.zeros(t5_vocab)
    return P


# ---- greedy answer generation (megatron/model/search_strategy.py:185-240, SampleOrGreedySearch with sample=False) ------------------------------
def greedy_decode(P, cfg, qext_ids, K, max_decode_len, bos_id, eos_id, return_margins=False):
    """Encode the K retrieved passages of every question once (EMDR2Model.forward, eval mode, emdr2_model.py:148-183), then decode token by
    token: each step re-runs the decoder on the whole prefix and takes the arg-max of the last position; stops when every question has
    produced [EOS]; answers are cut at their first [EOS], an empty answer becomes [1].  qext_ids [B*K, S]."""
    B = qext_ids.shape[0] // K
    H = cfg["hidden"]
    with torch.no_grad():
        enc = t5_encode(P, "language_model", cfg, qext_ids, ~make_attention_mask_3d(qext_ids, qext_ids)).reshape(B, -1, H)
        unflat = qext_ids.reshape(B, -1)
        y = torch.full((B, 1), bos_id, dtype=torch.int64)
        eos = torch.zeros(B, dtype=torch.int64)
        steps, margins = [], []
        for _ in range(max_decode_len):
            d_mask = ~(make_attention_mask_3d(y, y) * make_history_mask_3d(y))
            logits = t5_decode(P, "language_model", cfg, y, enc, d_mask, ~make_attention_mask_3d(y, unflat))[:, -1, :].float()
            top2 = torch.topk(logits, 2, dim=1)
            ys = top2.indices[:, 0]
            margins.append(top2.values[:, 0] - top2.values[:, 1])
            y = torch.cat([y, ys[:, None]], dim=1)
            steps.append(ys)
            eos += (ys == eos_id).long()
            if bool((eos > 0).all()):
                break
    out = []
    for row in torch.stack(steps, 1).tolist():
        if eos_id in row:
            row = row[:row.index(eos_id)]
        out.append(row if row else [1])
    return (out, torch.stack(margins, 1)) if return_margins else out
