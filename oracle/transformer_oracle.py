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
Tolerance: fp32 round-off (1e-5 relative); the HIP path is compared to this oracle at 2e-2 (bf16).

bf16-faithful mode (`with bf16_faithful():`): the same arithmetic with a round-to-nearest-even to bfloat16 at every point where the HIP
kernels store bfloat16 -- the bf16 working copies of the weights, every activation a kernel writes (embedding sum, LayerNorm output,
GEMM epilogue before AND after the residual add, GELU output, attention probabilities as they enter the P V product, attention output,
logits) -- and fp32 everywhere the kernels keep fp32 (accumulators, biases, LayerNorm statistics and gains, softmax sums, losses).
Activation gradients are rounded at the same points on their way back (the kernels hand bf16 gradients from op to op); weight gradients
stay fp32 (the weight-gradient GEMMs write fp32).  What is left between this mode and the HIP path is accumulation order, the hardware's
exp / erf, and the online softmax's reference point: a few bf16 ulps on isolated elements, which is what lets the parity tests run at
~3e-3 instead of 2e-2 (tests/test_parity_bf16_gpu.py).  The mode itself is pinned against the fp32 form (tests/test_oracle_transformer.py).
"""
import contextlib
import math

import torch
import torch.nn.functional as F


# ---- bf16-faithful mode ---------------------------------------------------------------------------------------------------
class _Mode(object):
    bf16 = False


@contextlib.contextmanager
def bf16_faithful(on=True):
    old, _Mode.bf16 = _Mode.bf16, bool(on)
    try:
        yield
    finally:
        _Mode.bf16 = old


def _act(x):
    """An activation a kernel stores as bf16 (its gradient comes back through the same rounding: the cast's own backward)."""
    return x.to(torch.bfloat16).to(torch.float32) if _Mode.bf16 else x


class _RoundValueOnly(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


def _w(p):
    """The bf16 working copy of an fp32 master parameter; its gradient stays fp32."""
    return _RoundValueOnly.apply(p) if _Mode.bf16 else p


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
    return _act(F.layer_norm(x, (x.shape[-1],), P[prefix + ".weight"], P[prefix + ".bias"], eps))


def embedding(P, prefix, ids, tokentype_ids=None):
    e = _w(P[prefix + ".word_embeddings.weight"])[ids] + _w(P[prefix + ".position_embeddings.weight"])[position_ids(ids)]
    if tokentype_ids is not None:
        e = e + _w(P[prefix + ".tokentype_embeddings.weight"])[tokentype_ids]
    return _act(e)


def attention(P, prefix, heads, x, mask, encoder_output=None):
    """x [b, sq, h]; mask bool [b, 1 or np, sq, sk] True = masked.  Self-attention if encoder_output is None."""
    b, sq, h = x.shape
    hn = h // heads
    if encoder_output is None:
        mixed = _act(F.linear(x, _w(P[prefix + ".query_key_value.weight"]), P[prefix + ".query_key_value.bias"]))
        mixed = mixed.view(b, sq, heads, hn, 3)
        q, k, v = mixed[..., 0], mixed[..., 1], mixed[..., 2]
    else:
        kv = _act(F.linear(encoder_output, _w(P[prefix + ".key_value.weight"]), P[prefix + ".key_value.bias"]))
        kv = kv.view(b, encoder_output.shape[1], heads, hn, 2)
        k, v = kv[..., 0], kv[..., 1]
        q = _act(F.linear(x, _w(P[prefix + ".query.weight"]), P[prefix + ".query.bias"])).view(b, sq, heads, hn)
    scores = torch.einsum("bqnd,bknd->bnqk", q, k) / math.sqrt(hn)
    scores = scores.masked_fill(mask, -10000.0)
    if _Mode.bf16:
        ctx = _FusedAttentionBF16.apply(q, k, v, mask, 1.0 / math.sqrt(hn)).reshape(b, sq, h)
    else:
        probs = torch.softmax(scores, dim=-1)
        ctx = torch.einsum("bnqk,bknd->bqnd", probs, v).reshape(b, sq, h)
    return F.linear(ctx, _w(P[prefix + ".dense.weight"])), P[prefix + ".dense.bias"]    # bias added by the caller (skip_bias_add)


_L2E = 1.4426950408889634


def _attention_constants(scale):
    f32 = torch.float32
    sc = float(torch.tensor(scale, dtype=f32) * torch.tensor(_L2E, dtype=f32))
    masked2 = float(torch.tensor(-10000.0, dtype=f32) * torch.tensor(_L2E, dtype=f32))
    return sc, masked2


def _kept_keys(mask_i):
    """Keys some query of this batch element attends to (the packed layout does not store the others: padding)."""
    keep = (~mask_i).any(dim=0).any(dim=0)
    if not bool(keep.any()):
        keep = torch.ones_like(keep)
    return torch.nonzero(keep).squeeze(1)


def _attention_probs_v_bf16(raw, mask, v, scale):
    """softmax(mask(raw * scale)) V the way the fused forward kernel rounds it (csrc/attention.hip).  The probabilities enter the P V product
    as bf16( exp2(s2 - m) ), s2 = score in log2 units, m = the kernel's LAZY running maximum: keys are consumed in steps of 32, and the
    reference point m of a wave's 32 queries moves (to ceil(max(m, step maximum)), per query) only in a step where some query of the wave exceeds
    its m by more than 8 -- so which bf16 a probability rounds to depends on that sequence, and this function walks it.  The normaliser is
    the fp32 sum of the UNROUNDED exponentials, applied as a reciprocal to the fp32 product.  Keys that no query may see (padding) are not
    part of the sequence: the packed layout does not store them, and a trailing run of them changes nothing in the dense layout either.
    raw [b, np, sq, sk] = q k^T (fp32), mask bool [b, np, sq, sk] True = masked, v [b, sk, np, hn]
    -> out [b, sq, np, hn] (bf16 values), m, l [b, np, sq] (log2-domain reference point and normaliser, as the kernel hands them to its backward)."""
    b, heads, sq, sk = raw.shape
    f32 = torch.float32
    sc, masked2 = _attention_constants(scale)
    nwave = (sq + 31) // 32
    outs, ms, ls = [], [], []
    for i in range(b):
        idx = _kept_keys(mask[i])
        s2 = torch.where(mask[i][:, :, idx], torch.full((), masked2, dtype=f32), raw[i][:, :, idx] * sc)        # [np, sq, nk]
        vi = v[i][idx]                                                   # [nk, np, hn]
        nk = idx.numel()
        m = torch.full((heads, sq), -3.0e38, dtype=f32)
        l = torch.zeros((heads, sq), dtype=f32)
        o = torch.zeros((heads, sq, v.shape[-1]), dtype=f32)
        for k0 in range(0, nk, 32):
            sj = s2[:, :, k0:k0 + 32]
            bmax = sj.amax(dim=-1)
            trig = bmax > m + 8.0                                        # per query; the kernel decides per wave of 32 queries
            tw = F.pad(trig, (0, nwave * 32 - sq)).reshape(heads, nwave, 32).any(dim=-1, keepdim=True).expand(-1, -1, 32)
            tw = tw.reshape(heads, nwave * 32)[:, :sq]
            mnew = torch.where(tw, torch.ceil(torch.maximum(m, bmax)), m)     # (r05: a whole number of binades, csrc/attention.hip)
            alpha = torch.exp2(m - mnew)
            l, o, m = l * alpha, o * alpha[..., None], mnew
            pj = torch.exp2(sj - m[..., None])
            l = l + pj.sum(dim=-1)
            o = o + torch.einsum("nqk,knd->nqd", _act(pj), vi[k0:k0 + 32])
        outs.append(_act(o * (1.0 / l)[..., None]).transpose(0, 1))     # [sq, np, hn]
        ms.append(m)
        ls.append(l)
    return torch.stack(outs, 0), torch.stack(ms, 0), torch.stack(ls, 0)


class _FusedAttentionBF16(torch.autograd.Function):
    """The fused attention of the HIP path in its own arithmetic, forward (above) AND backward (csrc/attention_bwd.hip), so that the
    gradient noise of the bf16-faithful oracle is the kernels' gradient noise:
      P = exp2(s2 - (m + log2 l)) rebuilt from the forward's statistics;  dP = dO V^T;  D = rowsum(dO o O) with the STORED (bf16) output;
      dS = P o (dP - D), zero at masked positions, rounded to bf16 as the operand of  dQ = scale dS K,  dK = scale dS^T Q;  dV = bf16(P)^T dO;
      dQ, dK, dV stored as bf16.
    q [b, sq, np, hn], k, v [b, sk, np, hn] (bf16 values in fp32 tensors), mask bool [b, 1 or np, sq, sk] True = masked."""

    @staticmethod
    def forward(ctx, q, k, v, mask, scale):
        b, sq, heads, hn = q.shape
        mask = mask.expand(b, heads, sq, k.shape[1])
        raw = torch.einsum("bqnd,bknd->bnqk", q, k)
        out, m, l = _attention_probs_v_bf16(raw, mask, v, scale)
        ctx.save_for_backward(q, k, v, mask, out, m, l)
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, mask, out, m, l = ctx.saved_tensors
        f32 = torch.float32
        rnd = lambda x: x.to(torch.bfloat16).to(f32)
        sc, masked2 = _attention_constants(ctx.scale)
        dout = rnd(dout)                                                 # the incoming gradient is a bf16 tensor on the HIP path
        dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
        for i in range(q.shape[0]):
            idx = _kept_keys(mask[i])
            mi = mask[i][:, :, idx]                                      # [np, sq, nk]
            ki, vi = k[i][idx], v[i][idx]                                # [nk, np, hn]
            raw = torch.einsum("qnd,knd->nqk", q[i], ki)
            pm = m[i] + torch.log2(l[i])
            p = torch.exp2(torch.where(mi, torch.full((), masked2, dtype=f32), raw * sc) - pm[..., None])
            dp = torch.einsum("qnd,knd->nqk", dout[i], vi)
            d = (dout[i] * out[i]).sum(dim=-1).transpose(0, 1)           # [np, sq]
            ds = rnd(torch.where(mi, torch.zeros((), dtype=f32), p * (dp - d[..., None])))
            dq[i] = rnd(torch.einsum("nqk,knd->qnd", ds, ki) * ctx.scale)
            dk[i][idx] = rnd(torch.einsum("nqk,qnd->knd", ds, q[i]) * ctx.scale)
            dv[i][idx] = rnd(torch.einsum("nqk,qnd->knd", rnd(p), dout[i]))
        return dq, dk, dv, None, None


def mlp(P, prefix, x):
    inter = _act(F.gelu(F.linear(x, _w(P[prefix + ".dense_h_to_4h.weight"])) + P[prefix + ".dense_h_to_4h.bias"]))
    return F.linear(inter, _w(P[prefix + ".dense_4h_to_h.weight"])), P[prefix + ".dense_4h_to_h.bias"]


def _bias_add_residual(y, bias, x):
    """bias-dropout-add at p = 0 (transformer.py:397-413).  bf16 mode: the GEMM epilogue rounds acc + bias on its way through the LDS staging
    buffer and again after adding the residual row (csrc/gemm8.hip)."""
    return _act(x + _act(y + bias))


def transformer_layer(P, prefix, heads, x, mask, encoder_output=None, enc_dec_mask=None):
    ln = layer_norm(x, P, prefix + ".input_layernorm")
    a, ab = attention(P, prefix + ".self_attention", heads, ln, mask)
    x = _bias_add_residual(a, ab, x)
    ln = layer_norm(x, P, prefix + ".post_attention_layernorm")
    if encoder_output is not None:
        a, ab = attention(P, prefix + ".inter_attention", heads, ln, enc_dec_mask, encoder_output=encoder_output)
        x = _bias_add_residual(a, ab, x)
        ln = layer_norm(x, P, prefix + ".post_inter_attention_layernorm")
    m, mb = mlp(P, prefix + ".mlp", ln)
    return _bias_add_residual(m, mb, x)


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
    return _act(F.linear(y, _w(P[lm + ".embedding.word_embeddings.weight"]), P[prefix + ".lm_head.bias"]))


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


# ---- beam search (megatron/model/search_strategy.py:124-182 `BeamSearch`; scores :20-41, beam bookkeeping :44-101, final pick :104-121) ---------
def _length_penalty(n, alpha):
    """PolynomialNormalization.lp (search_strategy.py:27-28): ((5 + n) / 6) ** alpha."""
    return pow(5 + n, alpha) / pow(5 + 1, alpha)


def beam_decode(P, cfg, qext_ids, K, max_decode_len, bos_id, eos_id, beam, alpha=0.6, return_gaps=False):
    """Encode once, then grow `beam` hypotheses per question, one token per step (at most max_decode_len steps).
      step 0      the `beam` best first tokens by log-probability; their scores are the log-probabilities;
      step t > 0  every live hypothesis proposes its `beam` best continuations with the length-normalised score
                  (score * lp(n - 1) + logp) / lp(n), n = tokens held incl. [BOS]; a hypothesis that already holds [EOS] proposes itself
                  once with its score unchanged (and `beam` - 1 fillers at score - 10000), its next token is [EOS] again; the `beam` best
                  of the beam * beam proposals of a question survive, best first;
      stop        when every hypothesis holds [EOS], or after max_decode_len steps;
      answer      the best-scoring hypothesis of a question (the first one on equal scores) without [BOS], cut before its first [EOS]
                  (it may be empty: unlike the greedy decoder, search_strategy.py:104-121 does not substitute [1]).
    With return_gaps: per question the smallest distance, over all steps, between the worst surviving proposal and the best rejected one
    (taken over ALL beam * V continuations, not only the proposed ones) and the final best-vs-second distance -- what a reduced-precision
    decoder must stay inside to be asked for the same answer.  qext_ids [B*K, S]."""
    B = qext_ids.shape[0] // K
    H = cfg["hidden"]
    with torch.no_grad():
        enc = t5_encode(P, "language_model", cfg, qext_ids, ~make_attention_mask_3d(qext_ids, qext_ids)).reshape(B, -1, H)
        unflat = qext_ids.reshape(B, -1)

        def logp_last(y, e, u):
            d_mask = ~(make_attention_mask_3d(y, y) * make_history_mask_3d(y))
            lg = t5_decode(P, "language_model", cfg, y, e, d_mask, ~make_attention_mask_3d(y, u))[:, -1, :].float()
            return torch.log_softmax(lg, dim=1)

        y = torch.full((B, 1), bos_id, dtype=torch.int64)
        lp0 = logp_last(y, enc, unflat)                                             # [B, V]
        top = torch.topk(lp0, beam + 1, dim=1)
        gaps = (top.values[:, beam - 1] - top.values[:, beam]).clone()
        total = top.values[:, :beam].reshape(-1)                                    # [B * beam], best first
        outs = torch.cat([torch.full((B * beam, 1), bos_id, dtype=torch.int64), top.indices[:, :beam].reshape(-1, 1)], dim=1)
        enc, unflat = enc.repeat_interleave(beam, 0), unflat.repeat_interleave(beam, 0)
        base = (torch.arange(B) * beam)[:, None]
        for _ in range(1, max_decode_len):
            if bool((outs == eos_id).any(1).all()):
                break
            n = outs.shape[1]
            lpv = logp_last(outs, enc, unflat)                                      # [B * beam, V]
            ended = (outs == eos_id).any(1)
            live_all = (total[:, None] * _length_penalty(n - 1, alpha) + lpv) / _length_penalty(n, alpha)
            sc, tok = torch.topk(lpv, beam, dim=1)
            cand = (total[:, None] * _length_penalty(n - 1, alpha) + sc) / _length_penalty(n, alpha)
            filler = torch.zeros_like(sc); filler[:, 1:] = -10000.0
            cand = torch.where(ended[:, None], total[:, None] + filler, cand)
            tok = torch.where(ended[:, None], torch.full_like(tok, eos_id), tok)
            best, arg = torch.topk(cand.view(B, beam * beam), beam, dim=1)
            # the rejected proposals' best, over every continuation of every live hypothesis and the single proposal of every ended one
            full = torch.where(ended[:, None], torch.full_like(live_all, -3.0e38), live_all)
            full[:, 0] = torch.where(ended, total, full[:, 0])
            allv = torch.topk(full.view(B, -1), beam + 1, dim=1).values
            gaps = torch.minimum(gaps, allv[:, beam - 1] - allv[:, beam])
            parent = (arg // beam + base).reshape(-1)
            total = best.reshape(-1)
            outs = torch.cat([outs[parent], tok.view(B, beam * beam).gather(1, arg).reshape(-1, 1)], dim=1)
        total = total.view(B, beam)
        pick = torch.argmax(total, dim=1)                                           # first maximum
        if beam > 1:
            t2 = torch.topk(total, 2, dim=1).values
            gaps = torch.minimum(gaps, t2[:, 0] - t2[:, 1])
    answers = []
    for q in range(B):
        row = outs[q * beam + int(pick[q]), 1:].tolist()
        if eos_id in row:
            row = row[:row.index(eos_id)]
        answers.append(row)
    return (answers, gaps) if return_gaps else answers
