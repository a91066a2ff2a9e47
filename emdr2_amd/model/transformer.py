"""Encoder / reader modules of the EMDR2 hot path on the HIP kernels of include/emdr2_ops.h, with the reference's module
tree and parameter names so its checkpoints load unchanged (`state_dict` keys identical to the reference's):

  ParallelMLP, ParallelAttention, ParallelTransformerLayer, ParallelTransformer   megatron/model/transformer.py:58-699
  Embedding, TransformerLanguageModel                                             megatron/model/language_model.py:98-358
  PretrainedBertModel, DualEncoderModel                                           megatron/model/dualencoder_model.py:40-194
  T5LMHead, T5Model                                                               megatron/model/t5_model.py:54-154

Differences that are deliberate: activations are [b, s, h] bf16 (the reference: [s, b, h] fp16); parameters are fp32 masters whose
bf16 working copies are refreshed after each optimizer step; attention masks are not materialised (the kernels derive them from token
ids: pad id 0, plus the history mask for decoder self-attention) - `forward` therefore takes token ids where the reference takes masks;
tensor model parallelism is not built (asserted 1 in the reference, dualencoder_model.py:15).  Dropout (embedding, attention probabilities,
bias-dropout-add; active only in `.train()` mode) is counter-based inside the kernels (csrc/rng.h, kernels.DROPOUT): masks are regenerated
in the backward / activation recompute, never stored.

Packed layout (kernels.PACKING, csrc/seqpack.hip): the encoder stacks run over [rows, h] = only the real tokens of their sequences,
stored back to back (`kernels.PackedSeqs`); `ids` arguments of the layers are then a PackedSeqs instead of a [b, s] id tensor.  The dense
entry points (`encode`, `forward`) keep the reference's shapes: they unpack (zeros at pad rows) on the way out.
"""
import math

import torch

from emdr2_amd.model import kernels as K


def _normal(shape, std, gen=None):
    return torch.nn.Parameter(torch.empty(shape, dtype=torch.float32, device="cuda").normal_(0.0, std, generator=gen))


class Config(object):
    """The architecture flags of megatron/arguments.py that the hot path reads."""

    def __init__(self, num_layers=12, hidden_size=768, num_attention_heads=12, ffn_hidden_size=3072, max_position_embeddings=512,
                 layernorm_epsilon=1e-5, init_method_std=0.02, hidden_dropout=0.0, attention_dropout=0.0, compute_dtype="bf16"):
        """compute_dtype: "bf16" (the product: bf16 operands and activations, fp32 accumulation / statistics / masters -- what `--fp16` of the
        shipped scripts maps to) or "fp32" (VALIDATION ONLY: the reference's arithmetic without --fp16, megatron/training.py:55-56, on the
        fp32 matrix cores; slow, dense layouts, no dropout -- kernels_f32.py, tests/test_parity_fp32_gpu.py)."""
        if not (0.0 <= hidden_dropout < 1.0 and 0.0 <= attention_dropout < 1.0):
            raise ValueError("dropout probabilities must be in [0, 1)")
        if compute_dtype not in ("bf16", "fp32"):
            raise ValueError("compute_dtype must be 'bf16' or 'fp32'")
        self.compute_dtype = compute_dtype
        self.hidden_dropout, self.attention_dropout = float(hidden_dropout), float(attention_dropout)
        self.num_layers, self.hidden_size, self.num_attention_heads = num_layers, hidden_size, num_attention_heads
        self.ffn_hidden_size, self.max_position_embeddings = ffn_hidden_size, max_position_embeddings
        self.layernorm_epsilon, self.init_method_std = layernorm_epsilon, init_method_std
        self.kv_channels = hidden_size // num_attention_heads


class Linear(torch.nn.Module):
    def __init__(self, n_in, n_out, std):
        super().__init__()
        self.weight = _normal((n_out, n_in), std)
        self.bias = torch.nn.Parameter(torch.zeros(n_out, dtype=torch.float32, device="cuda"))


class LayerNorm(torch.nn.Module):
    def __init__(self, h, eps):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.ones(h, dtype=torch.float32, device="cuda"))
        self.bias = torch.nn.Parameter(torch.zeros(h, dtype=torch.float32, device="cuda"))
        self.eps = eps

    def forward(self, x, with_residual=False):
        if with_residual:
            return K.layer_norm_residual(x, self.weight, self.bias, self.eps)
        return K.layer_norm(x, self.weight, self.bias, self.eps)


def _deinterleave_perm(heads, hn, parts, device):
    """row r' = t*h + n*hn + d of the de-interleaved weight  <-  row n*hn*parts + d*parts + t of the checkpoint layout
    (transformer.py:231-240: the projection output is viewed as [..., np, hn, 3])."""
    t, n, d = torch.meshgrid(torch.arange(parts), torch.arange(heads), torch.arange(hn), indexing="ij")
    return (n * hn * parts + d * parts + t).reshape(-1).to(device)


class ParallelMLP(torch.nn.Module):
    def __init__(self, cfg, out_std):
        super().__init__()
        self.dense_h_to_4h = Linear(cfg.hidden_size, cfg.ffn_hidden_size, cfg.init_method_std)
        self.dense_4h_to_h = Linear(cfg.ffn_hidden_size, cfg.hidden_size, out_std)
        self.hidden_dropout, self._site = cfg.hidden_dropout, K.DROPOUT.new_site()

    def forward(self, x, residual):
        p = self.hidden_dropout if self.training else 0.0
        # both linears, bias + erf-GELU, bias-dropout-add and (in the backward) gelu' in GEMM epilogues: one autograd node
        return K.mlp(x, self.dense_h_to_4h.weight, self.dense_h_to_4h.bias, self.dense_4h_to_h.weight, self.dense_4h_to_h.bias, residual,
                     drop_p=p, seed=K.DROPOUT.seed(self._site) if p else 0)


class ParallelAttention(torch.nn.Module):
    def __init__(self, cfg, out_std, attention_type="self"):
        super().__init__()
        h = cfg.hidden_size
        self.heads, self.hn, self.attention_type = cfg.num_attention_heads, cfg.kv_channels, attention_type
        if attention_type == "self":
            self.query_key_value = Linear(h, 3 * h, cfg.init_method_std)
            self.register_buffer("_perm", _deinterleave_perm(self.heads, self.hn, 3, "cuda"), persistent=False)
        else:
            self.query = Linear(h, h, cfg.init_method_std)
            self.key_value = Linear(h, 2 * h, cfg.init_method_std)
            self.register_buffer("_perm", _deinterleave_perm(self.heads, self.hn, 2, "cuda"), persistent=False)
        self.dense = Linear(h, h, out_std)
        self.hidden_dropout, self.attention_dropout = cfg.hidden_dropout, cfg.attention_dropout
        self._site_attn, self._site_out = K.DROPOUT.new_site(), K.DROPOUT.new_site()
        self.kv_cache = None            # (encoder_output, kv): set by `cross_kv_cache` during greedy decoding

    def forward(self, x, ids_q, ids_k, causal, residual, encoder_output=None):
        """x [b, sq, h] over dense ids, or [rows, h] over a PackedSeqs (likewise encoder_output / ids_k for cross-attention)."""
        lead = tuple(x.shape[:-1])
        pa = self.attention_dropout if self.training else 0.0
        ph = self.hidden_dropout if self.training else 0.0
        seed = K.DROPOUT.seed(self._site_attn) if pa else 0
        if self.attention_type == "self":
            mixed = K.linear(x, self.query_key_value.weight, self.query_key_value.bias, row_perm=self._perm)
            return self.attend(mixed, ids_q, ids_k, causal, residual)
        else:
            if self.kv_cache is not None and self.kv_cache[0] is encoder_output:
                kv = self.kv_cache[1]                                                     # K/V of the 25,600 encoder tokens, projected once
            else:
                kv = K.linear(encoder_output, self.key_value.weight, self.key_value.bias, row_perm=self._perm).view(
                    tuple(encoder_output.shape[:-1]) + (2, self.heads, self.hn))
                if self.kv_cache is not None:
                    self.kv_cache = (encoder_output, kv)
            q = K.linear(x, self.query.weight, self.query.bias).view(lead + (self.heads, self.hn))
            ctx = K.attention_core(q, kv, ids_q, ids_k, causal, drop_p=pa, seed=seed, site=self._site_attn).view(x.shape)
        return K.linear(ctx, self.dense.weight, self.dense.bias, residual=residual, drop_p=ph, seed=K.DROPOUT.seed(self._site_out) if ph else 0)


    def attend(self, mixed, ids_q, ids_k, causal, residual):
        """Self-attention from the packed QKV projection `mixed` [..., 3 h] on: fused attention, output projection, bias-dropout-add."""
        lead = tuple(mixed.shape[:-1])
        pa = self.attention_dropout if self.training else 0.0
        ph = self.hidden_dropout if self.training else 0.0
        ctx = K.attention_core(mixed.view(lead + (3, self.heads, self.hn)), None, ids_q, ids_k, causal, drop_p=pa,
                               seed=K.DROPOUT.seed(self._site_attn) if pa else 0, site=self._site_attn).view(lead + (self.heads * self.hn,))
        return K.linear(ctx, self.dense.weight, self.dense.bias, residual=residual, drop_p=ph, seed=K.DROPOUT.seed(self._site_out) if ph else 0)

    def step_self(self, x1, ids_cur, ids_block, pos, cache, residual):
        """One decoding step of SELF-attention with a K/V cache (the reference's `layer_past` / `get_key_value` plumbing,
        transformer.py:273-280,321-332): x1 [b, 1, h] is the hidden state at position `pos`; its K and V are written into
        cache [b, Lc, 2, np, hn] and the query attends to the whole cache -- positions behind `pos` hold token id 0 in `ids_block` [b, Lc]
        and are masked like padding, which IS the history mask for the newest position."""
        b = x1.shape[0]
        mixed = K.linear(x1, self.query_key_value.weight, self.query_key_value.bias, row_perm=self._perm).view(b, 1, 3, self.heads, self.hn)
        cache[:, pos, 0] = mixed[:, 0, 1]
        cache[:, pos, 1] = mixed[:, 0, 2]
        ctx = K.attention_core(mixed[:, :, 0], cache, ids_cur, ids_block, False).view(b, 1, self.heads * self.hn)
        return K.linear(ctx, self.dense.weight, self.dense.bias, residual=residual)


class ParallelTransformerLayer(torch.nn.Module):
    def __init__(self, cfg, out_std, layer_type="encoder"):
        super().__init__()
        self.layer_type = layer_type
        self.input_layernorm = LayerNorm(cfg.hidden_size, cfg.layernorm_epsilon)
        self.self_attention = ParallelAttention(cfg, out_std, "self")
        self.post_attention_layernorm = LayerNorm(cfg.hidden_size, cfg.layernorm_epsilon)
        if layer_type == "decoder":
            self.inter_attention = ParallelAttention(cfg, out_std, "cross")
            self.post_inter_attention_layernorm = LayerNorm(cfg.hidden_size, cfg.layernorm_epsilon)
        self.mlp = ParallelMLP(cfg, out_std)

    def forward_selective(self, x, ids):
        """The encoder layer with SELECTIVE activation retention (no per-layer checkpoint around it): what stays in HBM for the backward is
        the layer input, the packed QKV projection, the attention output (+ its row statistics) and the attention block's output -- 6
        [tokens, h] tensors instead of ~16; both LayerNorm outputs and the FFN's pre-activation / GELU output (9 of the 16) are rebuilt in
        the backward by the LayerNorm kernel and one h -> 4h GEMM (kernels.LNLinearFn, kernels.LNMLPFn).  Same kernels, same dropout
        streams, same results as `forward`."""
        att, mlp = self.self_attention, self.mlp
        ln1, ln2 = self.input_layernorm, self.post_attention_layernorm
        mixed, res = K.ln_linear(x, ln1.weight, ln1.bias, ln1.eps, att.query_key_value.weight, att.query_key_value.bias, att._perm)
        x = att.attend(mixed, ids, ids, False, residual=res)
        p = mlp.hidden_dropout if self.training else 0.0
        return K.ln_mlp(x, ln2.weight, ln2.bias, ln2.eps, mlp.dense_h_to_4h.weight, mlp.dense_h_to_4h.bias, mlp.dense_4h_to_h.weight,
                        mlp.dense_4h_to_h.bias, drop_p=p, seed=K.DROPOUT.seed(mlp._site) if p else 0)

    def forward(self, x, ids, causal, encoder_output=None, enc_ids=None):
        # every LayerNorm hands the residual stream through, so the stream's gradient is folded into the LayerNorm backward kernel
        ln, x = self.input_layernorm(x, with_residual=True)
        x = self.self_attention(ln, ids, ids, causal, residual=x)
        ln, x = self.post_attention_layernorm(x, with_residual=True)
        if self.layer_type == "decoder":
            x = self.inter_attention(ln, ids, enc_ids, False, residual=x, encoder_output=encoder_output)
            ln, x = self.post_inter_attention_layernorm(x, with_residual=True)
        return self.mlp(ln, residual=x)


    def step(self, x1, ids_cur, ids_block, pos, cache, encoder_output, enc_ids):
        """Decoder layer for ONE new position (no-grad decoding): cached self-attention, cross-attention over the cached K/V projection of
        the encoder output (`cross_kv_cache`), MLP."""
        ln, x = self.input_layernorm(x1, with_residual=True)
        x = self.self_attention.step_self(ln, ids_cur, ids_block, pos, cache, residual=x)
        ln, x = self.post_attention_layernorm(x, with_residual=True)
        x = self.inter_attention(ln, ids_cur, enc_ids, False, residual=x, encoder_output=encoder_output)
        ln, x = self.post_inter_attention_layernorm(x, with_residual=True)
        return self.mlp(ln, residual=x)


class _CheckpointedLayer(object):
    """The callable handed to torch.utils.checkpoint for ONE layer invocation: its first call is the forward proper, its second the
    re-run inside the backward.  It switches the attention stash (kernels.ATTN_STASH) between storing and consuming accordingly."""

    def __init__(self, layer):
        self.layer, self.calls = layer, 0

    def __call__(self, *args):
        self.calls += 1
        K.ATTN_STASH.mode, K.ATTN_STASH.key = ('store' if self.calls == 1 else 'consume'), id(self)
        try:
            return self.layer(*args)
        finally:
            K.ATTN_STASH.mode = K.ATTN_STASH.key = None


class ParallelTransformer(torch.nn.Module):
    def __init__(self, cfg, layer_type="encoder", checkpoint_activations=False):
        super().__init__()
        out_std = cfg.init_method_std / math.sqrt(2.0 * cfg.num_layers)                   # scaled_init_method_normal (model/utils.py)
        self.layers = torch.nn.ModuleList([ParallelTransformerLayer(cfg, out_std, layer_type) for _ in range(cfg.num_layers)])
        self.final_layernorm = LayerNorm(cfg.hidden_size, cfg.layernorm_epsilon)
        self.checkpoint_activations = checkpoint_activations
        # With 288 GB of HBM not every layer has to be re-run in the backward (the reference's --checkpoint-activations re-runs all of them,
        # mpu/random.py:245-319, transformer.py:621-646).  Counted from the END of the stack (produced last, consumed first: the least peak
        # memory per layer): the last `keep_last` layers keep ALL their activations (~16 [tokens, h] tensors each); the `selective` layers
        # before those keep 6 and rebuild the rest with a third of a layer forward (ParallelTransformerLayer.forward_selective; encoder
        # stacks only); whatever is left in front is re-run whole.  0 / 0 = the reference's behaviour.
        self.keep_last = 0
        self.selective = 0

    def forward(self, x, ids, causal=False, encoder_output=None, enc_ids=None):
        if encoder_output is not None:
            encoder_output = K.fan_in(encoder_output)             # every layer's K/V projection reads it: one gradient buffer, no pairwise sums
        n_layers = len(self.layers)
        n_keep = min(int(self.keep_last), n_layers)
        n_sel = min(int(self.selective), n_layers - n_keep) if encoder_output is None and not causal else 0
        n_recompute = n_layers - n_keep - n_sel
        for li, layer in enumerate(self.layers):
            if self.checkpoint_activations and torch.is_grad_enabled() and n_recompute <= li < n_recompute + n_sel:
                x = layer.forward_selective(x, ids)
            elif self.checkpoint_activations and torch.is_grad_enabled() and li < n_recompute:
                # determinism_check off: the first run of a layer deliberately saves placeholders for tensors only its re-run needs (the
                # FFN pre-activation), so saved-tensor shapes differ between the two runs by design; the COUNT and order stay equal
                x = torch.utils.checkpoint.checkpoint(_CheckpointedLayer(layer), x, ids, causal, encoder_output, enc_ids, use_reentrant=False,
                                                      determinism_check="none")
            else:
                x = layer(x, ids, causal, encoder_output, enc_ids)
        return self.final_layernorm(x)


class _Table(torch.nn.Module):
    def __init__(self, n, h, std):
        super().__init__()
        self.weight = _normal((n, h), std)


class Embedding(torch.nn.Module):
    def __init__(self, cfg, vocab_size, num_tokentypes):
        super().__init__()
        self.word_embeddings = _Table(vocab_size, cfg.hidden_size, cfg.init_method_std)
        self.position_embeddings = _Table(cfg.max_position_embeddings, cfg.hidden_size, cfg.init_method_std)
        self.tokentype_embeddings = _Table(num_tokentypes, cfg.hidden_size, cfg.init_method_std) if num_tokentypes > 0 else None
        self.embedding_dropout, self._site = cfg.hidden_dropout, K.DROPOUT.new_site()          # language_model.py:289: hidden_dropout
        self.fp32 = getattr(cfg, "compute_dtype", "bf16") == "fp32"

    def forward(self, ids, tokentype_ids=None, seqs=None):
        """[b, s, h] from ids (+ token types) [b, s]; or, with `seqs` (a PackedSeqs built from them), the packed [rows, h]."""
        with_types = (seqs.types if seqs is not None else tokentype_ids) is not None
        T = self.tokentype_embeddings.weight if with_types else None
        p = self.embedding_dropout if self.training else 0.0
        return K.embedding(ids, tokentype_ids if with_types else None, self.word_embeddings.weight, self.position_embeddings.weight, T, drop_p=p,
                           seed=K.DROPOUT.seed(self._site) if p else 0, seqs=seqs, fp32=self.fp32)


class TransformerLanguageModel(torch.nn.Module):
    def __init__(self, cfg, vocab_size, num_tokentypes=0, add_decoder=False, checkpoint_activations=False):
        super().__init__()
        self.embedding = Embedding(cfg, vocab_size, num_tokentypes)
        self.encoder = ParallelTransformer(cfg, "encoder", checkpoint_activations)
        self.add_decoder = add_decoder
        if add_decoder:
            self.decoder = ParallelTransformer(cfg, "decoder", checkpoint_activations)

    def packs(self):
        """Whether this stack runs its encoder over packed rows: the switch is on and the fused attention kernels take the head size."""
        return K.PACKING.enabled and self.encoder.layers[0].self_attention.hn == 64 and not self.embedding.fp32

    def encode_packed(self, enc_ids, tokentype_ids=None):
        """(hidden [rows, h], PackedSeqs): the encoder over the real tokens only."""
        seqs = K.PackedSeqs(enc_ids, tokentype_ids)
        return self.encoder(self.embedding(None, None, seqs=seqs), seqs), seqs

    def encode(self, enc_ids, tokentype_ids=None):
        """[b, s, h] like the reference (language_model.py:319-342); with packing on, rows of dropped pad positions are zeros."""
        if self.packs():
            x, seqs = self.encode_packed(enc_ids, tokentype_ids)
            return K.unpack_rows(x, seqs)
        return self.encoder(self.embedding(enc_ids, tokentype_ids), enc_ids)

    def decode(self, dec_ids, encoder_output, enc_ids):
        """encoder_output [b, sk, h] with enc_ids [b, sk]; or the packed [rows, h] with a PackedSeqs of b (groups of) sequences."""
        return self.decoder(self.embedding(dec_ids), dec_ids, causal=True, encoder_output=encoder_output, enc_ids=enc_ids)

    # ---- incremental decoding (search_strategy.py:185-240 re-decodes the whole prefix for every token) --------------------
    def init_decode_state(self, batch, max_len, device="cuda"):
        """Per-layer self-attention K/V cache [b, Lc, 2, np, hn] (Lc = max_len rounded up to 32: the P V product wants K % 32 == 0)."""
        if self.embedding.fp32:
            raise ValueError("incremental decoding is a bf16 path (the fp32 compute mode is for validating forward / loss / backward)")
        lc = (max_len + 31) // 32 * 32
        att = self.decoder.layers[0].self_attention
        return {"len": lc, "kv": [torch.zeros((batch, lc, 2, att.heads, att.hn), dtype=torch.bfloat16, device=device) for _ in self.decoder.layers]}

    def decode_step(self, ids_cur, pos, ids_block, encoder_output, enc_ids, state):
        """Hidden state [b, 1, h] of position `pos` given the tokens so far: ids_cur [b, 1] = ids_block[:, pos]; ids_block [b, Lc] holds the
        prefix and zeros behind it.  Use inside `cross_kv_cache` so the encoder K/V are projected once."""
        emb = self.embedding
        b = ids_cur.shape[0]
        h = emb.word_embeddings.weight.shape[1]
        x = torch.empty((b, 1, h), dtype=torch.bfloat16, device=ids_cur.device)
        pos_row = K.w_bf16(emb.position_embeddings.weight)[pos:pos + 1]          # the table row of THIS position (the kernel indexes by s % S, S = 1)
        K._native.check(K._lib().emdr2_embedding_fwd(ids_cur.contiguous().data_ptr(), None, K.w_bf16(emb.word_embeddings.weight).data_ptr(),
                                                     pos_row.data_ptr(), None, x.data_ptr(), b, 1, h, 0.0, 0, K._sp()), "embedding_fwd")
        for layer, cache in zip(self.decoder.layers, state["kv"]):
            x = layer.step(x, ids_cur, ids_block, pos, cache, encoder_output, enc_ids)
        return self.decoder.final_layernorm(x)


class cross_kv_cache(object):
    """Context manager: inside it (no_grad, eval) every decoder cross-attention keeps the K/V projection of the encoder output it saw
    first and reuses it while the same tensor is passed again -- the reference re-projects all K * S encoder tokens for every generated
    token (search_strategy.py:185-240 calls the full model once per position)."""

    def __init__(self, module):
        self.attn = [m for m in module.modules() if isinstance(m, ParallelAttention) and m.attention_type != "self"]

    def __enter__(self):
        for a in self.attn:
            a.kv_cache = (None, None)
        return self

    def expand(self, old_encoder_output, new_encoder_output, parent):
        """Beam search: the encoder states were re-gathered along the batch (new = old[parent]); carry the cached K/V along instead of
        projecting them again."""
        for a in self.attn:
            if a.kv_cache[0] is old_encoder_output:
                a.kv_cache = (new_encoder_output, a.kv_cache[1].index_select(0, parent))

    def __exit__(self, *exc):
        for a in self.attn:
            a.kv_cache = None


class PretrainedBertModel(torch.nn.Module):
    """BERT tower -> hidden state of token 0 (no pooler; dualencoder_model.py:166-181)."""

    def __init__(self, cfg, vocab_size, num_tokentypes=2, checkpoint_activations=False):
        super().__init__()
        self.language_model = TransformerLanguageModel(cfg, vocab_size, num_tokentypes, False, checkpoint_activations)

    def forward(self, input_ids, tokentype_ids=None):
        if self.language_model.packs():
            x, seqs = self.language_model.encode_packed(input_ids, tokentype_ids)
            return K.first_rows(x, seqs)
        return self.language_model.encode(input_ids, tokentype_ids)[:, 0, :]


class DualEncoderModel(torch.nn.Module):
    def __init__(self, cfg, vocab_size, checkpoint_activations=False):
        super().__init__()
        self.query_model = PretrainedBertModel(cfg, vocab_size, 2, checkpoint_activations)
        self.context_model = PretrainedBertModel(cfg, vocab_size, 2, checkpoint_activations)

    @staticmethod
    def embed_text(model, tokens, types):
        return model(tokens, types)


class T5LMHead(torch.nn.Module):
    def __init__(self, vocab_size):
        super().__init__()
        self.bias = torch.nn.Parameter(torch.zeros(vocab_size, dtype=torch.float32, device="cuda"))

    def forward(self, hidden, word_embeddings_weight):
        return K.linear(hidden, word_embeddings_weight, self.bias)                        # tied LM head (language_model.py:28-41)


class T5Model(torch.nn.Module):
    """Megatron-style encoder-decoder reader (t5_model.py:84-154): learned absolute positions, pre-LN, GELU, tied LM head + bias."""

    def __init__(self, cfg, vocab_size, num_tokentypes=2, checkpoint_activations=False):
        super().__init__()
        self.language_model = TransformerLanguageModel(cfg, vocab_size, num_tokentypes, True, checkpoint_activations)
        self.lm_head = T5LMHead(vocab_size)

    def encode(self, encoder_input_ids):
        return self.language_model.encode(encoder_input_ids)

    def encode_packed(self, encoder_input_ids):
        """(hidden [rows, h], PackedSeqs) -- hand both to `decode` / `decode_hidden` (group the layout for FiD: `seqs.grouped(K)`)."""
        return self.language_model.encode_packed(encoder_input_ids)

    def decode_hidden(self, decoder_input_ids, enc_hidden_states, enc_ids):
        """Decoder output before the LM head (for consumers that fuse the head with what follows it)."""
        return self.language_model.decode(decoder_input_ids, enc_hidden_states, enc_ids)

    def decode(self, decoder_input_ids, enc_hidden_states, enc_ids):
        dec = self.decode_hidden(decoder_input_ids, enc_hidden_states, enc_ids)
        return self.lm_head(dec, self.language_model.embedding.word_embeddings.weight)

    def decode_step(self, ids_cur, pos, ids_block, enc_hidden_states, enc_ids, state):
        """LM logits [b, 1, V] of the next token after position `pos` (incremental decoding with K/V caches)."""
        dec = self.language_model.decode_step(ids_cur, pos, ids_block, enc_hidden_states, enc_ids, state)
        return self.lm_head(dec, self.language_model.embedding.word_embeddings.weight)

    def forward(self, encoder_input_ids, decoder_input_ids):
        if self.language_model.packs():
            enc, seqs = self.encode_packed(encoder_input_ids)
            return self.decode(decoder_input_ids, enc, seqs), K.unpack_rows(enc, seqs)
        enc = self.encode(encoder_input_ids)
        return self.decode(decoder_input_ids, enc, encoder_input_ids), enc
