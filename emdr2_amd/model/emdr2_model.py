"""Retriever plug-in of EMDR2 (reference: megatron/model/emdr2_model.py:379-470,
`PreComputedEvidenceDocsRetriever`), MI355X-native:

* every rank owns one row shard of the evidence index (no node-first-rank special case, no broadcasts:
  INTEGRATION.md section 1);
* `get_topk` keeps the reference's return structure (host lists) for drop-in use under the reference's
  `EMDR2Model.forward`;
* `get_topk_assembled` is the fused path: MIPS search -> device-side evidence fetch + token assembly, returning
  the four tensors the reference's `postprocess` builds (emdr2_model.py:250-303) without leaving the GPU.
"""
import torch

from emdr2_amd.data.emdr2_index import DistributedBruteForceIndex, OpenRetreivalDataStore


class PreComputedEvidenceDocsRetriever(object):
    def __init__(self, args=None, evidence_arena=None, process_group=None, embed_data=None):
        """`PreComputedEvidenceDocsRetriever()` -- the reference's form (emdr2_model.py:379-408): flags from `get_args()`, the evidence
        (`--indexed-evidence-data-path`, `--indexed-title-data-path`, `--evidence-data-path`) loaded into an EvidenceArena in HBM, the
        embeddings of `--embedding-path` into this rank's index shard.  Explicit arguments override: `args` = a namespace with the
        reference's flags (topk_retrievals, hidden_size, allow_trivial_doc, embedding_path, faiss_use_gpu, seq_length, seq_length_ret),
        `evidence_arena` = a prebuilt EvidenceArena, `embed_data` = an OpenRetreivalDataStore."""
        if args is None:
            from emdr2_amd.global_vars import get_args
            args = get_args()
        if evidence_arena is None:
            from emdr2_amd.tasks.openqa.e2eqa.run import build_evidence_arena
            evidence_arena = build_evidence_arena(args)
        self.args = args
        self.topk = args.topk_retrievals
        self.embedding_size = args.hidden_size
        self.evidence_embedder_obj = None
        self.mips_index = None
        self.process_group = process_group
        self.arena = evidence_arena
        self.allow_trivial_doc = args.allow_trivial_doc
        if not args.allow_trivial_doc:
            self.topk = self.topk + 1                       # emdr2_model.py:389-391
        self.precomputed_index_wrapper(embed_data)

    def get_evidence_embedding(self, path):
        # not loaded here: DistributedBruteForceIndex takes an unloaded store through the flat twin of its file (rank 0 converts the pickle
        # once, every rank maps its own rows; emdr2_index.ensure_flat_embedding_file) -- the reference unpickles on the node-first rank
        # only (emdr2_model.py:414-423), which one process per GPU would turn into 8 x 70 GB of host memory
        self.evidence_embedder_obj = OpenRetreivalDataStore(path, load_from_path=False)
        self.evidence_embedder_obj._lazy = True                # a missing file is an error when the index asks for it

    def precomputed_index_wrapper(self, embed_data=None):
        if embed_data is None:
            self.get_evidence_embedding(self.args.embedding_path)
        else:
            self.evidence_embedder_obj = embed_data
        self.mips_index = DistributedBruteForceIndex(embed_size=self.embedding_size, embed_data=self.evidence_embedder_obj,
                                                     use_gpu=getattr(self.args, "faiss_use_gpu", True),
                                                     process_group=self.process_group)
        self._barrier()

    def update_evidence_embedding(self, from_refresh=False):
        """Swap in new evidence embeddings after an indexer job (emdr2_model.py:426-432): reload --embedding-path like the reference, or
        (`from_refresh`) commit the image an in-HBM refresh has just finished (indexer_emdr2.IndexBuilder.build_into_index,
        async_indexer.AsyncIndexBuilder.maybe_swap do this themselves)."""
        if from_refresh:
            self.mips_index.commit_refresh()
        else:
            self.mips_index.update_index()
        self._barrier()

    def _barrier(self):
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.barrier(self.process_group)

    def _world(self):
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            return torch.distributed.get_rank(self.process_group), torch.distributed.get_world_size(self.process_group)
        return 0, 1

    def _search(self, query_tensor):
        """all-gather queries (emdr2_model.py:438-439) -> sharded search + merge -> this rank's rows."""
        local_bsize = query_tensor.shape[0]
        rank, world = self._world()
        q = query_tensor.detach().contiguous()
        if world > 1:
            allq = torch.empty((world * local_bsize, q.shape[1]), dtype=q.dtype, device=q.device)
            torch.distributed.all_gather_into_tensor(allq, q, group=self.process_group)
        else:
            allq = q
        distance, topkindex = self.mips_index.search_mips_index(allq, top_k=self.topk, reconstruct=False)
        sl = slice(rank * local_bsize, (rank + 1) * local_bsize)
        self.searches = getattr(self, "searches", 0) + 1        # (the forward's collectives are behind us: bench_e2e's out-of-memory recovery asks)
        return distance[sl], topkindex[sl]

    def get_topk(self, query_tensor):
        """Reference structure: ([(ids[k], [(doc_list, main_doc_idx, title_ids)] * k)] * b, distance[b, k])."""
        distance, topkindex = self._search(query_tensor)
        topk_data = []
        for topkarray in topkindex.tolist():
            text_list = []
            for idx in topkarray:
                doc_idxs, main_doc_idx = self.arena.neighbour_paragraphs(idx)
                text_list.append(([self.arena.passage(d) for d in doc_idxs], main_doc_idx, self.arena.title(idx)))
            topk_data.append((topkarray, text_list))
        return topk_data, distance

    def get_topk_assembled(self, query_tensor, query_uid, query_ids_t5, query_ids_t5_len, cls_id, sep_id, pad_id):
        """(all_context_ids [b,K,S_ret], all_context_types, all_query_extended_context_ids [b*K,S],
        query_one_context_ids [b*K,S], kept doc ids [b,K], distance) on the device."""
        distance, topkindex = self._search(query_tensor)
        out = self.arena.assemble(topkindex, self.args.topk_retrievals, query_uid, query_ids_t5, query_ids_t5_len,
                                  self.args.seq_length_ret, self.args.seq_length, cls_id, sep_id, pad_id)
        return out + (distance,)


# =====================================================================================================================
# EMDR2Model: retriever towers + MIPS + reader, and the EMDR2 objective
# =====================================================================================================================
import math  # noqa: E402

from emdr2_amd.model import kernels as K  # noqa: E402
from emdr2_amd.model.transformer import Config, DualEncoderModel, T5Model  # noqa: E402


class OneContextLogits(object):
    """`lm_logits_one_context` [B, K, L, V] of the reference (emdr2_model.py:185-210) in unevaluated form: the decoder states of the
    no-grad one-context pass plus the tied LM head's parameters.  The EMDR2 loss only ever needs log_softmax(logits)[gold label]
    (train_e2eqa.py:79-96), which `gold_logprob` computes inside the LM-head GEMM without writing the 6.3 GB logits tensor;
    `materialize()` returns the reference's tensor for any other consumer."""

    def __init__(self, hidden, weight, bias):
        self.hidden, self.weight, self.bias = hidden, weight, bias                      # hidden [B, K, L, H] bf16

    @property
    def shape(self):
        return tuple(self.hidden.shape[:3]) + (self.weight.shape[0],)

    def gold_logprob(self, labels_bkl):
        with torch.no_grad():
            return K.lm_head_gold_logprob(self.hidden, self.weight, self.bias, labels_bkl)

    def materialize(self):
        with torch.no_grad():
            return K.linear(self.hidden, self.weight, self.bias)


class EMDR2Model(torch.nn.Module):
    """reference: megatron/model/emdr2_model.py:31-247.  Same sub-module names (`language_model`, `retriever_model`), same
    checkpoint keys ('encoder/t5_model', 'retriever/biencoder_model'), same training-mode return triple
    (lm_logits [B,L,V], topk_log_probs [B,K], lm_logits_one_context [B,K,L,V]).  Masks are derived from token ids inside the
    kernels, so `query_mask_bert` is accepted for signature compatibility and ignored."""

    def __init__(self, evidence_retriever, cfg=None, t5_vocab_size=None, bert_vocab_size=None, topk=None, seq_length=None, seq_length_ret=None,
                 cls_id=None, sep_id=None, pad_id=0, update_retriever=True, retriever_score_scaling=True, checkpoint_activations=False,
                 disable_retriever_dropout=False, no_query_embedder_training=False, no_context_embedder_training=False):
        """`EMDR2Model(evidence_retriever)` -- the reference's form (emdr2_model.py:31-61): architecture, sequence lengths, top-k and the
        training switches from `get_args()`, vocabulary sizes and special ids from the global tokenizers.  Passing `cfg` and the rest
        explicitly bypasses the globals (tests, bench)."""
        super().__init__()
        if cfg is None:
            from emdr2_amd.global_vars import get_args, get_t5_tokenizer
            a = get_args()
            t5_tok = get_t5_tokenizer()                                             # also fills a.{bert,t5}_padded_vocab_size
            cfg = Config(num_layers=a.num_layers, hidden_size=a.hidden_size, num_attention_heads=a.num_attention_heads,
                         ffn_hidden_size=a.ffn_hidden_size, max_position_embeddings=a.max_position_embeddings,
                         layernorm_epsilon=a.layernorm_epsilon, init_method_std=a.init_method_std, hidden_dropout=a.hidden_dropout,
                         attention_dropout=a.attention_dropout, compute_dtype=getattr(a, "compute_dtype", "bf16"))
            t5_vocab_size, bert_vocab_size = a.t5_padded_vocab_size, a.bert_padded_vocab_size
            topk, seq_length, seq_length_ret = a.topk_retrievals, a.seq_length, a.seq_length_ret
            cls_id, sep_id, pad_id = t5_tok.cls, t5_tok.sep, t5_tok.pad
            update_retriever, retriever_score_scaling = a.update_retriever, a.retriever_score_scaling
            checkpoint_activations, disable_retriever_dropout = a.checkpoint_activations, a.disable_retriever_dropout
            no_query_embedder_training, no_context_embedder_training = a.no_query_embedder_training, a.no_context_embedder_training
        self.topk = topk
        self.language_model = T5Model(cfg, t5_vocab_size, 2, checkpoint_activations)
        self._language_model_key = 'encoder/t5_model'
        self.retriever_model = DualEncoderModel(cfg, bert_vocab_size, checkpoint_activations)
        self._retriever_model_key = 'retriever/biencoder_model'
        self.evidence_retriever = evidence_retriever
        self.hidden_size = cfg.hidden_size
        self.seq_length, self.seq_length_ret = seq_length, seq_length_ret
        self.cls_id, self.sep_id, self.pad_id = cls_id, sep_id, pad_id
        self.update_retriever, self.retriever_score_scaling = update_retriever, retriever_score_scaling
        self.disable_retriever_dropout = disable_retriever_dropout                  # --disable-retriever-dropout (arguments.py:558)
        self.no_query_embedder_training = no_query_embedder_training              # emdr2_model.py:103-104
        self.no_context_embedder_training = no_context_embedder_training          # emdr2_model.py:130-131
        self.fuse_lm_head_loss = True          # one-context pass: LM head + log-softmax + gather in one GEMM (OneContextLogits); False = reference tensor

    def retriever_embedder(self, tokens, mask, types, embedder_type, disable_dropout=False):
        tower = self.retriever_model.query_model if embedder_type == "query" else self.retriever_model.context_model
        if disable_dropout:
            tower.eval()                                                            # emdr2_model.py:69-78
        return self.retriever_model.embed_text(tower, tokens, types)

    def forward(self, query_uid, query_ids_bert, query_types, query_mask_bert, query_ids_t5, query_ids_t5_len, dec_ids,
                all_query_context_hidden_states=None, all_query_context_ids_unflat=None, topk_log_probs=None):
        """Training mode: (lm_logits, topk_log_probs, lm_logits_one_context).  Eval mode (emdr2_model.py:211-214): (lm_logits,
        topk_log_probs, all_query_context_hidden_states [B, K*S, H], all_query_context_ids_unflat [B, K*S]); passing the last two back
        in skips retrieval and the encoder, which is how the greedy decoder iterates (search_strategy.py:203-213).  `dec_ids=None` (eval
        mode only, not in the reference) runs retrieval + encoder without a decoder pass: lm_logits is None."""
        if all_query_context_hidden_states is not None:
            lm_logits = self.language_model.decode(dec_ids, all_query_context_hidden_states, all_query_context_ids_unflat)
            return lm_logits, topk_log_probs, all_query_context_hidden_states, all_query_context_ids_unflat
        query_logits = self.retriever_embedder(query_ids_bert, None, query_types, "query", self.disable_retriever_dropout)
        if self.no_query_embedder_training:
            query_logits = query_logits.detach()
        with torch.no_grad():                                            # emdr2_model.py:107-115 on the device
            ctx_ids, ctx_types, qext, qone, _, _ = self.evidence_retriever.get_topk_assembled(
                query_logits.detach(), query_uid, query_ids_t5, query_ids_t5_len, self.cls_id, self.sep_id, self.pad_id)
        return self.forward_assembled(query_logits, ctx_ids, ctx_types, qext, qone, dec_ids)

    def forward_assembled(self, query_logits, ctx_ids, ctx_types, qext, qone, dec_ids):
        B, Kk = ctx_ids.shape[:2]
        H = self.hidden_size
        ctx_logits = self.retriever_embedder(ctx_ids.reshape(B * Kk, -1), None, ctx_types.reshape(B * Kk, -1), "context",
                                             self.disable_retriever_dropout).reshape(B, Kk, H)
        if self.no_context_embedder_training:
            ctx_logits = ctx_logits.detach()
        # fresh retriever scores -> prior over the K passages (emdr2_model.py:134-145): fp32 dot products, /sqrt(H), log-softmax, one HIP kernel
        topk_log_probs = K.retriever_prior(query_logits, ctx_logits, 1.0 / math.sqrt(H) if self.retriever_score_scaling else 1.0)

        S = qext.shape[1]
        packed = self.language_model.language_model.packs()
        if packed:
            # the reader encoder over real tokens only; a question's K passages are consecutive sequences of the packed buffer, so the FiD
            # concatenation (:159-161) is the same rows seen as B groups of K sequences -- no pad rows between the passages
            enc, seqs = self.language_model.encode_packed(qext)
            lm_logits = self.language_model.decode(dec_ids, enc, seqs.grouped(Kk)) if dec_ids is not None else None
        else:
            enc = self.language_model.encode(qext).reshape(B, Kk * S, H)                      # K passages concatenated (FiD), :148-164
            lm_logits = self.language_model.decode(dec_ids, enc, qext.reshape(B, Kk * S)) if dec_ids is not None else None   # :166-183

        one = None
        if dec_ids is None and self.training:
            raise ValueError("dec_ids=None (encoder only) is an evaluation-mode call")
        if self.training and self.update_retriever:
            with torch.no_grad():                                                             # :185-210
                dec_rep = torch.repeat_interleave(dec_ids, Kk, dim=0)
                K.DROPOUT.pass_id = 1                                                         # a second, independent draw of every dropout site
                try:
                    enc1, ids1 = self.language_model.encode_packed(qone) if packed else (self.language_model.encode(qone), qone)
                    if self.fuse_lm_head_loss:
                        hid = self.language_model.decode_hidden(dec_rep, enc1, ids1)
                        one = OneContextLogits(hid.reshape(B, Kk, dec_ids.shape[1], H), self.language_model.language_model.embedding.word_embeddings.weight,
                                               self.language_model.lm_head.bias)
                    else:
                        one = self.language_model.decode(dec_rep, enc1, ids1).reshape(B, Kk, dec_ids.shape[1], -1)
                finally:
                    K.DROPOUT.pass_id = 0
        if not self.training:
            if packed:                                                                       # the reference's shapes for the eval-mode return (:211-214)
                enc = K.unpack_rows(enc, seqs).reshape(B, Kk * S, H)
            return lm_logits, topk_log_probs, enc, qext.reshape(B, Kk * S)
        return lm_logits, topk_log_probs, one

    def forward_backward(self, query_uid, query_ids_bert, query_types, query_mask_bert, query_ids_t5, query_ids_t5_len, dec_ids, labels,
                         loss_mask, eos_id, micro_batches=1, ret_kldiv=False, on_group=None):
        """Forward, EMDR2 loss AND backward of one training step with the B questions of the batch run in `micro_batches` groups; returns
        (loss, stats) of the whole batch, gradients delivered (the reference's step is `model(...)` -> `get_loss_and_retriever_utility` ->
        `backward_step` over the undivided batch: emdr2_model.py:87-214, train_e2eqa.py:126-181, megatron/training.py:165-200).

        Why it is the same step: only the query tower and the MIPS search see the batch as a whole (ONE search per step: its cost is the
        index scan, not the number of queries) -- they run once, for all B questions.  Everything after the search is per question: a
        question's K passages go through the context tower, the prior is a softmax over ITS K scores (:134-145), FiD cross-attention
        concatenates ITS K passages (:159-161), both losses are sums over questions divided by batch-wide token counts (`loss_totals`).
        So each group's loss is taken with the whole batch's denominators and back-propagated at once; parameter gradients add up in the
        optimizer's flat fp32 buckets (training.FlatAdam counts m contributions per parameter and releases a bucket to its all-reduce when
        the last one has arrived), the query embeddings' gradient is collected per group and sent through the query tower once at the end.
        What it buys: the activations alive at any time are those of B / m questions, so NO layer has to be re-run in the backward
        (--checkpoint-activations off: zero recompute) inside 288 GB even at top-k 100.  Dropout: group i draws its masks from seeds
        hashed with i (kernels.DROPOUT.micro), so an undivided step (m = 1) is bit-identical to `forward` + `emdr2_loss` + `backward`.
        B need not be a multiple of `micro_batches` (the short last batch of an epoch under --keep-last, train_e2eqa.py:272): the groups are
        then `torch.tensor_split`-sized (the first B % m groups hold one question more; fewer groups than asked when B < m) -- the
        batch-wide denominators make uneven groups exactly as right as even ones.
        `on_group(i)`: called between group i's forward and its backward (test hook: injected allocation failures)."""
        B = query_ids_bert.shape[0]
        m = int(micro_batches)
        if m < 1:
            raise ValueError("micro_batches must be >= 1 (got %d)" % m)
        if not self.training:
            raise ValueError("forward_backward is a training-mode call")
        totals = loss_totals(labels, loss_mask, eos_id)
        query_logits = self.retriever_embedder(query_ids_bert, None, query_types, "query", self.disable_retriever_dropout)
        if self.no_query_embedder_training:
            query_logits = query_logits.detach()
        with torch.no_grad():                                            # ONE search + device-side fetch / assembly for all B questions
            ctx_ids, ctx_types, qext, qone, _, _ = self.evidence_retriever.get_topk_assembled(
                query_logits.detach(), query_uid, query_ids_t5, query_ids_t5_len, self.cls_id, self.sep_id, self.pad_id)
        Kk = ctx_ids.shape[1]
        q_leaf = query_logits.detach().requires_grad_(query_logits.requires_grad)
        bounds = question_group_bounds(B, m)
        loss_sum, stats_sum = None, {}
        try:
            for i, (lo, hi) in enumerate(bounds):
                K.DROPOUT.micro = i
                lm, tlp, one = self.forward_assembled(q_leaf[lo:hi], ctx_ids[lo:hi], ctx_types[lo:hi], qext[lo * Kk:hi * Kk], qone[lo * Kk:hi * Kk],
                                                      dec_ids[lo:hi])
                loss, stats = emdr2_loss(lm, tlp, one, labels[lo:hi], loss_mask[lo:hi], eos_id, ret_kldiv=ret_kldiv, totals=totals)
                del lm, tlp, one
                if on_group is not None:
                    on_group(i)
                loss.backward()                                          # frees this group's activations before the next group's forward
                loss_sum = loss.detach() if loss_sum is None else loss_sum + loss.detach()
                for k_, v in stats.items():
                    stats_sum[k_] = v if k_ not in stats_sum else stats_sum[k_] + v
                del loss, stats
        finally:
            K.DROPOUT.micro = 0
        if q_leaf.grad is not None:
            query_logits.backward(q_leaf.grad)                           # the query tower's backward: once, over all B questions
        return loss_sum, stats_sum

    def state_dict_for_save_checkpoint(self):
        """The reference's nested checkpoint dict (emdr2_model.py:217-226; layout in emdr2_amd/checkpointing.py)."""
        from emdr2_amd import checkpointing
        return checkpointing.emdr2_state_dict(self)

    def load_state_dict_from_checkpoint(self, state, strict=True):
        """emdr2_model.py:228-231: the nested checkpoint dict."""
        from emdr2_amd import checkpointing
        checkpointing.load_emdr2_state_dict(self, state, strict)
        K.WEIGHTS.invalidate()

    def set_recompute_keep_last(self, reader_encoder_layers):
        """Under --checkpoint-activations, keep the activations of the last n reader-encoder layers instead of re-running them in the backward
        (the reader encoder holds B * K * S tokens: its layers are the expensive ones to re-run).  Gradients are unchanged
        (tests/test_model_gpu.py); ~33 GB of HBM per layer at B = 64, K = 50, S = 512."""
        self.language_model.language_model.encoder.keep_last = max(0, int(reader_encoder_layers))

    def set_selective_retention(self, reader_encoder_layers, context_tower_layers=0, query_tower_layers=0):
        """Under --checkpoint-activations, run the last n layers of an encoder stack with SELECTIVE retention (6 [tokens, h] tensors kept per
        layer, LayerNorm outputs and the FFN intermediates rebuilt in the backward by one h -> 4h GEMM) instead of re-running the whole
        layer: ~8 GB per reader-encoder layer at the benchmark shape for two thirds of that layer's recompute.  Gradients unchanged."""
        self.language_model.language_model.encoder.selective = max(0, int(reader_encoder_layers))
        self.retriever_model.context_model.language_model.encoder.selective = max(0, int(context_tower_layers))
        self.retriever_model.query_model.language_model.encoder.selective = max(0, int(query_tower_layers))

    def load_state_dict(self, state_dict, strict=True, **kw):
        """The reference's name for the above (emdr2_model.py:228-231 overrides nn.Module.load_state_dict with the nested form);
        a flat torch state dict still goes to nn.Module."""
        if self._language_model_key in state_dict or self._retriever_model_key in state_dict:
            return self.load_state_dict_from_checkpoint(state_dict, strict)
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        K.WEIGHTS.invalidate()
        return out

    def init_state_dict_from_dpr_and_t5(self, pretrained_t5_load, pretrained_dpr_load):
        """emdr2_model.py:233-247: iteration-0 initialisation from the pre-trained reader and dual encoder."""
        from emdr2_amd import checkpointing
        if pretrained_t5_load is None or pretrained_dpr_load is None:
            import warnings
            warnings.warn("Pretrained Checkpoints are not found. Initializing from random weights")
            return
        checkpointing.load_t5_checkpoint(self.language_model, pretrained_t5_load)
        checkpointing.load_dualencoder_checkpoint(self.retriever_model, pretrained_dpr_load)


def question_group_bounds(batch, micro_batches):
    """[(lo, hi)] of the question groups of a step: min(micro_batches, batch) groups, sizes as `torch.tensor_split` deals them (the first
    batch % m groups get one question more).  The same function of (batch, m) on every rank: the number of gradient contributions per
    parameter -- what FlatAdam's bucket protocol counts -- follows from it."""
    m = max(1, min(int(micro_batches), int(batch)))
    base, extra = divmod(int(batch), m)
    bounds, lo = [], 0
    for i in range(m):
        hi = lo + base + (1 if i < extra else 0)
        bounds.append((lo, hi))
        lo = hi
    return bounds


def loss_totals(labels, loss_mask, eos_id):
    """The three batch-wide denominators of `emdr2_loss` (device scalars): sum of the loss mask, sum of the utility mask (answer tokens
    before [EOS], train_e2eqa.py:113-118), batch size.  A micro-batch's losses divided by THESE add up to the full batch's."""
    mask = loss_mask.float()
    lab = labels.masked_fill(~loss_mask.to(torch.bool), 0)
    return mask.sum(), mask.masked_fill(lab >= eos_id, 0).sum(), labels.shape[0]


def emdr2_loss(lm_logits, topk_log_probs, lm_logits_one_context, labels, loss_mask, eos_id, ret_kldiv=False, totals=None):
    """_cross_entropy_forward_step + get_loss_and_retriever_utility (tasks/openqa/e2eqa/train_e2eqa.py:72-181).
    The two vocabulary-sized log-softmax + gather passes run in the HIP kernel; what is left in torch acts on [B,L] / [B,K,L].
    `totals` (from `loss_totals` of the whole batch): the arguments are a group of the batch's questions and every mean is taken with the
    whole batch's denominator, so losses, statistics and gradients of the groups SUM to those of the undivided batch."""
    mask = loss_mask.float()
    mask_sum, util_sum, batch = totals if totals is not None else (mask.sum(), None, labels.shape[0])
    gold = K.lse_gather(lm_logits, labels)                                                    # [B, L] fp32
    lm_loss = -torch.sum(gold * mask * (labels != 0)) / mask_sum                              # CrossEntropyLoss(ignore_index=0) * loss_mask
    stats = {"lm_loss": lm_loss.detach()}
    retriever_loss = torch.zeros((), device=lm_logits.device)
    if lm_logits_one_context is not None:
        Kk = lm_logits_one_context.shape[1]
        lab = labels.masked_fill(~loss_mask.to(torch.bool), 0)
        labk = lab.unsqueeze(1).expand(-1, Kk, -1).contiguous()
        if isinstance(lm_logits_one_context, OneContextLogits):
            gold1 = lm_logits_one_context.gold_logprob(labk)                                                   # [B, K, L], logits never stored
        else:
            gold1 = K.lse_gather(lm_logits_one_context, labk).detach()
        if ret_kldiv:                                                                         # --ret-kldiv (train_e2eqa.py:184-214)
            teacher_log = torch.sum(gold1 * mask.unsqueeze(1), dim=2) / torch.sum(mask.unsqueeze(1), dim=2)
            retriever_loss = torch.nn.functional.kl_div(topk_log_probs.float(), torch.softmax(teacher_log, dim=1), reduction='sum') / batch   # 'batchmean'
            stats["retriever_loss"] = retriever_loss.detach()
            return lm_loss + retriever_loss, stats
        marginal = K.marginal_logsumexp(topk_log_probs, gold1)                                  # [B, L] = logsumexp_k(prior + gold), HIP kernel pair
        retriever_loss = -torch.sum(marginal * mask) / mask_sum
        util_mask = mask.masked_fill(lab >= eos_id, 0)
        stats["retriever_utility"] = (torch.sum((marginal - gold1[:, -1, :]) * util_mask) / (util_mask.sum() if util_sum is None else util_sum)).detach()
        stats["null_block_lm_loss"] = (-torch.sum(gold1[:, -1, :] * mask) / mask_sum).detach()
    stats["retriever_loss"] = retriever_loss.detach()
    return lm_loss + retriever_loss, stats
