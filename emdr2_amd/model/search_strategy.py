"""Greedy answer generation for exact-match evaluation (reference: megatron/model/search_strategy.py:185-240, `SampleOrGreedySearch`
with sample=False; the EM scorer in train_e2eqa.py:216-266 uses it when --beam-size 1, the setting of every shipped script).

Same contract: `generate_output(model, query_uid, query_ids_bert, query_types, query_mask_bert, query_ids_t5, query_ids_t5_len)` ->
list of token-id lists without [EOS] (an empty answer becomes [1], like the reference).  The first model call retrieves, assembles and
encodes.  What changes on the MI355X side, without altering results (tests/test_decode_gpu.py): the reference re-decodes the whole prefix
and re-projects all K*S encoder tokens to K/V for every generated token; here every step runs ONE position through the decoder with
per-layer self-attention K/V caches (`T5Model.decode_step`, the reference's dormant layer_past / get_key_value plumbing,
transformer.py:273-280,321-332) and the cross-attention K/V are projected once per batch (`cross_kv_cache`).  `incremental=False` keeps
the block form (decoder over the fixed [B, max_decode_len] block padded with id 0 at every step) for comparison."""
import numpy as np
import torch

from emdr2_amd.model.transformer import cross_kv_cache


class SampleOrGreedySearch(object):
    def __init__(self, max_decode_len, bos_id, eos_id, sample=False, topk_evidence=-1, incremental=True, keep_logits=False):
        self.max_decode_length, self.bos_id, self.eos_id, self.sample, self.incremental = max_decode_len, bos_id, eos_id, sample, incremental
        self.keep_logits = keep_logits         # tests only: hold every step's [b, V] logits in `last_logits` (compares the two decoding forms)
        assert topk_evidence >= 1, "this code is customized for retrieval tasks"
        if sample:
            raise NotImplementedError("sampling is not used by the reference's evaluation scripts (--beam-size 1 -> greedy)")

    def generate_output(self, model, query_uid, query_ids_bert, query_types, query_mask_bert, query_ids_t5, query_ids_t5_len):
        batch = query_ids_bert.shape[0]
        L = self.max_decode_length
        y_block = torch.zeros((batch, L), dtype=torch.int64, device=query_ids_bert.device)
        y_block[:, 0] = self.bos_id
        eos_flags = np.zeros((batch,), dtype=np.int32)
        result = []
        hidden = ids_unflat = topk_log_probs = None
        self.last_logits = []
        with torch.no_grad(), cross_kv_cache(model):
            state = None
            for i in range(L):
                if i == 0 and self.incremental:
                    # retrieval + assembly + reader encoder only (dec_ids None: no decoder pass -- the reference's first call decodes the whole
                    # block and this loop would then redo position 0); the decoder runs position by position below
                    _, topk_log_probs, hidden, ids_unflat = model(query_uid, query_ids_bert, query_types, query_mask_bert, query_ids_t5,
                                                                  query_ids_t5_len, None)
                elif not self.incremental:
                    logits, topk_log_probs, hidden, ids_unflat = model(query_uid, query_ids_bert, query_types, query_mask_bert, query_ids_t5,
                                                                       query_ids_t5_len, y_block, all_query_context_hidden_states=hidden,
                                                                       all_query_context_ids_unflat=ids_unflat, topk_log_probs=topk_log_probs)
                    step_logits = logits[:, i, :]
                if self.incremental:
                    reader = getattr(model, "module", model).language_model
                    if state is None:                                       # retrieval + encoder done by the call above; decoder from here on: one position per step
                        state = reader.language_model.init_decode_state(batch, L, y_block.device)
                        y_pad = torch.zeros((batch, state["len"]), dtype=torch.int64, device=y_block.device)
                    y_pad[:, :L] = y_block
                    step_logits = reader.decode_step(y_block[:, i:i + 1].contiguous(), i, y_pad, hidden, ids_unflat, state)[:, 0, :]
                if self.keep_logits:
                    self.last_logits.append(step_logits.float())
                ys = torch.argmax(step_logits.float(), dim=1)             # argmax of log_softmax == argmax of the logits
                if i + 1 < L:
                    y_block[:, i + 1] = ys
                ys = ys.cpu().numpy()
                result.append(ys)
                eos_flags += (ys == self.eos_id)
                if np.all(eos_flags):
                    break
        result = np.stack(result).T
        outs = []
        for y in result:
            inds = np.argwhere(y == self.eos_id)
            if len(inds) > 0:
                y = y[:inds[0, 0]]
            if len(y) == 0:
                y = np.array([1], 'i')
            outs.append(y.tolist())
        return outs
