"""Greedy answer generation for exact-match evaluation (reference: megatron/model/search_strategy.py:185-240, `SampleOrGreedySearch`
with sample=False; the EM scorer in train_e2eqa.py:216-266 uses it when --beam-size 1, the setting of every shipped script).

Same contract: `generate_output(model, query_uid, query_ids_bert, query_types, query_mask_bert, query_ids_t5, query_ids_t5_len)` ->
list of token-id lists without [EOS] (an empty answer becomes [1], like the reference).  The first model call retrieves, assembles and
encodes; later calls pass the encoder states back in.  Two MI355X-side changes that do not alter results: the decoder always runs on the
fixed [B, max_decode_len] block padded with id 0 (the causal + padding masks make position i blind to everything behind it, so logits at
position i equal those of the length-(i+1) prefix), and the cross-attention K/V of the K*S encoder tokens are projected once per batch
instead of once per generated token (`cross_kv_cache`)."""
import numpy as np
import torch

from emdr2_amd.model.transformer import cross_kv_cache


class SampleOrGreedySearch(object):
    def __init__(self, max_decode_len, bos_id, eos_id, sample=False, topk_evidence=-1):
        self.max_decode_length, self.bos_id, self.eos_id, self.sample = max_decode_len, bos_id, eos_id, sample
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
        with torch.no_grad(), cross_kv_cache(model):
            for i in range(L):
                logits, topk_log_probs, hidden, ids_unflat = model(query_uid, query_ids_bert, query_types, query_mask_bert, query_ids_t5,
                                                                   query_ids_t5_len, y_block, all_query_context_hidden_states=hidden,
                                                                   all_query_context_ids_unflat=ids_unflat, topk_log_probs=topk_log_probs)
                ys = torch.argmax(logits[:, i, :].float(), dim=1)        # argmax of log_softmax == argmax of the logits
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
