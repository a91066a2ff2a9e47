"""Answer generation for exact-match evaluation (reference: megatron/model/search_strategy.py): `SampleOrGreedySearch` (:185-240; greedy
is what the EM scorer in train_e2eqa.py:216-266 runs at --beam-size 1, the setting of every shipped script; sample=True draws from the
softmax instead) and `BeamSearch` (:124-182 with the length-normalised scores of :20-41, --beam-size > 1).

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
                if self.sample:                                             # search_strategy.py:213-218: one draw from the softmax (torch's generator)
                    ys = torch.multinomial(torch.softmax(step_logits.float(), dim=1), num_samples=1).squeeze(1)
                else:
                    ys = torch.argmax(step_logits.float(), dim=1)         # argmax of log_softmax == argmax of the logits
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


class BeamSearch(object):
    """`beam_size` hypotheses per question, grown one token per step (search_strategy.py:124-182; oracle/transformer_oracle.py::beam_decode
    states the rules and is pinned on the reference's own outputs, tests/golden/decode_beam_ref.npz).  Same contract as the reference:
    `generate_output(...)` -> list of token-id lists, the best hypothesis of every question without [BOS], cut before its first [EOS]
    (possibly empty).  The MI355X side of it: the reference re-runs the whole model call on [batch * beam] prefixes every step and
    re-gathers the [batch * beam, K * S, h] encoder states by parent index every step; here a step decodes ONE position through the
    per-layer self-attention K/V caches, which are what gets re-ordered by parent (a few KB per hypothesis), while the encoder states and
    their cross-attention K/V projection are expanded once (all hypotheses of a question share them, so re-ordering them is a no-op)."""

    def __init__(self, max_decode_len, bos_id, eos_id, beam_size=5, alpha=0.6, topk_evidence=-1):
        self.max_decode_length, self.bos_id, self.eos_id, self.k, self.alpha = max_decode_len, bos_id, eos_id, beam_size, alpha
        assert topk_evidence >= 1, "this code is customized for retrieval tasks"
        assert beam_size >= 1

    def _lp(self, n):
        return pow(5 + n, self.alpha) / pow(5 + 1, self.alpha)               # PolynomialNormalization.lp

    def generate_output(self, model, query_uid, query_ids_bert, query_types, query_mask_bert, query_ids_t5, query_ids_t5_len):
        B, k, L, dev = query_ids_bert.shape[0], self.k, self.max_decode_length, query_ids_bert.device
        with torch.no_grad(), cross_kv_cache(model) as kvc:
            _, _, hidden, ids_unflat = model(query_uid, query_ids_bert, query_types, query_mask_bert, query_ids_t5, query_ids_t5_len, None)
            reader = getattr(model, "module", model).language_model
            state = reader.language_model.init_decode_state(B, L, dev)
            Lc = state["len"]

            def logp(tokens, pos, n, enc, enc_ids):
                """log-softmax over the vocabulary after position `pos` of hypotheses holding tokens[:, :n]."""
                y = torch.zeros((tokens.shape[0], Lc), dtype=torch.int64, device=dev)
                y[:, :n] = tokens[:, :n]
                return torch.log_softmax(reader.decode_step(tokens[:, pos:pos + 1].contiguous(), pos, y, enc, enc_ids, state)[:, 0, :].float(), dim=1)

            outs = torch.full((B, L + 1), self.bos_id, dtype=torch.int64, device=dev)
            total, tok = torch.topk(logp(outs, 0, 1, hidden, ids_unflat), k, dim=1)           # step 0: the k best first tokens, best first
            parent = torch.arange(B, device=dev).repeat_interleave(k)
            hidden_k, ids_k = hidden.index_select(0, parent), ids_unflat.index_select(0, parent)
            kvc.expand(hidden, hidden_k, parent)
            state["kv"] = [c.index_select(0, parent) for c in state["kv"]]
            outs = outs.index_select(0, parent)
            outs[:, 1] = tok.reshape(-1)
            total, n = total.reshape(-1), 2
            base = (torch.arange(B, device=dev) * k)[:, None]
            for pos in range(1, L):
                ended = (outs[:, :n] == self.eos_id).any(1)
                if bool(ended.all()):
                    break
                sc, tok = torch.topk(logp(outs, pos, n, hidden_k, ids_k), k, dim=1)
                cand = (total[:, None] * self._lp(n - 1) + sc) / self._lp(n)
                filler = torch.zeros_like(sc)
                filler[:, 1:] = -10000.0                                                   # an ended hypothesis proposes itself once
                cand = torch.where(ended[:, None], total[:, None] + filler, cand)
                tok = torch.where(ended[:, None], torch.full_like(tok, self.eos_id), tok)
                best, arg = torch.topk(cand.view(B, k * k), k, dim=1)
                parent = (arg // k + base).reshape(-1)
                state["kv"] = [c.index_select(0, parent) for c in state["kv"]]
                outs = outs.index_select(0, parent)
                outs[:, n] = tok.view(B, k * k).gather(1, arg).reshape(-1)
                total, n = best.reshape(-1), n + 1
            total, outs = total.view(B, k).cpu().numpy(), outs[:, 1:n].view(B, k, n - 1).cpu().numpy()
        answers = []
        for q in range(B):
            row = outs[q, int(np.argmax(total[q]))].tolist()                              # the first maximum, like finish_beam's strict '<'
            if self.eos_id in row:
                row = row[:row.index(self.eos_id)]
            answers.append(row)
        return answers
