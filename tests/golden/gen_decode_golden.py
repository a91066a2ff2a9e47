"""Generate tests/golden/decode_ref.npz -- greedy answer generation of the REFERENCE (megatron/model/search_strategy.py:185-240,
`SampleOrGreedySearch(sample=False)`, the decoder of every shipped evaluation) on the toy EMDR2 model of model_ref.npz with its injected
retriever: 16 questions, decoded token ids, and per step the margin between the best and the second-best logit (the bf16 HIP path can
only be asked to reproduce an argmax whose margin exceeds its round-off); and tests/golden/decode_beam_ref.npz -- the reference's
`BeamSearch` (search_strategy.py:124-182, length-normalised scores of :20-41, beam bookkeeping of :44-101, best-beam pick of :104-121) on
the same model and questions at beam sizes 2 and 3: the decoded ids.  Build container only; tensors only."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402
import gen_model_golden as gm  # noqa: E402

NQ = 16
READER_GAIN = float(os.environ.get("READER_GAIN", "8.0"))


def sharpen(name, w):
    """The random toy reader of model_ref.npz decodes one constant token whatever the input (its logits are dominated by the LM-head bias).
    For a decoding fixture that depends on the retrieved evidence and on the prefix, the reader's layer weight matrices (not its embedding tables) are scaled up and the
    LM-head bias dropped -- the same deterministic transformation is applied by the test to the same stored weights."""
    if not name.startswith("language_model."):
        return w
    if name.endswith("lm_head.bias"):
        return np.zeros_like(w)
    if name.endswith(".weight") and w.ndim == 2 and "layernorm" not in name and "embedding" not in name:
        return (w * READER_GAIN).astype(w.dtype)
    return w


def main():
    args = gm.setup()
    from megatron import get_tokenizer, get_t5_tokenizer
    from megatron.model import EMDR2Model
    from megatron.model.search_strategy import SampleOrGreedySearch
    from megatron.data.mask_creation_utils import make_attention_mask_3d
    from tools.inverted_title_index import WikiTitleDocMap
    import assembly_cases
    d = gm.DIMS
    g = np.load(os.path.join(HERE, "model_ref.npz"))
    c = np.load(os.path.join(HERE, "model_corpus.npz"), allow_pickle=True)
    passages, titles = [list(x) for x in c["passages"]], [list(x) for x in c["titles"]]
    case = assembly_cases.build()
    wmap = WikiTitleDocMap.__new__(WikiTitleDocMap)
    wmap.docid2title = {dd: (min(gg),) for dd, gg in case["group_of_doc"].items()}
    wmap.title2docs = {}
    for dd, gg in case["group_of_doc"].items():
        wmap.title2docs[wmap.docid2title[dd]] = list(gg)
    K = d["topk"]
    rng = np.random.default_rng(77)
    n_docs = len(passages)
    topk_ids = np.stack([rng.choice(np.arange(1, n_docs + 1), size=K, replace=False) for _ in range(NQ)]).astype(np.int32)

    class FakeRetriever(object):
        def get_topk(self, query_tensor):
            data = []
            for row in topk_ids.tolist():
                texts = []
                for idx in row:
                    doc_idxs, main = wmap.get_neighbour_paragraphs(idx)
                    texts.append(([passages[x - 1] for x in doc_idxs], main, titles[idx - 1]))
                data.append((row, texts))
            return data, torch.zeros(NQ, K)

        def update_evidence_embedding(self):
            pass

    torch.manual_seed(99)
    model = EMDR2Model(FakeRetriever()).float()
    sd = model.state_dict()
    with torch.no_grad():
        for k, v in sd.items():
            v.copy_(torch.from_numpy(sharpen(k, g["emdr2." + k])))
    model.eval()
    bert_tok, t5_tok = get_tokenizer(), get_t5_tokenizer()
    qb = rng.integers(5, 245, size=(NQ, d["seq_ret"]))
    for r in qb:
        n = int(rng.integers(4, d["seq_ret"] // 2))
        r[n:] = 0
        r[0] = bert_tok.cls
        r[n - 1] = bert_tok.sep
    qb = torch.from_numpy(qb.astype(np.int64))
    q_len = (qb != 0).sum(1)
    q_types = torch.zeros_like(qb)
    q_mask = make_attention_mask_3d(qb, qb) < 0.5
    uid = -torch.arange(1, NQ + 1, dtype=torch.int64)

    margins = []
    orig_forward = model.forward

    def spy(*a, **kw):                                   # record the argmax margin of every decoding step
        out = orig_forward(*a, **kw)
        top2 = torch.topk(out[0][:, -1, :].float(), 2, dim=1).values
        margins.append((top2[:, 0] - top2[:, 1]).detach().numpy())
        return out
    model.forward = spy
    search = SampleOrGreedySearch(max_decode_len=d["dec"], bos_id=t5_tok.bos_token_id, eos_id=t5_tok.eos_token_id, sample=False, topk_evidence=K)
    with torch.no_grad(), _ref_import.cuda_calls_on_cpu():
        outs = search.generate_output(model, uid, qb, q_types, q_mask, qb.clone(), q_len)
    L = max(len(o) for o in outs)
    ids = np.full((NQ, L), -1, dtype=np.int64)
    for i, o in enumerate(outs):
        ids[i, :len(o)] = o
    np.savez_compressed(os.path.join(HERE, "decode_ref.npz"), topk_ids=topk_ids, query_ids=qb.numpy(), query_len=q_len.numpy(), query_uid=uid.numpy(),
                        decoded=ids, margins=np.stack(margins).T, reader_gain=np.float64(READER_GAIN), max_decode_len=np.int64(d["dec"]),
                        bos=np.int64(t5_tok.bos_token_id), eos=np.int64(t5_tok.eos_token_id))
    print("decoded", ids.tolist())
    print("min margin per question", np.stack(margins).T.min(1).round(4).tolist())

    # ---- beam search of the reference on the same model and questions ---------------------------------------------------------------
    from megatron.model.search_strategy import BeamSearch
    model.forward = orig_forward
    beams = {}
    EOS_BIAS = 0.34                       # second variant: an LM-head bias on [EOS] that makes beams END (the flat toy reader never emits it)
    for tag, eos_bias in (("", 0.0), ("_eos", EOS_BIAS)):
        with torch.no_grad():
            model.state_dict()["language_model.lm_head.bias"][t5_tok.eos_token_id] = eos_bias
        for k in (2, 3):
            bs = BeamSearch(max_decode_len=d["dec"], bos_id=t5_tok.bos_token_id, eos_id=t5_tok.eos_token_id, beam_size=k, alpha=0.6, topk_evidence=K)
            with torch.no_grad(), _ref_import.cuda_calls_on_cpu():
                o = bs.generate_output(model, uid, qb, q_types, q_mask, qb.clone(), q_len)
            Lb = max(max(len(x) for x in o), 1)
            a = np.full((NQ, Lb), -1, dtype=np.int64)
            for i, x in enumerate(o):
                a[i, :len(x)] = x
            beams["beam%d%s" % (k, tag)] = a
            print("beam", k, tag, a.tolist())
    np.savez_compressed(os.path.join(HERE, "decode_beam_ref.npz"), alpha=np.float64(0.6), eos_bias=np.float64(EOS_BIAS), **beams)


if __name__ == "__main__":
    main()
