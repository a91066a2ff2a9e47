"""Generate tests/golden/model_ref.npz: weights, inputs and outputs of the REFERENCE's own modules run on CPU
in fp32 (SURVEY.md Appendix A recipe) at tiny dimensions:

  F1/F3  T5Model: encoder-only output, full enc-dec logits          megatron/model/t5_model.py:112-154
  F2     DualEncoderModel.embed_text (query + context towers)       megatron/model/dualencoder_model.py:77-82,166-181
  F4     EMDR2Model.forward (train, update_retriever) with an injected retriever, the EMDR2 loss and its
         parameter gradients                                        megatron/model/emdr2_model.py:87-214,
                                                                    tasks/openqa/e2eqa/train_e2eqa.py:72-181
  F6     AnnealingLR table                                           megatron/learning_rates.py:51-71
  F8     make_attention_mask_3d / make_history_mask_3d               megatron/data/mask_creation_utils.py:17-42

Build container only; nothing of the reference's source is stored, only tensors."""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402

DIMS = dict(layers=2, hidden=32, heads=2, kv=16, ffn=128, max_pos=64, seq=48, seq_ret=24, dec=8, topk=3, batch=2, vocab_file=250)


def setup():
    _ref_import.install_import_shims()
    vf = os.path.join(tempfile.mkdtemp(), "vocab.txt")
    with open(vf, "w") as f:
        f.write("\n".join(["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + ["tok%d" % i for i in range(DIMS["vocab_file"] - 5)]) + "\n")
    d = DIMS
    sys.argv = ["gen", "--num-layers", str(d["layers"]), "--hidden-size", str(d["hidden"]), "--num-attention-heads", str(d["heads"]),
                "--kv-channels", str(d["kv"]), "--ffn-hidden-size", str(d["ffn"]), "--max-position-embeddings", str(d["max_pos"]),
                "--seq-length", str(d["seq"]), "--seq-length-ret", str(d["seq_ret"]), "--decoder-seq-length", str(d["dec"]),
                "--vocab-file", vf, "--tokenizer-type", "BertWordPieceLowerCase", "--use-cpu-initialization",
                "--topk-retrievals", str(d["topk"]), "--batch-size", str(d["batch"]), "--hidden-dropout", "0.0",
                "--attention-dropout", "0.0", "--update-retriever", "--retriever-score-scaling",
                "--allow-trivial-doc", "--lr", "2e-5", "--warmup", "0.01", "--lr-decay-style", "linear", "--train-iters", "1000"]
    from megatron.global_vars import set_global_variables
    set_global_variables()
    from megatron import mpu, get_args
    mpu.set_model_parallel_world_size(1)
    mpu.set_model_parallel_rank(0)
    import contextlib
    from megatron.mpu import random as mrandom
    mrandom.CudaRNGStatesTracker.fork = lambda self, name=None: contextlib.nullcontext()
    torch.cuda.current_device = lambda: "cpu"
    torch.cuda.LongTensor = torch.LongTensor
    torch.cuda.FloatTensor = torch.FloatTensor
    return get_args()


def flat_state(module):
    return {k: v.detach().clone().numpy() for k, v in module.state_dict().items()}


def main():
    args = setup()
    from megatron import get_tokenizer, get_t5_tokenizer
    from megatron.tokenizer.tokenizer import vocab_size_with_padding
    from megatron.model import T5Model, EMDR2Model
    from megatron.model.dualencoder_model import dualencoder_model_provider
    from megatron.data.mask_creation_utils import make_attention_mask_3d, make_history_mask_3d
    from tasks.openqa.e2eqa.train_e2eqa import get_loss_and_retriever_utility
    d = DIMS
    out = {}
    bert_tok, t5_tok = get_tokenizer(), get_t5_tokenizer()
    bert_vocab = vocab_size_with_padding(bert_tok.vocab_size, args)
    t5_vocab = vocab_size_with_padding(t5_tok.vocab_size, args)
    out["meta"] = np.array([bert_vocab, t5_vocab, t5_tok.cls, t5_tok.sep, t5_tok.pad, t5_tok.bos_token_id, t5_tok.eos_token_id], dtype=np.int64)
    rng = np.random.default_rng(2024)

    def ids(shape, lo=5, hi=245, pad_tail=True):
        x = rng.integers(lo, hi, size=shape)
        if pad_tail:                                  # ragged padding (pad id 0) at the end of each row
            for r in x.reshape(-1, shape[-1]):
                n = int(rng.integers(shape[-1] // 2, shape[-1] + 1))
                r[n:] = 0
        return torch.from_numpy(x.astype(np.int64))

    # ---------------- F8 masks ----------------
    a, b = ids((2, 6)), ids((2, 9))
    out["mask_src"], out["mask_tgt"] = a.numpy(), b.numpy()
    out["mask_3d"] = make_attention_mask_3d(a, b).numpy()
    out["mask_hist"] = make_history_mask_3d(a).numpy()

    # ---------------- F4 EMDR2 forward + loss + grads ----------------
    import assembly_cases
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), ".."))
    case = assembly_cases.build()
    small = lambda toks: [int(t) % 240 + 5 for t in toks]
    passages = [small(p)[:30] for p in case["passages"]]
    titles = [small(t) for t in case["titles"]]
    from tools.inverted_title_index import WikiTitleDocMap
    wmap = WikiTitleDocMap.__new__(WikiTitleDocMap)
    wmap.docid2title = {dd: (min(g),) for dd, g in case["group_of_doc"].items()}
    wmap.title2docs = {}
    for dd, g in case["group_of_doc"].items():
        wmap.title2docs[wmap.docid2title[dd]] = list(g)
    B, K = d["batch"], d["topk"]
    topk_ids = case["topk_ids"][:B, :K].tolist()

    class FakeRetriever(object):
        def get_topk(self, query_tensor):
            data = []
            for row in topk_ids:
                texts = []
                for idx in row:
                    doc_idxs, main = wmap.get_neighbour_paragraphs(idx)
                    texts.append(([passages[x - 1] for x in doc_idxs], main, titles[idx - 1]))
                data.append((row, texts))
            return data, torch.zeros(B, K)

        def update_evidence_embedding(self):
            pass

    torch.manual_seed(99)
    model = EMDR2Model(FakeRetriever()).float()
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.02 * torch.randn_like(p))
    for k, v in flat_state(model).items():
        out["emdr2." + k] = v
    # ---------------- F3 reader towers of the same model ----------------
    t5 = model.language_model
    t5.eval()
    enc_ids, dec_ids = ids((3, d["seq"])), ids((3, d["dec"]))
    dec_ids[:, 0] = t5_tok.bos_token_id
    enc_mask = make_attention_mask_3d(enc_ids, enc_ids) < 0.5
    dec_mask = (make_attention_mask_3d(dec_ids, dec_ids) * make_history_mask_3d(dec_ids)) < 0.5
    ed_mask = make_attention_mask_3d(dec_ids, enc_ids) < 0.5
    with torch.no_grad():
        enc_out = t5(encoder_input_ids=enc_ids, decoder_input_ids=dec_ids, encoder_attn_mask=enc_mask, decoder_attn_mask=None,
                     encoder_decoder_attn_mask=None, output_enc_hidden=True)
        logits, enc_again = t5(enc_ids, dec_ids, encoder_attn_mask=enc_mask, decoder_attn_mask=dec_mask, encoder_decoder_attn_mask=ed_mask)
    out["t5_enc_ids"], out["t5_dec_ids"] = enc_ids.numpy(), dec_ids.numpy()
    out["t5_enc_out"], out["t5_logits"] = enc_out.numpy(), logits.numpy()

    # ---------------- F2 retriever towers of the same model ----------------
    de = model.retriever_model
    de.eval()
    q_ids = ids((3, d["seq_ret"])); q_ids[:, 0] = bert_tok.cls
    q_types0 = torch.zeros_like(q_ids)
    q_mask0 = make_attention_mask_3d(q_ids, q_ids) < 0.5
    with torch.no_grad():
        q_emb = de.embed_text(de.query_model, q_ids, q_mask0, q_types0)
        c_emb = de.embed_text(de.context_model, q_ids, q_mask0, q_types0)
    out["de_ids"], out["de_query_emb"], out["de_context_emb"] = q_ids.numpy(), q_emb.numpy(), c_emb.numpy()
    model.train()

    query_uid = torch.tensor([-1, -2], dtype=torch.int64)
    qb = ids((B, d["seq_ret"])); qb[:, 0] = bert_tok.cls
    q_len = (qb != 0).sum(1)
    for i in range(B):
        qb[i, q_len[i] - 1] = bert_tok.sep
    q_types = torch.zeros_like(qb)
    q_mask = make_attention_mask_3d(qb, qb) < 0.5
    dec = ids((B, d["dec"]), pad_tail=True); dec[:, 0] = t5_tok.bos_token_id
    labels = torch.roll(dec, -1, 1); labels[:, -1] = 0
    for i in range(B):                                   # labels = answer + [EOS] then pad, like train_data_utils.py:60-81
        n = int((dec[i] != 0).sum())
        labels[i, n - 1] = t5_tok.eos_token_id
        labels[i, n:] = 0
    loss_mask = (labels != 0).float()
    lm_logits, topk_log_probs, one = model(query_uid, qb, q_types, q_mask, qb.clone(), q_len, dec)
    lm = lm_logits.float().view(B * d["dec"], -1)
    ce = torch.nn.CrossEntropyLoss(reduction='none', ignore_index=0)(lm, labels.view(-1))
    lm_loss = torch.sum(ce * loss_mask.reshape(-1)) / loss_mask.sum()
    r_loss, r_util, null_loss = get_loss_and_retriever_utility(one, topk_log_probs, labels, loss_mask, t5_tok.eos_token_id)
    net = lm_loss + r_loss
    net.backward()
    out["e_topk_ids"] = np.array(topk_ids, dtype=np.int32)
    out["e_query_uid"], out["e_query_ids"], out["e_query_len"] = query_uid.numpy(), qb.numpy(), q_len.numpy()
    out["e_dec_ids"], out["e_labels"], out["e_loss_mask"] = dec.numpy(), labels.numpy(), loss_mask.numpy()
    out["e_lm_logits"], out["e_topk_log_probs"], out["e_one_context_logits"] = lm_logits.detach().numpy(), topk_log_probs.detach().numpy(), one.detach().numpy()
    out["e_losses"] = np.array([lm_loss.item(), r_loss.item(), r_util.item(), null_loss.item()], dtype=np.float64)
    for k, p in model.named_parameters():
        out["grad." + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
    np.savez_compressed(os.path.join(HERE, "model_corpus.npz"), passages=np.array(passages, dtype=object), titles=np.array(titles, dtype=object), allow_pickle=True)

    # ---------------- F6 learning-rate table ----------------
    from megatron.learning_rates import AnnealingLR
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=2e-5)
    sched = AnnealingLR(opt, start_lr=2e-5, warmup_iter=10, total_iters=1000, decay_style="linear", last_iter=0, min_lr=0.0,
                        use_checkpoint_lr_scheduler=True, override_lr_scheduler=False)
    lrs = []
    for it in range(1000):
        sched.step()
        lrs.append(opt.param_groups[0]["lr"])
    out["lr_table"] = np.array(lrs, dtype=np.float64)

    path = os.path.join(HERE, "model_ref.npz")
    np.savez_compressed(path, **out)
    print("saved", path, "%.1f KB" % (os.path.getsize(path) / 1e3), "losses", out["e_losses"])


if __name__ == "__main__":
    main()


def a1_fixture():
    """F-a1: the reference's QA batch builder (tasks/openqa/e2eqa/train_data_utils.py:27-81) on edge cases."""
    _ref_import.install_import_shims()
    from tasks.openqa.e2eqa.train_data_utils import build_tokens_types_paddings_from_ids as ref
    rng = np.random.default_rng(3)
    cases, outs = [], []
    for qn, an in ((5, 2), (30, 1), (22, 9), (40, 7), (3, 12), (23, 8)):
        q, a = rng.integers(5, 200, size=qn).tolist(), rng.integers(5, 200, size=an).tolist()
        cases.append((q, a))
        outs.append(ref(q, a, 24, 8, 2, 3, 0, 250, 251))
    np.savez_compressed(os.path.join(HERE, "a1_ref.npz"), cases=np.array(cases, dtype=object), outs=np.array(outs, dtype=object), allow_pickle=True)
    print("saved a1_ref.npz")


if __name__ == "__main__" and os.environ.get("EMDR2_GEN_A1"):
    a1_fixture()
