"""A LEARNABLE planted open-domain QA task for the training entry point (the reference's deliverable is exact match and
retrieval accuracy: tasks/openqa/e2eqa/train_e2eqa.py:72-123,217-292, tasks/openqa/dense_retriever/evaluation/evaluate.py:42-134).

World: N passages, passage d = title "xa<i> xb<j> xc<l>" (a three-word key; every word is frequent, the combination belongs to d alone) +
text "ya<p> yb<q> yc<r> f.. f.. f.." (a three-word answer, likewise unique to d, + filler words); question d = "what is xa<i> xb<j> xc<l> ?",
answer ["ya<p> yb<q> yc<r>"].  So every question's answer string sits in exactly ONE passage and the `has_answer`-style retrieval accuracy @k
of the reference's evaluator IS the recall@k of the gold passage.  Nothing can be answered closed-book from a few passes over the questions
(N random key -> value pairs): the reader must copy the value from a retrieved passage, and the only training signal the
retriever ever receives is the EMDR2 objective -- the reader's likelihood of the answer given each retrieved passage
(emdr2_model.py:185-210, train_e2eqa.py:72-123).

Start: like the reference's recipe (README / examples/openqa/emdr2_nq.sh: an ICT- / MSS-pre-trained dual encoder with modest recall and a
pre-trained T5 reader, loaded by `--pretrained-dpr-load` / `--pretrained-t5-load`, emdr2_model.py:233-247), EMDR2 training starts from
"pre-trained" checkpoints, which this script makes itself with this package's modules, kernels and optimizer and writes in the reference's
checkpoint layout:
  * a WEAK dual encoder: both towers from one set of weights, then 5 steps of in-batch-negative contrastive training (the upstream ICT / DPR
    stage, out of SURVEY section 8's scope) -- the gold passage is in the top-20 for ~3 of 4 questions and on top for ~1 of 6;
  * a reader that can READ ONE PASSAGE: 400 steps on (question, gold passage) pairs -- it copies the answer out of the passage it is given.
The initial `--embedding-path` pickle is the indexer job's output for the weak dual encoder (megatron/indexer_emdr2.py:77-114).  Training
then runs through `emdr2_amd.tasks.run` with the flag set of examples/openqa/emdr2_nq.sh (tiny sizes, top-k 16), `--async-indexer` refreshing
the index from the live context tower.  With `--update-retriever` the retriever improves from the reader's signal alone -- the likelihood of
the answer under each retrieved passage, read one at a time (the no-grad one-context pass) -- and recall@1 / @5 / @20 go to ~1.0; without it
the retriever receives no gradient and recall stays where the warm-up left it.  Exact match is scored with the reference's scorer
(train_e2eqa.py:216-283) on what the reader generates from the passage the retriever ranks FIRST (`em_top1`): 0.16 -> 0.99 with the retriever
update, 0.16 without.  (`em_fid`, the same scorer over all 16 retrieved passages at once, stays ~0 in both arms: picking the passage whose
title is the question's key out of K concatenated ones is a skill this 4-layer reader does not acquire in a test-sized budget -- measured while
this task was designed: 3,000 FiD steps at K = 2 sit at the cannot-choose loss, K = 16 never leaves the unigram plateau; a reader that
cannot choose is also what makes the retriever's job visible in EM.  What did NOT work as a start, for the record: a weak retriever with
recall@16 of 0.3 or less -- most training contexts then lack the answer and FiD training un-teaches the reader to read within 100 steps,
after which Adam turns the retriever's noise gradients into a random walk and recall collapses to chance.)

    python tools/planted_task.py /tmp/planted [--steps 300] [--no-update-retriever]

prints recall@k before / after and the final EM; tests/test_planted_task_gpu.py asserts on them."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

LAYERS, HIDDEN, HEADS, FFN = 4, 256, 4, 1024
S_RET, S, L = 32, 64, 32
INIT_STD = float(os.environ.get("PLANTED_INIT_STD", "0.1"))
READER_LR = float(os.environ.get("PLANTED_READER_LR", "1e-3"))
READER_K = int(os.environ.get("PLANTED_READER_K", "1"))          # passages per question in the reader warm-up (0 = the task's top-k)
# curriculum of the reader warm-up: "until step : passages per question" -- reading ONE passage (copying the answer) is learnt in ~300 steps,
# choosing the passage whose title is the question's key among several is learnt one doubling at a time
CURRICULUM = [tuple(int(v) for v in item.split(":")) for item in os.environ.get("PLANTED_CURRICULUM", "").split(",") if item]


ALPHA = 13          # symbols per position: 13^3 = 2,197 distinct three-word keys / answers


def _triples(n, rng):
    """n distinct (i, j, l) triples over ALPHA symbols, in random order."""
    if n > ALPHA ** 3:
        raise ValueError("at most %d passages" % ALPHA ** 3)
    code = rng.permutation(ALPHA ** 3)[:n]
    return [(int(c) // (ALPHA * ALPHA), (int(c) // ALPHA) % ALPHA, int(c) % ALPHA) for c in code]


ANSWER_KEY = os.environ.get("PLANTED_ANSWER_KEY", "0") == "1"   # the answer repeats the key before the value
KEY_STYLE = os.environ.get("PLANTED_KEY", "triple")      # "word": one key word per passage (k0001 ..), "triple": three frequent words


def key_of(world_keys, d):
    return ("k%04d" % d) if KEY_STYLE == "word" else "xa%02d xb%02d xc%02d" % tuple(world_keys[d - 1])


def make_world(tmp, n_docs=2000, n_valid=200, fillers=50, seed=0):
    """vocab, evidence TSV + memory-mapped token files (the reference's preprocess output), train / valid QA files.  Passage d: title = a
    three-word key "xa.. xb.. xc.." no other passage has, text = a three-word answer "ya.. yb.. yc.." no other passage has + filler words;
    every word of a key / an answer is frequent (13 symbols per position), the COMBINATION is unique."""
    from emdr2_amd.data.indexed_dataset import MMapIndexedDatasetBuilder
    from emdr2_amd.tokenizer import BertWordPieceTokenizer
    os.makedirs(tmp, exist_ok=True)
    rng = np.random.default_rng(seed)
    vocab = os.path.join(tmp, "vocab.txt")
    words = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "what", "is", "?"] + ["f%03d" % i for i in range(fillers)] + \
            ["%s%02d" % (p, i) for p in ("xa", "xb", "xc", "ya", "yb", "yc") for i in range(ALPHA)] + ["k%04d" % d for d in range(1, n_docs + 1)]
    with open(vocab, "w") as f:
        f.write("\n".join(words) + "\n")
    t = BertWordPieceTokenizer(vocab)
    keys, vals = _triples(n_docs, rng), _triples(n_docs, rng)
    # the answer repeats the key ("k0017 ya03 yb07 yc11", as in "k0017 is ..."): once the decoder has emitted the key it holds the very
    # token the gold passage starts with -- choosing among the K retrieved passages is a one-hop key lookup for the cross-attention
    answer = lambda d: ("%s ya%02d yb%02d yc%02d" % ((key_of(keys, d),) + tuple(vals[d - 1]))) if ANSWER_KEY else ("ya%02d yb%02d yc%02d" % tuple(vals[d - 1]))
    ev = os.path.join(tmp, "psgs.tsv")
    pb = MMapIndexedDatasetBuilder(os.path.join(tmp, "text.bin")); tb = MMapIndexedDatasetBuilder(os.path.join(tmp, "title.bin"))
    with open(ev, "w") as f:
        f.write("id\ttext\ttitle\n")
        for d in range(1, n_docs + 1):
            title = key_of(keys, d)
            text = " ".join([answer(d)] + ["f%03d" % i for i in rng.integers(0, fillers, size=int(rng.integers(3, 7)))])
            f.write("%d\t%s\t%s\n" % (d, text, title))
            ids_t, ids_p = t.tokenize(title), t.tokenize(text)
            assert len(ids_t) in (1, 3) and 1 not in ids_p and 1 not in ids_t  # whole words, nothing unknown
            pb.add_item(ids_p); pb.end_document()
            tb.add_item(ids_t); tb.end_document()
    pb.finalize(os.path.join(tmp, "text.idx")); tb.finalize(os.path.join(tmp, "title.idx"))
    order = rng.permutation(n_docs) + 1
    for name, docs in (("train", order), ("valid", order[:n_valid])):
        with open(os.path.join(tmp, name + ".tsv"), "w") as f:
            for d in docs:
                f.write("what is %s ?\t%s\n" % (key_of(keys, d), json.dumps([answer(d)])))
    with open(os.path.join(tmp, "world.json"), "w") as f:
        json.dump({"keys": keys, "vals": vals}, f)
    return vocab, ev


def _cfg(dropout=0.0):
    from emdr2_amd.model.transformer import Config
    return Config(num_layers=LAYERS, hidden_size=HIDDEN, num_attention_heads=HEADS, ffn_hidden_size=FFN, max_position_embeddings=64,
                  init_method_std=INIT_STD, hidden_dropout=dropout, attention_dropout=dropout)


def _save_pretrained(tmp, name, sd):
    from emdr2_amd import checkpointing
    d = os.path.join(tmp, "pretrained_" + name)
    path = checkpointing.get_checkpoint_name(d, 1)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save({"checkpoint_version": 1.0, "iteration": 1, "model": sd}, path)
    with open(checkpointing.get_checkpoint_tracker_filename(d), "w") as f:
        f.write("1")
    return d


def _questions(t, world, docs, device="cuda"):
    """[CLS] what is <key of d> ? [SEP] pad rows (the dataset's encoder rows, train_data_utils.py:27-81) for a list of doc ids -> (ids, lengths)."""
    from emdr2_amd.tasks.openqa.e2eqa.train_data_utils import build_tokens_types_paddings_from_ids
    rows = [build_tokens_types_paddings_from_ids(t.tokenize("what is %s ?" % key_of(world["keys"], d)), [], S_RET, 2, t.cls, t.sep, t.pad, t.bos_token_id,
                                                 t.eos_token_id) for d in docs]
    return (torch.tensor([r[0] for r in rows], dtype=torch.int64, device=device), torch.tensor([r[2] for r in rows], dtype=torch.int64, device=device))


def write_pretrained(tmp, vocab, ev, n_docs, seed=1, retriever_steps=40, reader_steps=1200, topk=16, log=print):
    """The "pre-trained" checkpoints of --pretrained-dpr-load / --pretrained-t5-load in the reference's directory layout
    (checkpointing.py:267-340), made with this package's own modules, kernels and optimizer: a WEAK dual encoder (towers from one set of
    weights + `retriever_steps` of in-batch-negative contrastive training) and a reader trained to read (`reader_steps` of FiD training on
    gold passage + random distractors).  Returns (dual encoder, dpr dir, t5 dir)."""
    from emdr2_amd import checkpointing
    from emdr2_amd.indexer_emdr2 import IndexBuilder
    from emdr2_amd.model import kernels as K
    from emdr2_amd.model.emdr2_model import emdr2_loss
    from emdr2_amd.model.transformer import DualEncoderModel, T5Model
    from emdr2_amd.tokenizer import BertWordPieceTokenizer, vocab_size_with_padding
    from emdr2_amd.training import FlatAdam
    t = BertWordPieceTokenizer(vocab)
    t5_tok = BertWordPieceTokenizer(vocab, vocab_extra_ids=100)
    bert_vocab = vocab_size_with_padding(BertWordPieceTokenizer(vocab, vocab_extra_ids=0).vocab_size)
    t5_vocab = vocab_size_with_padding(t5_tok.vocab_size)
    arena = _arena(tmp, ev)
    world = json.load(open(os.path.join(tmp, "world.json")))
    world["keys"], world["vals"] = [tuple(k) for k in world["keys"]], [tuple(v) for v in world["vals"]]
    g = torch.Generator(device="cuda").manual_seed(seed)
    torch.manual_seed(seed)
    K.PACKING.sticky = False
    # ---- the weak dual encoder ------------------------------------------------------------------------------------------------------
    de = DualEncoderModel(_cfg(), bert_vocab)
    de.context_model.load_state_dict(de.query_model.state_dict())
    K.WEIGHTS.invalidate()
    de.train()
    opt = K.GRAD_SINK = FlatAdam(de, lr=5e-4, weight_decay=0.0, clip_grad=1.0, bucket_bytes=64 << 20)
    builder = IndexBuilder(de.context_model, arena, S_RET, t.cls, t.sep, t.pad, batch_size=128, log_interval=1 << 30)
    try:
        for step in range(retriever_steps):
            docs = (torch.randperm(n_docs, generator=g, device="cuda")[:64] + 1)
            q_ids, _ = _questions(t, world, docs.tolist())
            c_ids, c_types = builder.context_inputs(docs.to(torch.int32))
            opt.zero_grad()
            q = de.query_model(q_ids, torch.zeros_like(q_ids))
            c = de.context_model(c_ids, c_types)
            scores = q.float() @ c.float().t() / (HIDDEN ** 0.5)                      # in-batch negatives (the upstream DPR / ICT objective)
            loss = torch.nn.functional.cross_entropy(scores, torch.arange(64, device="cuda"))
            loss.backward()
            opt.finish(); opt.step()
        log("dual-encoder warm-up: %d steps, contrastive loss %.3f" % (retriever_steps, float(loss)) if retriever_steps else "dual-encoder warm-up: none")
    finally:
        K.GRAD_SINK = None
    de.eval()
    dpr = _save_pretrained(tmp, "dpr", checkpointing.dualencoder_state_dict(de))
    # ---- a reader that can read -----------------------------------------------------------------------------------------------------
    t5 = T5Model(_cfg(), t5_vocab).train()
    opt = K.GRAD_SINK = FlatAdam(t5, lr=READER_LR, weight_decay=0.0, clip_grad=1.0, bucket_bytes=64 << 20)
    B = 16
    topk_final = READER_K or topk
    try:
        for step in range(reader_steps):
            topk = next((k for until, k in CURRICULUM if step < until), topk_final)
            docs = torch.randperm(n_docs, generator=g, device="cuda")[:B] + 1
            others = torch.randint(1, n_docs + 1, (B, topk), generator=g, device="cuda")
            slot = torch.randint(0, topk, (B,), generator=g, device="cuda")
            others[torch.arange(B, device="cuda"), slot] = docs                        # the gold passage at a random rank
            q_ids, q_len = _questions(t, world, docs.tolist())
            _, _, qext, _, _ = arena.assemble(others.to(torch.int32), topk, -docs, q_ids, q_len, S_RET, S, t.cls, t.sep, t.pad)
            ans = [t5_tok.tokenize(("%s ya%02d yb%02d yc%02d" % ((key_of(world["keys"], d),) + tuple(world["vals"][d - 1]))) if ANSWER_KEY else
                                   ("ya%02d yb%02d yc%02d" % tuple(world["vals"][d - 1]))) for d in docs.tolist()]
            dec = torch.zeros((B, L), dtype=torch.int64, device="cuda"); labels = torch.zeros_like(dec)
            for i, a in enumerate(ans):
                dec[i, 0] = t5_tok.bos_token_id; dec[i, 1:1 + len(a)] = torch.tensor(a); labels[i, :len(a)] = torch.tensor(a); labels[i, len(a)] = t5_tok.eos_token_id
            mask = (labels != 0).float()
            opt.zero_grad()
            enc, seqs = t5.encode_packed(qext)
            logits = t5.decode(dec, enc, seqs.grouped(topk))
            loss, _ = emdr2_loss(logits, None, None, labels, mask, t5_tok.eos_token_id)
            loss.backward()
            opt.finish(); opt.step(lr=READER_LR * min(1.0, (step + 1) / 50.0))
            if (step + 1) % 200 == 0:
                log("reader warm-up step %d: %d passages per question, lm loss %.3f" % (step + 1, topk, float(loss)))
    finally:
        K.GRAD_SINK = None
    t5.eval()
    t5d = _save_pretrained(tmp, "t5", checkpointing.t5_state_dict(t5))
    K.WEIGHTS.invalidate()
    return de, dpr, t5d


def _arena(tmp, ev):
    from emdr2_amd.data import indexed_dataset
    from emdr2_amd.data.evidence_arena import EvidenceArena
    from emdr2_amd.tasks.openqa.e2eqa.run import read_evidence_titles
    passages = indexed_dataset.make_dataset(os.path.join(tmp, "text"), impl="infer", skip_warmup=True)
    titles = indexed_dataset.make_dataset(os.path.join(tmp, "title"), impl="infer", skip_warmup=True)
    return EvidenceArena.from_indexed(passages, titles, read_evidence_titles(ev)).to_device()


def index_and_recall(tmp, vocab, ev, dual_encoder, name, ks=(1, 5, 20)):
    """The indexer job (megatron/indexer_emdr2.py:77-114: every passage through the context tower -> --embedding-path pickle) followed by the
    retrieval evaluator (evaluate.py:42-134) on the validation questions -> ({k: accuracy}, embedding path)."""
    from emdr2_amd.data.emdr2_index import OpenRetreivalDataStore
    from emdr2_amd.indexer_emdr2 import IndexBuilder
    from emdr2_amd.tasks.openqa.dense_retriever.evaluation.evaluate import OpenRetrievalEvaluator, read_evidence_text
    from emdr2_amd.tokenizer import BertWordPieceTokenizer
    t = BertWordPieceTokenizer(vocab)
    emb = os.path.join(tmp, name + ".pkl")
    IndexBuilder(dual_encoder.context_model, _arena(tmp, ev), S_RET, t.cls, t.sep, t.pad, batch_size=128, log_interval=1 << 30).build_and_save_index(emb)
    store = OpenRetreivalDataStore(emb, load_from_path=True)
    evaluator = OpenRetrievalEvaluator(dual_encoder.query_model, t, store, read_evidence_text(ev), hidden_size=HIDDEN, seq_length_ret=S_RET, batch_size=128,
                                       topk_retrievals=max(ks), report_topk_accuracies=ks)
    acc, _ = evaluator.evaluate(os.path.join(tmp, "valid.tsv"), split="valid (%s)" % name)
    return acc, emb


def argv(tmp, vocab, ev, emb, dpr, t5, epochs, batch=16, topk=16, lr=1e-3, update_retriever=True, reload_interval=25, extra=()):
    a = ["--task", "OPENQA", "--num-layers", str(LAYERS), "--hidden-size", str(HIDDEN), "--num-attention-heads", str(HEADS), "--kv-channels",
         str(HIDDEN // HEADS), "--ffn-hidden-size", str(FFN), "--model-parallel-size", "1", "--train-data", os.path.join(tmp, "train.tsv"),
         "--valid-data", os.path.join(tmp, "valid.tsv"), "--evidence-data-path", ev, "--indexed-evidence-data-path", os.path.join(tmp, "text"),
         "--indexed-title-data-path", os.path.join(tmp, "title"), "--embedding-path", emb, "--pretrained-dpr-load", dpr, "--pretrained-t5-load", t5,
         "--log-interval", "25", "--eval-interval", "100000", "--weight-decay", "1.0e-2", "--seq-length", str(S), "--seq-length-ret", str(S_RET),
         "--decoder-seq-length", str(L), "--max-decode-len", "8", "--max-position-embeddings", "64", "--fp16", "--vocab-file", vocab, "--num-workers", "0",
         "--distributed-backend", "nccl", "--tokenizer-type", "BertWordPieceLowerCase", "--epochs", str(epochs), "--sample-rate", "1.0",
         "--batch-size", str(batch), "--eval-batch-size", "50", "--beam-size", "1", "--lr", str(lr), "--warmup", "0.05", "--DDP-impl", "local",
         "--lr-decay-style", "linear", "--max-training-rank", "1", "--faiss-use-gpu", "--topk-retrievals", str(topk), "--emdr2-training",
         "--retriever-score-scaling", "--allow-trivial-doc", "--async-indexer", "--index-reload-interval", str(reload_interval),
         "--indexer-batch-size", "128", "--init-method-std", str(INIT_STD), "--hidden-dropout", "0.0", "--attention-dropout", "0.0", "--clip-grad", "1.0"]
    if update_retriever:
        a.append("--update-retriever")
    return a + list(extra)


def prepare(tmp, n_docs=2000, topk=8, seed=0, retriever_steps=3, reader_steps=400):
    """World + pre-trained checkpoints + the initial index: everything `train` needs, shared by a run and its control."""
    vocab, ev = make_world(tmp, n_docs=n_docs, seed=seed)
    de0, dpr, t5 = write_pretrained(tmp, vocab, ev, n_docs, retriever_steps=retriever_steps, reader_steps=reader_steps, topk=topk)
    before, emb = index_and_recall(tmp, vocab, ev, de0, "emb_initial")
    del de0
    return dict(vocab=vocab, ev=ev, dpr=dpr, t5=t5, emb=emb, recall_before=before, n_docs=n_docs, topk=topk)


def train(tmp, world, steps=300, update_retriever=True, batch=16, lr=2e-4, extra=()):
    """`emdr2_amd.tasks.run` for `steps` steps from the prepared world -> dict(recall_before, recall_after, em, questions, steps)."""
    from emdr2_amd.tasks import run as task_run
    vocab, ev, dpr, t5, emb, before, n_docs, topk = (world[k] for k in ("vocab", "ev", "dpr", "t5", "emb", "recall_before", "n_docs", "topk"))
    iters_per_epoch = n_docs // batch
    epochs = max(1, (steps + iters_per_epoch - 1) // iters_per_epoch)
    a = argv(tmp, vocab, ev, emb, dpr, t5, epochs, batch=batch, topk=topk, lr=lr, update_retriever=update_retriever,
             extra=["--exit-interval", str(steps)] + list(extra))
    model, results = task_run.main(a)
    stats, total = results["validation"]
    after, _ = index_and_recall(tmp, vocab, ev, model.retriever_model, "emb_final_%d" % int(update_retriever))
    em_top1 = em_reading_the_top_passage(model, os.path.join(tmp, "valid.tsv"))
    from emdr2_amd.model import kernels as K
    K.GRAD_SINK = None
    return {"recall_before": before, "recall_after": after, "em_top1": em_top1, "em_fid": float(stats["Exact Match Score"]) / max(int(total), 1),
            "questions": int(total), "steps": steps, "update_retriever": bool(update_retriever)}


def em_reading_the_top_passage(model, qa_file):
    """Exact match of the greedy answers (train_e2eqa.py:216-283, the reference's EM scorer) when the reader is handed the ONE passage the
    retriever ranks first -- the open-book setting this world's reader was pre-trained for (choosing among K passages is a skill the 4-layer
    reader does not acquire in a test-sized budget: 3,000 steps at K = 2 plateau at the cannot-choose loss, tools/planted_task.py history)."""
    from emdr2_amd.global_vars import get_args, get_t5_tokenizer, get_tokenizer
    from emdr2_amd.tasks.openqa.e2eqa.train_data_utils import OpenQADataset
    from emdr2_amd.tasks.openqa.e2eqa.train_e2eqa import build_data_loader, reader_em_score
    args = get_args()
    retr = model.evidence_retriever
    saved = (retr.topk, retr.args.topk_retrievals, model.topk)
    retr.topk, retr.args.topk_retrievals, model.topk = 1, 1, 1
    try:
        ds = OpenQADataset("OPENQA_DATASET", "valid", [qa_file], get_tokenizer(), args.seq_length_ret, args.decoder_seq_length, seed=args.seed)
        loader = build_data_loader(ds, args.eval_batch_size, num_workers=0, drop_last=False, shuffle=False)
        stats, total = reader_em_score(model, loader, 1, get_t5_tokenizer())
    finally:
        retr.topk, retr.args.topk_retrievals, model.topk = saved
    return float(stats["Exact Match Score"]) / max(int(total), 1)


def run(tmp, steps=300, update_retriever=True, n_docs=2000, batch=16, topk=8, lr=2e-4, seed=0, retriever_steps=3, reader_steps=400):
    return train(tmp, prepare(tmp, n_docs, topk, seed, retriever_steps, reader_steps), steps, update_retriever, batch, lr)


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--docs", type=int, default=2000)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--topk", type=int, default=16)
    ap.add_argument("--lr", type=float, default=2e-4)
    ap.add_argument("--retriever-steps", type=int, default=5)
    ap.add_argument("--reader-steps", type=int, default=400)
    ap.add_argument("--no-update-retriever", action="store_true")
    ap.add_argument("--both", action="store_true", help="the run AND its control from the same prepared world")
    o = ap.parse_args()
    w = prepare(o.dir, o.docs, o.topk, 0, o.retriever_steps, o.reader_steps)
    for upd in ((True, False) if o.both else (not o.no_update_retriever,)):
        print(json.dumps(train(o.dir, w, o.steps, upd, o.batch, o.lr)), flush=True)
