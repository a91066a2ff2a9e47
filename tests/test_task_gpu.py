"""GPU: the task entry point end to end on a synthetic corpus: `emdr2_amd.tasks.run` with the flag set of
examples/openqa/emdr2_nq.sh -- tokenizer, QA files, memory-mapped evidence, pickled embeddings, training with side-stream re-indexing,
checkpoints in the reference's layout, greedy-decoding EM evaluation, resume."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
WORDS = ("the of and in to was is for on as with by that at from his he it an are which this were be or had first one their has new its who not "
         "but also after two they have been other when during all into there time may more years over only school city world emperor capital "
         "river paris france what year did war end where born wrote song play").split()


def _make_world(tmp, n_docs=300, n_train=24, n_valid=8, dim=128):
    from emdr2_amd.data.emdr2_index import OpenRetreivalDataStore
    from emdr2_amd.data.indexed_dataset import MMapIndexedDatasetBuilder
    from emdr2_amd.tokenizer import BertWordPieceTokenizer
    rng = np.random.default_rng(0)
    vocab = os.path.join(GOLD, "tokenizer_vocab.txt")
    t = BertWordPieceTokenizer(vocab)
    sent = lambda n: " ".join(rng.choice(WORDS, size=n))
    ev = os.path.join(tmp, "psgs.tsv")
    pb = MMapIndexedDatasetBuilder(os.path.join(tmp, "text.bin")); tb = MMapIndexedDatasetBuilder(os.path.join(tmp, "title.bin"))
    with open(ev, "w") as f:
        f.write("id\ttext\ttitle\n")
        title = sent(2)
        for d in range(1, n_docs + 1):
            if rng.random() < 0.4:
                title = sent(int(rng.integers(1, 4)))
            text = sent(int(rng.integers(20, 40)))
            f.write("%d\t%s\t%s\n" % (d, text, title))
            pb.add_item(t.tokenize(text)); pb.end_document()
            tb.add_item(t.tokenize(title)); tb.end_document()
    pb.finalize(os.path.join(tmp, "text.idx")); tb.finalize(os.path.join(tmp, "title.idx"))
    for name, n in (("train", n_train), ("valid", n_valid)):
        with open(os.path.join(tmp, name + ".tsv"), "w") as f:
            for _ in range(n):
                f.write("%s ?\t%s\n" % (sent(int(rng.integers(4, 9))), json.dumps([sent(int(rng.integers(1, 3))) for _ in range(int(rng.integers(1, 3)))])))
    emb = os.path.join(tmp, "emb.pkl")
    store = OpenRetreivalDataStore(emb, load_from_path=False, rank=0)
    store.add_block_data(np.arange(1, n_docs + 1), rng.standard_normal((n_docs, dim)).astype(np.float16))
    store.save_shard(); store.merge_shards_and_save()
    return vocab, ev, emb


def _argv(tmp, vocab, ev, emb, extra=()):
    return ["--task", "OPENQA", "--num-layers", "2", "--hidden-size", "128", "--num-attention-heads", "2", "--kv-channels", "64",
            "--ffn-hidden-size", "256", "--model-parallel-size", "1", "--train-data", os.path.join(tmp, "train.tsv"), "--valid-data",
            os.path.join(tmp, "valid.tsv"), "--test-data", os.path.join(tmp, "valid.tsv"), "--evidence-data-path", ev,
            "--indexed-evidence-data-path", os.path.join(tmp, "text"), "--indexed-title-data-path", os.path.join(tmp, "title"),
            "--save-interval", "500", "--save", os.path.join(tmp, "ckpt"), "--load", os.path.join(tmp, "ckpt"), "--embedding-path", emb,
            "--log-interval", "2", "--eval-interval", "500", "--eval-iters", "10", "--weight-decay", "1.0e-1", "--seq-length", "64",
            "--seq-length-ret", "32", "--decoder-seq-length", "32", "--max-decode-len", "32", "--max-position-embeddings", "64", "--fp16",
            "--vocab-file", vocab, "--num-workers", "0", "--distributed-backend", "nccl", "--checkpoint-activations", "--tokenizer-type",
            "BertWordPieceLowerCase", "--epochs", "1", "--sample-rate", "1.0", "--batch-size", "4", "--eval-batch-size", "4", "--beam-size", "1",
            "--lr", "2e-4", "--warmup", "0.01", "--DDP-impl", "local", "--lr-decay-style", "linear", "--max-training-rank", "1", "--faiss-use-gpu",
            "--topk-retrievals", "4", "--emdr2-training", "--retriever-score-scaling", "--update-retriever", "--allow-trivial-doc",
            "--async-indexer", "--index-reload-interval", "2", "--indexer-batch-size", "128", "--init-method-std", "0.05"] + list(extra)


def test_task_entry_point_trains_reindexes_checkpoints_evaluates_and_resumes(tmp_path, capsys):
    from emdr2_amd import checkpointing
    from emdr2_amd.tasks import run as task_run
    tmp = str(tmp_path)
    vocab, ev, emb = _make_world(tmp)
    model, results = task_run.main(_argv(tmp, vocab, ev, emb))
    out = capsys.readouterr().out
    assert "MIPS Index Updated" in out and "lm_loss" in out and "Exact Match Score" in out
    stats, total = results["validation"]
    # (this world's answers are random words no passage contains: nothing here can be learnt, only counted.  That training LEARNS -- retrieval
    # recall and exact match rise, and only with the retriever loss -- is tests/test_planted_task_gpu.py)
    assert total == 8 and float(stats["Exact Match Score"]) == int(stats["Exact Match Score"])
    it, release = checkpointing.read_tracker(os.path.join(tmp, "ckpt"))
    assert it == 6 and not release                                      # 24 questions / batch 4, one epoch
    state = torch.load(checkpointing.get_checkpoint_name(os.path.join(tmp, "ckpt"), it), map_location="cpu", weights_only=False)
    assert set(state) >= {"iteration", "model", "optimizer", "lr_scheduler"} and set(state["model"]) == {"encoder/t5_model", "retriever/biencoder_model"}
    w_saved = state["model"]["encoder/t5_model"]["language_model"]["encoder"]["layers.0.mlp.dense_h_to_4h.weight"]
    assert torch.equal(w_saved, model.language_model.language_model.encoder.layers[0].mlp.dense_h_to_4h.weight.detach().cpu())
    # resume: a second run loads iteration 6, has nothing left to train in epoch 1, and evaluates identically (eval mode is deterministic)
    model2, results2 = task_run.main(_argv(tmp, vocab, ev, emb))
    from emdr2_amd.global_vars import get_args
    assert get_args().iteration == 6
    for (k, p), (_, p2) in zip(model.named_parameters(), model2.named_parameters()):
        assert torch.equal(p, p2), k
    # the first run's evaluation used the index refreshed by the side stream; the resumed run loads the original pickle, so only the
    # question count is comparable
    assert results2["validation"][1] == 8


def test_checkpoint_layout_matches_the_reference_model(tmp_path):
    from emdr2_amd import checkpointing
    from emdr2_amd.model.emdr2_model import EMDR2Model
    from emdr2_amd.model.transformer import Config
    ref = json.load(open(os.path.join(GOLD, "ckpt_layout.json")))["layout"]
    meta = np.load(os.path.join(GOLD, "model_ref.npz"))["meta"]
    cfg = Config(num_layers=2, hidden_size=32, num_attention_heads=2, ffn_hidden_size=128, max_position_embeddings=64)
    model = EMDR2Model(None, cfg, int(meta[1]), int(meta[0]), 3, 48, 24, cls_id=2, sep_id=3)
    mine = {}

    def walk(d, prefix):
        for k, v in d.items():
            if isinstance(v, dict):
                walk(v, prefix + [k])
            else:
                mine["\t".join(prefix + [k])] = list(v.shape)
    nested = model.state_dict_for_save_checkpoint()
    walk(nested, [])
    assert mine == ref
    # round trip through a file in the reference's directory layout
    checkpointing.save_checkpoint(str(tmp_path), 7, model)
    other = EMDR2Model(None, cfg, int(meta[1]), int(meta[0]), 3, 48, 24, cls_id=2, sep_id=3)
    with torch.no_grad():
        for p in other.parameters():
            p.add_(1.0)
    assert checkpointing.load_checkpoint(str(tmp_path), other) == 7
    for (k, p), (_, q) in zip(model.named_parameters(), other.named_parameters()):
        assert torch.equal(p, q), k
    checkpointing.load_dualencoder_checkpoint(other.retriever_model, str(tmp_path), key_list=["retriever/biencoder_model"])
    checkpointing.load_t5_state_dict(other.language_model, nested["encoder/t5_model"])


def test_retrieval_evaluator_reports_topk_accuracy_on_a_planted_corpus(tmp_path):
    """f3: questions -> token ids -> query tower -> FaissMIPSIndex (fp32 scores) -> answer-string validation -> top-k accuracy.  Planted
    geometry: a stub tower embeds a sequence by its first word, every passage is indexed under its first word, and each question repeats
    its passage -- so the passage is retrieved among the (few) passages sharing that first word and top-5 accuracy must be exactly 1.
    The real BERT tower is run through the same evaluator afterwards (plumbing check: shapes, eval mode, monotone accuracies)."""
    from emdr2_amd.data.emdr2_index import OpenRetreivalDataStore
    from emdr2_amd.model.transformer import Config, PretrainedBertModel
    from emdr2_amd.tasks.openqa.dense_retriever.evaluation.evaluate import OpenRetrievalEvaluator, read_evidence_text
    from emdr2_amd.tokenizer import BertWordPieceTokenizer
    tmp = str(tmp_path)
    vocab, ev, emb = _make_world(tmp, n_docs=200)
    t = BertWordPieceTokenizer(vocab)
    docs = read_evidence_text(ev)
    g = torch.Generator(device="cuda").manual_seed(0)
    table = torch.randn((t.vocab_size, 128), generator=g, device="cuda").half()

    class FirstWordTower(torch.nn.Module):
        def forward(self, ids, types):
            return table[ids[:, 1]]

    store = OpenRetreivalDataStore(os.path.join(tmp, "planted.pkl"), load_from_path=False, rank=0)
    first = [t.tokenize(docs[d][0])[0] for d in range(1, 201)]
    store.add_block_data(list(range(1, 201)), table[torch.tensor(first, device="cuda")].cpu().numpy())
    qa = os.path.join(tmp, "retr.tsv")
    with open(qa, "w") as f:
        for d in range(1, 41):
            words = docs[d][0].split()
            f.write("%s\t%s\n" % (docs[d][0], json.dumps([" ".join(words[3:9])])))
    evaluator = OpenRetrievalEvaluator(FirstWordTower(), t, store, docs, hidden_size=128, seq_length_ret=64, batch_size=16, topk_retrievals=20,
                                       report_topk_accuracies=(1, 5, 20))
    acc, stats = evaluator.evaluate(qa)
    same_first = max(first.count(w) for w in set(first))
    assert same_first <= 20
    assert set(acc) == {1, 5, 20} and acc[1] <= acc[5] <= acc[20] == 1.0
    for qi, d in enumerate(range(1, 41)):                                # the passage itself is in the list, at a position below the tie group size
        ids = evaluator.mips_index.search_mips_index(table[first[d - 1]][None], top_k=20, reconstruct=False)[1][0].tolist()
        assert d in ids[:first.count(first[d - 1])]
    # the real tower through the same evaluator
    torch.manual_seed(1)
    cfg = Config(num_layers=2, hidden_size=128, num_attention_heads=2, ffn_hidden_size=256, max_position_embeddings=64, init_method_std=0.2,
                 hidden_dropout=0.1, attention_dropout=0.1)
    tower = PretrainedBertModel(cfg, 384).train()
    store2 = OpenRetreivalDataStore(os.path.join(tmp, "planted2.pkl"), load_from_path=False, rank=0)
    store2.add_block_data(list(range(1, 201)), table[torch.tensor(first, device="cuda")].cpu().numpy())
    ev2 = OpenRetrievalEvaluator(tower, t, store2, docs, hidden_size=128, seq_length_ret=64, batch_size=16, topk_retrievals=20, report_topk_accuracies=(1, 20))
    q1, q2 = ev2.generate_query_vectors([docs[1][0], docs[2][0]]), ev2.generate_query_vectors([docs[1][0], docs[2][0]])
    assert q1.shape == (2, 128) and torch.equal(q1, q2) and tower.training      # eval mode inside (no dropout), training mode restored
    acc2, _ = ev2.evaluate(qa)
    assert 0.0 <= acc2[1] <= acc2[20] <= 1.0


def test_weight_decay_groups_of_our_model_are_the_reference_groups():
    """F6 (second half) on the real thing: our EMDR2Model registers the parameters of the reference's EMDR2Model under the same names, and
    get_params_for_weight_decay_optimization / FlatAdam's decayed-first bucket layout split them as the reference's function does
    (megatron/model/utils.py:64-83; tests/golden/optim_groups.json written by the reference's function)."""
    from emdr2_amd.model.emdr2_model import EMDR2Model
    from emdr2_amd.model.transformer import Config
    from emdr2_amd.training import FlatAdam, get_params_for_weight_decay_optimization
    ref = json.load(open(os.path.join(GOLD, "optim_groups.json")))
    meta = np.load(os.path.join(GOLD, "model_ref.npz"))["meta"]
    cfg = Config(num_layers=2, hidden_size=32, num_attention_heads=2, ffn_hidden_size=128, max_position_embeddings=64)
    m = EMDR2Model(None, cfg, int(meta[1]), int(meta[0]), 3, 48, 24, cls_id=2, sep_id=3)
    names = {id(p): n for n, p in m.named_parameters()}
    assert sorted(names.values()) == sorted(ref["weight_decay"] + ref["no_weight_decay"])
    decay, no_decay = get_params_for_weight_decay_optimization(m)
    assert sorted(names[id(p)] for p in decay["params"]) == ref["weight_decay"]
    assert sorted(names[id(p)] for p in no_decay["params"]) == ref["no_weight_decay"] and no_decay["weight_decay"] == 0.0
    opt = FlatAdam(m)
    for b in opt.buckets:                                   # inside a bucket: decayed parameters first, `split` marks the boundary
        for p in b["params"]:
            o = opt.slot[p][1]
            assert (o < b["split"]) == (names[id(p)] in ref["weight_decay"]) or not p.requires_grad


def test_task_entry_point_on_two_ranks_with_question_groups_and_the_refresher(tmp_path):
    """The task driver as the shipped scripts launch it, on TWO ranks (gloo transport, both on the test GPU): `--embedding-path` is unpickled by
    rank 0 only and loaded through its flat twin, the index is row-sharded, every search is all-gather(queries) + record all-gather + merge,
    the step runs in question groups (`--question-micro-batches 2`: forward + backward per group, bucketed bf16 gradient exchange on the last
    group's backward), the side-stream refreshers of the two ranks agree on the swap (MIN all-reduce in `maybe_swap`), checkpoints and the
    EM evaluation run on both ranks."""
    import subprocess
    import sys
    from emdr2_amd import checkpointing
    tmp = str(tmp_path)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EMDR2_SINGLE_DEVICE="1", EMDR2_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "tools", "dryrun_task.py"), tmp, "--question-micro-batches", "2"]
    out = subprocess.run(cmd, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    text = out.stdout.decode()
    assert out.returncode == 0, text[-3000:]
    assert "rank 0 done" in text and "rank 1 done" in text
    assert "MIPS Index Updated" in text and "lm_loss" in text and "Exact Match Score" in text
    assert os.path.exists(os.path.join(tmp, "emb.flat"))                 # written once by rank 0, mapped by both
    it, release = checkpointing.read_tracker(os.path.join(tmp, "ckpt"))
    assert it == 3 and not release                                        # 24 questions / (batch 4 x 2 ranks), one epoch
