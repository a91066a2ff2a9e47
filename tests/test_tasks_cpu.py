"""CPU: QA batch layout (a1) against the reference's own builder, LR schedule and weight-decay grouping of the optimizer step."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_build_tokens_matches_reference_builder():
    from emdr2_amd.tasks.openqa.e2eqa.train_data_utils import build_tokens_types_paddings_from_ids
    g = np.load(os.path.join(GOLD, "a1_ref.npz"), allow_pickle=True)
    for (q, a), ref in zip(g["cases"], g["outs"]):
        got = build_tokens_types_paddings_from_ids(list(q), list(a), 24, 8, 2, 3, 0, 250, 251)
        assert [list(x) if isinstance(x, (list, tuple)) else x for x in got] == [list(x) if isinstance(x, (list, tuple)) else x for x in ref]


def test_collate_has_the_ten_reference_keys():
    from emdr2_amd.tasks.openqa.e2eqa.train_data_utils import build_sample, collate
    items = [build_sample(-(i + 1), [7, 8, 9 + i], [11, 12], 24, 8, 2, 3, 0, 250, 251, reference=["x"]) for i in range(3)]
    b = collate(items)
    assert list(b.keys()) == ["query_uid", "query_ids_bert", "query_types", "query_mask_bert", "query_ids_t5", "query_ids_t5_len",
                              "dec_ids", "labels", "loss_mask", "reference"]
    assert b["query_ids_bert"].shape == (3, 24) and b["query_mask_bert"].shape == (3, 24, 24) and b["loss_mask"].dtype == torch.float32
    assert b["dec_ids"][0].tolist()[:3] == [250, 11, 12] and b["labels"][0].tolist()[:3] == [11, 12, 251]


def test_annealing_lr_matches_reference_table():
    from emdr2_amd.training import AnnealingLR
    g = np.load(os.path.join(GOLD, "model_ref.npz"))
    s = AnnealingLR(2e-5, 10, 1000)
    ours = [s.step() for _ in range(1000)]
    np.testing.assert_allclose(ours, g["lr_table"], rtol=1e-12, atol=0)


def test_weight_decay_groups():
    from emdr2_amd.training import get_params_for_weight_decay_optimization

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.dense = torch.nn.Linear(4, 4)
            self.input_layernorm = torch.nn.LayerNorm(4)
    groups = get_params_for_weight_decay_optimization(M())
    assert len(groups[0]["params"]) == 1 and len(groups[1]["params"]) == 3 and groups[1]["weight_decay"] == 0.0


def test_exact_match_metric_matches_reference_pairs():
    import json
    import os
    from emdr2_amd.tasks.openqa.e2eqa.eval_utils import exact_match_score, metric_max_over_ground_truths
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ckpt_layout.json")))
    got = [bool(metric_max_over_ground_truths(exact_match_score, h, r)) for h, r in ref["em_pairs"]]
    assert got == ref["em"]


def test_argument_parser_accepts_the_reference_script_flags_and_rejects_unknown_ones():
    import pytest
    from emdr2_amd import arguments
    base = ("--task OPENQA --num-layers 12 --hidden-size 768 --num-attention-heads 12 --kv-channels 64 --ffn-hidden-size 3072 --model-parallel-size 1 "
            "--train-data a.tsv --valid-data b.tsv --test-data c.tsv --evidence-data-path e --indexed-evidence-data-path x --indexed-title-data-path y "
            "--save-interval 500 --save s --load s --pretrained-t5-load p --pretrained-dpr-load d --stale-checkpoint-path d --embedding-path e "
            "--log-interval 20 --eval-interval 500 --eval-iters 10 --weight-decay 1.0e-1 --seq-length 512 --seq-length-ret 256 --decoder-seq-length 32 "
            "--max-decode-len 32 --max-position-embeddings 512 --fp16 --vocab-file v --num-workers 2 --distributed-backend nccl "
            "--checkpoint-activations --tokenizer-type BertWordPieceLowerCase --epochs 10 --sample-rate 1.0 --batch-size 8 --eval-batch-size 8 "
            "--beam-size 1 --lr 2e-5 --warmup 0.01 --DDP-impl local --lr-decay-style linear --max-training-rank 8 --faiss-use-gpu "
            "--topk-retrievals 50 --emdr2-training --retriever-score-scaling --update-retriever --allow-trivial-doc --async-indexer "
            "--index-reload-interval 500").split()
    a = arguments.parse_args(base)
    assert (a.topk_retrievals, a.hidden_dropout, a.attention_dropout, a.weight_decay, a.clip_grad, a.seed) == (50, 0.1, 0.1, 0.1, 1.0, 1234)
    assert a.train_data == ["a.tsv"] and a.index_reload_interval == 500 and a.indexer_batch_size == 128 and a.async_indexer
    with pytest.raises(SystemExit):
        arguments.parse_args(base + ["--no-such-flag"])
    with pytest.raises(NotImplementedError):
        arguments.parse_args(base[:-2] + ["--model-parallel-size", "2"])


def test_openqa_dataset_reads_the_reference_file_format(tmp_path):
    import os
    from emdr2_amd.tasks.openqa.e2eqa.train_data_utils import OpenQADataset, collate
    from emdr2_amd.tokenizer import BertWordPieceTokenizer
    t = BertWordPieceTokenizer(os.path.join(os.path.dirname(__file__), "golden", "tokenizer_vocab.txt"))
    p = tmp_path / "qa.tsv"
    p.write_text('who was the first emperor\t["the emperor", "emperor"]\nwhat year did the war end\t["1999"]\n')
    ds = OpenQADataset("OPENQA", "t", [str(p)], t, 16, 8, seed=1)
    assert len(ds) == 2 and ds.samples[1]["uid"] == -2 and ds.samples[0]["answers"] == ["the emperor", "emperor"]
    s = ds[0]
    assert s["query_uid"] == -1 and s["query_ids_bert"][0] == t.cls and len(s["query_ids_bert"]) == 16 and len(s["dec_ids"]) == 8
    assert s["dec_ids"][0] == t.bos_token_id and t.eos_token_id in s["labels"] and s["reference"] == ["the emperor", "emperor"]
    b = collate([ds[0], ds[1]])
    assert b["query_ids_bert"].shape == (2, 16) and b["loss_mask"].dtype.is_floating_point and b["query_uid"].tolist() == [-1, -2]


def test_checkpoint_files_carry_version_and_release_checkpoints_restart_the_run(tmp_path, monkeypatch):
    """(i) files written here must say checkpoint_version 1.0 (the reference re-orders the QKV rows of version-0 files on load,
    megatron/checkpointing.py:236 + transformer.py:225-248); (ii) a 'release' tracker means iteration 0 and no optimizer / schedule state
    (checkpointing.py:196-204,243-256); (iii) version-0 files are refused instead of being read with scrambled rows."""
    import pytest
    from emdr2_amd import checkpointing as ck

    monkeypatch.setattr(ck, "emdr2_state_dict", lambda model: {"w": torch.ones(2)})
    monkeypatch.setattr(ck, "_invalidate_weight_caches", lambda: None)
    loaded = {}
    monkeypatch.setattr(ck, "load_emdr2_state_dict", lambda model, sd, strict=True: loaded.update(sd))

    class Opt(object):
        def __init__(self): self.got = None
        def state_dict(self): return {"step": 3, "state": {}}
        def load_state_dict(self, sd): self.got = sd

    class Sched(Opt):
        def state_dict(self): return {"num_iters": 7}

    d = str(tmp_path / "ckpt")
    ck.save_checkpoint(d, 40, object(), Opt(), Sched(), args={"lr": 1.0})
    raw = torch.load(ck.get_checkpoint_name(d, 40), weights_only=False)
    assert raw["checkpoint_version"] == 1.0 and raw["iteration"] == 40 and raw["args"] == {"lr": 1.0}
    opt, sched = Opt(), Sched()
    assert ck.load_checkpoint(d, object(), opt, sched) == 40 and opt.got["step"] == 3 and sched.got["num_iters"] == 7
    # without the optimizer the schedule is not restored either (reference: both or neither)
    sched2 = Sched()
    assert ck.load_checkpoint(d, object(), None, sched2) == 40 and sched2.got is None
    # release: weights only, iteration 0
    import os, shutil
    os.makedirs(os.path.dirname(ck.get_checkpoint_name(d, 0, release=True)))
    shutil.copy(ck.get_checkpoint_name(d, 40), ck.get_checkpoint_name(d, 0, release=True))
    open(ck.get_checkpoint_tracker_filename(d), "w").write("release")
    opt, sched = Opt(), Sched()
    assert ck.load_checkpoint(d, object(), opt, sched) == 0 and opt.got is None and sched.got is None and "w" in loaded
    # version 0 is refused
    raw.pop("checkpoint_version")
    torch.save(raw, ck.get_checkpoint_name(d, 0, release=True))
    with pytest.raises(ValueError):
        ck.load_checkpoint(d, object())
    # ... but only on the resume path: the pretrained-model loaders read a file without the key as version 1.0, like the reference's
    # (checkpointing.py:267-340 never consult the version) -- ADVICE r2
    seen = {}
    monkeypatch.setattr(ck, "load_t5_state_dict", lambda model, sd: seen.setdefault("t5", sd))
    monkeypatch.setattr(ck, "load_dualencoder_state_dict", lambda model, sd, **kw: seen.setdefault("de", sd))
    ck.load_t5_checkpoint(object(), d)
    ck.load_dualencoder_checkpoint(object(), d)
    assert "t5" in seen and "de" in seen
    # a reference-format optimizer state gives a pointer to --no-load-optim instead of a KeyError
    class BadOpt(Opt):
        def load_state_dict(self, sd): raise KeyError("step")
    open(ck.get_checkpoint_tracker_filename(d), "w").write("40")
    with pytest.raises(ValueError, match="no-load-optim"):
        ck.load_checkpoint(d, object(), BadOpt(), None)


def test_weight_decay_predicate_reproduces_the_reference_groups():
    """F6 (second half): the parameter names of the two optimizer groups the REFERENCE's get_params_for_weight_decay_optimization
    (megatron/model/utils.py:64-83) forms on its EMDR2Model (tests/golden/optim_groups.json, gen_store_optim_golden.py): our name
    predicate -- used by get_params_for_weight_decay_optimization AND by FlatAdam's bucket layout -- splits the same names the same way.
    (tests/test_task_gpu.py builds our EMDR2Model and checks the groups of its actual parameters.)"""
    import json
    from emdr2_amd.training import FlatAdam
    ref = json.load(open(os.path.join(GOLD, "optim_groups.json")))
    every = sorted(ref["weight_decay"] + ref["no_weight_decay"])
    assert len(every) == len(set(every)) == 130 and ref["no_weight_decay_value"] == 0.0
    assert [n for n in every if FlatAdam.is_no_decay(n)] == ref["no_weight_decay"]
    assert [n for n in every if not FlatAdam.is_no_decay(n)] == ref["weight_decay"]


def test_retention_guard_reruns_then_thins_the_plan_in_a_fixed_order():
    """training.RetentionGuard (bench_e2e.py and the training task): a step that runs out of HBM is run again once with the allocator's blocks
    given back, then with less and less retained -- context tower, kept layers, the reader's selective layers -- and gives up only when
    the reference's full recompute does not fit either."""
    from emdr2_amd.training import RetentionGuard

    class Model:
        def set_recompute_keep_last(self, n): self.keep = n
        def set_selective_retention(self, r, c=0, q=0): self.sel = (r, c, q)

    class Opt:
        zeroed = 0
        def zero_grad(self): self.zeroed += 1
        def abort_step(self): pass
    m, o = Model(), Opt()
    g = RetentionGuard(m, o, keep=1, reader=4, context=3, query=2)
    assert m.keep == 1 and m.sel == (4, 3, 2)
    fails = [9]

    def step():
        if fails[0] > 0:
            fails[0] -= 1
            raise torch.cuda.OutOfMemoryError("injected")
        return "ok"
    fails[0] = 1
    assert g.run(step) == "ok" and g.reruns == 1 and g.plan["thinned"] == 0 and m.sel == (4, 3, 2)      # first failure: same plan again
    fails[0] = 5
    assert g.run(step) == "ok"
    assert g.plan["thinned"] == 4 and m.keep == 0 and m.sel == (2, 0, 0)          # context 3 -> 1 -> 0 (query with it), keep 1 -> 0, reader 4 -> 2
    fails[0] = 99
    with pytest.raises(torch.cuda.OutOfMemoryError):
        g.run(step)
    assert m.sel == (0, 0, 0)


def test_retention_guard_splits_the_step_finer_when_there_is_no_plan_to_thin():
    """r05: with question micro-batches nothing is re-run and nothing can be thinned: a step that still runs out of HBM is re-run once as it
    is, then with twice the groups (up to one question per group: r06, the groups need not divide the batch), by the same rule on every rank;
    the step function reads `guard.micro` or is told through `on_micro_change` (the task's --question-micro-batches flag)."""
    from emdr2_amd.training import RetentionGuard

    class Model:
        def set_recompute_keep_last(self, n): pass
        def set_selective_retention(self, r, c=0, q=0): pass

    class Opt:
        def zero_grad(self): pass
        def abort_step(self): pass
    g = RetentionGuard(Model(), Opt(), micro=4, batch=64)
    seen, fails = [], [3]

    def step():
        seen.append(g.micro)
        if fails[0] > 0:
            fails[0] -= 1
            raise torch.cuda.OutOfMemoryError("injected")
        return "ok"
    assert g.run(step) == "ok"
    assert seen == [4, 4, 8, 16] and g.reruns == 3 and g.plan["thinned"] == 2
    told = []
    g2 = RetentionGuard(Model(), Opt(), micro=4, batch=12, on_micro_change=told.append)      # 12 questions: 8 groups do not divide them -- allowed
    fails[0] = 99
    with pytest.raises(torch.cuda.OutOfMemoryError):
        g2.run(step)
    assert g2.micro == 12 and told == [8, 12]                            # 4 -> 8 -> 12 (one question per group), then nothing is left to split
