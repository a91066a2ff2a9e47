"""CPU: the drop-in surface named by BASELINE.json's north_star ("keeping the tasks/openqa entry points and megatron.model retriever/reader
module API so it drops in under examples/openqa/emdr2_*.sh"):

* the verbatim flag lists of the three launch scripts (tests/golden/script_flags.json, produced from the reference's scripts by
  tests/golden/gen_script_flags.py) parse, with torch.distributed.launch's --local_rank appended;
* `tasks/run.py` exists at the repository root and dispatches --task OPENQA (reference tasks/run.py:49-67);
* the reference's import paths resolve: megatron.get_args, megatron.model.{EMDR2Model, PreComputedEvidenceDocsRetriever},
  megatron.data.emdr2_index.{OpenRetreivalDataStore, DistributedBruteForceIndex, FaissMIPSIndex}, tasks.openqa.e2eqa.{run, train_e2eqa};
* constructor / method signatures match the reference's call sites (tasks/openqa/e2eqa/run.py:33-39: `EMDR2Model(evidence_retriever)`,
  `PreComputedEvidenceDocsRetriever()`; emdr2_model.py:228: `load_state_dict(state_dict, strict=True)`).
"""
import inspect
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = json.load(open(os.path.join(ROOT, "tests", "golden", "script_flags.json")))


@pytest.mark.parametrize("script", sorted(FLAGS))
def test_launch_script_flags_parse(script, capsys):
    from emdr2_amd import arguments
    argv = FLAGS[script]["argv"] + ["--local_rank=3"]                       # what torch.distributed.launch appends per process
    args = arguments.parse_args(argv)
    assert args.task == "OPENQA" and args.num_layers == 12 and args.hidden_size == 768 and args.num_attention_heads == 12
    assert args.seq_length == 512 and args.seq_length_ret == 256 and args.decoder_seq_length == 32 and args.topk_retrievals == 50
    assert args.batch_size == (4 if "webq" in script else 8) and args.local_rank == 3 and args.checkpoint_activations and args.async_indexer
    assert args.emdr2_training and args.update_retriever and args.retriever_score_scaling and args.allow_trivial_doc
    assert args.weight_decay == 0.1 and args.lr == 2e-5 and args.warmup == 0.01 and args.clip_grad == 1.0 and args.index_reload_interval == 500
    assert args.embedding_path.endswith(".pkl") and args.train_data == [FLAGS[script]["argv"][1]]
    # --fp16 is accepted and its meaning here is stated, not silent
    assert args.fp16 and args.params_dtype == "bf16" and args.master_dtype == "fp32"
    assert "--fp16 requested -> bf16" in capsys.readouterr().out


def test_unknown_flag_is_rejected():
    from emdr2_amd import arguments
    with pytest.raises(SystemExit):
        arguments.parse_args(FLAGS["emdr2_nq.sh"]["argv"] + ["--no-such-flag"])


def test_reference_import_paths_and_signatures():
    sys.path.insert(0, ROOT)
    import megatron
    from megatron.model import EMDR2Model, PreComputedEvidenceDocsRetriever, DualEncoderModel, T5Model  # noqa: F401
    from megatron.data.emdr2_index import OpenRetreivalDataStore, DistributedBruteForceIndex, FaissMIPSIndex
    from megatron.checkpointing import save_checkpoint, load_checkpoint  # noqa: F401
    from tasks.openqa.e2eqa.run import main  # noqa: F401
    from tasks.openqa.e2eqa.train_e2eqa import train, _cross_entropy_forward_step  # noqa: F401
    assert callable(megatron.get_args) and callable(megatron.print_rank_0)
    # EMDR2Model(evidence_retriever): every other parameter is optional
    ps = list(inspect.signature(EMDR2Model.__init__).parameters.values())[1:]
    assert ps[0].name == "evidence_retriever" and all(p.default is not inspect.Parameter.empty for p in ps[1:])
    # forward keeps the reference's argument list (emdr2_model.py:87-92)
    assert list(inspect.signature(EMDR2Model.forward).parameters)[1:] == [
        "query_uid", "query_ids_bert", "query_types", "query_mask_bert", "query_ids_t5", "query_ids_t5_len", "dec_ids",
        "all_query_context_hidden_states", "all_query_context_ids_unflat", "topk_log_probs"]
    for name in ("state_dict_for_save_checkpoint", "load_state_dict", "init_state_dict_from_dpr_and_t5"):
        assert callable(getattr(EMDR2Model, name))
    assert list(inspect.signature(EMDR2Model.load_state_dict).parameters)[1:3] == ["state_dict", "strict"]
    # PreComputedEvidenceDocsRetriever(): no required arguments; the reference's plug-in methods
    assert all(p.default is not inspect.Parameter.empty for p in list(inspect.signature(PreComputedEvidenceDocsRetriever.__init__).parameters.values())[1:])
    for name in ("get_topk", "update_evidence_embedding"):
        assert callable(getattr(PreComputedEvidenceDocsRetriever, name))
    # index operator API (emdr2_index.py:200-305) and store API (:16-100)
    for cls in (DistributedBruteForceIndex, FaissMIPSIndex):
        for name in ("search_mips_index", "update_index", "reset_index", "add_embed_data"):
            assert callable(getattr(cls, name)), (cls, name)
    assert list(inspect.signature(DistributedBruteForceIndex.search_mips_index).parameters)[1:] == ["query_embeds", "top_k", "reconstruct"]
    assert list(inspect.signature(OpenRetreivalDataStore.__init__).parameters)[1:] == ["embedding_path", "load_from_path", "rank"]
    for name in ("add_block_data", "save_shard", "merge_shards_and_save", "clear", "load_from_file", "state"):
        assert callable(getattr(OpenRetreivalDataStore, name))


def test_tasks_run_py_is_the_launch_target():
    """`python tasks/run.py ...` from the repository root (what the scripts execute): without a GPU it must get as far as the loud
    'needs a GPU' error of the product path -- i.e. the script exists, parses the reference's flags and dispatches -- and never fall back."""
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", RANK="0", WORLD_SIZE="1")
    out = subprocess.run([sys.executable, "tasks/run.py"] + FLAGS["emdr2_nq.sh"]["argv"], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, timeout=300).stdout.decode()
    assert "bf16 activations" in out                                        # flags parsed, precision mode announced
    import torch
    if not torch.cuda.is_available():
        assert "needs a GPU" in out and "NativeError" in out


def test_data_store_round_trip_and_merge(tmp_path):
    """OpenRetreivalDataStore (emdr2_index.py:16-100): pickle layout, fp16 cast, overwrite guard, shard + merge with disjointness check."""
    import pickle
    import numpy as np
    from emdr2_amd.data.emdr2_index import OpenRetreivalDataStore
    path = str(tmp_path / "emb.pkl")
    stores = [OpenRetreivalDataStore(path, load_from_path=False, rank=r) for r in range(3)]
    rng = np.random.default_rng(0)
    for r, st in enumerate(stores):
        st.add_block_data(range(10 * r + 1, 10 * r + 11), rng.standard_normal((10, 8)).astype(np.float32))
        with pytest.raises(ValueError):
            st.add_block_data([10 * r + 1], np.zeros((1, 8)))
        st.save_shard()
    assert sorted(os.listdir(stores[0].temp_dir_name)) == ["0.pkl", "1.pkl", "2.pkl"] and stores[0].temp_dir_name == str(tmp_path / "emb_tmp")
    stores[1].merge_shards_and_save()
    raw = pickle.load(open(path, "rb"))
    assert list(raw) == ["embed_data"] and sorted(raw["embed_data"]) == list(range(1, 31))
    assert all(v.dtype == np.float16 and v.shape == (8,) for v in raw["embed_data"].values())
    assert not os.path.exists(stores[0].temp_dir_name)
    again = OpenRetreivalDataStore(path, load_from_path=True, rank=0)
    assert np.array_equal(again.embed_data[17], stores[1].embed_data[17])
    # overlapping shards are refused
    a, b = OpenRetreivalDataStore(path, False, 0), OpenRetreivalDataStore(path, False, 1)
    a.add_block_data([1], np.zeros((1, 8))); b.add_block_data([1], np.ones((1, 8)))
    a.save_shard(); b.save_shard()
    with pytest.raises(AssertionError):
        a.merge_shards_and_save()


def test_store_files_written_by_the_reference_are_read_and_merged(tmp_path):
    """F7: tests/golden/store_ref.pkl / store_ref_shard1.pkl were WRITTEN by the reference's OpenRetreivalDataStore.save_shard /
    merge_shards_and_save (emdr2_index.py:63-100; gen_store_optim_golden.py).  Ours must read the merged file (same ids, same insertion
    order, rows = fp16 of what went in), merge a reference-written shard with one of its own, and write a file the same bytes decode from."""
    import pickle
    import shutil
    import numpy as np
    from emdr2_amd.data.emdr2_index import OpenRetreivalDataStore
    gold = os.path.join(os.path.dirname(__file__), "golden")
    ref = np.load(os.path.join(gold, "store_ref.npz"))
    rows16 = ref["rows"].astype(np.float16)
    path = str(tmp_path / "emb.pkl")
    shutil.copy(os.path.join(gold, "store_ref.pkl"), path)
    st = OpenRetreivalDataStore(path, load_from_path=True, rank=0)
    assert list(st.embed_data) == ref["merged_order"].tolist() == ref["ids"].tolist()
    for i, doc in enumerate(ref["ids"].tolist()):
        v = st.embed_data[doc]
        assert v.dtype == np.float16 and np.array_equal(v.view(np.uint16), rows16[i].view(np.uint16))
    # our rank 0 + the reference's rank-1 shard file -> the reference's merged store
    mine = OpenRetreivalDataStore(path, load_from_path=False, rank=0)
    mine.add_block_data(ref["ids"][:60].tolist(), ref["rows"][:60])
    mine.save_shard()
    shutil.copy(os.path.join(gold, "store_ref_shard1.pkl"), os.path.join(mine.temp_dir_name, "1.pkl"))
    mine.merge_shards_and_save()
    ours, theirs = pickle.load(open(path, "rb")), pickle.load(open(os.path.join(gold, "store_ref.pkl"), "rb"))
    assert list(ours) == list(theirs) == ["embed_data"] and list(ours["embed_data"]) == list(theirs["embed_data"])
    assert all(np.array_equal(ours["embed_data"][k].view(np.uint16), theirs["embed_data"][k].view(np.uint16)) for k in theirs["embed_data"])
