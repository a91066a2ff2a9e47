"""GPU, world size 8 on ONE device: the rehearsal of BASELINE configs[3] / [4].  The driver's 8-GPU commands -- the PLAIN
`python bench.py --gpus 8 ...` and `python bench_e2e.py --gpus 8 ...`, no launcher around them -- run as eight gloo ranks sharing cuda:0
(EMDR2_SINGLE_DEVICE=1 EMDR2_DIST_BACKEND=gloo): everything an RCCL run executes except the transport -- self-launch, the 8-way
`shard_bounds` of the 21,015,324-row index (2,626,916 rows per rank, the last one 2,626,912), the query all-gather, the record all-gather +
merge of the sharded search, `FlatAdam`'s bucket protocol at world 8 with the bf16 exchange, question groups, the out-of-memory abort
protocol, the side-stream refresher and its swap handshake.  Reference collectives: megatron/model/emdr2_model.py:435-470,
megatron/model/distributed.py:53-62, tasks/openqa/e2eqa/async_indexer.py:116-144.

Per-rank batch: B = 8 questions in 8 question groups (one question's activations alive at a time).  Eight ranks share ONE GPU's 288 GB:
8 x (optimizer state + working copies + exchange buffer of 440 M parameters 9.7 GB + index shard 4.0 GB + 21M-passage corpus 6.3 GB) = 160 GB
are fixed, a question group's activations at K = 50 are ~9 GB per rank; B = 64 per rank (configs[3] as written) needs the eight GPUs."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROWS = 21015324


def _env(extra=None):
    env = dict(os.environ, EMDR2_SINGLE_DEVICE="1", EMDR2_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra or {}))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def _one_line(cmd, env, timeout):
    import torch
    torch.cuda.empty_cache()                                             # (eight more processes are about to share this device)
    out = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    assert out.returncode == 0, out.stderr.decode()[-4000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                                        # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_plain_bench_command_with_eight_ranks_full_index_full_depth_and_an_injected_allocation_failure():
    """configs[3] dry: `python bench.py --gpus 8 --rows 21015324 --queries 512 --no-e2e-k100` (+ the per-rank batch that fits eight ranks on
    one GPU).  (1) MIPS half: N/8 row shards, ONE 16-byte-record all-gather, merged top-50 of all 512 queries bit-identical on the eight ranks
    and to a SINGLE-shard search of the same 21M rows (digest over scores + doc ids); nothing unproven.  (2) e2e half: 12 layers x 4
    stacks, K = 50, question groups with nothing recomputed, rank 5 hits an (injected) allocation failure in its FIRST step between a
    group's forward and backward: all eight give the step up together, re-run it, and after the timed steps hold bit-identical parameters."""
    import torch
    cmd = [sys.executable, "bench.py", "--gpus", "8", "--rows", str(ROWS), "--queries", "512", "--no-e2e-k100", "--steps", "2", "--warmup", "1",
           "--batch", "8", "--micro-batches", "8", "--e2e-steps", "2", "--e2e-warmup", "1", "--no-cpu-baseline", "--e2e-timeout", "900"]
    r = _one_line(cmd, _env({"EMDR2_BENCH_INJECT_OOM": "5,1"}), timeout=1500)
    from emdr2_amd.data.emdr2_index import shard_bounds
    c = r["config"]
    assert r["metric"] == "mips_queries_per_sec" and r["n_gpus"] == 8 and r["value"] > 0 and r["scaling"] == "strong"
    assert c["rows_per_rank"] == [b - a for a, b in shard_bounds(ROWS, 8)] == [2626916] * 7 + [2626912]
    assert c["unproven_queries"] == 0 and c["result_identical_on_all_ranks"] is True
    assert c["allgather_bytes_per_rank"] == 512 * 50 * 16 and c["allgather_plus_merge_ms"] > 0
    assert "error" not in r["clustered"], r["clustered"]
    assert r["clustered"]["unproven_queries"] <= 5 and r["clustered"]["fast_path_equals_exact_path_on_8_queries"]
    e = r["e2e"]
    assert "error" not in e, e
    ec = e["config"]
    assert e["n_gpus"] == 8 and e["steps_per_s"] > 0 and ec["global_batch"] == 64 and ec["params"] == 440388096
    assert "21015324-row index" in ec["workload"] and "12 layers" in ec["workload"] and ec["parallelism"] == "dp8 (index row-sharded x8)"
    assert ec["question_micro_batches"] == 8 and ec["recompute_tflop_per_step"] == 0
    assert ec["steps_rerun_after_out_of_memory"] == 1                    # the injected failure on rank 5, recovered by all eight
    cs = ec["replica_parameter_checksums"]
    assert len(cs) == 8 and len(set(cs)) == 1, cs
    import math
    assert math.isfinite(float(ec["loss"]))
    # the single-shard search of the same index and queries, in this process: same digest
    sys.path.insert(0, ROOT)
    import bench
    import bench_e2e
    index = bench_e2e.build_index(ROWS, 0, 1)
    gq = torch.Generator(device="cuda").manual_seed(4321)
    queries = torch.randn((512, bench.DIM), generator=gq, device="cuda", dtype=torch.float32).to(torch.float16)
    d, i = index.search_mips_index(queries, 50)
    single = bench.result_digest(d, i)
    del index, d, i
    torch.cuda.empty_cache()
    assert c["result_sha256"] == single


def test_plain_bench_e2e_command_with_eight_ranks_and_the_refresher_on():
    """configs[4] dry, the training half: `python bench_e2e.py --gpus 8 --reindex-rows-per-step 5254` -- every rank re-embeds rows of ITS
    shard on its side stream (42 batches of 128 per step: the 8-GPU pace for a 500-step reload interval) while the eight-rank step runs.
    An 8M-row index here: the second (spare) image of every shard and the refresher's weight snapshot would not fit eight ranks at 21M rows on
    one GPU; the refresher's work per step does not depend on the index size."""
    cmd = [sys.executable, "bench_e2e.py", "--gpus", "8", "--steps", "2", "--warmup", "1", "--rows", "8000000", "--batch", "8", "--micro-batches", "8",
           "--layers", "12", "--reindex-rows-per-step", "5254"]
    r = _one_line(cmd, _env(), timeout=1200)
    c = r["config"]
    assert r["metric"] == "qa_train_steps_per_sec" and r["n_gpus"] == 8 and r["value"] > 0 and r["scaling"] == "weak"
    assert c["global_batch"] == 64 and c["params"] == 440388096 and c["question_micro_batches"] == 8 and c["reindex_rows_per_step"] == 5254
    assert c["recompute_tflop_per_step"] == 0 and c["steps_rerun_after_out_of_memory"] == 0
    cs = c["replica_parameter_checksums"]
    assert len(cs) == 8 and len(set(cs)) == 1, cs


# ---- the swap boundary when the pass is ready on SOME ranks only ---------------------------------------------------
S_RET, CLS, SEP, PAD = 64, 101, 102, 0


def _swap_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    from emdr2_amd.data.emdr2_index import DistributedBruteForceIndex
    from emdr2_amd.data.evidence_arena import EvidenceArena
    from emdr2_amd.indexer_emdr2 import IndexBuilder
    from emdr2_amd.model.transformer import Config, PretrainedBertModel
    from emdr2_amd.tasks.openqa.e2eqa.async_indexer import AsyncIndexBuilder
    torch.manual_seed(3)
    cfg = Config(num_layers=2, hidden_size=128, num_attention_heads=2, ffn_hidden_size=256, max_position_embeddings=S_RET, init_method_std=0.2)
    model = PretrainedBertModel(cfg, 2000)
    n_docs = 1900                                       # torch.chunk bounds at world 8: 238 rows per rank, the last one 234
    arena = EvidenceArena.synthetic(n_docs, seed=5, vocab=2000)
    ids = np.arange(1, n_docs + 1, dtype=np.int32)
    index = DistributedBruteForceIndex(128, None)
    index.add_arrays(ids, np.zeros((n_docs, 128), dtype=np.float16))
    lo, hi = index.local_rows()
    # 32-row batches: 8 per rank (238 or 234 rows).  Pace: ranks pump (1 + rank % 3) batches per step, so after the 3-step reload interval
    # the fast ranks (2, 5: 3 per step -> 9 >= 8) have finished their pass and the others have not: the boundary the review asks about
    indexer = AsyncIndexBuilder(model, arena, index, S_RET, CLS, SEP, PAD, batch_size=32, index_reload_interval=3, batches_per_pump=1 + rank % 3)
    g = torch.Generator(device="cuda").manual_seed(11)
    q = torch.randn((16, 128), generator=g, device="cuda").half()
    log = []
    for it in range(1, 12):
        if indexer.pump():
            indexer.stream.synchronize()                # deterministic: "enqueued" means "finished" in this test
        d, i = index.search_mips_index(q, 10)           # a training step's search: a collective of its own, between the swap handshakes
        ready_here = bool(indexer.ready())
        swapped = indexer.maybe_swap(it)
        log.append((it, ready_here, bool(swapped)))
        if swapped:
            break
    d, i = index.search_mips_index(q, 10)
    # what a synchronous rebuild from the same weights gives (eval mode: embeddings do not depend on the batch composition)
    sync = IndexBuilder(model, arena, S_RET, CLS, SEP, PAD, batch_size=32)
    ref = DistributedBruteForceIndex(128, None)
    ref.add_arrays(ids, np.zeros((n_docs, 128), dtype=np.float16))
    sync.build_into_index(ref)
    d2, i2 = ref.search_mips_index(q, 10)
    same = bool(torch.equal(d.view(torch.int16), d2.view(torch.int16)) and torch.equal(i, i2))
    torch.save({"log": log, "same": same, "rows": (lo, hi), "refreshes": indexer.refreshes, "ids": i.cpu()}, os.path.join(out_dir, "s%d.pt" % rank))
    torch.distributed.destroy_process_group()


def test_refresher_pass_ready_on_some_ranks_only_at_the_swap_boundary(tmp_path):
    """tasks/openqa/e2eqa/async_indexer.py:116-144 (the NEW_INDEX_READY handshake) at world 8: when the reload interval has gone by and the
    pass over the shard is complete on ranks 2 and 5 only, NO rank swaps (one MIN all-reduce of the ready flags; nobody hangs, the training
    step's own collectives keep running between the handshakes), the ready ranks keep serving the old image, and all eight swap at the SAME
    later step boundary -- the first one at which the slowest rank is done -- to an image equal to a synchronous rebuild."""
    import torch
    torch.cuda.empty_cache()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_swap_worker, args=(8, port, str(tmp_path)), nprocs=8, join=True)
    res = [torch.load(os.path.join(str(tmp_path), "s%d.pt" % r)) for r in range(8)]
    assert [r["rows"] for r in res] == [(238 * k, min(238 * (k + 1), 1900)) for k in range(8)]
    swap_steps = [[it for it, _, sw in r["log"] if sw] for r in res]
    assert all(s_ == swap_steps[0] and len(s_) == 1 for s_ in swap_steps), swap_steps      # the same step on every rank, once
    step = swap_steps[0][0]
    assert step == 9                                                     # slowest ranks: 1 batch per step, 8 batches, the pass ends with the 9th pump
    at_boundary = [dict((it, rd) for it, rd, _ in r["log"])[3] for r in res]              # the first step at which the interval has passed
    assert at_boundary == [False, False, True, False, False, True, False, False]         # ready on SOME ranks only ...
    assert all(not sw for r in res for it, _, sw in r["log"] if it < step)               # ... and nobody swapped before everybody was
    assert all(r["same"] and r["refreshes"] == 1 for r in res)
    assert all(torch.equal(r["ids"], res[0]["ids"]) for r in res)


def test_task_entry_point_on_eight_ranks(tmp_path):
    """The training task as the shipped scripts launch it (`python -m torch.distributed.run --nproc-per-node 8 tasks/run.py --task OPENQA ...`,
    examples/openqa/emdr2_nq.sh:35,106) on EIGHT ranks: the embedding pickle unpickled by the node-first rank only and mapped by all through its
    flat twin, 8-way row shards (300 rows: 38 per rank, the last one 34), all-gather(queries) + record all-gather + merge per search, bucketed bf16
    gradient exchange, the refreshers' swap handshake, checkpoint written by rank 0, EM evaluation de-duplicated over the ranks."""
    from emdr2_amd import checkpointing
    tmp = str(tmp_path)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tools", "dryrun_task.py"), tmp, "--batch-size", "1", "--question-micro-batches", "2"]
    import torch
    torch.cuda.empty_cache()
    out = subprocess.run(cmd, env=_env(), cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    text = out.stdout.decode()
    assert out.returncode == 0, text[-3000:]
    assert all(("rank %d done" % r) in text for r in range(8))
    assert "MIPS Index Updated" in text and "lm_loss" in text and "Exact Match Score" in text
    assert os.path.exists(os.path.join(tmp, "emb.flat"))
    it, release = checkpointing.read_tracker(os.path.join(tmp, "ckpt"))
    assert it == 3 and not release                                        # 24 questions / (batch 1 x 8 ranks), one epoch
