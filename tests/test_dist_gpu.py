"""GPU, world size 2 on ONE device (gloo transport, both ranks on cuda:0): the multi-rank search path with the real HIP kernels --
per-rank shard scan, all-gather of the per-shard top-k, HIP merge kernel -- must equal the single-shard oracle on every rank, for the
canonical fp16 search and for the FaissMIPSIndex fp32-score search.  The same two workers also run over RCCL ("nccl" backend, one GPU
per rank) whenever the box has two GPUs -- the first contact with RCCL is then a test, not the driver's 8-GPU benchmark."""
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _init(rank, world, port, backend):
    """gloo: both ranks share cuda:0 (CPU transport).  nccl (= RCCL): one GPU per rank."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    dev = rank if backend == "nccl" else 0
    torch.cuda.set_device(dev)
    extra = {"device_id": torch.device("cuda", dev)} if backend == "nccl" else {}
    torch.distributed.init_process_group(backend, rank=rank, world_size=world, **extra)


def _worker(rank, world, port, backend="gloo"):
    _init(rank, world, port, backend)
    import numpy as np
    import torch
    from emdr2_amd.data.emdr2_index import DistributedBruteForceIndex, FaissMIPSIndex
    from oracle import mips_oracle as mo
    rng = np.random.default_rng(0)
    n, d, nq, k = 30011, 128, 37, 100
    rows = rng.standard_normal((n, d)).astype(np.float16)
    rows[rng.integers(0, n, size=500)] = rows[rng.integers(0, n, size=500)]        # cross-shard duplicates: ties broken by global row
    q = rng.standard_normal((nq, d)).astype(np.float16)
    ids = (rng.permutation(n) + 1).astype(np.int64)
    f = FaissMIPSIndex(d, None, use_gpu=True)
    f.add_with_ids(rows, ids)
    lo, hi = f.local_rows()
    assert (lo, hi) == ((0, 15006) if rank == 0 else (15006, 30011))
    D, I = f.search_mips_index(torch.from_numpy(q), k, reconstruct=False)
    od, oi = mo.topk_f32(rows, q, k, ids=ids)
    assert np.array_equal(D.view(np.uint32), np.ascontiguousarray(od).view(np.uint32)) and np.array_equal(I, oi)
    b = DistributedBruteForceIndex(d, None, use_gpu=True)
    b.add_arrays(ids.astype(np.int32), rows)
    dist, idx = b.search_mips_index(torch.from_numpy(q).cuda(), 50)
    od2, oi2 = mo.topk(rows, q, 50, ids=ids.astype(np.int32))
    assert np.array_equal(dist.cpu().numpy().view(np.uint16), od2.view(np.uint16)) and np.array_equal(idx.cpu().numpy(), oi2)
    torch.distributed.destroy_process_group()


def test_two_rank_sharded_search_equals_single_shard_oracle():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port), nprocs=2, join=True)


def _train_worker(rank, world, port, out_dir, backend="gloo"):
    _init(rank, world, port, backend)
    import numpy as np
    import torch
    from emdr2_amd.model import kernels as K
    from emdr2_amd.model.transformer import Config, T5Model
    from emdr2_amd.training import FlatAdam
    torch.manual_seed(0)                                                 # same initial weights on both ranks, different data
    cfg = Config(num_layers=2, hidden_size=128, num_attention_heads=2, ffn_hidden_size=256, max_position_embeddings=128, init_method_std=0.05)
    m = T5Model(cfg, 512, checkpoint_activations=True).train()
    opt = sink = K.GRAD_SINK = FlatAdam(m, lr=1e-3, bucket_bytes=1 << 20)   # flat buckets, bf16 exchange (the training path of the task / bench)
    rng = np.random.default_rng(100 + rank)
    local_grads = None
    for step in range(3):
        enc = torch.from_numpy(rng.integers(5, 512, size=(8, 64))).cuda(); dec = torch.from_numpy(rng.integers(5, 512, size=(8, 32))).cuda()
        opt.zero_grad()
        logits, _ = m(enc, dec)
        logits.float().square().mean().backward()
        sink.finish()
        opt.step()
    assert sink.launched_early > 0
    assert all(b["xchg"] is not None and b["xchg"].dtype == torch.bfloat16 for b in sink.buckets)      # 16 bits per gradient on the wire
    assert opt.optimizer_launches <= 30
    torch.save([p.detach().cpu() for p in m.parameters()], os.path.join(out_dir, "p%d.pt" % rank))
    K.GRAD_SINK = None
    torch.distributed.destroy_process_group()


def test_two_rank_data_parallel_training_keeps_replicas_identical(tmp_path):
    """Bucketed, overlapped gradient averaging with the real backward kernels: after three optimizer steps on different data the two
    replicas hold bit-identical parameters (every rank applied the same averaged gradients), and they moved away from the initial ones."""
    import torch
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_train_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    p0, p1 = torch.load(os.path.join(str(tmp_path), "p0.pt")), torch.load(os.path.join(str(tmp_path), "p1.pt"))
    assert len(p0) == len(p1) > 10
    for a, b in zip(p0, p1):
        assert torch.equal(a, b)


def _two_gpus():
    import torch
    return torch.cuda.is_available() and torch.cuda.device_count() >= 2


@pytest.mark.skipif(not _two_gpus(), reason="needs two GPUs (RCCL)")
def test_two_rank_sharded_search_over_rccl():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, "nccl"), nprocs=2, join=True)


@pytest.mark.skipif(not _two_gpus(), reason="needs two GPUs (RCCL)")
def test_two_rank_data_parallel_training_over_rccl(tmp_path):
    """bf16 bucket all-reduce on the communicator's stream overlapped with the backward, on real xGMI links: replicas bit-identical."""
    import torch
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_train_worker, args=(2, port, str(tmp_path), "nccl"), nprocs=2, join=True)
    p0, p1 = torch.load(os.path.join(str(tmp_path), "p0.pt")), torch.load(os.path.join(str(tmp_path), "p1.pt"))
    for a, b in zip(p0, p1):
        assert torch.equal(a, b)


def test_bench_under_the_drivers_launcher_is_not_launched_twice():
    """The other form of the multi-GPU command (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`, the task statement's):
    inside a launcher's rank WORLD_SIZE is set and `dist_util.self_launch` must do nothing."""
    import json
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, EMDR2_SINGLE_DEVICE="1", EMDR2_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--rows", "1000000", "--no-e2e", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["rows_per_rank"] == [500000, 500000] and r["config"]["unproven_queries"] == 0


def _plain_bench(extra_env=None, extra_args=()):
    """`python bench.py --gpus 2 ...` exactly as the driver types it -- NO launcher around it (bench.py starts its own ranks through
    dist_util.self_launch) -- as 2 gloo ranks sharing cuda:0 (EMDR2_SINGLE_DEVICE) at a reduced size."""
    import json
    import subprocess
    env = dict(os.environ, EMDR2_SINGLE_DEVICE="1", EMDR2_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--rows", "2000000", "--e2e-steps", "1", "--e2e-warmup", "1",
           "--batch", "4", "--layers", "2", "--keep-last-layers", "0", "--no-cpu-baseline"] + list(extra_args)
    out = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                                      # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_plain_bench_command_with_two_gpus_launches_its_own_ranks_and_prints_both_halves():
    """The driver's multi-GPU command, dry (the reference's one-line launch, examples/openqa/emdr2_nq.sh:35,106).
    Everything an RCCL run executes runs here except the transport: self-launch, process-group bring-up through dist_util (the backend is
    one string), row-sharded scan + ONE all-gather + merge, data-parallel EMDR2 step with the bf16 bucket exchange, the JSON contract."""
    r = _plain_bench()
    assert r["metric"] == "mips_queries_per_sec" and r["n_gpus"] == 2 and r["value"] > 0 and r["scaling"] == "strong"
    assert r["config"]["rows_per_rank"] == [1000000, 1000000] and r["config"]["unproven_queries"] == 0
    assert r["config"]["allgather_bytes_per_rank"] == 512 * 50 * 16 and r["config"]["allgather_plus_merge_ms"] > 0      # one 16-byte record per (query, slot)
    assert r["roofline"]["frac"] > 0
    e = r["e2e"]
    assert "error" not in e, e
    assert e["n_gpus"] == 2 and e["steps_per_s"] > 0 and e["config"]["global_batch"] == 8 and e["config"]["packed_sequences"]
    cs = e["config"]["replica_parameter_checksums"]
    assert len(cs) == 2 and cs[0] == cs[1], cs                          # the two replicas hold bit-identical parameters after the steps
    assert e["roofline"]["per_step"]["gemm_nt"]["launches"] > 0
    assert 0 < e["roofline"]["executed_mfu"]["frac"] < 1 and e["roofline"]["padded_work_rate"]["tflops"] > 0


def test_a_rank_running_out_of_memory_mid_step_is_recovered_by_all_ranks_together():
    """rank 1's second step raises an allocation failure after its forward while rank 0 carries on into the backward and
    its bucket all-reduces.  Rank 1 completes the step's collectives (FlatAdam.abort_step), both ranks discard the step, run it again, and
    end with bit-identical replicas; nothing hangs, one re-run is reported."""
    r = _plain_bench({"EMDR2_BENCH_INJECT_OOM": "1,2"}, ["--selective-layers", "2,0", "--micro-batches", "1"])
    e = r["e2e"]
    assert "error" not in e, e
    assert e["config"]["steps_rerun_after_out_of_memory"] == 1
    cs = e["config"]["replica_parameter_checksums"]
    assert len(cs) == 2 and cs[0] == cs[1], cs


def test_out_of_memory_in_the_very_first_step_between_two_question_groups():
    """the failure comes in rank 1's FIRST step -- before any bucket has ever been exchanged, so the exchange buffers and the
    flags buffer must already exist (FlatAdam._ensure_exchange_buffers: abort_step allocates nothing) -- and in the micro-batched step
    (EMDR2Model.forward_backward), between group 0's forward and its backward, while rank 0 runs all its groups."""
    r = _plain_bench({"EMDR2_BENCH_INJECT_OOM": "1,1"}, ["--micro-batches", "2"])
    e = r["e2e"]
    assert "error" not in e, e
    assert e["config"]["steps_rerun_after_out_of_memory"] == 1 and e["config"]["question_micro_batches"] == 2
    assert e["config"]["recompute_tflop_per_step"] == 0
    cs = e["config"]["replica_parameter_checksums"]
    assert len(cs) == 2 and cs[0] == cs[1], cs


def test_plain_bench_e2e_command_with_two_gpus():
    import json
    import subprocess
    env = dict(os.environ, EMDR2_SINGLE_DEVICE="1", EMDR2_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "bench_e2e.py", "--gpus", "2", "--steps", "1", "--warmup", "1", "--rows", "500000", "--batch", "4", "--layers", "2",
           "--keep-last-layers", "0"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    r = json.loads(lines[0])
    assert r["metric"] == "qa_train_steps_per_sec" and r["n_gpus"] == 2 and r["value"] > 0 and r["scaling"] == "weak"
    cs = r["config"]["replica_parameter_checksums"]
    assert len(cs) == 2 and cs[0] == cs[1], cs


def test_config3_surrogate_two_ranks_full_depth_full_index():
    """BASELINE configs[3] as far as a 1-GPU box goes ("configs[3] only as world-size-2 surrogates at B = 4 / 2 layers"): the PLAIN
    `bench_e2e.py --gpus 2` command at the benchmark's depth and index -- all 12 layers of the four stacks (440 M parameters), the whole
    21,015,324-row index row-sharded over the two ranks (10.5 M rows each), top-k 50, S_ret 256 / S 512 / L 32, bf16, question groups with
    nothing recomputed -- at B = 16 questions per rank (two processes share the one GPU's 288 GB; 8 ranks x B = 64 need 8 GPUs).  Every
    collective of the step runs: all-gather of the queries, the record all-gather + merge of the sharded search, the bucketed bf16 gradient
    all-reduce overlapped with the last group's backward; the two replicas must end with bit-identical parameters."""
    import json
    import subprocess
    env = dict(os.environ, EMDR2_SINGLE_DEVICE="1", EMDR2_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "bench_e2e.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--rows", "21015324", "--batch", "16", "--layers", "12"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    r = json.loads(lines[0])
    c = r["config"]
    assert r["n_gpus"] == 2 and r["value"] > 0 and c["global_batch"] == 32 and c["params"] == 440388096
    assert "21015324-row index" in c["workload"] and "12 layers" in c["workload"] and c["parallelism"] == "dp2 (index row-sharded x2)"
    assert c["question_micro_batches"] == 4 and c["recompute_tflop_per_step"] == 0 and c["steps_rerun_after_out_of_memory"] == 0
    cs = c["replica_parameter_checksums"]
    assert len(cs) == 2 and cs[0] == cs[1], cs
    assert np_isfinite(c["loss"])


def np_isfinite(x):
    import math
    return math.isfinite(float(x))
