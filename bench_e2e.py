#!/usr/bin/env python
"""bench_e2e.py -- one EMDR2 training step on MI355X (BASELINE.json configs[2]: NQ end-to-end step, retriever + MIPS over the
21M-row index + FiD reader forward/backward, B = 64 questions/GPU, top-k 50, S_ret 256, S 512, L 32, bf16, synthetic data).

One step = query tower -> MIPS search over the resident index -> device-side evidence fetch + token assembly -> context tower ->
reader encoder/decoder + the no-grad one-context reader pass -> EMDR2 loss -> backward (per-layer recompute) -> DP all-reduce ->
Adam.  Prints ONE JSON line.  The driver's default benchmark stays `bench.py` (MIPS, configs[1]); this script reports the second
half of BASELINE.json's metric ("QA train steps/sec").  Dropout is 0 (not built yet) - stated in `config`.
"""
import argparse
import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
MFMA_PEAK_TFLOPS = 2500.0


def flops_per_step(B, K, S_ret, S, L, H, V, layers):
    """Dense-GEMM flops, MFU convention: 3x for every grad-enabled pass, 1x for the no-grad one-context pass."""
    lin = 24 * H * H

    def enc(tokens, s):
        return tokens * layers * (lin + 4 * s * H)
    A = enc(B * S_ret, S_ret)
    Bc = enc(B * K * S_ret, S_ret)
    C = enc(B * K * S, S)
    dec_tok = B * L
    D = dec_tok * layers * (lin + 4 * L * H + 4 * H * H + 4 * K * S * H) + (B * K * S) * layers * 4 * H * H + dec_tok * 2 * H * V
    dec1 = B * K * L
    E = C + dec1 * layers * (lin + 4 * L * H + 4 * H * H + 4 * S * H) + (B * K * S) * layers * 4 * H * H + dec1 * 2 * H * V
    return 3 * (A + Bc + C + D) + E


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--topk", type=int, default=50)
    ap.add_argument("--rows", type=int, default=21_015_324)
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--seq", type=int, default=512)
    ap.add_argument("--seq-ret", type=int, default=256)
    ap.add_argument("--dropout", type=float, default=0.1, help="hidden and attention dropout (the reference's default, arguments.py)")
    ap.add_argument("--reindex-rows-per-step", type=int, default=0,
                    help="BASELINE configs[5]: re-embed this many evidence rows per training step on a side stream into the spare index image "
                         "(N / (8 ranks * 500-step reload interval) = 5254 is the 8-GPU pace)")
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("EMDR2_SINGLE_DEVICE"):          # dry run of the N-rank code path on a 1-GPU box: all ranks share cuda:0 (use with gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("EMDR2_DIST_BACKEND", "nccl")
        torch.distributed.init_process_group(backend=backend, **({"device_id": torch.device("cuda", local_rank)} if backend == "nccl" else {}))

    from emdr2_amd.data.emdr2_index import DistributedBruteForceIndex, shard_bounds
    from emdr2_amd.data.evidence_arena import EvidenceArena
    from emdr2_amd.model.emdr2_model import EMDR2Model, PreComputedEvidenceDocsRetriever, emdr2_loss
    from emdr2_amd.model.transformer import Config
    from emdr2_amd.training import FusedAdam, AnnealingLR, allreduce_gradients, get_params_for_weight_decay_optimization
    import bench as mips_bench

    H, L, V_T5, V_BERT = 768, 32, 30720, 30592
    B, K, S, S_ret = args.batch, args.topk, args.seq, args.seq_ret
    # index shard + corpus (setup, untimed)
    index = DistributedBruteForceIndex(embed_size=H, embed_data=None, use_gpu=True)
    lo, hi = shard_bounds(args.rows, world)[rank]
    index.num_rows = args.rows
    index.shard = index._make_shard(H, hi - lo, lo)
    for block in mips_bench.synth_rows(lo, hi):
        index.shard.append_rows(block)
    index.shard.set_ids(torch.arange(lo + 1, hi + 1, dtype=torch.int32, device="cuda"))
    arena = EvidenceArena.synthetic(args.rows)
    retr = PreComputedEvidenceDocsRetriever.__new__(PreComputedEvidenceDocsRetriever)
    retr.args = types.SimpleNamespace(topk_retrievals=K, seq_length=S, seq_length_ret=S_ret)
    retr.topk, retr.mips_index, retr.arena, retr.process_group = K, index, arena, None

    torch.manual_seed(1234)
    cfg = Config(num_layers=args.layers, hidden_size=H, num_attention_heads=12, ffn_hidden_size=3072, max_position_embeddings=512, init_method_std=0.02,
                 hidden_dropout=args.dropout, attention_dropout=args.dropout)
    model = EMDR2Model(retr, cfg, V_T5, V_BERT, K, S, S_ret, cls_id=101, sep_id=102, checkpoint_activations=True)
    model.train()
    opt = FusedAdam(get_params_for_weight_decay_optimization(model), lr=2e-5, weight_decay=0.1, clip_grad=1.0)
    sched = AnnealingLR(2e-5, 10, 1000)
    n_params = sum(p.numel() for p in model.parameters())

    sink = None
    if world > 1:                                                      # bucketed gradient all-reduce overlapped with the backward
        from emdr2_amd.model import kernels as Kmod
        from emdr2_amd.training import GradientBuckets
        sink = Kmod.GRAD_SINK = GradientBuckets(model.parameters())
    indexer = None
    if args.reindex_rows_per_step > 0:
        from emdr2_amd.tasks.openqa.e2eqa.async_indexer import AsyncIndexBuilder
        indexer = AsyncIndexBuilder(model.retriever_model.context_model, arena, index, S_ret, 101, 102, 0, batch_size=128, log_interval=1 << 30,
                                    index_reload_interval=1 << 30, batches_per_pump=(args.reindex_rows_per_step + 127) // 128)

    g = torch.Generator(device="cuda").manual_seed(99 + rank)

    def make_batch():
        qlen = torch.randint(10, 27, (B,), generator=g, device="cuda")
        q = torch.randint(5, 30522, (B, S_ret), generator=g, device="cuda")
        q[:, 0] = 101
        ar = torch.arange(S_ret, device="cuda")[None, :]
        q = torch.where(ar < qlen[:, None], q, torch.zeros_like(q))
        q[torch.arange(B), qlen - 1] = 102
        alen = torch.randint(2, 8, (B,), generator=g, device="cuda")
        ans = torch.randint(5, 30522, (B, L), generator=g, device="cuda")
        arl = torch.arange(L, device="cuda")[None, :]
        dec = torch.where(arl < alen[:, None], ans, torch.zeros_like(ans)); dec[:, 0] = 30522                  # [BOS] a pad
        labels = torch.roll(dec, -1, 1); labels[:, -1] = 0
        labels[torch.arange(B), alen - 1] = 30523                                                               # a [EOS] pad
        return dict(uid=-torch.arange(1, B + 1, device="cuda"), q=q, types=torch.zeros_like(q), qlen=qlen.to(torch.int64), dec=dec,
                    labels=labels, mask=(labels != 0).float())

    def step():
        if indexer is not None:
            indexer.pump()                                                # side stream: overlaps with the training kernels below
        bt = make_batch()
        opt.zero_grad()
        if sink is not None:
            sink.begin_step()
        lm, tlp, one = model(bt["uid"], bt["q"], bt["types"], None, bt["q"], bt["qlen"], bt["dec"])
        loss, stats = emdr2_loss(lm, tlp, one, bt["labels"], bt["mask"], eos_id=30523)
        loss.backward()
        if sink is not None:
            sink.finish()
        else:
            allreduce_gradients(model)
        opt.step(lr=sched.step())
        return loss

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        fl = flops_per_step(B, K, S_ret, S, L, H, V_T5, args.layers)
        sps = args.steps / elapsed
        tf = fl * world * sps / 1e12
        print(json.dumps({
            "metric": "qa_train_steps_per_sec", "value": sps, "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: EMDR2 end-to-end step, B=%d/GPU, top-k %d, S_ret %d, S %d, L %d, %d-row index, %d layers"
                                   % (B, K, S_ret, S, L, args.rows, args.layers),
                       "global_batch": B * world, "params": n_params, "parallelism": "dp%d (index row-sharded x%d)" % (world, world),
                       "dropout": args.dropout, "activation_recompute": "per layer", "loss": float(loss.detach()),
                       "reindex_rows_per_step": args.reindex_rows_per_step,
                       "peak_hbm_gb": round(torch.cuda.max_memory_allocated() / 1e9, 1)},
            "roofline": {"bound": "mfma", "achieved": tf / world, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / world / MFMA_PEAK_TFLOPS,
                         "traffic": None, "flops_per_step_per_gpu": fl, "convention": "dense-GEMM flops, no recompute (SURVEY 8d)"},
        }), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
