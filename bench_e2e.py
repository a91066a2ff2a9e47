#!/usr/bin/env python
"""bench_e2e.py -- one EMDR2 training step on MI355X (BASELINE.json configs[2]: NQ end-to-end step, retriever + MIPS over the
21M-row index + FiD reader forward/backward, B = 64 questions/GPU, top-k 50, S_ret 256, S 512, L 32, bf16, synthetic data).

One step = query tower -> MIPS search over the resident index -> device-side evidence fetch + token assembly -> context tower ->
reader encoder/decoder + the no-grad one-context reader pass -> EMDR2 loss -> backward (per-layer recompute) -> DP all-reduce ->
Adam (train loop of the reference: tasks/openqa/e2eqa/train_e2eqa.py:468-513).  Hidden / attention dropout 0.1 as in the reference scripts.

`bench.py` (the driver's benchmark) imports `setup` / `run` from here and reports this step as its `e2e` object next to the MIPS
numbers; run directly, this script prints the step as its own ONE JSON line:

    python bench_e2e.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--topk K] [--rows R] [--reindex-rows-per-step n]
"""
import argparse
import ctypes
import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
MFMA_PEAK_TFLOPS = 2500.0
H, L, V_T5, V_BERT = 768, 32, 30720, 30592


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


def add_args(ap):
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--seq", type=int, default=512)
    ap.add_argument("--seq-ret", type=int, default=256)
    ap.add_argument("--dropout", type=float, default=0.1, help="hidden and attention dropout (the reference's default, arguments.py)")
    ap.add_argument("--keep-last-layers", default="auto",
                    help="reader-encoder layers whose activations are kept instead of re-run in the backward: a number, or 'auto' = as many as "
                         "fit the HBM left over after a first full-recompute step with 25 GB to spare (falls back to 0 on an allocation failure)")
    ap.add_argument("--selective-layers", default="auto",
                    help="encoder layers run with selective activation retention (6 of ~16 [tokens, h] tensors kept, FFN intermediates rebuilt in "
                         "the backward): 'auto' = reader encoder first, then the context tower, as HBM allows; or 'R,C' (reader, context tower)")
    ap.add_argument("--micro-batches", type=int, default=-1,
                    help="question micro-batches per step (EMDR2Model.forward_backward): the batch's questions run their post-search forward + "
                         "backward in this many groups and NO layer is re-run in the backward (no --checkpoint-activations).  -1 = by top-k "
                         "(4 up to 50, 8 above); 1 = the undivided step with per-layer recompute / selective retention (rounds 1-4)")
    ap.add_argument("--no-packing", action="store_true",
                    help="run the encoder stacks over the reference's padded [batch, S] grids instead of the packed real tokens (A/B)")
    ap.add_argument("--reindex-rows-per-step", type=int, default=0,
                    help="BASELINE configs[5]: re-embed this many evidence rows per training step on a side stream into the spare index image "
                         "(N / (8 ranks * 0.9 * 500-step reload interval) = 5838 is the 8-GPU pace with 10 % slack)")


def build_index(rows, rank, world):
    """This rank's row shard of the synthetic evidence index (same generator as bench.py), inside a DistributedBruteForceIndex."""
    from emdr2_amd.data.emdr2_index import DistributedBruteForceIndex, shard_bounds
    import bench as mips_bench
    index = DistributedBruteForceIndex(embed_size=H, embed_data=None, use_gpu=True)
    lo, hi = shard_bounds(rows, world)[rank]
    index.num_rows = rows
    index.shard = index._make_shard(H, hi - lo, lo)
    for block in mips_bench.synth_rows(lo, hi):
        index.shard.append_rows(block)
    index.shard.set_ids(torch.arange(lo + 1, hi + 1, dtype=torch.int32, device="cuda"))
    return index


def setup(args, rank, world, index=None, topk=50):
    """Model, optimizer, corpus, synthetic batch generator; returns a context with a `step()` callable."""
    from emdr2_amd.data.evidence_arena import EvidenceArena
    from emdr2_amd.model.emdr2_model import EMDR2Model, PreComputedEvidenceDocsRetriever, emdr2_loss
    from emdr2_amd.model.transformer import Config
    from emdr2_amd.training import FlatAdam, AnnealingLR
    from emdr2_amd.model import kernels as Kmod

    B, K, S, S_ret = args.batch, topk, args.seq, args.seq_ret
    micro = int(getattr(args, "micro_batches", -1))
    if micro < 1:
        micro = 4 if K <= 50 else 8
        while B % micro:
            micro //= 2
    Kmod.PACKING.enabled = not getattr(args, "no_packing", False)
    Kmod.PACKING.sticky, Kmod.PACKING.capacity = True, {}    # constant activation sizes from step to step (allocator reuse)
    Kmod.PREMASK.enabled = os.environ.get("EMDR2_PREMASK", "1") != "0"          # A/B switch: dropout masks of the backward from the LayerNorm backward (default) or their own launches
    if index is None:
        index = build_index(args.rows, rank, world)
    arena = EvidenceArena.synthetic(args.rows)
    retr = PreComputedEvidenceDocsRetriever.__new__(PreComputedEvidenceDocsRetriever)
    retr.args = types.SimpleNamespace(topk_retrievals=K, seq_length=S, seq_length_ret=S_ret)
    retr.topk, retr.mips_index, retr.arena, retr.process_group, retr.searches = K, index, arena, None, 0

    torch.manual_seed(1234)
    cfg = Config(num_layers=args.layers, hidden_size=H, num_attention_heads=12, ffn_hidden_size=3072, max_position_embeddings=512, init_method_std=0.02,
                 hidden_dropout=args.dropout, attention_dropout=args.dropout)
    model = EMDR2Model(retr, cfg, V_T5, V_BERT, K, S, S_ret, cls_id=101, sep_id=102, checkpoint_activations=(micro == 1))
    model.train()
    # flat buckets: masters / gradients / moments / bf16 working copies back to back; ~20 optimizer launches per step; with world > 1 the
    # buckets are all-reduced in bf16 (0.88 GB on the wire) as their last gradient arrives from the backward
    opt = Kmod.GRAD_SINK = FlatAdam(model, lr=2e-5, weight_decay=0.1, clip_grad=1.0)
    sched = AnnealingLR(2e-5, 10, 1000)
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

    from emdr2_amd.training import RetentionGuard
    # the retention plan in force (run() fills it in) and the all-ranks-together recovery from a step that runs out of HBM
    guard = RetentionGuard(model, opt, forward_progress=lambda: retr.searches, micro=micro, batch=B)
    plan = guard.plan

    def step():
        return guard.run(step_once)

    inject = tuple(int(v) for v in os.environ.get("EMDR2_BENCH_INJECT_OOM", "-1,-1").split(","))    # "rank,call": dry-run hook of the tests
    calls = [0]

    def step_once():
        calls[0] += 1
        if indexer is not None:
            indexer.pump()                                                # side stream: overlaps with the training kernels below
        bt = make_batch()
        opt.zero_grad()

        def injected(group=0):
            if inject == (rank, calls[0]) and group == 0:
                raise torch.cuda.OutOfMemoryError("injected by EMDR2_BENCH_INJECT_OOM (tests/test_dist_gpu.py)")
        if guard.micro > 1:
            # the batch's questions in groups: one search, then forward + backward per group, every activation kept (zero recompute)
            loss, stats = model.forward_backward(bt["uid"], bt["q"], bt["types"], None, bt["q"], bt["qlen"], bt["dec"], bt["labels"], bt["mask"],
                                                 30523, micro_batches=guard.micro, on_group=injected)
        else:
            lm, tlp, one = model(bt["uid"], bt["q"], bt["types"], None, bt["q"], bt["qlen"], bt["dec"])
            loss, stats = emdr2_loss(lm, tlp, one, bt["labels"], bt["mask"], eos_id=30523)
            injected()
            loss.backward()
        opt.finish()                                                      # waits for the bucket all-reduces launched from inside the backward
        opt.step(lr=sched.get_lr())
        sched.step()
        return loss

    return types.SimpleNamespace(micro=micro, indexer=indexer, make_batch=make_batch, retriever=retr, sched=sched, plan=plan, guard=guard, keep_last_arg=getattr(args, "keep_last_layers", "auto"), selective_arg=getattr(args, "selective_layers", "auto"), keep_last=0, step=step, model=model, opt=opt, n_params=sum(p.numel() for p in model.parameters()), B=B, K=K, S=S, S_ret=S_ret,
                                 layers=args.layers, rows=args.rows, dropout=args.dropout, reindex=args.reindex_rows_per_step)


def choose_retention(ctx, world):
    """How the HBM left over after a full-recompute step is spent on NOT recomputing (all ranks take the minimum):
      selective retention (transformer.ParallelTransformerLayer.forward_selective): +4 [tokens, h] tensors per layer over a checkpointed
        layer, saves 2/3 of that layer's re-run -- reader encoder first (the largest stack), then the context tower;
      keep_last: what is still free upgrades the LAST reader-encoder layers to keeping everything (+10 more tensors, saves the last third).
    Returns (keep_last, selective_reader, selective_context)."""
    from emdr2_amd.model import kernels as Kmod
    L = ctx.layers
    want_keep, want_sel = getattr(ctx, "keep_last_arg", "auto"), getattr(ctx, "selective_arg", "auto")
    free, total = torch.cuda.mem_get_info()
    capacity = free + torch.cuda.memory_reserved()            # what this process can have: its own pool + what is still free
    # what the full-recompute step needed, plus the caching allocator's overhead (measured: reserved = 1.08 x allocated at the peak), plus a
    # margin: packed token counts -- and with them every activation size -- move by a fraction of a percent from step to step
    peak = int(torch.cuda.max_memory_allocated() * 1.08)
    spare = (25 << 30) if not Kmod.PACKING.enabled else (12 << 30)
    budget = max(0, int((capacity - spare - peak) / 1.08))
    hist = [(S, rows) for n, S, rows in Kmod.PACKING.history if n == ctx.B * ctx.K] if Kmod.PACKING.enabled else []
    rows_reader = max([r for S, r in hist if S == ctx.S] + [0]) or ctx.B * ctx.K * ctx.S                 # (the larger of the two S-long stacks: qext)
    rows_ctx = max([r for S, r in hist if S == ctx.S_ret] + [0]) or ctx.B * ctx.K * ctx.S_ret
    unit_r, unit_c = rows_reader * H * 2, rows_ctx * H * 2                                                 # one [tokens, h] bf16 tensor
    if want_sel != "auto":
        sel_r, sel_c = (int(v) for v in str(want_sel).split(","))
    else:
        # measured at B = 64, K = 50: 6.3 GB per selective reader-encoder layer (3.2 tensors of [1.29M, 768] bf16 over a checkpointed one),
        # 3.0 GB per context-tower layer (4.4 tensors of [0.44M, 768]); a fully kept layer 14.5 GB (7.3 tensors): budgeted at 3.3 / 4.6 / 8.0
        sel_r = int(min(L, budget // (3.3 * unit_r)))
        budget -= sel_r * 3.3 * unit_r
        sel_c = int(min(L, budget // (4.6 * unit_c))) if sel_r == L else 0
        budget -= sel_c * 4.6 * unit_c
    if want_keep != "auto":
        keep = max(0, min(int(want_keep), L))
    else:
        keep = int(min(sel_r, budget // (8.0 * unit_r)))      # a kept layer replaces a selective one
    sel_r = min(sel_r, L - keep)
    if world > 1:
        t = torch.tensor([keep, sel_r, sel_c], device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MIN)
        keep, sel_r, sel_c = (int(v) for v in t.tolist())
    return keep, sel_r, sel_c


def run(ctx, steps, warmup, world):
    """W untimed + K timed steps between fences (barrier + synchronize), MAX over ranks; GEMM / attention time from the library's
    per-launch hipEvents (recorded on the launch stream inside the timed region)."""
    from emdr2_amd import _native
    lib = _native.lib()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    loss = None
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()                     # (the corpus / index builders of setup() peak higher than a training step)
    for _ in range(warmup):
        loss = ctx.step()
    fence()
    # question micro-batches: every activation of a group is kept, nothing is re-run, there is no retention plan to choose
    keep, sel_r, sel_c = choose_retention(ctx, world) if ctx.guard.micro == 1 else (0, 0, 0)
    full_ms = None
    if keep + sel_r + sel_c > 0:
        t0 = time.perf_counter()                             # for the record: one step with the reference's full per-layer recompute
        loss = ctx.step()
        fence()
        full_ms = (time.perf_counter() - t0) * 1e3
        warmup += 1
        # then one more untimed step so the allocator has grown before the timed region; if the estimate was too optimistic on this box the
        # plan is thinned out (context tower first, then half of the reader layers, then the reference's full recompute)
        # (a plan also counts as too tight when the caching allocator had to give blocks back to the driver and ask again during the trial
        # step -- `num_alloc_retries` -- : such a step runs, but at 1.2-1.3 x the time)
        retries = lambda: int(torch.cuda.memory_stats().get("num_alloc_retries", 0)) + ctx.guard.reruns

        def any_rank(flag):                                  # every rank takes the same branch below
            if world == 1:
                return bool(flag)
            t = torch.tensor([int(bool(flag))], device="cuda")
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            return bool(int(t.item()))
        for plan in ((keep, sel_r, sel_c), (0, sel_r, max(sel_c - 3, 0)), (0, sel_r, 0), (0, sel_r // 2, 0), (0, 0, 0)):
            keep, sel_r, sel_c = plan
            ctx.guard.set(keep, sel_r, sel_c, ctx.layers if sel_c else 0)
            torch.cuda.empty_cache()                         # blocks cached for the previous retention pattern do not fit the new one
            before = retries()
            loss = ctx.step()                                # (an allocation failure inside is handled there, by all ranks together, and counted)
            warmup += 1
            if not any_rank(retries() != before) or plan == (0, 0, 0):
                break
            keep, sel_r, sel_c = ctx.plan["keep"], ctx.plan["reader"], ctx.plan["context"]     # (the step may have thinned the plan itself)
        keep, sel_r, sel_c = ctx.plan["keep"], ctx.plan["reader"], ctx.plan["context"]
        fence()
    ctx.keep_last, ctx.selective, ctx.full_recompute_ms = keep, (sel_r, sel_c), (full_ms if keep + sel_r + sel_c > 0 else None)
    from emdr2_amd.model import kernels as Kmod
    alloc_retries = lambda: int(torch.cuda.memory_stats().get("num_alloc_retries", 0))
    timed_reruns = 0
    while True:
        lib.emdr2_ops_set_timing(1)
        Kmod.PACKING.real_tokens = Kmod.PACKING.grid_tokens = 0
        Kmod.RECOMPUTE.flops = 0.0
        before = alloc_retries()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]      # per-step boundaries on the stream (no host sync in the region)
        growths_before = Kmod.PACKING.growths
        t0 = time.perf_counter()
        for i in range(steps):
            marks[i].record()
            loss = ctx.step()
        marks[steps].record()
        fence()
        elapsed = time.perf_counter() - t0
        step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
        capacity_growths = Kmod.PACKING.growths - growths_before
        ms = (ctypes.c_double * 4)(); fl = (ctypes.c_double * 4)(); nl = (ctypes.c_int64 * 4)()
        _native.check(lib.emdr2_ops_timing_collect(ms, fl, nl, 4), "ops_timing_collect")
        lib.emdr2_ops_set_timing(0)
        # A packed stack that met a new maximum of real tokens grew its (sticky) row capacity inside these steps and, this close to the HBM
        # limit, the caching allocator gave its blocks back to the driver and asked again: seconds, once per new maximum (they stop coming
        # after the first tens of steps).  That is warm-up, not the step: the K steps are timed again, once, and the line says so.
        # r06: the same holds when the allocator did not have to retry: a capacity that grows changes EVERY activation size of its stack, none
        # of the cached blocks fit, and the step pays ~1.6 s of fresh hipMallocs (measured: one 3,166 ms step among nine of 1,540)
        retried = alloc_retries() > before or capacity_growths > 0
        if world > 1:
            t = torch.tensor([int(retried)], device="cuda")
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            retried = bool(int(t.item()))
        if not retried or timed_reruns >= 1:
            break
        timed_reruns += 1
        warmup += steps
    replicas = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
        # data-parallel replicas must hold bit-identical parameters after the timed steps (same averaged gradients, deterministic norm)
        cs = torch.stack([b["master"].double().sum() for b in ctx.opt.buckets]).sum().reshape(1)
        allcs = [torch.empty_like(cs) for _ in range(world)]
        torch.distributed.all_gather(allcs, cs)
        replicas = [float(c.item()) for c in allcs]
    # HBM traffic of the step's GEMM launches: per-kernel FETCH_SIZE / WRITE_SIZE (separate rocprofv3 --pmc passes over the kernels one by one,
    # tools/r06_evidence.sh), each access shape scaled by the factor measured on a known byte count on the same box
    # (profiles/r06_fetch_calibration.json), weighted by the algorithmic bytes of every GEMM launch of a step -- a committed summary of THIS
    # round's kernels, quoted only if it was taken from the very library that is loaded now
    traffic, traffic_note = None, "profiles/r06_gemm_summary.json not found"
    prof = os.path.join(ROOT, "profiles", "r06_gemm_summary.json")
    if os.path.exists(prof):
        import hashlib
        pj = json.load(open(prof))
        sw = pj.get("step_weighted_traffic")
        with open(_native.LIB_PATH, "rb") as fh:
            loaded = hashlib.sha256(fh.read()).hexdigest()
        if pj.get("library_sha256") != loaded:
            traffic_note = ("profiles/r06_gemm_summary.json was measured on another build of libemdr2_hip.so (sha256 %s..., loaded %s...): not quoted"
                            % (str(pj.get("library_sha256"))[:12], loaded[:12]))
        elif sw:
            cal = pj.get("calibration", {})
            traffic = {"hbm_read_gb_per_step": sw["measured_read_gb"], "algorithmic_read_gb_per_step": sw["algorithmic_read_gb"], "read_ratio": sw["read_ratio"],
                       "hbm_write_gb_per_step": sw["measured_write_gb_uncalibrated"], "algorithmic_write_gb_per_step": sw["algorithmic_write_gb"],
                       "write_ratio": sw["write_ratio_uncalibrated"], "calibration": cal}
            traffic_note = ("step-weighted over the GEMM launches of one step (%.0f of %.0f ms covered) from profiles/r06_gemm_summary.json (same library: sha256 "
                            "%s...): per (kind, N, K, epilogue) FETCH_SIZE / WRITE_SIZE, the LDS-DMA operand stream, the epilogue's residual-row reads and "
                            "its row-segment stores each scaled by the factor measured for THAT access shape on a known byte count "
                            "(profiles/r06_fetch_calibration.json; calibrated_on_this_box = %s)" % (sw["covered_ms"], sw["all_ms"], loaded[:12],
                                                                                               cal.get("calibrated_on_this_box")))
    ctx.keep_last, ctx.selective = ctx.plan["keep"], (ctx.plan["reader"], ctx.plan["context"])       # (thinned if a step ran out of memory)
    fl_step = flops_per_step(ctx.B, ctx.K, ctx.S_ret, ctx.S, L, H, V_T5, ctx.layers)
    sps = steps / elapsed
    gemm_ms, gemm_fl = ms[0] + ms[1], fl[0] + fl[1]
    gemm_tf = gemm_fl / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    kinds = ("gemm_nt", "gemm_tn", "attention_fwd", "attention_bwd")
    return {
        "steps_per_s": sps * 1.0, "ms_per_step": elapsed / steps * 1e3, "steps": steps, "warmup": warmup, "n_gpus": world, "dtype": "bf16", "scaling": "weak",
        # every timed step by itself (stream events at the step boundaries, rank 0): ms_per_step is their mean; a packed stack that meets a new
        # maximum of real tokens grows its row capacity once and that step pays for fresh allocations of every activation size
        "step_ms": [round(x, 1) for x in step_ms], "median_step_ms": sorted(step_ms)[len(step_ms) // 2], "packed_capacity_growths_in_timed_steps": capacity_growths,
        "config": {"workload": "BASELINE configs[2]: EMDR2 end-to-end step, B=%d/GPU, top-k %d, S_ret %d, S %d, L %d, %d-row index, %d layers"
                               % (ctx.B, ctx.K, ctx.S_ret, ctx.S, L, ctx.rows, ctx.layers),
                   "global_batch": ctx.B * world, "params": ctx.n_params, "parallelism": "dp%d (index row-sharded x%d)" % (world, world),
                   "packed_sequences": bool(Kmod.PACKING.enabled),
                   # encoder-stack tokens per step (query tower, context tower, reader encoder, one-context pass): real = what the packed
                   # layout runs, padded = the reference's [batch, S] grids
                   "tokens_real": (Kmod.PACKING.real_tokens // steps) if Kmod.PACKING.enabled else None,
                   "tokens_padded": (Kmod.PACKING.grid_tokens // steps) if Kmod.PACKING.enabled else
                                    ctx.B * ctx.S_ret + ctx.B * ctx.K * ctx.S_ret + 2 * ctx.B * ctx.K * ctx.S,
                   "question_micro_batches": ctx.guard.micro,
                   # north_star quotes two tolerances (1e-3 fp32 / 2e-2 bf16).  What is measured here is the bf16 product; the fp32 half is
                   # exercised by a validation-only fp32 compute mode (Config(compute_dtype="fp32") / --fp32-validation: fp32 MFMA GEMM, composed
                   # attention, dense layouts, no dropout; tests/test_parity_fp32_gpu.py: logits 7e-7, gradients <= 3e-5 of the fp32 oracle)
                   "fp32_mode": "validation only (tests/test_parity_fp32_gpu.py; never timed)",
                   "dropout": ctx.dropout, "activation_recompute": ("none: forward + backward in %d groups of %d questions, every activation of a group kept" % (ctx.guard.micro, ctx.B // ctx.guard.micro)) if ctx.guard.micro > 1 else "per layer" + (", except the last %d reader-encoder layers (all activations kept in HBM)" % ctx.keep_last if ctx.keep_last else "") +
                                           ("; selective retention (6 of ~16 [tokens, h] tensors kept, LayerNorm outputs + FFN intermediates rebuilt in the backward) "
                                            "on %d reader-encoder and %d context-tower layers" % ctx.selective if sum(ctx.selective) else ""),
                   "recompute_tflop_per_step": Kmod.RECOMPUTE.flops / steps / 1e12, "steps_rerun_after_out_of_memory": ctx.guard.reruns, "retention_thinned_after_out_of_memory": ctx.plan["thinned"],
                   "timed_region_reruns_after_allocator_retry": timed_reruns,
                   "ms_per_step_full_recompute": ctx.full_recompute_ms,      # one step timed before the switch (None when nothing is kept)
                   "loss": float(loss.detach()), "replica_parameter_checksums": replicas,
                   "reindex_rows_per_step": ctx.reindex, "peak_hbm_gb": round(torch.cuda.max_memory_allocated() / 1e9, 1),
                   "allocator_retries": int(torch.cuda.memory_stats().get("num_alloc_retries", 0)),       # cached blocks freed and re-requested: > 0 = the plan is too tight
                   "optimizer_launches_per_step": getattr(ctx.opt, "optimizer_launches", None),
                   "gradient_exchange": "bf16 all-reduce of %d flat buckets, %.2f GB per step" % (len(ctx.opt.buckets), sum(b["n"] for b in ctx.opt.buckets) * 2 / 1e9)},
        # dominant kernels of the step: the dense linears (NT GEMM forward / input gradients, TN GEMM weight gradients)
        "roofline": {"bound": "mfma", "achieved": gemm_tf, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": gemm_tf / MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_note": traffic_note,
                     "kernel": "gemm_nt (gemm8_kernel / gemm_nt_kernel) + gemm_tn_kernel: executed flops / summed per-launch hipEvent time (rank 0)",
                     "per_step": {k: {"ms": ms[i] / steps, "tflops": (fl[i] / (ms[i] * 1e-3) / 1e12 if ms[i] > 0 else 0.0), "launches": int(nl[i] // steps)}
                                  for i, k in enumerate(kinds)},
                     "executed_tflop_per_step": {"gemm": gemm_fl / steps / 1e12, "attention": (fl[2] + fl[3]) / steps / 1e12},
                     # EXECUTED dense flops (GEMM + attention launches of the timed steps, recompute included) over the whole step time: the
                     # fraction of the MFMA peak the step actually sustains
                     "executed_mfu": {"tflops": (gemm_fl + fl[2] + fl[3]) / elapsed / 1e12, "frac": (gemm_fl + fl[2] + fl[3]) / elapsed / 1e12 / MFMA_PEAK_TFLOPS},
                     # the rate at which the REFERENCE's work is retired: dense-GEMM flops of its padded [batch, S] grids, no recompute
                     # (SURVEY 8d's 1,906 TFLOP per step), over the step time -- not an achieved-MFMA fraction when sequences are packed
                     "padded_work_rate": {"tflops": fl_step * sps / 1e12, "frac_of_peak": fl_step * sps / 1e12 / MFMA_PEAK_TFLOPS,
                                          "flops_per_step_per_gpu": fl_step}},
    }


def release():
    """Drop what a finished `setup` / `run` left in module state (gradient sink, stashes) and give the cached HBM back."""
    import gc
    from emdr2_amd.model import kernels as Kmod
    Kmod.GRAD_SINK = None
    Kmod.ATTN_STASH.store.clear()
    Kmod.FANIN.clear()
    Kmod.PREMASK.clear()
    gc.collect()
    torch.cuda.empty_cache()


def run_k100(args, rank, world, index=None, steps=2, topk=100, reload_interval=500):
    """BASELINE configs[4] (TriviaQA shape: top-k 100 + continuous evidence re-embedding) at its PER-RANK shape: this rank's N/8-row index
    shard (with `index` None -- the 1-GPU run -- a shard-sized index is built: ceil(rows / 8) rows), B questions per GPU in 8 question
    micro-batches, the step timed twice: WITHOUT the refresher, then WITH `AsyncIndexBuilder` re-embedding N / (8 ranks x reload interval)
    rows per step on its side stream into the spare index image (tasks/openqa/e2eqa/async_indexer.py:84-144, train_e2eqa.py:436-508,
    megatron/indexer_emdr2.py:77-114).  SURVEY 8d: step-time inflation vs no refresh, refresh wall-time for the rank's rows."""
    import copy
    ranks = max(world, 8)
    shard_rows = (args.rows + ranks - 1) // ranks if index is None else None
    a = copy.copy(args)
    a.micro_batches = getattr(args, "micro_batches_k100", 8)
    # rows per step and rank: one pass over the rank's N / ranks rows within 90 % of the reload interval (async_indexer.PACE_MARGIN: the swap
    # waits for the slowest rank's pass, so the pace leaves slack instead of needing 489 of the 500 steps)
    from emdr2_amd.tasks.openqa.e2eqa.async_indexer import PACE_MARGIN
    paced_steps = max(1, int(reload_interval * PACE_MARGIN))
    pace = (args.rows + ranks * paced_steps - 1) // (ranks * paced_steps)
    out = {}
    if index is None:
        a.rows = shard_rows
        index = build_index(shard_rows, rank, world)
    for label, rows_per_step in (("without_refresh", 0), ("with_refresh", pace)):
        a.reindex_rows_per_step = rows_per_step
        ctx = setup(a, rank, world, index=index, topk=topk)
        res = run(ctx, steps, 1, world)
        lo, hi = ctx.retriever.mips_index.local_rows()
        out[label] = {"ms_per_step": res["ms_per_step"], "steps_per_s": res["steps_per_s"], "peak_hbm_gb": res["config"]["peak_hbm_gb"],
                      "recompute_tflop_per_step": res["config"]["recompute_tflop_per_step"], "question_micro_batches": res["config"]["question_micro_batches"],
                      "gemm_tflops": res["roofline"]["achieved"], "per_step": res["roofline"]["per_step"],
                      "steps_rerun_after_out_of_memory": res["config"]["steps_rerun_after_out_of_memory"], "loss": res["config"]["loss"]}
        if rows_per_step:
            batches = (rows_per_step + 127) // 128
            out[label]["rows_reembedded_per_step"] = batches * 128
            out[label]["refresher_batches_done"] = int(ctx.indexer.iteration)
        rank_rows = hi - lo
        del ctx, res
        release()
    w, wo = out["with_refresh"]["ms_per_step"], out["without_refresh"]["ms_per_step"]
    per_step = out["with_refresh"]["rows_reembedded_per_step"]
    steps_per_pass = (rank_rows + per_step - 1) // per_step
    out.update({
        "workload": "BASELINE configs[4]: EMDR2 step, B=%d/GPU, top-k %d, S_ret %d, S %d, %d-row index shard per rank (N/%d of %d), refresher at "
                    "N / (%d ranks x 0.9 x %d-step reload interval) = %d rows per step and rank" % (args.batch, topk, args.seq_ret, args.seq, rank_rows, ranks,
                                                                                             args.rows, ranks, reload_interval, pace),
        "n_gpus": world, "steps": steps, "ms_per_step": w, "steps_per_s": 1e3 / w,
        "inflation_from_refresh": w / wo - 1.0,
        "refresh_pass": {"rows_per_rank": rank_rows, "steps": steps_per_pass, "wall_s_at_this_pace": steps_per_pass * w / 1e3,
                         "gpu_s_spent_on_it_per_rank": steps_per_pass * (w - wo) / 1e3,
                         "note": "one pass re-embeds every row of the rank's shard once, spread over `steps` training steps; the swap is a pointer "
                                 "exchange at a step boundary (the reference: a second set of GPUs + 32 GB of pickles reloaded every 500 steps)"},
    })
    return out


def cpu_baseline_subprocess(seconds=12.0, limit=120.0, threads=0):
    """cpu_baseline_model in a child process with a hard wall-clock limit (a slow host must not cost the benchmark its JSON line).
    `threads`: the count the MIPS leg's sweep found fastest on this host (0 = decide here between 32 and 64)."""
    import subprocess
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--cpu-seconds", str(seconds), "--cpu-threads", str(threads)],
                             env=env, timeout=limit,
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout.decode()
        return json.loads(out.strip().splitlines()[-1])
    except Exception as exc:
        return {"error": "%s: %s" % (type(exc).__name__, exc)}


def cpu_baseline_model(seconds=12.0, layers=12, threads=0):
    """The model path on the host cores: the torch-fp32 oracle restatement of the reference's forward / loss (oracle.transformer_oracle,
    pinned on the reference's modules) + autograd backward at the BASELINE architecture (all 12 layers of every stack, H = 768, 12 heads,
    FFN 3072, S_ret 256, S 512, L 32, full vocabularies) on a BOUNDED sample -- B = 1 question, K = 2 passages -- so the factor to the
    B = 64, K = 50 step is batch only (dense-GEMM flops).  kind "port".  Threads: `threads` (what the MIPS leg's sweep found fastest on
    this host), else 32 -- GEMMs of a 2,000-token batch on hundreds of threads run slower than on a few dozen."""
    from oracle import transformer_oracle as to
    cores = min(os.cpu_count() or 1, threads if threads > 0 else 32)
    torch.set_num_threads(cores)
    cfg = dict(layers=layers, hidden=H, heads=12, ffn=3072)
    P = {k: v.requires_grad_(True) for k, v in to.random_params(cfg, V_BERT, V_T5).items()}
    B, K, S_ret, S = 1, 2, 256, 512
    g = torch.Generator().manual_seed(5)

    def ids(shape, n_real):
        x = torch.randint(5, 30522, shape, generator=g)
        x[..., n_real:] = 0
        return x
    qb, ctx, ext, one, dec = ids((B, S_ret), 20), ids((B, K, S_ret), 180), ids((B * K, S), 400), ids((B * K, S), 220), ids((B, L), 6)
    labels = torch.roll(dec, -1, 1)
    mask = (labels != 0).float()

    def step():
        for p in P.values():
            p.grad = None
        lm, tlp, oc = to.emdr2_forward(P, cfg, qb, torch.zeros_like(qb), ~to.make_attention_mask_3d(qb, qb), ctx, torch.zeros_like(ctx), ext, one, dec)
        loss = to.reader_ce_loss(lm, labels, mask) + to.retriever_loss_and_utility(oc, tlp, labels, mask, 30523)[0]
        loss.backward()
    t0 = time.perf_counter(); step(); t_first = time.perf_counter() - t0          # includes allocator warm-up
    reps, t = 0, 0.0
    while reps < 1 or (t_first + t / reps * (reps + 1) < seconds and reps < 3):
        t0 = time.perf_counter(); step(); t += time.perf_counter() - t0; reps += 1
    per = t / reps
    scale = flops_per_step(64, 50, S_ret, S, L, H, V_T5, 12) / flops_per_step(B, K, S_ret, S, L, H, V_T5, layers)
    return {"value": 1.0 / (per * scale), "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": "%d step(s) of the fp32 oracle (forward + loss + autograd backward, Adam excluded) at B=%d, K=%d, S_ret %d, S %d, L %d, "
                      "%d of 12 layers per stack, torch CPU, %d threads: %.2f s/step (first, untimed: %.2f s); scaled x%.0f by dense-GEMM flops to "
                      "B=64, K=50.  BASELINE configs[0] AS WRITTEN (10,000-row index, B = 8, K = 50, fp32) measured once, not scaled: 477 s per step "
                      "on 64 host threads (profiles/r03_config0.json; its forward alone 376 s on the 8 cores of the build container, "
                      "tests/golden/gen_config0_golden.py)" % (reps, B, K, S_ret, S, L, layers, cores, per, t_first, scale)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--topk", type=int, default=50)
    ap.add_argument("--rows", type=int, default=21_015_324)
    ap.add_argument("--cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="print the host-core baseline of the model path as JSON and exit (no GPU)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--cpu-threads", type=int, default=0)
    add_args(ap)
    args = ap.parse_args()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline_model(args.cpu_seconds, threads=args.cpu_threads)), flush=True)
        return
    from emdr2_amd import dist_util
    dist_util.self_launch(args.gpus)      # plain `python bench_e2e.py --gpus N`: becomes N ranks under torch.distributed.run
    rank, world, _ = dist_util.init_distributed()
    ctx = setup(args, rank, world, topk=args.topk)
    res = run(ctx, args.steps, args.warmup, world)
    if rank == 0:
        out = {"metric": "qa_train_steps_per_sec", "value": res["steps_per_s"], "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": res["warmup"],
               "ms_per_step": res["ms_per_step"], "step_ms": res["step_ms"], "median_step_ms": res["median_step_ms"],
               "packed_capacity_growths_in_timed_steps": res["packed_capacity_growths_in_timed_steps"],
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
               "data": "synthetic", "config": res["config"], "roofline": res["roofline"]}
        if args.cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline_subprocess()
        print(json.dumps(out), flush=True)
    dist_util.shutdown()


if __name__ == "__main__":
    main()
