"""BASELINE configs[4] AT ITS PER-RANK SHAPE under -m gpu: the TriviaQA-shaped EMDR2 step -- B = 64 questions, top-k 100,
S_ret 256, S 512, L 32, all 12 layers of the four stacks, bf16 -- over the N/8-row index shard one of 8 ranks holds (2,626,916 of the
21,015,324 rows), with the side-stream refresher re-embedding 5,254 evidence rows per step (N / (8 ranks x 500-step reload interval):
tasks/openqa/e2eqa/async_indexer.py:84-144, train_e2eqa.py:436-508, megatron/indexer_emdr2.py:77-114), built by `bench_e2e.setup` like the
benchmark.  The step runs as 8 question micro-batches (EMDR2Model.forward_backward): no layer is re-run in the backward.  The oracle cannot
run this size, so the checks are properties: retrieved ids equal the all-exact integer path; the refresher really advances under the
training steps; nothing is recomputed and the step stays inside 250 GB; the gradients do not depend on how the questions are grouped (8 vs
16 groups at the full batch; 1 vs 4 groups at B = 8, where the undivided step fits); and an image swapped in at a step boundary after
being filled UNDER real training steps equals a synchronous rebuild from the snapshot weights."""
import copy
import os
import sys
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ROWS, K100, PACE = 2_626_916, 100, 5254


def _set_dropout(model, p):
    for m in model.modules():
        for name in ("hidden_dropout", "attention_dropout", "embedding_dropout"):
            if hasattr(m, name):
                setattr(m, name, p)


def _rel(ga, gb):
    num = sum(float((x - y).double().pow(2).sum()) for x, y in zip(ga, gb)) ** 0.5
    den = sum(float(y.double().pow(2).sum()) for y in gb) ** 0.5
    return num / den


def test_config4_k100_step_with_the_refresher_pumping_at_the_per_rank_shape():
    import bench_e2e
    from emdr2_amd.data.emdr2_index import DistributedBruteForceIndex
    from emdr2_amd.indexer_emdr2 import IndexBuilder
    from emdr2_amd.model import kernels as K
    from emdr2_amd.tasks.openqa.e2eqa.async_indexer import AsyncIndexBuilder
    args = types.SimpleNamespace(batch=64, layers=12, seq=512, seq_ret=256, dropout=0.1, keep_last_layers="0", selective_layers="0,0", no_packing=False,
                                 reindex_rows_per_step=PACE, rows=ROWS, micro_batches=8)
    ctx = bench_e2e.setup(args, 0, 1, topk=K100)
    model, opt, retr, indexer = ctx.model, ctx.opt, ctx.retriever, ctx.indexer
    try:
        assert ctx.guard.micro == 8 and not model.language_model.language_model.encoder.checkpoint_activations
        bt = ctx.make_batch()

        # (1) top-100 ids of the step's own queries: fast path == all-exact integer path (8 of the 64 queries), nothing left unproven
        with torch.no_grad():
            q = model.retriever_embedder(bt["q"], None, bt["types"], "query").to(torch.float16).contiguous()
        shard = retr.mips_index.shard
        d, i, r, f = shard.search(q, K100, exact_fallback=False)
        assert int(f.abs().sum()) == 0
        sel = torch.tensor([0, 7, 13, 21, 34, 47, 55, 63], dtype=torch.int32, device="cuda")
        d2, i2, r2, f2 = d.clone(), i.clone(), r.clone(), f.clone()
        d2[sel.long()] = 0; i2[sel.long()] = -7; r2[sel.long()] = -7
        shard.search_exact(q, sel, K100, d2, i2, r2, f2)
        assert torch.equal(d.view(torch.int16), d2.view(torch.int16)) and torch.equal(i, i2) and torch.equal(r, r2)
        assert int(i.min()) >= 1 and int(i.max()) <= ROWS

        # (2) two full training steps (search, assembly, 8 groups of forward + backward, clip + Adam) with the refresher pumping on its stream
        torch.cuda.reset_peak_memory_stats()
        K.RECOMPUTE.flops = 0.0
        w0 = model.retriever_model.context_model.language_model.encoder.layers[0].mlp.dense_h_to_4h.weight.detach().clone()
        losses = [float(ctx.step()) for _ in range(2)]
        torch.cuda.synchronize()
        assert all(np.isfinite(losses)) and opt.step_count == 2
        assert K.RECOMPUTE.flops == 0.0                                      # nothing was re-run in a backward
        assert ctx.guard.reruns == 0 and ctx.plan["thinned"] == 0
        assert torch.cuda.max_memory_allocated() < 250e9, torch.cuda.max_memory_allocated() / 1e9
        batches = (PACE + 127) // 128
        assert indexer.iteration == 2 * batches and shard._refreshed == 2 * batches * 128      # the refresher advanced under the steps ...
        assert not indexer.ready()                                                              # ... and its pass over the shard is far from done
        w1 = model.retriever_model.context_model.language_model.encoder.layers[0].mlp.dense_h_to_4h.weight.detach()
        assert not torch.equal(w0, w1)                                       # the live context encoder moved; the refresher's snapshot did not:
        snap = indexer.model.language_model.encoder.layers[0].mlp.dense_h_to_4h.weight
        assert torch.equal(snap, w0)

        def grads(batch, micro):
            opt.zero_grad()
            loss, stats = model.forward_backward(batch["uid"], batch["q"], batch["types"], None, batch["q"], batch["qlen"], batch["dec"], batch["labels"],
                                                 batch["mask"], 30523, micro_batches=micro)
            opt.finish()
            torch.cuda.synchronize()
            return float(loss), float(stats["lm_loss"]), float(stats["retriever_loss"]), [b["grad"].clone() for b in opt.buckets]

        # (3) the grouping of the questions is immaterial (dropout 0: masks are keyed by the group): 8 vs 16 groups of the full batch ...
        _set_dropout(model, 0.0)
        l8, lm8, rl8, g8 = grads(bt, 8)
        l16, lm16, rl16, g16 = grads(bt, 16)
        assert abs(l8 - l16) < 2e-6 * abs(l8) and abs(lm8 - lm16) < 2e-6 * abs(lm8) and abs(rl8 - rl16) < 2e-6 * abs(rl8), (l8, l16)
        assert _rel(g16, g8) < 2e-5, _rel(g16, g8)
        for g in g8:
            assert bool(torch.isfinite(g).all())
        del g8, g16
        # ... and at B = 8 x top-100 (one group's worth), where the UNDIVIDED step fits without recompute: 1 group (= forward + emdr2_loss +
        # backward, tests/test_microbatch_gpu.py) vs 4 groups of 2 questions
        b8 = {k: v[:8].contiguous() for k, v in bt.items()}
        l1, _, _, g1 = grads(b8, 1)
        l4, _, _, g4 = grads(b8, 4)
        assert abs(l1 - l4) < 2e-6 * abs(l1), (l1, l4)
        assert _rel(g4, g1) < 2e-5, _rel(g4, g1)
        del g1, g4
        _set_dropout(model, 0.1)

        # (4) swap at a step boundary == synchronous rebuild.  A 65,536-row index over the first passages of the same corpus, refreshed by
        # the benchmark-size context encoder (12 layers, S_ret 256, batches of 128) in 128-batch pumps WHILE full K = 100 training steps move
        # the live weights; the image that is swapped in must be the one a synchronous pass from the SNAPSHOT weights builds.
        n_small = 65536
        ids = np.arange(1, n_small + 1, dtype=np.int32)
        small = DistributedBruteForceIndex(768, None)
        small.add_arrays(ids, np.zeros((n_small, 768), dtype=np.float16))
        side = AsyncIndexBuilder(model.retriever_model.context_model, ctx.retriever.arena, small, 256, 101, 102, 0, batch_size=128, log_interval=1 << 30,
                                 index_reload_interval=2, batches_per_pump=128)
        snapshot = copy.deepcopy(side.model)                                  # what the pass embeds with, whatever training does meanwhile
        gq = torch.Generator(device="cuda").manual_seed(11)
        qs = torch.randn((32, 768), generator=gq, device="cuda").half()
        swapped_at = None
        for it in range(1, 9):
            if side.pump():
                side.stream.synchronize()
            ctx.step()
            if side.maybe_swap(it):
                swapped_at = it
                break
        assert swapped_at in (4, 5), swapped_at                               # 512 batches at 128 per step (+ the call that finds the pass exhausted)
        d_side, i_side = small.search_mips_index(qs, 50)
        ref = DistributedBruteForceIndex(768, None)
        ref.add_arrays(ids, np.zeros((n_small, 768), dtype=np.float16))
        IndexBuilder(snapshot, ctx.retriever.arena, 256, 101, 102, 0, batch_size=128, log_interval=1 << 30).build_into_index(ref)
        d_ref, i_ref = ref.search_mips_index(qs, 50)
        assert torch.equal(d_side.view(torch.int16), d_ref.view(torch.int16)) and torch.equal(i_side, i_ref)
        # (the live encoder is elsewhere by now: its embeddings of the same passages differ)
        live = IndexBuilder(model.retriever_model.context_model, ctx.retriever.arena, 256, 101, 102, 0, batch_size=128).embed(torch.arange(1, 129))
        snap_rows = IndexBuilder(snapshot, ctx.retriever.arena, 256, 101, 102, 0, batch_size=128).embed(torch.arange(1, 129))
        assert not torch.equal(live, snap_rows)
    finally:
        K.GRAD_SINK = None
