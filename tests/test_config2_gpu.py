"""BASELINE configs[2] AT ITS BENCHMARK SHAPE under -m gpu: B = 64 questions, top-k 50, S_ret 256, S 512, L 32, all 12
layers of the four stacks (440 M parameters), bf16 -- the step `bench.py` times, built by the same `bench_e2e.setup`.  The oracle cannot
run this size, so the checks are properties: the retrieved ids equal the all-exact integer path, the packed layout equals the dense
[batch, S] layout (the reference's, train_e2eqa.py:126-181) in both losses, every parameter that gets a gradient at the oracle-checked
B = 2 size (tests/test_parity_bf16_gpu.py) gets a finite non-zero one here, and the step is a pure function of its inputs: run twice from
the same state it returns the same losses bit for bit and the same gradients to fp32 round-off.  The evidence index has 2,000,000 rows here (the 21,015,324-row search of the same kernels is
`test_mips_gpu.py::test_full_21m_row_index_the_bench_configuration`); everything else is the benchmark's."""
import os
import sys
import types

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _set_dropout(model, p):
    for m in model.modules():
        for name in ("hidden_dropout", "attention_dropout", "embedding_dropout"):
            if hasattr(m, name):
                setattr(m, name, p)


def test_config2_step_at_benchmark_shape_ids_layouts_gradients_and_bit_reproducibility():
    import bench_e2e
    from emdr2_amd.model import kernels as K
    from emdr2_amd.model.emdr2_model import emdr2_loss
    args = types.SimpleNamespace(batch=64, layers=12, seq=512, seq_ret=256, dropout=0.1, keep_last_layers="0", selective_layers="12,6", no_packing=False,
                                 reindex_rows_per_step=0, rows=2_000_000, micro_batches=1)
    ctx = bench_e2e.setup(args, 0, 1, topk=50)
    model, opt, retr = ctx.model, ctx.opt, ctx.retriever
    bt = ctx.make_batch()

    # (1) retrieved ids of the step's own queries: fast path == all-exact integer path (8 of the 64 queries), nothing left unproven
    with torch.no_grad():
        q = model.retriever_embedder(bt["q"], None, bt["types"], "query").to(torch.float16).contiguous()
    shard = retr.mips_index.shard
    d, i, r, f = shard.search(q, 50, exact_fallback=False)
    assert int(f.abs().sum()) == 0
    sel = torch.tensor([0, 7, 13, 21, 34, 47, 55, 63], dtype=torch.int32, device="cuda")
    d2, i2, r2, f2 = d.clone(), i.clone(), r.clone(), f.clone()
    d2[sel.long()] = 0; i2[sel.long()] = -7; r2[sel.long()] = -7
    shard.search_exact(q, sel, 50, d2, i2, r2, f2)
    assert torch.equal(d.view(torch.int16), d2.view(torch.int16)) and torch.equal(i, i2) and torch.equal(r, r2)
    assert int(i.min()) >= 1 and int(i.max()) <= args.rows                      # 1-based doc ids (emdr2_model.py:464)

    def step_grads(packing, retention):
        """forward + loss + backward of the SAME batch from the SAME parameters; (lm loss, retriever loss, gradient buckets)."""
        K.PACKING.enabled = packing
        model.set_recompute_keep_last(0)
        model.set_selective_retention(*retention)
        opt.zero_grad()
        lm, tlp, one = model(bt["uid"], bt["q"], bt["types"], None, bt["q"], bt["qlen"], bt["dec"])
        loss, stats = emdr2_loss(lm, tlp, one, bt["labels"], bt["mask"], eos_id=30523)
        loss.backward()
        opt.finish()
        torch.cuda.synchronize()
        return float(stats["lm_loss"]), float(stats["retriever_loss"]), [b["grad"].clone() for b in opt.buckets]

    # (2) the step twice, dropout 0.1, selective retention on (the benchmark's plan shape): bit-identical losses and gradients
    lm_a, rl_a, g_a = step_grads(True, (12, 6, 12))
    lm_b, rl_b, g_b = step_grads(True, (12, 6, 12))
    assert lm_a == lm_b and rl_a == rl_b                                         # the forward is bit-reproducible (dropout bits included)
    # ... the gradients to fp32 round-off: the weight-gradient GEMM adds its reduction slices, and the embedding backward its rows, with fp32
    # atomics whose order is not fixed (the reference's embedding / cuBLAS split-K backward is not bit-reproducible either); data-parallel
    # replicas still end bit-identical because they all apply the same all-reduced sums (tests/test_dist_gpu.py)
    num = sum(float((x - y).double().pow(2).sum()) for x, y in zip(g_a, g_b)) ** 0.5
    den = sum(float(y.double().pow(2).sum()) for y in g_b) ** 0.5
    assert num / den < 1e-6, num / den
    # ... and the retention plan does not change them beyond bf16 rounding of the rebuilt activations (full recompute = the reference)
    lm_c, rl_c, g_c = step_grads(True, (0, 0, 0))
    assert abs(lm_c - lm_a) < 1e-5 * abs(lm_a) and abs(rl_c - rl_a) < 1e-5 * abs(rl_a)          # (the forward does not depend on what is kept)
    num = sum(float((x - y).double().pow(2).sum()) for x, y in zip(g_a, g_c)) ** 0.5
    den = sum(float(y.double().pow(2).sum()) for y in g_c) ** 0.5
    assert num / den < 2e-2, num / den

    # (3) every parameter gradient finite; non-zero exactly where the model's structure says so (the reader's token-type table is unused)
    names = {id(p): n for n, p in model.named_parameters()}
    idle = []
    for b_, g in zip(opt.buckets, g_c):
        assert bool(torch.isfinite(g).all())
        for p in b_["params"]:
            _, o, n = opt.slot[p]
            if float(g[o:o + n].abs().max()) == 0.0:
                idle.append(names[id(p)])
    assert idle == ["language_model.language_model.embedding.tokentype_embeddings.weight"], idle
    del g_a, g_b, g_c

    # (4) packed == dense layout at dropout 0 (dropout bits are keyed by the element's row in ITS layout, so masks differ between layouts)
    _set_dropout(model, 0.0)
    lm_p, rl_p, g_p = step_grads(True, (0, 0, 0))
    gp = torch.cat([g.reshape(-1) for g in g_p]); del g_p
    lm_d, rl_d, g_d = step_grads(False, (0, 0, 0))
    gd = torch.cat([g.reshape(-1) for g in g_d]); del g_d
    K.PACKING.enabled = True
    assert abs(lm_p - lm_d) < 1e-3 * max(1.0, abs(lm_d)) and abs(rl_p - rl_d) < 1e-3 * max(1.0, abs(rl_d)), (lm_p, lm_d, rl_p, rl_d)
    rel = float((gp - gd).double().norm() / gd.double().norm())
    assert rel < 2e-2, rel
    del gd

    # (5) r05: the benchmark's default step -- 4 groups of 16 questions, every activation kept, NO layer re-run (EMDR2Model.forward_backward) --
    # against the undivided step with the reference's full per-layer recompute, same batch, same parameters, dropout 0: same losses, same
    # gradients (a re-run layer rebuilds bit-identical activations, so what differs is the order of fp32 additions of the weight gradients)
    for stack in (model.language_model.language_model.encoder, model.language_model.language_model.decoder,
                  model.retriever_model.query_model.language_model.encoder, model.retriever_model.context_model.language_model.encoder):
        stack.checkpoint_activations = False
    torch.cuda.empty_cache()
    opt.zero_grad()
    K.RECOMPUTE.flops = 0.0
    loss4, stats4 = model.forward_backward(bt["uid"], bt["q"], bt["types"], None, bt["q"], bt["qlen"], bt["dec"], bt["labels"], bt["mask"], 30523,
                                           micro_batches=4)
    opt.finish()
    torch.cuda.synchronize()
    assert K.RECOMPUTE.flops == 0.0
    g4 = torch.cat([b["grad"].reshape(-1) for b in opt.buckets])
    assert abs(float(stats4["lm_loss"]) - lm_p) < 1e-5 * abs(lm_p) and abs(float(stats4["retriever_loss"]) - rl_p) < 1e-5 * abs(rl_p), (float(stats4["lm_loss"]), lm_p)
    rel = float((g4 - gp).double().norm() / gp.double().norm())
    assert rel < 1e-4, rel
    K.GRAD_SINK = None


def test_config2_benchmark_step_over_the_full_21m_row_index():
    """the step `bench.py` times AS IT RUNS THERE -- B = 64, top-k 50, 12 layers, 4 question groups with every activation kept,
    the FULL 21,015,324-row index and the 21M-passage corpus resident next to it (the test above compares the step's variants and keeps a
    2,000,000-row index: its undivided selective-retention steps need the HBM).  Properties: the step's own queries retrieve what the
    all-exact integer path retrieves, nothing unproven; nothing is re-run in the backward; the step fits in 250 GB; run twice from the same
    parameters and batch it returns bit-identical losses and gradients equal to fp32 round-off; every parameter gradient is finite."""
    import gc
    import bench_e2e
    from emdr2_amd.model import kernels as K
    K.GRAD_SINK = None
    gc.collect(); torch.cuda.empty_cache()
    args = types.SimpleNamespace(batch=64, layers=12, seq=512, seq_ret=256, dropout=0.1, keep_last_layers="0", selective_layers="0,0", no_packing=False,
                                 reindex_rows_per_step=0, rows=21_015_324, micro_batches=4)
    try:
        ctx = bench_e2e.setup(args, 0, 1, topk=50)
        model, opt, retr = ctx.model, ctx.opt, ctx.retriever
        assert retr.mips_index.shard.n_rows == 21_015_324 and ctx.guard.micro == 4
        bt = ctx.make_batch()
        with torch.no_grad():
            q = model.retriever_embedder(bt["q"], None, bt["types"], "query").to(torch.float16).contiguous()
        shard = retr.mips_index.shard
        d, i, r, f = shard.search(q, 50, exact_fallback=False)
        assert int(f.abs().sum()) == 0
        sel = torch.tensor([1, 9, 17, 25, 33, 41, 49, 57], dtype=torch.int32, device="cuda")
        d2, i2, r2, f2 = d.clone(), i.clone(), r.clone(), f.clone()
        d2[sel.long()] = 0; i2[sel.long()] = -7; r2[sel.long()] = -7
        shard.search_exact(q, sel, 50, d2, i2, r2, f2)
        assert torch.equal(d.view(torch.int16), d2.view(torch.int16)) and torch.equal(i, i2) and torch.equal(r, r2)
        assert int(i.min()) >= 1 and int(i.max()) <= args.rows

        def step():
            opt.zero_grad()
            K.RECOMPUTE.flops = 0.0
            loss, stats = model.forward_backward(bt["uid"], bt["q"], bt["types"], None, bt["q"], bt["qlen"], bt["dec"], bt["labels"], bt["mask"], 30523,
                                                 micro_batches=4)
            opt.finish()
            torch.cuda.synchronize()
            assert K.RECOMPUTE.flops == 0.0                                   # zero recompute
            return float(stats["lm_loss"]), float(stats["retriever_loss"]), torch.cat([b["grad"].reshape(-1) for b in opt.buckets])
        torch.cuda.reset_peak_memory_stats()
        lm_a, rl_a, g_a = step()
        lm_b, rl_b, g_b = step()
        assert lm_a == lm_b and rl_a == rl_b
        assert bool(torch.isfinite(g_a).all()) and float(g_a.abs().max()) > 0
        rel = float((g_a - g_b).double().norm() / g_b.double().norm())
        assert rel < 1e-6, rel
        assert torch.cuda.max_memory_allocated() < 250e9, torch.cuda.max_memory_allocated() / 1e9
    finally:
        K.GRAD_SINK = None
        K.PACKING.sticky, K.PACKING.capacity = False, {}
        ctx = model = opt = retr = shard = None
        gc.collect(); torch.cuda.empty_cache()
