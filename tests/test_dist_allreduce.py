"""CPU, gloo, world size 2: the data-parallel gradient all-reduce (LocalDDP.allreduce_params, megatron/model/distributed.py:35-62):
pre-divided by the world size, every parameter averaged, ranks end up identical."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from emdr2_amd.training import allreduce_gradients
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.LayerNorm(16), torch.nn.Linear(16, 4))
    g = torch.Generator().manual_seed(100 + rank)
    for p in m.parameters():
        p.grad = torch.randn(p.shape, generator=g)
    m[2].bias.grad = None                                   # a parameter without gradient on this step is skipped consistently
    allreduce_gradients(m)
    torch.save([None if p.grad is None else p.grad.clone() for p in m.parameters()], os.path.join(out_dir, "g%d.pt" % rank))
    dist.destroy_process_group()


def test_gradients_are_averaged_across_ranks(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g0, g1 = torch.load(os.path.join(str(tmp_path), "g0.pt")), torch.load(os.path.join(str(tmp_path), "g1.pt"))
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.LayerNorm(16), torch.nn.Linear(16, 4))
    gens = [torch.Generator().manual_seed(100), torch.Generator().manual_seed(101)]
    for a, b, p in zip(g0, g1, m.parameters()):
        r0, r1 = torch.randn(p.shape, generator=gens[0]), torch.randn(p.shape, generator=gens[1])
        if a is None:
            assert b is None
            continue
        assert torch.allclose(a, b) and torch.allclose(a, (r0 + r1) / 2, atol=1e-6)


def _bucket_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from emdr2_amd.model import kernels as K
    from emdr2_amd.training import FlatAdam
    torch.manual_seed(0)
    names = ["emb", "w1", "b1", "w2", "unused", "w3"]
    shapes = [(50, 16), (64, 16), (64,), (16, 64), (3, 16), (300, 300)]
    mod = torch.nn.Module()
    for n, s_ in zip(names, shapes):
        mod.register_parameter(n, torch.nn.Parameter(torch.zeros(s_)))
    params = [getattr(mod, n) for n in names]
    # the product's optimizer / gradient sink on HOST tensors: only its exchange bookkeeping runs (fp32 buckets over gloo)
    sink = FlatAdam(mod, bucket_bytes=64 * 16 * 4 + 300, exchange_dtype="fp32")    # small buckets: several of them, one holding a single big param
    K.GRAD_SINK = sink
    assert len(sink.buckets) >= 3
    record = []
    for step in range(3):
        g = torch.Generator().manual_seed(1000 * step + rank)
        sink.begin_step()
        assert all(p.grad is None for p in params)
        contrib = {}
        # "backward": w3, w2, b1, w1, then the tied embedding twice (decoder side first, encoder side last); `unused` never gets a gradient
        for i in (5, 3, 2, 1, 0, 0):
            gi = torch.randn(shapes[i], generator=g)
            contrib[i] = contrib.get(i, 0) + gi
            K._accum_grad(params[i], gi)
        early = sink.launched_early
        sink.finish()
        record.append(([None if p.grad is None else p.grad.clone() for p in params], [contrib.get(i) for i in range(len(params))], early))
        assert params[4].grad is None and any(q is params[4] for q in sink.inactive)        # stays out of the optimizer step, like `grad is None`
    # a CHANGED contribution pattern: w3 gets a second, late contribution after its bucket has left, `unused` wakes up, b1 gets
    # nothing -- the step must still deliver the plain average of everything, and the pattern is re-learned
    g = torch.Generator().manual_seed(7000 + rank)
    sink.begin_step()
    contrib = {}
    for i in (5, 3, 1, 0, 0, 5, 4):
        gi = torch.randn(shapes[i], generator=g)
        contrib[i] = contrib.get(i, 0) + gi
        K._accum_grad(params[i], gi)
    sink.finish()
    assert sink.pattern_changes == 1 and sink.expected[params[5]] == 2 and sink.expected[params[2]] == 0
    assert any(q is params[2] for q in sink.inactive) and not any(q is params[4] for q in sink.inactive)
    record.append(([None if p.grad is None else p.grad.clone() for p in params], [contrib.get(i) for i in range(len(params))], sink.launched_early))
    torch.save(record, os.path.join(out_dir, "b%d.pt" % rank))
    try:
        sink.step()
        ok = False
    except Exception as exc:                                                  # no host optimizer: the step is HIP kernels only
        ok = "HIP" in str(exc)
    assert ok
    K.GRAD_SINK = None
    dist.destroy_process_group()


def test_bucketed_overlapped_gradient_averaging_matches_plain_average(tmp_path):
    """training.FlatAdam as the gradient sink of two gloo ranks (host tensors, fp32 exchange: its bookkeeping without its kernels): p.grad
    views into flat buckets, buckets reduced as soon as their last expected contribution arrives (from the second step on), tied parameters
    with two contributions, parameters without gradients, and a step whose contribution pattern differs from the learned one."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_bucket_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(os.path.join(str(tmp_path), "b0.pt")), torch.load(os.path.join(str(tmp_path), "b1.pt"))
    for step in range(4):
        g0, c0, early0 = r0[step]
        g1, c1, early1 = r1[step]
        for a, b, x, y in zip(g0, g1, c0, c1):
            if x is None:
                assert (a is None and b is None) or (float(a.abs().max()) == 0.0 and float(b.abs().max()) == 0.0)
                continue
            assert torch.allclose(a, b) and torch.allclose(a, (x + y) / 2, atol=1e-6)
        if step == 0:
            assert early0 == 0 and early1 == 0                               # the first step learns the contribution counts
    assert r0[2][2] > r0[1][2] > 0                                           # later steps launch buckets from inside the "backward"


def _abort_worker(rank, world, port, out_dir, abort_rank=1, late_rank=1):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from emdr2_amd.model import kernels as K
    from emdr2_amd.training import FlatAdam, StepAborted
    torch.manual_seed(0)
    shapes = [(50, 16), (64, 16), (64,), (16, 64), (300, 300)]
    mod = torch.nn.Module()
    for i, s_ in enumerate(shapes):
        mod.register_parameter("p%d" % i, torch.nn.Parameter(torch.zeros(s_)))
    params = [getattr(mod, "p%d" % i) for i in range(len(shapes))]
    sink = K.GRAD_SINK = FlatAdam(mod, bucket_bytes=64 * 16 * 4 + 300, exchange_dtype="fp32")
    assert len(sink.buckets) >= 3
    order = (4, 3, 2, 1, 0)

    def backward(seed, upto=None, extra=()):
        g = torch.Generator().manual_seed(seed + rank)
        contrib = {}
        for n, i in enumerate(order + tuple(extra)):
            if upto is not None and n >= upto:
                break
            gi = torch.randn(shapes[i], generator=g)
            contrib[i] = contrib.get(i, 0) + gi
            K._accum_grad(params[i], gi)
        return contrib

    record = {}
    for step in range(2):                                                       # learn the pattern, then a step with early launches
        sink.begin_step(); backward(100 * step); sink.finish()
    assert sink.launched_early > 0
    # step 2: rank `abort_rank` "runs out of memory" after two contributions (one bucket has already left); the others complete their
    # backward.  All must come out of the step with StepAborted / a returned abort_step(), none may hang
    sink.begin_step()
    if rank == abort_rank:
        backward(200, upto=2)
        sink.abort_step()
        aborted = True
    else:
        backward(200)
        try:
            sink.finish()
            aborted = False
        except StepAborted:
            aborted = True
    assert aborted
    # step 3: business as usual -- the plain average again
    sink.begin_step(); c = backward(300); sink.finish()
    record["after_abort"] = ([p.grad.clone() for p in params], [c[i] for i in range(len(params))])
    # step 4: only rank `late_rank` sees a late second contribution to the first parameter that leaves (pattern disagreement between
    # ranks): the others join the late-buffer all-reduce with zeros instead of deadlocking it
    sink.begin_step(); c = backward(400, extra=(4,) if rank == late_rank else ()); sink.finish()
    record["late_on_one_rank"] = ([p.grad.clone() for p in params], [c[i] for i in range(len(params))])
    torch.save(record, os.path.join(out_dir, "a%d.pt" % rank))
    K.GRAD_SINK = None
    dist.destroy_process_group()


def test_a_rank_that_gives_a_step_up_takes_its_peers_with_it_and_the_next_step_is_clean(tmp_path):
    """FlatAdam.abort_step (bench_e2e's out-of-memory recovery with world > 1): the rank that cannot finish its backward still
    issues the bucket all-reduces its peers issue, every rank learns of the abort in finish() (StepAborted), the following step averages
    correctly; and a late contribution seen by ONE rank only is exchanged by all."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_abort_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(os.path.join(str(tmp_path), "a0.pt")), torch.load(os.path.join(str(tmp_path), "a1.pt"))
    for key in ("after_abort", "late_on_one_rank"):
        (g0, c0), (g1, c1) = r0[key], r1[key]
        for a, b, x, y in zip(g0, g1, c0, c1):
            assert torch.allclose(a, b) and torch.allclose(a, (x + y) / 2, atol=1e-6), key


def test_abort_and_late_contribution_protocol_at_world_eight(tmp_path):
    """the same protocol with EIGHT ranks (the world size of BASELINE configs[3] / [4]; the largest any test had started
    was 3): rank 5 gives a step up in mid-backward, all eight discard it and the next step is the plain average over eight; a late
    contribution seen by rank 3 only is exchanged by all eight."""
    world = 8
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_abort_worker, args=(world, port, str(tmp_path), 5, 3), nprocs=world, join=True)
    rec = [torch.load(os.path.join(str(tmp_path), "a%d.pt" % r)) for r in range(world)]
    for key in ("after_abort", "late_on_one_rank"):
        grads = [r[key][0] for r in rec]
        contribs = [r[key][1] for r in rec]
        for i in range(len(grads[0])):
            mean = sum(c[i] for c in contribs) / world
            for g in grads:
                assert torch.equal(g[i], grads[0][i])                            # every replica received the same averaged gradient
            assert torch.allclose(grads[0][i], mean, atol=1e-6), key


def test_bucketed_gradient_averaging_at_world_eight(tmp_path):
    """FlatAdam's bucket bookkeeping (learned contribution counts, buckets leaving in index order from inside the backward, a changed
    pattern) with eight gloo ranks: every rank ends every step with the plain average over the eight."""
    world = 8
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_bucket_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    rec = [torch.load(os.path.join(str(tmp_path), "b%d.pt" % r)) for r in range(world)]
    for step in range(4):
        for i in range(6):
            xs = [r[step][1][i] for r in rec]
            gs = [r[step][0][i] for r in rec]
            if xs[0] is None:
                assert all(g is None or float(g.abs().max()) == 0.0 for g in gs)
                continue
            mean = sum(xs) / world
            assert all(torch.equal(g, gs[0]) for g in gs) and torch.allclose(gs[0], mean, atol=1e-6), (step, i)
    assert rec[0][2][2] > rec[0][1][2] > 0
