"""GPU: question micro-batching of the training step (EMDR2Model.forward_backward; the reference runs the undivided batch:
emdr2_model.py:87-214 + train_e2eqa.py:72-181 + megatron/training.py:165-200).  Everything after the MIPS search is per question, both
losses are sums over questions with batch-wide denominators -- so m groups of B / m questions must reproduce the undivided step: same
losses and statistics, the same parameter gradients up to the order of fp32 additions, with or without the flat-bucket gradient sink."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
CFG = dict(layers=2, hidden=128, heads=2, ffn=256, max_pos=128)


def _cfg(p=0.0):
    from emdr2_amd.model.transformer import Config
    return Config(num_layers=CFG["layers"], hidden_size=CFG["hidden"], num_attention_heads=CFG["heads"], ffn_hidden_size=CFG["ffn"],
                  max_position_embeddings=CFG["max_pos"], init_method_std=0.05, hidden_dropout=p, attention_dropout=p)


def _ids(rng, shape, vocab):
    x = rng.integers(5, vocab, size=shape)
    for r in x.reshape(-1, shape[-1]):
        r[int(rng.integers(shape[-1] // 2, shape[-1] + 1)):] = 0
    return torch.from_numpy(x.astype(np.int64)).cuda()


class _FixedRetriever(object):
    """Stands where PreComputedEvidenceDocsRetriever stands: hands back prepared `postprocess` tensors for the whole batch and counts the
    searches (ONE per step, however many groups)."""

    def __init__(self, ctx, typ, qext, qone):
        self.out, self.searches = (ctx, typ, qext, qone, None, None), 0

    def get_topk_assembled(self, query_tensor, *a):
        assert not query_tensor.requires_grad and query_tensor.shape[0] == self.out[0].shape[0]
        self.searches += 1
        return self.out


def _case(p=0.0, seed=7, B=8, Kk=4, S_ret=32, S=64, L=32, V=640):
    from emdr2_amd.model import kernels as K
    from emdr2_amd.model.emdr2_model import EMDR2Model
    rng = np.random.default_rng(seed)
    qb, ctx = _ids(rng, (B, S_ret), 512), _ids(rng, (B, Kk, S_ret), 512)
    qext, qone = _ids(rng, (B * Kk, S), 600), _ids(rng, (B * Kk, S), 600)
    dec = _ids(rng, (B, L), 600)
    labels = torch.roll(dec, -1, 1); labels[:, -1] = 0
    retr = _FixedRetriever(ctx, torch.zeros_like(ctx), qext, qone)
    torch.manual_seed(0)
    K.DROPOUT._sites = 0
    m = EMDR2Model(retr, _cfg(p), V, 512, Kk, S, S_ret, cls_id=2, sep_id=3)
    g = torch.Generator(device="cuda").manual_seed(5)
    with torch.no_grad():
        for q in m.parameters():
            q.add_(0.05 * torch.randn(q.shape, generator=g, device="cuda"))
    m.train()
    batch = dict(uid=-torch.arange(1, B + 1, device="cuda"), q=qb, types=torch.zeros_like(qb), qlen=(qb != 0).sum(1), dec=dec, labels=labels,
                 mask=(labels != 0).float())
    return m, retr, batch


def _grads(m):
    return {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}


def _clear(m):
    for p in m.parameters():
        p.grad = None


def _fb(m, bt, micro, **kw):
    return m.forward_backward(bt["uid"], bt["q"], bt["types"], None, bt["q"], bt["qlen"], bt["dec"], bt["labels"], bt["mask"], 601,
                              micro_batches=micro, **kw)


def _rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


@pytest.mark.parametrize("packing", [True, False])
def test_groups_of_questions_reproduce_the_undivided_step(packing):
    from emdr2_amd.model import kernels as K
    from emdr2_amd.model.emdr2_model import emdr2_loss
    K.PACKING.enabled = packing
    try:
        m, retr, bt = _case()
        # the undivided step as the task driver runs it: forward, loss, backward
        lm, tlp, one = m(bt["uid"], bt["q"], bt["types"], None, bt["q"], bt["qlen"], bt["dec"])
        loss0, stats0 = emdr2_loss(lm, tlp, one, bt["labels"], bt["mask"], 601)
        loss0.backward()
        g0 = _grads(m)
        assert retr.searches == 1
        for micro in (1, 2, 4, 8):
            _clear(m)
            before = retr.searches
            loss, stats = _fb(m, bt, micro)
            assert retr.searches == before + 1                     # one search per step, however many groups
            assert not loss.requires_grad
            g = _grads(m)
            assert set(g) == set(g0)
            tol = 0.0 if micro == 1 else 2e-6
            assert abs(float(loss) - float(loss0)) <= tol * abs(float(loss0)), (micro, float(loss), float(loss0))
            for k in ("lm_loss", "retriever_loss", "retriever_utility", "null_block_lm_loss"):
                assert abs(float(stats[k]) - float(stats0[k])) <= max(tol, 1e-12) * max(1.0, abs(float(stats0[k]))) + (0 if micro == 1 else 2e-6), (micro, k)
            # rows of a GEMM do not depend on how many other rows the launch has, so every activation is bit-identical; parameter
            # gradients differ by the order in which fp32 partial sums meet (atomics within a launch, groups across launches)
            worst = max(_rel(g[k], g0[k]) for k in g0)
            assert worst < 2e-5, (micro, worst)
    finally:
        K.PACKING.enabled = True


def test_a_batch_that_the_group_count_does_not_divide():
    """under --keep-last the short final batch of an epoch is normally not a multiple of --question-micro-batches
    (train_e2eqa.py:272: drop_last = not keep_last); the reference takes any batch size.  8 questions in 3 groups (3 + 3 + 2), 5 groups
    (2 + 2 + 2 + 1 + 1) and in more groups than questions (one each) reproduce the undivided step like the even splits do."""
    from emdr2_amd.model.emdr2_model import emdr2_loss, question_group_bounds
    assert question_group_bounds(8, 3) == [(0, 3), (3, 6), (6, 8)] and question_group_bounds(3, 8) == [(0, 1), (1, 2), (2, 3)]
    m, retr, bt = _case()
    lm, tlp, one = m(bt["uid"], bt["q"], bt["types"], None, bt["q"], bt["qlen"], bt["dec"])
    loss0, stats0 = emdr2_loss(lm, tlp, one, bt["labels"], bt["mask"], 601)
    loss0.backward()
    g0 = _grads(m)
    for micro, groups in ((3, 3), (5, 5), (12, 8)):
        _clear(m)
        seen = []
        loss, stats = _fb(m, bt, micro, on_group=seen.append)
        assert seen == list(range(groups))
        assert abs(float(loss) - float(loss0)) <= 2e-6 * abs(float(loss0)), (micro, float(loss), float(loss0))
        for k in ("lm_loss", "retriever_loss", "retriever_utility", "null_block_lm_loss"):
            assert abs(float(stats[k]) - float(stats0[k])) <= 4e-6 * max(1.0, abs(float(stats0[k]))), (micro, k)
        g = _grads(m)
        assert set(g) == set(g0) and max(_rel(g[k], g0[k]) for k in g0) < 2e-5, micro


def test_a_short_last_batch_through_the_flat_bucket_sink():
    """The same through training.FlatAdam: two full steps of 8 questions in 4 groups teach the sink 4 contributions per parameter, the
    third step holds 3 questions (3 groups of one): the contribution pattern deviates, nothing is lost (pattern_changes counts it), and the
    step's gradients are those of the undivided 3-question batch."""
    from emdr2_amd.model import kernels as K
    from emdr2_amd.model.emdr2_model import emdr2_loss
    from emdr2_amd.training import FlatAdam
    try:
        m, retr, bt = _case()
        opt = K.GRAD_SINK = FlatAdam(m, lr=0.0, weight_decay=0.0, clip_grad=1.0, bucket_bytes=1 << 20)
        for step in range(2):
            opt.zero_grad()
            _fb(m, bt, 4)
            opt.finish()
            opt.step()
        short = {k: v[:3] for k, v in bt.items()}
        full_out = retr.out
        retr.out = tuple(t[:3] if (t is not None and t.shape[0] == 8) else (t[:3 * 4] if t is not None else None) for t in full_out)
        opt.zero_grad()
        m.forward_backward(short["uid"], short["q"], short["types"], None, short["q"], short["qlen"], short["dec"], short["labels"], short["mask"], 601,
                           micro_batches=4)
        opt.finish()
        g_groups = torch.cat([b["grad"].clone() for b in opt.buckets])
        assert opt.pattern_changes == 1
        opt.zero_grad()
        lm, tlp, one = m(short["uid"], short["q"], short["types"], None, short["q"], short["qlen"], short["dec"])
        loss, _ = emdr2_loss(lm, tlp, one, short["labels"], short["mask"], 601)
        loss.backward()
        opt.finish()
        g_whole = torch.cat([b["grad"].clone() for b in opt.buckets])
        assert _rel(g_groups, g_whole) < 2e-5
    finally:
        K.GRAD_SINK = None


def test_groups_through_the_flat_bucket_sink_and_one_optimizer_step():
    """With training.FlatAdam as the gradient sink: m contributions per parameter land in the fp32 buckets, the learned pattern is m times
    the undivided one, and the updated parameters equal those of the undivided step."""
    from emdr2_amd.model import kernels as K
    from emdr2_amd.model.emdr2_model import emdr2_loss
    from emdr2_amd.training import FlatAdam
    res = {}
    try:
        for micro in (1, 4):
            m, retr, bt = _case()
            opt = K.GRAD_SINK = FlatAdam(m, lr=1e-3, weight_decay=0.1, clip_grad=1.0, bucket_bytes=1 << 20)
            for step in range(2):
                opt.zero_grad()
                if micro == 1:
                    lm, tlp, one = m(bt["uid"], bt["q"], bt["types"], None, bt["q"], bt["qlen"], bt["dec"])
                    loss, _ = emdr2_loss(lm, tlp, one, bt["labels"], bt["mask"], 601)
                    loss.backward()
                else:
                    loss, _ = _fb(m, bt, micro)
                opt.finish()
                grads = torch.cat([b["grad"].clone() for b in opt.buckets])
                opt.step()
            res[micro] = (float(loss), grads, torch.cat([b["master"].clone() for b in opt.buckets]), dict((id(p), c) for p, c in opt.expected.items()),
                          [id(p) for p in opt.params], opt.pattern_changes)
            K.GRAD_SINK = None
        (l1, g1, w1, e1, o1, pc1), (l4, g4, w4, e4, o4, pc4) = res[1], res[4]
        assert pc1 == 0 and pc4 == 0                               # the pattern learned in step 1 held in step 2
        c1, c4 = [e1[i] for i in o1], [e4[i] for i in o4]
        # everything behind the search runs 4 times; the query tower's backward once
        assert all(b in (a, 4 * a) for a, b in zip(c1, c4)) and any(b == 4 * a and a > 0 for a, b in zip(c1, c4))
        assert abs(l1 - l4) < 1e-4 * abs(l1)
        # step 2 starts from parameters that already differ by round-off: its gradients agree to 4e-5 (measured), and Adam turns a relative
        # gradient difference on near-zero-gradient elements into up to 2 x lr of update: the parameters agree to 1.0e-5 (measured; the bound
        # leaves a decade)
        assert _rel(g4, g1) < 1e-3 and _rel(w4, w1) < 1e-4
    finally:
        K.GRAD_SINK = None


def test_dropout_draws_differ_between_groups_and_group_zero_is_the_undivided_stream():
    from emdr2_amd.model import kernels as K
    from emdr2_amd.model.emdr2_model import emdr2_loss
    m, retr, bt = _case(p=0.1)
    K.DROPOUT.step = 3
    lm, tlp, one = m(bt["uid"], bt["q"], bt["types"], None, bt["q"], bt["qlen"], bt["dec"])
    loss0, _ = emdr2_loss(lm, tlp, one, bt["labels"], bt["mask"], 601)
    loss0.backward()
    g0 = _grads(m)
    _clear(m)
    loss1, _ = _fb(m, bt, 1)
    assert float(loss1) == float(loss0)                            # m = 1 IS the undivided step, dropout bits included
    assert max(_rel(_grads(m)[k], g0[k]) for k in g0) < 2e-5
    # two groups holding the SAME questions: with a shared mask their losses would be equal
    seeds = []
    site = m.language_model.language_model.encoder.layers[0].mlp._site
    for micro_id in (0, 1, 2):
        K.DROPOUT.micro = micro_id
        seeds.append(K.DROPOUT.seed(site))
    K.DROPOUT.micro = 0
    assert len(set(seeds)) == 3
    dup = {k: torch.cat([v[:4], v[:4]]) for k, v in bt.items()}
    m.evidence_retriever.out = tuple(torch.cat([t[:t.shape[0] // 2], t[:t.shape[0] // 2]]) if t is not None else None for t in m.evidence_retriever.out)
    _clear(m)
    per_group = []
    m.forward_backward(dup["uid"], dup["q"], dup["types"], None, dup["q"], dup["qlen"], dup["dec"], dup["labels"], dup["mask"], 601,
                       micro_batches=2, on_group=lambda i: per_group.append(i))
    assert per_group == [0, 1] and K.DROPOUT.micro == 0


def test_kl_div_variant_and_argument_checks():
    m, retr, bt = _case()
    l1, s1 = _fb(m, bt, 1, ret_kldiv=True)
    g1 = _grads(m)
    _clear(m)
    l2, s2 = _fb(m, bt, 2, ret_kldiv=True)
    assert abs(float(l1) - float(l2)) < 2e-6 * abs(float(l1)) and abs(float(s1["retriever_loss"]) - float(s2["retriever_loss"])) < 1e-6
    assert max(_rel(_grads(m)[k], g1[k]) for k in g1) < 2e-5
    with pytest.raises(ValueError):
        _fb(m, bt, 0)
    m.eval()
    with pytest.raises(ValueError):
        _fb(m, bt, 2)
