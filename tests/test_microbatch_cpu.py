"""CPU: the mathematics behind question micro-batching (EMDR2Model.forward_backward) on the ORACLE -- the fp32 restatement that is
pinned on the reference's modules and loss functions (tests/test_oracle_transformer.py): everything after retrieval is per question and both
EMDR2 losses are sums over questions divided by batch-wide token counts (tasks/openqa/e2eqa/train_e2eqa.py:72-181), so running the reference's
forward on groups of questions and taking every mean with the WHOLE batch's denominator reproduces the undivided loss, its statistics and
every parameter gradient.  (The HIP path is held to this in tests/test_microbatch_gpu.py / test_config4_gpu.py.)"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import transformer_oracle as to

CFG = dict(layers=2, hidden=32, heads=2, ffn=64)


def _ids(rng, shape, vocab):
    x = rng.integers(5, vocab, size=shape)
    for r in x.reshape(-1, shape[-1]):
        r[int(rng.integers(shape[-1] // 2, shape[-1] + 1)):] = 0
    return torch.from_numpy(x.astype(np.int64))


def _group_losses(lm, tlp, one, labels, mask, eos, totals, kldiv):
    """The two losses of a group of questions with the batch's denominators (mask sum, batch size)."""
    mask_sum, batch = totals
    ce = F.cross_entropy(lm.float().reshape(-1, lm.shape[-1]), labels.reshape(-1), reduction="none", ignore_index=0)
    lm_loss = torch.sum(ce * mask.reshape(-1)) / mask_sum
    logp = F.log_softmax(one.float(), dim=-1)
    lab = labels.masked_fill(~mask.to(torch.bool), 0)
    gold = torch.gather(logp, -1, lab[:, None, :, None].expand(-1, one.shape[1], -1, 1)).squeeze(-1)
    if kldiv:
        teacher = torch.sum(gold * mask.unsqueeze(1), dim=2) / torch.sum(mask.unsqueeze(1), dim=2)
        return lm_loss, F.kl_div(tlp.float(), torch.softmax(teacher, dim=1), reduction="sum") / batch
    marginal = torch.logsumexp(tlp.float().unsqueeze(-1) + gold, dim=1)
    return lm_loss, -torch.sum(marginal * mask) / mask_sum


@pytest.mark.parametrize("kldiv", [False, True])
def test_groups_of_questions_reproduce_the_reference_loss_and_gradients(kldiv):
    rng = np.random.default_rng(3)
    B, K, S_ret, S, L, V = 8, 3, 16, 32, 8, 96
    P0 = to.random_params(CFG, V, V, max_pos=64, std=0.2, seed=5)
    qb, ctx = _ids(rng, (B, S_ret), V - 4), _ids(rng, (B, K, S_ret), V - 4)
    qext, qone, dec = _ids(rng, (B * K, S), V - 4), _ids(rng, (B * K, S), V - 4), _ids(rng, (B, L), V - 4)
    labels = torch.roll(dec, -1, 1); labels[:, -1] = 0
    mask = (labels != 0).float()
    eos = V - 2

    def run(groups):
        P = {k: v.clone().requires_grad_(True) for k, v in P0.items()}
        totals = (mask.sum(), B)
        total = 0.0
        g = B // groups
        for i in range(groups):
            sl = slice(i * g, (i + 1) * g)
            sk = slice(i * g * K, (i + 1) * g * K)
            lm, tlp, one = to.emdr2_forward(P, CFG, qb[sl], torch.zeros_like(qb[sl]), ~to.make_attention_mask_3d(qb[sl], qb[sl]), ctx[sl],
                                            torch.zeros_like(ctx[sl]), qext[sk], qone[sk], dec[sl])
            a, b = _group_losses(lm, tlp, one, labels[sl], mask[sl], eos, totals, kldiv)
            (a + b).backward()                                       # gradients of the groups add up in .grad, like in the flat fp32 buckets
            total += float((a + b).detach())
        return total, {k: v.grad.clone() for k, v in P.items() if v.grad is not None}

    # the undivided batch through the oracle's own loss functions (= the reference's) ...
    P = {k: v.clone().requires_grad_(True) for k, v in P0.items()}
    lm, tlp, one = to.emdr2_forward(P, CFG, qb, torch.zeros_like(qb), ~to.make_attention_mask_3d(qb, qb), ctx, torch.zeros_like(ctx), qext, qone, dec)
    ref = to.reader_ce_loss(lm, labels, mask) + (to.retriever_kl_div_loss(one, tlp, labels, mask) if kldiv
                                                 else to.retriever_loss_and_utility(one, tlp, labels, mask, eos)[0])
    ref.backward()
    g_ref = {k: v.grad.clone() for k, v in P.items() if v.grad is not None}
    # ... equals 1, 2, 4 and 8 groups with batch-wide denominators
    for groups in (1, 2, 4, 8):
        total, grads = run(groups)
        assert abs(total - float(ref)) < 1e-5 * abs(float(ref)), (groups, total, float(ref))
        assert set(grads) == set(g_ref)
        for k in g_ref:
            assert float((grads[k] - g_ref[k]).abs().max()) <= 1e-5 * float(g_ref[k].abs().max()) + 1e-7, (groups, k)


def test_question_group_bounds_cover_any_batch():
    """a batch the group count does not divide (the short last batch of an epoch under --keep-last) is split like
    `torch.tensor_split`; more groups than questions degrade to one question per group."""
    import torch
    from emdr2_amd.model.emdr2_model import question_group_bounds
    for batch in range(1, 20):
        for m in range(1, 12):
            b = question_group_bounds(batch, m)
            assert len(b) == min(m, batch) and b[0][0] == 0 and b[-1][1] == batch
            assert all(b[i][1] == b[i + 1][0] for i in range(len(b) - 1))
            sizes = [hi - lo for lo, hi in b]
            assert sizes == [t.numel() for t in torch.tensor_split(torch.arange(batch), min(m, batch))]
