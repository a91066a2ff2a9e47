"""GPU: north_star's fp32 tolerance -- "reader logits within 1e-3 fp32" -- exercised.

The reference runs in fp32 when `--fp16` is not given (megatron/training.py:55-56,92: FP16_Module / FP16_Optimizer only under args.fp16).
`transformer.Config(compute_dtype="fp32")` runs the SAME module tree through the validation-only fp32 path (emdr2_amd/model/kernels_f32.py,
csrc/fp32_ops.hip: fp32 activations, the fp32 masters as weights, fp32 MFMA `v_mfma_f32_32x32x2_f32`, composed masked-softmax attention, fp32
LayerNorm / GELU / log-softmax).  Against the fp32 oracle (oracle/transformer_oracle.py, pinned on the reference's modules run in fp32):

  * the tiny EMDR2 model end to end: reader logits, one-context logits, prior, both losses at 1e-3 (max-normalised), EVERY parameter gradient
    at 1e-3 RMS-normalised (and 2e-3 max-normalised);
  * one base-size layer (H = 768, 12 heads, FFN 3072, S = 512) of each kind, encoder and decoder with cross-attention: output, input
    gradients and every parameter gradient at 1e-3.

Measured (MI355X): logits 7e-7, worst parameter gradient 2.6e-5 RMS / 2.1e-5 max -- fp32 summation-order noise; the bounds are north_star's."""
import numpy as np
import pytest
import torch

from oracle import transformer_oracle as to

pytestmark = pytest.mark.gpu
CFG = dict(layers=2, hidden=128, heads=2, ffn=256, max_pos=128)
TOL = 1e-3


def _cfg(**kw):
    from emdr2_amd.model.transformer import Config
    d = dict(num_layers=CFG["layers"], hidden_size=CFG["hidden"], num_attention_heads=CFG["heads"], ffn_hidden_size=CFG["ffn"],
             max_position_embeddings=CFG["max_pos"], init_method_std=0.05, compute_dtype="fp32")
    d.update(kw)
    return Config(**d)


def _params_cpu(module):
    return {k: v.detach().float().cpu() for k, v in module.state_dict().items()}


def _ids(rng, shape, vocab):
    x = rng.integers(5, vocab, size=shape)
    for r in x.reshape(-1, shape[-1]):
        r[int(rng.integers(shape[-1] // 2, shape[-1] + 1)):] = 0
    return torch.from_numpy(x.astype(np.int64))


def _max_rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _rms_rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _perturb(m, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.05 * torch.randn(p.shape, generator=g, device="cuda"))


def test_tiny_emdr2_model_end_to_end_in_fp32_vs_the_fp32_oracle():
    """a9-a14 in fp32: EMDR2Model.forward_assembled (training, update_retriever) + emdr2_loss + backward on assembled inputs."""
    from emdr2_amd.model import kernels as K
    from emdr2_amd.model.emdr2_model import EMDR2Model, emdr2_loss
    rng = np.random.default_rng(7)
    B, Kk, S_ret, S, L, V = 4, 8, 32, 64, 32, 640
    torch.manual_seed(0)
    m = EMDR2Model(None, _cfg(), V, 512, Kk, S, S_ret, cls_id=2, sep_id=3)
    _perturb(m, 5)
    m.train()
    assert not m.language_model.language_model.packs()                   # dense layouts in the fp32 mode
    qb = _ids(rng, (B, S_ret), 512); ctx = _ids(rng, (B, Kk, S_ret), 512); typ = torch.zeros_like(ctx)
    qext, qone = _ids(rng, (B * Kk, S), 600), _ids(rng, (B * Kk, S), 600)
    dec = _ids(rng, (B, L), 600)
    labels = torch.roll(dec, -1, 1); labels[:, -1] = 0
    loss_mask = (labels != 0).float()
    q_logits = m.retriever_embedder(qb.cuda(), None, torch.zeros_like(qb).cuda(), "query")
    assert q_logits.dtype == torch.float32
    lm, tlp, one = m.forward_assembled(q_logits, ctx.cuda(), typ.cuda(), qext.cuda(), qone.cuda(), dec.cuda())
    assert lm.dtype == torch.float32 and tlp.dtype == torch.float32
    loss, stats = emdr2_loss(lm, tlp, one, labels.cuda(), loss_mask.cuda(), eos_id=601)
    loss.backward()

    P = {k: v.requires_grad_(True) for k, v in _params_cpu(m).items()}
    lm_r, tlp_r, one_r = to.emdr2_forward(P, CFG, qb, torch.zeros_like(qb), ~to.make_attention_mask_3d(qb, qb), ctx, typ, qext, qone, dec)
    lm_loss_r = to.reader_ce_loss(lm_r, labels, loss_mask)
    r_loss_r, util_r, null_r = to.retriever_loss_and_utility(one_r, tlp_r, labels, loss_mask, 601)
    (lm_loss_r + r_loss_r).backward()
    one_t = one.materialize()
    assert one_t.dtype == torch.float32 and tuple(one_t.shape) == tuple(one_r.shape)
    # ALL positions, padded ones included: the dense fp32 path computes what the reference computes everywhere
    assert _max_rel(lm.cpu(), lm_r) < TOL, _max_rel(lm.cpu(), lm_r)
    assert _max_rel(one_t.cpu(), one_r) < TOL, _max_rel(one_t.cpu(), one_r)
    assert float((tlp.detach().cpu() - tlp_r.detach()).abs().max()) < TOL
    assert abs(float(stats["lm_loss"]) - float(lm_loss_r)) < TOL * float(lm_loss_r)
    assert abs(float(stats["retriever_loss"]) - float(r_loss_r)) < TOL * abs(float(r_loss_r))
    assert abs(float(stats["retriever_utility"]) - float(util_r)) < TOL * max(1.0, abs(float(util_r)))
    worst, zeros = {}, []
    gscale = max(float(P[k].grad.abs().max()) for k in P if P[k].grad is not None)
    for k, p in m.named_parameters():
        g_ref = P[k].grad
        if g_ref is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        assert p.grad is not None, k
        if float(g_ref.abs().max()) < 1e-5 * gscale:
            # analytically-zero gradients -- the K third of a QKV bias (softmax is invariant to a per-query shift of the scores), the context
            # tower's final LayerNorm bias (the prior's softmax over K is invariant to adding one vector to every passage embedding): both
            # sides hold fp32 round-off, which is held to the same smallness, not compared digit by digit
            assert float(p.grad.abs().max()) < 1e-5 * gscale, (k, float(p.grad.abs().max()), gscale)
            zeros.append(k)
            continue
        worst[k] = (_rms_rel(p.grad.cpu(), g_ref), _max_rel(p.grad.cpu(), g_ref))
    assert len(worst) > 100 and len(zeros) < 12, (len(worst), zeros)
    bad = {k: v for k, v in worst.items() if v[0] > TOL or v[1] > 2 * TOL}
    assert not bad, bad
    print("fp32 parity: logits %.2e, one-context logits %.2e, worst gradient (rms, max) %s; analytically-zero gradients: %s"
          % (_max_rel(lm.cpu(), lm_r), _max_rel(one_t.cpu(), one_r), max(worst.values()), zeros))
    K.WEIGHTS.invalidate()


@pytest.mark.parametrize("kind", ["encoder", "decoder"])
def test_one_base_size_layer_in_fp32_vs_the_fp32_oracle(kind):
    """a13 at the benchmark's layer shape (H 768, 12 heads of 64, FFN 3072, S 512 -- decoder: 32 queries over 512 encoder positions): output,
    input gradient(s) and every parameter gradient of ParallelTransformerLayer against the oracle's transformer_layer, both in fp32."""
    from emdr2_amd.model.transformer import ParallelTransformerLayer
    cfg = _cfg(num_layers=12, hidden_size=768, num_attention_heads=12, ffn_hidden_size=3072, max_position_embeddings=512, init_method_std=0.02)
    torch.manual_seed(1)
    layer = ParallelTransformerLayer(cfg, 0.02 / (24 ** 0.5), kind)
    _perturb(layer, 9)
    rng = np.random.default_rng(3)
    b, S, L = 2, 512, 32
    g = torch.Generator().manual_seed(2)
    enc_ids = _ids(rng, (b, S), 1000)
    if kind == "encoder":
        x = torch.randn((b, S, 768), generator=g)
        ids, causal, enc, mask = enc_ids, False, None, ~to.make_attention_mask_3d(enc_ids, enc_ids)
        ed_mask = None
    else:
        x = torch.randn((b, L, 768), generator=g)
        ids = _ids(rng, (b, L), 1000)
        enc = torch.randn((b, S, 768), generator=g)
        causal = True
        mask = ~(to.make_attention_mask_3d(ids, ids) * to.make_history_mask_3d(ids))
        ed_mask = ~to.make_attention_mask_3d(ids, enc_ids)
    xg = x.cuda().requires_grad_(True)
    eg = enc.cuda().requires_grad_(True) if enc is not None else None
    out = layer(xg, ids.cuda(), causal, eg, enc_ids.cuda() if enc is not None else None)
    assert out.dtype == torch.float32
    w = torch.randn(out.shape, generator=g)
    (out * w.cuda()).sum().backward()

    P = {"l." + k: v.requires_grad_(True) for k, v in _params_cpu(layer).items()}
    xr = x.clone().requires_grad_(True)
    er = enc.clone().requires_grad_(True) if enc is not None else None
    ref = to.transformer_layer(P, "l", 12, xr, mask[:, None], er, ed_mask[:, None] if ed_mask is not None else None)
    (ref * w).sum().backward()
    assert _max_rel(out.cpu(), ref) < TOL, _max_rel(out.cpu(), ref)
    assert _rms_rel(xg.grad.cpu(), xr.grad) < TOL
    if enc is not None:
        assert _rms_rel(eg.grad.cpu(), er.grad) < TOL
    bad = {}
    gscale = max(float(v.grad.abs().max()) for v in P.values() if v.grad is not None)
    for k, p in layer.named_parameters():
        g_ref = P["l." + k].grad
        assert p.grad is not None and g_ref is not None, k
        if float(g_ref.abs().max()) < 1e-5 * gscale:                     # (the K third of the QKV bias is hidden inside a non-zero tensor here)
            assert float(p.grad.abs().max()) < 1e-5 * gscale, k
            continue
        r = (_rms_rel(p.grad.cpu(), g_ref), _max_rel(p.grad.cpu(), g_ref))
        if r[0] > TOL or r[1] > 2 * TOL:
            bad[k] = r
    assert not bad, bad


def test_fp32_mode_refuses_what_it_does_not_implement():
    """Dropout and incremental decoding are not part of the validation mode: they raise instead of silently computing something else."""
    from emdr2_amd.model.transformer import T5Model
    m = T5Model(_cfg(hidden_dropout=0.1, attention_dropout=0.1), 640).train()
    ids = torch.randint(5, 600, (2, 64), device="cuda")
    with pytest.raises(ValueError):
        m(ids, ids[:, :32].contiguous())
    m.eval()
    logits, enc = m(ids, ids[:, :32].contiguous())                       # eval: dropout off -> runs
    assert logits.dtype == torch.float32 and bool(torch.isfinite(logits).all())
    with pytest.raises(ValueError):
        m.language_model.init_decode_state(2, 32)
