"""GPU: the HIP path against the bf16-FAITHFUL form of the oracle (oracle/transformer_oracle.py `bf16_faithful`: the fp32 oracle, pinned on the
reference's modules, with a round-to-bf16 wherever the kernels store bf16 and the fused attention's lazy running maximum walked step by step).

Every tensor is measured in TWO metrics -- max-normalised max|a-b| / max|b| and RMS-normalised ||a-b|| / ||b|| (systematic errors in
small-magnitude tensors: LayerNorm / bias gradients, late-position logits, show in the second) -- and must satisfy, in each,

        error  <=  max( TOL * scale(b),  NOISE_X_{ACT,GRAD} * | b - b32 | )

where b is the bf16-faithful oracle and b32 the SAME oracle in plain fp32: the distance between two valid roundings of one computation IS
the bf16 noise of that tensor, cancellation included (the K-third of a QKV bias gradient is analytically zero -- softmax ignores a common
shift of the keys -- so both paths hold pure round-off there; a LayerNorm bias gradient inherits it through W^T).  A systematic error
larger than that noise fails whatever the tensor's magnitude; nothing is excused by the scale of OTHER tensors.

* TEACHER-FORCED, layer by layer (`test_every_layer_*`): every transformer layer of every stack is re-run in the oracle from the HIP path's
  own input and upstream gradient; output, input gradient and parameter gradients are compared.  One layer deep the two paths differ by
  isolated one-ulp flips (1e-4 .. 2e-3 RMS measured).  A bf16 ulp is 2^-8 .. 2^-7 of its element, so ONE flip on the largest element costs
  up to 7.8e-3 in the max-normalised metric -- that bound cannot go below ~1e-2; the RMS bound is the sharp one.
* END TO END (`test_*_vs_bf16_faithful_oracle`): logits, losses, all gradients through the whole depth.  A store turns a perturbation of
  relative size eps into noise of size ~sqrt(eps * ulp) (an element within eps of a rounding boundary lands one ulp away), so two
  implementations that are not bit-identical in their fp32 accumulation order drift to the bf16 noise floor within a few layers
  (measured: 1e-4 per op, 7e-4 after one layer, 2.6e-3 after two, 6e-3 on the logits of a 2 + 2 layer reader, 1.1e-2 at 12 + 12 layers).

Reference lines: megatron/model/transformer.py:474-563 (layer), emdr2_model.py:87-214 (forward), train_e2eqa.py:72-181 (loss).
north_star quotes 1e-3 for fp32 compute; this build computes in bf16 only, so these are the tight comparisons."""
import numpy as np
import pytest
import torch

from oracle import transformer_oracle as to

pytestmark = pytest.mark.gpu
CFG = dict(layers=2, hidden=128, heads=2, ffn=256, max_pos=128)
LAYER_ACT = (1.2e-2, 3e-3)           # (max-normalised, RMS-normalised): one layer from the same input, its output
LAYER_GRAD = (2e-2, 8e-3)            # ... its input gradient and parameter gradients from the same upstream gradient
E2E_ACT = (2e-2, 1.5e-2)             # end to end: activations / logits
E2E_GRAD = (4e-2, 2.5e-2)            # end to end: parameter gradients
# ... or this multiple of the tensor's own bf16 noise |oracle_bf16 - oracle_fp32| (measured: the HIP path sits at 0.8 .. 1.2 x that noise
# on every tensor that needs it, once the oracle's attention backward follows the kernels' -- D = rowsum(dO o O) from the STORED bf16 output
# amplifies round-off into the cross-attention query gradients whenever the values of a question's keys share a large common component)
NOISE_X_ACT, NOISE_X_GRAD = 3.0, 3.0


def _cfg(layers=2, hidden=128, heads=2, ffn=256, max_pos=128, std=0.05):
    from emdr2_amd.model.transformer import Config
    return Config(num_layers=layers, hidden_size=hidden, num_attention_heads=heads, ffn_hidden_size=ffn, max_position_embeddings=max_pos,
                  init_method_std=std)


def _ids(rng, shape, vocab, lo_frac=0.5):
    x = rng.integers(5, vocab, size=shape)
    for r in x.reshape(-1, shape[-1]):
        r[int(rng.integers(max(1, int(shape[-1] * lo_frac)), shape[-1] + 1)):] = 0
    return torch.from_numpy(x.astype(np.int64))


def _perturb(m, seed, scale=0.05):
    g = torch.Generator(device="cuda").manual_seed(seed)
    with torch.no_grad():
        for p in m.parameters():
            p.add_(scale * torch.randn(p.shape, generator=g, device="cuda"))


def _check(report, name, a, b, tol, b32=None):
    """Record the two metrics of `a` (HIP) against `b` (bf16-faithful oracle); `b32` (fp32 oracle) supplies the tensor's own noise."""
    NOISE_X = NOISE_X_GRAD if tol in (LAYER_GRAD, E2E_GRAD) else NOISE_X_ACT
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    d = a - b
    e_max, e_rms = float(d.abs().max()), float(d.norm())
    s_max, s_rms = float(b.abs().max()) + 1e-30, float(b.norm()) + 1e-30
    n_max = n_rms = 0.0
    if b32 is not None:
        n = b - b32.detach().float().cpu()
        n_max, n_rms = float(n.abs().max()), float(n.norm())
    ok = e_max <= max(tol[0] * s_max, NOISE_X * n_max) and e_rms <= max(tol[1] * s_rms, NOISE_X * n_rms)
    report.append(dict(name=name, max=e_max / s_max, rms=e_rms / s_rms, noise_max=n_max / s_max, noise_rms=n_rms / s_rms, tol=tol, ok=ok))


def _assert_report(report, max_admitted=0):
    """All tensors are measured first, then every violator is listed (one run shows the whole picture).  `max_admitted` bounds how many tensors
    may pass by the noise clause only: the count measured when the bound was set (r04) plus a small margin -- a change that pushes more
    tensors out of the plain tolerance fails here even if each of them is still within 3 x its own bf16 noise."""
    fmt = lambda r: "%s: max %.3g rms %.3g (tol %.3g / %.3g; own bf16 noise %.3g / %.3g)" % (r["name"], r["max"], r["rms"], r["tol"][0], r["tol"][1],
                                                                                         r["noise_max"], r["noise_rms"])
    within = [r for r in report if r["max"] <= r["tol"][0] and r["rms"] <= r["tol"][1]]
    print("%d tensors, %d inside the plain tolerance, %d admitted by their own noise" % (len(report), len(within), sum(r["ok"] for r in report) - len(within)))
    print("worst RMS-normalised:", [fmt(r) for r in sorted(report, key=lambda r: -r["rms"])[:3]])
    print("worst max-normalised:", [fmt(r) for r in sorted(report, key=lambda r: -r["max"])[:3]])
    bad = [fmt(r) for r in report if not r["ok"]]
    for b in bad:
        print("VIOLATION", b)
    assert not bad, "%d tensors outside their bounds (listed above)" % len(bad)
    admitted = sum(r["ok"] for r in report) - len(within)
    assert admitted <= max_admitted, "%d tensors pass only by their own-noise clause (bound %d)" % (admitted, max_admitted)


def _check_grads(report, module, P, P32, prefix, tol):
    for k, p in module.named_parameters():
        ref = P[prefix + k].grad
        if ref is None:                                          # e.g. the reader's token-type table: never used (t5_model.py:124-137)
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        assert p.grad is not None, k
        _check(report, "grad " + k, p.grad, ref, tol, P32[prefix + k].grad)


def _both_forms(params, fn):
    """Run `fn(P)` (forward + backward on a fresh copy of the parameters) in the bf16-faithful and in the fp32 form of the oracle."""
    out = []
    for bf16 in (True, False):
        P = {k: v.detach().float().cpu().requires_grad_(True) for k, v in params.items()}
        with to.bf16_faithful(bf16):
            out.append((P, fn(P)))
    return out


def test_bert_tower_vs_bf16_faithful_oracle():
    from emdr2_amd.model.transformer import PretrainedBertModel
    torch.manual_seed(0)
    m = PretrainedBertModel(_cfg(), 512)
    _perturb(m, 1)
    rng = np.random.default_rng(0)
    ids = _ids(rng, (64, 64), 512)
    types = torch.from_numpy(rng.integers(0, 2, size=(64, 64)).astype(np.int64))
    out = m(ids.cuda(), types.cuda())
    w = torch.randn((64, CFG["hidden"]), generator=torch.Generator().manual_seed(3))
    (out.float() * w.cuda()).sum().backward()

    def oracle(P):
        ref = to.bert_embed(P, "bert", CFG, ids, ~to.make_attention_mask_3d(ids, ids), types)
        (ref * w).sum().backward()
        return ref
    (P, ref), (P32, ref32) = _both_forms({"bert." + k: v for k, v in m.state_dict().items()}, oracle)
    report = []
    _check(report, "cls embedding", out, ref, E2E_ACT, ref32)
    _check_grads(report, m, P, P32, "bert.", E2E_GRAD)
    _assert_report(report)


def test_reader_vs_bf16_faithful_oracle():
    from emdr2_amd.model.transformer import T5Model
    torch.manual_seed(0)
    m = T5Model(_cfg(), 640)
    _perturb(m, 2)
    rng = np.random.default_rng(1)
    enc_ids, dec_ids = _ids(rng, (32, 96), 640), _ids(rng, (32, 32), 640, 0.1)
    logits, enc = m(enc_ids.cuda(), dec_ids.cuda())
    real, dreal = enc_ids != 0, dec_ids != 0
    w = torch.randn(logits.shape, generator=torch.Generator().manual_seed(4)) * 0.1 * dreal[..., None]
    (logits.float() * w.cuda()).sum().backward()

    def oracle(P):
        e_ref = to.t5_encode(P, "t5", CFG, enc_ids, ~to.make_attention_mask_3d(enc_ids, enc_ids))
        d_mask = ~(to.make_attention_mask_3d(dec_ids, dec_ids) * to.make_history_mask_3d(dec_ids))
        l_ref = to.t5_decode(P, "t5", CFG, dec_ids, e_ref, d_mask, ~to.make_attention_mask_3d(dec_ids, enc_ids))
        (l_ref * w).sum().backward()
        return e_ref, l_ref
    (P, (e_ref, l_ref)), (P32, (e32, l32)) = _both_forms({"t5." + k: v for k, v in m.state_dict().items()}, oracle)
    report = []
    _check(report, "encoder output", enc.cpu()[real], e_ref[real], E2E_ACT, e32[real])
    _check(report, "logits", logits.cpu()[dreal], l_ref[dreal], E2E_ACT, l32[dreal])
    _check_grads(report, m, P, P32, "t5.", E2E_GRAD)
    _assert_report(report)


# ---- the EMDR2 step: end to end and teacher-forced layer by layer -------------------------------------------------------------------------
class _LayerTap(object):
    """Forward hooks on every ParallelTransformerLayer of a module: records, for each grad-enabled call, the layer's input, its side inputs,
    its output and (through tensor hooks) the gradients that arrive at the output and leave at the input."""

    def __init__(self, module):
        from emdr2_amd.model.transformer import ParallelTransformerLayer
        self.records, self.handles = [], []
        for name, mod in module.named_modules():
            if isinstance(mod, ParallelTransformerLayer):
                self.handles.append(mod.register_forward_hook(self._hook(name)))

    def _hook(self, name):
        def fn(mod, args, out):
            if not (torch.is_grad_enabled() and out.requires_grad):
                return
            rec = dict(name=name, mod=mod, x=args[0].detach(), ids=args[1], causal=bool(args[2]) if len(args) > 2 else False,
                       enc=args[3].detach() if len(args) > 3 and args[3] is not None else None, enc_ids=args[4] if len(args) > 4 else None,
                       out=out.detach())
            out.register_hook(lambda g, r=rec: r.__setitem__("g_out", g.detach()))
            if args[0].requires_grad:
                args[0].register_hook(lambda g, r=rec: r.__setitem__("g_in", g.detach()))
            self.records.append(rec)
        return fn

    def close(self):
        for h in self.handles:
            h.remove()


def _dense(t, info):
    """A layer-side tensor in the reference's [b, s, h] shape on the CPU (packed -> zeros at the dropped pad rows), and its token ids."""
    from emdr2_amd.model import kernels as K
    if isinstance(info, K.PackedSeqs):
        base = info
        ids = info.dense_ids
        if info.group != 1:                                   # FiD: the same rows, K consecutive sequences per question
            base = object.__new__(K.PackedSeqs); base.__dict__.update(info.__dict__)
            base.group, base.n = 1, ids.shape[0]
        d = K.unpack_rows(t, base)
        if info.group != 1:
            d = d.reshape(info.n, info.group * info.S, -1)
            ids = ids.reshape(info.n, -1)
        return d.float().cpu(), ids.cpu()
    return t.float().cpu(), info.cpu()


def _check_layers(report, tap, heads):
    """Re-run every recorded layer in the oracle (both forms) from the HIP path's own input / upstream gradient."""
    assert tap.records
    for rec in tap.records:
        mod, name = rec["mod"], rec["name"]
        x, ids = _dense(rec["x"], rec["ids"])
        out, _ = _dense(rec["out"], rec["ids"])
        g_out, _ = _dense(rec["g_out"], rec["ids"])
        real = ids != 0
        self_mask = to.make_attention_mask_3d(ids, ids)
        if rec["causal"]:
            self_mask = self_mask * to.make_history_mask_3d(ids)
        enc = ed_mask = None
        if rec["enc"] is not None:
            enc, enc_ids = _dense(rec["enc"], rec["enc_ids"])
            ed_mask = (~to.make_attention_mask_3d(ids, enc_ids))[:, None]

        def oracle(P):
            xr = x.clone().requires_grad_(True)
            ref = to.transformer_layer(P, name, heads, xr, (~self_mask)[:, None], enc, ed_mask)
            ref.backward(g_out * real[..., None])
            return ref.detach(), xr.grad
        (P, (ref, gx)), (P32, (ref32, gx32)) = _both_forms({name + "." + k: v for k, v in mod.state_dict().items()}, oracle)
        _check(report, name + " out", out[real], ref[real], LAYER_ACT, ref32[real])
        if "g_in" in rec:
            g_in, _ = _dense(rec["g_in"], rec["ids"])
            _check(report, name + " d input", g_in[real], gx[real], LAYER_GRAD, gx32[real])
        for k, p in mod.named_parameters():
            _check(report, name + " d " + k, p.grad, P[name + "." + k].grad, LAYER_GRAD, P32[name + "." + k].grad)


def _fp32_mode_against_the_fp32_oracle(m, cfg_kw, shapes, inputs, ref32, P32, tol=1e-3):
    """r06: the SAME weights and batch through the validation-only fp32 compute mode (Config(compute_dtype="fp32"), kernels_f32.py) against
    the fp32 form of the oracle that the bf16 comparison has just computed: north_star's "logits within 1e-3 fp32" at the BASELINE
    architecture (12 layers per stack, H 768, full vocabularies).  Every position, padded ones included (dense layouts)."""
    from emdr2_amd.model import kernels as K
    from emdr2_amd.model.emdr2_model import EMDR2Model, emdr2_loss
    from emdr2_amd.model.transformer import Config
    V_t5, V_bert, Kk, S, S_ret, eos = shapes
    qb, ctx, typ, qext, qone, dec, labels, loss_mask = inputs
    lm32, tlp32, one32, lm_loss32, r_loss32 = ref32
    c = _cfg(**cfg_kw)
    cfg32 = Config(num_layers=c.num_layers, hidden_size=c.hidden_size, num_attention_heads=c.num_attention_heads, ffn_hidden_size=c.ffn_hidden_size,
                   max_position_embeddings=c.max_position_embeddings, init_method_std=c.init_method_std, compute_dtype="fp32")
    m32 = EMDR2Model(None, cfg32, V_t5, V_bert, Kk, S, S_ret, cls_id=2, sep_id=3)
    m32.load_state_dict(m.state_dict())
    m32.train()
    q_logits = m32.retriever_embedder(qb.cuda(), None, torch.zeros_like(qb).cuda(), "query")
    lm, tlp, one = m32.forward_assembled(q_logits, ctx.cuda(), typ.cuda(), qext.cuda(), qone.cuda(), dec.cuda())
    loss, stats = emdr2_loss(lm, tlp, one, labels.cuda(), loss_mask.cuda(), eos_id=eos)
    loss.backward()
    mx = lambda a, b: float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))
    rms = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
    worst_act = max(mx(lm.detach().cpu(), lm32), mx(one.materialize().cpu(), one32))
    assert worst_act < tol, worst_act
    assert float((tlp.detach().cpu() - tlp32).abs().max()) < tol
    assert abs(float(stats["lm_loss"]) - lm_loss32) < tol * abs(lm_loss32) and abs(float(stats["retriever_loss"]) - r_loss32) < tol * abs(r_loss32)
    gscale = max(float(v.grad.abs().max()) for v in P32.values() if v.grad is not None)
    worst, n, rows = 0.0, 0, []
    for k, p in m32.named_parameters():
        g_ref = P32[k].grad
        if g_ref is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        if float(g_ref.abs().max()) < 1e-5 * gscale:                     # analytically-zero gradients: fp32 round-off on both sides
            assert float(p.grad.abs().max()) < 1e-5 * gscale, k
            continue
        # RMS error over the tensor's own RMS -- but not below 1e-4 of the largest gradient element: the few near-cancelling gradients (the last
        # context-tower layers' output biases, 3e-5 of the largest: the prior's softmax over K removes almost all of a shift common to every passage
        # embedding) carry fp32 cancellation noise of 1e-3 of THEMSELVES through twelve layers (measured 1.05e-3; everything else <= 2.7e-4)
        n_el = float(g_ref.numel()) ** 0.5
        own = float(g_ref.double().norm()) / n_el
        r = float((p.grad.cpu().double() - g_ref.double()).norm()) / n_el / max(own, 1e-4 * gscale)
        rows.append((r, k, float(g_ref.abs().max()) / gscale))
        worst, n = max(worst, r), n + 1
    rows.sort(reverse=True)
    print("fp32 mode vs fp32 oracle: activations %.2e (max-normalised), worst of %d parameter gradients %.2e (RMS-normalised)" % (worst_act, n, worst))
    print("worst five (rms error, tensor, its max / the largest gradient):", rows[:5])
    assert rows[0][0] < tol, rows[:5]
    del m32
    K.WEIGHTS.invalidate()


def _emdr2_case(cfg_kw, oracle_cfg, B, Kk, S_ret, S, L, V_t5, V_bert, seed, perturb, layerwise, max_admitted=0, fp32_too=False):
    from emdr2_amd.model.emdr2_model import EMDR2Model, emdr2_loss
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    m = EMDR2Model(None, _cfg(**cfg_kw), V_t5, V_bert, Kk, S, S_ret, cls_id=2, sep_id=3)
    _perturb(m, seed + 1, perturb)
    m.train()
    qb = _ids(rng, (B, S_ret), V_bert - 8, 0.05); ctx = _ids(rng, (B, Kk, S_ret), V_bert - 8, 0.3); typ = torch.zeros_like(ctx)
    qext, qone = _ids(rng, (B * Kk, S), V_t5 - 8, 0.5), _ids(rng, (B * Kk, S), V_t5 - 8, 0.2)
    dec = _ids(rng, (B, L), V_t5 - 8, 0.1)
    eos = V_t5 - 7
    labels = torch.roll(dec, -1, 1); labels[:, -1] = 0
    loss_mask = (labels != 0).float()
    tap = _LayerTap(m) if layerwise else None
    try:
        q_logits = m.retriever_embedder(qb.cuda(), None, torch.zeros_like(qb).cuda(), "query")
        lm, tlp, one = m.forward_assembled(q_logits, ctx.cuda(), typ.cuda(), qext.cuda(), qone.cuda(), dec.cuda())
        loss, stats = emdr2_loss(lm, tlp, one, labels.cuda(), loss_mask.cuda(), eos_id=eos)
        loss.backward()
    finally:
        if tap is not None:
            tap.close()
    report = []
    if layerwise:
        _check_layers(report, tap, oracle_cfg["heads"])
        _assert_report(report, max_admitted)
        return len(tap.records)

    def oracle(P):
        lm_r, tlp_r, one_r = to.emdr2_forward(P, oracle_cfg, qb, torch.zeros_like(qb), ~to.make_attention_mask_3d(qb, qb), ctx, typ, qext, qone, dec)
        lm_loss_r = to.reader_ce_loss(lm_r, labels, loss_mask)
        r_loss_r = to.retriever_loss_and_utility(one_r, tlp_r, labels, loss_mask, eos)[0]
        (lm_loss_r + r_loss_r).backward()
        return lm_r.detach(), tlp_r.detach(), one_r, float(lm_loss_r), float(r_loss_r)
    (P, (lm_r, tlp_r, one_r, lm_loss_r, r_loss_r)), (P32, (lm32, tlp32, one32, lm_loss32, r_loss32)) = _both_forms(m.state_dict(), oracle)
    dreal = dec != 0
    dk = dreal[:, None, :].expand(-1, Kk, -1)
    _check(report, "lm_logits", lm.cpu()[dreal], lm_r[dreal], E2E_ACT, lm32[dreal])
    _check(report, "one-context logits", one.materialize().cpu()[dk], one_r[dk], E2E_ACT, one32[dk])
    _check(report, "topk_log_probs", tlp, tlp_r, E2E_ACT, tlp32)
    for name, got, ref, ref32 in (("lm_loss", float(stats["lm_loss"]), lm_loss_r, lm_loss32), ("retriever_loss", float(stats["retriever_loss"]), r_loss_r, r_loss32)):
        assert abs(got - ref) <= max(2e-3 * abs(ref), NOISE_X_ACT * abs(ref - ref32)), (name, got, ref, ref32)
    _check_grads(report, m, P, P32, "", E2E_GRAD)
    _assert_report(report, max_admitted)
    if fp32_too:
        _fp32_mode_against_the_fp32_oracle(m, cfg_kw, (V_t5, V_bert, Kk, S, S_ret, eos), (qb, ctx, typ, qext, qone, dec, labels, loss_mask),
                                           (lm32, tlp32, one32, lm_loss32, r_loss32), P32)
    return 0


def test_emdr2_step_vs_bf16_faithful_oracle():
    """Rows a9-a14 at test dimensions: forward triple, both losses and every parameter gradient."""
    _emdr2_case({}, CFG, B=4, Kk=8, S_ret=32, S=64, L=32, V_t5=640, V_bert=512, seed=7, perturb=0.05, layerwise=False,
                max_admitted=16)                                  # r04: 12 of 132 tensors needed the noise clause


def test_every_layer_teacher_forced_vs_bf16_faithful_oracle():
    """All four stacks of the EMDR2 step at test dimensions (query tower, context tower, reader encoder, FiD decoder): each layer's output,
    input gradient and parameter gradients from the HIP path's own input and upstream gradient."""
    n = _emdr2_case({}, CFG, B=4, Kk=8, S_ret=32, S=64, L=32, V_t5=640, V_bert=512, seed=7, perturb=0.05, layerwise=True,
                    max_admitted=22)                              # r04: 16 of 128
    assert n == 8                                               # 2 layers x (query, context, reader encoder, reader decoder)


BASE = dict(layers=12, hidden=768, heads=12, ffn=3072)


def test_base_size_emdr2_step_vs_bf16_faithful_oracle():
    """The BASELINE architecture itself -- H 768, 12 heads, FFN 3072, 12 + 12 + 12 + 12 layers, the full vocabularies, S_ret 256, S 512,
    L 32 -- at B = 2, K = 4: EMDR2Model forward + EMDR2 loss + backward against the oracle run on the module's own weights
    (megatron/model/transformer.py:474-563 twelve times per stack, emdr2_model.py:87-214, train_e2eqa.py:72-181).  r06: the same weights and
    batch also run through the validation-only fp32 compute mode and are held to 1e-3 against the oracle's fp32 form (north_star's fp32 bar)."""
    torch.set_num_threads(min(64, torch.get_num_threads() or 1))
    _emdr2_case(dict(max_pos=512, std=0.02, **BASE), BASE, B=2, Kk=4, S_ret=256, S=512, L=32, V_t5=30720, V_bert=30592, seed=11, perturb=0.01,
                layerwise=False, max_admitted=340, fp32_too=True)   # r04: 313 of 692 (gradients through twelve bf16 layers per stack)


def test_every_layer_of_the_base_size_model_teacher_forced():
    """The same model and batch: all 48 layers, each from the HIP path's own input and upstream gradient."""
    torch.set_num_threads(min(64, torch.get_num_threads() or 1))
    n = _emdr2_case(dict(max_pos=512, std=0.02, **BASE), BASE, B=2, Kk=4, S_ret=256, S=512, L=32, V_t5=30720, V_bert=30592, seed=11, perturb=0.01,
                    layerwise=True, max_admitted=135)             # r04: 119 of 768
    assert n == 48
