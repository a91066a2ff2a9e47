"""GPU: the packed ("varlen") sequence layout (csrc/seqpack.hip, the cu_q / cu_k forms of the attention kernels, kernels.PackedSeqs)
against the dense [batch, S] path it replaces (reference: transformer.py:283-381 over padded grids, emdr2_model.py:148-210).

The claim under test: dropping a sequence's trailing [PAD] rows changes no CONSUMED value -- real positions, logits, losses and every
parameter gradient agree with the dense path (and therefore with the oracle, which the dense path is tested against elsewhere)."""
import numpy as np
import pytest
import torch

from oracle import transformer_oracle as to

pytestmark = pytest.mark.gpu
CFG = dict(layers=2, hidden=128, heads=2, ffn=256, max_pos=128)


def _cfg(**kw):
    from emdr2_amd.model.transformer import Config
    return Config(num_layers=CFG["layers"], hidden_size=CFG["hidden"], num_attention_heads=CFG["heads"], ffn_hidden_size=CFG["ffn"],
                  max_position_embeddings=CFG["max_pos"], init_method_std=0.05, **kw)


def _ragged_ids(rng, n, S, vocab, lo=1):
    x = rng.integers(5, vocab, size=(n, S))
    lens = rng.integers(lo, S + 1, size=n)
    for r, ln in zip(x, lens):
        r[ln:] = 0
    return x.astype(np.int64), lens


def _rel(a, b):
    a, b = a.detach().float(), b.detach().float()
    return float((a - b).abs().max() / (b.abs().max() + 1e-6))


# ---- the layout itself -------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,S", [(1, 8), (7, 64), (300, 100), (3200, 256), (5000, 33)])
def test_layout_matches_numpy(n, S):
    from emdr2_amd.model import kernels as K
    rng = np.random.default_rng(n * 1000 + S)
    ids, lens = _ragged_ids(rng, n, S, 1000)
    ids[0, :] = 0; lens[0] = S                               # an all-pad row keeps every position (the reference's uniform attention)
    if n > 2 and S >= 8:
        ids[2, 1] = 0                                        # an interior zero stays inside its sequence
        ids[2, 5] = 7; lens[2] = max(lens[2], 6)
    types = rng.integers(0, 2, size=(n, S)).astype(np.int64)
    seqs = K.PackedSeqs(torch.from_numpy(ids).cuda(), torch.from_numpy(types).cuda())
    cu = np.concatenate([[0], np.cumsum(lens)])
    assert seqs.total == int(cu[-1]) and seqs.pairs == int((lens.astype(np.int64) ** 2).sum()) and seqs.max_len == int(lens.max())
    assert seqs.rows % K.PackedSeqs.ROW_MULTIPLE == 0 and 0 <= seqs.rows - seqs.total < (K.PackedSeqs.ROW_MULTIPLE if seqs.total < 65536 else 16384)
    assert np.array_equal(seqs.cu.cpu().numpy(), cu)
    rowmap = np.full(seqs.rows, -1, dtype=np.int64)
    inverse = np.full(n * S, -1, dtype=np.int64)
    for i in range(n):
        rowmap[cu[i]:cu[i + 1]] = i * S + np.arange(lens[i])
        inverse[i * S:i * S + lens[i]] = cu[i] + np.arange(lens[i])
    assert np.array_equal(seqs.rowmap.cpu().numpy(), rowmap)
    assert np.array_equal(seqs.inverse.cpu().numpy(), inverse)
    flat_ids, flat_types = ids.reshape(-1), types.reshape(-1)
    want_ids = np.where(rowmap >= 0, flat_ids[np.maximum(rowmap, 0)], 0)
    assert np.array_equal(seqs.ids.cpu().numpy(), want_ids)
    assert np.array_equal(seqs.types.cpu().numpy(), np.where(rowmap >= 0, flat_types[np.maximum(rowmap, 0)], 0))
    g = seqs.grouped(1)
    assert g is seqs
    if n % 5 == 0:
        g5 = seqs.grouped(5)
        assert g5.n == n // 5 and g5.max_len == 5 * int(lens.max()) and np.array_equal(g5.cu.cpu().numpy(), cu[::5])


def test_pack_unpack_first_rows_and_their_gradients():
    from emdr2_amd.model import kernels as K
    rng = np.random.default_rng(3)
    n, S, H = 37, 48, 128
    ids, lens = _ragged_ids(rng, n, S, 500)
    seqs = K.PackedSeqs(torch.from_numpy(ids).cuda())
    real = torch.from_numpy(ids != 0).cuda()
    x = torch.randn((n, S, H), device="cuda").bfloat16().requires_grad_(True)
    xp = K.pack_rows(x, seqs)
    assert xp.shape == (seqs.rows, H) and float(xp[seqs.total:].abs().max() if seqs.rows > seqs.total else 0) == 0.0
    back = K.unpack_rows(xp, seqs)
    assert torch.equal(back[real], x.detach()[real]) and float(back[~real].abs().max()) == 0.0
    first = K.first_rows(xp, seqs)
    assert torch.equal(first, x.detach()[:, 0])
    w1, w2 = torch.randn_like(back, dtype=torch.float32), torch.randn_like(first, dtype=torch.float32)
    ((back.float() * w1).sum() + (first.float() * w2).sum()).backward()
    want = w1.bfloat16().float() * real[..., None]
    want[:, 0] += w2.bfloat16().float()
    assert torch.allclose(x.grad.float(), want, atol=4e-2, rtol=2e-2)           # (two bf16 roundings of the two contributions)
    assert float(x.grad[~real].abs().max()) == 0.0


def test_packed_embedding_equals_dense_at_real_positions():
    from emdr2_amd.model import kernels as K
    rng = np.random.default_rng(4)
    n, S, H, V = 21, 64, 128, 300
    ids, _ = _ragged_ids(rng, n, S, V)
    types = rng.integers(0, 2, size=(n, S)).astype(np.int64)
    ids_t, types_t = torch.from_numpy(ids).cuda(), torch.from_numpy(types).cuda()
    real = ids_t != 0
    g = torch.Generator(device="cuda").manual_seed(1)
    mk = lambda *s: torch.nn.Parameter(torch.randn(s, generator=g, device="cuda"))
    W, P, T = mk(V, H), mk(S, H), mk(2, H)
    dense = K.embedding(ids_t, types_t, W, P, T)
    wgt = torch.randn(dense.shape, generator=g, device="cuda") * real[..., None]
    (dense.float() * wgt).sum().backward()
    gd = [p.grad.clone() for p in (W, P, T)]
    for p in (W, P, T):
        p.grad = None
    seqs = K.PackedSeqs(ids_t, types_t)
    packed = K.embedding(None, None, W, P, T, seqs=seqs)
    assert torch.equal(K.unpack_rows(packed, seqs)[real], dense[real])
    (K.unpack_rows(packed, seqs).float() * wgt).sum().backward()
    for p, ref in zip((W, P, T), gd):
        assert torch.allclose(p.grad, ref, rtol=1e-4, atol=1e-3)


# ---- attention over packed operands -------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,S,heads,causal", [(5, 64, 2, False), (9, 256, 3, False), (4, 512, 2, False), (6, 96, 2, True), (3, 32, 1, False)])
def test_packed_self_attention_equals_dense(n, S, heads, causal):
    """Same kernels, cu_q = cu_k: outputs and q/k/v gradients at the real positions agree with the dense launch (which is tested against
    fp32 torch in test_ops_gpu.py) -- lengths are arbitrary (not multiples of 32), one sequence has a single token."""
    from emdr2_amd.model import kernels as K
    rng = np.random.default_rng(n * S)
    ids, lens = _ragged_ids(rng, n, S, 500)
    ids[1, 1:] = 0; lens[1] = 1
    ids_t = torch.from_numpy(ids).cuda()
    real = ids_t != 0
    g = torch.Generator(device="cuda").manual_seed(S)
    qkv = torch.randn((n, S, 3, heads, 64), generator=g, device="cuda").bfloat16().requires_grad_(True)
    out_d = K.attention_core(qkv, None, ids_t, ids_t, causal)
    w = torch.randn(out_d.shape, generator=g, device="cuda") * real[..., None, None]
    (out_d.float() * w).sum().backward()
    grad_d = qkv.grad.clone(); qkv.grad = None
    seqs = K.PackedSeqs(ids_t)
    qkv_p = K.pack_rows(qkv.reshape(n, S, -1), seqs).reshape(seqs.rows, 3, heads, 64)
    out_p = K.attention_core(qkv_p, None, seqs, seqs, causal)
    assert out_p.shape == (seqs.rows, heads, 64)
    if seqs.rows > seqs.total:
        assert float(out_p[seqs.total:].abs().max()) == 0.0
    back = K.unpack_rows(out_p.reshape(seqs.rows, -1), seqs).reshape(n, S, heads, 64)
    assert _rel(back[real], out_d[real]) < 1e-5, _rel(back[real], out_d[real])
    (back.float() * w).sum().backward()
    assert _rel(qkv.grad[real], grad_d[real]) < 2e-3, _rel(qkv.grad[real], grad_d[real])
    assert float(qkv.grad[~real].abs().max()) == 0.0


@pytest.mark.parametrize("causal", [False, True])
def test_packed_self_attention_at_every_block_boundary(causal):
    """One packed batch whose sequence lengths sit on, one below and one above every tiling boundary of the attention kernels (32-key
    softmax steps, 64-key staged blocks, 128-query / 128-key workgroups): the ragged ends are where the one-path forward of round 4 masks
    raw scores, zeroes keys past the end after the exponential and where the backward kernels switch to their masked arms."""
    from emdr2_amd.model import kernels as K
    lens = [1, 2, 31, 32, 33, 63, 64, 65, 95, 96, 97, 127, 128, 129, 160, 191, 192, 193, 255, 256]
    n, S, heads = len(lens), 256, 2
    rng = np.random.default_rng(5)
    ids = np.zeros((n, S), dtype=np.int64)
    for i, ln in enumerate(lens):
        ids[i, :ln] = rng.integers(1, 500, ln)
    ids_t = torch.from_numpy(ids).cuda()
    real = ids_t != 0
    g = torch.Generator(device="cuda").manual_seed(9)
    qkv = torch.randn((n, S, 3, heads, 64), generator=g, device="cuda").bfloat16().requires_grad_(True)
    out_d = K.attention_core(qkv, None, ids_t, ids_t, causal)
    w = torch.randn(out_d.shape, generator=g, device="cuda") * real[..., None, None]
    (out_d.float() * w).sum().backward()
    grad_d = qkv.grad.clone(); qkv.grad = None
    seqs = K.PackedSeqs(ids_t)
    qkv_p = K.pack_rows(qkv.reshape(n, S, -1), seqs).reshape(seqs.rows, 3, heads, 64)
    out_p = K.attention_core(qkv_p, None, seqs, seqs, causal)
    back = K.unpack_rows(out_p.reshape(seqs.rows, -1), seqs).reshape(n, S, heads, 64)
    (back.float() * w).sum().backward()
    for i, ln in enumerate(lens):                                                 # per sequence: a failure names the length
        assert _rel(back[i, :ln], out_d[i, :ln]) < 1e-5, (ln, _rel(back[i, :ln], out_d[i, :ln]))
        assert _rel(qkv.grad[i, :ln], grad_d[i, :ln]) < 2e-3, (ln, _rel(qkv.grad[i, :ln], grad_d[i, :ln]))
    assert float(qkv.grad[~real].abs().max()) == 0.0
    # and the dense launch itself against fp32 torch at the same lengths (padded queries and keys inside one grid)
    from tests.test_ops_gpu import _attention_reference
    qf = qkv.detach().float()
    ref = _attention_reference(qf[:, :, 0], qf[:, :, 1], qf[:, :, 2], ids_t, ids_t, causal, None)
    assert _rel(out_d[real].float(), ref[real]) < 1e-2, _rel(out_d[real].float(), ref[real])


@pytest.mark.parametrize("B,Kk,S,L,heads", [(2, 3, 64, 32, 2), (3, 5, 96, 32, 2), (2, 50, 512, 32, 1)])
def test_cross_attention_over_packed_keys_equals_dense(B, Kk, S, L, heads):
    """FiD: dense decoder queries [B, L] against the K passages of a question concatenated -- dense keys [B, K * S] (pad rows masked) vs the
    packed buffer grouped by K (no pad rows at all)."""
    from emdr2_amd.model import kernels as K
    rng = np.random.default_rng(B * Kk + S)
    ids, _ = _ragged_ids(rng, B * Kk, S, 500, lo=S // 4)
    ids_t = torch.from_numpy(ids).cuda()
    real = ids_t != 0
    dec, _ = _ragged_ids(rng, B, L, 500, lo=2)
    dec_t = torch.from_numpy(dec).cuda()
    g = torch.Generator(device="cuda").manual_seed(S)
    q = torch.randn((B, L, heads, 64), generator=g, device="cuda").bfloat16().requires_grad_(True)
    kv = torch.randn((B * Kk, S, 2, heads, 64), generator=g, device="cuda").bfloat16().requires_grad_(True)
    out_d = K.attention_core(q, kv.reshape(B, Kk * S, 2, heads, 64), dec_t, ids_t.reshape(B, Kk * S), False)
    w = torch.randn(out_d.shape, generator=g, device="cuda") * (dec_t != 0)[..., None, None]
    (out_d.float() * w).sum().backward()
    gq, gkv = q.grad.clone(), kv.grad.clone()
    q.grad = kv.grad = None
    seqs = K.PackedSeqs(ids_t)
    kv_p = K.pack_rows(kv.reshape(B * Kk, S, -1), seqs).reshape(seqs.rows, 2, heads, 64)
    out_p = K.attention_core(q, kv_p, dec_t, seqs.grouped(Kk), False)
    dreal = dec_t != 0
    # r05: launches with few queries and thousands of keys deal their key blocks to several workgroups (emdr2_attention_fwd_splitkv), and the
    # packed / dense layouts cut their key ranges at different places: the fp32 partial sums meet in another order, a bf16 output may move by
    # one step (largest shape); smaller shapes do not split and agree to fp32 round-off
    assert _rel(out_p[dreal], out_d[dreal]) < (8e-3 if Kk * S > 4096 else 1e-5)
    # (measured r05 on the largest shape: 0.6 % of the outputs move by one bf16 step, mean relative difference 2e-5 -- 2e-3 before the softmax
    # reference points were put on whole binades, when every probability rounded differently in the two layouts)
    assert float((out_p[dreal].float() - out_d[dreal].float()).abs().mean() / out_d[dreal].float().abs().mean()) < 2e-4
    (out_p.float() * w).sum().backward()
    assert _rel(q.grad[dreal], gq[dreal]) < 2e-3
    assert _rel(kv.grad[real], gkv[real]) < 2e-3 and float(kv.grad[~real].abs().max()) == 0.0


@pytest.mark.parametrize("drop_p", [0.0, 0.1])
def test_split_key_launches_equal_the_unsplit_launch(drop_p):
    """r05: few dense queries over very many keys -- the FiD cross-attention of a group of questions, cached decoding steps -- deal the key
    blocks of a (question, head) to several workgroups and fold fp32 partials (emdr2_attention_fwd_splitkv / _bwd_splitkv).  Same outputs,
    same statistics, same gradients as ONE workgroup walking all keys, up to the order of fp32 additions: packed grouped keys with one
    question that has almost no keys (most of its splits are empty), dense keys, padded decoder positions (uniform rows), dropout."""
    from emdr2_amd.model import kernels as K
    rng = np.random.default_rng(3)
    B, Kk, S, L, heads = 3, 50, 512, 32, 2
    ids, _ = _ragged_ids(rng, B * Kk, S, 500, lo=S // 4)
    ids[Kk:2 * Kk, 1:] = 0                                               # question 1: one token per passage, 50 keys in all
    ids_t = torch.from_numpy(ids).cuda()
    dec, _ = _ragged_ids(rng, B, L, 500, lo=2)
    dec_t = torch.from_numpy(dec).cuda()
    g = torch.Generator(device="cuda").manual_seed(4)
    q = torch.randn((B, L, heads, 64), generator=g, device="cuda").bfloat16().requires_grad_(True)
    seqs = K.PackedSeqs(ids_t)
    kv_p = torch.randn((seqs.rows, 2, heads, 64), generator=g, device="cuda").bfloat16().requires_grad_(True)
    kv_d = torch.randn((B, 25600, 2, heads, 64), generator=g, device="cuda").bfloat16().requires_grad_(True)
    ids_d = torch.randint(1, 100, (B, 25600), generator=g, device="cuda")
    ids_d[0, 20000:] = 0; ids_d[2, 37:] = 0
    w = torch.randn((B, L, heads, 64), generator=g, device="cuda")

    def run(kv, ids_k, force_unsplit):
        K._SPLITKV_PLANS.clear()
        sk = ids_k.max_len if isinstance(ids_k, K.PackedSeqs) else ids_k.shape[1]
        plan = K._splitkv_plan(B, heads, L, sk)
        assert plan[0] > 1, plan                                         # this shape does split on the device
        if force_unsplit:
            K._SPLITKV_PLANS[(B, heads, L, sk)] = (1, 0, 0)
        q.grad = kv.grad = None
        out = K.attention_core(q, kv, dec_t, ids_k, False, drop_p=drop_p, seed=77)
        (out.float() * w).sum().backward()
        K._SPLITKV_PLANS.clear()
        return out.detach().clone(), q.grad.clone(), kv.grad.clone()

    for kv, ids_k in ((kv_p, seqs.grouped(Kk)), (kv_d, ids_d)):
        o1, dq1, dkv1 = run(kv, ids_k, True)
        o2, dq2, dkv2 = run(kv, ids_k, False)
        # (bf16 outputs: a different order of the fp32 additions may move a value by one bf16 step)
        assert _rel(o2, o1) < 8e-3 and bool(torch.isfinite(o2.float()).all()), _rel(o2, o1)
        assert float((o2.float() - o1.float()).abs().mean() / o1.float().abs().mean()) < 2e-4
        assert _rel(dq2, dq1) < 1e-2, _rel(dq2, dq1)
        assert _rel(dkv2, dkv1) < 1e-2, _rel(dkv2, dkv1)                 # (dk / dv see the statistics the split forward left)
        o3, dq3, dkv3 = run(kv, ids_k, False)                            # partials are folded in split order: bit-reproducible
        assert torch.equal(o3, o2) and torch.equal(dq3, dq2)


def test_split_key_launch_with_a_key_sequence_that_has_no_keys():
    """a packed key sequence of ZERO keys (only a direct caller of the C ABI can build one: cu_k[b] == cu_k[b + 1]).  The
    unsplit launch leaves that question's output / dQ rows untouched; the split launch's combine kernels sum every split's slot of an
    uninitialised workspace, so the early-return path must leave defined partials there: the other questions' results equal the unsplit
    launch's, the keyless question's output is untouched and its dQ zero, nothing is NaN, dK / dV are unaffected."""
    from emdr2_amd.model import kernels as K
    rng = np.random.default_rng(5)
    B, Kk, S, L, heads = 3, 50, 512, 32, 2
    ids, _ = _ragged_ids(rng, B * Kk, S, 500, lo=S // 4)
    ids_t = torch.from_numpy(ids).cuda()
    dec, _ = _ragged_ids(rng, B, L, 500, lo=2)
    dec_t = torch.from_numpy(dec).cuda()
    g = torch.Generator(device="cuda").manual_seed(6)
    q = torch.randn((B, L, heads, 64), generator=g, device="cuda").bfloat16().requires_grad_(True)
    seqs = K.PackedSeqs(ids_t)
    gk = seqs.grouped(Kk)
    cu = gk.cu.clone()
    cu[1] = cu[2]                                                        # question 1: no keys; question 0 takes its passages as well
    gk.cu, gk.max_len = cu.contiguous(), 2 * S * Kk
    kv = torch.randn((seqs.rows, 2, heads, 64), generator=g, device="cuda").bfloat16().requires_grad_(True)
    w = torch.randn((B, L, heads, 64), generator=g, device="cuda")
    w[1] = 0

    def run(force_unsplit):
        K._SPLITKV_PLANS.clear()
        plan = K._splitkv_plan(B, heads, L, gk.max_len)
        assert plan[0] > 1, plan
        if force_unsplit:
            K._SPLITKV_PLANS[(B, heads, L, gk.max_len)] = (1, 0, 0)
        q.grad = kv.grad = None
        # poison what torch.empty would hand out next: a path that reads uninitialised workspace sees NaNs
        junk = torch.full((64 << 20,), float("nan"), device="cuda"); del junk
        out = K.attention_core(q, kv, dec_t, gk, False, drop_p=0.0, seed=1)
        (out.float() * w).sum().backward()
        K._SPLITKV_PLANS.clear()
        return out.detach().clone(), q.grad.clone(), kv.grad.clone()

    o1, dq1, dkv1 = run(True)
    o2, dq2, dkv2 = run(False)
    for i in (0, 2):
        assert bool(torch.isfinite(o2[i].float()).all()) and _rel(o2[i], o1[i]) < 8e-3
        assert bool(torch.isfinite(dq2[i].float()).all()) and _rel(dq2[i], dq1[i]) < 1e-2
    assert float(dq2[1].float().abs().max()) == 0.0                       # zero partials, folded: no gradient for a question without keys
    assert bool(torch.isfinite(dkv2.float()).all()) and _rel(dkv2, dkv1) < 1e-2


def test_packed_attention_dropout_mask_exact_at_block_boundaries():
    """Dropout on the packed launch, element by element: the keep bit of (head n, packed row r, key k of r's sequence) is the site hash at
    (row n * rows + r, column k) -- the same generator as every other dropout site -- so a torch reference with that mask pins forward, dq
    and dk / dv (mask-free and masked arms, the ragged last steps) at lengths around every tiling boundary."""
    from emdr2_amd.model import kernels as K
    from tests.test_ops_gpu import _attention_reference, _dropout_mask
    lens = [1, 31, 32, 33, 64, 65, 96, 127, 128, 129, 200, 256]
    n, S, heads, p, seed = len(lens), 256, 2, 0.2, 4711
    rng = np.random.default_rng(6)
    ids = np.zeros((n, S), dtype=np.int64)
    for i, ln in enumerate(lens):
        ids[i, :ln] = rng.integers(1, 500, ln)
    ids_t = torch.from_numpy(ids).cuda()
    seqs = K.PackedSeqs(ids_t)
    g = torch.Generator(device="cuda").manual_seed(10)
    qkv = torch.randn((n, S, 3, heads, 64), generator=g, device="cuda").bfloat16()
    qkv_p = K.pack_rows(qkv.reshape(n, S, -1), seqs).reshape(seqs.rows, 3, heads, 64).detach().requires_grad_(True)
    out = K.attention_core(qkv_p, None, seqs, seqs, False, drop_p=p, seed=seed)
    w = torch.randn(out.shape, generator=g, device="cuda")
    w[seqs.total:] = 0
    (out.float() * w).sum().backward()
    mask = _dropout_mask((heads * seqs.rows, S), p, seed).reshape(heads, seqs.rows, S)
    cu = seqs.cu.cpu().numpy()
    ref_out, ref_grad = torch.zeros_like(out, dtype=torch.float32), torch.zeros_like(qkv_p, dtype=torch.float32)
    for i, ln in enumerate(lens):
        c0 = int(cu[i])
        x = qkv_p.detach()[c0:c0 + ln].float()[None].requires_grad_(True)             # [1, ln, 3, heads, 64]
        one = torch.ones((1, ln), dtype=torch.long, device="cuda")
        ref = _attention_reference(x[:, :, 0], x[:, :, 1], x[:, :, 2], one, one, False, mask[None, :, c0:c0 + ln, :ln])
        (ref * w[None, c0:c0 + ln]).sum().backward()
        ref_out[c0:c0 + ln], ref_grad[c0:c0 + ln] = ref[0].detach(), x.grad[0]
    # per sequence (a failure names the length), against the scale of the whole tensor: the one-token sequence's dq is exactly 0 in the
    # reference and rounding noise of D = rowsum(dO o O) here
    so = float(ref_out.abs().max())
    sg = [float(ref_grad[:, j].abs().max()) for j in range(3)]
    for i, ln in enumerate(lens):
        c0 = int(cu[i])
        e = float((out.detach()[c0:c0 + ln].float() - ref_out[c0:c0 + ln]).abs().max()) / so
        assert e < 2e-2, (ln, "out", e)
        for j, name in enumerate(("dq", "dk", "dv")):
            e = float((qkv_p.grad[c0:c0 + ln, j].float() - ref_grad[c0:c0 + ln, j]).abs().max()) / sg[j]
            assert e < 3e-2, (ln, name, e)


def test_packed_attention_dropout_statistics_and_backward_consistency():
    """With dropout the packed launch draws its own mask (keyed by the packed row), so it cannot equal the dense launch element-wise: check
    the keep rate through the output's expectation and that forward and backward use the SAME mask (gradient of a linear functional)."""
    from emdr2_amd.model import kernels as K
    rng = np.random.default_rng(11)
    n, S, heads, p = 16, 128, 2, 0.25
    ids, _ = _ragged_ids(rng, n, S, 500, lo=64)
    seqs = K.PackedSeqs(torch.from_numpy(ids).cuda())
    g = torch.Generator(device="cuda").manual_seed(2)
    qkv = torch.zeros((seqs.rows, 3, heads, 64), device="cuda")
    qkv[:, 2] = 1.0                                                               # V = 1, scores 0: every output element = (kept probability mass) / (1 - p)
    qkv = qkv.bfloat16().requires_grad_(True)
    out = K.attention_core(qkv, None, seqs, seqs, False, drop_p=p, seed=77)
    o = out[:seqs.total].float()
    assert abs(float(o.mean()) - 1.0) < 2e-2 and float(o.std()) > 1e-3
    assert torch.equal(out, K.attention_core(qkv, None, seqs, seqs, False, drop_p=p, seed=77))
    # d(sum out)/dV[key] = sum over queries of the dropped probability: its total equals sum(out) when V = 1
    out.float().sum().backward()
    dv = qkv.grad[:seqs.total, 2].float()
    assert abs(float(dv.sum()) - float(o.sum())) / float(o.sum()) < 2e-2


# ---- whole modules: packing ON vs OFF, and ON vs the oracle ---------------------------------------------------------------------
def _with_packing(flag, fn):
    from emdr2_amd.model import kernels as K
    old = K.PACKING.enabled
    K.PACKING.enabled = flag
    try:
        return fn()
    finally:
        K.PACKING.enabled = old


def _grads(m):
    out = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    for p in m.parameters():
        p.grad = None
    return out


def test_bert_tower_packed_equals_dense_and_oracle():
    from emdr2_amd.model.transformer import PretrainedBertModel
    torch.manual_seed(0)
    m = PretrainedBertModel(_cfg(), 512)
    rng = np.random.default_rng(0)
    ids, _ = _ragged_ids(rng, 48, 64, 512, lo=3)
    ids_t = torch.from_numpy(ids)
    types = torch.zeros_like(ids_t)
    w = torch.randn((48, CFG["hidden"]), generator=torch.Generator().manual_seed(3))

    def run():
        out = m(ids_t.cuda(), types.cuda())
        (out.float() * w.cuda()).sum().backward()
        return out.detach(), _grads(m)
    out_d, g_d = _with_packing(False, run)
    out_p, g_p = _with_packing(True, run)
    assert _rel(out_p, out_d) < 1e-5
    for k in g_d:
        assert _rel(g_p[k], g_d[k]) < 2e-3, (k, _rel(g_p[k], g_d[k]))
    P = {"bert." + k: v.detach().float().cpu().requires_grad_(True) for k, v in m.state_dict().items()}
    ref = to.bert_embed(P, "bert", CFG, ids_t, ~to.make_attention_mask_3d(ids_t, ids_t), types)
    assert _rel(out_p.cpu(), ref) < 2e-2


def test_reader_packed_equals_dense():
    from emdr2_amd.model.transformer import T5Model
    torch.manual_seed(0)
    m = T5Model(_cfg(), 640)
    rng = np.random.default_rng(1)
    enc, _ = _ragged_ids(rng, 32, 96, 640, lo=5)
    dec, _ = _ragged_ids(rng, 32, 32, 640, lo=2)
    enc_t, dec_t = torch.from_numpy(enc).cuda(), torch.from_numpy(dec).cuda()
    w = torch.randn((32, 32, 640), generator=torch.Generator().manual_seed(4)).cuda() * 0.1 * (dec_t != 0)[..., None]

    def run():
        logits, e = m(enc_t, dec_t)
        (logits.float() * w).sum().backward()
        return logits.detach(), e.detach(), _grads(m)
    l_d, e_d, g_d = _with_packing(False, run)
    l_p, e_p, g_p = _with_packing(True, run)
    real, dreal = enc_t != 0, dec_t != 0
    assert _rel(e_p[real], e_d[real]) < 1e-5 and float(e_p[~real].abs().max()) == 0.0
    assert _rel(l_p[dreal], l_d[dreal]) < 1e-5
    assert set(g_p) == set(g_d)
    for k in g_d:
        assert _rel(g_p[k], g_d[k]) < 2e-3, (k, _rel(g_p[k], g_d[k]))


@pytest.mark.parametrize("recompute", [False, True])
def test_emdr2_step_packed_equals_dense(recompute):
    """EMDR2Model.forward_assembled + the EMDR2 objective + backward: FiD over the grouped packed encoder output, the one-context pass and
    the context tower all packed, against the dense path on the same weights (dropout 0); also under per-layer recompute."""
    from emdr2_amd.model.emdr2_model import EMDR2Model, emdr2_loss
    rng = np.random.default_rng(7)
    B, Kk, S_ret, S, L, V = 4, 8, 32, 64, 32, 640
    torch.manual_seed(0)
    m = EMDR2Model(None, _cfg(), V, 512, Kk, S, S_ret, cls_id=2, sep_id=3, checkpoint_activations=recompute)
    m.train()
    t = lambda a: torch.from_numpy(a).cuda()
    qb = t(_ragged_ids(rng, B, S_ret, 512, lo=3)[0]); ctx = t(_ragged_ids(rng, B * Kk, S_ret, 512, lo=3)[0]).reshape(B, Kk, S_ret)
    typ = torch.zeros_like(ctx)
    qext, qone = t(_ragged_ids(rng, B * Kk, S, 600, lo=8)[0]), t(_ragged_ids(rng, B * Kk, S, 600, lo=8)[0])
    dec = t(_ragged_ids(rng, B, L, 600, lo=2)[0])
    labels = torch.roll(dec, -1, 1); labels[:, -1] = 0
    loss_mask = (labels != 0).float()

    def run():
        q_logits = m.retriever_embedder(qb, None, torch.zeros_like(qb), "query")
        lm, tlp, one = m.forward_assembled(q_logits, ctx, typ, qext, qone, dec)
        loss, stats = emdr2_loss(lm, tlp, one, labels, loss_mask, eos_id=601)
        loss.backward()
        return loss.detach(), lm.detach(), tlp.detach(), _grads(m)
    loss_d, lm_d, tlp_d, g_d = _with_packing(False, run)
    loss_p, lm_p, tlp_p, g_p = _with_packing(True, run)
    # The context tower and the one-context pass see the same 64-key blocks in both layouts (tight agreement); the FiD cross-attention does
    # not -- its key blocks now tile the real tokens instead of the K * S padded grid, so the online softmax rounds its bf16 probabilities
    # against other running maxima: agreement at the level of a bf16 ulp of the logits.
    assert abs(float(loss_p) - float(loss_d)) < 2e-3 * abs(float(loss_d)), (float(loss_p), float(loss_d))
    assert _rel(tlp_p, tlp_d) < 1e-5, _rel(tlp_p, tlp_d)
    assert _rel(lm_p[dec != 0], lm_d[dec != 0]) < 1e-2, _rel(lm_p[dec != 0], lm_d[dec != 0])
    assert set(g_p) == set(g_d)
    gmax = max(float(g.abs().max()) for g in g_d.values())
    worst = max((float((g_p[k] - g_d[k]).abs().max()) / max(float(g_d[k].abs().max()), 1e-2 * gmax), k) for k in g_d)
    assert worst[0] < 3e-2, worst


def test_twenty_five_thousand_sequences_pack_like_the_reference_loop():
    """B = 256 questions at top-k 100 is 25,600 sequences per stack (the length scan used to stop at 16,000)."""
    from emdr2_amd.model import kernels as K
    n, S = 25600, 64
    g = torch.Generator(device="cuda").manual_seed(5)
    lens = torch.randint(1, S + 1, (n,), generator=g, device="cuda")
    ids = (torch.arange(S, device="cuda")[None, :] < lens[:, None]).long() * 9
    seqs = K.PackedSeqs(ids)
    cu = torch.zeros(n + 1, dtype=torch.int64, device="cuda"); cu[1:] = torch.cumsum(lens, 0)
    assert torch.equal(seqs.cu.long(), cu) and seqs.total == int(lens.sum()) and seqs.max_len == int(lens.max())
    assert seqs.pairs == int((lens.long() ** 2).sum())
