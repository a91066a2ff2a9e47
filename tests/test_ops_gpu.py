"""GPU: transformer-block kernels (through the C ABI) against a plain PyTorch fp32 reference of the same op.
Tolerances: bf16 inputs/outputs, fp32 accumulate -> 2e-2 relative on outputs of O(1) (north_star: 2e-2 bf16)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from emdr2_amd import _native
    return _native, _native.lib()


def _gemm(A, B, bias=None, gelu=False, residual=None, alpha=1.0, out_f32=False, want_pre=False):
    nat, lib = _lib()
    M, K = A.shape[-2:]
    N = B.shape[-2]
    C = torch.empty(A.shape[:-2] + (M, N), dtype=torch.float32 if out_f32 else torch.bfloat16, device="cuda")
    pre = torch.empty_like(C, dtype=torch.bfloat16) if want_pre else None
    batch = int(np.prod(A.shape[:-2])) if A.dim() > 2 else 1
    nat.check(lib.emdr2_gemm_nt_bf16(A.data_ptr(), K, B.data_ptr(), K, C.data_ptr(), N, M, N, K, batch, M * K, (N * K if B.dim() > 2 else 0), M * N,
                                     1, 0, 0, 0, alpha, bias.data_ptr() if bias is not None else None, int(gelu),
                                     pre.data_ptr() if pre is not None else None, residual.data_ptr() if residual is not None else None, 0,
                                     int(out_f32), 1, 0.0, 0, nat.stream_ptr()), "gemm")
    torch.cuda.synchronize()
    return (C, pre) if want_pre else C


@pytest.mark.parametrize("M,N,K", [(256, 256, 32), (512, 768, 768), (300, 2304, 768), (1000, 128, 64), (64, 3072, 768), (777, 40, 96), (4096, 3072, 768)])
def test_gemm_nt_plain(M, N, K):
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn((M, K), generator=g, device="cuda").bfloat16()
    B = torch.randn((N, K), generator=g, device="cuda").bfloat16()
    ref = A.float() @ B.float().T
    C = _gemm(A, B, out_f32=True)
    assert torch.allclose(C, ref, rtol=1e-3, atol=1e-2 * (K ** 0.5) * 0.01 + 1e-3)      # fp32 accumulate of exact bf16 products
    Cb = _gemm(A, B)
    assert torch.allclose(Cb.float(), ref, rtol=2e-2, atol=2e-2 * (K ** 0.5))


def test_gemm_asymmetric_identity_catches_transposes():
    A = torch.eye(256, device="cuda").bfloat16()
    B = (torch.arange(256 * 256, device="cuda").reshape(256, 256) % 251).float().bfloat16()
    C = _gemm(A, B, out_f32=True)
    assert torch.equal(C, B.float().T)


def test_gemm_epilogues_bias_gelu_residual_alpha():
    g = torch.Generator(device="cuda").manual_seed(5)
    M, N, K = 384, 512, 256
    A = torch.randn((M, K), generator=g, device="cuda").bfloat16() * 0.5
    B = torch.randn((N, K), generator=g, device="cuda").bfloat16() * 0.1
    bias = torch.randn(N, generator=g, device="cuda")
    R = torch.randn((M, N), generator=g, device="cuda").bfloat16()
    pre_ref = 0.125 * (A.float() @ B.float().T) + bias
    ref = torch.nn.functional.gelu(pre_ref) + R.float()
    C, pre = _gemm(A, B, bias=bias, gelu=True, residual=R, alpha=0.125, want_pre=True)
    assert torch.allclose(pre.float(), pre_ref, rtol=2e-2, atol=2e-2)
    assert torch.allclose(C.float(), ref, rtol=2e-2, atol=3e-2)


def test_gemm_batched_two_levels_strided():
    """[b, np] batches addressed through strides inside one [b, s, 3, np, hn] QKV buffer (attention QK^T)."""
    nat, lib = _lib()
    g = torch.Generator(device="cuda").manual_seed(11)
    b, s, heads, hn = 3, 128, 4, 64
    qkv = torch.randn((b, s, 3, heads, hn), generator=g, device="cuda").bfloat16()
    scores = torch.empty((b, heads, s, s), dtype=torch.bfloat16, device="cuda")
    q, k = qkv[:, :, 0], qkv[:, :, 1]
    ld = 3 * heads * hn
    nat.check(lib.emdr2_gemm_nt_bf16(q.data_ptr(), ld, k.data_ptr(), ld, scores.data_ptr(), s, s, s, hn, b, s * ld, s * ld, heads * s * s,
                                     heads, hn, hn, s * s, 0.125, None, 0, None, None, 0, 0, 1, 0.0, 0, nat.stream_ptr()), "gemm")
    torch.cuda.synchronize()
    ref = torch.einsum("bqnd,bknd->bnqk", q.float(), k.float()) * 0.125
    assert torch.allclose(scores.float(), ref, rtol=2e-2, atol=5e-2)


def test_transpose_and_colsum():
    nat, lib = _lib()
    g = torch.Generator(device="cuda").manual_seed(2)
    x = torch.randn((3, 2, 200, 96), generator=g, device="cuda").bfloat16()
    out = torch.empty((3, 2, 96, 200), dtype=torch.bfloat16, device="cuda")
    cs = torch.zeros(96, device="cuda")
    nat.check(lib.emdr2_transpose_bf16(x.data_ptr(), 96, out.data_ptr(), 200, 200, 96, 3, 2 * 200 * 96, 2 * 96 * 200, 2, 200 * 96, 96 * 200,
                                       cs.data_ptr(), nat.stream_ptr()), "transpose")
    torch.cuda.synchronize()
    assert torch.equal(out, x.transpose(-1, -2).contiguous())
    assert torch.allclose(cs, x.float().sum((0, 1, 2)), rtol=1e-4, atol=1e-3)


def test_weight_gradient_gemm_split_k():
    from emdr2_amd.model import kernels as K
    g = torch.Generator(device="cuda").manual_seed(21)
    M, N, Kd = 65536, 768, 512
    dyT = torch.randn((N, M), generator=g, device="cuda").bfloat16()
    xT = torch.randn((Kd, M), generator=g, device="cuda").bfloat16()
    dW = K.weight_grad_nt(dyT, xT)
    torch.cuda.synchronize()
    ref = dyT.float() @ xT.float().T
    assert torch.allclose(dW, ref, rtol=2e-3, atol=0.5)


# ---- fused attention + dropout ------------------------------------------------------------------------------------------
def _dropout_mask(shape, p, seed):
    """The multiplicative mask (0 or 1/(1-p)) of a dropout site, through the same kernel the backward uses."""
    nat, lib = _lib()
    ones = torch.ones(shape, dtype=torch.bfloat16, device="cuda")
    out = torch.empty_like(ones)
    nat.check(lib.emdr2_dropout(ones.data_ptr(), out.data_ptr(), ones.numel(), shape[-1], p, seed, nat.stream_ptr()), "dropout")
    torch.cuda.synchronize()
    return out.float()


def _attention_reference(q, k, v, ids_q, ids_k, causal, mask=None):
    """fp32 torch restatement with the reference's semantics (transformer.py:283-381): masked scores replaced by -10000."""
    hn = q.shape[-1]
    s = torch.einsum("bqnd,bknd->bnqk", q, k) / hn ** 0.5
    m = (ids_q[:, None, :, None] == 0) | (ids_k[:, None, None, :] == 0)
    if causal:
        sq, sk = q.shape[1], k.shape[1]
        m = m | (torch.arange(sk, device=q.device)[None, None, None, :] > torch.arange(sq, device=q.device)[None, None, :, None])
    p = torch.softmax(s.masked_fill(m, -10000.0), dim=-1)
    if mask is not None:
        p = p * mask
    return torch.einsum("bnqk,bknd->bqnd", p, v)


def _attention_case(b, heads, sq, sk, causal, drop_p, seed, gen_seed):
    from emdr2_amd.model import kernels as K
    g = torch.Generator(device="cuda").manual_seed(gen_seed)
    hn = 64
    packed = sq == sk                                                   # self-attention: one packed [b, s, 3, np, hn] QKV tensor
    if packed:
        qkv = torch.randn((b, sq, 3, heads, hn), generator=g, device="cuda").bfloat16().requires_grad_(True)
    else:
        q = torch.randn((b, sq, heads, hn), generator=g, device="cuda").bfloat16().requires_grad_(True)
        kv = torch.randn((b, sk, 2, heads, hn), generator=g, device="cuda").bfloat16().requires_grad_(True)
    ids_q = torch.randint(1, 100, (b, sq), generator=g, device="cuda")
    ids_k = ids_q if sq == sk else torch.randint(1, 100, (b, sk), generator=g, device="cuda")
    for i in range(b):                                                  # ragged padding; row b-1 of ids_q fully padded exercises uniform rows
        ids_k[i, sk - 7 * i - 3:] = 0
        if sq != sk:
            ids_q[i, sq - i - 1:] = 0
    out = K.attention_core(qkv, None, ids_q, ids_k, causal, drop_p=drop_p, seed=seed) if packed else \
        K.attention_core(q, kv, ids_q, ids_k, causal, drop_p=drop_p, seed=seed)
    w = torch.randn(out.shape, generator=g, device="cuda")
    (out.float() * w).sum().backward()
    mask = _dropout_mask((b, heads, sq, sk), drop_p, seed) if drop_p > 0 else None
    if packed:
        qkvf = qkv.detach().float().requires_grad_(True)
        ref = _attention_reference(qkvf[:, :, 0], qkvf[:, :, 1], qkvf[:, :, 2], ids_q, ids_k, causal, mask)
    else:
        qf = q.detach().float().requires_grad_(True)
        kvf = kv.detach().float().requires_grad_(True)
        ref = _attention_reference(qf, kvf[:, :, 0], kvf[:, :, 1], ids_q, ids_k, causal, mask)
    (ref * w).sum().backward()
    rel = lambda a, r: float((a.detach().float() - r.detach()).abs().max() / (r.detach().abs().max() + 1e-6))
    assert rel(out, ref) < 2e-2, ("out", rel(out, ref))
    if packed:
        for i, name in enumerate(("dq", "dk", "dv")):
            assert rel(qkv.grad[:, :, i], qkvf.grad[:, :, i]) < 3e-2, (name, rel(qkv.grad[:, :, i], qkvf.grad[:, :, i]))
    else:
        assert rel(q.grad, qf.grad) < 3e-2, ("dq", rel(q.grad, qf.grad))
        assert rel(kv.grad, kvf.grad) < 3e-2, ("dkv", rel(kv.grad, kvf.grad))


@pytest.mark.parametrize("b,heads,sq,sk,causal", [(3, 4, 64, 64, False), (2, 3, 512, 512, False), (2, 2, 256, 256, True), (3, 2, 32, 1024, False),
                                                  (2, 2, 288, 320, False), (2, 2, 32, 32, True), (2, 2, 96, 96, False), (2, 2, 32, 25600, False),
                                                  (5, 3, 32, 32, False), (2, 2, 96, 96, True), (3, 2, 40, 160, False), (2, 12, 32, 32, True)])
def test_attention_core_forward_backward_vs_torch(b, heads, sq, sk, causal):
    """Fused kernel (sk % 32 == 0: the decoder's 32 x 32 causal self-attention included) and the GEMM + softmax composition (the rest) against
    fp32 torch, forward and q/k/v gradients."""
    _attention_case(b, heads, sq, sk, causal, 0.0, 0, 100 + sq + sk)


@pytest.mark.parametrize("sq,sk", [(128, 128), (32, 640), (96, 96), (32, 32)])
def test_attention_dropout_mask_is_the_same_in_forward_backward_and_both_paths(sq, sk):
    _attention_case(2, 3, sq, sk, sq == sk and sq == 32, 0.1, 0xC0FFEE + sq, 7 + sq)


def test_dropout_statistics_and_determinism():
    p = 0.1
    m1 = _dropout_mask((4096, 768), p, 17)
    keep = float((m1 != 0).float().mean())
    n = m1.numel()
    assert abs(keep - (1 - p)) < 5 * (p * (1 - p) / n) ** 0.5
    vals = torch.unique(m1)
    assert vals.numel() == 2 and float(vals[0]) == 0.0 and abs(float(vals[1]) - 1 / (1 - p)) < 1e-2
    assert torch.equal(m1, _dropout_mask((4096, 768), p, 17))
    m2 = _dropout_mask((4096, 768), p, 18)
    agree = float(((m1 != 0) == (m2 != 0)).float().mean())
    assert abs(agree - ((1 - p) ** 2 + p ** 2)) < 5e-3                  # independent streams
    # no structure along rows / columns
    assert float((m1 != 0).float().mean(0).std()) < 3 * (p * (1 - p) / 4096) ** 0.5 * 1.5
    assert float((m1 != 0).float().mean(1).std()) < 3 * (p * (1 - p) / 768) ** 0.5 * 1.5


def test_linear_bias_dropout_add_forward_and_backward():
    from emdr2_amd.model import kernels as K
    g = torch.Generator(device="cuda").manual_seed(9)
    M, N, Kd, p, seed = 512, 768, 256, 0.1, 4242
    x = (torch.randn((M, Kd), generator=g, device="cuda") * 0.5).bfloat16().requires_grad_(True)
    res = torch.randn((M, N), generator=g, device="cuda").bfloat16().requires_grad_(True)
    W = torch.nn.Parameter(torch.randn((N, Kd), generator=g, device="cuda") * 0.1)
    bias = torch.nn.Parameter(torch.randn(N, generator=g, device="cuda"))
    y = K.linear(x, W, bias, residual=res, drop_p=p, seed=seed)
    w = torch.randn(y.shape, generator=g, device="cuda")
    (y.float() * w).sum().backward()
    mask = _dropout_mask((M, N), p, seed)
    xf, rf = x.detach().float().requires_grad_(True), res.detach().float().requires_grad_(True)
    Wf, bf = W.detach().bfloat16().float().requires_grad_(True), bias.detach().clone().requires_grad_(True)
    ref = (xf @ Wf.T + bf) * mask + rf
    (ref * w).sum().backward()
    rel = lambda a, r: float((a.detach().float() - r.detach()).abs().max() / (r.detach().abs().max() + 1e-6))
    assert torch.equal((y.float() - res.float() == 0), (ref - rf == 0)) or rel(y, ref) < 2e-2
    assert rel(y, ref) < 2e-2
    assert rel(x.grad, xf.grad) < 3e-2 and rel(res.grad, rf.grad) < 1e-2
    assert rel(W.grad, Wf.grad) < 3e-2 and rel(bias.grad, bf.grad) < 3e-2


def test_embedding_dropout_forward_and_backward():
    from emdr2_amd.model import kernels as K
    g = torch.Generator(device="cuda").manual_seed(10)
    b, s, H, V, p, seed = 4, 64, 128, 300, 0.1, 99
    Wt = torch.nn.Parameter(torch.randn((V, H), generator=g, device="cuda"))
    Pt = torch.nn.Parameter(torch.randn((s, H), generator=g, device="cuda"))
    ids = torch.randint(0, V, (b, s), generator=g, device="cuda")
    out = K.embedding(ids, None, Wt, Pt, None, drop_p=p, seed=seed)
    w = torch.randn(out.shape, generator=g, device="cuda")
    (out.float() * w).sum().backward()
    mask = _dropout_mask((b, s, H), p, seed)
    Wf, Pf = Wt.detach().bfloat16().float().requires_grad_(True), Pt.detach().bfloat16().float().requires_grad_(True)
    ref = (Wf[ids] + Pf[None, :s]) * mask
    (ref * w).sum().backward()
    rel = lambda a, r: float((a.detach().float() - r.detach()).abs().max() / (r.detach().abs().max() + 1e-6))
    assert rel(out, ref) < 1e-2
    assert rel(Wt.grad, Wf.grad) < 2e-2 and rel(Pt.grad, Pf.grad) < 2e-2


@pytest.mark.parametrize("M,N,Kd", [(64, 256, 256), (128, 512, 264), (384, 768, 768), (4096, 768, 768), (65536, 768, 512), (2048, 2304, 768), (1024, 40, 72), (8192, 3072, 768),
                                    (36864, 520, 1032)])
def test_weight_gradient_gemm_tn_with_bias_gradient(M, N, Kd):
    """dW = dy^T x through the LDS transpose reads (no HBM transposes), asymmetric random operands (a transposed or permuted fragment
    cannot pass), plus the fused column sums of dy."""
    from emdr2_amd.model import kernels as K
    g = torch.Generator(device="cuda").manual_seed(M + N + Kd)
    dy = torch.randn((M, N), generator=g, device="cuda").bfloat16()
    x = torch.randn((M, Kd), generator=g, device="cuda").bfloat16()
    db = torch.zeros(N, device="cuda")
    dW = K.weight_grad_tn(dy, x, colsum=db)
    torch.cuda.synchronize()
    ref = dy.float().T @ x.float()
    assert torch.allclose(dW, ref, rtol=2e-3, atol=2e-3 * (M ** 0.5))
    assert torch.allclose(db, dy.float().sum(0), rtol=1e-3, atol=1e-3 * (M ** 0.5))


def test_weight_gradient_gemm_tn_strided_operands_and_single_slice():
    """The persistent TN GEMM (csrc/gemm8t.hip) on column slices of wider activation matrices (lda, ldb > I, J: the packed QKV gradient is
    consumed like this), written into a column slice of a wider fp32 output, with one reduction slice (plain stores) and with many (atomics)."""
    from emdr2_amd import _native
    lib = _native.lib()
    g = torch.Generator(device="cuda").manual_seed(11)
    M, NF, KF, I, J = 6144, 1024, 1280, 520, 776
    dy_full = torch.randn((M, NF), generator=g, device="cuda").bfloat16()
    x_full = torch.randn((M, KF), generator=g, device="cuda").bfloat16()
    dy, x = dy_full[:, 256:256 + I], x_full[:, 8:8 + J]
    ref = dy.float().T @ x.float()
    for split in (1, 7):
        out = torch.full((I, 1024), 7.0, device="cuda")
        c = out[:, 128:128 + J]
        if split > 1:
            c.zero_()
        colsum = torch.zeros(I, device="cuda")
        rc = lib.emdr2_gemm_tn_bf16(dy.data_ptr(), NF, x.data_ptr(), KF, c.data_ptr(), 1024, I, J, M, split, colsum.data_ptr(), None)
        assert rc == 0
        torch.cuda.synchronize()
        assert torch.allclose(c, ref, rtol=2e-3, atol=2e-3 * (M ** 0.5))
        assert torch.allclose(colsum, dy.float().sum(0), rtol=1e-3, atol=1e-3 * (M ** 0.5))
        assert float(out[:, :128].min()) == 7.0 and float(out[:, 128 + J:].min()) == 7.0 and float(out[:, :128].max()) == 7.0      # nothing outside the slice


def test_gelu_epilogue_accuracy_vs_exact_erf():
    """The epilogue's erf (Abramowitz-Stegun 7.1.26, |err| <= 1.5e-7) against torch's exact-erf GELU in fp32: identity GEMM passes x through."""
    x = torch.linspace(-12.0, 12.0, 256 * 256, device="cuda").reshape(256, 256).bfloat16()
    eye = torch.eye(256, device="cuda").bfloat16()
    y = _gemm(eye, x.T.contiguous(), gelu=True, out_f32=True)            # C = gelu(I x) in fp32
    ref = torch.nn.functional.gelu(x.float())
    err = (y - ref).abs()
    assert float(err.max()) < 2e-6 and float((err / (ref.abs() + 1e-3)).max()) < 2e-4


@pytest.mark.parametrize("rows,H", [(1000, 768), (7, 768), (4096, 128), (513, 1024), (33, 3072)])
def test_layernorm_forward_backward_vs_torch(rows, H):
    """Generic kernel and the H = 768 half-wave-per-row fast path, odd row counts included."""
    from emdr2_amd.model import kernels as K
    g = torch.Generator(device="cuda").manual_seed(rows + H)
    x = (torch.randn((rows, H), generator=g, device="cuda") * 2 + 0.5).bfloat16().requires_grad_(True)
    gamma = torch.nn.Parameter(torch.randn(H, generator=g, device="cuda"))
    beta = torch.nn.Parameter(torch.randn(H, generator=g, device="cuda"))
    y = K.layer_norm(x, gamma, beta, 1e-5)
    w = torch.randn(y.shape, generator=g, device="cuda")
    (y.float() * w).sum().backward()
    xf = x.detach().float().requires_grad_(True)
    gf, bf = gamma.detach().clone().requires_grad_(True), beta.detach().clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xf, (H,), gf, bf, 1e-5)
    (ref * w).sum().backward()
    rel = lambda a, r: float((a.detach().float() - r.detach()).abs().max() / (r.detach().abs().max() + 1e-6))
    assert rel(y, ref) < 1e-2
    assert rel(x.grad, xf.grad) < 2e-2 and rel(gamma.grad, gf.grad) < 2e-2 and rel(beta.grad, bf.grad) < 2e-2


def test_fused_mlp_forward_backward_vs_torch():
    """gelu(x W1^T + b1) W2^T + b2 with dropout and residual as one node; gelu' applied in the epilogue of the dy W2 GEMM."""
    from emdr2_amd.model import kernels as K
    g = torch.Generator(device="cuda").manual_seed(13)
    M, H, F, p, seed = 512, 256, 1024, 0.1, 777
    x = (torch.randn((M, H), generator=g, device="cuda")).bfloat16().requires_grad_(True)
    res = torch.randn((M, H), generator=g, device="cuda").bfloat16().requires_grad_(True)
    W1 = torch.nn.Parameter(torch.randn((F, H), generator=g, device="cuda") * 0.08); b1 = torch.nn.Parameter(torch.randn(F, generator=g, device="cuda") * 0.5)
    W2 = torch.nn.Parameter(torch.randn((H, F), generator=g, device="cuda") * 0.05); b2 = torch.nn.Parameter(torch.randn(H, generator=g, device="cuda") * 0.5)
    y = K.mlp(x, W1, b1, W2, b2, res, drop_p=p, seed=seed)
    w = torch.randn(y.shape, generator=g, device="cuda")
    (y.float() * w).sum().backward()
    mask = _dropout_mask((M, H), p, seed)
    f = lambda t: t.detach().float().requires_grad_(True)
    xf, rf = f(x), f(res)
    W1f, W2f = W1.detach().bfloat16().float().requires_grad_(True), W2.detach().bfloat16().float().requires_grad_(True)
    b1f, b2f = f(b1), f(b2)
    inter = torch.nn.functional.gelu(xf @ W1f.T + b1f)
    ref = (inter @ W2f.T + b2f) * mask + rf
    (ref * w).sum().backward()
    rel = lambda a, r: float((a.detach().float() - r.detach()).abs().max() / (r.detach().abs().max() + 1e-6))
    assert rel(y, ref) < 2e-2
    for name, a, r in (("dx", x.grad, xf.grad), ("dres", res.grad, rf.grad), ("dW1", W1.grad, W1f.grad), ("db1", b1.grad, b1f.grad),
                       ("dW2", W2.grad, W2f.grad), ("db2", b2.grad, b2f.grad)):
        assert rel(a, r) < 3e-2, (name, rel(a, r))


def test_gemm_random_shapes_and_epilogues_property():
    """Seeded sweep: ragged M / N (tile tails, XCD-remapped 1-D grids included), K multiples of 32, random epilogue combinations, NT and TN."""
    from emdr2_amd.model import kernels as K
    rng = np.random.default_rng(7)
    g = torch.Generator(device="cuda").manual_seed(7)
    for case in range(20):
        M = int(rng.integers(1, 5000)); N = int(rng.integers(1, 200)) * 8; Kd = int(rng.integers(1, 24)) * 32
        a = (torch.randn((M, Kd), generator=g, device="cuda") * 0.5).bfloat16()
        b = (torch.randn((N, Kd), generator=g, device="cuda") * 0.2).bfloat16()
        use_bias, use_gelu, use_res = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        alpha = float(rng.choice([1.0, 0.125]))
        bias = torch.randn(N, generator=g, device="cuda") if use_bias else None
        res = torch.randn((M, N), generator=g, device="cuda").bfloat16() if use_res else None
        out = K.matmul_nt(a, b, alpha=alpha, bias=bias, gelu=use_gelu, residual=res)
        ref = alpha * (a.float() @ b.float().T)
        if use_bias:
            ref = ref + bias
        if use_gelu:
            ref = torch.nn.functional.gelu(ref)
        if use_res:
            ref = ref + res.float()
        err = float((out.float() - ref).abs().max() / (ref.abs().max() + 1e-6))
        assert err < 2e-2, (case, M, N, Kd, use_bias, use_gelu, use_res, err)
    for case in range(8):
        R = int(rng.integers(1, 300)) * 32; I = int(rng.integers(1, 120)) * 8; J = int(rng.integers(1, 120)) * 8
        dy = torch.randn((R, I), generator=g, device="cuda").bfloat16(); x = torch.randn((R, J), generator=g, device="cuda").bfloat16()
        cs = torch.zeros(I, device="cuda")
        dw = K.weight_grad_tn(dy, x, colsum=cs)
        ref = dy.float().T @ x.float()
        assert torch.allclose(dw, ref, rtol=2e-3, atol=2e-3 * R ** 0.5), (case, R, I, J)
        assert torch.allclose(cs, dy.float().sum(0), rtol=1e-3, atol=1e-3 * R ** 0.5)


def test_attention_random_shapes_property():
    """Seeded sweep over (batch, heads, sq, sk, causal, dropout): fused and composed paths, forward and all input gradients, vs fp32 torch."""
    rng = np.random.default_rng(11)
    for case in range(12):
        b, heads = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        if rng.integers(0, 2):
            sq = sk = int(rng.integers(1, 9)) * 32
        else:
            sq, sk = int(rng.integers(1, 5)) * 32, int(rng.integers(1, 12)) * 64
        causal = bool(sq == sk and rng.integers(0, 2))
        drop = float(rng.choice([0.0, 0.1]))
        _attention_case(b, heads, sq, sk, causal, drop, 1000 + case, 50 + case)


def test_embedding_backward_with_token_types_and_positions():
    """Word rows by atomic scatter, position rows by owner threads, type rows by register partials: all three against torch."""
    from emdr2_amd.model import kernels as K
    g = torch.Generator(device="cuda").manual_seed(21)
    b, s, H, V = 37, 48, 96, 200
    Wt = torch.nn.Parameter(torch.randn((V, H), generator=g, device="cuda")); Pt = torch.nn.Parameter(torch.randn((64, H), generator=g, device="cuda"))
    Tt = torch.nn.Parameter(torch.randn((2, H), generator=g, device="cuda"))
    ids = torch.randint(0, V, (b, s), generator=g, device="cuda"); types = torch.randint(0, 2, (b, s), generator=g, device="cuda")
    out = K.embedding(ids, types, Wt, Pt, Tt)
    w = torch.randn(out.shape, generator=g, device="cuda")
    (out.float() * w).sum().backward()
    f = lambda p: p.detach().bfloat16().float().requires_grad_(True)
    Wf, Pf, Tf = f(Wt), f(Pt), f(Tt)
    ref = Wf[ids] + Pf[None, :s] + Tf[types]
    (ref * w.bfloat16().float()).sum().backward()
    rel = lambda a, r: float((a.detach().float() - r.detach()).abs().max() / (r.detach().abs().max() + 1e-6))
    assert rel(out, ref) < 1e-2
    assert rel(Wt.grad, Wf.grad) < 1e-2 and rel(Pt.grad, Pf.grad) < 1e-2 and rel(Tt.grad, Tf.grad) < 1e-2
    assert float(Pt.grad[s:].abs().max()) == 0.0                          # rows beyond the sequence length untouched


@pytest.mark.parametrize("B,Kk,H,scaled", [(64, 50, 768, True), (8, 101, 768, True), (3, 7, 128, False), (5, 128, 64, True), (4, 129, 64, True), (2, 1000, 128, True)])
def test_retriever_prior_kernel_value_and_gradients(B, Kk, H, scaled):
    """a9 (emdr2_model.py:134-145): log_softmax_k(<q, c_k> / sqrt(H)) and its gradients against torch fp32 autograd of the same formula."""
    from emdr2_amd.model import kernels as K
    g = torch.Generator(device="cuda").manual_seed(B * Kk + H)
    q = torch.randn((B, H), generator=g, device="cuda").bfloat16().requires_grad_(True)
    c = torch.randn((B, Kk, H), generator=g, device="cuda").bfloat16().requires_grad_(True)
    w = torch.randn((B, Kk), generator=g, device="cuda")
    scale = 1.0 / H ** 0.5 if scaled else 1.0
    out = K.retriever_prior(q, c, scale)
    (out * w).sum().backward()
    qr, cr = q.detach().float().requires_grad_(True), c.detach().float().requires_grad_(True)
    ref = torch.log_softmax(torch.bmm(qr[:, None, :], cr.transpose(1, 2))[:, 0] * scale, dim=1)
    (ref * w).sum().backward()
    assert out.dtype == torch.float32 and torch.allclose(out, ref.detach(), rtol=1e-4, atol=1e-4)
    assert torch.allclose(q.grad.float(), qr.grad, rtol=2e-2, atol=2e-2 * float(qr.grad.abs().max()))
    assert torch.allclose(c.grad.float(), cr.grad, rtol=2e-2, atol=2e-2 * float(cr.grad.abs().max()))


@pytest.mark.parametrize("B,Kk,L", [(64, 50, 32), (4, 101, 32), (2, 3, 8)])
def test_marginal_logsumexp_kernel_value_and_gradient(B, Kk, L):
    """a14 tail (train_e2eqa.py:98-123): logsumexp_k(prior + gold) and d/d prior against torch."""
    from emdr2_amd.model import kernels as K
    g = torch.Generator(device="cuda").manual_seed(B + Kk + L)
    prior = torch.log_softmax(torch.randn((B, Kk), generator=g, device="cuda"), dim=1).requires_grad_(True)
    gold = -torch.rand((B, Kk, L), generator=g, device="cuda") * 12
    w = torch.randn((B, L), generator=g, device="cuda")
    out = K.marginal_logsumexp(prior, gold)
    (out * w).sum().backward()
    pr = prior.detach().clone().requires_grad_(True)
    ref = torch.logsumexp(pr.unsqueeze(-1) + gold, dim=1)
    (ref * w).sum().backward()
    assert torch.allclose(out, ref.detach(), rtol=1e-5, atol=1e-5)
    assert torch.allclose(prior.grad, pr.grad, rtol=1e-4, atol=1e-5)
