"""GPU: the persistent 256 x 256 x 64 GEMM (csrc/gemm8.hip) behind emdr2_gemm_nt_bf16 at the shapes it takes (M, N multiples of 256,
K multiple of 128, M >= 4096), every epilogue recipe the step uses, against a plain PyTorch fp32 reference of the same op.
Tolerances as in test_ops_gpu.py (bf16 outputs, fp32 accumulation: 2e-2).  Also: tile seams (several tiles per workgroup), the grouped
n-tile order (N = 3072, K = 768), run-to-run bit stability (a race in the LDS ring or the hand-over between the two wave halves would
show as flicker), and identity / asymmetric operands that catch transposes and mis-placed fragments."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _gemm(A, B, bias=None, gelu=False, residual=None, rmode=0, alpha=1.0, want_pre=False, drop_p=0.0, seed=0):
    from emdr2_amd import _native as nat
    lib = nat.lib()
    M, K = A.shape
    N = B.shape[0]
    C = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    pre = torch.empty_like(C) if want_pre else None
    nat.check(lib.emdr2_gemm_nt_bf16(A.data_ptr(), K, B.data_ptr(), K, C.data_ptr(), N, M, N, K, 1, 0, 0, 0, 1, 0, 0, 0, alpha,
                                     bias.data_ptr() if bias is not None else None, int(gelu), pre.data_ptr() if pre is not None else None,
                                     residual.data_ptr() if residual is not None else None, rmode, 0, 1, float(drop_p), int(seed),
                                     nat.stream_ptr()), "gemm")
    torch.cuda.synchronize()
    return (C, pre) if want_pre else C


def _rand(shape, gen, scale=1.0):
    return (torch.randn(shape, generator=gen, device="cuda") * scale).bfloat16()


@pytest.mark.parametrize("M,N,K", [(4096, 256, 128), (4096, 768, 768), (8192, 2304, 768), (4096, 768, 3072), (65536, 768, 768),
                                   (32768, 3072, 768), (4096, 30720, 768)])
def test_gemm8_plain(M, N, K):
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A, B = _rand((M, K), g), _rand((N, K), g)
    C = _gemm(A, B)
    for lo in range(0, M, 16384):                                   # reference in row blocks (fp32 [M, N] of the largest case is 2 GB otherwise)
        ref = A[lo:lo + 16384].float() @ B.float().T
        assert torch.allclose(C[lo:lo + 16384].float(), ref, rtol=2e-2, atol=2e-2 * (K ** 0.5)), (M, N, K, lo)


def test_gemm8_identity_and_asymmetric_operands():
    """A = [I | 0] picks rows of B^T: any transposed fragment, swapped half-tile or mis-swizzled 16-byte group shows as a wrong value."""
    M, N, K = 4096, 512, 256
    A = torch.zeros((M, K), device="cuda")
    A[torch.arange(M), torch.arange(M) % K] = 1.0
    B = ((torch.arange(N * K, device="cuda").reshape(N, K) * 7 + 3) % 251).float()
    C = _gemm(A.bfloat16(), B.bfloat16())
    ref = B.bfloat16().float().T[torch.arange(M) % K]
    assert torch.equal(C.float(), ref)


def test_gemm8_every_output_lands_in_its_own_place():
    """C[m, n] = m_code + n_code with distinct codes: catches row / column permutations of the epilogue's LDS round trip."""
    M, N, K = 4096, 768, 128
    A = torch.zeros((M, K), device="cuda"); B = torch.zeros((N, K), device="cuda")
    A[:, 0] = (torch.arange(M, device="cuda") % 128).float(); A[:, 1] = 1.0
    B[:, 0] = 1.0; B[:, 1] = (torch.arange(N, device="cuda") % 64).float() * 0.5
    C = _gemm(A.bfloat16(), B.bfloat16())
    ref = A[:, 0:1] + B[:, 1][None, :]
    assert torch.equal(C.float(), ref.bfloat16().float())


def test_gemm8_epilogues():
    g = torch.Generator(device="cuda").manual_seed(7)
    M, N, K = 8192, 768, 256
    A, B = _rand((M, K), g, 0.5), _rand((N, K), g, 0.1)
    bias = torch.randn(N, generator=g, device="cuda")
    R = _rand((M, N), g)
    acc = A.float() @ B.float().T
    # bias
    assert torch.allclose(_gemm(A, B, bias=bias).float(), acc + bias, rtol=2e-2, atol=2e-2)
    # bias + GELU, with and without the pre-activation output
    pre_ref = acc + bias
    C = _gemm(A, B, bias=bias, gelu=True)
    assert torch.allclose(C.float(), torch.nn.functional.gelu(pre_ref), rtol=2e-2, atol=2e-2)
    C2, pre = _gemm(A, B, bias=bias, gelu=True, want_pre=True)
    assert torch.equal(C2, C)
    assert torch.allclose(pre.float(), pre_ref, rtol=2e-2, atol=2e-2)
    # bias + residual
    assert torch.allclose(_gemm(A, B, bias=bias, residual=R).float(), pre_ref + R.float(), rtol=2e-2, atol=3e-2)
    # fused GELU backward: (A B^T) * gelu'(R)
    x = R.float()
    gelu_grad = 0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2 * torch.pi) ** 0.5
    assert torch.allclose(_gemm(A, B, residual=R, rmode=1).float(), acc * gelu_grad, rtol=2e-2, atol=3e-2)
    # the pair the fused MLP runs: gelu = 2 writes gelu'(pre-activation) next to the activation, residual_mode = 2 multiplies it back in
    C3, der = _gemm(A, B, bias=bias, gelu=2, want_pre=True)
    assert torch.equal(C3, C)
    der_ref = 0.5 * (1 + torch.erf(pre_ref / 2 ** 0.5)) + pre_ref * torch.exp(-0.5 * pre_ref * pre_ref) / (2 * torch.pi) ** 0.5
    assert torch.allclose(der.float(), der_ref, rtol=1e-2, atol=6e-3)
    assert torch.allclose(_gemm(A, B, residual=R, rmode=2).float(), acc * R.float(), rtol=2e-2, atol=3e-2)
    # ... on the general kernel too (a shape the persistent one does not take)
    As, Rs = A[:1000].contiguous(), R[:1000].contiguous()
    Cs, ders = _gemm(As, B, bias=bias, gelu=2, want_pre=True)
    assert torch.allclose(Cs.float(), torch.nn.functional.gelu(pre_ref[:1000]), rtol=2e-2, atol=2e-2)
    assert torch.allclose(ders.float(), der_ref[:1000], rtol=1e-2, atol=6e-3)
    assert torch.allclose(_gemm(As, B, residual=Rs, rmode=2).float(), acc[:1000] * Rs.float(), rtol=2e-2, atol=3e-2)


def test_gemm8_bias_dropout_add_uses_the_shared_mask():
    """bias-dropout-add (transformer.py:397-413): the epilogue's keep bits are those of emdr2_dropout (what the backward regenerates)."""
    from emdr2_amd import _native as nat
    lib = nat.lib()
    g = torch.Generator(device="cuda").manual_seed(9)
    M, N, K, p, seed = 4096, 768, 128, 0.1, 12345
    A, B = _rand((M, K), g, 0.5), _rand((N, K), g, 0.2)
    bias = torch.randn(N, generator=g, device="cuda")
    R = _rand((M, N), g)
    ones = torch.ones((M, N), dtype=torch.bfloat16, device="cuda")
    mask = torch.empty_like(ones)
    nat.check(lib.emdr2_dropout(ones.data_ptr(), mask.data_ptr(), ones.numel(), N, p, seed, nat.stream_ptr()), "dropout")
    C = _gemm(A, B, bias=bias, residual=R, drop_p=p, seed=seed)
    ref = (A.float() @ B.float().T + bias) * mask.float() + R.float()
    assert abs(float((mask == 0).float().mean()) - p) < 5e-3
    assert torch.allclose(C.float(), ref, rtol=2e-2, atol=3e-2)


def test_gemm8_is_bit_stable_across_runs():
    g = torch.Generator(device="cuda").manual_seed(3)
    M, N, K = 65536, 2304, 768
    A, B = _rand((M, K), g), _rand((N, K), g)
    bias = torch.randn(N, generator=g, device="cuda")
    first = _gemm(A, B, bias=bias)
    for _ in range(4):
        assert torch.equal(_gemm(A, B, bias=bias), first)


def test_gemm8_agrees_with_the_general_kernel():
    """Same operands through gemm.hip (reached with an M that is not a multiple of 256) and gemm8.hip: equal up to the one extra bf16 rounding
    the row-order residual add makes."""
    g = torch.Generator(device="cuda").manual_seed(5)
    M, N, K = 8192, 768, 768
    A, B = _rand((M + 8, K), g), _rand((N, K), g)
    bias = torch.randn(N, generator=g, device="cuda")
    general = _gemm(A, B, bias=bias)[:M]
    fast = _gemm(A[:M].contiguous(), B, bias=bias)
    assert float((general != fast).float().mean()) < 1e-3
    assert torch.allclose(general.float(), fast.float(), rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("M,V,H", [(4096, 30720, 768), (2048, 1024, 128), (256, 512, 256)])
def test_lm_head_fused_with_log_softmax_gather(M, V, H):
    """The LSE epilogue (LM head + log-softmax + gold gather, logits never stored) against the unfused kernels on the same operands, and
    against an fp32 torch reference (language_model.py:28-41 + train_e2eqa.py:79-96)."""
    from emdr2_amd.model import kernels as K
    g = torch.Generator(device="cuda").manual_seed(M + V)
    hid = _rand((M, H), g)
    W = torch.nn.Parameter(torch.randn((V, H), generator=g, device="cuda") * 0.05)
    bias = torch.nn.Parameter(torch.randn(V, generator=g, device="cuda") * 0.1)
    labels = torch.randint(0, V, (M,), generator=g, device="cuda")
    labels[:7] = torch.tensor([0, 1, 63, 64, 255, 256, V - 1], device="cuda")          # block / tile boundaries
    with torch.no_grad():
        fused = K.lm_head_gold_logprob(hid, W, bias, labels)
        logits = K.linear(hid, W, bias)
        unfused = K.lse_gather(logits, labels)
        ref = torch.log_softmax(hid.float() @ W.bfloat16().float().T + bias, dim=-1).gather(1, labels[:, None])[:, 0]
    assert torch.allclose(fused, unfused, rtol=0, atol=2e-4), float((fused - unfused).abs().max())   # same bf16-rounded logits, different summation order
    assert torch.allclose(fused, ref, rtol=2e-2, atol=5e-2)


def test_lm_head_fused_loss_with_labels_outside_the_vocabulary():
    """a label the epilogue never meets (negative: an ignore_index such as -100; or >= V) must not read uninitialised memory:
    gold is 0 for such a row, so the result is exactly -logsumexp(row), deterministically; in-range rows are unaffected."""
    from emdr2_amd.model import kernels as K
    g = torch.Generator(device="cuda").manual_seed(5)
    M, V, H = 512, 1024, 128
    hid = _rand((M, H), g)
    W = torch.nn.Parameter(torch.randn((V, H), generator=g, device="cuda") * 0.05)
    bias = torch.nn.Parameter(torch.randn(V, generator=g, device="cuda") * 0.1)
    labels = torch.randint(0, V, (M,), generator=g, device="cuda")
    labels[3], labels[100], labels[511] = -100, V, V + 7
    with torch.no_grad():
        torch.empty(M * 64, device="cuda").fill_(float("nan"))           # poison what the allocator hands out next
        a = K.lm_head_gold_logprob(hid, W, bias, labels)
        b = K.lm_head_gold_logprob(hid, W, bias, labels)
        lse = torch.logsumexp(K.linear(hid, W, bias).float(), dim=-1)
    assert torch.equal(a, b) and bool(torch.isfinite(a).all())
    out = torch.tensor([3, 100, 511], device="cuda")
    assert torch.allclose(a[out], -lse[out], atol=2e-4)
    inside = torch.ones(M, dtype=torch.bool, device="cuda"); inside[out] = False
    ref = K.lse_gather(K.linear(hid, W, bias), labels.clamp(0, V - 1))
    assert torch.allclose(a[inside], ref[inside], atol=2e-4)


def test_persistent_gemm_tile_order_at_three_million_rows():
    """M = 3,276,800 rows (B = 64 x top-k 100 x S 512: the TriviaQA shape) x N = 1536: 76,800 tiles in ONE n-group.  The tile walk divides by
    the rounded-up reciprocal of 6 x 12,800; uncorrected, that quotient is one too large from tile 59,075 on (an out-of-range tile: a
    memory fault at top-k 100).  Sampled row blocks against fp32 torch."""
    from emdr2_amd.model import kernels as K
    g = torch.Generator(device="cuda").manual_seed(11)
    M, N, Kd = 64 * 100 * 512, 1536, 128
    a = (torch.randn((M, Kd), generator=g, device="cuda") * 0.5).bfloat16()
    b = (torch.randn((N, Kd), generator=g, device="cuda") * 0.5).bfloat16()
    bias = torch.randn(N, generator=g, device="cuda")
    c = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    K.gemm_nt(a, Kd, b, Kd, c, N, M, N, Kd, bias=bias)
    torch.cuda.synchronize()
    for r0 in (0, 59_075 // 6 * 256, M // 2, M - 256):
        ref = a[r0:r0 + 256].float() @ b.float().T + bias
        got = c[r0:r0 + 256].float()
        assert bool(torch.isfinite(got).all()), r0
        assert torch.allclose(got, ref, rtol=2e-2, atol=5e-2), (r0, float((got - ref).abs().max()))
    assert bool(torch.isfinite(c[::4099].float()).all())                      # every tile was written (no NaN left from the fill)
