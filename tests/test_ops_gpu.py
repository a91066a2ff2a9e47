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
                                     pre.data_ptr() if pre is not None else None, residual.data_ptr() if residual is not None else None,
                                     int(out_f32), 1, nat.stream_ptr()), "gemm")
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
                                     heads, hn, hn, s * s, 0.125, None, 0, None, None, 0, 1, nat.stream_ptr()), "gemm")
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
