"""CPU: the C-ABI library loads and exports every symbol include/emdr2_mips.h declares; argument
validation works without touching a GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    from emdr2_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        g.build()
    return _native.lib()


def _declared():
    names = set()
    for h in sorted(os.listdir(os.path.join(ROOT, "include"))):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\bint\s+(emdr2_\w+)\s*\(", text))
    return sorted(names)


def test_every_declared_symbol_is_exported_and_bound(lib):
    from emdr2_amd import _native
    names = _declared()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), "library does not export %s" % n
        assert n in _native.SIGNATURES, "ctypes binding missing for %s" % n
    assert sorted(_native.SIGNATURES) == names, "binding table and header drifted"


def test_abi_version(lib):
    assert lib.emdr2_abi_version() == 4          # r06: the validation-only fp32 compute path (include/emdr2_ops_f32.h); r05: packed records


def test_layout_bytes_and_argument_validation(lib):
    n = ctypes.c_size_t()
    assert lib.emdr2_mips_layout_bytes(21015324, 768, ctypes.byref(n)) == 0
    assert n.value == ((21015324 + 511) // 512 * 512) * 768 * 2
    assert lib.emdr2_mips_layout_bytes(10, 100, ctypes.byref(n)) == -1        # dim % 32 != 0
    assert lib.emdr2_mips_layout_bytes(10, 32, ctypes.byref(n)) == -1         # dim < 64
    assert lib.emdr2_mips_workspace_bytes(512, 768, 50, ctypes.byref(n)) == 0 and n.value > 512 * 16384 * 8
    assert lib.emdr2_mips_workspace_bytes(512, 768, 121, ctypes.byref(n)) == -1
    # null pointers are rejected before any launch
    assert lib.emdr2_mips_search(None, 10, 768, 0, None, None, 1, 5, None, None, None, None, None, None, 0, None) == -1
    assert lib.emdr2_mips_merge(None, None, None, 2, 4, 5, None, None, None, None) == -1


def test_missing_library_fails_loudly(monkeypatch):
    from emdr2_amd import _native
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", "/nonexistent/libemdr2_hip.so")
    with pytest.raises(_native.NativeError):
        _native.lib()


def test_index_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from emdr2_amd import _native
    from emdr2_amd.data.emdr2_index import HipIndexShard
    with pytest.raises(_native.NativeError):
        HipIndexShard(768, 100, 0)


def test_ops_entry_points_validate_arguments_without_a_gpu(lib):
    """Bad shapes / null pointers come back as status codes before any launch (no exceptions across the ABI)."""
    n = ctypes.c_size_t()
    assert lib.emdr2_mips_exact_workspace_bytes_f32(1000, 4, ctypes.byref(n)) == 0 and n.value >= 8 * 1000 * 4
    assert lib.emdr2_mips_search_f32(None, 10, 768, 0, None, None, 1, 5, None, None, None, None, None, None, 0, None) == -1
    assert lib.emdr2_mips_merge_f32(None, None, None, 2, 4, 5, None, None, None, None) == -1
    one = ctypes.c_void_p(4096)                                          # a non-null, 16-byte aligned fake pointer: shapes are checked first
    # GEMM: K must be a multiple of 32, split-K only with fp32 output and no epilogue, dropout only unbatched
    assert lib.emdr2_gemm_nt_bf16(one, 64, one, 64, one, 64, 16, 16, 40, 1, 0, 0, 0, 1, 0, 0, 0, 1.0, None, 0, None, None, 0, 0, 1, 0.0, 0, None) == -1
    assert lib.emdr2_gemm_nt_bf16(one, 64, one, 64, one, 64, 16, 16, 64, 1, 0, 0, 0, 1, 0, 0, 0, 1.0, None, 0, None, None, 0, 0, 2, 0.0, 0, None) == -1
    assert lib.emdr2_gemm_nt_bf16(one, 64, one, 64, one, 64, 16, 16, 64, 2, 0, 0, 0, 1, 0, 0, 0, 1.0, None, 0, None, None, 0, 0, 1, 0.1, 7, None) == -1
    assert lib.emdr2_gemm_nt_bf16(one, 64, one, 64, one, 64, 16, 16, 64, 1, 0, 0, 0, 1, 0, 0, 0, 1.0, None, 0, None, None, 3, 0, 1, 0.0, 0, None) == -1      # residual_mode is 0, 1 or 2
    assert lib.emdr2_gemm_nt_bf16(one, 64, one, 64, one, 64, 16, 16, 64, 1, 0, 0, 0, 1, 0, 0, 0, 1.0, None, 2, None, None, 0, 0, 1, 0.0, 0, None) == -1      # gelu = 2 needs pre_act
    assert lib.emdr2_gemm_tn_bf16(one, 64, one, 64, one, 64, 64, 64, 48, 1, None, None) == -1      # R % 32
    assert lib.emdr2_gemm_tn_bf16(None, 64, one, 64, one, 64, 64, 64, 64, 1, None, None) == -1
    # attention: head dim 64 and sk % 32 == 0 only (-4 = unsupported shape, the caller falls back to the composed path)
    args_f = (one, 64, 64, 64, one, 64, 64, 64, one, 64, 64, 64, one, one, one, 1, 1, 32)
    assert lib.emdr2_attention_fwd(*args_f, 72, 64, 0, 0.125, 0.0, 0, None, None, None) == -4
    assert lib.emdr2_attention_fwd(*args_f, 128, 32, 0, 0.125, 0.0, 0, None, None, None) == -4
    assert lib.emdr2_attention_fwd(*args_f, 128, 64, 0, 0.125, 1.5, 0, None, None, None) == -1
    assert lib.emdr2_dropout(one, one, 64, 12, 0.1, 1, None) == -1          # cols % 8
    assert lib.emdr2_layernorm_fwd(one, one, one, one, one, one, 4, 12, 1e-5, None) == -1


def test_r05_entry_points_validate_their_arguments(lib):
    """The packed-record exchange and the split-key attention entries (ABI 3): bad pointers are refused without touching a GPU, and the
    split-key plan -- a pure function of the query / key extents, never of the batch -- says what include/emdr2_ops.h says."""
    assert lib.emdr2_mips_search_records(None, 10, 64, 0, None, None, 1, 1, None, 0, None, None, None, 0, None) == -1
    assert lib.emdr2_mips_merge_records(None, 2, 4, 5, 0, None, None, None, None) == -1
    assert lib.emdr2_mips_pack_records(None, None, None, None, 0, 5, 0, None, None) == -1
    ks, fb, bb = ctypes.c_int(), ctypes.c_size_t(), ctypes.c_size_t()

    def plan(batch, heads, sq, sk):
        assert lib.emdr2_attention_splitkv_plan(batch, heads, sq, sk, ctypes.byref(ks), ctypes.byref(fb), ctypes.byref(bb)) == 0
        return ks.value, fb.value, bb.value
    assert plan(16, 12, 32, 25600) == (13, 13 * 16 * 12 * 32 * 66 * 4, 13 * 16 * 12 * 32 * 64 * 4)       # 400 key blocks in ranges of 32
    assert plan(64, 12, 32, 25600)[0] == plan(1, 12, 32, 25600)[0] == 13                               # the batch does not enter
    assert plan(16, 12, 32, 4095) == (1, 0, 0) and plan(16, 12, 32, 4096)[0] == 2                       # short key sets never split
    assert plan(16, 12, 129, 25600) == (1, 0, 0) and plan(16, 12, 128, 65536)[0] == 32                 # one query block only
    assert lib.emdr2_attention_splitkv_plan(0, 12, 32, 25600, ctypes.byref(ks), ctypes.byref(fb), ctypes.byref(bb)) == -1


def test_fp32_validation_entry_points_validate_their_arguments(lib):
    """include/emdr2_ops_f32.h (ABI 4): null pointers / empty shapes are refused before any launch."""
    one = ctypes.c_void_p(4096)
    assert lib.emdr2_f32_gemm(None, 1, 1, 0, 0, one, 1, 1, 0, 0, one, 1, 1, 0, 0, 4, 4, 4, 1, 1, 1.0, None, None, 0, None) == -1
    assert lib.emdr2_f32_gemm(one, 1, 1, 0, 0, one, 1, 1, 0, 0, one, 1, 1, 0, 0, 4, 4, 0, 1, 1, 1.0, None, None, 0, None) == -1
    assert lib.emdr2_f32_layernorm_fwd(one, one, one, one, one, None, 4, 16, 1e-5, None) == -1
    assert lib.emdr2_f32_layernorm_bwd(one, one, one, one, one, None, one, one, None, 4, 16, None) == -1
    assert lib.emdr2_f32_softmax_mask_fwd(one, one, None, 1, 1, 4, 4, 0, None) == -1
    assert lib.emdr2_f32_softmax_mask_bwd(one, None, one, one, 1, 1, 4, 4, 0, None) == -1
    assert lib.emdr2_f32_gelu_fwd(one, None, 8, None) == -1 and lib.emdr2_f32_gelu_bwd(one, one, one, 0, None) == -1
    assert lib.emdr2_f32_embedding_fwd(one, None, one, one, one, one, 4, 4, 16, None) == -1      # a token-type table without token types
    assert lib.emdr2_f32_embedding_bwd(one, None, one, one, None, None, 4, 4, 16, None) == -1
    assert lib.emdr2_f32_lse_gather_fwd(one, one, one, None, 4, 16, None) == -1
    assert lib.emdr2_f32_lse_gather_bwd(one, one, one, one, None, 4, 16, None) == -1
    assert lib.emdr2_f32_retriever_prior_fwd(one, one, one, one, 2, 2000, 16, 1.0, None) == -1    # K <= 1024
    assert lib.emdr2_f32_retriever_prior_bwd(one, one, one, one, one, None, 2, 4, 16, 1.0, None) == -1
