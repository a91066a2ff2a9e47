"""ctypes binding of libemdr2_hip.so (C ABI: include/emdr2_mips.h, include/emdr2_assembly.h).

The library is built in-tree by `__graft_entry__.build()` / `make -C emdr2_amd/csrc`.  If it is
missing this module raises at first use -- the product path never falls back to a CPU or eager
implementation.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libemdr2_hip.so")     # tools/ may point this at lib/libemdr2_hip_exp.so (`make exp`) before first use

EMDR2_ABI_VERSION = 4
FLAG_AMBIGUOUS = 1
FLAG_OVERFLOW = 2
MAX_TOPK = 120

_ERRORS = {-1: "bad argument", -2: "workspace too small", -3: "HIP launch/runtime error", -4: "unsupported shape"}

# name -> (restype, argtypes); mirrors include/emdr2_mips.h one to one
_vp, _i32, _i64, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t
SIGNATURES = {
    "emdr2_abi_version": (_i32, []),
    "emdr2_device_cu_count": (_i32, []),
    "emdr2_mips_layout_bytes": (_i32, [_i64, _i32, ctypes.POINTER(_sz)]),
    "emdr2_mips_pack_rows": (_i32, [_vp, _i64, _i32, _i64, _i64, _vp, _vp, _vp]),
    "emdr2_mips_unpack_rows": (_i32, [_vp, _i64, _i32, _vp, _i64, _vp, _vp]),
    "emdr2_mips_workspace_bytes": (_i32, [_i32, _i32, _i32, ctypes.POINTER(_sz)]),
    "emdr2_mips_search": (_i32, [_vp, _i64, _i32, _i64, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "emdr2_mips_exact_workspace_bytes": (_i32, [_i64, _i32, ctypes.POINTER(_sz)]),
    "emdr2_mips_search_exact": (_i32, [_vp, _i64, _i32, _i64, _vp, _i32, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "emdr2_mips_merge": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "emdr2_mips_search_f32": (_i32, [_vp, _i64, _i32, _i64, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "emdr2_mips_exact_workspace_bytes_f32": (_i32, [_i64, _i32, ctypes.POINTER(_sz)]),
    "emdr2_mips_search_exact_f32": (_i32, [_vp, _i64, _i32, _i64, _vp, _i32, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "emdr2_mips_merge_f32": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "emdr2_mips_search_records": (_i32, [_vp, _i64, _i32, _i64, _vp, _vp, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _sz, _vp]),
    "emdr2_mips_merge_records": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "emdr2_mips_pack_records": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "emdr2_mips_debug_scores": (_i32, [_vp, _i64, _i32, _vp, _i32, _vp, _vp, _sz, _vp]),
    "emdr2_mips_set_timing": (_i32, [_i32]),
    "emdr2_mips_timing_collect": (_i32, [ctypes.POINTER(ctypes.c_float), ctypes.POINTER(_i64), _i32, ctypes.POINTER(_i32)]),
}

SIGNATURES["emdr2_assemble_evidence"] = (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32,
                                                  _vp, _vp, _vp, _vp, _vp, _vp])


_f32 = ctypes.c_float
_u32 = ctypes.c_uint32
SIGNATURES["emdr2_gemm_nt_bf16"] = (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _i32, _i64, _i64, _i64,
                                             _f32, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _f32, _u32, _vp])
SIGNATURES["emdr2_gemm_tn_bf16"] = (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp])
SIGNATURES["emdr2_dropout"] = (_i32, [_vp, _vp, _i64, _i32, _f32, _u32, _vp])
SIGNATURES["emdr2_transpose_bf16"] = (_i32, [_vp, _i64, _vp, _i64, _i32, _i32, _i32, _i64, _i64, _i32, _i64, _i64, _vp, _vp])


SIGNATURES.update({
    "emdr2_layernorm_fwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32, _vp]),
    "emdr2_layernorm_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "emdr2_layernorm_bwd_mask": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _f32, _u32, _vp]),
    "emdr2_softmax_mask_fwd": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _f32, _u32, _vp]),
    "emdr2_softmax_mask_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _u32, _vp, _vp]),
    "emdr2_softmax_mask_t": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _u32, _vp]),
    "emdr2_gelu_bwd": (_i32, [_vp, _vp, _vp, _i64, _vp]),
    "emdr2_embedding_fwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _f32, _u32, _vp]),
    "emdr2_embedding_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _f32, _u32, _vp]),
    "emdr2_attention_fwd": (_i32, [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32,
                                   _f32, _f32, _u32, _vp, _vp, _vp]),
    "emdr2_attention_bwd": (_i32, [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _i64, _i64,
                                   _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _u32, _vp]),
    "emdr2_lse_gather_fwd": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "emdr2_lse_gather_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "emdr2_sumsq_f32": (_i32, [_vp, _i64, _vp, _vp, _vp]),
    "emdr2_adam_step": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _i32, _vp, _f32, _vp]),
    "emdr2_adam_step_flat": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _f32, _f32, _f32, _f32, _f32, _i32, _vp, _f32, _vp]),
    "emdr2_scale_cast_f32_to_bf16": (_i32, [_vp, _vp, _i64, _f32, _vp]),
    "emdr2_widen_bf16_to_f32": (_i32, [_vp, _vp, _i64, _vp]),
    "emdr2_cast_f32_to_bf16": (_i32, [_vp, _vp, _i64, _vp]),
    "emdr2_accum_bf16_to_f32": (_i32, [_vp, _vp, _i64, _f32, _vp]),
})


SIGNATURES.update({
    "emdr2_seq_lengths": (_i32, [_vp, _i32, _i32, _vp, _vp, _vp]),
    "emdr2_seq_pack_ids": (_i32, [_vp, _vp, _vp, _i32, _i32, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "emdr2_gather_rows": (_i32, [_vp, _vp, _vp, _i64, _i32, _vp]),
    "emdr2_scatter_rows": (_i32, [_vp, _vp, _vp, _i64, _i32, _vp]),
    "emdr2_embedding_packed_fwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _f32, _u32, _vp]),
    "emdr2_embedding_packed_bwd": (_i32, [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _f32, _u32, _vp]),
    "emdr2_attention_varlen_fwd": (_i32, [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i64,
                                          _i32, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _u32, _vp, _vp, _vp]),
    "emdr2_attention_varlen_bwd": (_i32, [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _i64, _i64,
                                          _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _u32, _vp]),
})
SIGNATURES["emdr2_attention_splitkv_plan"] = (_i32, [_i32, _i32, _i32, _i32, ctypes.POINTER(_i32), ctypes.POINTER(_sz), ctypes.POINTER(_sz)])
SIGNATURES["emdr2_attention_fwd_splitkv"] = (_i32, [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _i64,
                                                      _i32, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _u32, _vp, _vp, _i32, _vp, _sz, _vp])
SIGNATURES["emdr2_attention_bwd_splitkv"] = (_i32, [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _vp, _vp,
                                                      _i64, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _u32,
                                                      _i32, _vp, _sz, _vp])
SIGNATURES["emdr2_gemm_nt_lse_bf16"] = (_i32, [_vp, _i64, _vp, _i64, _i32, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _vp])
SIGNATURES["emdr2_retriever_prior_fwd"] = (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp])
SIGNATURES["emdr2_retriever_prior_bwd"] = (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp])
SIGNATURES["emdr2_marginal_fwd"] = (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _vp])
SIGNATURES["emdr2_marginal_bwd"] = (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp])
SIGNATURES["emdr2_lse_combine"] = (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp])
SIGNATURES["emdr2_ops_set_timing"] = (_i32, [_i32])
SIGNATURES["emdr2_ops_timing_collect"] = (_i32, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_i64), _i32])


# include/emdr2_ops_f32.h (ABI 4): the validation-only fp32 compute path
SIGNATURES.update({
    "emdr2_f32_gemm": (_i32, [_vp, _i64, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _f32,
                              _vp, _vp, _i32, _vp]),
    "emdr2_f32_layernorm_fwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32, _vp]),
    "emdr2_f32_layernorm_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "emdr2_f32_softmax_mask_fwd": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "emdr2_f32_softmax_mask_bwd": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "emdr2_f32_gelu_fwd": (_i32, [_vp, _vp, _i64, _vp]),
    "emdr2_f32_gelu_bwd": (_i32, [_vp, _vp, _vp, _i64, _vp]),
    "emdr2_f32_embedding_fwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "emdr2_f32_embedding_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "emdr2_f32_lse_gather_fwd": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "emdr2_f32_lse_gather_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "emdr2_f32_retriever_prior_fwd": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp]),
    "emdr2_f32_retriever_prior_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp]),
})


class EvidenceArenaStruct(ctypes.Structure):
    """include/emdr2_assembly.h: emdr2_evidence_arena (device pointers)."""
    _fields_ = [("passage_tokens", _vp), ("passage_off", _vp), ("title_tokens", _vp), ("title_off", _vp),
                ("group_docs", _vp), ("group_off", _vp), ("doc_group", _vp), ("doc_pos", _vp), ("n_docs", _i64)]


_lib = None


class NativeError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises NativeError when the library is absent."""
    global _lib
    if _lib is None:
        # torch first: PyTorch-ROCm ships its own libamdhip64; if this library were loaded before it, the dynamic linker would bind us to
        # the system HIP runtime and torch to its bundled one -- two runtimes in one process, torch's device pointers invalid in ours
        # (seen as "HIP launch/runtime error" on the first kernel when build() and smoke() ran in one interpreter).
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise NativeError(
                "libemdr2_hip.so not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C emdr2_amd/csrc`. There is no CPU fallback for the EMDR2 hot path." % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)       # AttributeError here == header/library drift
            fn.restype, fn.argtypes = res, args
        if handle.emdr2_abi_version() != EMDR2_ABI_VERSION:
            raise NativeError("libemdr2_hip.so ABI version mismatch")
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        raise NativeError("%s failed: %s (%d)" % (what, _ERRORS.get(rc, "unknown error"), rc))


def stream_ptr(stream=None):
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return ctypes.c_void_p(s.cuda_stream)
