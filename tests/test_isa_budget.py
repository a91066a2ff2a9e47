"""Register / scratch budget of the kernels in the built library (no GPU: read from the gfx950 code objects inside libemdr2_hip.so).

The numbers pinned here are the ones the kernels' designs rest on: the occupancy each attention kernel is laid out for, no
scratch traffic in any hot kernel, and -- the defect found in round 4 by reading the ISA -- no copying of the O accumulators around the PV
product of the attention forward (LLVM structurizes a wave-uniform if / else with several conditional children like a divergent one; the
32 v_mov_b64 per 32-key step that came of it cost the forward 8-9 %).  A compiler or source change that brings any of this back fails here,
on the CPU, before a benchmark has to notice it."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources as kr  # noqa: E402

pytestmark = pytest.mark.skipif(not (os.path.exists(kr.DEFAULT_LIB) and os.path.exists(os.path.join(kr.LLVM, "llvm-readelf"))),
                                reason="needs the built library and llvm-readelf")


@pytest.fixture(scope="module")
def table():
    return kr.kernels()


def _sel(table, sub):
    rows = [e for e in table if sub in e["name"]]
    assert rows, "no kernel named *%s* in the library" % sub
    return rows


def test_library_holds_the_hot_kernels(table):
    for sub in ("mips_scan8_kernel", "gemm8_kernel", "gemm8t_kernel", "attention_fwd_kernel", "attention_bwd_dq_kernel", "attention_bwd_dkv_kernel",
                "layernorm_fwd768_kernel", "layernorm_bwd768_kernel", "finalize_kernel", "select_kernel"):
        _sel(table, sub)
    assert len(table) > 60


def test_no_kernel_spills_vector_registers_and_only_the_unaligned_gemm_fallback_has_a_stack(table):
    for e in table:
        assert e["vgpr_spill_count"] == 0, e
        if e["private_segment_fixed_size"]:
            # gemm_nt_kernel<.., VEC = false, ..>: the element-wise staging of operands whose K or pointers are not 16-byte aligned (odd test
            # shapes; no launch of a training step or a search takes it) indexes a small local array
            assert "gemm_nt_kernel" in e["name"] and "Lb0E" in e["name"], e


def test_attention_kernels_fit_the_occupancy_they_are_laid_out_for(table):
    # forward: four workgroups of four waves per CU = 4 waves per SIMD = at most 128 registers (dropout + causal mask: three, 168)
    for e in _sel(table, "attention_fwd_kernel"):
        both = "ILb1ELb1E" in e["name"]
        assert e["vgpr_count"] <= (168 if both else 128), e
        assert e["agpr_count"] == 0 and e["sgpr_spill_count"] == 0, e
    # backward: three workgroups per CU = at most 168 registers; the dK / dV kernel's static LDS (K / V tiles + two Q / dO stages + statistics)
    # must leave room for three workgroups in 160 KiB
    for sub in ("attention_bwd_dq_kernel", "attention_bwd_dkv_kernel"):
        for e in _sel(table, sub):
            assert e["vgpr_count"] <= 168, e
    for e in _sel(table, "attention_bwd_dkv_kernel"):
        assert 3 * e["group_segment_fixed_size"] <= 160 * 1024, e


def test_persistent_gemm_and_scan_kernels_fit_two_waves_per_simd(table):
    for sub in ("gemm8_kernel", "gemm8t_kernel", "mips_scan8_kernel"):
        for e in _sel(table, sub):
            assert e["vgpr_count"] + e["agpr_count"] <= 256, e


def test_attention_forward_does_not_copy_its_accumulators():
    # 19 - 36 v_mov_b64 in the whole kernel today (prologue and the peeled first block); the defect was 140, 32 of them in every 32-key step
    counts = kr.count_opcode(kr.DEFAULT_LIB, "v_mov_b64", "attention_fwd_kernel")
    assert len(counts) == 4, counts
    for name, c in counts.items():
        assert c <= 48, (name, c)


def test_power_bound_kernels_use_the_16x16x32_mfma_shape():
    # at the board's power cap the 16 x 16 x 32 shape does ~19 % more flops than 32 x 32 x 16 (half the accumulator registers moved
    # per flop); the MIPS scan and both persistent GEMMs are built on it
    for sub, small, big in (("mips_scan8_kernel", "v_mfma_f32_16x16x32_f16", "v_mfma_f32_32x32x16_f16"),
                            ("gemm8_kernel", "v_mfma_f32_16x16x32_bf16", "v_mfma_f32_32x32x16_bf16"),
                            ("gemm8t_kernel", "v_mfma_f32_16x16x32_bf16", "v_mfma_f32_32x32x16_bf16")):
        n_small = kr.count_opcode(kr.DEFAULT_LIB, small, sub)
        n_big = kr.count_opcode(kr.DEFAULT_LIB, big, sub)
        assert n_small and all(c >= 64 for c in n_small.values()), (sub, n_small)
        assert all(c == 0 for c in n_big.values()), (sub, n_big)
