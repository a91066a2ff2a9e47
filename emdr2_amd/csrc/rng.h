// emdr2_amd/csrc/rng.h -- counter-based dropout bits shared by every kernel that drops (GEMM epilogue, attention, softmax, embedding).
// keep(seed, row, col) is a pure function of the call-site seed and the element's (row, column), so the backward regenerates the
// forward's mask instead of storing it and an activation-recompute pass (--checkpoint-activations) sees the same bits.
// One 32-bit hash serves the two elements of a column pair (16 bits each): drop iff bits16 < round(p * 65536), i.e. the effective
// probability is p quantised to 2^-16 and the survivors are scaled by 65536 / (65536 - thr) so the expectation is exact.
// Reference: torch.nn.Dropout at transformer.py:262,386-394 / language_model.py:180 (Philox there; any i.i.d. Bernoulli(1-p) stream is
// an equally valid sample -- the reference's own mask is not reproducible across GPU counts either).
#ifndef EMDR2_RNG_H
#define EMDR2_RNG_H
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ uint32_t emdr2_mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ uint32_t emdr2_drop_thr(float drop_p) { return (uint32_t)(drop_p * 65536.f + 0.5f); }
__host__ __device__ __forceinline__ float emdr2_keep_scale(float drop_p) { return 65536.f / (65536.f - (float)emdr2_drop_thr(drop_p)); }
// per-row part of the hash (computed once per row / lane)
__device__ __forceinline__ uint32_t emdr2_row_hash(uint32_t seed, unsigned long long row)
{
    return emdr2_mix32((uint32_t)row ^ emdr2_mix32((uint32_t)(row >> 32) + seed * 0x9e3779b9u + 0x85ebca6bu));
}
// 32 random bits for the column pair (col & ~1, col | 1): low half for the even column, high half for the odd one.
// The row hash above is a full two-round mix; the per-pair step is deliberately light (xor-shift, one FULL-RATE 24-bit multiply, xor-shift:
// 6 VALU ops where the earlier two-multiply finaliser cost ~22 issue slots with its quarter-rate 32-bit multiplies): in a GEMM epilogue
// the pair hash is paid 4,096 times per lane-tile and was 60 % of the bias-dropout-add epilogue.  Keep rate, row / column sum variance,
// lag-1..32 auto-correlation and cross-seed correlation of the resulting masks are at the sampling-noise level (same as the old hash).
#define EMDR2_PAIR_MUL 0x9e3779b1u
__device__ __forceinline__ uint32_t emdr2_pair_bits_prod(uint32_t rowhash, uint32_t pair_prod)     // pair_prod = (col >> 1) * EMDR2_PAIR_MUL
{
    uint32_t x = pair_prod ^ rowhash;
    x ^= x >> 16;
    x = (x & 0xffffffu) * 0xa2d2e5u;          // both factors < 2^24: v_mul_u32_u24
    x ^= x >> 15;
    return x;
}
__device__ __forceinline__ uint32_t emdr2_pair_bits(uint32_t rowhash, uint32_t col) { return emdr2_pair_bits_prod(rowhash, (col >> 1) * EMDR2_PAIR_MUL); }
__device__ __forceinline__ bool emdr2_keep(uint32_t rowhash, uint32_t col, uint32_t thr)
{
    const uint32_t b = emdr2_pair_bits(rowhash, col);
    return ((col & 1u) ? (b >> 16) : (b & 0xffffu)) >= thr;
}
#endif
