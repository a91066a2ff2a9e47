// emdr2_amd/csrc/rng.h -- counter-based dropout bits shared by every kernel that drops (GEMM epilogue, attention, softmax, embedding).
// keep(seed, i) is a pure function of the call-site seed and the element's linear index, so the backward regenerates the forward's
// mask instead of storing it and an activation-recompute pass (--checkpoint-activations) sees the same bits.
// Reference: torch.nn.Dropout at transformer.py:262,386-394 / language_model.py:180 (Philox there; any i.i.d. Bernoulli(1-p) stream is
// an equally valid sample -- the reference's own mask is not reproducible across GPUs either).
#ifndef EMDR2_RNG_H
#define EMDR2_RNG_H
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ uint32_t emdr2_mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// uniform in [0, 1) with 24 bits from (seed, 64-bit element index)
__device__ __forceinline__ float emdr2_uniform01(uint32_t seed, unsigned long long idx)
{
    const uint32_t h = emdr2_mix32((uint32_t)idx ^ emdr2_mix32((uint32_t)(idx >> 32) + seed * 0x9e3779b9u + 0x85ebca6bu));
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}
__device__ __forceinline__ bool emdr2_keep(uint32_t seed, unsigned long long idx, float drop_p) { return emdr2_uniform01(seed, idx) >= drop_p; }
#endif
