// emdr2_amd/csrc/gemm_common.h -- scalar helpers shared by the bf16 GEMM kernels (gemm.hip, gemm8.hip): bf16 packing, the erf-form GELU of the
// reference (transformer.py:80,103-104: F.gelu, not the tanh fusion) and its derivative, streaming stores.
#ifndef EMDR2_GEMM_COMMON_H
#define EMDR2_GEMM_COMMON_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rng.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float floatx2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f)
{
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40); // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);                                             // round to nearest even
    return (uint16_t)(u >> 16);
}
// two fp32 -> packed bf16 pair with the hardware converter (v_cvt_pk_bf16_f32, round to nearest even, NaN preserved)
__device__ __forceinline__ uint32_t pack2_bf16(float a, float b)
{
    const floatx2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
// outputs are written once and not re-read by the producing kernel: streaming stores keep them from evicting the operand panels the
// other tiles of the XCD are about to reuse
__device__ __forceinline__ void store_stream(uint16_t *dst, uint4 v)
{
    const u32x4_t x = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(x, (u32x4_t *)dst);
}
// erf-form GELU 0.5 x (1 + erf(x / sqrt 2)).  erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16 output grid) on one
// v_rcp and one v_exp; the library erff costs ~3x.  With erf(|z|) = 1 - P(t) exp(-z^2), t = 1 / (1 + p |z|), both signs of x collapse into
//   gelu(x) = max(x, 0) - |x| * (P(t) / 2) * exp(-x^2 / 2)
// (x >= 0: x - x P e / 2;  x < 0: x P e / 2): 11 full-rate VALU ops (|x| and the negations ride as source modifiers), the 1/2 folded into the
// polynomial's coefficients, exp(-x^2 / 2) = exp2(-(x sqrt(log2(e) / 2))^2).
__device__ __forceinline__ float gelu_half_poly_exp(float x, float &e)
{
    const float t = __builtin_amdgcn_rcpf(fmaf(fabsf(x), 0.3275911f * 0.70710678118654752f, 1.0f));
    const float w = x * 0.84932180028801904f;                                      // sqrt(log2(e) / 2)
    e = __builtin_amdgcn_exp2f(-(w * w));                                          // exp(-x^2 / 2)
    return t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f), 0.5f * 1.421413741f), 0.5f * -0.284496736f), 0.5f * 0.254829592f);
}
__device__ __forceinline__ float gelu_erf(float x)
{
    float e;
    const float hp = gelu_half_poly_exp(x, e);
    return fmaf(-fabsf(x), hp * e, fmaxf(x, 0.0f));
}
// both at once (one rcp, one exp2, one polynomial): the activation and its derivative of the same pre-activation
__device__ __forceinline__ float gelu_erf_with_grad(float x, float &grad)
{
    float e;
    const float hp = gelu_half_poly_exp(x, e);
    grad = fmaf(e, fmaf(x, 0.3989422804014327f, -copysignf(hp, x)), x >= 0.0f ? 1.0f : 0.0f);
    return fmaf(-fabsf(x), hp * e, fmaxf(x, 0.0f));
}
// d/dx of the erf-form GELU: Phi(x) + x phi(x) = step(x) + exp(-x^2 / 2) (x / sqrt(2 pi) - sign(x) P(t) / 2), same erf approximation and
// the same exponential as the forward
__device__ __forceinline__ float gelu_erf_grad(float x)
{
    float e;
    const float hp = gelu_half_poly_exp(x, e);
    const float step = x >= 0.0f ? 1.0f : 0.0f;
    return fmaf(e, fmaf(x, 0.3989422804014327f, -copysignf(hp, x)), step);
}
#endif
