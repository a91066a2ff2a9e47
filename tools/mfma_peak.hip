// MFMA issue-rate probe: 8 waves/CU x 256 CUs, N back-to-back independent MFMAs per wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ void __launch_bounds__(512) probe(float *out, int iters, float seed)
{
    floatx16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    half8 a, b; bf8 ab, bb;
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int j = 0; j < 8; ++j) {
        h = h * 1664525u + 1013904223u; float u1 = ((h >> 8) + 1) * (1.0f / 16777217.0f);
        h = h * 1664525u + 1013904223u; float u2 = (h >> 8) * (1.0f / 16777216.0f);
        float n1 = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2), n2 = sqrtf(-2.f * logf(u1)) * sinf(6.2831853f * u2);
        if (seed < 0.3f) { n1 = seed + j; n2 = seed * 0.5f + j; }
        a[j] = (_Float16)n1; b[j] = (_Float16)n2; ab[j] = (__bf16)n1; bb[j] = (__bf16)n2;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
            if (KIND == 1) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[i], 0, 0, 0);
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][7];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND>
__global__ void __launch_bounds__(512) probe16(float *out, int iters, float seed)
{
    floatx4 acc[16];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    half8 a, b; bf8 ab, bb;
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int j = 0; j < 8; ++j) {
        h = h * 1664525u + 1013904223u; float u1 = ((h >> 8) + 1) * (1.0f / 16777217.0f);
        h = h * 1664525u + 1013904223u; float u2 = (h >> 8) * (1.0f / 16777216.0f);
        float n1 = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2), n2 = sqrtf(-2.f * logf(u1)) * sinf(6.2831853f * u2);
        if (seed < 0.3f) { n1 = seed + threadIdx.x * 0.001f + j; n2 = seed * 0.5f + j; }
        a[j] = (_Float16)n1; b[j] = (_Float16)n2; ab[j] = (__bf16)n1; bb[j] = (__bf16)n2;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
            if (KIND == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[i], 0, 0, 0);
        }
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += acc[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
    float *out; hipMalloc(&out, 256 * 512 * 4 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 60000;
    for (int kind = 0; kind < 7; ++kind) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (kind == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(512), 0, 0, out, iters, 0.25f);
            if (kind == 1) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(512), 0, 0, out, iters, 0.25f);
            if (kind == 3) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(512), 0, 0, out, iters, 1.0f);
            if (kind == 4) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(512), 0, 0, out, iters, 1.0f);
            if (kind == 2) hipLaunchKernelGGL(probe16<0>, dim3(256), dim3(512), 0, 0, out, iters, 0.25f);
            if (kind == 5) hipLaunchKernelGGL(probe16<1>, dim3(256), dim3(512), 0, 0, out, iters, 1.0f);
            if (kind == 6) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(512), 0, 0, out, iters, 1.0f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flops = (kind == 2 || kind == 5) ? 256.0 * 8 * iters * 16 * (2.0 * 16 * 16 * 32) : 256.0 * 8 * iters * 8 * (2.0 * 32 * 32 * 16);
            printf("kind %d (%s) rep %d: %.3f ms -> %.1f TFLOP/s\n", kind, kind == 0 ? "32x32x16 f16 const" : kind == 1 ? "32x32x16 bf16 const" : kind == 2 ? "16x16x32 f16 const" : kind == 3 ? "32x32x16 f16 N(0,1)" : kind == 5 ? "16x16x32 bf16 N(0,1)" : "32x32x16 bf16 N(0,1)", rep, ms, flops / ms / 1e9);
        }
    }
    return 0;
}
