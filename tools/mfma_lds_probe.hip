// What do the LDS fragment reads cost next to the MFMAs at the power cap?  8 waves per CU, per iteration 32 x v_mfma_f32_16x16x32_bf16 on 32
// accumulators fed by R ds_read_b128 of N(0,1) bf16 data (R = 0: operands stay in registers; 8: the reads a 128 x 128 wave tile needs per
// 32 MFMAs; 12: the 128 x 64 wave tile of gemm8.hip; 16).       hipcc -O3 --offload-arch=gfx950 tools/mfma_lds_probe.hip -o tools/mfma_lds_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
template <int R, int D>
__global__ void __launch_bounds__(512) probe(float *out, int iters, const char *src, unsigned span_mask)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int i = threadIdx.x; i < 65536 / 2; i += 512) {
        h = h * 1664525u + 1013904223u; const float u1 = ((h >> 8) + 1) * (1.0f / 16777217.0f);
        h = h * 1664525u + 1013904223u; const float u2 = (h >> 8) * (1.0f / 16777216.0f);
        ((__bf16 *)smem)[i] = (__bf16)(sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2));
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const unsigned base = (unsigned)(uintptr_t)smem + wave * 8192 + l15 * 128 + (((lq) ^ ((l15 >> 1) & 7)) << 4);     // gemm8's conflict-free pattern
    floatx4 acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    bf8 f[16];
    for (int i = 0; i < 16; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[i]) : "v"(base), "n"((i & 3) * 2048 + (i >> 2) * 64));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < R; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[i]) : "v"(base + (it & 1) * 16), "n"((i & 3) * 2048 + (i >> 2) * 64));
        if (D) {                                                  // D LDS-DMA pieces of 1 KiB per iteration (gemm8.hip: 4 per 32 MFMAs), sources walking a window of span_mask + 1 bytes
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const unsigned o = (((unsigned)it * D + d) * 8192u * 256u + blockIdx.x * 8192u + wave * 1024u + lane * 16u) & span_mask;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) unsigned *)(src + o),
                                                 (__attribute__((address_space(3))) unsigned *)(smem + 32768 + ((it & 1) * D + d) * 8192 + wave * 1024), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * D) : "memory");
        }
        if (R) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]), "+v"(f[9]),
                                                      "+v"(f[10]), "+v"(f[11]), "+v"(f[12]), "+v"(f[13]), "+v"(f[14]), "+v"(f[15])::"memory");
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[i & 7], f[8 + (i >> 3) + 4 * ((i >> 2) & 1)], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main()
{
    float *out; (void)hipMalloc(&out, 256 * 512 * 4);
    char *src; (void)hipMalloc(&src, 1ull << 31); (void)hipMemset(src, 0x3c, 1ull << 31);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 30000;
    for (int rep = 0; rep < 2; ++rep)
        for (int k = 0; k < 7; ++k) {
            (void)hipEventRecord(e0);
            if (k == 0) hipLaunchKernelGGL((probe<0, 0>), dim3(256), dim3(512), 65536, 0, out, iters, src, 0u);
            if (k == 1) hipLaunchKernelGGL((probe<8, 0>), dim3(256), dim3(512), 65536, 0, out, iters, src, 0u);
            if (k == 2) hipLaunchKernelGGL((probe<12, 0>), dim3(256), dim3(512), 65536, 0, out, iters, src, 0u);
            if (k == 3) hipLaunchKernelGGL((probe<16, 0>), dim3(256), dim3(512), 65536, 0, out, iters, src, 0u);
            if (k == 4) hipLaunchKernelGGL((probe<12, 4>), dim3(256), dim3(512), 65536, 0, out, iters, src, (16u << 20) - 1);      // L2 / Infinity-Cache hot window
            if (k == 5) hipLaunchKernelGGL((probe<12, 4>), dim3(256), dim3(512), 65536, 0, out, iters, src, (1u << 31) - 1);       // 2 GB window: HBM
            if (k == 6) hipLaunchKernelGGL((probe<12, 2>), dim3(256), dim3(512), 65536, 0, out, iters, src, (1u << 31) - 1);       // half the fill
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const char *names[7] = {"0 reads", "8 reads", "12 reads", "16 reads", "12 reads + 4 DMA (16 MB window)", "12 reads + 4 DMA (2 GB window)", "12 reads + 2 DMA (2 GB window)"};
            printf("%-34s %.3f ms -> %.0f TFLOP/s\n", names[k], ms, 256.0 * 8 * iters * 32 * (2.0 * 16 * 16 * 32) / ms / 1e9);
        }
    return 0;
}
