// Which SIMD does each wave of a 512-thread workgroup land on?  (HW_REG_HW_ID: wave_id[3:0] simd_id[5:4] cu_id[11:8])
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void __launch_bounds__(512) probe(unsigned *out)
{
    extern __shared__ char smem[];
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (15 << 11)); // HW_ID bits 15:0
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hw;
    smem[threadIdx.x] = 1;
}
int main()
{
    unsigned *out; hipMalloc(&out, 64 * 8 * 4);
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(probe, dim3(64), dim3(512), 150 * 1024, 0, out);
    unsigned h[64 * 8]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 6; ++b) {
        printf("block %d:", b);
        for (int w = 0; w < 8; ++w) printf("  w%d: simd %u wave_slot %u cu %u", w, (h[b * 8 + w] >> 4) & 3, h[b * 8 + w] & 15, (h[b * 8 + w] >> 8) & 15);
        printf("\n");
    }
    return 0;
}
