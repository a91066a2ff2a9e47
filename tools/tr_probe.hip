// Probe of ds_read_b64_tr_b16 on gfx950: every LDS bf16 element holds its own element index; each lane passes an address and we print what
// comes back, to pin the lane/element mapping before building operand fragments on it.  hipcc --offload-arch=gfx950 tools/tr_probe.hip -o tools/tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(uint16_t *out, int mode)
{
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x;
    // mode 0: lane l points at element 4*l (contiguous 8-B granules).  mode 1: [4][16] blocks per 16-lane group, row pitch 64 elements
    int elem = mode == 0 ? 4 * l : ((l & 15) >> 2) * 64 + (l & 3) * 4 + (l >> 4) * 16;
    uint32_t addr = (uint32_t)(uintptr_t)(lds) + elem * 2;
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[l * 4 + 0] = v.x & 0xffff; out[l * 4 + 1] = v.x >> 16; out[l * 4 + 2] = v.y & 0xffff; out[l * 4 + 3] = v.y >> 16;
}
int main()
{
    uint16_t *d, h[256];
    hipMalloc(&d, 512);
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    }
    return 0;
}
