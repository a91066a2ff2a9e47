// LDS bank-conflict probe for ds_read_b128 address patterns (rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE): one kernel per pattern,
// so the counters can be read per kernel name.  Patterns are the fragment reads of mips_scan8.hip over its 64-byte-row LDS image
// (16-byte groups XOR-swizzled with (row >> 2) & 3): P0 = the 32 x 32 x 16 form (row l31, group (2 j + hi) ^ swz), P1 = the first 16 x 16 x 32
// form (row l15, group lq ^ swz), P2.. = candidates.       hipcc -O3 --offload-arch=gfx950 tools/lds_conflict_probe.hip -o tools/lds_conflict_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float float4_t __attribute__((ext_vector_type(4)));
template <int P>
__global__ void __launch_bounds__(512) probe(float *out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16384; i += 512) ((float *)smem)[i] = (float)i;
    __syncthreads();
    const int l31 = lane & 31, hi = lane >> 5, l15 = lane & 15, lq = lane >> 4;
    int off;
    if (P == 0) off = l31 * 64 + (((0 * 2 + hi) ^ ((l31 >> 2) & 3)) << 4);
    if (P == 1) off = l15 * 64 + ((lq ^ ((l15 >> 2) & 3)) << 4);
    if (P == 2) { const int row = (l15 & 3) * 4 + (l15 >> 2); off = row * 64 + ((lq ^ ((row >> 2) & 3)) << 4); }            // rows transposed inside the tile
    if (P == 3) { const int row = 2 * l15 + (lq & 1); off = row * 64 + (((lq >> 1) * 2 ^ ((row >> 2) & 3)) << 4); }              // (not a legal operand; footprint test)
    if (P == 4) { const int row = l15 + 16 * (lq & 1); off = row * 64 + (((lq >> 1)) ^ ((row >> 2) & 3)) * 16; }                // 32 rows x 2 groups, 16x16 lanes
    if (P == 5) { const int row = (l15 >> 1) + 8 * (l15 & 1); off = row * 64 + ((lq ^ ((row >> 2) & 3)) << 4); }            // even / odd lanes 8 rows apart
    if (P == 6) { const int row = l15 ^ (lq << 2); off = row * 64 + ((lq ^ ((row >> 2) & 3)) << 4); }                        // row order rotated per k group
    off += wave * 2048;
    float4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        float4_t v;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(uintptr_t)smem + off) : "memory");
        acc += v;
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
int main()
{
    float *out; (void)hipMalloc(&out, 256 * 512 * 4);
#define RUN(P) hipLaunchKernelGGL(probe<P>, dim3(256), dim3(512), 65536, 0, out, 2000); (void)hipDeviceSynchronize();
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6)
    printf("done\n");
    return 0;
}
