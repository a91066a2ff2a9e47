// What a ONE-KERNEL attention backward (5 matrix products: S and dP built once) would pay for its dQ: key-parallel
// workgroups (one per 128 keys of a (sequence, head), as attention_bwd_dkv_kernel) each hold dK / dV of their keys and must ADD their
// contribution to dQ of every query of the sequence -- fp32 atomics into a [rows, heads * 64] buffer, one add per (query, d) and key block.
// This probe issues exactly that traffic and nothing else, at the reader encoder's shape of one question group (327,680 packed rows, 12
// heads, sequences of 400 tokens = 3.125 key blocks -> every dQ element receives `adds` contributions), so the fused kernel's floor can be
// put next to what it would save (the second build of S / dP and of the softmax + dropout bits: 2 of today's 7 products).
//   dq_atomic_f32   a wave adds a [32 queries x 64 d] fp32 tile per step with global_atomic_add_f32 (no return), queries strided by 768 floats
//   dq_store_f32    the same tiles plainly stored (what "per-key-range partials + a combine kernel" would write, before its extra read pass)
//   dq_store_bf16   the same rows written ONCE as bf16 (what attention_bwd_dq_kernel writes today)
// build: hipcc -O3 -munsafe-fp-atomics --offload-arch=gfx950 tools/atomic_dq_probe.hip -o tools/atomic_dq_probe ; run: tools/atomic_dq_probe [adds]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define ROWS 327680
#define HEADS 12
#define SEQ 400              // tokens per sequence (reader encoder mean: 402)
#define H (HEADS * 64)

// workgroup = (sequence, head, key block): 4 waves, each walks the sequence's queries in 32-query steps (wave w takes steps w, w + 4, ..)
template <int MODE>
__global__ void __launch_bounds__(256) dq_kernel(float *dq, uint16_t *dq16, int nseq, int kblocks)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int kb = blockIdx.x % kblocks, head = (blockIdx.x / kblocks) % HEADS, seq = blockIdx.x / (kblocks * HEADS);
    if (seq >= nseq) return;
    const int l31 = lane & 31, hi = lane >> 5;
    for (int q0 = wave * 32; q0 < SEQ; q0 += 128) {
        const int q = q0 + l31;
        if (q >= SEQ) continue;
        const size_t row = (size_t)seq * SEQ + q;
        // the dq accumulator layout of attention_bwd_dq_kernel: lane (query l31, half hi) holds d = j * 32 + 8 g + 4 hi .. + 3 for j < 2, g < 4
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = j * 32 + 8 * g + 4 * hi;
                float *p = dq + row * H + head * 64 + d;
                const float v = (float)(kb + 1) * 0.001f + (float)d;
                if (MODE == 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) atomicAdd(p + e, v);               // global_atomic_add_f32, no return
                } else if (MODE == 1) {
                    *(float4 *)(p + (size_t)kb * 0) = make_float4(v, v, v, v);       // (same address per key block: the stores of a real partial go to [kb][row])
                } else if (kb == 0) {
                    *(uint2 *)(dq16 + row * H + head * 64 + d) = make_uint2(0x3f803f80u, 0x3f803f80u);
                }
            }
    }
}

int main(int argc, char **argv)
{
    const int nseq = ROWS / SEQ;                                                    // 819 sequences
    const int adds = argc > 1 ? atoi(argv[1]) : 3;
    float *dq; uint16_t *dq16;
    hipMalloc(&dq, (size_t)ROWS * H * 4 * (size_t)(adds > 1 ? 1 : 1)); hipMalloc(&dq16, (size_t)ROWS * H * 2);
    hipMemset(dq, 0, (size_t)ROWS * H * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const char *names[3] = {"dq_atomic_f32", "dq_store_f32", "dq_store_bf16"};
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(a);
            const dim3 grid((unsigned)(nseq * HEADS * adds));
            if (mode == 0) hipLaunchKernelGGL(dq_kernel<0>, grid, dim3(256), 0, 0, dq, dq16, nseq, adds);
            if (mode == 1) hipLaunchKernelGGL(dq_kernel<1>, grid, dim3(256), 0, 0, dq, dq16, nseq, adds);
            if (mode == 2) hipLaunchKernelGGL(dq_kernel<2>, grid, dim3(256), 0, 0, dq, dq16, nseq, adds);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        const double bytes = mode == 2 ? (double)nseq * SEQ * H * 2 : (double)nseq * SEQ * H * 4 * adds;
        printf("%-14s adds/element %d: %.3f ms  (%.2f GB issued, %.0f GB/s)\n", names[mode], mode == 2 ? 1 : adds, best, bytes / 1e9, bytes / best / 1e6);
    }
    return 0;
}
