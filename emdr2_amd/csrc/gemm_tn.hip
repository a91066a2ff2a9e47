// emdr2_amd/csrc/gemm_tn.hip -- bf16 MFMA GEMM in "TN" form for weight gradients (include/emdr2_ops.h: emdr2_gemm_tn_bf16).
//
//   C[i, j] (fp32) = sum_r A[r, i] * B[r, j]          A [R, I], B [R, J] row-major: dW[n, k] = sum_tokens dy[token, n] * x[token, k]
//
// Both operands are stored with the REDUCTION index as the slow dimension (the layout the forward activations and their gradients
// already have), while an MFMA fragment wants 8 consecutive reduction elements per lane.  Instead of transposing dy and x through HBM
// (two extra read+write passes per linear layer, 11 % of the round-1 training step), tiles are DMA'd as they are ([32 r][256 cols] per
// stage, 16-B granules XOR-swizzled by (r & 3) << 2) and the fragments are gathered with the LDS transpose read of gfx950:
// ds_read_b64_tr_b16 hands lane t of each 16-lane group column t of a [4 r][16 cols] block whose four rows are addressed by lanes
// 4j..4j+3 (layout pinned by tools/tr_probe.hip).  Two such reads build the 8-element k-run of one 32x32x16 operand fragment.
// The swizzle makes every read conflict-free: a 32-lane service group touches 4 rows x 4 granules = 16 distinct 16-B slots.
// Reference op: the autograd weight gradient of F.linear (mpu/layers.py:255,353) = dy^T x.
// Optional fused bias gradient: colsum[i] += sum_r A[r, i] (the fp32 column sums of dy), taken from the A fragments already in
// registers by the workgroups of the first j-tile.  split_k > 1: reduction slices accumulate with fp32 atomics into a pre-zeroed C.
#include "../../include/emdr2_ops.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "ops_timing.h"
#include "exp_hooks.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

int emdr2_gemm8t_try(const void *A, int64_t lda, const void *B, int64_t ldb, float *C, int64_t ldc, int I, int J, int R, int split_k,
                     float *colsum_a, hipStream_t stream);       // gemm8t.hip

namespace {

struct TnParams {
    const char *A, *B;
    float *C, *colsum;
    long long lda, ldb, ldc;
    int I, J, R, splitk;
    int tiles_i, tiles_j;
};

#define TN_STAGES 3
#define TN_OPER 16384                       // one operand stage: 32 reduction rows x 256 columns x 2 B
#define TN_STAGE (2 * TN_OPER)

#define TR_READ(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

__global__ void __launch_bounds__(512) gemm_tn_kernel(TnParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int t = lane & 15, colgrp = (lane >> 4) & 1;
    // 1-D grid, XCD-aware: workgroup ids go round-robin to the 8 XCDs (one L2 each); every XCD gets a contiguous range of
    // (reduction slice, tile) pairs with the tile index fastest, so the tiles_i * tiles_j workgroups that stream the SAME token rows of
    // dy and x run together on one L2 and both operands leave HBM once per slice.
    const int tiles = p.tiles_i * p.tiles_j, total = tiles * p.splitk;
    const int per_xcd = (total + 7) >> 3;
    const int t_id = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (t_id >= total) return;
    const int zs = t_id / tiles, tile = t_id - zs * tiles;
    const int tj = tile / p.tiles_i, ti = tile - tj * p.tiles_i;
    const int i0 = ti * 256, j0 = tj * 256;

    // fragment addresses: lane -> row (t >> 2) of its group's [4][16] block, column segment (t & 3) * 4
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    uint32_t a_addr[2], b_addr[4];
    const int rbase = 8 * hi + (t >> 2), sw = (t >> 2) << 2;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int col = wm * 64 + mi * 32 + colgrp * 16 + (t & 3) * 4;
        a_addr[mi] = lds0 + rbase * 512 + ((((col >> 3) ^ sw)) << 4) + (col & 7) * 2;
    }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int col = wn * 128 + ni * 32 + colgrp * 16 + (t & 3) * 4;
        b_addr[ni] = lds0 + TN_OPER + rbase * 512 + ((((col >> 3) ^ sw)) << 4) + (col & 7) * 2;
    }

    // LDS-DMA: an operand stage is 16 pieces of 1 KiB; wave w moves pieces 2w, 2w+1 of A and of B.
    // LDS slot sl = piece * 64 + lane -> (r = sl >> 5, granule slot gs = sl & 31); source granule = gs ^ ((r & 3) << 2)
    const char *a_src[2], *b_src[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int sl = (wave * 2 + j) * 64 + lane;
        const int r = sl >> 5, g = (sl & 31) ^ ((r & 3) << 2);
        int ca = i0 + g * 8; if (ca > p.I - 8) ca = p.I - 8;          // overhang columns re-read the last granule (never stored)
        int cb = j0 + g * 8; if (cb > p.J - 8) cb = p.J - 8;
        a_src[j] = p.A + ((long long)r * p.lda + ca) * 2;
        b_src[j] = p.B + ((long long)r * p.ldb + cb) * 2;
    }
    const int nch_all = p.R >> 5;
    const int per = (nch_all + p.splitk - 1) / p.splitk;
    const int c_begin = zs * per;
    const int nch = (c_begin + per <= nch_all ? per : (nch_all > c_begin ? nch_all - c_begin : 0));
    if (nch == 0) return;
    const long long a_step = p.lda * 64, b_step = p.ldb * 64;          // 32 rows in bytes
    int pf_c = 0, pf_stage = 0;
    auto issue = [&]() {
        const long long c = c_begin + (pf_c < nch ? pf_c : nch - 1);    // past the end: harmless re-read, keeps the vmcnt arithmetic fixed
        char *sb = smem + pf_stage * TN_STAGE;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t *)(a_src[j] + c * a_step), (lptr_t *)(sb + (wave * 2 + j) * 1024), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t *)(b_src[j] + c * b_step), (lptr_t *)(sb + TN_OPER + (wave * 2 + j) * 1024), 16, 0, 0);
        ++pf_c;
        pf_stage = (pf_stage == TN_STAGES - 1) ? 0 : pf_stage + 1;
    };

    floatx16 acc[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    float csum[2] = {0.f, 0.f};
    const bool want_colsum = p.colsum && tj == 0 && wn == 0;

    issue();
    issue();
    int cs = 0;
    for (int c = 0; c < nch; ++c) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue();
        const uint32_t so = cs * TN_STAGE;
        cs = (cs == TN_STAGES - 1) ? 0 : cs + 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint2 al[2], ah[2], bl[4], bh[4];
            if (ks == 0) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) { TR_READ(al[mi], a_addr[mi] + so, 0); TR_READ(ah[mi], a_addr[mi] + so, 4 * 512); }
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) { TR_READ(bl[ni], b_addr[ni] + so, 0); TR_READ(bh[ni], b_addr[ni] + so, 4 * 512); }
            } else {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) { TR_READ(al[mi], a_addr[mi] + so, 16 * 512); TR_READ(ah[mi], a_addr[mi] + so, 20 * 512); }
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) { TR_READ(bl[ni], b_addr[ni] + so, 16 * 512); TR_READ(bh[ni], b_addr[ni] + so, 20 * 512); }
            }
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(al[0]), "+v"(al[1]), "+v"(ah[0]), "+v"(ah[1]), "+v"(bl[0]), "+v"(bl[1]), "+v"(bl[2]), "+v"(bl[3]), "+v"(bh[0]),
                           "+v"(bh[1]), "+v"(bh[2]), "+v"(bh[3])
                         :
                         : "memory");
            bf16x8 a[2], b[4];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) a[mi] = __builtin_bit_cast(bf16x8, make_uint4(al[mi].x, al[mi].y, ah[mi].x, ah[mi].y));
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) b[ni] = __builtin_bit_cast(bf16x8, make_uint4(bl[ni].x, bl[ni].y, bh[ni].x, bh[ni].y));
            if (want_colsum) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    csum[mi] += (bf_lo(al[mi].x) + bf_hi(al[mi].x)) + (bf_lo(al[mi].y) + bf_hi(al[mi].y)) + (bf_lo(ah[mi].x) + bf_hi(ah[mi].x)) +
                                (bf_lo(ah[mi].y) + bf_hi(ah[mi].y));
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the two speculative chunks

    if (want_colsum) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const float tot = csum[mi] + __shfl_xor(csum[mi], 32);
            const int i = i0 + wm * 64 + mi * 32 + l31;
            if (hi == 0 && i < p.I) atomicAdd(&p.colsum[i], tot);
        }
    }
    // C layout of the 32x32 MFMA: column j = lane&31, row i = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int irow0 = i0 + wm * 64 + 4 * hi;
    const int jcol0 = j0 + wn * 128 + l31;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int j = jcol0 + ni * 32;
        if (j >= p.J) continue;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = irow0 + mi * 32 + (r & 3) + 8 * (r >> 2);
                if (i >= p.I) continue;
                float *dst = p.C + (long long)i * p.ldc + j;
                if (p.splitk > 1) atomicAdd(dst, acc[mi][ni][r]);
                else *dst = acc[mi][ni][r];
            }
    }
}

} // namespace

extern "C" int emdr2_gemm_tn_bf16(const void *A, int64_t lda, const void *B, int64_t ldb, float *C, int64_t ldc, int I, int J, int R, int split_k,
                                  float *colsum_a, void *stream)
{
    if (!A || !B || !C || I < 8 || J < 8 || R < 32 || (R & 31) || split_k < 1) return -1;
    if ((I & 7) || (J & 7) || (lda & 7) || (ldb & 7) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return -1;
    OpsTimer timer(OPS_GEMM_TN, 2.0 * I * (double)J * R, (hipStream_t)stream);
    bool general_only = false;
    EXP_TN_GENERAL_ONLY(general_only)
    if (!general_only) {
        const int rc = emdr2_gemm8t_try(A, lda, B, ldb, C, ldc, I, J, R, split_k, colsum_a, (hipStream_t)stream);
        if (rc != -4) return rc;
    }
    TnParams p;
    p.A = (const char *)A; p.B = (const char *)B; p.C = C; p.colsum = colsum_a;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.I = I; p.J = J; p.R = R; p.splitk = split_k;
    constexpr int LDS = TN_STAGES * TN_STAGE;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void *)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -3;
        attr_done = true;
    }
    p.tiles_i = (I + 255) / 256; p.tiles_j = (J + 255) / 256;
    dim3 grid((unsigned)(((p.tiles_i * p.tiles_j * split_k + 7) / 8) * 8));
    hipLaunchKernelGGL(gemm_tn_kernel, grid, dim3(512), LDS, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
