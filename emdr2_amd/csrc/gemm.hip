// emdr2_amd/csrc/gemm.hip -- bf16 MFMA GEMM, "NT" form, with fused epilogues (include/emdr2_ops.h).
//
//   C[m, n] = epilogue( alpha * sum_k A[m, k] * B[n, k] )         A: [M, K] row-major (lda), B: [N, K] row-major (ldb)
//
// This is the shape of every linear layer of the reference (F.linear: weight is [out, in],
// megatron/mpu/layers.py:255,353), of QK^T (transformer.py:309-312) and - with pre-transposed operands - of every
// backward GEMM.  Same engine as the index scan (mips_scan.hip): 8 waves, wave tile 64 x 128 = 2x4
// v_mfma_f32_32x32x16_bf16 accumulators, operands staged through a 3-deep LDS ring by LDS-DMA in 16-row x 64-B
// pieces whose 16-B groups are XOR-swizzled on the SOURCE address (conflict-free ds_read_b128), counted vmcnt.
// Epilogue: alpha, + bias[n], exact-erf GELU (optionally also storing the pre-activation for the backward),
// + residual[m, n], bf16 or fp32 output.  Two batch levels with independent strides cover [batch, head] attention GEMMs.
#include "../../include/emdr2_ops.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "gemm_common.h"
#include "exp_hooks.h"
#include "ops_timing.h"

#define GNST 3

int emdr2_gemm8_try(const void *A, int64_t lda, const void *B, int64_t ldb, void *C, int64_t ldc, int M, int N, int K, float alpha,
                    const float *bias, int gelu, void *pre_act, const void *residual, int residual_mode, float drop_p, uint32_t seed,
                    hipStream_t stream);   // gemm8.hip

struct GemmParams {
    const char *A, *B;
    char *C, *C2;             // C2: optional pre-activation output (bf16)
    const float *bias;
    const char *R;            // optional residual (bf16, same indexing as C)
    int rmode;                // 0: + R;  1: * gelu'(R) -- R is the saved GELU pre-activation (backward of the FFN's first linear, fused)
    long long lda, ldb, ldc;  // in elements
    long long sA1, sB1, sC1, sA2, sB2, sC2; // batch strides in elements
    int M, N, K, batch2;
    float alpha;
    int gelu, out_f32;
    float drop_p;             // dropout on (acc*alpha + bias [gelu]) before the residual add: bias_dropout_add (transformer.py:397-413)
    uint32_t seed;            // keep bit = emdr2_keep(row_hash(seed, m), n, thr)
    int ablate;               // -DEMDR2_EXPERIMENTS builds only (EMDR2_GEMM_ABLATE): 1 = no epilogue, 2 = no k-loop, 3 = epilogue without global stores
    int ngroup;               // n-tiles per L2-resident group of B panels (order >= 1)
    int tiles_m, tiles_n, order; // order 1: 1-D grid, n-tiles fastest inside a per-XCD contiguous tile range (operand A read once from HBM)
    int splitk;               // > 1: blockIdx.z also enumerates K slices; fp32 output accumulated with atomics (C pre-zeroed)
};

// VEC: bf16 output staged through LDS per wave and written as 16-byte row segments (bias / GELU / pre-activation / residual
// applied on 8-element vectors); otherwise (fp32 output, split-K atomics, unaligned leading dimensions) the scalar epilogue.
template <int WM, int WN, bool VEC, int MI = 2>
__global__ void __launch_bounds__(WM * WN * 64) gemm_nt_kernel(GemmParams p)
{
    constexpr int BM = WM * 32 * MI, BN = WN * 128;                   // a wave owns (32 MI) x 128 outputs: MI = 2 -> 64 x 128, MI = 4 -> 128 x 128
    constexpr int A_STAGE = BM * 64, B_STAGE = BN * 64, STAGE = A_STAGE + B_STAGE;
    constexpr int NW = WM * WN;                                       // waves per workgroup: each owns a 64 x 128 output sub-tile
    constexpr int A_PW = BM / (16 * NW), B_PW = BN / (16 * NW), PPW = A_PW + B_PW;   // 1-KiB DMA pieces per wave and stage
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hi = lane >> 5;
    const int swz = (l31 >> 2) & 3;

    int tm = blockIdx.x, tn = blockIdx.y;
    if (p.order >= 1) {
        // Workgroup ids are dealt round-robin to the 8 XCDs, each with its own 4 MB L2.  Give every XCD one contiguous range of the tile
        // sequence, and order the sequence so that what is shared stays in that L2: n-tiles are taken in groups of `ngroup` B panels
        // that fit the L2 (a 256 x K panel of the weight matrix each), and inside a group the walk is n-fastest, so the tiles sharing an
        // A panel run back to back.  A then leaves HBM once per group (N = 3072, K = 768: twice instead of 12 times) and B stays
        // L2-resident instead of cycling through a working set larger than the cache.
        const int total = p.tiles_m * p.tiles_n, id = blockIdx.x;
        const int per = (total + 7) >> 3;
        int t = (id & 7) * per + (id >> 3);
        if (t >= total) return;                                        // ragged tail of the last XCD ranges (grid is padded to 8 * per)
        const int full = p.ngroup * p.tiles_m;                         // tiles in one full n-group
        const int g = t / full, r = t - g * full;
        const int gsize = (g + 1) * p.ngroup <= p.tiles_n ? p.ngroup : p.tiles_n - g * p.ngroup;   // the last group may be narrower
        tm = r / gsize; tn = g * p.ngroup + (r - tm * gsize);
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int zb = blockIdx.z / p.splitk, zs = blockIdx.z % p.splitk;
    const int b1 = zb / p.batch2, b2 = zb % p.batch2;
    const char *A = p.A + ((long long)b1 * p.sA1 + (long long)b2 * p.sA2) * 2;
    const char *B = p.B + ((long long)b1 * p.sB1 + (long long)b2 * p.sB2) * 2;
    const long long coff = (long long)b1 * p.sC1 + (long long)b2 * p.sC2;

    int a_off[2], b_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int sp = ((ks * 2 + hi) ^ swz) << 4;
        a_off[ks] = (wm * 32 * MI + l31) * 64 + sp;
        b_off[ks] = A_STAGE + (wn * 128 + l31) * 64 + sp;
    }

    // LDS-DMA source addresses: piece = 16 rows x 64 B; lane -> (row = lane>>2, LDS slot = lane&3), source group = slot ^ swizzle(row)
    const int prow = lane >> 2, pslot = lane & 3;
    const char *a_src[A_PW];
    const char *b_src[B_PW];
#pragma unroll
    for (int j = 0; j < A_PW; ++j) {
        const int r = (wave + NW * j) * 16 + prow;                  // row inside the tile
        int gm = m0 + r; if (gm >= p.M) gm = p.M - 1;               // overhang rows re-read the last row (never stored)
        a_src[j] = A + ((long long)gm * p.lda + ((pslot ^ ((r >> 2) & 3)) << 3)) * 2;
    }
#pragma unroll
    for (int j = 0; j < B_PW; ++j) {
        const int r = (wave + NW * j) * 16 + prow;
        int gn = n0 + r; if (gn >= p.N) gn = p.N - 1;
        b_src[j] = B + ((long long)gn * p.ldb + ((pslot ^ ((r >> 2) & 3)) << 3)) * 2;
    }

    const int nch_all = p.K >> 5;
    const int per = (nch_all + p.splitk - 1) / p.splitk;
    const int c_begin = zs * per;
    const int nch = (c_begin + per <= nch_all ? per : (nch_all > c_begin ? nch_all - c_begin : 0));
    if (nch == 0) return;
    const int nch_run = EXP_GEMM_KLOOP_LEN(p, nch);
    int pf_c = 0, pf_stage = 0;
    auto issue = [&]() {
        const int c = c_begin + (pf_c < nch ? pf_c : nch - 1);      // past the end: harmless re-read, keeps the vmcnt arithmetic fixed
        char *sb = smem + pf_stage * STAGE;
#pragma unroll
        for (int j = 0; j < A_PW; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t *)(a_src[j] + c * 64), (lptr_t *)(sb + (wave + NW * j) * 1024), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < B_PW; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t *)(b_src[j] + c * 64), (lptr_t *)(sb + A_STAGE + (wave + NW * j) * 1024), 16, 0, 0);
        ++pf_c;
        pf_stage = (pf_stage == GNST - 1) ? 0 : pf_stage + 1;
    };

    floatx16 acc[MI][4];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    issue();
    issue();
    int cs = 0;
    for (int c = 0; c < nch_run; ++c) {
        if (PPW == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (PPW == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (PPW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue();
        const char *sb = smem + cs * STAGE;
        cs = (cs == GNST - 1) ? 0 : cs + 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[MI], b[4];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) a[mi] = *(const bf16x8 *)(sb + a_off[ks] + mi * 2048);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) b[ni] = *(const bf16x8 *)(sb + b_off[ks] + ni * 2048);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the two speculative chunks

    EXP_GEMM_AFTER_KLOOP(p, acc)
    if (VEC) {
        // ---- vector epilogue: acc -> LDS (fp32, wave-private 32 x 128 half tile, pitch 132) -> 8-wide row segments ----------
        __syncthreads();                                           // every wave is done with the operand ring
        float *wl = (float *)smem + wave * (32 * 132);
        const int seg = lane & 15, rsub = lane >> 4;               // 16 lanes cover one 128-column row, 4 rows per pass
        const int ncol = n0 + wn * 128 + seg * 8;
        float bv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) bv[j] = (p.bias && ncol + j < p.N) ? p.bias[ncol + j] : 0.f;
        const bool affine = p.bias != nullptr || p.alpha != 1.0f;       // plain GEMMs skip the per-element multiply-add
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            // residual rows of this half: all eight 16-byte loads go out BEFORE the LDS round trip, so their HBM latency overlaps the
            // staging instead of serialising one load per 4-row step (that chain cost more than the whole k-loop at K = 768)
            uint4 rr[8];
            if (p.R) {
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const int m = m0 + wm * 32 * MI + mi * 32 + g * 4 + rsub;
                    rr[g] = make_uint4(0, 0, 0, 0);
                    if (m < p.M && ncol < p.N) {
                        const u32x4_t t = __builtin_nontemporal_load((const u32x4_t *)((const uint16_t *)p.R + coff + (long long)m * p.ldc + ncol));
                        rr[g] = make_uint4(t.x, t.y, t.z, t.w);
                    }
                }
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    wl[((r & 3) + 8 * (r >> 2) + 4 * hi) * 132 + ni * 32 + l31] = acc[mi][ni][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int rl = g * 4 + rsub;
                const int m = m0 + wm * 32 * MI + mi * 32 + rl;
                const float4 lo = *(const float4 *)(wl + rl * 132 + seg * 8), hi4 = *(const float4 *)(wl + rl * 132 + seg * 8 + 4);
                float v[8] = {lo.x, lo.y, lo.z, lo.w, hi4.x, hi4.y, hi4.z, hi4.w};
                if (m < p.M && ncol < p.N) {
                    const long long o = coff + (long long)m * p.ldc + ncol;
                    if (affine) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = v[j] * p.alpha + bv[j];
                    }
                    if (p.gelu == 2) {                                   // activation + its derivative (the derivative goes to C2)
                        float d[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = gelu_erf_with_grad(v[j], d[j]);
                        uint32_t w[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) w[j] = pack2_bf16(d[2 * j], d[2 * j + 1]);
                        store_stream((uint16_t *)p.C2 + o, make_uint4(w[0], w[1], w[2], w[3]));
                    } else {
                        if (p.C2) {
                            uint32_t w[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) w[j] = pack2_bf16(v[2 * j], v[2 * j + 1]);
                            store_stream((uint16_t *)p.C2 + o, make_uint4(w[0], w[1], w[2], w[3]));
                        }
                        if (p.gelu) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) v[j] = gelu_erf(v[j]);
                        }
                    }
                    if (p.drop_p > 0.f) {
                        const float ik = emdr2_keep_scale(p.drop_p);
                        const uint32_t rh = emdr2_row_hash(p.seed, (unsigned long long)m), thr = emdr2_drop_thr(p.drop_p);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const uint32_t bits = emdr2_pair_bits(rh, (uint32_t)(ncol + 2 * j));
                            v[2 * j] = (bits & 0xffffu) >= thr ? v[2 * j] * ik : 0.f;
                            v[2 * j + 1] = (bits >> 16) >= thr ? v[2 * j + 1] * ik : 0.f;
                        }
                    }
                    if (p.R) {
                        const uint32_t rw[4] = {rr[g].x, rr[g].y, rr[g].z, rr[g].w};
                        // the value meets the residual as the bf16 the reference's F.linear would have stored (and as gemm8.hip, whose staging
                        // buffer holds bf16, sees it): one rounding before the add, one after -- the same in every GEMM kernel of the library
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = bf16_to_f32(f32_to_bf16(v[j]));
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float r0 = bf16_to_f32((uint16_t)(rw[j] & 0xffff)), r1 = bf16_to_f32((uint16_t)(rw[j] >> 16));
                            if (p.rmode == 0) { v[2 * j] += r0; v[2 * j + 1] += r1; }
                            else if (p.rmode == 2) { v[2 * j] *= r0; v[2 * j + 1] *= r1; }
                            else { v[2 * j] *= gelu_erf_grad(r0); v[2 * j + 1] *= gelu_erf_grad(r1); }
                        }
                    }
                    uint32_t w[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) w[j] = pack2_bf16(v[2 * j], v[2 * j + 1]);
                    EXP_GEMM_STORE_IF(p, w)
                    store_stream((uint16_t *)p.C + o, make_uint4(w[0], w[1], w[2], w[3]));
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads of this half are done before the next half overwrites
        }
        return;
    }

    // epilogue.  C layout of the 32x32 MFMA: column n = lane&31, row m = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int mrow0 = m0 + wm * 32 * MI + 4 * hi;
    const int ncol0 = n0 + wn * 128 + l31;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int n = ncol0 + ni * 32;
        if (n >= p.N) continue;
        const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mrow0 + mi * 32 + (r & 3) + 8 * (r >> 2);
                if (m >= p.M) continue;
                const long long o = coff + (long long)m * p.ldc + n;
                float v = acc[mi][ni][r] * p.alpha + bias;
                if (p.gelu == 2) { float d; v = gelu_erf_with_grad(v, d); ((uint16_t *)p.C2)[o] = f32_to_bf16(d); }
                else {
                    if (p.C2) ((uint16_t *)p.C2)[o] = f32_to_bf16(v);
                    if (p.gelu) v = gelu_erf(v);
                }
                if (p.drop_p > 0.f) v = emdr2_keep(emdr2_row_hash(p.seed, (unsigned long long)m), (uint32_t)n, emdr2_drop_thr(p.drop_p)) ? v * emdr2_keep_scale(p.drop_p) : 0.f;
                if (p.R) {
                    const float rv = bf16_to_f32(((const uint16_t *)p.R)[o]);
                    if (!p.out_f32) v = bf16_to_f32(f32_to_bf16(v));          // as above: bf16 before it meets the residual
                    v = p.rmode == 0 ? v + rv : (p.rmode == 2 ? v * rv : v * gelu_erf_grad(rv));
                }
                if (p.splitk > 1) atomicAdd(&((float *)p.C)[o], v);
                else if (p.out_f32) ((float *)p.C)[o] = v;
                else ((uint16_t *)p.C)[o] = f32_to_bf16(v);
            }
        }
    }
}

template <int WM, int WN, bool VEC, int MI = 2>
static int launch_gemm_v(const GemmParams &p, int batch, hipStream_t stream)
{
    constexpr int BM = WM * 32 * MI, BN = WN * 128;
    constexpr int RING = GNST * (BM + BN) * 64, STAGING = WM * WN * 32 * 132 * 4;
    constexpr int LDS = (VEC && STAGING > RING) ? STAGING : RING;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void *)gemm_nt_kernel<WM, WN, VEC, MI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -3;
        attr_done = true;
    }
    GemmParams q = p;
    q.tiles_m = (p.M + BM - 1) / BM; q.tiles_n = (p.N + BN - 1) / BN;
    EXP_GEMM_TILE_ORDER(order_env, ablate_env, l2_env)
    q.ablate = ablate_env;
    q.order = (order_env >= 1 && q.tiles_n > 1 && q.tiles_m > 8) ? 1 : 0;
    // B panels of BN x K bf16 that fit about half of a 4 MB L2 (the rest holds the A panels in flight and the output lines in transit)
    const long long panel = (long long)BN * p.K * 2;
    int ng = (int)((long long)l2_env * 1024 / (panel > 0 ? panel : 1));
    if (ng < 1) ng = 1;
    if (ng > q.tiles_n || order_env == 2 || 2 * ng < q.tiles_n || (long long)p.N * p.K * 2 <= (4ll << 20)) ng = q.tiles_n;   // one group = plain n-fastest: EMDR2_GEMM_ORDER=2, or B panels so
                                                                                  // large (K = 3072) that grouping would re-read A 3+ times, or a B that fits the 4 MB L2 whole
                                                                                  // (N = 2304, K = 768: measured 1.7 % better ungrouped; N = 3072: 2.3 % better grouped)
    // balance the groups: e.g. 9 n-tiles with room for 6 -> 5 + 4 rather than 6 + 3
    const int groups = (q.tiles_n + ng - 1) / ng;
    q.ngroup = (q.tiles_n + groups - 1) / groups;
    dim3 grid(q.tiles_m, q.tiles_n, batch * p.splitk);
    if (q.order == 1) grid = dim3(((q.tiles_m * q.tiles_n + 7) / 8) * 8, 1, batch * p.splitk);
    hipLaunchKernelGGL((gemm_nt_kernel<WM, WN, VEC, MI>), grid, dim3(WM * WN * 64), LDS, stream, q);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

template <int WM, int WN, int MI = 2>
static int launch_gemm(const GemmParams &p, int batch, hipStream_t stream)
{
    const bool vec = !p.out_f32 && p.splitk == 1 && !(p.ldc & 7) && !(p.N & 7) && !(p.sC1 & 7) && !(p.sC2 & 7) && !((uintptr_t)p.C & 15) &&
                     !((uintptr_t)p.C2 & 15) && !((uintptr_t)p.R & 15);
    return vec ? launch_gemm_v<WM, WN, true, MI>(p, batch, stream) : launch_gemm_v<WM, WN, false, MI>(p, batch, stream);
}

extern "C" int emdr2_gemm_nt_bf16(const void *A, int64_t lda, const void *B, int64_t ldb, void *C, int64_t ldc, int M, int N, int K,
                                  int batch1, int64_t sA1, int64_t sB1, int64_t sC1, int batch2, int64_t sA2, int64_t sB2, int64_t sC2,
                                  float alpha, const float *bias, int gelu, void *pre_act, const void *residual, int residual_mode,
                                  int out_f32, int split_k, float drop_p, uint32_t seed, void *stream)
{
    if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && (batch1 * batch2 != 1 || split_k > 1))) return -1;
    if (!A || !B || !C || M < 1 || N < 1 || K < 32 || (K & 31) || batch1 < 1 || batch2 < 1 || split_k < 1) return -1;
    if (split_k > 1 && (!out_f32 || bias || gelu || pre_act || residual)) return -1;
    if ((lda & 7) || (ldb & 7) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return -1;
    if ((sA1 & 7) || (sB1 & 7) || (sA2 & 7) || (sB2 & 7)) return -1;
    GemmParams p;
    p.A = (const char *)A; p.B = (const char *)B; p.C = (char *)C; p.C2 = (char *)pre_act;
    if (residual_mode < 0 || residual_mode > 2 || gelu < 0 || gelu > 2 || (gelu == 2 && !pre_act)) return -1;
    p.bias = bias; p.R = (const char *)residual; p.rmode = residual_mode;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.sA1 = sA1; p.sB1 = sB1; p.sC1 = sC1; p.sA2 = sA2; p.sB2 = sB2; p.sC2 = sC2;
    p.M = M; p.N = N; p.K = K; p.batch2 = batch2; p.alpha = alpha; p.gelu = gelu; p.out_f32 = out_f32; p.splitk = split_k; p.drop_p = drop_p; p.seed = seed;
    const int batch = batch1 * batch2;
    OpsTimer timer(OPS_GEMM_NT, 2.0 * M * (double)N * K * batch, (hipStream_t)stream);
    if (batch == 1 && split_k == 1 && !out_f32) {
        // large unbatched linears: the persistent 256 x 256 x 64 kernel (gemm8.hip); -4 = shape not covered there
        if (EXP_GEMM_USE_GEMM8()) {
            const int rc = emdr2_gemm8_try(A, lda, B, ldb, C, ldc, M, N, K, alpha, bias, gelu, pre_act, residual, residual_mode, drop_p, seed, (hipStream_t)stream);
            if (rc != -4) return rc;
        }
    }
    if (N <= 128) return launch_gemm<8, 1>(p, batch, (hipStream_t)stream);
    if (M <= 128) return launch_gemm<2, 4>(p, batch, (hipStream_t)stream);
    EXP_GEMM_TILE_VARIANTS(p, batch, split_k, stream)
    // few output tiles (the decoder's 2,048-row batches, decoding): a 256 x 256 tile per workgroup would leave most of the 256 CUs idle and
    // every workgroup with the whole K loop to itself -- 128 x 256 tiles (4 waves) while they still make ~128 workgroups, else 128 x 128
    // (2 waves).  M = 2,048: N = K = 768 26 -> 17 us, K = 3,072 77 -> 46 us, N = 3,072 29 -> 22 us.
    if (split_k == 1) {
        const long long t256 = (long long)batch * ((M + 255) / 256) * ((N + 255) / 256), t128 = (long long)batch * ((M + 127) / 128) * ((N + 255) / 256);
        if (t256 < 192) return t128 >= 128 ? launch_gemm<2, 2>(p, batch, (hipStream_t)stream) : launch_gemm<2, 1>(p, batch, (hipStream_t)stream);
    }
    return launch_gemm<4, 2>(p, batch, (hipStream_t)stream);
}
