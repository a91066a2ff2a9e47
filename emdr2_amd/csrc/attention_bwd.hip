// emdr2_amd/csrc/attention_bwd.hip -- fused attention backward for head dim 64 (include/emdr2_ops.h: emdr2_attention_bwd).
//
// Gradient of o = dropout(softmax(mask(q k^T scale))) v (reference: the autograd of transformer.py:283-381) without ever writing a
// [sq, sk] matrix.  The probabilities are rebuilt from the forward's row statistics (m, l) in both orientations:
//
//   dq kernel  (a wave owns 32 queries, streams 64-key blocks of K and V through LDS)
//       S^T = K Q^T, dP^T = V dO^T          MFMA A = K / V rows (ds_read_b128), B = Q / dO fragments held in registers
//       dS^T = P^T o (dP^T_eff - D)         D[q] = dO[q,:] . O[q,:] (computed here, written for the second kernel)
//       dQ^T += K^T dS^T                    MFMA A = K^T gathered from the SAME [key][d] tile with ds_read_b64_tr_b16, B = dS^T from the
//                                           accumulator registers (the key subset a lane owns is used as the k index on both sides)
//   dk/dv kernel (a wave owns 32 keys, streams 64-query blocks of Q and dO plus their row statistics through LDS)
//       S = Q K^T, dP = dO V^T              MFMA A = Q / dO rows, B = K / V fragments held in registers
//       dV^T += dO^T P_dropped, dK^T += Q^T dS     A = dO^T / Q^T by transpose reads of the same tiles
//
// Tiles are [64 rows][64 d] bf16 (128-B rows of eight 16-B granules), granule index XOR f(row) with
// f(row) = (((row >> 1) & 1) << 2) | ((row >> 2) & 3): conflict-free for the row reads (16 consecutive rows, one granule) and for the
// transpose reads (4 rows x 4 granules per 32-lane service group).  Masks from token ids (pad id 0) + optional history mask; masked
// positions get dS = 0 (masked_fill cuts the dependence on the score) while their (normally zero) probability still feeds dV, like
// the reference.  Dropout bits are regenerated from (seed, row, key) -- csrc/rng.h -- never stored.
#include "../../include/emdr2_ops.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "rng.h"

#include "attention_common.h"
#include "ops_timing.h"

namespace {

struct BwdParams {
    const char *q, *k, *v, *o, *dout;
    char *dq, *dk, *dv;
    const long long *ids_q, *ids_k;
    const float *m, *l;
    float *dstat;                  // per-query statistics the dq kernel leaves for the dk/dv kernel, [4][stat_n] (stat index as m / l):
                                   //   exp2 offset pm = m log2(e) + log2(l) | D (1 - p), D = rowsum(dO o O) | dropout row hash (bits) | 1 = real token (bits)
    long long stat_n;              // batch * heads * sq (dense) or heads * total_q (packed)
    long long q_sb, q_ss, q_sn;    // element strides of q (batch, sequence, head); k and v share k_*
    long long k_sb, k_ss, k_sn, v_sb, v_ss, v_sn;
    long long dq_sb, dq_ss, dkv_sb, dkv_ss;   // element strides (batch, sequence) of the gradient outputs; heads are 64 apart
    int heads, sq, sk, causal, batch;
    float scale, drop_p;
    uint32_t seed;
    float keep_scale, inv_keep_scale, scale2;      // 1 / (1 - p) of the surviving probabilities (and its inverse); scale * log2(e): host-computed -> SGPRs
    uint32_t drop_thr;
    // packed operands (attention.hip: AttnParams): sequence b owns rows [cu[b], cu[b+1]); statistics [heads, tq] when cu_q is given
    const int *cu_q, *cu_k;
    long long tq;
    // split keys (attention.hip: AttnParams): the dq kernel's key blocks dealt to `ksplit` workgroups, fp32 partial dQ [ksplit][stat_n][64]
    // folded by attention_dq_combine_kernel; the dk / dv kernel is parallel over keys already
    int ksplit;
    unsigned base_grid;
    float *part_dq;
};

// the sequence's extent and operand offsets, dense or packed (see attention.hip)
struct SeqExtent {
    int sq, sk;
    long long qrow0, krow0, q_off, k_off, v_off, dq_off, dkv_off, stat0;
};
__device__ __forceinline__ SeqExtent seq_extent(const BwdParams &p, int b, int n)
{
    SeqExtent e;
    e.sq = p.sq; e.sk = p.sk;
    e.qrow0 = (long long)b * p.sq; e.krow0 = (long long)b * p.sk;
    e.q_off = (long long)b * p.q_sb; e.k_off = (long long)b * p.k_sb; e.v_off = (long long)b * p.v_sb;
    e.dq_off = (long long)b * p.dq_sb; e.dkv_off = (long long)b * p.dkv_sb;
    e.stat0 = ((long long)b * p.heads + n) * p.sq;
    if (p.cu_q) {
        const int c0 = p.cu_q[b];
        e.sq = p.cu_q[b + 1] - c0; e.qrow0 = c0; e.q_off = (long long)c0 * p.q_ss; e.dq_off = (long long)c0 * p.dq_ss; e.stat0 = (long long)n * p.tq + c0;
    }
    if (p.cu_k) {
        const int c0 = p.cu_k[b];
        e.sk = p.cu_k[b + 1] - c0; e.krow0 = c0; e.k_off = (long long)c0 * p.k_ss; e.v_off = (long long)c0 * p.v_ss; e.dkv_off = (long long)c0 * p.dkv_ss;
    }
    return e;
}

// ============================================================ dq =====================================================================
// Like the forward (attention.hip): VALU-bound, so workgroups are FOUR waves (128 queries) at <= 168 VGPRs -- three per CU, each on its own
// barrier -- and a staged 64-key block is consumed as two 32-key steps (one S / dP accumulator pair live); DROP / CAUSAL are compile-time.
#define KSPLIT_BLOCKS 32       // key blocks per workgroup of a split-key launch (attention.hip)
#define BNW 4
#define KNW 4                  // waves (32 keys each) per workgroup of the dK / dV kernel
template <bool DROP, bool CAUSAL>
__global__ void __launch_bounds__(BNW * 64, 3) attention_bwd_dq_kernel(BwdParams p)
{
    // dynamic LDS: 3 stages x (K tile 8 KiB + V tile 8 KiB) + one key-mask word per 64-key block; two blocks of DMA in flight (attention.hip)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long *kmask_s = (unsigned long long *)(smem + 3 * 16384);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    int qblk, b, n;
    const int split = p.ksplit > 1 ? (int)(blockIdx.x / p.base_grid) : 0;
    if (!attn_decode(p.ksplit > 1 ? (int)(blockIdx.x - split * p.base_grid) : (int)blockIdx.x, (p.sq + BNW * 32 - 1) / (BNW * 32), p.batch * p.heads, p.heads, qblk, b, n)) return;
    const SeqExtent ex = seq_extent(p, b, n);
    const int sq = ex.sq, sk = ex.sk;
    if (qblk * (BNW * 32) >= sq) return;
    if (sk < 1) {
        // a key sequence without keys: the unsplit launch leaves dq untouched; a split launch's combine kernel sums EVERY split's slot of
        // the (uninitialised) partials workspace, so the slots of this workgroup's rows must hold zeros
        const int qz = qblk * (BNW * 32) + wave * 32 + l31;
        if (p.ksplit > 1 && qz < sq) {
            float4 *z = (float4 *)(p.part_dq + ((long long)split * p.stat_n + ex.stat0 + qz) * 64 + 32 * hi);
#pragma unroll
            for (int i = 0; i < 8; ++i) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    const int q0 = qblk * (BNW * 32) + wave * 32;
    const int qi = q0 + l31;
    const bool qvalid = qi < sq;
    const int qc = qvalid ? qi : sq - 1;
    const bool wave_live = q0 < sq;

    const int prow = wave * 8 + (lane >> 3), pslot = (lane & 7) ^ tile_swz(wave * 8 + (lane >> 3));
    const char *k_src = p.k + (ex.k_off + (long long)n * p.k_sn) * 2 + pslot * 16;
    const char *v_src = p.v + (ex.v_off + (long long)n * p.v_sn) * 2 + pslot * 16;
    const int nblk_all = (sk + 63) / 64;
    const int blk_lo = p.ksplit > 1 ? min(split * KSPLIT_BLOCKS, nblk_all) : 0;                      // this workgroup's key blocks [blk_lo, nblk)
    const int nblk = p.ksplit > 1 ? min((split + 1) * KSPLIT_BLOCKS, nblk_all) : nblk_all;
    auto issue = [&](int blk, int stage) {
        char *sb = smem + stage * 16384;
#pragma unroll
        for (int i = 0; i < 8 / BNW; ++i) {
            long long key = blk * 64 + prow + 8 * BNW * i;
            if (key >= sk) key = sk - 1;                                           // keys past sk: re-read the last row, masked below
            __builtin_amdgcn_global_load_lds((gptr_t *)(k_src + key * p.k_ss * 2), (lptr_t *)(sb + (wave + BNW * i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t *)(v_src + key * p.v_ss * 2), (lptr_t *)(sb + 8192 + (wave + BNW * i) * 1024), 16, 0, 0);
        }
    };
    if (blk_lo < nblk) issue(blk_lo, 0);    // before the fragment / statistics loads below: a short sequence must not pay the memory latency twice
    if (blk_lo + 1 < nblk) issue(blk_lo + 1, 1);

    bf16x8 qf[4], dof[4];
    load_row_frags(p.q + (ex.q_off + (long long)qc * p.q_ss + (long long)n * p.q_sn) * 2, hi, qf);
    const long long orow = ((ex.qrow0 + qc) * p.heads + n) * 64;
    load_row_frags(p.dout + orow * 2, hi, dof);
    const long long si = ex.stat0 + qc;
    float Dq;
    {   // D = dO . O over this row (two half-rows, one per half-wave)
        bf16x8 of[4];
        load_row_frags(p.o + orow * 2, hi, of);
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint4 a = __builtin_bit_cast(uint4, dof[t]), c = __builtin_bit_cast(uint4, of[t]);
            acc += bf_lo(a.x) * bf_lo(c.x) + bf_hi(a.x) * bf_hi(c.x) + bf_lo(a.y) * bf_lo(c.y) + bf_hi(a.y) * bf_hi(c.y) +
                   bf_lo(a.z) * bf_lo(c.z) + bf_hi(a.z) * bf_hi(c.z) + bf_lo(a.w) * bf_lo(c.w) + bf_hi(a.w) * bf_hi(c.w);
        }
        {
            const auto r_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc), __float_as_uint(acc), false, false);
            Dq = __uint_as_float(r_[0]) + __uint_as_float(r_[1]);
        }
    }
    const bool qpad = !qvalid || p.ids_q[ex.qrow0 + qc] == 0;
    const bool all_qpad = __builtin_amdgcn_ballot_w64(qpad) == ~0ull;  // a wave of padded queries: every dS is 0
    const float sc = p.scale2;
    const float pm = p.m[si] * L2E + __log2f(p.l[si]);                  // P = exp2(s2 - pm)

    for (int blk = blk_lo + wave; blk < nblk; blk += BNW) {
        const int key = blk * 64 + lane;
        const unsigned long long w = __builtin_amdgcn_ballot_w64(key < sk && p.ids_k[ex.krow0 + (key < sk ? key : sk - 1)] != 0);
        if (lane == 0) kmask_s[blk] = w;
    }
    uint32_t ktr[2][2], kra[4];
    tr_addresses((uint32_t)(uintptr_t)smem, lane, ktr);
    row_frag_addresses((uint32_t)(uintptr_t)smem, lane, kra);

    floatx16 dqacc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqacc[j][r] = 0.f;
    const float ik = DROP ? p.keep_scale : 1.f;
    const float Dk = DROP ? Dq * p.inv_keep_scale : Dq;
    const uint32_t thr = p.drop_thr;
    const uint32_t rh = emdr2_row_hash(p.seed, (unsigned long long)si);
    if (qvalid && hi == 0) {       // the dk/dv kernel stages these per 32-query step by LDS-DMA: nothing per query is recomputed there
        p.dstat[si] = pm; p.dstat[p.stat_n + si] = Dk;                          // (the dk/dv kernel works in units of 1 / keep_scale as well)
        ((uint32_t *)p.dstat)[2 * p.stat_n + si] = rh; ((uint32_t *)p.dstat)[3 * p.stat_n + si] = qpad ? 0u : 1u;
    }

    int stage = 0;
    for (int blk = blk_lo; blk < nblk; ++blk, stage = stage == 2 ? 0 : stage + 1) {
        if (blk + 1 < nblk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (8 / BNW)) : "memory");      // all but the next block's pieces
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                        // raw barrier (no fence: the look-ahead DMA stays in flight)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (blk + 2 < nblk) issue(blk + 2, stage == 0 ? 2 : stage - 1);
        const unsigned long long kmask = kmask_s[blk];
        const int key0 = blk * 64;
        // every (query of this wave, key of this block) pair masked -> dS == 0: nothing to add
        if (!wave_live || all_qpad || kmask == 0ull || (CAUSAL && key0 > q0 + 31)) continue;
        const uint32_t a0[2] = {ktr[0][0] + (uint32_t)(stage * 16384), ktr[0][1] + (uint32_t)(stage * 16384)};
        const uint32_t a1[2] = {ktr[1][0] + (uint32_t)(stage * 16384), ktr[1][1] + (uint32_t)(stage * 16384)};
#pragma unroll
        for (int j = 0; j < 2; ++j) {                                             // two 32-key steps per staged block
            const uint32_t km = (uint32_t)(kmask >> (32 * j));
            const int kb0 = key0 + 32 * j;
            if (kb0 >= sk || km == 0u || (CAUSAL && kb0 > q0 + 31)) continue;                  // every pair of this step masked: dS == 0
            const floatx16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            // row fragments of the K and V tiles by inline asm (attention_common.h: a plain load would wait for the next block's DMA first)
            const uint32_t so = (uint32_t)(stage * 16384);
            floatx16 sacc, pacc;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                bf16x8 kr, vr;
                if (j == 0) { LDS_READ128(kr, kra[t] + so, 0); LDS_READ128(vr, kra[t] + so, 8192); }
                else { LDS_READ128(kr, kra[t] + so, 4096); LDS_READ128(vr, kra[t] + so, 8192 + 4096); }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kr), "+v"(vr)::"memory");
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kr, qf[t], t == 0 ? zero : sacc, 0, 0, 0);
                pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vr, dof[t], t == 0 ? zero : pacc, 0, 0, 0);
            }
            // dS^T = P^T (dP^T_eff - D); masked -> 0
            const bool need_mask = km != 0xffffffffu || (CAUSAL && kb0 + 31 > q0) || __builtin_amdgcn_ballot_w64(qpad) != 0ull;   // wave-uniform
            const uint32_t prod0 = DROP ? ((uint32_t)(kb0 + 4 * hi) >> 1) * EMDR2_PAIR_MUL : 0u;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint32_t b0 = 0, b1 = 0;
                if (DROP) {
                    b0 = emdr2_pair_bits_prod(rh, prod0 + (uint32_t)(4 * g) * EMDR2_PAIR_MUL);
                    b1 = emdr2_pair_bits_prod(rh, prod0 + (uint32_t)(4 * g + 1) * EMDR2_PAIR_MUL);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    const float pr = __builtin_amdgcn_exp2f(fmaf(sacc[r], sc, -pm));
                    float gr = pacc[r];
                    if (DROP) {
                        const uint32_t bits = e < 2 ? b0 : b1;
                        gr = ((e & 1) ? (bits >> 16) : (bits & 0xffffu)) >= thr ? gr : 0.f;
                    }
                    sacc[r] = pr * (gr - Dk);                              // dS / keep_scale (dQ is scaled once, at the end)
                }
            }
            // masks are rare (padding, the causal diagonal): ONE wave-uniform branch around all sixteen selects.  Tested per score
            // (`if (need_mask)` inside the loops above) the compiler turns the branch into arithmetic and every score pays the bit test, the
            // compare and the select whether a mask exists or not: 3 of 18 VALU instructions per score.
            if (need_mask) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kl = 8 * (r >> 2) + 4 * hi + (r & 3);
                    const bool masked = qpad || !((km >> kl) & 1u) || (CAUSAL && kb0 + kl > qi);
                    sacc[r] = masked ? 0.f : sacc[r];
                }
            }
            // dQ^T += K^T dS^T
#pragma unroll
            for (int w2 = 0; w2 < 2; ++w2) {
                const int r0 = w2 * 8;
                const bf16x8 dsf = __builtin_bit_cast(bf16x8, make_uint4(pack_bf16(sacc[r0], sacc[r0 + 1]), pack_bf16(sacc[r0 + 2], sacc[r0 + 3]),
                                                                          pack_bf16(sacc[r0 + 4], sacc[r0 + 5]), pack_bf16(sacc[r0 + 6], sacc[r0 + 7])));
                bf16x8 kt0, kt1;
                if (j == 0 && w2 == 0) TR_FRAG2(kt0, a0, kt1, a1, 0);
                else if (j == 0) TR_FRAG2(kt0, a0, kt1, a1, 1);
                else if (w2 == 0) TR_FRAG2(kt0, a0, kt1, a1, 2);
                else TR_FRAG2(kt0, a0, kt1, a1, 3);
                dqacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt0, dsf, dqacc[0], 0, 0, 0);
                dqacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt1, dsf, dqacc[1], 0, 0, 0);
            }
        }
    }
    if (p.ksplit > 1) {                                                           // partial dQ of this key range (unscaled, fp32)
        if (qvalid) {
            float *prow = p.part_dq + ((long long)split * p.stat_n + si) * 64;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *(float4 *)(prow + j * 32 + 8 * g + 4 * hi) = make_float4(dqacc[j][4 * g], dqacc[j][4 * g + 1], dqacc[j][4 * g + 2], dqacc[j][4 * g + 3]);
        }
        return;
    }
    if (qvalid) {
        const float qsc = p.scale * ik;
        uint16_t *drow = (uint16_t *)p.dq + ex.dq_off + (long long)qi * p.dq_ss + n * 64;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = j * 32 + 8 * g + 4 * hi;
                *(uint2 *)(drow + d) = make_uint2(pack_bf16(dqacc[j][4 * g] * qsc, dqacc[j][4 * g + 1] * qsc),
                                                  pack_bf16(dqacc[j][4 * g + 2] * qsc, dqacc[j][4 * g + 3] * qsc));
            }
    }
}

// dQ = (sum of the splits' partials) x scale / (1 - p): dense-q launches only; one wave per query row, lane = d, partials in split order
__global__ void __launch_bounds__(256) attention_dq_combine_kernel(BwdParams p, float qsc)
{
    const long long si = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int d = threadIdx.x & 63;
    if (si >= p.stat_n) return;
    float acc = 0.f;
    for (int s = 0; s < p.ksplit; ++s) acc += p.part_dq[((long long)s * p.stat_n + si) * 64 + d];
    const long long bn = si / p.sq, q = si - bn * p.sq, b = bn / p.heads, n = bn - b * p.heads;
    ((__bf16 *)p.dq)[b * p.dq_sb + q * p.dq_ss + n * 64 + d] = (__bf16)(acc * qsc);
}

// ========================================================== dk, dv ===================================================================
// A wave owns 32 keys, a workgroup 128 (KNW waves); the queries stream past in steps of 32.  r04: THREE workgroups per CU (<= 168 VGPRs),
// like the forward and the dq kernel.  The kernel is VALU-bound (~20 VALU instructions per score against 16 MFMAs per 1,024 scores), and at
// two waves per SIMD (r03: 242 VGPRs) the vector pipe sat idle 44 % of the time -- every MFMA group waited on its own LDS fragment reads
// with only one other wave to cover (SQ_WAIT_ANY / SQ_WAVE_CYCLES 0.38-0.45; the dq kernel, same arithmetic at three waves: 0.29).  What
// made room: the wave's K / V fragments (B operands of S = Q K^T and dP = dO V^T, 32 registers) live in LDS next to the Q / dO stages and
// are re-read every step, and a step is 32 queries with a double-buffered 8 KiB stage (Q + dO rows) instead of 64 queries in a
// three-deep ring of 16 KiB stages: 32 KiB (K, V tiles of the workgroup's 128 keys) + 16 KiB (stages) + statistics = 49 KiB per workgroup.
#define QS 32                  // queries per step
template <bool DROP, bool CAUSAL>
__global__ void __launch_bounds__(KNW * 64, 3) attention_bwd_dkv_kernel(BwdParams p)
{
    // [0, 16K) K tile [128 keys][64 d], [16K, 32K) V tile, then 2 stages x (Q tile [32 q][64 d] 4 KiB + dO tile 4 KiB); all tiles swizzled as in
    // attention_common.h.  Per-query statistics of a step: pm, D, row hash (32 each) + the real-query bits
    __shared__ __attribute__((aligned(16))) char smem[32768 + 2 * 8192];
    __shared__ __attribute__((aligned(16))) float st_all[2][4][QS];       // [stage][pm, D, row hash (bits), real-token flag (bits)][query]
#define st_pm(stage, q) st_all[stage][0][q]
#define st_d(stage, q) st_all[stage][1][q]
#define st_rh(stage, q) st_all[stage][2][q]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    int kblk, b, n;
    if (!attn_decode(blockIdx.x, (p.sk + KNW * 32 - 1) / (KNW * 32), p.batch * p.heads, p.heads, kblk, b, n)) return;
    const SeqExtent ex = seq_extent(p, b, n);
    const int sq = ex.sq, sk = ex.sk;
    if (kblk * (KNW * 32) >= sk || sq < 1) return;
    const int k0 = kblk * (KNW * 32) + wave * 32;
    const int key = k0 + l31;
    const bool wave_live = k0 < sk;
    const bool kvalid = key < sk;                                       // (dense: sk % 32 == 0, so a live wave's keys are all valid)
    const int kc = kvalid ? key : sk - 1;

    // LDS-DMA pieces of 1 KiB = 8 rows x 128 B: lane -> (row = 8 piece + lane>>3, LDS granule = lane&7), source granule = granule ^ tile_swz(row)
    // (tile_swz looks at row bits 1..3: period 16)
    const int prow = wave * 8 + (lane >> 3), pslot = (lane & 7) ^ tile_swz(wave * 8 + (lane >> 3));
    // DMA sources as a wave-uniform base (SGPRs) + a 32-bit lane offset: two 64-bit pointers per lane cost registers this kernel does not have
    const char *q_base = p.q + (ex.q_off + (long long)n * p.q_sn) * 2;
    const char *o_base = p.dout + (ex.qrow0 * p.heads + n) * 128;
    const uint32_t q_row_bytes = (uint32_t)p.q_ss * 2u, o_row_bytes = (uint32_t)p.heads * 128u;     // (a sequence spans < 2^32 bytes: <= 65536 rows)
    const int nstep = (sq + QS - 1) / QS;
    // a step's per-query statistics (written by the dq kernel): 4 arrays x 32 queries = two LDS-DMA instructions of 64 x 4 bytes, issued by
    // waves 0 and 1 together with their tile pieces (lane -> array 2 wave + hi, query l31)
    const char *st_base = (const char *)(p.dstat + ex.stat0);
    const uint32_t st_arr = (uint32_t)(2 * (wave & 1) + hi) * (uint32_t)p.stat_n * 4u;               // (4 stat_n floats < 2^32 bytes: checked on the host)
    auto issue = [&](int step, int stage) {                                // a step's Q / dO rows: 4 + 4 pieces, one of each per wave
        char *sb = smem + 32768 + stage * 8192;
        const uint32_t qr = (uint32_t)min(step * QS + prow, sq - 1);            // overhang queries re-read the last row; masked out below
        __builtin_amdgcn_global_load_lds((gptr_t *)(q_base + (__umul24(qr, q_row_bytes) + (uint32_t)(pslot * 16))), (lptr_t *)(sb + wave * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t *)(o_base + (__umul24(qr, o_row_bytes) + (uint32_t)(pslot * 16))), (lptr_t *)(sb + 4096 + wave * 1024), 16, 0, 0);
        if (wave < 2) {
            const uint32_t qq = (uint32_t)min(step * QS + l31, sq - 1);
            __builtin_amdgcn_global_load_lds((gptr_t *)(st_base + (st_arr + qq * 4u)), (lptr_t *)((char *)&st_all[stage][0][0] + wave * 256), 4, 0, 0);
        }
    };
    issue(0, 0);                     // first Q / dO steps in flight before this wave's K / V rows and the statistics are fetched
    {   // this wave's own 32 rows of the K and V tiles (nobody else reads them: no barrier needed, the counted wait below is enough)
        const char *k_src = p.k + (ex.k_off + (long long)n * p.k_sn) * 2;
        const char *v_src = p.v + (ex.v_off + (long long)n * p.v_sn) * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int trow = 8 * i + (lane >> 3);                                 // row inside this wave's 32-row slice of the tiles
            const int slot = ((lane & 7) ^ tile_swz(trow)) * 16;
            long long kr = k0 + trow; if (kr >= sk) kr = sk - 1;
            __builtin_amdgcn_global_load_lds((gptr_t *)(k_src + kr * p.k_ss * 2 + slot), (lptr_t *)(smem + (wave * 4 + i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t *)(v_src + kr * p.v_ss * 2 + slot), (lptr_t *)(smem + 16384 + (wave * 4 + i) * 1024), 16, 0, 0);
        }
    }
    if (nstep > 1) issue(1, 1);
    const bool kpad = !kvalid || p.ids_k[ex.krow0 + kc] == 0;
    const float sc = p.scale2;

    uint32_t qtr[2][2], qra[4], kra[4];
    row_frag_addresses((uint32_t)(uintptr_t)smem + 32768, lane, qra);        // Q rows of stage 0; dO: + 4096; stage 1: + 8192
    tr_addresses((uint32_t)(uintptr_t)smem + 32768, lane, qtr);               // Q^T; dO^T: + 4096
    row_frag_addresses((uint32_t)(uintptr_t)smem + wave * 4096, lane, kra);  // this lane's key row of the K tile; V: + 16384

    floatx16 dkacc[2], dvacc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dkacc[j][r] = 0.f; dvacc[j][r] = 0.f; }
    const float ik = DROP ? p.keep_scale : 1.f;
    const uint32_t thr = p.drop_thr;
    const uint32_t colmul = ((uint32_t)kc >> 1) * 0x9e3779b1u;           // this lane's column-pair term of emdr2_pair_bits
    const uint32_t csh = (kc & 1) ? 16u : 0u;                             // which half of the pair's 32 bits is this key's

    for (int step = 0; step < nstep; ++step) {
        const int stage = step & 1;
        // this step's tiles and statistics (the newest DMA of this wave; in step 0 also the K / V rows and everything else the prologue asked for)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                        // raw barrier (no fence needed: nothing else is in flight)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // the NEXT step's rows go into the stage everybody has just left (step - 1's).  Its latency is covered by this step's arithmetic of
        // this wave and its two SIMD mates.  (step 1 was issued in the prologue)
        if (step >= 1 && step + 1 < nstep) issue(step + 1, stage ^ 1);
        const int qh0 = step * QS;
        // keys of this wave all ahead of every query of the step: P == 0 exactly and dS == 0
        if (!wave_live || (CAUSAL && k0 > qh0 + QS - 1)) continue;
        const uint32_t qr32 = (uint32_t)__builtin_amdgcn_ballot_w64(__float_as_uint(st_all[stage][3][l31]) != 0u);     // bit q: query q of the step is a real token
        const uint32_t so = (uint32_t)(stage * 8192);

        const floatx16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        // Five phases, so that S and dP are never live together with the statistics and hash temporaries of both (the single-pass form needed
        // ~185 registers; an accumulator spilled to scratch once per step costs 8 KiB of traffic per wave and step, four times what the step
        // stages): (1) S = Q K^T, (2) P and the dropped P from S -- the un-dropped P stays in the S registers with the keep bit as its SIGN --,
        // (3) dV^T += dO^T P_d, (4) dP = dO V^T, (5) dS = |P| (keep ? dP / (1 - p) : 0 - D), dK^T += Q^T dS.
        // Fragments by inline asm (attention_common.h: a plain load would make the compiler wait for the next step's DMA first): A = row
        // fragments of the Q / dO tiles, B = this lane's key row of the K / V tiles.
        // (Issuing a phase's fragment reads a phase early -- measured in four depths -- buys nothing at three waves per SIMD and costs the
        // registers that keep the loop free of scratch traffic: reads are issued where they are used, four at a time.)
        const uint32_t aq0[2] = {qtr[0][0] + so, qtr[0][1] + so}, aq1[2] = {qtr[1][0] + so, qtr[1][1] + so};
        const uint32_t ao0[2] = {aq0[0] + 4096, aq0[1] + 4096}, ao1[2] = {aq1[0] + 4096, aq1[1] + 4096};
        floatx16 sacc;
#pragma unroll
        for (int t = 0; t < 4; t += 2) {
            bf16x8 qr0, kr0, qr1, kr1;
            LDS_READ128(qr0, qra[t] + so, 0); LDS_READ128(kr0, kra[t], 0);
            LDS_READ128(qr1, qra[t + 1] + so, 0); LDS_READ128(kr1, kra[t + 1], 0);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(qr0), "+v"(kr0), "+v"(qr1), "+v"(kr1)::"memory");
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qr0, kr0, t == 0 ? zero : sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qr1, kr1, sacc, 0, 0, 0);
        }
        // this lane: one key, queries ql = 8g + 4hi + e.  Masks are rare (padding, the causal diagonal, the last query step): a wave-uniform
        // test picks the mask-free form of the element loops otherwise
        const bool need_mask = __builtin_amdgcn_ballot_w64(kpad) != 0ull || qr32 != 0xffffffffu || (CAUSAL && k0 + 31 > qh0) || qh0 + QS - 1 >= sq;
        const uint32_t qbits = qr32 >> (4 * hi);                               // bit 8g + e = "query ql is a real token" for this half-wave
        uint32_t pw[8];                                                        // dropped P, packed bf16 pairs (B operand of the dV product)
        auto probabilities = [&](auto masks) {
            constexpr bool MASKS = decltype(masks)::value;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ql0 = 8 * g + 4 * hi;
                const float4 pm4 = *(const float4 *)&st_pm(stage, ql0);
                uint4 rh4 = make_uint4(0, 0, 0, 0);
                if (DROP) rh4 = *(const uint4 *)&st_rh(stage, ql0);
                const float pmv[4] = {pm4.x, pm4.y, pm4.z, pm4.w};
                const uint32_t rhv[4] = {rh4.x, rh4.y, rh4.z, rh4.w};
                float pdv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e, qg = qh0 + 8 * g + 4 * hi + e;
                    bool masked = false;
                    float pr;
                    if (MASKS) {
                        masked = kpad || !((qbits >> (8 * g + e)) & 1u) || (CAUSAL && key > qg);
                        pr = __builtin_amdgcn_exp2f((masked ? MASKED2 : sacc[r] * sc) - pmv[e]);
                        pr = qg < sq ? pr : 0.f;                          // rows beyond sq do not exist
                    } else {
                        pr = __builtin_amdgcn_exp2f(fmaf(sacc[r], sc, -pmv[e]));
                    }
                    float pd = pr, keep_pr = pr;
                    if (DROP) {
                        const uint32_t bits = emdr2_pair_bits_prod(rhv[e], colmul);
                        const bool keep = ((bits >> csh) & 0xffffu) >= thr;
                        keep_pr = keep ? pr : -pr;                        // P >= 0: the sign carries the keep bit to phase 5
                        pd = fmaxf(keep_pr, 0.f);                         // dropped P in units of 1 / keep_scale (dV is scaled once, at the end)
                    }
                    sacc[r] = MASKS ? (masked ? 0.f : keep_pr) : keep_pr;  // masked: dS = 0 (masked_fill cuts the dependence on the score) ...
                    pdv[e] = pd;                                          // ... while the (normally zero) probability still feeds dV, like the reference
                }
                pw[2 * g] = pack_bf16(pdv[0], pdv[1]); pw[2 * g + 1] = pack_bf16(pdv[2], pdv[3]);
                __builtin_amdgcn_sched_barrier(0);     // one group of four queries at a time: interleaving the groups' hashes costs > 100 registers
            }
        };
        if (need_mask) probabilities(std::true_type{});
        else probabilities(std::false_type{});
        // dV^T += dO^T P_d  (k index = the 16 queries of k-step uu)
#pragma unroll
        for (int uu = 0; uu < 2; ++uu) {
            const bf16x8 pf = __builtin_bit_cast(bf16x8, make_uint4(pw[4 * uu], pw[4 * uu + 1], pw[4 * uu + 2], pw[4 * uu + 3]));
            bf16x8 ot0, ot1;
            if (uu == 0) TR_FRAG2(ot0, ao0, ot1, ao1, 0);
            else TR_FRAG2(ot0, ao0, ot1, ao1, 1);
            dvacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ot0, pf, dvacc[0], 0, 0, 0);
            dvacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ot1, pf, dvacc[1], 0, 0, 0);
        }
        floatx16 pacc;
#pragma unroll
        for (int t = 0; t < 4; t += 2) {
            bf16x8 or0, vr0, or1, vr1;
            LDS_READ128(or0, qra[t] + so, 4096); LDS_READ128(vr0, kra[t], 16384);
            LDS_READ128(or1, qra[t + 1] + so, 4096); LDS_READ128(vr1, kra[t + 1], 16384);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(or0), "+v"(vr0), "+v"(or1), "+v"(vr1)::"memory");
            pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(or0, vr0, t == 0 ? zero : pacc, 0, 0, 0);
            pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(or1, vr1, pacc, 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 d4 = *(const float4 *)&st_d(stage, 8 * g + 4 * hi);
            const float dv_[4] = {d4.x, d4.y, d4.z, d4.w};
            float dsv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g + e;
                // dS / keep_scale = P (keep ? dP : 0) - P D / keep_scale; kept: P = max(q, 0), always: P = |q| (q = sacc: P with the keep bit as sign)
                if (DROP) dsv[e] = fmaf(fmaxf(sacc[r], 0.f), pacc[r], -(__builtin_fabsf(sacc[r]) * dv_[e]));
                else dsv[e] = sacc[r] * (pacc[r] - dv_[e]);
            }
            pw[2 * g] = pack_bf16(dsv[0], dsv[1]); pw[2 * g + 1] = pack_bf16(dsv[2], dsv[3]);
        }
        // dK^T += Q^T dS
#pragma unroll
        for (int uu = 0; uu < 2; ++uu) {
            const bf16x8 dsf = __builtin_bit_cast(bf16x8, make_uint4(pw[4 * uu], pw[4 * uu + 1], pw[4 * uu + 2], pw[4 * uu + 3]));
            bf16x8 qt0, qt1;
            if (uu == 0) TR_FRAG2(qt0, aq0, qt1, aq1, 0);
            else TR_FRAG2(qt0, aq0, qt1, aq1, 1);
            dkacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt0, dsf, dkacc[0], 0, 0, 0);
            dkacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt1, dsf, dkacc[1], 0, 0, 0);
        }
    }
#undef st_pm
#undef st_d
#undef st_rh
    if (kvalid) {
        const float ksc = p.scale * ik;                                // the step loop accumulated dS and the dropped P in units of 1 / keep_scale
        uint16_t *krow = (uint16_t *)p.dk + ex.dkv_off + (long long)key * p.dkv_ss + n * 64;
        uint16_t *vrow = (uint16_t *)p.dv + ex.dkv_off + (long long)key * p.dkv_ss + n * 64;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = j * 32 + 8 * g + 4 * hi;
                *(uint2 *)(krow + d) = make_uint2(pack_bf16(dkacc[j][4 * g] * ksc, dkacc[j][4 * g + 1] * ksc),
                                                  pack_bf16(dkacc[j][4 * g + 2] * ksc, dkacc[j][4 * g + 3] * ksc));
                *(uint2 *)(vrow + d) = make_uint2(pack_bf16(dvacc[j][4 * g] * ik, dvacc[j][4 * g + 1] * ik), pack_bf16(dvacc[j][4 * g + 2] * ik, dvacc[j][4 * g + 3] * ik));
            }
    }
}

} // namespace

static int attention_bwd_launch(const void *q, int64_t q_sb, int64_t q_ss, int64_t q_sn, const void *k, int64_t k_sb, int64_t k_ss, int64_t k_sn,
                                const void *v, int64_t v_sb, int64_t v_ss, int64_t v_sn, const void *o, const void *dout, void *dq, int64_t dq_sb,
                                int64_t dq_ss, void *dk, void *dv, int64_t dkv_sb, int64_t dkv_ss, const int64_t *ids_q, const int64_t *ids_k, const float *m, const float *l, float *dstat, int batch,
                                int heads, int sq, int sk, int head_dim, int causal, float scale, float drop_p, uint32_t seed, const int32_t *cu_q, const int32_t *cu_k,
                                int64_t total_q, double pairs, void *stream, int ksplit = 1, void *ws = nullptr, size_t ws_bytes = 0)
{
    if (!q || !k || !v || !o || !dout || !dq || !dk || !dv || !ids_q || !ids_k || !m || !l || !dstat || batch < 1 || heads < 1 || sq < 1) return -1;
    if (ksplit < 1 || ksplit > 64 || (ksplit > 1 && (cu_q || !ws || ((uintptr_t)ws & 15) || sq > BNW * 32 || ksplit != ((sk + 63) / 64 + KSPLIT_BLOCKS - 1) / KSPLIT_BLOCKS))) return -1;
    if (head_dim != 64 || sk < 1 || sk > 65536) return -4;
    if (!cu_k && (sk < 32 || (sk & 31))) return -4;
    if (cu_q && total_q < 1) return -1;
    const int64_t strides[13] = {q_sb, q_ss, q_sn, k_sb, k_ss, k_sn, v_sb, v_ss, v_sn, dq_sb, dq_ss, dkv_sb, dkv_ss};
    for (int i = 0; i < 13; ++i)
        if (strides[i] & 7) return -4;
    if (((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)v & 15) || ((uintptr_t)o & 15) || ((uintptr_t)dout & 15) || ((uintptr_t)dq & 7) ||
        ((uintptr_t)dk & 7) || ((uintptr_t)dv & 7) || drop_p < 0.f || drop_p >= 1.f)
        return -1;
    BwdParams p;
    p.q = (const char *)q; p.k = (const char *)k; p.v = (const char *)v; p.o = (const char *)o; p.dout = (const char *)dout;
    p.dq = (char *)dq; p.dk = (char *)dk; p.dv = (char *)dv;
    p.ids_q = (const long long *)ids_q; p.ids_k = (const long long *)ids_k; p.m = m; p.l = l; p.dstat = dstat;
    p.q_sb = q_sb; p.q_ss = q_ss; p.q_sn = q_sn; p.k_sb = k_sb; p.k_ss = k_ss; p.k_sn = k_sn; p.v_sb = v_sb; p.v_ss = v_ss; p.v_sn = v_sn;
    p.dq_sb = dq_sb; p.dq_ss = dq_ss; p.dkv_sb = dkv_sb; p.dkv_ss = dkv_ss;
    p.heads = heads; p.sq = sq; p.sk = sk; p.causal = causal; p.scale = scale; p.drop_p = drop_p; p.seed = seed;
    p.batch = batch; p.cu_q = cu_q; p.cu_k = cu_k; p.tq = total_q;
    p.stat_n = cu_q ? (long long)heads * total_q : (long long)batch * heads * sq;
    if (p.stat_n * 16 >= (1ll << 32) || q_ss >= (1 << 22) || heads >= (1 << 16)) return -4;     // 32-bit lane offsets of the dk/dv kernel's staging
    p.keep_scale = emdr2_keep_scale(drop_p); p.inv_keep_scale = 1.f / p.keep_scale; p.scale2 = scale * L2E; p.drop_thr = emdr2_drop_thr(drop_p);
    p.ksplit = ksplit; p.base_grid = attn_grid((sq + BNW * 32 - 1) / (BNW * 32), batch * heads, heads); p.part_dq = (float *)ws;
    if (ksplit > 1 && ws_bytes < (size_t)ksplit * p.stat_n * 64 * sizeof(float)) return -2;
    OpsTimer timer(OPS_ATTN_BWD, 10.0 * heads * pairs * 64, (hipStream_t)stream);
    const dim3 dq_grid(p.base_grid * (unsigned)ksplit);
    const size_t dq_lds = 3 * 16384 + (size_t)((sk + 63) / 64) * 8;
    if (drop_p > 0.f && causal) hipLaunchKernelGGL((attention_bwd_dq_kernel<true, true>), dq_grid, dim3(BNW * 64), dq_lds, (hipStream_t)stream, p);
    else if (drop_p > 0.f) hipLaunchKernelGGL((attention_bwd_dq_kernel<true, false>), dq_grid, dim3(BNW * 64), dq_lds, (hipStream_t)stream, p);
    else if (causal) hipLaunchKernelGGL((attention_bwd_dq_kernel<false, true>), dq_grid, dim3(BNW * 64), dq_lds, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((attention_bwd_dq_kernel<false, false>), dq_grid, dim3(BNW * 64), dq_lds, (hipStream_t)stream, p);
    if (ksplit > 1)
        hipLaunchKernelGGL(attention_dq_combine_kernel, dim3((unsigned)((p.stat_n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p,
                           scale * (drop_p > 0.f ? p.keep_scale : 1.f));
    const dim3 kv_grid(attn_grid((sk + KNW * 32 - 1) / (KNW * 32), batch * heads, heads));
    if (drop_p > 0.f && causal) hipLaunchKernelGGL((attention_bwd_dkv_kernel<true, true>), kv_grid, dim3(KNW * 64), 0, (hipStream_t)stream, p);
    else if (drop_p > 0.f) hipLaunchKernelGGL((attention_bwd_dkv_kernel<true, false>), kv_grid, dim3(KNW * 64), 0, (hipStream_t)stream, p);
    else if (causal) hipLaunchKernelGGL((attention_bwd_dkv_kernel<false, true>), kv_grid, dim3(KNW * 64), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((attention_bwd_dkv_kernel<false, false>), kv_grid, dim3(KNW * 64), 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

extern "C" int emdr2_attention_bwd(const void *q, int64_t q_sb, int64_t q_ss, int64_t q_sn, const void *k, int64_t k_sb, int64_t k_ss, int64_t k_sn,
                                   const void *v, int64_t v_sb, int64_t v_ss, int64_t v_sn, const void *o, const void *dout, void *dq, int64_t dq_sb,
                                   int64_t dq_ss, void *dk, void *dv, int64_t dkv_sb, int64_t dkv_ss, const int64_t *ids_q, const int64_t *ids_k, const float *m, const float *l, float *dstat, int batch,
                                   int heads, int sq, int sk, int head_dim, int causal, float scale, float drop_p, uint32_t seed, void *stream)
{
    return attention_bwd_launch(q, q_sb, q_ss, q_sn, k, k_sb, k_ss, k_sn, v, v_sb, v_ss, v_sn, o, dout, dq, dq_sb, dq_ss, dk, dv, dkv_sb, dkv_ss, ids_q, ids_k, m, l,
                                dstat, batch, heads, sq, sk, head_dim, causal, scale, drop_p, seed, nullptr, nullptr, 0, (double)batch * sq * sk, stream);
}

extern "C" int emdr2_attention_bwd_splitkv(const void *q, int64_t q_sb, int64_t q_ss, int64_t q_sn, const void *k, int64_t k_sb, int64_t k_ss, int64_t k_sn,
                                           const void *v, int64_t v_sb, int64_t v_ss, int64_t v_sn, const void *o, const void *dout, void *dq, int64_t dq_sb,
                                           int64_t dq_ss, void *dk, void *dv, int64_t dkv_sb, int64_t dkv_ss, const int64_t *ids_q, const int64_t *ids_k,
                                           const int32_t *cu_k, int64_t pairs, const float *m, const float *l, float *dstat,
                                           int batch, int heads, int sq, int max_sk, int head_dim, int causal, float scale, float drop_p, uint32_t seed,
                                           int ksplit, void *ws, size_t ws_bytes, void *stream)
{
    return attention_bwd_launch(q, q_sb, q_ss, q_sn, k, k_sb, k_ss, k_sn, v, v_sb, v_ss, v_sn, o, dout, dq, dq_sb, dq_ss, dk, dv, dkv_sb, dkv_ss, ids_q, ids_k, m, l,
                                dstat, batch, heads, sq, max_sk, head_dim, causal, scale, drop_p, seed, nullptr, cu_k, 0, (double)pairs, stream, ksplit, ws, ws_bytes);
}

extern "C" int emdr2_attention_varlen_bwd(const void *q, int64_t q_sb, int64_t q_ss, int64_t q_sn, const void *k, int64_t k_sb, int64_t k_ss, int64_t k_sn,
                                          const void *v, int64_t v_sb, int64_t v_ss, int64_t v_sn, const void *o, const void *dout, void *dq, int64_t dq_sb,
                                          int64_t dq_ss, void *dk, void *dv, int64_t dkv_sb, int64_t dkv_ss, const int64_t *ids_q, const int64_t *ids_k,
                                          const int32_t *cu_q, const int32_t *cu_k, int64_t total_q, int64_t pairs, const float *m, const float *l, float *dstat,
                                          int batch, int heads, int max_sq, int max_sk, int head_dim, int causal, float scale, float drop_p, uint32_t seed, void *stream)
{
    if (!cu_q && !cu_k) return -1;
    return attention_bwd_launch(q, q_sb, q_ss, q_sn, k, k_sb, k_ss, k_sn, v, v_sb, v_ss, v_sn, o, dout, dq, dq_sb, dq_ss, dk, dv, dkv_sb, dkv_ss, ids_q, ids_k, m, l,
                                dstat, batch, heads, max_sq, max_sk, head_dim, causal, scale, drop_p, seed, cu_q, cu_k, total_q, (double)pairs, stream);
}
