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
    float *dstat;                  // D[q] = rowsum(dO o O), [b, heads, sq]
    long long q_sb, q_ss, q_sn;    // element strides of q (batch, sequence, head); k and v share k_*
    long long k_sb, k_ss, k_sn, v_sb, v_ss, v_sn;
    long long dq_sb, dq_ss, dkv_sb, dkv_ss;   // element strides (batch, sequence) of the gradient outputs; heads are 64 apart
    int heads, sq, sk, causal, batch;
    float scale, drop_p;
    uint32_t seed;
    // packed operands (attention.hip: AttnParams): sequence b owns rows [cu[b], cu[b+1]); statistics [heads, tq] when cu_q is given
    const int *cu_q, *cu_k;
    long long tq;
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
#define BNW 4
#ifndef DKV_OCC
#define DKV_OCC 2
#endif
#ifndef KNW
#define KNW 4                  // waves (32 keys each) per workgroup of the dK / dV kernel
#endif
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
    if (!attn_decode(blockIdx.x, (p.sq + BNW * 32 - 1) / (BNW * 32), p.batch * p.heads, p.heads, qblk, b, n)) return;
    const SeqExtent ex = seq_extent(p, b, n);
    const int sq = ex.sq, sk = ex.sk;
    if (qblk * (BNW * 32) >= sq || sk < 1) return;
    const int q0 = qblk * (BNW * 32) + wave * 32;
    const int qi = q0 + l31;
    const bool qvalid = qi < sq;
    const int qc = qvalid ? qi : sq - 1;
    const bool wave_live = q0 < sq;

    const int prow = wave * 8 + (lane >> 3), pslot = (lane & 7) ^ tile_swz(wave * 8 + (lane >> 3));
    const char *k_src = p.k + (ex.k_off + (long long)n * p.k_sn) * 2 + pslot * 16;
    const char *v_src = p.v + (ex.v_off + (long long)n * p.v_sn) * 2 + pslot * 16;
    const int nblk = (sk + 63) / 64;
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
    issue(0, 0);                     // before the fragment / statistics loads below: a short sequence must not pay the memory latency twice
    if (nblk > 1) issue(1, 1);

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
        if (qvalid && hi == 0) p.dstat[si] = Dq;
    }
    const bool qpad = !qvalid || p.ids_q[ex.qrow0 + qc] == 0;
    const bool all_qpad = __builtin_amdgcn_ballot_w64(qpad) == ~0ull;  // a wave of padded queries: every dS is 0
    const float sc = p.scale * L2E;
    const float pm = p.m[si] * L2E + __log2f(p.l[si]);                  // P = exp2(s2 - pm)

    for (int blk = wave; blk < nblk; blk += BNW) {
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
    const float ik = DROP ? emdr2_keep_scale(p.drop_p) : 1.f;
    const uint32_t thr = emdr2_drop_thr(p.drop_p);
    const uint32_t rh = emdr2_row_hash(p.seed, (unsigned long long)si);

    int stage = 0;
    for (int blk = 0; blk < nblk; ++blk, stage = stage == 2 ? 0 : stage + 1) {
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
                    const int kl = 8 * g + 4 * hi + e;
                    const float pr = __builtin_amdgcn_exp2f(fmaf(sacc[r], sc, -pm));
                    float gr = pacc[r];
                    if (DROP) {
                        const uint32_t bits = e < 2 ? b0 : b1;
                        gr = ((e & 1) ? (bits >> 16) : (bits & 0xffffu)) >= thr ? gr * ik : 0.f;
                    }
                    float ds = pr * (gr - Dq);
                    if (need_mask) {
                        const bool masked = qpad || !((km >> kl) & 1u) || (CAUSAL && kb0 + kl > qi);
                        ds = masked ? 0.f : ds;
                    }
                    sacc[r] = ds;
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
    if (qvalid) {
        uint16_t *drow = (uint16_t *)p.dq + ex.dq_off + (long long)qi * p.dq_ss + n * 64;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = j * 32 + 8 * g + 4 * hi;
                *(uint2 *)(drow + d) = make_uint2(pack_bf16(dqacc[j][4 * g] * p.scale, dqacc[j][4 * g + 1] * p.scale),
                                                  pack_bf16(dqacc[j][4 * g + 2] * p.scale, dqacc[j][4 * g + 3] * p.scale));
            }
    }
}

// ========================================================== dk, dv ===================================================================
template <bool DROP, bool CAUSAL>
__global__ void __launch_bounds__(KNW * 64, DKV_OCC) attention_bwd_dkv_kernel(BwdParams p)
{
    // 3 stages x (Q tile 8 KiB + dO tile 8 KiB): two blocks of DMA in flight while one is consumed (as in attention.hip); per-query
    // statistics of a block: pm, D, row hash, flags (64 each)
    __shared__ __attribute__((aligned(16))) char smem[3 * 16384];
    __shared__ __attribute__((aligned(16))) float st_pm[3][64], st_d[3][64];
    __shared__ __attribute__((aligned(16))) uint32_t st_rh[3][64];
    __shared__ unsigned long long st_qreal[3];
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

    const int prow = wave * 8 + (lane >> 3), pslot = (lane & 7) ^ tile_swz(wave * 8 + (lane >> 3));
    const char *q_src = p.q + (ex.q_off + (long long)n * p.q_sn) * 2 + pslot * 16;
    const char *o_src = p.dout + (ex.qrow0 * p.heads + n) * 128 + pslot * 16;
    const int nblk = (sq + 63) / 64;
    const long long sbase = ex.stat0;
    auto issue = [&](int blk, int stage) {
        char *sb = smem + stage * 16384;
#pragma unroll
        for (int i = 0; i < 8 / KNW; ++i) {
            long long qr = blk * 64 + prow + 8 * KNW * i; if (qr >= sq) qr = sq - 1;         // overhang queries re-read the last row; masked out below
            __builtin_amdgcn_global_load_lds((gptr_t *)(q_src + qr * p.q_ss * 2), (lptr_t *)(sb + (wave + KNW * i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t *)(o_src + qr * p.heads * 128), (lptr_t *)(sb + 8192 + (wave + KNW * i) * 1024), 16, 0, 0);
        }
    };
    issue(0, 0);                     // first Q / dO blocks in flight before this wave's K / V rows and the statistics are fetched
    if (nblk > 1) issue(1, 1);

    bf16x8 kf[4], vf[4];
    load_row_frags(p.k + (ex.k_off + (long long)kc * p.k_ss + (long long)n * p.k_sn) * 2, hi, kf);
    load_row_frags(p.v + (ex.v_off + (long long)kc * p.v_ss + (long long)n * p.v_sn) * 2, hi, vf);
    const bool kpad = !kvalid || p.ids_k[ex.krow0 + kc] == 0;
    const float sc = p.scale * L2E;

    // per-query statistics of a block, loaded by wave 0 one block ahead into registers and written to LDS a block later, so the
    // global-load latency never sits between a barrier and the tiles' DMA
    float nx_pm = 0.f, nx_d = 0.f;
    uint32_t nx_rh = 0;
    unsigned long long nx_real = 0;
    auto load_stats = [&](int blk) {
        const int qq = blk * 64 + lane;
        const bool valid = qq < sq;
        const long long si = sbase + (valid ? qq : sq - 1);
        nx_pm = p.m[si] * L2E + __log2f(p.l[si]);
        nx_d = p.dstat[si];
        nx_rh = emdr2_row_hash(p.seed, (unsigned long long)si);
        nx_real = __builtin_amdgcn_ballot_w64(valid && p.ids_q[ex.qrow0 + (valid ? qq : 0)] != 0);
    };
    auto store_stats = [&](int stage) {
        st_pm[stage][lane] = nx_pm; st_d[stage][lane] = nx_d; st_rh[stage][lane] = nx_rh;
        if (lane == 0) st_qreal[stage] = nx_real;
    };
    uint32_t qtr[2][2], otr[2][2], qra[4];
    row_frag_addresses((uint32_t)(uintptr_t)smem, lane, qra);
    tr_addresses((uint32_t)(uintptr_t)smem, lane, qtr);
    tr_addresses((uint32_t)(uintptr_t)smem + 8192, lane, otr);

    floatx16 dkacc[2], dvacc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dkacc[j][r] = 0.f; dvacc[j][r] = 0.f; }
    const float ik = DROP ? emdr2_keep_scale(p.drop_p) : 1.f;
    const uint32_t thr = emdr2_drop_thr(p.drop_p);
    const uint32_t colmul = ((uint32_t)kc >> 1) * 0x9e3779b1u;           // this lane's column-pair term of emdr2_pair_bits
    const bool codd = kc & 1;

    if (wave == 0) { load_stats(0); store_stats(0); if (nblk > 1) load_stats(1); }
    // everything fetched so far is complete here, in a form the compiler sees (cf. attention.hip): no hidden vmcnt(0) inside the loop
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]), "+v"(vf[0]), "+v"(vf[1]), "+v"(vf[2]), "+v"(vf[3]),
                 "+v"(nx_pm), "+v"(nx_d), "+v"(nx_rh)::"memory");
    int stage = 0;
    for (int blk = 0; blk < nblk; ++blk, stage = stage == 2 ? 0 : stage + 1) {
        // all but the newest block's DMA pieces (2 x 8 / KNW instructions of this wave): this block's tiles -- and, on wave 0, the statistics
        // of the next block, which were requested BEFORE that DMA -- have arrived
        if (blk + 1 < nblk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (8 / KNW)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                        // raw barrier: a fence would drain the look-ahead DMA
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        {
            const int nstage = stage == 2 ? 0 : stage + 1;
            if (blk + 1 < nblk && wave == 0) store_stats(nstage);                 // block blk + 1 (its row statistics came in an iteration ago)
            if (blk + 2 < nblk) {
                if (wave == 0) load_stats(blk + 2);                               // ... requested before the DMA of the same block
                issue(blk + 2, stage == 0 ? 2 : stage - 1);
            }
        }
        const int qb0 = blk * 64;
        // keys of this wave all ahead of every query of the block: P == 0 exactly and dS == 0
        if (!wave_live || (CAUSAL && k0 > qb0 + 63)) continue;
        const unsigned long long qreal = st_qreal[stage];

        // the block's 64 queries in two halves of 32 (keeps the live accumulators at dK, dV + one S / dP pair)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            __builtin_amdgcn_sched_barrier(0);                             // keeps the two halves' fragment reads from being hoisted together
            const floatx16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            // row fragments of the Q and dO tiles by inline asm (attention_common.h: a plain load would wait for the next block's DMA first)
            const uint32_t so = (uint32_t)(stage * 16384);
            floatx16 sacc, pacc;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                bf16x8 qr, orr;
                if (j == 0) { LDS_READ128(qr, qra[t] + so, 0); LDS_READ128(orr, qra[t] + so, 8192); }
                else { LDS_READ128(qr, qra[t] + so, 4096); LDS_READ128(orr, qra[t] + so, 8192 + 4096); }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(qr), "+v"(orr)::"memory");
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qr, kf[t], t == 0 ? zero : sacc, 0, 0, 0);
                pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(orr, vf[t], t == 0 ? zero : pacc, 0, 0, 0);
            }
            // this lane: one key, queries ql = j*32 + 8g + 4hi + e.  sacc <- dS, pacc <- dropped P.  Masks are rare (padding, the causal
            // diagonal, the last query block): a wave-uniform test picks the mask-free form of the element loop otherwise
            const uint32_t qr32 = (uint32_t)(qreal >> (32 * j));
            const int qh0 = qb0 + 32 * j;
            const bool need_mask = __builtin_amdgcn_ballot_w64(kpad) != 0ull || qr32 != 0xffffffffu || (CAUSAL && k0 + 31 > qh0) || qh0 + 31 >= sq;
            auto elements = [&](auto masks) {
                constexpr bool MASKS = decltype(masks)::value;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ql0 = j * 32 + 8 * g + 4 * hi;
                    const float4 pm4 = *(const float4 *)&st_pm[stage][ql0], d4 = *(const float4 *)&st_d[stage][ql0];
                    uint4 rh4 = make_uint4(0, 0, 0, 0);
                    if (DROP) rh4 = *(const uint4 *)&st_rh[stage][ql0];
                    const float pmv[4] = {pm4.x, pm4.y, pm4.z, pm4.w}, dv_[4] = {d4.x, d4.y, d4.z, d4.w};
                    const uint32_t rhv[4] = {rh4.x, rh4.y, rh4.z, rh4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * g + e, ql = 8 * g + 4 * hi + e, qg = qh0 + ql;
                        bool masked = false;
                        float pr;
                        if (MASKS) {
                            masked = kpad || !((qr32 >> ql) & 1u) || (CAUSAL && key > qg);
                            pr = __builtin_amdgcn_exp2f((masked ? MASKED2 : sacc[r] * sc) - pmv[e]);
                            pr = qg < sq ? pr : 0.f;                          // rows beyond sq do not exist
                        } else {
                            pr = __builtin_amdgcn_exp2f(fmaf(sacc[r], sc, -pmv[e]));
                        }
                        float gr = pacc[r], pd = pr;
                        if (DROP) {
                            const uint32_t bits = emdr2_pair_bits_prod(rhv[e], colmul);
                            const float km_ = (codd ? (bits >> 16) : (bits & 0xffffu)) >= thr ? ik : 0.f;
                            gr *= km_;
                            pd *= km_;
                        }
                        const float ds = pr * (gr - dv_[e]);
                        sacc[r] = MASKS ? (masked ? 0.f : ds) : ds;
                        pacc[r] = pd;
                    }
                }
            };
            if (need_mask) elements(std::true_type{});
            else elements(std::false_type{});
            __builtin_amdgcn_sched_barrier(0);
            // dV^T += dO^T P_d ; dK^T += Q^T dS  (k index = the 16 queries of k-step u = 2j, 2j+1)
#pragma unroll
            for (int uu = 0; uu < 2; ++uu) {
                const int r0 = uu * 8;
                const bf16x8 pf = __builtin_bit_cast(bf16x8, make_uint4(pack_bf16(pacc[r0], pacc[r0 + 1]), pack_bf16(pacc[r0 + 2], pacc[r0 + 3]),
                                                                         pack_bf16(pacc[r0 + 4], pacc[r0 + 5]), pack_bf16(pacc[r0 + 6], pacc[r0 + 7])));
                const bf16x8 dsf = __builtin_bit_cast(bf16x8, make_uint4(pack_bf16(sacc[r0], sacc[r0 + 1]), pack_bf16(sacc[r0 + 2], sacc[r0 + 3]),
                                                                          pack_bf16(sacc[r0 + 4], sacc[r0 + 5]), pack_bf16(sacc[r0 + 6], sacc[r0 + 7])));
                bf16x8 ot0, ot1, qt0, qt1;
                const uint32_t so = (uint32_t)(stage * 16384);
                const uint32_t ao0[2] = {otr[0][0] + so, otr[0][1] + so}, ao1[2] = {otr[1][0] + so, otr[1][1] + so};
                const uint32_t aq0[2] = {qtr[0][0] + so, qtr[0][1] + so}, aq1[2] = {qtr[1][0] + so, qtr[1][1] + so};
                switch (2 * j + uu) {
                case 0: TR_FRAG2(ot0, ao0, ot1, ao1, 0); break;
                case 1: TR_FRAG2(ot0, ao0, ot1, ao1, 1); break;
                case 2: TR_FRAG2(ot0, ao0, ot1, ao1, 2); break;
                default: TR_FRAG2(ot0, ao0, ot1, ao1, 3); break;
                }
                dvacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ot0, pf, dvacc[0], 0, 0, 0);
                dvacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ot1, pf, dvacc[1], 0, 0, 0);
                switch (2 * j + uu) {
                case 0: TR_FRAG2(qt0, aq0, qt1, aq1, 0); break;
                case 1: TR_FRAG2(qt0, aq0, qt1, aq1, 1); break;
                case 2: TR_FRAG2(qt0, aq0, qt1, aq1, 2); break;
                default: TR_FRAG2(qt0, aq0, qt1, aq1, 3); break;
                }
                dkacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt0, dsf, dkacc[0], 0, 0, 0);
                dkacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt1, dsf, dkacc[1], 0, 0, 0);
            }
        }
    }
    if (kvalid) {
        uint16_t *krow = (uint16_t *)p.dk + ex.dkv_off + (long long)key * p.dkv_ss + n * 64;
        uint16_t *vrow = (uint16_t *)p.dv + ex.dkv_off + (long long)key * p.dkv_ss + n * 64;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = j * 32 + 8 * g + 4 * hi;
                *(uint2 *)(krow + d) = make_uint2(pack_bf16(dkacc[j][4 * g] * p.scale, dkacc[j][4 * g + 1] * p.scale),
                                                  pack_bf16(dkacc[j][4 * g + 2] * p.scale, dkacc[j][4 * g + 3] * p.scale));
                *(uint2 *)(vrow + d) = make_uint2(pack_bf16(dvacc[j][4 * g], dvacc[j][4 * g + 1]), pack_bf16(dvacc[j][4 * g + 2], dvacc[j][4 * g + 3]));
            }
    }
}

} // namespace

static int attention_bwd_launch(const void *q, int64_t q_sb, int64_t q_ss, int64_t q_sn, const void *k, int64_t k_sb, int64_t k_ss, int64_t k_sn,
                                const void *v, int64_t v_sb, int64_t v_ss, int64_t v_sn, const void *o, const void *dout, void *dq, int64_t dq_sb,
                                int64_t dq_ss, void *dk, void *dv, int64_t dkv_sb, int64_t dkv_ss, const int64_t *ids_q, const int64_t *ids_k, const float *m, const float *l, float *dstat, int batch,
                                int heads, int sq, int sk, int head_dim, int causal, float scale, float drop_p, uint32_t seed, const int32_t *cu_q, const int32_t *cu_k,
                                int64_t total_q, double pairs, void *stream)
{
    if (!q || !k || !v || !o || !dout || !dq || !dk || !dv || !ids_q || !ids_k || !m || !l || !dstat || batch < 1 || heads < 1 || sq < 1) return -1;
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
    OpsTimer timer(OPS_ATTN_BWD, 10.0 * heads * pairs * 64, (hipStream_t)stream);
    const dim3 dq_grid(attn_grid((sq + BNW * 32 - 1) / (BNW * 32), batch * heads, heads));
    const size_t dq_lds = 3 * 16384 + (size_t)((sk + 63) / 64) * 8;
    if (drop_p > 0.f && causal) hipLaunchKernelGGL((attention_bwd_dq_kernel<true, true>), dq_grid, dim3(BNW * 64), dq_lds, (hipStream_t)stream, p);
    else if (drop_p > 0.f) hipLaunchKernelGGL((attention_bwd_dq_kernel<true, false>), dq_grid, dim3(BNW * 64), dq_lds, (hipStream_t)stream, p);
    else if (causal) hipLaunchKernelGGL((attention_bwd_dq_kernel<false, true>), dq_grid, dim3(BNW * 64), dq_lds, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((attention_bwd_dq_kernel<false, false>), dq_grid, dim3(BNW * 64), dq_lds, (hipStream_t)stream, p);
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
