// emdr2_amd/csrc/attention.hip -- fused attention forward for head dim 64 (include/emdr2_ops.h: emdr2_attention_fwd).
//
// softmax(mask(Q K^T / 8)) V for one (batch, head, 256-query block) per workgroup without materialising the [sq, sk] score
// matrix (reference: transformer.py:283-381 builds it with baddbmm, masked_fill(-10000), softmax, dropout, bmm: 20 GB per layer at
// 3,200 sequences of 512).  Flash-style online softmax in the "swapped" orientation, so nothing crosses lanes:
//   S^T[key, q] = K[key, :] . Q[q, :]     MFMA A = K block (LDS), B = Q fragments (registers)   -> a lane owns ONE query column
//   O^T[d, q]  += V^T[d, key] P^T[key, q]  MFMA A = V^T gathered from the [key][d] V tile with ds_read_b64_tr_b16, B = P^T straight from the
//                                          S^T accumulator registers
// The 32x32 C layout gives lane (q = lane&31, half = lane>>5) the keys {(r&3) + 8(r>>2) + 4*half}; the same key subset is used as
// the k-index of the second MFMA for both operands, so the accumulators of the first MFMA feed the second without a shuffle.
// Row max / row sum need one exchange between the two half-waves (lanes q and q+32 share a query).
// Masks come from token ids (pad id 0) + optional history mask; masked scores are REPLACED by -10000 like the reference.
// Optional dropout on the probabilities (after normalisation), counter-based: keep(seed, ((b*np+n)*sq+q)*sk+key).
// Inputs: q [b, sq, np, 64], k, v [b, sk, np, 64] strided views (last dim contiguous).
// Outputs: o [b, sq, np, 64] contiguous, m / l (row max, row sum of exp; fp32 [b, np, sq]) for the backward.
#include "../../include/emdr2_ops.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "rng.h"

#include "attention_common.h"
#include "ops_timing.h"

namespace {

struct AttnParams {
    const char *q, *k, *v;
    char *o;
    const long long *ids_q, *ids_k;
    float *m, *l;
    long long q_sb, q_ss, q_sn;   // element strides of q: batch, sequence, head
    long long k_sb, k_ss, k_sn, v_sb, v_ss, v_sn;
    int heads, sq, sk, causal, batch;
    float scale, drop_p;
    uint32_t seed;
    // packed ("varlen") operands: sequence b owns rows [cu[b], cu[b+1]) of a [rows, heads, 64] tensor (no batch stride, no padding rows);
    // NULL = the dense [batch, s] layout.  With cu_q the row statistics are laid out [heads, tq]; sq / sk are then the LONGEST sequence.
    const int *cu_q, *cu_k;
    long long tq;
    // split keys (few queries, very many keys: the FiD decoder's 32 positions over the ~20,000 packed keys of a question's K passages, where
    // one workgroup per (question, head) walking all key blocks leaves most of the chip idle): the key blocks of a (batch, head, query block)
    // are dealt to `ksplit` workgroups, each leaves its UNNORMALISED partial (O fp32, m in log2 units, l) in the workspace and
    // attention_combine_kernel folds them.  ksplit == 1: none of this.
    int ksplit;
    unsigned base_grid;           // workgroups of one split (attn_grid)
    float *part_o, *part_m, *part_l;   // [ksplit][stat_n][64], [ksplit][stat_n] x 2
    long long stat_n;
};

// exchange between lane i and lane i + 32 (the two halves of a wave hold the two key subsets of one query): v_permlane32_swap with the value
// in both operands leaves {lower half's x, upper half's x} in every lane -- one VALU op instead of a ds_bpermute round trip through the LDS
__device__ __forceinline__ float half_max(float x)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_sum(float x)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

#define KB 64   // keys per block
#define KSPLIT_BLOCKS 32   // key blocks (2,048 keys) per workgroup of a split-key launch: a sequence's ranges depend on ITS key count only
#define QW 32   // queries per wave
#define NW 4    // waves per workgroup
#define QB (NW * QW)  // queries per workgroup

// DROP / CAUSAL are compile-time: the instances without them carry neither the hash nor the history-mask arithmetic.  The kernel is VALU-bound
// (rocprofv3: ~35 VALU instructions per MFMA, VALU busy 4.6x the matrix pipe), so what matters is keeping the VALU issuing: a 64-key block is
// consumed as two 32-key online-softmax steps (16 score registers live instead of 32) and a workgroup is FOUR waves (128 queries), each
// workgroup with its own barrier, so that one workgroup's softmax arithmetic runs under another's MFMAs and DMA waits.  r04: FOUR workgroups per
// CU (4 waves per SIMD: <= 128 VGPRs -- 120 - 128 since the softmax section is one path, see there -- and 32 KiB of LDS each).
template <bool DROP, bool CAUSAL>
__global__ void __launch_bounds__(NW * 64, (DROP && CAUSAL) ? 3 : 4) attention_fwd_kernel(AttnParams p)   // (dropout + causal: 134 VGPRs)
{
    // LDS (dynamic): NST stages x (K tile + V tile, each [64 keys][64 d] 8 KiB, granules swizzled as in attention_common.h), then the
    // key-padding bits, one 64-bit word per 64-key block.  r03 ran THREE stages (two blocks of DMA in flight while one is consumed) at three
    // workgroups per CU; with the registers for a fourth, two stages and four workgroups measure better on every shape (context tower 1.07 ->
    // 0.99 ms, reader pairs 4.02 -> 3.92, dense 417 -> 428 TFLOP/s): the fetch of the next block is covered by three other workgroups now.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NST = 2;
    unsigned long long *kmask_s = (unsigned long long *)(smem + NST * 16384);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    int qblk, b, n;
    const int split = p.ksplit > 1 ? (int)(blockIdx.x / p.base_grid) : 0;
    if (!attn_decode(p.ksplit > 1 ? (int)(blockIdx.x - split * p.base_grid) : (int)blockIdx.x, (p.sq + QB - 1) / QB, p.batch * p.heads, p.heads, qblk, b, n)) return;
    // this sequence's extent: dense = [b * s, b * s + s) with a batch stride; packed = rows [cu[b], cu[b+1])
    int sq = p.sq, sk = p.sk;
    long long qrow0 = (long long)b * p.sq, krow0 = (long long)b * p.sk;            // first row of the sequence in ids / o / (k)
    long long q_off = (long long)b * p.q_sb, k_off = (long long)b * p.k_sb, v_off = (long long)b * p.v_sb;
    long long stat0 = ((long long)b * p.heads + n) * p.sq;
    if (p.cu_q) { const int c0 = p.cu_q[b]; sq = p.cu_q[b + 1] - c0; qrow0 = c0; q_off = (long long)c0 * p.q_ss; stat0 = (long long)n * p.tq + c0; }
    if (p.cu_k) { const int c0 = p.cu_k[b]; sk = p.cu_k[b + 1] - c0; krow0 = c0; k_off = (long long)c0 * p.k_ss; v_off = (long long)c0 * p.v_ss; }
    if (qblk * QB >= sq) return;                                                  // (packed: a block past the end of a short sequence)
    if (sk < 1) {
        // a key sequence without keys: the unsplit launch leaves o / m / l untouched.  A split launch's combine kernel reads EVERY split's
        // slot of the (uninitialised) workspace: leave the "empty range" partial (-3e38, 0, 0) there; with all ranges empty the combine
        // kernel finds L == 0 and leaves the row untouched as well
        const int qz = qblk * QB + wave * QW + l31;
        if (p.ksplit > 1 && qz < sq) {
            const long long si = (long long)split * p.stat_n + stat0 + qz;
            float4 *z = (float4 *)(p.part_o + si * 64 + 32 * hi);
#pragma unroll
            for (int i = 0; i < 8; ++i) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (hi == 0) { p.part_m[si] = -3.0e38f; p.part_l[si] = 0.f; }
        }
        return;
    }
    const int q0 = qblk * QB + wave * QW;
    const int qi = q0 + l31;                       // this lane's query
    const bool qvalid = qi < sq;
    const int qc = qvalid ? qi : sq - 1;
    const bool wave_live = q0 < sq;                 // a wave past the end of its sequence only helps to stage K / V (decoder: sq = 32 of 128)

    // LDS-DMA: per block 16 pieces of 1 KiB (8 for K, 8 for V); wave w moves pieces w, w + NW, .. of each: 8 rows x 128 B,
    // lane -> (row = 8 piece + lane>>3, LDS granule = lane&7), source granule = granule ^ tile_swz(row) (tile_swz looks at row bits 1..3: period 16, and a wave's pieces are 32 rows apart)
    const int prow = wave * 8 + (lane >> 3), pslot = (lane & 7) ^ tile_swz(wave * 8 + (lane >> 3));
    const char *k_src = p.k + (k_off + (long long)n * p.k_sn) * 2 + pslot * 16;        // + key * k_ss * 2
    const char *v_src = p.v + (v_off + (long long)n * p.v_sn) * 2 + pslot * 16;
    const int nblk_all = (sk + KB - 1) / KB;
    // this workgroup's key blocks [blk_lo, nblk): all of them, or its share of a split launch (possibly none: more splits than blocks)
    const int blk_lo = p.ksplit > 1 ? min(split * KSPLIT_BLOCKS, nblk_all) : 0;
    const int nblk = p.ksplit > 1 ? min((split + 1) * KSPLIT_BLOCKS, nblk_all) : nblk_all;
    auto issue = [&](int blk, int stage) {                                     // dense: sk % 32 == 0 (checked on the host); packed: any sk
        char *sb = smem + stage * 16384;
#pragma unroll
        for (int i = 0; i < 8 / NW; ++i) {
            long long key = blk * KB + prow + 8 * NW * i;
            if (key >= sk) key = sk - 1;                                           // keys past sk: re-read the last row, masked below (kmask bit 0)
            __builtin_amdgcn_global_load_lds((gptr_t *)(k_src + key * p.k_ss * 2), (lptr_t *)(sb + (wave + NW * i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t *)(v_src + key * p.v_ss * 2), (lptr_t *)(sb + 8192 + (wave + NW * i) * 1024), 16, 0, 0);
        }
    };
    // the first K / V block goes out before anything else touches global memory: short (packed) sequences have only 2 - 4 blocks per
    // workgroup, and a prologue that first waits for its Q rows and key ids and only then starts the DMA pays the memory latency twice
    if (blk_lo < nblk) issue(blk_lo, 0);

    // Q fragments (B operand of S^T = K Q^T): for k-step t the lane holds d = 16t + 8*half .. +8
    const char *qrow = p.q + (q_off + (long long)qc * p.q_ss + (long long)n * p.q_sn) * 2;
    bf16x8 qf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) qf[t] = *(const bf16x8 *)(qrow + (16 * t + 8 * hi) * 2);
    const bool qpad = !qvalid || p.ids_q[qrow0 + qc] == 0;

    uint32_t vtr[2][2], kra[4];
    tr_addresses((uint32_t)(uintptr_t)smem + 8192, lane, vtr);
    row_frag_addresses((uint32_t)(uintptr_t)smem, lane, kra);
    for (int blk = blk_lo + wave; blk < nblk; blk += NW) {
        const int key = blk * KB + lane;
        const unsigned long long w = __builtin_amdgcn_ballot_w64(key < sk && p.ids_k[krow0 + (key < sk ? key : sk - 1)] != 0);
        if (lane == 0) kmask_s[blk] = w;
    }

    floatx16 oacc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[j][r] = 0.f;
    // Softmax bookkeeping in the log2 domain: s2 = score * scale * log2(e), p = exp2(s2 - m2): one FMA + one v_exp per element.
    // A padded query has every score replaced by -10000: per-lane (scale, offset) = (0, -10000 log2 e) does that without a select.
    const float sc_q = qpad ? 0.f : p.scale * L2E, c_q = qpad ? MASKED2 : 0.f;
    const bool any_qpad = __builtin_amdgcn_ballot_w64(qpad) != 0ull;
    const float raw_masked = MASKED2 / (p.scale * L2E);                       // the raw score that a real query's affine map sends to the mask value
    float mrun = -3.0e38f, lrun = 0.f;
    const float ik = DROP ? emdr2_keep_scale(p.drop_p) : 1.f;
    const uint32_t thr = emdr2_drop_thr(p.drop_p);
    const uint32_t rh = emdr2_row_hash(p.seed, (unsigned long long)(stat0 + qc));

    // everything the prologue fetched is complete HERE, stated in a form the compiler sees (the asm "defines" the Q fragments): left to
    // itself it would put the wait for the Q rows at their first use inside the loop, as vmcnt(0) -- and drain the look-ahead DMA with it
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3])::"memory");
    int stage = 0;
    for (int blk = blk_lo; blk < nblk; ++blk, stage ^= 1) {
        // this block's pieces are the newest DMA of this wave (the next block's go out after the barrier)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // raw barrier: what must be ordered here is LDS only -- the mask words of the prologue and (through each wave's own wait above) the tiles
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                             // every wave's pieces have landed; everyone is done with the block before
        __builtin_amdgcn_sched_barrier(0);
        if (blk + 1 < nblk) issue(blk + 1, stage ^ 1);                            // into the stage the previous block has just left
        if (!wave_live) continue;
        const unsigned long long kmask = kmask_s[blk];
        const int key0 = blk * KB;
        // A block whose keys are all masked for every query of this wave contributes exp(-10000 - m) == 0 exactly once each query has
        // seen a real key (m > -7000): skip it.  Rows that are masked everywhere (padded queries) need every block, hence any_qpad.
        if ((kmask == 0ull || (CAUSAL && key0 > q0 + QW - 1)) && !any_qpad && blk > blk_lo &&
            __builtin_amdgcn_ballot_w64(mrun > -7000.f) == ~0ull)
            continue;
        const uint32_t av0[2] = {vtr[0][0] + (uint32_t)(stage * 16384), vtr[0][1] + (uint32_t)(stage * 16384)};
        const uint32_t av1[2] = {vtr[1][0] + (uint32_t)(stage * 16384), vtr[1][1] + (uint32_t)(stage * 16384)};

#pragma unroll
        for (int j = 0; j < 2; ++j) {                                             // two 32-key online-softmax steps per staged block
            const uint32_t km = (uint32_t)(kmask >> (32 * j));
            const int kb0 = key0 + 32 * j;
            if (kb0 >= sk) continue;                                              // the second half of the last block does not exist
            if ((km == 0u || (CAUSAL && kb0 > q0 + QW - 1)) && !any_qpad && (blk > blk_lo || j > 0) &&
                __builtin_amdgcn_ballot_w64(mrun > -7000.f) == ~0ull)
                continue;
            auto dropout_and_pv = [&](floatx16 &sacc) {
                if (DROP) {                                                       // attention dropout after the normaliser (l is un-dropped)
                    const uint32_t prod0 = ((uint32_t)(kb0 + 4 * hi) >> 1) * EMDR2_PAIR_MUL;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const uint32_t b0 = emdr2_pair_bits_prod(rh, prod0 + (uint32_t)(4 * g) * EMDR2_PAIR_MUL);
                        const uint32_t b1 = emdr2_pair_bits_prod(rh, prod0 + (uint32_t)(4 * g + 1) * EMDR2_PAIR_MUL);
                        sacc[4 * g] = (b0 & 0xffffu) >= thr ? sacc[4 * g] : 0.f; // (the 1 / (1 - p) of the survivors is applied once, to O)
                        sacc[4 * g + 1] = (b0 >> 16) >= thr ? sacc[4 * g + 1] : 0.f;
                        sacc[4 * g + 2] = (b1 & 0xffffu) >= thr ? sacc[4 * g + 2] : 0.f;
                        sacc[4 * g + 3] = (b1 >> 16) >= thr ? sacc[4 * g + 3] : 0.f;
                    }
                }
                // ---- O^T += V^T P^T : 2 d sub-blocks x 2 k-steps (16 keys each: registers 8w..8w+7) -----------------------------
#pragma unroll
                for (int w2 = 0; w2 < 2; ++w2) {
                    uint32_t pw[4];
#pragma unroll
                    for (int w = 0; w < 4; ++w) pw[w] = pack_bf16(sacc[w2 * 8 + 2 * w], sacc[w2 * 8 + 2 * w + 1]);
                    const bf16x8 pf = __builtin_bit_cast(bf16x8, make_uint4(pw[0], pw[1], pw[2], pw[3]));
                    // V^T fragments for the same key subset (rows 16u + 4*half + {0..3, 8..11} of the V tile, u = 2j + w2) by transpose reads
                    bf16x8 vt0, vt1;
                    if (j == 0 && w2 == 0) TR_FRAG2(vt0, av0, vt1, av1, 0);
                    else if (j == 0) TR_FRAG2(vt0, av0, vt1, av1, 1);
                    else if (w2 == 0) TR_FRAG2(vt0, av0, vt1, av1, 2);
                    else TR_FRAG2(vt0, av0, vt1, av1, 3);
                    oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vt0, pf, oacc[0], 0, 0, 0);
                    oacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vt1, pf, oacc[1], 0, 0, 0);
                }
            };
            // ---- S^T = K Q^T : 32 keys x 4 k-steps -------------------------------------------------------------------------------
            const floatx16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // inline-constant C operand: no v_mov per register
            // two fragments at a time: all four at once cost eight more live registers than the 168 of three waves per SIMD allow
            const uint32_t so = (uint32_t)(stage * 16384);
            bf16x8 kr[2];
            floatx16 sacc;
#pragma unroll
            for (int tp = 0; tp < 2; ++tp) {
                if (j == 0) { LDS_READ128(kr[0], kra[2 * tp] + so, 0); LDS_READ128(kr[1], kra[2 * tp + 1] + so, 0); }
                else { LDS_READ128(kr[0], kra[2 * tp] + so, 4096); LDS_READ128(kr[1], kra[2 * tp + 1] + so, 4096); }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kr[0]), "+v"(kr[1])::"memory");
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kr[0], qf[2 * tp], tp == 0 ? zero : sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kr[1], qf[2 * tp + 1], sacc, 0, 0, 0);
            }
            // ---- mask, online softmax (this lane: one query, 16 of the 32 keys; its half-wave partner holds the other 16) --------
            // ONE path with one-armed, wave-uniform `if`s around the rare parts, each of which touches the score registers only (or, the
            // rescale, O in place).  r01-r03 ran two instantiations of the whole section, `if (need_mask) A else B`, each with the
            // rescale and the PV product inside: LLVM structurizes a region with more than one conditional child even when every branch
            // is wave-uniform, a structurized if / else runs its arms one after the other as far as register allocation is concerned, O
            // had to survive the arm that "does not run" -- and every 32-key step copied the 32 registers of O to a second set and back
            // (32 v_mov_b64 of ~210 VALU instructions on the mask-free path).  Masks are applied to the RAW scores: a masked key gets the
            // raw value that the affine map below sends to -10000 log2(e) (a real query; a padded query has scale 0 and lands there
            // whatever the raw value), keys past the end of a packed sequence hold a copy of the last real key's score (the DMA re-reads
            // that row), so they cannot disturb the max, and are zeroed after the exponential: probability 0 even for a padded query,
            // so the forward agrees with the backward, which never stores such a key's dK / dV.
            if (km != 0xffffffffu || (CAUSAL && kb0 + 31 > q0)) {                  // wave-uniform
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kl = (r & 3) + 8 * (r >> 2) + 4 * hi;              // key inside the sub-block
                    const bool masked = !((km >> kl) & 1u) || (CAUSAL && kb0 + kl > qi);
                    sacc[r] = masked ? raw_masked : sacc[r];
                }
            }
            float mx = sacc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sacc[r]);
            const float bmax = half_max(fmaf(mx, sc_q, c_q));                    // sc_q >= 0: max commutes with the affine map
            // LAZY running max: the reference point of the exponentials only moves when some query's block max exceeds it by more than
            // 2^8 -- with 32 queries per wave the exact max moves in almost every step (and each move costs a rescale of the 32 O
            // registers: 3 of 10 VALU slots per score), a jump of 8 happens in the first step and then practically never.  Until then
            // exp2(s - m) <= 256: nothing for fp32 sums or bf16 probabilities (same relative precision), and O / l is unchanged
            // mathematically; (m, l) stay consistent for the backward, which only ever uses m + log2 l.
            if (__builtin_amdgcn_ballot_w64(bmax > mrun + 8.f) != 0ull) {
                // r05: the reference point is a WHOLE number of binades: exp2(s - m) then has the same mantissa whatever m a walk over
                // the keys arrives at, so the bf16 probabilities that enter P V -- and with them O up to the order of fp32 additions --
                // do not depend on where a walk starts: key-split launches, packed vs dense layouts and groups of any size agree
                const float mnew = __builtin_ceilf(fmaxf(mrun, bmax));
                const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
                lrun *= alpha;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[jj][r] *= alpha;
                mrun = mnew;
            }
            const float off = c_q - mrun;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = __builtin_amdgcn_exp2f(fmaf(sacc[r], sc_q, off));
            if (kb0 + 32 > sk) {                                                    // (packed: the ragged end of the sequence)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[r] = kb0 + (r & 3) + 8 * (r >> 2) + 4 * hi < sk ? sacc[r] : 0.f;
            }
            float bsum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) bsum += sacc[r];
            lrun += half_sum(bsum);
            dropout_and_pv(sacc);
        }
    }

    if (p.ksplit > 1) {
        // partial of this key range: unnormalised O, reference point m (log2 units), sum l -- (-3e38, 0, 0) when the range was empty
        if (qvalid) {
            const long long si = (long long)split * p.stat_n + stat0 + qi;
            float *orow = p.part_o + si * 64;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *(float4 *)(orow + j * 32 + 8 * g + 4 * hi) = make_float4(oacc[j][4 * g], oacc[j][4 * g + 1], oacc[j][4 * g + 2], oacc[j][4 * g + 3]);
            if (hi == 0) { p.part_m[si] = mrun; p.part_l[si] = lrun; }
        }
        return;
    }
    // ---- normalise and store O[q, d]: lane q holds d = j*32 + (r&3) + 8(r>>2) + 4*half --------------------------------------
    if (qvalid) {
        const float inv = ik / lrun;                                 // survivors of the attention dropout are scaled here, once
        uint16_t *orow = (uint16_t *)p.o + ((qrow0 + qi) * p.heads + n) * 64;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = j * 32 + 8 * g + 4 * hi;
                const uint32_t w0 = pack_bf16(oacc[j][4 * g] * inv, oacc[j][4 * g + 1] * inv);
                const uint32_t w1 = pack_bf16(oacc[j][4 * g + 2] * inv, oacc[j][4 * g + 3] * inv);
                *(uint2 *)(orow + d) = make_uint2(w0, w1);
            }
        if (hi == 0 && p.m) {
            const long long si = stat0 + qi;
            p.m[si] = mrun * 0.6931471805599453f; p.l[si] = lrun;        // back to natural-log units
        }
    }
}

// Fold the partials of a split launch: O = sum_s 2^(m_s - M) O_s / sum_s 2^(m_s - M) l_s (x 1 / (1 - p) of the attention dropout), statistics
// (M ln 2, L) exactly as one workgroup walking all the keys leaves them.  One wave per query row (dense-q launches only): lane = d.
__global__ void __launch_bounds__(256) attention_combine_kernel(AttnParams p, float keep_scale)
{
    const long long si = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int d = threadIdx.x & 63;
    if (si >= p.stat_n) return;
    float M = -3.0e38f;
    for (int s = 0; s < p.ksplit; ++s) M = fmaxf(M, p.part_m[(long long)s * p.stat_n + si]);
    float L = 0.f, acc = 0.f;
    for (int s = 0; s < p.ksplit; ++s) {
        const long long i = (long long)s * p.stat_n + si;
        const float w = __builtin_amdgcn_exp2f(p.part_m[i] - M);
        L += w * p.part_l[i];
        acc += w * p.part_o[i * 64 + d];
    }
    // si = (b * heads + n) * sq + q  ->  o row (b * sq + q) * heads + n
    if (!(L > 0.f)) return;                                                      // a sequence without keys: untouched, like the unsplit launch
    const long long bn = si / p.sq, q = si - bn * p.sq, b = bn / p.heads, n = bn - b * p.heads;
    ((__bf16 *)p.o)[((b * p.sq + q) * p.heads + n) * 64 + d] = (__bf16)(acc * (keep_scale / L));
    if (d == 0 && p.m) { p.m[si] = M * 0.6931471805599453f; p.l[si] = L; }
}

} // namespace

// Whether (and how many ways) a launch splits its keys: ONE query block (<= 128 dense queries) over sequences of 4,096 keys or more, in
// ranges of KSPLIT_BLOCKS blocks.  A function of the launch's query / key EXTENTS only, never of the batch: the same question gives the same
// bits whether it runs alone, in a group of 8 or in a batch of 64 (question micro-batching relies on it), and short-key launches -- every
// shape the bf16-faithful oracle walks step by step -- never split.
static int attention_ksplit(int max_sq, int max_sk)
{
    if (max_sq > QB || max_sk < 4096) return 1;
    const int nblk = (max_sk + KB - 1) / KB;
    return (nblk + KSPLIT_BLOCKS - 1) / KSPLIT_BLOCKS;                            // <= 32 (sk <= 65,536)
}

static int attention_fwd_launch(const void *q, int64_t q_sb, int64_t q_ss, int64_t q_sn, const void *k, int64_t k_sb, int64_t k_ss, int64_t k_sn,
                                const void *v, int64_t v_sb, int64_t v_ss, int64_t v_sn, void *o, const int64_t *ids_q, const int64_t *ids_k, int batch, int heads, int sq, int sk,
                                int head_dim, int causal, float scale, float drop_p, uint32_t seed, float *m, float *l, const int32_t *cu_q, const int32_t *cu_k,
                                int64_t total_q, double pairs, void *stream, int ksplit = 1, void *ws = nullptr, size_t ws_bytes = 0)
{
    if (!q || !k || !v || !o || !ids_q || !ids_k || batch < 1 || heads < 1 || sq < 1) return -1;
    if (ksplit < 1 || ksplit > 64 || (ksplit > 1 && (cu_q || !ws || ((uintptr_t)ws & 15) || ksplit != attention_ksplit(sq, sk)))) return -1;
    if (head_dim != 64 || sk < 1 || sk > 65536 || (q_ss & 7) || (k_ss & 7) || (q_sn & 7) || (k_sn & 7) || (q_sb & 7) || (k_sb & 7) || (v_sb & 7) || (v_ss & 7) || (v_sn & 7)) return -4;
    if (!cu_k && (sk < 32 || (sk & 31))) return -4;                                  // dense keys: whole 32-key steps (padded queries average over exactly sk keys)
    if (cu_q && total_q < 1) return -1;
    if (((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)v & 15) || ((uintptr_t)o & 7) || drop_p < 0.f || drop_p >= 1.f) return -1;
    AttnParams p;
    p.q = (const char *)q; p.k = (const char *)k; p.v = (const char *)v; p.o = (char *)o;
    p.ids_q = (const long long *)ids_q; p.ids_k = (const long long *)ids_k; p.m = m; p.l = l;
    p.q_sb = q_sb; p.q_ss = q_ss; p.q_sn = q_sn; p.k_sb = k_sb; p.k_ss = k_ss; p.k_sn = k_sn; p.v_sb = v_sb; p.v_ss = v_ss; p.v_sn = v_sn;
    p.heads = heads; p.sq = sq; p.sk = sk; p.causal = causal; p.scale = scale; p.drop_p = drop_p; p.seed = seed;
    p.batch = batch; p.cu_q = cu_q; p.cu_k = cu_k; p.tq = total_q;
    p.ksplit = ksplit; p.base_grid = attn_grid((sq + QB - 1) / QB, batch * heads, heads);
    p.stat_n = (long long)batch * heads * sq;
    p.part_o = p.part_m = p.part_l = nullptr;
    if (ksplit > 1) {
        if (ws_bytes < (size_t)ksplit * p.stat_n * 66 * sizeof(float)) return -2;
        p.part_o = (float *)ws; p.part_m = p.part_o + (size_t)ksplit * p.stat_n * 64; p.part_l = p.part_m + (size_t)ksplit * p.stat_n;
    }
    dim3 grid(p.base_grid * (unsigned)ksplit);
    const size_t lds = 2 * 16384 + (size_t)((sk + KB - 1) / KB) * 8;            // <= 40 KiB (sk <= 65536)
    OpsTimer timer(OPS_ATTN_FWD, 4.0 * heads * pairs * 64, (hipStream_t)stream);
    if (drop_p > 0.f && causal) hipLaunchKernelGGL((attention_fwd_kernel<true, true>), grid, dim3(NW * 64), lds, (hipStream_t)stream, p);
    else if (drop_p > 0.f) hipLaunchKernelGGL((attention_fwd_kernel<true, false>), grid, dim3(NW * 64), lds, (hipStream_t)stream, p);
    else if (causal) hipLaunchKernelGGL((attention_fwd_kernel<false, true>), grid, dim3(NW * 64), lds, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((attention_fwd_kernel<false, false>), grid, dim3(NW * 64), lds, (hipStream_t)stream, p);
    if (ksplit > 1)
        hipLaunchKernelGGL(attention_combine_kernel, dim3((unsigned)((p.stat_n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p,
                           drop_p > 0.f ? emdr2_keep_scale(drop_p) : 1.f);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

extern "C" int emdr2_attention_fwd(const void *q, int64_t q_sb, int64_t q_ss, int64_t q_sn, const void *k, int64_t k_sb, int64_t k_ss, int64_t k_sn,
                                   const void *v, int64_t v_sb, int64_t v_ss, int64_t v_sn, void *o, const int64_t *ids_q, const int64_t *ids_k, int batch, int heads, int sq, int sk,
                                   int head_dim, int causal, float scale, float drop_p, uint32_t seed, float *m, float *l, void *stream)
{
    return attention_fwd_launch(q, q_sb, q_ss, q_sn, k, k_sb, k_ss, k_sn, v, v_sb, v_ss, v_sn, o, ids_q, ids_k, batch, heads, sq, sk, head_dim, causal, scale, drop_p,
                                seed, m, l, nullptr, nullptr, 0, (double)batch * sq * sk, stream);
}

// Split-key plan of a launch with DENSE queries [batch, max_sq] over up to max_sk keys per sequence: number of splits (1 = do not split) and
// the workspace bytes the forward / backward launches then need (emdr2_ops.h).
extern "C" int emdr2_attention_splitkv_plan(int batch, int heads, int max_sq, int max_sk, int *ksplit, size_t *fwd_bytes, size_t *bwd_bytes)
{
    if (!ksplit || !fwd_bytes || !bwd_bytes || batch < 1 || heads < 1 || max_sq < 1 || max_sk < 1) return -1;
    const int ks = attention_ksplit(max_sq, max_sk);
    const size_t stat_n = (size_t)batch * heads * max_sq;
    *ksplit = ks;
    *fwd_bytes = ks > 1 ? (size_t)ks * stat_n * 66 * sizeof(float) : 0;
    *bwd_bytes = ks > 1 ? (size_t)ks * stat_n * 64 * sizeof(float) : 0;
    return 0;
}

extern "C" int emdr2_attention_fwd_splitkv(const void *q, int64_t q_sb, int64_t q_ss, int64_t q_sn, const void *k, int64_t k_sb, int64_t k_ss, int64_t k_sn,
                                                  const void *v, int64_t v_sb, int64_t v_ss, int64_t v_sn, void *o, const int64_t *ids_q, const int64_t *ids_k,
                                                  const int32_t *cu_k, int64_t pairs, int batch, int heads, int sq, int max_sk,
                                                  int head_dim, int causal, float scale, float drop_p, uint32_t seed, float *m, float *l, int ksplit, void *ws,
                                                  size_t ws_bytes, void *stream)
{
    return attention_fwd_launch(q, q_sb, q_ss, q_sn, k, k_sb, k_ss, k_sn, v, v_sb, v_ss, v_sn, o, ids_q, ids_k, batch, heads, sq, max_sk, head_dim, causal, scale,
                                drop_p, seed, m, l, nullptr, cu_k, 0, (double)pairs, stream, ksplit, ws, ws_bytes);
}

extern "C" int emdr2_attention_varlen_fwd(const void *q, int64_t q_sb, int64_t q_ss, int64_t q_sn, const void *k, int64_t k_sb, int64_t k_ss, int64_t k_sn,
                                          const void *v, int64_t v_sb, int64_t v_ss, int64_t v_sn, void *o, const int64_t *ids_q, const int64_t *ids_k,
                                          const int32_t *cu_q, const int32_t *cu_k, int64_t total_q, int64_t pairs, int batch, int heads, int max_sq, int max_sk,
                                          int head_dim, int causal, float scale, float drop_p, uint32_t seed, float *m, float *l, void *stream)
{
    if (!cu_q && !cu_k) return -1;
    return attention_fwd_launch(q, q_sb, q_ss, q_sn, k, k_sb, k_ss, k_sn, v, v_sb, v_ss, v_sn, o, ids_q, ids_k, batch, heads, max_sq, max_sk, head_dim, causal, scale,
                                drop_p, seed, m, l, cu_q, cu_k, total_q, (double)pairs, stream);
}
