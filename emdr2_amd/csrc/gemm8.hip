// emdr2_amd/csrc/gemm8.hip -- persistent bf16 NT GEMM for the large linears of the EMDR2 step (include/emdr2_ops.h: emdr2_gemm_nt_bf16
// dispatches here when M % 256 == N % 256 == 0, K % 128 == 0, unbatched, bf16 output).
//
//   C[m, n] = epilogue( alpha * sum_k A[m, k] * B[n, k] )         A: [M, K] row-major (lda), B: [N, K] row-major (ldb)
//
// Replaces F.linear + bias(+GELU) / bias-dropout-add of the reference (megatron/mpu/layers.py:255,353, transformer.py:94-108,397-413) at
// the shapes that carry the step: M = 1.6M / 0.8M tokens, N, K in {768, 1536, 2304, 3072}.
//
// Structure (one workgroup of 8 waves per CU, walking its share of the 256 x 256 output tiles):
//  * K-tile = 64: an LDS row is one full 128-byte line of the operand; two K-tile buffers of four 16 KiB half-tiles each
//    (A rows for the first / second 64-row half of every wave's 128 output rows, B rows for the first / second 32 columns of its 64).
//  * a wave owns 128 x 64 outputs = 8 x 4 accumulators of v_mfma_f32_16x16x32_bf16 (r04; 4 x 2 of the 32 x 32 x 16 shape before: same
//    fragment traffic, half the accumulator registers moved per flop, and at the board's power cap the 16 x 16 x 32 shape sustains 19 % more
//    flops, tools/mfma_peak.hip), computed as D^T (B fragment first): a lane then holds 4 consecutive COLUMNS of one output row per
//    accumulator, so the epilogue packs bf16 pairs without cross-lane traffic.
//  * four phases per K-tile, one C quadrant (64 x 32) x K = 64 each: 16 MFMAs behind 12 / 4 / 8 / 0 fragment reads; every phase also issues
//    one half-tile of LDS-DMA (2 x 1 KiB per wave) SIX half-tiles ahead of its consumer and waits with a counted vmcnt(8); a half-tile's slot is
//    re-used two or three phases after its last read, which is what lets two buffers run a prefetch distance of more than one K-tile.
//  * waves 0-3 and 4-7 (one of each per SIMD) run ONE BARRIER APART: while one half issues its MFMAs the other issues its LDS reads and DMA,
//    so the matrix pipe of a SIMD alternates between its two waves instead of both stalling on the same barrier.
//  * persistent: the DMA stream runs on across tile seams (the next tile's first six half-tiles are in flight during the epilogue), the
//    epilogue stages through its own 32 KiB of LDS (the ring stays live), both wave halves run their epilogues in the same barrier interval.
//  * XCD-aware tile order as in gemm.hip: each XCD walks one contiguous range of the n-fastest tile sequence, its 32 CUs take consecutive
//    tiles, so the A panel shared by the n-tiles of an m-row leaves HBM once and stays in that XCD's L2.
// Epilogue: alpha, + bias[n], optional pre-activation output, exact-erf GELU, dropout (csrc/rng.h), then through LDS into row order:
// + residual[m, n] (or * gelu'(residual): the fused GELU backward), bf16, 16-byte streaming stores of whole 128-byte row segments.
// Optional LSE mode (the tied LM head, language_model.py:28-41 + train_e2eqa.py:79-96): instead of storing the logits the epilogue reduces
// each row's 256-column tile to (max, sum exp) and picks the gold logit, so the [tokens, vocab] matrix never reaches HBM.
#include "../../include/emdr2_ops.h"
#include "gemm_common.h"
#include "exp_hooks.h"
#include "ops_timing.h"
#include <stdlib.h>

namespace {

struct G8Params {
    const char *A, *B;
    char *C, *C2;              // C2: optional pre-activation output (bf16)
    const float *bias;
    const char *R;             // optional residual (bf16, indexed like C)
    int rmode;                 // 0: + R;  1: * gelu'(R);  2: * R
    long long lda, ldb, ldc;   // elements
    int M, N, K;
    float alpha;
    int gelu;
    float drop_p;
    uint32_t seed;
    int tiles_m, tiles_n, ngroup, total, per;   // tile order: `per` consecutive sequence positions per XCD
    uint32_t mg_full, mg_group, mg_last;        // ceil(2^32 / d) for d = ngroup * tiles_m, ngroup, width of the last group (tile_coords corrects the +1)
    // LSE mode (lse_part != nullptr): nothing is stored to C; per (row, n-tile) partial max / sum-exp and the gold logit are written instead
    float *lse_max, *lse_sum, *lse_gold;
    const long long *labels;
    int stagger;               // XCD start stagger: 1/64ths of a 1,024-cycle nap per K-tile and XCD index, 0 = off
    int ablate;                // -DEMDR2_EXPERIMENTS builds only (EMDR2_G8_ABLATE): 1 = no epilogue, 2 = epilogue without global stores
    int nt_a;                  // A half-tiles with the non-temporal cache policy (an A panel is used by the n-tiles of ONE round of its XCD, then never again)
};

__device__ __forceinline__ void tile_coords(const G8Params &p, int pos, int &tm, int &tn)
{
    // n-tiles in groups of `ngroup` B panels that fit the L2; inside a group the walk is n-fastest (gemm.hip has the rationale)
    const int full = p.ngroup * p.tiles_m;
    // quotient by the rounded-up reciprocal ceil(2^32 / d): never too small, and too large by at most one (the excess is < pos / 2^32 < 1) --
    // which it IS once pos * (mg * d - 2^32) reaches 2^32, e.g. from tile 59,075 on for d = 6 x 12,800 tiles (M = 3.3M rows, N = 3072: top-k
    // 100): one compare puts it right for every pos < 2^32
    int g = (int)__umulhi((uint32_t)pos, p.mg_full);
    if (g * full > pos) --g;
    const int r = pos - g * full;
    const bool whole = (g + 1) * p.ngroup <= p.tiles_n;
    const int gsize = whole ? p.ngroup : p.tiles_n - g * p.ngroup;
    const uint32_t mg = whole ? p.mg_group : p.mg_last;
    tm = mg ? (int)__umulhi((uint32_t)r, mg) : r;             // mg == 0 encodes a divisor of 1
    if (tm * gsize > r) --tm;
    tn = g * p.ngroup + (r - tm * gsize);
}

#define G8_BUF 65536
#define G8_SLOT 16384
#define G8_STAGING (2 * G8_BUF)

// EPI: compile-time epilogue recipe (a runtime-branched epilogue kept so many paths live that the register allocator spilled the residual
// prefetch and the accumulators).  The dispatcher instantiates the combinations the EMDR2 step uses; anything else falls back to gemm.hip.
enum { G8_BIAS = 1, G8_GELU = 2, G8_DROP = 4, G8_RADD = 8, G8_RGELU = 16, G8_PRE = 32, G8_LSE = 64, G8_PREG = 128, G8_RMUL = 256 };   // PREG (with PRE): C2 receives gelu' of the pre-activation

template <int EPI>
__global__ void __launch_bounds__(512) gemm8_kernel(G8Params p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;                 // wave grid 2 (m) x 4 (n); wr is also the half that runs one barrier behind
    const int l31 = lane & 31, hi = lane >> 5;

    // ---- this workgroup's tiles: XCD x = id & 7 owns sequence positions [x * per, (x + 1) * per), its workgroups take them round-robin
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, wg_per_xcd = gridDim.x >> 3;
    const int seq_lo = xcd * p.per;
    int seq_hi = seq_lo + p.per; if (seq_hi > p.total) seq_hi = p.total;
    const int first = seq_lo + slot;
    if (first >= seq_hi) return;
    const int my_count = (seq_hi - first + wg_per_xcd - 1) / wg_per_xcd;
    const int KT = p.K >> 6;
    const long long lda2 = p.lda * 2, ldb2 = p.ldb * 2;

    // ---- LDS-DMA addressing.  A half-tile is 128 LDS rows of 128 B; a DMA instruction writes 8 rows (lane -> row lane>>3, 16-B slot lane&7);
    // wave w fills rows [8w, 8w+8) and [64+8w, 64+8w+8).  The 16-B groups of a row are XOR-swizzled with ((row >> 1) & 7) on the SOURCE side
    // (the destination of a DMA is lane-linear), which makes the 32-row ds_read_b128 fragment reads below conflict-free.
    const int si = lane >> 3;
    const int sgrp = (lane & 7) ^ (((wave & 1) * 4 + (si >> 1)) & 7);
    const uint32_t offA = (uint32_t)((8 * wave + si) * lda2 + sgrp * 16);                                  // + 64 rows: second half (mh), + 128: rows of wr = 1
    const uint32_t offB = (uint32_t)(((wave >> 2) * 64 + 8 * (wave & 3) + si) * ldb2 + sgrp * 16);      // + 32 rows: nh = 1, + 128: waves wc 2,3
    // stage cursor (wave-uniform): tile `s_i` of this workgroup's list, K-tile `s_kt`
    int s_i = 0, s_kt = 0;
    const char *sA, *sB;
    auto cursor_tile = [&](int i) {
        if (i >= my_count) i = my_count - 1;                  // past the end: harmless re-reads keep the vmcnt arithmetic fixed
        int tm, tn;
        tile_coords(p, first + i * wg_per_xcd, tm, tn);
        sA = p.A + (long long)tm * 256 * lda2;
        sB = p.B + (long long)tn * 256 * ldb2;
    };
    cursor_tile(0);
#define G8_STAGE(T, SB)                                                                                                                   \
    do {                                                                                                                                  \
        const char *src_ = ((T) == 0 || (T) == 3) ? sA + ((T) == 3 ? 64 * lda2 : 0) : sB + ((T) == 2 ? 32 * ldb2 : 0);                    \
        const uint32_t off_ = ((T) == 0 || (T) == 3) ? offA : offB;                                                                       \
        const long long half_ = ((T) == 0 || (T) == 3) ? 128 * lda2 : 128 * ldb2;                                                         \
        char *dst_ = smem + (SB) * G8_BUF + (T) * G8_SLOT + wave * 1024;                                                                   \
        if (((T) == 0 || (T) == 3) && p.nt_a) {                                                                                           \
            __builtin_amdgcn_global_load_lds((gptr_t *)(src_ + off_), (lptr_t *)dst_, 16, 0, 2);                                          \
            __builtin_amdgcn_global_load_lds((gptr_t *)(src_ + half_ + off_), (lptr_t *)(dst_ + 8192), 16, 0, 2);                         \
        } else {                                                                                                                          \
            __builtin_amdgcn_global_load_lds((gptr_t *)(src_ + off_), (lptr_t *)dst_, 16, 0, 0);                                          \
            __builtin_amdgcn_global_load_lds((gptr_t *)(src_ + half_ + off_), (lptr_t *)(dst_ + 8192), 16, 0, 0);                         \
        }                                                                                                                                 \
        if ((T) == 3) {                                                                                                                   \
            sA += 128; sB += 128;                                                                                                         \
            if (++s_kt == KT) { s_kt = 0; cursor_tile(++s_i); }                                                                           \
        }                                                                                                                                 \
    } while (0)

    // ---- fragment read addressing (v_mfma_f32_16x16x32_bf16): lane reads row l15 of a 16-row fragment, k = 32 c + 8 lq .. + 7 of the K-tile,
    // i.e. the 16-B group (4 c + lq) ^ ((row >> 1) & 7)
    const int l15 = lane & 15, lq = lane >> 4;
    int a_rd[2], b_rd[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int g = ((4 * c + lq) ^ ((l15 >> 1) & 7)) << 4;
        a_rd[c] = (wr * 64 + l15) * 128 + g;                      // + 16-row tile * 2048
        b_rd[c] = (wc * 32 + l15) * 128 + g;
    }
    bf16x8 av[2][4], b0v[4], b1v[4];                              // A: [k half][16-row tile of the 64-row half]; B: [k half * 2 + 16-column tile of the 32-column half]
#define G8_READ_A(BUF, MH)                                                                                                                \
    _Pragma("unroll") for (int c = 0; c < 2; ++c)                                                                                         \
        _Pragma("unroll") for (int rt = 0; rt < 4; ++rt)                                                                                  \
            av[c][rt] = *(const bf16x8 *)(smem + (BUF) * G8_BUF + ((MH) ? 3 * G8_SLOT : 0) + rt * 2048 + a_rd[c])
#define G8_READ_B(BUF, NH, DST)                                                                                                           \
    _Pragma("unroll") for (int c = 0; c < 2; ++c)                                                                                         \
        _Pragma("unroll") for (int ct = 0; ct < 2; ++ct)                                                                                  \
            DST[2 * c + ct] = *(const bf16x8 *)(smem + (BUF) * G8_BUF + ((NH) ? 2 * G8_SLOT : G8_SLOT) + ct * 2048 + b_rd[c])
    // D^T orientation: rows of the MFMA result = columns n of C (B fragment first), its columns = rows m of C
#define G8_MFMA(MH, NH, BV)                                                                                                               \
    _Pragma("unroll") for (int c = 0; c < 2; ++c)                                                                                         \
        _Pragma("unroll") for (int rt = 0; rt < 4; ++rt)                                                                                  \
            _Pragma("unroll") for (int ct = 0; ct < 2; ++ct)                                                                              \
                acc[4 * (MH) + rt][2 * (NH) + ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BV[2 * c + ct], av[c][rt], acc[4 * (MH) + rt][2 * (NH) + ct], 0, 0, 0)
// first K-tile of an output tile: each accumulator's first MFMA takes C = 0 as an inline constant, so the accumulators are never cleared by
// separate instructions (128 v_mov per wave and tile otherwise, paid inside the VALU-bound epilogues)
#define G8_MFMA_Z(MH, NH, BV)                                                                                                             \
    _Pragma("unroll") for (int c = 0; c < 2; ++c)                                                                                         \
        _Pragma("unroll") for (int rt = 0; rt < 4; ++rt)                                                                                  \
            _Pragma("unroll") for (int ct = 0; ct < 2; ++ct)                                                                              \
                acc[4 * (MH) + rt][2 * (NH) + ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BV[2 * c + ct], av[c][rt], c == 0 ? zero4 : acc[4 * (MH) + rt][2 * (NH) + ct], 0, 0, 0)
#define G8_SYNC_COMPUTE(MFMAS) G8_SYNC_COMPUTE_W(asm volatile("s_waitcnt vmcnt(8)" ::: "memory"), MFMAS)
// first K-tile after a tile seam: the epilogue's stores sit in the (in-order) vmcnt queue between the DMA issued before the seam and the
// DMA issued now.  "All DMA except the newest four half-tiles has landed" is then vmcnt(8 + stores): the write-backs of the previous
// tile drain behind the next tile's MFMAs instead of stalling its first phase.
#define G8_SYNC_COMPUTE_SEAM(MFMAS)                                                                                                       \
    G8_SYNC_COMPUTE_W(if (after_seam) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 + EPI_STORES) : "memory");                               \
                      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"), MFMAS)
#define G8_SYNC_COMPUTE_W(WAIT, MFMAS)                                                                                                    \
    WAIT;                                                                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                                                    \
    __builtin_amdgcn_s_barrier();                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                                                    \
    __builtin_amdgcn_s_setprio(1);                                                                                                        \
    MFMAS;                                                                                                                                \
    __builtin_amdgcn_s_setprio(0);                                                                                                        \
    __builtin_amdgcn_sched_barrier(0)
#define G8_BARRIER()                                                                                                                      \
    __builtin_amdgcn_s_barrier();                                                                                                         \
    __builtin_amdgcn_sched_barrier(0)

    floatx4 acc[8][4];                                          // [16-row tile of the wave's 128 rows][16-column tile of its 64 columns]
    const floatx4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // ---- XCD stagger.  Every tile takes the same time, so without this all 256 CUs reach their epilogues together and 33 MB of output hits
    // the memory system as one burst per tile period (the store issue then stalls for microseconds).  XCD x starts x/8 of a tile period late:
    // the eight L2s write back in turn, each absorbs its own 4 MB burst, and the CUs of one XCD (who share operand panels) stay in step.
    if (p.stagger) {
        const int naps = (xcd * KT * p.stagger) >> 6;             // s_sleep 16 = 1,024 cycles; stagger / 64 = naps per K-tile and XCD index
        for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(16);
    }

    // ---- prologue: the first six half-tiles of the stream, then everybody meets once; the second half then drops one barrier behind
    G8_STAGE(0, 0); G8_STAGE(1, 0); G8_STAGE(2, 0); G8_STAGE(3, 0); G8_STAGE(0, 1); G8_STAGE(1, 1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          // A0, B0 of K-tile 0 have landed (this wave's pieces)
    G8_BARRIER();
    if (wr == 1) { G8_BARRIER(); }

    // stores one wave issues per tile (LSE mode: data-dependent count, no relaxation)
    constexpr int EPI_STORES = (EPI & G8_LSE) ? 0 : ((EPI & G8_PRE) ? 32 : 16);
    for (int ti = 0; ti < my_count; ++ti) {
        // one pair of K-tiles (buffer 0, buffer 1); FIRST: the pair that opens an output tile (zero-C MFMAs, seam form of the DMA waits)
#define G8_KPAIR(FIRST, MF)                                                                                                               \
        do {                                                                                                                              \
            G8_READ_B(0, 0, b0v); G8_READ_A(0, 0);                                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                                                            \
            G8_STAGE(2, 1);                                                                                                               \
            G8_SYNC_COMPUTE_SEAM(MF(0, 0, b0v));                                                                                          \
            G8_BARRIER();                                                                                                                 \
            G8_READ_B(0, 1, b1v);                                                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                                            \
            G8_STAGE(3, 1);                                                                                                               \
            G8_SYNC_COMPUTE_SEAM(MF(0, 1, b1v));                                                                                          \
            G8_BARRIER();                                                                                                                 \
            G8_READ_A(0, 1);                                                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                                                            \
            G8_STAGE(0, 0);                                                                                                               \
            G8_SYNC_COMPUTE_SEAM(MF(1, 1, b1v));                                                                                          \
            G8_BARRIER();                                                                                                                 \
            G8_STAGE(1, 0);                                                                                                               \
            G8_SYNC_COMPUTE_SEAM(MF(1, 0, b0v));                                                                                          \
            G8_BARRIER();                                                                                                                 \
            G8_READ_B(1, 0, b0v); G8_READ_A(1, 0);                                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                                                            \
            G8_STAGE(2, 0);                                                                                                               \
            G8_SYNC_COMPUTE(G8_MFMA(0, 0, b0v));                                                                                          \
            G8_BARRIER();                                                                                                                 \
            G8_READ_B(1, 1, b1v);                                                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                                            \
            G8_STAGE(3, 0);                                                                                                               \
            G8_SYNC_COMPUTE(G8_MFMA(0, 1, b1v));                                                                                          \
            G8_BARRIER();                                                                                                                 \
            G8_READ_A(1, 1);                                                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                                                            \
            G8_STAGE(0, 1);                                                                                                               \
            G8_SYNC_COMPUTE(G8_MFMA(1, 1, b1v));                                                                                          \
            G8_BARRIER();                                                                                                                 \
            G8_STAGE(1, 1);                                                                                                               \
            G8_SYNC_COMPUTE(G8_MFMA(1, 0, b0v));                                                                                          \
        } while (0)
        {
            const bool after_seam = EPI_STORES > 0 && ti > 0;
            G8_KPAIR(1, G8_MFMA_Z);
            if (2 < KT) { G8_BARRIER(); }
        }
        for (int kt2 = 2; kt2 < KT; kt2 += 2) {
            constexpr bool after_seam = false;
            G8_KPAIR(0, G8_MFMA);
            if (kt2 + 2 < KT) { G8_BARRIER(); }
        }
        // ---- tile seam.  The leading half is past its last MFMAs one barrier interval before the trailing half: it takes the closing barrier of
        // the last phase first, the trailing half after its epilogue, so both epilogues run in the same interval (and overlap).
        if (wr == 0) { G8_BARRIER(); }

        int tm, tn;
        tile_coords(p, first + ti * wg_per_xcd, tm, tn);
        const int m_w = tm * 256 + wr * 128, n_w = tn * 256 + wc * 64;
        // lane-derived epilogue addresses are rebuilt per tile from a fresh lane id (v_mbcnt): hoisted to kernel entry they would be
        // live across the whole main loop and push its operands into scratch
        const int elane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        const int e15 = elane & 15, eq = elane >> 4;             // accumulator order: C row e15 of a 16-row tile, columns 4 eq .. 4 eq + 3 of a 16-column tile
        char *stg = smem + G8_STAGING + wave * 4096;            // wave-private: 32 rows x 64 bf16, 8-B slots XOR-swizzled with (row & 15)
        const int prow = elane >> 3, pc16 = elane & 7;           // row-order pass: 8 lanes cover one 128-byte row segment, 8 rows per pass

        EXP_G8_AFTER_KLOOP(p, acc, wr)
        constexpr bool LSE = (EPI & G8_LSE) != 0, HAS_BIAS = (EPI & G8_BIAS) != 0, HAS_RES = (EPI & (G8_RADD | G8_RGELU | G8_RMUL)) != 0;
        constexpr bool PREG = (EPI & G8_PREG) != 0;
        if constexpr (!LSE) {
            const float keep_scale = emdr2_keep_scale(p.drop_p);
            const uint32_t thr = emdr2_drop_thr(p.drop_p);
            constexpr int npass = (EPI & G8_PRE) ? 2 : 1;         // with a pre-activation output: one pass for it, one for the activation
            // this lane's place in the row-order pass: row prow (+ 8 per pass, + 32 per slab), 16-byte column group pc16; one 64-bit multiply
            // per tile, everything else is a wave-uniform multiple of the row pitch
            const long long lane_off = ((long long)(m_w + prow) * p.ldc + n_w + pc16 * 8) * 2;
            const long long pitch8 = p.ldc * 16;                  // bytes per 8 rows
            // bias of the 32 columns this lane holds in accumulator order: loaded ONCE per tile and BEFORE the residual rows (vmcnt retires
            // in order: a bias load issued behind them would make its first use wait for the whole residual block)
            float bcol[4][4];                                     // [16-column tile][register]
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (HAS_BIAS) t = *(const float4 *)(p.bias + n_w + nt * 16 + 4 * eq);
                bcol[nt][0] = t.x; bcol[nt][1] = t.y; bcol[nt][2] = t.z; bcol[nt][3] = t.w;
            }
            // residual rows in row order, fetched as far ahead as the registers allow: a slab is staged in ~0.7 us, a residual row takes 1-2 us
            // to arrive -- two slabs ahead (r02) every slab still waited for its rows.  All four at once where 64 registers are free (the
            // main loop's operand registers are, the dropout recipe's hash temporaries leave room for three)
            constexpr int RAHEAD = (EPI & G8_DROP) ? 3 : 4;
            uint4 rr[RAHEAD][4];
            auto load_res = [&](int mi, uint4(&dst)[4]) {
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    const u32x4_t t = __builtin_nontemporal_load((const u32x4_t *)(p.R + lane_off + (mi * 4 + ps) * pitch8));
                    dst[ps] = make_uint4(t.x, t.y, t.z, t.w);
                }
            };
            if constexpr (HAS_RES) {
#pragma unroll
                for (int a = 0; a < RAHEAD; ++a) load_res(a, rr[a]);
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                float act[PREG ? 8 : 1][4];                       // PREG: the activations wait here while their derivatives go out in pass 0
#pragma unroll
                for (int pass = 0; pass < npass; ++pass) {
                    const bool final_pass = pass == npass - 1;
                    // a 32-row slab = two 16-row accumulator tiles (h) x four 16-column tiles (nt); this lane: slab row 16 h + e15
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int srow = 16 * h + e15;
                        const int m_l = m_w + mi * 32 + srow;             // the C row this lane holds in accumulator order
                        uint32_t rh = 0;
                        if constexpr ((EPI & G8_DROP) != 0) rh = emdr2_row_hash(p.seed, (unsigned long long)m_l);
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) {
                            const floatx4 &a4 = acc[2 * mi + h][nt];
                            float v[4];
                            if constexpr (PREG) {
                                if (pass == 0) {
#pragma unroll
                                    for (int r = 0; r < 4; ++r) act[4 * h + nt][r] = gelu_erf_with_grad(fmaf(a4[r], p.alpha, bcol[nt][r]), v[r]);
                                } else {
#pragma unroll
                                    for (int r = 0; r < 4; ++r) v[r] = act[4 * h + nt][r];
                                }
                            } else {
#pragma unroll
                                for (int r = 0; r < 4; ++r) v[r] = fmaf(a4[r], p.alpha, bcol[nt][r]);
                            }
                            if (final_pass && !PREG) {
                                if constexpr ((EPI & G8_GELU) != 0) {
#pragma unroll
                                    for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
                                }
                                if constexpr ((EPI & G8_DROP) != 0) {
                                    // (col >> 1) * MUL of this lane's first pair of the tile; the second pair is the next one
                                    const uint32_t prod0 = (uint32_t)((n_w + nt * 16 + 4 * eq) >> 1) * EMDR2_PAIR_MUL;
#pragma unroll
                                    for (int r = 0; r < 4; r += 2) {
                                        const uint32_t bits = emdr2_pair_bits_prod(rh, prod0 + (uint32_t)(r >> 1) * EMDR2_PAIR_MUL);
                                        // a multiplicative mask (not a select on v): keeps the compiler from predicating the loads v depends on
                                        v[r] *= (bits & 0xffffu) >= thr ? keep_scale : 0.f;
                                        v[r + 1] *= (bits >> 16) >= thr ? keep_scale : 0.f;
                                    }
                                }
                            }
                            // accumulator order -> LDS: 4 consecutive columns = one 8-byte slot
                            const int c8 = nt * 4 + eq;
                            *(uint2 *)(stg + srow * 128 + ((c8 ^ (srow & 15)) << 3)) = make_uint2(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]));
                        }
                    }
                    asm volatile("" ::: "memory");                       // compiler fence only: the LDS executes one wave's writes and reads in order
                    // LDS -> row order -> global
                    char *outp = (final_pass ? p.C : p.C2) + lane_off;
#pragma unroll
                    for (int ps = 0; ps < 4; ++ps) {
                        const int row = ps * 8 + prow;
                        const uint2 lo = *(const uint2 *)(stg + row * 128 + (((2 * pc16) ^ (row & 15)) << 3));
                        const uint2 hi2 = *(const uint2 *)(stg + row * 128 + (((2 * pc16 + 1) ^ (row & 15)) << 3));
                        uint32_t w[4] = {lo.x, lo.y, hi2.x, hi2.y};
                        if (HAS_RES && final_pass) {
                            const uint32_t rw[4] = {rr[mi % RAHEAD][ps].x, rr[mi % RAHEAD][ps].y, rr[mi % RAHEAD][ps].z, rr[mi % RAHEAD][ps].w};
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                float x0 = bf16_to_f32((uint16_t)(w[q] & 0xffff)), x1 = bf16_to_f32((uint16_t)(w[q] >> 16));
                                const float r0 = bf16_to_f32((uint16_t)(rw[q] & 0xffff)), r1 = bf16_to_f32((uint16_t)(rw[q] >> 16));
                                if constexpr ((EPI & G8_RADD) != 0) { x0 += r0; x1 += r1; }
                                else if constexpr ((EPI & G8_RMUL) != 0) { x0 *= r0; x1 *= r1; }
                                else { x0 *= gelu_erf_grad(r0); x1 *= gelu_erf_grad(r1); }
                                w[q] = pack2_bf16(x0, x1);
                            }
                        }
                        EXP_G8_STORE_IF(p, w)
                        store_stream((uint16_t *)(outp + (mi * 4 + ps) * pitch8), make_uint4(w[0], w[1], w[2], w[3]));
                    }
                    asm volatile("" ::: "memory");                       // (the next pass's writes queue behind these reads in the same in-order LDS pipe)
                }
                if constexpr (HAS_RES) {
                    if (mi + RAHEAD < 4) load_res(mi + RAHEAD, rr[mi % RAHEAD]);
                }
            }
        } else {
            // ---- LSE mode: logits = alpha * acc + bias stay in registers.  Per C row (four lanes e15 + 16 eq hold its 64 columns of this
            // wave): max, sum exp(x - max), and the gold logit if the row's label falls into these columns.  The four waves of a row block
            // (wc = 0..3) write separate partials: slot index tn * 4 + wc of a [M, tiles_n * 4] table.
            const int nslots = p.tiles_n * 4;
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {                      // 16-row accumulator tiles; four lanes (eq = 0..3) share a C row
                const int m_l = m_w + mt * 16 + e15;
                const long long lab = p.labels[m_l];
                float mx = -3.0e38f, gold = 0.f;
                bool has = false;
                float x[4][4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int n = n_w + nt * 16 + 4 * eq + r;
                        // the logits the unfused path would have stored are bf16: round the same way so both paths agree to the bit
                        const float t = bf16_to_f32(f32_to_bf16(acc[mt][nt][r] * p.alpha + (HAS_BIAS ? p.bias[n] : 0.f)));
                        x[nt][r] = t;
                        mx = fmaxf(mx, t);
                        if ((long long)n == lab) { gold = t; has = true; }
                    }
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                float sm = 0.f;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sm += __builtin_amdgcn_exp2f((x[nt][r] - mx) * 1.4426950408889634f);
                sm += __shfl_xor(sm, 16);
                sm += __shfl_xor(sm, 32);
                if (eq == 0) {
                    p.lse_max[(long long)m_l * nslots + tn * 4 + wc] = mx;
                    p.lse_sum[(long long)m_l * nslots + tn * 4 + wc] = sm;
                }
                if (has) p.lse_gold[m_l] = gold;
            }
        }
        if (wr == 1) { G8_BARRIER(); }
    }
    if (wr == 0) { G8_BARRIER(); }                              // the leading half pays back the barrier the trailing half took at the start
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // speculative half-tiles past the end of the stream
}

int g8_cu_count()
{
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        cus = prop.multiProcessorCount;
    }
    return cus;
}

int grid_for(const G8Params &p)
{
    int grid = g8_cu_count() & ~7;
    if (grid < 8) grid = 8;
    if (grid > ((p.total + 7) & ~7)) grid = (p.total + 7) & ~7;
    return grid;
}

template <int EPI>
int g8_launch(G8Params &p, hipStream_t stream)
{
    constexpr int LDS = G8_STAGING + 8 * 4096;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void *)gemm8_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -3;
        attr_done = true;
    }
    p.tiles_m = p.M / 256; p.tiles_n = p.N / 256;
    p.total = p.tiles_m * p.tiles_n;
    p.per = (p.total + 7) >> 3;
    // B panels of 256 x K bf16 that fit about half of a 4 MB L2 (the rest holds the A panels in flight and the output lines in transit)
    const long long panel = 256ll * p.K * 2;
    int ng = (int)(2560ll * 1024 / panel);
    if (ng < 1) ng = 1;
    long long single_kb = 4096;                               // a B matrix up to this size is walked as ONE group (plain n-fastest)
    EXP_G8_HOST_L2(single_kb)
    if (ng > p.tiles_n || 2 * ng < p.tiles_n || (long long)p.N * p.K * 2 <= (single_kb << 10)) ng = p.tiles_n;
    EXP_G8_HOST_GROUPS(p, ng)
    const int groups = (p.tiles_n + ng - 1) / ng;
    p.ngroup = (p.tiles_n + groups - 1) / groups;
    auto magic = [](long long d) { return (uint32_t)(((1ull << 32) + (unsigned long long)d - 1) / (unsigned long long)d); };
    const int last = p.tiles_n - (groups - 1) * p.ngroup;
    if ((long long)p.total >= (1ll << 24)) return -4;
    p.mg_full = magic((long long)p.ngroup * p.tiles_m); p.mg_group = p.ngroup > 1 ? magic(p.ngroup) : 0; p.mg_last = last > 1 ? magic(last) : 0;
    // one K-tile takes ~3,200 shader cycles at the sustained rate; 1/8 of that per XCD index, in 1,024-cycle naps: 3200 / 8 / 1024 * 64 = 25
    p.stagger = p.total >= 4 * grid_for(p) ? 25 : 0;             // only when every workgroup has several tiles to amortise the late start
    EXP_G8_HOST_LAUNCH(p)
    const int grid = grid_for(p);
    hipLaunchKernelGGL((gemm8_kernel<EPI>), dim3(grid), dim3(512), LDS, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

} // namespace

// Called by emdr2_gemm_nt_bf16 (gemm.hip) for eligible shapes; returns -4 when the shape is not covered so the caller falls through to the
// general kernel.
int emdr2_gemm8_try(const void *A, int64_t lda, const void *B, int64_t ldb, void *C, int64_t ldc, int M, int N, int K, float alpha,
                    const float *bias, int gelu, void *pre_act, const void *residual, int residual_mode, float drop_p, uint32_t seed,
                    hipStream_t stream)
{
    if ((M & 255) || (N & 255) || (K & 127) || K < 128 || M < 256 * 16) return -4;
    if ((lda & 7) || (ldb & 7) || (ldc & 7) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((uintptr_t)C & 15) || ((uintptr_t)pre_act & 15) ||
        ((uintptr_t)residual & 15) || ((uintptr_t)bias & 15))
        return -4;
    if (256ll * lda * 2 + 256 >= (1ll << 32) || 256ll * ldb * 2 + 256 >= (1ll << 32)) return -4;      // 32-bit lane offsets inside a tile
    G8Params p = {};
    p.A = (const char *)A; p.B = (const char *)B; p.C = (char *)C; p.C2 = (char *)pre_act;
    p.bias = bias; p.R = (const char *)residual; p.rmode = residual_mode;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.alpha = alpha; p.gelu = gelu; p.drop_p = drop_p; p.seed = seed;
    const int epi = (bias ? G8_BIAS : 0) | (gelu ? G8_GELU : 0) | (drop_p > 0.f ? G8_DROP : 0) |
                    (residual ? (residual_mode == 0 ? G8_RADD : residual_mode == 2 ? G8_RMUL : G8_RGELU) : 0) | (pre_act ? G8_PRE : 0) | (gelu == 2 ? G8_PREG : 0);
    switch (epi) {
    case 0: return g8_launch<0>(p, stream);                                            // plain (input gradients)
    case G8_BIAS: return g8_launch<G8_BIAS>(p, stream);                                 // QKV / Q / KV projections
    case G8_BIAS | G8_GELU: return g8_launch<G8_BIAS | G8_GELU>(p, stream);             // FFN h -> 4h, no backward to follow
    case G8_BIAS | G8_GELU | G8_PRE: return g8_launch<G8_BIAS | G8_GELU | G8_PRE>(p, stream);
    case G8_RADD: return g8_launch<G8_RADD>(p, stream);                                 // data gradient added onto a running sum (C may be R: kernels.FanInFn)
    case G8_BIAS | G8_RADD: return g8_launch<G8_BIAS | G8_RADD>(p, stream);             // attention output / FFN 4h -> h, evaluation
    case G8_BIAS | G8_DROP | G8_RADD: return g8_launch<G8_BIAS | G8_DROP | G8_RADD>(p, stream);   // the same in training: bias-dropout-add
    case G8_RGELU: return g8_launch<G8_RGELU>(p, stream);                               // d(pre-activation) = (dy W2) * gelu'(pre)
    case G8_BIAS | G8_GELU | G8_PRE | G8_PREG: return g8_launch<G8_BIAS | G8_GELU | G8_PRE | G8_PREG>(p, stream);   // FFN h -> 4h, derivative kept
    case G8_RMUL: return g8_launch<G8_RMUL>(p, stream);                                 // d(pre-activation) = (dy W2) * saved gelu'
    default: return -4;
    }
}

extern "C" int emdr2_gemm_nt_lse_bf16(const void *A, int64_t lda, const void *B, int64_t ldb, int M, int N, int K, float alpha, const float *bias,
                                      const int64_t *labels, float *part_max, float *part_sum, float *gold, void *stream)
{
    if (!A || !B || !labels || !part_max || !part_sum || !gold) return -1;
    if ((M & 255) || (N & 255) || (K & 127) || K < 128) return -4;
    if ((lda & 7) || (ldb & 7) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return -1;
    if (256ll * lda * 2 + 256 >= (1ll << 32) || 256ll * ldb * 2 + 256 >= (1ll << 32)) return -4;
    G8Params p = {};
    p.A = (const char *)A; p.B = (const char *)B; p.bias = bias;
    p.lda = lda; p.ldb = ldb; p.ldc = N; p.M = M; p.N = N; p.K = K; p.alpha = alpha;
    p.lse_max = part_max; p.lse_sum = part_sum; p.lse_gold = gold; p.labels = (const long long *)labels;
    OpsTimer timer(OPS_GEMM_NT, 2.0 * M * (double)N * K, (hipStream_t)stream);
    return bias ? g8_launch<G8_LSE | G8_BIAS>(p, (hipStream_t)stream) : g8_launch<G8_LSE>(p, (hipStream_t)stream);
}
