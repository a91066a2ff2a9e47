// emdr2_amd/csrc/mips_scan8.hip -- the index scan for 129..512 queries per pass in the pipeline of the persistent GEMM (gemm8.hip): fused fp16 MFMA
// GEMM  S = E Q^T  + threshold filter, S never materialised.
//
// Replaces, like mips_scan.hip, the reference's dense  C = Q * E^T  + torch.topk (megatron/data/emdr2_index.py:281-295) for the filter
// segments (mode 0) of a search; same inputs (stripe-tiled index image, chunk-tiled query image, per-query thresholds) and the same
// survivor protocol (LDS queue -> per-query candidate buffers), so the select / finalize kernels downstream are unchanged and the results are
// bit-identical by construction (every (row, query) score is the same fp32 MFMA sum over k in the same order: chunks ascending, the four
// 16-wide k-steps of a K-tile ascending).
//
// What changes against mips_scan.hip (128 rows x 512 queries per workgroup, 32-wide chunks, 3-stage lockstep ring: 915 TFLOP/s, the LDS-DMA
// fill and the MFMA phase nearly additive):
//  * work item = 256 index rows x 256 queries (one half of the 512-query image): 64 KiB of L2->LDS fill per 8.4 MFLOP instead of 40 KiB per
//    4.2 MFLOP.  The two halves of a row tile are adjacent items, taken by two CUs of the same XCD in the same round: the index rows leave
//    HBM once, the partner reads them from L2.
//  * K-tile = 64 (two 8 KiB chunk images per stripe), 2 x 4 half-tile slots, four phases per K-tile with 8 MFMAs behind 12 / 4 / 8 / 0 fragment
//    reads and one half-tile of LDS-DMA six half-tiles ahead (vmcnt(8)), the two wave halves one barrier apart, persistent DMA stream across
//    items: the schedule of gemm8.hip, which the same measurements put at 1,300 TFLOP/s without an output epilogue.
//  * the epilogue is the filter: per accumulator block a max + one ballot in the common case (no survivor).
//  * the two workgroups that share a row tile pace themselves against each other through a progress counter (S8_COUPLE below): without it
//    they drift apart by more than the L2 holds and the index rows are fetched from the fabric 1.77 times; with it 1.015 times.
// LDS: 128 KiB operand ring + 32 KiB survivor queue.
#include "mips_device.h"
#include "mips_kernels.h"
#include "exp_hooks.h"

namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));

#define S8_BUF 65536
#define S8_SLOT 16384
#define S8_QCAP 2040              // survivor queue entries (16 B each); the counters sit behind them
#define S8_WCAP 255               // ... in eight wave-private regions: a wave reserves slots by adding to its OWN count, no LDS atomic, no round trip
#define S8_FLUSH_AT 128           // flush when some wave's region is half full

struct Scan8Params {
    ScanParams s;
    int t_begin, t_end;           // 256-row tiles
    int halves;                   // 256-query halves of the query image (1 or 2)
    int bn;                       // rows of the query image (256 or 512)
    int last_stripe;              // highest 128-row stripe that exists
    int total, per;               // items, items per XCD
    unsigned *prog;               // progress counters, one per pair of workgroups that share row tiles, 64 uints apart (zeroed by the caller), or nullptr
};

// A survivor's slot in query q's candidate set.  r04: one sub-list per XCD (ScanParams.cand8 / count8) and an atomic of WORKGROUP scope: it is
// performed by this XCD's L2 on a line no other XCD touches during the launch, instead of going out to the memory side like the agent-scope
// atomic on ONE counter per query did (eight L2s are not coherent among themselves: ~0.2 us each, 380,000 of them in the segment right after
// the dense one = 80 of its 215 us).  The select that follows a segment reads the main list and the eight sub-lists (mips_aux.hip).
__device__ __forceinline__ void s8_append(const ScanParams &p, unsigned xcc, unsigned q, unsigned score_bits, unsigned row)
{
    const unsigned slot = __hip_atomic_fetch_add(&p.count8[xcc * 512 + q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (slot < SUBCAP) { p.cand8[((size_t)q * 8 + xcc) * SUBCAP + slot] = make_uint2(score_bits, row); return; }
    // r05: a full sub-list SPILLS into the query's main list (CAPQ entries, shared by all XCDs: agent-scope atomic) instead of dropping the
    // survivor.  An XCD owns a CONTIGUOUS range of the row sequence, so in an index whose neighbouring rows are similar (consecutive passages
    // of one article) a query's survivors of a segment pile up in ONE sub-list; before, 1,024 of them sent the query to the all-exact
    // path (a host sync + an integer pass over every row) although the other seven sub-lists and the 16,384-entry main list stood empty.
    // The count keeps growing past SUBCAP (the select clamps it); only a full MAIN list loses candidates and flags the query.
    const unsigned s2 = __hip_atomic_fetch_add(&p.count[q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (s2 < p.capq) p.cand[(size_t)q * p.capq + s2] = make_uint2(score_bits, row);
}

// queue -> candidate sub-lists; cnt[w] = entries in wave w's region (all 512 threads take part: thread t drains region t >> 6)
__device__ __forceinline__ void s8_flush(const ScanParams &p, const char *qbuf, const unsigned *cnt, int tid, unsigned xcc)
{
    const int region = tid >> 6;
    unsigned m = ((const volatile __attribute__((address_space(3))) unsigned *)cnt)[region];
    if (m > S8_WCAP) m = S8_WCAP;
    for (unsigned i = tid & 63; i < m; i += 64) {
        const uint4 e = ((const uint4 *)qbuf)[region * S8_WCAP + i];
        s8_append(p, xcc, e.z, e.x, e.y);
    }
}

__global__ void __launch_bounds__(512) mips_scan8_kernel(Scan8Params P)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ScanParams &p = P.s;
    char *const qbuf = smem + 2 * S8_BUF;
    unsigned *const qcnt = (unsigned *)(qbuf + S8_QCAP * 16);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;                 // wave grid 2 (rows) x 4 (queries); wr is also the half that runs one barrier behind
    const int l31 = lane & 31, hi = lane >> 5;

    // ---- this workgroup's items: XCD x = id & 7 owns sequence positions [x * per, (x + 1) * per), its workgroups take them round-robin; an item
    // is (row tile, query half) with the half fastest, so both halves of a tile run at the same time on one L2
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, wg_per_xcd = gridDim.x >> 3;
    const int seq_lo = xcd * P.per;
    int seq_hi = seq_lo + P.per; if (seq_hi > P.total) seq_hi = P.total;
    const int first = seq_lo + slot;
    if (first >= seq_hi) return;
    const int my_count = (seq_hi - first + wg_per_xcd - 1) / wg_per_xcd;
    const int KT = p.nch >> 1;                                // K-tiles of 64 = pairs of 32-wide chunks; even (host)
    unsigned *const flagw = qcnt + 8;                          // LDS landing word of the partner-progress DMA (behind the eight region counts)
    if (tid < 8) qcnt[tid] = 0;
    if (tid == 0) *flagw = 0;
    unsigned wq = 0;                                           // entries in this wave's queue region (wave-uniform)
    // the XCD this workgroup really runs on (HW_REG_XCC_ID: id 20, bits 0..3), not the one its block id suggests: the sub-list protocol is
    // only correct if all appenders of a sub-list share an L2
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u;
    // the workgroups of an XCD stride the sequence by an even count (host) and `per` is even: a workgroup keeps ONE query half for all its items,
    // so its thresholds are loaded once (a load in the filter would wait out the whole DMA queue: vmcnt is in order)
    const int hq = P.halves == 2 ? first & 1 : 0;
    float tauv[4];                                             // the wave's four 16-query tiles: this lane's query of tile qt is wc * 64 + qt * 16 + l15
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
        const int q = hq * 256 + wc * 64 + qt * 16 + ((lane & 3) * 4 + ((lane & 15) >> 2));      // (tile rows are permuted, see the fragment reads)
        tauv[qt] = q < p.n_q ? p.tau[q] : __builtin_inff();
        EXP_SCAN8_TAU(p, tauv, qt)
    }

    // ---- LDS-DMA addressing.  The operand images in HBM are LDS images already (mips_device.h: 64-byte rows, 16-byte groups XOR-swizzled with
    // (row >> 2) & 3), so every piece is a linear 1 KiB copy.  Half-tile slots:
    //   A_h (rows [64 h, 64 h + 64) of both stripes):  [stripe 2][chunk 2][64 rows x 64 B]     piece pa = stripe * 8 + chunk * 4 + quarter
    //   B_h (queries [32 h, 32 h + 32) of every wave):  [wave column 4][chunk 2][32 rows x 64 B] piece pb = wc * 4 + chunk * 2 + half
    // wave w moves pieces w and w + 8 of every half-tile.
    const int ja = (wave >> 2) & 1, qa = wave & 3;            // A pieces w, w + 8: stripes 0 / 1, chunk ja, quarter qa
    const int jb = (wave >> 1) & 1, qb = wave & 1;            // B pieces w, w + 8: wave columns w >> 2 and 2 + (w >> 2), chunk jb, half qb
    const uint32_t offA = (uint32_t)(ja * STRIPE_CHUNK_BYTES + qa * 1024 + lane * 16);
    const uint32_t q_stage = (uint32_t)P.bn * 64;             // bytes of one 32-wide chunk of the query image
    const uint32_t offB = (uint32_t)(jb * q_stage + ((wave >> 2) * 64) * 64 + qb * 1024 + lane * 16);
    // stage cursor (wave-uniform): item `s_i` of this workgroup's list, K-tile `s_kt`
    int s_i = 0, s_kt = 0;
    const char *sA0, *sA1, *sB;
    auto cursor_item = [&](int i) {
        if (i >= my_count) i = my_count - 1;                  // past the end: harmless re-reads keep the vmcnt arithmetic fixed
        const int pos = first + i * wg_per_xcd;
        const int tile = P.t_begin + (P.halves == 2 ? pos >> 1 : pos);
        int st0 = tile * 2, st1 = tile * 2 + 1;
        if (st0 > P.last_stripe) st0 = P.last_stripe;         // rows past the end of the shard: re-read the last stripe (masked by row < n_rows)
        if (st1 > P.last_stripe) st1 = P.last_stripe;
        sA0 = p.e_tiled + (size_t)st0 * p.nch * STRIPE_CHUNK_BYTES;
        sA1 = p.e_tiled + (size_t)st1 * p.nch * STRIPE_CHUNK_BYTES;
        sB = p.q_tiled + (size_t)hq * 256 * 64;
    };
    cursor_item(0);
#define S8_STAGE(T, SB)                                                                                                                   \
    do {                                                                                                                                  \
        char *dst_ = smem + (SB) * S8_BUF + (T) * S8_SLOT + wave * 1024;                                                                   \
        if ((T) == 0 || (T) == 3) {                                                                                                       \
            const uint32_t o_ = offA + ((T) == 3 ? 4096 : 0);                                                                             \
            __builtin_amdgcn_global_load_lds((gptr_t *)(sA0 + o_), (lptr_t *)dst_, 16, 0, 0);                                             \
            __builtin_amdgcn_global_load_lds((gptr_t *)(sA1 + o_), (lptr_t *)(dst_ + 8192), 16, 0, 0);                                    \
        } else {                                                                                                                          \
            const uint32_t o_ = offB + ((T) == 2 ? 32 * 64 : 0);                                                                          \
            __builtin_amdgcn_global_load_lds((gptr_t *)(sB + o_), (lptr_t *)dst_, 16, 0, 0);                                              \
            __builtin_amdgcn_global_load_lds((gptr_t *)(sB + o_ + 128 * 64), (lptr_t *)(dst_ + 8192), 16, 0, 0);                          \
        }                                                                                                                                 \
        if ((T) == 3) {                                                                                                                   \
            sA0 += 2 * STRIPE_CHUNK_BYTES; sA1 += 2 * STRIPE_CHUNK_BYTES; sB += 2 * q_stage;                                              \
            if (++s_kt == KT) { s_kt = 0; cursor_item(++s_i); }                                                                           \
        }                                                                                                                                 \
    } while (0)

    // ---- fragment reads for v_mfma_f32_16x16x32_f16: a lane holds row (query) l15 of a 16-row tile and k = 8 lq .. 8 lq + 7 of the 32-wide
    // chunk, i.e. the 16-byte group lq ^ ((row >> 2) & 3) of that row's 64 bytes.  r04: the 16 x 16 x 32 shape instead of 32 x 32 x 16 -- same
    // fragment bytes and registers per flop, half the accumulator registers read and written per flop: with N(0,1) operands an MFMA-only loop
    // sustains 2,120 instead of 1,780 TFLOP/s at the board's power cap (tools/mfma_peak.hip), and the cap is what binds this kernel.
    // Operand lane l15 takes tile row prow = 4 (l15 & 3) + (l15 >> 2), not row l15: with consecutive rows on consecutive lanes every
    // ds_read_b128 of this pattern has a 2-way bank conflict on the 64-byte-row image (SQ_LDS_BANK_CONFLICT = half of SQ_LDS_IDX_ACTIVE;
    // tools/lds_conflict_probe.hip), with the rows of a tile dealt four apart none.  Output row m = 4 eq + r of a tile is then index row
    // 4 r + eq, output column e15 query 4 (e15 & 3) + (e15 >> 2): the filter below and the thresholds above follow.
    const int l15 = lane & 15, lq = lane >> 4;
    const int prow = (l15 & 3) * 4 + (l15 >> 2);
    const int frag_rd = prow * 64 + ((lq ^ ((prow >> 2) & 3)) << 4);
    const int a_rd = wr * 8192 + frag_rd;                       // + chunk * 4096 + row tile * 1024
    const int b_rd = wc * 4096 + frag_rd;                       // + chunk * 2048 + query tile * 1024
    half8 av[2][4], b0v[4], b1v[4];                              // A: [chunk][16-row tile of the 64-row half]; B: [chunk * 2 + 16-query tile of the 32-query half]
#define S8_READ_A(BUF, MH)                                                                                                                \
    _Pragma("unroll") for (int c = 0; c < 2; ++c)                                                                                         \
        _Pragma("unroll") for (int rt = 0; rt < 4; ++rt)                                                                                  \
            av[c][rt] = *(const half8 *)(smem + (BUF) * S8_BUF + ((MH) ? 3 * S8_SLOT : 0) + c * 4096 + rt * 1024 + a_rd)
#define S8_READ_B(BUF, NH, DST)                                                                                                           \
    _Pragma("unroll") for (int c = 0; c < 2; ++c)                                                                                         \
        _Pragma("unroll") for (int ct = 0; ct < 2; ++ct)                                                                                  \
            DST[2 * c + ct] = *(const half8 *)(smem + (BUF) * S8_BUF + ((NH) ? 2 * S8_SLOT : S8_SLOT) + c * 2048 + ct * 1024 + b_rd)
    // rows of the MFMA result = index rows (A fragment first), columns = queries: a lane holds ONE query and 4 rows per accumulator tile
#define S8_MFMA(MH, NH, BV)                                                                                                               \
    _Pragma("unroll") for (int c = 0; c < 2; ++c)                                                                                         \
        _Pragma("unroll") for (int rt = 0; rt < 4; ++rt)                                                                                  \
            _Pragma("unroll") for (int ct = 0; ct < 2; ++ct)                                                                              \
                acc[4 * (MH) + rt][2 * (NH) + ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[c][rt], BV[2 * c + ct], acc[4 * (MH) + rt][2 * (NH) + ct], 0, 0, 0)
// first K-tile of a row tile: each accumulator's first MFMA takes C = 0 as an inline constant -- the accumulators are never cleared by
// separate instructions (128 v_mov per wave and item otherwise, inside the filter's VALU time)
#define S8_MFMA_Z(MH, NH, BV)                                                                                                             \
    _Pragma("unroll") for (int c = 0; c < 2; ++c)                                                                                         \
        _Pragma("unroll") for (int rt = 0; rt < 4; ++rt)                                                                                  \
            _Pragma("unroll") for (int ct = 0; ct < 2; ++ct)                                                                              \
                acc[4 * (MH) + rt][2 * (NH) + ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[c][rt], BV[2 * c + ct], c == 0 ? zero4 : acc[4 * (MH) + rt][2 * (NH) + ct], 0, 0, 0)
#define S8_SYNC_COMPUTE(BETWEEN, MFMAS)                                                                                                   \
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                                                    \
    __builtin_amdgcn_s_barrier();                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                                                    \
    BETWEEN;                                                                                                                              \
    __builtin_amdgcn_s_setprio(1);                                                                                                        \
    MFMAS;                                                                                                                                \
    __builtin_amdgcn_s_setprio(0);                                                                                                        \
    __builtin_amdgcn_sched_barrier(0)
#define S8_BARRIER()                                                                                                                      \
    __builtin_amdgcn_s_barrier();                                                                                                         \
    __builtin_amdgcn_sched_barrier(0)
    // Queue high-water check.  Both wave halves run it in the SAME barrier interval -- the first one after all pushes of the finished item
    // (leading half: right behind the first barrier of the next item; trailing half: right behind its seam barrier) -- so the decision is
    // uniform, and the two barriers inside pair up half against half.
    auto maybe_flush = [&]() {
        unsigned n_ = 0;
#pragma unroll
        for (int w = 0; w < 8; ++w) n_ = max(n_, ((const volatile __attribute__((address_space(3))) unsigned *)qcnt)[w]);
        if (n_ >= S8_FLUSH_AT) {
            s8_flush(p, qbuf, qcnt, tid, xcc);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            S8_BARRIER();
            if (tid < 8) qcnt[tid] = 0;
            wq = 0;
            S8_BARRIER();
        }
    };
#define S8_MAYBE_FLUSH() maybe_flush()

    // ---- partner coupling.  The two workgroups that take the two query halves of the same row tiles (slots 2j, 2j + 1 of one XCD) should read
    // the index rows within the L2's residency of each other (~10 us of streaming); nothing else couples them (a miss does not slow the
    // leader down), and uncoupled they drift apart until every row tile is fetched from the fabric twice.  Once per two K-tiles wave 0 adds 1
    // to the pair's counter (no-return atomic) and has the counter DMA'd into an LDS word (no VGPR result, nothing to wait for); the word
    // read one round later gives the partner's progress as of ~4 us ago, and a workgroup that leads by two rounds or more naps in
    // proportion.  The follower never waits, so there is no way to deadlock; a finished workgroup adds 2^20.
    unsigned *const prog = (P.prog && P.halves == 2) ? P.prog + (xcd * (wg_per_xcd >> 1) + (slot >> 1)) * 64 : nullptr;      // 256 B apart: one L2 line each
    int ticks = 0;
#define S8_COUPLE()                                                                                                                       \
    if (prog && wave == 0) {                                                                                                              \
        const int tot_ = __builtin_amdgcn_readfirstlane((int)*(volatile __attribute__((address_space(3))) unsigned *)flagw);               \
        if (lane == 0) {                                                                                                                  \
            atomicAdd(prog, 1u);                                                                                                          \
            __builtin_amdgcn_global_load_lds((gptr_t *)prog, (lptr_t *)flagw, 4, 0, 16);      /* sc1 = agent scope: never from the CU's own L1 */   \
        }                                                                                                                                 \
        const int lead_ = 2 * ticks - tot_;                   /* my rounds minus the partner's, both as of the previous round */           \
        ++ticks;                                                                                                                          \
        if (!(P.s.tune & 64)) for (int i_ = 0; i_ < (lead_ > 6 ? 6 : lead_) - 1; ++i_) __builtin_amdgcn_s_sleep(16);     /* 1,024 cycles each */ \
    }

    floatx4 acc[8][4];                                         // [16-row tile of the wave's 128 rows][16-query tile of its 64 queries]
    const floatx4 zero4 = {0.f, 0.f, 0.f, 0.f};

    EXP_SCAN8_LATE_START(P, hq)
    // ---- prologue: the first six half-tiles of the stream, then everybody meets once; the second half then drops one barrier behind
    S8_STAGE(0, 0); S8_STAGE(1, 0); S8_STAGE(2, 0); S8_STAGE(3, 0); S8_STAGE(0, 1); S8_STAGE(1, 1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          // A0, B0 of K-tile 0 have landed (this wave's pieces)
    S8_BARRIER();                                             // (also publishes the queue counter reset)
    if (wr == 1) { S8_BARRIER(); }

    for (int ti = 0; ti < my_count; ++ti) {
        // one pair of K-tiles (buffer 0, buffer 1); HOOK runs behind the first barrier, MF is the MFMA form of the first K-tile
#define S8_KPAIR(HOOK, MF)                                                                                                                \
        do {                                                                                                                              \
            S8_READ_B(0, 0, b0v); S8_READ_A(0, 0);                                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                                                            \
            S8_STAGE(2, 1);                                                                                                               \
            S8_SYNC_COMPUTE(HOOK, MF(0, 0, b0v));                                                                                         \
            S8_BARRIER();                                                                                                                 \
            S8_READ_B(0, 1, b1v);                                                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                                            \
            S8_STAGE(3, 1);                                                                                                               \
            S8_SYNC_COMPUTE(, MF(0, 1, b1v));                                                                                             \
            S8_BARRIER();                                                                                                                 \
            S8_READ_A(0, 1);                                                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                                                            \
            S8_STAGE(0, 0);                                                                                                               \
            S8_SYNC_COMPUTE(, MF(1, 1, b1v));                                                                                             \
            S8_BARRIER();                                                                                                                 \
            S8_STAGE(1, 0);                                                                                                               \
            S8_SYNC_COMPUTE(, MF(1, 0, b0v));                                                                                             \
            S8_BARRIER();                                                                                                                 \
            S8_READ_B(1, 0, b0v); S8_READ_A(1, 0);                                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                                                            \
            S8_STAGE(2, 0);                                                                                                               \
            S8_SYNC_COMPUTE(, S8_MFMA(0, 0, b0v));                                                                                        \
            S8_BARRIER();                                                                                                                 \
            S8_READ_B(1, 1, b1v);                                                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                                            \
            S8_STAGE(3, 0);                                                                                                               \
            S8_SYNC_COMPUTE(, S8_MFMA(0, 1, b1v));                                                                                        \
            S8_BARRIER();                                                                                                                 \
            S8_READ_A(1, 1);                                                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                                                            \
            S8_STAGE(0, 1);                                                                                                               \
            S8_SYNC_COMPUTE(, S8_MFMA(1, 1, b1v));                                                                                        \
            S8_BARRIER();                                                                                                                 \
            S8_STAGE(1, 1);                                                                                                               \
            S8_SYNC_COMPUTE(, S8_MFMA(1, 0, b0v));                                                                                        \
            S8_COUPLE();                                                                                                                  \
        } while (0)
        S8_KPAIR(if (wr == 0 && ti > 0) S8_MAYBE_FLUSH(), S8_MFMA_Z);
        if (2 < KT) { S8_BARRIER(); }
        for (int kt2 = 2; kt2 < KT; kt2 += 2) {
            S8_KPAIR(, S8_MFMA);
            if (kt2 + 2 < KT) { S8_BARRIER(); }
        }
        // ---- item seam.  The leading half is past its last MFMAs one barrier interval before the trailing half: it takes the closing barrier of
        // the last phase first, the trailing half after its filter, so both filters run in the same interval.
        if (wr == 0) { S8_BARRIER(); }

        const int pos = first + ti * wg_per_xcd;
        const int tile = P.t_begin + (P.halves == 2 ? pos >> 1 : pos);
        // lane ids rebuilt per item (v_mbcnt): hoisted to kernel entry they would be live across the whole main loop
        const int elane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        const int e15 = elane & 15, eq = elane >> 4;
        const int row_w = tile * 256 + wr * 128 + eq;         // + 16 rt + 4 r
        const bool tail = (tile + 1) * 256 > p.n_rows;          // only the shard's last tile has rows that do not exist
        bool stored = false;
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            const unsigned q = (unsigned)(hq * 256 + wc * 64 + qt * 16 + (e15 & 3) * 4 + (e15 >> 2));
            const float tau = tauv[qt];
            // one max + one ballot per 128 x 16 accumulator column (31 v_max): the common case has no survivor.  A column with one looks into the
            // 16 x 16 tiles that hold one (their maxes are the partial results of the column's) and there into the four registers.
            float mr[8];
#pragma unroll
            for (int rt = 0; rt < 8; ++rt) mr[rt] = fmaxf(fmaxf(acc[rt][qt][0], acc[rt][qt][1]), fmaxf(acc[rt][qt][2], acc[rt][qt][3]));
            const float m = fmaxf(fmaxf(fmaxf(mr[0], mr[1]), fmaxf(mr[2], mr[3])), fmaxf(fmaxf(mr[4], mr[5]), fmaxf(mr[6], mr[7])));
            if (__builtin_amdgcn_ballot_w64(m >= tau) == 0) continue;              // the common case
            // Slots come out of this wave's OWN queue region: the reservation is a scalar add (r03: a ballot, an LDS atomic by lane 0 and a
            // readfirstlane round trip per register that held a survivor); registers without a survivor cost a compare and a scalar branch.
#pragma unroll
            for (int rt = 0; rt < 8; ++rt) {
                if (__builtin_amdgcn_ballot_w64(mr[rt] >= tau) == 0) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[rt][qt][r];
                    const int row = row_w + rt * 16 + 4 * r;
                    const unsigned long long mask = __builtin_amdgcn_ballot_w64((v >= tau) && (!tail || row < p.n_rows));
                    if (mask == 0) continue;
                    if ((mask >> elane) & 1ull) {
                        const unsigned mine = wq + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                        if (mine < S8_WCAP) {                                  // (two writes: no aligned register quad to assemble)
                            ((uint2 *)qbuf)[2 * (wave * S8_WCAP + mine)] = make_uint2(__float_as_uint(v), (unsigned)row);
                            ((unsigned *)qbuf)[4 * (wave * S8_WCAP + mine) + 2] = q;
                        } else {                                               // queue region full: straight to the sub-list
                            s8_append(p, xcc, q, __float_as_uint(v), (unsigned)row);
                            stored = true;
                        }
                    }
                    wq += (unsigned)__popcll(mask);
                }
            }
        }
        if (__builtin_amdgcn_ballot_w64(stored)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (wq > S8_WCAP) wq = S8_WCAP;                        // (the overflow went straight to the candidate buffers)
        if (elane == 0) qcnt[wave] = wq;                       // published before the barrier behind which both halves look at the counts
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (wr == 1) {
            S8_BARRIER();
            if (ti + 1 < my_count) S8_MAYBE_FLUSH();
        }
    }
    if (wr == 0) { S8_BARRIER(); }                            // the leading half pays back the barrier the trailing half took at the start
    if (prog && tid == 0) atomicAdd(prog, 1u << 20);          // done: the partner stops pacing itself against this workgroup
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // speculative half-tiles past the end of the stream
    S8_BARRIER();
    s8_flush(p, qbuf, qcnt, tid, xcc);
}

} // namespace

// Filter scan (mode 0) of rows [row_begin, row_end) for a query image of `bn` = 256 or 512 rows; -4 = not covered (the caller uses mips_scan.hip)
int mips_launch_scan8(const ScanParams &p, int bn, int64_t row_begin, int64_t row_end, int cus, unsigned *prog, hipStream_t stream)
{
    if ((bn != 256 && bn != 512) || (p.nch & 3) || p.nch < 4 || (row_begin & 255) || row_end <= row_begin || !p.cand8 || !p.count8) return -4;
    Scan8Params P;
    P.s = p;
    P.prog = prog;
    P.bn = bn; P.halves = bn / 256;
    P.t_begin = (int)(row_begin >> 8);
    P.t_end = (int)((row_end + 255) >> 8);
    P.last_stripe = (int)((p.n_rows + STRIPE_ROWS - 1) / STRIPE_ROWS) - 1;
    P.total = (P.t_end - P.t_begin) * P.halves;
    int grid = cus & ~15;                                     // a multiple of 8 XCDs x an even number of workgroups each
    if (grid < 16) grid = 16;
    if (P.total < grid) return -4;                            // short segments stay on the non-persistent kernel
    P.per = ((P.total + 7) >> 3);
    P.per = (P.per + 1) & ~1;                                 // both halves of a tile on the same XCD
    constexpr int LDS = 2 * S8_BUF + S8_QCAP * 16 + 128;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void *)mips_scan8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -3;
        attr_done = true;
    }
    hipLaunchKernelGGL(mips_scan8_kernel, dim3(grid), dim3(512), LDS, stream, P);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
