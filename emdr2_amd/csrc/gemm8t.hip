// emdr2_amd/csrc/gemm8t.hip -- the weight-gradient ("TN") GEMM in the pipeline of gemm8.hip (include/emdr2_ops.h: emdr2_gemm_tn_bf16 dispatches
// here when the reduction length is a multiple of 128 and the operands are 16-byte aligned).
//
//   C[i, j] (fp32) (+)= sum_r A[r, i] * B[r, j]        A [R, I], B [R, J] row-major bf16: dW[n, k] = sum_tokens dy[token, n] * x[token, k]
//
// Reference op: the autograd weight gradient of F.linear (megatron/mpu/layers.py:255,353) = dy^T x, R = 1.6M tokens in the EMDR2 step.
//
// What carries over from gemm8.hip: the 256 x 256 output tile held by 8 waves (2 x 4, 128 x 64 outputs = 8 x 4 accumulators of the 16 x 16 x 32
// MFMA shape each), K-tile = 64 reduction rows as four 16 KiB half-tiles (A0, B0, B1, A1) re-used half-tile by half-tile, four phases per K-tile with 16 MFMAs each behind
// one half-tile of LDS-DMA and a counted vmcnt, the two wave halves one barrier apart.
// What differs:
//  * the operands arrive with the REDUCTION index slow.  A half-tile is [64 r][128 columns]: full 256-byte rows straight from the activation
//    matrices (one DMA instruction = 4 rows), and the fragments are gathered with ds_read_b64_tr_b16 (two per 8-element k-run, layout as in
//    gemm_tn.hip / tools/tr_probe.hip).  16-byte granules are XOR-swizzled with (r & 3) << 2 on the source side: the four rows a 32-lane
//    service group touches land on 4 x 64 B = distinct banks.
//  * a wave's 128 i-columns are two separate blocks of 64 (block h in half-tile A_h), its 64 j-columns two blocks of 32, so that every half-
//    tile is one contiguous 128-column span of the operand.
//  * no epilogue staging is needed (the result goes out as fp32 atomics), so all 160 KiB of LDS are operand ring: ten half-tile slots, stages
//    eight half-tiles ahead, vmcnt(12).
//  * work item = (reduction slice, output tile), one per workgroup, tile index fastest and an XCD-contiguous item range: the tiles that stream
//    the same token rows run together on one L2; slices are interleaved K-tile by K-tile.  The epilogue is 128 fp32 atomics (or plain stores,
//    one slice) per wave once per ~900 K-tiles: no persistence needed.
//  * optional bias gradient: colsum[i] += sum_r A[r, i], from the A fragments of the waves wc == 0 of the tj == 0 tiles (v_dot2_f32_bf16
//    against ones: four VALU ops per fragment).
#include "../../include/emdr2_ops.h"
#include "gemm_common.h"
#include "exp_hooks.h"
#include "ops_timing.h"
#include <stdlib.h>

typedef short short4_t __attribute__((ext_vector_type(4)));
typedef short short8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) short4_t lds_s4_t;

namespace {

struct T8Params {
    const char *A, *B;
    float *C, *colsum;
    long long lda, ldb, ldc;
    int I, J;
    int tiles_i, tiles, items;      // items = tiles * slices
    int slices, total_kt;           // reduction slices; K-tiles (64 reduction rows) in all.  Slice z takes K-tiles z, z + slices, ...
    int atomic;                     // more than one slice: accumulate with fp32 atomics into a pre-zeroed C
    int ablate;                     // -DEMDR2_EXPERIMENTS builds only (EMDR2_T8_ABLATE): 1 = the DMA stream re-reads K-tile 0, 2 = no DMA
};

#define T8_SLOT 16384                 // one half-tile: 64 reduction rows x 128 columns x 2 B
#define T8_RING (10 * T8_SLOT)        // all 160 KiB of a CU's LDS

__global__ void __launch_bounds__(512) gemm8t_kernel(T8Params p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;                 // wave grid 2 (i) x 4 (j); wr is also the half that runs one barrier behind
    const int t = lane & 15, lq = lane >> 4;                  // v_mfma_f32_16x16x32_bf16: fragment column t, k = 8 lq .. 8 lq + 7 of a 32-row k-step

    // ---- this workgroup's item: XCD x = id & 7 owns items [x * per_xcd, (x + 1) * per_xcd), tile index fastest inside a slice
    const int per_xcd = (p.items + 7) >> 3;
    const int item = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (item >= p.items) return;
    const int zs = item / p.tiles, tile = item - zs * p.tiles;
    const int tj = tile / p.tiles_i, ti = tile - tj * p.tiles_i;
    const int i0 = ti * 256, j0 = tj * 256;
    // the slices are INTERLEAVED K-tile by K-tile: at any moment all workgroups read one window of `slices` consecutive K-tiles (a few MB that
    // slides down the operands) instead of `slices` streams a hundred MB apart
    const int KT = (p.total_kt - zs + p.slices - 1) / p.slices;                               // >= 1 (host: slices <= total_kt)
    const long long lda2 = p.lda * 2, ldb2 = p.ldb * 2;

    // ---- LDS-DMA addressing.  A half-tile is 64 LDS rows of 256 B; a DMA instruction writes 4 rows (lane -> row lane >> 4, 16-B slot lane & 15);
    // wave w fills rows [4w, 4w + 4) and [32 + 4w, 32 + 4w + 4).  Slot s of row r holds source granule s ^ ((r & 3) << 2) ^ (((r >> 3) & 1) << 1).
    const int dr = lane >> 4;
    // (r04: rows 8 apart -- the two 16-lane groups of a 32-lane service group of the 16 x 16 x 32 fragment reads -- additionally swap the two
    // 32-byte halves of a 64-byte slot: row r = 4 wave + dr (+ 32) has (r >> 3) & 1 = (wave >> 1) & 1)
    const int gsrc = (lane & 15) ^ (dr << 2) ^ (((wave >> 1) & 1) << 1);
    uint32_t offA[2], offB[2];                                // half-tile h: columns [128 h, 128 h + 128) of the tile
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        int ca = i0 + 128 * h + gsrc * 8; if (ca > p.I - 8) ca = p.I - 8;       // overhang columns re-read the last granule (never stored)
        int cb = j0 + 128 * h + gsrc * 8; if (cb > p.J - 8) cb = p.J - 8;
        offA[h] = (uint32_t)((4 * wave + dr) * lda2 + ca * 2);
        offB[h] = (uint32_t)((4 * wave + dr) * ldb2 + cb * 2);
    }
    EXP_T8_STREAM(p, zs, KT, zs_addr, KT_STREAM, T8_DMA_ON)     // product: zs_addr = zs, KT_STREAM = KT, T8_DMA_ON = true
    const char *sA = p.A + (long long)zs_addr * 64 * lda2, *sB = p.B + (long long)zs_addr * 64 * ldb2;
    const long long hopA = (long long)p.slices * 64 * lda2, hopB = (long long)p.slices * 64 * ldb2;
    int s_kt = 0;
    // The LDS is a ring of TEN half-tile slots filled in stream order (A0, B0, B1, A1 of K-tile 0, of K-tile 1, ...).  The stage of a phase
    // goes EIGHT half-tiles ahead of the half-tile that phase consumes, into the slot whose last read was two phases ago; "everything but the
    // newest six half-tiles has landed" = vmcnt(12) is what the next phase's reads need: six phases of latency cover instead of the four of the
    // 8-slot ring of gemm8.hip (every operand byte is a fresh HBM line here; measured +2.5 %).  Ring positions are wave-uniform byte offsets:
    // DMA destinations go through M0, fragment reads add the offset once per fragment column.
    int st_off = 0;
#define T8_STAGE(T)                                                                                                                       \
    do {                                                                                                                                  \
        const char *src_ = ((T) == 0 || (T) == 3) ? sA : sB;                                                                              \
        const uint32_t off_ = (T) == 0 ? offA[0] : (T) == 3 ? offA[1] : (T) == 1 ? offB[0] : offB[1];                                     \
        const long long half_ = ((T) == 0 || (T) == 3) ? 32 * lda2 : 32 * ldb2;                                                           \
        char *dst_ = smem + st_off + wave * 1024;                                                                                          \
        if (T8_DMA_ON) {                                                                                                                  \
            __builtin_amdgcn_global_load_lds((gptr_t *)(src_ + off_), (lptr_t *)dst_, 16, 0, 0);                                          \
            __builtin_amdgcn_global_load_lds((gptr_t *)(src_ + half_ + off_), (lptr_t *)(dst_ + 8192), 16, 0, 0);                         \
        }                                                                                                                                 \
        st_off = st_off == T8_RING - T8_SLOT ? 0 : st_off + T8_SLOT;                                                                       \
        if ((T) == 3 && s_kt + 1 < KT_STREAM) { ++s_kt; sA += hopA; sB += hopB; }   /* past the end: harmless re-reads of the last K-tile */ \
    } while (0)

    // (First version of this layout: the four 16-lane groups read the SAME 32-byte column run of rows 8 apart, which the (r & 3) swizzle maps
    // to the same banks -- a 2-way conflict inside every 32-lane service group that cost exactly what the MFMA shape gained: equal to the
    // 32 x 32 x 16 kernel in isolation, 2.5 % slower in the step.  With the (r >> 3) & 1 term of the swizzle: +3.5-6 % per shape.)
    // ---- transposed fragment reads (r04: for the 16 x 16 x 32 MFMA shape, see gemm8.hip): a 16-lane group addresses 4 rows x 16 columns --
    // lane t: row (t >> 2), the 8-byte run of columns (t & 3) * 4 -- and lane t receives column t of those four rows; the four groups of a wave
    // take rows 8 lq .. 8 lq + 3 (and + 4 for the second read) of a 32-row k-step, all of the same 16-column fragment.
    const int rbase = 8 * lq + (t >> 2), sw = ((t >> 2) << 2) ^ ((lq & 1) << 1);
    uint32_t a_rd[4], b_rd[2];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        const int col = wr * 64 + rt * 16 + (t & 3) * 4;
        a_rd[rt] = rbase * 256 + (((col >> 3) ^ sw) << 4) + (col & 7) * 2;
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int col = wc * 32 + ct * 16 + (t & 3) * 4;
        b_rd[ct] = rbase * 256 + (((col >> 3) ^ sw) << 4) + (col & 7) * 2;
    }
    // Fragment halves as raw 8-byte pairs.  The reads are inline asm on purpose: for the ds_read_tr builtin the compiler's wait-count pass
    // assumes the read may alias every LDS-DMA in flight and puts s_waitcnt vmcnt(0) in front of it -- in every phase, which collapsed the
    // eight-half-tile DMA pipeline to one (the first versions of this kernel ran like that).  The matching lgkmcnt wait is T8_WAIT_* below.
    uint2 al[2][4], ah[2][4], b0l[4], b0h[4], b1l[4], b1h[4];  // A: [k half][16-column tile of the 64-column block]; B: [k half * 2 + 16-column tile of the 32]
    int rd_off = 0;                                           // ring offset of the current K-tile's A0; B0, B1, A1 follow (with wrap-around)
    auto ring = [](int off) { return off >= T8_RING ? off - T8_RING : off; };
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
#define T8_TR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define T8_READ_A(SLOT)                                                                                                                   \
    _Pragma("unroll") for (int rt = 0; rt < 4; ++rt) {                                                                                    \
        const uint32_t base_ = lds0 + (uint32_t)ring(rd_off + (SLOT) * T8_SLOT) + a_rd[rt];                                               \
        T8_TR(al[0][rt], base_, 0); T8_TR(ah[0][rt], base_, 1024); T8_TR(al[1][rt], base_, 8192); T8_TR(ah[1][rt], base_, 9216);          \
    }
#define T8_READ_B(SLOT, L, H)                                                                                                             \
    _Pragma("unroll") for (int ct = 0; ct < 2; ++ct) {                                                                                    \
        const uint32_t base_ = lds0 + (uint32_t)ring(rd_off + (SLOT) * T8_SLOT) + b_rd[ct];                                               \
        T8_TR(L[ct], base_, 0); T8_TR(H[ct], base_, 1024); T8_TR(L[2 + ct], base_, 8192); T8_TR(H[2 + ct], base_, 9216);                  \
    }
#define T8_WAIT_A()                                                                                                                       \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                                                   \
                 : "+v"(al[0][0]), "+v"(ah[0][0]), "+v"(al[0][1]), "+v"(ah[0][1]), "+v"(al[0][2]), "+v"(ah[0][2]), "+v"(al[0][3]), "+v"(ah[0][3]),   \
                   "+v"(al[1][0]), "+v"(ah[1][0]), "+v"(al[1][1]), "+v"(ah[1][1]), "+v"(al[1][2]), "+v"(ah[1][2]), "+v"(al[1][3]), "+v"(ah[1][3])    \
                 :: "memory")
#define T8_WAIT_B(L, H)                                                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(L[0]), "+v"(H[0]), "+v"(L[1]), "+v"(H[1]), "+v"(L[2]), "+v"(H[2]), "+v"(L[3]), "+v"(H[3])::"memory")
#define T8_FR(L, H) __builtin_bit_cast(bf16x8, make_uint4((L).x, (L).y, (H).x, (H).y))
    // rows of the MFMA result = i (A fragment first), columns = j: a lane holds one j and 4 i, 16 lanes write 64 contiguous bytes of C
#define T8_MFMA(MH, NH, BL, BH)                                                                                                           \
    _Pragma("unroll") for (int c = 0; c < 2; ++c)                                                                                         \
        _Pragma("unroll") for (int rt = 0; rt < 4; ++rt)                                                                                  \
            _Pragma("unroll") for (int ct = 0; ct < 2; ++ct)                                                                              \
                acc[4 * (MH) + rt][2 * (NH) + ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(T8_FR(al[c][rt], ah[c][rt]), T8_FR(BL[2 * c + ct], BH[2 * c + ct]), acc[4 * (MH) + rt][2 * (NH) + ct], 0, 0, 0)
#define T8_SYNC_COMPUTE(WAIT, MFMAS)                                                                                                      \
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                                                    \
    __builtin_amdgcn_s_barrier();                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                                                    \
    WAIT;                                                                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                                                    \
    __builtin_amdgcn_s_setprio(1);                                                                                                        \
    MFMAS;                                                                                                                                \
    __builtin_amdgcn_s_setprio(0);                                                                                                        \
    __builtin_amdgcn_sched_barrier(0)
#define T8_BARRIER()                                                                                                                      \
    __builtin_amdgcn_s_barrier();                                                                                                         \
    __builtin_amdgcn_sched_barrier(0)

    floatx4 acc[8][4];                                         // [16-column tile of the wave's 128 i][16-column tile of its 64 j]
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mi][ni][r] = 0.f;
    float csum[2] = {0.f, 0.f};                               // [mh]: this wave sums the 16-column tile rt == wc of each 64-column block
    // Bias gradient colsum[i] += sum_r A[r, i].  Every A fragment must be summed exactly once across the tiles_j workgroups that read the same
    // A columns and the four waves (wc) that hold the same fragments: K-tile kt belongs to the tile with tj == kt % tiles_j, the 16-column
    // tile rt of a 64-column block to the wave with wc == rt.  (The first version gave all of it to the wc == 0 waves of the tj == 0 tiles:
    // 32 v_dot2 per K-tile on one wave in eight, at ~24 cycles apiece beside MFMAs -- those workgroups ran 26 % longer and the launch waits
    // for its slowest workgroup.)
    const int tiles_j = p.tiles / p.tiles_i;
    const bool want_colsum = p.colsum != nullptr;             // wave-uniform
    int cs_phase = tj;                                        // counts down to this tile's K-tiles
    const bf16x2_t ones2 = __builtin_bit_cast(bf16x2_t, 0x3f803f80u);
#define T8_COLSUM(MH)                                                                                                                     \
    if (want_colsum && cs_phase == 0) {                                                                                                   \
        _Pragma("unroll") for (int rt = 0; rt < 4; ++rt)                                                                                  \
            if (rt == wc) {                                                                                                               \
                _Pragma("unroll") for (int c = 0; c < 2; ++c) {                                                                           \
                    const uint4 w_ = make_uint4(al[c][rt].x, al[c][rt].y, ah[c][rt].x, ah[c][rt].y);                                      \
                    float s_ = csum[MH];                                                                                                  \
                    s_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w_.x), ones2, s_, false);                           \
                    s_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w_.y), ones2, s_, false);                           \
                    s_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w_.z), ones2, s_, false);                           \
                    s_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w_.w), ones2, s_, false);                           \
                    csum[MH] = s_;                                                                                                        \
                }                                                                                                                         \
            }                                                                                                                             \
    }

    // ---- prologue: the first eight half-tiles of the stream, then everybody meets once; the second half then drops one barrier behind
    T8_STAGE(0); T8_STAGE(1); T8_STAGE(2); T8_STAGE(3); T8_STAGE(0); T8_STAGE(1); T8_STAGE(2); T8_STAGE(3);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");         // A0, B0 of K-tile 0 have landed (this wave's pieces)
    T8_BARRIER();
    if (wr == 1) { T8_BARRIER(); }

    for (int kt = 0; kt < KT; ++kt) {
        T8_READ_B(1, b0l, b0h); T8_READ_A(0);
        __builtin_amdgcn_sched_barrier(0);
        T8_STAGE(0);
        T8_SYNC_COMPUTE(T8_WAIT_B(b0l, b0h); T8_WAIT_A(), T8_MFMA(0, 0, b0l, b0h));
        T8_COLSUM(0);
        T8_BARRIER();
        T8_READ_B(2, b1l, b1h);
        __builtin_amdgcn_sched_barrier(0);
        T8_STAGE(1);
        T8_SYNC_COMPUTE(T8_WAIT_B(b1l, b1h), T8_MFMA(0, 1, b1l, b1h));
        T8_BARRIER();
        T8_READ_A(3);
        __builtin_amdgcn_sched_barrier(0);
        T8_STAGE(2);
        T8_SYNC_COMPUTE(T8_WAIT_A(), T8_MFMA(1, 1, b1l, b1h));
        T8_COLSUM(1);
        cs_phase = cs_phase == 0 ? tiles_j - 1 : cs_phase - 1;
        T8_BARRIER();
        T8_STAGE(3);
        T8_SYNC_COMPUTE(, T8_MFMA(1, 0, b0l, b0h));
        rd_off = ring(rd_off + 4 * T8_SLOT);
        if (kt + 1 < KT) { T8_BARRIER(); }
    }
    if (wr == 0) { T8_BARRIER(); }                            // the leading half pays back the barrier the trailing half took at the start
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // speculative half-tiles past the end of the stream

    // ---- epilogue.  acc[4 mh + rt][2 nh + ct]: rows i = i0 + 128 mh + 64 wr + 16 rt + 4 lq + r, column j = j0 + 128 nh + 32 wc + 16 ct + t
    if (want_colsum) {
#pragma unroll
        for (int mh = 0; mh < 2; ++mh) {
            float tot = csum[mh];                                                       // the four k-groups of the wave hold partial sums of column t
            tot += __shfl_xor(tot, 16);
            tot += __shfl_xor(tot, 32);
            const int i = i0 + 128 * mh + 64 * wr + 16 * wc + t;
            if (lq == 0 && i < p.I) unsafeAtomicAdd(&p.colsum[i], tot);
        }
    }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int j = j0 + 128 * (ni >> 1) + 32 * wc + 16 * (ni & 1) + t;
        if (j >= p.J) continue;
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            const int ib = i0 + 128 * (mi >> 2) + 64 * wr + 16 * (mi & 3) + 4 * lq;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = ib + r;
                if (i >= p.I) continue;
                float *dst = p.C + (long long)i * p.ldc + j;
                if (p.atomic) unsafeAtomicAdd(dst, acc[mi][ni][r]);
                else *dst = acc[mi][ni][r];
            }
        }
    }
}

} // namespace

// Called by emdr2_gemm_tn_bf16 (gemm_tn.hip); -4 = shape not covered, the caller falls through to the general kernel.  `split_k` is the number
// of reduction slices (capped at the number of K-tiles).
int emdr2_gemm8t_try(const void *A, int64_t lda, const void *B, int64_t ldb, float *C, int64_t ldc, int I, int J, int R, int split_k,
                     float *colsum_a, hipStream_t stream)
{
    if ((R & 63) || R < 64 || I < 8 || J < 8 || (I & 7) || (J & 7)) return -4;
    if (40ll * lda * 2 + 2ll * I >= (1ll << 32) || 40ll * ldb * 2 + 2ll * J >= (1ll << 32)) return -4;       // 32-bit lane offsets inside a K-tile
    T8Params p;
    p.A = (const char *)A; p.B = (const char *)B; p.C = C; p.colsum = colsum_a;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.I = I; p.J = J;
    p.tiles_i = (I + 255) / 256;
    p.tiles = p.tiles_i * ((J + 255) / 256);
    p.total_kt = R >> 6;
    p.slices = split_k < p.total_kt ? split_k : p.total_kt;
    p.items = p.tiles * p.slices;
    p.ablate = 0;
    EXP_T8_HOST(p)
    p.atomic = split_k > 1;                                   // the caller zeroed C exactly when it asked for more than one slice
    constexpr int LDS = T8_RING;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void *)gemm8t_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -3;
        attr_done = true;
    }
    const unsigned grid = (unsigned)(((p.items + 7) / 8) * 8);
    hipLaunchKernelGGL(gemm8t_kernel, dim3(grid), dim3(512), LDS, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
