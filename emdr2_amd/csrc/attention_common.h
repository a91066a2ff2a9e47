// emdr2_amd/csrc/attention_common.h -- tile layout and fragment helpers shared by the fused attention kernels (attention.hip,
// attention_bwd.hip).  Tiles are [64 rows][64 d] bf16 (128-B rows of eight 16-B granules) filled by LDS-DMA with the granule index XOR
// f(row), f(row) = (((row >> 1) & 1) << 2) | ((row >> 2) & 3): conflict-free both for row reads (ds_read_b128: 16 consecutive rows, one
// granule) and for ds_read_b64_tr_b16 transpose reads (4 rows x 4 granules per 32-lane service group; layout pinned by tools/tr_probe.hip).
#ifndef EMDR2_ATTENTION_COMMON_H
#define EMDR2_ATTENTION_COMMON_H
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

#define L2E 1.4426950408889634f
#define MASKED2 (-10000.f * 1.4426950408889634f)
#define TR_READ(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

__device__ __forceinline__ int tile_swz(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ uint32_t pack_bf16(float a, float b)
{
    const floatx2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));     // v_cvt_pk_bf16_f32 (RNE)
}
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// Fragment of the row-major global matrix row: 8 bf16 at d = 16 t + 8 hi for t = 0..3 (the B operand of the "swapped" MFMAs)
__device__ __forceinline__ void load_row_frags(const char *row, int hi, bf16x8 (&f)[4])
{
#pragma unroll
    for (int t = 0; t < 4; ++t) f[t] = *(const bf16x8 *)(row + (16 * t + 8 * hi) * 2);
}

// Row fragment (MFMA A operand, 8 consecutive d of one tile row) from a swizzled tile
__device__ __forceinline__ bf16x8 tile_row_frag(const char *tile, int row, int gran)
{
    return *(const bf16x8 *)(tile + row * 128 + ((gran ^ tile_swz(row)) << 4));
}

// The same row fragments by inline asm, for the kernels that have an LDS-DMA in flight while they read (all of them: the next block is
// staged while this one is consumed).  For a plain load from the staging buffer the compiler's wait-count pass assumes it may alias the DMA
// destination and puts s_waitcnt vmcnt(0) in front of it: the prefetch of the next block was waited for BEFORE the current block was touched
// and the double buffering hid nothing.  row_frag_addresses: the lane's byte address (row = lane & 31, stage 0) for k-step t = 0..3; the
// second 32-row half of a tile is offset 4096, a stage 16384.
__device__ __forceinline__ void row_frag_addresses(uint32_t tile_lds, int lane, uint32_t (&a)[4])
{
    const int row = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int t = 0; t < 4; ++t) a[t] = tile_lds + row * 128 + (((2 * t + hi) ^ tile_swz(row)) << 4);
}
#define LDS_READ128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define ROW_FRAGS4(d, a, so, off)                                                                                   \
    do {                                                                                                            \
        LDS_READ128((d)[0], (a)[0] + (so), off); LDS_READ128((d)[1], (a)[1] + (so), off);                           \
        LDS_READ128((d)[2], (a)[2] + (so), off); LDS_READ128((d)[3], (a)[3] + (so), off);                           \
    } while (0)
#define LDS_WAIT4(d) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"((d)[0]), "+v"((d)[1]), "+v"((d)[2]), "+v"((d)[3])::"memory")
#define LDS_WAIT8(d, e)                                                                                             \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                             \
                 : "+v"((d)[0]), "+v"((d)[1]), "+v"((d)[2]), "+v"((d)[3]), "+v"((e)[0]), "+v"((e)[1]), "+v"((e)[2]), "+v"((e)[3])::"memory")

// Transposed fragments: for the 16-row k-step starting at tile row `rb` (multiple of 16) the lane (col c = jsub*32 + lane&31, half hi) gets rows
// rb + 4 hi + {0,1,2,3} and rb + 8 + 4 hi + {0,1,2,3} of column c -- the row subset a lane's accumulator registers 8(u&1)..8(u&1)+7 cover.
// tr_base[jsub][which] holds the lane's byte address for rb = 0.
__device__ __forceinline__ void tr_addresses(uint32_t tile_lds, int lane, uint32_t (&base)[2][2])
{
    const int t = lane & 15, colgrp = (lane >> 4) & 1, hi = lane >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = j * 32 + colgrp * 16 + (t & 3) * 4;
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            const int row = 4 * hi + 8 * w + (t >> 2);
            base[j][w] = tile_lds + row * 128 + ((((col >> 3) ^ tile_swz(row))) << 4) + (col & 7) * 2;
        }
    }
}
// k-step u = 0..3 covers tile rows 16 u .. 16 u + 15: tile_swz(row + 16 u) == tile_swz(row) ^ ... only bits (row>>1)&1 and (row>>2)&3 enter, and
// 16 u changes neither, so the address moves by the plain 2048-byte offset.
#define TR_FRAG(dst, base, u)                                                              \
    do {                                                                                   \
        uint2 lo_, hi_;                                                                    \
        TR_READ(lo_, (base)[0], (u) * 2048);                                               \
        TR_READ(hi_, (base)[1], (u) * 2048);                                               \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo_), "+v"(hi_)::"memory");             \
        dst = __builtin_bit_cast(bf16x8, make_uint4(lo_.x, lo_.y, hi_.x, hi_.y));          \
    } while (0)

// Two / four transposed fragments with ONE wait: all reads go out first (they are independent), so their LDS latencies overlap
#define TR_FRAG2(d0, b0, d1, b1, u)                                                                                  \
    do {                                                                                                             \
        uint2 l0_, h0_, l1_, h1_;                                                                                    \
        TR_READ(l0_, (b0)[0], (u) * 2048); TR_READ(h0_, (b0)[1], (u) * 2048);                                        \
        TR_READ(l1_, (b1)[0], (u) * 2048); TR_READ(h1_, (b1)[1], (u) * 2048);                                        \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(l0_), "+v"(h0_), "+v"(l1_), "+v"(h1_)::"memory");                 \
        d0 = __builtin_bit_cast(bf16x8, make_uint4(l0_.x, l0_.y, h0_.x, h0_.y));                                     \
        d1 = __builtin_bit_cast(bf16x8, make_uint4(l1_.x, l1_.y, h1_.x, h1_.y));                                     \
    } while (0)
#define TR_FRAG4(d0, b0, d1, b1, d2, b2, d3, b3, u)                                                                  \
    do {                                                                                                             \
        uint2 l0_, h0_, l1_, h1_, l2_, h2_, l3_, h3_;                                                                \
        TR_READ(l0_, (b0)[0], (u) * 2048); TR_READ(h0_, (b0)[1], (u) * 2048);                                        \
        TR_READ(l1_, (b1)[0], (u) * 2048); TR_READ(h1_, (b1)[1], (u) * 2048);                                        \
        TR_READ(l2_, (b2)[0], (u) * 2048); TR_READ(h2_, (b2)[1], (u) * 2048);                                        \
        TR_READ(l3_, (b3)[0], (u) * 2048); TR_READ(h3_, (b3)[1], (u) * 2048);                                        \
        asm volatile("s_waitcnt lgkmcnt(0)"                                                                          \
                     : "+v"(l0_), "+v"(h0_), "+v"(l1_), "+v"(h1_), "+v"(l2_), "+v"(h2_), "+v"(l3_), "+v"(h3_)::"memory"); \
        d0 = __builtin_bit_cast(bf16x8, make_uint4(l0_.x, l0_.y, h0_.x, h0_.y));                                     \
        d1 = __builtin_bit_cast(bf16x8, make_uint4(l1_.x, l1_.y, h1_.x, h1_.y));                                     \
        d2 = __builtin_bit_cast(bf16x8, make_uint4(l2_.x, l2_.y, h2_.x, h2_.y));                                     \
        d3 = __builtin_bit_cast(bf16x8, make_uint4(l3_.x, l3_.y, h3_.x, h3_.y));                                     \
    } while (0)

// 1-D grid decode shared by the attention kernels: workgroup ids go round-robin to the 8 XCDs (one L2 each); the blocks of one (batch,
// head) -- which stream the same K/V (or Q/dO) rows -- are given ids 8 apart so they land on the SAME XCD and the shared operand leaves HBM
// once.  With >= 64 sequences the HEADS of a sequence stay on one XCD as well (sequence b -> XCD b % 8, its heads and blocks consecutive
// there): a token's q / k / v for all heads are one contiguous row of the packed projection output, and short (packed) sequences are
// memory-bound -- 70 flop per byte at 140 tokens -- so it matters that the 128-byte head slices of a row are fetched by one L2 close
// together in time instead of by eight L2s at eight different moments.  Returns false for the padded tail.
__device__ __forceinline__ bool attn_decode(int id, int nblk, int total_bn, int heads, int &blk, int &b, int &n)
{
    const int xcd = id & 7, rest = id >> 3;
    blk = rest % nblk;
    const int r2 = rest / nblk;
    const int batch = total_bn / heads;
    if (batch >= 64) {
        n = r2 % heads;
        b = (r2 / heads) * 8 + xcd;
        return b < batch;
    }
    const int bn = r2 * 8 + xcd;
    if (bn >= total_bn) return false;
    b = bn / heads; n = bn - b * heads;
    return true;
}
__host__ __forceinline__ unsigned attn_grid(int nblk, int total_bn, int heads)
{
    const int batch = total_bn / heads;
    if (batch >= 64) return (unsigned)(((batch + 7) / 8) * 8 * heads * nblk);
    return (unsigned)(((total_bn + 7) / 8) * 8 * nblk);
}
#endif
