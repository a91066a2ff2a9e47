// emdr2_amd/csrc/mips_device.h -- device helpers shared by the MIPS kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

#define STRIPE_ROWS 128
#define CHUNK_K 32
#define STRIPE_CHUNK_BYTES (STRIPE_ROWS * CHUNK_K * 2) /* 8192 */

// Stripe-tiled HBM layout: stripe t = rows [128t, 128t+128), chunk c = k in
// [32c, 32c+32).  Block (t, c) is 8 KiB, stored at ((t*nch)+c)*8192, and is exactly the LDS image
// the scan kernel wants: 512 slots of 16 B, slot = row_in_stripe*4 + (s ^ ((row_in_stripe>>2)&3))
// where s = which 8-element group of the chunk.  The XOR makes the MFMA-operand ds_read_b128
// (16 lanes = 16 rows, same s) hit 16 distinct 16-B bank slots.
__device__ __forceinline__ size_t tiled_seg_offset(int64_t row, int seg /* k/8 */, int nch)
{
    const int64_t stripe = row >> 7;
    const int ri = (int)(row & 127);
    const int c = seg >> 2, s = seg & 3;
    const int sp = s ^ ((ri >> 2) & 3);
    return ((size_t)(stripe * nch + c) * 512 + (size_t)(ri * 4 + sp)) * 16;
}

// monotone maps: larger unsigned <=> larger value
__device__ __forceinline__ uint32_t f32_order(float f)
{
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float f32_unorder(uint32_t o)
{
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}
__device__ __forceinline__ uint32_t h16_order(uint16_t h)
{
    return (h & 0x8000) ? (uint32_t)((~h) & 0xffff) : (uint32_t)(h | 0x8000);
}
__device__ __forceinline__ uint16_t h16_unorder(uint32_t o)
{
    return (o & 0x8000) ? (uint16_t)(o & 0x7fff) : (uint16_t)((~o) & 0xffff);
}

// ---- exact integer dot product of fp16 vectors ------------------------------------------------
// every finite fp16 = X * 2^-24 with X = (1024+m) << (e-1) (normal) or m (subnormal).
// We keep (signed 12-bit mantissa, shift) and accumulate products split by total shift so both
// partial sums stay inside int64 for dim <= 8192:  T = hi * 2^28 + lo  (value = T * 2^-48).
struct HalfFix {
    int mant;  // signed, |mant| < 2048
    int shift; // 0..29
};
__device__ __forceinline__ HalfFix half_fix(uint16_t h)
{
    const int e = (h >> 10) & 0x1f, m = h & 0x3ff;
    HalfFix r;
    r.mant = e ? (1024 + m) : m;
    r.shift = e ? (e - 1) : 0;
    if (h & 0x8000) r.mant = -r.mant;
    return r;
}
__device__ __forceinline__ void exact_mac(int64_t &lo, int64_t &hi, HalfFix a, HalfFix b)
{
    const int64_t p = (int64_t)(a.mant * b.mant); // |p| < 2^22
    const int sh = a.shift + b.shift;             // 0..58
    if (sh < 28) lo += p << sh;                   // < 2^50 each
    else hi += p << (sh - 28);                    // < 2^52 each
}
__device__ __forceinline__ int64_t wave_sum_i64(int64_t v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const int lo_ = __shfl_xor((int)(uint32_t)v, o);
        const int hi_ = __shfl_xor((int)(uint32_t)((uint64_t)v >> 32), o);
        v += (int64_t)(((uint64_t)(uint32_t)hi_ << 32) | (uint32_t)lo_);
    }
    return v;
}
// exact T*2^-48 (T = hi*2^28 + lo) -> nearest-even fp16 bits; overflow -> inf
__device__ __forceinline__ uint16_t fixed_to_half(int64_t lo, int64_t hi)
{
    __int128 t = ((__int128)hi << 28) + (__int128)lo;
    uint16_t sign = 0;
    unsigned __int128 mag;
    if (t < 0) { sign = 0x8000; mag = (unsigned __int128)(-t); } else mag = (unsigned __int128)t;
    if (mag == 0) return 0;
    const uint64_t mh = (uint64_t)(mag >> 64), ml = (uint64_t)mag;
    const int p = mh ? (127 - __clzll(mh)) : (63 - __clzll(ml));
    int shift = p - 10;
    if (shift < 24) shift = 24;
    unsigned __int128 q = mag >> shift;
    const unsigned __int128 rem = mag & ((((unsigned __int128)1) << shift) - 1);
    const unsigned __int128 half = ((unsigned __int128)1) << (shift - 1);
    if (rem > half || (rem == half && ((uint32_t)q & 1))) q += 1;
    uint32_t qq = (uint32_t)q;
    if (qq == 0) return sign;
    if (shift == 24 && qq <= 1024) return (uint16_t)(sign | qq);
    if (qq == 2048) { qq = 1024; shift += 1; }
    const int field = (shift + 10 - 48) + 15;
    if (field >= 31) return (uint16_t)(sign | 0x7c00);
    return (uint16_t)(sign | (field << 10) | (qq - 1024));
}

// exact T*2^-48 -> nearest-even fp32 (FaissMIPSIndex-style scores, oracle: fixed48_to_float).  |T| < 2^92 here, far from fp32 overflow;
// results below 2^-48 * 2^24 are exact (fewer than 24 significant bits).
__device__ __forceinline__ float fixed_to_float(int64_t lo, int64_t hi)
{
    __int128 t = ((__int128)hi << 28) + (__int128)lo;
    const bool neg = t < 0;
    unsigned __int128 mag = neg ? (unsigned __int128)(-t) : (unsigned __int128)t;
    if (mag == 0) return 0.0f;
    const uint64_t mh = (uint64_t)(mag >> 64), ml = (uint64_t)mag;
    const int p = mh ? (127 - __clzll(mh)) : (63 - __clzll(ml));
    int shift = p - 23;
    uint32_t q;
    if (shift <= 0) { q = (uint32_t)ml; shift = 0; }
    else {
        unsigned __int128 qq = mag >> shift;
        const unsigned __int128 rem = mag & ((((unsigned __int128)1) << shift) - 1);
        const unsigned __int128 half = ((unsigned __int128)1) << (shift - 1);
        if (rem > half || (rem == half && ((uint32_t)qq & 1))) qq += 1;
        q = (uint32_t)qq;                                  // <= 2^24: exactly representable as float
    }
    const float v = ldexpf((float)q, shift - 48);
    return neg ? -v : v;
}

// fp32 -> fp16 rounding toward +inf (used only for the conservative validity bound)
__device__ __forceinline__ uint16_t f32_to_h16_roundup(float f)
{
    _Float16 h = (_Float16)f; // RNE
    if ((float)h < f) {       // step to the next representable value above
        uint16_t b = __builtin_bit_cast(uint16_t, h);
        if (b == 0x8000) b = 0;
        if (b & 0x8000) b -= 1; else b += 1;
        return b;
    }
    return __builtin_bit_cast(uint16_t, h);
}
