/*
 * oracle/mips_oracle.c -- TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
 *
 * CPU restatement of the EMDR2 training-path MIPS search,
 *   reference: megatron/data/emdr2_index.py:268-305 (DistributedBruteForceIndex.search_mips_index)
 *              megatron/data/emdr2_index.py:241-266 (add_embed_data: row order = dict order,
 *                                                    fp16 storage, id_map[row] = doc id)
 * under the canonical numerics fixed in DESIGN.md section 3:
 *
 *   score(q, r) = RNE_fp16( sum_d q[d] * E[r][d] )     with the sum taken EXACTLY in the reals
 *   result(q)   = first k rows of { r } ordered by (score desc, r asc), mapped through ids[r]
 *
 * The reference computes fp16(torch.matmul(fp16, fp16)) (fp32-accumulate GEMM whose summation
 * order is hardware specific) followed by torch.topk (tie order unspecified on GPU).  The exact
 * sum is the order-independent idealisation of that GEMM; (score desc, row asc) is the total
 * order we pin.  Parity pin: tests/golden/mips_ref_*.npz were produced by running the
 * reference's own search_mips_index in the survey container (tests/golden/gen_mips_golden.py)
 * and this file reproduces them (tests/test_oracle_mips.py).
 *
 * Exactness: every finite fp16 is an integer multiple of 2^-24 with |int| < 2^40, so a product
 * is an integer multiple of 2^-48 below 2^80 and a 768-term (any dim < 2^40) sum fits __int128.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef __int128 i128;
typedef unsigned __int128 u128;

/* fp16 bits -> integer X with value = X * 2^-24.  Returns 0 and sets *bad on inf/nan. */
static inline int64_t half_to_fixed(uint16_t h, int *bad)
{
    int e = (h >> 10) & 0x1f;
    int64_t m = h & 0x3ff;
    int64_t x;
    if (e == 0) x = m;
    else if (e == 31) { *bad = 1; return 0; }
    else x = (1024 + m) << (e - 1);
    return (h & 0x8000) ? -x : x;
}

/* exact value T * 2^-48  ->  nearest fp16 (ties to even), overflow -> inf */
static inline uint16_t fixed48_to_half(i128 t)
{
    uint16_t sign = 0;
    u128 mag;
    if (t < 0) { sign = 0x8000; mag = (u128)(-t); } else mag = (u128)t;
    if (mag == 0) return 0;
    int p = 127;
    while (!((mag >> p) & 1)) --p;            /* msb position */
    int shift = p - 10;                       /* keep 11 significant bits */
    if (shift < 24) shift = 24;               /* subnormal quantum 2^-24 */
    u128 q = mag >> shift;
    u128 rem = mag & ((((u128)1) << shift) - 1);
    u128 half = ((u128)1) << (shift - 1);
    if (rem > half || (rem == half && (q & 1))) q += 1;
    if (q == 0) return sign;                  /* rounds to (signed) zero */
    if (shift == 24 && q <= 1024) return (uint16_t)(sign | (uint16_t)q);   /* subnormal / min normal */
    if (q == 2048) { q = 1024; shift += 1; }
    int field = (shift + 10 - 48) + 15;       /* exponent field */
    if (field >= 31) return (uint16_t)(sign | 0x7c00);
    return (uint16_t)(sign | (field << 10) | (uint16_t)(q - 1024));
}

static inline i128 exact_dot(const int64_t *qx, const uint16_t *row, int dim, int *bad)
{
    i128 acc = 0;
    for (int d = 0; d < dim; ++d) {
        int64_t y = half_to_fixed(row[d], bad);
        acc += (i128)qx[d] * (i128)y;
    }
    return acc;
}

/* monotone map fp16 bits -> unsigned (larger = greater value) */
static inline uint32_t half_order(uint16_t h)
{
    return (h & 0x8000) ? (uint32_t)((~h) & 0xffff) : (uint32_t)(h | 0x8000);
}

/* canonical score matrix: out[nq][n] fp16 bits.  returns 0 ok, 1 if a non-finite input was met */
int emdr2_oracle_scores(const uint16_t *rows, int64_t n, int dim,
                        const uint16_t *queries, int nq, uint16_t *out)
{
    int bad_any = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(|:bad_any)
    for (int qi = 0; qi < nq; ++qi) {
        int bad = 0;
        int64_t *qx = (int64_t *)malloc(sizeof(int64_t) * (size_t)dim);
        for (int d = 0; d < dim; ++d) qx[d] = half_to_fixed(queries[(size_t)qi * dim + d], &bad);
        for (int64_t r = 0; r < n; ++r)
            out[(size_t)qi * n + r] = fixed48_to_half(exact_dot(qx, rows + (size_t)r * dim, dim, &bad));
        free(qx);
        bad_any |= bad;
    }
    return bad_any;
}

/* min-heap on 64-bit keys (key = order(score) << 32 | ~row : larger key = better) */
static void sift_down(uint64_t *h, int n, int i)
{
    for (;;) {
        int l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && h[l] < h[m]) m = l;
        if (r < n && h[r] < h[m]) m = r;
        if (m == i) return;
        uint64_t t = h[i]; h[i] = h[m]; h[m] = t; i = m;
    }
}

static int cmp_desc(const void *a, const void *b)
{
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return (x < y) - (x > y);
}

/*
 * Canonical top-k.  ids may be NULL (then indices are row numbers + row_base).
 * out_dist [nq][k] fp16 bits, out_idx [nq][k] int32, out_row [nq][k] int64 (may be NULL).
 * k > n: the tail is filled with dist = 0xfc00 (-inf), idx = -1.
 */
int emdr2_oracle_topk(const uint16_t *rows, int64_t n, int dim, int64_t row_base,
                      const uint16_t *queries, int nq, int k, const int32_t *ids,
                      uint16_t *out_dist, int32_t *out_idx, int64_t *out_row)
{
    int bad_any = 0;
    if (n >= ((int64_t)1 << 32)) return 2;
#pragma omp parallel for schedule(dynamic, 1) reduction(|:bad_any)
    for (int qi = 0; qi < nq; ++qi) {
        int bad = 0;
        int64_t *qx = (int64_t *)malloc(sizeof(int64_t) * (size_t)dim);
        uint64_t *heap = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(k > 0 ? k : 1));
        int hn = 0;
        for (int d = 0; d < dim; ++d) qx[d] = half_to_fixed(queries[(size_t)qi * dim + d], &bad);
        for (int64_t r = 0; r < n; ++r) {
            uint16_t h = fixed48_to_half(exact_dot(qx, rows + (size_t)r * dim, dim, &bad));
            uint64_t key = ((uint64_t)half_order(h) << 32) | (uint64_t)(0xffffffffu - (uint32_t)r);
            if (hn < k) {
                heap[hn++] = key;
                if (hn == k) for (int i = k / 2 - 1; i >= 0; --i) sift_down(heap, k, i);
            } else if (k > 0 && key > heap[0]) {
                heap[0] = key; sift_down(heap, k, 0);
            }
        }
        qsort(heap, (size_t)hn, sizeof(uint64_t), cmp_desc);
        for (int j = 0; j < k; ++j) {
            size_t o = (size_t)qi * k + j;
            if (j < hn) {
                uint32_t ord = (uint32_t)(heap[j] >> 32);
                uint16_t h = (ord & 0x8000) ? (uint16_t)(ord & 0x7fff) : (uint16_t)((~ord) & 0xffff);
                int64_t r = (int64_t)(0xffffffffu - (uint32_t)(heap[j] & 0xffffffffu));
                out_dist[o] = h;
                out_idx[o] = ids ? ids[r] : (int32_t)(r + row_base);
                if (out_row) out_row[o] = r + row_base;
            } else {
                out_dist[o] = 0xfc00; out_idx[o] = -1;
                if (out_row) out_row[o] = -1;
            }
        }
        free(heap); free(qx);
        bad_any |= bad;
    }
    return bad_any;
}

/*
 * FaissMIPSIndex restatement (reference: megatron/data/emdr2_index.py:164-197; faiss.IndexFlatIP +
 * IndexIDMap, fp32 inner product over fp16-representable stored vectors):
 *   score = RNE_fp32(exact dot), order (score desc, row asc), int64 ids.
 * faiss itself is not vendored / not installable here: "parity unpinned" for this entry point.
 */
static inline float fixed48_to_float(i128 t)
{
    int neg = t < 0;
    u128 mag = neg ? (u128)(-t) : (u128)t;
    if (mag == 0) return 0.0f;
    int p = 127;
    while (!((mag >> p) & 1)) --p;
    int shift = p - 23;
    double v;
    if (shift <= 0) v = (double)(uint64_t)mag;          /* < 2^24: exact */
    else {
        u128 q = mag >> shift;
        u128 rem = mag & ((((u128)1) << shift) - 1);
        u128 half = ((u128)1) << (shift - 1);
        if (rem > half || (rem == half && (q & 1))) q += 1;
        v = ldexp((double)(uint64_t)q, shift);
    }
    v = ldexp(v, -48);
    return (float)(neg ? -v : v);                        /* exact: <= 24 significant bits */
}

static inline uint32_t float_order(float f)
{
    uint32_t u; memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

int emdr2_oracle_topk_f32(const uint16_t *rows, int64_t n, int dim,
                          const uint16_t *queries, int nq, int k, const int64_t *ids,
                          float *out_dist, int64_t *out_idx)
{
    int bad_any = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(|:bad_any)
    for (int qi = 0; qi < nq; ++qi) {
        int bad = 0;
        int64_t *qx = (int64_t *)malloc(sizeof(int64_t) * (size_t)dim);
        uint64_t *keys = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(k > 0 ? k : 1));
        int hn = 0;
        for (int d = 0; d < dim; ++d) qx[d] = half_to_fixed(queries[(size_t)qi * dim + d], &bad);
        for (int64_t r = 0; r < n; ++r) {
            float s = fixed48_to_float(exact_dot(qx, rows + (size_t)r * dim, dim, &bad));
            if (s == 0.0f) s = 0.0f;
            uint64_t key = ((uint64_t)float_order(s) << 32) | (uint64_t)(0xffffffffu - (uint32_t)r);
            if (hn < k) {
                keys[hn++] = key;
                if (hn == k) for (int i = k / 2 - 1; i >= 0; --i) sift_down(keys, k, i);
            } else if (k > 0 && key > keys[0]) {
                keys[0] = key; sift_down(keys, k, 0);
            }
        }
        qsort(keys, (size_t)hn, sizeof(uint64_t), cmp_desc);
        for (int j = 0; j < k; ++j) {
            size_t o = (size_t)qi * k + j;
            if (j < hn) {
                uint32_t ord = (uint32_t)(keys[j] >> 32);
                uint32_t u = (ord & 0x80000000u) ? (ord & 0x7fffffffu) : ~ord;
                float s; memcpy(&s, &u, 4);
                int64_t r = (int64_t)(0xffffffffu - (uint32_t)(keys[j] & 0xffffffffu));
                out_dist[o] = s; out_idx[o] = ids ? ids[r] : r;
            } else { out_dist[o] = -INFINITY; out_idx[o] = -1; }
        }
        free(keys); free(qx);
        bad_any |= bad;
    }
    return bad_any;
}

/*
 * Timed CPU "port" used by bench.py's cpu_baseline leg: the arithmetic the reference performs on
 * its devices (fp32-accumulate dot, one rounding to fp16, top-k), written as a blocked loop the
 * compiler can vectorise.  Not exact-sum: used for timing and loose cross-checks only.
 */
static inline float half_to_float(uint16_t h)
{
    uint32_t s = (uint32_t)(h & 0x8000) << 16, e = (h >> 10) & 0x1f, m = h & 0x3ff, u;
    if (e == 0) {
        if (m == 0) u = s;
        else { int sh = 0; while (!(m & 0x400)) { m <<= 1; ++sh; } u = s | ((uint32_t)(113 - sh) << 23) | ((m & 0x3ff) << 13); }
    } else if (e == 31) u = s | 0x7f800000u | (m << 13);
    else u = s | ((e + 112) << 23) | (m << 13);
    float f; memcpy(&f, &u, 4); return f;
}

static inline uint16_t float_to_half_rne(float f)
{
    uint32_t x; memcpy(&x, &f, 4);
    uint16_t sign = (uint16_t)((x >> 16) & 0x8000);
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | (x > 0x7f800000u ? 0x7e00 : 0x7c00));
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00);      /* >= 65520 rounds to inf */
    if (x <= 0x33000000u) return sign;                           /* <= 2^-25 rounds to zero */
    int e = (int)(x >> 23) - 127;
    uint32_t m = (x & 0x7fffffu) | 0x800000u;
    int shift = (e < -14) ? (13 + (-14 - e)) : 13;
    uint32_t q = m >> shift, rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1))) q += 1;
    if (e < -14) return (uint16_t)(sign | q);
    uint32_t field = (uint32_t)(e + 15);
    if (q == 2048) { q = 1024; field += 1; }
    if (field >= 31) return (uint16_t)(sign | 0x7c00);
    return (uint16_t)(sign | (field << 10) | (q - 1024));
}

int emdr2_oracle_topk_fp32accum(const uint16_t *rows, int64_t n, int dim,
                                const uint16_t *queries, int nq, int k,
                                uint16_t *out_dist, int32_t *out_row)
{
    float *qf = (float *)malloc(sizeof(float) * (size_t)nq * dim);
    for (size_t i = 0; i < (size_t)nq * dim; ++i) qf[i] = half_to_float(queries[i]);
    uint64_t *heaps = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)nq * (k > 0 ? k : 1));
    int *hn = (int *)calloc((size_t)nq, sizeof(int));
    enum { RB = 64 };
#pragma omp parallel
    {
        float *ef = (float *)malloc(sizeof(float) * RB * (size_t)dim);
        /* each thread owns a slice of queries; rows streamed in blocks of RB (converted once per block per thread) */
#pragma omp for schedule(static)
        for (int qi = 0; qi < nq; ++qi) {
            uint64_t *heap = heaps + (size_t)qi * k;
            const float *q = qf + (size_t)qi * dim;
            for (int64_t r0 = 0; r0 < n; r0 += RB) {
                int rb = (int)((n - r0) < RB ? (n - r0) : RB);
                for (int i = 0; i < rb; ++i)
                    for (int d = 0; d < dim; ++d) ef[(size_t)i * dim + d] = half_to_float(rows[(size_t)(r0 + i) * dim + d]);
                for (int i = 0; i < rb; ++i) {
                    const float *e = ef + (size_t)i * dim;
                    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                    for (int d = 0; d + 8 <= dim; d += 8)
                        for (int j = 0; j < 8; ++j) acc[j] += q[d + j] * e[d + j];
                    float s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
                    for (int d = dim & ~7; d < dim; ++d) s += q[d] * e[d];
                    uint16_t h = float_to_half_rne(s);
                    uint64_t key = ((uint64_t)half_order(h) << 32) | (uint64_t)(0xffffffffu - (uint32_t)(r0 + i));
                    if (hn[qi] < k) {
                        heap[hn[qi]++] = key;
                        if (hn[qi] == k) for (int t = k / 2 - 1; t >= 0; --t) sift_down(heap, k, t);
                    } else if (k > 0 && key > heap[0]) { heap[0] = key; sift_down(heap, k, 0); }
                }
            }
            qsort(heap, (size_t)hn[qi], sizeof(uint64_t), cmp_desc);
            for (int j = 0; j < k; ++j) {
                size_t o = (size_t)qi * k + j;
                if (j < hn[qi]) {
                    uint32_t ord = (uint32_t)(heap[j] >> 32);
                    out_dist[o] = (ord & 0x8000) ? (uint16_t)(ord & 0x7fff) : (uint16_t)((~ord) & 0xffff);
                    out_row[o] = (int32_t)(0xffffffffu - (uint32_t)(heap[j] & 0xffffffffu));
                } else { out_dist[o] = 0xfc00; out_row[o] = -1; }
            }
        }
        free(ef);
    }
    free(hn); free(heaps); free(qf);
    return 0;
}
