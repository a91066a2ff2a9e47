// emdr2_amd/csrc/mips_aux.hip -- everything around the scan: HBM re-layout, query packing,
// candidate selection, exact integer re-scoring + validity proof, shard merge, all-exact fallback.
#include "mips_device.h"
#include "mips_kernels.h"

#define CHECK_LAUNCH() (hipGetLastError() == hipSuccess ? 0 : -3)

// ------------------------------------------------------------------------------------------------
// index re-layout (reference: add_embed_data uploads a dense row-major fp16 matrix,
// megatron/data/emdr2_index.py:248-256; here the upload is re-tiled into scan order)
// ------------------------------------------------------------------------------------------------
__global__ void pack_rows_kernel(const uint4 *__restrict__ rows_rm, int64_t n_chunk, int nseg, int64_t row_offset,
                                 char *__restrict__ tiled)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_chunk * nseg) return;
    const int64_t r = i / nseg;
    const int seg = (int)(i - r * nseg);
    *(uint4 *)(tiled + tiled_seg_offset(row_offset + r, seg, nseg >> 2)) = rows_rm[i];
}

__global__ void row_norm_max_kernel(const uint4 *__restrict__ rows_rm, int64_t n_chunk, int nseg, float *emax_sq)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    float best = 0.f;
    for (int64_t r = wave; r < n_chunk; r += nwaves) {
        float s = 0.f;
        for (int seg = lane; seg < nseg; seg += 64) {
            const uint4 v = rows_rm[r * nseg + seg];
            const half8 h = __builtin_bit_cast(half8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float x = (float)h[j]; s += x * x; }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
        best = fmaxf(best, s);
    }
    if (lane == 0 && best > 0.f) atomicMax((unsigned *)emax_sq, __float_as_uint(best)); // non-negative floats order as uints
}

int mips_launch_pack_rows(const void *rows_rm, int64_t n_chunk, int dim, int64_t row_offset, void *tiled,
                          float *emax_sq, hipStream_t stream)
{
    const int nseg = dim / 8;
    const int64_t total = n_chunk * nseg;
    if (total == 0) return 0;
    const int64_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const uint4 *)rows_rm, n_chunk, nseg,
                       row_offset, (char *)tiled);
    int nb = (int)((n_chunk + 3) / 4);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(row_norm_max_kernel, dim3(nb), dim3(256), 0, stream, (const uint4 *)rows_rm, n_chunk, nseg, emax_sq);
    return CHECK_LAUNCH();
}

__global__ void unpack_rows_kernel(const char *__restrict__ tiled, int nseg, const int64_t *__restrict__ row_ids,
                                   int64_t n_out, uint4 *__restrict__ rows_rm)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out * nseg) return;
    const int64_t r = i / nseg;
    const int seg = (int)(i - r * nseg);
    rows_rm[i] = *(const uint4 *)(tiled + tiled_seg_offset(row_ids[r], seg, nseg >> 2));
}

int mips_launch_unpack_rows(const void *tiled, int dim, const int64_t *row_ids, int64_t n_out, void *rows_rm,
                            hipStream_t stream)
{
    const int nseg = dim / 8;
    const int64_t total = n_out * nseg;
    if (total == 0) return 0;
    hipLaunchKernelGGL(unpack_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const char *)tiled, nseg,
                       row_ids, n_out, (uint4 *)rows_rm);
    return CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------
// query packing: [n_q, dim] row-major -> per chunk a [bn rows x 64 B] swizzled block (zero padded)
// ------------------------------------------------------------------------------------------------
__global__ void pack_queries_kernel(const uint4 *__restrict__ queries, int n_q, int nseg, int bn, char *__restrict__ q_tiled,
                                    float *__restrict__ qnorm)
{
    const int q = blockIdx.x; // one wave per padded query row
    const int lane = threadIdx.x;
    float s = 0.f;
    for (int seg = lane; seg < nseg; seg += 64) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (q < n_q) {
            v = queries[(size_t)q * nseg + seg];
            const half8 h = __builtin_bit_cast(half8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float x = (float)h[j]; s += x * x; }
        }
        const int c = seg >> 2, sp = (seg & 3) ^ ((q >> 2) & 3);
        *(uint4 *)(q_tiled + ((size_t)c * bn * 4 + (size_t)q * 4 + sp) * 16) = v;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0 && q < n_q) qnorm[q] = sqrtf(s) * 1.001f;
}

__global__ void pack_queries_frag_kernel(const uint4 *__restrict__ queries, int n_q, int nseg, char *__restrict__ q_frag)
{
    const int q = blockIdx.x; // 512 padded query rows, one wave each
    const int lane0 = threadIdx.x;
    const int wave = q >> 6, ni = (q >> 5) & 1, l31 = q & 31;
    for (int seg = lane0; seg < nseg; seg += 64) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (q < n_q) v = queries[(size_t)q * nseg + seg];
        const int c = seg >> 2, s = seg & 3, ks = s >> 1, hi = s & 1;
        const int lane = hi * 32 + l31;
        *(uint4 *)(q_frag + ((((size_t)(c * 8 + wave) * 2 + ks) * 2 + ni) * 64 + lane) * 16) = v;
    }
}

int mips_launch_pack_queries_frag(const void *queries, int n_q, int dim, void *q_frag, hipStream_t stream)
{
    hipLaunchKernelGGL(pack_queries_frag_kernel, dim3(512), dim3(64), 0, stream, (const uint4 *)queries, n_q, dim / 8, (char *)q_frag);
    return CHECK_LAUNCH();
}

int mips_launch_pack_queries(const void *queries, int n_q, int dim, int bn, void *q_tiled, float *qnorm,
                             hipStream_t stream)
{
    hipLaunchKernelGGL(pack_queries_kernel, dim3(bn), dim3(64), 0, stream, (const uint4 *)queries, n_q, dim / 8, bn, (char *)q_tiled,
                       qnorm);
    return CHECK_LAUNCH();
}

__global__ void init_kernel(float *tau, unsigned *count, unsigned *flags, int bn, int n_q, unsigned dense_count, uint4 *zero16, unsigned n_zero16)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < bn) {
        tau[i] = -__builtin_inff();
        count[i] = (i < n_q) ? dense_count : 0u;
    }
    if (i < n_q) flags[i] = 0u;
    // the pair-progress counters of the persistent scan launches (mips_scan8.hip), zeroed here instead of by a memset of their own
    for (unsigned j = (unsigned)i; j < n_zero16; j += gridDim.x * blockDim.x) zero16[j] = make_uint4(0u, 0u, 0u, 0u);
}

int mips_launch_init(float *tau, unsigned *count, unsigned *flags, int bn, int n_q, unsigned dense_count, unsigned *zero, size_t n_zero,
                     hipStream_t stream)
{
    if (n_zero & 3) return -1;
    const int blocks = zero && n_zero ? 128 : (bn + 255) / 256;
    hipLaunchKernelGGL(init_kernel, dim3(blocks < (bn + 255) / 256 ? (bn + 255) / 256 : blocks), dim3(256), 0, stream, tau, count, flags, bn, n_q, dense_count,
                       (uint4 *)zero, (unsigned)(zero ? n_zero / 4 : 0));
    return CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------
// candidate selection: keep the kp largest keys (fp32 score desc, row asc) of each query
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t cand_key(uint2 e)
{
    return ((uint64_t)f32_order(__uint_as_float(e.x)) << 32) | (uint64_t)(0xffffffffu - e.y);
}

// wave 0: find the digit whose descending cumulative count first reaches `need`
__device__ __forceinline__ void pick_digit(const unsigned *hist, unsigned need, unsigned *out_digit, unsigned *out_need)
{
    const int lane = threadIdx.x;
    unsigned h[4], s = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { h[j] = hist[255 - (4 * lane + j)]; s += h[j]; }
    unsigned incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    unsigned c = incl - s;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (c < need && need <= c + h[j]) { *out_digit = 255 - (4 * lane + j); *out_need = need - c; }
        c += h[j];
    }
}

// One block of 256 threads reduces query q's candidate list to its kp largest keys (radix select, 8 digits of 8 bits from the top) and sets
// tau[q] to the kp-th score.  r04: lists of up to SEL_LDS keys (the dense first segment has 8,192) are staged in LDS once instead of being
// re-read from L2 by each of the eight passes, and a thread adds RUNS of equal digits to the histogram with one atomic -- scores of one query
// share their sign / exponent byte, so the first passes used to serialise thousands of LDS atomics on one or two bins (96 us for the first
// select of a search, 18 us for the later ones: a third of what a search on an N / 8 row shard spends outside its scan).
#define SEL_LDS 8192
struct SelectShared {
    uint64_t key[SEL_LDS];
    unsigned hist[256];
    unsigned digit, need, out;
    uint2 keep[128];
};
// entry i of query q's whole candidate set: the main list [0, n_main) followed by the eight per-XCD sub-lists (lengths sub[x])
struct CandView {
    const uint2 *main_list, *sub_lists;   // [capq], [8][SUBCAP]
    unsigned n_main, sub_end[8];          // sub_end[x] = n_main + sub[0] + .. + sub[x]
    __device__ __forceinline__ uint2 at(unsigned i) const
    {
        if (i < n_main) return main_list[i];
        unsigned lo = n_main;
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            if (i < sub_end[x]) return sub_lists[x * SUBCAP + (i - lo)];
            lo = sub_end[x];
        }
        return main_list[0];
    }
};

__device__ __forceinline__ void select_block(SelectShared &sh, uint2 *cand_all, unsigned *count, uint2 *cand8, unsigned *count8, float *tau, unsigned *flags,
                                             unsigned capq, int kp, int q)
{
    const int tid = threadIdx.x;
    uint2 *cand = cand_all + (size_t)q * capq;
    CandView cv;
    cv.main_list = cand;
    cv.sub_lists = cand8 ? cand8 + (size_t)q * 8 * SUBCAP : nullptr;
    unsigned n = count[q];
    if (n > capq) { if (tid == 0) { atomicOr(&flags[q], 2u); count[q] = capq; } n = capq; }
    cv.n_main = n;
    bool any_sub = false;
#pragma unroll
    for (int x = 0; x < 8; ++x) {
        unsigned m = cand8 ? count8[x * 512 + q] : 0u;
        if (m > SUBCAP) m = SUBCAP;                                     // (a full sub-list spilled its later survivors into the main list: mips_scan8.hip s8_append)
        any_sub |= m != 0;
        n += m;
        cv.sub_end[x] = n;
    }
    __syncthreads();                                                  // (everybody has read the sub-list counts before they are reset below)
    if (any_sub && tid < 8) count8[tid * 512 + q] = 0;                // the next segment appends to empty sub-lists
    if (n <= (unsigned)kp) {
        if (any_sub) {                                                // fewer than kp candidates in all: move the sub-list entries behind the main ones
            for (unsigned i = cv.n_main + tid; i < n; i += 256) cand[i] = cv.at(i);
            if (tid == 0) count[q] = n;
        }
        return;
    }
    const bool staged = n <= SEL_LDS;
    if (staged) {
        // eight loads in flight per thread (one at a time, the 32 rounds of an 8,192-entry list each paid a full L2 latency)
        for (unsigned i0 = tid; i0 < n; i0 += 8 * 256) {
            uint2 e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { const unsigned i = i0 + j * 256; e[j] = cv.at(i < n ? i : n - 1); }
#pragma unroll
            for (int j = 0; j < 8; ++j) { const unsigned i = i0 + j * 256; if (i < n) sh.key[i] = cand_key(e[j]); }
        }
    }
    uint64_t prefix = 0, mask = 0;
    unsigned need = (unsigned)kp;
    for (int pass = 0; pass < 8; ++pass) {
        const int shift = 56 - 8 * pass;
        sh.hist[tid] = 0;
        __syncthreads();
        unsigned run_digit = 0xffffffffu, run = 0;
        for (unsigned i = tid; i < n; i += 256) {
            const uint64_t k = staged ? sh.key[i] : cand_key(cv.at(i));
            if ((k & mask) == prefix) {
                const unsigned d = (unsigned)(k >> shift) & 255u;
                if (d != run_digit) {
                    if (run) atomicAdd(&sh.hist[run_digit], run);
                    run_digit = d; run = 0;
                }
                ++run;
            }
        }
        if (run) atomicAdd(&sh.hist[run_digit], run);
        __syncthreads();
        if (tid < 64) pick_digit(sh.hist, need, &sh.digit, &sh.need);
        __syncthreads();
        prefix |= (uint64_t)sh.digit << shift;
        mask |= (uint64_t)0xff << shift;
        need = sh.need;
        const unsigned same = sh.hist[sh.digit];                      // keys that share the prefix so far
        __syncthreads();
        // r05: the upper half of a key is its fp32 score, the lower half only breaks ties by row.  Once the four score digits are fixed and ALL
        // keys with that score are wanted (no tie across the boundary: the usual case), every key >= (score, row bits 0) is in and the four
        // row passes have nothing to decide: half of every select's passes.
        if (pass == 3 && same == need) break;
    }
    // prefix is now the kp-th largest key; keys are unique, so exactly kp keys are >= prefix
    if (tid == 0) sh.out = 0;
    __syncthreads();
    for (unsigned i = tid; i < n; i += 256) {
        const uint64_t k = staged ? sh.key[i] : cand_key(cv.at(i));
        if (k >= prefix) {                                            // (a key is its entry: score bits from the ordered form, row from the low word)
            const unsigned s = atomicAdd(&sh.out, 1u);
            if (s < 128) sh.keep[s] = make_uint2(__float_as_uint(f32_unorder((uint32_t)(k >> 32))), 0xffffffffu - (uint32_t)k);
        }
    }
    __syncthreads();
    if (tid < kp) cand[tid] = sh.keep[tid];
    if (tid == 0) {
        count[q] = (unsigned)kp;
        tau[q] = f32_unorder((uint32_t)(prefix >> 32));
    }
}

__global__ void __launch_bounds__(256) select_kernel(uint2 *cand_all, unsigned *count, uint2 *cand8, unsigned *count8, float *tau, unsigned *flags, unsigned capq, int kp)
{
    __shared__ SelectShared sh;
    select_block(sh, cand_all, count, cand8, count8, tau, flags, capq, kp, blockIdx.x);
}

int mips_launch_select(uint2 *cand, unsigned *count, uint2 *cand8, unsigned *count8, float *tau, unsigned *flags, unsigned capq, int kp, int n_q,
                       hipStream_t stream)
{
    hipLaunchKernelGGL(select_kernel, dim3(n_q), dim3(256), 0, stream, cand, count, cand8, count8, tau, flags, capq, kp);
    return CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------
// finalize: exact integer re-scoring of <= kp candidates, canonical (fp16 score desc, row asc)
// order, proof that no pruned row can enter the top-k, output
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void exact_dot_wave(const char *e_tiled, int64_t row, const uint16_t *qrow, int nseg, int lane, int64_t &lo_out,
                                               int64_t &hi_out)
{
    int64_t lo = 0, hi = 0;
    for (int seg = lane; seg < nseg; seg += 64) {
        const uint4 ev = *(const uint4 *)(e_tiled + tiled_seg_offset(row, seg, nseg >> 2));
        const uint4 qv = *(const uint4 *)(qrow + seg * 8);
        const uint32_t ew[4] = {ev.x, ev.y, ev.z, ev.w}, qw[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            exact_mac(lo, hi, half_fix((uint16_t)(ew[j] & 0xffff)), half_fix((uint16_t)(qw[j] & 0xffff)));
            exact_mac(lo, hi, half_fix((uint16_t)(ew[j] >> 16)), half_fix((uint16_t)(qw[j] >> 16)));
        }
    }
    lo_out = wave_sum_i64(lo);
    hi_out = wave_sum_i64(hi);
}

// Four candidates of one wave at a time (dim <= 1024: a row is <= 2 segments per lane): all eight row loads go out before the first is
// used.  One candidate after the other, each of a wave's 16 - 32 candidates paid a full memory latency for its scattered 16-byte row segments
// (r04: ~50 of the finalize launch's 82 us on an N / 8 shard); the integer arithmetic itself is a few hundred instructions per candidate.
__device__ __forceinline__ void exact_dot_wave_x4(const char *e_tiled, const unsigned (&rows)[4], int n_valid, const uint16_t *qrow, int nseg, int lane,
                                                  int64_t (&lo_out)[4], int64_t (&hi_out)[4])
{
    const bool two = lane + 64 < nseg, one = lane < nseg;
    uint4 ev[4][2];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        ev[c][0] = ev[c][1] = make_uint4(0, 0, 0, 0);
        if (c < n_valid) {
            if (one) ev[c][0] = *(const uint4 *)(e_tiled + tiled_seg_offset((int64_t)rows[c], lane, nseg >> 2));
            if (two) ev[c][1] = *(const uint4 *)(e_tiled + tiled_seg_offset((int64_t)rows[c], lane + 64, nseg >> 2));
        }
    }
    uint4 qv[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
    if (one) qv[0] = *(const uint4 *)(qrow + lane * 8);
    if (two) qv[1] = *(const uint4 *)(qrow + (lane + 64) * 8);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        int64_t lo = 0, hi = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t ew[4] = {ev[c][h].x, ev[c][h].y, ev[c][h].z, ev[c][h].w}, qw[4] = {qv[h].x, qv[h].y, qv[h].z, qv[h].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {                          // (absent segments hold zeros: mantissa 0, nothing added)
                exact_mac(lo, hi, half_fix((uint16_t)(ew[j] & 0xffff)), half_fix((uint16_t)(qw[j] & 0xffff)));
                exact_mac(lo, hi, half_fix((uint16_t)(ew[j] >> 16)), half_fix((uint16_t)(qw[j] >> 16)));
            }
        }
        lo_out[c] = wave_sum_i64(lo);
        hi_out[c] = wave_sum_i64(hi);
    }
}

template <bool SELECT_FIRST>
__global__ void __launch_bounds__(256) finalize_kernel(FinalizeParams p)
{
    __shared__ uint64_t fkey[128];
    __shared__ uint32_t sh_kth;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if constexpr (SELECT_FIRST) {
        // the select that follows the LAST scan segment, run by the block that consumes its result (one launch less per search; the list, its
        // count and tau go through global memory as between two kernels: the fence orders them for this block's own reads below)
        __shared__ SelectShared sel;
        select_block(sel, (uint2 *)p.cand, (unsigned *)p.count, p.cand8, p.count8, (float *)p.tau, p.flags, p.capq, p.kp, q);
        __threadfence_block();
        __syncthreads();
    }
    const uint2 *cand = p.cand + (size_t)q * p.capq;
    unsigned cnt = p.count[q];
    if (cnt > (unsigned)p.kp) cnt = p.kp;
    const int nseg = p.dim / 8;
    const uint16_t *qrow = p.queries + (size_t)q * p.dim;

    if (nseg <= 128) {
        for (unsigned j0 = wave; j0 < cnt; j0 += 16) {                        // this wave's candidates j0, j0 + 4, j0 + 8, j0 + 12 together
            unsigned rows[4];
            int nv = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) { const unsigned j = j0 + 4 * c; rows[c] = j < cnt ? cand[j].y : 0u; nv += j < cnt; }
            int64_t lo[4], hi[4];
            exact_dot_wave_x4(p.e_tiled, rows, nv, qrow, nseg, lane, lo, hi);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const unsigned j = j0 + 4 * c;
                if (j < cnt && lane == 0) {
                    const uint32_t ord = p.f32 ? f32_order(fixed_to_float(lo[c], hi[c])) : h16_order(fixed_to_half(lo[c], hi[c]));
                    fkey[j] = ((uint64_t)ord << 32) | (uint64_t)(0xffffffffu - rows[c]);
                }
            }
        }
    } else {
        for (unsigned j = wave; j < cnt; j += 4) {
            const unsigned row = cand[j].y;
            int64_t lo, hi;
            exact_dot_wave(p.e_tiled, (int64_t)row, qrow, nseg, lane, lo, hi);
            const uint32_t ord = p.f32 ? f32_order(fixed_to_float(lo, hi)) : h16_order(fixed_to_half(lo, hi));
            if (lane == 0) fkey[j] = ((uint64_t)ord << 32) | (uint64_t)(0xffffffffu - row);
        }
    }
    if (tid == 0) sh_kth = 0;
    __syncthreads();

    const unsigned kk = cnt < (unsigned)p.k ? cnt : (unsigned)p.k;
    if ((unsigned)tid < cnt) {
        const uint64_t mine = fkey[tid];
        unsigned rank = 0;
        for (unsigned i = 0; i < cnt; ++i) rank += (fkey[i] > mine);
        if (rank < kk) {
            const uint32_t row = 0xffffffffu - (uint32_t)mine;
            const size_t o = (size_t)q * p.k + rank;
            const int64_t grow = p.row_base + (int64_t)row;
            const int32_t gid = p.ids ? p.ids[row] : (int32_t)grow;
            if (p.out_rec) {
                const uint32_t bits = p.f32 ? __float_as_uint(f32_unorder((uint32_t)(mine >> 32))) : (uint32_t)h16_unorder((uint32_t)(mine >> 32));
                p.out_rec[o] = make_uint4((uint32_t)grow, (uint32_t)((uint64_t)grow >> 32), (uint32_t)gid, bits);
            } else {
                if (p.f32) ((float *)p.out_dist)[o] = f32_unorder((uint32_t)(mine >> 32));
                else ((uint16_t *)p.out_dist)[o] = h16_unorder((uint32_t)(mine >> 32));
                p.out_row[o] = grow;
                p.out_idx[o] = gid;
            }
            if (rank == kk - 1) sh_kth = (uint32_t)(mine >> 32);
        }
    }
    for (unsigned j = kk + tid; j < (unsigned)p.k; j += 256) { // shard holds fewer than k rows
        const size_t o = (size_t)q * p.k + j;
        if (p.out_rec) { p.out_rec[o] = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, p.f32 ? 0xff800000u : 0xfc00u); continue; }
        if (p.f32) ((float *)p.out_dist)[o] = -INFINITY; else ((uint16_t *)p.out_dist)[o] = 0xfc00;
        p.out_row[o] = -1; p.out_idx[o] = -1;
    }
    __syncthreads();
    if (tid == 0 && p.n_rows > (int64_t)cnt) {
        // rows outside the candidate list have MFMA score S~ <= tau (the kp-th best S~), hence exact
        // score <= tau + eps with eps >= |S~ - exact| for a dim-term fp32 accumulation;
        // if even that bound rounds below the k-th canonical score the top-k is proven.
        bool ok = (cnt == (unsigned)p.kp) && (kk == (unsigned)p.k);
        if (ok) {
            const float eps = (float)p.dim * 2.3841858e-07f /* 2^-22 */ * p.qnorm[q] * (sqrtf(*p.emax_sq) * 1.001f);
            const float bound = p.tau[q] + eps * 1.0001f;
            // fp32 mode: a pruned row's exact score is <= bound (a float), so its RNE_fp32 is <= bound as well
            ok = (p.f32 ? f32_order(bound) : h16_order(f32_to_h16_roundup(bound))) < sh_kth;
        }
        if (!ok) atomicOr(&p.flags[q], 1u);
    }
}

int mips_launch_finalize(const FinalizeParams &p, bool select_first, hipStream_t stream)
{
    if (select_first) hipLaunchKernelGGL(finalize_kernel<true>, dim3(p.n_q), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(finalize_kernel<false>, dim3(p.n_q), dim3(256), 0, stream, p);
    return CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------
// shard merge: [S, n_q, k] per-shard canonical lists -> [n_q, k], (score desc, global row asc)
// ------------------------------------------------------------------------------------------------
#define MERGE_MAX 4096
__global__ void __launch_bounds__(256) merge_kernel(const uint16_t *dist_in, const int32_t *idx_in, const int64_t *row_in, int n_shards,
                                                    int n_q, int k, uint16_t *out_dist, int32_t *out_idx, int64_t *out_row)
{
    __shared__ uint64_t key[MERGE_MAX];
    __shared__ unsigned nvalid;
    const int q = blockIdx.x, tid = threadIdx.x;
    const int n = n_shards * k;
    if (tid == 0) nvalid = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        const int s = i / k, j = i - s * k;
        const size_t src = ((size_t)s * n_q + q) * k + j;
        const int64_t row = row_in[src];
        uint64_t kv = 0;
        if (row >= 0) {
            kv = ((uint64_t)h16_order(dist_in[src]) << 48) | (0xffffffffffffull - (uint64_t)row);
            atomicAdd(&nvalid, 1u);
        }
        key[i] = kv;
    }
    __syncthreads();
    const unsigned nv = nvalid;
    for (int i = tid; i < n; i += 256) {
        const uint64_t mine = key[i];
        if (mine == 0) continue;
        unsigned rank = 0;
        for (int t = 0; t < n; ++t) rank += (key[t] > mine);
        if (rank < (unsigned)k) {
            const int s = i / k, j = i - s * k;
            const size_t src = ((size_t)s * n_q + q) * k + j, o = (size_t)q * k + rank;
            out_dist[o] = dist_in[src]; out_idx[o] = idx_in[src]; out_row[o] = row_in[src];
        }
    }
    for (unsigned j = nv + tid; j < (unsigned)k; j += 256) {
        const size_t o = (size_t)q * k + j;
        out_dist[o] = 0xfc00; out_idx[o] = -1; out_row[o] = -1;
    }
}

int mips_launch_merge(const uint16_t *dist_in, const int32_t *idx_in, const int64_t *row_in, int n_shards, int n_q,
                      int k, uint16_t *out_dist, int32_t *out_idx, int64_t *out_row, hipStream_t stream)
{
    if (n_shards * k > MERGE_MAX) return -4;
    hipLaunchKernelGGL(merge_kernel, dim3(n_q), dim3(256), 0, stream, dist_in, idx_in, row_in, n_shards, n_q, k, out_dist, out_idx,
                       out_row);
    return CHECK_LAUNCH();
}

// fp32-score twin: key = (f32 order : 32 | ~row : 32); global rows < 2^32
__global__ void __launch_bounds__(256) merge_f32_kernel(const float *dist_in, const int32_t *idx_in, const int64_t *row_in, int n_shards, int n_q,
                                                        int k, float *out_dist, int32_t *out_idx, int64_t *out_row)
{
    __shared__ uint64_t key[MERGE_MAX];
    __shared__ unsigned nvalid;
    const int q = blockIdx.x, tid = threadIdx.x;
    const int n = n_shards * k;
    if (tid == 0) nvalid = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        const int s = i / k, j = i - s * k;
        const size_t src = ((size_t)s * n_q + q) * k + j;
        const int64_t row = row_in[src];
        uint64_t kv = 0;
        if (row >= 0) {
            kv = ((uint64_t)f32_order(dist_in[src]) << 32) | (uint64_t)(0xffffffffu - (uint32_t)row);
            atomicAdd(&nvalid, 1u);
        }
        key[i] = kv;
    }
    __syncthreads();
    const unsigned nv = nvalid;
    for (int i = tid; i < n; i += 256) {
        const uint64_t mine = key[i];
        if (mine == 0) continue;
        unsigned rank = 0;
        for (int t = 0; t < n; ++t) rank += (key[t] > mine);
        if (rank < (unsigned)k) {
            const int s = i / k, j = i - s * k;
            const size_t src = ((size_t)s * n_q + q) * k + j, o = (size_t)q * k + rank;
            out_dist[o] = dist_in[src]; out_idx[o] = idx_in[src]; out_row[o] = row_in[src];
        }
    }
    for (unsigned j = nv + tid; j < (unsigned)k; j += 256) {
        const size_t o = (size_t)q * k + j;
        out_dist[o] = -INFINITY; out_idx[o] = -1; out_row[o] = -1;
    }
}

int mips_launch_merge_f32(const float *dist_in, const int32_t *idx_in, const int64_t *row_in, int n_shards, int n_q, int k, float *out_dist,
                          int32_t *out_idx, int64_t *out_row, hipStream_t stream)
{
    if (n_shards * k > MERGE_MAX) return -4;
    hipLaunchKernelGGL(merge_f32_kernel, dim3(n_q), dim3(256), 0, stream, dist_in, idx_in, row_in, n_shards, n_q, k, out_dist, out_idx, out_row);
    return CHECK_LAUNCH();
}

// the merge over gathered 16-byte records [n_shards, n_q, k] (FinalizeParams::out_rec): what a sharded search exchanges in its ONE
// all-gather is the finalize kernel's own output, and this kernel reads the gathered buffer as it arrives -- no casts / stacks between.
// Same keys, same order, same outputs as merge_kernel / merge_f32_kernel.
template <bool F32>
__global__ void __launch_bounds__(256) merge_records_kernel(const uint4 *rec_in, int n_shards, int n_q, int k, void *out_dist, int32_t *out_idx,
                                                            int64_t *out_row)
{
    __shared__ uint64_t key[MERGE_MAX];
    __shared__ unsigned nvalid;
    const int q = blockIdx.x, tid = threadIdx.x;
    const int n = n_shards * k;
    if (tid == 0) nvalid = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        const int s = i / k, j = i - s * k;
        const uint4 r = rec_in[((size_t)s * n_q + q) * k + j];
        const int64_t row = (int64_t)(((uint64_t)r.y << 32) | r.x);
        uint64_t kv = 0;
        if (row >= 0) {
            kv = F32 ? (((uint64_t)f32_order(__uint_as_float(r.w)) << 32) | (uint64_t)(0xffffffffu - (uint32_t)row))
                     : (((uint64_t)h16_order((uint16_t)r.w) << 48) | (0xffffffffffffull - (uint64_t)row));
            atomicAdd(&nvalid, 1u);
        }
        key[i] = kv;
    }
    __syncthreads();
    const unsigned nv = nvalid;
    for (int i = tid; i < n; i += 256) {
        const uint64_t mine = key[i];
        if (mine == 0) continue;
        unsigned rank = 0;
        for (int t = 0; t < n; ++t) rank += (key[t] > mine);
        if (rank < (unsigned)k) {
            const int s = i / k, j = i - s * k;
            const uint4 r = rec_in[((size_t)s * n_q + q) * k + j];
            const size_t o = (size_t)q * k + rank;
            if (F32) ((float *)out_dist)[o] = __uint_as_float(r.w); else ((uint16_t *)out_dist)[o] = (uint16_t)r.w;
            out_idx[o] = (int32_t)r.z; out_row[o] = (int64_t)(((uint64_t)r.y << 32) | r.x);
        }
    }
    for (unsigned j = nv + tid; j < (unsigned)k; j += 256) {
        const size_t o = (size_t)q * k + j;
        if (F32) ((float *)out_dist)[o] = -INFINITY; else ((uint16_t *)out_dist)[o] = 0xfc00;
        out_idx[o] = -1; out_row[o] = -1;
    }
}

int mips_launch_merge_records(const uint4 *rec_in, int n_shards, int n_q, int k, int f32, void *out_dist, int32_t *out_idx, int64_t *out_row,
                              hipStream_t stream)
{
    if (n_shards * k > MERGE_MAX) return -4;
    if (f32) hipLaunchKernelGGL(merge_records_kernel<true>, dim3(n_q), dim3(256), 0, stream, rec_in, n_shards, n_q, k, out_dist, out_idx, out_row);
    else hipLaunchKernelGGL(merge_records_kernel<false>, dim3(n_q), dim3(256), 0, stream, rec_in, n_shards, n_q, k, out_dist, out_idx, out_row);
    return CHECK_LAUNCH();
}

__global__ void __launch_bounds__(128) pack_records_kernel(const void *dist, const int32_t *idx, const int64_t *row, const int32_t *sel, int k,
                                                           int f32, uint4 *rec)
{
    const int q = sel[blockIdx.x];
    for (int j = threadIdx.x; j < k; j += 128) {
        const size_t o = (size_t)q * k + j;
        const int64_t r = row[o];
        const uint32_t bits = f32 ? __float_as_uint(((const float *)dist)[o]) : (uint32_t)((const uint16_t *)dist)[o];
        rec[o] = make_uint4((uint32_t)r, (uint32_t)((uint64_t)r >> 32), (uint32_t)idx[o], bits);
    }
}

int mips_launch_pack_records(const void *dist, const int32_t *idx, const int64_t *row, const int32_t *sel, int n_sel, int k, int f32, uint4 *rec,
                             hipStream_t stream)
{
    if (n_sel < 1) return 0;
    hipLaunchKernelGGL(pack_records_kernel, dim3(n_sel), dim3(128), 0, stream, dist, idx, row, sel, k, f32, rec);
    return CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------
// all-exact fallback: canonical fp16 keys for EVERY row in integer arithmetic, then a two-level
// radix threshold + ordered collection (first rows win ties).  Up to 8 queries per index pass.
// ------------------------------------------------------------------------------------------------
#define XQ 8
__global__ void __launch_bounds__(256) exact_scores_kernel(const char *__restrict__ e_tiled, int64_t n_rows, int dim,
                                                           const uint16_t *__restrict__ queries, const int32_t *__restrict__ sel,
                                                           int n_sel, uint16_t *__restrict__ hkeys)
{
    extern __shared__ int qfix[]; // [n_sel][dim] packed (mant << 8) | shift
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < n_sel * dim; i += 256) {
        const int f = i / dim, d = i - f * dim;
        const HalfFix hf = half_fix(queries[(size_t)sel[f] * dim + d]);
        qfix[i] = (hf.mant << 8) | hf.shift;
    }
    __syncthreads();
    const int nseg = dim / 8;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < n_rows; row += (int64_t)gridDim.x * 4) {
        int64_t lo[XQ], hi[XQ];
#pragma unroll
        for (int f = 0; f < XQ; ++f) { lo[f] = 0; hi[f] = 0; }
        for (int seg = lane; seg < nseg; seg += 64) {
            const uint4 ev = *(const uint4 *)(e_tiled + tiled_seg_offset(row, seg, nseg >> 2));
            const uint32_t ew[4] = {ev.x, ev.y, ev.z, ev.w};
            HalfFix ef[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) { ef[2 * j] = half_fix((uint16_t)(ew[j] & 0xffff)); ef[2 * j + 1] = half_fix((uint16_t)(ew[j] >> 16)); }
#pragma unroll
            for (int f = 0; f < XQ; ++f) {
                if (f < n_sel) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int pq = qfix[f * dim + seg * 8 + j];
                        HalfFix qf; qf.mant = pq >> 8; qf.shift = pq & 0xff;
                        exact_mac(lo[f], hi[f], ef[j], qf);
                    }
                }
            }
        }
#pragma unroll
        for (int f = 0; f < XQ; ++f) {
            if (f < n_sel) {
                const int64_t l = wave_sum_i64(lo[f]), h = wave_sum_i64(hi[f]);
                if (lane == 0) hkeys[(size_t)f * n_rows + row] = (uint16_t)h16_order(fixed_to_half(l, h));
            }
        }
    }
}

int mips_launch_exact_scores(const char *e_tiled, int64_t n_rows, int dim, const uint16_t *queries, const int32_t *sel,
                             int n_sel, uint16_t *hkeys, hipStream_t stream)
{
    if (n_sel > XQ) return -1;
    int64_t nb = (n_rows + 3) / 4;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(exact_scores_kernel, dim3((unsigned)nb), dim3(256), (size_t)n_sel * dim * sizeof(int), stream, e_tiled, n_rows, dim,
                       queries, sel, n_sel, hkeys);
    return CHECK_LAUNCH();
}

#define XS_THREADS 1024
#define XS_R 8
__global__ void __launch_bounds__(XS_THREADS) exact_select_kernel(const uint16_t *__restrict__ hkeys, int64_t n_rows, int64_t row_base,
                                                                  const int32_t *__restrict__ sel, int k, const int32_t *__restrict__ ids,
                                                                  uint16_t *out_dist, int32_t *out_idx, int64_t *out_row, unsigned *flags)
{
    __shared__ unsigned hist[256];
    __shared__ unsigned sh_digit, sh_need, sh_above, sh_wsum[XS_THREADS / 64];
    __shared__ uint64_t okey[128];
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint16_t *hk = hkeys + (size_t)f * n_rows;
    const unsigned keff = (int64_t)k < n_rows ? (unsigned)k : (unsigned)n_rows;

    // level 1: high byte
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    for (int64_t i = tid; i < n_rows; i += XS_THREADS) atomicAdd(&hist[hk[i] >> 8], 1u);
    __syncthreads();
    if (tid < 64) pick_digit(hist, keff, &sh_digit, &sh_need);
    __syncthreads();
    const unsigned bhi = sh_digit, need1 = sh_need;
    __syncthreads();
    // level 2: low byte inside the boundary high-byte bucket
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    for (int64_t i = tid; i < n_rows; i += XS_THREADS) { const unsigned v = hk[i]; if ((v >> 8) == bhi) atomicAdd(&hist[v & 255u], 1u); }
    __syncthreads();
    if (tid < 64) pick_digit(hist, need1, &sh_digit, &sh_need);
    __syncthreads();
    const unsigned T = (bhi << 8) | sh_digit;   // k-th canonical key
    const unsigned need_eq = sh_need;           // how many rows with key == T belong to the top-k
    const unsigned n_above = keff - need_eq;
    if (tid == 0) sh_above = 0;
    __syncthreads();

    // collection in row order: all keys > T, and the first need_eq rows with key == T
    unsigned eq_done = 0;
    for (int64_t base = 0; base < n_rows; base += (int64_t)XS_THREADS * XS_R) {
        const int64_t r0 = base + (int64_t)tid * XS_R;
        unsigned eqm = 0, mycnt = 0;
        uint16_t v[XS_R];
#pragma unroll
        for (int j = 0; j < XS_R; ++j) {
            const int64_t r = r0 + j;
            v[j] = r < n_rows ? hk[r] : 0;
            if (r < n_rows && v[j] > T) { const unsigned s = atomicAdd(&sh_above, 1u); okey[s] = ((uint64_t)v[j] << 32) | (uint64_t)(0xffffffffu - (uint32_t)r); }
            if (r < n_rows && v[j] == T) { eqm |= 1u << j; ++mycnt; }
        }
        // block exclusive scan of mycnt in thread (= row) order
        unsigned incl = mycnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o); if (lane >= o) incl += t; }
        if (lane == 63) sh_wsum[wave] = incl;
        __syncthreads();
        unsigned woff = 0, total = 0;
        for (int w = 0; w < XS_THREADS / 64; ++w) { const unsigned s = sh_wsum[w]; if (w < wave) woff += s; total += s; }
        unsigned rank = eq_done + woff + incl - mycnt;
#pragma unroll
        for (int j = 0; j < XS_R; ++j)
            if (eqm & (1u << j)) { if (rank < need_eq) okey[n_above + rank] = ((uint64_t)T << 32) | (uint64_t)(0xffffffffu - (uint32_t)(r0 + j)); ++rank; }
        eq_done += total;
        __syncthreads();
    }
    __syncthreads();
    const int qi = sel[f];
    if ((unsigned)tid < keff) {
        const uint64_t mine = okey[tid];
        unsigned rank = 0;
        for (unsigned i = 0; i < keff; ++i) rank += (okey[i] > mine);
        const uint32_t row = 0xffffffffu - (uint32_t)mine;
        const size_t o = (size_t)qi * k + rank;
        out_dist[o] = h16_unorder((uint32_t)(mine >> 32));
        out_row[o] = row_base + (int64_t)row;
        out_idx[o] = ids ? ids[row] : (int32_t)(row_base + (int64_t)row);
    }
    for (unsigned j = keff + tid; j < (unsigned)k; j += XS_THREADS) {
        const size_t o = (size_t)qi * k + j;
        out_dist[o] = 0xfc00; out_row[o] = -1; out_idx[o] = -1;
    }
    if (tid == 0) flags[qi] = 0;
}

int mips_launch_exact_select(const uint16_t *hkeys, int64_t n_rows, int64_t row_base, const int32_t *sel, int n_sel,
                             int k, const int32_t *ids, uint16_t *out_dist, int32_t *out_idx, int64_t *out_row,
                             unsigned *flags, hipStream_t stream)
{
    hipLaunchKernelGGL(exact_select_kernel, dim3(n_sel), dim3(XS_THREADS), 0, stream, hkeys, n_rows, row_base, sel, k, ids, out_dist,
                       out_idx, out_row, flags);
    return CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------
// all-exact fallback, fp32-score mode: 32-bit ordered keys, four-level radix threshold, same ordered collection
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) exact_scores_f32_kernel(const char *__restrict__ e_tiled, int64_t n_rows, int dim,
                                                               const uint16_t *__restrict__ queries, const int32_t *__restrict__ sel, int n_sel,
                                                               uint32_t *__restrict__ keys)
{
    extern __shared__ int qfix[]; // [n_sel][dim] packed (mant << 8) | shift
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < n_sel * dim; i += 256) {
        const int f = i / dim, d = i - f * dim;
        const HalfFix hf = half_fix(queries[(size_t)sel[f] * dim + d]);
        qfix[i] = (hf.mant << 8) | hf.shift;
    }
    __syncthreads();
    const int nseg = dim / 8;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < n_rows; row += (int64_t)gridDim.x * 4) {
        int64_t lo[XQ], hi[XQ];
#pragma unroll
        for (int f = 0; f < XQ; ++f) { lo[f] = 0; hi[f] = 0; }
        for (int seg = lane; seg < nseg; seg += 64) {
            const uint4 ev = *(const uint4 *)(e_tiled + tiled_seg_offset(row, seg, nseg >> 2));
            const uint32_t ew[4] = {ev.x, ev.y, ev.z, ev.w};
            HalfFix ef[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) { ef[2 * j] = half_fix((uint16_t)(ew[j] & 0xffff)); ef[2 * j + 1] = half_fix((uint16_t)(ew[j] >> 16)); }
#pragma unroll
            for (int f = 0; f < XQ; ++f) {
                if (f < n_sel) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int pq = qfix[f * dim + seg * 8 + j];
                        HalfFix qf; qf.mant = pq >> 8; qf.shift = pq & 0xff;
                        exact_mac(lo[f], hi[f], ef[j], qf);
                    }
                }
            }
        }
#pragma unroll
        for (int f = 0; f < XQ; ++f) {
            if (f < n_sel) {
                const int64_t l = wave_sum_i64(lo[f]), h = wave_sum_i64(hi[f]);
                if (lane == 0) keys[(size_t)f * n_rows + row] = f32_order(fixed_to_float(l, h));
            }
        }
    }
}

int mips_launch_exact_scores_f32(const char *e_tiled, int64_t n_rows, int dim, const uint16_t *queries, const int32_t *sel, int n_sel,
                                 uint32_t *keys, hipStream_t stream)
{
    if (n_sel > XQ) return -1;
    int64_t nb = (n_rows + 3) / 4;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(exact_scores_f32_kernel, dim3((unsigned)nb), dim3(256), (size_t)n_sel * dim * sizeof(int), stream, e_tiled, n_rows, dim,
                       queries, sel, n_sel, keys);
    return CHECK_LAUNCH();
}

__global__ void __launch_bounds__(XS_THREADS) exact_select_f32_kernel(const uint32_t *__restrict__ keys, int64_t n_rows, int64_t row_base,
                                                                      const int32_t *__restrict__ sel, int k, const int32_t *__restrict__ ids,
                                                                      float *out_dist, int32_t *out_idx, int64_t *out_row, unsigned *flags)
{
    __shared__ unsigned hist[256];
    __shared__ unsigned sh_digit, sh_need, sh_above, sh_wsum[XS_THREADS / 64];
    __shared__ uint64_t okey[128];
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t *hk = keys + (size_t)f * n_rows;
    const unsigned keff = (int64_t)k < n_rows ? (unsigned)k : (unsigned)n_rows;

    // radix threshold, most significant byte first: after level L the k-th key is known to start with `prefix`
    uint32_t prefix = 0;
    unsigned need = keff;
    for (int level = 3; level >= 0; --level) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const int sh = level * 8;
        for (int64_t i = tid; i < n_rows; i += XS_THREADS) {
            const uint32_t v = hk[i];
            if (level == 3 || (v >> (sh + 8)) == (prefix >> (sh + 8))) atomicAdd(&hist[(v >> sh) & 255u], 1u);
        }
        __syncthreads();
        if (tid < 64) pick_digit(hist, need, &sh_digit, &sh_need);
        __syncthreads();
        prefix |= sh_digit << sh;
        need = sh_need;
        __syncthreads();
    }
    const uint32_t T = prefix;                  // k-th key
    const unsigned need_eq = need;              // how many rows with key == T belong to the top-k
    const unsigned n_above = keff - need_eq;
    if (tid == 0) sh_above = 0;
    __syncthreads();

    unsigned eq_done = 0;
    for (int64_t base = 0; base < n_rows; base += (int64_t)XS_THREADS * XS_R) {
        const int64_t r0 = base + (int64_t)tid * XS_R;
        unsigned eqm = 0, mycnt = 0;
#pragma unroll
        for (int j = 0; j < XS_R; ++j) {
            const int64_t r = r0 + j;
            const uint32_t v = r < n_rows ? hk[r] : 0;
            if (r < n_rows && v > T) { const unsigned s = atomicAdd(&sh_above, 1u); okey[s] = ((uint64_t)v << 32) | (uint64_t)(0xffffffffu - (uint32_t)r); }
            if (r < n_rows && v == T) { eqm |= 1u << j; ++mycnt; }
        }
        unsigned incl = mycnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o); if (lane >= o) incl += t; }
        if (lane == 63) sh_wsum[wave] = incl;
        __syncthreads();
        unsigned woff = 0, total = 0;
        for (int w = 0; w < XS_THREADS / 64; ++w) { const unsigned s = sh_wsum[w]; if (w < wave) woff += s; total += s; }
        unsigned rank = eq_done + woff + incl - mycnt;
#pragma unroll
        for (int j = 0; j < XS_R; ++j)
            if (eqm & (1u << j)) { if (rank < need_eq) okey[n_above + rank] = ((uint64_t)T << 32) | (uint64_t)(0xffffffffu - (uint32_t)(r0 + j)); ++rank; }
        eq_done += total;
        __syncthreads();
    }
    __syncthreads();
    const int qi = sel[f];
    if ((unsigned)tid < keff) {
        const uint64_t mine = okey[tid];
        unsigned rank = 0;
        for (unsigned i = 0; i < keff; ++i) rank += (okey[i] > mine);
        const uint32_t row = 0xffffffffu - (uint32_t)mine;
        const size_t o = (size_t)qi * k + rank;
        out_dist[o] = f32_unorder((uint32_t)(mine >> 32));
        out_row[o] = row_base + (int64_t)row;
        out_idx[o] = ids ? ids[row] : (int32_t)(row_base + (int64_t)row);
    }
    for (unsigned j = keff + tid; j < (unsigned)k; j += XS_THREADS) {
        const size_t o = (size_t)qi * k + j;
        out_dist[o] = -INFINITY; out_row[o] = -1; out_idx[o] = -1;
    }
    if (tid == 0) flags[qi] = 0;
}

int mips_launch_exact_select_f32(const uint32_t *keys, int64_t n_rows, int64_t row_base, const int32_t *sel, int n_sel, int k,
                                 const int32_t *ids, float *out_dist, int32_t *out_idx, int64_t *out_row, unsigned *flags, hipStream_t stream)
{
    hipLaunchKernelGGL(exact_select_f32_kernel, dim3(n_sel), dim3(XS_THREADS), 0, stream, keys, n_rows, row_base, sel, k, ids, out_dist,
                       out_idx, out_row, flags);
    return CHECK_LAUNCH();
}
