// emdr2_amd/csrc/seqpack.hip -- packed ("varlen") sequence layout of the encoder stacks (include/emdr2_ops.h: emdr2_seq_*).
//
// The reference runs every encoder over [batch, S] token grids (transformer.py:283-381, emdr2_model.py:148-210); at the NQ shapes about
// half of those rows are [PAD] (a 100-word passage is ~140 of the context encoder's 256 positions, ~160 of the one-context reader's 512).
// A padded key is masked to -10000 and contributes exp(-10000 - max) == 0 exactly to every real query; a padded query's output is
// never consumed (the towers read token 0, the FiD decoder masks padded encoder positions, padded rows receive exactly zero gradient).
// So the rows can simply be left out: sequence i keeps its first len[i] tokens, len[i] = 1 + index of its LAST non-pad token (interior
// zeros stay in the sequence and stay masked by the id test; an all-pad row keeps all S positions, which reproduces the reference's
// uniform attention for it), and the sequences are stored back to back:
//     packed row t = cu[i] + pos   <->   dense row i * S + pos,  pos < len[i],  cu = exclusive prefix sum of len
// GEMMs, LayerNorm and the epilogues see a [T, H] matrix with T = cu[n] rounded up to a multiple of 256 (tail rows are zeros); the
// attention kernels take cu (csrc/attention.hip); the kernels here build the layout and move rows between the two forms.
#include "../../include/emdr2_ops.h"
#include <hip/hip_runtime.h>
#include <mutex>
#include <stdint.h>

namespace {

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? 0 : -3)

// Two launches.  seq_len_kernel: a wave per sequence (ballot over 64-token chunks), 16 sequences per workgroup, len[i] -> cu[i] (the
// 13 MB of ids of a 3,200 x 512 block go through the whole chip, not through one CU).  seq_lengths_kernel: ONE workgroup turns the lengths
// in cu[] into their exclusive scan in place, and emits sum(len^2) (the attention score count of the packed self-attention, for flop
// accounting) and the longest length (the attention grids are sized by it) next to the total.
__global__ void __launch_bounds__(1024) seq_len_kernel(const long long *ids, int n, int S, int *cu)
{
    const int lane = threadIdx.x & 63, i = blockIdx.x * 16 + (threadIdx.x >> 6);
    if (i >= n) return;
    int last = -1;
    for (int c = 0; c < S; c += 64) {
        const int pos = c + lane;
        const unsigned long long w = __builtin_amdgcn_ballot_w64(pos < S && ids[(long long)i * S + (pos < S ? pos : 0)] != 0);
        if (w) last = c + 63 - __builtin_clzll(w);
    }
    if (lane == 0) cu[i] = last < 0 ? S : last + 1;
}

__global__ void __launch_bounds__(1024) seq_lengths_kernel(int n, int *cu, long long *totals)
{
    extern __shared__ int len_s[];               // n ints
    __shared__ int part[16], max_part[16];
    __shared__ long long sq_part[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < n; i += 1024) len_s[i] = cu[i];
    __syncthreads();
    // scan: thread t owns a contiguous chunk of sequences
    const int per = (n + 1023) / 1024, lo = tid * per, hi = min(n, lo + per);
    int sum = 0, mx = 0;
    long long sq = 0;
    for (int i = lo; i < hi; ++i) { sum += len_s[i]; sq += (long long)len_s[i] * len_s[i]; mx = max(mx, len_s[i]); }
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { sq += __shfl_xor(sq, d, 64); mx = max(mx, __shfl_xor(mx, d, 64)); }
    if (lane == 63) part[wave] = incl;
    if (lane == 0) { sq_part[wave] = sq; max_part[wave] = mx; }
    __syncthreads();
    int base = incl - sum;
    for (int w = 0; w < wave; ++w) base += part[w];
    for (int i = lo; i < hi; ++i) { cu[i] = base; base += len_s[i]; }
    if (tid == 1023) {
        cu[n] = base;
        long long tot = 0;
        int longest = 0;
        for (int w = 0; w < 16; ++w) { tot += sq_part[w]; longest = max(longest, max_part[w]); }
        totals[0] = base; totals[1] = tot; totals[2] = longest;
    }
}

// One workgroup per sequence (+ extra groups for the tail rows): the two row maps and the packed ids / token types.
__global__ void __launch_bounds__(256) seq_pack_ids_kernel(const long long *ids, const long long *types, const int *cu, int n, int S, long long t_pad,
                                                           int *rowmap, int *inverse, long long *ids_p, long long *types_p)
{
    const int i = blockIdx.x;
    if (i < n) {
        const int c0 = cu[i], len = cu[i + 1] - c0;
        for (int pos = threadIdx.x; pos < S; pos += 256) {
            const long long d = (long long)i * S + pos;
            if (pos < len) {
                rowmap[c0 + pos] = (int)d;
                ids_p[c0 + pos] = ids[d];
                if (types_p) types_p[c0 + pos] = types[d];
                inverse[d] = c0 + pos;
            } else {
                inverse[d] = -1;
            }
        }
    } else {
        const long long t = (long long)cu[n] + (long long)(i - n) * 256 + threadIdx.x;
        if (t < t_pad) {
            rowmap[t] = -1; ids_p[t] = 0;
            if (types_p) types_p[t] = 0;
        }
    }
}

// out[r, :] = map[r] >= 0 ? in[map[r], :] : 0 over bf16 rows of H (H % 8 == 0): 16 B per lane
__global__ void __launch_bounds__(256) gather_rows_kernel(const uint4 *in, const int *map, uint4 *out, long long rows, int h16)
{
    const long long r = (long long)blockIdx.x * 2 + (threadIdx.x >> 7);
    if (r >= rows) return;
    const int src = map[r];
    for (int c = threadIdx.x & 127; c < h16; c += 128) out[r * h16 + c] = src >= 0 ? in[(long long)src * h16 + c] : make_uint4(0, 0, 0, 0);
}

// out[map[r], :] = in[r, :] for map[r] >= 0 (distinct targets; the rest of `out` is the caller's: usually zeros)
__global__ void __launch_bounds__(256) scatter_rows_kernel(const uint4 *in, const int *map, uint4 *out, long long rows, int h16)
{
    const long long r = (long long)blockIdx.x * 2 + (threadIdx.x >> 7);
    if (r >= rows) return;
    const int dst = map[r];
    if (dst < 0) return;
    for (int c = threadIdx.x & 127; c < h16; c += 128) out[(long long)dst * h16 + c] = in[r * h16 + c];
}

} // namespace

extern "C" int emdr2_seq_lengths(const int64_t *ids, int n, int S, int32_t *cu, int64_t *totals, void *stream)
{
    if (!ids || !cu || !totals || n < 1 || S < 1) return -1;
    if (n > 38000) return -4;                                                    // the lengths live in LDS: 152 of the 160 KB (B = 256 at top-k 100 is 25,600)
    if (n > 16000) {                                                             // past the 64 KB a kernel gets without asking
        // the attribute belongs to (function, device): one flag per device ordinal, set under a lock (callers may be threads of several devices);
        // a device whose opt-in LDS limit is below the request (not gfx950) is an unsupported shape, not a runtime error
        static std::mutex attr_lock;
        static bool attr_done[64] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -3;
        std::lock_guard<std::mutex> guard(attr_lock);
        if (!attr_done[dev]) {
            int optin = 0;
            if (hipDeviceGetAttribute(&optin, hipDeviceAttributeSharedMemPerBlockOptin, dev) != hipSuccess) optin = 0;
            if (optin > 0 && optin < 38000 * (int)sizeof(int)) return -4;
            if (hipFuncSetAttribute((const void *)seq_lengths_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 38000 * (int)sizeof(int)) != hipSuccess) {
                (void)hipGetLastError();
                return -4;
            }
            attr_done[dev] = true;
        }
    }
    hipLaunchKernelGGL(seq_len_kernel, dim3((unsigned)(n + 15) / 16), dim3(1024), 0, (hipStream_t)stream, (const long long *)ids, n, S, (int *)cu);
    hipLaunchKernelGGL(seq_lengths_kernel, dim3(1), dim3(1024), (size_t)n * sizeof(int), (hipStream_t)stream, n, (int *)cu, (long long *)totals);
    return LAUNCH_OK();
}

extern "C" int emdr2_seq_pack_ids(const int64_t *ids, const int64_t *types, const int32_t *cu, int n, int S, int64_t total, int64_t rows_padded,
                                  int32_t *rowmap, int32_t *inverse, int64_t *ids_packed, int64_t *types_packed, void *stream)
{
    if (!ids || !cu || !rowmap || !inverse || !ids_packed || n < 1 || S < 1 || total < 1 || rows_padded < total || (types && !types_packed)) return -1;
    const unsigned tail_groups = (unsigned)((rows_padded - total + 255) / 256);
    hipLaunchKernelGGL(seq_pack_ids_kernel, dim3((unsigned)n + tail_groups), dim3(256), 0, (hipStream_t)stream, (const long long *)ids, (const long long *)types,
                       (const int *)cu, n, S, (long long)rows_padded, (int *)rowmap, (int *)inverse, (long long *)ids_packed,
                       types ? (long long *)types_packed : (long long *)nullptr);
    return LAUNCH_OK();
}

extern "C" int emdr2_gather_rows(const void *in, const int32_t *map, void *out, int64_t rows_out, int H, void *stream)
{
    if (!in || !map || !out || rows_out < 1 || H < 8 || (H & 7) || ((uintptr_t)in & 15) || ((uintptr_t)out & 15)) return -1;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((rows_out + 1) / 2)), dim3(256), 0, (hipStream_t)stream, (const uint4 *)in, (const int *)map,
                       (uint4 *)out, (long long)rows_out, H / 8);
    return LAUNCH_OK();
}

extern "C" int emdr2_scatter_rows(const void *in, const int32_t *map, void *out, int64_t rows_in, int H, void *stream)
{
    if (!in || !map || !out || rows_in < 1 || H < 8 || (H & 7) || ((uintptr_t)in & 15) || ((uintptr_t)out & 15)) return -1;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)((rows_in + 1) / 2)), dim3(256), 0, (hipStream_t)stream, (const uint4 *)in, (const int *)map,
                       (uint4 *)out, (long long)rows_in, H / 8);
    return LAUNCH_OK();
}
