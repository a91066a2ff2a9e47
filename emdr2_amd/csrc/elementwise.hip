// emdr2_amd/csrc/elementwise.hip -- HBM-bound kernels around the GEMMs (include/emdr2_ops.h).
#include "../../include/emdr2_ops.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rng.h"

namespace {

__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
typedef uint32_t ew_u32x4 __attribute__((ext_vector_type(4)));
// row streams that are touched once per launch (LayerNorm inputs / outputs: 2 GB each at the reader's shape): keep them out of the way of
// what the neighbouring GEMMs want to find in L2
__device__ __forceinline__ uint4 ld_stream(const uint16_t *p)
{
    const ew_u32x4 t = __builtin_nontemporal_load((const ew_u32x4 *)p);
    return make_uint4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void st_stream(uint16_t *p, uint4 v)
{
    const ew_u32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, (ew_u32x4 *)p);
}
__device__ __forceinline__ uint16_t f2bf(float f)
{
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// 64 x 64 tiles through LDS (+1 padding), 256 threads: coalesced 128-B rows in, 128-B rows out
__global__ void __launch_bounds__(256) transpose_kernel(const uint16_t *in, long long ld_in, uint16_t *out, long long ld_out, int rows,
                                                        int cols, int batch2, long long sI1, long long sO1, long long sI2, long long sO2,
                                                        float *colsum)
{
    __shared__ uint16_t tile[64][66];
    const int b1 = blockIdx.z / batch2, b2 = blockIdx.z % batch2;
    in += b1 * sI1 + b2 * sI2;
    out += b1 * sO1 + b2 * sO2;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    float cs = 0.f;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        uint16_t v = 0;
        if (r < rows && c < cols) v = in[(long long)r * ld_in + c];
        tile[i][tx] = v;
        cs += bf2f(v);
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows) out[(long long)c * ld_out + r] = tile[tx][i];
    }
    if (colsum) {
        __shared__ float part[4][64];
        part[ty][tx] = cs;
        __syncthreads();
        if (ty == 0 && c0 + tx < cols) atomicAdd(&colsum[c0 + tx], part[0][tx] + part[1][tx] + part[2][tx] + part[3][tx]);
    }
}


__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// ---- LayerNorm (torch.nn.LayerNorm semantics, eps inside the sqrt; mpu/layers.py:28-36) -----------------------
// one wave per row, H % 8 == 0, 16-B vector loads
__global__ void __launch_bounds__(256) layernorm_fwd_kernel(const uint16_t *x, const float *gamma, const float *beta, uint16_t *y,
                                                            float *mean, float *rstd, long long rows, int H, float eps)
{
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const uint16_t *xr = x + row * H;
    float s = 0.f, ss = 0.f;
    for (int i = lane * 8; i < H; i += 512) {
        const uint4 v = *(const uint4 *)(xr + i);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float a = bf2f((uint16_t)(w[j] & 0xffff)), b = bf2f((uint16_t)(w[j] >> 16)); s += a + b; ss += a * a + b * b; }
    }
    s = wave_sum(s); ss = wave_sum(ss);
    const float mu = s / H;
    const float var = fmaxf(ss / H - mu * mu, 0.f);
    const float rs = rsqrtf(var + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    uint16_t *yr = y + row * H;
    for (int i = lane * 8; i < H; i += 512) {
        const uint4 v = *(const uint4 *)(xr + i);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = (bf2f((uint16_t)(w[j] & 0xffff)) - mu) * rs * gamma[i + 2 * j] + beta[i + 2 * j];
            const float b = (bf2f((uint16_t)(w[j] >> 16)) - mu) * rs * gamma[i + 2 * j + 1] + beta[i + 2 * j + 1];
            o[j] = (uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16);
        }
        *(uint4 *)(yr + i) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// H == 768 (every LayerNorm of the base configuration): half a wave per row, the row's 96 16-byte granules held in registers (3 per
// lane, all lanes busy, one HBM read), reductions over 32 lanes, gamma / beta fetched once per lane for both rows of the wave.
__global__ void __launch_bounds__(256) layernorm_fwd768_kernel(const uint16_t *x, const float *gamma, const float *beta, uint16_t *y, float *mean,
                                                               float *rstd, long long rows, float eps)
{
    const int lane = threadIdx.x & 63, l31 = lane & 31;
    const long long row = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + (lane >> 5);
    const bool live = row < rows;
    const uint16_t *xr = x + (live ? row : rows - 1) * 768;
    uint4 v[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) v[i] = ld_stream(xr + (i * 32 + l31) * 8);
    float f[3][8], s = 0.f, ss = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const uint32_t w[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f[i][2 * j] = bf2f((uint16_t)(w[j] & 0xffff)); f[i][2 * j + 1] = bf2f((uint16_t)(w[j] >> 16));
            s += f[i][2 * j] + f[i][2 * j + 1];
            ss += f[i][2 * j] * f[i][2 * j] + f[i][2 * j + 1] * f[i][2 * j + 1];
        }
    }
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) { s += __shfl_xor(s, o); ss += __shfl_xor(ss, o); }   // within the 32 lanes of the row
    const float mu = s * (1.0f / 768.0f);
    const float var = fmaxf(ss * (1.0f / 768.0f) - mu * mu, 0.f);
    const float rs = rsqrtf(var + eps);
    if (live && l31 == 0) { mean[row] = mu; rstd[row] = rs; }
    if (!live) return;
    uint16_t *yr = y + row * 768;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = (i * 32 + l31) * 8;
        const float4 g0 = *(const float4 *)(gamma + c), g1 = *(const float4 *)(gamma + c + 4);
        const float4 b0 = *(const float4 *)(beta + c), b1 = *(const float4 *)(beta + c + 4);
        const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bt[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            o[j] = (uint32_t)f2bf(fmaf((f[i][2 * j] - mu) * rs, gm[2 * j], bt[2 * j])) |
                   ((uint32_t)f2bf(fmaf((f[i][2 * j + 1] - mu) * rs, gm[2 * j + 1], bt[2 * j + 1])) << 16);
        st_stream(yr + c, make_uint4(o[0], o[1], o[2], o[3]));
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy*gamma ; optional residual-branch gradient added on the way out.
// Block = 32 rows: phase 1 one wave per row computes the two row means into LDS; phase 2 every thread owns columns tid, tid+256, ...
// for all 32 rows (dgamma / dbeta partials stay in registers; one global atomicAdd per column per block).
#define LN_ROWS 32
__global__ void __launch_bounds__(256) layernorm_bwd_kernel(const uint16_t *dy, const uint16_t *x, const float *gamma, const float *mean,
                                                            const float *rstd, const uint16_t *dres, uint16_t *dx, float *dgamma,
                                                            float *dbeta, long long rows, int H, int rows_per_block)
{
    __shared__ float s1s[LN_ROWS], s2s[LN_ROWS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long r0 = (long long)blockIdx.x * LN_ROWS;
    for (int rr = wave; rr < LN_ROWS; rr += 4) {
        const long long row = r0 + rr;
        float s1 = 0.f, s2 = 0.f;
        if (row < rows) {
            const uint16_t *xr = x + row * H, *gr = dy + row * H;
            const float mu = mean[row], rs = rstd[row];
            for (int i = lane; i < H; i += 64) {
                const float xh = (bf2f(xr[i]) - mu) * rs, g = bf2f(gr[i]) * gamma[i];
                s1 += g; s2 += g * xh;
            }
        }
        s1 = wave_sum(s1); s2 = wave_sum(s2);
        if (lane == 0) { s1s[rr] = s1 / H; s2s[rr] = s2 / H; }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < H; c += 256) {
        const float gm = gamma[c];
        float ag = 0.f, ab = 0.f;
        for (int rr = 0; rr < LN_ROWS; ++rr) {
            const long long row = r0 + rr;
            if (row >= rows) break;
            const float mu = mean[row], rs = rstd[row];
            const float xh = (bf2f(x[row * H + c]) - mu) * rs, d = bf2f(dy[row * H + c]);
            float v = rs * (d * gm - s1s[rr] - xh * s2s[rr]);
            if (dres) v += bf2f(dres[row * H + c]);
            dx[row * H + c] = f2bf(v);
            ag += d * xh; ab += d;
        }
        atomicAdd(&dgamma[c], ag);
        atomicAdd(&dbeta[c], ab);
    }
}

// H == 768 backward: half a wave per row with dy and x in registers (one HBM read each), persistent waves accumulate the dgamma / dbeta
// partials of their 24 columns per lane in registers across all their rows; one LDS reduction and one atomicAdd per column per workgroup.
// DMASK: also write dx o keep(seed, row, col) / (1 - p) -- the bias-dropout-add below this LayerNorm wants exactly that tensor as the operand of
// its two backward GEMMs (they read operands by LDS-DMA, so the mask cannot ride in their loads), and r03 made it with a kernel of its own
// (read dx, write the masked copy: 21 ms of the step).  Same bits as that kernel: the mask is applied to the bf16-ROUNDED dx.
template <bool DMASK>
__global__ void __launch_bounds__(256) layernorm_bwd768_kernel(const uint16_t *dy, const uint16_t *x, const float *gamma, const float *mean,
                                                               const float *rstd, const uint16_t *dres, uint16_t *dx, float *dgamma, float *dbeta,
                                                               long long rows, uint16_t *dmask, float drop_p, uint32_t seed)
{
    __shared__ float red[2][4][768];
    const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5, wave = threadIdx.x >> 6;
    float gm[3][8], ag[3][8], ab[3][8];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = (i * 32 + l31) * 8;
        const float4 g0 = *(const float4 *)(gamma + c), g1 = *(const float4 *)(gamma + c + 4);
        gm[i][0] = g0.x; gm[i][1] = g0.y; gm[i][2] = g0.z; gm[i][3] = g0.w; gm[i][4] = g1.x; gm[i][5] = g1.y; gm[i][6] = g1.z; gm[i][7] = g1.w;
#pragma unroll
        for (int j = 0; j < 8; ++j) { ag[i][j] = 0.f; ab[i][j] = 0.f; }
    }
    const long long npairs = (rows + 1) >> 1, stride = (long long)gridDim.x * 4;
    for (long long rp = (long long)blockIdx.x * 4 + wave; rp < npairs; rp += stride) {
        const long long row = rp * 2 + half;
        if (row >= rows) continue;                                       // only the odd tail row's partner half-wave
        const uint16_t *xr = x + row * 768, *dr = dy + row * 768;
        uint4 xv[3], dv[3], rv[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) { xv[i] = ld_stream(xr + (i * 32 + l31) * 8); dv[i] = ld_stream(dr + (i * 32 + l31) * 8); }
        if (dres) {
#pragma unroll
            for (int i = 0; i < 3; ++i) rv[i] = ld_stream(dres + row * 768 + (i * 32 + l31) * 8);
        }
        const float mu = mean[row], rs = rstd[row];
        float xh[3][8], g[3][8], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const uint32_t xw[4] = {xv[i].x, xv[i].y, xv[i].z, xv[i].w}, dw[4] = {dv[i].x, dv[i].y, dv[i].z, dv[i].w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xe = bf2f((uint16_t)((j & 1) ? xw[j >> 1] >> 16 : xw[j >> 1] & 0xffff));
                const float de = bf2f((uint16_t)((j & 1) ? dw[j >> 1] >> 16 : dw[j >> 1] & 0xffff));
                xh[i][j] = (xe - mu) * rs;
                g[i][j] = de * gm[i][j];
                s1 += g[i][j]; s2 += g[i][j] * xh[i][j];
                ag[i][j] += de * xh[i][j]; ab[i][j] += de;
            }
        }
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
        const float m1 = s1 * (1.0f / 768.0f), m2 = s2 * (1.0f / 768.0f);
        uint16_t *xo = dx + row * 768;
        const float ik = DMASK ? emdr2_keep_scale(drop_p) : 1.f;
        const uint32_t thr = DMASK ? emdr2_drop_thr(drop_p) : 0u, rh = DMASK ? emdr2_row_hash(seed, (unsigned long long)row) : 0u;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const uint32_t rw[4] = {rv[i].x, rv[i].y, rv[i].z, rv[i].w};
            uint32_t o[4], om[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a = rs * (g[i][2 * j] - m1 - xh[i][2 * j] * m2), b = rs * (g[i][2 * j + 1] - m1 - xh[i][2 * j + 1] * m2);
                if (dres) { a += bf2f((uint16_t)(rw[j] & 0xffff)); b += bf2f((uint16_t)(rw[j] >> 16)); }
                const uint16_t ha = f2bf(a), hb = f2bf(b);
                o[j] = (uint32_t)ha | ((uint32_t)hb << 16);
                if (DMASK) {
                    const uint32_t bits = emdr2_pair_bits(rh, (uint32_t)((i * 32 + l31) * 8 + 2 * j));
                    const float lo = (bits & 0xffffu) >= thr ? bf2f(ha) * ik : 0.f, hi = (bits >> 16) >= thr ? bf2f(hb) * ik : 0.f;
                    om[j] = (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
                }
            }
            st_stream(xo + (i * 32 + l31) * 8, make_uint4(o[0], o[1], o[2], o[3]));
            if (DMASK) st_stream(dmask + row * 768 + (i * 32 + l31) * 8, make_uint4(om[0], om[1], om[2], om[3]));
        }
    }
    // both half-waves own the same columns: fold, then reduce the 4 waves through LDS
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float a = ag[i][j] + __shfl_xor(ag[i][j], 32), b = ab[i][j] + __shfl_xor(ab[i][j], 32);
            if (half == 0) { red[0][wave][(i * 32 + l31) * 8 + j] = a; red[1][wave][(i * 32 + l31) * 8 + j] = b; }
        }
    __syncthreads();
    for (int c = threadIdx.x; c < 768; c += 256) {
        atomicAdd(&dgamma[c], red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
        atomicAdd(&dbeta[c], red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
    }
}

// ---- masked softmax over the last dim (reference fallback path: fused_softmax.py:113-125, mask value -10000 REPLACES the score) --
// masked(b, q, k) = ids_q[b,q] == 0 || ids_k[b,k] == 0 || (causal && k > q)     (mask_creation_utils.py:17-42, pad id 0)
// scores [B, np, sq, sk] bf16 in place -> probs; stats m, l fp32 [B, np, sq] (row max and sum of exp), one wave per row
__global__ void __launch_bounds__(256) softmax_fwd_kernel(uint16_t *s, const long long *ids_q, const long long *ids_k, int np, int sq, int sk,
                                                          int causal, float *mstat, float *lstat, long long rows, float drop_p, uint32_t seed)
{
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int q = (int)(row % sq);
    const long long b = row / ((long long)sq * np);
    const bool qpad = ids_q[b * sq + q] == 0;
    uint16_t *sr = s + row * sk;
    const long long *kid = ids_k + b * sk;
    float m = -3.0e38f;
    for (int k = lane; k < sk; k += 64) {
        const bool masked = qpad || kid[k] == 0 || (causal && k > q);
        m = fmaxf(m, masked ? -10000.f : bf2f(sr[k]));
    }
    m = wave_max(m);
    float l = 0.f;
    for (int k = lane; k < sk; k += 64) {
        const bool masked = qpad || kid[k] == 0 || (causal && k > q);
        l += __expf((masked ? -10000.f : bf2f(sr[k])) - m);
    }
    l = wave_sum(l);
    const float inv = 1.f / l;
    for (int k = lane; k < sk; k += 64) {
        const bool masked = qpad || kid[k] == 0 || (causal && k > q);
        float pv = __expf((masked ? -10000.f : bf2f(sr[k])) - m) * inv;
        if (drop_p > 0.f) pv = emdr2_keep(emdr2_row_hash(seed, (unsigned long long)row), (uint32_t)k, emdr2_drop_thr(drop_p)) ? pv * emdr2_keep_scale(drop_p) : 0.f;
        sr[k] = f2bf(pv);
    }
    if (lane == 0 && mstat) { mstat[row] = m; lstat[row] = l; }
}

// register-resident variants (sk % 8 == 0, sk <= 2048): one HBM read + one write per element, 16-byte accesses
template <int NV>
__global__ void __launch_bounds__(256) softmax_fwd_reg_kernel(uint16_t *s, const long long *ids_q, const long long *ids_k, int np, int sq, int sk,
                                                              int causal, float *mstat, float *lstat, long long rows, float drop_p, uint32_t seed)
{
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int q = (int)(row % sq);
    const long long b = row / ((long long)sq * np);
    const bool qpad = ids_q[b * sq + q] == 0;
    uint16_t *sr = s + row * sk;
    const long long *kid = ids_k + b * sk;
    float v[NV][8];
    float m = -3.0e38f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int k0 = (i * 64 + lane) * 8;
        if (k0 < sk) {
            const uint4 x = *(const uint4 *)(sr + k0);
            const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + j;
                const bool masked = qpad || kid[k] == 0 || (causal && k > q);
                v[i][j] = masked ? -10000.f : bf2f((uint16_t)((j & 1) ? w[j >> 1] >> 16 : w[j >> 1] & 0xffff));
                m = fmaxf(m, v[i][j]);
            }
        }
    }
    m = wave_max(m);
    float l = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if ((i * 64 + lane) * 8 < sk) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[i][j] = __expf(v[i][j] - m); l += v[i][j]; }
        }
    l = wave_sum(l);
    const float inv = 1.f / l;
    const float ik = drop_p > 0.f ? emdr2_keep_scale(drop_p) : 1.f;
    const uint32_t rh = emdr2_row_hash(seed, (unsigned long long)row), thr = emdr2_drop_thr(drop_p);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int k0 = (i * 64 + lane) * 8;
        if (k0 < sk) {
            uint32_t w[4];
            if (drop_p > 0.f) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[i][j] = emdr2_keep(rh, (uint32_t)(k0 + j), thr) ? v[i][j] * ik : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = (uint32_t)f2bf(v[i][2 * j] * inv) | ((uint32_t)f2bf(v[i][2 * j + 1] * inv) << 16);
            *(uint4 *)(sr + k0) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    if (lane == 0 && mstat) { mstat[row] = m; lstat[row] = l; }
}

template <int NV>
__global__ void __launch_bounds__(256) softmax_bwd_reg_kernel(const uint16_t *p, uint16_t *dp, const long long *ids_q, const long long *ids_k, int np,
                                                              int sq, int sk, int causal, float *dstat, long long rows, const float *mstat,
                                                              const float *lstat, float drop_p, uint32_t seed)
{
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int q = (int)(row % sq);
    const long long b = row / ((long long)sq * np);
    const bool qpad = ids_q[b * sq + q] == 0;
    const long long *kid = ids_k + b * sk;
    const uint16_t *pr = p + row * sk;
    uint16_t *dr = dp + row * sk;
    float pv[NV][8], gv[NV][8];
    float d = 0.f;
    const float ms = mstat ? mstat[row] : 0.f, il = mstat ? 1.f / lstat[row] : 1.f;
    const float ik = drop_p > 0.f ? emdr2_keep_scale(drop_p) : 1.f;
    const uint32_t rh = emdr2_row_hash(seed, (unsigned long long)row), thr = emdr2_drop_thr(drop_p);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int k0 = (i * 64 + lane) * 8;
        if (k0 < sk) {
            const uint4 a = *(const uint4 *)(pr + k0), g = *(const uint4 *)(dr + k0);
            const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, gw[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                pv[i][j] = bf2f((uint16_t)((j & 1) ? aw[j >> 1] >> 16 : aw[j >> 1] & 0xffff));
                gv[i][j] = bf2f((uint16_t)((j & 1) ? gw[j >> 1] >> 16 : gw[j >> 1] & 0xffff));
                if (mstat) {                                             // p holds raw scaled scores: rebuild the probability from the row statistics
                    const int k = k0 + j;
                    const bool masked = qpad || kid[k] == 0 || (causal && k > q);
                    pv[i][j] = __expf((masked ? -10000.f : pv[i][j]) - ms) * il;
                }
                if (drop_p > 0.f) gv[i][j] = emdr2_keep(rh, (uint32_t)(k0 + j), thr) ? gv[i][j] * ik : 0.f;
                d += pv[i][j] * gv[i][j];
            }
        }
    }
    d = wave_sum(d);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int k0 = (i * 64 + lane) * 8;
        if (k0 < sk) {
            uint32_t w[4];
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + j;
                const bool masked = qpad || kid[k] == 0 || (causal && k > q);
                o[j] = masked ? 0.f : pv[i][j] * (gv[i][j] - d);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = (uint32_t)f2bf(o[2 * j]) | ((uint32_t)f2bf(o[2 * j + 1]) << 16);
            *(uint4 *)(dr + k0) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    if (lane == 0 && dstat) dstat[row] = d;
}

// dS = P * (dP - sum_k P*dP), in place on dP; masked positions get the same formula (their P is ~0 unless the row is fully
// masked, where the reference's autograd also flows a uniform-softmax gradient into the replaced (constant) scores: those
// are zeroed because masked_fill breaks the dependence on the score).  D[row] is also written.
__global__ void __launch_bounds__(256) softmax_bwd_kernel(const uint16_t *p, uint16_t *dp, const long long *ids_q, const long long *ids_k,
                                                          int np, int sq, int sk, int causal, float *dstat, long long rows, const float *mstat,
                                                          const float *lstat, float drop_p, uint32_t seed)
{
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int q = (int)(row % sq);
    const long long b = row / ((long long)sq * np);
    const bool qpad = ids_q[b * sq + q] == 0;
    const long long *kid = ids_k + b * sk;
    const uint16_t *pr = p + row * sk;
    uint16_t *dr = dp + row * sk;
    float d = 0.f;
    const float ms = mstat ? mstat[row] : 0.f, il = mstat ? 1.f / lstat[row] : 1.f;
    const float ik = drop_p > 0.f ? emdr2_keep_scale(drop_p) : 1.f;
    const uint32_t rh = emdr2_row_hash(seed, (unsigned long long)row), thr = emdr2_drop_thr(drop_p);
    auto prob = [&](int k, bool masked) { return mstat ? __expf((masked ? -10000.f : bf2f(pr[k])) - ms) * il : bf2f(pr[k]); };
    auto grad = [&](int k) { const float g = bf2f(dr[k]); return drop_p > 0.f ? (emdr2_keep(rh, (uint32_t)k, thr) ? g * ik : 0.f) : g; };
    for (int k = lane; k < sk; k += 64) d += prob(k, qpad || kid[k] == 0 || (causal && k > q)) * grad(k);
    d = wave_sum(d);
    for (int k = lane; k < sk; k += 64) {
        const bool masked = qpad || kid[k] == 0 || (causal && k > q);
        dr[k] = masked ? (uint16_t)0 : f2bf(prob(k, masked) * (grad(k) - d));
    }
    if (lane == 0 && dstat) dstat[row] = d;
}

// transposed twin used by the attention backward: from raw scaled scores S^T [B, np, sk, sq] and dP^T, with the forward's
// row statistics m, l and D (all indexed [B, np, sq]): P^T = exp(s - m[q]) / l[q] (masked: exp(-10000 - m[q]) / l[q]),
// dS^T = masked ? 0 : P^T * (dP^T - D[q]).  Writes P^T over S^T and dS^T over dP^T.  Elementwise, 8 elements per thread.
__global__ void __launch_bounds__(256) softmax_t_kernel(uint16_t *st, uint16_t *dpt, const long long *ids_q, const long long *ids_k,
                                                        const float *mstat, const float *lstat, const float *dstat, int np, int sq,
                                                        int sk, int causal, long long total, float drop_p, uint32_t seed)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int q = (int)(i % sq);
    const long long rk = i / sq;             // (b, n, k)
    const int k = (int)(rk % sk);
    const long long bn = rk / sk;
    const long long b = bn / np;
    const long long srow = bn * sq + q;      // stats index
    const bool masked = ids_q[b * sq + q] == 0 || ids_k[b * sk + k] == 0 || (causal && k > q);
    const float pt = __expf((masked ? -10000.f : bf2f(st[i])) - mstat[srow]) / lstat[srow];
    float g = bf2f(dpt[i]), pd = pt;
    if (drop_p > 0.f) {
        const bool keep = emdr2_keep(emdr2_row_hash(seed, (unsigned long long)srow), (uint32_t)k, emdr2_drop_thr(drop_p));
        const float ik = emdr2_keep_scale(drop_p);
        g = keep ? g * ik : 0.f;
        pd = keep ? pt * ik : 0.f;                                        // dropped probabilities (what multiplied V in the forward) for dV
    }
    st[i] = f2bf(pd);
    dpt[i] = masked ? (uint16_t)0 : f2bf(pt * (g - dstat[srow]));
}

// ---- GELU backward on the saved pre-activation (exact erf form; transformer.py:80,103-104) -------------------------------
__global__ void __launch_bounds__(256) gelu_bwd_kernel(const uint16_t *pre, const uint16_t *dact, uint16_t *dpre, long long n)
{
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i >= n) return;
    const uint4 a = *(const uint4 *)(pre + i), g = *(const uint4 *)(dact + i);
    const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, gw[4] = {g.x, g.y, g.z, g.w};
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float r[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float u = bf2f((uint16_t)(h ? aw[j] >> 16 : aw[j] & 0xffff)), d = bf2f((uint16_t)(h ? gw[j] >> 16 : gw[j] & 0xffff));
            const float cdf = 0.5f * (1.f + erff(u * 0.70710678118654752f));
            r[h] = d * (cdf + u * 0.3989422804014327f * __expf(-0.5f * u * u));
        }
        o[j] = (uint32_t)f2bf(r[0]) | ((uint32_t)f2bf(r[1]) << 16);
    }
    *(uint4 *)(dpre + i) = make_uint4(o[0], o[1], o[2], o[3]);
}

// ---- embedding (language_model.py:169-181): out = W[ids] + P[pos] (+ T[types]) ------------------------------------------------
__global__ void __launch_bounds__(256) embedding_fwd_kernel(const long long *ids, const long long *types, const uint16_t *W, const uint16_t *P,
                                                            const uint16_t *T, uint16_t *out, long long tokens, int S, int H, float drop_p, uint32_t seed,
                                                            const int *rowmap)
{
    // rowmap (packed layout, csrc/seqpack.hip): row t holds the token of dense row rowmap[t] = sequence * S + position; -1 = a tail row (zeros)
    const long long t = blockIdx.x;
    long long pos = t % S;
    if (rowmap) {
        const int src = rowmap[t];
        if (src < 0) {
            for (int i = threadIdx.x; i < H; i += 256) out[t * H + i] = 0;
            return;
        }
        pos = src % S;
    }
    const long long id = ids[t];
    const long long ty = types ? types[t] : 0;
    for (int i = threadIdx.x; i < H; i += 256) {
        float v = bf2f(W[id * H + i]) + bf2f(P[pos * H + i]);
        if (types) v += bf2f(T[ty * H + i]);
        if (drop_p > 0.f) v = emdr2_keep(emdr2_row_hash(seed, (unsigned long long)t), (uint32_t)i, emdr2_drop_thr(drop_p)) ? v * emdr2_keep_scale(drop_p) : 0.f;
        out[t * H + i] = f2bf(v);
    }
}

// word-embedding rows: atomic scatter (ids are spread over the vocabulary, little contention)
__global__ void __launch_bounds__(256) embedding_bwd_kernel(const long long *ids, const long long *types, const uint16_t *dout, float *dW, float *dP,
                                                            float *dT, long long tokens, int S, int H, float drop_p, uint32_t seed)
{
    const long long t = blockIdx.x;
    const long long id = ids[t];
    // In training, padding positions receive exactly zero gradient (masked as keys everywhere, nothing reads their outputs, every backward
    // kernel maps zero rows to zero rows) and half of the reader's tokens are padding: adding their zeros to row 0 serialised ~800k atomics
    // per column on one address.  A row of zeros is skipped whatever its id (adding it changes nothing, so this is exact for any caller).
    int nz = 0;
    for (int i = threadIdx.x; i < H; i += 256) nz |= (dout[t * H + i] & 0x7fff) != 0;
    if (!__syncthreads_or(nz)) return;
    const uint32_t rh = emdr2_row_hash(seed, (unsigned long long)t), thr = emdr2_drop_thr(drop_p);
    const float ik = drop_p > 0.f ? emdr2_keep_scale(drop_p) : 1.f;
    for (int i = threadIdx.x; i < H; i += 256) {
        float g = bf2f(dout[t * H + i]);
        if (drop_p > 0.f) g = emdr2_keep(rh, (uint32_t)i, thr) ? g * ik : 0.f;
        atomicAdd(&dW[id * H + i], g);
    }
}

// position and token-type rows: every sequence hits the same S position rows and the same one or two type rows, so an atomic per token
// serialises thousands of adds on a few addresses.  One thread owns (position, column) for ONE SLICE of the batch (gridDim.z slices: a
// single walker per (position, column) made 3,200 dependent loads in a row -- 2.6 ms at the reader's shape), walks its sequences, and adds
// its sums with one atomic per table entry: S x H x slices atomics instead of one per token.
__global__ void __launch_bounds__(256) embedding_bwd_pos_kernel(const long long *types, const uint16_t *dout, float *dP, float *dT, long long tokens,
                                                                int S, int H, int n_types, float drop_p, uint32_t seed, const int *cu, int nseq)
{
    // cu (packed layout): sequence b owns rows [cu[b], cu[b+1]); its position `pos` exists iff pos < its length
    const int pos = blockIdx.x, i = blockIdx.y * 256 + threadIdx.x;
    if (i >= H) return;
    const long long nb = cu ? nseq : tokens / S;
    const uint32_t thr = emdr2_drop_thr(drop_p);
    const float ik = drop_p > 0.f ? emdr2_keep_scale(drop_p) : 1.f;
    float ap = 0.f, at[4] = {0.f, 0.f, 0.f, 0.f};
    for (long long b = blockIdx.z; b < nb; b += gridDim.z) {
        long long t = b * S + pos;
        if (cu) {
            const int c0 = cu[b];
            if (pos >= cu[b + 1] - c0) continue;
            t = c0 + pos;
        }
        float g = bf2f(dout[t * H + i]);
        if (drop_p > 0.f) g = emdr2_keep(emdr2_row_hash(seed, (unsigned long long)t), (uint32_t)i, thr) ? g * ik : 0.f;
        ap += g;
        if (types) {
            const int ty = (int)types[t];
#pragma unroll
            for (int k = 0; k < 4; ++k) at[k] += (ty == k) ? g : 0.f;
        }
    }
    if (ap != 0.f) atomicAdd(&dP[(long long)pos * H + i], ap);
    if (types) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < n_types && at[k] != 0.f) atomicAdd(&dT[(long long)k * H + i], at[k]);
    }
}

// ---- dropout mask of a site re-applied to a contiguous bf16 tensor (backward of the fused bias-dropout-add epilogue) --------------
__global__ void __launch_bounds__(256) dropout_kernel(const uint16_t *x, uint16_t *out, long long n, int cols, float drop_p, uint32_t seed)
{
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 8;     // cols % 8 == 0: the 8 elements share a row
    if (i >= n) return;
    const float ik = emdr2_keep_scale(drop_p);
    const uint32_t thr = emdr2_drop_thr(drop_p), rh = emdr2_row_hash(seed, (unsigned long long)(i / cols)), c0 = (uint32_t)(i % cols);
    const uint4 a = *(const uint4 *)(x + i);
    const uint32_t aw[4] = {a.x, a.y, a.z, a.w};
    uint32_t w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t b = emdr2_pair_bits(rh, c0 + 2 * j);
        const float lo = (b & 0xffffu) >= thr ? bf2f((uint16_t)(aw[j] & 0xffff)) * ik : 0.f;
        const float hi = (b >> 16) >= thr ? bf2f((uint16_t)(aw[j] >> 16)) * ik : 0.f;
        w[j] = (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
    }
    *(uint4 *)(out + i) = make_uint4(w[0], w[1], w[2], w[3]);
}

// ---- log-softmax + gather over the vocabulary (train_e2eqa.py:72-123,152-160) --------------------------------------------------
// gold[row] = logits[row, label[row]] - logsumexp(logits[row, :]),  lse[row] kept for the backward; one block per row
__global__ void __launch_bounds__(256) lse_gather_kernel(const uint16_t *logits, const long long *labels, float *gold, float *lse, int V)
{
    __shared__ float red[4];
    const long long row = blockIdx.x;
    const uint16_t *lr = logits + row * V;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float m = -3.0e38f;
    for (int i = threadIdx.x; i < V; i += 256) m = fmaxf(m, bf2f(lr[i]));
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int i = threadIdx.x; i < V; i += 256) s += __expf(bf2f(lr[i]) - m);
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float l = m + __logf(red[0] + red[1] + red[2] + red[3]);
        lse[row] = l;
        gold[row] = bf2f(lr[labels[row]]) - l;
    }
}

// dlogits[row, v] = w[row] * (onehot(label) - softmax)[v]  where w = d(loss)/d(gold[row])
__global__ void __launch_bounds__(256) lse_gather_bwd_kernel(const uint16_t *logits, const long long *labels, const float *lse, const float *w,
                                                             uint16_t *dlogits, int V)
{
    const long long row = blockIdx.x;
    const float l = lse[row], ww = w[row];
    const long long lab = labels[row];
    for (int i = threadIdx.x; i < V; i += 256) {
        const float sm = __expf(bf2f(logits[row * V + i]) - l);
        dlogits[row * V + i] = f2bf(ww * ((i == lab ? 1.f : 0.f) - sm));
    }
}

// ---- optimizer: fp32 masters, decoupled weight decay Adam (apex FusedAdam defaults, training.py:89; SURVEY 8c: unpinned) -------
// Deterministic: block partials go to scratch[0 .. grid) and the LAST block to finish adds them up in index order, so the sum (and with
// it the clip factor every data-parallel replica derives from it) does not depend on the arrival order of atomics.  scratch: 1024
// partial slots + one counter word at [1024] (zero before the first use; the kernel resets it).
__global__ void __launch_bounds__(256) sumsq_kernel(const float *g, long long n, float *out, float *scratch)
{
    __shared__ float red[4];
    __shared__ bool last;
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += g[i] * g[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    unsigned *counter = (unsigned *)(scratch + 1024);                  // fixed slot: partials of launches with other grid sizes never alias it
    if (threadIdx.x == 0) {
        scratch[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
        __threadfence();
        last = atomicAdd(counter, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    float t = 0.f;
    for (unsigned i = threadIdx.x; i < gridDim.x; i += 256) t += ((volatile float *)scratch)[i];   // fixed assignment of partials to lanes
    t = wave_sum(t);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) { *out += (red[0] + red[1]) + (red[2] + red[3]); *counter = 0u; }    // launches on one stream are ordered
}

__global__ void __launch_bounds__(256) adam_kernel(float *master, const float *grad, float *m, float *v, uint16_t *param_bf16, long long n, float lr,
                                                   float b1, float b2, float eps, float wd, float bc1, float bc2, const float *gnorm_sq,
                                                   float clip)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float scale = 1.f;
    if (gnorm_sq && clip > 0.f) { const float gn = sqrtf(*gnorm_sq); const float c = clip / (gn + 1.0e-6f); if (c < 1.f) scale = c; }
    const float g = grad[i] * scale;
    const float mi = b1 * m[i] + (1.f - b1) * g;
    const float vi = b2 * v[i] + (1.f - b2) * g * g;
    m[i] = mi; v[i] = vi;
    const float upd = (mi / bc1) / (sqrtf(vi / bc2) + eps) + wd * master[i];
    const float w = master[i] - lr * upd;
    master[i] = w;
    if (param_bf16) param_bf16[i] = f2bf(w);
}

// Adam over one FLAT bucket (all tensors of the bucket back to back; n % 4 == 0 by construction of the buckets): four elements per thread,
// 16-byte accesses.  Elements [0, split) take the decoupled weight decay, [split, n) none (LayerNorm parameters and biases,
// megatron/model/utils.py:64-83): the bucket stores its decayed parameters first, so one launch covers both groups.
__global__ void __launch_bounds__(256) adam_flat_kernel(float4 *master, const float4 *grad, float4 *m, float4 *v, uint2 *work_bf16, long long n4,
                                                        long long split4, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2,
                                                        const float *gnorm_sq, float clip)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float scale = 1.f;
    if (gnorm_sq && clip > 0.f) { const float gn = sqrtf(*gnorm_sq); const float c = clip / (gn + 1.0e-6f); if (c < 1.f) scale = c; }
    const float wdi = i < split4 ? wd : 0.f;
    const float4 g4 = grad[i], m4 = m[i], v4 = v[i], w4 = master[i];
    float g[4] = {g4.x, g4.y, g4.z, g4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w}, w[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float gj = g[j] * scale;
        mm[j] = b1 * mm[j] + (1.f - b1) * gj;
        vv[j] = b2 * vv[j] + (1.f - b2) * gj * gj;
        const float upd = (mm[j] / bc1) / (sqrtf(vv[j] / bc2) + eps) + wdi * w[j];
        w[j] = w[j] - lr * upd;
    }
    m[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
    v[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
    master[i] = make_float4(w[0], w[1], w[2], w[3]);
    if (work_bf16) work_bf16[i] = make_uint2((uint32_t)f2bf(w[0]) | ((uint32_t)f2bf(w[1]) << 16), (uint32_t)f2bf(w[2]) | ((uint32_t)f2bf(w[3]) << 16));
}

// gradient exchange in bf16 (the reference all-reduces fp16 gradients pre-divided by the world size, model/distributed.py:53-62):
// dst_bf16 = bf16(scale * src), and the way back dst_f32 = float(src_bf16); n % 4 == 0
__global__ void __launch_bounds__(256) scale_cast_kernel(const float4 *src, uint2 *dst, long long n4, float scale)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 x = src[i];
    dst[i] = make_uint2((uint32_t)f2bf(x.x * scale) | ((uint32_t)f2bf(x.y * scale) << 16), (uint32_t)f2bf(x.z * scale) | ((uint32_t)f2bf(x.w * scale) << 16));
}

__global__ void __launch_bounds__(256) widen_kernel(const uint2 *src, float4 *dst, long long n4)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const uint2 x = src[i];
    dst[i] = make_float4(bf2f((uint16_t)(x.x & 0xffff)), bf2f((uint16_t)(x.x >> 16)), bf2f((uint16_t)(x.y & 0xffff)), bf2f((uint16_t)(x.y >> 16)));
}

__global__ void __launch_bounds__(256) cast_kernel(const float *src, uint16_t *dst, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = f2bf(src[i]);
}

__global__ void __launch_bounds__(256) accum_bf16_to_f32_kernel(const uint16_t *src, float *dst, long long n, float scale)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] += scale * bf2f(src[i]);
}

} // namespace

extern "C" int emdr2_transpose_bf16(const void *in, int64_t ld_in, void *out, int64_t ld_out, int rows, int cols, int batch1,
                                    int64_t sI1, int64_t sO1, int batch2, int64_t sI2, int64_t sO2, float *colsum, void *stream)
{
    if (!in || !out || rows < 1 || cols < 1 || batch1 < 1 || batch2 < 1) return -1;
    dim3 grid((cols + 63) / 64, (rows + 63) / 64, batch1 * batch2);
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t *)in, (long long)ld_in, (uint16_t *)out,
                       (long long)ld_out, rows, cols, batch2, (long long)sI1, (long long)sO1, (long long)sI2, (long long)sO2, colsum);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? 0 : -3)

extern "C" int emdr2_layernorm_fwd(const void *x, const float *gamma, const float *beta, void *y, float *mean, float *rstd, int64_t rows,
                                   int H, float eps, void *stream)
{
    if (!x || !gamma || !beta || !y || !mean || !rstd || rows < 1 || H < 8 || (H & 7)) return -1;
    if (H == 768 && !((uintptr_t)x & 15) && !((uintptr_t)y & 15) && !((uintptr_t)gamma & 15) && !((uintptr_t)beta & 15)) {
        hipLaunchKernelGGL(layernorm_fwd768_kernel, dim3((unsigned)((rows + 7) / 8)), dim3(256), 0, (hipStream_t)stream, (const uint16_t *)x, gamma,
                           beta, (uint16_t *)y, mean, rstd, (long long)rows, eps);
        return LAUNCH_OK();
    }
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const uint16_t *)x, gamma, beta,
                       (uint16_t *)y, mean, rstd, (long long)rows, H, eps);
    return LAUNCH_OK();
}

extern "C" int emdr2_layernorm_bwd(const void *dy, const void *x, const float *gamma, const float *mean, const float *rstd, const void *dres,
                                   void *dx, float *dgamma, float *dbeta, int64_t rows, int H, void *stream)
{
    if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || rows < 1 || H < 1 || H > 8192) return -1;
    const bool al = !(((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dx | (uintptr_t)dres | (uintptr_t)gamma) & 15);
    if (H == 768 && al) {
        long long blocks = (rows + 7) / 8;
        if (blocks > 2048) blocks = 2048;                               // persistent: each wave walks ~100 row pairs at the reader's shape
        hipLaunchKernelGGL(layernorm_bwd768_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t *)dy,
                           (const uint16_t *)x, gamma, mean, rstd, (const uint16_t *)dres, (uint16_t *)dx, dgamma, dbeta, (long long)rows,
                           (uint16_t *)nullptr, 0.f, 0u);
        return LAUNCH_OK();
    }
    const int rpb = LN_ROWS;
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)dy, (const uint16_t *)x, gamma, mean, rstd, (const uint16_t *)dres, (uint16_t *)dx, dgamma, dbeta,
                       (long long)rows, H, rpb);
    return LAUNCH_OK();
}

// LayerNorm backward that ALSO writes dmask = dropout_mask(seed) o dx / (1 - p): the operand the bias-dropout-add under this LayerNorm needs for
// its backward GEMMs (emdr2_dropout would make it from dx in a launch of its own).  H == 768 only (-4 otherwise: the caller keeps the two launches).
extern "C" int emdr2_layernorm_bwd_mask(const void *dy, const void *x, const float *gamma, const float *mean, const float *rstd, const void *dres, void *dx,
                                        float *dgamma, float *dbeta, int64_t rows, int H, void *dmask, float drop_p, uint32_t seed, void *stream)
{
    if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || !dmask || rows < 1 || drop_p <= 0.f || drop_p >= 1.f) return -1;
    if (H != 768 || (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dx | (uintptr_t)dres | (uintptr_t)gamma | (uintptr_t)dmask) & 15)) return -4;
    long long blocks = (rows + 7) / 8;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(layernorm_bwd768_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t *)dy,
                       (const uint16_t *)x, gamma, mean, rstd, (const uint16_t *)dres, (uint16_t *)dx, dgamma, dbeta, (long long)rows,
                       (uint16_t *)dmask, drop_p, seed);
    return LAUNCH_OK();
}

extern "C" int emdr2_softmax_mask_fwd(void *scores, const int64_t *ids_q, const int64_t *ids_k, int batch, int heads, int sq, int sk, int causal,
                                      float *m, float *l, float drop_p, uint32_t seed, void *stream)
{
    if (drop_p < 0.f || drop_p >= 1.f) return -1;
    if (!scores || !ids_q || !ids_k || batch < 1 || heads < 1 || sq < 1 || sk < 1) return -1;
    const long long rows = (long long)batch * heads * sq;
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (!(sk & 7) && sk <= 512 && !((uintptr_t)scores & 15))
        hipLaunchKernelGGL(softmax_fwd_reg_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, (uint16_t *)scores, (const long long *)ids_q,
                           (const long long *)ids_k, heads, sq, sk, causal, m, l, rows, drop_p, seed);
    else if (!(sk & 7) && sk <= 2048 && !((uintptr_t)scores & 15))
        hipLaunchKernelGGL(softmax_fwd_reg_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, (uint16_t *)scores, (const long long *)ids_q,
                           (const long long *)ids_k, heads, sq, sk, causal, m, l, rows, drop_p, seed);
    else
        hipLaunchKernelGGL(softmax_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, (uint16_t *)scores, (const long long *)ids_q,
                           (const long long *)ids_k, heads, sq, sk, causal, m, l, rows, drop_p, seed);
    return LAUNCH_OK();
}

extern "C" int emdr2_softmax_mask_bwd(const void *probs, void *dprobs, const int64_t *ids_q, const int64_t *ids_k, const float *m, const float *l,
                                      int batch, int heads, int sq, int sk, int causal, float drop_p, uint32_t seed, float *d, void *stream)
{
    if (drop_p < 0.f || drop_p >= 1.f || (m && !l) || (drop_p > 0.f && !m)) return -1;
    if (!probs || !dprobs || !ids_q || !ids_k || batch < 1 || heads < 1 || sq < 1 || sk < 1) return -1;
    const long long rows = (long long)batch * heads * sq;
    const dim3 grid((unsigned)((rows + 3) / 4));
    const bool al = !((uintptr_t)probs & 15) && !((uintptr_t)dprobs & 15);
    if (!(sk & 7) && sk <= 512 && al)
        hipLaunchKernelGGL(softmax_bwd_reg_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t *)probs, (uint16_t *)dprobs,
                           (const long long *)ids_q, (const long long *)ids_k, heads, sq, sk, causal, d, rows, m, l, drop_p, seed);
    else if (!(sk & 7) && sk <= 2048 && al)
        hipLaunchKernelGGL(softmax_bwd_reg_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t *)probs, (uint16_t *)dprobs,
                           (const long long *)ids_q, (const long long *)ids_k, heads, sq, sk, causal, d, rows, m, l, drop_p, seed);
    else
        hipLaunchKernelGGL(softmax_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t *)probs, (uint16_t *)dprobs,
                           (const long long *)ids_q, (const long long *)ids_k, heads, sq, sk, causal, d, rows, m, l, drop_p, seed);
    return LAUNCH_OK();
}

extern "C" int emdr2_softmax_mask_t(void *scores_t, void *dprobs_t, const int64_t *ids_q, const int64_t *ids_k, const float *m, const float *l,
                                    const float *d, int batch, int heads, int sq, int sk, int causal, float drop_p, uint32_t seed, void *stream)
{
    if (drop_p < 0.f || drop_p >= 1.f) return -1;
    if (!scores_t || !dprobs_t || !ids_q || !ids_k || !m || !l || !d || batch < 1 || heads < 1 || sq < 1 || sk < 1) return -1;
    const long long total = (long long)batch * heads * sq * sk;
    hipLaunchKernelGGL(softmax_t_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (uint16_t *)scores_t,
                       (uint16_t *)dprobs_t, (const long long *)ids_q, (const long long *)ids_k, m, l, d, heads, sq, sk, causal, total, drop_p, seed);
    return LAUNCH_OK();
}

extern "C" int emdr2_gelu_bwd(const void *pre, const void *dact, void *dpre, int64_t n, void *stream)
{
    if (!pre || !dact || !dpre || n < 8 || (n & 7)) return -1;
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint16_t *)pre,
                       (const uint16_t *)dact, (uint16_t *)dpre, (long long)n);
    return LAUNCH_OK();
}

extern "C" int emdr2_embedding_fwd(const int64_t *ids, const int64_t *types, const void *W, const void *P, const void *T, void *out, int64_t tokens,
                                   int S, int H, float drop_p, uint32_t seed, void *stream)
{
    if (!ids || !W || !P || !out || tokens < 1 || S < 1 || H < 1 || (types && !T) || drop_p < 0.f || drop_p >= 1.f) return -1;
    hipLaunchKernelGGL(embedding_fwd_kernel, dim3((unsigned)tokens), dim3(256), 0, (hipStream_t)stream, (const long long *)ids, (const long long *)types,
                       (const uint16_t *)W, (const uint16_t *)P, (const uint16_t *)T, (uint16_t *)out, (long long)tokens, S, H, drop_p, seed, (const int *)nullptr);
    return LAUNCH_OK();
}

extern "C" int emdr2_embedding_packed_fwd(const int64_t *ids, const int64_t *types, const int32_t *rowmap, const void *W, const void *P, const void *T, void *out,
                                          int64_t rows, int S, int H, float drop_p, uint32_t seed, void *stream)
{
    if (!ids || !rowmap || !W || !P || !out || rows < 1 || S < 1 || H < 1 || (types && !T) || drop_p < 0.f || drop_p >= 1.f) return -1;
    hipLaunchKernelGGL(embedding_fwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const long long *)ids, (const long long *)types,
                       (const uint16_t *)W, (const uint16_t *)P, (const uint16_t *)T, (uint16_t *)out, (long long)rows, S, H, drop_p, seed, (const int *)rowmap);
    return LAUNCH_OK();
}

// batch slices of the position-gradient kernel: ~256 sequences per walker, at most 16 slices
static unsigned pos_slices(long long nseq) { const long long s = (nseq + 255) / 256; return (unsigned)(s < 1 ? 1 : (s > 16 ? 16 : s)); }

extern "C" int emdr2_embedding_bwd(const int64_t *ids, const int64_t *types, const void *dout, float *dW, float *dP, float *dT, int64_t tokens, int S,
                                   int H, int n_types, float drop_p, uint32_t seed, void *stream)
{
    if (!ids || !dout || !dW || !dP || tokens < 1 || S < 1 || H < 1 || (types && !dT) || drop_p < 0.f || drop_p >= 1.f) return -1;
    if (tokens % S || (types && (n_types < 1 || n_types > 4))) return -4;
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3((unsigned)tokens), dim3(256), 0, (hipStream_t)stream, (const long long *)ids, (const long long *)types,
                       (const uint16_t *)dout, dW, dP, dT, (long long)tokens, S, H, drop_p, seed);
    hipLaunchKernelGGL(embedding_bwd_pos_kernel, dim3((unsigned)S, (unsigned)((H + 255) / 256), pos_slices(tokens / S)), dim3(256), 0, (hipStream_t)stream,
                       (const long long *)types, (const uint16_t *)dout, dP, dT, (long long)tokens, S, H, n_types, drop_p, seed, (const int *)nullptr, 0);
    return LAUNCH_OK();
}

extern "C" int emdr2_embedding_packed_bwd(const int64_t *ids, const int64_t *types, const int32_t *cu, int nseq, const void *dout, float *dW, float *dP, float *dT,
                                          int64_t rows, int S, int H, int n_types, float drop_p, uint32_t seed, void *stream)
{
    if (!ids || !cu || nseq < 1 || !dout || !dW || !dP || rows < 1 || S < 1 || H < 1 || (types && !dT) || drop_p < 0.f || drop_p >= 1.f) return -1;
    if (types && (n_types < 1 || n_types > 4)) return -4;
    // word rows: the same atomic scatter (a tail row carries a zero gradient and is skipped like any zero row)
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const long long *)ids, (const long long *)types,
                       (const uint16_t *)dout, dW, dP, dT, (long long)rows, S, H, drop_p, seed);
    hipLaunchKernelGGL(embedding_bwd_pos_kernel, dim3((unsigned)S, (unsigned)((H + 255) / 256), pos_slices(nseq)), dim3(256), 0, (hipStream_t)stream,
                       (const long long *)types, (const uint16_t *)dout, dP, dT, (long long)rows, S, H, n_types, drop_p, seed, (const int *)cu, nseq);
    return LAUNCH_OK();
}

extern "C" int emdr2_dropout(const void *x, void *out, int64_t n, int cols, float drop_p, uint32_t seed, void *stream)
{
    if (!x || !out || n < 8 || (n & 7) || cols < 8 || (cols & 7) || n % cols || drop_p <= 0.f || drop_p >= 1.f || ((uintptr_t)x & 15) || ((uintptr_t)out & 15)) return -1;
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint16_t *)x, (uint16_t *)out,
                       (long long)n, cols, drop_p, seed);
    return LAUNCH_OK();
}

// ---- retriever prior (emdr2_model.py:134-145): sim[b, k] = <q[b], c[b, k]> * scale, log-softmax over the K retrieved passages --------------------
// One workgroup per question; K <= 1024 (top-k 50 / 100 + 1 in the shipped scripts).  q, c bf16, everything else fp32.  prob (= exp(logp)) is
// kept for the backward.
#define PRIOR_MAX_K 1024
__global__ void __launch_bounds__(256) retriever_prior_fwd_kernel(const uint16_t *q, const uint16_t *c, float *logp, float *prob, int K, int H, float scale)
{
    __shared__ float sim[PRIOR_MAX_K];
    __shared__ float stat[2];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint16_t *qb = q + (long long)b * H;
    for (int k = wave; k < K; k += 4) {
        const uint16_t *ck = c + ((long long)b * K + k) * H;
        float s = 0.f;
        for (int i = lane * 8; i < H; i += 512) {
            const uint4 qa = *(const uint4 *)(qb + i), ca = *(const uint4 *)(ck + i);
            const uint32_t qw[4] = {qa.x, qa.y, qa.z, qa.w}, cw[4] = {ca.x, ca.y, ca.z, ca.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                s += bf2f((uint16_t)(qw[j] & 0xffff)) * bf2f((uint16_t)(cw[j] & 0xffff)) + bf2f((uint16_t)(qw[j] >> 16)) * bf2f((uint16_t)(cw[j] >> 16));
        }
        s = wave_sum(s);
        if (lane == 0) sim[k] = s * scale;
    }
    __syncthreads();
    if (wave == 0) {
        float a = -3.0e38f;
        for (int k = lane; k < K; k += 64) a = fmaxf(a, sim[k]);
        const float m = wave_max(a);
        float e = 0.f;
        for (int k = lane; k < K; k += 64) e += __expf(sim[k] - m);
        const float l = wave_sum(e);
        if (lane == 0) { stat[0] = m; stat[1] = __logf(l); }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += 256) {
        const float lp = sim[k] - stat[0] - stat[1];
        logp[(long long)b * K + k] = lp;
        prob[(long long)b * K + k] = __expf(lp);
    }
}

// d sim[k] = (g[k] - prob[k] * sum_j g[j]) * scale;  dq[h] = sum_k dsim[k] c[k, h];  dc[k, h] = dsim[k] q[h]   (bf16 gradients out)
__global__ void __launch_bounds__(256) retriever_prior_bwd_kernel(const float *g, const float *prob, const uint16_t *q, const uint16_t *c, uint16_t *dq,
                                                                  uint16_t *dc, int K, int H, float scale)
{
    __shared__ float dsim[PRIOR_MAX_K];
    __shared__ float gsum;
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *gb = g + (long long)b * K, *pb = prob + (long long)b * K;
    if (wave == 0) {
        float t = 0.f;
        for (int k = lane; k < K; k += 64) t += gb[k];
        t = wave_sum(t);
        if (lane == 0) gsum = t;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += 256) dsim[k] = (gb[k] - pb[k] * gsum) * scale;
    __syncthreads();
    const uint16_t *qb = q + (long long)b * H, *cb = c + (long long)b * K * H;
    for (int h = threadIdx.x; h < H; h += 256) {
        float acc = 0.f;
        const float qh = bf2f(qb[h]);
        for (int k = 0; k < K; ++k) {
            acc += dsim[k] * bf2f(cb[(long long)k * H + h]);
            if (dc) dc[((long long)b * K + k) * H + h] = f2bf(dsim[k] * qh);
        }
        if (dq) dq[(long long)b * H + h] = f2bf(acc);
    }
}

// ---- EMDR2 marginal (train_e2eqa.py:98-123): marginal[b, l] = logsumexp_k( prior[b, k] + gold[b, k, l] ) -------------------------------------------
__global__ void __launch_bounds__(64) marginal_fwd_kernel(const float *prior, const float *gold, float *marginal, int K, int L)
{
    const int b = blockIdx.x;
    for (int l = threadIdx.x; l < L; l += 64) {
        float m = -3.0e38f;
        for (int k = 0; k < K; ++k) m = fmaxf(m, prior[(long long)b * K + k] + gold[((long long)b * K + k) * L + l]);
        float s = 0.f;
        for (int k = 0; k < K; ++k) s += __expf(prior[(long long)b * K + k] + gold[((long long)b * K + k) * L + l] - m);
        marginal[(long long)b * L + l] = m + __logf(s);
    }
}

// d prior[b, k] = sum_l gm[b, l] * exp(prior[b, k] + gold[b, k, l] - marginal[b, l])      (gold is a constant of the loss: no-grad pass)
__global__ void __launch_bounds__(128) marginal_bwd_kernel(const float *prior, const float *gold, const float *marginal, const float *gm, float *dprior,
                                                           int K, int L)
{
    const int b = blockIdx.x;
    for (int k = threadIdx.x; k < K; k += 128) {
        const float pk = prior[(long long)b * K + k];
        float acc = 0.f;
        for (int l = 0; l < L; ++l) acc += gm[(long long)b * L + l] * __expf(pk + gold[((long long)b * K + k) * L + l] - marginal[(long long)b * L + l]);
        dprior[(long long)b * K + k] = acc;
    }
}

// out[row] = gold[row] - logsumexp over the row's partials (max_j, sum_j exp(x - max_j)) written by emdr2_gemm_nt_lse_bf16: one wave per row
__global__ void __launch_bounds__(256) lse_combine_kernel(const float *pmax, const float *psum, const float *gold, float *out, float *lse,
                                                           long long rows, int slots)
{
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float *pm = pmax + row * slots, *ps = psum + row * slots;
    float m = -3.0e38f;
    for (int i = lane; i < slots; i += 64) m = fmaxf(m, pm[i]);
    m = wave_max(m);
    float s = 0.f;
    for (int i = lane; i < slots; i += 64) s += ps[i] * __expf(pm[i] - m);
    s = wave_sum(s);
    if (lane == 0) {
        const float l = m + __logf(s);
        if (lse) lse[row] = l;
        out[row] = gold[row] - l;
    }
}

extern "C" int emdr2_retriever_prior_fwd(const void *q, const void *c, float *logp, float *prob, int batch, int K, int H, float scale, void *stream)
{
    if (!q || !c || !logp || !prob || batch < 1 || K < 1 || H < 8 || (H & 7) || ((uintptr_t)q & 15) || ((uintptr_t)c & 15)) return -1;
    if (K > PRIOR_MAX_K) return -4;
    hipLaunchKernelGGL(retriever_prior_fwd_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, (const uint16_t *)q, (const uint16_t *)c, logp, prob, K,
                       H, scale);
    return LAUNCH_OK();
}

extern "C" int emdr2_retriever_prior_bwd(const float *dlogp, const float *prob, const void *q, const void *c, void *dq, void *dc, int batch, int K, int H,
                                         float scale, void *stream)
{
    if (!dlogp || !prob || !q || !c || (!dq && !dc) || batch < 1 || K < 1 || H < 8) return -1;
    if (K > PRIOR_MAX_K) return -4;
    hipLaunchKernelGGL(retriever_prior_bwd_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, dlogp, prob, (const uint16_t *)q, (const uint16_t *)c,
                       (uint16_t *)dq, (uint16_t *)dc, K, H, scale);
    return LAUNCH_OK();
}

extern "C" int emdr2_marginal_fwd(const float *prior, const float *gold, float *marginal, int batch, int K, int L, void *stream)
{
    if (!prior || !gold || !marginal || batch < 1 || K < 1 || L < 1) return -1;
    hipLaunchKernelGGL(marginal_fwd_kernel, dim3(batch), dim3(64), 0, (hipStream_t)stream, prior, gold, marginal, K, L);
    return LAUNCH_OK();
}

extern "C" int emdr2_marginal_bwd(const float *prior, const float *gold, const float *marginal, const float *dmarginal, float *dprior, int batch, int K,
                                  int L, void *stream)
{
    if (!prior || !gold || !marginal || !dmarginal || !dprior || batch < 1 || K < 1 || L < 1) return -1;
    hipLaunchKernelGGL(marginal_bwd_kernel, dim3(batch), dim3(128), 0, (hipStream_t)stream, prior, gold, marginal, dmarginal, dprior, K, L);
    return LAUNCH_OK();
}

extern "C" int emdr2_lse_combine(const float *part_max, const float *part_sum, const float *gold, float *out, float *lse, int64_t rows, int slots,
                                 void *stream)
{
    if (!part_max || !part_sum || !gold || !out || rows < 1 || slots < 1) return -1;
    hipLaunchKernelGGL(lse_combine_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, part_max, part_sum, gold, out, lse,
                       (long long)rows, slots);
    return LAUNCH_OK();
}

extern "C" int emdr2_lse_gather_fwd(const void *logits, const int64_t *labels, float *gold, float *lse, int64_t rows, int V, void *stream)
{
    if (!logits || !labels || !gold || !lse || rows < 1 || V < 1) return -1;
    hipLaunchKernelGGL(lse_gather_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const uint16_t *)logits, (const long long *)labels,
                       gold, lse, V);
    return LAUNCH_OK();
}

extern "C" int emdr2_lse_gather_bwd(const void *logits, const int64_t *labels, const float *lse, const float *w, void *dlogits, int64_t rows, int V,
                                    void *stream)
{
    if (!logits || !labels || !lse || !w || !dlogits || rows < 1 || V < 1) return -1;
    hipLaunchKernelGGL(lse_gather_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const uint16_t *)logits,
                       (const long long *)labels, lse, w, (uint16_t *)dlogits, V);
    return LAUNCH_OK();
}

extern "C" int emdr2_sumsq_f32(const float *g, int64_t n, float *out, float *scratch, void *stream)
{
    if (!g || !out || !scratch || n < 1) return -1;
    int blocks = (int)((n + 2047) / 2048);                               // a function of n only: every replica uses the same grid
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, (long long)n, out, scratch);
    return LAUNCH_OK();
}

extern "C" int emdr2_adam_step(float *master, const float *grad, float *m, float *v, void *param_bf16, int64_t n, float lr, float beta1, float beta2,
                               float eps, float weight_decay, int step, const float *gnorm_sq, float clip, void *stream)
{
    if (!master || !grad || !m || !v || n < 1 || step < 1) return -1;
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, master, grad, m, v, (uint16_t *)param_bf16,
                       (long long)n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, gnorm_sq, clip);
    return LAUNCH_OK();
}

extern "C" int emdr2_adam_step_flat(float *master, const float *grad, float *m, float *v, void *work_bf16, int64_t n, int64_t decay_split, float lr,
                                    float beta1, float beta2, float eps, float weight_decay, int step, const float *gnorm_sq, float clip, void *stream)
{
    if (!master || !grad || !m || !v || n < 4 || (n & 3) || (decay_split & 3) || decay_split < 0 || decay_split > n || step < 1) return -1;
    if (((uintptr_t)master | (uintptr_t)grad | (uintptr_t)m | (uintptr_t)v) & 15 || ((uintptr_t)work_bf16 & 7)) return -1;
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    const long long n4 = n / 4;
    hipLaunchKernelGGL(adam_flat_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (float4 *)master, (const float4 *)grad,
                       (float4 *)m, (float4 *)v, (uint2 *)work_bf16, n4, (long long)(decay_split / 4), lr, beta1, beta2, eps, weight_decay, bc1, bc2,
                       gnorm_sq, clip);
    return LAUNCH_OK();
}

extern "C" int emdr2_scale_cast_f32_to_bf16(const float *src, void *dst, int64_t n, float scale, void *stream)
{
    if (!src || !dst || n < 4 || (n & 3) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 7)) return -1;
    hipLaunchKernelGGL(scale_cast_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float4 *)src, (uint2 *)dst,
                       (long long)(n / 4), scale);
    return LAUNCH_OK();
}

extern "C" int emdr2_widen_bf16_to_f32(const void *src, float *dst, int64_t n, void *stream)
{
    if (!src || !dst || n < 4 || (n & 3) || ((uintptr_t)dst & 15) || ((uintptr_t)src & 7)) return -1;
    hipLaunchKernelGGL(widen_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint2 *)src, (float4 *)dst,
                       (long long)(n / 4));
    return LAUNCH_OK();
}

extern "C" int emdr2_cast_f32_to_bf16(const float *src, void *dst, int64_t n, void *stream)
{
    if (!src || !dst || n < 1) return -1;
    hipLaunchKernelGGL(cast_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, (uint16_t *)dst, (long long)n);
    return LAUNCH_OK();
}

extern "C" int emdr2_accum_bf16_to_f32(const void *src, float *dst, int64_t n, float scale, void *stream)
{
    if (!src || !dst || n < 1) return -1;
    hipLaunchKernelGGL(accum_bf16_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint16_t *)src, dst,
                       (long long)n, scale);
    return LAUNCH_OK();
}
