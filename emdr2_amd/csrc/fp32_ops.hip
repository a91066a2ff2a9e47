// emdr2_amd/csrc/fp32_ops.hip -- the VALIDATION-ONLY fp32 compute path (include/emdr2_ops_f32.h).
//
// north_star: "reader logits within 1e-3 fp32 / 2e-2 bf16".  The product is bf16 (gemm8.hip, attention.hip, ...); this file is what lets the
// SAME module tree run with fp32 activations on the fp32 matrix cores so that the first half of that sentence can be tested.  Nothing here
// is tuned: one LDS-tiled GEMM with arbitrary element strides (64 x 64 output tile per workgroup, four waves of one 32 x 32
// v_mfma_f32_32x32x2_f32 accumulator each, K in steps of 16) serves every contraction of the forward and of the backward; the row kernels
// are one wave per row.  fp32 in, fp32 accumulate, fp32 out -- the reference's arithmetic when --fp16 is not given
// (megatron/training.py:55-56).
#include "../../include/emdr2_ops_f32.h"
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float floatx16 __attribute__((ext_vector_type(16)));

namespace {

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? 0 : -3)

__device__ __forceinline__ float wave_sum(float x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}
__device__ __forceinline__ float wave_max(float x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
    return x;
}

// ================================================================ GEMM =================================================================
struct GemmP {
    const float *A, *B;
    float *C;
    const float *bias, *residual;
    long long a_rs, a_cs, a_b1, a_b2, b_rs, b_cs, b_b1, b_b2, c_rs, c_cs, c_b1, c_b2;
    int M, N, K, batch2, accumulate;
    float alpha;
};

#define GT 64      // output tile edge
#define GK 16      // K step
__global__ void __launch_bounds__(256) gemm_f32_kernel(GemmP p)
{
    __shared__ float As[GT][GK + 1], Bs[GT][GK + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
    const int b1 = blockIdx.z / p.batch2, b2 = blockIdx.z - b1 * p.batch2;
    const float *A = p.A + b1 * p.a_b1 + b2 * p.a_b2;
    const float *B = p.B + b1 * p.b_b1 + b2 * p.b_b2;
    const long long coff = b1 * p.c_b1 + b2 * p.c_b2;
    // which index runs fastest over the threads of a load: the one with unit stride (rows of an x / W operand: k; of a transposed one: the row)
    const bool a_kfast = p.a_cs == 1 || p.a_rs != 1, b_kfast = p.b_cs == 1 || p.b_rs != 1;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < p.K; k0 += GK) {
#pragma unroll
        for (int i = 0; i < (GT * GK) / 256; ++i) {
            const int e = tid + 256 * i;
            {
                const int r = a_kfast ? e / GK : e % GT, kk = a_kfast ? e % GK : e / GT;
                const int gm = m0 + r, gk = k0 + kk;
                As[r][kk] = (gm < p.M && gk < p.K) ? A[gm * p.a_rs + gk * p.a_cs] : 0.f;
            }
            {
                const int r = b_kfast ? e / GK : e % GT, kk = b_kfast ? e % GK : e / GT;
                const int gn = n0 + r, gk = k0 + kk;
                Bs[r][kk] = (gn < p.N && gk < p.K) ? B[gn * p.b_rs + gk * p.b_cs] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < GK; kk += 2) {
            // v_mfma_f32_32x32x2_f32: A[i][k] in lane i + 32 k, B[k][j] in lane j + 32 k, D[i][j]: lane = j + 32 ((i >> 2) & 1), register = (i & 3) + 4 (i >> 3)
            const float a = As[wm + (lane & 31)][kk + (lane >> 5)];
            const float b = Bs[wn + (lane & 31)][kk + (lane >> 5)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    const int gn = n0 + wn + (lane & 31);
    if (gn >= p.N) return;
    const float bv = p.bias ? p.bias[gn] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int gm = m0 + wm + (r & 3) + 4 * (lane >> 5) + 8 * (r >> 2);
        if (gm >= p.M) continue;
        const long long ci = coff + gm * p.c_rs + gn * p.c_cs;
        float v = p.alpha * acc[r] + bv;
        if (p.residual) v += p.residual[ci];
        if (p.accumulate) v += p.C[ci];
        p.C[ci] = v;
    }
}

// ============================================================== LayerNorm ==============================================================
// one wave per row, four rows per block; two passes over the row (mean, then centred sum of squares: the numerically plain form)
__global__ void __launch_bounds__(256) ln_fwd_f32_kernel(const float *x, const float *gamma, const float *beta, float *y, float *mean, float *rstd,
                                                         long long rows, int H, float eps)
{
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float *xr = x + row * H;
    float s = 0.f;
    for (int i = lane; i < H; i += 64) s += xr[i];
    const float mu = wave_sum(s) / H;
    float v = 0.f;
    for (int i = lane; i < H; i += 64) { const float d = xr[i] - mu; v += d * d; }
    const float rs = rsqrtf(wave_sum(v) / H + eps);
    for (int i = lane; i < H; i += 64) y[row * H + i] = (xr[i] - mu) * rs * gamma[i] + beta[i];
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

__global__ void __launch_bounds__(256) ln_bwd_f32_kernel(const float *dy, const float *x, const float *gamma, const float *mean, const float *rstd,
                                                         const float *dres, float *dx, float *dgamma, float *dbeta, long long rows, int H)
{
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float mu = mean[row], rs = rstd[row];
    const float *xr = x + row * H, *dyr = dy + row * H;
    float s1 = 0.f, s2 = 0.f;
    for (int i = lane; i < H; i += 64) {
        const float g = dyr[i] * gamma[i], xh = (xr[i] - mu) * rs;
        s1 += g; s2 += g * xh;
    }
    s1 = wave_sum(s1) / H; s2 = wave_sum(s2) / H;
    for (int i = lane; i < H; i += 64) {
        const float xh = (xr[i] - mu) * rs, g = dyr[i] * gamma[i];
        float d = rs * (g - s1 - xh * s2);
        if (dres) d += dres[row * H + i];
        dx[row * H + i] = d;
        atomicAdd(dgamma + i, dyr[i] * xh);
        atomicAdd(dbeta + i, dyr[i]);
    }
}

// =============================================================== softmax ===============================================================
// row = (b, head, q) of scores [batch, heads, sq, sk]; masked(q, k) = ids_q[b][q] == 0 || ids_k[b][k] == 0 || (causal && k > q)
__global__ void __launch_bounds__(256) softmax_fwd_f32_kernel(float *scores, const long long *ids_q, const long long *ids_k, int heads, int sq, int sk,
                                                              int causal, long long rows)
{
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int q = (int)(row % sq);
    const long long b = row / ((long long)sq * heads);
    const bool qpad = ids_q[b * sq + q] == 0;
    const long long *kid = ids_k + b * sk;
    float *sr = scores + row * sk;
    float m = -3.0e38f;
    for (int k = lane; k < sk; k += 64) {
        const bool masked = qpad || kid[k] == 0 || (causal && k > q);
        const float v = masked ? -10000.0f : sr[k];
        sr[k] = v;
        m = fmaxf(m, v);
    }
    m = wave_max(m);
    float s = 0.f;
    for (int k = lane; k < sk; k += 64) { const float e = expf(sr[k] - m); sr[k] = e; s += e; }
    const float inv = 1.f / wave_sum(s);
    for (int k = lane; k < sk; k += 64) sr[k] *= inv;
}

__global__ void __launch_bounds__(256) softmax_bwd_f32_kernel(const float *probs, float *dprobs, const long long *ids_q, const long long *ids_k, int heads,
                                                              int sq, int sk, int causal, long long rows)
{
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int q = (int)(row % sq);
    const long long b = row / ((long long)sq * heads);
    const bool qpad = ids_q[b * sq + q] == 0;
    const long long *kid = ids_k + b * sk;
    const float *pr = probs + row * sk;
    float *dr = dprobs + row * sk;
    float s = 0.f;
    for (int k = lane; k < sk; k += 64) s += pr[k] * dr[k];
    s = wave_sum(s);
    for (int k = lane; k < sk; k += 64) {
        const bool masked = qpad || kid[k] == 0 || (causal && k > q);
        dr[k] = masked ? 0.f : pr[k] * (dr[k] - s);
    }
}

// ================================================================ GELU =================================================================
__global__ void __launch_bounds__(256) gelu_fwd_f32_kernel(const float *x, float *y, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const float v = x[i]; y[i] = 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }
}
__global__ void __launch_bounds__(256) gelu_bwd_f32_kernel(const float *x, const float *dy, float *dx, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const float v = x[i];
        const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752f));
        const float pdf = 0.39894228040143268f * expf(-0.5f * v * v);
        dx[i] = dy[i] * (cdf + v * pdf);
    }
}

// ============================================================== embedding ==============================================================
__global__ void __launch_bounds__(256) emb_fwd_f32_kernel(const long long *ids, const long long *types, const float *W, const float *P, const float *T,
                                                          float *out, long long tokens, int S, int H)
{
    const long long t = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (t >= tokens) return;
    const float *w = W + ids[t] * H, *pp = P + (t % S) * H, *tt = T ? T + types[t] * H : nullptr;
    for (int i = lane; i < H; i += 64) out[t * H + i] = w[i] + pp[i] + (tt ? tt[i] : 0.f);
}
__global__ void __launch_bounds__(256) emb_bwd_f32_kernel(const long long *ids, const long long *types, const float *dout, float *dW, float *dP, float *dT,
                                                          long long tokens, int S, int H)
{
    const long long t = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (t >= tokens) return;
    float *w = dW + ids[t] * H, *pp = dP + (t % S) * H, *tt = dT ? dT + types[t] * H : nullptr;
    for (int i = lane; i < H; i += 64) {
        const float g = dout[t * H + i];
        atomicAdd(w + i, g); atomicAdd(pp + i, g);
        if (tt) atomicAdd(tt + i, g);
    }
}

// ======================================================= log-softmax + gather =========================================================
__global__ void __launch_bounds__(256) lse_fwd_f32_kernel(const float *logits, const long long *labels, float *gold, float *lse, int V)
{
    __shared__ float red[4];
    const long long row = blockIdx.x;
    const float *lr = logits + row * V;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float m = -3.0e38f;
    for (int i = threadIdx.x; i < V; i += 256) m = fmaxf(m, lr[i]);
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int i = threadIdx.x; i < V; i += 256) s += expf(lr[i] - m);
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float l = m + logf((red[0] + red[1]) + (red[2] + red[3]));
        lse[row] = l;
        gold[row] = lr[labels[row]] - l;
    }
}
__global__ void __launch_bounds__(256) lse_bwd_f32_kernel(const float *logits, const long long *labels, const float *lse, const float *w, float *dlogits, int V)
{
    const long long row = blockIdx.x;
    const float l = lse[row], ww = w[row];
    const long long lab = labels[row];
    for (int i = threadIdx.x; i < V; i += 256) dlogits[row * V + i] = ww * ((i == lab ? 1.f : 0.f) - expf(logits[row * V + i] - l));
}

// =========================================================== retriever prior ==========================================================
// one workgroup per question; K <= 1024 scores staged in LDS
__global__ void __launch_bounds__(256) prior_fwd_f32_kernel(const float *q, const float *c, float *logp, float *prob, int K, int H, float scale)
{
    __shared__ float sc[1024];
    __shared__ float red[4];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *qb = q + (long long)b * H;
    for (int k = wave; k < K; k += 4) {
        const float *ck = c + ((long long)b * K + k) * H;
        float s = 0.f;
        for (int i = lane; i < H; i += 64) s += qb[i] * ck[i];
        s = wave_sum(s);
        if (lane == 0) sc[k] = s * scale;
    }
    __syncthreads();
    float m = -3.0e38f;
    for (int k = threadIdx.x; k < K; k += 256) m = fmaxf(m, sc[k]);
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) s += expf(sc[k] - m);
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float l = m + logf((red[0] + red[1]) + (red[2] + red[3]));
    for (int k = threadIdx.x; k < K; k += 256) {
        const float lp = sc[k] - l;
        logp[(long long)b * K + k] = lp;
        prob[(long long)b * K + k] = expf(lp);
    }
}
__global__ void __launch_bounds__(256) prior_bwd_f32_kernel(const float *dlogp, const float *prob, const float *q, const float *c, float *dq, float *dc,
                                                            int K, int H, float scale)
{
    __shared__ float ds[1024];
    __shared__ float red[4];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float s = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) s += dlogp[(long long)b * K + k];
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float tot = (red[0] + red[1]) + (red[2] + red[3]);
    for (int k = threadIdx.x; k < K; k += 256) ds[k] = (dlogp[(long long)b * K + k] - prob[(long long)b * K + k] * tot) * scale;
    __syncthreads();
    const float *qb = q + (long long)b * H;
    for (int i = threadIdx.x; i < H; i += 256) {
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc += ds[k] * c[((long long)b * K + k) * H + i];
        dq[(long long)b * H + i] = acc;
    }
    for (long long e = threadIdx.x; e < (long long)K * H; e += 256) {
        const int k = (int)(e / H), i = (int)(e - (long long)k * H);
        dc[((long long)b * K + k) * H + i] = ds[k] * qb[i];
    }
}

} // namespace

extern "C" int emdr2_f32_gemm(const float *A, int64_t a_rs, int64_t a_cs, int64_t a_b1, int64_t a_b2, const float *B, int64_t b_rs, int64_t b_cs,
                              int64_t b_b1, int64_t b_b2, float *C, int64_t c_rs, int64_t c_cs, int64_t c_b1, int64_t c_b2, int M, int N, int K,
                              int batch1, int batch2, float alpha, const float *bias, const float *residual, int accumulate, void *stream)
{
    if (!A || !B || !C || M < 1 || N < 1 || K < 1 || batch1 < 1 || batch2 < 1 || (long long)batch1 * batch2 > 65535) return -1;
    GemmP p{A, B, C, bias, residual, a_rs, a_cs, a_b1, a_b2, b_rs, b_cs, b_b1, b_b2, c_rs, c_cs, c_b1, c_b2, M, N, K, batch2, accumulate ? 1 : 0, alpha};
    const dim3 grid((unsigned)((N + GT - 1) / GT), (unsigned)((M + GT - 1) / GT), (unsigned)(batch1 * batch2));
    if (grid.y > 65535) return -1;
    hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    return LAUNCH_OK();
}

extern "C" int emdr2_f32_layernorm_fwd(const float *x, const float *gamma, const float *beta, float *y, float *mean, float *rstd, int64_t rows, int H,
                                       float eps, void *stream)
{
    if (!x || !gamma || !beta || !y || !mean || !rstd || rows < 1 || H < 1) return -1;
    hipLaunchKernelGGL(ln_fwd_f32_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, mean, rstd,
                       (long long)rows, H, eps);
    return LAUNCH_OK();
}

extern "C" int emdr2_f32_layernorm_bwd(const float *dy, const float *x, const float *gamma, const float *mean, const float *rstd, const float *dres,
                                       float *dx, float *dgamma, float *dbeta, int64_t rows, int H, void *stream)
{
    if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || rows < 1 || H < 1) return -1;
    hipLaunchKernelGGL(ln_bwd_f32_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dy, x, gamma, mean, rstd, dres, dx,
                       dgamma, dbeta, (long long)rows, H);
    return LAUNCH_OK();
}

extern "C" int emdr2_f32_softmax_mask_fwd(float *scores, const int64_t *ids_q, const int64_t *ids_k, int batch, int heads, int sq, int sk, int causal,
                                          void *stream)
{
    if (!scores || !ids_q || !ids_k || batch < 1 || heads < 1 || sq < 1 || sk < 1) return -1;
    const long long rows = (long long)batch * heads * sq;
    hipLaunchKernelGGL(softmax_fwd_f32_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, scores, (const long long *)ids_q,
                       (const long long *)ids_k, heads, sq, sk, causal, rows);
    return LAUNCH_OK();
}

extern "C" int emdr2_f32_softmax_mask_bwd(const float *probs, float *dprobs, const int64_t *ids_q, const int64_t *ids_k, int batch, int heads, int sq,
                                          int sk, int causal, void *stream)
{
    if (!probs || !dprobs || !ids_q || !ids_k || batch < 1 || heads < 1 || sq < 1 || sk < 1) return -1;
    const long long rows = (long long)batch * heads * sq;
    hipLaunchKernelGGL(softmax_bwd_f32_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, probs, dprobs,
                       (const long long *)ids_q, (const long long *)ids_k, heads, sq, sk, causal, rows);
    return LAUNCH_OK();
}

extern "C" int emdr2_f32_gelu_fwd(const float *x, float *y, int64_t n, void *stream)
{
    if (!x || !y || n < 1) return -1;
    hipLaunchKernelGGL(gelu_fwd_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, (long long)n);
    return LAUNCH_OK();
}

extern "C" int emdr2_f32_gelu_bwd(const float *x, const float *dy, float *dx, int64_t n, void *stream)
{
    if (!x || !dy || !dx || n < 1) return -1;
    hipLaunchKernelGGL(gelu_bwd_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, (long long)n);
    return LAUNCH_OK();
}

extern "C" int emdr2_f32_embedding_fwd(const int64_t *ids, const int64_t *types, const float *W, const float *P, const float *T, float *out,
                                       int64_t tokens, int S, int H, void *stream)
{
    if (!ids || !W || !P || !out || tokens < 1 || S < 1 || H < 1 || (T && !types)) return -1;
    hipLaunchKernelGGL(emb_fwd_f32_kernel, dim3((unsigned)((tokens + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const long long *)ids,
                       (const long long *)types, W, P, T, out, (long long)tokens, S, H);
    return LAUNCH_OK();
}

extern "C" int emdr2_f32_embedding_bwd(const int64_t *ids, const int64_t *types, const float *dout, float *dW, float *dP, float *dT, int64_t tokens,
                                       int S, int H, void *stream)
{
    if (!ids || !dout || !dW || !dP || tokens < 1 || S < 1 || H < 1 || (dT && !types)) return -1;
    hipLaunchKernelGGL(emb_bwd_f32_kernel, dim3((unsigned)((tokens + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const long long *)ids,
                       (const long long *)types, dout, dW, dP, dT, (long long)tokens, S, H);
    return LAUNCH_OK();
}

extern "C" int emdr2_f32_lse_gather_fwd(const float *logits, const int64_t *labels, float *gold, float *lse, int64_t rows, int V, void *stream)
{
    if (!logits || !labels || !gold || !lse || rows < 1 || V < 1) return -1;
    hipLaunchKernelGGL(lse_fwd_f32_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, logits, (const long long *)labels, gold, lse, V);
    return LAUNCH_OK();
}

extern "C" int emdr2_f32_lse_gather_bwd(const float *logits, const int64_t *labels, const float *lse, const float *w, float *dlogits, int64_t rows,
                                        int V, void *stream)
{
    if (!logits || !labels || !lse || !w || !dlogits || rows < 1 || V < 1) return -1;
    hipLaunchKernelGGL(lse_bwd_f32_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, logits, (const long long *)labels, lse, w, dlogits, V);
    return LAUNCH_OK();
}

extern "C" int emdr2_f32_retriever_prior_fwd(const float *q, const float *c, float *logp, float *prob, int batch, int K, int H, float scale,
                                             void *stream)
{
    if (!q || !c || !logp || !prob || batch < 1 || K < 1 || K > 1024 || H < 1) return -1;
    hipLaunchKernelGGL(prior_fwd_f32_kernel, dim3((unsigned)batch), dim3(256), 0, (hipStream_t)stream, q, c, logp, prob, K, H, scale);
    return LAUNCH_OK();
}

extern "C" int emdr2_f32_retriever_prior_bwd(const float *dlogp, const float *prob, const float *q, const float *c, float *dq, float *dc, int batch,
                                             int K, int H, float scale, void *stream)
{
    if (!dlogp || !prob || !q || !c || !dq || !dc || batch < 1 || K < 1 || K > 1024 || H < 1) return -1;
    hipLaunchKernelGGL(prior_bwd_f32_kernel, dim3((unsigned)batch), dim3(256), 0, (hipStream_t)stream, dlogp, prob, q, c, dq, dc, K, H, scale);
    return LAUNCH_OK();
}
