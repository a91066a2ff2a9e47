/* emdr2_ops_f32.h -- C ABI of the VALIDATION-ONLY fp32 compute path of libemdr2_hip.so (ABI 4).
 *
 * north_star states two tolerances for the reader: logits within 1e-3 in fp32, 2e-2 in bf16.  The product computes in bf16
 * (include/emdr2_ops.h: MFMA bf16 kernels; `--fp16` of every shipped script, examples/openqa/emdr2_*.sh).  The reference also runs in fp32
 * when `--fp16` is not given (megatron/training.py:55-56,92: the model is wrapped in FP16_Module / FP16_Optimizer only under args.fp16).
 * These entry points are that mode, for validation: the same module tree evaluated with fp32 activations, fp32 parameters (the masters
 * themselves, no working copies) and fp32 accumulation on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), so that the HIP path can be held to
 * the 1e-3 bar against the fp32 oracle (tests/test_parity_fp32_gpu.py).  Speed is a non-goal: one simple LDS-tiled GEMM with arbitrary
 * element strides serves every contraction (x W^T, dy W, dy^T x, Q K^T, P V and their gradients), dropout is not implemented (parity runs
 * need dropout 0: the reference's Philox stream cannot be matched), sequences are not packed.
 *
 * Conventions as in emdr2_ops.h: raw device pointers, element strides, caller-owned buffers, `stream` a hipStream_t, int status
 * (0 ok, -1 bad argument, -3 launch error); nothing is allocated inside.
 */
#ifndef EMDR2_OPS_F32_H
#define EMDR2_OPS_F32_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* C[b1][b2][m][n] = alpha * sum_k A[b1][b2][m][k] * B[b1][b2][n][k]  (+ bias[n]) (+ residual[m][n], same addressing as C) (+ C if accumulate)
 * with ELEMENT strides: X[b1][b2][r][c] at X + b1 * x_b1 + b2 * x_b2 + r * x_rs + c * x_cs.  Replaces F.linear / torch.matmul / baddbmm / bmm of
 * the reference's fp32 path (mpu/layers.py:255,353; transformer.py:283-381) and autograd's products of them. */
int emdr2_f32_gemm(const float *A, int64_t a_rs, int64_t a_cs, int64_t a_b1, int64_t a_b2,
                   const float *B, int64_t b_rs, int64_t b_cs, int64_t b_b1, int64_t b_b2,
                   float *C, int64_t c_rs, int64_t c_cs, int64_t c_b1, int64_t c_b2,
                   int M, int N, int K, int batch1, int batch2, float alpha, const float *bias, const float *residual, int accumulate,
                   void *stream);

/* torch.nn.LayerNorm (mpu/layers.py:28-36 fallback) over rows of H; mean / rstd kept for the backward.  bwd: dx (+= dres when given: the
 * gradient of the residual branch that by-passed the LayerNorm), dgamma / dbeta ACCUMULATED (atomics) into zero-initialised fp32 [H]. */
int emdr2_f32_layernorm_fwd(const float *x, const float *gamma, const float *beta, float *y, float *mean, float *rstd, int64_t rows, int H,
                            float eps, void *stream);
int emdr2_f32_layernorm_bwd(const float *dy, const float *x, const float *gamma, const float *mean, const float *rstd, const float *dres,
                            float *dx, float *dgamma, float *dbeta, int64_t rows, int H, void *stream);

/* In place over scores [batch, heads, sq, sk] (already scaled): masked_fill(mask, -10000) then softmax over sk, mask from the token ids as
 * the reference builds it (pad id 0 on either side: megatron/data/mask_creation_utils.py:17-26; `causal`: the history mask).
 * bwd, in place over dprobs: dscores = probs o (dprobs - rowsum(probs o dprobs)), zero at masked positions (masked_fill cuts the dependence). */
int emdr2_f32_softmax_mask_fwd(float *scores, const int64_t *ids_q, const int64_t *ids_k, int batch, int heads, int sq, int sk, int causal,
                               void *stream);
int emdr2_f32_softmax_mask_bwd(const float *probs, float *dprobs, const int64_t *ids_q, const int64_t *ids_k, int batch, int heads, int sq,
                               int sk, int causal, void *stream);

/* exact-erf GELU (transformer.py:94-108 with bias_gelu_fusion off: F.gelu) and its derivative: dx = dy * gelu'(x). */
int emdr2_f32_gelu_fwd(const float *x, float *y, int64_t n, void *stream);
int emdr2_f32_gelu_bwd(const float *x, const float *dy, float *dx, int64_t n, void *stream);

/* Embedding.forward (language_model.py:169-181): out[t] = W[ids[t]] + P[t % S] + (T ? T[types[t]] : 0); bwd scatters dout into dW / dP / dT
 * (atomics; zero-initialised or holding earlier contributions). */
int emdr2_f32_embedding_fwd(const int64_t *ids, const int64_t *types, const float *W, const float *P, const float *T, float *out, int64_t tokens,
                            int S, int H, void *stream);
int emdr2_f32_embedding_bwd(const int64_t *ids, const int64_t *types, const float *dout, float *dW, float *dP, float *dT, int64_t tokens, int S,
                            int H, void *stream);

/* gold[r] = log_softmax(logits[r])[labels[r]], lse[r] = logsumexp(logits[r]) (train_e2eqa.py:79-96,152-160);
 * bwd: dlogits[r][v] = w[r] * (onehot(labels[r])[v] - softmax(logits[r])[v]). */
int emdr2_f32_lse_gather_fwd(const float *logits, const int64_t *labels, float *gold, float *lse, int64_t rows, int V, void *stream);
int emdr2_f32_lse_gather_bwd(const float *logits, const int64_t *labels, const float *lse, const float *w, float *dlogits, int64_t rows, int V,
                             void *stream);

/* Fresh retriever scores -> prior (emdr2_model.py:134-145): logp[b] = log_softmax_k(scale * q[b] . c[b][k]); prob = exp(logp) kept for bwd.
 * bwd: ds = dlogp - prob * sum_k dlogp; dq[b] = scale * sum_k ds[k] c[b][k]; dc[b][k] = scale * ds[k] q[b]. */
int emdr2_f32_retriever_prior_fwd(const float *q, const float *c, float *logp, float *prob, int batch, int K, int H, float scale, void *stream);
int emdr2_f32_retriever_prior_bwd(const float *dlogp, const float *prob, const float *q, const float *c, float *dq, float *dc, int batch, int K,
                                  int H, float scale, void *stream);

#ifdef __cplusplus
}
#endif
#endif
