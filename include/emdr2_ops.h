/*
 * include/emdr2_ops.h -- C ABI of the transformer-block kernels (libemdr2_hip.so): what the encoder / reader
 * forward + backward of the EMDR2 hot path are made of.
 *
 * The reference runs these through torch / cuBLAS / apex from Python; the entry points below are what a binding
 * next to megatron/model/transformer.py would call.  Each comment names the reference code it replaces.
 * Conventions as in emdr2_mips.h: device pointers, element strides, int status (0 ok), enqueue on `stream`.
 * Activations and weights are bf16 (round-to-nearest-even), accumulation and statistics fp32.
 */
#ifndef EMDR2_OPS_H
#define EMDR2_OPS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/*
 * C[b1,b2][m,n] = epi( alpha * sum_k A[b1,b2][m,k] * B[b1,b2][n,k] ),  epi = (+bias[n]) -> (GELU) -> (dropout) -> (+residual[m,n])
 * Replaces F.linear (mpu/layers.py:255,353), baddbmm/bmm (transformer.py:309-312,371), bias+GELU
 * (transformer.py:103-104; exact erf), bias-dropout-add (transformer.py:397-413).
 * drop_p > 0 (unbatched, no split-K): element (m, n) is kept iff keep(seed, row m, column n) of csrc/rng.h (drop_p quantised to 2^-16);
 * survivors are scaled by 1/(1-p); emdr2_dropout() applies the same mask to the incoming gradient in the backward.
 * K % 32 == 0; lda, ldb and the A/B batch strides multiples of 8 elements; A, B 16-byte aligned.
 * pre_act (optional, bf16, indexed like C): value before GELU, kept for the backward.  gelu == 2 (needs pre_act): GELU as with 1, but
 * pre_act receives gelu'(value before GELU) instead -- the only thing the backward ever does with the pre-activation, computed here
 * from the fp32 value and next to the activation's own exponential (residual_mode 2 consumes it).
 * residual_mode 0: + residual (`residual` may be C itself: every element is read before it is written, by the same lane -- a data gradient
 * added onto a running sum in place).  1: the `residual` buffer holds a saved GELU pre-activation u and the result is MULTIPLIED by gelu'(u) --
 * the backward of bias-GELU fused into the GEMM that produces d(activation) (saves one 3 x [tokens, ffn] elementwise pass).
 * 2: the result is MULTIPLIED by residual[m, n] (a gelu' saved by gelu == 2: the same backward without the erf arithmetic).
 * split_k > 1: the reduction is cut into split_k slices accumulated with fp32 atomics into a PRE-ZEROED fp32 C (weight
 * gradients: few output tiles, very long K); no epilogue options in that mode.
 */
int emdr2_gemm_nt_bf16(const void *A, int64_t lda, const void *B, int64_t ldb, void *C, int64_t ldc, int M, int N, int K,
                       int batch1, int64_t sA1, int64_t sB1, int64_t sC1, int batch2, int64_t sA2, int64_t sB2, int64_t sC2,
                       float alpha, const float *bias, int gelu, void *pre_act, const void *residual, int residual_mode, int out_f32,
                       int split_k, float drop_p, uint32_t seed, void *stream);

/* Weight-gradient GEMM, "TN" form: C[i, j] (fp32) = sum_r A[r, i] * B[r, j], A [R, I] and B [R, J] row-major bf16 (dW = dy^T x without
 * transposing either operand through HBM: LDS transpose reads, see csrc/gemm_tn.hip).  Replaces the autograd weight gradient of F.linear
 * (mpu/layers.py:255,353).  R % 32 == 0, I, J, lda, ldb multiples of 8.  colsum_a (optional, fp32 [I]) accumulates the column sums
 * of A (the bias gradient).  split_k > 1 accumulates reduction slices with atomics into a PRE-ZEROED C. */
int emdr2_gemm_tn_bf16(const void *A, int64_t lda, const void *B, int64_t ldb, float *C, int64_t ldc, int I, int J, int R, int split_k,
                       float *colsum_a, void *stream);

/* out[r, c] = keep(seed, r, c) ? x[r, c] / (1 - p) : 0 over a contiguous bf16 [n / cols, cols] tensor (the mask of a dropout site,
 * regenerated; cols % 8 == 0). */
int emdr2_dropout(const void *x, void *out, int64_t n, int cols, float drop_p, uint32_t seed, void *stream);

/* out[b1,b2][c, r] = in[b1,b2][r, c] (bf16), optional fp32 column sums colsum[c] += sum_r in[r, c] over all batches
 * (bias gradients: the reduce of the reference's autograd over [s, b]). */
int emdr2_transpose_bf16(const void *in, int64_t ld_in, void *out, int64_t ld_out, int rows, int cols,
                         int batch1, int64_t sI1, int64_t sO1, int batch2, int64_t sI2, int64_t sO2, float *colsum,
                         void *stream);

/* LayerNorm over the last dim (torch.nn.LayerNorm / apex FusedLayerNorm, mpu/layers.py:28-36; eps 1e-5, arguments.py:199).
 * x, y bf16 [rows, H]; gamma, beta, mean, rstd fp32.  bwd: dx (+ dres, the residual-branch gradient, optional), dgamma / dbeta
 * ACCUMULATED into fp32 buffers. */
int emdr2_layernorm_fwd(const void *x, const float *gamma, const float *beta, void *y, float *mean, float *rstd, int64_t rows, int H,
                        float eps, void *stream);
int emdr2_layernorm_bwd(const void *dy, const void *x, const float *gamma, const float *mean, const float *rstd, const void *dres, void *dx,
                        float *dgamma, float *dbeta, int64_t rows, int H, void *stream);
/* The same, ALSO writing dmask = dx * keep(seed, row, col) / (1 - drop_p) (bf16 [rows, H], the bits of emdr2_dropout(dx, ...)): the operand of the
 * backward GEMMs of the bias-dropout-add (transformer.py:397-413) that produced x.  H == 768 only; -4 otherwise. */
int emdr2_layernorm_bwd_mask(const void *dy, const void *x, const float *gamma, const float *mean, const float *rstd, const void *dres, void *dx,
                             float *dgamma, float *dbeta, int64_t rows, int H, void *dmask, float drop_p, uint32_t seed, void *stream);

/* Scale-mask-softmax, the path the shipped scripts run (fused_softmax.py:113-125): masked scores are REPLACED by -10000
 * (bert/t5_attention_mask_func), masks derived on the fly from token ids (pad id 0; mask_creation_utils.py:17-42),
 * `causal` adds the history mask.  scores bf16 [batch, heads, sq, sk] in place; m, l: row max / sum-exp (fp32 [batch, heads, sq]).
 * drop_p > 0: attention dropout (transformer.py:262,366) applied to the stored probabilities AFTER normalisation (m, l are those of the
 * un-dropped softmax); keep bit = keep(seed, row, k) of csrc/rng.h, row = (b * heads + n) * sq + q.
 * _bwd: dprobs -> dscores in place, d = rowsum(P*dP_eff).  With m, l given, `probs` holds the raw scaled scores and P is rebuilt from
 * the statistics (the forward need not keep the [sq, sk] probabilities); dropout requires that form.
 * _t: transposed twin for the attention backward: S^T -> dropped P^T, dP^T -> dS^T (see elementwise.hip). */
int emdr2_softmax_mask_fwd(void *scores, const int64_t *ids_q, const int64_t *ids_k, int batch, int heads, int sq, int sk, int causal,
                           float *m, float *l, float drop_p, uint32_t seed, void *stream);
int emdr2_softmax_mask_bwd(const void *probs, void *dprobs, const int64_t *ids_q, const int64_t *ids_k, const float *m, const float *l,
                           int batch, int heads, int sq, int sk, int causal, float drop_p, uint32_t seed, float *d, void *stream);
int emdr2_softmax_mask_t(void *scores_t, void *dprobs_t, const int64_t *ids_q, const int64_t *ids_k, const float *m, const float *l,
                         const float *d, int batch, int heads, int sq, int sk, int causal, float drop_p, uint32_t seed, void *stream);

/* Fused attention forward for head_dim 64, sk % 32 == 0 (transformer.py:283-381 without the [sq, sk] score matrix): o = dropout(softmax(mask(q k^T
 * scale))) v.  q [b, sq, heads, 64] / k [b, sk, heads, 64] strided views (element strides: batch, sequence, head; last dim contiguous),
 * v likewise, o [b, sq, heads, 64] contiguous, masks from token ids (pad id 0) + causal, masked scores REPLACED by -10000.
 * m / l (row max and row sum of exp, fp32 [b, heads, sq]) feed the backward.  Dropout keep-bit = keep(seed, row (b*heads+n)*sq+q,
 * column key) of csrc/rng.h (same generator as emdr2_dropout / emdr2_softmax_mask_*).  Returns -4 for shapes it does not cover. */
int emdr2_attention_fwd(const void *q, int64_t q_sb, int64_t q_ss, int64_t q_sn, const void *k, int64_t k_sb, int64_t k_ss, int64_t k_sn,
                        const void *v, int64_t v_sb, int64_t v_ss, int64_t v_sn, void *o, const int64_t *ids_q, const int64_t *ids_k, int batch, int heads, int sq, int sk, int head_dim,
                        int causal, float scale, float drop_p, uint32_t seed, float *m, float *l, void *stream);

/* Fused attention backward for the same shapes (csrc/attention_bwd.hip): dq, dk, dv [b, s, heads, 64] with caller-given batch / sequence element
 * strides (heads 64 apart: they may be slices of one packed QKV gradient) from q, k, v (strided views),
 * the forward output o and its gradient dout (contiguous), and the forward's m, l.  dstat (fp32 [4, b, heads, sq], caller-owned scratch;
 * ABI 2: four times the ABI-1 size) receives what the dq kernel leaves for the dk / dv kernel per query: the exp2 offset
 * m log2(e) + log2(l), D = rowsum(dout * o), the dropout row hash and the real-token flag (the last two as bit patterns).
 * Same masks, scale and dropout stream as the forward. */
int emdr2_attention_bwd(const void *q, int64_t q_sb, int64_t q_ss, int64_t q_sn, const void *k, int64_t k_sb, int64_t k_ss, int64_t k_sn,
                        const void *v, int64_t v_sb, int64_t v_ss, int64_t v_sn, const void *o, const void *dout, void *dq, int64_t dq_sb, int64_t dq_ss,
                        void *dk, void *dv, int64_t dkv_sb, int64_t dkv_ss,
                        const int64_t *ids_q, const int64_t *ids_k, const float *m, const float *l, float *dstat, int batch, int heads, int sq,
                        int sk, int head_dim, int causal, float scale, float drop_p, uint32_t seed, void *stream);

/* ---- packed ("varlen") sequences: the encoder stacks without their [PAD] rows (csrc/seqpack.hip) -------------------------------------
 * The reference runs [batch, S] token grids through every layer (transformer.py:283-381, emdr2_model.py:148-210); a padded key adds
 * exp(-10000 - max) == 0 to every real query and a padded query's row is never consumed, so dropping those rows changes no consumed value.
 * Layout: sequence i keeps its first len[i] tokens, len[i] = 1 + index of its last non-pad (id != 0) token (S for an all-pad row), and
 * owns rows [cu[i], cu[i+1]) of a [rows, ...] tensor; cu = exclusive prefix sum of len (int32 [n+1]).
 *
 * emdr2_seq_lengths: cu from ids [n, S]; totals[0] = cu[n], totals[1] = sum len^2, totals[2] = max len (int64 [3], device).  n <= 38000 (the lengths are scanned in LDS).
 * emdr2_seq_pack_ids: rowmap[t] (int32 [rows_padded]) = dense row i*S+pos of packed row t (-1 for t >= total: tail rows up to a GEMM-friendly
 *   multiple), inverse[i*S+pos] (int32 [n*S]) = packed row or -1, ids_packed / types_packed (int64 [rows_padded], 0 in the tail).
 * emdr2_gather_rows: out[r, :] = map[r] >= 0 ? in[map[r], :] : 0 (bf16 rows of H, H % 8 == 0) -- dense -> packed with rowmap, packed -> dense
 *   (zeros at pad rows) with inverse, token-0 rows of a packed tensor with cu.  emdr2_scatter_rows: out[map[r], :] = in[r, :] (map[r] >= 0). */
int emdr2_seq_lengths(const int64_t *ids, int n, int S, int32_t *cu, int64_t *totals, void *stream);
int emdr2_seq_pack_ids(const int64_t *ids, const int64_t *types, const int32_t *cu, int n, int S, int64_t total, int64_t rows_padded,
                       int32_t *rowmap, int32_t *inverse, int64_t *ids_packed, int64_t *types_packed, void *stream);
int emdr2_gather_rows(const void *in, const int32_t *map, void *out, int64_t rows_out, int H, void *stream);
int emdr2_scatter_rows(const void *in, const int32_t *map, void *out, int64_t rows_in, int H, void *stream);

/* Embedding.forward / backward over packed rows (language_model.py:169-181): position = rowmap[t] % S; tail rows (rowmap < 0) are zeros.
 * bwd: the position sums walk the sequences through cu (position p exists in sequence i iff p < len[i]). */
int emdr2_embedding_packed_fwd(const int64_t *ids_packed, const int64_t *types_packed, const int32_t *rowmap, const void *W, const void *P, const void *T,
                               void *out, int64_t rows, int S, int H, float drop_p, uint32_t seed, void *stream);
int emdr2_embedding_packed_bwd(const int64_t *ids_packed, const int64_t *types_packed, const int32_t *cu, int nseq, const void *dout, float *dW, float *dP,
                               float *dT, int64_t rows, int S, int H, int n_types, float drop_p, uint32_t seed, void *stream);

/* The fused attention kernels over packed operands (same kernels as emdr2_attention_fwd / _bwd): cu_q and / or cu_k (int32 [batch+1], either
 * may be NULL = that side is dense [batch, s] with its batch stride) give sequence b's rows [cu[b], cu[b+1]) of q / o / dq (k, v / dk, dv);
 * ids_q / ids_k are indexed the same way (packed ids for a packed side).  max_sq / max_sk: the longest sequence (grid size; dense side: s);
 * a packed key side takes ANY lengths >= 1 (keys past the end are masked).  With cu_q the row statistics m, l are [heads, total_q] and dstat is [4, heads, total_q].
 * pairs = sum_b sq_b * sk_b (flop accounting of the timing hooks only).  Replaces transformer.py:283-381 for the packed stacks and the
 * FiD cross-attention of emdr2_model.py:166-183 over the packed encoder output (cu_k = every K-th entry of the encoder's cu). */
int emdr2_attention_varlen_fwd(const void *q, int64_t q_sb, int64_t q_ss, int64_t q_sn, const void *k, int64_t k_sb, int64_t k_ss, int64_t k_sn,
                               const void *v, int64_t v_sb, int64_t v_ss, int64_t v_sn, void *o, const int64_t *ids_q, const int64_t *ids_k,
                               const int32_t *cu_q, const int32_t *cu_k, int64_t total_q, int64_t pairs, int batch, int heads, int max_sq, int max_sk,
                               int head_dim, int causal, float scale, float drop_p, uint32_t seed, float *m, float *l, void *stream);
int emdr2_attention_varlen_bwd(const void *q, int64_t q_sb, int64_t q_ss, int64_t q_sn, const void *k, int64_t k_sb, int64_t k_ss, int64_t k_sn,
                               const void *v, int64_t v_sb, int64_t v_ss, int64_t v_sn, const void *o, const void *dout, void *dq, int64_t dq_sb,
                               int64_t dq_ss, void *dk, void *dv, int64_t dkv_sb, int64_t dkv_ss, const int64_t *ids_q, const int64_t *ids_k,
                               const int32_t *cu_q, const int32_t *cu_k, int64_t total_q, int64_t pairs, const float *m, const float *l, float *dstat,
                               int batch, int heads, int max_sq, int max_sk, int head_dim, int causal, float scale, float drop_p, uint32_t seed, void *stream);

/* Split-key launches of the same kernels for FEW queries over VERY MANY keys: the FiD decoder's cross-attention (emdr2_model.py:166-183:
 * batch questions x 32 decoder positions over the ~20,000 / 40,000 packed encoder tokens of a question's 50 / 100 passages) and the cached
 * decoding steps of search_strategy.py:185-240.  One workgroup per (batch, head) walking every key block leaves most of the chip idle when
 * batch x heads is a few hundred; here the key blocks of a (batch, head) are dealt, 32 blocks (2,048 keys) each, to `ksplit` workgroups whose
 * fp32 partials (forward: unnormalised O + (m, l); backward: dQ) go through the caller's workspace and are folded by a small second kernel,
 * in split order (deterministic).  Queries are DENSE [batch, sq <= 128] (q_sb / dq_sb strides); cu_k as above or NULL for dense keys.  Same
 * results as the unsplit launch up to the order of fp32 additions (the softmax reference points are whole binades, so the bf16
 * probabilities do not depend on where a walk over the keys starts).
 *   _plan: ksplit for these extents (1 = do not split: more than 128 queries or fewer than 4,096 keys; never a function of the batch, so a
 *   question's result does not depend on what it is batched with) and the workspace bytes of the two launches (16-byte aligned ws). */
int emdr2_attention_splitkv_plan(int batch, int heads, int max_sq, int max_sk, int *ksplit, size_t *fwd_bytes, size_t *bwd_bytes);
int emdr2_attention_fwd_splitkv(const void *q, int64_t q_sb, int64_t q_ss, int64_t q_sn, const void *k, int64_t k_sb, int64_t k_ss, int64_t k_sn,
                                const void *v, int64_t v_sb, int64_t v_ss, int64_t v_sn, void *o, const int64_t *ids_q, const int64_t *ids_k,
                                const int32_t *cu_k, int64_t pairs, int batch, int heads, int sq, int max_sk, int head_dim, int causal, float scale,
                                float drop_p, uint32_t seed, float *m, float *l, int ksplit, void *ws, size_t ws_bytes, void *stream);
int emdr2_attention_bwd_splitkv(const void *q, int64_t q_sb, int64_t q_ss, int64_t q_sn, const void *k, int64_t k_sb, int64_t k_ss, int64_t k_sn,
                                const void *v, int64_t v_sb, int64_t v_ss, int64_t v_sn, const void *o, const void *dout, void *dq, int64_t dq_sb,
                                int64_t dq_ss, void *dk, void *dv, int64_t dkv_sb, int64_t dkv_ss, const int64_t *ids_q, const int64_t *ids_k,
                                const int32_t *cu_k, int64_t pairs, const float *m, const float *l, float *dstat, int batch, int heads, int sq,
                                int max_sk, int head_dim, int causal, float scale, float drop_p, uint32_t seed, int ksplit, void *ws, size_t ws_bytes,
                                void *stream);

/* dpre = dact * gelu'(pre), exact-erf GELU (transformer.py:80,103-104; the tanh fusion of fused_bias_gelu.py is off in all scripts) */
int emdr2_gelu_bwd(const void *pre, const void *dact, void *dpre, int64_t n, void *stream);

/* Embedding.forward (language_model.py:169-181): out[t] = dropout(W[ids[t]] + P[t % S] (+ T[types[t]])); bwd scatter-adds the (masked)
 * gradient into fp32 grads (word rows: atomics; position rows: one owner thread per (position, column), no atomics; the <= 4 type rows:
 * register partials, one atomic per (position, column)).  tokens % S == 0.  Dropout element = (row t, column i). */
int emdr2_embedding_fwd(const int64_t *ids, const int64_t *types, const void *W, const void *P, const void *T, void *out, int64_t tokens, int S,
                        int H, float drop_p, uint32_t seed, void *stream);
int emdr2_embedding_bwd(const int64_t *ids, const int64_t *types, const void *dout, float *dW, float *dP, float *dT, int64_t tokens, int S, int H,
                        int n_types, float drop_p, uint32_t seed, void *stream);

/* log-softmax over the vocabulary + gather of the gold token (train_e2eqa.py:79-96,152-160): gold[row] = logits[row, label] - lse */
int emdr2_lse_gather_fwd(const void *logits, const int64_t *labels, float *gold, float *lse, int64_t rows, int V, void *stream);
int emdr2_lse_gather_bwd(const void *logits, const int64_t *labels, const float *lse, const float *w, void *dlogits, int64_t rows, int V,
                         void *stream);

/* Optimizer step on fp32 masters (FP16_Optimizer + apex FusedAdam, training.py:89-93, fp16/fp16.py:420-474): global-norm clip
 * (mpu/grads.py:74-127) folded in through gnorm_sq; decoupled weight decay; writes the bf16 working copy. */
/* *out += sum g[i]^2, deterministically (block partials summed in index order by the last block): data-parallel replicas derive
 * bit-identical clip factors.  scratch: 1025 floats, zero-initialised once by the caller (1024 partials + a counter the kernel resets). */
int emdr2_sumsq_f32(const float *g, int64_t n, float *out, float *scratch, void *stream);
int emdr2_adam_step(float *master, const float *grad, float *m, float *v, void *param_bf16, int64_t n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, int step, const float *gnorm_sq, float clip, void *stream);
/* The same update over one FLAT bucket of parameters (masters, gradients, both moments and the bf16 working copies of many tensors stored back
 * to back; apex runs FusedAdam through amp_C.multi_tensor_apply for the same reason, fp16/fp16.py:420-474): elements [0, decay_split) take
 * the weight decay, the rest none.  n, decay_split multiples of 4; fp32 buffers 16-byte aligned. */
int emdr2_adam_step_flat(float *master, const float *grad, float *m, float *v, void *work_bf16, int64_t n, int64_t decay_split, float lr,
                         float beta1, float beta2, float eps, float weight_decay, int step, const float *gnorm_sq, float clip, void *stream);
/* Gradient exchange in 16 bits (the reference all-reduces fp16 gradients pre-divided by the world size, model/distributed.py:53-62):
 * dst = bf16(scale * src) before the all-reduce, dst = float(src) after it.  n % 4 == 0. */
int emdr2_scale_cast_f32_to_bf16(const float *src, void *dst, int64_t n, float scale, void *stream);
int emdr2_widen_bf16_to_f32(const void *src, float *dst, int64_t n, void *stream);
int emdr2_cast_f32_to_bf16(const float *src, void *dst, int64_t n, void *stream);
int emdr2_accum_bf16_to_f32(const void *src, float *dst, int64_t n, float scale, void *stream);

/* C[m, n-tile partials]: the tied LM head without the logits (language_model.py:28-41 + the log-softmax / gather of
 * train_e2eqa.py:79-96).  For x[m, n] = bf16(alpha * sum_k A[m,k] B[n,k] + bias[n]) -- the value the unfused path would have stored -- writes per
 * row m and per 64-column block j (N / 64 of them): part_max[m, j] = max_n x, part_sum[m, j] = sum_n exp(x - part_max), and gold[m] =
 * x[m, labels[m]].  emdr2_lse_combine turns them into gold - logsumexp.  M % 256 == 0, N % 256 == 0, K % 128 == 0. */
int emdr2_gemm_nt_lse_bf16(const void *A, int64_t lda, const void *B, int64_t ldb, int M, int N, int K, float alpha, const float *bias,
                           const int64_t *labels, float *part_max, float *part_sum, float *gold, void *stream);

/* out[row] = gold[row] - logsumexp(row) from the per-block partials of emdr2_gemm_nt_lse_bf16 (slots = N / 64); lse optional. */
int emdr2_lse_combine(const float *part_max, const float *part_sum, const float *gold, float *out, float *lse, int64_t rows, int slots,
                      void *stream);

/* Retriever prior over the K retrieved passages (emdr2_model.py:134-145): logp[b, k] = log_softmax_k( <q[b], c[b, k]> * scale ), scale =
 * 1/sqrt(H) with --retriever-score-scaling; q bf16 [batch, H], c bf16 [batch, K, H], logp / prob (= exp(logp), kept for the backward) fp32
 * [batch, K]; K <= 1024 (-4 above).  bwd: bf16 gradients dq [batch, H], dc [batch, K, H] (either may be NULL: --no-query/context-embedder-training). */
int emdr2_retriever_prior_fwd(const void *q, const void *c, float *logp, float *prob, int batch, int K, int H, float scale, void *stream);
int emdr2_retriever_prior_bwd(const float *dlogp, const float *prob, const void *q, const void *c, void *dq, void *dc, int batch, int K, int H,
                              float scale, void *stream);
/* EMDR2 marginal likelihood (train_e2eqa.py:98-123): marginal[b, l] = logsumexp_k( prior[b, k] + gold[b, k, l] ), gold = per-passage gold
 * log-likelihoods of the no-grad one-context pass (constants).  bwd: dprior[b, k] = sum_l dmarginal[b, l] * posterior[b, k, l].  All fp32. */
int emdr2_marginal_fwd(const float *prior, const float *gold, float *marginal, int batch, int K, int L, void *stream);
int emdr2_marginal_bwd(const float *prior, const float *gold, const float *marginal, const float *dmarginal, float *dprior, int batch, int K, int L,
                       void *stream);

/* Measurement hooks (bench.py): with timing on, every GEMM / attention launch of this library is bracketed by hipEvents recorded on its launch
 * stream.  collect() waits for them and returns, per kind (0 NT GEMM, 1 TN GEMM, 2 attention forward, 3 attention backward), the summed
 * milliseconds, the summed algorithmic flops and the number of launches since the last set_timing / collect.  kinds >= 4. */
int emdr2_ops_set_timing(int enabled);
int emdr2_ops_timing_collect(double *ms, double *flops, int64_t *launches, int kinds);

#ifdef __cplusplus
}
#endif
#endif
