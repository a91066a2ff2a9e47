"""The VALIDATION-ONLY fp32 compute path (include/emdr2_ops_f32.h, csrc/fp32_ops.hip): the autograd glue of `kernels.py` once more, over
fp32 activations.  `kernels.linear / mlp / layer_norm / attention_core / embedding / lse_gather / retriever_prior / lm_head_gold_logprob`
dispatch here when their activation argument is fp32 (`transformer.Config(compute_dtype="fp32")` makes the embeddings emit fp32; everything
downstream follows the dtype), so the module tree, the parameter names, the loss code and the gradient-delivery protocol
(`kernels._accum_grad`: fp32 gradients to the parameter or to the flat-bucket sink) are the product's own.

What the reference does in this mode: without --fp16 the model is not wrapped in FP16_Module and every op runs in fp32
(megatron/training.py:55-56,92; transformer.py:283-381 with fp32 baddbmm / softmax / bmm).  What this mode is for: north_star's "reader logits
within 1e-3 fp32" (tests/test_parity_fp32_gpu.py).  What it is not: fast (one simple MFMA-fp32 GEMM kernel for every contraction, dense
[b, heads, sq, sk] score matrices, no sequence packing), or complete beyond that purpose -- dropout > 0 and incremental decoding raise.
Parameters are used as they are (fp32 masters; no working copies); QKV / KV weights are de-interleaved by a row gather like the bf16
path's `w_bf16_perm`."""
import math

import torch

from emdr2_amd import _native
from emdr2_amd.model import kernels as K

F32 = torch.float32


def _lib():
    return _native.lib()


def _sp():
    return _native.stream_ptr()


def _check(*ts):
    for t in ts:
        if t is not None and (t.dtype != F32 or not t.is_cuda):
            raise TypeError("expected a CUDA float32 tensor (fp32 validation path)")


def _no_dropout(p):
    if p:
        raise ValueError("the fp32 validation path has no dropout (parity runs need dropout 0: --hidden-dropout 0 --attention-dropout 0 or .eval())")


def gemm(A, a_rs, a_cs, B, b_rs, b_cs, C, c_rs, c_cs, M, N, Kd, batch1=1, a_b1=0, b_b1=0, c_b1=0, batch2=1, a_b2=0, b_b2=0, c_b2=0, alpha=1.0,
         bias=None, residual=None, accumulate=False):
    """C[m, n] (+)= alpha * sum_k A[m, k] B[n, k] (+ bias[n]) (+ residual[m, n]) with element strides (include/emdr2_ops_f32.h)."""
    _native.check(_lib().emdr2_f32_gemm(A.data_ptr(), a_rs, a_cs, a_b1, a_b2, B.data_ptr(), b_rs, b_cs, b_b1, b_b2, C.data_ptr(), c_rs, c_cs, c_b1, c_b2,
                                        M, N, Kd, batch1, batch2, float(alpha), K._ptr(bias), K._ptr(residual), int(bool(accumulate)), _sp()), "f32_gemm")
    return C


def _w(p, perm=None):
    """The fp32 master itself, or its row-gathered (de-interleaved) form, cached on the parameter like the bf16 path's derived forms."""
    if perm is None:
        return p.detach()
    return K.WEIGHTS.get(p, "perm_f32", lambda: p.detach()[perm].contiguous())


def _param_grads(dy2, x2, weight, bias, perm):
    """dW [N, K] = dy^T x, db = column sums of dy, delivered in the checkpoint's row order."""
    M, N = dy2.shape
    Kd = x2.shape[1]
    dW = torch.empty((N, Kd), dtype=F32, device=dy2.device)
    gemm(dy2, 1, N, x2, 1, Kd, dW, Kd, 1, N, Kd, M)                                   # A[n, m] = dy[m, n], B[k, m] = x[m, k]
    db = None
    if bias is not None:
        ones = torch.ones((1, M), dtype=F32, device=dy2.device)
        db = torch.empty((N,), dtype=F32, device=dy2.device)
        gemm(dy2, 1, N, ones, 0, 1, db, 1, 0, N, 1, M)                                # db[n] = sum_m dy[m, n]
    if perm is not None:
        un = torch.empty_like(dW); un[perm] = dW; dW = un
        if db is not None:
            ub = torch.empty_like(db); ub[perm] = db; db = ub
    K._accum_grad(weight, dW)
    if bias is not None:
        K._accum_grad(bias, db)


class LinearFn(torch.autograd.Function):
    """y = x W^T + b (+ exact-erf GELU) (+ residual): F.linear + bias(+gelu) / bias-dropout-add at p = 0 (mpu/layers.py:255,353,
    transformer.py:94-108,397-407), fp32."""

    @staticmethod
    def forward(ctx, x, weight, bias, gelu, residual, row_perm):
        _check(x, residual)
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).contiguous()
        M, Kd = x2.shape
        N = weight.shape[0]
        w = _w(weight, row_perm)
        b = None if bias is None else _w(bias, row_perm)
        res2 = residual.reshape(M, N).contiguous() if residual is not None else None
        y = torch.empty((M, N), dtype=F32, device=x.device)
        pre = None
        if gelu:
            pre = torch.empty_like(y)
            gemm(x2, Kd, 1, w, Kd, 1, pre, N, 1, M, N, Kd, bias=b)
            _native.check(_lib().emdr2_f32_gelu_fwd(pre.data_ptr(), y.data_ptr(), y.numel(), _sp()), "f32_gelu_fwd")
            if res2 is not None:
                raise ValueError("gelu with a residual is not a shape of the model")
        else:
            gemm(x2, Kd, 1, w, Kd, 1, y, N, 1, M, N, Kd, bias=b, residual=res2)
        ctx.save_for_backward(x2, pre)
        ctx.weight, ctx.bias, ctx.has_res, ctx.shp, ctx.row_perm = weight, bias, residual is not None, shp, row_perm
        return y.reshape(shp[:-1] + (N,))

    @staticmethod
    def backward(ctx, dy):
        x2, pre = ctx.saved_tensors
        weight, bias = ctx.weight, ctx.bias
        M, Kd = x2.shape
        N = weight.shape[0]
        dy2 = dy.reshape(M, N).contiguous()
        dres = dy if ctx.has_res else None
        if pre is not None:
            dpre = torch.empty_like(dy2)
            _native.check(_lib().emdr2_f32_gelu_bwd(pre.data_ptr(), dy2.data_ptr(), dpre.data_ptr(), dy2.numel(), _sp()), "f32_gelu_bwd")
            dy2 = dpre
        dx = None
        if ctx.needs_input_grad[0]:
            w = _w(weight, ctx.row_perm)
            dx = torch.empty((M, Kd), dtype=F32, device=dy.device)
            gemm(dy2, N, 1, w, 1, Kd, dx, Kd, 1, M, Kd, N)                            # dx[m, k] = sum_n dy[m, n] W[n, k]
            dx = dx.reshape(ctx.shp)
        if weight.requires_grad:
            _param_grads(dy2, x2, weight, bias, ctx.row_perm)
        return dx, None, None, None, dres, None


def linear(x, weight, bias=None, gelu=False, residual=None, row_perm=None, drop_p=0.0, seed=0):
    _no_dropout(drop_p)
    return LinearFn.apply(x, weight, bias, gelu, residual, row_perm)


def mlp(x, w1, b1, w2, b2, residual, drop_p=0.0, seed=0):
    """ParallelMLP + bias-dropout-add at p = 0 (transformer.py:58-108,397-413) as two fp32 linears."""
    _no_dropout(drop_p)
    return linear(linear(x, w1, b1, gelu=True), w2, b2, residual=residual)


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, passthrough):
        _check(x)
        H = x.shape[-1]
        x2 = x.reshape(-1, H).contiguous()
        rows = x2.shape[0]
        y = torch.empty_like(x2)
        mean = torch.empty(rows, dtype=F32, device=x.device)
        rstd = torch.empty_like(mean)
        _native.check(_lib().emdr2_f32_layernorm_fwd(x2.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                     rows, H, float(eps), _sp()), "f32_layernorm_fwd")
        ctx.save_for_backward(x2, mean, rstd)
        ctx.gamma, ctx.beta = gamma, beta
        if passthrough:
            return y.reshape(x.shape), x.view_as(x)
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy, dpass=None):
        x2, mean, rstd = ctx.saved_tensors
        gamma, beta = ctx.gamma, ctx.beta
        rows, H = x2.shape
        dy2 = dy.reshape(rows, H).contiguous()
        dres = dpass.reshape(rows, H).contiguous() if dpass is not None else None
        dx = torch.empty_like(x2)
        dg, db = torch.zeros(H, dtype=F32, device=dy.device), torch.zeros(H, dtype=F32, device=dy.device)
        _native.check(_lib().emdr2_f32_layernorm_bwd(dy2.data_ptr(), x2.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), K._ptr(dres),
                                                     dx.data_ptr(), dg.data_ptr(), db.data_ptr(), rows, H, _sp()), "f32_layernorm_bwd")
        K._accum_grad(gamma, dg)
        K._accum_grad(beta, db)
        return dx.reshape(dy.shape), None, None, None, None


def layer_norm(x, gamma, beta, eps=1e-5):
    return LayerNormFn.apply(x, gamma, beta, eps, False)


def layer_norm_residual(x, gamma, beta, eps=1e-5):
    return LayerNormFn.apply(x, gamma, beta, eps, True)


class AttentionCoreFn(torch.autograd.Function):
    """softmax(mask(Q K^T / sqrt(hn))) V for all heads (transformer.py:283-381), composed: batched fp32 GEMM -> masked softmax kernel ->
    batched fp32 GEMM, probabilities [b, heads, sq, sk] kept for the backward.  Operands as in kernels.AttentionCoreFn: `qsrc` = the packed
    [b, s, 3, np, hn] QKV projection (self-attention) or [b, sq, np, hn] with `kvsrc` = [b, sk, 2, np, hn]; dq / dk / dv are written into ONE
    packed gradient through strides."""

    @staticmethod
    def forward(ctx, qsrc, kvsrc, ids_q, ids_k, causal):
        _check(qsrc, kvsrc)
        if isinstance(ids_q, K.PackedSeqs) or isinstance(ids_k, K.PackedSeqs):
            raise ValueError("the fp32 validation path runs dense [b, s] layouts only")
        if kvsrc is None:
            q, k, v = qsrc.select(-3, 0), qsrc.select(-3, 1), qsrc.select(-3, 2)
        else:
            q, k, v = qsrc, kvsrc.select(-3, 0), kvsrc.select(-3, 1)
        b, sq, heads, hn = q.shape
        sk = k.shape[1]
        scale = 1.0 / math.sqrt(hn)
        P = torch.empty((b, heads, sq, sk), dtype=F32, device=q.device)
        gemm(q, q.stride(1), 1, k, k.stride(1), 1, P, sk, 1, sq, sk, hn, b, q.stride(0), k.stride(0), heads * sq * sk, heads, q.stride(2), k.stride(2),
             sq * sk, alpha=scale)
        _native.check(_lib().emdr2_f32_softmax_mask_fwd(P.data_ptr(), ids_q.data_ptr(), ids_k.data_ptr(), b, heads, sq, sk, int(causal), _sp()),
                      "f32_softmax_mask_fwd")
        out = torch.empty((b, sq, heads, hn), dtype=F32, device=q.device)
        gemm(P, sk, 1, v, 1, v.stride(1), out, heads * hn, 1, sq, hn, sk, b, heads * sq * sk, v.stride(0), sq * heads * hn, heads, sq * sk, v.stride(2), hn)
        ctx.save_for_backward(qsrc, kvsrc, P)
        ctx.ids_q, ctx.ids_k, ctx.causal = ids_q, ids_k, causal
        return out

    @staticmethod
    def backward(ctx, dout):
        qsrc, kvsrc, P = ctx.saved_tensors
        ids_q, ids_k = ctx.ids_q, ctx.ids_k
        if kvsrc is None:
            q, k, v = qsrc.select(-3, 0), qsrc.select(-3, 1), qsrc.select(-3, 2)
            dqsrc, dkvsrc = torch.empty_like(qsrc), None
            dq, dk, dv = dqsrc.select(-3, 0), dqsrc.select(-3, 1), dqsrc.select(-3, 2)
        else:
            q, k, v = qsrc, kvsrc.select(-3, 0), kvsrc.select(-3, 1)
            dqsrc, dkvsrc = torch.empty_like(qsrc), torch.empty_like(kvsrc)
            dq, dk, dv = dqsrc, dkvsrc.select(-3, 0), dkvsrc.select(-3, 1)
        b, sq, heads, hn = q.shape
        sk = k.shape[1]
        scale = 1.0 / math.sqrt(hn)
        dout = dout.contiguous()
        H = heads * hn
        # dV[b, k, n, d] = sum_q P[b, n, q, k] dout[b, q, n, d]
        gemm(P, 1, sk, dout, 1, H, dv, dv.stride(1), 1, sk, hn, sq, b, heads * sq * sk, sq * H, dv.stride(0), heads, sq * sk, hn, dv.stride(2))
        # dP[b, n, q, k] = sum_d dout[b, q, n, d] v[b, k, n, d], then dS = P o (dP - rowsum(P o dP)) in place (0 at masked positions)
        dS = torch.empty_like(P)
        gemm(dout, H, 1, v, v.stride(1), 1, dS, sk, 1, sq, sk, hn, b, sq * H, v.stride(0), heads * sq * sk, heads, hn, v.stride(2), sq * sk)
        _native.check(_lib().emdr2_f32_softmax_mask_bwd(P.data_ptr(), dS.data_ptr(), ids_q.data_ptr(), ids_k.data_ptr(), b, heads, sq, sk,
                                                        int(ctx.causal), _sp()), "f32_softmax_mask_bwd")
        # dQ[b, q, n, d] = scale sum_k dS[b, n, q, k] k[b, k, n, d];  dK[b, k, n, d] = scale sum_q dS[b, n, q, k] q[b, q, n, d]
        gemm(dS, sk, 1, k, 1, k.stride(1), dq, dq.stride(1), 1, sq, hn, sk, b, heads * sq * sk, k.stride(0), dq.stride(0), heads, sq * sk, k.stride(2),
             dq.stride(2), alpha=scale)
        gemm(dS, 1, sk, q, 1, q.stride(1), dk, dk.stride(1), 1, sk, hn, sq, b, heads * sq * sk, q.stride(0), dk.stride(0), heads, sq * sk, q.stride(2),
             dk.stride(2), alpha=scale)
        return dqsrc, dkvsrc, None, None, None


def attention_core(qsrc, kvsrc, ids_q, ids_k, causal=False, drop_p=0.0, seed=0, site=0):
    _no_dropout(drop_p)
    return AttentionCoreFn.apply(qsrc, kvsrc, ids_q, ids_k, bool(causal))


class EmbeddingFn(torch.autograd.Function):
    """Embedding.forward (language_model.py:169-181): word + position (+ token type), fp32 tables read as they are."""

    @staticmethod
    def forward(ctx, ids, types, W, P, T):
        ids = ids.contiguous()
        types = types.contiguous() if (types is not None and T is not None) else None
        b, S = ids.shape
        H = W.shape[1]
        out = torch.empty((b, S, H), dtype=F32, device=ids.device)
        _native.check(_lib().emdr2_f32_embedding_fwd(ids.data_ptr(), K._ptr(types), W.data_ptr(), P.data_ptr(), K._ptr(T if types is not None else None),
                                                     out.data_ptr(), b * S, S, H, _sp()), "f32_embedding_fwd")
        ctx.ids, ctx.types, ctx.W, ctx.P, ctx.T = ids, types, W, P, (T if types is not None else None)
        return out

    @staticmethod
    def backward(ctx, dout):
        ids, types, W, P, T = ctx.ids, ctx.types, ctx.W, ctx.P, ctx.T
        dout = dout.contiguous()
        b, S = ids.shape
        H = W.shape[1]
        dW, dP = torch.zeros_like(W), torch.zeros_like(P)
        dT = torch.zeros_like(T) if T is not None else None
        _native.check(_lib().emdr2_f32_embedding_bwd(ids.data_ptr(), K._ptr(types), dout.data_ptr(), dW.data_ptr(), dP.data_ptr(), K._ptr(dT), b * S, S, H,
                                                     _sp()), "f32_embedding_bwd")
        K._accum_grad(W, dW)
        K._accum_grad(P, dP)
        if T is not None:
            K._accum_grad(T, dT)
        return None, None, None, None, None


def embedding(ids, types, W, P, T, drop_p=0.0, seed=0, seqs=None):
    _no_dropout(drop_p)
    if seqs is not None:
        raise ValueError("the fp32 validation path runs dense [b, s] layouts only")
    out = EmbeddingFn.apply(ids, types, W, P, T)
    # the tables are parameters delivered through _accum_grad, not autograd inputs: the output needs a grad_fn of its own whenever they train
    return out


class LseGatherFn(torch.autograd.Function):
    """gold[...] = log_softmax(logits)[label] over the vocabulary (train_e2eqa.py:79-96,152-160), fp32 logits."""

    @staticmethod
    def forward(ctx, logits, labels):
        _check(logits)
        V = logits.shape[-1]
        l2 = logits.reshape(-1, V).contiguous()
        lab = labels.reshape(-1).contiguous()
        rows = l2.shape[0]
        gold = torch.empty(rows, dtype=F32, device=logits.device)
        lse = torch.empty_like(gold)
        _native.check(_lib().emdr2_f32_lse_gather_fwd(l2.data_ptr(), lab.data_ptr(), gold.data_ptr(), lse.data_ptr(), rows, V, _sp()), "f32_lse_gather_fwd")
        ctx.save_for_backward(l2, lab, lse)
        ctx.shp = logits.shape
        return gold.reshape(labels.shape)

    @staticmethod
    def backward(ctx, dgold):
        l2, lab, lse = ctx.saved_tensors
        rows, V = l2.shape
        w = dgold.reshape(-1).contiguous().float()
        dl = torch.empty_like(l2)
        _native.check(_lib().emdr2_f32_lse_gather_bwd(l2.data_ptr(), lab.data_ptr(), lse.data_ptr(), w.data_ptr(), dl.data_ptr(), rows, V, _sp()),
                      "f32_lse_gather_bwd")
        return dl.reshape(ctx.shp), None


def lse_gather(logits, labels):
    return LseGatherFn.apply(logits, labels)


def lm_head_gold_logprob(hidden, weight, bias, labels):
    """log_softmax(hidden W^T + b)[label] of the no-grad one-context pass (emdr2_model.py:185-210, train_e2eqa.py:79-96): here the logits are
    simply materialised in fp32 (the bf16 path fuses the head with the log-softmax; speed is a non-goal of this mode)."""
    with torch.no_grad():
        return lse_gather(linear(hidden, weight, bias), labels)


class RetrieverPriorFn(torch.autograd.Function):
    """topk_log_probs = log_softmax_k(scale * q . c_k) (emdr2_model.py:134-145), fp32 embeddings."""

    @staticmethod
    def forward(ctx, q, c, scale):
        _check(q, c)
        B, Kk, H = c.shape
        q, c = q.contiguous(), c.contiguous()
        logp = torch.empty((B, Kk), dtype=F32, device=q.device)
        prob = torch.empty_like(logp)
        _native.check(_lib().emdr2_f32_retriever_prior_fwd(q.data_ptr(), c.data_ptr(), logp.data_ptr(), prob.data_ptr(), B, Kk, H, float(scale), _sp()),
                      "f32_retriever_prior_fwd")
        ctx.save_for_backward(q, c, prob)
        ctx.scale = float(scale)
        return logp

    @staticmethod
    def backward(ctx, dlogp):
        q, c, prob = ctx.saved_tensors
        B, Kk, H = c.shape
        dq, dc = torch.empty_like(q), torch.empty_like(c)
        _native.check(_lib().emdr2_f32_retriever_prior_bwd(dlogp.contiguous().data_ptr(), prob.data_ptr(), q.data_ptr(), c.data_ptr(), dq.data_ptr(),
                                                           dc.data_ptr(), B, Kk, H, ctx.scale, _sp()), "f32_retriever_prior_bwd")
        return dq, dc, None


def retriever_prior(q, c, scale):
    return RetrieverPriorFn.apply(q, c, scale)
