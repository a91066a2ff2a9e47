"""Thin torch-tensor wrappers over the C ABI of include/emdr2_ops.h (libemdr2_hip.so) and the autograd glue that
chains them.  No torch compute lives here: every FLOP and every byte moved on the model path goes through a hand-written HIP
kernel; torch provides storage, views, the autograd tape and streams.  There is no CPU fallback (a missing library raises)."""
import math

import torch

from emdr2_amd import _native

BF16 = torch.bfloat16


def _lib():
    return _native.lib()


def _sp():
    return _native.stream_ptr()


def _ptr(t):
    return t.data_ptr() if t is not None else None


def _f32(t):
    """The validation-only fp32 compute path (kernels_f32.py, include/emdr2_ops_f32.h) is selected by the ACTIVATION dtype: a stack whose
    embeddings emit fp32 (transformer.Config(compute_dtype="fp32")) runs every op downstream in fp32."""
    return t is not None and t.dtype == torch.float32


def _F32():
    from emdr2_amd.model import kernels_f32
    return kernels_f32


def _check_bf16(*ts):
    for t in ts:
        if t is not None and (t.dtype != BF16 or not t.is_cuda):
            raise TypeError("expected a CUDA bfloat16 tensor")


# ---- dropout seeds -----------------------------------------------------------------------------------------------------------
class _DropoutState(object):
    """Every dropout site (a module instance x a role) owns a static site id; the seed a kernel gets is a hash of
    (base seed, optimizer step, pass id, site id).  It is a pure function of those four, so a layer recomputed by
    --checkpoint-activations regenerates exactly the bits of its first forward, and a backward regenerates the forward's mask instead of
    storing it.  `pass_id` separates the several forward passes one training step makes through the same layers (reader over K documents
    vs. the one-context pass); `micro` the question micro-batches of one step (EMDR2Model.forward_backward: every group of questions runs
    through the same sites with rows numbered from 0, so without it all groups would share one mask per site; micro 0 = the seeds of an
    undivided step).  Like the reference (megatron/initialize.py:_set_random_seed) all data-parallel ranks use the same seed."""

    def __init__(self):
        self.base_seed, self.step, self.pass_id, self.micro, self._sites = 1234, 0, 0, 0, 0

    def new_site(self):
        self._sites += 1
        return self._sites

    def seed(self, site):
        x = (self.base_seed * 0x9E3779B1 + self.step * 0x85EBCA77 + self.pass_id * 0xC2B2AE3D + site * 0x27D4EB2F + self.micro * 0x165667B1) & 0xFFFFFFFF
        x ^= x >> 16; x = (x * 0x7FEB352D) & 0xFFFFFFFF; x ^= x >> 15; x = (x * 0x846CA68B) & 0xFFFFFFFF; x ^= x >> 16
        return x


DROPOUT = _DropoutState()


# ---- raw kernels --------------------------------------------------------------------------------------------------------
class _RecomputeMeter(object):
    """GEMM flops spent RE-running forward work inside a backward: the re-run of a checkpointed layer (ATTN_STASH.mode == 'consume') or the
    FFN-1 / projection recomputation of the selective-retention functions (`active` > 0).  bench_e2e.py reports it per step."""

    def __init__(self):
        self.flops, self.active = 0.0, 0


RECOMPUTE = _RecomputeMeter()


def gemm_nt(A, lda, B, ldb, C, ldc, M, N, K, batch1=1, sA1=0, sB1=0, sC1=0, batch2=1, sA2=0, sB2=0, sC2=0, alpha=1.0, bias=None,
            gelu=False, pre_act=None, residual=None, split_k=1, drop_p=0.0, seed=0, residual_mode=0):
    if RECOMPUTE.active or ATTN_STASH.mode == 'consume':
        RECOMPUTE.flops += 2.0 * M * N * K * batch1 * batch2
    _native.check(_lib().emdr2_gemm_nt_bf16(A.data_ptr(), lda, B.data_ptr(), ldb, C.data_ptr(), ldc, M, N, K, batch1, sA1, sB1, sC1, batch2, sA2,
                                            sB2, sC2, alpha, _ptr(bias), int(gelu), _ptr(pre_act), _ptr(residual), int(residual_mode),
                                            int(C.dtype == torch.float32), split_k, float(drop_p), int(seed), _sp()), "gemm_nt_bf16")
    return C


def matmul_nt(a, b, out_dtype=BF16, **kw):
    """a [M, K] @ b[N, K]^T for contiguous 2-D bf16 operands."""
    M, K = a.shape
    N = b.shape[0]
    c = torch.empty((M, N), dtype=out_dtype, device=a.device)
    return gemm_nt(a, K, b, K, c, N, M, N, K, **kw)


def weight_grad_nt(dyT, xT):
    """dW [N, K] fp32 = dyT [N, M] @ xT [K, M]^T : few output tiles, reduction over all tokens -> split the reduction across the chip."""
    N, M = dyT.shape
    Kd = xT.shape[0]
    tiles = ((N + 255) // 256) * ((Kd + 255) // 256)
    split = max(1, min(512 // max(tiles, 1), M // 4096))
    if split == 1:
        return matmul_nt(dyT, xT, out_dtype=torch.float32)
    c = torch.zeros((N, Kd), dtype=torch.float32, device=dyT.device)
    return gemm_nt(dyT, M, xT, M, c, Kd, N, Kd, M, split_k=split)


def weight_grad_tn(dy, x, colsum=None, into=None):
    """dW [N, K] fp32 = dy [M, N]^T @ x [M, K] straight from the row-major activations (gemm_tn.hip); `colsum` (fp32 [N]) accumulates the
    bias gradient.  Few output tiles, reduction over all tokens -> the reduction is split across the chip.  `into` (fp32 [N, K], holding
    zeros or earlier contributions): accumulate there with the slices' atomics instead of returning a fresh tensor -- taken only when the
    reduction is split (a single slice stores plainly); the caller learns which from the returned tensor."""
    M, N = dy.shape
    Kd = x.shape[1]
    tiles = ((N + 255) // 256) * ((Kd + 255) // 256)
    split = max(1, min(512 // max(tiles, 1), M // 4096))
    if into is not None and split > 1:
        c = into
    else:
        c = (torch.zeros if split > 1 else torch.empty)((N, Kd), dtype=torch.float32, device=dy.device)
    _native.check(_lib().emdr2_gemm_tn_bf16(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), c.data_ptr(), Kd, N, Kd, M, split, _ptr(colsum),
                                            _sp()), "gemm_tn_bf16")
    return c


def transpose(x2d, colsum=None):
    """[R, C] bf16 contiguous -> [C, R]; optional fp32 column sums accumulated into `colsum`."""
    R, C = x2d.shape
    out = torch.empty((C, R), dtype=BF16, device=x2d.device)
    _native.check(_lib().emdr2_transpose_bf16(x2d.data_ptr(), C, out.data_ptr(), R, R, C, 1, 0, 0, 1, 0, 0, _ptr(colsum), _sp()), "transpose")
    return out


def head_transpose(x, b, s, heads, hn):
    """x: strided view [b, s, heads, hn] (last dim contiguous) -> contiguous [b, heads, hn, s]."""
    ld = x.stride(1)
    out = torch.empty((b, heads, hn, s), dtype=BF16, device=x.device)
    _native.check(_lib().emdr2_transpose_bf16(x.data_ptr(), ld, out.data_ptr(), s, s, hn, b, x.stride(0), heads * hn * s, heads, x.stride(2), hn * s,
                                              None, _sp()), "transpose")
    return out


def cast_bf16(src_f32):
    dst = torch.empty(src_f32.shape, dtype=BF16, device=src_f32.device)
    _native.check(_lib().emdr2_cast_f32_to_bf16(src_f32.data_ptr(), dst.data_ptr(), src_f32.numel(), _sp()), "cast")
    return dst


class _WeightCache(object):
    """bf16 working copies of the fp32 master parameters (plus transposed / row-permuted forms), kept ON the parameter object and
    rebuilt when the master changed: either its autograd version moved (in-place torch update) or the optimizer bumped `epoch`
    (the HIP Adam kernel writes the master through a raw pointer, which autograd cannot see)."""

    def __init__(self):
        self.epoch = 0
        self.listeners = []       # called by invalidate(): training.FlatAdam keeps its own bf16 copies current (weakly held would be nicer; one per run)

    def get(self, p, kind, build):
        cache = p.__dict__.setdefault("_emdr2_cache", {})
        ent = cache.get(kind)
        # a frozen parameter (weight snapshot of the side-stream indexer) is never touched by the optimizer: ignore the epoch
        stamp = (p._version, -1 if p.__dict__.get("_emdr2_frozen") else self.epoch)
        if ent is None or ent[0] != stamp:
            ent = (stamp, build())
            cache[kind] = ent
        return ent[1]

    def invalidate(self):
        self.epoch += 1
        for ref in list(self.listeners):                          # weak references to bound methods
            fn = ref()
            if fn is None:
                self.listeners.remove(ref)
            else:
                fn()


WEIGHTS = _WeightCache()


def w_bf16(p):
    flat = p.__dict__.get("_emdr2_flat")
    if flat is not None:                      # the parameter lives in a training.FlatAdam bucket: its bf16 copy is a view the Adam kernel writes
        return flat.work_view(p)
    return WEIGHTS.get(p, "bf16", lambda: cast_bf16(p.detach().contiguous()))


def w_bf16_t(p):
    return WEIGHTS.get(p, "bf16_t", lambda: transpose(w_bf16(p)))


def w_bf16_perm(p, perm):
    """bf16 copy with rows gathered as perm (de-interleaving the reference's [np, hn, 3] QKV row order once per weight update)."""
    return WEIGHTS.get(p, "perm", lambda: w_bf16(p)[perm].contiguous())


GRAD_SINK = None    # the training.FlatAdam that owns the parameters' flat gradient buckets (and overlaps their all-reduce with the backward)


def _accum_grad(p, g_f32):
    """Every parameter gradient of the model is delivered here by the autograd functions of this module (fp32, produced by the
    weight-gradient GEMMs / reduction kernels)."""
    if GRAD_SINK is not None and GRAD_SINK.owns(p):
        GRAD_SINK.accumulate(p, g_f32)
    elif p.grad is None:
        p.grad = g_f32
    else:
        p.grad += g_f32        # rare (a parameter used twice in one backward): tied embedding / LM head


def _grad_buffer(p, shape=None):
    """Where a kernel that ACCUMULATES (atomics, +=) should put the fp32 gradient of parameter p: the parameter's own slice of the
    optimizer's flat gradient bucket while that bucket still takes contributions (training.FlatAdam zeroes its buckets once per step: no
    per-parameter fill, no copy afterwards), else a fresh zeroed tensor.  Returns (buffer, direct)."""
    if GRAD_SINK is not None and GRAD_SINK.owns(p):
        v = GRAD_SINK.accumulation_target(p)
        if v is not None:
            return v, True
    return torch.zeros(p.shape if shape is None else shape, dtype=torch.float32, device=p.device), False


def _deliver_grad(p, buf, direct):
    if direct:
        GRAD_SINK.contributed(p)
    else:
        _accum_grad(p, buf)


def _linear_grads_into(dy2, x2, weight, bias):
    """dW = dy^T x and the bias gradient of a linear layer in its checkpoint row order, accumulated where the parameters' gradients live."""
    db, db_direct = _grad_buffer(bias) if bias is not None else (None, False)
    tw = GRAD_SINK.accumulation_target(weight) if GRAD_SINK is not None and GRAD_SINK.owns(weight) else None
    dW = weight_grad_tn(dy2, x2, colsum=db, into=tw)
    if tw is not None and dW is tw:                               # (a single-slice reduction stores plainly into a tensor of its own)
        GRAD_SINK.contributed(weight)
    else:
        _accum_grad(weight, dW)
    if bias is not None:
        _deliver_grad(bias, db, db_direct)


# ---- one input, many linear consumers: the input gradient is summed by the consumers' own GEMMs --------------------------------
class _FanInState(object):
    def __init__(self, node, numel):
        import weakref
        self.acc, self.node, self.numel = None, weakref.ref(node), numel


class _FanInTable(dict):
    """data_ptr of a tensor that went through `fan_in` -> its gradient accumulator (lives from that forward to its backward).  An entry
    whose graph has gone away without a backward (its autograd node is dead) is void: the allocator may have handed that address to another
    linear layer's input, whose gradient must not be swallowed -- `live` drops such entries instead of returning them."""

    def live(self, x2):
        st = self.get(x2.data_ptr())
        if st is not None and (st.node() is None or st.numel != x2.numel()):
            del self[x2.data_ptr()]
            return None
        return st


FANIN = _FanInTable()


class _PreMask(object):
    """The backward of a bias-dropout-add (y = x + dropout(z)) needs dy o mask as the operand of two GEMMs.  `dy` is produced by the backward of
    the LayerNorm that consumed y, so that launch can write the masked copy on its way (csrc/elementwise.hip: layernorm_bwd768_kernel<true>)
    instead of a dropout launch re-reading dy.  Forward: the bias-dropout-add registers (p, seed) under its output's address (`want`);
    backward: the LayerNorm backward whose INPUT has that address leaves (dx ITSELF, p, seed) -> masked tensor in the one `slot`; the
    bias-dropout-add's backward takes it if its dy IS that dx (same storage, same shape, not written since) with the same (p, seed), and
    falls back to the dropout kernel otherwise.  The slot holds dx -- not just its address -- on purpose: were y ever consumed
    by a second autograd node, autograd's input buffer could ADD that node's gradient into the dx buffer in place (it does when it holds
    the only reference to tensor and storage), same address, other values, and the masked copy would be mask(partial dx); with the slot's
    reference the sum goes to a fresh tensor (another address: no match), and `_version` catches any other in-place write.  One slot: the
    consumer is the next node autograd runs, anything else overwrites or clears it, so at most one extra [tokens, h] tensor is alive."""

    def __init__(self):
        self.want, self.slot, self.enabled, self.fused, self.unfused = {}, None, True, 0, 0

    def register(self, y, drop_p, seed):
        if self.enabled and drop_p > 0.0:
            self.want[y.data_ptr()] = (float(drop_p), int(seed), y.numel())

    def request_for(self, x2):
        req = self.want.pop(x2.data_ptr(), None)
        return req if (req is not None and req[2] == x2.numel()) else None

    def offer(self, dx, drop_p, seed, dmask):
        self.slot = (dx, dx._version, float(drop_p), int(seed), dmask)

    def take(self, dy2, drop_p, seed):
        slot, self.slot = self.slot, None
        if slot is not None:
            dx, version, p, sd, dmask = slot
            if (dx.data_ptr() == dy2.data_ptr() and dx._version == version and dy2._version == version and dx.numel() == dy2.numel()
                    and (p, sd) == (float(drop_p), int(seed)) and dmask.shape == dy2.shape):
                self.fused += 1
                return dmask
        self.unfused += 1
        return None

    def clear(self):
        self.want.clear()
        self.slot = None


PREMASK = _PreMask()


def _dropout_mask_of(dy2, N, drop_p, seed):
    """dy o keep(seed) / (1 - p) for the backward GEMMs of a bias-dropout-add: from the LayerNorm backward that made dy, else by its own kernel."""
    pre = PREMASK.take(dy2, drop_p, seed)
    if pre is not None:
        return pre
    dmask = torch.empty_like(dy2)
    _native.check(_lib().emdr2_dropout(dy2.data_ptr(), dmask.data_ptr(), dy2.numel(), N, drop_p, seed, _sp()), "dropout")
    return dmask


class FanInFn(torch.autograd.Function):
    """y = x for a tensor that feeds MANY linear layers (the FiD reader's encoder output: the K/V projection of every decoder layer reads all
    K * S encoder tokens).  Autograd would materialise each consumer's [tokens, h] input gradient and sum them pairwise (11 additions of
    2 GB tensors per step); instead the consumers' data-gradient GEMMs add their product onto ONE buffer in their epilogue
    (`LinearFn.backward`, + residual in place) and hand autograd nothing; this node delivers the buffer.  Same roundings as the pairwise
    bf16 sums (each product is rounded to bf16 before it meets the running sum, csrc/gemm8.hip), same order."""

    @staticmethod
    def forward(ctx, x):
        ctx.key = x.data_ptr()
        FANIN[ctx.key] = _FanInState(ctx, x.numel())
        ctx.set_materialize_grads(False)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        st = FANIN.pop(ctx.key, None)
        acc = st.acc if st is not None else None
        if g is not None:                                         # a consumer that is not one of ours
            acc = g if acc is None else acc + g.reshape(acc.shape)
        return acc


def fan_in(x):
    if _f32(x):
        return x                                                  # (fp32 validation path: autograd's own sums)
    return FanInFn.apply(x) if torch.is_grad_enabled() and x.requires_grad else x


# ---- packed ("varlen") sequence layout ---------------------------------------------------------------------------------------
class _Packing(object):
    """Switch for the packed layout of the encoder stacks (csrc/seqpack.hip): ON = the context tower, the reader encoder and the
    one-context pass run only over real tokens (every consumed value is unchanged, see PackedSeqs); OFF = the reference's [batch, S] grids."""

    def __init__(self):
        self.enabled = True
        self.real_tokens = self.grid_tokens = 0          # running totals over the layouts built so far (bench: tokens_real / tokens_padded)
        self.history = []                                # (n, S, rows) of the most recent layouts (bench: activation budget per stack)
        # Training loops set `sticky`: the row count of a large stack then never shrinks and grows to (new maximum + 1 %) in 1,024-row units, per
        # (n, S, fill octile) -- after a few steps every activation of the stack has the SAME size step after step, which is what lets the caching
        # allocator reuse its blocks (real-token counts move by a fraction of a percent per step; with free-running sizes two or three size
        # classes of every transient tensor pile up: 58-68 GB "reserved but unallocated" and an out-of-memory after ~10 steps at 230 GB)
        self.sticky = False
        self.capacity = {}
        self.growths = 0                                 # how often a sticky capacity had to grow (bench: inside the timed steps?)


PACKING = _Packing()


class PackedSeqs(object):
    """Layout of n sequences of <= S tokens stored back to back without their trailing [PAD] rows (include/emdr2_ops.h, "packed
    sequences"): sequence i owns rows [cu[i], cu[i+1]) of every [rows, ...] tensor of the stack; `rows` = the real token count `total`
    rounded up to ROW_MULTIPLE (tail rows hold zeros / token id 0, so the persistent GEMMs see whole 256-row tiles and weight gradients
    are unaffected).  Building it costs one host sync (the row count sizes every activation of the stack).
    Why this is exact: a padded key contributes exp(-10000 - max) == 0 to every real query of the reference's masked softmax
    (transformer.py:283-381 with bert/t5_attention_mask_func), and padded queries are consumed by nothing (token 0 of the towers,
    dualencoder_model.py:166-181; the FiD decoder masks padded encoder positions, emdr2_model.py:166-183) and receive zero gradient."""
    ROW_MULTIPLE = 256

    def __init__(self, ids, types=None):
        if ids.dim() != 2 or ids.dtype != torch.int64 or not ids.is_cuda:
            raise TypeError("token ids must be a CUDA int64 [n, S] tensor")
        ids = ids.contiguous()
        types = types.contiguous() if types is not None else None
        n, S = ids.shape
        dev = ids.device
        lib = _lib()
        self.n, self.S, self.max_len, self.group = n, S, S, 1
        self.dense_ids = ids                                  # the [n, S] grid this layout was built from (consumers that want the reference's shapes)
        self.cu = torch.empty(n + 1, dtype=torch.int32, device=dev)
        totals = torch.empty(3, dtype=torch.int64, device=dev)
        _native.check(lib.emdr2_seq_lengths(ids.data_ptr(), n, S, self.cu.data_ptr(), totals.data_ptr(), _sp()), "seq_lengths")
        self.total, self.pairs, self.max_len = (int(v) for v in totals.tolist())    # the layout's one host sync (max_len sizes the attention grids)
        PACKING.real_tokens += self.total
        PACKING.grid_tokens += n * S
        # rows: whole GEMM tiles; for large stacks a coarser granule (<= 1.6 % more rows) so that the activation sizes of successive training
        # steps -- whose real-token counts differ by a fraction of a percent -- repeat and the caching allocator reuses its blocks
        m = self.ROW_MULTIPLE if self.total < (1 << 16) else (8192 if self.total < (1 << 20) else 16384)
        self.rows = (self.total + m - 1) // m * m
        if PACKING.sticky and self.total >= (1 << 16):
            # training: a stack's row count is a STICKY capacity -- the same from step to step, so every activation size repeats and the caching
            # allocator reuses its blocks.  r06: the capacity is the largest token count seen so far + 1 % (the step-to-step spread of a stack's
            # real tokens is 0.1 - 0.3 %), in 1,024-row units.  (r03 - r05: whole 8,192 / 16,384-row granules plus 16,384 rows on every growth --
            # sized for the undivided 1.3 M-row stack; on a question group's 0.33 M rows one growth was + 7.5 % rows for the rest of the run:
            # every step after it 3.5 % slower (same box, 10 timed steps: 1,555 ms without a growth, 1,610 after one; this policy 1,564 - 1,572,
            # profiles/r06_e2e_capacity_policy.txt) -- and the growth step itself pays ~1.6 s of fresh hipMallocs for every activation size, so
            # growths should be rare AND small.)
            key = (n, S, int(8.0 * self.total / (n * S)))
            cap = PACKING.capacity.get(key, 0)
            need = (self.total + 255) // 256 * 256
            if need > cap:
                cap = (int(self.total * 1.01) + 1023) // 1024 * 1024
                PACKING.capacity[key] = cap
                PACKING.growths += 1
            self.rows = cap
        PACKING.history = PACKING.history[-15:] + [(n, S, self.rows)]
        self.rowmap = torch.empty(self.rows, dtype=torch.int32, device=dev)         # packed row -> dense row (i * S + pos), -1 in the tail
        self.inverse = torch.empty(n * S, dtype=torch.int32, device=dev)            # dense row -> packed row, -1 at dropped pad rows
        self.ids = torch.empty(self.rows, dtype=torch.int64, device=dev)
        self.types = torch.empty(self.rows, dtype=torch.int64, device=dev) if types is not None else None
        _native.check(lib.emdr2_seq_pack_ids(ids.data_ptr(), _ptr(types), self.cu.data_ptr(), n, S, self.total, self.rows, self.rowmap.data_ptr(),
                                             self.inverse.data_ptr(), self.ids.data_ptr(), _ptr(self.types), _sp()), "seq_pack_ids")

    def grouped(self, k):
        """The same rows seen as n / k sequences, each the concatenation of k consecutive ones: the FiD decoder's keys (the K passages of a
        question concatenated, emdr2_model.py:159-161) -- contiguous in the packed buffer, no pad rows in between."""
        if k == 1:
            return self
        if self.group != 1 or self.n % k:
            raise ValueError("grouping needs an ungrouped layout of a multiple of k sequences")
        g = object.__new__(PackedSeqs)
        g.__dict__.update(self.__dict__)
        g.n, g.max_len, g.group = self.n // k, min(self.S, self.max_len) * k, k
        g.cu = self.cu[::k].contiguous()
        g.pairs = None
        return g

    def zero_tail(self, t):
        if self.rows > self.total:
            t[self.total:].zero_()
        return t


class GatherRowsFn(torch.autograd.Function):
    """out[r] = x[fwd_map[r]] (zeros where the map is negative) over bf16 rows; backward = the gather through `bwd_map` when the caller has
    the inverse map (pack <-> unpack), else a scatter into zeros (token-0 rows of a packed tensor)."""

    @staticmethod
    def forward(ctx, x, fwd_map, rows_out, bwd_map):
        _check_bf16(x)
        H = x.shape[-1]
        x2 = x.reshape(-1, H)
        out = torch.empty((rows_out, H), dtype=BF16, device=x.device)
        _native.check(_lib().emdr2_gather_rows(x2.data_ptr(), fwd_map.data_ptr(), out.data_ptr(), rows_out, H, _sp()), "gather_rows")
        ctx.fwd_map, ctx.bwd_map, ctx.in_shape = fwd_map, bwd_map, x.shape
        return out

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        H = dy.shape[-1]
        rows_in = 1
        for d in ctx.in_shape[:-1]:
            rows_in *= d
        if ctx.bwd_map is not None:
            dx = torch.empty((rows_in, H), dtype=BF16, device=dy.device)
            _native.check(_lib().emdr2_gather_rows(dy.data_ptr(), ctx.bwd_map.data_ptr(), dx.data_ptr(), rows_in, H, _sp()), "gather_rows")
        else:
            dx = torch.zeros((rows_in, H), dtype=BF16, device=dy.device)
            _native.check(_lib().emdr2_scatter_rows(dy.data_ptr(), ctx.fwd_map.data_ptr(), dx.data_ptr(), dy.shape[0], H, _sp()), "scatter_rows")
        return dx.reshape(ctx.in_shape), None, None, None


def unpack_rows(x, seqs):
    """packed [rows, H] -> dense [n, S, H] with zeros at the dropped pad positions (the reference's shape, for consumers that want it)."""
    if seqs.group != 1:
        raise ValueError("unpack through the ungrouped layout")
    return GatherRowsFn.apply(x, seqs.inverse, seqs.n * seqs.S, seqs.rowmap).reshape(seqs.n, seqs.S, x.shape[-1])


def pack_rows(x, seqs):
    """dense [n, S, H] -> packed [rows, H]."""
    return GatherRowsFn.apply(x, seqs.rowmap, seqs.rows, seqs.inverse)


def first_rows(x, seqs):
    """[n, H]: the row of token 0 of every sequence of a packed tensor (the towers' output, dualencoder_model.py:166-181)."""
    return GatherRowsFn.apply(x, seqs.cu, seqs.n, None)


# ---- autograd functions ----------------------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    """y = x W^T + b, optional exact-erf GELU, optional residual add (the reference's F.linear + bias(+gelu) / bias-dropout-add at
    p = 0: mpu/layers.py:255,353, transformer.py:94-108,397-407).  W, b are fp32 masters; GEMMs run on bf16 copies."""

    @staticmethod
    def forward(ctx, x, weight, bias, gelu, residual, row_perm=None, drop_p=0.0, seed=0, grad_on=True):
        _check_bf16(x, residual)
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        if not x2.is_contiguous():
            raise ValueError("linear input must be contiguous")
        M, K = x2.shape
        N = weight.shape[0]
        y = torch.empty((M, N), dtype=BF16, device=x.device)
        pre = torch.empty_like(y) if gelu else None
        res2 = residual.reshape(M, N) if residual is not None else None
        wb = w_bf16(weight) if row_perm is None else w_bf16_perm(weight, row_perm)
        bb = None
        if bias is not None:
            bb = bias.detach() if row_perm is None else WEIGHTS.get(bias, "perm", lambda: bias.detach()[row_perm].contiguous())
        gemm_nt(x2, K, wb, K, y, N, M, N, K, bias=bb, gelu=gelu, pre_act=pre, residual=res2, drop_p=drop_p, seed=seed)
        if grad_on and ATTN_STASH.mode != 'store':           # a backward will really run from THIS forward (cf. MLPFn.forward)
            PREMASK.register(y, drop_p, seed)
        ctx.save_for_backward(x2, pre)
        ctx.drop_p, ctx.seed = drop_p, seed
        ctx.weight, ctx.bias, ctx.gelu, ctx.has_res, ctx.shp, ctx.row_perm = weight, bias, gelu, residual is not None, shp, row_perm
        return y.reshape(shp[:-1] + (N,))

    @staticmethod
    def backward(ctx, dy):
        x2, pre = ctx.saved_tensors
        weight, bias = ctx.weight, ctx.bias
        M, K = x2.shape
        N = weight.shape[0]
        dy2 = dy.reshape(M, N)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dres = dy if ctx.has_res else None
        if ctx.drop_p > 0.0:                                                              # the epilogue's dropout mask, regenerated
            dy2 = _dropout_mask_of(dy2, N, ctx.drop_p, ctx.seed)
        if ctx.gelu:
            dpre = torch.empty_like(dy2)
            _native.check(_lib().emdr2_gelu_bwd(pre.data_ptr(), dy2.data_ptr(), dpre.data_ptr(), dy2.numel(), _sp()), "gelu_bwd")
            dy2 = dpre
        dx = None
        if ctx.needs_input_grad[0]:
            wt = w_bf16_t(weight) if ctx.row_perm is None else WEIGHTS.get(weight, "perm_t", lambda: transpose(w_bf16_perm(weight, ctx.row_perm)))
            shared = FANIN.live(x2)
            if shared is None:
                dx = matmul_nt(dy2, wt).reshape(ctx.shp)                                  # [M,N] x [K,N]^T
            elif shared.acc is None:
                shared.acc = matmul_nt(dy2, wt).reshape(ctx.shp)                          # first consumer of a fanned-in input (FanInFn) ...
            else:
                acc2 = shared.acc.reshape(M, K)                                           # ... the others add onto it in their epilogue, in place
                gemm_nt(dy2, N, wt, N, acc2, K, M, K, N, residual=acc2)
        if weight.requires_grad:
            if M % 32:
                raise ValueError("token count must be a multiple of 32 for the weight-gradient GEMM")
            _linear_param_grads(dy2, x2, weight, bias, ctx.row_perm)                      # [N, K] fp32 = dy^T x (+ bias gradient), no transposes
        return dx, None, None, None, dres, None, None, None, None


def linear(x, weight, bias=None, gelu=False, residual=None, row_perm=None, drop_p=0.0, seed=0):
    if _f32(x):
        return _F32().linear(x, weight, bias, gelu, residual, row_perm, drop_p, seed)
    return LinearFn.apply(x, weight, bias, gelu, residual, row_perm, drop_p, seed, torch.is_grad_enabled())


class MLPFn(torch.autograd.Function):
    """ParallelMLP + bias-dropout-add as one node (transformer.py:58-108,397-413): y = residual + dropout(gelu(x W1^T + b1) W2^T + b2).
    Fusing the two linears lets the backward fold gelu'(pre) into the epilogue of the GEMM that produces d(gelu output) instead of
    running a separate 3 x [tokens, ffn] elementwise pass between two autograd nodes."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, residual, drop_p=0.0, seed=0, grad_on=True):
        _check_bf16(x, residual)
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        if not x2.is_contiguous():
            raise ValueError("MLP input must be contiguous")
        M, H = x2.shape
        F = w1.shape[0]
        # gelu'(pre-activation) -- all the backward ever needs of the pre-activation, written by the FFN-1 GEMM's epilogue next to the
        # activation (gelu = 2) and multiplied in by the epilogue of the GEMM that produces d(activation) (residual_mode = 2) -- is only needed
        # by a backward that will really run from THIS forward: not in no-grad passes (the one-context
        # pass, evaluation) and not in the first run of a checkpointed layer, whose saved tensors are dropped and rebuilt by the re-run
        # (`grad_on` is the caller's grad mode: inside Function.forward grad is always off, and needs_input_grad reflects requires_grad of
        # the parameters even under torch.no_grad())
        need_pre = grad_on and any(ctx.needs_input_grad) and ATTN_STASH.mode != 'store'
        pre = torch.empty((M, F), dtype=BF16, device=x.device) if need_pre else None
        inter = torch.empty((M, F), dtype=BF16, device=x.device)
        gemm_nt(x2, H, w_bf16(w1), H, inter, F, M, F, H, bias=b1.detach(), gelu=2 if need_pre else 1, pre_act=pre)
        y = torch.empty((M, H), dtype=BF16, device=x.device)
        gemm_nt(inter, F, w_bf16(w2), F, y, H, M, H, F, bias=b2.detach(), residual=residual.reshape(M, H), drop_p=drop_p, seed=seed)
        if need_pre:
            PREMASK.register(y, drop_p, seed)
        # (a placeholder keeps the NUMBER of saved tensors equal between a checkpointed layer's first run and its re-run, which is how
        # torch.utils.checkpoint pairs them up)
        ctx.save_for_backward(x2, pre if pre is not None else x2.new_empty(0), inter)
        ctx.params, ctx.shp, ctx.drop_p, ctx.seed = (w1, b1, w2, b2), shp, drop_p, seed
        return y.reshape(shp)

    @staticmethod
    def backward(ctx, dy):
        x2, pre, inter = ctx.saved_tensors
        if pre.numel() == 0:
            raise RuntimeError("MLP backward without the saved pre-activation (forward ran in a mode that does not keep it)")
        w1, b1, w2, b2 = ctx.params
        M, H = x2.shape
        F = w1.shape[0]
        if M % 32:
            raise ValueError("token count must be a multiple of 32 for the weight-gradient GEMM")
        dy2 = dy.reshape(M, H)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dres = dy
        if ctx.drop_p > 0.0:
            dy2 = _dropout_mask_of(dy2, H, ctx.drop_p, ctx.seed)
        # d(pre) = (dy W2) * gelu'(pre): the multiply rides in the GEMM epilogue (`pre` holds the derivative, see forward)
        dpre = torch.empty((M, F), dtype=BF16, device=dy.device)
        gemm_nt(dy2, H, w_bf16_t(w2), H, dpre, F, M, F, H, residual=pre, residual_mode=2)
        _linear_grads_into(dy2, inter, w2, b2)
        dx = matmul_nt(dpre, w_bf16_t(w1)).reshape(ctx.shp) if ctx.needs_input_grad[0] else None
        _linear_grads_into(dpre, x2, w1, b1)
        return dx, None, None, None, None, dres, None, None, None


def mlp(x, w1, b1, w2, b2, residual, drop_p=0.0, seed=0):
    if _f32(x):
        return _F32().mlp(x, w1, b1, w2, b2, residual, drop_p, seed)
    return MLPFn.apply(x, w1, b1, w2, b2, residual, drop_p, seed, torch.is_grad_enabled())


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, passthrough=False):
        """passthrough: also return x itself as a second output (the residual stream).  Its gradient then arrives here instead of at an
        autograd add node, and the backward kernel folds it into dx (one pass instead of LN-backward + a 3-tensor elementwise add)."""
        _check_bf16(x)
        H = x.shape[-1]
        x2 = x.reshape(-1, H)
        rows = x2.shape[0]
        y = torch.empty_like(x2)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        _native.check(_lib().emdr2_layernorm_fwd(x2.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                 rows, H, eps, _sp()), "layernorm_fwd")
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
        return _ln_backward(dy2, x2, gamma, beta, mean, rstd, dres).reshape(dy.shape), None, None, None, None


def _ln_forward(x2, gamma, beta, eps):
    rows, H = x2.shape
    y = torch.empty_like(x2)
    mean = torch.empty(rows, dtype=torch.float32, device=x2.device)
    rstd = torch.empty_like(mean)
    _native.check(_lib().emdr2_layernorm_fwd(x2.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                             rows, H, eps, _sp()), "layernorm_fwd")
    return y, mean, rstd


def _ln_backward(dy2, x2, gamma, beta, mean, rstd, dres):
    """dx of LayerNorm (+ dres, the gradient of a residual branch that by-passed it); accumulates the gain / bias gradients."""
    rows, H = x2.shape
    dx = torch.empty_like(x2)
    (dg, dg_direct), (db, db_direct) = _grad_buffer(gamma), _grad_buffer(beta)            # the kernel adds its column sums atomically
    req = PREMASK.request_for(x2)          # x2 is the output of a bias-dropout-add whose backward runs next and wants dx o mask
    rc = -4
    if req is not None:
        dmask = torch.empty_like(x2)
        rc = _lib().emdr2_layernorm_bwd_mask(dy2.data_ptr(), x2.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _ptr(dres), dx.data_ptr(),
                                             dg.data_ptr(), db.data_ptr(), rows, H, dmask.data_ptr(), req[0], req[1], _sp())
        if rc == 0:
            PREMASK.offer(dx, req[0], req[1], dmask)
        elif rc != -4:
            _native.check(rc, "layernorm_bwd_mask")
    if rc == -4:
        _native.check(_lib().emdr2_layernorm_bwd(dy2.data_ptr(), x2.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _ptr(dres),
                                                 dx.data_ptr(), dg.data_ptr(), db.data_ptr(), rows, H, _sp()), "layernorm_bwd")
    _deliver_grad(gamma, dg, dg_direct)
    _deliver_grad(beta, db, db_direct)
    return dx


def _linear_param_grads(dy2, x2, weight, bias, row_perm):
    """dW = dy^T x (+ bias gradient) of a linear layer, delivered to the parameter (rows permuted back to the checkpoint's order)."""
    N = weight.shape[0]
    if dy2.shape[0] % 32:
        raise ValueError("token count must be a multiple of 32 for the weight-gradient GEMM")
    if row_perm is None:
        _linear_grads_into(dy2, x2, weight, bias)
        return
    db = torch.zeros(N, dtype=torch.float32, device=dy2.device) if bias is not None else None
    dW = weight_grad_tn(dy2, x2, colsum=db)
    un = torch.empty_like(dW); un[row_perm] = dW; dW = un
    if bias is not None:
        ub = torch.empty_like(db); ub[row_perm] = db; db = ub
    _accum_grad(weight, dW)
    if bias is not None:
        _accum_grad(bias, db)


class LNLinearFn(torch.autograd.Function):
    """(Linear(LayerNorm(x)), x): the pre-LN projection of a transformer block (transformer.py:474-490: input_layernorm ->
    query_key_value) as ONE node that keeps only x and the row statistics.  The normalised activations -- a [tokens, h] tensor per layer
    that `LayerNormFn` + `LinearFn` would hold for the weight gradient -- are rebuilt by the (HBM-bound) LayerNorm kernel in the backward.
    The second output hands the residual stream through, so its gradient is folded into the LayerNorm backward kernel.  Part of the
    selective activation retention of `ParallelTransformerLayer` (mode 'selective')."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, weight, bias, row_perm):
        _check_bf16(x)
        H = x.shape[-1]
        x2 = x.reshape(-1, H)
        if not x2.is_contiguous():
            raise ValueError("input must be contiguous")
        M, N = x2.shape[0], weight.shape[0]
        ln, mean, rstd = _ln_forward(x2, gamma, beta, eps)
        y = torch.empty((M, N), dtype=BF16, device=x.device)
        wb = w_bf16(weight) if row_perm is None else w_bf16_perm(weight, row_perm)
        bb = None
        if bias is not None:
            bb = bias.detach() if row_perm is None else WEIGHTS.get(bias, "perm", lambda: bias.detach()[row_perm].contiguous())
        gemm_nt(ln, H, wb, H, y, N, M, N, H, bias=bb)
        ctx.save_for_backward(x2, mean, rstd)
        ctx.gamma, ctx.beta, ctx.eps, ctx.weight, ctx.bias, ctx.row_perm, ctx.shp = gamma, beta, eps, weight, bias, row_perm, x.shape
        return y.reshape(x.shape[:-1] + (N,)), x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dpass):
        x2, mean, rstd = ctx.saved_tensors
        weight, bias, row_perm = ctx.weight, ctx.bias, ctx.row_perm
        M, H = x2.shape
        N = weight.shape[0]
        dy2 = dy.reshape(M, N)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        ln, _, _ = _ln_forward(x2, ctx.gamma, ctx.beta, ctx.eps)                           # rebuilt, not stored
        _linear_param_grads(dy2, ln, weight, bias, row_perm)
        del ln
        wt = w_bf16_t(weight) if row_perm is None else WEIGHTS.get(weight, "perm_t", lambda: transpose(w_bf16_perm(weight, row_perm)))
        dln = matmul_nt(dy2, wt)
        dres = dpass.reshape(M, H).contiguous() if dpass is not None else None
        dx = _ln_backward(dln, x2, ctx.gamma, ctx.beta, mean, rstd, dres)
        return dx.reshape(ctx.shp), None, None, None, None, None, None


def ln_linear(x, gamma, beta, eps, weight, bias, row_perm=None):
    return LNLinearFn.apply(x, gamma, beta, eps, weight, bias, row_perm)


class LNMLPFn(torch.autograd.Function):
    """x + dropout(gelu(LayerNorm(x) W1^T + b1) W2^T + b2): post-attention LayerNorm + ParallelMLP + bias-dropout-add (transformer.py:94-108,
    397-413,545-563) as ONE node that keeps only x and the row statistics.  The three [tokens, h] / [tokens, 4h] tensors a backward needs --
    normalised input, FFN pre-activation, GELU output: 9 of the ~16 [tokens, h]-sized tensors a layer would keep -- are rebuilt in the
    backward by the LayerNorm kernel and ONE GEMM (h -> 4h with the bias + GELU (+ pre-activation) epilogue): a third of the layer's forward
    flops instead of all of them (per-layer recompute) or none (keeping 32 GB per layer at the benchmark shape)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, w1, b1, w2, b2, drop_p, seed, grad_on=True):
        _check_bf16(x)
        H = x.shape[-1]
        x2 = x.reshape(-1, H)
        if not x2.is_contiguous():
            raise ValueError("input must be contiguous")
        M, F = x2.shape[0], w1.shape[0]
        ln, mean, rstd = _ln_forward(x2, gamma, beta, eps)
        inter = torch.empty((M, F), dtype=BF16, device=x.device)
        gemm_nt(ln, H, w_bf16(w1), H, inter, F, M, F, H, bias=b1.detach(), gelu=True)
        del ln
        y = torch.empty((M, H), dtype=BF16, device=x.device)
        gemm_nt(inter, F, w_bf16(w2), F, y, H, M, H, F, bias=b2.detach(), residual=x2, drop_p=drop_p, seed=seed)
        if grad_on:
            PREMASK.register(y, drop_p, seed)
        ctx.save_for_backward(x2, mean, rstd)
        ctx.params, ctx.eps, ctx.shp, ctx.drop_p, ctx.seed = (gamma, beta, w1, b1, w2, b2), eps, x.shape, drop_p, seed
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, mean, rstd = ctx.saved_tensors
        gamma, beta, w1, b1, w2, b2 = ctx.params
        M, H = x2.shape
        F = w1.shape[0]
        if M % 32:
            raise ValueError("token count must be a multiple of 32 for the weight-gradient GEMM")
        dy2 = dy.reshape(M, H)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dres = dy2                                                                        # the residual branch: y = x + ...
        if ctx.drop_p > 0.0:
            dy2 = _dropout_mask_of(dy2, H, ctx.drop_p, ctx.seed)
        # rebuild LayerNorm output, FFN pre-activation and GELU output
        RECOMPUTE.active += 1
        try:
            ln, _, _ = _ln_forward(x2, gamma, beta, ctx.eps)
            pre = torch.empty((M, F), dtype=BF16, device=dy.device)
            inter = torch.empty((M, F), dtype=BF16, device=dy.device)
            gemm_nt(ln, H, w_bf16(w1), H, inter, F, M, F, H, bias=b1.detach(), gelu=2, pre_act=pre)        # pre <- gelu'(pre-activation)
        finally:
            RECOMPUTE.active -= 1
        _linear_grads_into(dy2, inter, w2, b2)
        del inter
        dpre = torch.empty((M, F), dtype=BF16, device=dy.device)
        gemm_nt(dy2, H, w_bf16_t(w2), H, dpre, F, M, F, H, residual=pre, residual_mode=2)         # (dy W2) * gelu'(pre) in the epilogue
        del pre
        _linear_grads_into(dpre, ln, w1, b1)
        del ln
        dln = matmul_nt(dpre, w_bf16_t(w1))
        del dpre
        dx = _ln_backward(dln, x2, gamma, beta, mean, rstd, dres)
        return dx.reshape(ctx.shp), None, None, None, None, None, None, None, None, None, None


def ln_mlp(x, gamma, beta, eps, w1, b1, w2, b2, drop_p=0.0, seed=0):
    return LNMLPFn.apply(x, gamma, beta, eps, w1, b1, w2, b2, drop_p, seed, torch.is_grad_enabled())


def layer_norm(x, gamma, beta, eps=1e-5):
    if _f32(x):
        return _F32().layer_norm(x, gamma, beta, eps)
    return LayerNormFn.apply(x, gamma, beta, eps)


def layer_norm_residual(x, gamma, beta, eps=1e-5):
    """(LayerNorm(x), x): use the second output as the residual operand of the block that consumes the first."""
    if _f32(x):
        return _F32().layer_norm_residual(x, gamma, beta, eps)
    return LayerNormFn.apply(x, gamma, beta, eps, True)


class _AttentionStash(object):
    """Selective activation recompute: under per-layer checkpointing the whole layer is re-run in the backward, but the fused attention
    forward is the one piece whose output is small ([b, s, h] + two fp32 statistics per row) next to its cost (it runs three times per
    layer and step otherwise).  The checkpoint wrapper (transformer._CheckpointedLayer) puts the stash in 'store' mode for the first run
    of a layer and in 'consume' mode for its re-run; the key ties an entry to that one wrapper call, so other passes through the same
    module (the no-grad one-context pass) never see it.  Costs ~45 GB of the 288 GB at the benchmark shape, saves one attention forward
    per layer and step."""

    def __init__(self):
        self.mode, self.key, self.store, self.enabled = None, None, {}, True


ATTN_STASH = _AttentionStash()


def _attn_side(x, info):
    """Launch description of one attention operand: x [b, s, np, hn] over dense token ids [b, s], or x [rows, np, hn] over a PackedSeqs.
    -> (ids, cu, batch stride, sequence stride, head stride, batch, longest sequence)."""
    if isinstance(info, PackedSeqs):
        if x.dim() != 3 or x.shape[0] != info.rows:
            raise ValueError("a packed attention operand must be [rows, heads, head_dim]")
        return info.ids, info.cu, 0, x.stride(0), x.stride(1), info.n, info.max_len
    if info.dtype != torch.int64 or not info.is_contiguous() or not info.is_cuda:
        raise TypeError("token ids must be contiguous CUDA int64 tensors (the kernels derive the masks from them)")
    if x.dim() != 4:
        raise ValueError("a dense attention operand must be [batch, s, heads, head_dim]")
    return info, None, x.stride(0), x.stride(1), x.stride(2), x.shape[0], x.shape[1]


_SPLITKV_PLANS = {}


def _splitkv_plan(batch, heads, sq, sk):
    """(ksplit, forward workspace bytes, backward workspace bytes) of a fused attention launch with dense queries [batch, sq] over up to sk
    keys per sequence (include/emdr2_ops.h: emdr2_attention_splitkv_plan); ksplit 1 = the launch fills the chip by itself."""
    key = (batch, heads, sq, sk)
    plan = _SPLITKV_PLANS.get(key)
    if plan is None:
        import ctypes
        ks, fb, bb = ctypes.c_int(), ctypes.c_size_t(), ctypes.c_size_t()
        _native.check(_lib().emdr2_attention_splitkv_plan(batch, heads, sq, sk, ctypes.byref(ks), ctypes.byref(fb), ctypes.byref(bb)), "attention_splitkv_plan")
        plan = _SPLITKV_PLANS[key] = (ks.value, fb.value, bb.value)
    return plan


class AttentionCoreFn(torch.autograd.Function):
    """dropout(softmax(mask(Q K^T / sqrt(hn)))) V for all heads (transformer.py:283-381).
    Inputs are the projection outputs themselves: self-attention passes `qsrc` = the packed [b, s, 3, np, hn] QKV tensor (kvsrc None),
    cross-attention `qsrc` = [b, sq, np, hn] and `kvsrc` = the packed [b, sk, 2, np, hn] KV tensor.  The kernels read q, k, v as strided
    slices and the backward writes dq, dk, dv straight into ONE packed gradient (no select_backward zero-fill + add chains).
    `ids_q` / `ids_k` are dense token ids [b, s] (masks: pad id 0 + optional history mask) or a PackedSeqs, in which case that side's
    tensors are [rows, ...] without a batch dimension (sequence b = rows cu[b] .. cu[b+1]).  hn == 64 runs the fused kernels
    (attention.hip, attention_bwd.hip; dense keys need sk % 32 == 0); other dense shapes run QK^T GEMM + softmax kernel + PV GEMM.
    Only the row statistics (max, sum-exp) and the output are kept; the backward rebuilds the probabilities from them."""

    @staticmethod
    def forward(ctx, qsrc, kvsrc, ids_q, ids_k, causal, drop_p=0.0, seed=0, site=0):
        _check_bf16(qsrc, kvsrc)
        if kvsrc is None:
            q, k, v = qsrc.select(-3, 0), qsrc.select(-3, 1), qsrc.select(-3, 2)
        else:
            q, k, v = qsrc, kvsrc.select(-3, 0), kvsrc.select(-3, 1)
        pq, pk = isinstance(ids_q, PackedSeqs), isinstance(ids_k, PackedSeqs)
        iq, cq, q_sb, q_ss, q_sn, b, sq = _attn_side(q, ids_q)
        ik, ck, k_sb, k_ss, k_sn, bk, sk = _attn_side(k, ids_k)
        _, _, v_sb, v_ss, v_sn, _, _ = _attn_side(v, ids_k)
        if b != bk:
            raise ValueError("query and key batches differ")
        heads, hn = q.shape[-2:]
        dev = q.device
        scale = 1.0 / math.sqrt(hn)
        ctx.causal, ctx.drop_p, ctx.seed = causal, float(drop_p), int(seed)
        ctx.ids_q, ctx.ids_k = ids_q, ids_k
        stash_key = (ATTN_STASH.key, site) if (ATTN_STASH.enabled and ATTN_STASH.mode and site) else None
        stashed = ATTN_STASH.store.pop(stash_key, None) if (stash_key and ATTN_STASH.mode == 'consume') else None
        if stashed is not None:                                                           # the layer's re-run: reuse the first run's output
            ctxo, m, l = stashed
            ctx.save_for_backward(qsrc, kvsrc, m, l, ctxo)
            ctx.ksplit = _splitkv_plan(b, heads, sq, sk)[0] if (hn == 64 and not pq and sk <= 65536 and (pk or sk % 32 == 0)) else 1
            return ctxo
        m = torch.empty((heads, ids_q.rows) if pq else (b, heads, sq), dtype=torch.float32, device=dev)
        l = torch.empty_like(m)
        ctxo = torch.empty(tuple(q.shape[:-2]) + (heads, hn), dtype=BF16, device=dev)
        fused = hn == 64 and sk <= 65536 and (pk or sk % 32 == 0)
        if (pq or pk) and not fused:
            raise ValueError("packed attention operands need head dim 64 (the fused kernels)")
        # few dense queries over very many keys (the FiD decoder's cross-attention, cached decoding steps): keys dealt to several workgroups
        ksplit, ws_f, _ = _splitkv_plan(b, heads, sq, sk) if (fused and not pq) else (1, 0, 0)
        ctx.ksplit = ksplit
        if ksplit > 1:
            ws = torch.empty(ws_f, dtype=torch.uint8, device=dev)
            pairs = sq * ids_k.total if pk else b * sq * sk
            _native.check(_lib().emdr2_attention_fwd_splitkv(q.data_ptr(), q_sb, q_ss, q_sn, k.data_ptr(), k_sb, k_ss, k_sn, v.data_ptr(), v_sb, v_ss, v_sn,
                                                             ctxo.data_ptr(), iq.data_ptr(), ik.data_ptr(), _ptr(ck), int(pairs), b, heads, sq, sk, hn,
                                                             int(causal), scale, float(drop_p), int(seed), m.data_ptr(), l.data_ptr(), ksplit,
                                                             ws.data_ptr(), ws_f, _sp()), "attention_fwd_splitkv")
        elif pq or pk:
            pairs = ids_q.pairs if (pq and ids_k is ids_q) else (sq * ids_k.total if not pq else 0)
            _native.check(_lib().emdr2_attention_varlen_fwd(q.data_ptr(), q_sb, q_ss, q_sn, k.data_ptr(), k_sb, k_ss, k_sn, v.data_ptr(), v_sb, v_ss, v_sn,
                                                            ctxo.data_ptr(), iq.data_ptr(), ik.data_ptr(), _ptr(cq), _ptr(ck), ids_q.rows if pq else 0,
                                                            int(pairs or 0), b, heads, sq, sk, hn, int(causal), scale, float(drop_p), int(seed),
                                                            m.data_ptr(), l.data_ptr(), _sp()), "attention_varlen_fwd")
            if pq:
                ids_q.zero_tail(ctxo)
        elif fused:
            _native.check(_lib().emdr2_attention_fwd(q.data_ptr(), q_sb, q_ss, q_sn, k.data_ptr(), k_sb, k_ss,
                                                     k_sn, v.data_ptr(), v_sb, v_ss, v_sn, ctxo.data_ptr(), iq.data_ptr(), ik.data_ptr(), b, heads, sq,
                                                     sk, hn, int(causal), scale, float(drop_p), int(seed), m.data_ptr(), l.data_ptr(), _sp()),
                          "attention_fwd")
        else:
            if sk % 32:
                raise ValueError("dense attention operands need a key length that is a multiple of 32 (got %d: K x --seq-length; the reference's "
                                 "sequence lengths 256 / 512 are; packed layouts take any length)" % sk)
            vT = head_transpose(v, b, sk, heads, hn)
            S = torch.empty((b, heads, sq, sk), dtype=BF16, device=dev)
            gemm_nt(q, q.stride(1), k, k.stride(1), S, sk, sq, sk, hn, b, q.stride(0), k.stride(0), heads * sq * sk, heads, q.stride(2),
                    k.stride(2), sq * sk, alpha=scale)
            _native.check(_lib().emdr2_softmax_mask_fwd(S.data_ptr(), iq.data_ptr(), ik.data_ptr(), b, heads, sq, sk, int(causal),
                                                        m.data_ptr(), l.data_ptr(), float(drop_p), int(seed), _sp()), "softmax_fwd")
            gemm_nt(S, sk, vT, sk, ctxo, heads * hn, sq, hn, sk, b, heads * sq * sk, heads * hn * sk, sq * heads * hn, heads, sq * sk, hn * sk, hn)
        ctx.save_for_backward(qsrc, kvsrc, m, l, ctxo)
        if stash_key and ATTN_STASH.mode == 'store':
            ATTN_STASH.store[stash_key] = (ctxo, m, l)
        return ctxo

    @staticmethod
    def backward(ctx, dctx):
        qsrc, kvsrc, m, l, ctxo = ctx.saved_tensors
        ids_q, ids_k = ctx.ids_q, ctx.ids_k
        if kvsrc is None:
            q, k, v = qsrc.select(-3, 0), qsrc.select(-3, 1), qsrc.select(-3, 2)
            dqsrc, dkvsrc = torch.empty_like(qsrc), None
            dq, dk, dv = dqsrc.select(-3, 0), dqsrc.select(-3, 1), dqsrc.select(-3, 2)
        else:
            q, k, v = qsrc, kvsrc.select(-3, 0), kvsrc.select(-3, 1)
            dqsrc, dkvsrc = torch.empty_like(qsrc), torch.empty_like(kvsrc)
            dq, dk, dv = dqsrc, dkvsrc.select(-3, 0), dkvsrc.select(-3, 1)
        pq, pk = isinstance(ids_q, PackedSeqs), isinstance(ids_k, PackedSeqs)
        iq, cq, q_sb, q_ss, q_sn, b, sq = _attn_side(q, ids_q)
        ik, ck, k_sb, k_ss, k_sn, _, sk = _attn_side(k, ids_k)
        _, _, v_sb, v_ss, v_sn, _, _ = _attn_side(v, ids_k)
        _, _, dq_sb, dq_ss, _, _, _ = _attn_side(dq, ids_q)
        _, _, dkv_sb, dkv_ss, _, _, _ = _attn_side(dk, ids_k)
        heads, hn = q.shape[-2:]
        dev = q.device
        causal = int(ctx.causal)
        scale = 1.0 / math.sqrt(hn)
        dctx = dctx.contiguous()
        H = heads * hn
        lib = _lib()
        # scratch of the fused kernels: four per-query statistics (exp2 offset, D = rowsum(dout * o), dropout row hash, real-token flag) the dq
        # kernel leaves for the dk / dv kernel (include/emdr2_ops.h); the unfused path below uses the first [b, heads, sq] as D
        D = torch.empty((4,) + tuple(m.shape), dtype=torch.float32, device=dev)
        if getattr(ctx, "ksplit", 1) > 1:
            _, _, ws_b = _splitkv_plan(b, heads, sq, sk)
            ws = torch.empty(ws_b, dtype=torch.uint8, device=dev)
            pairs = sq * ids_k.total if pk else b * sq * sk
            _native.check(lib.emdr2_attention_bwd_splitkv(q.data_ptr(), q_sb, q_ss, q_sn, k.data_ptr(), k_sb, k_ss, k_sn, v.data_ptr(), v_sb, v_ss, v_sn,
                                                          ctxo.data_ptr(), dctx.data_ptr(), dq.data_ptr(), dq_sb, dq_ss, dk.data_ptr(), dv.data_ptr(), dkv_sb,
                                                          dkv_ss, iq.data_ptr(), ik.data_ptr(), _ptr(ck), int(pairs), m.data_ptr(), l.data_ptr(), D.data_ptr(),
                                                          b, heads, sq, sk, hn, causal, scale, ctx.drop_p, ctx.seed, ctx.ksplit, ws.data_ptr(), ws_b, _sp()),
                          "attention_bwd_splitkv")
            if pk and dkvsrc is not None:
                ids_k.zero_tail(dkvsrc)
            return dqsrc, dkvsrc, None, None, None, None, None, None
        if pq or pk:
            pairs = ids_q.pairs if (pq and ids_k is ids_q) else (sq * ids_k.total if not pq else 0)
            _native.check(lib.emdr2_attention_varlen_bwd(q.data_ptr(), q_sb, q_ss, q_sn, k.data_ptr(), k_sb, k_ss, k_sn, v.data_ptr(), v_sb, v_ss, v_sn,
                                                         ctxo.data_ptr(), dctx.data_ptr(), dq.data_ptr(), dq_sb, dq_ss, dk.data_ptr(), dv.data_ptr(), dkv_sb,
                                                         dkv_ss, iq.data_ptr(), ik.data_ptr(), _ptr(cq), _ptr(ck), ids_q.rows if pq else 0, int(pairs or 0),
                                                         m.data_ptr(), l.data_ptr(), D.data_ptr(), b, heads, sq, sk, hn, causal, scale, ctx.drop_p,
                                                         ctx.seed, _sp()), "attention_varlen_bwd")
            if pq:                                                                        # tail rows feed the weight-gradient GEMMs: zeros
                ids_q.zero_tail(dqsrc)
            if pk and dkvsrc is not None:
                ids_k.zero_tail(dkvsrc)
            return dqsrc, dkvsrc, None, None, None, None, None, None
        if hn == 64 and sk % 32 == 0 and sk <= 65536:                                     # fused: no [sq, sk] matrix, no operand transposes
            _native.check(lib.emdr2_attention_bwd(q.data_ptr(), q.stride(0), q.stride(1), q.stride(2), k.data_ptr(), k.stride(0), k.stride(1),
                                                  k.stride(2), v.data_ptr(), v.stride(0), v.stride(1), v.stride(2), ctxo.data_ptr(), dctx.data_ptr(),
                                                  dq.data_ptr(), dq.stride(0), dq.stride(1), dk.data_ptr(), dv.data_ptr(), dk.stride(0),
                                                  dk.stride(1), iq.data_ptr(), ik.data_ptr(), m.data_ptr(), l.data_ptr(), D.data_ptr(), b,
                                                  heads, sq, sk, hn, causal, scale, ctx.drop_p, ctx.seed, _sp()), "attention_bwd")
            return dqsrc, dkvsrc, None, None, None, None, None, None
        ids_q, ids_k = iq, ik
        # main orientation: S = scale Q K^T (recomputed), dP = dctx V^T, dS = P (dP_eff - D) with P rebuilt from (m, l)
        S = torch.empty((b, heads, sq, sk), dtype=BF16, device=dev)
        gemm_nt(q, q.stride(1), k, k.stride(1), S, sk, sq, sk, hn, b, q.stride(0), k.stride(0), heads * sq * sk, heads, q.stride(2), k.stride(2),
                sq * sk, alpha=scale)
        dP = torch.empty((b, heads, sq, sk), dtype=BF16, device=dev)
        gemm_nt(dctx, H, v, v.stride(1), dP, sk, sq, sk, hn, b, sq * H, v.stride(0), heads * sq * sk, heads, hn, v.stride(2), sq * sk)
        _native.check(lib.emdr2_softmax_mask_bwd(S.data_ptr(), dP.data_ptr(), ids_q.data_ptr(), ids_k.data_ptr(), m.data_ptr(), l.data_ptr(), b,
                                                 heads, sq, sk, causal, ctx.drop_p, ctx.seed, D.data_ptr(), _sp()), "softmax_bwd")
        del S
        kT = head_transpose(k, b, sk, heads, hn)
        gemm_nt(dP, sk, kT, sk, dq, dq.stride(1), sq, hn, sk, b, heads * sq * sk, heads * hn * sk, dq.stride(0), heads, sq * sk, hn * sk, hn,
                alpha=scale)
        del dP
        # transposed orientation: S^T, dP^T recomputed in the layout dK / dV need (no [sq, sk] transposes)
        St = torch.empty((b, heads, sk, sq), dtype=BF16, device=dev)
        gemm_nt(k, k.stride(1), q, q.stride(1), St, sq, sk, sq, hn, b, k.stride(0), q.stride(0), heads * sk * sq, heads, k.stride(2), q.stride(2),
                sk * sq, alpha=scale)
        dPt = torch.empty_like(St)
        gemm_nt(v, v.stride(1), dctx, H, dPt, sq, sk, sq, hn, b, v.stride(0), sq * H, heads * sk * sq, heads, v.stride(2), hn, sk * sq)
        _native.check(lib.emdr2_softmax_mask_t(St.data_ptr(), dPt.data_ptr(), ids_q.data_ptr(), ids_k.data_ptr(), m.data_ptr(), l.data_ptr(),
                                               D.data_ptr(), b, heads, sq, sk, causal, ctx.drop_p, ctx.seed, _sp()), "softmax_t")
        qT = head_transpose(q, b, sq, heads, hn)
        gemm_nt(dPt, sq, qT, sq, dk, dk.stride(1), sk, hn, sq, b, heads * sk * sq, heads * hn * sq, dk.stride(0), heads, sk * sq, hn * sq, hn,
                alpha=scale)
        dctxT = head_transpose(dctx.view(b, sq, heads, hn), b, sq, heads, hn)
        gemm_nt(St, sq, dctxT, sq, dv, dv.stride(1), sk, hn, sq, b, heads * sk * sq, heads * hn * sq, dv.stride(0), heads, sk * sq, hn * sq, hn)
        return dqsrc, dkvsrc, None, None, None, None, None, None


def attention_core(qsrc, kvsrc, ids_q, ids_k, causal=False, drop_p=0.0, seed=0, site=0):
    """qsrc packed [b, s, 3, np, hn] with kvsrc None (self-attention), or qsrc [b, sq, np, hn] + kvsrc packed [b, sk, 2, np, hn]."""
    if _f32(qsrc):
        return _F32().attention_core(qsrc, kvsrc, ids_q, ids_k, causal, drop_p, seed, site)
    return AttentionCoreFn.apply(qsrc, kvsrc, ids_q, ids_k, causal, drop_p, seed, site)


class EmbeddingFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, types, W, P, T, drop_p=0.0, seed=0, seqs=None):
        H = W.shape[1]
        lib = _lib()
        if seqs is not None:                                     # packed rows: ids / types come from the layout, positions from its row map
            ids, types, S = seqs.ids, seqs.types, seqs.S
            if types is not None and T is None:
                raise ValueError("the packed layout carries token types but no type table was passed")
            out = torch.empty((seqs.rows, H), dtype=BF16, device=ids.device)
            _native.check(lib.emdr2_embedding_packed_fwd(ids.data_ptr(), _ptr(types), seqs.rowmap.data_ptr(), w_bf16(W).data_ptr(), w_bf16(P).data_ptr(),
                                                         w_bf16(T).data_ptr() if types is not None else None, out.data_ptr(), seqs.rows, S, H,
                                                         float(drop_p), int(seed), _sp()), "embedding_packed_fwd")
        else:
            b, s = ids.shape
            out = torch.empty((b, s, H), dtype=BF16, device=ids.device)
            ids = ids.contiguous()
            types = types.contiguous() if types is not None else None
            _native.check(lib.emdr2_embedding_fwd(ids.data_ptr(), _ptr(types), w_bf16(W).data_ptr(), w_bf16(P).data_ptr(),
                                                  w_bf16(T).data_ptr() if types is not None else None, out.data_ptr(), b * s, s, H, float(drop_p),
                                                  int(seed), _sp()), "embedding_fwd")
        ctx.ids, ctx.types, ctx.W, ctx.P, ctx.T, ctx.drop_p, ctx.seed, ctx.seqs = ids, types, W, P, T, drop_p, seed, seqs
        return out

    @staticmethod
    def backward(ctx, dout):
        ids, types, W, P, T, seqs = ctx.ids, ctx.types, ctx.W, ctx.P, ctx.T, ctx.seqs
        H = W.shape[1]
        dout = dout.contiguous()
        (dW, dW_direct), (dP, dP_direct) = _grad_buffer(W), _grad_buffer(P)            # the kernels scatter-add their rows atomically
        dT, dT_direct = _grad_buffer(T) if types is not None else (None, False)
        if seqs is not None:
            _native.check(_lib().emdr2_embedding_packed_bwd(ids.data_ptr(), _ptr(types), seqs.cu.data_ptr(), seqs.n, dout.data_ptr(), dW.data_ptr(),
                                                            dP.data_ptr(), _ptr(dT), seqs.rows, seqs.S, H, T.shape[0] if dT is not None else 0,
                                                            float(ctx.drop_p), int(ctx.seed), _sp()), "embedding_packed_bwd")
        else:
            b, s = ids.shape
            _native.check(_lib().emdr2_embedding_bwd(ids.data_ptr(), _ptr(types), dout.data_ptr(), dW.data_ptr(), dP.data_ptr(), _ptr(dT), b * s, s, H,
                                                     T.shape[0] if dT is not None else 0, float(ctx.drop_p), int(ctx.seed), _sp()), "embedding_bwd")
        _deliver_grad(W, dW, dW_direct)
        _deliver_grad(P, dP, dP_direct)
        if dT is not None:
            _deliver_grad(T, dT, dT_direct)
        return None, None, None, None, None, None, None, None


def embedding(ids, types, W, P, T, drop_p=0.0, seed=0, seqs=None, fp32=False):
    """ids [b, s] (+ types) -> [b, s, H]; or `seqs` = a PackedSeqs -> [rows, H] (ids / types are taken from the layout).
    `fp32`: emit fp32 activations -- the validation-only fp32 path (kernels_f32.py), which everything downstream then follows."""
    if fp32:
        return _F32().embedding(ids, types, W, P, T, drop_p, seed, seqs)
    return EmbeddingFn.apply(ids, types, W, P, T, drop_p, seed, seqs)


class LseGatherFn(torch.autograd.Function):
    """gold[row] = log_softmax(logits[row])[label[row]] over the vocabulary (fp32 out)."""

    @staticmethod
    def forward(ctx, logits, labels):
        _check_bf16(logits)
        V = logits.shape[-1]
        l2 = logits.reshape(-1, V)
        lab = labels.reshape(-1).contiguous()
        rows = l2.shape[0]
        gold = torch.empty(rows, dtype=torch.float32, device=logits.device)
        lse = torch.empty_like(gold)
        _native.check(_lib().emdr2_lse_gather_fwd(l2.data_ptr(), lab.data_ptr(), gold.data_ptr(), lse.data_ptr(), rows, V, _sp()), "lse_gather_fwd")
        ctx.save_for_backward(l2, lab, lse)
        ctx.shape = logits.shape
        return gold.reshape(labels.shape)

    @staticmethod
    def backward(ctx, dgold):
        l2, lab, lse = ctx.saved_tensors
        rows, V = l2.shape
        w = dgold.reshape(-1).to(torch.float32).contiguous()
        dl = torch.empty_like(l2)
        _native.check(_lib().emdr2_lse_gather_bwd(l2.data_ptr(), lab.data_ptr(), lse.data_ptr(), w.data_ptr(), dl.data_ptr(), rows, V, _sp()),
                      "lse_gather_bwd")
        return dl.reshape(ctx.shape), None


def lse_gather(logits, labels):
    if _f32(logits):
        return _F32().lse_gather(logits, labels)
    return LseGatherFn.apply(logits, labels)


def lm_head_gold_logprob(hidden, weight, bias, labels):
    """gold[row] = log_softmax(hidden[row] W^T + b)[labels[row]] (fp32) WITHOUT materialising the [rows, vocab] logits: the tied LM head
    (language_model.py:28-41) and the log-softmax + gather of train_e2eqa.py:79-96 in one GEMM whose epilogue keeps, per row and 64-column
    block, only (max, sum exp) and the gold logit (csrc/gemm8.hip LSE mode) plus a small combine kernel.  Logits are rounded to bf16 inside
    the epilogue exactly as the unfused path stores them, so both paths agree.  No gradient: this is the no-grad one-context pass
    (emdr2_model.py:185-210), whose [B, K, L, V] logits are 6.3 GB at the benchmark shape."""
    if _f32(hidden):
        return _F32().lm_head_gold_logprob(hidden, weight, bias, labels)
    _check_bf16(hidden)
    V, H = weight.shape
    h2 = hidden.reshape(-1, H)
    M = h2.shape[0]
    lab = labels.reshape(-1).contiguous()
    if lab.numel() != M or lab.dtype != torch.int64:
        raise ValueError("one int64 label per row expected")
    if M % 256 or V % 256 or H % 128 or not h2.is_contiguous():
        return lse_gather(linear(h2, weight, bias), lab).reshape(labels.shape)           # shapes the fused kernel does not take
    slots = V // 64
    dev = hidden.device
    pmax = torch.empty((M, slots), dtype=torch.float32, device=dev)
    psum = torch.empty_like(pmax)
    # the epilogue writes gold[row] only when the label falls on one of the V columns: zero-initialised, so a label outside [0, V) (an
    # ignore_index such as -100) yields gold = 0 - logsumexp instead of whatever the allocator handed out; callers mask such rows anyway
    gold = torch.zeros(M, dtype=torch.float32, device=dev)
    out = torch.empty_like(gold)
    _native.check(_lib().emdr2_gemm_nt_lse_bf16(h2.data_ptr(), H, w_bf16(weight).data_ptr(), H, M, V, H, 1.0, _ptr(bias.detach() if bias is not None else None),
                                                lab.data_ptr(), pmax.data_ptr(), psum.data_ptr(), gold.data_ptr(), _sp()), "gemm_nt_lse")
    _native.check(_lib().emdr2_lse_combine(pmax.data_ptr(), psum.data_ptr(), gold.data_ptr(), out.data_ptr(), None, M, slots, _sp()), "lse_combine")
    return out.reshape(labels.shape)


class RetrieverPriorFn(torch.autograd.Function):
    """topk_log_probs [B, K] = log_softmax_k(<q_b, c_bk> * scale) (emdr2_model.py:134-145) on the HIP kernel pair of elementwise.hip."""

    @staticmethod
    def forward(ctx, q, c, scale):
        _check_bf16(q, c)
        q, c = q.contiguous(), c.contiguous()
        B, Kk, H = c.shape
        logp = torch.empty((B, Kk), dtype=torch.float32, device=q.device)
        prob = torch.empty_like(logp)
        _native.check(_lib().emdr2_retriever_prior_fwd(q.data_ptr(), c.data_ptr(), logp.data_ptr(), prob.data_ptr(), B, Kk, H, float(scale), _sp()),
                      "retriever_prior_fwd")
        ctx.save_for_backward(q, c, prob)
        ctx.scale = float(scale)
        return logp

    @staticmethod
    def backward(ctx, dlogp):
        q, c, prob = ctx.saved_tensors
        B, Kk, H = c.shape
        dq = torch.empty_like(q) if ctx.needs_input_grad[0] else None
        dc = torch.empty_like(c) if ctx.needs_input_grad[1] else None
        if dq is None and dc is None:
            return None, None, None
        g = dlogp.to(torch.float32).contiguous()
        _native.check(_lib().emdr2_retriever_prior_bwd(g.data_ptr(), prob.data_ptr(), q.data_ptr(), c.data_ptr(), _ptr(dq), _ptr(dc), B, Kk, H, ctx.scale,
                                                       _sp()), "retriever_prior_bwd")
        return dq, dc, None


def retriever_prior(q, c, scale):
    if _f32(q):
        return _F32().retriever_prior(q, c, scale)
    return RetrieverPriorFn.apply(q, c, scale)


class MarginalFn(torch.autograd.Function):
    """marginal [B, L] = logsumexp_k(prior[b, k] + gold[b, k, l]) (train_e2eqa.py:98-123); gold comes from the no-grad pass (constant)."""

    @staticmethod
    def forward(ctx, prior, gold):
        prior, gold = prior.to(torch.float32).contiguous(), gold.to(torch.float32).contiguous()
        B, Kk, L = gold.shape
        out = torch.empty((B, L), dtype=torch.float32, device=gold.device)
        _native.check(_lib().emdr2_marginal_fwd(prior.data_ptr(), gold.data_ptr(), out.data_ptr(), B, Kk, L, _sp()), "marginal_fwd")
        ctx.save_for_backward(prior, gold, out)
        return out

    @staticmethod
    def backward(ctx, dm):
        prior, gold, out = ctx.saved_tensors
        B, Kk, L = gold.shape
        dprior = torch.empty_like(prior)
        dm = dm.to(torch.float32).contiguous()
        _native.check(_lib().emdr2_marginal_bwd(prior.data_ptr(), gold.data_ptr(), out.data_ptr(), dm.data_ptr(), dprior.data_ptr(), B, Kk, L, _sp()),
                      "marginal_bwd")
        return dprior, None


def marginal_logsumexp(prior, gold):
    return MarginalFn.apply(prior, gold)
