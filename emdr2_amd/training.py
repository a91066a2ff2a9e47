"""Optimizer step of the hot path (reference: megatron/training.py:73-230, megatron/fp16/fp16.py, apex FusedAdam, mpu/grads.py,
megatron/model/distributed.py, megatron/learning_rates.py), bf16 edition: parameters ARE the fp32 masters, gradients arrive in fp32
from the weight-gradient GEMMs, so there is no loss scaling, no overflow check and no master-copy round trip."""
import torch

from emdr2_amd import _native


def _no_weight_decay(name):
    """megatron/model/utils.py:64-83 walks the modules: every parameter of a LayerNorm and every parameter NAMED `bias` goes without weight
    decay.  All LayerNorm modules of the EMDR2 model have 'layernorm' in their name; pinned on the reference's own grouping of its
    EMDR2Model by tests/golden/optim_groups.json."""
    return name == "bias" or name.endswith(".bias") or "layernorm" in name


def get_params_for_weight_decay_optimization(module):
    """(decayed group, undecayed group) like the reference's function (megatron/model/utils.py:64-83)."""
    decay, no_decay = [], []
    for name, p in module.named_parameters():
        (no_decay if _no_weight_decay(name) else decay).append(p)
    return {"params": decay}, {"params": no_decay, "weight_decay": 0.0}


class AnnealingLR(object):
    """Linear warm-up then linear decay (learning_rates.py:51-71, decay_style 'linear')."""

    def __init__(self, start_lr, warmup_iter, total_iters, min_lr=0.0):
        self.start_lr, self.warmup_iter, self.end_iter, self.min_lr, self.num_iters = start_lr, warmup_iter, total_iters, min_lr, 0

    def get_lr(self):
        n_ = min(self.num_iters, self.end_iter - self.warmup_iter)
        if self.warmup_iter > 0 and self.num_iters <= self.warmup_iter:
            return float(self.start_lr) * n_ / self.warmup_iter
        n_ = n_ - self.warmup_iter
        return max(self.start_lr * (self.end_iter - n_) / self.end_iter, self.min_lr)

    def step(self):
        self.num_iters += 1
        return self.get_lr()

    def state_dict(self):
        """learning_rates.py:86-95 (same keys)."""
        return {'start_lr': self.start_lr, 'warmup_iter': self.warmup_iter, 'num_iters': self.num_iters, 'decay_style': 'linear',
                'end_iter': self.end_iter, 'min_lr': self.min_lr}

    def load_state_dict(self, sd):
        self.num_iters = sd['num_iters']


def allreduce_gradients(module, group=None):
    """LocalDDP.allreduce_params (megatron/model/distributed.py:35-62): one flattened all-reduce of every gradient, pre-divided by the
    data-parallel world size.  RCCL over xGMI on the GPU box, gloo in the CPU tests."""
    if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
        return
    world = torch.distributed.get_world_size(group)
    if world == 1:
        return
    grads = [p.grad for p in module.parameters() if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    flat /= world
    torch.distributed.all_reduce(flat, group=group)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


class StepAborted(RuntimeError):
    """Raised by FlatAdam.finish() on EVERY rank of the data-parallel group when any rank gave the step up (FlatAdam.abort_step(): an
    allocation failure in its forward / backward): the gradient exchange of the step has been completed consistently -- the same
    collectives, in the same order, on all ranks -- and its result is to be discarded by all of them together."""


class FlatAdam(object):
    """The optimizer step AND the data-parallel gradient exchange over FLAT buckets (the reference: apex FusedAdam through
    amp_C.multi_tensor_apply on fp32 masters, fp16/fp16.py:332-354,420-474; global-norm clip via amp_C.multi_tensor_l2norm, mpu/grads.py:74-127;
    one flattened, pre-divided fp16 all-reduce, model/distributed.py:35-62).

    Parameters are re-homed, in reverse registration order (roughly the order their gradients become final), into a few large buckets; each
    bucket holds, back to back for all its tensors: fp32 masters (`p.data` becomes a view), fp32 gradients (`p.grad` is a view), both Adam
    moments, the bf16 working copies the GEMMs read (`kernels.w_bf16(p)` is a view) and, when data parallel, a bf16 exchange buffer.  Inside a
    bucket the decayed parameters come first, so ONE Adam launch per bucket applies the decoupled weight decay to exactly those.  A step is
    then: one sum-of-squares launch and one Adam launch per bucket (the Adam kernel also writes the bf16 working copies: no cast launches) --
    about 20 launches instead of ~6,000 per-parameter ones.

    Data parallel: a bucket whose parameters have all received their last gradient contribution of the step is pre-divided by the world size,
    cast to bf16, all-reduced asynchronously (RCCL over xGMI: 0.88 GB on the wire per step like the reference's fp16 buffer, half of an fp32
    exchange) while the backward keeps computing, and widened back to fp32 in `finish()`.  Every rank receives the same all-reduced bf16 values
    and the global norm is summed deterministically (block partials in index order, buckets in launch order), so replicas stay bit-identical.
    `exchange_dtype="fp32"` all-reduces the fp32 buckets themselves instead (twice the bytes; 24 mantissa bits where bf16 has 8 and the
    reference's fp16 buffer 11 -- the choice is numerics against wire time, the default follows the reference's 16 bits).
    Which parameters receive how many contributions is learned in the first step; if a later step deviates (a branch of the model switched
    on or off), nothing is lost: a contribution that arrives after its bucket left is collected in a side buffer and all-reduced in
    `finish()`, and the pattern is re-learned from that step.  Ranks agree on this: with world > 1 `finish()` starts with ONE small
    all-reduce (MAX) of [abort flag, late-buffer flag per bucket], every rank joins the late-buffer all-reduce of a bucket if any rank has
    one (zeros where it has none), and every rank all-reduces EVERY bucket every step -- the sequence of collectives never depends on what
    a single rank saw.  `abort_step()` is how a rank that cannot finish its backward (allocation failure) leaves the step without leaving
    its peers in a collective: it issues the bucket all-reduces it still owes (buckets always leave in index order), raises the abort
    flag, and every rank's `finish()` raises `StepAborted`.

    Parameters that never receive a gradient (the reader's unused token-type table) are left untouched, like apex / torch skip `grad is None`."""

    def __init__(self, module, lr=2e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.1, clip_grad=1.0, group=None, bucket_bytes=512 << 20,
                 exchange_dtype="bf16"):
        from emdr2_amd.model import kernels
        if exchange_dtype not in ("bf16", "fp32"):
            raise ValueError("exchange_dtype must be 'bf16' or 'fp32'")
        self.lr, self.betas, self.eps, self.weight_decay, self.clip_grad, self.group = lr, betas, eps, weight_decay, clip_grad, group
        self.exchange_dtype, self.pattern_changes = exchange_dtype, 0
        self.step_count = 0
        no_decay = set(id(p) for n, p in module.named_parameters() if self.is_no_decay(n))      # model/utils.py:64-83
        self.params = [p for p in module.parameters() if p.requires_grad][::-1]
        self.slot = {}
        self.buckets = []
        cur, cur_n = [], 0
        for p in self.params:
            n = (p.numel() + 7) // 8 * 8
            if cur and (cur_n + n) * 4 > bucket_bytes:
                self._close(cur, no_decay)
                cur, cur_n = [], 0
            cur.append(p); cur_n += n
        if cur:
            self._close(cur, no_decay)
        dev = self.params[0].device
        # Host tensors: only the gradient-sink / exchange bookkeeping below runs (torch.distributed plumbing, fp32 exchange) -- that is how
        # the N-rank logic is exercised over gloo without a GPU (tests/test_dist_allreduce.py).  There is no host optimizer: step() raises.
        self.on_device = dev.type == "cuda"
        if not self.on_device and exchange_dtype != "fp32":
            raise _native.NativeError("FlatAdam on host tensors is the exchange bookkeeping only: exchange_dtype must be 'fp32' (the bf16 "
                                      "exchange and the optimizer step are HIP kernels)")
        self._gsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self._scratch = torch.zeros(1025, dtype=torch.float32, device=dev)
        self._stale = False
        self.expected = None                  # {param: contributions per step}, learned in the first step
        self.count = {p: 0 for p in self.params}
        self.launched_early = 0
        self.launches_last_step = 0
        self.next_bucket = 0                  # buckets leave in INDEX order, every step, on every rank: the sequence of collectives is static
        self._epoch = 0
        import weakref
        kernels.WEIGHTS.listeners.append(weakref.WeakMethod(self._on_invalidate))
        self.refresh_working_copies()
        self.begin_step()

    is_no_decay = staticmethod(_no_weight_decay)

    # ---- layout ---------------------------------------------------------------------------------------------------------------------
    def _close(self, plist, no_decay):
        dev = plist[0].device
        ordered = [p for p in plist if id(p) not in no_decay] + [p for p in plist if id(p) in no_decay]
        offs, off, split = [], 0, 0
        for p in ordered:
            offs.append(off)
            off += (p.numel() + 7) // 8 * 8                     # 32-byte aligned fp32 / 16-byte aligned bf16 views
            if id(p) not in no_decay:
                split = off
        f32 = lambda: torch.zeros(off, dtype=torch.float32, device=dev)
        b = {"params": ordered, "n": off, "split": split, "master": f32(), "grad": f32(), "m": f32(), "v": f32(),
             "work": torch.zeros(off, dtype=torch.bfloat16, device=dev), "xchg": None, "late": None, "pending": 0, "handle": None,
             "launched": False, "ready": False, "idx": len(self.buckets)}
        for p, o in zip(ordered, offs):
            n = p.numel()
            b["master"][o:o + n].view(p.shape).copy_(p.data)
            p.data = b["master"][o:o + n].view(p.shape)          # the parameter IS its slice of the flat master buffer from here on
            p.grad = None
            self.slot[p] = (b, o, n)
            p.__dict__["_emdr2_flat"] = self
        self.buckets.append(b)

    def _world(self):
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            return torch.distributed.get_world_size(self.group)
        return 1

    # ---- bf16 working copies (kernels.w_bf16) -----------------------------------------------------------------------------------------
    def _on_invalidate(self):
        self._epoch += 1                                          # someone changed masters behind our back (checkpoint load): re-cast lazily

    def work_view(self, p):
        b, o, n = self.slot[p]
        view = b["work"][o:o + n].view(p.shape)
        stamp = (p._version, self._epoch)
        if p.__dict__.get("_emdr2_work_stamp") != stamp:
            _native.check(_native.lib().emdr2_cast_f32_to_bf16(p.data_ptr(), view.data_ptr(), n, _native.stream_ptr()), "cast")
            p.__dict__["_emdr2_work_stamp"] = stamp
        return view

    def refresh_working_copies(self):
        if not self.on_device:
            return
        lib, sp = _native.lib(), _native.stream_ptr()
        for b in self.buckets:
            _native.check(lib.emdr2_cast_f32_to_bf16(b["master"].data_ptr(), b["work"].data_ptr(), b["n"], sp), "cast")
        self._stamp_all()

    def _stamp_all(self):
        for p in self.params:
            p.__dict__["_emdr2_work_stamp"] = (p._version, self._epoch)

    # ---- gradient sink (kernels.GRAD_SINK protocol) -------------------------------------------------------------------------------------
    def owns(self, p):
        return p in self.slot

    def zero_grad(self):
        self.begin_step()
        from emdr2_amd.model import kernels
        kernels.ATTN_STASH.store.clear()      # entries of a forward whose backward never ran must not outlive the step
        kernels.FANIN.clear()
        kernels.PREMASK.clear()

    def _ensure_exchange_buffers(self):
        """world > 1: the bf16 exchange copy of every bucket and the flags buffer of finish() exist BEFORE the first backward, so that a rank
        that runs out of memory in its very first step can still complete the step's collectives (abort_step) without allocating."""
        if self._world() == 1 or getattr(self, "_flags", None) is not None:
            return
        dev = self.buckets[0]["grad"].device
        if self.exchange_dtype == "bf16":
            for b in self.buckets:
                if b["xchg"] is None:
                    b["xchg"] = torch.empty(b["n"], dtype=torch.bfloat16, device=dev)
        self._flags = torch.zeros(1 + len(self.buckets), dtype=torch.int32, device=dev)
        self._flags_host = torch.zeros(1 + len(self.buckets), dtype=torch.int32)
        if self.on_device:
            self._flags_host = self._flags_host.pin_memory()

    def begin_step(self):
        self._stale = False
        self._ensure_exchange_buffers()
        for b in self.buckets:
            b["grad"].zero_()                 # ONE fill per bucket: the producing kernels accumulate straight into their slices (accumulation_target)
            b["handle"], b["launched"], b["ready"] = None, False, False
            b["late"] = None
            b["pending"] = sum(1 for p in b["params"] if self.expected and self.expected.get(p, 0) > 0)
            b["ready"] = bool(self.expected) and b["pending"] == 0      # (nothing expected: it leaves, carrying zeros, when its turn comes)
        for p in self.params:
            self.count[p] = 0
            p.grad = None
        self.next_bucket = 0

    def grad_view(self, p):
        b, o, n = self.slot[p]
        return b["grad"][o:o + n].view(p.shape)

    def accumulation_target(self, p):
        """The parameter's slice of its flat gradient bucket, for a kernel that ADDS its contribution there (weight-gradient GEMM slices,
        bias / LayerNorm column sums, embedding scatter-adds) -- zeroed with the bucket at the start of the step, so neither a per-parameter
        fill nor a copy is needed; None once the bucket has left for its all-reduce (the contribution then goes through `accumulate`).
        The caller reports the finished contribution with `contributed(p)`."""
        if self._stale:
            self.begin_step()                 # a backward after step() without zero_grad(): the buckets still hold the last step's sums
        b, o, n = self.slot[p]
        return None if b["launched"] else b["grad"][o:o + n].view(p.shape)

    def contributed(self, p):
        b, o, n = self.slot[p]
        if p.grad is None:
            p.grad = self.grad_view(p)
        self._count_contribution(p, b)

    def _count_contribution(self, p, b):
        self.count[p] += 1
        if self.expected is not None:
            exp = self.expected.get(p, 0)                          # (an unexpected or surplus contribution is simply added; finish() re-learns)
            if self.count[p] == exp:
                b["pending"] -= 1
                if b["pending"] == 0:
                    # complete: it leaves as soon as every bucket before it has left (reverse registration order is roughly the order in which
                    # gradients become final, so this rarely holds a bucket back -- and an aborting rank knows what its peers have issued)
                    b["ready"] = True
                    while self.next_bucket < len(self.buckets) and self.buckets[self.next_bucket]["ready"]:
                        self._launch(self.buckets[self.next_bucket])
                        self.launched_early += 1

    def accumulate(self, p, g):
        if self._stale:
            self.begin_step()
        b, o, n = self.slot[p]
        if b["launched"]:
            # the contribution pattern changed since it was learned: this bucket has left for its all-reduce.  Collect the late-comer in a
            # side buffer that finish() all-reduces and adds (mean of the early part + mean of the late part = mean of the whole)
            if b["late"] is None:
                b["late"] = torch.zeros(b["n"], dtype=torch.float32, device=b["grad"].device)
            b["late"][o:o + n].view(p.shape).add_(g.view(p.shape))
            self.count[p] += 1
            if p.grad is None:
                p.grad = self.grad_view(p)
            return
        v = self.grad_view(p)
        v.add_(g.view_as(v))                                       # (the slice holds zeros or the step's earlier contributions)
        p.grad = v
        self._count_contribution(p, b)

    def _launch(self, b):
        assert b["idx"] == self.next_bucket
        b["launched"] = True
        self.next_bucket += 1
        world = self._world()
        if world > 1 and self.exchange_dtype == "fp32":
            b["grad"].mul_(1.0 / world)                            # pre-divide, then sum (distributed.py:56-58)
            self.launches_last_step += 1
            b["handle"] = torch.distributed.all_reduce(b["grad"], group=self.group, async_op=True)
        elif world > 1:
            if b["xchg"] is None:                                  # (normally preallocated: _ensure_exchange_buffers)
                b["xchg"] = torch.empty(b["n"], dtype=torch.bfloat16, device=b["grad"].device)
            _native.check(_native.lib().emdr2_scale_cast_f32_to_bf16(b["grad"].data_ptr(), b["xchg"].data_ptr(), b["n"], 1.0 / world, _native.stream_ptr()),
                          "scale_cast")                            # pre-divide, then sum (distributed.py:56-58), 16 bits on the wire
            self.launches_last_step += 1
            b["handle"] = torch.distributed.all_reduce(b["xchg"], group=self.group, async_op=True)

    def finish(self, abort=False):
        """After loss.backward(): reduce what is left, wait for everything, widen the exchanged gradients back to fp32.
        world > 1: every bucket is all-reduced every step and the ranks first agree (one small MAX all-reduce) on whether the step stands and
        on which buckets carry late contributions; raises StepAborted on every rank if any rank called abort_step()."""
        world = self._world()
        if not abort and (self.expected is None or any(self.count[p] != self.expected.get(p, 0) for p in self.params)):
            if self.expected is not None:
                self.pattern_changes += 1                          # (every rank runs the same model code: they all see the change in the same step)
            self.expected = dict(self.count)
            idle = [p for p in self.params if self.count[p] == 0]
            for p in idle:                                         # no gradient this step: zero its slice (a previous step may have written it)
                self.grad_view(p).zero_()
            self.inactive = idle                                   # ... and leave it untouched by the update, like `grad is None` in apex / torch
        # what has not left yet, in index order.  world > 1: EVERY bucket is exchanged (a bucket without gradients carries zeros), so the
        # collectives of a step are the same on a rank that is aborting as on its peers
        for b in self.buckets[self.next_bucket:]:
            if world > 1 or any(self.count[p] for p in b["params"]):
                self._launch(b)
            else:
                self.next_bucket += 1
        late = [b["late"] is not None for b in self.buckets]
        aborted = bool(abort)
        flag_handle = None
        if world > 1:
            # [abort, late-buffer flag per bucket], MAX over the ranks: persistent buffers (nothing is allocated here: a rank may be out of
            # memory), the all-reduce in flight while the bucket results are waited for and widened; the host reads it afterwards
            self._ensure_exchange_buffers()
            self._flags_host.copy_(torch.tensor([int(aborted)] + [int(x) for x in late], dtype=torch.int32))
            self._flags.copy_(self._flags_host, non_blocking=True)
            flag_handle = torch.distributed.all_reduce(self._flags, op=torch.distributed.ReduceOp.MAX, group=self.group, async_op=True)
        for b in self.buckets:
            if b["handle"] is not None:
                b["handle"].wait()
                b["handle"] = None
                if self.exchange_dtype == "bf16":                  # (of an aborted step too: its buckets are discarded anyway)
                    _native.check(_native.lib().emdr2_widen_bf16_to_f32(b["xchg"].data_ptr(), b["grad"].data_ptr(), b["n"], _native.stream_ptr()), "widen")
                    self.launches_last_step += 1
        if flag_handle is not None:
            flag_handle.wait()
            flags = self._flags.tolist()                           # the step's one host read of the exchange (everything above is enqueued)
            aborted, late = bool(flags[0]), [bool(x) for x in flags[1:]]
        if aborted:                                                # every rank knows: the late buffers' all-reduces are skipped by all of them
            late = [False] * len(self.buckets)
            for b in self.buckets:
                b["late"] = None
        for b, has_late in zip(self.buckets, late):
            if has_late:                                           # contributions that missed their bucket's all-reduce (pattern change) on SOME rank
                if b["late"] is None:
                    b["late"] = torch.zeros(b["n"], dtype=torch.float32, device=b["grad"].device)
                if world > 1:
                    b["late"].mul_(1.0 / world)
                    torch.distributed.all_reduce(b["late"], group=self.group)
                b["grad"].add_(b["late"])
                b["late"] = None
        if aborted:
            self._stale = True                                     # the buckets hold a discarded step: the next contribution starts afresh
            raise StepAborted("a rank of the data-parallel group gave this step up; its gradients are discarded on all ranks")

    def abort_step(self):
        """This rank cannot complete the step's backward (allocation failure): complete the step's gradient collectives anyway so that the
        peers, which are waiting in theirs, and this rank stay in lockstep, and tell them to discard the step (their finish() raises
        StepAborted).  Single process: nothing to do."""
        if self._world() == 1:
            self._stale = True
            return
        try:
            self.finish(abort=True)
        except StepAborted:
            pass

    # ---- the update ---------------------------------------------------------------------------------------------------------------------
    def step(self, lr=None):
        if not self.on_device:
            raise _native.NativeError("the optimizer step runs on HIP kernels only (no host implementation)")
        lib, sp = _native.lib(), _native.stream_ptr()
        lr = self.lr if lr is None else lr
        if self.expected is None:
            self.finish()                                          # single-process use without an explicit finish()
        self.step_count += 1
        active = [b for b in self.buckets if any(self.count[p] for p in b["params"])]
        if not active:
            return 0.0
        self._gsq.zero_()
        # parameters without a gradient this step are untouched by the update -- master AND both moments -- like `grad is None` in apex / torch
        saved = [(p, p.data.clone()) + tuple(t.clone() for t in self._moments(p)) for p in getattr(self, "inactive", [])]
        for b in active:
            _native.check(lib.emdr2_sumsq_f32(b["grad"].data_ptr(), b["n"], self._gsq.data_ptr(), self._scratch.data_ptr(), sp), "sumsq")
        for b in active:
            _native.check(lib.emdr2_adam_step_flat(b["master"].data_ptr(), b["grad"].data_ptr(), b["m"].data_ptr(), b["v"].data_ptr(), b["work"].data_ptr(),
                                                   b["n"], b["split"], lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, self.step_count,
                                                   self._gsq.data_ptr(), self.clip_grad, sp), "adam_step_flat")
        for p, old, m_old, v_old in saved:
            p.data.copy_(old)
            m, v = self._moments(p)
            m.copy_(m_old); v.copy_(v_old)
        self.launches_last_step += 1 + 2 * len(active) + 6 * len(saved)
        self.optimizer_launches = self.launches_last_step
        self.launches_last_step = 0
        self._stale = True                    # the gradient buckets are spent: the next contribution starts a new step if zero_grad() does not
        from emdr2_amd.model import kernels
        kernels.WEIGHTS.invalidate()          # drops the derived forms (transposed / row-permuted copies) of every weight ...
        self._stamp_all()                     # ... while the bf16 working copies were just written by the Adam kernel itself
        for p, *_ in saved:                   # restored masters: their working copies were overwritten by the kernel
            p.__dict__["_emdr2_work_stamp"] = None
        kernels.DROPOUT.step = self.step_count
        return self._gsq

    # ---- checkpoint format of FusedAdam (moments in parameter order) ---------------------------------------------------------------------
    def _ordered(self):
        return self.params[::-1]

    def _moments(self, p):
        b, o, n = self.slot[p]
        return b["m"][o:o + n].view(p.shape), b["v"][o:o + n].view(p.shape)

    def state_dict(self):
        return {'step': self.step_count,
                'state': {i: {'exp_avg': self._moments(p)[0], 'exp_avg_sq': self._moments(p)[1]} for i, p in enumerate(self._ordered())}}

    def load_state_dict(self, sd):
        self.step_count = sd['step']
        ordered = self._ordered()
        for i, st in sd['state'].items():
            m, v = self._moments(ordered[int(i)])
            m.copy_(st['exp_avg']); v.copy_(st['exp_avg_sq'])
        from emdr2_amd.model import kernels
        kernels.DROPOUT.step = self.step_count


class RetentionGuard(object):
    """Runs a training step under the activation-retention plan of an EMDR2Model and keeps it running when the plan stops fitting.

    The plan -- `keep` reader-encoder layers whole, `reader` / `context` / `query` encoder layers with selective retention
    (EMDR2Model.set_recompute_keep_last / set_selective_retention; everything else is re-run per layer like the reference's
    --checkpoint-activations) -- is sized for the first steps; packed token counts, and with them every retained tensor, still grow a little
    over the first tens of steps.  A step that runs out of HBM is given up ON ALL RANKS TOGETHER: the rank that failed completes the step's
    gradient collectives (FlatAdam.abort_step), its peers learn of it in finish() (StepAborted), everybody frees its cached blocks and runs
    the same step again; a second failure thins the plan (context tower first, then kept layers, then the reader's selective layers) by
    the same rule on every rank.  `forward_progress()` is a counter that moves once the forward's own all-gathers (the retriever's) are
    behind this rank -- before that point of an attempt its peers cannot be told and the failure is raised.  Used by bench_e2e.py and by
    the training task."""

    def __init__(self, model, optimizer, keep=0, reader=0, context=0, query=0, forward_progress=None, log=None, micro=1, batch=None,
                 on_micro_change=None):
        self.model, self.opt = model, optimizer
        self.plan = {"keep": int(keep), "reader": int(reader), "context": int(context), "query": int(query), "thinned": 0}
        # question micro-batches of a step (EMDR2Model.forward_backward; the step function reads `guard.micro`): with m > 1 nothing is
        # re-run and there is no retention plan to thin -- a step that still does not fit is split finer (m doubles, up to one question per
        # group: forward_backward takes group counts that do not divide the batch).  `on_micro_change(m)`: how a step function that reads the
        # group count from somewhere else (the task's --question-micro-batches flag) is told
        self.micro, self.batch, self.on_micro_change = max(1, int(micro)), batch, on_micro_change
        self.reruns = 0
        self.forward_progress = forward_progress
        self.log = log or (lambda msg: None)
        self.apply()

    def apply(self):
        self.model.set_recompute_keep_last(self.plan["keep"])
        self.model.set_selective_retention(self.plan["reader"], self.plan["context"], self.plan["query"])

    def set(self, keep, reader, context, query=None):
        self.plan.update(keep=int(keep), reader=int(reader), context=int(context))
        if query is not None:
            self.plan["query"] = int(query)
        self.apply()

    def thin(self):
        p = self.plan
        if p["context"] > 0:
            p["context"] = max(0, p["context"] - 2)
            if p["context"] == 0:
                p["query"] = 0
        elif p["keep"] > 0:
            p["keep"] -= 1
        elif p["reader"] > 0:
            p["reader"] = max(0, p["reader"] - 2)
        elif self.micro > 1 and self.batch and self.micro < self.batch:
            self.micro = min(2 * self.micro, int(self.batch))
            if self.on_micro_change is not None:
                self.on_micro_change(self.micro)
            p["thinned"] += 1
            self.log("step split into %d question micro-batches after an allocation failure" % self.micro)
            return True
        else:
            return False
        p["thinned"] += 1
        self.apply()
        self.log("retention plan thinned after an allocation failure: %s" % {k: v for k, v in p.items() if k != "thinned"})
        return True

    def run(self, step_once):
        import gc
        world = torch.distributed.get_world_size() if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 1
        for attempt in range(8):
            mark = self.forward_progress() if self.forward_progress else None
            failed = False
            try:
                return step_once()
            except torch.cuda.OutOfMemoryError:
                if world > 1 and self.forward_progress and self.forward_progress() == mark:
                    raise
                if world > 1 and getattr(self.opt, "abort_step", None) is None:
                    raise                                          # (an optimizer without the abort protocol cannot take its peers along)
                failed = True
            except StepAborted:
                pass
            if failed:
                # outside the except block: the failed step's activations were held by the traceback; drop them BEFORE the exchange is
                # completed (abort_step itself allocates nothing: FlatAdam._ensure_exchange_buffers)
                gc.collect()
                torch.cuda.empty_cache()
                abort = getattr(self.opt, "abort_step", None)
                if abort is not None:
                    abort()
            self.opt.zero_grad()
            gc.collect()
            torch.cuda.empty_cache()                               # first: allocator fragmentation -- give the blocks back, same step again
            self.reruns += 1
            if attempt >= 1 and not self.thin():
                raise torch.cuda.OutOfMemoryError("the step does not fit with the reference's full per-layer recompute either")
        return step_once()
