"""Optimizer step of the hot path (reference: megatron/training.py:73-230, megatron/fp16/fp16.py, apex FusedAdam, mpu/grads.py,
megatron/model/distributed.py, megatron/learning_rates.py), bf16 edition: parameters ARE the fp32 masters, gradients arrive in fp32
from the weight-gradient GEMMs, so there is no loss scaling, no overflow check and no master-copy round trip."""
import torch

from emdr2_amd import _native


def get_params_for_weight_decay_optimization(module):
    """No weight decay on LayerNorm parameters and biases (megatron/model/utils.py:64-83)."""
    decay, no_decay = [], []
    for name, p in module.named_parameters():
        (no_decay if (name.endswith(".bias") or "layernorm" in name) else decay).append(p)
    return [{"params": decay}, {"params": no_decay, "weight_decay": 0.0}]


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


class FusedAdam(object):
    """Adam with decoupled weight decay on fp32 masters (apex FusedAdam(adam_w_mode=True) defaults betas (0.9, 0.999), eps 1e-8;
    SURVEY.md 8c: apex is unpinned in the reference, this is the documented choice), global-norm clipping folded into the update."""

    def __init__(self, param_groups, lr=2e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.1, clip_grad=1.0):
        self.groups = [dict(g) for g in param_groups]
        for g in self.groups:
            g.setdefault("weight_decay", weight_decay)
        self.lr, self.betas, self.eps, self.clip_grad, self.step_count = lr, betas, eps, clip_grad, 0
        self.state = {}

    def zero_grad(self):
        for g in self.groups:
            for p in g["params"]:
                p.grad = None
        from emdr2_amd.model import kernels
        kernels.ATTN_STASH.store.clear()      # entries of a forward whose backward never ran must not outlive the step

    def state_dict(self):
        """Moments in parameter order (torch-optimizer style: {'step', 'state': {index: {'exp_avg', 'exp_avg_sq'}}})."""
        params = [p for g in self.groups for p in g["params"]]
        return {'step': self.step_count,
                'state': {i: {'exp_avg': self.state[p][0], 'exp_avg_sq': self.state[p][1]} for i, p in enumerate(params) if p in self.state}}

    def load_state_dict(self, sd):
        params = [p for g in self.groups for p in g["params"]]
        self.step_count = sd['step']
        for i, st in sd['state'].items():
            p = params[int(i)]
            self.state[p] = (st['exp_avg'].to(p.device, torch.float32).clone(), st['exp_avg_sq'].to(p.device, torch.float32).clone())
        from emdr2_amd.model import kernels
        kernels.DROPOUT.step = self.step_count

    def step(self, lr=None):
        lib = _native.lib()
        sp = _native.stream_ptr()
        lr = self.lr if lr is None else lr
        self.step_count += 1
        params = [p for g in self.groups for p in g["params"] if p.grad is not None]
        if not params:
            return 0.0
        gsq = torch.zeros(1, dtype=torch.float32, device=params[0].device)
        if getattr(self, "_scratch", None) is None:
            self._scratch = torch.zeros(1025, dtype=torch.float32, device=params[0].device)
        for p in params:
            _native.check(lib.emdr2_sumsq_f32(p.grad.data_ptr(), p.grad.numel(), gsq.data_ptr(), self._scratch.data_ptr(), sp), "sumsq")
        for g in self.groups:
            for p in g["params"]:
                if p.grad is None:
                    continue
                st = self.state.get(p)
                if st is None:
                    st = self.state[p] = (torch.zeros_like(p), torch.zeros_like(p))
                _native.check(lib.emdr2_adam_step(p.data_ptr(), p.grad.data_ptr(), st[0].data_ptr(), st[1].data_ptr(), None, p.numel(), lr,
                                                  self.betas[0], self.betas[1], self.eps, g["weight_decay"], self.step_count, gsq.data_ptr(),
                                                  self.clip_grad, sp), "adam_step")
        from emdr2_amd.model import kernels
        kernels.WEIGHTS.invalidate()      # masters were written through raw pointers: bf16 working copies are rebuilt lazily
        kernels.DROPOUT.step = self.step_count   # next iteration draws fresh dropout masks
        return gsq


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


class GradientBuckets(object):
    """Data-parallel gradient averaging, bucketed and overlapped with the backward (the reference does one blocking flattened
    all-reduce after the backward, megatron/model/distributed.py:35-62).

    Parameters are laid out, in reverse registration order (roughly the order their gradients become final), into flat fp32 buckets;
    `p.grad` is a view into its bucket, so nothing is copied before or after the collective.  The autograd functions hand every gradient
    to `accumulate`; once a bucket has received the last expected contribution of each of its parameters it is pre-divided by the world
    size and all-reduced asynchronously (RCCL over xGMI on the GPU box: the collective runs on the communicator's stream while the
    backward keeps computing), `finish()` waits for the handles.  How many contributions a parameter receives per step (1, or more for
    the tied embedding / LM-head weights) is learned in the first step, during which all buckets are reduced in `finish()`.
    Result = `allreduce_gradients` (tests/test_dist_allreduce.py, gloo, world size 2)."""

    def __init__(self, params, group=None, bucket_bytes=128 << 20):
        self.group = group
        self.params = [p for p in params if p.requires_grad][::-1]
        self.buckets, self.bucket_of, self.view = [], {}, {}
        cur, cur_n = [], 0
        for p in self.params:
            if cur and (cur_n + p.numel()) * 4 > bucket_bytes:
                self._close(cur)
                cur, cur_n = [], 0
            cur.append(p); cur_n += p.numel()
        if cur:
            self._close(cur)
        self.expected = None                  # {param: contributions per step}, learned in the first step
        self.count = {p: 0 for p in self.params}
        self.launched_early = 0
        self.begin_step()

    def _close(self, plist):
        dev = plist[0].device
        flat = torch.zeros(sum(p.numel() for p in plist), dtype=torch.float32, device=dev)
        b = {"params": plist, "flat": flat, "pending": 0, "handle": None, "launched": False}
        off = 0
        for p in plist:
            self.view[p] = flat[off:off + p.numel()].view(p.shape)
            self.bucket_of[p] = b
            off += p.numel()
        self.buckets.append(b)

    def _world(self):
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            return torch.distributed.get_world_size(self.group)
        return 1

    def owns(self, p):
        return p in self.view

    def begin_step(self):
        """After optimizer.zero_grad(): clear the buckets and the per-step bookkeeping."""
        for b in self.buckets:
            b["flat"].zero_()
            b["handle"], b["launched"] = None, False
            b["pending"] = sum(1 for p in b["params"] if self.expected and self.expected.get(p, 0) > 0)
        for p in self.params:
            self.count[p] = 0
            p.grad = None

    def accumulate(self, p, g):
        b = self.bucket_of[p]
        if b["launched"]:
            raise RuntimeError("gradient for a parameter whose bucket was already reduced (the contribution pattern changed between steps)")
        v = self.view[p]
        if self.count[p] == 0:
            v.copy_(g.view_as(v))
            p.grad = v
        else:
            v.add_(g.view_as(v))
        self.count[p] += 1
        if self.expected is not None:
            exp = self.expected.get(p, 0)
            if exp == 0 or self.count[p] > exp:
                raise RuntimeError("unexpected gradient contribution (the contribution pattern changed between steps)")
            if self.count[p] == exp:
                b["pending"] -= 1
                if b["pending"] == 0:
                    self._launch(b)
                    self.launched_early += 1

    def _launch(self, b):
        b["launched"] = True
        world = self._world()
        if world > 1:
            b["flat"].div_(world)                                          # pre-divide, then sum (distributed.py:56-58)
            b["handle"] = torch.distributed.all_reduce(b["flat"], group=self.group, async_op=True)

    def finish(self):
        """After loss.backward(): reduce what is left, wait for everything."""
        if self.expected is None:
            self.expected = dict(self.count)
        elif any(self.count[p] != self.expected.get(p, 0) for p in self.params):
            raise RuntimeError("gradient contributions differ from the first step")
        for b in self.buckets:
            if not b["launched"] and any(self.count[p] for p in b["params"]):
                self._launch(b)
        for b in self.buckets:
            if b["handle"] is not None:
                b["handle"].wait()
                b["handle"] = None


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

    Parameters that never receive a gradient (the reader's unused token-type table) are left untouched, like apex / torch skip `grad is None`."""

    def __init__(self, module, lr=2e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.1, clip_grad=1.0, group=None, bucket_bytes=512 << 20):
        from emdr2_amd.model import kernels
        self.lr, self.betas, self.eps, self.weight_decay, self.clip_grad, self.group = lr, betas, eps, weight_decay, clip_grad, group
        self.step_count = 0
        no_decay = set(id(p) for n, p in module.named_parameters() if (n.endswith(".bias") or "layernorm" in n))      # model/utils.py:64-83
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
        self._gsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self._scratch = torch.zeros(1025, dtype=torch.float32, device=dev)
        self.expected = None                  # {param: contributions per step}, learned in the first step
        self.count = {p: 0 for p in self.params}
        self.launched_early = 0
        self.launches_last_step = 0
        self._epoch = 0
        import weakref
        kernels.WEIGHTS.listeners.append(weakref.WeakMethod(self._on_invalidate))
        self.refresh_working_copies()
        self.begin_step()

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
             "work": torch.zeros(off, dtype=torch.bfloat16, device=dev), "xchg": None, "pending": 0, "handle": None, "launched": False}
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

    def begin_step(self):
        for b in self.buckets:
            b["handle"], b["launched"] = None, False
            b["pending"] = sum(1 for p in b["params"] if self.expected and self.expected.get(p, 0) > 0)
        for p in self.params:
            self.count[p] = 0
            p.grad = None

    def grad_view(self, p):
        b, o, n = self.slot[p]
        return b["grad"][o:o + n].view(p.shape)

    def accumulate(self, p, g):
        b = self.slot[p][0]
        if b["launched"]:
            raise RuntimeError("gradient for a parameter whose bucket was already reduced (the contribution pattern changed between steps)")
        v = self.grad_view(p)
        if self.count[p] == 0:
            if g.data_ptr() != v.data_ptr():
                v.copy_(g.view_as(v))
            p.grad = v
        else:
            v.add_(g.view_as(v))
        self.count[p] += 1
        if self.expected is not None:
            exp = self.expected.get(p, 0)
            if exp == 0 or self.count[p] > exp:
                raise RuntimeError("unexpected gradient contribution (the contribution pattern changed between steps)")
            if self.count[p] == exp:
                b["pending"] -= 1
                if b["pending"] == 0:
                    self._launch(b)
                    self.launched_early += 1

    def _launch(self, b):
        b["launched"] = True
        world = self._world()
        if world > 1:
            if b["xchg"] is None:
                b["xchg"] = torch.empty(b["n"], dtype=torch.bfloat16, device=b["grad"].device)
            _native.check(_native.lib().emdr2_scale_cast_f32_to_bf16(b["grad"].data_ptr(), b["xchg"].data_ptr(), b["n"], 1.0 / world, _native.stream_ptr()),
                          "scale_cast")                            # pre-divide, then sum (distributed.py:56-58), 16 bits on the wire
            self.launches_last_step += 1
            b["handle"] = torch.distributed.all_reduce(b["xchg"], group=self.group, async_op=True)

    def finish(self):
        """After loss.backward(): reduce what is left, wait for everything, widen the exchanged gradients back to fp32."""
        first = self.expected is None
        if first:
            self.expected = dict(self.count)
            never = [p for p in self.params if self.count[p] == 0]
            for p in never:                                        # stays zero forever: nobody writes it
                self.grad_view(p).zero_()
            self.inactive = never
        elif any(self.count[p] != self.expected.get(p, 0) for p in self.params):
            raise RuntimeError("gradient contributions differ from the first step")
        for b in self.buckets:
            if not b["launched"] and any(self.count[p] for p in b["params"]):
                self._launch(b)
        for b in self.buckets:
            if b["handle"] is not None:
                b["handle"].wait()
                b["handle"] = None
                _native.check(_native.lib().emdr2_widen_bf16_to_f32(b["xchg"].data_ptr(), b["grad"].data_ptr(), b["n"], _native.stream_ptr()), "widen")
                self.launches_last_step += 1

    # ---- the update ---------------------------------------------------------------------------------------------------------------------
    def step(self, lr=None):
        lib, sp = _native.lib(), _native.stream_ptr()
        lr = self.lr if lr is None else lr
        if self.expected is None:
            self.finish()                                          # single-process use without an explicit finish()
        self.step_count += 1
        active = [b for b in self.buckets if any(self.count[p] for p in b["params"])]
        if not active:
            return 0.0
        self._gsq.zero_()
        saved = [(p, p.data.clone()) for p in getattr(self, "inactive", [])]      # untouched by the update, like `grad is None` in apex / torch
        for b in active:
            _native.check(lib.emdr2_sumsq_f32(b["grad"].data_ptr(), b["n"], self._gsq.data_ptr(), self._scratch.data_ptr(), sp), "sumsq")
        for b in active:
            _native.check(lib.emdr2_adam_step_flat(b["master"].data_ptr(), b["grad"].data_ptr(), b["m"].data_ptr(), b["v"].data_ptr(), b["work"].data_ptr(),
                                                   b["n"], b["split"], lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, self.step_count,
                                                   self._gsq.data_ptr(), self.clip_grad, sp), "adam_step_flat")
        for p, old in saved:
            p.data.copy_(old)
        self.launches_last_step += 1 + 2 * len(active) + 2 * len(saved)
        self.optimizer_launches = self.launches_last_step
        self.launches_last_step = 0
        from emdr2_amd.model import kernels
        kernels.WEIGHTS.invalidate()          # drops the derived forms (transposed / row-permuted copies) of every weight ...
        self._stamp_all()                     # ... while the bf16 working copies were just written by the Adam kernel itself
        for p, old in saved:                  # restored masters: their working copies were overwritten by the kernel
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
