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
        for p in params:
            _native.check(lib.emdr2_sumsq_f32(p.grad.data_ptr(), p.grad.numel(), gsq.data_ptr(), sp), "sumsq")
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
