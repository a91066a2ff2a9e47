"""TEST INFRASTRUCTURE: a per-parameter Adam on the `emdr2_adam_step` kernel (one launch per tensor), kept as an independent check of
`emdr2_amd.training.FlatAdam` (the product optimizer: flat buckets, one launch per bucket) and itself checked against `torch.optim.AdamW` +
`clip_grad_norm_` (tests/test_model_gpu.py).  It was the round-1 product optimizer (`training.FusedAdam`); nothing under emdr2_amd/ uses it."""
import torch

from emdr2_amd import _native


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
