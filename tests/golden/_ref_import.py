"""Import shims that let the read-only reference at /root/reference be imported and run on CPU in
the build container, for GENERATING golden fixtures only.

Never imported by tests, bench.py, smoke() or the product: /root/reference does not exist on the
GPU box.  Nothing from the reference is copied; this file only installs stand-in *modules* for
third-party packages that are not installed (apex, amp_C) and patches device calls to CPU.
"""
import contextlib
import sys
import types

import numpy as np
import torch

REFERENCE_ROOT = "/root/reference"


def install_import_shims():
    sys.dont_write_bytecode = True
    # the repo root carries alias packages named `megatron` and `tasks` (drop-in names over emdr2_amd): the REFERENCE's packages must win here
    if sys.path[:1] != [REFERENCE_ROOT]:
        if REFERENCE_ROOT in sys.path:
            sys.path.remove(REFERENCE_ROOT)
        sys.path.insert(0, REFERENCE_ROOT)
    for name in [n for n in sys.modules if n.split(".")[0] in ("megatron", "tasks")]:
        if not str(getattr(sys.modules[name], "__file__", "") or "").startswith(REFERENCE_ROOT):
            del sys.modules[name]
    if "torch._six" not in sys.modules:
        six = types.ModuleType("torch._six")
        six.inf = float("inf")
        sys.modules["torch._six"] = six
    if not hasattr(np, "float"):
        np.float = float
    for name in ("apex", "apex.optimizers", "apex.multi_tensor_apply", "amp_C"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["apex.optimizers"].FusedAdam = torch.optim.AdamW
    sys.modules["apex.multi_tensor_apply"].multi_tensor_applier = None
    sys.modules["apex"].optimizers = sys.modules["apex.optimizers"]
    sys.modules["apex"].multi_tensor_apply = sys.modules["apex.multi_tensor_apply"]


@contextlib.contextmanager
def cuda_calls_on_cpu(device_count=1):
    """Route the reference's hard-coded CUDA placement to CPU tensors while it runs."""
    orig_to, orig_cuda, orig_zeros = torch.Tensor.to, torch.Tensor.cuda, torch.zeros
    orig_count, orig_cur = torch.cuda.device_count, torch.cuda.current_device

    def _is_cuda(d):
        return isinstance(d, str) and d.startswith("cuda") or (isinstance(d, torch.device) and d.type == "cuda")

    def to(self, *args, **kw):
        args = tuple(a for a in args if not _is_cuda(a))
        if _is_cuda(kw.get("device")):
            kw.pop("device")
        return orig_to(self, *args, **kw) if (args or kw) else self

    def zeros(*args, **kw):
        if _is_cuda(kw.get("device")):
            kw.pop("device")
        return orig_zeros(*args, **kw)

    torch.Tensor.to, torch.Tensor.cuda, torch.zeros = to, (lambda self, *a, **k: self), zeros
    torch.cuda.device_count = lambda: device_count
    torch.cuda.current_device = lambda: "cpu"
    try:
        yield
    finally:
        torch.Tensor.to, torch.Tensor.cuda, torch.zeros = orig_to, orig_cuda, orig_zeros
        torch.cuda.device_count, torch.cuda.current_device = orig_count, orig_cur
