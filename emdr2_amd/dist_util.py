"""Process-group bring-up shared by every multi-rank entry point (bench.py, bench_e2e.py, tasks/run.py, tools/rccl_probe.py): one process per
GPU, `torch.distributed` over RCCL ("nccl" on ROCm) -- the reference's initialize.py:_initialize_distributed without its model-parallel
groups (tensor / pipeline parallelism is asserted 1 there, dualencoder_model.py:15).

The backend is chosen in ONE place so that the dry runs this repository can do on a single GPU (EMDR2_DIST_BACKEND=gloo with
EMDR2_SINGLE_DEVICE=1: all ranks share cuda:0) execute the same lines as an RCCL run; the two differ by the backend string and by the
`device_id` hint that lets RCCL bind its communicator eagerly."""
import datetime
import os

import torch


def backend_name():
    return os.environ.get("EMDR2_DIST_BACKEND", "nccl")


def local_device(local_rank=None):
    """cuda:<LOCAL_RANK>, or cuda:0 for every rank under EMDR2_SINGLE_DEVICE (N-rank dry run on a 1-GPU box)."""
    if local_rank is None:
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("EMDR2_SINGLE_DEVICE"):
        local_rank = 0
    return torch.device("cuda", local_rank)


def init_distributed(timeout_s=1800.0, rank=None, world=None, local_rank=None):
    """(rank, world, device).  Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the launcher's environment (or takes them from the
    caller's parsed flags), selects the device, and -- for world > 1 -- creates the default process group with a COLLECTIVE timeout: a rank
    whose peers have gone away fails inside the collective after `timeout_s` instead of waiting forever (no per-rank kill timers needed)."""
    rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
    dev = local_device(local_rank)
    torch.cuda.set_device(dev)
    if world > 1 and not torch.distributed.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "6000")
        backend = backend_name()
        extra = {"device_id": dev} if backend == "nccl" else {}
        torch.distributed.init_process_group(backend=backend, world_size=world, rank=rank, timeout=datetime.timedelta(seconds=timeout_s), **extra)
    return rank, world, dev


def shutdown():
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
