"""Process-group bring-up shared by every multi-rank entry point (bench.py, bench_e2e.py, tasks/run.py, tools/rccl_probe.py): one process per
GPU, `torch.distributed` over RCCL ("nccl" on ROCm) -- the reference's initialize.py:_initialize_distributed without its model-parallel
groups (tensor / pipeline parallelism is asserted 1 there, dualencoder_model.py:15).

The backend is chosen in ONE place so that the dry runs this repository can do on a single GPU (EMDR2_DIST_BACKEND=gloo with
EMDR2_SINGLE_DEVICE=1: all ranks share cuda:0) execute the same lines as an RCCL run; the two differ by the backend string and by the
`device_id` hint that lets RCCL bind its communicator eagerly."""
import datetime
import os
import socket
import subprocess
import sys

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


def self_launch(gpus, argv=None):
    """`python bench.py --gpus N` as typed, without a launcher (the reference's one-line launch is `python -m torch.distributed.launch
    --nproc_per_node N tasks/run.py ...`, examples/openqa/emdr2_nq.sh:35,106): when no launcher has set WORLD_SIZE (or it says 1) and more
    than one GPU is asked for, re-run the same command line as N ranks under `torch.distributed.run` -- one process per LOCAL_RANK device,
    rendezvous on 127.0.0.1 and a free port -- and exit with ITS status: non-zero as soon as any rank fails (the launcher then stops the
    others) or a collective times out.  Only rank 0 prints, so stdout still carries the ONE JSON line.  Returns when nothing is to do (one
    GPU, or already inside a launcher's rank)."""
    if gpus <= 1 or int(os.environ.get("WORLD_SIZE", "1")) > 1:
        return
    if not os.environ.get("EMDR2_SINGLE_DEVICE") and torch.cuda.is_available() and torch.cuda.device_count() < gpus:
        raise SystemExit("--gpus %d but only %d device(s) visible (EMDR2_SINGLE_DEVICE=1 EMDR2_DIST_BACKEND=gloo runs all ranks on cuda:0 as a dry run)"
                         % (gpus, torch.cuda.device_count()))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")              # the host driver does dmabuf IPC only (RCCL's intra-node transport)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or gpus) // gpus)))
    argv = list(sys.argv if argv is None else argv)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + argv
    raise SystemExit(subprocess.call(cmd, env=env))


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
