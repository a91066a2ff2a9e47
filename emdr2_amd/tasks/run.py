"""Entry point: `python -m emdr2_amd.tasks.run --task OPENQA ...` with the flags of examples/openqa/emdr2_*.sh
(reference: tasks/run.py:24-89, megatron/initialize.py:30-100).  One process per GPU: launch with
`python -m torch.distributed.run --nproc-per-node N -m emdr2_amd.tasks.run ...` (rendezvous on 127.0.0.1)."""
import os
import sys

import torch

from emdr2_amd import arguments
from emdr2_amd.global_vars import set_args


def initialize(argv=None):
    args = arguments.parse_args(argv)
    set_args(args)
    if not torch.cuda.is_available():
        from emdr2_amd import _native
        raise _native.NativeError("the QA task needs a GPU; there is no CPU fallback")
    if os.environ.get("EMDR2_SINGLE_DEVICE"):                          # dry run of the N-rank path on a 1-GPU box (all ranks on cuda:0, gloo)
        args.local_rank = 0
    torch.cuda.set_device(args.local_rank)
    if args.world_size > 1 and not torch.distributed.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "6000")
        backend = os.environ.get("EMDR2_DIST_BACKEND", "nccl")         # "nccl" is RCCL on ROCm
        extra = {"device_id": torch.device("cuda", args.local_rank)} if backend == "nccl" else {}
        torch.distributed.init_process_group(backend=backend, world_size=args.world_size, rank=args.rank, **extra)
    from emdr2_amd.model import kernels
    kernels.DROPOUT.base_seed = args.seed                                # megatron/initialize.py:_set_random_seed: same seed on all DP ranks
    torch.manual_seed(args.seed)
    return args


def main(argv=None):
    args = initialize(argv)
    if args.task in ('OPENQA',):
        from emdr2_amd.tasks.openqa.e2eqa.run import main as task_main
    else:
        raise NotImplementedError('Task {} is not implemented.'.format(args.task))
    return task_main()


if __name__ == '__main__':
    main(sys.argv[1:])
