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
    from emdr2_amd import dist_util
    _, _, dev = dist_util.init_distributed(rank=args.rank, world=args.world_size, local_rank=args.local_rank)     # one process per GPU, RCCL
    args.local_rank = dev.index
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
