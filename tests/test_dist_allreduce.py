"""CPU, gloo, world size 2: the data-parallel gradient all-reduce (LocalDDP.allreduce_params, megatron/model/distributed.py:35-62):
pre-divided by the world size, every parameter averaged, ranks end up identical."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from emdr2_amd.training import allreduce_gradients
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.LayerNorm(16), torch.nn.Linear(16, 4))
    g = torch.Generator().manual_seed(100 + rank)
    for p in m.parameters():
        p.grad = torch.randn(p.shape, generator=g)
    m[2].bias.grad = None                                   # a parameter without gradient on this step is skipped consistently
    allreduce_gradients(m)
    torch.save([None if p.grad is None else p.grad.clone() for p in m.parameters()], os.path.join(out_dir, "g%d.pt" % rank))
    dist.destroy_process_group()


def test_gradients_are_averaged_across_ranks(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g0, g1 = torch.load(os.path.join(str(tmp_path), "g0.pt")), torch.load(os.path.join(str(tmp_path), "g1.pt"))
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.LayerNorm(16), torch.nn.Linear(16, 4))
    gens = [torch.Generator().manual_seed(100), torch.Generator().manual_seed(101)]
    for a, b, p in zip(g0, g1, m.parameters()):
        r0, r1 = torch.randn(p.shape, generator=gens[0]), torch.randn(p.shape, generator=gens[1])
        if a is None:
            assert b is None
            continue
        assert torch.allclose(a, b) and torch.allclose(a, (r0 + r1) / 2, atol=1e-6)
