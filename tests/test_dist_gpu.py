"""GPU, world size 2 on ONE device (gloo transport, both ranks on cuda:0): the multi-rank search path with the real HIP kernels --
per-rank shard scan, all-gather of the per-shard top-k, HIP merge kernel -- must equal the single-shard oracle on every rank, for the
canonical fp16 search and for the FaissMIPSIndex fp32-score search.  (RCCL itself is exercised only by the driver's multi-GPU runs.)"""
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import numpy as np
    import torch
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    from emdr2_amd.data.emdr2_index import DistributedBruteForceIndex, FaissMIPSIndex
    from oracle import mips_oracle as mo
    rng = np.random.default_rng(0)
    n, d, nq, k = 30011, 128, 37, 100
    rows = rng.standard_normal((n, d)).astype(np.float16)
    rows[rng.integers(0, n, size=500)] = rows[rng.integers(0, n, size=500)]        # cross-shard duplicates: ties broken by global row
    q = rng.standard_normal((nq, d)).astype(np.float16)
    ids = (rng.permutation(n) + 1).astype(np.int64)
    f = FaissMIPSIndex(d, None, use_gpu=True)
    f.add_with_ids(rows, ids)
    lo, hi = f.local_rows()
    assert (lo, hi) == ((0, 15006) if rank == 0 else (15006, 30011))
    D, I = f.search_mips_index(torch.from_numpy(q), k, reconstruct=False)
    od, oi = mo.topk_f32(rows, q, k, ids=ids)
    assert np.array_equal(D.view(np.uint32), np.ascontiguousarray(od).view(np.uint32)) and np.array_equal(I, oi)
    b = DistributedBruteForceIndex(d, None, use_gpu=True)
    b.add_arrays(ids.astype(np.int32), rows)
    dist, idx = b.search_mips_index(torch.from_numpy(q).cuda(), 50)
    od2, oi2 = mo.topk(rows, q, 50, ids=ids.astype(np.int32))
    assert np.array_equal(dist.cpu().numpy().view(np.uint16), od2.view(np.uint16)) and np.array_equal(idx.cpu().numpy(), oi2)
    torch.distributed.destroy_process_group()


def test_two_rank_sharded_search_equals_single_shard_oracle():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port), nprocs=2, join=True)
