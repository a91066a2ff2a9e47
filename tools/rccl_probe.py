"""First contact with the RCCL transport on whatever GPUs the box has (GPU): the exact calls the N-rank paths make -- process group through
emdr2_amd.dist_util (device_id hint), an async bf16 all-reduce of a gradient-bucket-sized buffer, `all_gather_into_tensor` of a packed
top-k block, barrier.   usage: python tools/rccl_probe.py            (1 rank)
                               python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/rccl_probe.py"""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emdr2_amd import dist_util  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
rank, world, dev = dist_util.init_distributed(timeout_s=120.0)
if world == 1:                                             # a 1-rank group still goes through RCCL's communicator set-up
    dist.init_process_group(dist_util.backend_name(), rank=0, world_size=1, device_id=dev)
x = torch.ones(256 << 20, dtype=torch.bfloat16, device=dev)                # one 512 MB gradient bucket in bf16
torch.cuda.synchronize(); t0 = time.perf_counter()
h = dist.all_reduce(x, async_op=True); h.wait(); torch.cuda.synchronize()
t_ar = time.perf_counter() - t0
y = torch.arange(3 * 512 * 50, dtype=torch.int64, device=dev) + rank       # the packed (score, id, row) block of one 512-query search
out = torch.empty(world * y.numel(), dtype=torch.int64, device=dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
dist.all_gather_into_tensor(out, y); torch.cuda.synchronize()
t_ag = time.perf_counter() - t0
# r05: the sharded search's exchange -- every rank's 16-byte records written into ITS slice of the gather buffer, gathered IN PLACE (the slice
# is the send buffer: ncclAllGather's in-place form, sendbuff == recvbuff + rank * count), then merged by the HIP kernel
nq, k = 512, 50
gathered = torch.zeros((world, nq, k, 16), dtype=torch.uint8, device=dev)
gathered[rank] = (torch.arange(nq * k * 16, device=dev) % 251 + rank).to(torch.uint8).view(nq, k, 16)
expect = torch.stack([(torch.arange(nq * k * 16, device=dev) % 251 + r).to(torch.uint8).view(nq, k, 16) for r in range(world)])
torch.cuda.synchronize(); t0 = time.perf_counter()
dist.all_gather_into_tensor(gathered.view(-1, k, 16), gathered[rank]); torch.cuda.synchronize()
t_inplace = time.perf_counter() - t0
inplace_ok = bool(torch.equal(gathered, expect))
dist.barrier(); torch.cuda.synchronize()
if rank == 0:
    print(json.dumps({"backend": dist_util.backend_name(), "world": world, "rccl_version": list(torch.cuda.nccl.version()),
                      "allreduce_512MB_bf16_ms": t_ar * 1e3, "allreduce_ok": bool(float(x[0]) == world), "allgather_topk_block_ms": t_ag * 1e3,
                      "allgather_ok": bool(int(out[-1]) == y.numel() - 1 + world - 1),
                      "allgather_records_in_place_ms": t_inplace * 1e3, "allgather_records_in_place_ok": inplace_ok}))
dist_util.shutdown()
