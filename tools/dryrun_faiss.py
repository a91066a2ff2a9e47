"""2-rank dry run (1 GPU, gloo) of the sharded FaissMIPSIndex: every rank must return the single-shard oracle result.
    EMDR2_SINGLE_DEVICE=1 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/dryrun_faiss.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
torch.cuda.set_device(0)
torch.distributed.init_process_group(backend="gloo")
from emdr2_amd.data.emdr2_index import DistributedBruteForceIndex, FaissMIPSIndex
from oracle import mips_oracle as mo
rng = np.random.default_rng(0)
n, d, nq, k = 30011, 128, 37, 100
rows = rng.standard_normal((n, d)).astype(np.float16); q = rng.standard_normal((nq, d)).astype(np.float16)
ids = (rng.permutation(n) + 1).astype(np.int64)
f = FaissMIPSIndex(d, None, use_gpu=True); f.add_with_ids(rows, ids)
D, I = f.search_mips_index(torch.from_numpy(q), k, reconstruct=False)
od, oi = mo.topk_f32(rows, q, k, ids=ids)
assert np.array_equal(D.view(np.uint32), np.ascontiguousarray(od).view(np.uint32)) and np.array_equal(I, oi)
b = DistributedBruteForceIndex(d, None, use_gpu=True); b.add_arrays(ids.astype(np.int32), rows)
dist, idx = b.search_mips_index(torch.from_numpy(q).cuda(), 50)
od2, oi2 = mo.topk(rows, q, 50, ids=ids.astype(np.int32))
assert np.array_equal(dist.cpu().numpy().view(np.uint16), od2.view(np.uint16)) and np.array_equal(idx.cpu().numpy(), oi2)
print("rank %d: sharded fp32-score and fp16 searches equal the single-shard oracle (shard rows %d..%d)" % ((torch.distributed.get_rank(),) + f.local_rows()), flush=True)
torch.distributed.destroy_process_group()
