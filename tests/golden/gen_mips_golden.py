"""Generate tests/golden/mips_ref_<case>.npz by RUNNING THE REFERENCE's own
DistributedBruteForceIndex.search_mips_index (megatron/data/emdr2_index.py:200-305) in the build
container, on CPU, with device placement shimmed (tests/golden/_ref_import.py).

Run here only:  python tests/golden/gen_mips_golden.py
The .npz files hold OUTPUTS (distances, doc ids) + an input digest; inputs are rebuilt from seeds
by tests/golden/mips_cases.py.  Nothing of the reference's source is stored.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402
import mips_cases  # noqa: E402


def run_reference(case, device_count):
    _ref_import.install_import_shims()
    from megatron.data.emdr2_index import OpenRetreivalDataStore, DistributedBruteForceIndex
    store = OpenRetreivalDataStore(embedding_path="/tmp/_unused_embed.pkl", load_from_path=False, rank=0)
    # insertion order of the dict == row order of the matrix (emdr2_index.py:245)
    store.add_block_data([int(i) for i in case["ids"]], case["rows"])
    with _ref_import.cuda_calls_on_cpu(device_count=device_count):
        index = DistributedBruteForceIndex(embed_size=case["rows"].shape[1], embed_data=store, use_gpu=True)
        dist, idx = index.search_mips_index(torch.from_numpy(case["queries"]), case["k"], reconstruct=False)
    return dist.numpy().copy(), idx.numpy().copy()


def main():
    torch.set_num_threads(8)
    for fn in mips_cases.ALL_CASES:
        case = fn()
        d1, i1 = run_reference(case, 1)
        d3, i3 = run_reference(case, 3)      # 3 "devices": chunked matmul + concat path (emdr2_index.py:252-292)
        same = np.array_equal(d1.view(np.uint16), d3.view(np.uint16)) and np.array_equal(i1, i3)
        out = os.path.join(HERE, "mips_ref_%s.npz" % case["name"])
        np.savez_compressed(out, dist=d1.view(np.uint16), idx=i1.astype(np.int32),
                            dist_3dev=d3.view(np.uint16), idx_3dev=i3.astype(np.int32),
                            digest=np.array(mips_cases.digest(case)), torch_version=np.array(torch.__version__))
        print(case["name"], "saved", out, "1dev==3dev:", same, d1.shape, d1.dtype, i1.dtype)


if __name__ == "__main__":
    main()
