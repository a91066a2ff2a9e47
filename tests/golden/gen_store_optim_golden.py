"""Fixture generator (build container only; imports /root/reference) for SURVEY 8c's F6 (second half) and F7:

  F6  get_params_for_weight_decay_optimization (megatron/model/utils.py:64-83) on the REFERENCE's EMDR2Model at tiny dims:
      the parameter names of its two groups  ->  tests/golden/optim_groups.json
  F7  files WRITTEN by the reference's OpenRetreivalDataStore (megatron/data/emdr2_index.py:16-100): two ranks add_block_data + save_shard,
      rank 0 merge_shards_and_save  ->  tests/golden/store_ref.pkl (the merged 100-row store), store_ref_shard1.pkl (rank 1's shard as the
      reference wrote it), store_ref.npz (the fp32 rows that went in, the ids, the insertion order of the merged dict)

    python tests/golden/gen_store_optim_golden.py
"""
import json
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_model_golden as g  # noqa: E402


def main():
    g.setup()
    from megatron.model.emdr2_model import EMDR2Model
    from megatron.model.utils import get_params_for_weight_decay_optimization

    class FakeRetriever:
        pass
    with g._ref_import.cuda_calls_on_cpu():
        model = EMDR2Model(FakeRetriever())
    name_of = {id(p): n for n, p in model.named_parameters()}
    decay, no_decay = get_params_for_weight_decay_optimization(model)
    groups = {"weight_decay": sorted(set(name_of[id(p)] for p in decay["params"])),
              "no_weight_decay": sorted(set(name_of[id(p)] for p in no_decay["params"])),
              "no_weight_decay_value": no_decay["weight_decay"]}
    assert len(groups["weight_decay"]) + len(groups["no_weight_decay"]) == len(name_of)
    json.dump(groups, open(os.path.join(HERE, "optim_groups.json"), "w"), indent=0)

    from megatron.data.emdr2_index import OpenRetreivalDataStore
    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, "emb.pkl")
    rng = np.random.default_rng(77)
    rows = rng.standard_normal((100, 16)).astype(np.float32) * 3.0
    ids = (rng.permutation(100) + 1).astype(np.int64)                   # 1-based doc ids in a shuffled order (dict insertion order matters)
    stores = [OpenRetreivalDataStore(path, load_from_path=False, rank=r) for r in range(2)]
    stores[0].add_block_data([int(i) for i in ids[:60]], rows[:60])
    stores[1].add_block_data([int(i) for i in ids[60:]], rows[60:])
    for st in stores:
        st.save_shard()
    shutil.copy(os.path.join(stores[1].temp_dir_name, "1.pkl"), os.path.join(HERE, "store_ref_shard1.pkl"))
    stores[0].merge_shards_and_save()
    shutil.copy(path, os.path.join(HERE, "store_ref.pkl"))
    np.savez_compressed(os.path.join(HERE, "store_ref.npz"), rows=rows, ids=ids, merged_order=np.array(list(stores[0].embed_data), dtype=np.int64))
    shutil.rmtree(tmp, ignore_errors=True)
    print(len(groups["weight_decay"]), len(groups["no_weight_decay"]), os.path.getsize(os.path.join(HERE, "store_ref.pkl")))


if __name__ == "__main__":
    main()
