"""CPU: QA batch layout (a1) against the reference's own builder, LR schedule and weight-decay grouping of the optimizer step."""
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_build_tokens_matches_reference_builder():
    from emdr2_amd.tasks.openqa.e2eqa.train_data_utils import build_tokens_types_paddings_from_ids
    g = np.load(os.path.join(GOLD, "a1_ref.npz"), allow_pickle=True)
    for (q, a), ref in zip(g["cases"], g["outs"]):
        got = build_tokens_types_paddings_from_ids(list(q), list(a), 24, 8, 2, 3, 0, 250, 251)
        assert [list(x) if isinstance(x, (list, tuple)) else x for x in got] == [list(x) if isinstance(x, (list, tuple)) else x for x in ref]


def test_collate_has_the_ten_reference_keys():
    from emdr2_amd.tasks.openqa.e2eqa.train_data_utils import build_sample, collate
    items = [build_sample(-(i + 1), [7, 8, 9 + i], [11, 12], 24, 8, 2, 3, 0, 250, 251, reference=["x"]) for i in range(3)]
    b = collate(items)
    assert list(b.keys()) == ["query_uid", "query_ids_bert", "query_types", "query_mask_bert", "query_ids_t5", "query_ids_t5_len",
                              "dec_ids", "labels", "loss_mask", "reference"]
    assert b["query_ids_bert"].shape == (3, 24) and b["query_mask_bert"].shape == (3, 24, 24) and b["loss_mask"].dtype == torch.float32
    assert b["dec_ids"][0].tolist()[:3] == [250, 11, 12] and b["labels"][0].tolist()[:3] == [11, 12, 251]


def test_annealing_lr_matches_reference_table():
    from emdr2_amd.training import AnnealingLR
    g = np.load(os.path.join(GOLD, "model_ref.npz"))
    s = AnnealingLR(2e-5, 10, 1000)
    ours = [s.step() for _ in range(1000)]
    np.testing.assert_allclose(ours, g["lr_table"], rtol=1e-12, atol=0)


def test_weight_decay_groups():
    from emdr2_amd.training import get_params_for_weight_decay_optimization

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.dense = torch.nn.Linear(4, 4)
            self.input_layernorm = torch.nn.LayerNorm(4)
    groups = get_params_for_weight_decay_optimization(M())
    assert len(groups[0]["params"]) == 1 and len(groups[1]["params"]) == 3 and groups[1]["weight_decay"] == 0.0
