"""Fixture generator (build container only): the nested key layout of the REFERENCE's EMDR2 checkpoint
(`EMDR2Model.state_dict_for_save_checkpoint()`, what megatron/checkpointing.py:94-135 stores under 'model') at tiny dims, as
{path: shape} in tests/golden/ckpt_layout.json, plus the EM-metric answers of the reference's tasks/openqa/e2eqa/eval_utils.py on a few string pairs.

    python tests/golden/gen_ckpt_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_model_golden as g  # noqa: E402


def walk(d, prefix, out):
    for k, v in d.items():
        p = prefix + [k]
        if isinstance(v, dict):
            walk(v, p, out)
        else:
            out["\t".join(p)] = list(v.shape)


def main():
    args = g.setup()
    import torch
    from megatron.model.emdr2_model import EMDR2Model

    class FakeRetriever:
        pass
    with g._ref_import.cuda_calls_on_cpu():
        model = EMDR2Model(FakeRetriever())
    out = {}
    walk(model.state_dict_for_save_checkpoint(), [], out)
    from tasks.openqa.e2eqa.eval_utils import exact_match_score, metric_max_over_ground_truths
    pairs = [("The Eiffel Tower", ["eiffel tower"]), ("an  apple, a day!", ["Apple day"]), ("Paris", ["London", "paris."]), ("1999", ["1,999"]),
             ("new-york", ["new york"]), ("", ["the"]), ("A", ["b"])]
    em = [bool(metric_max_over_ground_truths(exact_match_score, h, r)) for h, r in pairs]
    # --ret-kldiv retriever loss (train_e2eqa.py:184-214) on small random inputs
    from tasks.openqa.e2eqa.train_e2eqa import get_kl_div_retriever
    gen = torch.Generator().manual_seed(7)
    B, K, L, V = 3, 4, 6, 11
    one = torch.randn((B, K, L, V), generator=gen)
    tlp = torch.log_softmax(torch.randn((B, K), generator=gen), dim=1)
    labels = torch.randint(1, V, (B, L), generator=gen)
    mask = (torch.rand((B, L), generator=gen) > 0.3).float(); mask[:, 0] = 1.0
    kl = get_kl_div_retriever(one, tlp, labels, mask)
    kl_case = {"one": one.tolist(), "tlp": tlp.tolist(), "labels": labels.tolist(), "mask": mask.tolist(), "loss": float(kl)}
    json.dump({"layout": out, "em_pairs": pairs, "em": em, "kl": kl_case}, open(os.path.join(HERE, "ckpt_layout.json"), "w"), indent=0)
    print(len(out), em)


if __name__ == "__main__":
    main()
