"""BASELINE.json configs[0] AS WRITTEN (SURVEY.md 8d "Config 1"): a 10,000-passage toy index (10,000 x 768 rows fp16 N(0,1), seed 1234), 128 synthetic
QA pairs (questions U[8,24] tokens, answers U[1,6], token ids U[5,30521]), passages U[100,160] tokens, titles U[2,8], title groups of 1-10
consecutive ids, B = 8 questions per step, K = 50, S_ret 256, S 512, L 32, BERT-base towers + the 12 + 12-layer reader, fp32, dropout 0,
random-init weights (seed 1234, std 0.02).

  CPU leg  -- the reference's CPU-runnable case restated by the oracles (oracle/: query tower -> exact MIPS in C -> evidence fetch + token
              assembly -> context tower + reader + one-context pass -> EMDR2 loss -> autograd backward), timed on the host cores: steps/s.
  HIP leg  -- the SAME weights and the SAME batch through the product path (HipIndexShard search of the oracle's own fp16 queries, device
              evidence assembly, EMDR2Model.forward_assembled + emdr2_loss + backward) for parity: retrieved ids and scores bit-identical,
              assembled token tensors identical, losses within bf16 tolerance; its step time for the record.

One step of the CPU leg is ~260 TFLOP of fp32 GEMMs (minutes on any host): this is a one-off measurement tool, not part of bench.py; its
output is committed under profiles/.     usage: python tools/config0.py [--threads N] [--skip-cpu-backward] [--out profiles/r03_config0.json]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import assembly_oracle as ao  # noqa: E402
from oracle import mips_oracle as mo  # noqa: E402
from oracle import transformer_oracle as to  # noqa: E402

N_DOCS, DIM, B, K, S_RET, S, L = 10_000, 768, 8, 50, 256, 512, 32
V_BERT, V_T5, CLS, SEP, PAD, BOS, EOS = 30592, 30720, 101, 102, 0, 30522, 30523
CFG = dict(layers=12, hidden=768, heads=12, ffn=3072)


def world(seed=1234):
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    rows = torch.randn((N_DOCS, DIM)).to(torch.float16).numpy()
    sizes, left = [], N_DOCS
    while left > 0:
        s = int(min(left, rng.integers(1, 11))); sizes.append(s); left -= s
    groups, d = [], 1
    for s in sizes:
        groups.append(list(range(d, d + s))); d += s
    group_of_doc = {doc: g for g in groups for doc in g}
    passages = [rng.integers(5, 30522, size=int(rng.integers(100, 161))).tolist() for _ in range(N_DOCS)]
    titles = [None] * N_DOCS
    for g in groups:
        t = rng.integers(5, 30522, size=int(rng.integers(2, 9))).tolist()
        for doc in g:
            titles[doc - 1] = t
    qa = []
    for i in range(128):
        q = [CLS] + rng.integers(5, 30522, size=int(rng.integers(8, 25))).tolist() + [SEP]
        a = rng.integers(5, 30522, size=int(rng.integers(1, 7))).tolist()
        qa.append((-(i + 1), q, a))
    return rows, passages, titles, group_of_doc, groups, qa


def batch(qa, i0):
    qb = np.zeros((B, S_RET), dtype=np.int64); dec = np.zeros((B, L), dtype=np.int64); labels = np.zeros((B, L), dtype=np.int64)
    uid, qlen = [], []
    for r, (u, q, a) in enumerate(qa[i0:i0 + B]):
        qb[r, :len(q)] = q; uid.append(u); qlen.append(len(q))
        dec[r, 0] = BOS; dec[r, 1:1 + len(a)] = a
        labels[r, :len(a)] = a; labels[r, len(a)] = EOS
    return np.array(uid), torch.from_numpy(qb), np.array(qlen), torch.from_numpy(dec), torch.from_numpy(labels), (torch.from_numpy(labels) != 0).float()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=32)
    ap.add_argument("--skip-cpu-backward", action="store_true")
    ap.add_argument("--out", default="")
    ap.add_argument("--quick", action="store_true", help="plumbing check: B = 2, K = 4 (not configs[0])")
    args = ap.parse_args()
    if args.quick:
        global B, K
        B, K = 2, 4
    torch.set_num_threads(min(args.threads, os.cpu_count() or 1))
    rows, passages, titles, group_of_doc, groups, qa = world()
    P = to.random_params(CFG, V_BERT, V_T5, seed=1234)
    uid, qb, qlen, dec, labels, mask = batch(qa, 0)
    res = {"config": "BASELINE configs[0]: %d x %d fp16 index, 128 QA pairs, B=%d, K=%d, S_ret %d, S %d, L %d, 12-layer towers + 12+12-layer reader, fp32, "
                     "dropout 0" % (N_DOCS, DIM, B, K, S_RET, S, L), "host_threads": torch.get_num_threads(), "host_cores": os.cpu_count()}

    # ---- CPU leg --------------------------------------------------------------------------------------------------------------------
    Pg = {k: v.clone().requires_grad_(not args.skip_cpu_backward) for k, v in P.items()}
    t0 = time.perf_counter()
    with torch.no_grad():
        q_emb = to.bert_embed(Pg, "retriever_model.query_model", CFG, qb, ~to.make_attention_mask_3d(qb, qb), torch.zeros_like(qb))
    q16 = q_emb.to(torch.float16).numpy()
    t1 = time.perf_counter()
    dist, ids = mo.topk(rows, q16, K + 1, ids=np.arange(1, N_DOCS + 1, dtype=np.int32))          # --topk-retrievals 50 (+1: trivial-doc slot)
    t2 = time.perf_counter()
    corpus = ao.Corpus(passages, titles, group_of_doc)
    ctx, typ, ext, one, kept = ao.postprocess(uid.tolist(), qb.tolist(), qlen.tolist(), ids.tolist(), corpus, K, S_RET, S, CLS, SEP, PAD)
    tt = lambda x: torch.tensor(x, dtype=torch.int64)
    ctx, typ, ext, one = tt(ctx), tt(typ), tt(ext), tt(one)
    t3 = time.perf_counter()
    grad_ctx = torch.enable_grad() if not args.skip_cpu_backward else torch.no_grad()
    with grad_ctx:
        lm, tlp, oc = to.emdr2_forward(Pg, CFG, qb, torch.zeros_like(qb), ~to.make_attention_mask_3d(qb, qb), ctx, typ, ext, one, dec)
        lm_loss = to.reader_ce_loss(lm, labels, mask)
        r_loss = to.retriever_loss_and_utility(oc, tlp, labels, mask, EOS)[0]
        t4 = time.perf_counter()
        if not args.skip_cpu_backward:
            (lm_loss + r_loss).backward()
    t5 = time.perf_counter()
    step_s = t5 - t0
    res["cpu"] = {"query_tower_s": t1 - t0, "mips_exact_s": t2 - t1, "assembly_s": t3 - t2, "forward_loss_s": t4 - t3, "backward_s": t5 - t4,
                  "step_s": step_s, "steps_per_s": 1.0 / step_s, "backward_included": not args.skip_cpu_backward,
                  "lm_loss": float(lm_loss), "retriever_loss": float(r_loss), "kind": "port (oracle restatement of the reference's CPU path)"}
    print("CPU leg:", json.dumps(res["cpu"]), flush=True)

    # ---- HIP leg: same weights, same batch -----------------------------------------------------------------------------------------
    if torch.cuda.is_available():
        from emdr2_amd.data.emdr2_index import HipIndexShard
        from emdr2_amd.data.evidence_arena import EvidenceArena
        from emdr2_amd.model import kernels as Kmod
        from emdr2_amd.model.emdr2_model import EMDR2Model, emdr2_loss
        from emdr2_amd.model.transformer import Config
        torch.cuda.set_device(0)
        shard = HipIndexShard(DIM, N_DOCS, 0)
        shard.append_rows(rows)
        shard.set_ids(torch.arange(1, N_DOCS + 1, dtype=torch.int32, device="cuda"))
        gd, gi, _, flags = shard.search(torch.from_numpy(q16).cuda(), K + 1)
        torch.cuda.synchronize()
        ids_equal = bool(np.array_equal(gi.cpu().numpy(), ids)) and bool(np.array_equal(gd.cpu().numpy().view(np.uint16), dist.view(np.uint16)))
        arena = EvidenceArena(passages, titles, title_keys=[tuple(group_of_doc[d + 1]) for d in range(N_DOCS)])
        out = arena.assemble(gi, K, torch.from_numpy(uid).cuda(), qb.cuda(), torch.from_numpy(qlen).cuda(), S_RET, S, CLS, SEP, PAD)
        g_ctx, g_typ, g_ext, g_one = out[0], out[1], out[2], out[3]
        asm_equal = bool(torch.equal(g_ctx.cpu(), ctx) and torch.equal(g_typ.cpu(), typ) and torch.equal(g_ext.cpu(), ext) and torch.equal(g_one.cpu(), one))
        cfg = Config(num_layers=12, hidden_size=768, num_attention_heads=12, ffn_hidden_size=3072, max_position_embeddings=512, init_method_std=0.02)
        m = EMDR2Model(None, cfg, V_T5, V_BERT, K, S, S_RET, cls_id=CLS, sep_id=SEP)
        m.load_state_dict({k: v for k, v in P.items()})
        m.train()

        def step():
            for p in m.parameters():
                p.grad = None
            ql = m.retriever_embedder(qb.cuda(), None, torch.zeros_like(qb).cuda(), "query")
            lmg, tlpg, oneg = m.forward_assembled(ql, g_ctx, g_typ, g_ext, g_one, dec.cuda())
            loss, stats = emdr2_loss(lmg, tlpg, oneg, labels.cuda(), mask.cuda(), eos_id=EOS)
            loss.backward()
            return stats
        stats = step(); torch.cuda.synchronize()
        t0 = time.perf_counter(); stats = step(); torch.cuda.synchronize(); g_step = time.perf_counter() - t0
        res["hip"] = {"retrieved_ids_and_scores_bit_identical": ids_equal, "unproven_queries": int(flags.abs().sum()), "assembled_tokens_identical": asm_equal,
                      "lm_loss": float(stats["lm_loss"]), "retriever_loss": float(stats["retriever_loss"]),
                      "lm_loss_rel_diff": abs(float(stats["lm_loss"]) - float(lm_loss)) / abs(float(lm_loss)),
                      "retriever_loss_rel_diff": abs(float(stats["retriever_loss"]) - float(r_loss)) / abs(float(r_loss)),
                      "model_step_s": g_step, "note": "model forward + loss + backward of the same batch (bf16, packed sequences: %s); the search "
                                                      "and the assembly are checked, not timed, here" % Kmod.PACKING.enabled}
        print("HIP leg:", json.dumps(res["hip"]), flush=True)
        assert ids_equal and asm_equal
    if args.out:
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
