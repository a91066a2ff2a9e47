"""The decoder-side attention launches of the step at micro-batch sizes (GPU): FiD cross-attention of b questions x 32 decoder positions over
the K packed passages of each question (~20k keys), the one-context pass's cross-attention (b*K sequences x 32 over ~160 keys), and the three
encoder stacks at b*K sequences.   usage: python tools/attn_cross_bench.py [--questions 64,16,8] [--topk 50] [--drop 0.1]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emdr2_amd.model import kernels as K  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--questions", default="64,16,8")
ap.add_argument("--topk", type=int, default=50)
ap.add_argument("--drop", type=float, default=0.1)
ap.add_argument("--only-fid-cross", action="store_true", help="time only the FiD cross-attention (for per-kernel traces)")
args = ap.parse_args()
heads, hn, L = 12, 64, 32
g = torch.Generator(device="cuda").manual_seed(0)


def timed(fn, reps=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def case(name, q, kv, ids_q, ids_k, pairs, causal=False):
    out = K.attention_core(q, kv, ids_q, ids_k, causal, drop_p=args.drop, seed=1)
    dy = torch.randn_like(out)
    qd, kvd = q.detach(), (kv.detach() if kv is not None else None)
    t_f = timed(lambda: K.attention_core(qd, kvd, ids_q, ids_k, causal, drop_p=args.drop, seed=1))

    def fb():
        o = K.attention_core(q, kv, ids_q, ids_k, causal, drop_p=args.drop, seed=1)
        o.backward(dy)
        q.grad = None
        if kv is not None:
            kv.grad = None
    t_b = timed(fb) - t_f
    fl = 4.0 * heads * pairs * hn
    print("%-46s fwd %7.3f ms %4.0f TF/s | bwd %7.3f ms %4.0f TF/s" % (name, t_f * 1e3, fl / t_f / 1e12, t_b * 1e3, 2.5 * fl / t_b / 1e12), flush=True)


for b in (int(v) for v in args.questions.split(",")):
    n = b * args.topk
    print("---- %d questions x top-%d = %d sequences" % (b, args.topk, n), flush=True)
    for name, S, lo, hi in (("context tower", 256, 105, 171), ("one-context encoder", 512, 125, 200), ("reader encoder", 512, 300, 512)):
        lens = torch.randint(lo, hi + 1, (n,), generator=g, device="cuda").clamp(max=S)
        ids = (torch.arange(S, device="cuda")[None, :] < lens[:, None]).long() * 7
        seqs = K.PackedSeqs(ids)
        qkv = torch.randn((seqs.rows, 3, heads, hn), generator=g, device="cuda").bfloat16().requires_grad_(True)
        if not args.only_fid_cross:
            case("%s (self, %d rows)" % (name, seqs.rows), qkv, None, seqs, seqs, seqs.pairs)
        if name != "context tower" and (name == "reader encoder" or not args.only_fid_cross):
            # the decoder's cross-attention over these encoder rows: FiD = b questions x L positions over the K passages of a question;
            # one-context = b*K sequences x L positions over one passage each
            grp = seqs.grouped(args.topk) if name == "reader encoder" else seqs
            nq = grp.n
            alen = torch.randint(2, 8, (nq,), generator=g, device="cuda")
            dec_ids = (torch.arange(L, device="cuda")[None, :] < alen[:, None]).long() * 9
            q = torch.randn((nq, L, heads, hn), generator=g, device="cuda").bfloat16().requires_grad_(True)
            kv = torch.randn((seqs.rows, 2, heads, hn), generator=g, device="cuda").bfloat16().requires_grad_(True)
            case("  cross-attention %d x %d queries over it" % (nq, L), q, kv, dec_ids, grp, float(L) * seqs.total)
            del q, kv
        del qkv
    dec = torch.randint(2, 8, (b,), generator=g, device="cuda")
    torch.cuda.empty_cache()
