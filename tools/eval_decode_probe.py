"""Greedy answer generation (the EM evaluation path) at the benchmark shape on the GPU: retrieval + assembly + reader encoder once
(EMDR2Model eval forward with dec_ids=None), then incremental decoding with K/V caches.   usage: python tools/eval_decode_probe.py [--batch 64] [--rows N]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_e2e

ap = argparse.ArgumentParser()
bench_e2e.add_args(ap)
ap.add_argument("--rows", type=int, default=2_626_916)
ap.add_argument("--topk", type=int, default=50)
ap.add_argument("--max-decode-len", type=int, default=32)
ap.add_argument("--no-split-keys", action="store_true", help="A/B: the cross-attention of the decoding steps as ONE workgroup per (question, head) (r04)")
args = ap.parse_args()
torch.cuda.set_device(0)
ctx = bench_e2e.setup(args, 0, 1, topk=args.topk)
if args.no_split_keys:
    from emdr2_amd.model import kernels as _K
    _K._splitkv_plan = lambda *a: (1, 0, 0)
from emdr2_amd.model.search_strategy import SampleOrGreedySearch
m = ctx.model.eval()
B = args.batch
g = torch.Generator(device="cuda").manual_seed(3)
qlen = torch.randint(10, 27, (B,), generator=g, device="cuda")
q = torch.randint(5, 30522, (B, args.seq_ret), generator=g, device="cuda")
q[:, 0] = 101
q = torch.where(torch.arange(args.seq_ret, device="cuda")[None, :] < qlen[:, None], q, torch.zeros_like(q))
q[torch.arange(B), qlen - 1] = 102
uid = -torch.arange(1, B + 1, device="cuda")
for incremental in (True, False):
    s = SampleOrGreedySearch(args.max_decode_len, 30522, 30523, sample=False, topk_evidence=args.topk, incremental=incremental)
    s.eos_id = -1                                            # random weights never emit [EOS] on cue: decode all positions
    outs = s.generate_output(m, uid, q, torch.zeros_like(q), None, q, qlen.to(torch.int64)); torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = s.generate_output(m, uid, q, torch.zeros_like(q), None, q, qlen.to(torch.int64)); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("greedy decode, B=%d, K=%d, %d positions, %s: %.2f s per batch (%d answers of %d tokens)" %
          (B, args.topk, args.max_decode_len, "incremental (K/V caches)" if incremental else "block form (reference's re-decode)", dt, len(outs), len(outs[0])), flush=True)
    if incremental:
        first = outs
print("same tokens in both forms:", first == outs)
