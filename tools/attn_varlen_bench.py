"""Fused attention over PACKED sequences at the step's three encoder shapes (GPU): forward and backward time, TFLOP/s on the real
(query, key) pairs, against the dense launch over the padded grid.   usage: python tools/attn_varlen_bench.py [--drop 0.1] [--lib path]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emdr2_amd import _native  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--drop", type=float, default=0.1)
ap.add_argument("--lib", default="")
ap.add_argument("--dense", action="store_true", help="also time the dense launch over the padded grid")
args = ap.parse_args()
if args.lib:
    _native.LIB_PATH = os.path.abspath(args.lib)
from emdr2_amd.model import kernels as K  # noqa: E402

heads, hn = 12, 64
g = torch.Generator(device="cuda").manual_seed(0)


def lengths(n, lo, hi, S):
    return torch.randint(lo, hi + 1, (n,), generator=g, device="cuda").clamp(max=S)


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for name, n, S, lo, hi in (("context tower", 3200, 256, 105, 171), ("one-context reader", 3200, 512, 125, 200), ("reader encoder", 3200, 512, 300, 512)):
    lens = lengths(n, lo, hi, S)
    ids = (torch.arange(S, device="cuda")[None, :] < lens[:, None]).long() * 7
    seqs = K.PackedSeqs(ids)
    qkv = torch.randn((seqs.rows, 3, heads, hn), generator=g, device="cuda").bfloat16().requires_grad_(True)
    pairs = seqs.pairs
    out = K.attention_core(qkv, None, seqs, seqs, False, drop_p=args.drop, seed=1)
    dy = torch.randn_like(out)
    t_f = timed(lambda: K.attention_core(qkv.detach(), None, seqs, seqs, False, drop_p=args.drop, seed=1))

    def fb():
        o = K.attention_core(qkv, None, seqs, seqs, False, drop_p=args.drop, seed=1)
        o.backward(dy)
        qkv.grad = None
    t_fb = timed(fb)
    t_b = t_fb - t_f
    fl = 4.0 * heads * pairs * hn
    print("%-20s packed: %d seqs, mean len %.0f of %d, rows %d | fwd %.3f ms %.0f TF/s | bwd %.3f ms %.0f TF/s" %
          (name, n, float(lens.float().mean()), S, seqs.rows, t_f * 1e3, fl / t_f / 1e12, t_b * 1e3, 2.5 * fl / t_b / 1e12), flush=True)
    if args.dense:
        qd = torch.randn((n, S, 3, heads, hn), generator=g, device="cuda").bfloat16().requires_grad_(True)
        od = K.attention_core(qd, None, ids, ids, False, drop_p=args.drop, seed=1)
        dyd = torch.randn_like(od)
        t_fd = timed(lambda: K.attention_core(qd.detach(), None, ids, ids, False, drop_p=args.drop, seed=1))

        def fbd():
            o = K.attention_core(qd, None, ids, ids, False, drop_p=args.drop, seed=1)
            o.backward(dyd)
            qd.grad = None
        t_bd = timed(fbd) - t_fd
        print("%-20s dense : fwd %.3f ms (%.0f TF/s on real pairs) | bwd %.3f ms (%.0f TF/s on real pairs)" %
              ("", t_fd * 1e3, fl / t_fd / 1e12, t_bd * 1e3, 2.5 * fl / t_bd / 1e12), flush=True)
        del qd, od, dyd
    del qkv, out, dy
    torch.cuda.empty_cache()
