"""Loss of the first N end-to-end EMDR2 training steps at the benchmark configuration (B = 64, top-k 50, 12 layers, full index; GPU):
every step draws a fresh synthetic batch (random answers), so what can be learned is the answer-token statistics: a working forward /
backward / optimizer chain drives the loss from ~21 (random init) to ~17 within the learning-rate warm-up and holds it there.
usage: python tools/e2e_loss_curve.py [--steps 16] [--out profiles/r02_e2e_loss_curve.json]  (+ bench_e2e flags)"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_e2e

ap = argparse.ArgumentParser()
bench_e2e.add_args(ap)
ap.add_argument("--rows", type=int, default=21_015_324)
ap.add_argument("--topk", type=int, default=50)
ap.add_argument("--steps", type=int, default=16)
ap.add_argument("--out", default="")
args = ap.parse_args()
torch.cuda.set_device(0)
ctx = bench_e2e.setup(args, 0, 1, topk=args.topk)
if args.selective_layers != "auto":                           # e.g. --selective-layers 12,8: the plan bench.py picks on a 288 GB box
    sel_r, sel_c = (int(v) for v in args.selective_layers.split(","))
    ctx.model.set_selective_retention(sel_r, sel_c, ctx.layers if sel_c else 0)
losses, t0 = [], time.perf_counter()
for i in range(args.steps):
    loss = ctx.step()
    losses.append(float(loss.detach()))
torch.cuda.synchronize()
from emdr2_amd.model import kernels as Kmod
res = {"workload": "bench_e2e.py step, B=%d, top-k %d, %d layers, %d-row index, lr warm-up 10 steps to 2e-5, dropout %.1f, packed sequences %s, question "
                   "micro-batches %d (1 = undivided, selective retention %s)"
                   % (ctx.B, ctx.K, ctx.layers, ctx.rows, ctx.dropout, Kmod.PACKING.enabled, ctx.guard.micro, args.selective_layers),
       "loss_per_step": losses, "seconds": time.perf_counter() - t0}
print(json.dumps(res))
if args.out:
    json.dump(res, open(args.out, "w"), indent=1)
