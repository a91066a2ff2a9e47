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
losses, t0 = [], time.perf_counter()
for i in range(args.steps):
    loss = ctx.step()
    losses.append(float(loss.detach()))
torch.cuda.synchronize()
res = {"workload": "bench_e2e.py step, B=%d, top-k %d, %d layers, %d-row index, lr warm-up 10 steps to 2e-5, dropout %.1f" % (ctx.B, ctx.K, ctx.layers, ctx.rows, ctx.dropout),
       "loss_per_step": losses, "seconds": time.perf_counter() - t0}
print(json.dumps(res))
if args.out:
    json.dump(res, open(args.out, "w"), indent=1)
