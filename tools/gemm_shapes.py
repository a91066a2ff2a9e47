"""Per-shape GEMM table of one end-to-end EMDR2 step (GPU): every dense-GEMM launch of a timed step, bracketed by events on the launch stream,
grouped by (kind, M, N, K, epilogue) -> launches, ms, TFLOP/s, fraction of the dense bf16 MFMA peak, kernel that takes the shape.

usage: python tools/gemm_shapes.py [bench_e2e flags] [--out profiles/r02_gemm_summary.json]
The step is the bench_e2e.py step (B = 64, top-k 50, 12 layers, full index); launches are timed one by one, so the step itself runs
slower than in bench.py (events serialise nothing, but ~1,700 event pairs are recorded and read back)."""
import argparse
import collections
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_e2e  # noqa: E402

PEAK = bench_e2e.MFMA_PEAK_TFLOPS


def main():
    ap = argparse.ArgumentParser()
    bench_e2e.add_args(ap)
    ap.add_argument("--rows", type=int, default=21_015_324)
    ap.add_argument("--topk", type=int, default=50)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    torch.cuda.set_device(0)
    from emdr2_amd.model import kernels as K
    ctx = bench_e2e.setup(args, 0, 1, topk=args.topk)
    if args.selective_layers != "auto":                       # the retention plan bench.py's auto policy picks on a 288 GB box: pass it explicitly
        sel_r, sel_c = (int(v) for v in args.selective_layers.split(","))
        ctx.model.set_selective_retention(sel_r, sel_c, ctx.layers if sel_c else 0)
    if args.keep_last_layers != "auto":
        ctx.model.set_recompute_keep_last(int(args.keep_last_layers))
    records = []
    live = {"on": False}

    def timed(key, flops, call):
        if not live["on"]:
            return call()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = call()
        b.record()
        records.append((key, flops, a, b))
        return out

    nt_raw, tn_raw, lse_raw = K.gemm_nt, K.weight_grad_tn, K.lm_head_gold_logprob

    def gemm_nt(A, lda, B, ldb, C, ldc, M, N, Kd, batch1=1, sA1=0, sB1=0, sC1=0, batch2=1, sA2=0, sB2=0, sC2=0, alpha=1.0, bias=None, gelu=False,
                pre_act=None, residual=None, split_k=1, drop_p=0.0, seed=0, residual_mode=0):
        epi = "+".join(n for n, on in (("bias", bias is not None), ("gelu", gelu), ("pre", pre_act is not None and gelu != 2), ("gelu'out", gelu == 2),
                                      ("drop", drop_p > 0), ("res", residual is not None and residual_mode == 0),
                                      ("gelu'", residual is not None and residual_mode == 1), ("x gelu'saved", residual is not None and residual_mode == 2),
                                      ("f32out", C.dtype == torch.float32), ("splitk%d" % split_k, split_k > 1)) if on) or "plain"
        nb = batch1 * batch2
        g8 = nb == 1 and split_k == 1 and C.dtype != torch.float32 and M % 256 == 0 and N % 256 == 0 and Kd % 128 == 0 and M >= 4096
        key = ("nt", M, N, Kd, nb, epi, "gemm8_kernel" if g8 else "gemm_nt_kernel")
        return timed(key, 2.0 * M * N * Kd * nb, lambda: nt_raw(A, lda, B, ldb, C, ldc, M, N, Kd, batch1, sA1, sB1, sC1, batch2, sA2, sB2, sC2, alpha, bias,
                                                              gelu, pre_act, residual, split_k, drop_p, seed, residual_mode))

    def weight_grad_tn(dy, x, colsum=None, into=None):
        M, N = dy.shape
        Kd = x.shape[1]
        key = ("tn", M, N, Kd, 1, "colsum" if colsum is not None else "plain", "gemm8t_kernel" if M % 64 == 0 else "gemm_tn_kernel")
        return timed(key, 2.0 * M * N * Kd, lambda: tn_raw(dy, x, colsum, into=into))

    def lm_head_gold_logprob(hidden, weight, bias, labels):
        V, H = weight.shape
        M = hidden.numel() // H
        fused = M % 256 == 0 and V % 256 == 0 and H % 128 == 0
        key = ("nt", M, V, H, 1, "lse(max,sumexp,gold)" + ("+bias" if bias is not None else ""), "gemm8_kernel<LSE> + lse_combine" if fused else "unfused")
        return timed(key, 2.0 * M * V * H, lambda: lse_raw(hidden, weight, bias, labels))

    K.gemm_nt, K.weight_grad_tn, K.lm_head_gold_logprob = gemm_nt, weight_grad_tn, lm_head_gold_logprob
    import emdr2_amd.model.emdr2_model as em
    if hasattr(em, "K"):
        em.K.lm_head_gold_logprob = lm_head_gold_logprob
    for _ in range(2):
        ctx.step()
    torch.cuda.synchronize()
    live["on"] = True
    ctx.step()
    torch.cuda.synchronize()
    live["on"] = False
    table = collections.OrderedDict()
    for key, flops, a, b in records:
        e = table.setdefault(key, {"launches": 0, "ms": 0.0, "flops": 0.0})
        e["launches"] += 1; e["ms"] += a.elapsed_time(b); e["flops"] += flops
    rows = []
    for (kind, M, N, Kd, nb, epi, kernel), e in table.items():
        tf = e["flops"] / (e["ms"] * 1e-3) / 1e12
        rows.append({"kind": kind, "M": M, "N": N, "K": Kd, "batch": nb, "epilogue": epi, "kernel": kernel, "launches": e["launches"], "ms": round(e["ms"], 3),
                     "tflops": round(tf, 1), "frac_of_peak": round(tf / PEAK, 3)})
    rows.sort(key=lambda r: -r["ms"])
    tot_ms, tot_fl = sum(e["ms"] for e in table.values()), sum(e["flops"] for e in table.values())
    summary = {"workload": "one EMDR2 end-to-end step (bench_e2e.py: B=%d, top-k %d, %d layers, %d-row index), per-launch events" % (ctx.B, ctx.K, ctx.layers, ctx.rows),
               "peak_tflops": PEAK, "gemm_ms_per_step": round(tot_ms, 1), "gemm_tflops": round(tot_fl / (tot_ms * 1e-3) / 1e12, 1),
               "gemm_frac_of_peak": round(tot_fl / (tot_ms * 1e-3) / 1e12 / PEAK, 3), "launches": len(records), "shapes": rows}
    print("%-3s %8s %6s %6s %5s  %-28s %-30s %5s %9s %8s" % ("", "M", "N", "K", "batch", "epilogue", "kernel", "n", "ms", "TFLOP/s"))
    for r in rows:
        print("%-3s %8d %6d %6d %5d  %-28s %-30s %5d %9.2f %8.1f" % (r["kind"], r["M"], r["N"], r["K"], r["batch"], r["epilogue"], r["kernel"], r["launches"],
                                                                   r["ms"], r["tflops"]))
    print("total: %.1f ms, %.1f TFLOP/s (%.3f of %.0f)" % (tot_ms, summary["gemm_tflops"], summary["gemm_frac_of_peak"], PEAK))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(summary, f, indent=1)


if __name__ == "__main__":
    main()
