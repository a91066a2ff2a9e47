"""PMC evidence for the reader GEMMs (GPU).  Two modes:

  python tools/gemm_pmc.py run
      the workload rocprofv3 wraps: for every configuration of CONFIGS, LAUNCHES launches of the persistent NT GEMM (csrc/gemm8.hip) or of the
      weight-gradient GEMM (csrc/gemm8t.hip), in a fixed order, nothing else on the device in between
  python tools/gemm_pmc.py summarize <rocprof dir> [<rocprof dir> ...] [--shapes <gemm_shapes.json>] --out profiles/r02_gemm_summary.json
      maps the kernel dispatches of each pass (in order) back to CONFIGS and writes, per configuration: mean duration, TFLOP/s, and every
      counter collected (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16-byte streaming reads on gfx950)

Collect with one pass per counter group (never together with the hip/hsa trace domains), e.g.
  tools/pmc_pass.sh gemm "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES" "TCC_HIT_sum TCC_MISS_sum" \
      -- python $PWD/tools/gemm_pmc.py run"""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# token rows per launch: r02-r04 profiled the undivided batch (3,200 x 512); since r05 the step runs its questions in groups
# (EMDR2Model.forward_backward) and a reader-encoder group of 16 questions x 50 passages packs to 327,680 rows
M_FULL = int(os.environ.get("EMDR2_GEMM_PMC_M", 327680))
LAUNCHES = 3
# (kind, M, N, K, epilogue)
CONFIGS = [("nt", M_FULL, N, K, e) for N, K in ((768, 768), (2304, 768), (3072, 768), (768, 3072))
           for e in ("plain", "bias", "bias+gelu", "bias+gelu+pre", "bias+gelu+gelu'out", "bias+res", "bias+drop+res", "gelu'", "x gelu'saved")] + \
          [("tn", M_FULL, N, K, "colsum") for N, K in ((768, 768), (2304, 768), (3072, 768), (768, 3072))]


def run():
    import torch
    from emdr2_amd.model import kernels as K
    g = torch.Generator(device="cuda").manual_seed(0)

    def rnd(*s):
        return (torch.randn(s, generator=g, device="cuda") * 0.5).bfloat16()
    last = None
    for kind, M, N, Kd, epi in CONFIGS:
        if (kind, N, Kd) != last:
            a = rnd(M, Kd)
            b = rnd(N, Kd) if kind == "nt" else rnd(M, N)
            c = torch.empty((M, N), dtype=torch.bfloat16, device="cuda") if kind == "nt" else None
            bias, r = torch.zeros(N, device="cuda"), (rnd(M, N) if kind == "nt" else None)
            last = (kind, N, Kd)
        for _ in range(LAUNCHES):
            if kind == "tn":
                K.weight_grad_tn(b, a, colsum=bias)          # dW [N, K] = dy [M, N]^T x [M, K]
            else:
                kw = {}
                if "pre" in epi.split("+"): kw["pre_act"] = r                   # (the pre-activation output: same size as the residual buffer)
                if "bias" in epi: kw["bias"] = bias
                if "gelu" in epi.split("+"): kw["gelu"] = True
                if "res" in epi.split("+"): kw["residual"] = r
                if "drop" in epi: kw.update(drop_p=0.1, seed=7)
                if epi == "gelu'": kw.update(residual=r, residual_mode=1)
                if "gelu'out" in epi.split("+"): kw.update(gelu=2, pre_act=r)     # gelu' of the pre-activation as the second output
                if epi == "x gelu'saved": kw.update(residual=r, residual_mode=2)
                K.gemm_nt(a, Kd, b, Kd, c, N, M, N, Kd, **kw)
        torch.cuda.synchronize()


def library_sha256():
    from emdr2_amd import _native
    with open(_native.LIB_PATH, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def summarize(dirs, shapes_json, out, calibration_json=""):
    cal = {"read_stream": 2.0, "read_rowseg": 2.0, "read_lds_dma": 2.0, "write_stream": 1.0, "write_rowseg": 1.0}
    calibrated = False
    if calibration_json and os.path.exists(calibration_json):
        f_ = json.load(open(calibration_json)).get("factors", {})
        cal.update({k: v for k, v in f_.items() if v})
        calibrated = all(f_.get(k) for k in ("read_rowseg", "read_lds_dma", "write_rowseg"))
    per = [dict(kind=k, M=M, N=N, K=Kd, epilogue=e, counters={}) for k, M, N, Kd, e in CONFIGS]
    for d in dirs:
        cc = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
        if not cc:
            continue
        # counter_collection.csv alone (kernel name, timestamps, one row per dispatch and counter): its dispatch ids are not the kernel trace's
        rows = [r for r in csv.DictReader(open(max(cc, key=os.path.getmtime))) if "gemm8" in r["Kernel_Name"]]     # newest run in the directory
        order, ctr = [], collections.defaultdict(dict)
        for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
            if r["Dispatch_Id"] not in ctr:
                order.append((r["Dispatch_Id"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
            ctr[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
        assert len(order) == len(CONFIGS) * LAUNCHES, (d, len(order))
        for i, cfg in enumerate(per):
            mine = order[i * LAUNCHES:(i + 1) * LAUNCHES]
            cfg.setdefault("_dur", []).extend(ms for _, ms in mine)
            names = set().union(*(ctr[k].keys() for k, _ in mine))
            for n in names:
                v = [ctr[k][n] for k, _ in mine if n in ctr[k]]
                cfg["counters"][n] = sum(v) / len(v)
    for cfg in per:
        dur = cfg.pop("_dur", [])
        if not dur:
            continue
        ms = sum(dur) / len(dur)
        fl = 2.0 * cfg["M"] * cfg["N"] * cfg["K"]
        cfg["ms_per_launch_under_pmc"] = round(ms, 3)
        cfg["tflops_under_pmc"] = round(fl / (ms * 1e-3) / 1e12, 1)
        c = cfg["counters"]
        # algorithmic bytes, reads and writes apart: operands (+ the residual / saved pre-activation the epilogue reads) in, outputs out
        if cfg["kind"] == "nt":
            rd = 2.0 * (cfg["M"] * cfg["K"] + cfg["N"] * cfg["K"])
            wr = 2.0 * cfg["M"] * cfg["N"] * (2 if ("pre" in cfg["epilogue"].split("+") or "gelu'out" in cfg["epilogue"].split("+")) else 1)
            if "res" in cfg["epilogue"].split("+") or cfg["epilogue"] in ("gelu'", "x gelu'saved"):
                rd += 2.0 * cfg["M"] * cfg["N"]
        else:
            rd, wr = 2.0 * cfg["M"] * (cfg["N"] + cfg["K"]), 4.0 * cfg["N"] * cfg["K"]
        cfg["algorithmic_read_gb"], cfg["algorithmic_write_gb"] = round(rd / 1e9, 3), round(wr / 1e9, 3)
        cfg["algorithmic_hbm_gb"] = round((rd + wr) / 1e9, 3)
        if "FETCH_SIZE" in c:
            # FETCH_SIZE (KiB) counts each access shape at its own rate (profiles/r05_fetch_calibration.json): the epilogue's residual /
            # saved-gelu' rows are read exactly once (every element by one lane) -- their share of the counter is their bytes / read_rowseg --
            # and what is left is the LDS-DMA operand stream, scaled by read_lds_dma.  TN: both operands are LDS-DMA streams.
            resid = (rd - 2.0 * (cfg["M"] * cfg["K"] + cfg["N"] * cfg["K"])) if cfg["kind"] == "nt" else 0.0
            counted = c["FETCH_SIZE"] * 1024.0
            operands = max(0.0, counted - resid / cal["read_rowseg"]) * cal["read_lds_dma"]
            cfg["hbm_read_gb"] = round((operands + resid) / 1e9, 3)
            cfg["read_ratio"] = round(cfg["hbm_read_gb"] / cfg["algorithmic_read_gb"], 3)
            cfg["operand_read_ratio"] = round(operands / (rd - resid), 3)
        if "WRITE_SIZE" in c:
            wf = cal["write_rowseg"] if cfg["kind"] == "nt" else cal["write_stream"]
            cfg["hbm_write_gb_uncalibrated"] = round(c["WRITE_SIZE"] * 1024 * wf / 1e9, 3)          # (key kept; calibrated when `calibrated` below says so)
            cfg["write_ratio_uncalibrated"] = round(cfg["hbm_write_gb_uncalibrated"] / cfg["algorithmic_write_gb"], 3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
            # MFMA-busy cycles are summed over the 1,024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs
            cfg["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (c["GRBM_GUI_ACTIVE"] / 8.0), 3)
            cfg["shader_clock_ghz"] = round(c["GRBM_GUI_ACTIVE"] / 8.0 / (ms * 1e6), 3)
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
            cfg["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 3)
        cfg["counters"] = {k: round(v, 1) for k, v in sorted(c.items())}
    res = {"what": "reader GEMM kernels one by one at the end-to-end step's shapes (M = %d token rows): %d launches per configuration, mean per launch; "
                   "counters from separate rocprofv3 --pmc passes (kernel-trace only)" % (M_FULL, LAUNCHES),
           "library_sha256": library_sha256(), "calibration": dict(cal, calibrated_on_this_box=calibrated, source=os.path.basename(calibration_json or "")),
           "notes": ["tflops_under_pmc is measured while counters are being collected (a few % slower than free-running)",
                     "hbm_read_gb = 2 x FETCH_SIZE (gfx950 correction of MI355X_MICROARCH.md, HBM section); WRITE_SIZE is reported as is",
                     "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs): the fraction of cycles the matrix pipe works, at the "
                     "clock the launch actually ran at (shader_clock_ghz; the 2.5 PFLOP/s peak assumes 2.4 GHz)"],
           "per_kernel": per}
    if shapes_json and os.path.exists(shapes_json):
        step = json.load(open(shapes_json))
        res["end_to_end_step"] = step
        # step-weighted traffic: every GEMM launch of the step takes the read / write ratio measured for its (kind, N, K, epilogue) above,
        # weighted by its own algorithmic bytes (M differs per stack: the ratios are per byte, not per launch)
        by = {(c["kind"], c["N"], c["K"], c["epilogue"]): c for c in per if "read_ratio" in c}
        tot = {"algorithmic_read_gb": 0.0, "measured_read_gb": 0.0, "algorithmic_write_gb": 0.0, "measured_write_gb_uncalibrated": 0.0, "covered_ms": 0.0, "all_ms": 0.0}
        for r in step["shapes"]:
            tot["all_ms"] += r["ms"]
            epi = {"colsum": "colsum", "plain": "plain"}.get(r["epilogue"], r["epilogue"])
            c = by.get((r["kind"], r["N"], r["K"], epi))
            if c is None or r.get("batch", 1) != 1:
                continue
            if r["kind"] == "nt":
                rd = 2.0 * (r["M"] * r["K"] + r["N"] * r["K"]) + (2.0 * r["M"] * r["N"] if ("res" in epi.split("+") or epi in ("gelu'", "x gelu'saved")) else 0.0)
                wr = 2.0 * r["M"] * r["N"] * (2 if ("pre" in epi.split("+") or "gelu'out" in epi.split("+")) else 1)
            else:
                rd, wr = 2.0 * r["M"] * (r["N"] + r["K"]), 4.0 * r["N"] * r["K"]
            n = r["launches"]
            tot["algorithmic_read_gb"] += n * rd / 1e9; tot["measured_read_gb"] += n * rd / 1e9 * c["read_ratio"]
            tot["algorithmic_write_gb"] += n * wr / 1e9; tot["measured_write_gb_uncalibrated"] += n * wr / 1e9 * c.get("write_ratio_uncalibrated", 1.0)
            tot["covered_ms"] += r["ms"]
        tot = {k: round(v, 2) for k, v in tot.items()}
        tot["read_ratio"] = round(tot["measured_read_gb"] / max(tot["algorithmic_read_gb"], 1e-9), 3)
        tot["write_ratio_uncalibrated"] = round(tot["measured_write_gb_uncalibrated"] / max(tot["algorithmic_write_gb"], 1e-9), 3)
        res["step_weighted_traffic"] = tot
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    for cfg in per:
        print("%-2s N=%-5d K=%-5d %-14s %8s ms %7s TF  mfma busy %-6s clock %-6s read %-7s GB (algorithmic total %-7s GB)  L2 hit %s" % (
            cfg["kind"], cfg["N"], cfg["K"], cfg["epilogue"], cfg.get("ms_per_launch_under_pmc"), cfg.get("tflops_under_pmc"), cfg.get("mfma_busy_frac"),
            cfg.get("shader_clock_ghz"), cfg.get("hbm_read_gb"), cfg.get("algorithmic_hbm_gb"), cfg.get("l2_hit_rate")))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "run":
        run()
    elif len(sys.argv) > 1 and sys.argv[1] == "summarize":
        args = sys.argv[2:]
        shapes = args[args.index("--shapes") + 1] if "--shapes" in args else ""
        calib = args[args.index("--calibration") + 1] if "--calibration" in args else ""
        out = args[args.index("--out") + 1]
        dirs = [a for i, a in enumerate(args) if not a.startswith("--") and (i == 0 or args[i - 1] not in ("--shapes", "--out", "--calibration"))]
        summarize(dirs, shapes, out, calib)
    else:
        print(__doc__)
