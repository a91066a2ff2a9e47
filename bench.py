#!/usr/bin/env python
"""bench.py -- both halves of BASELINE.json's metric on MI355X: MIPS queries/s of the EMDR2 evidence search (configs[1], the top-level
fields of the JSON line) and QA train steps/s of the end-to-end EMDR2 step (configs[2], the `e2e` object; bench_e2e.py has the step).

One "step" = one call of the OPERATOR the reference's seam calls, `DistributedBruteForceIndex.search_mips_index(queries, 50)`
(megatron/data/emdr2_index.py:268-305): 512 fp16 queries against the 21,015,324 x 768 fp16
evidence index resident in HBM, top-50, including the dtype check, query packing, the fused scan, candidate
selection, exact re-scoring, the host read of the per-query proof flags (+ the all-exact path for flagged queries) and (N > 1) the
all-gather + merge.  `config.inner_sequence_ms_per_step` is the kernel sequence alone (no flag read), timed right after.  Inputs are resident in HBM when
the timed region starts.  With --gpus N the index is row-sharded N ways (one process per GPU, RCCL);
total work is fixed, so scaling is "strong".

    python bench.py [--gpus N] [--steps K] [--warmup W] [--rows R] [--queries Q] [--topk k]

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the scan launch over the
last, largest row segment), timed with hipEvents on the launch stream inside the library;
`cpu_baseline` times the oracle's CPU port (fp32-accumulate GEMM + top-k, all host cores) on a
bounded row sample (rank 0, N = 1 only).

`e2e` (after the MIPS region, same resident index): `steps_per_s` / `ms_per_step` of --e2e-steps training steps at B = 64 questions per
GPU, top-k 50 (data-parallel over the N ranks, gradients averaged over RCCL); its `roofline` is for the step's dominant kernels, the
dense linears: executed GEMM flops / per-launch hipEvent time summed inside the timed steps, against the 2.5 PFLOP/s bf16 MFMA peak, plus
the whole-step MFU; its `cpu_baseline` is the fp32 oracle of the model path on the host cores (bounded sample, N = 1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_ROWS_FULL = 21_015_324          # psgs_w100 passages
DIM = 768
HBM_PEAK_GBPS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0         # dense fp16/bf16 MFMA peak
GEN_CHUNK = 1 << 19


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=N_ROWS_FULL)
    ap.add_argument("--queries", type=int, default=512)
    ap.add_argument("--topk", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-e2e", action="store_true", help="MIPS half only")
    ap.add_argument("--e2e-steps", type=int, default=10)
    ap.add_argument("--e2e-warmup", type=int, default=1)
    ap.add_argument("--e2e-timeout", type=float, default=600.0, help="seconds after which rank 0 prints the line without the unfinished e2e objects and exits")
    ap.add_argument("--no-e2e-k100", action="store_true",
                    help="skip the `e2e_k100` object: BASELINE configs[4] (top-k 100 + continuous re-embedding on a side stream) at its per-rank "
                         "shape -- the N/8-row index shard of an 8-GPU run -- timed with and without the refresher")
    ap.add_argument("--k100-steps", type=int, default=5)
    ap.add_argument("--no-clustered", action="store_true", help="skip the `clustered` object: the same search over a topic-contiguous, anisotropic corpus")
    import bench_e2e
    bench_e2e.add_args(ap)
    return ap.parse_args()


def synth_rows(lo, hi, seed=1234):
    """Rows [lo, hi) of the synthetic index: fp16(N(0,1)), generated on the device chunk by chunk from a
    (seed, chunk) keyed generator so that a row's content does not depend on the sharding."""
    c0, c1 = lo // GEN_CHUNK, (hi + GEN_CHUNK - 1) // GEN_CHUNK
    for c in range(c0, c1):
        g = torch.Generator(device="cuda").manual_seed(seed * 1_000_003 + c)
        block = torch.randn((GEN_CHUNK, DIM), generator=g, device="cuda", dtype=torch.float32).to(torch.float16)
        a, b = max(lo, c * GEN_CHUNK), min(hi, (c + 1) * GEN_CHUNK)
        yield block[a - c * GEN_CHUNK: b - c * GEN_CHUNK]


TOPIC_ROWS = 32


def _topic_block(chunk, seed):
    """(centers fp32 [GEN_CHUNK / 32, DIM], norm scales [GEN_CHUNK / 32]) of the topics of generator chunk `chunk` of the clustered corpus."""
    g = torch.Generator(device="cuda").manual_seed(seed * 1_000_003 + 7_000_000 + chunk)
    nt = GEN_CHUNK // TOPIC_ROWS
    centers = torch.randn((nt, DIM), generator=g, device="cuda", dtype=torch.float32)
    scales = torch.exp(0.25 * torch.randn((nt,), generator=g, device="cuda", dtype=torch.float32))
    return centers, scales


def synth_rows_clustered(lo, hi, seed=1234, alpha=0.8):
    """Rows [lo, hi) of the CLUSTERED synthetic index: the opposite of i.i.d. rows in the three respects that matter to
    a threshold filter walking the rows in order -- (1) topic-contiguous: every 32 consecutive rows share a topic centre (cosine 0.64 to it:
    consecutive passages of one article), (2) anisotropic: a topic's rows carry a log-normal norm factor (sigma 0.25), (3) queries
    (`clustered_queries`) sit near topics at the very END of the row order, so the rows that matter arrive after every threshold was set on
    unrelated ones.  Chunk-keyed like `synth_rows`: a row does not depend on the sharding."""
    c0, c1 = lo // GEN_CHUNK, (hi + GEN_CHUNK - 1) // GEN_CHUNK
    beta = (1.0 - alpha * alpha) ** 0.5
    for c in range(c0, c1):
        centers, scales = _topic_block(c, seed)
        g = torch.Generator(device="cuda").manual_seed(seed * 1_000_003 + c)
        block = torch.randn((GEN_CHUNK, DIM), generator=g, device="cuda", dtype=torch.float32)
        block = block.view(-1, TOPIC_ROWS, DIM).mul_(beta).add_(alpha * centers[:, None, :]).mul_(scales[:, None, None]).view(GEN_CHUNK, DIM).to(torch.float16)
        a, b = max(lo, c * GEN_CHUNK), min(hi, (c + 1) * GEN_CHUNK)
        yield block[a - c * GEN_CHUNK: b - c * GEN_CHUNK]


def clustered_queries(n_rows, nq, seed=1234, per_topic=8):
    """nq fp16 queries near nq / per_topic distinct topics taken from the last 2 % of the rows (8 questions about each late article)."""
    g = torch.Generator(device="cuda").manual_seed(seed + 99)
    n_topics = max(1, nq // per_topic)
    first = max(0, int(n_rows * 0.98) // TOPIC_ROWS)
    last = max(first + 1, (n_rows - 1) // TOPIC_ROWS)
    topics = first + torch.randperm(max(last - first, 1), generator=g, device="cuda")[:n_topics]
    topics = topics.repeat_interleave(per_topic)[:nq]
    if topics.numel() < nq:
        topics = torch.cat([topics, topics[:nq - topics.numel()]])
    out = torch.empty((nq, DIM), dtype=torch.float32, device="cuda")
    per_chunk = GEN_CHUNK // TOPIC_ROWS
    for c in torch.unique(topics // per_chunk).tolist():
        centers, _ = _topic_block(int(c), seed)
        m = (topics // per_chunk) == c
        out[m] = centers[(topics[m] % per_chunk)]
    out += 0.3 * torch.randn((nq, DIM), generator=g, device="cuda", dtype=torch.float32)
    return out.to(torch.float16)


def clustered_leg(args, rank, world, steps=5):
    """The search over the clustered corpus (same size, same kernels, same operator): queries/s, how many queries the fast path could not
    prove (they take the all-exact path inside the operator), the time of one all-exact pass (8 queries over this rank's rows), and the
    bound on a search they imply.  Rank-local timing; N > 1: every rank searches its shard, exchange included."""
    from emdr2_amd.data.emdr2_index import DistributedBruteForceIndex, shard_bounds
    index = DistributedBruteForceIndex(embed_size=DIM, embed_data=None, use_gpu=True)
    lo, hi = shard_bounds(args.rows, world)[rank]
    index.num_rows = args.rows
    index.shard = index._make_shard(DIM, hi - lo, lo)
    for block in synth_rows_clustered(lo, hi):
        index.shard.append_rows(block)
    index.shard.set_ids(torch.arange(lo + 1, hi + 1, dtype=torch.int32, device="cuda"))
    q = clustered_queries(args.rows, args.queries)
    k = args.topk
    for _ in range(2):
        index.search_mips_index(q, k)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        index.search_mips_index(q, k)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    d, i, r, f = index.shard.search(q, k, exact_fallback=False)
    flagged = int((f != 0).sum().item())
    overflow = int(((f & 2) != 0).sum().item())
    sel = torch.arange(8, dtype=torch.int32, device="cuda")
    d2, i2, r2, f2 = d.clone(), i.clone(), r.clone(), f.clone()
    index.shard.search_exact(q, sel, k, d2, i2, r2, f2)                   # (also the correctness net's own check below)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    index.shard.search_exact(q, sel, k, d2, i2, r2, f2)
    torch.cuda.synchronize()
    exact_ms = (time.perf_counter() - t0) * 1e3
    ok = f[:8].abs().sum().item() != 0 or (torch.equal(d[:8].view(torch.int16), d2[:8].view(torch.int16)) and torch.equal(i[:8], i2[:8]))
    # where the top-50 of these queries live: the share found in the last 2 % of the rows
    late = float((r >= int(args.rows * 0.98)).float().mean().item())
    passes = (args.queries + 7) // 8
    out = {"data": "clustered: 32-row topic runs (cosine 0.64 to the topic centre), log-normal topic norms (sigma 0.25), %d queries near %d topics of the "
                   "last 2 %% of the rows" % (args.queries, max(1, args.queries // 8)),
           "queries_per_s": args.queries / (ms * 1e-3), "ms_per_search": ms, "unproven_queries": flagged, "candidate_overflow_queries": overflow,
           "share_of_topk_from_the_last_2pct_rows": late, "fast_path_equals_exact_path_on_8_queries": bool(ok),
           "exact_pass_ms_per_8_queries": exact_ms,
           "worst_case_bound_ms": ms + passes * exact_ms,
           "worst_case_note": "a query the fast path cannot prove costs its share of an all-exact pass (integer arithmetic over every row, 8 queries "
                              "per pass): at most %d passes for %d queries on top of the fast path" % (passes, args.queries)}
    del index
    return out


def result_digest(dist, idx):
    """sha256 over the (scores fp16 [Q, k], doc ids int32 [Q, k]) a search returned: the canonical result is a function of the index and
    the queries alone, so the digest of an N-shard run must equal the single-shard run's (tests/test_world8_gpu.py)."""
    import hashlib
    h = hashlib.sha256()
    h.update(dist.detach().to(torch.float16).contiguous().cpu().numpy().tobytes())
    h.update(idx.detach().to(torch.int32).contiguous().cpu().numpy().tobytes())
    return h.hexdigest()


def cpu_baseline(args, queries_cpu):
    """Reference arithmetic on the host cores: fp32-accumulate GEMM (torch/MKL), one rounding to fp16, top-k --
    oracle.mips_oracle.topk_blas -- over >= 1M synthetic rows, in the metric's unit by scaling rows/s to the full index.
    The thread count is PICKED by a short sweep (32 / 64 / all cores: a GEMM of 512 x 768 panels on hundreds of threads runs slower than
    on 64), the sweep is part of `sample`."""
    from oracle import mips_oracle as mo
    cores = os.cpu_count() or 1
    nq = queries_cpu.shape[0]
    probe_rows = 65536
    g = torch.Generator().manual_seed(7)
    rows = torch.randn((probe_rows, DIM), generator=g).to(torch.float16).numpy()
    sweep = {}
    for threads in sorted(set(min(cores, t) for t in (32, 64, cores))):
        torch.set_num_threads(threads)
        mo.topk_blas(rows[:8192], queries_cpu, args.topk)                   # warm-up at this thread count
        t0 = time.perf_counter(); mo.topk_blas(rows, queries_cpu, args.topk); sweep[threads] = time.perf_counter() - t0
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    reps = max(16, min(64, int(args.cpu_seconds * 0.5 / max(sweep[best], 1e-3))))       # >= 16 x 65,536 = 1,048,576 rows
    t0 = time.perf_counter()
    for _ in range(reps):                                                     # same rows re-scanned: only the rate matters
        mo.topk_blas(rows, queries_cpu, args.topk)
    t = time.perf_counter() - t0
    row_queries_per_s = (reps * probe_rows) * nq / t
    return {"value": row_queries_per_s / args.rows, "unit": "queries/s", "cores": best, "kind": "port",
            "sample": "%d queries x %d rows (fp32 GEMM + fp16 round + top-%d, torch CPU) in %.1f s on %d threads = %.0f GFLOP/s; threads swept on a "
                      "%d-row probe: %s (host has %d cores); rate scaled to the %d-row index"
                      % (nq, reps * probe_rows, args.topk, t, best, 2.0 * nq * reps * probe_rows * DIM / t / 1e9, probe_rows,
                         ", ".join("%d -> %.2f s" % kv for kv in sorted(sweep.items())), cores, args.rows)}


def main():
    args = parse()
    from emdr2_amd import dist_util
    dist_util.self_launch(args.gpus)      # plain `python bench.py --gpus N`: becomes N ranks under torch.distributed.run (no-op inside a launcher)
    # one process per GPU over RCCL; a collective whose peers are gone times out (e2e budget + slack) instead of hanging the launcher
    rank, world, _ = dist_util.init_distributed(timeout_s=args.e2e_timeout + 60.0)

    from emdr2_amd import _native
    from emdr2_amd.data.emdr2_index import shard_bounds
    import bench_e2e
    lib = _native.lib()

    lo, hi = shard_bounds(args.rows, world)[rank]
    index = bench_e2e.build_index(args.rows, rank, world)              # this rank's row shard, shared by both halves of the benchmark
    shard = index.shard
    gq = torch.Generator(device="cuda").manual_seed(4321)
    queries = torch.randn((args.queries, DIM), generator=gq, device="cuda", dtype=torch.float32).to(torch.float16)
    nq, k = args.queries, args.topk

    index.exchange_events = [] if world > 1 else None      # (start, end) hipEvent pairs around all-gather + merge, recorded by the index

    def step():
        # the plug-in operator itself: fp16 check, shard search, host read of the proof flags (+ all-exact path for flagged queries),
        # N > 1: records straight into the gather buffer, ONE all-gather, ONE merge launch (emdr2_index.py:_search_exchange_merge)
        return index.search_mips_index(queries, k)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    fence()
    lib.emdr2_mips_set_timing(1)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t0
    digest = result_digest(out[0], out[1])
    digests_agree = None
    if world > 1:                                           # every rank must hold the same merged result
        every = [None] * world
        torch.distributed.all_gather_object(every, digest)
        digests_agree = len(set(every)) == 1

    # per-launch scan timings recorded with hipEvents on the launch stream during the timed region
    cap = 2048
    ms = (ctypes.c_float * cap)(); rows_l = (ctypes.c_int64 * cap)(); n_l = ctypes.c_int()
    _native.check(lib.emdr2_mips_timing_collect(ms, rows_l, cap, ctypes.byref(n_l)), "timing_collect")
    lib.emdr2_mips_set_timing(0)
    launches = [(ms[i], rows_l[i]) for i in range(n_l.value)]
    # the kernel sequence of a search alone (query packing, scan segments, selects, finalize; no flag read, no exchange), same step count:
    # what rounds 1-4 reported as the headline; and the number of queries the fast path left unproven (the operator re-does those exactly)
    fence()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        inner = shard.search(queries, k, exact_fallback=False)
    fence()
    inner_ms = (time.perf_counter() - t1) / max(args.steps, 1) * 1e3
    flags_total = int(inner[3].abs().sum().item())
    big = max(r for _, r in launches)
    dom = [m for m, r in launches if r == big]
    dom_ms = sum(dom) / len(dom)
    scan_ms_per_step = sum(m for m, _ in launches) / max(args.steps, 1)

    # secondary roofline in the HBM-bound regime (Q = 64, the per-GPU training batch): same index, same kernel family
    hbm_regime = None
    if nq > 128:
        q64 = queries[:64].contiguous()
        for _ in range(2):
            shard.search(q64, k, exact_fallback=False)
        fence()
        lib.emdr2_mips_set_timing(1)
        for _ in range(5):
            shard.search(q64, k, exact_fallback=False)
        fence()
        _native.check(lib.emdr2_mips_timing_collect(ms, rows_l, cap, ctypes.byref(n_l)), "timing_collect")
        lib.emdr2_mips_set_timing(0)
        l64 = [(ms[i], rows_l[i]) for i in range(n_l.value)]
        big64 = max(r for _, r in l64)
        d64 = [m for m, r in l64 if r == big64]
        ms64 = sum(d64) / len(d64)
        gb64 = float(big64) * DIM * 2 / (ms64 * 1e-3) / 1e9
        hbm_regime = {"queries": 64, "bound": "hbm", "achieved": gb64, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                      "frac": gb64 / HBM_PEAK_GBPS, "kernel_ms": ms64, "rows_per_launch": int(big64)}

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    exchange_ms = None
    if world > 1 and index.exchange_events:
        timed = index.exchange_events[-args.steps:]
        exchange_ms = sum(a.elapsed_time(b) for a, b in timed) / len(timed)
    index.exchange_events = None

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        flops = 2.0 * nq * big * DIM
        bytes_alg = float(big) * DIM * 2
        tflops = flops / (dom_ms * 1e-3) / 1e12
        gbps = bytes_alg / (dom_ms * 1e-3) / 1e9
        bound = "mfma" if nq > 312 else "hbm"        # machine balance 2.5e15 / 8e12
        roofline = {"bound": bound,
                    "achieved": tflops if bound == "mfma" else gbps,
                    "peak": MFMA_PEAK_TFLOPS if bound == "mfma" else HBM_PEAK_GBPS,
                    "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
                    "frac": (tflops / MFMA_PEAK_TFLOPS) if bound == "mfma" else (gbps / HBM_PEAK_GBPS),
                    "traffic": None,
                    "kernel": "mips_scan8_kernel (last row segment: %d rows/launch)" % big,
                    "kernel_ms": dom_ms, "scan_ms_per_step": scan_ms_per_step,
                    "hbm_gbps": gbps, "hbm_frac": gbps / HBM_PEAK_GBPS,
                    "mfma_tflops": tflops, "mfma_frac": tflops / MFMA_PEAK_TFLOPS}
        # HBM bytes per launch of the same kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x2 per the
        # gfx950 correction + WRITE_SIZE); only quoted when the profile was taken on this exact launch shape.
        prof = os.path.join(ROOT, "profiles", "r06_mips_summary.json")
        if os.path.exists(prof):
            import hashlib
            pj = json.load(open(prof))
            loaded = hashlib.sha256(open(_native.LIB_PATH, "rb").read()).hexdigest()
            if pj.get("library_sha256") != loaded:
                roofline["traffic_source"] = "profiles/r06_mips_summary.json was measured on another build of libemdr2_hip.so: not quoted"
            elif pj.get("algorithmic_bytes_last_segment") == int(bytes_alg) and nq == 512:
                roofline["traffic"] = pj["traffic_bytes"]
                roofline["traffic_source"] = ("profiles/r06_mips_summary.json, same library (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE; L2->fabric requests, "
                                              "Infinity-Cache hits included)")
        result = {
            "metric": "mips_queries_per_sec", "value": nq * args.steps / elapsed, "unit": "queries/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f16",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: MIPS-only top-%d, %d queries/search over a %d x %d fp16 index resident in HBM"
                                   % (k, nq, args.rows, DIM),
                       "rows": args.rows, "dim": DIM, "queries_per_step": nq, "top_k": k,
                       "parallelism": "index row-sharded x%d, all-gather(top-k) + merge" % world,
                       "cus": int(lib.emdr2_device_cu_count()), "unproven_queries": flags_total,
                       # scores + doc ids of all queries of the last timed search: equal for any number of shards / ranks
                       "result_sha256": digest, "result_identical_on_all_ranks": digests_agree,
                       "timed_call": "DistributedBruteForceIndex.search_mips_index (flag read + exact fallback + exchange included)",
                       "inner_sequence_ms_per_step": inner_ms,
                       # what makes the 1 -> N curve interpretable: rows scanned per rank, bytes every rank contributes to the ONE
                       # all-gather of a search, and the time from the all-gather to the merged result (rank 0, hipEvents)
                       "rows_per_rank": [b - a for a, b in shard_bounds(args.rows, world)],
                       "allgather_bytes_per_rank": (nq * k * 16) if world > 1 else 0,
                       "allgather_plus_merge_ms": exchange_ms},
            "roofline": roofline,
        }
        if hbm_regime is not None:
            result["roofline_hbm_regime"] = hbm_regime
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args, queries.cpu().numpy())
    else:
        result = None

    if not args.no_clustered:
        try:
            cl = clustered_leg(args, rank, world)
        except Exception as exc:
            cl = {"error": "%s: %s" % (type(exc).__name__, exc)}
        import gc
        gc.collect(); torch.cuda.empty_cache()
        if rank == 0:
            result["clustered"] = cl

    # ---- second half of the metric: the end-to-end training step over the same resident index ---------------------------------------
    if not args.no_e2e:
        watchdog = None
        if rank == 0:
            import threading

            def give_up():
                result.setdefault("e2e", {"error": "end-to-end step did not finish within %.0f s" % args.e2e_timeout})
                if not args.no_e2e_k100:
                    result.setdefault("e2e_k100", {"error": "not finished within %.0f s" % args.e2e_timeout})
                print(json.dumps(result), flush=True)
                os._exit(0)
            watchdog = threading.Timer(args.e2e_timeout, give_up)
            watchdog.daemon = True
            watchdog.start()
        # (ranks > 0 need no timer: a collective whose peer has given up fails by the process group's own timeout, dist_util.init_distributed)
        try:
            del queries
            ctx = bench_e2e.setup(args, rank, world, index=index, topk=k)
            e2e = bench_e2e.run(ctx, args.e2e_steps, args.e2e_warmup, world)
            if rank == 0 and world == 1 and not args.no_cpu_baseline:
                e2e["cpu_baseline"] = bench_e2e.cpu_baseline_subprocess(min(args.cpu_seconds, 10.0), threads=result.get("cpu_baseline", {}).get("cores", 0))
        except Exception as exc:                      # the MIPS half is still reported
            e2e = {"error": "%s: %s" % (type(exc).__name__, exc)}
        if rank == 0:
            result["e2e"] = e2e
        if not args.no_e2e_k100:
            # BASELINE configs[4] at its per-rank shape: top-k 100 over the N/8-row shard one of 8 ranks holds (N = 1: a shard-sized index of its
            # own; N > 1: this rank's shard of --rows), the step timed WITHOUT and WITH the side-stream refresher at the 8-GPU pace
            try:
                ctx = None
                bench_e2e.release()
                k100 = bench_e2e.run_k100(args, rank, world, index if world > 1 else None, steps=args.k100_steps)
            except Exception as exc:
                k100 = {"error": "%s: %s" % (type(exc).__name__, exc)}
            if rank == 0:
                result["e2e_k100"] = k100
        if watchdog is not None:
            watchdog.cancel()
    if rank == 0:
        print(json.dumps(result), flush=True)
    dist_util.shutdown()


if __name__ == "__main__":
    main()
