"""Kernel timeline of ONE MIPS search on a row shard (what an 8-GPU rank runs per step: N / 8 rows, 512 queries): every launch of the last of a
few searches with its start offset, duration and the idle gap before it, from `rocprofv3 --kernel-trace`.
usage: python tools/mips_timeline.py [rows] [queries] [k]         (runs itself under rocprofv3; the summary goes to stdout)"""
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("EMDR2_TIMELINE_CHILD"):
    sys.path.insert(0, ROOT)
    import torch
    import bench
    from emdr2_amd import _native
    if os.environ.get("EMDR2_TIMELINE_EXP"):                 # experiments build: EMDR2_MIPS_TUNE / _GROWTH / _SEG0 are live (timing experiments only)
        _native.LIB_PATH = _native.LIB_PATH.replace("libemdr2_hip.so", "libemdr2_hip_exp.so")
    from emdr2_amd.data.emdr2_index import HipIndexShard
    rows, nq, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    sh = HipIndexShard(768, rows, 0)
    for blk in bench.synth_rows(0, rows):
        sh.append_rows(blk)
    q = torch.randn((nq, 768), device="cuda", generator=torch.Generator(device="cuda").manual_seed(4321)).half()
    for _ in range(6):
        out = sh.search(q, k, exact_fallback=False)
    torch.cuda.synchronize()
    sys.exit(0)

rows = sys.argv[1] if len(sys.argv) > 1 else "2626916"
nq = sys.argv[2] if len(sys.argv) > 2 else "512"
k = sys.argv[3] if len(sys.argv) > 3 else "50"
out = "/tmp/mips_timeline"
subprocess.run(["rm", "-rf", out])
env = dict(os.environ, EMDR2_TIMELINE_CHILD="1", TMPDIR="/tmp")
subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__), rows, nq, k],
               env=env, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
recs = []
for f in glob.glob(out + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        recs.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
recs.sort()
starts = [i for i, r in enumerate(recs) if "pack_queries" in r[2]]
last = recs[starts[-1]:]
t0, prev_end = last[0][0], last[0][0]
print("one search: %s rows, %s queries, top-%s" % (rows, nq, k))
print("%10s %10s %8s  %s" % ("start us", "dur us", "gap us", "kernel"))
for s, e, name in last:
    print("%10.1f %10.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name.replace("(anonymous namespace)::", "").split("(")[0][-60:]))
    prev_end = e
print("total %.1f us from the first launch's start to the last one's end; kernels busy %.1f us" % ((last[-1][1] - t0) / 1e3, sum(e - s for s, e, _ in last) / 1e3))
