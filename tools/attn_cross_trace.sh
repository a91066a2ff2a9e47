#!/bin/bash
# Per-kernel times of the decoder-side attention launches of ONE question group (rocprofv3 --kernel-trace --stats over tools/attn_cross_bench.py).
# usage: bash tools/attn_cross_trace.sh [questions] [topk]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/attn_cross_trace
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/attn_cross_trace -- python $R/tools/attn_cross_bench.py --questions ${1:-16} --topk ${2:-50} --only-fid-cross > /tmp/attn_cross_trace.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/attn_cross_trace/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-100s calls %5s avg %9.1f us total %8.2f ms" % (r["Name"].replace("(anonymous namespace)::", "")[:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
