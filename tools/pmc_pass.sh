#!/bin/bash
# rocprofv3 PMC passes over one command, one CSV summary per pass (per kernel: launches, mean counter value per launch).
# usage: tools/pmc_pass.sh <tag> "<counters pass 1>" ["<counters pass 2>" ...] -- <command ...>
# Counters are collected with --kernel-trace only (never with the hip/hsa/memory trace domains).
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
passes=()
while [ "$1" != "--" ]; do passes+=("$1"); shift; done
shift
i=0
for ctrs in "${passes[@]}"; do
  i=$((i+1))
  out=$R/gpurun_out/pmc_${tag}_p$i
  rm -rf $out
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -- "$@" > $out.log 2>&1
  python - "$out" <<'PY'
import csv, glob, sys, collections
files = glob.glob(sys.argv[1] + '/*/*counter_collection.csv')
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
print("kernel,counter,launches,mean_per_launch")
for k in sorted(acc, key=lambda k: -sum(len(v) for v in acc[k].values())):
    for c, v in sorted(acc[k].items()):
        print('"%s",%s,%d,%.6g' % (k, c, len(v), sum(v) / len(v)))
PY
done
