#!/bin/bash
# HBM traffic of the NT GEMM (FETCH_SIZE / WRITE_SIZE, KB) for one shape under a given environment.  usage: tools/pmc_gemm.sh "<ENV>" N K
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cfg="$1"; N=$2; K=$3
for ctr in FETCH_SIZE WRITE_SIZE; do
  out=$R/gpurun_out/pmcg_${ctr}_$$
  env $cfg rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -- python $R/tools/gemm_loop_once.py $N $K > $out.log 2>&1
  python - "$out" "$cfg" $ctr <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*counter_collection.csv')[0]
vals=[float(r['Counter_Value']) for r in csv.DictReader(open(f)) if 'gemm_nt' in r['Kernel_Name']]
print("%-28s %-11s per launch: %.3f GB (x2 gfx950 correction for FETCH_SIZE: %.3f GB)" % (sys.argv[2], sys.argv[3], max(vals)*1024/1e9, 2*max(vals)*1024/1e9))
PY
done
