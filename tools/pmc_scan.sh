#!/bin/bash
# PMC pass over the bench for the scan kernel. usage: tools/pmc_scan.sh "<ENV settings>" COUNTER [COUNTER...]
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cfg="$1"; shift
out=$R/gpurun_out/pmc_$$
env $cfg rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out.log 2>&1
python - "$out" "$cfg" <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/*/*counter_collection.csv')[0]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'mips_scan' in r['Kernel_Name'] and int(r['Grid_Size'])>=256*512:
        acc[(r['Kernel_Name'][:40],r['Counter_Name'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']),float(r['Counter_Value'])))
print(sys.argv[2])
for (k,c),v in sorted(acc.items()):
    v.sort(); d,val=v[-1]
    print("  %-42s %-28s %14.0f  (launch %.3f ms)"%(k,c,val,d/1e6))
PY
