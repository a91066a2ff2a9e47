#!/bin/bash
# Samples rocm-smi power / clocks while a GPU workload runs.  usage: tools/power_probe.sh <label> <command...>
label=$1; shift
( for i in $(seq 1 40); do rocm-smi --showpower --showclocks --json 2>/dev/null | python -c "
import sys,json
try:
    d=json.load(sys.stdin)['card0']
    print('$label', {k:v for k,v in d.items() if 'ower' in k or 'sclk' in k or 'mclk' in k})
except Exception as e: print('smi parse error', e)
"; sleep 0.25; done ) > gpurun_out/power_$label.txt 2>&1 &
SMI=$!
"$@" > gpurun_out/power_${label}_cmd.txt 2>&1
kill $SMI 2>/dev/null
tail -n +8 gpurun_out/power_$label.txt | head -12
