#!/bin/bash
# Segment schedule of the MIPS search on an N/8 row shard (experiments library: EMDR2_MIPS_SEG0 / EMDR2_MIPS_GROWTH are live): kernel timeline per setting.
# usage: bash tools/mips_segment_sweep.sh [rows]
R=${GRAFT_REPO_ROOT:-/root/repo}
rows=${1:-2626916}
for cfg in "8192 8" "8192 16" "8192 32" "8192 64" "4096 8" "4096 16" "16384 8" "16384 16" "2048 16"; do
  set -- $cfg
  echo "==== seg0 $1 growth $2"
  EMDR2_TIMELINE_EXP=1 EMDR2_MIPS_SEG0=$1 EMDR2_MIPS_GROWTH=$2 python $R/tools/mips_timeline.py $rows 512 50 | tail -n +3
done
