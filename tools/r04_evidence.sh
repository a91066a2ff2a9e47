#!/bin/bash
# Round-4 measurement run on the MI355X box: everything profiles/r04_* is made from.  usage: bash tools/r04_evidence.sh   (writes gpurun_out/r04/)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
export PYTHONPATH=$R
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --rows 2626916 --no-e2e --no-cpu-baseline > $O/bench_shard.json 2>> $O/bench.err
python tools/mips_timeline.py 2626916 512 50 > $O/mips_timeline_shard.txt 2>&1
python tools/mips_timeline.py 21015324 512 50 > $O/mips_timeline_full.txt 2>&1
python tools/attn_bench.py 800 512 0.1 0 > $O/attn_dense.txt 2>&1
python tools/attn_varlen_bench.py > $O/attn_varlen.txt 2>&1
( cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/r04_stats; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r04_stats -- python $R/bench.py --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.err; cp $(ls /tmp/r04_stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv )
bash tools/pmc_pass.sh scan8 "FETCH_SIZE" "WRITE_SIZE" -- python $R/tools/scan_launches.py 21015324 512 > $O/mips_pmc.csv 2>&1
python tools/mips_pmc_summary.py gpurun_out/pmc_scan8_p1 gpurun_out/pmc_scan8_p2 --out $O/mips_summary.json > /dev/null 2>&1
bash tools/pmc_pass.sh attn_r04 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" -- python $R/tools/attn_bench.py 800 512 0.1 0 > $O/attention_pmc.csv 2>&1
rm -rf $R/gpurun_out/pmc_scan8_p? $R/gpurun_out/pmc_attn_r04_p?
ls -la $O
