#!/bin/bash
# Round-6 measurement run on the MI355X box: everything profiles/r06_* is made from.  Run it AFTER the last kernel change of the round: the GEMM
# traffic summary records the sha256 of libemdr2_hip.so and bench_e2e.py quotes it only for that very library.
# usage: bash tools/r06_evidence.sh [quick]   (writes gpurun_out/r06/)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
export PYTHONPATH=$R
# (1) counter calibration on known byte counts in the GEMMs' access shapes, then the per-shape GEMM table of a step + PMC passes over the kernels
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/fetch_calib.hip -o tools/fetch_calib > $O/fetch_calib_build.log 2>&1
bash tools/pmc_pass.sh calib "FETCH_SIZE" "WRITE_SIZE" -- $R/tools/fetch_calib > $O/fetch_calib_pmc.csv 2>&1
python tools/fetch_calib_summary.py gpurun_out/pmc_calib_p1 gpurun_out/pmc_calib_p2 --out $O/fetch_calibration.json > $O/fetch_calibration.txt 2>&1
python tools/gemm_shapes.py --out $O/gemm_shapes.json > $O/gemm_shapes.txt 2>&1
bash tools/pmc_pass.sh gemm5 "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES" "TCC_HIT_sum TCC_MISS_sum" -- python $R/tools/gemm_pmc.py run > $O/gemm_pmc.csv 2>&1
python tools/gemm_pmc.py summarize gpurun_out/pmc_gemm5_p1 gpurun_out/pmc_gemm5_p2 gpurun_out/pmc_gemm5_p3 gpurun_out/pmc_gemm5_p4 \
    --shapes $O/gemm_shapes.json --calibration $O/fetch_calibration.json --out $O/gemm_summary.json > $O/gemm_summary.txt 2>&1
mkdir -p $R/profiles; cp $O/gemm_summary.json $R/profiles/r06_gemm_summary.json; cp $O/fetch_calibration.json $R/profiles/r06_fetch_calibration.json   # (bench_e2e.py reads these)
if [ "$1" != "quick" ]; then
  bash tools/pmc_pass.sh scan8 "FETCH_SIZE" "WRITE_SIZE" -- python $R/tools/scan_launches.py 21015324 512 > $O/mips_pmc.csv 2>&1
  python tools/mips_pmc_summary.py gpurun_out/pmc_scan8_p1 gpurun_out/pmc_scan8_p2 --out $O/mips_summary.json > /dev/null 2>&1
  cp $O/mips_summary.json $R/profiles/r06_mips_summary.json
fi
# (2) the benchmark line (both halves + the k = 100 and clustered legs), the N/8-shard search, kernel timelines, attention shapes
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --rows 2626916 --no-e2e --no-cpu-baseline --no-clustered > $O/bench_shard.json 2>> $O/bench.err
python tools/mips_timeline.py 2626916 512 50 > $O/mips_timeline_shard.txt 2>&1
python tools/mips_timeline.py 21015324 512 50 > $O/mips_timeline_full.txt 2>&1
python tools/attn_varlen_bench.py > $O/attn_varlen.txt 2>&1
python tools/attn_cross_bench.py > $O/attn_cross.txt 2>&1
if [ "$1" != "quick" ]; then
  python tools/attn_bench.py 800 512 0.1 0 > $O/attn_dense.txt 2>&1
  ( cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/r06_stats; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r06_stats -- python $R/bench.py --no-cpu-baseline --no-clustered --no-e2e-k100 > $O/bench_profiled.json 2> $O/bench_profiled.err; cp $(ls /tmp/r06_stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv )
fi
# (3) round 6: the 8-rank dry runs of configs[3] / [4] on this one GPU (gloo), the planted task, the fp32 validation mode, the dQ-atomics probe
EMDR2_SINGLE_DEVICE=1 EMDR2_DIST_BACKEND=gloo python bench.py --gpus 8 --rows 21015324 --queries 512 --no-e2e-k100 --steps 5 --warmup 2 --batch 8 --micro-batches 8 --e2e-steps 3 --e2e-warmup 1 --no-cpu-baseline > $O/bench_world8_gloo.json 2> $O/bench_world8_gloo.err
python tools/planted_task.py /tmp/planted_r06 --steps 300 --both 2> /dev/null | grep "^{" > $O/planted_task.jsonl
python -m pytest tests/test_parity_fp32_gpu.py -q -s 2>&1 | grep "fp32 parity\|passed\|failed" > $O/fp32_parity.txt
/opt/rocm/bin/hipcc -O3 -munsafe-fp-atomics --offload-arch=gfx950 tools/atomic_dq_probe.hip -o tools/atomic_dq_probe > /dev/null 2>&1 && tools/atomic_dq_probe 3 > $O/atomic_dq_probe.txt 2>&1
rm -rf $R/gpurun_out/pmc_scan8_p? $R/gpurun_out/pmc_gemm5_p? $R/gpurun_out/pmc_calib_p?
ls -la $O
