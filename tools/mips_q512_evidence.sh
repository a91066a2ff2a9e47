#!/bin/bash
# Raw evidence behind DESIGN.md 5.3 (what limits the MIPS scan at Q = 512), written under gpurun_out/r02_mips_q512/ for copying into profiles/.
#   1. MFMA-only micro-benchmark (tools/mfma_peak.hip)
#   2. ablation table of the lockstep scan kernel (experiments build: EMDR2_MIPS_ABLATE; thresholds +inf so the filter is idle):
#        0 full kernel | 4 thresholds +inf | 3 no LDS-DMA fill | 1 fill + fragment reads, no MFMA | 2 fill only
#   3. package power / shader clock sampled by rocm-smi during a sustained Q = 512 search loop and during the NT GEMM loop
#   4. SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE / FETCH_SIZE of the scan kernel (rocprofv3 --pmc, kernel-trace only)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; export PYTHONPATH=$R
O=$R/gpurun_out/r02_mips_q512; mkdir -p $O
ROWS=${1:-21015324}
( [ -x tools/mfma_peak ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/mfma_peak ) && tools/mfma_peak > $O/mfma_peak.txt 2>&1
for a in 0 4 3 1 2; do
  echo "== EMDR2_MIPS_ABLATE=$a (libemdr2_hip_exp.so), $ROWS rows, Q = 512" >> $O/ablation.txt
  EMDR2_MIPS_ABLATE=$a python tools/scan_launches.py --exp $ROWS 512 >> $O/ablation.txt 2>&1
done
echo "== production library (no switches)" >> $O/ablation.txt
python tools/scan_launches.py $ROWS 512 >> $O/ablation.txt 2>&1
bash tools/power_probe.sh mips_q512 python bench.py --steps 60 --warmup 2 --no-e2e --no-cpu-baseline > $O/power_mips_q512.txt 2>&1
cp gpurun_out/power_mips_q512.txt $O/power_mips_q512_samples.txt; tail -1 gpurun_out/power_mips_q512_cmd.txt > $O/power_mips_q512_bench_line.json
bash tools/power_probe.sh gemm_nt python tools/gemm_loop.py 768 3072 > $O/power_gemm_nt.txt 2>&1
cp gpurun_out/power_gemm_nt.txt $O/power_gemm_nt_samples.txt; tail -3 gpurun_out/power_gemm_nt_cmd.txt > $O/power_gemm_nt_cmd_tail.txt
bash tools/pmc_pass.sh scan "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES" "FETCH_SIZE" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY" -- python $R/tools/scan_launches.py $ROWS 512 > $O/pmc_scan_q512.csv 2>&1
ls -la $O
