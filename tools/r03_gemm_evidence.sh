#!/bin/bash
# Regenerates profiles/r03_gemm_summary.json on a GPU box: the per-shape GEMM table of one end-to-end step (packed sequences, selective
# retention) + rocprofv3 PMC passes over the GEMM kernels one by one (FETCH_SIZE / WRITE_SIZE / MFMA-busy / L2 hit), each counter group in
# its own pass with --kernel-trace only.   usage: bash tools/r03_gemm_evidence.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
python $R/tools/gemm_shapes.py --selective-layers 12,11 --keep-last-layers 0 --out $R/gpurun_out/r03_gemm_shapes.json > $R/gpurun_out/r03_gemm_shapes.txt 2>&1
bash $R/tools/pmc_pass.sh gemm3 "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES" "TCC_HIT_sum TCC_MISS_sum" -- python $R/tools/gemm_pmc.py run
python $R/tools/gemm_pmc.py summarize $R/gpurun_out/pmc_gemm3_p1 $R/gpurun_out/pmc_gemm3_p2 $R/gpurun_out/pmc_gemm3_p3 $R/gpurun_out/pmc_gemm3_p4 \
    --shapes $R/gpurun_out/r03_gemm_shapes.json --out $R/gpurun_out/r03_gemm_summary.json > $R/gpurun_out/r03_gemm_summary.txt 2>&1
tail -40 $R/gpurun_out/r03_gemm_summary.txt
